"""Blocked-NHWC device tensors and thin wrappers over the C ABI (one Python call = one kernel launch
sequence on the current HIP stream)."""
import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F32, MmifTensor, check, lib

_TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}
_CODE = {torch.float32: F32, torch.bfloat16: BF16}


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def require_device(t, what):
    """Every launch goes to the CURRENT device's current stream: refuse CPU tensors and tensors of another GPU
    (model.to('cuda:1') without torch.cuda.set_device(1) would otherwise launch on device 0 against device-1 pointers)."""
    if not t.is_cuda:
        raise RuntimeError(f"mmif: {what} must live on the GPU (got a {t.device} tensor); the HIP engine has no CPU path")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"mmif: {what} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"call torch.cuda.set_device({t.device.index}) (one process per GPU) before running the model")


class BT:
    """A view [cb_off, cb_off+cb) of a blocked allocation [n][cb_total][h+2*halo][w+2*halo][8]."""
    __slots__ = ("buf", "n", "h", "w", "halo", "cb_total", "cb_off", "cb", "code", "flags", "_d")

    def __init__(self, buf, n, h, w, halo, cb_total, cb_off, cb, code, flags=0):
        self.buf, self.n, self.h, self.w, self.halo = buf, n, h, w, halo
        self.cb_total, self.cb_off, self.cb, self.code, self.flags = cb_total, cb_off, cb, code, flags
        self._d = MmifTensor(buf.data_ptr(), code, n, h, w, halo, cb_total, cb_off, cb, flags)

    @staticmethod
    def alloc(n, c, h, w, dtype, device, halo=0, zero=False):
        cb = (c + 7) // 8
        shape = (n, cb, h + 2 * halo, w + 2 * halo, 8)
        buf = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        return BT(buf, n, h, w, halo, cb, 0, cb, _CODE[dtype])

    def view(self, cb_off, cb):
        assert 0 <= cb_off and cb_off + cb <= self.cb, (cb_off, cb, self.cb)
        return BT(self.buf, self.n, self.h, self.w, self.halo, self.cb_total, self.cb_off + cb_off, cb, self.code, self.flags)

    def as_folded(self):
        """Same view, flagged 'halo already folded + zeroed' (after fold_halo): readers skip fold-on-load."""
        return BT(self.buf, self.n, self.h, self.w, self.halo, self.cb_total, self.cb_off, self.cb, self.code, _lib.T_FOLDED)

    def fold_halo_(self):
        check(lib.mmif_fold_halo(self.d, stream_ptr()), "fold_halo")
        return self.as_folded()

    @property
    def d(self):
        return C.byref(self._d)

    @property
    def channels(self):
        return self.cb * 8

    @property
    def dtype(self):
        return _TORCH_DTYPE[self.code]

    # ---- boundary conversions (reference tensors are NCHW fp32) ----
    @staticmethod
    def from_nchw(x, dtype, halo=0):
        require_device(x, "input")
        x = x.contiguous().float()
        n, c, h, w = x.shape
        t = BT.alloc(n, c, h, w, dtype, x.device, halo, zero=halo > 0)
        check(lib.mmif_nchw_to_blocked(_ptr(x), c, t.d, stream_ptr()), "nchw_to_blocked")
        return t

    def to_nchw(self, c=None):
        c = self.channels if c is None else c
        out = torch.empty((self.n, c, self.h, self.w), dtype=torch.float32, device=self.buf.device)
        check(lib.mmif_blocked_to_nchw(self.d, _ptr(out), c, stream_ptr()), "blocked_to_nchw")
        return out

    def zero_(self):
        check(lib.mmif_zero(self.d, stream_ptr()), "zero")
        return self


class PackedWeights:
    """MFMA operand images of one conv layer's weights (forward + dgrad).  fmt = BF16: the bf16 kernels' images (bf16 tensors);
    fmt = F32: the split-bf16 "x3" images (successive bf16 pieces of every weight) of the fp32-grade kernels that run fp32 tensors'
    3x3 and 1x1 layers on the matrix pipe (csrc/conv_x3.hip)."""
    __slots__ = ("fwd", "dgrad", "cout", "cin", "k", "fmt")

    def __init__(self, cout, cin, k, device, fmt=BF16):
        assert fmt in (BF16, F32)
        nbytes = lib.mmif_packed_weight_bytes_x3(cout, cin, k) if fmt == F32 else lib.mmif_packed_weight_bytes(cout, cin, k)
        self.fwd = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.dgrad = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.cout, self.cin, self.k, self.fmt = cout, cin, k, fmt

    @property
    def usable(self):
        return self.fmt == BF16 or self.k in (1, 3)

    def pack(self, w):
        if self.fmt == F32:
            if self.k in (1, 3):
                check(lib.mmif_pack_weights_x3(_ptr(w), self.cout, self.cin, self.k, _ptr(self.fwd), _ptr(self.dgrad), stream_ptr()),
                      "pack_weights_x3")
            return
        check(lib.mmif_pack_weights(_ptr(w), self.cout, self.cin, self.k, _ptr(self.fwd), _ptr(self.dgrad), stream_ptr()),
              "pack_weights")


def _image(packed, which, code):
    """pointer to a layer's operand image when it is in the format the tensors' dtype needs (else NULL: the kernels that take the fp32
    master weights run)"""
    if packed is None or packed.fmt != code or not packed.usable:
        return None
    return _ptr(getattr(packed, which))


def pack_many(pairs):
    """[(PackedWeights, fp32 weight tensor), ...] -> every operand image in ONE launch (mmif_pack_weights_multi)."""
    if not pairs:
        return
    jobs = (_lib.MmifPackJob * len(pairs))()
    for j, (pk, w) in zip(jobs, pairs):
        j.w, j.cout, j.cin, j.ksize = w.data_ptr(), pk.cout, pk.cin, pk.k
        j.format = _lib.PACK_X3 if pk.fmt == F32 else _lib.PACK_BF16
        j.packed_fwd, j.packed_dgrad = pk.fwd.data_ptr(), pk.dgrad.data_ptr()
    check(lib.mmif_pack_weights_multi(jobs, len(pairs), stream_ptr()), "pack_weights_multi")


# ------------------------------------------------------------------ per-op HIP-event timing (bench.py roofline)
PROFILE_TAGS = set()     # op tags ("<layer>:fwd|dgrad|wgrad") to time; "*" = every tagged op (tools/layer_times.py)
PROFILE_EVENTS = {}      # tag -> [(start_event, end_event), ...] recorded on the launch stream
PROFILE_SHAPES = {}      # tag -> (n, h, w, cin, cout, k) of the last timed conv call with that tag


PROFILE_EVENT_POOL = {}   # tag -> [(start_event, end_event), ...] created (and recorded once) ahead of time: prealloc_events()


def prealloc_events(tags, n):
    """n event pairs per tag, created and recorded once NOW (outside any timed region): a torch.cuda.Event only creates its HIP event at its first
    record(), and that creation inside bench.py's timed loop stalled the host for tens of milliseconds once in a while (round 6: one default
    run in five lost 10-20 %; the kernels' own durations never moved)"""
    for tag in tags:
        pool = PROFILE_EVENT_POOL.setdefault(tag, [])
        while len(pool) < n:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            pool.append((a, b))


class _timed:
    __slots__ = ("tag", "e0", "e1")

    def __init__(self, tag, shape=None):
        if shape is not None and tag is not None and PROFILE_TAGS:
            PROFILE_SHAPES[tag] = shape
        self.tag = tag if (tag is not None and (tag in PROFILE_TAGS or "*" in PROFILE_TAGS)) else None

    def __enter__(self):
        if self.tag is not None:
            pool = PROFILE_EVENT_POOL.get(self.tag)
            self.e0, self.e1 = pool.pop() if pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.e0.record()

    def __exit__(self, *a):
        if self.tag is not None:
            self.e1.record()
            PROFILE_EVENTS.setdefault(self.tag, []).append((self.e0, self.e1))


# ------------------------------------------------------------------ op wrappers
def conv_fwd(x, w, bias, y, cin, cout, k, relu, packed=None, impl=_lib.IMPL_AUTO, tag=None):
    with _timed(tag, (x.n, x.h, x.w, cin, cout, k)):
        check(lib.mmif_conv2d_reflect_fwd(x.d, _ptr(w), _image(packed, "fwd", x.code), _ptr(bias), y.d,
                                          cin, cout, k, int(relu), impl, stream_ptr()), "conv2d_reflect_fwd")


def conv_dgrad(gy, w, x, gx, cin, cout, k, mask_bits=0, accum_bits=0, packed=None, impl=_lib.IMPL_AUTO, tag=None, fold=False):
    """fold=True: dgrad + fold_halo(gx) in one call (mmif_conv2d_reflect_dgrad_folded; gx's halo ring must be zero on entry);
    returns the folded view."""
    fn = lib.mmif_conv2d_reflect_dgrad_folded if fold else lib.mmif_conv2d_reflect_dgrad
    with _timed(tag, (gy.n, gy.h, gy.w, cin, cout, k)):
        check(fn(gy.d, _ptr(w), _image(packed, "dgrad", gy.code), x.d if x is not None else None, gx.d, cin, cout,
                 k, mask_bits, accum_bits, impl, stream_ptr()), "conv2d_reflect_dgrad")
    return gx.as_folded() if fold else gx


def dgrad_onto_supported(gy, gx, cin, cout, k):
    return bool(lib.mmif_conv2d_dgrad_onto_supported(gy.d, gx.d, cin, cout, k))


def conv_dgrad_dup_supported(gy, gx, cin, cout, k):
    return bool(lib.mmif_conv2d_dgrad_dup_supported(gy.d, gx.d, cin, cout, k))


def conv_dgrad_dup(gy, gx, cin, cout, k, packed, dup_out, dup_mask, frag, tag=None):
    """folded dgrad (nothing masked / accumulated) + the two masked copies of fragment `frag` in dup_out (csrc/conv_mfma.hip struct DupOut);
    returns gx as a folded view"""
    with _timed(tag):
        check(lib.mmif_conv2d_reflect_dgrad_folded_dup(gy.d, _image(packed, "dgrad", gy.code), gx.d, cin, cout, k, dup_out.d, dup_mask.d, frag,
                                                       stream_ptr()), "conv_dgrad_dup")
    return gx.as_folded()


def conv_dgrad_onto(gy, x, gx_old, gx, cin, cout, k, mask_bits, accum_bits, packed, tag=None):
    """gx = [mask](fold(dgrad(gy)) + gx_old) on the blocks in accum_bits: the accumulate operand comes from another tensor; returns the
    folded gx view (gx's halo ring must be zero on entry)"""
    with _timed(tag, (gy.n, gy.h, gy.w, cin, cout, k)):
        check(lib.mmif_conv2d_reflect_dgrad_folded_onto(gy.d, _ptr(packed.dgrad), x.d if x is not None else None, gx_old.d, gx.d, cin, cout, k,
                                                        mask_bits, accum_bits, stream_ptr()), "conv2d_reflect_dgrad_folded_onto")
    return gx.as_folded()


def conv_wgrad(x, gy, dw, db, cin, cout, k, ws, accumulate=False, impl=_lib.IMPL_AUTO, tag=None):
    with _timed(tag, (x.n, x.h, x.w, cin, cout, k)):
        check(lib.mmif_conv2d_reflect_wgrad(x.d, gy.d, _ptr(dw), _ptr(db), cin, cout, k, int(accumulate), _ptr(ws),
                                            ws.numel() * ws.element_size(), impl, stream_ptr()), "conv2d_reflect_wgrad")


def bwd_pair_supported(cin, cout, k):
    return bool(lib.mmif_conv2d_bwd_pair_supported(cin, cout, k))


def conv_bwd_pair(gy, x, gx, dw, db, cin, cout, k, packed, ws, accumulate=False, tag=None):
    """dgrad (all blocks masked by x, folded) + wgrad of one thin 3x3 layer in one launch; returns the folded gx view."""
    with _timed(tag, (gy.n, gy.h, gy.w, cin, cout, k)):
        check(lib.mmif_conv2d_reflect_bwd_pair(gy.d, _ptr(packed.dgrad), x.d, gx.d, _ptr(dw), _ptr(db), cin, cout, k, int(accumulate), _ptr(ws),
                                               ws.numel() * ws.element_size(), stream_ptr()), "conv2d_reflect_bwd_pair")
    return gx.as_folded()


def bwd_wide_supported(cin, cout, k, dtype=torch.bfloat16):
    if dtype == torch.float32:
        return bool(lib.mmif_conv2d_bwd_wide_supported_f32(cin, cout, k))
    return bool(lib.mmif_conv2d_bwd_wide_supported(cin, cout, k))


def bwd_wide_signs_bytes(n, cin, h, w):
    return lib.mmif_conv2d_bwd_wide_signs_bytes(n, cin, h, w)


def conv_bwd_wide(gy, x, gx, dw, db, cin, cout, k, packed, mask_bits, ws, signs, accumulate=False, tag=None):
    """wgrad + dgrad (blocks in mask_bits masked by x, folded) of one wide 3x3 layer; the wgrad kernel leaves x's ReLU sign bytes in
    `signs` (uint8 scratch, bwd_wide_signs_bytes) for the dgrad kernel; returns the folded gx view.  When the profiler asks for the
    layer's ':wgrad' / ':dgrad' tag the two halves run as two calls (phase bits of `accumulate`, mmif.h) -- the same two kernels in the
    same order, each bracketed by its own events."""
    def call(acc):
        check(lib.mmif_conv2d_reflect_bwd_wide(gy.d, _ptr(packed.dgrad), x.d, gx.d, _ptr(dw), _ptr(db), cin, cout, k, mask_bits, acc,
                                               _ptr(ws), ws.numel() * ws.element_size(), _ptr(signs), signs.numel(), stream_ptr()),
              "conv2d_reflect_bwd_wide")
    shape = (gy.n, gy.h, gy.w, cin, cout, k)
    base = tag[:-4] if (tag is not None and tag.endswith(":bwd")) else None
    if base is not None and PROFILE_TAGS and (base + ":wgrad" in PROFILE_TAGS or base + ":dgrad" in PROFILE_TAGS):
        with _timed(base + ":wgrad", shape):
            call(int(accumulate) | 2)
        with _timed(base + ":dgrad", shape):
            call(4)
    else:
        with _timed(tag, shape):
            call(int(accumulate))
    return gx.as_folded()


def image_in_fwd(img, w, bias, y, cout, k, relu):
    check(lib.mmif_conv2d_image_in_fwd(_ptr(img), _ptr(w), _ptr(bias), y.d, cout, k, int(relu), stream_ptr()), "image_in_fwd")


def _encoder_structs(branches):
    structs = []
    for img, w0, b0, packed, biases, out in branches:
        e = _lib.MmifDenseEncoder()
        e.img, e.w0, e.b0 = img.data_ptr(), w0.data_ptr(), (b0.data_ptr() if b0 is not None else None)
        for i in range(3):
            assert (packed[i].cout, packed[i].cin, packed[i].k) == (16, 16 + 16 * i, 3), "DenseBlock(16, 16) operand images expected"
            e.packed[i] = packed[i].fwd.data_ptr()
            e.bias[i] = biases[i].data_ptr() if biases[i] is not None else None
        structs.append((e, out))
    return structs


def dense_encoder_fwd(branches, tag=None):
    """ConvLayer(1,16) + DenseBlock(16,16) of one or two branches as ONE streaming launch (csrc/enc_stream.hip).
    branches: [(img fp32 [n,1,h,w], w0, b0, (PackedWeights x 3), (bias x 3), out 8-block bf16 view), ...]"""
    structs = _encoder_structs(branches)
    a, b = structs[0], (structs[1] if len(structs) > 1 else (None, None))
    with _timed(tag):
        check(lib.mmif_dense_encoder_fwd(C.byref(a[0]), a[1].d, C.byref(b[0]) if b[0] is not None else None,
                                         b[1].d if b[1] is not None else None, stream_ptr()), "dense_encoder_fwd")


def dense_encoder_fwd_sum(branches, out_sum, tag=None):
    """two branches that share ONE encoder (DenseFuse) + out_sum = out_a + out_b in one launch (csrc/enc_stream2.hip, dual form); returns False
    (nothing launched) when the library does not take the case -- the caller then runs dense_encoder_fwd + fuse_elem_fwd"""
    (ea, oa), (eb, ob) = _encoder_structs(branches)
    if not lib.mmif_dense_encoder_fwd_sum_supported(C.byref(ea), C.byref(eb), oa.n, oa.h, oa.w):
        return False
    with _timed(tag):
        check(lib.mmif_dense_encoder_fwd_sum(C.byref(ea), oa.d, C.byref(eb), ob.d, out_sum.d, stream_ptr()), "dense_encoder_fwd_sum")
    return True


def pack_dense_chain(w1, w2, w3, device, fmt=BF16):
    """dgrad operand images of the DenseBlock's three virtual gather layers (x0 <- convs 1..3, x1 <- convs 2..3, x2 <- conv 3);
    fmt = F32: the split-operand (x3) images for fp32 tensors."""
    pk = [PackedWeights(16 * (3 - k), 16, 3, device, fmt) for k in range(3)]
    repack_dense_chain(pk, w1, w2, w3)
    return pk


def repack_dense_chain(pk, w1, w2, w3):
    fn = lib.mmif_pack_dense_chain_x3 if pk[0].fmt == F32 else lib.mmif_pack_dense_chain
    check(fn(_ptr(w1), _ptr(w2), _ptr(w3), _ptr(pk[0].dgrad), _ptr(pk[1].dgrad), _ptr(pk[2].dgrad), stream_ptr()), "pack_dense_chain")


def repack_dense_chain_pair(pk_a, ws_a, pk_b, ws_b):
    """bf16 chain images of two encoder branches in ONE launch (mmif_pack_dense_chain_pair)"""
    assert pk_a[0].fmt == BF16 and pk_b[0].fmt == BF16
    arr = lambda ptrs: (C.c_void_p * 3)(*ptrs)
    keep = [arr([_ptr(w) for w in ws_a]), arr([_ptr(p.dgrad) for p in pk_a]), arr([_ptr(w) for w in ws_b]), arr([_ptr(p.dgrad) for p in pk_b])]
    check(lib.mmif_pack_dense_chain_pair(*[C.cast(a, C.c_void_p) for a in keep], stream_ptr()), "pack_dense_chain_pair")


def dense_encoder_chain(branches, tag=None):
    """The DenseBlock's backward gradient chain of one or two branches as ONE streaming launch (csrc/enc_chain.hip).
    branches: [(g3 2-block view, glow 6-block view, x 6-block view, chain images (pack_dense_chain), out 8-block view), ...]"""
    structs, keep = [], []
    for g3, glow, x, pk, out in branches:
        e = _lib.MmifDenseChain()
        e.g3, e.glow, e.x, e.out = C.pointer(g3._d), C.pointer(glow._d), C.pointer(x._d), C.pointer(out._d)
        for k in range(3):
            assert pk[k].fmt == BF16 and (pk[k].cout, pk[k].cin, pk[k].k) == (16 * (3 - k), 16, 3), "pack_dense_chain images expected"
            e.packed[k] = pk[k].dgrad.data_ptr()
        structs.append(e)
        keep.append((g3, glow, x, out))
    with _timed(tag):
        check(lib.mmif_dense_encoder_chain(C.byref(structs[0]), C.byref(structs[1]) if len(structs) > 1 else None, stream_ptr()), "dense_encoder_chain")


def dense_encoder_bwd_workspace_bytes():
    return lib.mmif_dense_encoder_bwd_workspace()


def dense_encoder_bwd(branches, ws, tag=None):
    """The whole backward of ConvLayer(1,16) + DenseBlock(16,16) of one or two branches as ONE streaming launch (csrc/enc_bwd.hip): the
    gradient chain and all four layers' weight gradients; nothing but dW / db is written.
    branches: [(g3 2-block view, glow 6-block view, x 6-block view, chain images (pack_dense_chain), img fp32 [n,1,h,w],
                [(dw0, db0), (dw1, db1), (dw2, db2), (dw3, db3)], accumulate), ...]"""
    structs, ptrs, keep = [], [], []
    for g3, glow, x, pk, img, grads, acc in branches:
        e = _lib.MmifDenseChain()
        e.g3, e.glow, e.x, e.out = C.pointer(g3._d), C.pointer(glow._d), C.pointer(x._d), None
        for k in range(3):
            assert pk[k].fmt == BF16 and (pk[k].cout, pk[k].cin, pk[k].k) == (16 * (3 - k), 16, 3), "pack_dense_chain images expected"
            e.packed[k] = pk[k].dgrad.data_ptr()
        arr = (C.c_void_p * 8)(*[t.data_ptr() if t is not None else None for pair in grads for t in pair])
        structs.append(e)
        ptrs.append(arr)
        keep.append((g3, glow, x, img, grads))
    a, b = 0, (1 if len(structs) > 1 else None)
    with _timed(tag):
        check(lib.mmif_dense_encoder_bwd(C.byref(structs[a]), _ptr(branches[a][4]), ptrs[a], int(branches[a][6]),
                                         C.byref(structs[b]) if b is not None else None, _ptr(branches[b][4]) if b is not None else None,
                                         ptrs[b] if b is not None else None, int(branches[b][6]) if b is not None else 0,
                                         _ptr(ws), ws.numel() * ws.element_size(), stream_ptr()), "dense_encoder_bwd")


def dense_encoder_wgrad_workspace_bytes():
    return lib.mmif_dense_encoder_wgrad_workspace()


def dense_encoder_wgrad(img, x, gz, grads, ws, accumulate=False, tag=None):
    """dW, db of ConvLayer(1,16) + DenseBlock(16,16) in one pass (csrc/enc_wgrad.hip).  grads = [(dw0, db0), (dw1, db1), (dw2, db2), (dw3, db3)]"""
    flat = [_ptr(t) for pair in grads for t in pair]
    with _timed(tag):
        check(lib.mmif_dense_encoder_wgrad(_ptr(img), x.d, gz.d, *flat, int(accumulate), _ptr(ws), ws.numel() * ws.element_size(), stream_ptr()),
              "dense_encoder_wgrad")


def image_in_wgrad(img, gy, dw, db, cout, k, ws, accumulate=False):
    check(lib.mmif_conv2d_image_in_wgrad(_ptr(img), gy.d, _ptr(dw), _ptr(db), cout, k, int(accumulate), _ptr(ws),
                                         ws.numel() * ws.element_size(), stream_ptr()), "image_in_wgrad")


def image_out_fwd(x, w, bias, img, cin, k, relu):
    check(lib.mmif_conv2d_image_out_fwd(x.d, _ptr(w), _ptr(bias), _ptr(img), cin, k, int(relu), stream_ptr()), "image_out_fwd")


def image_out_dgrad(gimg, yimg, w, x, gx, cin, k, mask_bits=0, accum_bits=0):
    check(lib.mmif_conv2d_image_out_dgrad(_ptr(gimg), _ptr(yimg), _ptr(w), x.d if x is not None else None, gx.d, cin, k,
                                          mask_bits, accum_bits, stream_ptr()), "image_out_dgrad")


def image_out_wgrad(x, gimg, yimg, dw, db, cin, k, ws, accumulate=False):
    check(lib.mmif_conv2d_image_out_wgrad(x.d, _ptr(gimg), _ptr(yimg), _ptr(dw), _ptr(db), cin, k, int(accumulate), _ptr(ws),
                                          ws.numel() * ws.element_size(), stream_ptr()), "image_out_wgrad")


def image_out_bwd_supported(x, cin, k):
    """the fused backward of the 16 -> 1 layer (csrc/image_bwd.hip) takes this activation tensor"""
    return bool(lib.mmif_conv2d_image_out_bwd_supported(x.code, cin, k, x.h, x.w)) and x.halo == 0 and x.cb == 2


def image_out_bwd(x, gimg, yimg, w, gx, dw, db, cin, k, ws, accumulate=False):
    """dL/dx (reflect adjoint applied, masked by x > 0, into the interior of the zero-ringed halo-1 tensor gx), dW, db of the Cout = 1 layer in
    ONE launch; returns gx as a folded view"""
    check(lib.mmif_conv2d_image_out_bwd(x.d, _ptr(gimg), _ptr(yimg), _ptr(w), gx.d, _ptr(dw), _ptr(db), cin, k, int(accumulate), _ptr(ws),
                                        ws.numel() * ws.element_size(), stream_ptr()), "image_out_bwd")
    return gx.as_folded()


def fuse_elem_fwd(a, b, out, mode):
    check(lib.mmif_fuse_elem_fwd(a.d, b.d, out.d, mode, stream_ptr()), "fuse_elem_fwd")


def fuse_elem_bwd(a, b, g, ga, gb, mode, relu_mask):
    check(lib.mmif_fuse_elem_bwd(a.d if a is not None else None, b.d if b is not None else None, g.d, ga.d, gb.d, mode,
                                 int(relu_mask), stream_ptr()), "fuse_elem_bwd")


def wgrad_workspace_bytes(cin, cout, k):
    return lib.mmif_conv2d_wgrad_workspace(cin, cout, k)


def image_wgrad_workspace_bytes(c, k):
    return lib.mmif_conv2d_image_wgrad_workspace(c, k)


def maxpool_fwd(x, y):
    check(lib.mmif_maxpool2x2_fwd(x.d, y.d, stream_ptr()), "maxpool2x2_fwd")


def maxpool_bwd(x, g, gx, accumulate, relu=False):
    """relu=True: also gx *= [x > 0] (x is a ReLU output and this is the last contribution to its gradient)"""
    fn = lib.mmif_maxpool2x2_bwd_relu if relu else lib.mmif_maxpool2x2_bwd
    check(fn(x.d, g.d, gx.d, int(accumulate), stream_ptr()), "maxpool2x2_bwd")


def upsample_fwd(x, y):
    check(lib.mmif_upsample2x_fwd(x.d, y.d, stream_ptr()), "upsample2x_fwd")


def upsample_bwd(g, gx, accumulate, relu_of=None):
    """relu_of: the ReLU output whose gradient gx is -- gx *= [relu_of > 0] after this (last) contribution"""
    if relu_of is not None:
        check(lib.mmif_upsample2x_bwd_relu(g.d, gx.d, int(accumulate), relu_of.d, stream_ptr()), "upsample2x_bwd_relu")
    else:
        check(lib.mmif_upsample2x_bwd(g.d, gx.d, int(accumulate), stream_ptr()), "upsample2x_bwd")


def relu_mask_(x, g):
    check(lib.mmif_relu_mask(x.d, g.d, stream_ptr()), "relu_mask")


ATTN_MODES = {"sa": 0, "ca": 1, "sca": 2}


def attn_workspace(n, c, device):
    return torch.empty(lib.mmif_fuse_attn_workspace(n, c) // 4 + 1, dtype=torch.float32, device=device)


def attn_fwd(a, b, out, mode, ws):
    check(lib.mmif_fuse_attn_fwd(a.d, b.d, out.d, mode, _ptr(ws), ws.numel() * 4, stream_ptr()), "fuse_attn_fwd")


def attn_bwd(a, b, g, ga, gb, mode, accumulate, ws, cached=False):
    """cached=True: ws is the workspace attn_fwd ran on for these a, b, mode and has not been written since (channel sums reused)"""
    fn = lib.mmif_fuse_attn_bwd_cached if cached else lib.mmif_fuse_attn_bwd
    check(fn(a.d, b.d, g.d, ga.d, gb.d, mode, int(accumulate), _ptr(ws), ws.numel() * 4, stream_ptr()), "fuse_attn_bwd")


def _d(t):
    return t.d if t is not None else None


def pairconv_fwd(a, b, w, bias, nout, oa, ob, relu, res1=None, res2=None):
    check(lib.mmif_pairconv_fwd(a.d, b.d, _ptr(w), _ptr(bias), nout, oa.d, _d(ob), int(relu), _d(res1), _d(res2), stream_ptr()), "pairconv_fwd")


def pairconv_dgrad(ga, gb, w, nout, xa, xb, gxa, gxb, mask_bits=0, add=None):
    check(lib.mmif_pairconv_dgrad(ga.d, _d(gb), _ptr(w), nout, _d(xa), _d(xb), gxa.d, gxb.d, mask_bits, _d(add), stream_ptr()), "pairconv_dgrad")


def pairconv_wgrad_workspace_bytes():
    return lib.mmif_pairconv_wgrad_workspace()


def pairconv_wgrad(xa, xb, ga, gb, nout, dw, db, ws, accumulate=False):
    check(lib.mmif_pairconv_wgrad(xa.d, xb.d, ga.d, _d(gb), nout, _ptr(dw), _ptr(db), int(accumulate), _ptr(ws), ws.numel() * ws.element_size(),
                                  stream_ptr()), "pairconv_wgrad")


def pairconv_bwd(ga, gb, w, nout, xa, xb, gxa, gxb, dw, db, ws, mask_bits=0, add=None, accumulate=False):
    """pairconv_dgrad + pairconv_wgrad of one layer in one pass over (g, x)"""
    check(lib.mmif_pairconv_bwd(ga.d, _d(gb), _ptr(w), nout, xa.d, xb.d, gxa.d, gxb.d, mask_bits, _d(add), _ptr(dw), _ptr(db), int(accumulate),
                                _ptr(ws), ws.numel() * ws.element_size(), stream_ptr()), "pairconv_bwd")


# ------------------------------------------------------------------ general ConvLayer primitives (plain NCHW fp32; row n4)
def _f32c(t, name):
    require_device(t, name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"mmif: {name} must be a contiguous fp32 tensor")
    return t


def gconv_out_size(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def gconv_fwd(x, w, bias, stride, padding, reflect, relu):
    """nn.Conv2d(k in 1/3/5/7, stride 1/2, reflect | zero padding) (+ ReLU) on NCHW fp32."""
    _f32c(x, "x"), _f32c(w, "weight")
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    y = torch.empty((n, cout, gconv_out_size(h, k, stride, padding), gconv_out_size(wd, k, stride, padding)), dtype=torch.float32, device=x.device)
    check(lib.mmif_gconv_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, cin, cout, h, wd, k, stride, padding, int(reflect), int(relu),
                             stream_ptr()), "gconv_fwd")
    return y


def gconv_dgrad(gy, w, x_shape, stride, padding, reflect):
    _f32c(gy, "gy"), _f32c(w, "weight")
    n, cin, h, wd = x_shape
    cout, _, k, _ = w.shape
    dx = torch.empty(x_shape, dtype=torch.float32, device=gy.device)
    nb = lib.mmif_gconv_dgrad_workspace(n, cin, h, wd, padding, int(reflect))
    ws = torch.empty(nb // 4 + 1, dtype=torch.float32, device=gy.device)
    check(lib.mmif_gconv_dgrad(_ptr(gy), _ptr(w), _ptr(dx), n, cin, cout, h, wd, k, stride, padding, int(reflect), _ptr(ws), ws.numel() * 4,
                               stream_ptr()), "gconv_dgrad")
    return dx


def gconv_wgrad(x, gy, k, stride, padding, reflect, want_bias=True):
    _f32c(x, "x"), _f32c(gy, "gy")
    n, cin, h, wd = x.shape
    cout = gy.shape[1]
    dw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    ws = torch.empty(lib.mmif_gconv_wgrad_workspace(cin, cout, k) // 4 + 1, dtype=torch.float32, device=x.device)
    check(lib.mmif_gconv_wgrad(_ptr(x), _ptr(gy), _ptr(dw), _ptr(db), n, cin, cout, h, wd, k, stride, padding, int(reflect), _ptr(ws),
                               ws.numel() * 4, stream_ptr()), "gconv_wgrad")
    return dw, db


def gconvt_fwd(x, w, bias, stride, padding, output_padding, relu=False):
    """nn.ConvTranspose2d (weight [cin][cout][k][k]) (+ ReLU) on NCHW fp32."""
    _f32c(x, "x"), _f32c(w, "weight")
    n, cin, h, wd = x.shape
    _, cout, k, _ = w.shape
    ho, wo = (h - 1) * stride - 2 * padding + k + output_padding, (wd - 1) * stride - 2 * padding + k + output_padding
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
    check(lib.mmif_gconvt_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, cin, cout, h, wd, k, stride, padding, output_padding, int(relu),
                              stream_ptr()), "gconvt_fwd")
    return y


def gconvt_dgrad(gy, w, x_shape, stride, padding, output_padding):
    _f32c(gy, "gy"), _f32c(w, "weight")
    n, cin, h, wd = x_shape
    _, cout, k, _ = w.shape
    dx = torch.empty(x_shape, dtype=torch.float32, device=gy.device)
    check(lib.mmif_gconvt_dgrad(_ptr(gy), _ptr(w), _ptr(dx), n, cin, cout, h, wd, k, stride, padding, output_padding, stream_ptr()), "gconvt_dgrad")
    return dx


def gconvt_wgrad(x, gy, k, stride, padding, output_padding):
    _f32c(x, "x"), _f32c(gy, "gy")
    n, cin, h, wd = x.shape
    cout = gy.shape[1]
    dw = torch.empty((cin, cout, k, k), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.mmif_gconv_wgrad_workspace(cout, cin, k) // 4 + 1, dtype=torch.float32, device=x.device)
    check(lib.mmif_gconvt_wgrad(_ptr(x), _ptr(gy), _ptr(dw), None, n, cin, cout, h, wd, k, stride, padding, output_padding, _ptr(ws),
                                ws.numel() * 4, stream_ptr()), "gconvt_wgrad")
    return dw


def relu_bwd(g, y):
    _f32c(g, "g"), _f32c(y, "y")
    out = torch.empty_like(g)
    check(lib.mmif_relu_bwd(_ptr(g), _ptr(y), _ptr(out), g.numel(), stream_ptr()), "relu_bwd")
    return out


def channel_sum(x):
    """[n][c][h][w] fp32 -> per-channel sums [c] (deterministic)."""
    _f32c(x, "x")
    n, c = x.shape[0], x.shape[1]
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib.mmif_channel_sum(_ptr(x), _ptr(out), n, c, x[0, 0].numel(), stream_ptr()), "channel_sum")
    return out


def bilinear_up_fwd(x, scale):
    """nn.Upsample(scale_factor=scale, mode='bilinear', align_corners=True) on NCHW fp32."""
    _f32c(x, "x")
    n, c, h, w = x.shape
    out = torch.empty((n, c, h * scale, w * scale), dtype=torch.float32, device=x.device)
    check(lib.mmif_bilinear_up_fwd(_ptr(x), _ptr(out), n * c, h, w, h * scale, w * scale, stream_ptr()), "bilinear_up_fwd")
    return out


def bilinear_up_bwd(g, in_hw):
    _f32c(g, "g")
    n, c, H, W = g.shape
    dx = torch.empty((n, c, in_hw[0], in_hw[1]), dtype=torch.float32, device=g.device)
    check(lib.mmif_bilinear_up_bwd(_ptr(g), _ptr(dx), n * c, in_hw[0], in_hw[1], H, W, stream_ptr()), "bilinear_up_bwd")
    return dx


# ------------------------------------------------------------------ norm + activation epilogues (row n4)
NORM_BN_TRAIN, NORM_BN_EVAL, NORM_GN = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH, ACT_RELU6 = 0, 1, 2, 3, 4


def _norm_ws(n, c, dev):
    return torch.empty(lib.mmif_norm_workspace(n, c) // 4 + 2, dtype=torch.float32, device=dev)


def norm_act_fwd(x, gamma, beta, running_mean, running_var, kind, eps, momentum, act, slope=0.2):
    """y = act(norm(x)) (BatchNorm2d train / eval, GroupNorm(c, c)); returns (y, stats)."""
    _f32c(x, "x")
    n, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    y = torch.empty_like(x)
    stats = torch.empty(2 * (n * c if kind == NORM_GN else c), dtype=torch.float32, device=x.device)
    ws = _norm_ws(n, c, x.device)
    check(lib.mmif_norm_act_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), _ptr(running_mean), _ptr(running_var), n, c, hw, kind,
                                float(eps), float(momentum), act, float(slope), _ptr(ws), ws.numel() * 4, stream_ptr()), "norm_act_fwd")
    return y, stats


def norm_act_bwd(x, y, gy, stats, gamma, kind, act, slope=0.2, want_affine=True):
    _f32c(x, "x"), _f32c(y, "y"), _f32c(gy, "gy")
    n, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    dx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    ws = _norm_ws(n, c, x.device)
    check(lib.mmif_norm_act_bwd(_ptr(x), _ptr(y), _ptr(gy), _ptr(stats), _ptr(gamma), _ptr(dx), _ptr(dg), _ptr(db), n, c, hw, kind, act,
                                float(slope), _ptr(ws), ws.numel() * 4, stream_ptr()), "norm_act_bwd")
    return dx, dg, db


# cross-rank BatchNorm: statistics and apply stages (the caller all-reduces `chan` = [c][2] fp64 sums in between; mmif/dist.py)
def bn_moments(x):
    _f32c(x, "x")
    n, c = x.shape[0], x.shape[1]
    chan = torch.empty(2 * c + 1, dtype=torch.float64, device=x.device)   # [c][2] sums + the element count (rides in the all-reduce)
    ws = _norm_ws(n, c, x.device)
    check(lib.mmif_bn_moments(_ptr(x), _ptr(chan), n, c, x[0, 0].numel(), _ptr(ws), ws.numel() * 4, stream_ptr()), "bn_moments")
    return chan


def bn_apply_fwd(x, chan, gamma, beta, running_mean, running_var, eps, momentum, act, slope=0.2):
    n, c = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    stats = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    check(lib.mmif_bn_apply_fwd(_ptr(x), _ptr(chan), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), _ptr(running_mean),
                                _ptr(running_var), n, c, x[0, 0].numel(), float(eps), float(momentum), act, float(slope), stream_ptr()),
          "bn_apply_fwd")
    return y, stats


def bn_bwd_sums(x, y, gy, stats, act, slope=0.2, want_affine=True):
    _f32c(x, "x"), _f32c(y, "y"), _f32c(gy, "gy")
    n, c = x.shape[0], x.shape[1]
    chan = torch.empty(2 * c, dtype=torch.float64, device=x.device)
    dg = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    ws = _norm_ws(n, c, x.device)
    check(lib.mmif_bn_bwd_sums(_ptr(x), _ptr(y), _ptr(gy), _ptr(stats), _ptr(chan), _ptr(dg), _ptr(db), n, c, x[0, 0].numel(), act,
                               float(slope), _ptr(ws), ws.numel() * 4, stream_ptr()), "bn_bwd_sums")
    return chan, dg, db


def bn_apply_bwd(x, y, gy, stats, gamma, chan, count, act, slope=0.2):
    """count: 1-element fp64 device tensor (the forward's all-reduced element count)"""
    n, c = x.shape[0], x.shape[1]
    dx = torch.empty_like(x)
    check(lib.mmif_bn_apply_bwd(_ptr(x), _ptr(y), _ptr(gy), _ptr(stats), _ptr(gamma), _ptr(chan), _ptr(count), _ptr(dx), n, c,
                                x[0, 0].numel(), act, float(slope), stream_ptr()), "bn_apply_bwd")
    return dx


def act_fwd(x, act, slope=0.2):
    _f32c(x, "x")
    y = torch.empty_like(x)
    check(lib.mmif_act_fwd(_ptr(x), _ptr(y), x.numel(), act, float(slope), stream_ptr()), "act_fwd")
    return y


def act_bwd(gy, y, act, slope=0.2):
    _f32c(gy, "gy"), _f32c(y, "y")
    dx = torch.empty_like(gy)
    check(lib.mmif_act_bwd(_ptr(gy), _ptr(y), _ptr(dx), gy.numel(), act, float(slope), stream_ptr()), "act_bwd")
    return dx


# ------------------------------------------------------------------ depth-wise conv (groups == channels)
def dwconv_fwd(x, w, bias, reflect):
    _f32c(x, "x"), _f32c(w, "weight")
    n, c, h, wd = x.shape
    y = torch.empty_like(x)
    check(lib.mmif_dwconv_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, c, h, wd, w.shape[2], int(reflect), stream_ptr()), "dwconv_fwd")
    return y


def dwconv_dgrad(gy, w, reflect):
    _f32c(gy, "gy"), _f32c(w, "weight")
    n, c, h, wd = gy.shape
    dx = torch.empty_like(gy)
    check(lib.mmif_dwconv_dgrad(_ptr(gy), _ptr(w), _ptr(dx), n, c, h, wd, w.shape[2], int(reflect), stream_ptr()), "dwconv_dgrad")
    return dx


def dwconv_wgrad(x, gy, k, reflect, want_bias):
    _f32c(x, "x"), _f32c(gy, "gy")
    n, c, h, wd = x.shape
    dw = torch.empty((c, 1, k, k), dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_bias else None
    check(lib.mmif_dwconv_wgrad(_ptr(x), _ptr(gy), _ptr(dw), _ptr(db), n, c, h, wd, k, int(reflect), stream_ptr()), "dwconv_wgrad")
    return dw, db


# ------------------------------------------------------------------ resampling glue on NCHW fp32 (max-pool, nearest up-sampling, reflect pad / crop)
def maxpool_nchw_fwd(x, k):
    _f32c(x, "x")
    n, c, h, w = x.shape
    y = torch.empty((n, c, h // k, w // k), dtype=torch.float32, device=x.device)
    idx = torch.empty((n, c, h // k, w // k), dtype=torch.uint8, device=x.device)
    check(lib.mmif_maxpool_nchw_fwd(_ptr(x), _ptr(y), _ptr(idx), n * c, h, w, k, stream_ptr()), "maxpool_nchw_fwd")
    return y, idx


def maxpool_nchw_bwd(g, idx, in_hw, k):
    _f32c(g, "g")
    n, c = g.shape[0], g.shape[1]
    dx = torch.empty((n, c, in_hw[0], in_hw[1]), dtype=torch.float32, device=g.device)
    check(lib.mmif_maxpool_nchw_bwd(_ptr(g), _ptr(idx), _ptr(dx), n * c, in_hw[0], in_hw[1], k, stream_ptr()), "maxpool_nchw_bwd")
    return dx


def nearest_up_fwd(x, scale):
    _f32c(x, "x")
    n, c, h, w = x.shape
    y = torch.empty((n, c, h * scale, w * scale), dtype=torch.float32, device=x.device)
    check(lib.mmif_nearest_up_fwd(_ptr(x), _ptr(y), n * c, h, w, scale, stream_ptr()), "nearest_up_fwd")
    return y


def nearest_up_bwd(g, scale):
    _f32c(g, "g")
    n, c, H, W = g.shape
    dx = torch.empty((n, c, H // scale, W // scale), dtype=torch.float32, device=g.device)
    check(lib.mmif_nearest_up_bwd(_ptr(g), _ptr(dx), n * c, H // scale, W // scale, scale, stream_ptr()), "nearest_up_bwd")
    return dx


def reflect_pad_fwd(x, pads):
    """pads = (left, right, top, bottom) as nn.ReflectionPad2d; negative amounts crop."""
    _f32c(x, "x")
    n, c, h, w = x.shape
    l, r, t, b = pads
    y = torch.empty((n, c, h + t + b, w + l + r), dtype=torch.float32, device=x.device)
    check(lib.mmif_reflect_pad_fwd(_ptr(x), _ptr(y), n * c, h, w, l, r, t, b, stream_ptr()), "reflect_pad_fwd")
    return y


def reflect_pad_bwd(g, in_hw, pads):
    _f32c(g, "g")
    n, c = g.shape[0], g.shape[1]
    l, r, t, b = pads
    dx = torch.empty((n, c, in_hw[0], in_hw[1]), dtype=torch.float32, device=g.device)
    check(lib.mmif_reflect_pad_bwd(_ptr(g), _ptr(dx), n * c, in_hw[0], in_hw[1], l, r, t, b, stream_ptr()), "reflect_pad_bwd")
    return dx
