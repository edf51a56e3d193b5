"""Whole-model execution engine: runs the encoder / fusion / decoder forward and backward of the
fusion nets (reference core/model.py) as a fixed sequence of HIP kernel launches on blocked-NHWC
buffers, with torch.cat replaced by channel-block views of one allocation (zero-copy concat).

Gradient convention (DESIGN.md "padded-domain gradients"): a dgrad kernel writes the gradient of
the reflect-PADDED input ([h+2][w+2], halo = 1); a tiny fold kernel (mmif_fold_halo) then adds the
halo onto the interior (adjoint of reflect padding) and zeroes it, so readers see an ordinary
gradient.  The ReLU mask of a producer layer is applied by the LAST kernel that contributes to that
tensor's gradient (mask_bits), accumulation of several consumers' contributions by accum_bits.
"""
import os

import torch

from . import _lib
from . import dist as D
from . import tensor as T
from .tensor import BT, PackedWeights

class _Cfg:
    """process-wide settings (one process per GPU; a forward on any thread sees set_compute_dtype())"""
    dtype = None


_cfg = _Cfg()

# base data_ptr -> flat gradient buffer (lets mmif.optim find the buffer behind a parameter's .grad view)
FLAT_BUFFERS = {}
# None, or a list that ModelEngine.run() fills with (parameter, stand-in leaf) pairs: while set, the autograd node of a forward hangs off
# fresh leaves instead of the parameters, so that the gradients are accumulated by AccumulateGrad nodes created NOW, on the current
# stream (a parameter's own accumulator lives -- bound to the stream it was created on -- as long as any older autograd graph does)
FRESH_LEAVES = [None]
# bumped by every fused optimiser step: the packed bf16 weight images must be rebuilt
WEIGHTS_EPOCH = [0]
# id()s of pooled scratch buffers whose addresses are baked into a captured hipGraph (never evicted from their pools)
GRAPH_PINNED = set()
GRAD_TAIL = 8  # scalar slots after the gradients (loss values ride along in the gradient all-reduce)


def compute_dtype():
    """Storage dtype of feature maps: torch.float32 (parity path, VALU kernels) or torch.bfloat16
    (throughput path, MFMA kernels).  Default from $MMIF_DTYPE, else float32."""
    d = getattr(_cfg, "dtype", None)
    if d is None:
        d = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16}.get(os.environ.get("MMIF_DTYPE", "fp32").lower(), torch.float32)
        _cfg.dtype = d
    return d


def set_compute_dtype(dtype):
    if isinstance(dtype, str):
        dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[dtype.lower()]
    assert dtype in (torch.float32, torch.bfloat16)
    _cfg.dtype = dtype


# A/B switches ($MMIF_...): read from the environment ONCE and cached -- the per-step host path only looks them up in this dict.  A process
# that changes one at run time (the tests do) calls reload_switches().
_SW = {}


def switch(name, default="1"):
    v = _SW.get(name)
    if v is None:
        v = _SW[name] = os.environ.get(name, default) != "0"
    return v


def switch_int(name, default):
    v = _SW.get(name)
    if v is None:
        v = _SW[name] = int(os.environ.get(name, str(default)))
    return v


def reload_switches():
    _SW.clear()


def conv_impl():
    """MMIF_IMPL_AUTO unless $MMIF_CONV_IMPL = valu | mfma | x3 (cross-checking the kernel families)."""
    v = _SW.get("MMIF_CONV_IMPL")
    if v is None:
        v = _SW["MMIF_CONV_IMPL"] = {"valu": _lib.IMPL_VALU, "mfma": _lib.IMPL_MFMA, "x3": _lib.IMPL_X3}.get(
            os.environ.get("MMIF_CONV_IMPL", "auto").lower(), _lib.IMPL_AUTO)
    return v


def x3_enabled():
    """fp32 tensors: 3x3 layers on the matrix pipe as split-bf16 products (csrc/conv_x3.hip, ~1e-5 of the fp32 FMA kernels' results);
    $MMIF_X3=0 keeps them on the fp32 FMA kernels.  The LIBRARY owns the switch (it reads the environment once per process): asking it
    keeps the engine's launch plan and the kernels' dispatch in step even if the environment changes later."""
    return bool(_lib.lib.mmif_get_x3_enabled())


def set_x3_forward_pieces(pieces):
    """Operand format of the x3 FORWARD kernels: 16 (default) = two scaled fp16 pieces, three products per tap, fp32-grade; 3 = three
    bf16 pieces, six products, fp32-grade (weights beyond |w| >= 64 too); 2 = two bf16 pieces, three products, activations within ~1e-5 of
    fp32.  Process wide; operand images are re-packed on the next forward."""
    assert pieces in (2, 3, 16)
    if _lib.lib.mmif_get_x3_forward_pieces() != pieces:
        _lib.lib.mmif_set_x3_forward_pieces(pieces)
        WEIGHTS_EPOCH[0] += 1


def wants_packed(dtype, impl, standalone=False):
    """do the conv kernels this (storage dtype, impl) pair selects take operand images?  standalone: a ConvLayer called on its own
    (core/block.py: the layer-by-layer nets of row n4) -- those keep fp32 tensors on the exact fp32 FMA kernels unless asked
    ($MMIF_X3_LAYERS=1 / impl x3): their BatchNorm / GroupNorm layers amplify the split products' 1e-5 (DIFNet at batch 2, 32 x 32:
    5e-3 on the first layer's weight gradient, over the 2e-3 its golden test allows), the model engines have no normalisation."""
    if impl == _lib.IMPL_VALU:
        return False
    if dtype == torch.bfloat16:
        return True
    if standalone and impl != _lib.IMPL_X3 and not switch("MMIF_X3_LAYERS", "0"):
        return False
    return x3_enabled()


def bits(*blocks):
    m = 0
    for b in blocks:
        m |= 1 << b
    return m


def all_bits(n):
    return (1 << n) - 1


class ConvSpec:
    """One ConvLayer as the engine sees it: fp32 master weight/bias (nn.Parameter), geometry, the
    packed bf16 operand images and where its gradients go.

    split = Cf > 0: the layer's reference input is cat(feats[Cf], up[Cu]) but the engine's buffer holds
    [up | feats] (NestDecoder rows, so every concat is a contiguous PREFIX of one allocation): the kernels
    see the input-channel-permuted weights `wperm`, and dW is permuted back into the parameter's order."""

    def __init__(self, name, conv, relu, split=0):
        self.name = name
        self.conv = conv  # nn.Conv2d holding .weight / .bias
        self.cout, self.cin, self.k = conv.weight.shape[0], conv.weight.shape[1], conv.weight.shape[2]
        self.relu = relu
        self.split = split
        self.pair = False   # PFNetv2 fuse layer: runs on the pair-conv kernels (fp32 weights, nothing to pack)
        self.wperm = None
        self.dw_tmp = None
        self.packed = None
        self.packed_version = None
        self.perm_version = None
        self.dw = None  # views into the flat gradient buffer, set per backward
        self.db = None

    @property
    def w(self):
        return self.wperm if self.split else self.conv.weight

    @property
    def b(self):
        return self.conv.bias

    def _key(self, version):
        w = self.conv.weight
        return (version, WEIGHTS_EPOCH[0], w._version, w.data_ptr())

    def refresh(self, version, pack, batch=None, fmt=_lib.BF16):
        """Bring the derived weight images (channel permutation, MFMA operand packing in the format of the storage dtype) up to
        date.  With `batch` (a list) the packing itself is left to the caller: (PackedWeights, weight) pairs for ONE T.pack_many
        launch."""
        key = self._key(version)
        if self.split and self.perm_version != key:
            w = self.conv.weight.detach()
            self.wperm = torch.cat((w[:, self.split:], w[:, :self.split]), dim=1).contiguous()
            self.perm_version = key
        if pack and self.cin > 1 and self.cout > 1 and not self.pair:
            w = self.conv.weight
            if self.packed is None or self.packed.fwd.device != w.device or self.packed.fmt != fmt:
                self.packed = PackedWeights(self.cout, self.cin, self.k, w.device, fmt)
                self.packed_version = None
            if self.packed_version != key:
                if batch is not None:
                    batch.append((self.packed, self.w.detach()))
                else:
                    self.packed.pack(self.w.detach())
                self.packed_version = key

    def ensure_packed(self, version):
        self.refresh(version, True)

    def grad_target(self):
        """Where the wgrad kernel writes dW: the parameter's gradient view, or a scratch in buffer order."""
        if not self.split:
            return self.dw
        if self.dw_tmp is None or self.dw_tmp.device != self.dw.device:
            self.dw_tmp = torch.empty_like(self.dw)
        return self.dw_tmp

    def finish_grad(self, accumulate):
        if not self.split:
            return
        cu = self.cin - self.split   # buffer order [up(cu) | feats(split)] -> parameter order [feats | up]
        t = self.dw_tmp
        if accumulate:
            self.dw[:, :self.split] += t[:, cu:]
            self.dw[:, self.split:] += t[:, :cu]
        else:
            self.dw[:, :self.split] = t[:, cu:]
            self.dw[:, self.split:] = t[:, :cu]


class Lease:
    """Per-call activation / gradient buffers; reused across iterations through the engine's pool."""

    def __init__(self):
        self.bufs = {}
        self.key = None
        self.imgs = None
        self.out = None
        self.busy = False


class ModelEngine:
    """Base: buffer pool, flat parameter / gradient storage, autograd glue."""

    def __init__(self, module, specs):
        self.module = module
        self.specs = specs  # ordered list of ConvSpec (any order; grads are matched by parameter)
        self.pool = {}
        self.weights_version = 0
        self.flat_grads = [None, None]
        self.flat_idx = 0
        self._ws = None
        self._params = None

    # ---- parameters -------------------------------------------------------------------------
    def params(self):
        if self._params is None:
            self._params = list(self.module.parameters())
        return self._params

    def bump_weights(self):
        self.weights_version += 1

    def _grad_buffer(self, device):
        """A flat fp32 gradient buffer that no live .grad aliases (two are rotated)."""
        ps = self.params()
        total = sum(p.numel() for p in ps)
        live = {p.grad.data_ptr() for p in ps if p.grad is not None}
        for i in (self.flat_idx, 1 - self.flat_idx):
            buf = self.flat_grads[i]
            if buf is None or buf.device != device or buf.numel() != total + GRAD_TAIL:
                if buf is not None:
                    FLAT_BUFFERS.pop(buf.data_ptr(), None)
                buf = torch.zeros(total + GRAD_TAIL, dtype=torch.float32, device=device)
                FLAT_BUFFERS[buf.data_ptr()] = buf
                self.flat_grads[i] = buf
            lo, hi = buf.data_ptr(), buf.data_ptr() + buf.numel() * 4
            if not any(lo <= a < hi for a in live):
                self.flat_idx = 1 - i
                return buf
        return torch.zeros(total + GRAD_TAIL, dtype=torch.float32, device=device)  # both aliased: fresh allocation

    def _assign_grad_views(self, device):
        flat = self._grad_buffer(device)
        views, off = {}, 0
        self._goff = {}
        for p in self.params():
            views[id(p)] = flat[off:off + p.numel()].view(p.shape)
            self._goff[id(p)] = (off, p.numel())
            off += p.numel()
        for s in self.specs:
            s.dw = views[id(s.conv.weight)]
            s.db = views[id(s.conv.bias)]
        # hand autograd FRESH view objects (sole owners), so AccumulateGrad can adopt them as .grad without a copy
        return flat, [views[id(p)].view(p.shape) for p in self.params()]

    def early_reduce(self, flat, specs):
        """Data parallel: start the all-reduce of these layers' finished gradients (a contiguous range of the flat buffer, + the staged
        loss scalars when the range ends at the tail) while the rest of the backward runs -- mmif/dist.py.  Plain flow only: a
        backward that finds .grad already set is accumulating, and its gradients are not final."""
        if not D.early_reduce_armed() or any(p.grad is not None for p in self.params()):
            return False
        rng = [self._goff[id(t)] for s in specs for t in (s.conv.weight, s.conv.bias) if t is not None and id(t) in self._goff]
        if not rng:
            return False
        lo, hi = min(o for o, _ in rng), max(o + n for o, n in rng)
        if sum(n for _, n in rng) != hi - lo:      # not contiguous in the parameter order: leave it to the optimizer
            return False
        if FLAT_BUFFERS.get(flat.data_ptr()) is not flat:    # a one-off allocation (both rotating buffers aliased by live gradients)
            return False
        return D.early_allreduce(flat, lo, hi, tail_at=flat.numel() - GRAD_TAIL)

    def workspace(self, device):
        need = 0
        for s in self.specs:
            if s.pair:
                need = max(need, T.pairconv_wgrad_workspace_bytes())
            elif s.cin == 1 or s.cout == 1:
                need = max(need, T.image_wgrad_workspace_bytes(max(s.cin, s.cout), s.k))
            else:
                need = max(need, T.wgrad_workspace_bytes(s.cin, s.cout, s.k))
        if isinstance(self, DenseEncoderMixin):
            need = max(need, T.dense_encoder_wgrad_workspace_bytes(), T.dense_encoder_bwd_workspace_bytes())
        if self._ws is None or self._ws.device != device or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
        return self._ws

    def defer_reduces(self, device, layers):
        """queue the fixed-order reduce launches of `layers`' weight gradients (csrc/reduce_defer.hip) until flush_reduces(): their partial sums
        go to an arena of the layers' summed workspaces.  OFF by default ($MMIF_DEFER_REDUCE=1 switches it on; bit-identical either way):
        measured at the headline size, the one launch takes 29 us against 32 us for the five it replaces -- a reduce right after its producer
        reads the partial sums (113 MB for decode.0) out of the last-level cache, the deferred one reads 220 MB from HBM after the other
        backward kernels have evicted them."""
        if not switch("MMIF_DEFER_REDUCE", "0"):
            return False
        need = 0
        for s in layers:
            if s.pair:
                return False
            b = T.image_wgrad_workspace_bytes(max(s.cin, s.cout), s.k) if (s.cin == 1 or s.cout == 1) else T.wgrad_workspace_bytes(s.cin, s.cout, s.k)
            need += (b + 255) // 256 * 256
        arena = getattr(self, "_arena", None)
        if arena is None or arena.device != device or arena.numel() * 4 < need:
            arena = self._arena = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
        T.check(_lib.lib.mmif_reduce_defer_begin(arena.data_ptr(), arena.numel() * 4), "reduce_defer_begin")
        return True

    @staticmethod
    def flush_reduces(keep=False):
        T.check(_lib.lib.mmif_reduce_defer_flush(1 if keep else 0, T.stream_ptr()), "reduce_defer_flush")

    # ---- buffers ----------------------------------------------------------------------------
    def lease(self, key, device):
        lst = self.pool.setdefault(key, [])
        for l in lst:
            if not l.busy:
                l.busy = True
                return l
        l = Lease()
        l.key = key
        l.busy = True
        lst.append(l)
        return l

    @staticmethod
    def buf(lease, name, n, c, h, w, dtype, device, halo=0):
        b = lease.bufs.get(name)
        if b is None:
            # gradient (halo-1) buffers start zeroed: writers either cover the whole padded domain and then fold + zero the
            # halo (mmif_fold_halo) or touch the interior only, so 'folded => halo ring is zero' holds for every reader
            b = BT.alloc(n, c, h, w, dtype, device, halo, zero=halo > 0)
            lease.bufs[name] = b
        return b

    # ---- entry point ------------------------------------------------------------------------
    def run(self, img1, img2=None):
        T.require_device(img1, "img1")
        if img2 is not None:
            T.require_device(img2, "img2")
        ps = self.params()
        need_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in ps) or img1.requires_grad)
        if need_grad:
            leaves = ps
            if FRESH_LEAVES[0] is not None:      # graph capture (mmif/graph.py): fresh leaf identities, their gradients handed over by the caller
                leaves = [p.detach().requires_grad_(p.requires_grad) for p in ps]
                FRESH_LEAVES[0].extend(zip(ps, leaves))
            return _EngineFn.apply(self, img1, img2, *leaves)
        with torch.no_grad():
            out, lease = self.forward(img1, img2)
            lease.busy = False
            return out

    def prepare(self, imgs):
        dtype = compute_dtype()
        impl = conv_impl()
        use_mfma = wants_packed(dtype, impl)
        fmt = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
        stale = []
        for s in self.specs:
            s.refresh(self.weights_version, use_mfma, stale, fmt)
        check_range = bool(stale) and fmt == _lib.F32 and self._x3_range_due()
        if check_range:
            _lib.lib.mmif_x3_pack_saturations(1)     # (count THIS engine's packs only: the counter is process wide)
        T.pack_many(stale)   # all stale operand images of the model in one launch
        if check_range and self._x3_range_check():
            # weights beyond the scaled-fp16 images' range (|w| >= ~63.5) were clamped: the forward switched to three bf16 pieces (fp32's
            # exponent range, six products per tap) -- re-pack in that format before anything runs on the clamped images
            stale = []
            for s in self.specs:
                s.refresh(self.weights_version, use_mfma, stale, fmt)
            T.pack_many(stale)
        imgs = [None if i is None else i.detach().contiguous().float() for i in imgs]
        n, c, h, w = imgs[0].shape
        if c != 1:
            raise ValueError(f"fusion nets take single-channel images [B,1,H,W]; got {tuple(imgs[0].shape)}")
        for i in imgs[1:]:
            if i is not None and i.shape != imgs[0].shape:
                raise ValueError("img1 and img2 must have the same shape")
        return imgs, n, h, w, dtype, impl

    def _x3_range_due(self):
        n = self._x3_packs = getattr(self, "_x3_packs", -1) + 1
        return n % 512 == 0 and _lib.lib.mmif_get_x3_forward_pieces() == 16 and not torch.cuda.is_current_stream_capturing()

    def _x3_range_check(self):
        """fp32 path, scaled-fp16 forward images: ask the library whether the last packs clamped a weight (that used to be
        silent).  The query synchronises, so it runs on the first pack of an engine and every 512th after (never under graph capture);
        on a hit the process switches to the three-piece bf16 forward (set_x3_forward_pieces(3)) with a warning.  True = re-pack now."""
        nsat = _lib.lib.mmif_x3_pack_saturations(1)
        if nsat <= 0:
            return False
        import warnings
        warnings.warn(f"mmif: {nsat} weight value(s) with |w| >= ~63.5 do not fit the scaled-fp16 operand images of the fp32 forward; "
                      "switching the process to three bf16 pieces per operand (mmif.engine.set_x3_forward_pieces(3): same accuracy, twice "
                      "the forward MFMA work)", RuntimeWarning)
        set_x3_forward_pieces(3)
        return True

    # conv helpers operating on ConvSpecs ------------------------------------------------------
    @staticmethod
    def c_fwd(s, x, y, impl):
        T.conv_fwd(x, s.w.detach(), s.b.detach(), y, s.cin, s.cout, s.k, s.relu, s.packed, impl, s.name + ":fwd")

    @staticmethod
    def c_dgrad(s, gy, x, gx, mask_bits, accum_bits, impl):
        """dgrad into the padded-domain view gx + fold of its halo (one call; the DMA-staged kernels fold inside the dgrad);
        returns the folded view.  Every gradient buffer starts zeroed and is folded after each contribution, so gx's halo ring
        is zero on entry as mmif_conv2d_reflect_dgrad_folded requires."""
        return T.conv_dgrad(gy, s.w.detach(), x, gx, s.cin, s.cout, s.k, mask_bits, accum_bits, s.packed, impl, s.name + ":dgrad", fold=True)

    @staticmethod
    def pair_ok(s, dtype, impl, h, w):
        """this layer's backward runs as ONE launch (csrc/conv_mfma.hip bwd_pair_kernel; $MMIF_BWD_PAIR=0: dgrad and wgrad apart)"""
        return (dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and not s.split and s.relu and s.packed is not None and h >= 4 and w >= 4
                and T.bwd_pair_supported(s.cin, s.cout, s.k) and switch("MMIF_BWD_PAIR"))

    @staticmethod
    def wide_ok(s, dtype, impl, x, gx):
        """this layer's backward runs as wgrad (leaving ReLU sign bytes) + dgrad reading them (csrc/conv_mfma.hip bwd_wide;
        $MMIF_BWD_WIDE=0: the two calls apart, the dgrad reading the activations)"""
        if dtype == torch.float32:   # split-operand kernels (csrc/conv_x3.hip): any layer they take; the map is 1/32 of the bytes of x
            return (impl != _lib.IMPL_VALU and x3_enabled() and not s.split and s.packed is not None and s.packed.fmt == _lib.F32
                    and x.halo == 0 and gx.halo == 1 and x.cb == (s.cin + 7) // 8 and gx.cb == x.cb and x.h >= 2 and x.w >= 2
                    and T.bwd_wide_supported(s.cin, s.cout, s.k, dtype) and switch("MMIF_BWD_WIDE"))
        return (dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and not s.split and s.packed is not None and x.h >= 4 and x.w >= 4
                and x.halo == 0 and gx.halo == 1 and x.cb * 8 == s.cin and T.bwd_wide_supported(s.cin, s.cout, s.k)
                and (x.h + 2) * (x.w + 2) * 128 < (1 << 31) and switch("MMIF_BWD_WIDE"))

    def c_bwd_wide(self, s, gy, x, gx, mask_bits, ws):
        need = T.bwd_wide_signs_bytes(x.n, s.cin, x.h, x.w)
        sg = getattr(self, "_signs", None)
        if sg is None or sg.numel() < need or sg.device != x.buf.device:
            sg = self._signs = torch.empty(need, dtype=torch.uint8, device=x.buf.device)
        return T.conv_bwd_wide(gy, x, gx, s.dw, s.db, s.cin, s.cout, s.k, s.packed, mask_bits, ws, sg, False, s.name + ":bwd")

    @staticmethod
    def c_bwd_pair(s, gy, x, gx, ws):
        return T.conv_bwd_pair(gy, x, gx, s.dw, s.db, s.cin, s.cout, s.k, s.packed, ws, False, s.name + ":bwd")

    @staticmethod
    def tag_dgrad(gy, x, gx, cin, cout, packed, tag):
        """dgrad of a virtual layer (operand image only, no fp32 weights): accumulate onto gx + ReLU mask of x, folded"""
        impl = _lib.IMPL_MFMA if packed.fmt == _lib.BF16 else _lib.IMPL_X3
        return T.conv_dgrad(gy, None, x, gx, cin, cout, 3, all_bits(gx.cb), all_bits(gx.cb), packed, impl, tag, fold=True)

    @staticmethod
    def c_wgrad(s, x, gy, ws, impl, accumulate=False):
        if s.split:
            T.conv_wgrad(x, gy, s.grad_target(), s.db, s.cin, s.cout, s.k, ws, False, impl, s.name + ":wgrad")
            # db went straight to the parameter's view (accumulate unsupported for permuted layers: used once per step)
            s.finish_grad(accumulate)
        else:
            T.conv_wgrad(x, gy, s.dw, s.db, s.cin, s.cout, s.k, ws, accumulate, impl, s.name + ":wgrad")

    @staticmethod
    def image_layer_bwd(last, x, gout, yout, g, ws):
        """backward of the decoder's last layer (Cout = 1): dW, db and the folded, masked input gradient in g.  bf16, 16 channels, 3x3: ONE
        launch (csrc/image_bwd.hip, round 6; $MMIF_IMAGE_BWD=0: weight gradient, input gradient and fold as three)"""
        if switch("MMIF_IMAGE_BWD") and T.image_out_bwd_supported(x, last.cin, last.k) and g.cb == 2:
            return T.image_out_bwd(x, gout, yout, last.w.detach(), g, last.dw, last.db, last.cin, last.k, ws)
        T.image_out_wgrad(x, gout, yout, last.dw, last.db, last.cin, last.k, ws)
        T.image_out_dgrad(gout, yout, last.w.detach(), x, g, last.cin, last.k, all_bits(g.cb), 0)
        return g.fold_halo_() if last.k > 1 else g.as_folded()

    def forward(self, img1, img2):
        raise NotImplementedError

    def backward(self, lease, gout):
        raise NotImplementedError


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, img1, img2, *params):
        out, lease = engine.forward(img1, img2)
        ctx.engine, ctx.lease = engine, lease
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, gout):
        lease = ctx.lease
        if lease is None:
            raise RuntimeError("mmif: backward called twice on the same forward (buffers were recycled)")
        D.begin_backward()
        grads = ctx.engine.backward(lease, gout.contiguous().float())
        lease.busy = False
        ctx.lease = None
        return (None, None, None) + tuple(grads)


# =============================================================================================
class DenseEncoderMixin:
    """Conv(1->16) + DenseBlock(16,16) into an 8-block (64 ch) slice of a feature buffer
    (reference core/model.py:73-80 + core/block.py:137-151)."""

    @staticmethod
    def enc_fwd(specs, img, F, base, impl):
        first, c0, c1, c2 = specs
        T.image_in_fwd(img, first.w.detach(), first.b.detach(), F.view(base, 2), first.cout, first.k, first.relu)
        ModelEngine.c_fwd(c0, F.view(base, 2), F.view(base + 2, 2), impl)
        ModelEngine.c_fwd(c1, F.view(base, 4), F.view(base + 4, 2), impl)
        ModelEngine.c_fwd(c2, F.view(base, 6), F.view(base + 6, 2), impl)

    @staticmethod
    def enc_fwd_all(branches, F, dtype, impl):
        """branches: [(specs, img, first channel block in F), ...] (one or two).  bf16 / MFMA: ONE streaming launch for all four
        layers of all branches (csrc/enc_stream.hip; bit-identical to the layer-wise launches, $MMIF_ENC_STREAM=0 selects those)."""
        stream = (dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and switch("MMIF_ENC_STREAM")
                  and all(s.relu and s.k == 3 for specs, _, _ in branches for s in specs)
                  and all([(s.cin, s.cout) for s in specs] == [(1, 16), (16, 16), (32, 16), (48, 16)] for specs, _, _ in branches))
        if not stream:
            for specs, img, base in branches:
                DenseEncoderMixin.enc_fwd(specs, img, F, base, impl)
            return
        T.dense_encoder_fwd([(img, specs[0].w.detach(), specs[0].b.detach(), [s.packed for s in specs[1:]],
                              [s.b.detach() for s in specs[1:]], F.view(base, 8)) for specs, img, base in branches], tag="encode:fwd")

    @staticmethod
    def chain_images(specs, fmt=_lib.BF16):
        """dgrad operand images of the DenseBlock's virtual gather layers, re-packed when the weights changed (one launch)"""
        first, c0, c1, c2 = specs
        key = (WEIGHTS_EPOCH[0], fmt) + tuple((s.conv.weight._version, s.conv.weight.data_ptr()) for s in (c0, c1, c2))
        cached = getattr(first, "_chain", None)
        ws = [s.conv.weight.detach() for s in (c0, c1, c2)]
        if cached is None or cached[1][0].dgrad.device != ws[0].device or cached[1][0].fmt != fmt:
            cached = (key, T.pack_dense_chain(*ws, ws[0].device, fmt))
            first._chain = cached
        elif cached[0] != key:
            T.repack_dense_chain(cached[1], *ws)
            cached = (key, cached[1])
            first._chain = cached
        return cached[1]

    @staticmethod
    def chain_out(first, F, slot=0):
        """[g0 | g1 | g2 | g3] of one encoder branch: the streaming chain's output (it is not an in-place kernel), read by the branch's
        weight-gradient pass afterwards -- one 8-block view per branch slot of a buffer kept with the encoder's first layer"""
        # one 64-channel buffer per (shape, branch slot), never freed while the spec lives: a captured GraphedStep has the address baked
        # into its launches, and an eager step of another shape in between must not recycle it
        key = (F.n, F.h, F.w, F.buf.device, slot)
        pool = first.__dict__.setdefault("_gz", {})
        gz = pool.pop(key, None)
        if gz is None:
            gz = BT.alloc(F.n, 64, F.h, F.w, torch.bfloat16, F.buf.device)
            # bounded (variable-size crops grew this without limit): the least recently used shapes go once more than 8 are held --
            # except those a live GraphedStep captured (mmif/graph.py pins the buffers of its capture in GRAPH_PINNED)
            while len(pool) >= 8:
                victim = next((k for k in pool if id(pool[k].buf) not in GRAPH_PINNED), None)
                if victim is None:
                    break
                del pool[victim]
        pool[key] = gz          # (re-inserted last = most recently used)
        if torch.cuda.is_current_stream_capturing():
            GRAPH_PINNED.add(id(gz.buf))
        return gz

    @staticmethod
    def chain_streams(specs, F, impl):
        """this encoder's backward chain runs as the streaming launch (csrc/enc_chain.hip)"""
        return (F.dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and all(s.k == 3 for s in specs)
                and [(s.cin, s.cout) for s in specs] == [(1, 16), (16, 16), (32, 16), (48, 16)]
                and F.h >= 4 and F.w >= 4 and switch("MMIF_ENC_CHAIN") and switch("MMIF_ENC_CHAIN_STREAM"))

    @staticmethod
    def bwd_fused(branches, F, ws, impl):
        """branches: [(specs, img, fbase, g3 2-block view, glow 6-block view, accumulate), (...)] -- the WHOLE backward of the encoder branches
        (gradient chain + the four layers' weight gradients) as ONE streaming launch (csrc/enc_bwd.hip, round 5; $MMIF_ENC_BWD_FUSED=0: the chain and
        weight-gradient launches of rounds 2 / 4 it replaces -- 0.40 ms against 0.64 ms at the headline size, profiles/r05_*).  Returns False when the kernel does not apply (the caller takes the other path)."""
        if not (switch("MMIF_ENC_BWD_FUSED") and switch("MMIF_ENC_CHAIN") and switch("MMIF_ENC_CHAIN_STREAM") and switch("MMIF_ENC_WGRAD")
                and F.dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and F.h >= 4 and F.w >= 4 and F.halo == 0
                and all(_lib.lib.mmif_dense_encoder_bwd_fits(t.cb_total, t.h, t.w, t.halo) for br in branches for t in (F, br[3], br[4]))
                and all(all(s.k == 3 for s in specs) and [(s.cin, s.cout) for s in specs] == [(1, 16), (16, 16), (32, 16), (48, 16)]
                        for specs, *_ in branches)):
            return False
        args = []
        if len(branches) == 2:
            DenseEncoderMixin.chain_images_pair(branches[0][0], branches[1][0])
        for specs, img, fbase, g3, glow, acc in branches:
            pk = DenseEncoderMixin.chain_images(specs, _lib.BF16)
            args.append((g3, glow, F.view(fbase, 6), pk, img, [(s.dw, s.db) for s in specs], acc))
        T.dense_encoder_bwd(args, ws, tag="encode:bwd")
        return True

    @staticmethod
    def chain_images_pair(specs_a, specs_b):
        """two encoder branches with their own weights (PFNetv1): when both sets of bf16 chain images exist and are stale, re-pack them in ONE
        launch; chain_images() then finds them current.  Anything else (first use, shared weights, one side current) is left to chain_images()."""
        ca, cb = getattr(specs_a[0], "_chain", None), getattr(specs_b[0], "_chain", None)
        if ca is None or cb is None or specs_a[0] is specs_b[0]:
            return
        keys, wss = [], []
        for specs in (specs_a, specs_b):
            c = specs[1:]
            keys.append((WEIGHTS_EPOCH[0], _lib.BF16) + tuple((s.conv.weight._version, s.conv.weight.data_ptr()) for s in c))
            wss.append([s.conv.weight.detach() for s in c])
        if ca[0] == keys[0] or cb[0] == keys[1] or ca[1][0].fmt != _lib.BF16 or cb[1][0].fmt != _lib.BF16 \
                or ca[1][0].dgrad.device != wss[0][0].device or cb[1][0].dgrad.device != wss[1][0].device:
            return
        T.repack_dense_chain_pair(ca[1], wss[0], cb[1], wss[1])
        specs_a[0]._chain = (keys[0], ca[1])
        specs_b[0]._chain = (keys[1], cb[1])

    @staticmethod
    def chain_all(branches, F, GF, impl):
        """branches: [(specs, fbase, gbase, onto), (...)] -- the backward chains of BOTH encoder branches as ONE streaming launch; returns
        their [g0 | g1 | g2 | g3] views for enc_bwd(gz=...), or None per branch when the streaming kernel does not apply"""
        if len(branches) != 2 or not all(DenseEncoderMixin.chain_streams(sp, F, impl) for sp, _, _, _ in branches):
            return [None] * len(branches)
        GFf = GF.as_folded()
        args, outs = [], []
        for slot, (specs, fbase, gbase, onto) in enumerate(branches):
            pk = DenseEncoderMixin.chain_images(specs, _lib.BF16)
            gz = DenseEncoderMixin.chain_out(specs[0], F, slot)
            glow = onto.view(0, 6) if onto is not None else GFf.view(gbase, 6)
            args.append((GFf.view(gbase + 6, 2), glow, F.view(fbase, 6), pk, gz))
            outs.append(gz)
        T.dense_encoder_chain(args, tag="encode.chain:dgrad")
        return outs

    @staticmethod
    def enc_bwd(specs, img, F, GF, fbase, gbase, ws, impl, accumulate_w=False, onto=None, gz=None):
        """GF[gbase:gbase+8] holds dL/d(encoder output) (padded domain), top 2 blocks already masked.
        onto (8-block folded view): blocks 0..5 of GF are NOT initialised -- the chain adds its contributions to onto's blocks
        instead and writes the sums to GF (DenseFuse / VIFNet: both branches start from the one gradient of f1 + f2)."""
        first, c0, c1, c2 = specs
        GF = GF.as_folded()   # every contribution so far has been folded; each dgrad below re-folds what it adds
        # bf16 / MFMA: the four layers' weight gradients in ONE pass over [x0 | x1 | x2] and the finished [g0 | g1 | g2 | g3]
        # (csrc/enc_wgrad.hip; $MMIF_ENC_WGRAD=0 selects the layer-wise kernels)
        hot = (F.dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and all(s.k == 3 for s in specs)
               and [(s.cin, s.cout) for s in specs] == [(1, 16), (16, 16), (32, 16), (48, 16)])
        # fp32 on the split-operand kernels: the same fused weight-gradient call (wgrad_x3_dense_kernel); the dgrad chain stays per layer
        hot32 = (F.dtype == torch.float32 and impl != _lib.IMPL_VALU and x3_enabled() and all(s.k == 3 for s in specs)
                 and [(s.cin, s.cout) for s in specs] == [(1, 16), (16, 16), (32, 16), (48, 16)] and F.halo == 0)
        fused = (hot or hot32) and switch("MMIF_ENC_WGRAD")
        # ... and the dgrad chain per DESTINATION (gather form: one launch per x_k on the stacked virtual layer, fp32 sum of all
        # contributions, one rounding) instead of per source layer (read-modify-write of the lower blocks); $MMIF_ENC_CHAIN=0: scatter
        gather = (hot or (hot32 and onto is None)) and switch("MMIF_ENC_CHAIN")
        # bf16: the three destinations as ONE streaming launch (csrc/enc_chain.hip: line-buffer pipeline, reflect adjoint in place) that
        # leaves [g0 | g1 | g2 | g3] in a buffer of its own; $MMIF_ENC_CHAIN_STREAM=0: one gather-form dgrad launch per destination, in place
        chain_done = gz is not None          # (chain_all ran this branch's chain already, in one launch with the other branch's)
        if chain_done:
            pass
        elif gather and hot and F.h >= 4 and F.w >= 4 and switch("MMIF_ENC_CHAIN_STREAM"):
            pk = DenseEncoderMixin.chain_images(specs, _lib.BF16)
            gz = DenseEncoderMixin.chain_out(first, F)
            glow = onto.view(0, 6) if onto is not None else GF.view(gbase, 6)
            T.dense_encoder_chain([(GF.view(gbase + 6, 2), glow, F.view(fbase, 6), pk, gz)], tag=f"{first.name}.chain:dgrad")
        elif gather:
            gz = GF.view(gbase, 8)
            pk = DenseEncoderMixin.chain_images(specs, _lib.F32 if hot32 else _lib.BF16)
            for k in (2, 1, 0):
                gy, xk, dst = GF.view(gbase + 2 * (k + 1), 2 * (3 - k)), F.view(fbase + 2 * k, 2), GF.view(gbase + 2 * k, 2)
                if onto is not None:
                    T.conv_dgrad_onto(gy, xk, onto.view(2 * k, 2), dst, 16, 16 * (3 - k), 3, all_bits(2), all_bits(2), pk[k],
                                      f"{first.name}.chain{k}:dgrad")
                else:
                    ModelEngine.tag_dgrad(gy, xk, dst, 16, 16 * (3 - k), pk[k], f"{first.name}.chain{k}:dgrad")
        if gz is None:
            gz = GF.view(gbase, 8)
        scatter = not (gather or chain_done)
        for s, nin in ((c2, 6), (c1, 4), (c0, 2)):
            g = gz.view(nin, 2)
            x = F.view(fbase, nin)
            if not fused:
                T.conv_wgrad(x, g, s.dw, s.db, s.cin, s.cout, s.k, ws, accumulate_w, impl, s.name + ":wgrad")
            # accumulate into the lower blocks; this conv is the LAST contributor of its top 2 input blocks
            if scatter:
                ModelEngine.c_dgrad(s, g, x, GF.view(gbase, nin), bits(nin - 2, nin - 1), all_bits(nin), impl)
        if fused:
            T.dense_encoder_wgrad(img, F.view(fbase, 6), gz, [(s.dw, s.db) for s in specs], ws, accumulate_w, tag="encode:wgrad")
        else:
            T.image_in_wgrad(img, gz.view(0, 2), first.dw, first.db, first.cout, first.k, ws, accumulate_w)


class PFNetv1Engine(ModelEngine, DenseEncoderMixin):
    """reference core/model.py:69-111."""

    def __init__(self, module):
        m = module
        def cs(name, layer):
            return ConvSpec(name, layer.layers[0], layer.act is not None)
        self.enc = []
        for e, seq in enumerate((m.encode1, m.encode2)):
            self.enc.append([cs(f"encode{e + 1}.0", seq[0])] + [cs(f"encode{e + 1}.1.{i}", l) for i, l in enumerate(seq[1].layers)])
        self.dec = [cs(f"decode.{i}", l) for i, l in enumerate(m.decode)]
        super().__init__(module, self.enc[0] + self.enc[1] + self.dec)

    def forward(self, img1, img2):
        (img1, img2), n, h, w, dtype, impl = self.prepare((img1, img2))
        dev = img1.device
        L = self.lease((n, h, w, dtype), dev)
        L.imgs = (img1, img2)
        F = self.buf(L, "F", n, 128, h, w, dtype, dev)
        self.enc_fwd_all([(self.enc[0], img1, 0), (self.enc[1], img2, 8)], F, dtype, impl)
        x = F
        for i, s in enumerate(self.dec[:-1]):
            y = self.buf(L, f"D{i}", n, s.cout, h, w, dtype, dev)
            self.c_fwd(s, x, y, impl)
            x = y
        last = self.dec[-1]
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
        T.image_out_fwd(x, last.w.detach(), last.b.detach(), out, last.cin, last.k, last.relu)
        L.out = out.detach() if last.relu else None     # (an alias: the returned object gets a grad_fn, which must not be kept alive by the lease)
        return out, L

    def backward(self, L, gout):
        n, h, w, dtype = L.key
        dev = gout.device
        impl = conv_impl()
        flat, grads = self._assign_grad_views(dev)
        ws = self.workspace(dev)
        img1, img2 = L.imgs
        F = L.bufs["F"]
        acts = [F] + [L.bufs[f"D{i}"] for i in range(len(self.dec) - 1)]
        last = self.dec[-1]
        x = acts[-1]
        # the decoder's weight-gradient reduces as ONE launch at the end of the decoder's backward (not with the side-stream experiment: its
        # producer runs on another stream)
        deferred = dtype == torch.bfloat16 and self.defer_reduces(dev, self.dec)
        try:      # (a failing launch must not leave the library's deferred-reduce queue open: ADVICE r5)
            g = self.buf(L, f"G{len(acts) - 1}", n, last.cin, h, w, dtype, dev, halo=1)
            g = self.image_layer_bwd(last, x, gout, L.out, g, ws)
            for i in range(len(self.dec) - 2, -1, -1):
                s, x = self.dec[i], acts[i]
                gx = self.buf(L, f"G{i}", n, s.cin, h, w, dtype, dev, halo=1)
                if i > 0 and self.pair_ok(s, dtype, impl, h, w):
                    g = self.c_bwd_pair(s, g, x, gx, ws)      # thin layer: dgrad (every block masked) + wgrad in one launch
                    continue
                # gradient w.r.t. the concatenated encoder features (i == 0): only each encoder's last DenseBlock output
                # (blocks 6,7 / 14,15) has no further contributor
                mb = all_bits(gx.cb) if i > 0 else bits(6, 7, 14, 15)
                if self.wide_ok(s, dtype, impl, x, gx):
                    g = self.c_bwd_wide(s, g, x, gx, mb, ws)  # wide layer: the wgrad leaves the ReLU sign bytes the dgrad masks with
                    continue
                self.c_wgrad(s, x, g, ws, impl)
                g = self.c_dgrad(s, g, x, gx, mb, 0, impl)
        finally:
            if deferred:
                self.flush_reduces()
        self.early_reduce(flat, self.dec)      # (data parallel) the decoder's gradients leave while the encoder's backward runs
        self.enc_bwd_all(img1, img2, F, g, ws, impl)
        return grads

    def enc_bwd_all(self, img1, img2, F, g, ws, impl):
        gf = g.as_folded()
        if self.bwd_fused([(self.enc[0], img1, 0, gf.view(6, 2), gf.view(0, 6), False), (self.enc[1], img2, 8, gf.view(14, 2), gf.view(8, 6), False)], F, ws, impl):
            return
        gz = self.chain_all([(self.enc[0], 0, 0, None), (self.enc[1], 8, 8, None)], F, g, impl)
        self.enc_bwd(self.enc[0], img1, F, g, 0, 0, ws, impl, gz=gz[0])
        self.enc_bwd(self.enc[1], img2, F, g, 8, 8, ws, impl, gz=gz[1])


class VIFNetEngine(PFNetv1Engine):
    """reference core/model.py:189-206: PFNetv1's graph with ONE encoder applied to both images (its weight gradients are
    the sum of the two branches: the second branch accumulates)."""

    def __init__(self, module):
        m = module
        def cs(name, layer):
            return ConvSpec(name, layer.layers[0], layer.act is not None)
        shared = [cs("encode.0", m.encode[0])] + [cs(f"encode.1.{i}", l) for i, l in enumerate(m.encode[1].layers)]
        self.enc = [shared, shared]
        self.dec = [cs(f"decode.{i}", l) for i, l in enumerate(m.decode)]
        ModelEngine.__init__(self, module, shared + self.dec)

    def enc_bwd_all(self, img1, img2, F, g, ws, impl):
        gf = g.as_folded()
        if self.bwd_fused([(self.enc[0], img1, 0, gf.view(6, 2), gf.view(0, 6), False), (self.enc[1], img2, 8, gf.view(14, 2), gf.view(8, 6), True)], F, ws, impl):
            return
        gz = self.chain_all([(self.enc[0], 0, 0, None), (self.enc[1], 8, 8, None)], F, g, impl)
        self.enc_bwd(self.enc[0], img1, F, g, 0, 0, ws, impl, accumulate_w=False, gz=gz[0])
        self.enc_bwd(self.enc[1], img2, F, g, 8, 8, ws, impl, accumulate_w=True, gz=gz[1])


class DenseFuseEngine(ModelEngine, DenseEncoderMixin):
    """reference core/model.py:165-186 (+ _FusionModel :27-63): shared encoder run on both images,
    element 'sum' fusion, 4-conv decoder.  Also the auto-encoder mode forward(img1)."""

    fusion_mode = _lib.FUSE_SUM

    def __init__(self, module):
        m = module
        def cs(name, layer):
            return ConvSpec(name, layer.layers[0], layer.act is not None)
        self.enc = [cs("encode.0", m.encode[0])] + [cs(f"encode.1.{i}", l) for i, l in enumerate(m.encode[1].layers)]
        self.dec = [cs(f"decode.{i}", l) for i, l in enumerate(m.decode)]
        super().__init__(module, self.enc + self.dec)

    def forward(self, img1, img2):
        (img1, img2), n, h, w, dtype, impl = self.prepare((img1, img2))
        dev = img1.device
        single = img2 is None
        L = self.lease((n, h, w, dtype, single), dev)
        L.imgs = (img1, img2)
        F = self.buf(L, "F", n, 64 if single else 128, h, w, dtype, dev)
        if single:
            self.enc_fwd_all([(self.enc, img1, 0)], F, dtype, impl)
            x = F
        else:
            x = self.buf(L, "S", n, 64, h, w, dtype, dev)
            if not self.enc_fwd_sum(img1, img2, F, x, dtype, impl):
                self.enc_fwd_all([(self.enc, img1, 0), (self.enc, img2, 8)], F, dtype, impl)
                self.fusion_fwd(L, F, x)
        for i, s in enumerate(self.dec[:-1]):
            y = self.buf(L, f"D{i}", n, s.cout, h, w, dtype, dev)
            self.c_fwd(s, x, y, impl)
            x = y
        last = self.dec[-1]
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
        T.image_out_fwd(x, last.w.detach(), last.b.detach(), out, last.cin, last.k, last.relu)
        L.out = out.detach() if last.relu else None     # (an alias: the returned object gets a grad_fn, which must not be kept alive by the lease)
        return out, L

    def backward(self, L, gout):
        n, h, w, dtype, single = L.key
        dev = gout.device
        impl = conv_impl()
        flat, grads = self._assign_grad_views(dev)
        ws = self.workspace(dev)
        img1, img2 = L.imgs
        F = L.bufs["F"]
        first_in = F if single else L.bufs["S"]
        acts = [first_in] + [L.bufs[f"D{i}"] for i in range(len(self.dec) - 1)]
        last = self.dec[-1]
        x = acts[-1]
        # the decoder's weight-gradient reduces as ONE launch at the end of the decoder's backward (not with the side-stream experiment: its
        # producer runs on another stream)
        deferred = dtype == torch.bfloat16 and self.defer_reduces(dev, self.dec)
        try:
            g = self.buf(L, f"G{len(acts) - 1}", n, last.cin, h, w, dtype, dev, halo=1)
            g = self.image_layer_bwd(last, x, gout, L.out, g, ws)
            for i in range(len(self.dec) - 2, -1, -1):
                s, x = self.dec[i], acts[i]
                gx = self.buf(L, f"G{i}", n, s.cin, h, w, dtype, dev, halo=1)
                if i > 0 and self.pair_ok(s, dtype, impl, h, w):
                    g = self.c_bwd_pair(s, g, x, gx, ws)
                    continue
                # i == 0: the auto-encoder's input is the DenseBlock output (only its last conv, blocks 6,7, has no further contributor);
                # x = f1 + f2 is not a ReLU output
                mb = all_bits(gx.cb) if i > 0 else (bits(6, 7) if single else 0)
                if i == 0 and not single and self.dup_ok(s, g, gx, L, dtype, impl):
                    # 'sum' fusion: decode.0's dgrad also leaves the copies of its blocks 6, 7 that each encoder branch masks with its own x3
                    # (csrc/conv_mfma.hip struct DupOut, round 6: mmif_fuse_elem_bwd's launch disappears); the weight gradient apart
                    GF = self.buf(L, "GF", n, 128, h, w, dtype, dev, halo=1)
                    self.c_wgrad(s, x, g, ws, impl)
                    g = T.conv_dgrad_dup(g, gx, s.cin, s.cout, s.k, s.packed, GF, F, 3, s.name + ":dgrad")
                    L.dup_done = True
                    continue
                if self.wide_ok(s, dtype, impl, x, gx):
                    g = self.c_bwd_wide(s, g, x, gx, mb, ws)
                    continue
                self.c_wgrad(s, x, g, ws, impl)
                g = self.c_dgrad(s, g, x, gx, mb, 0, impl)
        finally:
            if deferred:
                self.flush_reduces()
        self.early_reduce(flat, self.dec)
        return self._encoder_backward(L, F, g, ws, impl, dtype, dev, n, h, w, single, img1, img2, grads)

    def _encoder_backward(self, L, F, g, ws, impl, dtype, dev, n, h, w, single, img1, img2, grads):
        if single:
            gf = g.as_folded()
            if not self.bwd_fused([(self.enc, img1, 0, gf.view(6, 2), gf.view(0, 6), False)], F, ws, impl):
                self.enc_bwd(self.enc, img1, F, g, 0, 0, ws, impl)
            return grads
        # fusion backward: d(f1+f2) -> each encoder's own gradient buffer (they diverge below);
        # ReLU mask only on the DenseBlock's last conv output (blocks 6,7), the rest are masked by
        # their last contributor inside enc_bwd
        GF = self.buf(L, "GF", n, 128, h, w, dtype, dev, halo=1)
        if self.share_fused_grad(g, GF, dtype, impl):
            # 'sum' fusion: d(f1 + f2) IS each branch's gradient.  Only the masked top blocks (6,7 / 14,15) are materialised per branch;
            # the chain reads the lower blocks' starting values straight from g ($MMIF_FUSE_SHARE=0: copy them per branch first)
            if not getattr(L, "dup_done", False):     # (decode.0's dgrad left them already: dup_ok)
                T.fuse_elem_bwd(F.view(6, 2), F.view(14, 2), g.view(6, 2), GF.view(6, 2), GF.view(14, 2), self.fusion_mode, True)
            L.dup_done = False
            gff = GF.as_folded()
            if self.bwd_fused([(self.enc, img1, 0, gff.view(6, 2), g.view(0, 6), False), (self.enc, img2, 8, gff.view(14, 2), g.view(0, 6), True)], F, ws, impl):
                return grads
            gz = self.chain_all([(self.enc, 0, 0, g), (self.enc, 8, 8, g)], F, GF, impl)
            self.enc_bwd(self.enc, img1, F, GF, 0, 0, ws, impl, accumulate_w=False, onto=g, gz=gz[0])
            self.enc_bwd(self.enc, img2, F, GF, 8, 8, ws, impl, accumulate_w=True, onto=g, gz=gz[1])
            return grads
        self.fusion_bwd(L, F, g, GF, ws)
        gff = GF.as_folded()
        if self.bwd_fused([(self.enc, img1, 0, gff.view(6, 2), gff.view(0, 6), False), (self.enc, img2, 8, gff.view(14, 2), gff.view(8, 6), True)], F, ws, impl):
            return grads
        gz = self.chain_all([(self.enc, 0, 0, None), (self.enc, 8, 8, None)], F, GF, impl)
        self.enc_bwd(self.enc, img1, F, GF, 0, 0, ws, impl, accumulate_w=False, gz=gz[0])
        self.enc_bwd(self.enc, img2, F, GF, 8, 8, ws, impl, accumulate_w=True, gz=gz[1])
        return grads

    def enc_fwd_sum(self, img1, img2, F, S, dtype, impl):
        """bf16, 'sum' fusion: the shared encoder on both images AND f1 + f2 as ONE launch (csrc/enc_stream2.hip's dual form, round 6;
        $MMIF_ENC_SUM=0: the two-branch encoder launch + mmif_fuse_elem_fwd).  False = not taken."""
        specs = self.enc
        if not (dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and switch("MMIF_ENC_STREAM") and switch("MMIF_ENC_SUM")
                and type(self).fusion_fwd is DenseFuseEngine.fusion_fwd and self.fusion_mode == _lib.FUSE_SUM
                and all(s.relu and s.k == 3 for s in specs) and [(s.cin, s.cout) for s in specs] == [(1, 16), (16, 16), (32, 16), (48, 16)]):
            return False
        br = [(img, specs[0].w.detach(), specs[0].b.detach(), [s.packed for s in specs[1:]], [s.b.detach() for s in specs[1:]], F.view(base, 8))
              for img, base in ((img1, 0), (img2, 8))]
        return T.dense_encoder_fwd_sum(br, S, tag="encode:fwd")

    def dup_ok(self, s, gy, gx, L, dtype, impl):
        """decode.0's input gradient also writes the two masked copies of its top blocks ($MMIF_DGRAD_DUP=0: mmif_fuse_elem_bwd does)"""
        if not (switch("MMIF_DGRAD_DUP") and dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and s.packed is not None and not s.split
                and (s.cin, s.k) == (64, 3) and T.conv_dgrad_dup_supported(gy, gx, s.cin, s.cout, s.k)):
            return False
        n, h, w = gx.n, gx.h, gx.w
        GF = self.buf(L, "GF", n, 128, h, w, dtype, gx.buf.device, halo=1)
        return self.share_fused_grad(gx, GF, dtype, impl)

    def share_fused_grad(self, g, GF, dtype, impl):
        specs = self.enc
        hot = (dtype == torch.bfloat16 and impl != _lib.IMPL_VALU and all(s.k == 3 for s in specs)
               and [(s.cin, s.cout) for s in specs] == [(1, 16), (16, 16), (32, 16), (48, 16)])
        return (hot and type(self).fusion_bwd is DenseFuseEngine.fusion_bwd and self.fusion_mode == _lib.FUSE_SUM
                and switch("MMIF_ENC_CHAIN") and switch("MMIF_FUSE_SHARE")
                and all(T.dgrad_onto_supported(GF.as_folded().view(2 * (k + 1), 2 * (3 - k)), GF.view(2 * k, 2), 16, 16 * (3 - k), 3)
                        for k in (2, 1, 0)))

    def fusion_fwd(self, L, F, out):
        T.fuse_elem_fwd(F.view(0, 8), F.view(8, 8), out, self.fusion_mode)

    def fusion_bwd(self, L, F, g, GF, ws):
        # blocks 0..5: plain copy scaled by the fusion derivative; blocks 6,7: also ReLU-masked
        T.fuse_elem_bwd(F.view(0, 6), F.view(8, 6), g.view(0, 6), GF.view(0, 6), GF.view(8, 6), self.fusion_mode, False)
        T.fuse_elem_bwd(F.view(6, 2), F.view(14, 2), g.view(6, 2), GF.view(6, 2), GF.view(14, 2), self.fusion_mode, True)


class PFNetv2Engine(DenseFuseEngine):
    """reference core/model.py:114-141: DenseFuse's encoder / decoder around the self-learned fusion -- the shared
    conv stack fuse = (2->2, 2->2, 2->1) applied to every channel pair (feat1[:, i], feat2[:, i]) plus feat1 + feat2.
    Each fuse layer is ONE pair-conv launch over the blocked feature buffers (csrc/pair.hip) instead of the
    reference's 64-iteration Python loop; the intermediates H1, H2 are 2 x 64-channel buffers [a-half | b-half]."""

    def __init__(self, module):
        super().__init__(module)
        self.fuse = [ConvSpec(f"fuse.{i}", l.layers[0], l.act is not None) for i, l in enumerate(module.fuse)]
        for s in self.fuse:
            assert (s.cin, s.k) == (2, 3) and s.cout in (1, 2), "PFNetv2.fuse layers are 3x3 convs on channel pairs"
            s.pair = True
        self.specs = self.enc + self.fuse + self.dec

    def fusion_fwd(self, L, F, out):
        n, h, w, dtype = L.key[:4]
        dev = F.buf.device
        f0, f1, f2 = self.fuse
        H1 = self.buf(L, "H1", n, 128, h, w, dtype, dev)
        H2 = self.buf(L, "H2", n, 128, h, w, dtype, dev)
        a, b = F.view(0, 8), F.view(8, 8)
        T.pairconv_fwd(a, b, f0.w.detach(), f0.b.detach(), 2, H1.view(0, 8), H1.view(8, 8), f0.relu)
        T.pairconv_fwd(H1.view(0, 8), H1.view(8, 8), f1.w.detach(), f1.b.detach(), 2, H2.view(0, 8), H2.view(8, 8), f1.relu)
        T.pairconv_fwd(H2.view(0, 8), H2.view(8, 8), f2.w.detach(), f2.b.detach(), 1, out, None, f2.relu, a, b)

    def fusion_bwd(self, L, F, g, GF, ws):
        n, h, w, dtype = L.key[:4]
        dev = F.buf.device
        f0, f1, f2 = self.fuse
        H1, H2 = L.bufs["H1"], L.bufs["H2"]
        GH1 = self.buf(L, "GH1", n, 128, h, w, dtype, dev, halo=1)
        GH2 = self.buf(L, "GH2", n, 128, h, w, dtype, dev, halo=1)
        a, b = F.view(0, 8), F.view(8, 8)
        h1a, h1b, h2a, h2b = H1.view(0, 8), H1.view(8, 8), H2.view(0, 8), H2.view(8, 8)
        if switch("MMIF_PAIR_BWD"):   # dgrad + wgrad of each layer in one pass (csrc/pair.hip pairconv_bwd_kernel)
            T.pairconv_bwd(g, None, f2.w.detach(), 1, h2a, h2b, GH2.view(0, 8), GH2.view(8, 8), f2.dw, f2.db, ws, all_bits(8))
            g2 = GH2.fold_halo_()
            T.pairconv_bwd(g2.view(0, 8), g2.view(8, 8), f1.w.detach(), 2, h1a, h1b, GH1.view(0, 8), GH1.view(8, 8), f1.dw, f1.db, ws, all_bits(8))
            g1 = GH1.fold_halo_()
            T.pairconv_bwd(g1.view(0, 8), g1.view(8, 8), f0.w.detach(), 2, a, b, GF.view(0, 8), GF.view(8, 8), f0.dw, f0.db, ws, bits(6, 7), add=g)
            GF.fold_halo_()
            return
        # fuse.2 (2 -> 1, no activation): g is dL/d(fused features), not masked
        T.pairconv_wgrad(h2a, h2b, g, None, 1, f2.dw, f2.db, ws)
        T.pairconv_dgrad(g, None, f2.w.detach(), 1, h2a, h2b, GH2.view(0, 8), GH2.view(8, 8), all_bits(8))
        g2 = GH2.fold_halo_()
        T.pairconv_wgrad(h1a, h1b, g2.view(0, 8), g2.view(8, 8), 2, f1.dw, f1.db, ws)
        T.pairconv_dgrad(g2.view(0, 8), g2.view(8, 8), f1.w.detach(), 2, h1a, h1b, GH1.view(0, 8), GH1.view(8, 8), all_bits(8))
        g1 = GH1.fold_halo_()
        T.pairconv_wgrad(a, b, g1.view(0, 8), g1.view(8, 8), 2, f0.dw, f0.db, ws)
        # + g: the residual "+ feat1 + feat2"; ReLU mask only where the gradient is complete (DenseBlock's last conv output)
        T.pairconv_dgrad(g1.view(0, 8), g1.view(8, 8), f0.w.detach(), 2, a, b, GF.view(0, 8), GF.view(8, 8), bits(6, 7), add=g)
        GF.fold_halo_()
