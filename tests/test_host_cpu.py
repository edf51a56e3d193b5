"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/mmif.h
declares, the Python mirror reproduces the reference's module tree / state_dict layout and error
behaviour, and the training plumbing (meters, warm-up, flags) behaves like the reference's common.py."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    import mmif
    from mmif._lib import LIB_PATH, SIGNATURES
    hdr = open(os.path.join(ROOT, "include", "mmif.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mmif_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mmif.h but not exported by {LIB_PATH}"
        assert name in SIGNATURES, f"{name} has no ctypes prototype in mmif/_lib.py"
    assert "gfx950" in mmif.version()


def test_c_abi_argument_validation_without_gpu():
    """Validation runs before any launch, so error paths are testable on CPU."""
    from mmif._lib import MmifTensor, lib
    t = MmifTensor(None, 0, 1, 8, 8, 0, 1, 0, 1, 0)
    assert lib.mmif_zero(ctypes.byref(t), None) == -1
    assert b"null tensor" in lib.mmif_last_error()
    buf = (ctypes.c_float * 64)()
    t = MmifTensor(ctypes.addressof(buf), 0, 1, 8, 8, 0, 1, 0, 2, 0)   # view wider than the allocation
    assert lib.mmif_zero(ctypes.byref(t), None) == -1
    assert lib.mmif_packed_weight_bytes(128, 128, 3) == 9 * 16 * 128 * 16
    assert lib.mmif_conv2d_wgrad_workspace(128, 128, 3) > 0 and lib.mmif_loss_workspace(2, 64, 64) > 0


def test_c_abi_validation_of_the_widened_entry_points_without_gpu():
    """pair conv / SSIM modes / TV / patch feed: bad arguments are rejected before anything is launched (no GPU needed),
    with the reference's messages where it has one."""
    from mmif._lib import MmifTensor, lib
    buf = (ctypes.c_char * 4096)()
    base = ctypes.addressof(buf)
    ok = MmifTensor(base, 1, 1, 4, 4, 0, 1, 0, 1, 0)            # bf16 [1][1 block][4][4][8] = 256 B
    h1 = MmifTensor(base, 1, 1, 4, 4, 1, 1, 0, 1, 0)            # halo-1 view, NOT folded
    w = (ctypes.c_float * 36)()
    r = ctypes.byref
    assert lib.mmif_pairconv_fwd(r(ok), r(ok), w, None, 3, r(ok), r(ok), 1, None, None, None) == -1
    assert b"nout must be 1 or 2" in lib.mmif_last_error()
    assert lib.mmif_pairconv_fwd(r(ok), r(ok), None, None, 2, r(ok), r(ok), 1, None, None, None) == -1
    assert lib.mmif_pairconv_dgrad(r(h1), None, w, 1, None, None, r(h1), r(h1), 0, None, None) == -1
    assert b"folded first" in lib.mmif_last_error()
    assert lib.mmif_pairconv_wgrad(r(ok), r(ok), r(ok), None, 1, w, w, 0, None, 0, None) == -3            # workspace
    assert lib.mmif_pairconv_wgrad_workspace() > 0
    assert lib.mmif_pairconv_bwd(r(ok), None, w, 1, r(ok), r(ok), r(h1), r(h1), 0, None, w, w, 0, None, 0, None) == -3   # workspace
    assert lib.mmif_pairconv_bwd(r(ok), None, w, 2, r(ok), r(ok), r(h1), r(h1), 0, None, w, w, 0, None, 0, None) == -1   # gb missing
    f = (ctypes.c_float * 16)()
    assert lib.mmif_ssim_loss_mode(f, f, f, 1, 32, 32, 1.0, 1.0, 7, f, None, f, 1 << 20, None) == -1
    assert b"only supported ['ssim', 'w-ssim', 'ms-ssim', 'msw-ssim'] mode" in lib.mmif_last_error()
    assert lib.mmif_ssim_loss_mode(f, f, f, 1, 8, 8, 1.0, 1.0, 1, f, None, f, 1 << 20, None) == -1        # < 11x11
    assert lib.mmif_ssim_loss_mode(f, f, f, 1, 32, 32, 1.0, 1.0, 1, f, None, f, 16, None) == -3
    assert lib.mmif_ssim_loss_mode_workspace(2, 256, 256, 2) > lib.mmif_ssim_loss_mode_workspace(2, 256, 256, 1) > 0
    assert lib.mmif_tv_loss(f, 1, 1, 8, 1.0, 0, f, None, f, 1 << 16, None) == -1                          # needs 2x2
    assert lib.mmif_tv_loss(f, 1, 8, 8, 1.0, 0, f, None, f, 16, None) == -3
    assert lib.mmif_patch_feed(f, 4, 8, f, None, 2, 5, f, None) == -1
    assert b"only supported ['min-max', 'z-score'] mode" in lib.mmif_last_error()
    assert lib.mmif_patch_feed(None, 4, 8, f, None, 2, 0, f, None) == -1
    assert lib.mmif_fuse_attn_workspace(2, 64) > 0
    assert lib.mmif_gconv_fwd(f, f, None, f, 1, 8, 8, 16, 16, 4, 1, 2, 1, 0, None) == -1
    assert b"ksize must be 1, 3, 5 or 7" in lib.mmif_last_error()
    assert lib.mmif_gconv_fwd(f, f, None, f, 1, 8, 8, 16, 16, 3, 3, 1, 1, 0, None) == -1
    assert b"stride must be 1 or 2" in lib.mmif_last_error()
    assert lib.mmif_gconv_fwd(f, f, None, f, 1, 8, 8, 3, 16, 7, 1, 3, 1, 0, None) == -1           # reflect 3 on a 3-row image
    assert lib.mmif_gconv_dgrad(f, f, f, 1, 8, 8, 16, 16, 5, 1, 2, 1, None, 0, None) == -3
    assert lib.mmif_gconv_dgrad_workspace(2, 8, 16, 16, 2, 1) == 2 * 8 * 20 * 20 * 4 and lib.mmif_gconv_dgrad_workspace(2, 8, 16, 16, 2, 0) == 0
    assert lib.mmif_gconv_wgrad(f, f, f, None, 1, 8, 8, 16, 16, 5, 1, 2, 1, f, 64, None) == -3
    assert lib.mmif_gconvt_fwd(f, f, None, f, 1, 8, 8, 4, 4, 3, 2, 1, 2, 0, None) == -1           # output_padding >= stride
    assert lib.mmif_relu_bwd(None, f, f, 4, None) == -1 and lib.mmif_channel_sum(f, f, 0, 1, 1, None) == -1
    assert lib.mmif_reflect_pad_fwd(f, f, 1, 4, 4, 4, 0, 0, 0, None) == -1      # padding must be smaller than the input
    assert lib.mmif_maxpool_nchw_fwd(f, f, f, 1, 3, 8, 4, None) == -1 and lib.mmif_nearest_up_fwd(f, f, 1, 2, 2, 0, None) == -1
    from mmif._lib import MmifPackJob
    jobs = (MmifPackJob * 1)()
    assert lib.mmif_pack_weights_multi(jobs, 0, None) == -1
    jobs[0].w, jobs[0].cout, jobs[0].cin, jobs[0].ksize = base, 16, 16, 5
    assert lib.mmif_pack_weights_multi(jobs, 1, None) == -1
    assert b"ksize must be 1 or 3" in lib.mmif_last_error()


@pytest.mark.parametrize("name", ["PFNetv1", "PFNetv2", "DenseFuse", "VIFNet", "NestFuse", "RFNNest", "DeepFuse", "DBNet", "SEDRFuse", "IFCNN",
                                  "DIFNet", "PMGI", "UNFusion", "MAFusion", "Res2Fusion"])
def test_state_dict_manifest_and_init(name):
    import core.model as M
    man = json.load(open(os.path.join(G, {"VIFNet": "f10_manifest.json", "DeepFuse": "f12_manifest.json", "DBNet": "f12_manifest.json",
                                          "SEDRFuse": "f13_manifest.json", "IFCNN": "f13_manifest.json", "DIFNet": "f13_manifest.json",
                                          "PMGI": "f13_manifest.json", "UNFusion": "f14_manifest.json", "MAFusion": "f14_manifest.json", "Res2Fusion": "f15_manifest.json"}.get(name, "f5_manifest.json"))))
    torch.manual_seed(0)
    m = getattr(M, name)()
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == man[name]
    # reference init (core/block.py:101-118): zero biases; kaiming-normal weights for ReLU layers
    for k, v in m.state_dict().items():
        if k.endswith("bias"):
            assert float(v.abs().max()) == 0.0
    if name in ("NestFuse", "RFNNest", "DeepFuse", "SEDRFuse", "IFCNN", "DIFNet", "PMGI", "UNFusion", "MAFusion", "Res2Fusion"):
        return
    w = m.state_dict()["decode.0.layers.0.weight"]
    fan_in = w.shape[1] * 9
    assert abs(float(w.std()) - (2.0 / fan_in) ** 0.5) / (2.0 / fan_in) ** 0.5 < 0.05


def test_reference_init_is_bitwise_reproduced_when_reference_is_present():
    ref_root = "/root/reference"
    if not os.path.isdir(ref_root):
        pytest.skip("reference not mounted (GPU box)")
    import importlib.util
    import sys
    import core.model as M
    torch.manual_seed(0)
    mine = M.PFNetv1().state_dict()
    # import the reference's core package under a private name
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "core" or k.startswith("core.")}
    sys.path.insert(0, ref_root)
    try:
        sys.dont_write_bytecode = True
        import core.model as R
        torch.manual_seed(0)
        ref = R.PFNetv1().state_dict()
    finally:
        sys.path.remove(ref_root)
        for k in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
            sys.modules.pop(k)
        sys.modules.update(saved)
    assert list(mine) == list(ref)
    for k in ref:
        assert torch.equal(mine[k], ref[k]), k


def test_conv_layer_signature_and_fallback_rules():
    from core.block import ConvBlock, ConvLayer, DenseBlock, NestDecoder, RFN
    assert ConvLayer(16, 16)._hip and ConvLayer(8, 64, ksize=1)._hip and ConvLayer(16, 1, act=None)._hip
    # argument combinations outside the hot path stay stock torch modules
    # the general kernels (k 5/7, stride 2, zero padding, ConvTranspose2d) -- still HIP, fp32 NCHW
    for lay in (ConvLayer(16, 16, stride=2), ConvLayer(16, 16, ksize=5), ConvLayer(1, 16, ksize=7), ConvLayer(16, 16, padding_mode='zeros'),
                ConvLayer(16, 8, stride=2, layer=nn.ConvTranspose2d)):
        assert lay._gen and not lay._hip
    # BatchNorm / GroupNorm(c, c) and LeakyReLU / Tanh: HIP conv + the norm / activation epilogue kernels
    for lay in (ConvLayer(16, 16, norm=nn.BatchNorm2d), ConvLayer(16, 16, act=nn.Tanh), ConvLayer(16, 16, norm=nn.GroupNorm, stride=2),
                ConvLayer(3, 16, ksize=5, norm=nn.BatchNorm2d, act=nn.LeakyReLU)):
        assert lay._epilogue and not lay._hip and not lay._gen
    # everything else stays the stock torch modules
    assert ConvLayer(16, 16, groups=16, bias=False, act=None)._depthwise and ConvLayer(16, 64, ksize=1, bias=False, act=nn.ReLU6)._epilogue
    for lay in (ConvLayer(16, 16, dilation=2, padding=2), ConvLayer(16, 16, pre_norm=nn.BatchNorm2d), ConvLayer(16, 16, act=nn.Sigmoid),
                ConvLayer(16, 16, groups=2)):
        assert not lay._hip and not lay._gen and not lay._epilogue
    assert list(DenseBlock(16, 16).state_dict())[0] == "layers.0.layers.0.weight"
    assert sum(p.numel() for p in ConvBlock(16, 64).parameters()) == 16 * 8 * 9 + 8 + 8 * 64 + 64
    assert len(list(RFN(16).parameters())) == 12
    assert "DB1_3.layers.1.layers.0.bias" in NestDecoder(ConvBlock, [8, 16, 24, 32], "nearest").state_dict()


def test_errors_match_reference_behaviour():
    import core.fusion as F
    import core.loss as L
    import core.model as M
    x = torch.zeros(1, 8, 4, 4)
    with pytest.raises(ValueError, match="sum"):
        F.element_fusion(x, x, "nope")
    with pytest.raises(ValueError):
        F.attention_fusion(x, x, "nope")
    with pytest.raises(ValueError):
        F.spatial_pooling(x, "nope")
    with pytest.raises(ValueError):
        F.channel_pooling(x, "nope")
    with pytest.raises(ValueError):
        L.NormLoss("l3")(x)
    with pytest.raises(ValueError):
        M.DenseFuse().fusion(x, x, mode="nope")
    with pytest.raises(NotImplementedError):
        M._FusionModel().fusion(x, x)
    # no silent CPU fallback for the hot path
    with pytest.raises(RuntimeError, match="GPU"):
        M.DenseFuse()(torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="GPU"):
        F.element_fusion(x, x, "sum")


def test_fusion_function_compositions_on_cpu_match_oracle():
    """attention / pooling functions are tensor-level compositions (valid on any device)."""
    import core.fusion as F
    from oracle import fusion_oracle as O
    s = (2, 16, 6, 10)
    a, b = O.closed_form_signed(s, 0.15), O.closed_form_signed(s, 1.25)
    for mode in ("sa", "ca", "sca"):
        got = F.attention_fusion(torch.from_numpy(a), torch.from_numpy(b), mode).numpy()
        np.testing.assert_allclose(got, O.attention_fusion(a, b, mode), rtol=2e-5, atol=2e-6)
    z = torch.zeros(s)
    assert float(F.attention_fusion(z, z, "sca").abs().max()) == 0.0


def test_common_plumbing():
    import common as C
    m = C.AverageMeter()
    assert m.is_empty()
    m.update(2.0, 4)
    m.update(4.0, 4)
    assert m.avg == 3.0 and m.count == 8
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = C.WarmupLR(opt, 0.001, 10)   # the reference's call form (train.py:323): (optimizer, warmup_factor, warmup_iters)
    lrs = []
    for _ in range(12):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    want = [0.001 + 0.999 * i / 10 for i in range(10)] + [1.0, 1.0]   # common.py:156-166 'linear'
    assert np.allclose(lrs, want, atol=1e-12) and lrs == sorted(lrs)
    opt2 = torch.optim.SGD([p], lr=2.0)
    C.WarmupLR(opt2, warmup_factor=0.25, warmup_iters=3, warmup_method="constant")
    assert opt2.param_groups[0]["lr"] == 0.5
    with pytest.raises(ValueError):
        C.WarmupLR(torch.optim.SGD([p], lr=1.0), 0.1, 3, "cosine")
    img = torch.tensor([[[-0.5, 0.0], [0.5, 1.5]]])
    assert C.denorm(img).tolist() == [[[0], [0]], [[127], [255]]]   # truncation, as data/transform.py:32-35
    assert C.save_result(img).tolist() == C.denorm(img).tolist()
    assert abs(float(C.norm(np.array([255.0]))[0]) - 1.0) < 1e-7
    import sys
    argv, sys.argv = sys.argv, ["train.py", "--data", "roadscene", "--bs", "8", "--model", "DenseFuse"]
    try:
        a = C.get_train_args()
    finally:
        sys.argv = argv
    assert a.data == "roadscene" and a.bs == 8 and a.lr == 1e-4 and a.epoch == 12 and a.clip_grad is True and a.model == "DenseFuse"
    argv, sys.argv = sys.argv, ["test.py"]
    try:
        t = C.get_test_args()
    finally:
        sys.argv = argv
    assert t.data == "roadscene" and t.ckpt == "2023-02-26_23-15" and t.use_gpu is True


def test_make_logger_layout(tmp_path):
    """common.py:200-210: <root>/../checkpoints/<time>/train.log, returns (log_dir, logger)"""
    import common as C
    root = tmp_path / "pkg"
    root.mkdir()
    log_dir, logger = C.make_logger(str(root))
    assert os.path.isfile(os.path.join(log_dir, "train.log"))
    assert os.path.normpath(os.path.dirname(log_dir)) == str(tmp_path / "checkpoints")
    logger.info("hello")
    for h in list(logger.handlers):
        h.flush()
        logger.removeHandler(h)
    assert "hello" in open(os.path.join(log_dir, "train.log")).read()


def test_bench_self_launch_refuses_a_smaller_box(monkeypatch):
    """bench.py --gpus N on a box with fewer GPUs must not report an N-GPU line: self_launch exits with the "refusing" message before it
    starts anything (this container has 0 GPUs); the rendezvous port comes from the kernel (bind to port 0), not from the pid."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse()
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    with pytest.raises(SystemExit) as ei:
        bench.self_launch(args)
    assert "refusing" in str(ei.value) and "--gpus 8" in str(ei.value)
    with pytest.raises(SystemExit) as ei:       # the whole entry point takes that exit too
        bench.main()
    assert "refusing" in str(ei.value)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--one-device"])     # debugging aid: gloo only
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert "gloo" in str(ei.value)
    p1, p2 = bench.free_port(), bench.free_port()
    assert 1024 < p1 < 65536 and 1024 < p2 < 65536
    # SURVEY 8(d)'s ideal for the headline config: 1 / max(42.8 us, 45.8 us) = 21.8 k pairs/s
    assert abs(bench.ideal_pairs_per_s("PFNetv1", 256, 256, "bf16") - 21850) < 100
