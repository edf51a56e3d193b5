#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec of the PFNet train step at 256x256 (BASELINE.json).

A "step" = one pass of the hot path over one synthetic batch already resident in HBM:
PFNetv1 forward -> SSIM + max-pixel + Sobel-gradient losses -> backward -> (RCCL gradient
all-reduce when N > 1) -> clip_grad_norm_(5) + Adam, i.e. train.py:61-75 of the reference, run
through the drop-in API (core.model.PFNetv1, core.loss.*, mmif.optim.FusedClipAdam).
Workload = BASELINE.json configs[1]: PFNet 256x256 IR/visible pairs, bf16 feature maps, batch 32
per GPU (weak scaling: per-GPU work is fixed as N grows).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher environment starts the N ranks itself (a child `python -m torch.distributed.run`, started before
this process touches the GPU) and relays rank 0's line; a box with fewer than N GPUs is an error, never a silent 1-GPU run.

Rank 0 prints ONE JSON line (contract in the task description) including
  roofline     -- the dominant kernel timed live with HIP events inside the timed region
  parity_path  -- the same step on the parity-grade path (fp32 tensors, 3x3 / 1x1 layers on the matrix pipe as split-operand products,
                  csrc/conv_x3.hip) timed the same way, with its error against the CPU oracle on a small sample (N = 1 only)
  cpu_baseline -- the torch-CPU restatement of the step (oracle/torch_cpu_step.py, "port") timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

MODEL_FLOPS_TRAIN = {"PFNetv1": 107.04e9, "PFNetv2": 36.82e9, "DenseFuse": 34.56e9}  # per pair @256x256
MODEL_BYTES_TRAIN_BF16 = {"PFNetv1": 366.1e6}
PEAK_MFMA_BF16 = 2.5e15   # dense, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU")
    ap.add_argument("--size", type=int, default=256, help="image height (and width unless --width)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="infer = forward only under no_grad (test.py path; BASELINE config 5: --mode infer --batch 1 --size 1024 --width 1224)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="PFNetv1", choices=["PFNetv1", "PFNetv2", "DenseFuse", "VIFNet", "NestFuse", "RFNNest", "DeepFuse", "DBNet", "SEDRFuse", "IFCNN", "DIFNet", "PMGI", "UNFusion", "MAFusion", "Res2Fusion"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-losses", action="store_true", help="the three loss modules + torch additions instead of core.loss.FusionLoss (one call)")
    ap.add_argument("--graph", action="store_true",
                    help="capture forward + losses + backward in ONE hipGraph and replay it per step (launch-bound small batches); "
                         "the gradient all-reduce and clip+Adam stay outside the graph")
    ap.add_argument("--cpu-sample", type=int, default=8, help="image pairs per step of the CPU sample")
    ap.add_argument("--no-parity-path", action="store_true", help="skip the second timed leg on the parity-grade fp32 / split-bf16 path")
    ap.add_argument("--parity-steps", type=int, default=16)
    ap.add_argument("--roofline-tag", default="decode.0:fwd", help="engine op timed with HIP events for `roofline`")
    ap.add_argument("--hbm-tag", default="auto", help="an HBM-bound engine op timed the same way for `roofline_hbm` ('' = none)")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(model_name, size, n_pairs, budget_s=12.0):
    """The torch-CPU restatement of the SAME train step (oracle/torch_cpu_step.py: stock torch CPU ops, pinned to the reference's goldens
    F5 / F6 by tests/test_torch_cpu_step.py) on this box's host cores, with the intra-op thread count torch picks: one untimed step,
    then timed steps of n_pairs pairs until budget_s is spent (at most 4)."""
    from oracle import torch_cpu_step as TC
    if model_name not in ("PFNetv1", "DenseFuse"):
        return None
    m = TC.TorchCpuModel(model_name)
    P = m.init_params(0)
    opt = TC.make_optimizer(P)
    g = torch.Generator().manual_seed(0)
    shape = (n_pairs, 1, size, size)
    i1, i2 = torch.rand(shape, generator=g), torch.rand(shape, generator=g)
    TC.train_step(m, P, opt, i1[:max(1, n_pairs // 4)], i2[:max(1, n_pairs // 4)])   # thread pool / allocator warm-up
    t0, steps = time.time(), 0
    while steps < 4 and (steps == 0 or time.time() - t0 < budget_s):
        TC.train_step(m, P, opt, i1, i2)
        steps += 1
    dt = time.time() - t0
    return {"value": n_pairs * steps / dt, "unit": "image-pairs/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{steps} train step(s) of {model_name} on {n_pairs} pairs {size}x{size} fp32, torch {torch.__version__} CPU ops "
                      f"(oracle/torch_cpu_step.py), {dt:.1f} s; torch.get_num_threads() = {torch.get_num_threads()}, os.cpu_count() = {os.cpu_count()}, "
                      f"CPU: {_cpu_model()}"}


def lib_sha256():
    import hashlib
    from mmif import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def measured_traffic(tag, match):
    """HBM bytes per launch of the kernel behind `tag` from the TCC counter passes on file (profiles/r03_traffic.json, written by
    tools/make_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command) -- only when those passes ran on
    THIS library build (sha256 of the .so) and this workload; otherwise (None, reason)."""
    tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
    if not os.path.isfile(tpath):
        return None, "no counter pass on file"
    tj = json.load(open(tpath))
    if tj.get("workload") != match:
        return None, f"counter pass on file is for {tj.get('workload')!r}"
    if tj.get("lib_sha256") != lib_sha256():
        return None, "counter pass on file was taken on another library build (stale)"
    ent = tj.get("kernels", {}).get(tag)
    if not ent:
        return None, "kernel not in the counter pass on file"
    return ent, f"profiles/r03_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, lib {tj['lib_sha256'][:12]})"


def parity_leg(args, dev, img1, img2):
    """The same train step on the parity-grade path: fp32 feature maps, 3x3 / 1x1 layers on the matrix pipe as split-operand products (three
    per tap: fp16 pieces forward, bf16 pieces backward; csrc/conv_x3.hip), everything else fp32.  Timed like the main leg; its error is taken against the CPU
    oracle on a small closed-form sample (2 pairs of 64 x 64: fused image, total loss, every parameter gradient)."""
    import numpy as np
    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss
    from mmif import engine as E
    from mmif._lib import lib
    from mmif.optim import FusedClipAdam
    from oracle import fusion_oracle as O
    prev = E.compute_dtype()
    E.set_compute_dtype("fp32")
    try:
        l_ssim, l_pix, l_grad = SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev)
        l_all = None if args.separate_losses else FusionLoss(l_ssim, l_pix, l_grad, 'max', 'max')

        def one(model, opt, a, b):
            opt.zero_grad(set_to_none=True)
            f = model(a, b)
            tot = l_all(a, b, f) if l_all is not None else l_ssim(a, b, f) + l_pix(a, b, f, mode='max') + l_grad(a, b, f, mode='max')
            tot.backward()
            opt.step()
            return f, tot
        torch.manual_seed(0)
        model = getattr(M, args.model)().to(dev)
        opt = FusedClipAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), max_norm=5.0)
        for _ in range(6):     # (warm-up as the main leg's: packing, leases, the clock's ramp after the idle oracle / setup phase)
            one(model, opt, img1, img2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.parity_steps):
            one(model, opt, img1, img2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del model, opt
        # error against the oracle, small sample
        shape = (2, 1, 64, 64)
        i1n, i2n = O.closed_form_image(shape, 0.3), O.closed_form_image(shape, 1.7)
        om = O.MODELS[args.model]()
        P = om.init_params(seed=1)
        ref = O.train_step(om, P, O.AdamState(P), i1n, i2n, clip=None)
        m = getattr(M, args.model)()
        m.load_state_dict({k: torch.from_numpy(O.closed_form_param(i, k, tuple(v.shape), 1)) for i, (k, v) in enumerate(m.state_dict().items())})
        m = m.to(dev)
        o2 = FusedClipAdam(m.parameters(), lr=1e-4, max_norm=0.0)
        i1, i2 = torch.from_numpy(i1n).to(dev), torch.from_numpy(i2n).to(dev)
        o2.zero_grad(set_to_none=True)
        f = m(i1, i2)
        tot = l_ssim(i1, i2, f) + l_pix(i1, i2, f, mode='max') + l_grad(i1, i2, f, mode='max')
        tot.backward()
        torch.cuda.synchronize()
        err_img = float(np.abs(f.detach().cpu().numpy() - ref["imgf"]).max() / np.abs(ref["imgf"]).max())
        gerr = max(float(np.abs(p.grad.cpu().numpy() - ref["grads"][k]).max() / max(np.abs(ref["grads"][k]).max(), 1e-12)) for k, p in m.named_parameters())
        B = img1.shape[0]
        mode = lib.mmif_get_x3_forward_pieces()
        fwd = {16: "3 products of scaled fp16 pieces", 3: "6 products of bf16 pieces", 2: "3 products of bf16 pieces"}[mode]
        return {"dtype": f"fp32 storage; 3x3 / 1x1 layers as split-operand MFMA products (forward: {fwd}; backward: 3 products of bf16 pieces), fp32 accumulate",
                "value": B * args.parity_steps / dt, "unit": "image-pairs/s", "ms_per_step": dt / args.parity_steps * 1e3, "steps": args.parity_steps,
                "rel_err_vs_oracle": err_img, "loss_abs_err_vs_oracle": abs(float(tot.item()) - float(ref["losses"][3])), "grad_rel_err_vs_oracle": gerr,
                "oracle_sample": "2 pairs 64x64, closed-form weights / images: max |fused image - oracle| / max|oracle|; max over parameters of max |grad - oracle| / max|oracle grad|",
                "tolerance": "BASELINE north star: 1e-3 relative"}
    finally:
        E.set_compute_dtype(prev)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher environment: start the N ranks as a CHILD torch.distributed.run -- this
    process has not initialised the GPU (torch.cuda.device_count() does not) and never will -- relay the output, return its code."""
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this box has {have} visible GPU(s); refusing to report a {have}-GPU run as {args.gpus}")
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} rank(s)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # $MMIF_FORCE_DIST=1: take the RCCL code path (init, parameter broadcast, gradient all-reduce, barriers) with one rank too
    use_dist = world > 1 or os.environ.get("MMIF_FORCE_DIST", "0") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss
    from mmif import engine as E
    from mmif import tensor as T
    from mmif.dist import broadcast_parameters
    from mmif.optim import FusedClipAdam

    E.set_compute_dtype(args.dtype)
    torch.manual_seed(0)
    model = getattr(M, args.model)().to(dev)
    if use_dist:
        broadcast_parameters(model, 0)
    opt = FusedClipAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), max_norm=5.0)
    l_ssim, l_pix, l_grad = SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev)
    # the three terms + their sum + the three gradient contributions as ONE device call (same kernels; --separate-losses: three modules
    # and torch additions, as the reference's train.py:64-69 writes it)
    l_all = None if args.separate_losses else FusionLoss(l_ssim, l_pix, l_grad, 'max', 'max')

    B, S = args.batch, args.size
    Wd = args.width or S
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    img1 = torch.rand(B, 1, S, Wd, generator=gen).to(dev)  # synthetic IR / visible pairs in [0,1)
    img2 = torch.rand(B, 1, S, Wd, generator=gen).to(dev)

    def infer_step():
        with torch.no_grad():
            return model(img1, img2).mean()

    def step():
        if args.mode == "infer":
            return infer_step()
        opt.zero_grad(set_to_none=True)
        f = model(img1, img2)
        if l_all is not None:
            tot = l_all(img1, img2, f)
            vals = l_all.values
        else:
            a, b, c = l_ssim(img1, img2, f), l_pix(img1, img2, f, mode='max'), l_grad(img1, img2, f, mode='max')
            tot = a + b + c
            vals = [tot, a, b, c]
        opt.stage_scalars(vals)      # (data parallel) the loss values ride in the early gradient all-reduce
        tot.backward()
        opt.step(scalars=vals)
        return tot

    for _ in range(args.warmup):
        step()
    if args.graph and args.mode == "train":
        # same kernels, same work per step -- only the ~100 launches of forward / losses / backward become one graph launch
        from mmif.graph import GraphedStep

        def losses(i1, i2, f):
            if l_all is not None:
                tot = l_all(i1, i2, f)
                return (tot,) + tuple(l_all.values[1:4].unbind(0))
            a, b, c = l_ssim(i1, i2, f), l_pix(i1, i2, f, mode='max'), l_grad(i1, i2, f, mode='max')
            return a + b + c, a, b, c
        gstep = GraphedStep(model, losses, opt, img1, img2)

        def step():   # noqa: F811
            return gstep(img1, img2)[0]
        for _ in range(2):
            step()
    hbm_tag = args.hbm_tag
    enc_stream = args.dtype == "bf16" and os.environ.get("MMIF_ENC_STREAM", "1") != "0" and args.model in ("PFNetv1", "DenseFuse", "PFNetv2", "VIFNet")
    if hbm_tag == "auto":   # the encoder: ONE streaming launch for its 2 x 4 layers, or (layer-wise) its widest thin layer 48 -> 16
        hbm_tag = "encode:fwd" if enc_stream else ({"PFNetv1": "encode1.1.2:fwd", "DenseFuse": "encode.1.2:fwd", "PFNetv2": "encode.1.2:fwd", "VIFNet": "encode.1.2:fwd"}.get(args.model, "") if args.mode == "train" else "")
    T.PROFILE_TAGS = {args.roofline_tag} | ({hbm_tag} if hbm_tag else set())
    T.PROFILE_EVENTS.clear()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tot = step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    T.PROFILE_TAGS = set()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(tot.item())

    if rank == 0:
        pairs = B * world * args.steps
        value = pairs / dt
        workload_id = f"{args.model} {args.mode} B={B} {Wd}x{S} {args.dtype}"
        # dominant kernel, timed live (HIP events on the launch stream, inside the timed region)
        evs = T.PROFILE_EVENTS.get(args.roofline_tag, [])
        roof = None
        if evs:
            ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
            spec = [s for s in model._engine.specs if args.roofline_tag.startswith(s.name + ":")]
            if spec:
                s = spec[0]
                flops = 2.0 * B * S * Wd * s.cin * s.cout * s.k * s.k
                ach = flops / (ms * 1e-3)
                dma = os.environ.get("MMIF_CONV_DMA", "1") != "0" and s.k == 3 and s.cout >= 49 and args.dtype == "bf16"
                kname = "conv_dma_kernel" if dma else "conv_mfma_kernel"
                if args.dtype == "fp32":
                    # fp32 tensors: the split-bf16 kernel issues `prods` bf16 MFMA products per algorithmic product; its roofline is the
                    # bf16 matrix peak over the MFMA flops it really executes
                    from mmif._lib import lib as _l
                    prods = {16: 3, 3: 6, 2: 3}[_l.mmif_get_x3_forward_pieces()] if args.roofline_tag.endswith(":fwd") else 3
                    kname = f"conv_x3_kernel ({prods} half-precision MFMA products per tap)"
                    ach *= prods
                roof = {"bound": "mfma", "kernel": f"{kname} {s.cin}->{s.cout} k{s.k} ({args.roofline_tag})",
                        "achieved": ach / 1e12, "peak": PEAK_MFMA_BF16 / 1e12,
                        "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_BF16,
                        "avg_launch_ms": ms, "launches": len(evs), "traffic": None}
                # HBM bytes per launch of this kernel from the TCC PMC passes of the same command (separate rocprofv3 --pmc FETCH_SIZE /
                # WRITE_SIZE runs, gfx950 correction applied, tools/prof_pmc.sh + tools/make_traffic.py) -- only when they were taken on
                # THIS library build; a stale file prints null
                roof["algorithmic_bytes"] = float(B) * S * Wd * (s.cin + s.cout) * (2 if args.dtype == "bf16" else 4)
                ent, src = measured_traffic(args.roofline_tag, workload_id)
                roof["traffic"] = ent["hbm_bytes_per_launch"] if ent else None
                roof["traffic_source"] = src
        roof_hbm = None
        evh = T.PROFILE_EVENTS.get(hbm_tag, []) if hbm_tag else []
        if evh:
            ms = sum(a.elapsed_time(b) for a, b in evh) / len(evh)
            spec = [s for s in model._engine.specs if hbm_tag.startswith(s.name + ":")]
            if hbm_tag == "encode:fwd":
                # algorithmic bytes of the 2 x 4 conv passes it replaces (H*W*(Cin+Cout)*sizeof per pass, the image in fp32)
                # -- the launch itself only moves the two images and the 128 output planes (`fused_bytes`)
                # -- the launch itself only moves the images and the output planes (`fused_bytes`): achieved / frac are the bytes the launch
                # REALLY moves (the counter figure when one is on file for this build, else fused_bytes) over its duration; the rate in
                # terms of the layer-wise passes it replaces is reported apart (it is not a bandwidth and may exceed the peak)
                nbr = 2 if args.model in ("PFNetv1", "VIFNet", "DenseFuse", "PFNetv2") else 1
                per_px = nbr * ((1 * 4 + 16 * 2) + (16 + 16) * 2 + (32 + 16) * 2 + (48 + 16) * 2)
                nbytes = float(B) * S * Wd * per_px
                fused = float(B) * S * Wd * nbr * (4 + 64 * 2)
                ent, src = measured_traffic(hbm_tag, workload_id)
                moved = ent["hbm_bytes_per_launch"] if ent else fused
                roof_hbm = {"bound": "hbm", "kernel": f"enc_stream_fwd_kernel {nbr} x (1->16, 16->16, 32->16, 48->16) k3 (encode:fwd)",
                            "achieved": moved / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": moved / (ms * 1e-3) / PEAK_HBM,
                            "avg_launch_ms": ms, "launches": len(evh), "traffic": ent["hbm_bytes_per_launch"] if ent else None, "traffic_source": src,
                            "fused_bytes": fused, "algorithmic_bytes_layerwise": nbytes, "equivalent_layerwise_GBps": nbytes / (ms * 1e-3) / 1e9}
            elif spec:
                s = spec[0]
                esz = 2 if args.dtype == "bf16" else 4
                nbytes = float(B) * S * Wd * (s.cin + s.cout) * esz      # algorithmic bytes of one conv pass
                ent, src = measured_traffic(hbm_tag, workload_id)
                roof_hbm = {"bound": "hbm", "kernel": f"{'conv_x3_kernel' if args.dtype == 'fp32' else 'conv_mfma_kernel'} {s.cin}->{s.cout} k{s.k} ({hbm_tag})",
                            "achieved": nbytes / (ms * 1e-3) / 1e9,
                            "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / PEAK_HBM, "avg_launch_ms": ms,
                            "launches": len(evh), "traffic": ent["hbm_bytes_per_launch"] if ent else None, "traffic_source": src, "algorithmic_bytes": nbytes}
        out = {
            "metric": "image-pairs/sec at 256x256, PFNet train step" if args.mode == "train" else f"image-pairs/sec at {Wd}x{S}, {args.model} inference", "value": value, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "graph": bool(args.graph and args.mode == "train"),
            "config": {"workload": (f"{args.model} train step (fwd + SSIM/pixel/grad losses + bwd + clip + Adam)" if args.mode == "train" else f"{args.model} forward (no_grad, test.py path)") + f", {Wd}x{S} synthetic IR/visible pairs, "
                                   f"batch {B} per GPU, {args.dtype} feature maps / fp32 accumulate, random-init weights (seed 0)",
                       "global_batch": B * world, "parallelism": f"dp{world}" if world > 1 else "single",
                       "step_model_tflops": MODEL_FLOPS_TRAIN.get(args.model, 0) * value / 1e12 if (S == 256 and Wd == 256 and args.mode == "train") else None},
            "final_loss": loss,
            "roofline": roof,
            "roofline_hbm": roof_hbm,
            "parity_path": None,
            "cpu_baseline": None,
        }
        assert out["n_gpus"] == args.gpus
        if world == 1 and not args.no_parity_path and args.mode == "train" and args.dtype == "bf16" and args.model in ("PFNetv1", "DenseFuse", "PFNetv2", "VIFNet"):
            out["parity_path"] = parity_leg(args, dev, img1, img2)
        if world == 1 and not args.no_cpu_baseline and args.mode == "train" and Wd == S:
            out["cpu_baseline"] = cpu_baseline(args.model, S, args.cpu_sample)
    else:
        out = None
    if use_dist:
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which -- stdout being a pipe -- sits in
    # libc's buffer until exit, i.e. would land AFTER a line printed here; every rank flushes it out now, rank 0 prints a moment later
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if out is not None:
        if use_dist:
            time.sleep(0.3)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
