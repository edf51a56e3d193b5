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

Rank 0 prints ONE JSON line (contract in the task description) including
  roofline     -- the dominant kernel timed live with HIP events inside the timed region
  cpu_baseline -- the CPU oracle ("port") timed on this box's host cores on a bounded sample
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
    ap.add_argument("--graph", action="store_true",
                    help="capture forward + losses + backward in ONE hipGraph and replay it per step (launch-bound small batches); "
                         "the gradient all-reduce and clip+Adam stay outside the graph")
    ap.add_argument("--cpu-sample", type=int, default=2, help="image pairs in the CPU-oracle sample")
    ap.add_argument("--roofline-tag", default="decode.0:fwd", help="engine op timed with HIP events for `roofline`")
    ap.add_argument("--hbm-tag", default="auto", help="an HBM-bound engine op timed the same way for `roofline_hbm` ('' = none)")
    return ap.parse_args()


def cpu_baseline(model_name, size, n_pairs):
    """The CPU oracle (numpy restatement, oracle/fusion_oracle.py) running the SAME train step on the host."""
    import numpy as np
    from oracle import fusion_oracle as O
    m = O.MODELS[model_name]()
    P = m.init_params(seed=0)
    st = O.AdamState(P)
    rng = np.random.default_rng(0)
    shape = (n_pairs, 1, size, size)
    i1, i2 = rng.random(shape, dtype=np.float32), rng.random(shape, dtype=np.float32)
    t0 = time.time()
    O.train_step(m, P, st, i1, i2)
    dt = time.time() - t0
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    return {"value": n_pairs / dt, "unit": "image-pairs/s", "cores": int(threads), "kind": "port",
            "sample": f"1 train step of {model_name} on {n_pairs} pairs {size}x{size} fp32 (numpy oracle), {dt:.1f} s; host has {os.cpu_count()} logical CPUs"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    from core.loss import GradLoss, PixelLoss, SSIMLoss
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
        a, b, c = l_ssim(img1, img2, f), l_pix(img1, img2, f, mode='max'), l_grad(img1, img2, f, mode='max')
        tot = a + b + c
        opt.stage_scalars([tot, a, b, c])      # (data parallel) the loss values ride in the early gradient all-reduce
        tot.backward()
        opt.step(scalars=[tot, a, b, c])
        return tot

    for _ in range(args.warmup):
        step()
    if args.graph and args.mode == "train":
        # same kernels, same work per step -- only the ~100 launches of forward / losses / backward become one graph launch
        from mmif.graph import GraphedStep

        def losses(i1, i2, f):
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
                roof = {"bound": "mfma", "kernel": f"{'conv_dma_kernel' if dma else 'conv_mfma_kernel'} {s.cin}->{s.cout} k{s.k} ({args.roofline_tag})",
                        "achieved": ach / 1e12, "peak": PEAK_MFMA_BF16 / 1e12 if args.dtype == "bf16" else 157.3,
                        "unit": "TFLOP/s", "frac": ach / (PEAK_MFMA_BF16 if args.dtype == "bf16" else 157.3e12),
                        "avg_launch_ms": ms, "launches": len(evs), "traffic": None}
                # HBM bytes per launch of this kernel from the TCC PMC passes of the same command (separate
                # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied; profiles/r02_traffic.json, tools/prof_pmc.sh)
                tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
                if os.path.isfile(tpath) and B == 32 and S == 256 and Wd == 256 and args.dtype == "bf16" and args.mode == "train":
                    tj = json.load(open(tpath)).get(args.roofline_tag)
                    if tj:
                        roof["traffic"] = tj["hbm_bytes_per_launch"]
                        roof["algorithmic_bytes"] = tj["algorithmic_bytes_per_launch"]
        roof_hbm = None
        evh = T.PROFILE_EVENTS.get(hbm_tag, []) if hbm_tag else []
        if evh:
            ms = sum(a.elapsed_time(b) for a, b in evh) / len(evh)
            spec = [s for s in model._engine.specs if hbm_tag.startswith(s.name + ":")]
            if hbm_tag == "encode:fwd":
                # algorithmic bytes of the 2 x 4 conv passes it replaces (H*W*(Cin+Cout)*sizeof per pass, the image in fp32)
                # -- the launch itself only moves the two images and the 128 output planes (`fused_bytes`)
                per_px = 2 * ((1 * 4 + 16 * 2) + (16 + 16) * 2 + (32 + 16) * 2 + (48 + 16) * 2)
                nbytes = float(B) * S * Wd * per_px
                roof_hbm = {"bound": "hbm", "kernel": "enc_stream_fwd_kernel 2 x (1->16, 16->16, 32->16, 48->16) k3 (encode:fwd)",
                            "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / PEAK_HBM,
                            "avg_launch_ms": ms, "launches": len(evh), "traffic": None, "algorithmic_bytes": nbytes,
                            "fused_bytes": float(B) * S * Wd * 2 * (4 + 64 * 2)}
                tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
                if os.path.isfile(tpath) and B == 32 and S == 256 and Wd == 256 and args.mode == "train" and args.model == "PFNetv1":
                    tj = json.load(open(tpath)).get(hbm_tag)
                    if tj:
                        roof_hbm["traffic"] = tj["hbm_bytes_per_launch"]
            elif spec:
                s = spec[0]
                esz = 2 if args.dtype == "bf16" else 4
                nbytes = float(B) * S * Wd * (s.cin + s.cout) * esz      # algorithmic bytes of one conv pass
                roof_hbm = {"bound": "hbm", "kernel": f"conv_mfma_kernel {s.cin}->{s.cout} k{s.k} ({hbm_tag})", "achieved": nbytes / (ms * 1e-3) / 1e9,
                            "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / PEAK_HBM, "avg_launch_ms": ms,
                            "launches": len(evh), "traffic": None}
                tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
                if os.path.isfile(tpath) and B == 32 and S == 256 and Wd == 256 and args.dtype == "bf16" and args.mode == "train":
                    tj = json.load(open(tpath)).get(hbm_tag if hbm_tag in ("encode1.1.2:fwd",) else "")
                    if tj:   # (PFNetv1's entry; the other models' tags have no counter pass on file)
                        roof_hbm["traffic"] = tj["hbm_bytes_per_launch"]
                        roof_hbm["algorithmic_bytes"] = tj["algorithmic_bytes_per_launch"]
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
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "train" and Wd == S:
            out["cpu_baseline"] = cpu_baseline(args.model, S, args.cpu_sample)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
