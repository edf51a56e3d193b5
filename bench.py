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
  other_configs -- BASELINE.json's configs 3, 4, 5 on this GPU (DenseFuse B=32 256^2, NestFuse / RFN-Nest B=4 512^2 train steps, PFNetv1
                  inference on one 1224x1024 pair): 20 timed steps each after 6 warm-up steps, value / ms_per_step / fraction of ideal (N = 1 only)
  cpu_baseline -- the torch-CPU restatement of the step (oracle/torch_cpu_step.py, "port") timed on this box's host cores
"""
import argparse
import gc
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

# algorithmic work of one train step per image pair (SURVEY.md section 8d / BASELINE.md section 3): (GFLOP, MB of bf16 tensors, side)
MODEL_WORK_TRAIN = {"PFNetv1": (107.04e9, 366.1e6, 256), "PFNetv2": (36.82e9, 517.1e6, 256), "DenseFuse": (34.56e9, 240.3e6, 256),
                    "NestFuse": (1828.4e9, 4006.9e6, 512), "RFNNest": (2665.9e9, 6324.5e6, 512)}
MODEL_FLOPS_TRAIN = {k: v[0] for k, v in MODEL_WORK_TRAIN.items() if v[2] == 256}
# forward only: (GFLOP, MB bf16) per image pair at the model's side; scaled by the pixel count for other frames
MODEL_WORK_INFER = {"PFNetv1": (35.69e9, 122.0e6, 256), "DenseFuse": (11.53e9, 80.1e6, 256), "NestFuse": (609.5e9, 1335.6e6, 512), "RFNNest": (888.6e9, 2108.2e6, 512)}
PEAK_MFMA_BF16 = 2.5e15   # dense, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_MATRIX_FP32 = 157e12
PEAK_HBM = 8.0e12


def ideal_pairs_per_s(model, h, w, dtype, products=1, mode="train"):
    """SURVEY 8(d)'s "ideal pairs/s/GPU": 1 / max(t_MFMA, t_HBM) of the step's algorithmic flops and bytes at the peaks above.  fp32
    storage doubles the bytes; `products` = half-precision MFMA products the path issues per algorithmic product (3 on the split-operand
    fp32 path), priced at the bf16 matrix peak"""
    table = MODEL_WORK_TRAIN if mode == "train" else MODEL_WORK_INFER
    if model not in table:
        return None
    fl, by, side = table[model]
    scale = (h * w) / float(side * side)
    return 1.0 / max(products * fl * scale / PEAK_MFMA_BF16, by * scale * (2 if dtype == "fp32" else 1) / PEAK_HBM)


def settle_gc():
    """Before a timed region: collect, then move everything alive into the permanent generation (gc.freeze).  Round 6 found a ~75-150 ms pause of
    the host inside one default run in three: CPython's full (generation-2) collection walking the ~million container objects torch's import
    leaves behind, triggered by allocation count -- at a fixed step of the loop (step 14 of the headline leg), i.e. while the host is only
    ~35 ms ahead of the GPU, which then idles: 7.0-7.6 k instead of 8.6-9.0 k pairs/s with identical kernel times (tools/diag_stall*.py;
    DESIGN.md section 4.2).  The collector stays ON; it just no longer re-walks start-up objects.  train.py does the same after its set-up."""
    gc.collect()
    gc.freeze()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.4 s of timed steps after 0.1 s of warm-up (30 / 5 measured 1 % lower on the same box: the clocks are still settling)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
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
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` block (BASELINE configs 3, 4, 5 on this GPU)")
    ap.add_argument("--other-steps", type=int, default=20)
    ap.add_argument("--other-warmup", type=int, default=6)
    ap.add_argument("--roofline-layer", default="decode.0", help="engine layer whose forward / dgrad / wgrad launches are timed with HIP events; "
                    "`roofline` reports the one with the largest share of the step, `roofline_kernels` all three")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend for N > 1 (nccl = RCCL; gloo: debugging)")
    ap.add_argument("--one-device", action="store_true", help="debug: every rank on cuda:0 (gloo only) -- exercises the N > 1 plumbing on a 1-GPU box")
    ap.add_argument("--hbm-tag", default="auto", help="an HBM-bound engine op timed the same way for `roofline_hbm` ('' = none)")
    ap.add_argument("--roofline-every", type=int, default=0, help="time the roofline launches on every N-th timed step (0 = auto: five samples "
                    "per kernel, every step when --steps <= 5).  Each HIP event pair around a launch idles the GPU for ~10 us (a marker packet "
                    "before and after: 4 timed launches per step were ~45 us = 1.2 %% of the step the events are there to describe)")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(model_name, size, n_pairs, budget_s=24.0):
    """The torch-CPU restatement of the SAME train step (oracle/torch_cpu_step.py: stock torch CPU ops, pinned to the reference's goldens
    F5 / F6 by tests/test_torch_cpu_step.py) on this box's host cores with min(32, cores) intra-op threads (128 threads on a small
    conv step measure the thread pool, not the cores: round 3's one-step sample moved +-70 % run to run): one untimed step, then at
    least 3 timed steps of n_pairs pairs (more while budget_s lasts, at most 8); `value` is the BEST step, the median is beside it."""
    from oracle import torch_cpu_step as TC
    if model_name not in ("PFNetv1", "DenseFuse"):
        return None
    prev_threads = torch.get_num_threads()
    threads = max(1, min(32, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    try:
        m = TC.TorchCpuModel(model_name)
        P = m.init_params(0)
        opt = TC.make_optimizer(P)
        g = torch.Generator().manual_seed(0)
        shape = (n_pairs, 1, size, size)
        i1, i2 = torch.rand(shape, generator=g), torch.rand(shape, generator=g)
        TC.train_step(m, P, opt, i1, i2)   # thread pool / allocator warm-up at the timed shape
        times, t_start = [], time.time()
        while len(times) < 3 or (len(times) < 8 and time.time() - t_start < budget_s):
            t0 = time.time()
            TC.train_step(m, P, opt, i1, i2)
            times.append(time.time() - t0)
        ts = sorted(times)
        med = ts[len(ts) // 2]
        return {"value": n_pairs / ts[0], "unit": "image-pairs/s", "cores": threads, "kind": "port", "median_value": n_pairs / med,
                "threads_note": "min(32, os.cpu_count()) intra-op threads: on an 8-pair 256x256 step torch's CPU convolutions stop scaling there -- with the box's "
                                "full 128 / 256 threads round 3's sample measured the thread pool (+-70 % run to run), not the cores",
                "steps": len(times), "step_seconds": [round(t, 3) for t in times],
                "sample": f"best of {len(times)} train steps of {model_name} on {n_pairs} pairs {size}x{size} fp32 (one untimed warm-up step first), "
                          f"torch {torch.__version__} CPU ops (oracle/torch_cpu_step.py), {sum(times):.1f} s; torch.set_num_threads({threads}), "
                          f"os.cpu_count() = {os.cpu_count()}, CPU: {_cpu_model()}"}
    finally:
        torch.set_num_threads(prev_threads)


def lib_sha256():
    import hashlib
    from mmif import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def measured_traffic(tag, match):
    """HBM bytes per launch of the kernel behind `tag` from the TCC counter passes on file (profiles/r06_traffic.json, written by
    tools/make_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command) -- only when those passes ran on
    THIS library build (sha256 of the .so) and this workload; otherwise (None, reason)."""
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
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
    return ent, f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, lib {tj['lib_sha256'][:12]})"


TRAFFIC_FILE = "r06_traffic.json"
KIND_KERNEL_BF16 = {"fwd": "conv_dma_kernel<false>", "dgrad": "conv_dma_kernel<true> (ReLU sign bytes, in-tile reflect fold)",
                    "wgrad": "wgrad_dma_kernel + wgrad_dma_reduce"}
KIND_KERNEL_FP32 = {"fwd": "conv_x3_kernel<fwd>", "dgrad": "conv_x3_kernel<dgrad> + fold", "wgrad": "wgrad_x3_kernel + wgrad_x3_reduce"}


def conv_rooflines(T, engine, layer, B, S, Wd, dtype, step_ms, workload_id=None):
    """`roofline` objects of the forward / dgrad / wgrad launches of one conv layer from the HIP events bench recorded inside the timed
    region (T.PROFILE_EVENTS): ALGORITHMIC flops (2 B H W Cin Cout k^2 per pass) over the average launch duration.  bf16:
    against the dense bf16 MFMA peak.  fp32 tensors (split-operand kernels): the EXECUTED MFMA rate (products_per_tap x the algorithmic
    flops) against the same bf16 peak, the algorithmic fp32 rate as a separate field."""
    spec = [s for s in engine.specs if s.name == layer]
    if not spec:
        return {}
    sp = spec[0]
    flops = 2.0 * B * S * Wd * sp.cin * sp.cout * sp.k * sp.k
    out = {}
    for kind in ("fwd", "dgrad", "wgrad"):
        evs = T.PROFILE_EVENTS.get(f"{layer}:{kind}", [])
        if not evs:
            continue
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        ach = flops / (ms * 1e-3)
        r = {"bound": "mfma", "kernel": f"{(KIND_KERNEL_FP32 if dtype == 'fp32' else KIND_KERNEL_BF16)[kind]} {sp.cin}->{sp.cout} k{sp.k} ({layer}:{kind})",
             "achieved": ach / 1e12, "peak": PEAK_MFMA_BF16 / 1e12, "unit": "TFLOP/s",
             "frac": ach / PEAK_MFMA_BF16, "avg_launch_ms": ms, "launches": len(evs),
             "step_share": ms / step_ms if step_ms else None, "traffic": None,
             "algorithmic_bytes": float(B) * S * Wd * (sp.cin + sp.cout) * (2 if dtype == "bf16" else 4),
             "clock_note": "peak is at the nominal 2.4 GHz; s_memtime cycles against wall time put this kernel family at ~1.65 GHz on Gaussian data "
                           "(power budget) with the matrix pipe busy ~0.65 of the cycles: profiles/r06_ubench_conv_clock.txt, DESIGN.md 4.2"}
        if dtype == "fp32":
            # the split-operand kernels issue products_per_tap half-precision MFMA products per algorithmic fp32 product: `achieved` / `frac`
            # are the EXECUTED matrix work against the dense bf16 MFMA peak (the pipe these kernels run on); the algorithmic fp32 rate is
            # beside it (gfx950's own fp32 matrix rate is 157 TFLOP/s -- these kernels beat it, which is their point, not a roofline)
            from mmif._lib import lib as _l
            prods = {16: 3, 3: 6, 2: 3}[_l.mmif_get_x3_forward_pieces()] if kind == "fwd" else 3
            r["products_per_tap"] = prods
            r["algorithmic_fp32_tflops"] = ach / 1e12
            r["achieved"] = r["executed_mfma_tflops"] = ach * prods / 1e12
            r["frac"] = r["executed_frac_of_bf16_mfma_peak"] = ach * prods / PEAK_MFMA_BF16
            r["peak_note"] = "achieved = executed half-precision MFMA work (products_per_tap x the algorithmic fp32 flops); peak = dense bf16 MFMA"
        if workload_id is not None:
            # HBM bytes per launch from the TCC PMC passes of the same command (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs,
            # gfx950 correction applied, tools/prof_pmc.sh + tools/make_traffic.py) -- only when taken on THIS library build
            ent, src = measured_traffic(f"{layer}:{kind}", workload_id)
            r["traffic"] = ent["hbm_bytes_per_launch"] if ent else None
            r["traffic_source"] = src
        out[kind] = r
    return out


def oracle_error_sample(model_name, shape, dev, losses):
    """one train step's fused image, total loss and every parameter gradient on the current compute dtype against the CPU oracle"""
    import numpy as np
    import core.model as M
    from mmif.optim import FusedClipAdam
    from oracle import fusion_oracle as O
    l_ssim, l_pix, l_grad = losses
    i1n, i2n = O.closed_form_image(shape, 0.3), O.closed_form_image(shape, 1.7)
    om = O.MODELS[model_name]()
    P = om.init_params(seed=1)
    ref = O.train_step(om, P, O.AdamState(P), i1n, i2n, clip=None)
    m = getattr(M, model_name)()
    m.load_state_dict({k: torch.from_numpy(O.closed_form_param(i, k, tuple(v.shape), 1)) for i, (k, v) in enumerate(m.state_dict().items())})
    m = m.to(dev)
    o2 = FusedClipAdam(m.parameters(), lr=1e-4, max_norm=0.0)
    i1, i2 = torch.from_numpy(i1n).to(dev), torch.from_numpy(i2n).to(dev)
    o2.zero_grad(set_to_none=True)
    f = m(i1, i2)
    tot = l_ssim(i1, i2, f) + l_pix(i1, i2, f, mode='max') + l_grad(i1, i2, f, mode='max')
    tot.backward()
    torch.cuda.synchronize()
    assert float(np.abs(ref["imgf"]).max()) > 0
    err_img = float(np.abs(f.detach().cpu().numpy() - ref["imgf"]).max() / np.abs(ref["imgf"]).max())
    gerr = max(float(np.abs(p.grad.cpu().numpy() - ref["grads"][k]).max() / max(np.abs(ref["grads"][k]).max(), 1e-12)) for k, p in m.named_parameters())
    return {"shape": "x".join(str(v) for v in (shape[0],) + shape[2:]), "rel_err_vs_oracle": err_img,
            "loss_abs_err_vs_oracle": abs(float(tot.item()) - float(ref["losses"][3])), "grad_rel_err_vs_oracle": gerr}


def parity_leg(args, dev, img1, img2):
    """The same train step on the parity-grade path: fp32 feature maps, 3x3 / 1x1 layers on the matrix pipe as split-operand products (three
    per tap: fp16 pieces forward, bf16 pieces backward; csrc/conv_x3.hip), everything else fp32.  Timed like the main leg, with the same
    per-kernel HIP events on --roofline-layer; its error is taken against the CPU oracle on two small closed-form samples (2 pairs of
    64 x 64 and one ragged 37 x 53 pair: fused image, total loss, every parameter gradient)."""
    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
    from mmif import engine as E
    from mmif import tensor as T
    from mmif._lib import lib
    from mmif.optim import FusedClipAdam
    prev = E.compute_dtype()
    E.set_compute_dtype("fp32")
    try:
        l_ssim, l_pix, l_grad = SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev)
        l_all = None if args.separate_losses else FusionLoss(l_ssim, l_pix, l_grad, 'max', 'max')

        def one(model, opt, a, b):
            opt.zero_grad(set_to_none=True)
            f = model(a, b)
            tot = l_all(a, b, f) if l_all is not None else l_ssim(a, b, f) + l_pix(a, b, f, mode='max') + l_grad(a, b, f, mode='max')
            tot.backward(unit_gradient(tot))
            opt.step()
            return f, tot
        torch.manual_seed(0)
        model = getattr(M, args.model)().to(dev)
        opt = FusedClipAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), max_norm=5.0)
        for _ in range(6):     # (warm-up as the main leg's: packing, leases, the clock's ramp after the idle oracle / setup phase)
            one(model, opt, img1, img2)
        tags = {f"{args.roofline_layer}:{k}" for k in ("fwd", "dgrad", "wgrad")}
        every = args.roofline_every if args.roofline_every > 0 else max(1, args.parity_steps // 5)
        T.prealloc_events(tags, (args.parity_steps + every - 1) // every + 2)
        T.PROFILE_EVENTS.clear()
        settle_gc()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.parity_steps):
            T.PROFILE_TAGS = tags if i % every == 0 else set()   # (as the main leg: events on a sample of the timed steps)
            one(model, opt, img1, img2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.unfreeze()
        T.PROFILE_TAGS = set()
        B, S, Wd = img1.shape[0], img1.shape[2], img1.shape[3]
        value = B * args.parity_steps / dt
        roofs = conv_rooflines(T, model._engine, args.roofline_layer, B, S, Wd, "fp32", dt / args.parity_steps * 1e3)
        T.PROFILE_EVENTS.clear()
        del model, opt
        samples = [oracle_error_sample(args.model, shp, dev, (l_ssim, l_pix, l_grad)) for shp in ((2, 1, 64, 64), (1, 1, 37, 53))]
        mode = lib.mmif_get_x3_forward_pieces()
        fwd = {16: "3 products of scaled fp16 pieces", 3: "6 products of bf16 pieces", 2: "3 products of bf16 pieces"}[mode]
        ideal = ideal_pairs_per_s(args.model, S, Wd, "fp32", products=3)
        dom = max(roofs.values(), key=lambda r: r["step_share"]) if roofs else None
        return {"dtype": f"fp32 storage; 3x3 / 1x1 layers as split-operand MFMA products (forward: {fwd}; backward: 3 products of bf16 pieces), fp32 accumulate",
                "value": value, "unit": "image-pairs/s", "ms_per_step": dt / args.parity_steps * 1e3, "steps": args.parity_steps,
                "step_frac_of_ideal": value / ideal if ideal else None, "ideal_pairs_per_s_per_gpu": ideal,
                "ideal_note": "ideal = SURVEY 8(d) 1 / max(t_MFMA, t_HBM) with 3 half-precision products per algorithmic product at the bf16 MFMA peak and fp32 bytes",
                "roofline": dom, "roofline_kernels": roofs,
                "rel_err_vs_oracle": max(e["rel_err_vs_oracle"] for e in samples), "loss_abs_err_vs_oracle": max(e["loss_abs_err_vs_oracle"] for e in samples),
                "grad_rel_err_vs_oracle": max(e["grad_rel_err_vs_oracle"] for e in samples), "oracle_samples": samples,
                "oracle_sample": "2 pairs 64x64 and 1 pair 37x53, closed-form weights / images: max |fused image - oracle| / max|oracle|; max over parameters of max |grad - oracle| / max|oracle grad| (worst of the two samples)",
                "tolerance": "BASELINE north star: 1e-3 relative"}
    finally:
        E.set_compute_dtype(prev)


# the other BASELINE.json configs, timed by the same process after the headline leg (verdict r5 item 1d): (key, model, mode, batch, H, W, config)
OTHER_CONFIGS = (("DenseFuse_b32_256", "DenseFuse", "train", 32, 256, 256, "configs[2] per-GPU share: DenseFuse 256x256 bf16, batch 32 per GPU"),
                 ("NestFuse_b4_512", "NestFuse", "train", 4, 512, 512, "configs[3]: NestFuse 512x512 bf16, batch 4 per GPU"),
                 ("RFNNest_b4_512", "RFNNest", "train", 4, 512, 512, "configs[3]: RFN-Nest 512x512 bf16, batch 4 per GPU"),
                 # (batch scaling of the same build: BASELINE's config 4 names no batch; at 4 pairs the 128 x 128 and 64 x 64 levels give a launch fewer
                 # tiles than the chip has CUs)
                 ("NestFuse_b8_512", "NestFuse", "train", 8, 512, 512, "configs[3] at batch 8 per GPU (batch scaling): NestFuse 512x512 bf16"),
                 ("RFNNest_b8_512", "RFNNest", "train", 8, 512, 512, "configs[3] at batch 8 per GPU (batch scaling): RFN-Nest 512x512 bf16"),
                 ("infer_1224x1024", "PFNetv1", "infer", 1, 1024, 1224, "configs[4]: PFNetv1 forward (no_grad, test.py path) on one 1224x1024 pair, bf16"))


OTHER_MIN_TIMED_S = 0.2
OTHER_MAX_STEPS = 400


def other_configs_leg(args, dev):
    """BASELINE.json's configs 3, 4 and 5 on this GPU, each at least `--other-steps` timed steps after at least `--other-warmup` untimed ones (both
    stretched to >= 0.2 s of timed work for sub-millisecond steps; device synchronised on both sides), bf16 feature maps, through the same drop-in API and the same step function as the headline leg: value
    (image-pairs/s), ms_per_step and the fraction of SURVEY 8(d)'s ideal.  About 2 s of GPU time in total."""
    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
    from mmif import engine as E
    from mmif.optim import FusedClipAdam
    prev = E.compute_dtype()
    E.set_compute_dtype("bf16")
    res = {}
    try:
        for key, name, mode, B, H, Wd, what in OTHER_CONFIGS:
            torch.manual_seed(0)
            model = getattr(M, name)().to(dev)
            gen = torch.Generator(device="cpu").manual_seed(0)
            a = torch.rand(B, 1, H, Wd, generator=gen).to(dev)
            b = torch.rand(B, 1, H, Wd, generator=gen).to(dev)
            if mode == "train":
                opt = FusedClipAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), max_norm=5.0)
                l_all = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev), 'max', 'max')

                def one():
                    opt.zero_grad(set_to_none=True)
                    f = model(a, b)
                    tot = l_all(a, b, f)
                    tot.backward(unit_gradient(tot))
                    opt.step(scalars=l_all.values)
                    return tot
            else:
                def one():
                    with torch.no_grad():
                        return model(a, b).mean()
            for _ in range(args.other_warmup):
                one()
            # a leg of sub-millisecond steps (config 5) is over before the clocks have settled: probe the step time and stretch both
            # the warm-up and the timed region to at least OTHER_MIN_TIMED_S of work (the step counts used are reported)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            probe = (time.perf_counter() - t0) / 3
            steps = min(OTHER_MAX_STEPS, max(args.other_steps, int(OTHER_MIN_TIMED_S / probe) + 1))
            warm = args.other_warmup + 3
            for _ in range(max(0, steps // 3 - warm)):
                one()
                warm += 1
            settle_gc()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tot = one()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            gc.unfreeze()
            value = B * steps / dt
            ideal = ideal_pairs_per_s(name, H, Wd, "bf16", 1, mode)
            res[key] = {"config": what, "model": name, "mode": mode, "batch": B, "height": H, "width": Wd, "dtype": "bf16", "value": value,
                        "unit": "image-pairs/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warm,
                        "step_frac_of_ideal": value / ideal if ideal else None, "ideal_pairs_per_s_per_gpu": ideal, "final_value": float(tot.item())}
            del model, a, b
            if mode == "train":
                del opt, l_all
    finally:
        E.set_compute_dtype(prev)
    return res


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher environment: start the N ranks as a CHILD torch.distributed.run -- this
    process has not initialised the GPU (torch.cuda.device_count() does not) and never will -- relay the output, return its code.
    The rendezvous port is one the kernel just handed out (bind to port 0)."""
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not (args.one_device and args.backend == "gloo"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this box has {have} visible GPU(s); refusing to report a {have}-GPU run as {args.gpus}")
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.one_device and args.backend != "gloo":
        raise SystemExit("bench.py: --one-device is a debugging aid for --backend gloo (RCCL wants one device per rank)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} rank(s)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        # stdout belongs to rank 0's JSON line: whatever the other ranks (or the libraries in them: RCCL's banner goes through C stdio and
        # sits in libc's buffer until exit) print goes to stderr from here on
        sys.stdout.flush()
        os.dup2(2, 1)
    # $MMIF_FORCE_DIST=1: take the RCCL code path (init, parameter broadcast, gradient all-reduce, barriers) with one rank too
    use_dist = world > 1 or os.environ.get("MMIF_FORCE_DIST", "0") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
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
    # manual_seed(0); img1, img2 = rand(global batch, 1, H, W) in that order; rank r takes the slice
    # [r B, (r + 1) B) of the global batch (the DistributedSampler split of train.py:204-209 on synthetic data)
    gen = torch.Generator(device="cpu").manual_seed(0)
    img1 = torch.rand(B * world, 1, S, Wd, generator=gen)[rank * B:(rank + 1) * B].to(dev)  # synthetic IR / visible pairs in [0,1)
    img2 = torch.rand(B * world, 1, S, Wd, generator=gen)[rank * B:(rank + 1) * B].to(dev)

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
        tot.backward(unit_gradient(tot))   # (= tot.backward(); core/loss.py: a cached ones tensor, as train.py does)
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
    tags = {f"{args.roofline_layer}:{k}" for k in ("fwd", "dgrad", "wgrad")} | ({hbm_tag} if hbm_tag else set())
    if hbm_tag == "encode:fwd" and args.mode == "train":
        tags |= {"encode:bwd"}      # the fused encoder backward (csrc/enc_bwd.hip), when the engine takes it
    every = args.roofline_every if args.roofline_every > 0 else max(1, args.steps // 5)
    # the sampled steps' events exist before the clock starts (created and recorded once), and one untimed step runs the sampled form of the
    # backward (wgrad and dgrad of the wide layers as two calls): nothing is created or first-used inside the timed region
    T.prealloc_events(tags, (args.steps + every - 1) // every + 2)
    T.PROFILE_TAGS = tags
    step()
    T.PROFILE_TAGS = set()
    for evs in T.PROFILE_EVENTS.values():     # (the untimed sampled step's events go back to the pool)
        for pair in evs:
            T.PROFILE_EVENT_POOL.setdefault("_spare", []).append(pair)
    T.PROFILE_EVENTS.clear()
    settle_gc()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    diag = os.environ.get("BENCH_DIAG", "0") == "1"       # (diagnostics: host stamps per step + a GPU event every tenth step)
    if diag:
        dev_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps // 10 + 1)]
        for e in dev_ev:
            e.record()
        torch.cuda.synchronize()
        hts = [time.perf_counter()]
        t0 = hts[0]
    for i in range(args.steps):
        if diag and i % 10 == 0:
            dev_ev[i // 10].record()
        T.PROFILE_TAGS = tags if i % every == 0 else set()   # (live, inside the timed region -- on a sample of its steps)
        tot = step()
        if diag:
            hts.append(time.perf_counter())
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    if diag:
        hd = [(hts[k + 1] - hts[k]) * 1e3 for k in range(args.steps)]
        gd = [dev_ev[k].elapsed_time(dev_ev[k + 1]) for k in range(len(dev_ev) - 1)]
        print(f"DIAG ms/step {dt / args.steps * 1e3:.3f}; GPU ms per 10 steps {[round(v, 1) for v in gd]}; host worst {sorted([(round(h, 1), k) for k, h in enumerate(hd)])[-4:]}", file=sys.stderr)
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
        # the wide layer's three launches, timed live (HIP events on the launch stream, inside the timed region); `roofline` is the one
        # with the largest share of the step
        roofs = conv_rooflines(T, model._engine, args.roofline_layer, B, S, Wd, args.dtype, dt / args.steps * 1e3, workload_id) if getattr(model, "_engine", None) is not None else {}
        roof = max(roofs.values(), key=lambda r: r["step_share"]) if roofs else None
        roof_hbm = None
        evh = T.PROFILE_EVENTS.get(hbm_tag, []) if hbm_tag else []
        if evh:
            ms = sum(a.elapsed_time(b) for a, b in evh) / len(evh)
            spec = [s for s in model._engine.specs if hbm_tag.startswith(s.name + ":")]
            if hbm_tag == "encode:fwd":
                # algorithmic bytes of the 2 x 4 conv passes it replaces (H*W*(Cin+Cout)*sizeof per pass, the image in fp32)
                # -- the launch itself only moves the images and the output planes (`fused_bytes`): achieved / frac are the bytes the launch
                # REALLY moves (the counter figure when one is on file for this build, else fused_bytes) over its duration; the rate in
                # terms of the layer-wise passes it replaces is reported apart (it is not a bandwidth and may exceed the peak)
                nbr = 2 if args.model in ("PFNetv1", "VIFNet", "DenseFuse", "PFNetv2") else 1
                per_px = nbr * ((1 * 4 + 16 * 2) + (16 + 16) * 2 + (32 + 16) * 2 + (48 + 16) * 2)
                nbytes = float(B) * S * Wd * per_px
                fused = float(B) * S * Wd * nbr * (4 + 64 * 2)
                ent, src = measured_traffic(hbm_tag, workload_id)
                moved = ent["hbm_bytes_per_launch"] if ent else fused
                roof_hbm = {"bound": "hbm", "kernel": f"enc_stream2_fwd_kernel {nbr} x (1->16, 16->16, 32->16, 48->16) k3 (encode:fwd)",
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
        roof_hbm_bwd = None
        evb = T.PROFILE_EVENTS.get("encode:bwd", [])
        if evb:
            # bytes the launch moves: per branch the incoming gradient G (64 planes) and x0..x2 (48 planes) in bf16, the fp32 image; nothing
            # written but one partial sum per block.  The two launches it replaces moved 2.4 GB
            ms = sum(a.elapsed_time(b) for a, b in evb) / len(evb)
            nbr = 2
            moved_alg = float(B) * S * Wd * nbr * (4 + (64 + 48) * 2)
            ent, src = measured_traffic("encode:bwd", workload_id)
            moved = ent["hbm_bytes_per_launch"] if ent else moved_alg
            roof_hbm_bwd = {"bound": "hbm", "kernel": "enc_bwd_fused_kernel 2 x (gradient chain + dW, db of 1->16, 16->16, 32->16, 48->16) (encode:bwd)",
                            "achieved": moved / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": moved / (ms * 1e-3) / PEAK_HBM,
                            "avg_launch_ms": ms, "launches": len(evb), "traffic": ent["hbm_bytes_per_launch"] if ent else None, "traffic_source": src,
                            "algorithmic_bytes": moved_alg,
                            "note": "LDS / issue bound, not bandwidth bound: 133 MFMA per 32-pixel row step and wave pair (DESIGN section 4)"}
        ideal = ideal_pairs_per_s(args.model, S, Wd, args.dtype, 3 if args.dtype == "fp32" else 1) if args.mode == "train" else None
        out = {
            "metric": "image-pairs/sec at 256x256, PFNet train step" if args.mode == "train" else f"image-pairs/sec at {Wd}x{S}, {args.model} inference", "value": value, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "graph": bool(args.graph and args.mode == "train"),
            "config": {"workload": (f"{args.model} train step (fwd + SSIM/pixel/grad losses + bwd + clip + Adam)" if args.mode == "train" else f"{args.model} forward (no_grad, test.py path)") + f", {Wd}x{S} synthetic IR/visible pairs, "
                                   f"batch {B} per GPU, {args.dtype} feature maps / fp32 accumulate, random-init weights (seed 0)",
                       "global_batch": B * world, "parallelism": f"dp{world}" if world > 1 else "single",
                       "backend": (args.backend if use_dist else None),
                       "step_model_tflops": MODEL_FLOPS_TRAIN.get(args.model, 0) * value / 1e12 if (S == 256 and Wd == 256 and args.mode == "train") else None},
            "final_loss": loss,
            # value per GPU over SURVEY 8(d)'s ideal pairs/s/GPU = 1 / max(t_MFMA, t_HBM) of the step's algorithmic flops / bytes at the peaks
            "step_frac_of_ideal": (value / world) / ideal if ideal else None,
            "ideal_pairs_per_s_per_gpu": ideal,
            "roofline": roof,
            "roofline_kernels": roofs or None,
            "roofline_sampling": {"every_nth_timed_step": every, "timed_steps": args.steps,
                                  "note": "HIP events around the named launches, recorded inside the timed region on every n-th step"},
            "roofline_hbm": roof_hbm,
            "roofline_hbm_bwd": roof_hbm_bwd,
            "parity_path": None,
            "other_configs": None,
            "cpu_baseline": None,
        }
        assert out["n_gpus"] == args.gpus
        if world == 1 and not args.no_parity_path and args.mode == "train" and args.dtype == "bf16" and args.model in ("PFNetv1", "DenseFuse", "PFNetv2", "VIFNet"):
            out["parity_path"] = parity_leg(args, dev, img1, img2)
        headline = (args.model, args.mode, args.dtype, B, S, Wd) == ("PFNetv1", "train", "bf16", 32, 256, 256)
        if world == 1 and not args.no_other_configs and headline and not args.graph:
            del tot
            out["other_configs"] = other_configs_leg(args, dev)
        if world == 1 and not args.no_cpu_baseline and args.mode == "train" and Wd == S:
            out["cpu_baseline"] = cpu_baseline(args.model, S, args.cpu_sample)
    else:
        out = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: ranks other than 0 have had their stdout pointed at stderr since start-up, and rank 0
    # empties Python's and libc's buffers (RCCL's banner goes through C stdio) before it prints
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
