"""Time the streaming encoder launch alone (needs a GPU):  python tools/bench_enc.py [B H W]
Runs the product library (layer-wise and streaming), then every ab/lib*.so variant ($MMIF_LIB) in a child process."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(B, H, W, modes):
    for p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")):
        sys.path.insert(0, p)
    import torch
    import core.model as M
    from mmif import engine as E, tensor as T
    E.set_compute_dtype("bf16")
    torch.manual_seed(0)
    m = M.PFNetv1().cuda()
    eng = E.PFNetv1Engine(m)
    i1, i2 = torch.rand(B, 1, H, W).cuda(), torch.rand(B, 1, H, W).cuda()
    (i1, i2), _, _, _, dtype, impl = eng.prepare((i1, i2))
    F = T.BT.alloc(B, 128, H, W, dtype, "cuda")
    br = [(eng.enc[0], i1, 0), (eng.enc[1], i2, 8)]
    GF = T.BT.alloc(B, 128, H, W, dtype, "cuda", halo=1, zero=True).as_folded()
    GF.buf.normal_()
    ws = eng.workspace(torch.device("cuda"))
    grads = [(torch.zeros_like(s.conv.weight), torch.zeros_like(s.conv.bias)) for s in eng.enc[0]]

    def fwd():
        eng.enc_fwd_all(br, F, dtype, impl)

    def wgrad():   # one branch
        T.dense_encoder_wgrad(i1, F.view(0, 6), GF.view(0, 8), grads, ws)

    from mmif._lib import lib
    for label, env in modes:
        lib.mmif_debug_set_enc_stream2(int(env.get("_ES2", "2")))
        env = {k: v for k, v in env.items() if not k.startswith("_")}
        os.environ.update(env)
        for name, fn in (("fwd (2 branches)", fwd), ("wgrad (1 branch)", wgrad)):
            if name.startswith("wgrad") and env.get("MMIF_ENC_STREAM") == "0":
                continue
            for _ in range(150):     # (the first launches of a process run ~15 % slower than the sustained rate: clock ramp)
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"{label:28s} {name:18s} {e0.elapsed_time(e1) / 200 * 1e3:8.1f} us", flush=True)
        for k in env:
            os.environ.pop(k)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, H, W = (int(a) for a in args[:3]) if len(args) >= 3 else (32, 256, 256)
    if "--child" in sys.argv:
        one(B, H, W, [("stream " + os.path.basename(os.environ.get("MMIF_LIB", "?")), {})])
    else:
        one(B, H, W, [("stream2 32 px (default)", {}), ("stream2 64 px", {"_ES2": "1"}), ("stream (round 2)", {"_ES2": "0"})])
        for lib in sorted(glob.glob(os.path.join(ROOT, "ab", "lib*.so"))):
            subprocess.run([sys.executable, __file__, str(B), str(H), str(W), "--child"], env=dict(os.environ, MMIF_LIB=lib))
