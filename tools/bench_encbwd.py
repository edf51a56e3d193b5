"""Time the encoder backward launches alone (needs a GPU):  python tools/bench_encbwd.py [B H W]
The fused launch (csrc/enc_bwd.hip, two branches) beside the chain + 2 x weight-gradient launches it replaces, then every
ab/libmmif_eb_*.so ablation build ($MMIF_LIB) in a child process."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(B, H, W, label, fused_only):
    for p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")):
        sys.path.insert(0, p)
    import torch
    from mmif import tensor as T
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    br = []
    for b in range(2):
        xs = torch.randn(B, 64, H, W, generator=g).clamp_(min=0)
        G = torch.randn(B, 64, H, W, generator=g)
        ws = [torch.randn(16, 16 * (i + 1), 3, 3, generator=g) * (0.25 / (i + 1)) for i in range(3)]
        img = torch.rand(B, 1, H, W, generator=g).to(dev)
        F = T.BT.from_nchw(xs.to(dev), torch.bfloat16)
        GF = T.BT.from_nchw(G.to(dev), torch.bfloat16, halo=1).as_folded()
        pk = T.pack_dense_chain(*[t.to(dev) for t in ws], dev)
        shapes = [((16, 1, 3, 3), (16,)), ((16, 16, 3, 3), (16,)), ((16, 32, 3, 3), (16,)), ((16, 48, 3, 3), (16,))]
        grads = [(torch.zeros(a, device=dev), torch.zeros(b_, device=dev)) for a, b_ in shapes]
        out = T.BT.alloc(B, 64, H, W, torch.bfloat16, dev)
        br.append((F, GF, pk, img, grads, out))
    wsf = torch.empty(T.dense_encoder_bwd_workspace_bytes() // 4 + 1, dtype=torch.float32, device=dev)
    wsw = torch.empty(T.dense_encoder_wgrad_workspace_bytes() // 4 + 1, dtype=torch.float32, device=dev)

    def fused():
        T.dense_encoder_bwd([(GF.view(6, 2), GF.view(0, 6), F.view(0, 6), pk, img, grads, False) for F, GF, pk, img, grads, out in br], wsf)

    def chain():
        T.dense_encoder_chain([(GF.view(6, 2), GF.view(0, 6), F.view(0, 6), pk, out) for F, GF, pk, img, grads, out in br])

    def wgrads():
        for F, GF, pk, img, grads, out in br:
            T.dense_encoder_wgrad(img, F.view(0, 6), out, grads, wsw, False)

    for name, fn in (("fused (2 branches)", fused),) + (() if fused_only else (("chain (2 branches)", chain), ("wgrad x 2", wgrads))):
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{label:28s} {name:20s} {e0.elapsed_time(e1) / 100 * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, H, W = (int(a) for a in args[:3]) if len(args) >= 3 else (32, 256, 256)
    if "--child" in sys.argv:
        one(B, H, W, os.path.basename(os.environ.get("MMIF_LIB", "?")), True)
    else:
        one(B, H, W, "product library", False)
        for lib in sorted(glob.glob(os.path.join(ROOT, "ab", "libmmif_eb_*.so"))):
            subprocess.run([sys.executable, __file__, str(B), str(H), str(W), "--child"], env=dict(os.environ, MMIF_LIB=lib))
