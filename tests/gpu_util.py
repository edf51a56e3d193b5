"""Helpers shared by the -m gpu parity tests (HIP path vs CPU oracle / golden fixtures)."""
import os

import numpy as np
import torch

from oracle import fusion_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


def tg(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def close(a, b, rtol, what=""):
    e = rel_err(a, b)
    assert e <= rtol, f"{what}: max err / max|ref| = {e:.3e} > {rtol:.1e}"
    return e


def digest(a):
    f = np.asarray(a, dtype=np.float32).reshape(-1)
    n = f.size
    idx = (np.arange(16) * max(1, n // 16)) % n
    return np.concatenate(([f.astype(np.float64).sum(), np.abs(f.astype(np.float64)).sum()],
                           f[:16] if n >= 16 else np.pad(f, (0, 16 - n)), f[idx])).astype(np.float64)


def close_digest(arr, dg, rtol, what=""):
    mine = digest(arr)
    arr = np.asarray(arr)
    scale = max(np.abs(arr).max(), 1e-12)
    n = arr.size
    assert abs(mine[0] - dg[0]) <= rtol * scale * max(1.0, np.sqrt(n)) * 4, (what, "sum", mine[0], dg[0])
    assert abs(mine[1] - dg[1]) <= rtol * max(dg[1], scale), (what, "abs-sum", mine[1], dg[1])
    assert np.abs(mine[2:] - dg[2:]).max() <= rtol * scale, (what, "samples", np.abs(mine[2:] - dg[2:]).max(), scale)


def load_closed_form(module, seed):
    sd = module.state_dict()
    module.load_state_dict({k: torch.from_numpy(O.closed_form_param(i, k, tuple(v.shape), seed)) for i, (k, v) in enumerate(sd.items())})
    return module


def reload_switches():
    """the engine caches its $MMIF_... A/B switches: call after changing one at run time"""
    from mmif import engine as E
    E.reload_switches()


class dtype_ctx:
    """with dtype_ctx('bf16', impl='mfma'): ... -- select the engine's storage dtype / kernel family."""

    def __init__(self, dtype, impl="auto"):
        self.dtype, self.impl = dtype, impl

    def __enter__(self):
        from mmif import engine as E
        self.prev = E.compute_dtype()
        self.prev_impl = os.environ.get("MMIF_CONV_IMPL")
        E.set_compute_dtype(self.dtype)
        os.environ["MMIF_CONV_IMPL"] = self.impl
        E.reload_switches()

    def __exit__(self, *a):
        from mmif import engine as E
        E.set_compute_dtype(self.prev)
        if self.prev_impl is None:
            os.environ.pop("MMIF_CONV_IMPL", None)
        else:
            os.environ["MMIF_CONV_IMPL"] = self.prev_impl
        E.reload_switches()
