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


def rel_err(a, b, allow_zero=False):
    """max|a - b| / max|b|.  An all-zero reference pins nothing (0 == 0 passes whatever the kernels wrote elsewhere): it is refused
    unless the case is zero by design (allow_zero=True)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert allow_zero or b.size == 0 or np.abs(b).max() > 0.0, "all-zero reference: the comparison would be vacuous"
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def close(a, b, rtol, what="", allow_zero=False):
    try:
        e = rel_err(a, b, allow_zero)
    except AssertionError as ex:
        raise AssertionError(f"{what}: {ex}") from None
    assert e <= rtol, f"{what}: max err / max|ref| = {e:.3e} > {rtol:.1e}"
    return e


def digest(a):
    f = np.asarray(a, dtype=np.float32).reshape(-1)
    n = f.size
    idx = (np.arange(16) * max(1, n // 16)) % n
    return np.concatenate(([f.astype(np.float64).sum(), np.abs(f.astype(np.float64)).sum()],
                           f[:16] if n >= 16 else np.pad(f, (0, 16 - n)), f[idx])).astype(np.float64)


def close_digest(arr, dg, rtol, what="", allow_zero=False):
    assert allow_zero or np.abs(np.asarray(dg)).max() > 0.0, f"{what}: all-zero reference digest: the comparison would be vacuous"
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


def load_live(module, name):
    """the live closed-form parameter set of NestFuse / RFN-Nest (oracle.LIVE_PARAMS: the final ReLU passes 30-60 % of the pixels)"""
    sd = module.state_dict()
    module.load_state_dict({k: torch.from_numpy(O.live_param(name, i, k, tuple(v.shape))) for i, (k, v) in enumerate(sd.items())})
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
