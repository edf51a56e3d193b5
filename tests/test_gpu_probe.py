"""Silicon probes: check the lane mappings conv_mfma.hip assumes for ds_read_b64_tr_b16 and
v_mfma_f32_16x16x32_bf16 against the hardware itself; dump what was observed to gpurun_out/."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _out_dir():
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d


def test_tr16_lane_mapping():
    from mmif._lib import check, lib
    perm = torch.arange(64, dtype=torch.int32, device="cuda:0")
    out = torch.zeros(256, dtype=torch.int16, device="cuda:0")
    check(lib.mmif_probe_tr16(C.c_void_p(perm.data_ptr()), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(64, 4).astype(np.int64)
    json.dump(got.tolist(), open(os.path.join(_out_dir(), "probe_tr16.json"), "w"))
    # hypothesis H1 (conv_mfma.hip wgrad): lane l supplies elements [4l, 4l+4); inside each 16-lane
    # group G these form a 4x16 row-major matrix (row r = lanes 4r..4r+3); lane s receives column s:
    # element j = M[j][s] = 64G + 16j + s
    l = np.arange(64)
    want = np.stack([64 * (l >> 4) + 16 * j + (l & 15) for j in range(4)], axis=1)
    assert (got == want).all(), f"ds_read_b64_tr_b16 mapping differs from H1:\n{got[:16]}"


def test_mfma_16x16x32_bf16_layout():
    from mmif._lib import check, lib
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, size=(16, 32)).astype(np.float32)      # asymmetric small integers: exact in bf16
    B = rng.integers(-4, 5, size=(32, 16)).astype(np.float32)
    a = torch.from_numpy(A).to("cuda:0").bfloat16().contiguous()
    b = torch.from_numpy(B).to("cuda:0").bfloat16().contiguous()
    out = torch.zeros(16, 16, dtype=torch.float32, device="cuda:0")
    check(lib.mmif_probe_mfma(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    np.save(os.path.join(_out_dir(), "probe_mfma.npy"), got)
    assert np.array_equal(got, A @ B), "v_mfma_f32_16x16x32_bf16 operand / result lane mapping differs from the assumed one"


def test_lds_dma_lane_mapping():
    """global_load_lds_dwordx4: lane l of a wave writes its (arbitrary) source granule to LDS at the wave-uniform
    base + 16*l -- the destination rule the DMA-staged conv kernel relies on."""
    from mmif._lib import check, lib
    rng = np.random.default_rng(1)
    src = torch.arange(4096 * 4, dtype=torch.int32, device="cuda:0").reshape(4096, 4)   # granule g = [4g, 4g+1, 4g+2, 4g+3]
    idx_np = rng.permutation(4096)[:256].astype(np.int32)
    slot_np = np.array([2, 0, 3, 1], dtype=np.int32)
    idx, slot = torch.from_numpy(idx_np).cuda(), torch.from_numpy(slot_np).cuda()
    out = torch.zeros(256, 4, dtype=torch.int32, device="cuda:0")
    check(lib.mmif_probe_dma(C.c_void_p(src.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(slot.data_ptr()),
                             C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = np.zeros((256, 4), dtype=np.int64)
    for w in range(4):
        for l in range(64):
            want[slot_np[w] * 64 + l] = 4 * idx_np[64 * w + l] + np.arange(4)
    assert np.array_equal(got, want), f"LDS-DMA destination mapping differs:\n{got[:8]}\n{want[:8]}"
