"""The torch-CPU restatement that bench.py times as `cpu_baseline` (oracle/torch_cpu_step.py) against the reference's golden vectors:
F5 (forward + every parameter gradient of PFNetv1 / DenseFuse) and F6 (3-step train trajectories: losses, pre-clip gradient norm,
final weights).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from oracle import torch_cpu_step as TC
from gpu_util import G, close, close_digest


def _load(model, seed):
    P = model.init_params(0)
    with torch.no_grad():
        for i, (k, v) in enumerate(P.items()):
            v.copy_(torch.from_numpy(O.closed_form_param(i, k, tuple(v.shape), seed)))
    return P


@pytest.mark.parametrize("name,shape", [("PFNetv1", (2, 1, 32, 32)), ("DenseFuse", (2, 1, 32, 32)), ("PFNetv1", (1, 1, 37, 53))])
def test_torch_cpu_models_vs_golden_f5(name, shape):
    ref = np.load(os.path.join(G, "f5_models.npz"))
    man = json.load(open(os.path.join(G, "f5_manifest.json")))
    tag = f"{name}_{shape[0]}x{shape[2]}x{shape[3]}"
    m = TC.TorchCpuModel(name)
    assert [[k, list(s)] for k, s in m.shapes.items()] == man[name]     # the reference's state_dict keys and shapes
    P = _load(m, 1)
    i1, i2 = torch.from_numpy(O.closed_form_image(shape, 0.3)), torch.from_numpy(O.closed_form_image(shape, 1.7))
    y = m.forward(P, i1, i2)
    close(y.detach().numpy(), ref[tag + "__y"], 5e-5, "y")
    y.backward(torch.from_numpy(O.closed_form_signed(shape, 0.9, 1.0)))
    for k, v in P.items():
        close_digest(v.grad.numpy(), ref[f"{tag}__dp_{k}"], 1e-4, k)


def load_live(model, name):
    """oracle.LIVE_PARAMS: the closed-form set whose final ReLU passes 30-60 % of the pixels (what golden F5 holds for the nested nets)"""
    P = model.init_params(0)
    with torch.no_grad():
        for i, (k, v) in enumerate(P.items()):
            v.copy_(torch.from_numpy(O.live_param(name, i, k, tuple(v.shape))))
    return P


@pytest.mark.parametrize("name,shape", [("NestFuse", (1, 1, 32, 32)), ("RFNNest", (1, 1, 32, 32)), ("NestFuse", (2, 1, 36, 44)),
                                        ("RFNNest", (2, 1, 36, 44))])
def test_torch_cpu_nested_models_vs_live_golden_f5(name, shape):
    """NestFuse / RFN-Nest (reference core/model.py:319-384) in the torch-CPU restatement: fused image + all 44 / 92 parameter
    gradients against the LIVE golden cases, incl. the odd pyramid 36x44 -> 18x22 -> 9x11 -> 4x5 (Upsample._pad)."""
    ref = np.load(os.path.join(G, "f5_models.npz"))
    man = json.load(open(os.path.join(G, "f5_manifest.json")))
    tag = f"{name}_{shape[0]}x{shape[2]}x{shape[3]}"
    m = TC.TorchCpuModel(name)
    assert [[k, list(s)] for k, s in m.shapes.items()] == man[name]
    P = load_live(m, name)
    i1, i2 = torch.from_numpy(O.closed_form_image(shape, 0.3)), torch.from_numpy(O.closed_form_image(shape, 1.7))
    y = m.forward(P, i1, i2)
    yr = ref[tag + "__y"]
    assert 0.3 <= float((yr > 0).mean()) <= 0.7
    close(y.detach().numpy(), yr, 5e-5, "y")
    y.backward(torch.from_numpy(O.closed_form_image(shape, 0.9)))
    for k, v in P.items():
        close_digest(v.grad.numpy(), ref[f"{tag}__dp_{k}"], 2e-4, k)


@pytest.mark.parametrize("name", ["PFNetv1", "DenseFuse"])
def test_torch_cpu_train_trajectory_vs_golden_f6(name):
    ref = np.load(os.path.join(G, "f6_traj.npz"))
    m = TC.TorchCpuModel(name)
    P = _load(m, 2)
    opt = TC.make_optimizer(P)
    shape = (4, 1, 64, 64)
    rows = ref[name + "__rows"]
    for step in range(3):
        i1, i2 = torch.from_numpy(O.closed_form_image(shape, 0.21 + step)), torch.from_numpy(O.closed_form_image(shape, 1.43 + step))
        r = TC.train_step(m, P, opt, i1, i2)
        if step == 0:
            close(r["imgf"].numpy(), ref[name + "__imgf0"], 5e-5, "imgf")
        np.testing.assert_allclose(list(r["losses"]) + [r["grad_norm"]], rows[step], rtol=2e-4, atol=2e-6, err_msg=f"step {step}")
    for k, v in P.items():
        close_digest(v.detach().numpy(), ref[f"{name}__w_{k}"], 2e-5, k)


def test_torch_cpu_init_matches_reference_statistics():
    """init_params: kaiming-normal std sqrt(2 / fan_in) on the ReLU layers, zero biases (core/block.py:101-118)"""
    m = TC.TorchCpuModel("PFNetv1")
    P = m.init_params(0)
    w = P["decode.0.layers.0.weight"].detach()
    assert abs(float(w.std()) - (2.0 / (128 * 9)) ** 0.5) < 2e-3
    assert all(float(v.abs().max()) == 0.0 for k, v in P.items() if k.endswith("bias"))
