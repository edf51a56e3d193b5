"""train.py end to end on synthetic patch pairs (reference train.py:151-343: loaders, model, three losses, clip + Adam, warm-up and
multi-step schedules, validation pass, checkpoints, log lines) -- as a subprocess, the way a user runs it."""
import glob
import os
import shutil
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multi-modal-image-fusion_amd")


@pytest.mark.parametrize("extra", [["--dtype", "fp32", "--warmup", "True"], ["--dtype", "bf16", "--graph", "True"]], ids=["fp32-warmup", "bf16-graph"])
def test_train_py_runs_on_synthetic_patches(extra):
    before = set(glob.glob(os.path.join(ROOT, "checkpoints", "*")))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "train.py", "--synthetic", "24", "--bs", "4", "--epoch", "2", "--lr", "1e-3"] + extra,
                       cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
    new = sorted(set(glob.glob(os.path.join(ROOT, "checkpoints", "*"))) - before)
    try:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert len(new) == 1, new
        log = open(os.path.join(new[0], "train.log")).read()
        assert "training done" in log and "epoch: 02, train loss:" in log, log[-1500:]
        losses = [float(l.split("train loss: ")[1].split(",")[0]) for l in log.splitlines() if "train loss:" in l and "valid loss" in l]
        assert len(losses) == 2 and all(0.0 < v < 2.0 for v in losses), losses
        sd = torch.load(os.path.join(new[0], "epoch_last.pth"), map_location="cpu")
        assert "decode.0.layers.0.weight" in sd and all(torch.isfinite(v).all() for v in sd.values())
        assert os.path.isfile(os.path.join(new[0], "train", "02.png")) and os.path.isfile(os.path.join(new[0], "valid", "02.png"))
    finally:
        for d in new:
            shutil.rmtree(d, ignore_errors=True)
