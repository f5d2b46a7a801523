"""-m gpu: the whole training loop in the reference's shape (Trainer.train_one_epoch, nerf/utils.py:1455-1500) on the HIP
path: update_extra_state every 16 steps, march sized by mean_count, fused field / criterion / Adam + GradScaler, learning
rate decay -- a fresh network fitted to images rendered by a teacher network (tools/fit_scene.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_student_fits_teacher_scene():
    import fit_scene
    lines = []
    psnr = fit_scene.fit(steps=200, log=lines.append)
    first = float(lines[0].split("loss")[1].split()[0])
    last = float(lines[-2].split("loss")[1].split()[0])
    assert last < 0.05 * first, lines                       # the loss falls by more than an order of magnitude
    assert psnr > 35.0, lines                                # held-out rays (400 steps reach ~50 dB)
    assert "skipped by the scaler: 0" in lines[-1]
    occ = [float(l.split("occupied")[1]) for l in lines[:-1]]
    assert occ[-1] < occ[0]                                  # the student's occupancy grid culls empty space as it learns
