"""What the reference does with the generator is train it (models/model.py:72-86 under Adam, :239-247): 200 Adam steps of the
shipped DN net in every math mode of the engine next to torch float64 / float32 of the same graph (tools/trajectory.py).
One-step parity cannot show how a mode's rounding accumulates; this does."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_training_trajectory_every_math_mode_vs_float64():
    """DN 32 x 4, fixed batch of 4 tiles of 96 x 96, seeded reference init, 200 steps of L1 + Adam(1e-4).  At steps 50 / 100 /
    200 the engine's loss and the PSNR of two held-out tiles sit within 2 x torch-float32's own distance to float64 (float32 on
    the GPU = MIOpen and on the host = oneDNN: the larger of the two), and within 0.01 dB at step 200 -- for f16x3 (the
    headline mode) and for bf16x6 / fp32 alike."""
    import trajectory as tj
    assert torch.cuda.is_available()
    res, cps = tj.run_all(steps=200, size=96, checkpoints=(50, 100, 200), cpu_f32=True, log=lambda s: print(s, flush=True))
    text, rows = tj.report(res, cps, 200, 96)
    print(text)
    # the run is a real optimisation: the loss falls by more than a third and every leg agrees on that
    for leg, (losses, _) in res.items():
        assert losses[-1] < 0.67 * losses[0], (leg, losses[0], losses[-1])
    assert {r[0] for r in rows} == set(tj.ENGINE_MODES)
    bad = [r for r in rows if not r[-1]]
    assert not bad, bad
    # absolute statement beside the relative one: every mode's PSNR at step 200 within 0.01 dB of float64's
    for leg in tj.ENGINE_MODES:
        d = max(abs(a - b) for a, b in zip(res[leg][1][cps[-1]], res["float64"][1][cps[-1]]))
        assert d <= 0.01, (leg, d)
