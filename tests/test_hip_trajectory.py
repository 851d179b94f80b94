"""What the reference does with the generator is train it (models/model.py:72-86 under Adam, :239-247): 200 Adam steps of the
shipped DN net in every math mode of the engine next to torch float64 of the same graph -- as ENSEMBLES (tools/trajectory.py).
One-step parity cannot show how a mode's rounding accumulates; a single long trajectory cannot either, because the optimisation
is chaotic (float64 from a start moved by one fp32 ulp ends 0.01 - 0.5 dB away): past step ~50 a single run carries no information
about the arithmetic.  Round 5's per-run bars at steps 25 / 50 / 100 / 200 were passed by the 16-bit negative control and are gone."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_training_trajectory_ensembles_every_math_mode_vs_float64():
    """DN 32 x 4, fixed batch of 4 tiles of 64 x 64, seeded reference init, 200 steps of L1 + Adam(1e-4); 24 members per arithmetic
    (member 0: the seeded start; members 1..23: every start weight moved to a neighbouring float32 or left, seeded; the same 24 starts
    for torch float64, f16x3, bf16x6 and the engine's fp32; the 16-bit-operand control runs member 0 only here, its ensemble is in
    profiles/r06_trajectory_ensemble.txt).  Why 24: the step-200 PSNR is skewed -- about one start in five ends 0.3 - 0.5 dB low, in every
    arithmetic -- and a sample of 16 holds none of those every ~35th time (bf16x6 did: sd ratio 0.45), a sample of 24 every ~200th.  Held:
      (1) step 10, member 0 -- the divergence still at rounding level, every fp32-class arithmetic at ~4e-6 dB --: loss within 2e-6
          and the PSNR of two held-out tiles within 1e-4 dB of float64's, outright, for f16x3 (the headline mode), bf16x6 and fp32.
          The bar has teeth: the negative control (torch float32 with conv operands rounded to 16 significant bits, the arithmetic
          of the two-term bf16 modes removed in round 3) sits at 1e-3 dB / 3e-5 there and must FAIL it;
      (2) step 200, distributions: for the loss and for the held-out PSNR, |mean(mode) - mean(float64)| <= 3 pooled standard errors
          and sd(mode) / sd(float64) in [0.5, 2], for every engine mode -- a systematic bias of an arithmetic would shift the mean, extra
          noise would widen the spread.  What the control does under (2) is printed and recorded (profiles/r06_trajectory_ensemble.txt),
          not asserted."""
    import trajectory as tj
    assert torch.cuda.is_available()
    K = 24
    ens, cps = tj.run_ensemble(steps=200, size=64, members=K, control_members=1, log=lambda s: print(s, flush=True))
    text, early, rows = tj.report(ens, cps, 200, 64)
    print(text)
    assert cps == (10, 200)
    for leg in (tj.REF, tj.CONTROL) + tj.ENGINE_MODES:
        assert len(ens[leg]) == (1 if leg == tj.CONTROL else K), leg
        # the runs are real optimisations: the loss falls by more than a third in every member of every leg
        for losses, _ in ens[leg]:
            assert losses[-1] < 0.67 * losses[0], (leg, losses[0], losses[-1])
    # step 1 is one forward from identical weights: every engine mode within 1e-6 of float64's loss, member by member
    for leg in tj.ENGINE_MODES:
        for (l, _), (lr, _) in zip(ens[leg], ens[tj.REF]):
            assert abs(l[0] - lr[0]) < 1e-6 * lr[0] + 1e-7, leg
    # the members differ: an ulp-moved start is a different trajectory by step 200 (the spread the bar is made of is not zero)
    ref_last = [r[0][-1] for r in ens[tj.REF]]
    assert max(ref_last) - min(ref_last) > 1e-6
    # (1) the absolute bar and its teeth
    e = {r[0]: r for r in early}
    for leg in tj.ENGINE_MODES:
        assert e[leg][-1] and e[leg][1] <= tj.ABS_BAR_LOSS and e[leg][2] <= tj.ABS_BAR_DB, e[leg]
    assert not e[tj.CONTROL][-1], e[tj.CONTROL]
    assert e[tj.CONTROL][2] > 3 * tj.ABS_BAR_DB or e[tj.CONTROL][1] > 3 * tj.ABS_BAR_LOSS, e[tj.CONTROL]
    # (2) the distribution bar
    got = {(r[0], r[1]): r for r in rows}
    for leg in tj.ENGINE_MODES:
        for which in ("loss", "psnr"):
            r = got[(leg, which)]
            assert abs(r[5]) <= tj.Z_BAR and tj.SPREAD_BAR[0] <= r[6] <= tj.SPREAD_BAR[1] and r[-1], r
    assert tj.passed(early, rows)
