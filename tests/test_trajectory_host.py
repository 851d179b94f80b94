"""Host logic of tools/trajectory.py (the bar of tests/test_hip_trajectory.py) on synthetic records: no GPU, no training."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _leg(loss_off, psnr_off, steps=200, cps=(10, 25, 50, 100, 200)):
    """a run whose loss / PSNR differ from the reference run's by the given offsets at every checkpoint ({checkpoint: offset} or a number)"""
    lo = (lambda c: loss_off[c]) if isinstance(loss_off, dict) else (lambda c: loss_off)
    po = (lambda c: psnr_off[c]) if isinstance(psnr_off, dict) else (lambda c: psnr_off)
    losses = [0.1 / (1 + 0.05 * s) for s in range(steps)]
    for c in cps:
        losses[c - 1] += lo(c)
    return losses, {c: [28.0 + 0.01 * c + po(c), 27.0 + 0.01 * c - po(c)] for c in cps}


def test_verdict_applies_the_absolute_bar_early_and_the_relative_one_later():
    import trajectory as tj
    cps = (10, 25, 50, 100, 200)
    res = {
        "float64": _leg(0.0, 0.0),
        "float64_ulp0": _leg(1e-7, {10: 1e-6, 25: 1e-5, 50: 1e-3, 100: 0.05, 200: 0.2}),
        "float32_cpu": _leg(2e-7, {10: 4e-6, 25: 5e-5, 50: 2e-3, 100: 0.3, 200: 0.1}),
        "float32": _leg(2e-7, {10: 4e-6, 25: 5e-5, 50: 3e-3, 100: 0.1, 200: 0.7}),
        tj.CONTROL: _leg(3e-5, 1e-3),
        # f16x3: inside everything; bf16x6: 1.3 dB out at step 200 while the yard-sticks' largest is 0.7 (bar 1.4): inside; fp32: out at step 10
        "f16x3": _leg(1.5e-7, {10: 4e-6, 25: 6e-5, 50: 5e-3, 100: 0.5, 200: 0.9}),
        "bf16x6": _leg(1.5e-7, {10: 4e-6, 25: 6e-5, 50: 9e-3, 100: 0.59, 200: 1.3}),
        "fp32": _leg(1.5e-7, {10: 2e-4, 25: 6e-5, 50: 5e-3, 100: 0.5, 200: 0.9}),
    }
    assert tj.yard_sticks(res) == ["float64_ulp0", "float32_cpu", "float32"]          # neither the reference, nor the control, nor an engine mode
    rows = {(r[0], r[1]): r for r in tj.verdict(res, cps)}
    assert all(rows[("f16x3", c)][-1] for c in cps)
    assert all(rows[("bf16x6", c)][-1] for c in cps)
    assert not rows[("fp32", 10)][-1] and all(rows[("fp32", c)][-1] for c in cps[1:])     # 2e-4 dB at step 10 misses the absolute bar
    assert rows[("f16x3", 10)][4:6] == (tj.ABS_BAR_LOSS, tj.ABS_BAR_DB)                    # the absolute bars, whatever the yard-sticks do
    assert rows[("f16x3", 50)][5] == 0.01 and rows[("f16x3", 100)][5] == 1.0               # allowances where 2 x yard is smaller ...
    assert abs(rows[("bf16x6", 200)][5] - 1.4) < 1e-12                                     # ... 2 x the largest yard-stick where it is larger
    assert not rows[(tj.CONTROL, 10)][-1]                                                  # the negative control is outside the step-10 bar
    text, _ = tj.report(res, cps, 200, 64)
    assert "BAR MISSED: fp32@10" in text and "CAUGHT by the absolute bar" in text
    res["fp32"] = res["f16x3"]
    text, _ = tj.report(res, cps, 200, 64)
    assert "ALL ENGINE MODES WITHIN THE BAR" in text


def test_operand_rounding_of_the_negative_control():
    import torch
    import trajectory as tj
    v = torch.tensor([1.0 + 2.0 ** -20, 3.14159265, -1e-3, 0.0], dtype=torch.float64, requires_grad=True)
    r = tj._round_sig(v, 16)
    assert r[0].item() == 1.0 and r[3].item() == 0.0
    assert abs(r[1].item() - 3.14159265) <= 3.14159265 * 2.0 ** -16 and r[1].item() != 3.14159265
    r.sum().backward()
    assert torch.equal(v.grad, torch.ones_like(v))                                          # identity gradient
