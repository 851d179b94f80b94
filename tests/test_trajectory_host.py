"""Host logic of tools/trajectory.py (the bar of tests/test_hip_trajectory.py) on synthetic records: no GPU, no training."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

STEPS = 200
N = 16      # members per leg (two samples of 8 from ONE normal distribution miss sd ratio in [0.5, 2] every ~11th time; of 16, every ~100th)


def _member(loss10, psnr10, loss200, psnr200):
    """one run's record: (loss per step, {10: [psnr tile 0, tile 1], 200: [...]})"""
    losses = [0.1 / (1 + 0.05 * s) for s in range(STEPS)]
    losses[9] = loss10
    losses[-1] = loss200
    return losses, {10: [16.0 + psnr10, 15.0 - psnr10], STEPS: [psnr200 + 0.5, psnr200 - 0.5]}


def _leg(n, seed, d10_loss, d10_psnr, mean_l, sd_l, mean_p, sd_p):
    g = np.random.default_rng(seed)
    ls = mean_l + sd_l * g.standard_normal(n)
    ps = mean_p + sd_p * g.standard_normal(n)
    return [_member(0.05 + (d10_loss if i == 0 else 0.0), d10_psnr if i == 0 else 0.0, float(ls[i]), float(ps[i])) for i in range(n)]


def _ens():      # (seeds chosen so that each 16-member sample has a typical mean and spread)
    import trajectory as tj
    return {
        tj.REF: _leg(N, 23, 0.0, 0.0, 0.0240, 4e-4, 29.0, 0.20),
        tj.CONTROL: _leg(N, 2, 3e-5, 1e-3, 0.0241, 4e-4, 29.05, 0.20),      # outside the step-10 bar, inside the step-200 distribution
        "f16x3": _leg(N, 9, 1.5e-7, 4e-6, 0.0240, 4e-4, 29.0, 0.20),
        "bf16x6": _leg(N, 15, 1.5e-7, 4e-6, 0.0241, 5e-4, 28.95, 0.25),
        "fp32": _leg(N, 33, 1.5e-7, 4e-6, 0.0240, 4e-4, 29.0, 0.20),
    }


def test_early_bar_is_absolute_and_the_control_must_fail_it():
    import trajectory as tj
    ens = _ens()
    rows = {r[0]: r for r in tj.early_verdict(ens)}
    assert all(rows[m][-1] for m in tj.ENGINE_MODES)
    assert not rows[tj.CONTROL][-1] and abs(rows[tj.CONTROL][2] - 1e-3) < 1e-9
    ens["fp32"][0] = _member(0.05, 2e-4, 0.024, 29.0)      # 2e-4 dB at step 10: outside the absolute bar
    rows = {r[0]: r for r in tj.early_verdict(ens)}
    assert not rows["fp32"][-1] and rows["f16x3"][-1]


def test_ensemble_bar_catches_a_shifted_mean_and_a_changed_spread():
    import trajectory as tj
    ens = _ens()
    rows = {(r[0], r[1]): r for r in tj.ensemble_verdict(ens, STEPS)}
    assert all(rows[(m, w)][-1] for m in tj.ENGINE_MODES for w in ("loss", "psnr")), rows
    assert rows[(tj.REF, "loss")][2] == N and rows[(tj.REF, "psnr")][5] == 0.0
    # z is Welch's: (mean - mean_ref) / sqrt(sd^2 / n + sd_ref^2 / n_ref); the PSNR statistic is the mean of the held-out tiles
    ref_p = [float(np.mean(r[1][STEPS])) for r in ens[tj.REF]]
    f_p = [float(np.mean(r[1][STEPS])) for r in ens["f16x3"]]
    z = (np.mean(f_p) - np.mean(ref_p)) / np.sqrt(np.var(f_p, ddof=1) / N + np.var(ref_p, ddof=1) / N)
    assert abs(rows[("f16x3", "psnr")][5] - z) < 1e-9
    # a biased arithmetic: the same spread, the mean 0.4 dB (2 sd) lower -> |z| ~ 2 sd / (sd sqrt(2 / 16)) = 5.7
    ens["f16x3"] = _leg(N, 9, 1.5e-7, 4e-6, 0.0240, 4e-4, 28.6, 0.20)
    rows = {(r[0], r[1]): r for r in tj.ensemble_verdict(ens, STEPS)}
    assert not rows[("f16x3", "psnr")][-1] and abs(rows[("f16x3", "psnr")][5]) > tj.Z_BAR and rows[("f16x3", "loss")][-1]
    # a noisy arithmetic: the right mean, three times the spread
    ens["bf16x6"] = _leg(N, 15, 1.5e-7, 4e-6, 0.0240, 1.2e-3, 29.0, 0.20)
    rows = {(r[0], r[1]): r for r in tj.ensemble_verdict(ens, STEPS)}
    assert not rows[("bf16x6", "loss")][-1] and rows[("bf16x6", "loss")][6] > tj.SPREAD_BAR[1]
    text, early, rws = tj.report(ens, (10, STEPS), STEPS, 64)
    assert "BAR MISSED" in text and "f16x3:psnr@200" in text and "bf16x6:loss@200" in text
    assert "CAUGHT by the absolute bar" in text and not tj.passed(early, rws)


def test_report_and_exit_status():
    import trajectory as tj
    ens = _ens()
    text, early, rows = tj.report(ens, (10, STEPS), STEPS, 64)
    assert "ALL ENGINE MODES WITHIN THE BAR" in text and "CAUGHT by the absolute bar" in text
    assert "does NOT separate from float64" in text          # this synthetic control sits inside the step-200 distribution
    assert tj.passed(early, rows)
    # a control that passes the step-10 bar means the bar has no teeth: the tool's exit status says so
    ens[tj.CONTROL][0] = _member(0.05, 0.0, 0.0241, 29.05)
    text, early, rows = tj.report(ens, (10, STEPS), STEPS, 64)
    assert "NOT caught" in text and not tj.passed(early, rows)


def test_perturbed_start_moves_every_weight_by_at_most_one_float32_ulp():
    import torch
    import trajectory as tj
    st = {"w": torch.randn(1000, generator=torch.Generator().manual_seed(0)), "b": torch.tensor([0.0, 1.0, -1.0])}
    assert all(torch.equal(tj.perturbed_start(st, None)[k], st[k]) for k in st)
    p = tj.perturbed_start(st, 7)
    up = torch.nextafter(st["w"], torch.full_like(st["w"], float("inf")))
    dn = torch.nextafter(st["w"], torch.full_like(st["w"], float("-inf")))
    assert bool(((p["w"] == st["w"]) | (p["w"] == up) | (p["w"] == dn)).all())
    moved = (p["w"] != st["w"]).float().mean().item()
    assert 0.5 < moved < 0.8                                       # two of three weights move
    assert not torch.equal(tj.perturbed_start(st, 8)["w"], p["w"]) and torch.equal(tj.perturbed_start(st, 7)["w"], p["w"])
    assert tj.member_seed(0) is None and tj.member_seed(3) == 103


def test_batched_members_equal_one_run_per_member():
    """run_torch_members (K members in one graph: im2col + einsum, stacked Adam) against run_torch (F.conv2d, one member at a time)
    on a small net in float64: the same losses and held-out PSNR, member by member -- the float64 ensemble IS K float64 runs"""
    import torch
    import trajectory as tj
    from xmm_superres_denoise.models import GeneratorRRDB_DN
    old = tj.NF, tj.BLOCKS
    tj.NF, tj.BLOCKS = 8, 1
    try:
        torch.manual_seed(0)
        m = GeneratorRRDB_DN(1, 1, 8, 1)
        state = {k: v.detach().clone() for k, v in m.state_dict().items()}
        x, t = tj.denoise_pairs(2, 16, 11)
        xh, th = tj.denoise_pairs(2, 16, 23)
        seeds = [None, 101, 102]
        got = tj.run_torch_members("float64", state, seeds, x, t, xh, th, 6, (3, 6), device="cpu")
        for i, sd in enumerate(seeds):
            l, p = tj.run_torch("float64", state, x, t, xh, th, 6, (3, 6), device="cpu", perturb_seed=sd)
            assert max(abs(a - b) for a, b in zip(l, got[i][0])) < 1e-12
            for c in (3, 6):
                assert max(abs(a - b) for a, b in zip(p[c], got[i][1][c])) < 1e-9
        assert got[0][0][-1] != got[1][0][-1]                    # the members are different runs
        ctl = tj.run_torch_members("float32", state, [None], x, t, xh, th, 4, (4,), device="cpu", sig_bits=16)
        l, _ = tj.run_torch("float32", state, x, t, xh, th, 4, (4,), device="cpu", sig_bits=16)
        assert max(abs(a - b) for a, b in zip(l, ctl[0][0])) < 1e-6
    finally:
        tj.NF, tj.BLOCKS = old


def test_operand_rounding_of_the_negative_control():
    import torch
    import trajectory as tj
    v = torch.tensor([1.0 + 2.0 ** -20, 3.14159265, -1e-3, 0.0], dtype=torch.float64, requires_grad=True)
    r = tj._round_sig(v, 16)
    assert r[0].item() == 1.0 and r[3].item() == 0.0
    assert abs(r[1].item() - 3.14159265) <= 3.14159265 * 2.0 ** -16 and r[1].item() != 3.14159265
    r.sum().backward()
    assert torch.equal(v.grad, torch.ones_like(v))                                          # identity gradient
