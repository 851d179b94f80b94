"""CPU: the loss oracle (oracle/loss.py, hand-derived backward) against tests/golden/loss.npz, whose values and
gradients come from torch autograd over a torch restatement of the torchmetrics algorithm (make_golden_loss.py; l1 and
poisson straight from torch.nn.functional).  Also the host-side factory arithmetic of create_loss."""
import os

import numpy as np
import pytest

import make_golden_loss as mg
from oracle import loss as ol

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss.npz"))


@pytest.mark.parametrize("case", sorted(mg.CASES))
@pytest.mark.parametrize("term", ol.TERMS)
def test_oracle_term_matches_autograd(case, term):
    B, H, W, seed = mg.CASES[case]
    p, t = mg.loss_inputs(B, H, W, seed)
    v, g = ol._FUNCS[term](p, t)
    ref_v, ref_g = G[f"{case}_{term}_f64_value"], G[f"{case}_{term}_f64_grad_sub"]
    assert abs(v - ref_v) <= 1e-12 * max(1.0, abs(ref_v))
    assert np.abs(g[:, ::mg.SUB, ::mg.SUB] - ref_g).max() <= 1e-11 * np.abs(ref_g).max()
    sums = G[f"{case}_{term}_f64_grad_sum"]
    assert abs(g.sum() - sums[0]) <= 1e-9 * sums[1]
    # the fp32 evaluation (what torchmetrics would run) agrees with the float64 one to fp32 accuracy
    assert abs(G[f"{case}_{term}_f32_value"] - ref_v) <= 2e-6 * max(1.0, abs(ref_v))


@pytest.mark.parametrize("C", [2, 3])
def test_oracle_multi_channel_reductions_match_autograd(C):
    """[B,C,H,W] batches (the reference's constructors take 3 -> 2 / 1 -> 3 generators): MS-SSIM averages every scale's statistic
    over a sample's channels before the product over scales, the Poisson term divides by the number of SAMPLES
    (metrics/metrics.py:30-39), ssim / psnr / l1 are what they are over the folded images -- the numpy oracle against torch
    autograd over the torch restatement of torchmetrics' reduction (make_golden_loss.ssim_update(channels=C))."""
    import torch
    import torch.nn.functional as F
    B, H, W = 2, 304, 320
    p3, t3 = mg.loss_inputs(B * C, H, W, 40 + C)
    p4, t4 = p3.reshape(B, C, H, W), t3.reshape(B, C, H, W)
    tp = torch.from_numpy(p3).double()[:, None].requires_grad_(True)      # folded [B*C,1,H,W]
    tt = torch.from_numpy(t3).double()[:, None]
    cases = {
        "ms_ssim": lambda: mg.ms_ssim(tp, tt, channels=C),
        "ssim": lambda: mg.ssim_update(tp, tt, channels=C)[0].mean(),
        "poisson": lambda: F.poisson_nll_loss(tp, tt, log_input=False, reduction="mean") / B,
        "psnr": lambda: mg.psnr(tp, tt),
        "l1": lambda: F.l1_loss(tp, tt),
    }
    for term, fn in cases.items():
        tp.grad = None
        v = fn()
        v.backward()
        g_ref = tp.grad[:, 0].numpy().reshape(B, C, H, W)
        v_o, g_o = ol._FUNCS[term](p4, t4)
        assert abs(v_o - v.item()) <= 1e-12 * max(1.0, abs(v.item())), term
        assert g_o.shape == (B, C, H, W) and np.abs(g_o - g_ref).max() <= 1e-10 * np.abs(g_ref).max(), term
    # and the per-sample reduction is NOT the per-image one (what folding the channels into the batch would compute)
    v_fold, _ = ol.ms_ssim(p3, t3)
    v_grp, _ = ol.ms_ssim(p4, t4)
    assert abs(v_fold - v_grp) > 1e-8
    assert abs(ol.poisson(p3, t3)[0] * C - ol.poisson(p4, t4)[0]) < 1e-15


def test_create_loss_arithmetic_and_size_checks():
    sc = {"psnr": {"scaling": -0.11938872970391594, "correction": 3.6491165234001905},
          "ms_ssim": {"scaling": -2.85143997718848, "correction": 2.737382378100941}}
    w, corr = ol.effective_weights(dict(l1=0.0, poisson=0.0, psnr=0.5, ssim=0.0, ms_ssim=0.5), sc)
    assert w == {"psnr": 0.5 * sc["psnr"]["scaling"], "ms_ssim": 0.5 * sc["ms_ssim"]["scaling"]}
    assert corr == sc["psnr"]["correction"] + sc["ms_ssim"]["correction"]
    B, H, W, seed = mg.CASES["a"]
    p, t = mg.loss_inputs(B, H, W, seed)
    total, values, grad = ol.loss_and_grad(p, t, w, corr)
    expect = w["psnr"] * G["a_psnr_f64_value"] + w["ms_ssim"] * G["a_ms_ssim_f64_value"] + corr
    assert abs(total - expect) < 1e-12
    assert grad.shape == p.shape
    with pytest.raises(ValueError):
        ol.ms_ssim(p[:, :100, :100], t[:, :100, :100])
    with pytest.raises(AssertionError):
        ol.effective_weights(dict(l1=0.0), None)
