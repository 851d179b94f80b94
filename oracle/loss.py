"""CPU oracle of the training loss.  TEST INFRASTRUCTURE ONLY.

Restates, in numpy float64 on fp32 inputs, what the reference's `create_loss` builds
(xmm_superres_denoise/utils/loss_functions.py:11-47; weights from res/configs/loss_functions.toml:5-42):
a weighted sum of torchmetrics metrics evaluated per batch by `Metric.forward` (models/model.py:78),
    total = sum_i p_i * scaling_i * metric_i(preds, target)  (+ sum_i correction_i  if that sum is > 0).

PARITY UNPINNED for psnr / ssim / ms_ssim: their arithmetic lives in torchmetrics (poetry.lock pins 0.11.4, the code
needs >= 1.0), which is not installed here and is not under /root/reference.  The functions below restate the
published torchmetrics 1.x algorithm (functional/image/ssim.py `_ssim_update`, `_multiscale_ssim_update`;
functional/image/psnr.py `_psnr_compute`), including its quirks:
  * with gaussian_kernel=True the window is int(3.5*sigma+0.5)*2+1 = 19 taps for sigma=2.5 (the kernel_size=13
    argument is only used by the MS-SSIM size check), reflect-padded and then cropped by the same amount, so only the
    "valid" interior (H-18)x(W-18) contributes and the padding never matters;
  * data_range=None -> max(preds.max()-preds.min(), target.max()-target.min()) of the CURRENT scale, a differentiable
    function of preds (gradient goes to the arg-max / arg-min pixels, split evenly over ties like torch.max());
  * variances are clamped at 0; MS-SSIM uses normalize="relu", betas (0.0448, 0.2856, 0.3001, 0.2363, 0.1333),
    2x2 average pooling between scales, per-image product, batch mean;
  * PSNR: data_range = max(target.max(), 0) - min(target.min(), 0) (metric states start at 0), base 10.
l1 (MeanAbsoluteError) and poisson (metrics/metrics.py:30-39: F.poisson_nll_loss(log_input=False, mean) / batch size)
are pinned against torch.nn.functional by tests/golden/make_golden_loss.py, which also cross-checks every gradient
here against torch autograd of the same restatement.
"""
from __future__ import annotations

import numpy as np

BETAS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
TERMS = ("l1", "poisson", "psnr", "ssim", "ms_ssim")


def gaussian_taps(sigma: float) -> np.ndarray:
    size = int(3.5 * sigma + 0.5) * 2 + 1
    dist = np.arange((1 - size) / 2, (1 + size) / 2, 1.0)
    g = np.exp(-((dist / sigma) ** 2) / 2)
    return g / g.sum()


def _filt_valid(a, g):
    """separable valid correlation over the last two axes"""
    R2 = len(g) - 1
    H, W = a.shape[-2:]
    tmp = sum(g[k] * a[..., :, k:W - R2 + k] for k in range(len(g)))
    return sum(g[k] * tmp[..., k:H - R2 + k, :] for k in range(len(g)))


def _filt_full_T(a, g):
    """transpose of _filt_valid: (.., H-2R, W-2R) -> (.., H, W)"""
    R2 = len(g) - 1
    Hv, Wv = a.shape[-2:]
    tmp = np.zeros(a.shape[:-2] + (Hv + R2, Wv), a.dtype)
    for k in range(len(g)):
        tmp[..., k:k + Hv, :] += g[k] * a
    out = np.zeros(a.shape[:-2] + (Hv + R2, Wv + R2), a.dtype)
    for k in range(len(g)):
        out[..., :, k:k + Wv] += g[k] * tmp
    return out


def _data_range(p, t):
    rp = p.max() - p.min()
    rt = t.max() - t.min()
    return (rp, True) if rp >= rt else (rt, False)


def ssim_scale(p, t, sigma=2.5, k1=0.01, k2=0.05):
    """One `_ssim_update(..., return_contrast_sensitivity=True)`: p, t [B,H,W] float64.
    Returns sim[B], cs[B] and a backward closure (g_sim[B], g_cs[B]) -> d/dp [B,H,W]."""
    g = gaussian_taps(sigma)
    dr, from_p = _data_range(p, t)
    c1, c2 = (k1 * dr) ** 2, (k2 * dr) ** 2
    mp, mt = _filt_valid(p, g), _filt_valid(t, g)
    epp, ett, ept = _filt_valid(p * p, g), _filt_valid(t * t, g), _filt_valid(p * t, g)
    vp_raw, vt_raw = epp - mp * mp, ett - mt * mt
    vp, vt = np.maximum(vp_raw, 0), np.maximum(vt_raw, 0)
    cov = ept - mp * mt
    U, L = 2 * cov + c2, vp + vt + c2
    A, Bq = 2 * mp * mt + c1, mp * mp + mt * mt + c1
    cs_map, lum = U / L, A / Bq
    n = cs_map.shape[-1] * cs_map.shape[-2]
    sim = (lum * cs_map).reshape(len(p), -1).mean(-1)
    cs = cs_map.reshape(len(p), -1).mean(-1)

    def backward(g_sim, g_cs):
        d_ssim = (np.asarray(g_sim) / n)[:, None, None]
        d_cs = (np.asarray(g_cs) / n)[:, None, None] + d_ssim * lum
        d_lum = d_ssim * cs_map
        dA, dB = d_lum / Bq, -d_lum * A / (Bq * Bq)
        dU, dL = d_cs / L, -d_cs * U / (L * L)
        d_cov = 2 * dU
        d_vp = np.where(vp_raw > 0, dL, 0.0)
        d_mp = dA * 2 * mt + dB * 2 * mp - d_cov * mt - d_vp * 2 * mp
        dp = _filt_full_T(d_mp, g) + 2 * p * _filt_full_T(d_vp, g) + t * _filt_full_T(d_cov, g)
        if from_p:
            d_dr = 2 * k1 * k1 * dr * (dA + dB).sum() + 2 * k2 * k2 * dr * (dU + dL).sum()
            hi, lo = p == p.max(), p == p.min()
            dp = dp + d_dr * (hi / hi.sum() - lo / lo.sum())
        return dp

    return sim, cs, backward


def _pool(a):
    H, W = a.shape[-2] // 2 * 2, a.shape[-1] // 2 * 2
    a = a[..., :H, :W]
    return 0.25 * (a[..., 0::2, 0::2] + a[..., 0::2, 1::2] + a[..., 1::2, 0::2] + a[..., 1::2, 1::2])


def _unpool(g, shape):
    out = np.zeros(g.shape[:-2] + tuple(shape), g.dtype)
    H, W = g.shape[-2] * 2, g.shape[-1] * 2
    out[..., :H, :W] = 0.25 * np.repeat(np.repeat(g, 2, -2), 2, -1)
    return out


def check_ms_ssim_size(H, W, kernel_size=13, nscales=len(BETAS)):
    """the ValueErrors `_multiscale_ssim_update` raises"""
    if min(H, W) < 2 ** nscales:
        raise ValueError("image too small for the number of MS-SSIM scales")
    div = max(1, nscales - 1) ** 2
    if H // div <= kernel_size - 1 or W // div <= kernel_size - 1:
        raise ValueError("image too small for the MS-SSIM kernel size")


def _fold(p, t):
    """[B,C,H,W] -> ([B*C,H,W], [B*C,H,W], C, original shape); [B,H,W] passes with C = 1.  torchmetrics filters every channel on its
    own (grouped conv) and takes data_range over the whole tensors, so a multi-channel batch IS its B*C images -- except for
    the per-SAMPLE reductions, which the callers do over groups of C consecutive images."""
    p = np.asarray(p, np.float64)
    t = np.asarray(t, np.float64)
    shape = p.shape
    C = shape[1] if p.ndim == 4 else 1
    return p.reshape((-1,) + shape[-2:]), t.reshape((-1,) + shape[-2:]), C, shape


def ms_ssim(p, t, sigma=2.5, k1=0.01, k2=0.05, betas=BETAS, want_grad=True):
    """[B,H,W] or [B,C,H,W].  Per scale the statistic of a SAMPLE is the mean over its C channels (`_ssim_update`:
    `.reshape(B, -1).mean(-1)` over C, H, W), relu, then the product over scales per sample, then the batch mean."""
    p, t, C, shape = _fold(p, t)
    check_ms_ssim_size(*p.shape[-2:])
    ps, vals, backs = [p], [], []
    for s in range(len(betas)):
        sim, cs, back = ssim_scale(p, t, sigma, k1, k2)
        raw = (sim if s == len(betas) - 1 else cs).reshape(-1, C).mean(1)       # per sample
        vals.append(np.maximum(raw, 0))
        backs.append(back)
        if s + 1 < len(betas):
            p, t = _pool(p), _pool(t)
            ps.append(p)
    M = np.prod([v ** b for v, b in zip(vals, betas)], axis=0)     # per sample
    value = M.mean()
    if not want_grad:
        return value, None
    S = len(M)
    grad = None
    for s in reversed(range(len(betas))):
        gv = np.where(vals[s] > 0, betas[s] * M / np.where(vals[s] > 0, vals[s], 1.0), 0.0) / S
        gv = np.repeat(gv / C, C)                                   # every channel image of the sample gets 1/C of it
        zero = np.zeros_like(gv)
        local = backs[s](gv, zero) if s == len(betas) - 1 else backs[s](zero, gv)
        grad = local if grad is None else local + _unpool(grad, ps[s].shape[-2:])
    return value, grad.reshape(shape)


def ssim(p, t, sigma=2.5, k1=0.01, k2=0.05, want_grad=True):
    p, t, _, shape = _fold(p, t)         # mean over samples of the channel means = mean over all B*C images
    sim, _, back = ssim_scale(p, t, sigma, k1, k2)
    B = len(sim)
    return sim.mean(), (back(np.full(B, 1.0 / B), np.zeros(B)).reshape(shape) if want_grad else None)


def psnr(p, t, want_grad=True):
    p = np.asarray(p, np.float64)
    t = np.asarray(t, np.float64)
    dr = max(t.max(), 0.0) - min(t.min(), 0.0)
    d = p - t
    mse = (d * d).sum() / d.size
    value = (10 / np.log(10.0)) * (2 * np.log(dr) - np.log(mse))
    return value, (-(10 / np.log(10.0)) * 2 * d / (d.size * mse) if want_grad else None)


def l1(p, t, want_grad=True):
    p = np.asarray(p, np.float64)
    t = np.asarray(t, np.float64)
    d = p - t
    return np.abs(d).sum() / d.size, (np.sign(d) / d.size if want_grad else None)


def poisson(p, t, want_grad=True):
    p = np.asarray(p, np.float64)
    t = np.asarray(t, np.float64)
    B = p.shape[0]
    value = (p - t * np.log(p + 1e-8)).mean() / B
    return value, ((1 - t / (p + 1e-8)) / (p.size * B) if want_grad else None)


_FUNCS = {"l1": l1, "poisson": poisson, "psnr": psnr, "ssim": ssim, "ms_ssim": ms_ssim}


def effective_weights(loss_cfg: dict, sc_dict: dict | None):
    """the loop of create_loss (loss_functions.py:25-36): returns ({term: weight}, correction)"""
    weights, correction = {}, 0.0
    for name in TERMS:
        pw = float(loss_cfg.get(name, 0.0))
        if pw > 0.0:
            if sc_dict is not None and name in sc_dict:
                pw = pw * sc_dict[name]["scaling"]
                correction = correction + sc_dict[name]["correction"]
            weights[name] = pw
    assert weights
    return weights, correction


def loss_and_grad(p, t, weights: dict, correction: float = 0.0, sigma=2.5, k1=0.01, k2=0.05):
    """p, t: [B,H,W] or [B,C,H,W].  Returns (total, {term: value}, dtotal/dp)."""
    total, grad, values = 0.0, 0.0, {}
    for name, w in weights.items():
        if name in ("ssim", "ms_ssim"):
            v, g = _FUNCS[name](p, t, sigma, k1, k2)
        else:
            v, g = _FUNCS[name](p, t)
        values[name] = float(v)
        total += w * v
        grad = grad + w * g
    if correction > 0.0:
        total += correction
    return float(total), values, grad


def metric_epoch(batches, stretch=lambda a: a):
    """Epoch-level values of the reference's validation metric set (metrics/xmm_metric_collection.py:14-38) over a list
    of (preds, target) batches, accumulated the way the torchmetrics classes accumulate their states."""
    sse = n = abs_sum = 0.0
    tmin = tmax = 0.0
    ssim_sum = ms_sum = po_sum = 0.0
    nimg = 0
    for p, t in batches:
        p = stretch(np.asarray(p, np.float64))
        t = stretch(np.asarray(t, np.float64))
        d = p - t
        sse += (d * d).sum(); abs_sum += np.abs(d).sum(); n += d.size
        tmin, tmax = min(tmin, t.min()), max(tmax, t.max())
        B = p.shape[0]
        ssim_sum += ssim(p, t, want_grad=False)[0] * B
        ms_sum += ms_ssim(p, t, want_grad=False)[0] * B
        po_sum += poisson(p, t, want_grad=False)[0] * B
        nimg += B
    mse = sse / n
    return {"psnr": 10 * np.log10((tmax - tmin) ** 2 / mse), "ssim": ssim_sum / nimg, "ms_ssim": ms_sum / nimg,
            "l1": abs_sum / n, "l2": mse, "poisson": po_sum / nimg}
