#!/usr/bin/env python3
"""Golden fixture for the loss terms (tests/golden/loss.npz).

The reference builds its loss from torchmetrics classes (utils/loss_functions.py:11-47); torchmetrics is not installed
in the build container, so this script restates the published torchmetrics 1.x algorithm with torch ops (F.conv2d,
F.avg_pool2d, reflect pad + crop exactly as `_ssim_update` does) and lets torch AUTOGRAD produce the gradients.
l1 / poisson use torch.nn.functional directly (those two terms are therefore pinned to the real arithmetic; psnr, ssim
and ms_ssim stay "parity unpinned", see oracle/loss.py).  Inputs are regenerated from seeds by loss_inputs().
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

BETAS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
SUB = 8   # gradients are stored on a [::SUB, ::SUB] lattice plus their full sums


def loss_inputs(B, H, W, seed):
    """pred / target pairs with exact ties at 0 and 1 (clamped network outputs look like this)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:H, 0:W]
    t = np.stack([0.45 + 0.5 * np.sin(xx / (9.0 + b) + b) * np.cos(yy / (7.0 + 2 * b)) for b in range(B)])
    t = np.clip(t + 0.15 * rng.standard_normal((B, H, W)), 0, 1).astype(np.float32)
    p = np.clip(t + 0.08 * rng.standard_normal((B, H, W)), 0, 1).astype(np.float32)
    return p, t


def _gauss(sigma, dtype):
    size = int(3.5 * sigma + 0.5) * 2 + 1
    dist = torch.arange((1 - size) / 2, (1 + size) / 2, 1, dtype=dtype)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    g = (g / g.sum()).unsqueeze(0)
    return torch.matmul(g.t(), g)[None, None], (size - 1) // 2


def ssim_update(p, t, sigma=2.5, k1=0.01, k2=0.05, channels=1):
    """p, t: [B,1,H,W]; returns per-image (ssim, contrast sensitivity).  channels = C: p, t hold the B*C channel images of a
    [B,C,H,W] batch (torchmetrics filters every channel on its own: a grouped conv = the same conv over the folded images) and
    the statistics are per SAMPLE: `.reshape(B, -1).mean(-1)` runs over C, H and W."""
    data_range = max(p.max() - p.min(), t.max() - t.min())
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    kern, pad = _gauss(sigma, p.dtype)
    pp = F.pad(p, (pad, pad, pad, pad), mode="reflect")
    tp = F.pad(t, (pad, pad, pad, pad), mode="reflect")
    out = F.conv2d(torch.cat((pp, tp, pp * pp, tp * tp, pp * tp)), kern).split(p.shape[0])
    mu_pp, mu_tt, mu_pt = out[0].pow(2), out[1].pow(2), out[0] * out[1]
    s_pp = torch.clamp(out[2] - mu_pp, min=0.0)
    s_tt = torch.clamp(out[3] - mu_tt, min=0.0)
    s_pt = out[4] - mu_pt
    upper, lower = 2 * s_pt + c2, s_pp + s_tt + c2
    full = ((2 * mu_pt + c1) * upper) / ((mu_pp + mu_tt + c1) * lower)
    ssim_idx = full[..., pad:-pad, pad:-pad]
    cs = (upper / lower)[..., pad:-pad, pad:-pad]
    return ssim_idx.reshape(p.shape[0] // channels, -1).mean(-1), cs.reshape(p.shape[0] // channels, -1).mean(-1)


def ms_ssim(p, t, **kw):
    mcs = []
    for _ in BETAS:
        sim, cs = ssim_update(p, t, **kw)
        mcs.append(torch.relu(cs))
        p, t = F.avg_pool2d(p, (2, 2)), F.avg_pool2d(t, (2, 2))
    mcs[-1] = torch.relu(sim)
    betas = torch.tensor(BETAS, dtype=p.dtype).view(-1, 1)
    return torch.prod(torch.stack(mcs) ** betas, 0).mean()


def psnr(p, t):
    zero = torch.zeros((), dtype=t.dtype)
    dr = torch.maximum(t.max(), zero) - torch.minimum(t.min(), zero)
    sse, n = torch.sum(torch.pow(p - t, 2)), t.numel()
    return (2 * torch.log(dr) - torch.log(sse / n)) * (10 / torch.log(torch.tensor(10.0, dtype=t.dtype)))


TERMS = {
    "l1": lambda p, t: F.l1_loss(p, t),
    "poisson": lambda p, t: F.poisson_nll_loss(p, t, log_input=False, reduction="mean") / p.shape[0],
    "psnr": psnr,
    "ssim": lambda p, t: ssim_update(p, t)[0].mean(),
    "ms_ssim": ms_ssim,
}

CASES = {"a": (2, 320, 336, 11), "b": (3, 304, 304, 12)}


def main():
    out = {}
    for cname, (B, H, W, seed) in CASES.items():
        p_np, t_np = loss_inputs(B, H, W, seed)
        out[f"{cname}_shape"] = np.array([B, H, W, seed])
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            for name, fn in TERMS.items():
                p = torch.from_numpy(p_np).to(dtype)[:, None].requires_grad_(True)
                t = torch.from_numpy(t_np).to(dtype)[:, None]
                v = fn(p, t)
                v.backward()
                g = p.grad[:, 0].numpy()
                out[f"{cname}_{name}_{tag}_value"] = np.array(v.item(), np.float64)
                out[f"{cname}_{name}_{tag}_grad_sub"] = g[:, ::SUB, ::SUB].astype(np.float64 if tag == "f64" else np.float32)
                out[f"{cname}_{name}_{tag}_grad_sum"] = np.array([g.astype(np.float64).sum(), np.abs(g.astype(np.float64)).sum()])
                print(cname, tag, name, v.item(), np.abs(g).max())
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **out)


if __name__ == "__main__":
    main()
