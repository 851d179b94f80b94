"""Validation metric set of the reference (metrics/xmm_metric_collection.py:14-38,67-91,114-143): psnr, ssim, ms_ssim
(kernel_size=13, sigma=2.5, k2=0.05), l1, l2, poisson, evaluated once per scaling normalizer ("linear", "sqrt", ...)
on images whose dataset stretch is first undone and which are then re-stretched with that normalizer.

All values come from one `xsd_loss_eval` call per (batch, stretch mode) (include/xsd.h); the per-batch states it returns
(sum of squared errors, target range, per-image similarity sums) are accumulated on the device exactly as the
torchmetrics classes accumulate theirs, so `compute()` gives epoch-level values, not a mean of batch values:
  psnr    10*log10(data_range^2 / (sum_sq_err / n)), data_range = max(target) - min(target) over the epoch with both
          states starting at 0 (PeakSignalNoiseRatio(data_range=None));
  ssim / ms_ssim   sum of per-image values / number of images;
  l1, l2  sum of absolute / squared errors / n;
  poisson sum of per-batch means / number of images (metrics/metrics.py:30-39 as written).
`denorm` / `norm` are the bare stretch functions of Normalize (transforms/normalize.py:55-62), applied to [0,1] images as
in the reference's `update` (:136-143).  The piq / VIF "extended" collections (get_ext_metrics) are third-party arithmetic outside the hot path.
"""
from __future__ import annotations

from typing import List

import torch

from ..utils.loss_functions import Loss

NAMES = ("psnr", "ssim", "ms_ssim", "l1", "l2", "poisson")


class _State:
    def __init__(self):
        self.acc = None      # device tensor: [sse, n, tmin, tmax, ssim_sum, ms_sum, nimg, abs_sum, poisson_sum]

    def add(self, out: torch.Tensor, n: int, nimg: int):
        # out: [total, l1, poisson, psnr, ssim, ms_ssim, mse, tmin, tmax, ...] of one batch
        cur = torch.stack([out[6] * n, out.new_tensor(float(n)), out[7], out[8], out[4] * nimg, out[5] * nimg,
                           out.new_tensor(float(nimg)), out[1] * n, out[2] * nimg]).double()
        if self.acc is None:
            zero = torch.zeros((), dtype=torch.float64, device=out.device)
            cur[2] = torch.minimum(cur[2], zero)     # metric states start at 0
            cur[3] = torch.maximum(cur[3], zero)
            self.acc = cur
        else:
            a = self.acc
            self.acc = torch.stack([a[0] + cur[0], a[1] + cur[1], torch.minimum(a[2], cur[2]), torch.maximum(a[3], cur[3]),
                                    a[4] + cur[4], a[5] + cur[5], a[6] + cur[6], a[7] + cur[7], a[8] + cur[8]])

    def compute(self) -> dict:
        a = self.acc
        mse = a[0] / a[1]
        dr = a[3] - a[2]
        return {"psnr": 10.0 * (2 * torch.log10(dr) - torch.log10(mse)), "ssim": a[4] / a[6], "ms_ssim": a[5] / a[6],
                "l1": a[7] / a[1], "l2": mse, "poisson": a[8] / a[6]}


class XMMMetricCollection:
    """reference signature: XMMMetricCollection(metrics, dataset_normalizer, scaling_normalizers, prefix); `metrics` is the
    tuple of metric names (optionally prefixed, e.g. "in/psnr") instead of a torchmetrics MetricCollection."""

    def __init__(self, metrics, dataset_normalizer, scaling_normalizers: List, prefix: str):
        self.names = tuple(metrics)
        for n in self.names:
            if n.split("/")[-1] not in NAMES:
                raise NotImplementedError(f"metric {n}: only {NAMES} run on the MI355X engine (SURVEY.md section 8f-4)")
        self.dataset_normalizer = dataset_normalizer
        self.normalizer_dict = {n.stretch_mode: n for n in scaling_normalizers}
        self.prefix = prefix
        self._eval = Loss({"l1": 1.0, "poisson": 1.0, "psnr": 1.0, "ssim": 1.0, "ms_ssim": 1.0})
        self.reset()

    def reset(self):
        self.states = {mode: _State() for mode in self.normalizer_dict}

    @torch.no_grad()
    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        preds = self.dataset_normalizer.denorm(preds)
        target = self.dataset_normalizer.denorm(target)
        nimg = preds.shape[0]
        for mode, normalizer in self.normalizer_dict.items():
            p = normalizer.norm(preds).contiguous()
            t = normalizer.norm(target).contiguous()
            out, _ = self._eval._eval(p, t, False)
            self.states[mode].add(out, p.numel(), nimg)

    def compute(self) -> dict:
        res = {}
        for mode, st in self.states.items():
            vals = st.compute()
            for n in self.names:
                head, _, base = n.rpartition("/")
                key = f"{self.prefix}/{mode}/{n}"
                res[key] = vals[base].float()
        return res


def get_metrics(dataset_normalizer, scaling_normalizers: List, prefix: str) -> XMMMetricCollection:
    return XMMMetricCollection(NAMES, dataset_normalizer, scaling_normalizers, prefix)


def get_in_metrics(dataset_normalizer, scaling_normalizers: List, prefix: str) -> XMMMetricCollection:
    return XMMMetricCollection(tuple("in/" + n for n in NAMES), dataset_normalizer, scaling_normalizers, prefix)
