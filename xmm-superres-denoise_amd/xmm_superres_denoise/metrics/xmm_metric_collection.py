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

from ..utils.loss_functions import EpochState, Loss

NAMES = ("psnr", "ssim", "ms_ssim", "l1", "l2", "poisson")


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
        self.states = {mode: EpochState() for mode in self.normalizer_dict}

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

    def sync(self, group=None) -> None:
        """all-reduce the per-mode epoch states over the ranks (sum / min / max), see EpochState.sync"""
        for st in self.states.values():
            st.sync(group)

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
