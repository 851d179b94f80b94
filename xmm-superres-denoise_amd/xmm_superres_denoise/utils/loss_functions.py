"""Host mirror of the reference's loss factory (xmm_superres_denoise/utils/loss_functions.py:11-47).

`create_loss(sc_dict, loss_config)` keeps the reference's signature and arithmetic: for every term with a relative
percentage p > 0, weight = p * scaling (when a scaling table is given) and the corrections are summed; the result is
`sum_i weight_i * metric_i(preds, target)` plus the correction when that sum is > 0.  What it returns is a `Loss`: a
callable `(preds, target) -> scalar tensor` (like the torchmetrics CompositionalMetric's forward, models/model.py:78),
differentiable with respect to preds, evaluated entirely by the HIP kernels behind `xsd_loss_eval` (include/xsd.h).
There is no CPU fallback.
"""
from __future__ import annotations

import ctypes

import torch
from torch.autograd.function import once_differentiable

from ..config.config import LossCfg
from ..engine._lib import XsdError, check, load
from ..engine.engine import _require_cuda_f32, _stream_ptr

TERMS = ("l1", "poisson", "psnr", "ssim", "ms_ssim")

# res/configs/loss_functions.toml:5-42 of the reference: default percentages and the paper's scaling/correction tables
LOSS_TOML = {
    "loss": dict(use_scaling=True, l1=0.0, poisson=0.0, psnr=0.5, ssim=0.0, ms_ssim=0.5),
    "scaling": {
        "linear": {"l1": (27.404768429706774, -0.5746779939709512), "poisson": (6.583278472679395, -1.187623436471363),
                   "psnr": (-0.11938872970391594, 3.6491165234001905), "ssim": (-2.97441998810232, 2.1469363474122547),
                   "ms_ssim": (-2.85143997718848, 2.737382378100941)},
        "sqrt": {"l1": (9.65623792970259, -0.5189262263422172), "poisson": (12.269938650306754, -5.137423312883438),
                 "psnr": (-0.121713729308666, 2.7966163583252186), "ssim": (-3.0684258975145746, 1.417919607241485),
                 "ms_ssim": (-3.0165912518853695, 2.636500754147813)},
        "asinh": {"l1": (5.651952749675013, -0.4542474424913807), "poisson": (0.4388467108439022, -0.22920963707377018),
                  "psnr": (-0.11042402826855124, 2.1554770318021204), "ssim": (-3.2824552765468566, 1.2020351222714591),
                  "ms_ssim": (-1.6189088554314395, 1.3368949328152826)},
        "log": {"l1": (4.071661237785016, -0.4364820846905537), "poisson": (0.39835876190096803, -0.2616021989403656),
                "psnr": (-0.1108524553818867, 1.8665336437202082), "ssim": (-3.414600833162603, 1.176671447107833),
                "ms_ssim": (-2.043318348998774, 1.6309767061708214)},
    },
}


class _LossConfigC(ctypes.Structure):
    _fields_ = [("w_l1", ctypes.c_float), ("w_poisson", ctypes.c_float), ("w_psnr", ctypes.c_float),
                ("w_ssim", ctypes.c_float), ("w_ms_ssim", ctypes.c_float), ("correction", ctypes.c_float),
                ("sigma", ctypes.c_float), ("k1", ctypes.c_float), ("k2", ctypes.c_float), ("kernel_size", ctypes.c_int32)]


def load_loss_config(scaling: str = "linear", **overrides):
    """What train.py:46-53 of the reference does with loss_functions.toml: returns (sc_dict | None, LossCfg)."""
    d = dict(LOSS_TOML["loss"])
    d.update(overrides)
    sc = None
    if d.pop("use_scaling"):
        sc = {k: {"scaling": v[0], "correction": v[1]} for k, v in LOSS_TOML["scaling"][scaling].items()}
    return sc, LossCfg(**d)


class EpochState:
    """Per-epoch states of the torchmetrics classes behind the loss terms / validation metrics, accumulated on the
    device from the per-batch result vector of xsd_loss_eval: sum of squared errors and element count, running
    min/max of the target (both start at 0, PeakSignalNoiseRatio(data_range=None)), per-image ssim / ms_ssim sums,
    absolute-error sum, sum of per-batch poisson means (metrics/metrics.py:30-39 divides that by the image count)."""

    def __init__(self):
        self.acc = None      # [sse, n, tmin, tmax, ssim_sum, ms_sum, nimg, abs_sum, poisson_sum]

    def add(self, out: torch.Tensor, n: int, nimg: int):
        # out: [total, l1, poisson, psnr, ssim, ms_ssim, mse, tmin, tmax, ...] of one batch
        cur = torch.stack([out[6] * n, out.new_tensor(float(n)), out[7], out[8], out[4] * nimg, out[5] * nimg,
                           out.new_tensor(float(nimg)), out[1] * n, out[2] * nimg]).double()
        if self.acc is None:
            zero = torch.zeros((), dtype=torch.float64, device=out.device)
            cur[2] = torch.minimum(cur[2], zero)
            cur[3] = torch.maximum(cur[3], zero)
            self.acc = cur
        else:
            a = self.acc
            self.acc = torch.stack([a[0] + cur[0], a[1] + cur[1], torch.minimum(a[2], cur[2]), torch.maximum(a[3], cur[3]),
                                    a[4] + cur[4], a[5] + cur[5], a[6] + cur[6], a[7] + cur[7], a[8] + cur[8]])

    def sync(self, group=None, device=None) -> None:
        """Reduce the STATES over the ranks before compute(), as torchmetrics does (dist_reduce_fx: "sum" for the sums and
        counts, "min" / "max" for the target range; reference metrics/metrics.py:16-21): PSNR of the pooled squared error,
        not a mean of per-rank PSNRs.  A rank that saw no batch (uneven validation shards) still takes part in every
        collective, with the identity state -- all zeros: the sums' identity, and the target range starts at [0, 0] on every
        rank anyway -- so the ranks' collective sequences always match.  If NO rank saw a batch the state stays empty."""
        import torch.distributed as dist
        from xmm_superres_denoise.parallel import collectives_on
        if not collectives_on(group):      # one rank (unless XSD_FORCE_DP=1 asks for the collectives anyway)
            return
        if self.acc is not None:
            a = self.acc.clone()
        else:
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            a = torch.zeros(9, dtype=torch.float64, device=device)
        sums = a[[0, 1, 4, 5, 6, 7, 8]].contiguous()
        lo, hi = a[2:3].contiguous(), a[3:4].contiguous()
        from xmm_superres_denoise.parallel import all_reduce_any
        all_reduce_any(sums, dist.ReduceOp.SUM, group)
        all_reduce_any(lo, dist.ReduceOp.MIN, group)
        all_reduce_any(hi, dist.ReduceOp.MAX, group)
        if float(sums[1]) == 0.0:
            self.acc = None
            return
        self.acc = torch.stack([sums[0], sums[1], lo[0], hi[0], sums[2], sums[3], sums[4], sums[5], sums[6]])

    def compute(self) -> dict:
        a = self.acc
        mse = a[0] / a[1]
        dr = a[3] - a[2]
        return {"psnr": 10.0 * (2 * torch.log10(dr) - torch.log10(mse)), "ssim": a[4] / a[6], "ms_ssim": a[5] / a[6],
                "l1": a[7] / a[1], "l2": mse, "poisson": a[8] / a[6]}


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, target, loss, need_grad):
        out, dy = loss._eval(preds, target, need_grad)
        loss.last_values = out
        if need_grad:
            ctx.save_for_backward(dy)
        return out[0].clone()

    @staticmethod
    @once_differentiable      # the backward is kernels, not torch ops: a second differentiation (create_graph=True) is refused by name
    def backward(ctx, g):
        (dy,) = ctx.saved_tensors
        return dy * g, None, None, None


class Loss:
    """weights: {term: effective weight}; correction: summed corrections (added when > 0, loss_functions.py:44-45)."""

    def __init__(self, weights: dict, correction: float = 0.0, kernel_size: int = 13, sigma: float = 2.5,
                 k1: float = 0.01, k2: float = 0.05):
        unknown = set(weights) - set(TERMS)
        if unknown:
            raise XsdError(f"unknown loss terms {sorted(unknown)}")
        self.weights = {k: float(weights.get(k, 0.0)) for k in TERMS}
        self.correction = float(correction)
        self._ctor = dict(weights=dict(self.weights), correction=self.correction, kernel_size=kernel_size, sigma=sigma, k1=k1, k2=k2)
        self.L = load()
        cfg = _LossConfigC(*[self.weights[k] for k in TERMS], self.correction, sigma, k1, k2, kernel_size)
        h = ctypes.c_void_p()
        check(self.L.xsd_loss_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        self.last_values = None   # device tensor [12] of the last call: total, l1, poisson, psnr, ssim, ms_ssim, mse, min/max(target)
        self._epoch = EpochState()

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.xsd_loss_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __getstate__(self):      # a copy / an unpickled loss creates its own handle (the handle is a pointer of this process)
        return dict(self._ctor)

    def __setstate__(self, st):
        self.__init__(**st)

    def _eval(self, preds, target, want_grad):
        if isinstance(preds, torch.Tensor) and preds.is_cuda:      # the C side allocates its workspace and launches on the current device
            with torch.cuda.device(preds.device):
                return self._eval_on_device(preds, target, want_grad)
        return self._eval_on_device(preds, target, want_grad)

    def _eval_on_device(self, preds, target, want_grad):
        _require_cuda_f32(preds, "preds")
        _require_cuda_f32(target, "target")
        if preds.shape != target.shape:
            raise XsdError(f"shape mismatch {tuple(preds.shape)} vs {tuple(target.shape)}")
        if preds.dim() == 4:
            # [B, C, H, W]: the kernels take one-channel images, so the channels fold into the batch (a contiguous NCHW tensor is
            # B*C images) and the loss is told how many images form a sample (xsd_loss_set_channels): l1, psnr (means over all
            # elements) and ssim (mean over samples of the channel means) do not care; the Poisson term divides by the number of
            # SAMPLES (metrics/metrics.py:30-39) and MS-SSIM averages each scale's statistic over a sample's channels before the
            # product over scales, as torchmetrics does.
            B, C, H, W = preds.shape
            channels = C
            B = B * C
        elif preds.dim() == 3:
            B, H, W = preds.shape
            channels = 1
        else:
            raise XsdError(f"expected [B,C,H,W] or [B,H,W], got {tuple(preds.shape)}")
        out = torch.empty(12, device=preds.device, dtype=torch.float32)
        dy = torch.empty_like(preds) if want_grad else None
        check(self.L.xsd_loss_set_channels(self.h, channels))
        check(self.L.xsd_loss_eval(self.h, preds.data_ptr(), target.data_ptr(), dy.data_ptr() if want_grad else None,
                                   out.data_ptr(), B, H, W, _stream_ptr(preds.device)))
        return out, dy

    def value_and_grad(self, preds, target):
        """engine-level call used by the training driver: (total [scalar tensor], d total / d preds)"""
        out, dy = self._eval(preds, target, True)
        self.last_values = out
        return out[0], dy

    def __call__(self, preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        need = torch.is_grad_enabled() and preds.requires_grad
        return _LossFn.apply(preds, target, self, need)

    # ---- epoch-level protocol used by the validation / test branches of Model._on_step (models/model.py:87-88,109-120):
    # the composed metric's compute() is the same weighted sum over the terms' EPOCH-level values
    @torch.no_grad()
    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        out, _ = self._eval(preds.contiguous(), target.contiguous(), False)
        self.last_values = out
        self._epoch.add(out, preds.numel(), preds.shape[0])

    def sync(self, group=None) -> None:
        self._epoch.sync(group)

    def compute(self) -> torch.Tensor:
        vals = self._epoch.compute()
        total = sum(w * vals[k] for k, w in self.weights.items() if w != 0.0)
        if self.correction > 0.0:
            total = total + self.correction
        return total.float()

    def reset(self) -> None:
        self._epoch = EpochState()

    def term_values(self) -> dict:
        v = self.last_values.tolist()
        return {"total": v[0], **{k: v[1 + i] for i, k in enumerate(TERMS) if self.weights[k] != 0.0}}

    def __repr__(self):
        terms = " + ".join(f"{w:g}*{k}" for k, w in self.weights.items() if w != 0.0)
        return f"Loss({terms}{f' + {self.correction:g}' if self.correction > 0 else ''})"


def create_loss(sc_dict: dict | None, loss_config: LossCfg) -> Loss:
    """reference signature: create_loss(sc_dict: dict[str, dict[str, float]] | None, loss_config: LossCfg) -> Metric"""
    correction = 0.0
    weights = {}
    for loss, p in iter(loss_config):
        if p > 0.0:
            if sc_dict is not None and loss in sc_dict:
                p = p * sc_dict[loss]["scaling"]
                correction = correction + sc_dict[loss]["correction"]
            weights[loss] = p
    assert weights
    return Loss(weights, correction)
