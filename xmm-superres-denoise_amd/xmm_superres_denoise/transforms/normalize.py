"""Normalize with the reference's API (transforms/normalize.py:35-107), computed by the HIP normalize kernel.

Deviation, documented: the reference's denormalize_lr_image/_hr_image raise IndexError as written (they index the
0-dim self.lr_max with [:, None, None, None], normalize.py:88,103-107); here they work and equal
denormalize_image(image, max_val.expand(B)).
"""
from __future__ import annotations

import functools

import torch

from xmm_superres_denoise.engine import normalize as _hip_normalize

_MODES = ("linear", "sqrt", "asinh", "log")


def _stretch(x: torch.Tensor, mode: str, inverse: bool) -> torch.Tensor:
    return _hip_normalize(x.contiguous(), 1.0, mode, inverse=inverse)


class Normalize:
    def __init__(self, lr_max: float, hr_max: float, stretch_mode: str = "linear"):
        assert isinstance(stretch_mode, str)
        if stretch_mode not in _MODES:
            raise ValueError(f"Stretching function {stretch_mode} is not implemented")
        self.stretch_mode = stretch_mode
        self.lr_max: torch.Tensor = torch.tensor(lr_max)
        self.hr_max: torch.Tensor = torch.tensor(hr_max)
        # The reference exposes the bare stretch functions as .norm/.denorm (used on [0,1] images by
        # XMMMetricCollection.update, metrics/xmm_metric_collection.py:136-143).  On [0,1] inputs the fused kernel with
        # max_val = 1 is exactly the stretch: clamp(0,1) and /1 are identities there.
        # (functools.partial of a module-level function, not a lambda: the object stays picklable like the reference's)
        self.norm = functools.partial(_stretch, mode=self.stretch_mode, inverse=False)
        self.denorm = functools.partial(_stretch, mode=self.stretch_mode, inverse=True)

    def normalize_image(self, image: torch.Tensor, max_val) -> torch.Tensor:
        mv = float(max_val)
        if mv <= 0:  # reference :72-74 -- no clamp, divide by the image maximum
            image = image / image.max()
            mv = 1.0
        return _hip_normalize(image.contiguous(), mv, self.stretch_mode, inverse=False)

    def denormalize_image(self, image: torch.Tensor, max_val):
        mv = torch.as_tensor(max_val, dtype=torch.float32).flatten()
        if mv.numel() == 1 or bool((mv == mv[0]).all()):
            return _hip_normalize(image.contiguous(), float(mv[0]), self.stretch_mode, inverse=True)
        outs = [_hip_normalize(image[i].contiguous(), float(mv[i]), self.stretch_mode, inverse=True)
                for i in range(image.shape[0])]
        return torch.stack(outs, 0)

    def normalize_lr_image(self, image: torch.Tensor) -> torch.Tensor:
        return self.normalize_image(image, max_val=self.lr_max)

    def normalize_hr_image(self, image):
        if image is None:
            return None
        return self.normalize_image(image, max_val=self.hr_max)

    def denormalize_lr_image(self, image: torch.Tensor):
        return self.denormalize_image(image, max_val=self.lr_max)

    def denormalize_hr_image(self, image):
        return self.denormalize_image(image, max_val=self.hr_max)
