"""Host-side pieces of the input path the engine needs around its kernels (reference data/tools.py:79-126,
data/dataset.py:24-49): a dependency-free FITS primary-HDU reader (the reference uses astropy) and the fused
detector-mask * pad (* normalize) entry point."""
from __future__ import annotations

import gzip

import numpy as np
import torch

from xmm_superres_denoise.engine import mask_pad_normalize as _hip_mask_pad_normalize


def load_fits(fits_path) -> torch.Tensor:
    """PRIMARY HDU image as float32 [1,H,W] (reference data/tools.py:79-86)."""
    a = read_fits_primary(fits_path)
    return torch.from_numpy(a.astype(np.float32)).unsqueeze(0)


def read_fits_primary(path) -> np.ndarray:
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        raw = f.read()
    hdr, off, done = {}, 0, False
    while not done:
        blk = raw[off:off + 2880]
        if len(blk) < 2880:
            raise ValueError(f"{path}: truncated FITS header")
        off += 2880
        for i in range(36):
            card = blk[i * 80:(i + 1) * 80].decode("ascii", "replace")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] == "= ":
                hdr[key] = card[10:].split("/")[0].strip().strip("'").strip()
    dt = {8: "u1", 16: ">i2", 32: ">i4", -32: ">f4", -64: ">f8"}[int(hdr["BITPIX"])]
    n1, n2 = int(hdr["NAXIS1"]), int(hdr["NAXIS2"])
    a = np.frombuffer(raw, dtype=dt, count=n1 * n2, offset=off).reshape(n2, n1)
    bz, bs = float(hdr.get("BZERO", 0.0)), float(hdr.get("BSCALE", 1.0))
    if bz != 0.0 or bs != 1.0:
        a = a.astype(np.float64) * bs + bz
    return np.ascontiguousarray(a.astype(a.dtype.newbyteorder("=")))


def reshape_img_to_res(res: int, img: torch.Tensor) -> torch.Tensor:
    """Centred zero pad / crop of [1,H,W] (or [B,H,W]) to [*,res,res] on the GPU (reference data/tools.py:103-126)."""
    return _hip_mask_pad_normalize(img.contiguous(), None, res, None)[:, 0]


def load_and_prepare(counts: torch.Tensor, det_mask: torch.Tensor | None, res: int, max_val: float | None = None,
                     stretch: str = "linear") -> torch.Tensor:
    """counts [B,Hin,Win] (int32 or float32, CUDA) -> img *= mask -> pad to res -> optional normalize, one kernel
    (reference data/dataset.py:41-47 + :267-268)."""
    return _hip_mask_pad_normalize(counts.contiguous(), det_mask, res, max_val, stretch)
