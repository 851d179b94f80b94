"""Inference on one det-xy FITS image: the engine-side counterpart of the reference's `utils/run_inference_on_file.py`
(:101-200, stale in the snapshot) and of the WCS header surgery in `utils/filehandling.py:131-247`.

FITS in -> (detector mask, centred pad, Normalize) in one HIP kernel -> generator -> denormalize (HIP) -> FITS out with
the reference's WCS bookkeeping.  No astropy: a minimal primary-HDU reader/writer lives here (2880-byte blocks).
The XMM-SAS steps that produce the det-xy image (`utils/xmmsas_tools.py`) are out of scope (external ESA toolchain).
"""
from __future__ import annotations

import gzip
import math
import os
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch

# header keys the reference drops when it copies the input header (utils/filehandling.py:148-196)
_OMIT = {"SIMPLE", "BITPIX", "NAXIS", "NAXIS1", "NAXIS2", "EXTEND", "XPROC0", "XDAL0", "CREATOR", "DATE",
         "CTYPE1L", "CRPIX1L", "CRVAL1L", "CDELT1L", "LTV1", "LTM1_1", "CTYPE2L", "CRPIX2L", "CRVAL2L", "CDELT2L",
         "LTV2", "LTM2_2", "LTM1_2", "LTM2_1", "EXPOSURE", "DURATION"} | {f"ONTIME{i:02d}" for i in range(1, 13)}


def _parse_value(txt: str):
    t = txt.strip()
    if t.startswith("'"):
        return t.strip("'").rstrip()
    if t in ("T", "F"):
        return t == "T"
    try:
        return int(t)
    except ValueError:
        try:
            return float(t.replace("D", "E"))
        except ValueError:
            return t


def read_fits(path) -> tuple[np.ndarray, "OrderedDict[str, object]"]:
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        raw = f.read()
    hdr: "OrderedDict[str, object]" = OrderedDict()
    off, done = 0, False
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
                val = card[10:]
                if val.lstrip().startswith("'"):
                    end = val.find("'", val.find("'") + 1)
                    val = val[:end + 1]
                else:
                    val = val.split("/")[0]
                hdr[key] = _parse_value(val)
    dt = {8: "u1", 16: ">i2", 32: ">i4", -32: ">f4", -64: ">f8"}[int(hdr["BITPIX"])]
    n1, n2 = int(hdr["NAXIS1"]), int(hdr["NAXIS2"])
    a = np.frombuffer(raw, dtype=dt, count=n1 * n2, offset=off).reshape(n2, n1)
    return a, hdr


def _card(key: str, value, comment: str = "") -> bytes:
    if isinstance(value, bool):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, int):
        v = f"{value:>20d}"
    elif isinstance(value, float):
        v = f"{value:>20.12G}"
    else:
        s = str(value).replace("'", "''")
        v = f"'{s:<8}'"
    c = f"{key:<8}= {v}"
    if comment:
        c += f" / {comment}"
    return c[:80].ljust(80).encode("ascii", "replace")


def write_fits(path, img: np.ndarray, header: "OrderedDict[str, object]", comments=()) -> None:
    img = np.ascontiguousarray(img, dtype=">f4")
    cards = [_card("SIMPLE", True), _card("BITPIX", -32), _card("NAXIS", 2), _card("NAXIS1", img.shape[1]), _card("NAXIS2", img.shape[0])]
    for k, v in header.items():
        if k in ("SIMPLE", "BITPIX", "NAXIS", "NAXIS1", "NAXIS2", "END") or v is None:
            continue
        cards.append(_card(k, v))
    for c in comments:
        cards.append(f"COMMENT {c}"[:80].ljust(80).encode("ascii", "replace"))
    cards.append(b"END".ljust(80))
    blob = b"".join(cards)
    blob += b" " * (-len(blob) % 2880)
    data = img.tobytes()
    data += b"\0" * (-len(data) % 2880)
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wb") as f:
        f.write(blob + data)


def wcs_header(in_header, source_file_name: str, res_mult: int, exposure) -> "OrderedDict[str, object]":
    """Header of the output image: the input header minus the reference's omit list, reference pixel shifted by the
    centred pad (+6, +2; 403x411 -> 416x416) and, for the 2x model, rescaled WCS with a CD matrix from PA_PNT
    (utils/filehandling.py:199-226)."""
    h: "OrderedDict[str, object]" = OrderedDict()
    h["IMG_FILE"] = source_file_name
    for k, v in in_header.items():
        if k not in _OMIT:
            h[k] = v
    h["EXPOSURE"] = exposure
    if "CRPIX1" in h and "CRPIX2" in h:
        c1, c2 = float(h["CRPIX1"]) + 6, float(h["CRPIX2"]) + 2
        h["CRPIX1"], h["CRPIX2"] = c1, c2
        if res_mult == 2:
            h["CRPIX1"], h["CRPIX2"] = res_mult * c1 + 0.5, res_mult * c2 + 0.5
            d1, d2 = float(h["CDELT1"]) / res_mult, float(h["CDELT2"]) / res_mult
            h["CDELT1"], h["CDELT2"] = d1, d2
            rot = 90.0 - float(h["PA_PNT"])
            h["CROT2"] = rot
            r = math.radians(rot)
            h["CD1_1"], h["CD1_2"] = d1 * math.cos(r), -1.0 * d2 * math.sin(r)
            h["CD2_1"], h["CD2_2"] = d1 * math.sin(r), d2 * math.cos(r)
    return h


@torch.no_grad()
def infer_file(fits_path, model, det_mask: torch.Tensor | None, out_dir, lr_res: int = 416, lr_max: float = 0.0022336,
               hr_max: float = 0.0005584, stretch: str = "sqrt", device="cuda:0", write_input: bool = True):
    """Run one det-xy image through `model` (GeneratorRRDB_DN / _SR on `device`).  Returns (prediction [Hout,Wout] in
    physical units, output path).  Mirrors run_inference_on_file.py:127-199 without the exposure bookkeeping against SAS."""
    from xmm_superres_denoise.engine import compose_input, normalize
    data, hdr = read_fits(fits_path)
    exposure = hdr.get("EXPOSURE", 0)
    kind32 = data.dtype.kind == "i" and data.dtype.itemsize == 4
    raw = torch.from_numpy(np.ascontiguousarray(data).view(np.int32 if kind32 else np.float32).copy())[None].to(device) \
        if data.dtype.itemsize == 4 and data.dtype.kind in "if" else torch.from_numpy(data.astype(np.float32))[None].to(device)
    big_endian = data.dtype.itemsize == 4 and data.dtype.kind in "if" and data.dtype.byteorder == ">"
    x = compose_input(raw, None, None, det_mask, lr_res, lr_max, stretch, big_endian=big_endian)
    with torch.no_grad():       # the inference plan (9 recycled planes), not the training plan that keeps every activation
        y = model(x)
    res_mult = y.shape[-1] // x.shape[-1]
    y_phys = normalize(y.contiguous(), hr_max if res_mult > 1 else lr_max, stretch, inverse=True)[0, 0].cpu().numpy()
    os.makedirs(out_dir, exist_ok=True)
    base = os.path.basename(str(fits_path)).replace(".gz", "").replace(".fits", "")
    stamp = datetime.now().strftime("%d/%m/%Y %H:%M:%S")
    out_path = os.path.join(out_dir, f"{base}_{'sr' if res_mult > 1 else 'dn'}_predict.fits.gz")
    write_fits(out_path, y_phys, wcs_header(hdr, os.path.basename(str(fits_path)), res_mult, exposure),
               comments=("MI355X engine output (xmm-superres-denoise_amd)", f"File created on {stamp}"))
    if write_input:
        x_phys = normalize(x.contiguous(), lr_max, stretch, inverse=True)[0, 0].cpu().numpy()
        write_fits(os.path.join(out_dir, f"{base}_input.fits.gz"), x_phys, wcs_header(hdr, os.path.basename(str(fits_path)), 1, exposure))
    return y_phys, out_path
