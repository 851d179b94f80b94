"""Shared, reference-free helpers for the golden fixtures.

Everything here is numpy-only and is used by BOTH
  * tests/golden/make_golden.py  (runs only in the build container; imports the reference), and
  * the tests / smoke / bench (run anywhere; never touch /root/reference).

The fixtures do not store network weights: weights and inputs are regenerated from numpy's PCG64
stream (deterministic across platforms), in the reference's state-dict order
(reference key names: SURVEY.md section 5, verified against the reference's state_dict() by make_golden.py).
"""
from __future__ import annotations

import gzip
from collections import OrderedDict

import numpy as np


def rrdb_param_shapes(kind: str, nf: int, blocks: int, in_ch: int = 1, out_ch: int = 1,
                      num_upsample: int = 1) -> "OrderedDict[str, tuple]":
    """Parameter names/shapes of GeneratorRRDB_DN ('dn') / GeneratorRRDB_SR ('sr').

    Mirrors the registration order of the reference constructors
    (models/modules/generator_rrdb.py:10-64,73-101; rrdb_blocks.py:23-32,60-64).
    """
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(name, cout, cin):
        s[name + ".weight"] = (cout, cin, 3, 3)
        s[name + ".bias"] = (cout,)

    conv("conv_first", nf, in_ch)
    for i in range(blocks):
        for r in (1, 2, 3):
            for c in (1, 2, 3, 4, 5):
                conv(f"rrdb.{i}.RDB{r}.conv{c}", nf, nf * c)  # gc == nf in the reference factory
    conv("trunk_conv", nf, nf)
    conv("conv_last", out_ch, nf)
    if kind == "sr":
        for u in range(num_upsample):
            conv(f"upsampling.{3 * u}", 4 * nf, nf)
        conv("HRconv", nf, nf)
    elif kind != "dn":
        raise ValueError(kind)
    return s


def make_state(kind: str, nf: int, blocks: int, seed: int, num_upsample: int = 1,
               gain: float = 1.0, last_bias: float | None = None, in_ch: int = 1, out_ch: int = 1) -> "OrderedDict[str, np.ndarray]":
    """Deterministic fp32 weights: U(-b, b), b = gain/sqrt(fan_in), drawn in state-dict order."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    shapes = rrdb_param_shapes(kind, nf, blocks, in_ch=in_ch, out_ch=out_ch, num_upsample=num_upsample)
    fan_in = 1
    for name, shp in shapes.items():
        if name.endswith(".weight"):
            fan_in = shp[1] * 9
        b = gain / np.sqrt(fan_in)
        out[name] = rng.uniform(-b, b, size=shp).astype(np.float32)
    if last_bias is not None:
        out["conv_last.bias"] = np.full_like(out["conv_last.bias"], last_bias)
    return out


def make_input(shape, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).uniform(0.0, 1.0, size=shape).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# Minimal FITS primary-HDU reader (the reference uses astropy, data/tools.py:79-86; astropy is not
# installed here).  Handles BITPIX 8/16/32/-32/-64, 2-D images, optional gzip.
# ----------------------------------------------------------------------------------------------
def read_fits_primary(path: str):
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        raw = f.read()
    hdr = {}
    off = 0
    done = False
    while not done:
        blk = raw[off:off + 2880]
        off += 2880
        for i in range(36):
            card = blk[i * 80:(i + 1) * 80].decode("ascii", "replace")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] == "= ":
                hdr[key] = card[10:].split("/")[0].strip().strip("'").strip()
    bitpix = int(hdr["BITPIX"])
    n1, n2 = int(hdr["NAXIS1"]), int(hdr["NAXIS2"])
    dt = {8: "u1", 16: ">i2", 32: ">i4", -32: ">f4", -64: ">f8"}[bitpix]
    a = np.frombuffer(raw, dtype=dt, count=n1 * n2, offset=off).reshape(n2, n1)
    bzero = float(hdr.get("BZERO", 0.0))
    bscale = float(hdr.get("BSCALE", 1.0))
    if bzero != 0.0 or bscale != 1.0:
        a = a.astype(np.float64) * bscale + bzero
    return np.ascontiguousarray(a), hdr


def psnr(pred: np.ndarray, target: np.ndarray, data_range: float = 1.0) -> float:
    mse = float(np.mean((pred.astype(np.float64) - target.astype(np.float64)) ** 2))
    return 10.0 * np.log10(data_range ** 2 / mse) if mse > 0 else float("inf")
