"""CPU restatement of the arithmetic of math mode "f16x3" (csrc/xsd_split.h: scale_for_amax, split2_f16x4) in numpy, and the
properties DESIGN.md section 4 claims for it.  The device code itself is exercised by the GPU tests (test_hip_precision.py:
error against float64, range robustness); this file pins the algorithm they implement."""
import numpy as np


def scale_for_amax(amax: np.float32) -> float:
    """power of two s with amax * s in [2^13, 2^14); 1 for 0 / subnormal / inf / nan; exponent clamped to +-60"""
    u = np.float32(amax).view(np.uint32)
    ex = int((u >> 23) & 0xFF)
    k = 0 if ex in (0, 255) else 140 - ex
    k = max(-60, min(60, k))
    return float(2.0 ** k)


def split2(x: np.ndarray, s: float):
    """x * s = h + l * 2^-11 with h, l fp16 (round to nearest even at each step; x * s and the residual are exact in fp32)"""
    xs = (x.astype(np.float32) * np.float32(s)).astype(np.float32)
    h = xs.astype(np.float16)
    r = (xs - h.astype(np.float32)).astype(np.float32)
    l = (r * np.float32(2048.0)).astype(np.float16)
    return h, l, xs


def test_scale_puts_the_maximum_at_the_top_of_the_fp16_range():
    rng = np.random.default_rng(0)
    for e in range(-40, 41, 3):
        amax = np.float32(rng.uniform(1.0, 2.0) * 2.0 ** e)
        s = scale_for_amax(amax)
        assert np.log2(s) == round(np.log2(s))                 # a power of two: scaling and un-scaling are exact
        assert 2.0 ** 13 <= float(amax) * s < 2.0 ** 14, (amax, s)
    assert scale_for_amax(np.float32(0.0)) == 1.0
    assert scale_for_amax(np.float32(np.inf)) == 1.0
    assert scale_for_amax(np.float32(np.nan)) == 1.0
    assert scale_for_amax(np.float32(2.0 ** 100)) == 2.0 ** -60   # clamped (such a tensor overflows fp32 products anyway)


def test_two_term_split_carries_22_bits_over_28_octaves():
    rng = np.random.default_rng(1)
    x = (rng.normal(size=200000) * np.exp2(rng.uniform(-27, 0, size=200000))).astype(np.float32)   # 27 octaves of dynamic range
    x[0] = np.float32(1.9999)                                                                   # the maximum
    s = scale_for_amax(np.abs(x).max())
    h, l, xs = split2(x, s)
    assert np.isfinite(h.astype(np.float32)).all() and np.isfinite(l.astype(np.float32)).all()
    rec = h.astype(np.float64) + l.astype(np.float64) * 2.0 ** -11
    rel = np.abs(rec - xs.astype(np.float64)) / np.maximum(np.abs(xs.astype(np.float64)), 1e-300)
    big = np.abs(xs) >= 2.0 ** -14                               # h is a normal fp16 number there: elements above 2^-28 of the maximum
    assert big.mean() > 0.9
    assert rel[big].max() <= 2.0 ** -22                          # half an ulp of l: 22-23 significant bits
    assert np.sqrt(np.mean(rel[big] ** 2)) < 2.0 ** -23.5
    # below that the error is absolute, not relative: 2^-36 in scaled units = 2^-50 of the maximum
    assert np.abs(rec - xs.astype(np.float64))[~big].max() <= 2.0 ** -36


def test_power_of_two_rescaling_changes_nothing():
    rng = np.random.default_rng(2)
    x = rng.normal(size=4096).astype(np.float32)
    h0, l0, _ = split2(x, scale_for_amax(np.abs(x).max()))
    for k in (-40, -7, 11, 40):        # (inside the +-60 clamp of the scale's exponent)
        y = (x * np.float32(2.0 ** k)).astype(np.float32)
        h1, l1, _ = split2(y, scale_for_amax(np.abs(y).max()))
        assert np.array_equal(h0.view(np.uint16), h1.view(np.uint16)) and np.array_equal(l0.view(np.uint16), l1.view(np.uint16))


def test_three_products_reproduce_the_fp32_product_to_2_pow_minus_21():
    rng = np.random.default_rng(3)
    x = rng.normal(size=50000).astype(np.float32)
    w = (rng.normal(size=50000) * 0.05).astype(np.float32)
    sx, sw = scale_for_amax(np.abs(x).max()), scale_for_amax(np.abs(w).max())
    hx, lx, _ = split2(x, sx)
    hw, lw, _ = split2(w, sw)
    hx, lx, hw, lw = (a.astype(np.float64) for a in (hx, lx, hw, lw))
    prod = (hw * hx + (hw * lx + lw * hx) * 2.0 ** -11) / (sx * sw)       # the three MFMA products, un-scaled
    exact = x.astype(np.float64) * w.astype(np.float64)
    rel = np.abs(prod - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -21                                         # dropped l*l and two half-ulps of l
    assert np.sqrt(np.mean(rel ** 2)) < 2.0 ** -22.5
