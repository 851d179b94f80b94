"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Three parts:
  * ctypes bindings of oracle/libxsd_oracle.so (C restatement of the RRDB generators, xsd_oracle.c);
  * numpy restatement of the input transforms (detector mask, centred pad, Normalize, ImageUpsample);
  * a torch.nn.functional restatement of the same op graph, used ONLY as the timed CPU baseline
    ("cpu_baseline.kind = port") because oneDNN-backed torch convs are the strongest CPU baseline here.
All are pinned against tests/golden/*.npz (generated from the reference itself) by tests/test_oracle_pinned.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
KIND = {"dn": 0, "sr": 1}


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libxsd_oracle.so")
    src = os.path.join(_HERE, "xsd_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "libxsd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        L.xsd_oracle_param_count.restype = ctypes.c_long
        L.xsd_oracle_param_count.argtypes = [ctypes.c_int] * 4
        L.xsd_oracle_param_count_c.restype = ctypes.c_long
        L.xsd_oracle_param_count_c.argtypes = [ctypes.c_int] * 6
        L.xsd_oracle_forward.argtypes = [ctypes.c_int] * 4 + [fp, fp] + [ctypes.c_int] * 3 + [fp]
        L.xsd_oracle_forward_c.argtypes = [ctypes.c_int] * 6 + [fp, fp] + [ctypes.c_int] * 3 + [fp]
        L.xsd_oracle_l1_train_c.argtypes = ([ctypes.c_int] * 6 + [fp, fp, fp] + [ctypes.c_int] * 3 +
                                            [fp, ctypes.POINTER(ctypes.c_double), fp, fp])
        L.xsd_oracle_l1_train.argtypes = ([ctypes.c_int] * 4 + [fp, fp, fp] + [ctypes.c_int] * 3 +
                                          [fp, ctypes.POINTER(ctypes.c_double), fp, fp])
        L.xsd_oracle_conv3x3.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 5
        L.xsd_oracle_conv3x3_bwd.argtypes = [fp] * 6 + [ctypes.c_int] * 5
        L.xsd_oracle_adam.argtypes = [fp, fp, fp, fp, ctypes.c_long, ctypes.c_int] + [ctypes.c_float] * 4
        for f in (L.xsd_oracle_conv3x3, L.xsd_oracle_conv3x3_bwd, L.xsd_oracle_adam):
            f.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def flatten_state(state) -> np.ndarray:
    """state: ordered mapping name->array in state_dict order -> flat fp32 vector."""
    return np.concatenate([_f32(v).ravel() for v in state.values()])


def unflatten(flat: np.ndarray, shapes) -> dict:
    out, off = {}, 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    assert off == flat.size
    return out


def forward(kind, nf, blocks, params, x, num_upsample=1, out_ch=1):
    """x: [B, in_ch, H, W] -> [B, out_ch, sH, sW]; any image channel counts the reference's constructors take
    (generator_rrdb.py:10-16; DN: in_ch == out_ch, or a one-channel x broadcast by `out + x`, :134)"""
    x = _f32(x)
    B, C, H, W = x.shape
    s = 2 ** num_upsample if kind == "sr" else 1
    y = np.empty((B, out_ch, H * s, W * s), np.float32)
    params = _f32(params)
    assert params.size == lib().xsd_oracle_param_count_c(KIND[kind], nf, blocks, num_upsample, C, out_ch)
    rc = lib().xsd_oracle_forward_c(KIND[kind], nf, blocks, num_upsample, C, out_ch, _p(params), _p(x), B, H, W, _p(y))
    if rc != 0:
        raise ValueError(f"oracle: a DN generator cannot add a {C}-channel input to {out_ch} output channels")
    return y


def l1_train(kind, nf, blocks, params, x, target, num_upsample=1):
    """returns y, loss, dx, flat grads; the channel counts are read off x and target"""
    x, target, params = _f32(x), _f32(target), _f32(params)
    B, C, H, W = x.shape
    out_ch = target.shape[1]
    s = 2 ** num_upsample if kind == "sr" else 1
    assert target.shape == (B, out_ch, H * s, W * s)
    assert params.size == lib().xsd_oracle_param_count_c(KIND[kind], nf, blocks, num_upsample, C, out_ch)
    y = np.empty_like(target)
    dx = np.empty_like(x)
    grads = np.zeros_like(params)
    loss = ctypes.c_double(0.0)
    rc = lib().xsd_oracle_l1_train_c(KIND[kind], nf, blocks, num_upsample, C, out_ch, _p(params), _p(x), _p(target), B, H, W,
                                     _p(y), ctypes.byref(loss), _p(dx), _p(grads))
    if rc != 0:
        raise ValueError(f"oracle: a DN generator cannot add a {C}-channel input to {out_ch} output channels")
    return y, loss.value, dx, grads


def conv3x3(x, w, b=None):
    x, w = _f32(x), _f32(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = np.empty((B, Cout, H, W), np.float32)
    lib().xsd_oracle_conv3x3(_p(x), _p(w), _p(None if b is None else _f32(b)), _p(y), B, Cin, Cout, H, W)
    return y


def conv3x3_bwd(x, w, dy):
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty((Cout,), np.float32)
    lib().xsd_oracle_conv3x3_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), B, Cin, Cout, H, W)
    return dx, dw, db


def adam(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """in-place on fp32 contiguous arrays"""
    lib().xsd_oracle_adam(_p(p), _p(g), _p(m), _p(v), p.size, step, lr, b1, b2, eps)


# ---------------------------------------------------------------------------------------------------
# Input transforms, numpy restatement
# ---------------------------------------------------------------------------------------------------
def reshape_img_to_res(img: np.ndarray, res: int) -> np.ndarray:
    """Centred zero-pad / crop of [C,H,W] to [C,res,res]  (data/tools.py:103-126; F.pad with negative pad crops)."""
    C, H, W = img.shape
    yd, xd = res - H, res - W
    yt = int(np.floor(yd / 2.0)); yb = yd - yt
    xl = int(np.floor(xd / 2.0)); xr = xd - xl
    # crop first (negative pads), then pad
    y0, y1 = max(0, -yt), H - max(0, -yb)
    x0, x1 = max(0, -xl), W - max(0, -xr)
    a = img[:, y0:y1, x0:x1]
    return np.pad(a, ((0, 0), (max(0, yt), max(0, yb)), (max(0, xl), max(0, xr))), mode="constant")


def mask_pad(counts: np.ndarray, mask: np.ndarray, res: int) -> np.ndarray:
    """load_fits -> float32, img *= mask, centred pad (data/dataset.py:41-47).  counts,mask: [H,W]."""
    img = counts.astype(np.float32)[None] * mask.astype(np.float32)[None]
    return reshape_img_to_res(img, res)


def _stretch(x, mode):
    if mode == "linear":
        return x
    if mode == "sqrt":
        return np.sqrt(x)
    if mode == "log":   # transforms/normalize.py:23-26
        return (np.log(np.float32(1000) * x + np.float32(1)) / np.log(np.float32(1000))).astype(np.float32)
    if mode == "asinh":  # transforms/normalize.py:4-11
        a = np.float32(0.02)
        return (np.arcsinh(x / a) / np.arcsinh(np.float32(1.0) / a)).astype(np.float32)
    raise ValueError(mode)


def _stretch_inv(x, mode):
    if mode == "linear":
        return x
    if mode == "sqrt":
        return np.square(x)
    if mode == "log":   # normalize.py:29-32
        return ((np.power(np.float32(1000), x) - np.float32(1)) / np.float32(1000)).astype(np.float32)
    if mode == "asinh":  # normalize.py:14-20
        a = np.float32(0.02)
        return (a * np.sinh(x * np.arcsinh(np.float32(1.0) / a))).astype(np.float32)
    raise ValueError(mode)


def normalize(img: np.ndarray, max_val: float, mode: str) -> np.ndarray:
    """Normalize.normalize_image (transforms/normalize.py:66-82)."""
    img = img.astype(np.float32)
    mv = np.float32(max_val)
    if mv > 0:
        img = np.clip(img, np.float32(0), mv) / mv
    else:
        img = img / img.max()
    return np.clip(_stretch(img, mode), np.float32(0), np.float32(1)).astype(np.float32)


def denormalize(img: np.ndarray, max_val, mode: str) -> np.ndarray:
    """Normalize.denormalize_image (normalize.py:84-92); max_val scalar or [B]."""
    mv = np.asarray(max_val, np.float32).reshape(-1, 1, 1, 1)
    out = mv * _stretch_inv(img.astype(np.float32), mode)
    return np.minimum(np.maximum(out, np.float32(0)), mv).astype(np.float32)


def image_upsample(x: np.ndarray, s: int) -> np.ndarray:
    """ImageUpsample (transforms/imageupsample.py:10-26): nearest x s then / s^2."""
    return (np.repeat(np.repeat(x, s, axis=-2), s, axis=-1) / np.float32(s ** 2)).astype(np.float32)


# ---------------------------------------------------------------------------------------------------
# torch.nn.functional restatement (CPU baseline only)
# ---------------------------------------------------------------------------------------------------
def torch_forward(kind, nf, blocks, state, x, num_upsample=1):
    """Same op graph as the reference modules, written with torch.nn.functional; state: name->torch tensor."""
    import torch
    import torch.nn.functional as F

    def conv(name, t):
        return F.conv2d(t, state[name + ".weight"], state[name + ".bias"], stride=1, padding=1)

    fea = conv("conv_first", x)
    cur = fea
    for i in range(blocks):
        rin = cur
        for r in (1, 2, 3):
            pre = f"rrdb.{i}.RDB{r}."
            xs = [cur]
            for c in (1, 2, 3, 4):
                xs.append(F.leaky_relu(conv(pre + f"conv{c}", torch.cat(xs, 1)), 0.2))
            cur = conv(pre + "conv5", torch.cat(xs, 1)) * 0.2 + cur
        cur = cur * 0.2 + rin
    fea = fea + conv("trunk_conv", cur)
    if kind == "sr":
        for u in range(num_upsample):
            fea = F.pixel_shuffle(F.leaky_relu(conv(f"upsampling.{3 * u}", fea), 0.01), 2)
        out = conv("conv_last", F.leaky_relu(conv("HRconv", fea), 0.2))
    else:
        out = conv("conv_last", fea) + x
    return torch.clamp(torch.clamp(out, 0.0, 1.0), 0.0, 1.0)
