"""Helpers shared by the GPU parity tests (not a test module)."""
import ctypes
import os

import numpy as np
import torch

import gen_common as gc

G = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name, kind):
    z = np.load(os.path.join(G, name + ".npz"))
    nf, blocks, nup, wseed, xseed, tseed = [int(v) for v in z["meta"][:6]]
    xshape = tuple(int(v) for v in z["meta"][6:])
    lb = float(z["last_bias"][0])
    in_ch, out_ch = (int(v) for v in z["chan"]) if "chan" in z.files else (1, 1)     # image channel counts (round-4 width goldens)
    assert in_ch == xshape[1]
    state = gc.make_state(kind, nf, blocks, wseed, num_upsample=nup, last_bias=None if np.isnan(lb) else lb, in_ch=in_ch, out_ch=out_ch)
    x = gc.make_input(xshape, xseed)
    s = 2 ** nup if kind == "sr" else 1
    t = gc.make_input((xshape[0], out_ch, xshape[2] * s, xshape[3] * s), tseed)
    return z, nf, blocks, nup, state, x, t


def build_module(kind, blocks, nup, state, device="cuda", nf=32, in_ch=1, out_ch=1):
    from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
    m = GeneratorRRDB_DN(in_ch, out_ch, nf, blocks) if kind == "dn" else GeneratorRRDB_SR(in_ch, out_ch, nf, blocks, num_upsample=nup)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    return m.to(device)


def nchw_to_planes(a: np.ndarray):
    """[B,C,H,W] -> list of C/32 contiguous NHWC planes [B,H,W,32] (torch cuda)"""
    B, C, H, W = a.shape
    out = []
    for k in range(C // 32):
        out.append(torch.from_numpy(np.ascontiguousarray(a[:, 32 * k:32 * k + 32].transpose(0, 2, 3, 1))).cuda())
    return out


def planes_to_nchw(planes):
    return np.concatenate([p.cpu().numpy().transpose(0, 3, 1, 2) for p in planes], axis=1)


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def flip_candidates(kind, blocks, state, x, t, nup=1, eps=4e-6, return_clamp_mask=False, device="cpu"):
    """Where can two fp32-accurate implementations legitimately take different branches of the network's step functions?
    A float64 evaluation of the reference graph (rrdb_blocks.py:37-54, generator_rrdb.py:66-137) records, for every conv
    that feeds a LeakyReLU, the output channels that hold a pre-activation within eps * rms(plane) of zero, plus whether any
    output pixel sits within eps of a clamp bound or of its target (clamp mask, sign(y - t) of the L1 loss).
    Returns ({param prefix: set(channels)}, n_output_candidates), or with return_clamp_mask the boolean mask of the output pixels
    whose pre-clamp value sits within 8 eps of a clamp bound instead of the count.
    The dict also names the rows of dL/dx a candidate can reach, under the key "__dx_rows__" (row index of dx reshaped to
    [-1, W]): a decision that falls the other way at pixel (y, x) of a conv `depth` 3 x 3 convs behind the input changes dL/dx
    inside that pixel's receptive field, the image rows y - depth .. y + depth (low-resolution coordinates)."""
    import torch.nn.functional as F
    st = {k: torch.from_numpy(v).double().to(device) for k, v in state.items()}     # device="cuda": the float64 pass of a full-size batch takes seconds instead of minutes
    cand = {}
    B, C_in, H_in = x.shape[0], x.shape[1], x.shape[2]
    dx_rows = set()
    depth = [0]     # 3 x 3 convs between the network input and the conv whose output is being examined (inclusive)

    def conv(name, inp):
        depth[0] += 1
        return F.conv2d(inp, st[name + ".weight"], st[name + ".bias"], padding=1)

    def act(name, pre, slope):
        rms = pre.pow(2).mean().sqrt()
        close = pre.abs() < eps * rms
        near = close.any(dim=0).any(dim=-1).any(dim=-1)      # per output channel
        ch = set(int(c) for c in torch.nonzero(near).flatten())
        if ch:
            cand[name] = ch
            sc = pre.shape[2] // H_in                         # 1 on the trunk, 2 / 4 behind the pixel shuffles
            for b in range(B):
                ys = torch.nonzero(close[b].any(dim=0).any(dim=-1)).flatten()
                for yy in ys.tolist():
                    lo, hi = max(0, yy // sc - depth[0]), min(H_in - 1, yy // sc + depth[0])
                    for c in range(C_in):
                        dx_rows.update(range((b * C_in + c) * H_in + lo, (b * C_in + c) * H_in + hi + 1))
        return F.leaky_relu(pre, slope)

    xt = torch.from_numpy(x).double().to(device)
    fea = conv("conv_first", xt)
    cur = fea
    for i in range(blocks):
        rin = cur
        for r in (1, 2, 3):
            pre = f"rrdb.{i}.RDB{r}."
            xs = [cur]
            for c in (1, 2, 3, 4):
                xs.append(act(pre + f"conv{c}", conv(pre + f"conv{c}", torch.cat(xs, 1)), 0.2))
            cur = conv(pre + "conv5", torch.cat(xs, 1)) * 0.2 + cur
        cur = cur * 0.2 + rin
    fea = fea + conv("trunk_conv", cur)
    if kind == "sr":
        for u in range(nup):
            fea = F.pixel_shuffle(act(f"upsampling.{3 * u}", conv(f"upsampling.{3 * u}", fea), 0.01), 2)
        out = conv("conv_last", act("HRconv", conv("HRconv", fea), 0.2))
    else:
        out = conv("conv_last", fea) + xt
    if dx_rows:
        cand["__dx_rows__"] = dx_rows
    tt = torch.from_numpy(t).double().to(device)
    y = out.clamp(0, 1)
    n_out = int(((out.abs() < eps) | ((out - 1).abs() < eps) | ((y - tt).abs() < eps)).sum())
    if return_clamp_mask:      # output pixels whose PRE-clamp value is within 8 eps of a clamp bound (either side of it)
        return cand, ((out.abs() < 8 * eps) | ((out - 1).abs() < 8 * eps)).cpu().numpy()
    return cand, n_out


FLIP_LOG = []     # one record per (case, tensor) that needed the flip allowance: printed by the tests, asserted to stay explainable


def layer_order(prefix):
    """position of a conv in the FORWARD pass (the state-dict order lists conv_last before the SR head)"""
    if prefix == "conv_first":
        return (0,)
    if prefix.startswith("rrdb."):
        _, i, r, c = prefix.split(".")
        return (1, int(i), int(r[3:]), int(c[4:]))
    if prefix == "trunk_conv":
        return (2,)
    if prefix.startswith("upsampling."):
        return (3, int(prefix.split(".")[1]))
    return (4,) if prefix == "HRconv" else (5,)


def assert_grad_close(g, ref, name, tight=2e-4, loose=2e-2, max_flip_frac=0.15, candidates=None, n_out_candidates=0,
                      strict=1e-3, l2_bar=None):
    """Gradient comparison, relative to the tensor's largest entry.

    LeakyReLU', the clamp mask and sign(y - t) are step functions: an activation within ~1e-6 of zero can take a different
    branch in two fp32-accurate implementations.  That changes ONE (pixel, channel) term, i.e. the weight / bias gradient
    ROW of that output channel of that conv by O(1/sqrt(N)), and sends a ripple well below 1e-3 upstream.
      * `candidates` given (flip-aware mode, used for the fp32-class math modes): a row over `tight` must be EXPLAINED -- either
        it is named by flip_candidates() for this tensor (that conv's output channel holds a pre-activation within rounding
        of zero: up to `loose`), or a candidate sits DOWNSTREAM of this tensor in the forward pass (or at an output pixel):
        the decision that falls the other way there changes one term of the backward pass and ripples into every gradient
        upstream of it, far below north_star's 1e-3 (`strict`; the tests scale it for images of under 1000 pixels).  A
        candidate upstream of the tensor explains nothing (the backward pass reaches this tensor first).  Rows that are
        neither are counted as `unexplained` in FLIP_LOG and fail.
      * `candidates` None (16-bit tolerance-only modes): at most `max_flip_frac` of the rows may exceed `tight`, none
        `loose`, relative L2 error below `loose`/2."""
    g = np.asarray(g, np.float64).reshape(ref.shape)
    r = np.asarray(ref, np.float64)
    scale = np.abs(r).max() + 1e-30
    err = np.abs(g - r) / scale
    if err.max() <= tight:
        return
    rows = err.reshape(err.shape[0], -1).max(axis=1) if err.ndim > 1 else err
    bad = np.nonzero(rows > tight)[0]
    if candidates is not None:
        prefix = name.rsplit(".", 1)[0]
        allowed = candidates.get(prefix, set()) if err.ndim >= 1 and name.rsplit(".", 1)[-1] in ("weight", "bias") else set()
        if name == "dx":       # rows of dL/dx inside the receptive field of a candidate pixel (flip_candidates)
            allowed = candidates.get("__dx_rows__", set())
        not_named = [int(b) for b in bad if int(b) not in allowed]
        # a candidate DOWNSTREAM of this tensor (dL/dx: every layer is) or at an output pixel (clamp bound / L1 sign) ripples into it
        here = (-1,) if name == "dx" else layer_order(prefix)
        downstream = n_out_candidates > 0 or any(k != "__dx_rows__" and layer_order(k) > here for k in candidates)
        ripple = [b for b in not_named if downstream and rows[b] <= strict]
        unexplained = [b for b in not_named if b not in set(ripple)]
        FLIP_LOG.append({"tensor": name, "rows_over_tight": len(bad), "candidate_rows": len(allowed), "ripple_rows": len(ripple),
                         "unexplained": len(unexplained), "max_rel_err": float(err.max()), "output_candidates": n_out_candidates})
        assert not unexplained, \
            f"{name}: rows {unexplained[:8]} exceed {tight} (max {max(rows[b] for b in unexplained):.3e}) with no flip candidate on them or downstream of them"
        assert err.max() <= loose, f"{name}: max rel err {err.max():.3e} > loose {loose}"
        return
    nbad = len(bad)
    assert err.max() <= loose, f"{name}: max rel err {err.max():.3e} > loose {loose}"
    assert nbad <= max(1, int(max_flip_frac * rows.size)), f"{name}: {nbad}/{rows.size} rows exceed {tight}"
    l2 = np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-30)
    assert l2 <= (loose / 2 if l2_bar is None else l2_bar), f"{name}: rel L2 {l2:.3e}"
