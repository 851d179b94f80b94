#!/usr/bin/env python3
"""Generate the golden fixtures from the REFERENCE's own PyTorch modules.

Runs ONLY in the build container (needs /root/reference).  The reference is imported by path
(SURVEY.md section 8c recipe): `rrdb_blocks.py` needs only torch; `generator_rrdb.py` does
`from models.modules import RRDB, make_layer`, so stub packages `models` / `models.modules` are
registered first (a normal `import models` would pull in lightning, which is not installed).
`transforms/` imports cleanly; `data/tools.py::reshape_img_to_res` loads once astropy/loguru are stubbed.

Outputs (small .npz files, committed):  tests/golden/*.npz.  No reference source is copied.
Weights/inputs are NOT stored: they are regenerated from gen_common.make_state / make_input.
"""
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_common as gc  # noqa: E402

REF = "/root/reference"
REFPKG = os.path.join(REF, "xmm_superres_denoise")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    rb = _load("_ref_rrdb_blocks", os.path.join(REFPKG, "models/modules/rrdb_blocks.py"))
    pkg = types.ModuleType("models")
    pkg.__path__ = []
    sub = types.ModuleType("models.modules")
    sub.RRDB = rb.RRDB
    sub.make_layer = rb.make_layer
    pkg.modules = sub
    sys.modules["models"] = pkg
    sys.modules["models.modules"] = sub
    gen = _load("_ref_generator_rrdb", os.path.join(REFPKG, "models/modules/generator_rrdb.py"))
    # transforms: torch/numpy only
    sys.path.insert(0, REFPKG)
    import transforms as ref_transforms  # noqa
    # data/tools.py needs astropy.io.fits + loguru at import time only
    for name in ("astropy", "astropy.io", "astropy.io.fits", "loguru", "pandas_stub"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["astropy.io"].fits = sys.modules["astropy.io.fits"]
    sys.modules["astropy"].io = sys.modules["astropy.io"]
    sys.modules["loguru"].logger = types.SimpleNamespace(
        info=print, warning=print, error=print, debug=print, success=print)
    tools = _load("_ref_tools", os.path.join(REFPKG, "data/tools.py"))
    return rb, gen, ref_transforms, tools


def build_ref(gen, kind, nf, blocks, num_upsample=1, in_ch=1, out_ch=1):
    """the reference's own constructors (generator_rrdb.py:73-81,114-121) at any widths / image channel counts"""
    if kind == "dn":
        m = gen.GeneratorRRDB_DN(in_ch, out_ch, nf, blocks)
    else:
        m = gen.GeneratorRRDB_SR(in_ch, out_ch, nf, blocks, num_upsample=num_upsample)
    return m


def load_np_state(m, state):
    sd = m.state_dict()
    assert list(sd.keys()) == list(state.keys()), (list(sd.keys())[:5], list(state.keys())[:5])
    for k, v in state.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})


def run_case(gen, name, kind, nf, blocks, xshape, wseed, xseed, tseed, num_upsample=1,
             full_grads=False, last_bias=None, selected=(), out_ch=1):
    """xshape = (B, in_ch, H, W); out_ch output channels (the target's).  Cases with image channel counts other than 1 / 1
    store them under "chan"."""
    torch.manual_seed(0)
    in_ch = xshape[1]
    m = build_ref(gen, kind, nf, blocks, num_upsample, in_ch, out_ch)
    state = gc.make_state(kind, nf, blocks, wseed, num_upsample=num_upsample, last_bias=last_bias, in_ch=in_ch, out_ch=out_ch)
    load_np_state(m, state)
    x = torch.from_numpy(gc.make_input(xshape, xseed)).requires_grad_(True)
    scale = 1 if kind == "dn" else 2 ** num_upsample
    tshape = (xshape[0], out_ch, xshape[2] * scale, xshape[3] * scale)
    t = torch.from_numpy(gc.make_input(tshape, tseed))
    # Model.forward clamps a second time (models/model.py:48-49)
    y = torch.clamp(m(x), min=0.0, max=1.0)
    loss = torch.nn.functional.l1_loss(y, t)
    loss.backward()
    out = OrderedDict()
    out["meta"] = np.array([nf, blocks, num_upsample, wseed, xseed, tseed] + list(xshape), dtype=np.int64)
    out["last_bias"] = np.array([np.nan if last_bias is None else last_bias], dtype=np.float64)
    if (in_ch, out_ch) != (1, 1):
        out["chan"] = np.array([in_ch, out_ch], dtype=np.int64)
    out["y"] = y.detach().numpy()
    out["loss"] = np.array([loss.item()], dtype=np.float64)
    out["dx"] = x.grad.numpy()
    names = [n for n, _ in m.named_parameters()]
    gs = np.zeros((len(names), 2), dtype=np.float64)
    for i, (n, p) in enumerate(m.named_parameters()):
        g = p.grad.detach().numpy().astype(np.float64)
        gs[i, 0] = g.sum()
        gs[i, 1] = np.abs(g).sum()
        if full_grads or n.rsplit(".", 1)[0] in selected:
            out["grad." + n] = p.grad.detach().numpy()
    out["grad_sums"] = gs
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    frac_clamped = float(((y <= 0) | (y >= 1)).float().mean())
    print(f"{name}: loss={loss.item():.6f} clamped={frac_clamped:.3f} y[{y.min().item():.3f},{y.max().item():.3f}]")


def init_parity(gen):
    """Reference constructors under torch.manual_seed(0): per-tensor sums + first values."""
    out = {}
    for kind in ("dn", "sr"):
        torch.manual_seed(0)
        m = build_ref(gen, kind, 32, 4, 1)
        sd = m.state_dict()
        names = list(sd.keys())
        stats = np.zeros((len(names), 6), dtype=np.float64)
        for i, k in enumerate(names):
            v = sd[k].double().flatten()
            stats[i, 0] = v.sum()
            stats[i, 1] = v.abs().sum()
            stats[i, 2:2 + min(4, v.numel())] = v[:4].numpy()
        out[kind + "_names"] = np.array(names)
        out[kind + "_stats"] = stats
        out[kind + "_nparams"] = np.array([sum(p.numel() for p in m.parameters())])
    np.savez_compressed(os.path.join(HERE, "init_parity.npz"), **out)
    print("init parity:", out["dn_nparams"], out["sr_nparams"])


def transforms_golden(ref_transforms, tools):
    rng = np.random.default_rng(77)
    out = {}
    # Normalize: 4 stretch modes on an input with negatives, zeros, > max values
    img = (rng.uniform(-0.1, 1.4, size=(2, 1, 9, 11)) * 0.0022336).astype(np.float32)
    img[0, 0, 0, :4] = [0.0, 0.0022336, 1e-9, 5.0]
    out["norm_in"] = img
    for mode in ("linear", "sqrt", "asinh", "log"):
        nz = ref_transforms.Normalize(lr_max=0.0022336, hr_max=0.0005584, stretch_mode=mode)
        out[f"norm_lr_{mode}"] = nz.normalize_lr_image(torch.from_numpy(img.copy())).numpy()
        out[f"norm_hr_{mode}"] = nz.normalize_hr_image(torch.from_numpy(img.copy())).numpy()
        u = rng.uniform(0, 1, size=(2, 1, 9, 11)).astype(np.float32)
        u[0, 0, 0, :3] = [0.0, 1.0, 0.5]
        out[f"denorm_in_{mode}"] = u
        # NOTE: denormalize_lr_image/_hr_image raise IndexError in the reference as written (they pass the
        # 0-dim self.lr_max into max_val[:, None, None, None], transforms/normalize.py:88,103-107);
        # the working contract is denormalize_image(image[B,C,H,W], max_val[B]).
        mv_lr = torch.full((u.shape[0],), 0.0022336)
        mv_hr = torch.full((u.shape[0],), 0.0005584)
        out[f"denorm_lr_{mode}"] = nz.denormalize_image(torch.from_numpy(u.copy()), mv_lr).numpy()
        out[f"denorm_hr_{mode}"] = nz.denormalize_image(torch.from_numpy(u.copy()), mv_hr).numpy()
    # max_val <= 0 branch: divide by image max (transforms/normalize.py:72-74)
    nz = ref_transforms.Normalize(lr_max=0.0, hr_max=0.0, stretch_mode="sqrt")
    pos = rng.uniform(0, 3, size=(1, 5, 7)).astype(np.float32)
    out["norm_auto_in"] = pos
    out["norm_auto_sqrt"] = nz.normalize_lr_image(torch.from_numpy(pos.copy())).numpy()
    # ImageUpsample (transforms/imageupsample.py:10-26), 3-D and 4-D inputs
    up = ref_transforms.ImageUpsample(scale_factor=2)
    a3 = rng.uniform(0, 5, size=(1, 5, 7)).astype(np.float32)
    a4 = rng.uniform(0, 5, size=(2, 1, 4, 6)).astype(np.float32)
    out["up_in3"], out["up_out3"] = a3, up(torch.from_numpy(a3)).numpy()
    out["up_in4"], out["up_out4"] = a4, up(torch.from_numpy(a4)).numpy()
    # reshape_img_to_res (data/tools.py:103-126): pad 411x403 -> 416, 822x806 -> 832, and a crop case
    for (h, w, res) in ((411, 403, 416), (822, 806, 832), (20, 13, 16), (9, 9, 12)):
        a = rng.integers(0, 50, size=(1, h, w)).astype(np.float32)
        r = tools.reshape_img_to_res(res=res, img=torch.from_numpy(a)).numpy()
        out[f"pad_in_{h}x{w}_{res}"] = a.astype(np.int16)
        out[f"pad_out_{h}x{w}_{res}"] = r.astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "transforms.npz"), **out)
    print("transforms golden written")


def example_data_golden(gen, ref_transforms, tools):
    """Config 1: real 20ks tile -> mask -> pad 416 -> sqrt-normalize -> DN forward (full model).
    Also sim 20ks 1x img -> SR forward -> PSNR vs normalized 100ks 2x target (data/dataset.py:24-49,258-270).
    Inputs are stored as int32 counts (data); masks stored bit-packed."""
    import glob
    out = {}
    m1, _ = gc.read_fits_primary(os.path.join(REF, "res/detector_mask/pn_mask_500_2000_detxy_1x.ds"))
    m2, _ = gc.read_fits_primary(os.path.join(REF, "res/detector_mask/pn_mask_500_2000_detxy_2x.ds"))
    out["mask1x_bits"] = np.packbits(m1.astype(np.uint8))
    out["mask2x_bits"] = np.packbits(m2.astype(np.uint8))
    out["mask1x_shape"] = np.array(m1.shape)
    out["mask2x_shape"] = np.array(m2.shape)
    nz = ref_transforms.Normalize(lr_max=0.0022336, hr_max=0.0022336, stretch_mode="sqrt")
    nz_sr = ref_transforms.Normalize(lr_max=0.0022336, hr_max=0.0005584, stretch_mode="sqrt")

    def prep(counts, mask, res, normfn):
        img = torch.from_numpy(counts.astype(np.float32)).unsqueeze(0)  # load_fits
        img *= torch.from_numpy(mask.astype(np.float32)).unsqueeze(0)   # dataset.py:41-42
        img = tools.reshape_img_to_res(res=res, img=img)                # dataset.py:47
        return normfn(img)                                               # dataset.py:267-268

    # --- DN on two real tiles
    real = sorted(glob.glob(os.path.join(REF, "data/example_data/real/20ks/*.fits*")))[:2]
    real50 = sorted(glob.glob(os.path.join(REF, "data/example_data/real/50ks/*.fits*")))[:2]
    torch.manual_seed(0)
    dn = build_ref(gen, "dn", 32, 4)
    load_np_state(dn, gc.make_state("dn", 32, 4, 1234))
    for i, (p20, p50) in enumerate(zip(real, real50)):
        c20, _ = gc.read_fits_primary(p20)
        c50, _ = gc.read_fits_primary(p50)
        x = prep(c20, m1, 416, nz.normalize_lr_image)[None]
        t = prep(c50, m1, 416, nz.normalize_hr_image)[None]
        with torch.no_grad():
            y = torch.clamp(dn(x), 0.0, 1.0)
        out[f"dn_counts20_{i}"] = c20.astype(np.int32)
        out[f"dn_counts50_{i}"] = c50.astype(np.int32)
        out[f"dn_x_sum_{i}"] = np.array([x.double().sum().item()])
        yn = y.numpy()[0, 0]
        out[f"dn_y_sub_{i}"] = yn[::5, ::5].copy()
        out[f"dn_y_stats_{i}"] = np.array([yn.astype(np.float64).sum(), (yn.astype(np.float64) ** 2).sum()])
        out[f"dn_psnr_{i}"] = np.array([gc.psnr(yn, t.numpy()[0, 0])])
        print(f"example DN {i}: psnr={out[f'dn_psnr_{i}'][0]:.4f} dB  y.sum={yn.sum():.3f}")
    # --- SR on two sim tiles (img only; no agn/background so the pairing is deterministic)
    lr_files = sorted(glob.glob(os.path.join(REF, "data/example_data/sim/20ks/img/1x/*.fits*")))
    hr_files = sorted(glob.glob(os.path.join(REF, "data/example_data/sim/100ks/img/2x/*.fits*")))
    torch.manual_seed(0)
    sr = build_ref(gen, "sr", 32, 4, 1)
    load_np_state(sr, gc.make_state("sr", 32, 4, 4321, last_bias=0.05))
    n = 0
    for p in lr_files:
        key = os.path.basename(p).split("_mult_")[0]
        match = [h for h in hr_files if os.path.basename(h).split("_mult_")[0] == key]
        if not match:
            continue
        c_lr, _ = gc.read_fits_primary(p)
        c_hr, _ = gc.read_fits_primary(match[0])
        x = prep(c_lr, m1, 416, nz_sr.normalize_lr_image)[None]
        t = prep(c_hr, m2, 832, nz_sr.normalize_hr_image)[None]
        with torch.no_grad():
            y = torch.clamp(sr(x), 0.0, 1.0)
        yn = y.numpy()[0, 0]
        out[f"sr_counts_lr_{n}"] = c_lr.astype(np.int32)
        out[f"sr_counts_hr_{n}"] = c_hr.astype(np.int32)
        out[f"sr_y_sub_{n}"] = yn[::9, ::9].copy()
        out[f"sr_y_stats_{n}"] = np.array([yn.astype(np.float64).sum(), (yn.astype(np.float64) ** 2).sum()])
        out[f"sr_psnr_{n}"] = np.array([gc.psnr(yn, t.numpy()[0, 0])])
        print(f"example SR {n}: psnr={out[f'sr_psnr_{n}'][0]:.4f} dB")
        n += 1
        if n == 2:
            break
    np.savez_compressed(os.path.join(HERE, "example_data.npz"), **out)


def width_cases(gen):
    """Round 4: the reference at OTHER widths and image channel counts than the shipped 32 / 1 / 1 (its constructors take any,
    generator_rrdb.py:10-54; the dense block's own default is nf = 64, rrdb_blocks.py:23), so that the engine's multi-plane,
    zero-padded and per-image-channel paths are pinned to the reference itself rather than to a restatement.  Small images;
    the wide nets keep full tensors of a selection of layers plus the per-tensor sums of every gradient."""
    sel = ("conv_first", "conv_last", "trunk_conv", "rrdb.0.RDB1.conv1", "rrdb.0.RDB2.conv3", "rrdb.0.RDB3.conv5", "upsampling.3", "HRconv")
    run_case(gen, "dn_nf64_b1", "dn", 64, 1, (1, 1, 20, 24), 101, 102, 103, selected=sel)                        # two planes per tensor
    run_case(gen, "sr_nf32_c3x2_b1", "sr", 32, 1, (2, 3, 18, 35), 111, 112, 113, full_grads=True, last_bias=0.4, out_ch=2)   # RGB in, 2 out
    run_case(gen, "dn_nf16_c1x3_b1", "dn", 16, 1, (2, 1, 16, 16), 121, 122, 123, full_grads=True, out_ch=3)      # broadcast skip (:134), zero-padded width
    run_case(gen, "sr_nf64_b1_up2", "sr", 64, 1, (1, 1, 9, 11), 131, 132, 133, num_upsample=2, selected=sel, last_bias=0.4)
    run_case(gen, "dn_nf48_b1", "dn", 48, 1, (1, 1, 21, 45), 141, 142, 143, selected=sel)                        # zero-padded to 64: two planes, one half empty


def main():
    torch.set_num_threads(8)
    rb, gen, ref_transforms, tools = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "widths":      # only the round-4 additions (the older fixtures stay byte-identical)
        return width_cases(gen)
    width_cases(gen)
    init_parity(gen)
    transforms_golden(ref_transforms, tools)
    sel = ("conv_first", "conv_last", "trunk_conv", "rrdb.0.RDB1.conv1", "rrdb.0.RDB1.conv5",
           "rrdb.1.RDB2.conv3", "rrdb.3.RDB3.conv5", "rrdb.3.RDB3.conv4", "upsampling.0", "HRconv")
    # reduced models (oracle generality; full tensors)
    run_case(gen, "dn_nf8_b1", "dn", 8, 1, (2, 1, 12, 20), 11, 12, 13, full_grads=True)
    run_case(gen, "sr_nf8_b1", "sr", 8, 1, (2, 1, 12, 20), 21, 22, 23, full_grads=True, last_bias=0.4)
    run_case(gen, "sr_nf8_b1_up2", "sr", 8, 1, (1, 1, 6, 10), 31, 32, 33, num_upsample=2, full_grads=True,
             last_bias=0.4)
    # full architecture (nf=32, 4 RRDB), small images incl. ragged sizes vs the 8x32 HIP tile
    run_case(gen, "dn_nf32_b4_32x32", "dn", 32, 4, (1, 1, 32, 32), 41, 42, 43, selected=sel)
    run_case(gen, "dn_nf32_b4_24x40", "dn", 32, 4, (2, 1, 24, 40), 51, 52, 53, selected=sel)
    run_case(gen, "sr_nf32_b4_24x40", "sr", 32, 4, (2, 1, 24, 40), 61, 62, 63, selected=sel, last_bias=0.4)
    run_case(gen, "sr_nf32_b4_17x45", "sr", 32, 4, (1, 1, 17, 45), 71, 72, 73, selected=sel, last_bias=0.4)
    run_case(gen, "dn_nf32_b1_64x64", "dn", 32, 1, (1, 1, 64, 64), 81, 82, 83, full_grads=True)
    example_data_golden(gen, ref_transforms, tools)


if __name__ == "__main__":
    main()
