"""GPU parity of the input-transform kernels against the golden vectors from the reference's transforms/ and
data/tools.py.  Bit-exact: detector mask * pad, ImageUpsample, linear stretch.  sqrt: <= 1 ulp (the reference's own
torch CPU sqrt is 1 ulp off IEEE on some inputs).  asinh/log: 2e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from util_hip import G

pytestmark = pytest.mark.gpu


def test_normalize_all_modes():
    from xmm_superres_denoise.transforms import Normalize
    z = np.load(os.path.join(G, "transforms.npz"))
    img = torch.from_numpy(z["norm_in"]).cuda()
    for mode, tol in (("linear", 0.0), ("sqrt", 1.2e-7), ("asinh", 2e-6), ("log", 2e-6)):
        nz = Normalize(lr_max=0.0022336, hr_max=0.0005584, stretch_mode=mode)
        assert np.abs(nz.normalize_lr_image(img.clone()).cpu().numpy() - z[f"norm_lr_{mode}"]).max() <= tol
        assert np.abs(nz.normalize_hr_image(img.clone()).cpu().numpy() - z[f"norm_hr_{mode}"]).max() <= tol
        u = torch.from_numpy(z[f"denorm_in_{mode}"]).cuda()
        assert np.abs(nz.denormalize_lr_image(u).cpu().numpy() - z[f"denorm_lr_{mode}"]).max() <= max(tol, 1e-7) * 0.0022336 * 4
        assert np.abs(nz.denormalize_hr_image(u).cpu().numpy() - z[f"denorm_hr_{mode}"]).max() <= max(tol, 1e-7) * 0.0005584 * 4
    nz = Normalize(lr_max=0.0, hr_max=0.0, stretch_mode="sqrt")
    out = nz.normalize_lr_image(torch.from_numpy(z["norm_auto_in"]).cuda()).cpu().numpy()
    assert np.abs(out - z["norm_auto_sqrt"]).max() <= 1.2e-7


def test_image_upsample_bit_exact():
    from xmm_superres_denoise.transforms import ImageUpsample
    z = np.load(os.path.join(G, "transforms.npz"))
    up = ImageUpsample(scale_factor=2)
    assert np.array_equal(up(torch.from_numpy(z["up_in3"]).cuda()).cpu().numpy(), z["up_out3"])
    assert np.array_equal(up(torch.from_numpy(z["up_in4"]).cuda()).cpu().numpy(), z["up_out4"])


def test_pad_crop_bit_exact():
    from xmm_superres_denoise.data.tools import reshape_img_to_res
    z = np.load(os.path.join(G, "transforms.npz"))
    for key in [k for k in z.files if k.startswith("pad_in_")]:
        res = int(key.split("_")[-1])
        a = torch.from_numpy(z[key].astype(np.float32)).cuda()
        r = reshape_img_to_res(res=res, img=a).cpu().numpy()
        assert np.array_equal(r, z[key.replace("pad_in_", "pad_out_")].astype(np.float32)), key


def test_detector_mask_pad_bit_exact_full_size():
    """The 'detector-mask index op': real 411x403 tile x real mask -> 416x416, bit-exact vs the oracle; int32 and
    float32 count inputs agree; batch of 2."""
    from xmm_superres_denoise.data.tools import load_and_prepare
    z = np.load(os.path.join(G, "example_data.npz"))
    m1 = np.unpackbits(z["mask1x_bits"])[: int(np.prod(z["mask1x_shape"]))].reshape(z["mask1x_shape"])
    counts = np.stack([z["dn_counts20_0"], z["dn_counts20_1"]])
    ref = np.stack([oracle.mask_pad(c, m1, 416) for c in counts])
    md = torch.from_numpy(m1).cuda()
    out_i = load_and_prepare(torch.from_numpy(counts).cuda(), md, 416).cpu().numpy()
    out_f = load_and_prepare(torch.from_numpy(counts.astype(np.float32)).cuda(), md, 416).cpu().numpy()
    assert np.array_equal(out_i, ref) and np.array_equal(out_f, ref)
    # property at full size: pixels outside the mask/pad region are exactly zero, inside equal the counts
    assert out_i[:, 0, :2].sum() == 0 and out_i[:, 0, -3:].sum() == 0
    fused = load_and_prepare(torch.from_numpy(counts).cuda(), md, 416, 0.0022336, "sqrt").cpu().numpy()
    assert np.abs(fused - np.stack([oracle.normalize(r, 0.0022336, "sqrt") for r in ref])).max() <= 1.2e-7


def test_compose_input_from_raw_fits_words_bit_exact():
    """XmmDataset sample composition in one kernel: (img + agn + bkg) * mask -> pad, from big-endian FITS words, and
    the real-data path img * mask -> nearest x2 / 4 -> pad 832 (data/dataset.py:24-49); bit-exact vs the numpy oracle."""
    from xmm_superres_denoise.engine import compose_input
    z = np.load(os.path.join(G, "example_data.npz"))
    m1 = np.unpackbits(z["mask1x_bits"])[: int(np.prod(z["mask1x_shape"]))].reshape(z["mask1x_shape"])
    a, b, c = z["sr_counts_lr_0"], z["sr_counts_lr_1"], z["dn_counts20_0"]
    ref = oracle.mask_pad((a.astype(np.float32) + b.astype(np.float32) + c.astype(np.float32)), m1, 416)
    md = torch.from_numpy(m1).cuda()
    be = [torch.from_numpy(v.astype(">i4").view(np.int32).copy())[None].cuda() for v in (a, b, c)]  # raw FITS byte order
    out = compose_input(be[0], be[1], be[2], md, 416, None, big_endian=True).cpu().numpy()
    assert np.array_equal(out[0], ref)
    ne = [torch.from_numpy(v.astype(np.int32))[None].cuda() for v in (a, b, c)]
    assert np.array_equal(compose_input(ne[0], ne[1], ne[2], md, 416, None).cpu().numpy()[0], ref)
    # real-data HR path: mask, nearest x2 with /4, pad to 832
    masked = c.astype(np.float32) * m1.astype(np.float32)
    ref_up = oracle.reshape_img_to_res(oracle.image_upsample(masked[None], 2), 832)
    got = compose_input(ne[2], None, None, md, 832, None, upsample=2).cpu().numpy()[0]
    assert np.array_equal(got, ref_up)
