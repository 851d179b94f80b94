"""GPU parity of the individual HIP kernels against the CPU oracle, through the C ABI (single-layer test hooks).
Tolerance: |hip - oracle| <= 2e-5 * max|oracle| (fp32 MFMA is a k-ordered fmaf chain; the oracle sums in a
different order).  Shapes are ragged against the 8x32 tile on purpose."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle
from util_hip import nchw_to_planes, planes_to_nchw, ptr_array

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp32", "bf16x6", "f16x3"])
def eng(request):
    from xmm_superres_denoise.engine import Engine
    math = request.param
    e = Engine("dn", 1, 1, 32, 1)
    e.set_math(math)
    e.tol = 2e-5
    return e


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("n_in,n_out,shape", [
    (1, 1, (1, 8, 32)), (1, 1, (2, 19, 45)), (2, 1, (1, 16, 64)), (3, 1, (2, 9, 33)), (4, 1, (1, 24, 40)),
    (5, 1, (2, 17, 70)), (1, 2, (1, 11, 37)), (1, 4, (2, 8, 32)), (1, 5, (1, 25, 31)),
])
def test_conv3x3_forward(eng, n_in, n_out, shape):
    from xmm_superres_denoise.engine._lib import check
    B, H, W = shape
    rng = np.random.default_rng(100 * n_in + n_out)
    x = rng.normal(size=(B, 32 * n_in, H, W)).astype(np.float32)
    w = (rng.normal(size=(32 * n_out, 32 * n_in, 3, 3)) / np.sqrt(288 * n_in)).astype(np.float32)
    b = rng.normal(size=(32 * n_out,)).astype(np.float32)
    ref = oracle.conv3x3(x, w, b)
    ref = np.where(ref > 0, ref, 0.2 * ref)
    xin = nchw_to_planes(x)
    outs = [torch.full((B, H, W, 32), float("nan"), device="cuda") for _ in range(n_out)]
    wd, bd = torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()
    check(eng.L.xsd_test_conv3x3(eng.h, ptr_array(xin), n_in, wd.data_ptr(), bd.data_ptr(), ptr_array(outs), n_out,
                                 0.2, B, H, W, None))
    got = planes_to_nchw(outs)
    assert np.isfinite(got).all()
    assert _rel(got, ref) < eng.tol


@pytest.mark.parametrize("n_in,shape", [(1, (1, 8, 32)), (1, (3, 5, 7)), (4, (1, 9, 31)), (2, (2, 19, 45)), (3, (1, 40, 33)), (5, (2, 17, 70))])
def test_conv3x3_backward(eng, n_in, shape):
    from xmm_superres_denoise.engine._lib import check
    B, H, W = shape
    rng = np.random.default_rng(7 * n_in)
    x = rng.normal(size=(B, 32 * n_in, H, W)).astype(np.float32)
    w = (rng.normal(size=(32, 32 * n_in, 3, 3)) / np.sqrt(288 * n_in)).astype(np.float32)
    g = rng.normal(size=(B, 32, H, W)).astype(np.float32)
    dx_ref, dw_ref, db_ref = oracle.conv3x3_bwd(x, w, g)
    xin = nchw_to_planes(x)
    gp = nchw_to_planes(g)[0]
    dxs = [torch.full((B, H, W, 32), float("nan"), device="cuda") for _ in range(n_in)]
    wd = torch.from_numpy(w).cuda()
    dw = torch.full_like(wd, float("nan"))
    db = torch.full((32,), float("nan"), device="cuda")
    check(eng.L.xsd_test_conv3x3_bwd(eng.h, ptr_array(xin), n_in, wd.data_ptr(), gp.data_ptr(), ptr_array(dxs),
                                     dw.data_ptr(), db.data_ptr(), B, H, W, None))
    assert _rel(planes_to_nchw(dxs), dx_ref) < eng.tol
    assert _rel(dw.cpu().numpy(), dw_ref) < max(5e-5, eng.tol)
    assert _rel(db.cpu().numpy(), db_ref) < max(5e-5, eng.tol)


def test_wgrad_is_bitwise_reproducible(eng):
    from xmm_superres_denoise.engine._lib import check
    B, H, W, n_in = 2, 33, 65, 3
    rng = np.random.default_rng(3)
    xin = nchw_to_planes(rng.normal(size=(B, 96, H, W)).astype(np.float32))
    gp = nchw_to_planes(rng.normal(size=(B, 32, H, W)).astype(np.float32))[0]
    wd = torch.zeros((32, 96, 3, 3), device="cuda")
    res = []
    for _ in range(2):
        dxs = [torch.empty((B, H, W, 32), device="cuda") for _ in range(n_in)]
        dw, db = torch.empty_like(wd), torch.empty((32,), device="cuda")
        check(eng.L.xsd_test_conv3x3_bwd(eng.h, ptr_array(xin), n_in, wd.data_ptr(), gp.data_ptr(), ptr_array(dxs),
                                         dw.data_ptr(), db.data_ptr(), B, H, W, None))
        res.append((dw.cpu().numpy().copy(), db.cpu().numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_adam_kernel_matches_oracle(eng):
    rng = np.random.default_rng(9)
    n = 100003
    p = rng.normal(size=n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    pd, md, vd = torch.from_numpy(p.copy()).cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = (rng.normal(size=n) * 0.01).astype(np.float32)
        oracle.adam(p, g, m, v, step)
        eng.adam_step(pd, torch.from_numpy(g).cuda(), md, vd, step)
        assert np.abs(pd.cpu().numpy() - p).max() < 3e-7


def test_l1_loss_kernel(eng):
    rng = np.random.default_rng(4)
    y = rng.uniform(size=(2, 1, 37, 53)).astype(np.float32); t = rng.uniform(size=y.shape).astype(np.float32)
    y[0, 0, 0, :5] = t[0, 0, 0, :5]
    loss, dy = eng.l1_loss(torch.from_numpy(y).cuda(), torch.from_numpy(t).cuda())
    assert abs(loss.item() - np.abs(y.astype(np.float64) - t).mean()) < 1e-6
    ref = np.sign(y - t) / y.size
    assert np.allclose(dy.cpu().numpy(), ref, rtol=1e-6, atol=0)
