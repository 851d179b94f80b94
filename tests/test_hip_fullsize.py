"""BASELINE.json's single-GPU workloads at their full sizes, under assertions (the oracle is far too slow here, so these
are size-independent properties):
  configs[1]  SR 2x forward, batch 16, 1x512x512 -> 1x1024x1024
  configs[2]  DN train step, batch 32 (L1 + Adam) -- also with the reference's shipped loss (0.5 PSNR + 0.5 MS-SSIM)
  configs[3]/[4] per-GPU share: SR train step, and a DN train step fed by the on-GPU input pipeline
              (int32 counts -> detector mask -> pad -> sqrt-normalize) as `bench.py --input-pipeline` does.
Properties: bitwise determinism, batch independence (a tile's output does not depend on its neighbours), agreement of the
fp32-class math modes, outputs inside [0,1], finite and reproducible gradients, the workspace fits, the update moves."""
import os

import numpy as np
import pytest
import torch

import gen_common as gc
from util_hip import G, build_module

pytestmark = pytest.mark.gpu

HEADLINE = "f16x3"


def _tiles(shape, seed):
    return torch.from_numpy(gc.make_input(shape, seed)).cuda()


def test_sr_forward_batch16_full_size():
    state = gc.make_state("sr", 32, 4, 3101, last_bias=0.05)
    x = _tiles((16, 1, 512, 512), 3102)
    m = build_module("sr", 4, 1, state).set_math(HEADLINE)
    with torch.no_grad():
        y1 = m(x)
        y2 = m(x)
        y_one = m(x[5:6].contiguous())
    assert y1.shape == (16, 1, 1024, 1024)
    assert torch.equal(y1, y2)                                  # determinism
    assert torch.equal(y1[5:6], y_one)                          # batch independence
    assert float(y1.min()) >= 0.0 and float(y1.max()) <= 1.0 and torch.isfinite(y1).all()
    assert float(y1.std()) > 1e-3                               # not a constant image
    m32 = build_module("sr", 4, 1, state).set_math("fp32")
    with torch.no_grad():
        y32 = m32(x[:2].contiguous())
    assert float((y1[:2] - y32).abs().max()) < 5e-6             # the two fp32-class modes agree to a few ulp of 1.0


def _train_step_properties(kind, batch, loss=None, steps=2):
    from xmm_superres_denoise.parallel import DataParallelTrainer
    s = 2 if kind == "sr" else 1
    state = gc.make_state(kind, 32, 4, 3201, last_bias=0.05 if kind == "sr" else None)
    x = _tiles((batch, 1, 512, 512), 3202)
    t = _tiles((batch, 1, 512 * s, 512 * s), 3203)
    runs = []
    for _ in range(2):      # the same two steps twice from the same start: bitwise reproducible
        m = build_module(kind, 4, 1, state).set_math(HEADLINE)
        tr = DataParallelTrainer(m, lr=1e-4, loss=loss)
        losses = [float(tr.train_step(x, t)) for _ in range(steps)]
        runs.append((losses, tr.grads.clone(), tr.flat.clone()))
        del tr, m
        torch.cuda.empty_cache()
    (l1, g1, p1), (l2, g2, p2) = runs
    assert l1 == l2 and torch.equal(g1, g2) and torch.equal(p1, p2)
    assert all(np.isfinite(l1)) and torch.isfinite(g1).all() and torch.isfinite(p1).all()
    start = np.concatenate([v.ravel() for v in state.values()])
    moved = np.abs(p1.cpu().numpy() - start)
    assert moved.max() > 5e-5 and moved.max() < 2.5e-4 * steps      # Adam: |update| <= lr per step
    assert float(g1.abs().max()) > 0
    return l1


def test_dn_train_batch32_full_size():
    losses = _train_step_properties("dn", 32)
    assert losses[1] < losses[0] + 1e-4


def test_sr_train_batch32_full_size():
    _train_step_properties("sr", 32)


def test_dn_train_paper_loss_full_size():
    """the reference's shipped training loss (utils/loss_functions.py:11-47 with res/configs/loss_functions.toml, 'linear')"""
    from xmm_superres_denoise.utils import create_loss, load_loss_config
    _train_step_properties("dn", 16, loss=create_loss(*load_loss_config("linear")))


def test_input_pipeline_train_step_full_size():
    """configs[4]: every step starts from int32 count tiles (411 x 403) and runs detector-mask * pad * sqrt-normalize on the
    GPU; the composed tile equals mask_pad_normalize of the same counts bit for bit, and the step trains."""
    from xmm_superres_denoise.engine import compose_input, mask_pad_normalize
    from xmm_superres_denoise.parallel import DataParallelTrainer
    B = 16
    z = np.load(os.path.join(G, "example_data.npz"))
    m1 = np.unpackbits(z["mask1x_bits"])[: int(np.prod(z["mask1x_shape"]))].reshape(z["mask1x_shape"])
    mask = torch.from_numpy(m1).cuda()
    counts = torch.from_numpy(np.random.default_rng(2).poisson(0.1, size=(B, 411, 403)).astype(np.int32)).cuda()
    xin = compose_input(counts, None, None, mask, 512, 0.0022336, "sqrt")
    assert torch.equal(xin, mask_pad_normalize(counts, mask, 512, 0.0022336, "sqrt"))
    assert xin.shape == (B, 1, 512, 512) and float(xin.min()) >= 0 and float(xin.max()) <= 1
    inside = xin[:, :, 50:461, 54:457]           # 411 x 403 centred in 512 x 512: top 50, left 54
    assert float(xin.sum()) == float(inside.sum())          # everything outside the detector frame is padding
    m = build_module("dn", 4, 1, gc.make_state("dn", 32, 4, 3301)).set_math(HEADLINE)
    tr = DataParallelTrainer(m, lr=1e-4)
    t = _tiles((B, 1, 512, 512), 3302)
    l0 = float(tr.train_step(compose_input(counts, None, None, mask, 512, 0.0022336, "sqrt"), t))
    l1 = float(tr.train_step(compose_input(counts, None, None, mask, 512, 0.0022336, "sqrt"), t))
    assert np.isfinite([l0, l1]).all() and l1 < l0 + 1e-4


@pytest.mark.parametrize("math", ["f16x3", "bf16x6", "fp32"])
def test_backward_is_batch_independent_at_the_bench_batch(math):
    """BASELINE configs[2] batch (32 tiles of 512 x 512, 4 blocks): the input gradient of a tile must not depend on its batch
    neighbours.  dL/dx of tiles 0, 13 and 31 out of the batch-32 backward equals the dL/dx the same tiles get in a batch of
    their own (the forward's batch independence is asserted in test_hip_network.py; this is the backward's, at the batch
    where every persistent workgroup walks 64 tiles across many batch slices) -- BITWISE in the modes without operand
    scales (bf16x6, fp32).  f16x3 scales every operand plane by a power of two taken from the plane's max |x| over the
    WHOLE batch: a power of two commutes with every rounding except where an element's fp16 terms go subnormal (elements
    below 2^-28 of the plane's maximum -- gradient planes have them), so a different batch can move single results by an
    ulp; there the bar is 2e-6 of max |dL/dx| and at most one element in a thousand that differs at all.  The parameter
    gradients of the small batch are a partial sum of the big one's and only bound it."""
    state = gc.make_state("dn", 32, 4, 31337)
    m = build_module("dn", 4, 1, state).set_math(math)
    eng = m._get_engine(torch.device("cuda", 0))
    eng.pack(m.flat_parameters())
    B = 32
    x = _tiles((B, 1, 512, 512), 5).cuda()
    dy = ((_tiles((B, 1, 512, 512), 6) - 0.5) / (B * 512 * 512)).cuda()
    y = eng.forward(x, save_for_backward=True)
    g = torch.empty_like(m.flat_parameters())
    dx = eng.backward(dy, g, need_dx=True).clone()
    y = y.clone()
    assert torch.isfinite(dx).all() and torch.isfinite(g).all()
    pick = [0, 13, 31]
    xs, dys = x[pick].contiguous(), dy[pick].contiguous()
    del x, dy
    y3 = eng.forward(xs, save_for_backward=True)
    g3 = torch.empty_like(g)
    dx3 = eng.backward(dys, g3, need_dx=True)
    assert torch.equal(y3, y[pick])
    if math == "f16x3":
        d = (dx3 - dx[pick]).abs()
        ndiff, worst = int((d > 0).sum()), float(d.max() / dx.abs().max())
        print(f"f16x3 batch-32 vs batch-3 dL/dx: {ndiff} of {d.numel()} elements differ, worst {worst:.2e} of max |dL/dx|")
        assert worst <= 2e-6 and ndiff <= d.numel() // 1000
    else:
        assert torch.equal(dx3, dx[pick])
    assert float(g3.abs().max()) <= float(g.abs().max()) * 3.0 + 1e-30


@pytest.mark.parametrize("math", ["f16x3", "bf16x6"])
def test_results_do_not_depend_on_the_persistent_grid(math):
    """csrc/xsd_kernels.h: persistent_grid (round 6) picks HOW MANY workgroups walk the tiles of a conv launch from the tile count:
    416 x 416 at batch 1 = 338 tiles -> 169 workgroups, batch 4 = 1352 -> 226, batch 8 = 2704 -> the full 256; 512 x 512: full rounds, 256.  A tile's arithmetic must not depend on which workgroup runs it: the output and
    dL/dx of one image are BITWISE the same at every batch size when its batch neighbours are copies of it (identical planes: identical
    operand scales in f16x3), at the reference's tile and at BASELINE's."""
    state = gc.make_state("dn", 32, 4, 2718)
    m = build_module("dn", 4, 1, state).set_math(math)
    eng = m._get_engine(torch.device("cuda", 0))
    eng.pack(m.flat_parameters())
    g = torch.empty_like(m.flat_parameters())
    for T, batches in ((416, (1, 4, 8)), (512, (1, 2))):
        x1 = _tiles((1, 1, T, T), 41).cuda()
        dy1 = ((_tiles((1, 1, T, T), 42) - 0.5) / (T * T)).cuda()
        ref = None
        for B in batches:
            x, dy = x1.expand(B, 1, T, T).contiguous(), dy1.expand(B, 1, T, T).contiguous()
            y = eng.forward(x, save_for_backward=True).clone()
            dx = eng.backward(dy, g, need_dx=True).clone()
            assert torch.isfinite(y).all() and torch.isfinite(dx).all()
            for b in range(1, B):                                   # every copy alike within the batch
                assert torch.equal(y[b], y[0]) and torch.equal(dx[b], dx[0]), (T, B, b)
            if ref is None:
                ref = (y[0].clone(), dx[0].clone())
            else:
                assert torch.equal(y[0], ref[0]) and torch.equal(dx[0], ref[1]), (T, B)
