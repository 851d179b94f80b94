"""Error behaviour of the C ABI on the GPU box (include/xsd.h: negative status + xsd_last_error, never a crash or a silent
fallback).  The reference raises Python exceptions in the same situations (shape errors from torch, ValueError from
configure_model); the host layer turns the status codes into XsdError."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_status_codes_and_messages():
    from xmm_superres_denoise.engine import Engine, XsdError
    from xmm_superres_denoise.engine._lib import XsdConfig, load
    L = load()
    h = ctypes.c_void_p()
    for bad in (dict(kind=2), dict(num_filters=0), dict(in_channels=3, out_channels=2), dict(num_res_blocks=0), dict(kind=1, num_upsample=3)):
        cfg = dict(kind=0, in_channels=1, out_channels=1, num_filters=32, num_res_blocks=1, num_upsample=1, memory_efficient=0, reserved=0)
        cfg.update(bad)
        assert L.xsd_create(ctypes.byref(XsdConfig(**cfg)), ctypes.byref(h)) < 0, bad
        assert len(L.xsd_last_error()) > 0
    e = Engine("dn", 1, 1, 32, 1)
    x = torch.rand(1, 1, 16, 32, device="cuda")
    with pytest.raises(XsdError, match="xsd_pack_weights must be called"):
        e.forward(x)
    flat = torch.zeros(e.nparams, device="cuda")
    e.pack(flat)
    with pytest.raises(XsdError, match="x must be"):
        e.forward(torch.rand(1, 2, 16, 32, device="cuda"))
    with pytest.raises(XsdError, match="CUDA"):
        e.forward(torch.rand(1, 1, 16, 32))
    with pytest.raises(XsdError, match="float32"):
        e.forward(x.double())
    with pytest.raises(XsdError, match="wider than"):
        e.forward(torch.rand(1, 1, 8, 4100, device="cuda"))
    with pytest.raises(XsdError, match="elements"):
        e.pack(torch.zeros(e.nparams - 1, device="cuda"))
    with pytest.raises(XsdError, match="preceding forward"):
        e.backward(torch.zeros(1, 1, 16, 32, device="cuda"), torch.zeros_like(flat))
    assert L.xsd_set_math(e.h, 7) < 0 and b"math mode" in L.xsd_last_error()
    assert L.xsd_set_math(e.h, 1) < 0 and L.xsd_set_math(e.h, 2) < 0            # the 16-bit-significand modes are gone
    # the split modes address a plane's batch slice with 32-bit BYTE offsets: 2^24 pixels x 128 B no longer fit -> refused
    # up front, with a message, instead of range-checked loads returning zeros; the exact-fp32 mode takes it
    e.set_math("f16x3")
    with pytest.raises(XsdError, match="too large for math modes"):
        e.forward(torch.zeros(1, 1, 4096, 4096, device="cuda"))
    e.set_math("bf16x6")
    e.pack(flat)
    with pytest.raises(XsdError, match="too large for math modes"):
        e.forward(torch.zeros(1, 1, 4096, 4096, device="cuda"))
    e.set_math("f16x3")
    e.pack(flat)
    assert L.xsd_forward(e.h, None, None, 1, 16, 32, 0, None) < 0
    assert L.xsd_normalize(None, None, 0, 1.0, 0, 0, None) < 0
    assert L.xsd_mask_pad_normalize(None, 1, None, None, 1, 8, 8, 16, 1, -1.0, 9, None) < 0
    # and the engine still works after all of that
    y = e.forward(x)
    assert y.shape == x.shape and torch.isfinite(y).all()


def test_ragged_and_tiny_shapes():
    """1 x 1 pixel, single row / column, sizes that are not multiples of any tile: outputs equal the oracle's"""
    import numpy as np
    import gen_common as gc
    from oracle import oracle
    from util_hip import build_module
    state = gc.make_state("dn", 32, 1, 611)
    m = build_module("dn", 1, 1, state)
    for shape in [(1, 1, 1, 1), (2, 1, 1, 37), (1, 1, 35, 1), (3, 1, 5, 3), (1, 1, 17, 33)]:
        x = gc.make_input(shape, 612)
        yo = oracle.forward("dn", 32, 1, oracle.flatten_state(state), x)
        with torch.no_grad():
            y = m(torch.from_numpy(x).cuda()).cpu().numpy()
        assert y.shape == yo.shape and np.abs(y - yo).max() < 1e-5, shape


def test_batch_that_cannot_fit_fails_cleanly_and_memory_efficient_trains_it():
    """rrdb_blocks.py:39-47: the reference's way out of a batch whose activations do not fit is `memory_efficient` (recompute).
    A training step keeps 62 planes of 128 B per pixel = 2.1 GB per 512 x 512 tile: 160 tiles = 333 GiB, more than the device has.
    The engine must (a) refuse it with XSD_ERR_NOMEM and a message that names memory_efficient, (b) stay usable: a batch that fits
    then gives results bit-equal to a fresh engine's, (c) train the SAME batch with memory_efficient=True."""
    import numpy as np
    import gen_common as gc
    from util_hip import build_module
    from xmm_superres_denoise.engine import XsdError
    from xmm_superres_denoise.models import GeneratorRRDB_DN
    from xmm_superres_denoise.parallel import DataParallelTrainer
    total = torch.cuda.get_device_properties(0).total_memory
    B, T = 160, 512
    assert B * 62 * 128 * T * T > total, "this device would hold the batch: raise B"
    state = gc.make_state("dn", 32, 4, 4242)
    m = build_module("dn", 4, 1, state)
    g = torch.Generator().manual_seed(5)
    xs = torch.rand((2, 1, 96, 64), generator=g).cuda()
    ts = torch.rand((2, 1, 96, 64), generator=g).cuda()
    tr = DataParallelTrainer(m, lr=1e-4)
    x = torch.rand((B, 1, T, T), generator=g).cuda()
    t = torch.rand((B, 1, T, T), generator=g).cuda()
    before = tr.flat.clone()
    with pytest.raises(XsdError, match="memory_efficient") as ei:
        tr.train_step(x, t)
    assert "GiB" in str(ei.value) and "does not fit" in str(ei.value)
    assert torch.equal(tr.flat, before) and tr.step_count == 0           # nothing was updated by the refused step
    # (b) the engine that refused goes on: same small step as a fresh module, bit for bit (loss, every gradient, updated weights)
    l1 = tr.train_step(xs, ts)
    m2 = build_module("dn", 4, 1, state)
    tr2 = DataParallelTrainer(m2, lr=1e-4)
    l2 = tr2.train_step(xs, ts)
    torch.cuda.synchronize()
    assert torch.equal(l1, l2) and torch.equal(tr.grads, tr2.grads) and torch.equal(tr.flat, tr2.flat)
    # a forward-only batch of the same size fits (9 recycled planes) and still works after the refusal
    del tr, tr2, m2
    with torch.no_grad():
        y = m(x[:64])
    assert y.shape == (64, 1, T, T) and bool(torch.isfinite(y).all())
    del y, m
    torch.cuda.empty_cache()
    # (c) the same 160-tile batch trains with memory_efficient=True (8 tiles at a time are recomputed and back-propagated)
    mme = GeneratorRRDB_DN(1, 1, 32, 4, memory_efficient=True)
    mme.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    mme = mme.cuda()
    trm = DataParallelTrainer(mme, lr=1e-4)
    w0 = trm.flat.clone()
    la = float(trm.train_step(x, t))
    lb = float(trm.train_step(x, t))
    torch.cuda.synchronize()
    assert np.isfinite(la) and np.isfinite(lb) and bool(torch.isfinite(trm.grads).all())
    assert not torch.equal(trm.flat, w0) and lb < la                   # two Adam steps on a fixed batch: the loss falls
    # ... and its gradient is the full-batch gradient: against 16 of the tiles kept whole on a second engine (chunks are independent)
    m16 = build_module("dn", 4, 1, state)
    tr16 = DataParallelTrainer(m16, lr=1e-4)
    trm2 = DataParallelTrainer(GeneratorRRDB_DN(1, 1, 32, 4, memory_efficient=True).cuda(), lr=1e-4)
    trm2.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    trm2.flat = trm2.model.flat_parameters()
    tr16.train_step(x[:16].contiguous(), t[:16].contiguous())
    trm2.train_step(x[:16].contiguous(), t[:16].contiguous())
    torch.cuda.synchronize()
    scale = float(tr16.grads.abs().max())
    assert float((tr16.grads - trm2.grads).abs().max()) <= 2e-6 * scale + 1e-12


def test_non_finite_first_pixel_does_not_leak_through_the_zero_padding():
    """The 32 -> 1 edge conv fetched out-of-image taps from the image's FIRST pixel and multiplied them by 0 -- 0 * nan = nan: a NaN at
    pixel (0, 0) came out along every border of the tile.  Now a select.  The set of non-finite outputs equals the oracle's: the 19 x 19
    corner the 18 convs reach from (0, 0)."""
    import numpy as np
    import gen_common as gc
    from oracle import oracle
    from util_hip import build_module
    state = gc.make_state("dn", 32, 1, 977)
    m = build_module("dn", 1, 1, state)
    x = gc.make_input((2, 1, 80, 72), 978)
    x[1, 0, 0, 0] = np.nan
    with np.errstate(invalid="ignore", over="ignore"):
        yo = oracle.forward("dn", 32, 1, oracle.flatten_state(state), x)
    bad_o = ~np.isfinite(yo)
    assert bad_o[0].sum() == 0 and bad_o[1].sum() == 19 * 19
    for mode in ("f16x3", "bf16x6", "fp32"):
        m.set_math(mode)
        xd = torch.from_numpy(x).cuda().requires_grad_(True)
        y = m(xd)
        assert np.array_equal(~np.isfinite(y.detach().cpu().numpy()), bad_o), mode
        # ... and backward: dL/dx of the clean tile is finite everywhere (conv_first's input-gradient is the same 32 -> 1 kernel)
        y[0].sum().backward()
        g = xd.grad.cpu().numpy()
        assert np.isfinite(g[0]).all(), mode


@pytest.mark.parametrize("bad", ["nan", "inf", "-inf"])
def test_non_finite_input_propagates_like_the_reference(bad):
    """One non-finite input pixel: torch propagates it through every conv it reaches (a 3 x 3 conv spreads it by one pixel, 0 * nan = nan,
    inf - inf = nan), LeakyReLU and the residual sums keep it, torch.clamp keeps nan and sends +-inf to the bounds
    (generator_rrdb.py:130-137, models/model.py:48-49).  The set of non-finite OUTPUT pixels of the engine equals the oracle's in all
    three math modes, the finite ones stay within the usual tolerance, and the tile next to it in the batch is untouched.
    (Until round 6 the output clamp was fminf(fmaxf()), which returns the non-nan operand: a nan pixel came out as 0.)"""
    import numpy as np
    import gen_common as gc
    from oracle import oracle
    from util_hip import build_module
    state = gc.make_state("dn", 32, 1, 977)
    m = build_module("dn", 1, 1, state)
    x = gc.make_input((2, 1, 80, 72), 978)
    x[0, 0, 40, 30] = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf}[bad]
    with np.errstate(invalid="ignore", over="ignore"):
        yo = oracle.forward("dn", 32, 1, oracle.flatten_state(state), x)
    bad_o = ~np.isfinite(yo)
    # 18 convs between input and output (conv_first, 15 in the dense blocks, trunk_conv, conv_last): everything within 18 pixels of (40, 30) in tile 0, nothing in tile 1
    assert bad_o[1].sum() == 0 and bad_o[0].sum() == 37 * 37, bad_o.sum()
    for mode in ("f16x3", "bf16x6", "fp32"):
        m.set_math(mode)
        with torch.no_grad():
            y = m(torch.from_numpy(x).cuda()).cpu().numpy()
        bad_e = ~np.isfinite(y)
        assert np.array_equal(bad_e, bad_o), (mode, int(bad_e.sum()), int(bad_o.sum()), int((bad_e ^ bad_o).sum()))
        assert np.isnan(y[bad_e]).all() and np.isnan(yo[bad_o]).all()            # clamp leaves no inf behind, in either
        ok = ~bad_o
        assert np.abs(y[ok] - yo[ok]).max() < 2e-5, (mode, float(np.abs(y[ok] - yo[ok]).max()))
        # and the engine is not poisoned: the same module on a clean batch right afterwards
        xc = gc.make_input((2, 1, 80, 72), 979)
        with torch.no_grad():
            yc = m(torch.from_numpy(xc).cuda()).cpu().numpy()
        assert np.isfinite(yc).all() and np.abs(yc - oracle.forward("dn", 32, 1, oracle.flatten_state(state), xc)).max() < 1e-5, mode


@pytest.mark.parametrize("kind", ["dn", "sr"])
def test_empty_batch_like_torch(kind):
    """The reference's modules are torch convs: an empty batch gives an empty output of the right shape and zero gradients (it is what a
    rank with an empty shard sees).  The C ABI refuses B < 1 (include/xsd.h), so the module answers without a launch -- and a real batch
    through the same module afterwards is untouched by it."""
    import gen_common as gc
    from util_hip import build_module
    state = gc.make_state(kind, 32, 1, 731)
    m = build_module(kind, 1, 1, state)
    s = 2 if kind == "sr" else 1
    x = torch.zeros(0, 1, 24, 40, device="cuda", requires_grad=True)
    y = m(x)
    assert tuple(y.shape) == (0, 1, 24 * s, 40 * s) and y.requires_grad
    y.sum().backward()
    assert tuple(x.grad.shape) == (0, 1, 24, 40)
    for n, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape and not p.grad.any(), n
    with torch.no_grad():
        assert tuple(m(torch.zeros(0, 1, 8, 8, device="cuda")).shape) == (0, 1, 8 * s, 8 * s)
    from xmm_superres_denoise.engine import XsdError
    with pytest.raises(XsdError, match="x must be"):
        m(torch.zeros(0, 2, 8, 8, device="cuda"))
    xr = torch.from_numpy(gc.make_input((2, 1, 24, 40), 732)).cuda()
    with torch.no_grad():
        y1 = m(xr)
        m(torch.zeros(0, 1, 24, 40, device="cuda"))
        y2 = m(xr)
    assert torch.equal(y1, y2)


def test_second_differentiation_is_refused_by_name():
    """The backward is HIP kernels, not torch ops: a gradient of a gradient (create_graph=True, e.g. a gradient penalty -- the reference has
    none) cannot be formed.  torch must refuse (the first gradient carries no graph; where the incoming gradient itself requires grad,
    once_differentiable names the function) instead of returning a second gradient that silently ignores the path."""
    import gen_common as gc
    from util_hip import build_module
    m = build_module("dn", 1, 1, gc.make_state("dn", 32, 1, 781))
    x = torch.rand(1, 1, 16, 32, device="cuda", requires_grad=True)
    (g,) = torch.autograd.grad(m(x).sum(), x, create_graph=True)
    assert g.shape == x.shape
    with pytest.raises(RuntimeError, match="once_differentiable|does not require grad"):
        g.sum().backward()
    w = torch.ones(1, 1, 16, 32, device="cuda", requires_grad=True)
    (g2,) = torch.autograd.grad((m(x) * w).sum(), x, create_graph=True)      # now the incoming gradient requires grad
    with pytest.raises(RuntimeError, match="once_differentiable"):
        g2.sum().backward()
