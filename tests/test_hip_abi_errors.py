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
