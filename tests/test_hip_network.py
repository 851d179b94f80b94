"""GPU parity of the whole generators (forward, L1 backward) against (1) the golden vectors generated from the
reference modules and (2) the CPU oracle on fresh seeded inputs, through the drop-in nn.Module API.
north_star tolerance: 1e-3 relative fp32; measured errors are ~1e-6..1e-5, the asserts use 1e-4."""
import os

import numpy as np
import pytest
import torch

import gen_common as gc
from oracle import oracle
from util_hip import FLIP_LOG, G, assert_grad_close, build_module, flip_candidates, load_case

pytestmark = pytest.mark.gpu

CASES = [("dn_nf32_b4_32x32", "dn"), ("dn_nf32_b4_24x40", "dn"), ("sr_nf32_b4_24x40", "sr"),
         ("sr_nf32_b4_17x45", "sr"), ("dn_nf32_b1_64x64", "dn")]
# the reference's goldens at other widths: 8 filters (reduced model of SURVEY 7.1; full gradient tensors stored; runs
# zero-padded to one plane on the split-precision kernels since round 3) and the round-4 set generated from the reference's own
# constructors at 64 / 48 / 16 filters and 3 -> 2 / 1 -> 3 image channels (tests/golden/make_golden.py:width_cases) -- every
# one of them in every math mode, held to the same flip-aware bar as the shipped width
NF8_CASES = [("dn_nf8_b1", "dn"), ("sr_nf8_b1", "sr"), ("sr_nf8_b1_up2", "sr")]
WIDTH_CASES = [("dn_nf64_b1", "dn"), ("sr_nf32_c3x2_b1", "sr"), ("dn_nf16_c1x3_b1", "dn"), ("sr_nf64_b1_up2", "sr"), ("dn_nf48_b1", "dn")]


def _relmax(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


MATHS = ["fp32", "bf16x6", "f16x3"]
FP32_CLASS = tuple(MATHS)   # every mode is held to the flip-aware comparison: only rows named by a flip candidate may deviate
# gradient tolerances (see util_hip.assert_grad_close), the same for every math mode: 2e-4 of the tensor's largest entry on
# every row that has no flip candidate (north_star: 1e-3)
TIGHT = {m: 2e-4 for m in MATHS}
DX_TIGHT = {m: 4e-4 for m in MATHS}


def _ripple_bar(x):
    """What a flip candidate may do to the rows it does NOT name.  One LeakyReLU' decision that falls the other way changes one
    (pixel, channel) term of the backward pass; every gradient upstream of it moves by about that term's share of the sum over
    the pixels.  On the 1000+-pixel cases that ripple is far below north_star's 1e-3, which is the bar there; on the
    9 x 11-pixel golden (sr_nf64_b1_up2: 99 pixels) one decision is 1 / 99 of everything -- measured 5e-3 on conv_first.weight --
    so the bar is max(1e-3, 1 / pixels).  It applies only when the float64 evaluation names a candidate at all."""
    return max(1e-3, 1.0 / float(x.shape[0] * x.shape[2] * x.shape[3]))


def _report_flips(tag, before):
    recs = FLIP_LOG[before:]
    if recs:
        print(f"[flip report] {tag}: " + "; ".join(f"{r['tensor']}: {r['rows_over_tight']} rows over tight, {r['candidate_rows']} candidate rows, "
                                                   f"{r.get('ripple_rows', 0)} rippled from a downstream candidate, {r['unexplained']} unexplained (max {r['max_rel_err']:.1e})" for r in recs))


@pytest.mark.parametrize("name,kind,math", [(n, k, m) for n, k in CASES + NF8_CASES + WIDTH_CASES for m in MATHS])
def test_golden_forward_backward(name, kind, math):
    z, nf, blocks, nup, state, x, t = load_case(name, kind)
    m = build_module(kind, blocks, nup, state, nf=nf, in_ch=x.shape[1], out_ch=t.shape[1]).set_math(math)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    y = m(xd)
    assert np.abs(y.detach().cpu().numpy() - z["y"]).max() < 1e-4
    loss = torch.nn.functional.l1_loss(y, torch.from_numpy(t).cuda())
    assert abs(loss.item() - float(z["loss"][0])) < 1e-5
    loss.backward()
    cand, n_out = flip_candidates(kind, blocks, state, x, t, nup) if math in FP32_CLASS else (None, 0)
    mark = len(FLIP_LOG)
    strict = _ripple_bar(x)
    assert_grad_close(xd.grad.cpu().numpy().reshape(-1, x.shape[-1]), z["dx"].reshape(-1, x.shape[-1]), "dx", tight=DX_TIGHT[math], loose=5e-2, max_flip_frac=0.3,
                      candidates=cand, n_out_candidates=n_out, strict=strict)
    names = [str(n) for n in z["param_names"]]
    params = dict(m.named_parameters())
    assert list(params.keys()) == names
    for i, n in enumerate(names):
        g = params[n].grad.cpu().numpy().astype(np.float64)
        s_ref, a_ref = z["grad_sums"][i]
        assert abs(np.abs(g).sum() - a_ref) <= 1e-2 * a_ref + 1e-9, n
        assert abs(g.sum() - s_ref) <= 1e-2 * a_ref + 1e-9, n
        if "grad." + n in z.files:
            assert_grad_close(g, z["grad." + n], n, tight=TIGHT[math], candidates=cand, n_out_candidates=n_out, strict=strict)
    _report_flips(f"{name} {math}", mark)


def test_forward_no_grad_matches_train_forward_and_reuses_planes():
    z, nf, blocks, nup, state, x, t = load_case("sr_nf32_b4_24x40", "sr")
    m = build_module("sr", blocks, nup, state)
    with torch.no_grad():
        y = m(torch.from_numpy(x).cuda())
    assert np.abs(y.cpu().numpy() - z["y"]).max() < 1e-4


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("kind,shape,nup", [("dn", (3, 1, 41, 67), 1), ("sr", (2, 1, 33, 35), 1), ("sr", (2, 1, 19, 22), 2)])
def test_fresh_inputs_vs_oracle(kind, shape, nup, math):
    """nup = 2 is the class default of GeneratorRRDB_SR (generator_rrdb.py:79): two conv + PixelShuffle stages, 4x output"""
    state = gc.make_state(kind, 32, 2, 900, num_upsample=nup, last_bias=0.3 if kind == "sr" else None)
    x = gc.make_input(shape, 901)
    s = 2 ** nup if kind == "sr" else 1
    t = gc.make_input((shape[0], 1, shape[2] * s, shape[3] * s), 902)
    yo, lo, dxo, go = oracle.l1_train(kind, 32, 2, oracle.flatten_state(state), x, t, num_upsample=nup)
    m = build_module(kind, 2, nup, state).set_math(math)
    eng = m._get_engine(torch.device("cuda", 0))
    eng.pack(m.flat_parameters())
    xd = torch.from_numpy(x).cuda()
    y = eng.forward(xd, save_for_backward=True)
    loss, dy = eng.l1_loss(y, torch.from_numpy(t).cuda())
    grads = torch.empty_like(m.flat_parameters())
    dx = eng.backward(dy, grads, need_dx=True)
    assert np.abs(y.cpu().numpy() - yo).max() < 1e-4
    assert abs(loss.item() - lo) < 1e-5
    cand, n_out = flip_candidates(kind, 2, state, x, t, nup) if math in FP32_CLASS else (None, 0)
    mark = len(FLIP_LOG)
    assert_grad_close(dx.cpu().numpy().reshape(-1, dxo.shape[-1]), dxo.reshape(-1, dxo.shape[-1]), "dx", tight=DX_TIGHT[math], loose=5e-2, max_flip_frac=0.3,
                      candidates=cand, n_out_candidates=n_out)
    g = grads.cpu().numpy()
    shapes = gc.rrdb_param_shapes(kind, 32, 2, num_upsample=nup)
    off = 0
    for n, shp in shapes.items():
        k = int(np.prod(shp))
        assert_grad_close(g[off:off + k].reshape(shp), go[off:off + k].reshape(shp), n, tight=TIGHT[math], candidates=cand, n_out_candidates=n_out)
        off += k
    _report_flips(f"{kind}{shape} nup={nup} {math}", mark)


def test_staged_backward_equals_monolithic():
    z, nf, blocks, nup, state, x, t = load_case("dn_nf32_b4_24x40", "dn")
    m = build_module("dn", blocks, nup, state)
    eng = m._get_engine(torch.device("cuda", 0))
    eng.pack(m.flat_parameters())
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    y = eng.forward(xd, save_for_backward=True)
    _, dy = eng.l1_loss(y, td)
    g1 = torch.zeros_like(m.flat_parameters())
    eng.backward(dy, g1)
    g2 = torch.full_like(g1, float("nan"))
    covered = 0
    for st in range(eng.num_stages):
        eng.backward_stage(st, dy, g2)
        off, cnt = eng.grad_range(st)
        covered += cnt
        torch.cuda.synchronize()
        assert torch.isfinite(g2[off:off + cnt]).all()
    assert covered == g1.numel()
    assert torch.equal(g1, g2)


# Gradient floor (relative to the largest gradient) above which a parameter's accumulated Adam update must agree with the
# oracle's to 5e-8, and the share of parameters that have to be above it: ONE rule for every math mode, set by the least
# precise one (tools/calib_adam.py).  Adam divides by |g|, so an absolute gradient error d moves the update by lr * d / |g|: with
# f16x3's 22-23 significant operand bits a parameter whose gradient sits at 1e-6 of the largest moves 1.3e-6 differently,
# from 1e-5 up (65 % of the parameters) all three modes agree with the oracle to 1.9e-8.
ADAM_SOLID = (1e-5, 0.6)


@pytest.mark.parametrize("math", FP32_CLASS)
def test_train_step_adam_matches_oracle(math):
    """3 optimisation steps (L1 + Adam lr 1e-4) from the same weights: engine vs oracle."""
    kind, blocks = "dn", 1
    state = gc.make_state(kind, 32, blocks, 77)
    x = gc.make_input((2, 1, 24, 40), 78)
    t = gc.make_input((2, 1, 24, 40), 79)
    p = oracle.flatten_state(state).copy()
    mo, vo = np.zeros_like(p), np.zeros_like(p)
    m = build_module(kind, blocks, 1, state).set_math(math)
    eng = m._get_engine(torch.device("cuda", 0))
    flat = m.flat_parameters()
    md, vd = torch.zeros_like(flat), torch.zeros_like(flat)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    grads = torch.empty_like(flat)
    g_min = None
    for step in range(1, 4):
        _, lo, _, go = oracle.l1_train(kind, 32, blocks, p, x, t)
        g_min = np.abs(go) if g_min is None else np.minimum(g_min, np.abs(go))
        oracle.adam(p, go, mo, vo, step)
        eng.pack(flat)
        y = eng.forward(xd, save_for_backward=True)
        loss, dy = eng.l1_loss(y, td)
        eng.backward(dy, grads)
        eng.adam_step(flat, grads, md, vd, step)
        assert abs(loss.item() - lo) < 1e-5
    start = oracle.flatten_state(state)
    d_eng = flat.cpu().numpy().astype(np.float64) - start
    d_ora = p.astype(np.float64) - start
    # Adam normalises the step (about lr per step whatever the gradient scale), so a parameter whose gradient is within
    # rounding of zero may legitimately move either way.  Everywhere else -- gradients that stay above ADAM_SOLID's floor
    # in all three steps -- the accumulated updates must agree to 5e-8 = 0.05 % of one step (lr = 1e-4; measured 7.5e-9 to
    # 1.9e-8, one to two ulp of a weight).
    floor, share = ADAM_SOLID
    solid = g_min > floor * np.abs(go).max()
    assert solid.mean() > share
    assert np.abs(d_eng[solid] - d_ora[solid]).max() < 5e-8, np.abs(d_eng[solid] - d_ora[solid]).max()
    assert np.abs(d_eng - d_ora).max() < 2.5e-4          # nobody moves by more than the three steps allow
    assert np.median(np.abs(d_ora[solid])) > 1e-4        # and the solid ones really moved (about 3 steps of lr)


@pytest.mark.parametrize("math", MATHS)
def test_example_data_psnr_parity(math):
    """PSNR(engine) within 0.01 dB of PSNR(reference) on example_data tiles with identical seeded weights."""
    from xmm_superres_denoise.data.tools import load_and_prepare
    z = np.load(os.path.join(G, "example_data.npz"))
    m1 = np.unpackbits(z["mask1x_bits"])[: int(np.prod(z["mask1x_shape"]))].reshape(z["mask1x_shape"])
    m2 = np.unpackbits(z["mask2x_bits"])[: int(np.prod(z["mask2x_shape"]))].reshape(z["mask2x_shape"])
    m1d, m2d = torch.from_numpy(m1).cuda(), torch.from_numpy(m2).cuda()
    dn = build_module("dn", 4, 1, gc.make_state("dn", 32, 4, 1234)).set_math(math)
    sr = build_module("sr", 4, 1, gc.make_state("sr", 32, 4, 4321, last_bias=0.05)).set_math(math)
    with torch.no_grad():
        for i in range(2):
            x = load_and_prepare(torch.from_numpy(z[f"dn_counts20_{i}"]).cuda()[None], m1d, 416, 0.0022336, "sqrt")
            t = load_and_prepare(torch.from_numpy(z[f"dn_counts50_{i}"]).cuda()[None], m1d, 416, 0.0022336, "sqrt")
            assert abs(float(x.double().sum()) - float(z[f"dn_x_sum_{i}"][0])) < 1e-3
            y = dn(x)[0, 0].cpu().numpy()
            assert np.abs(y[::5, ::5] - z[f"dn_y_sub_{i}"]).max() < 1e-4
            assert abs(gc.psnr(y, t[0, 0].cpu().numpy()) - float(z[f"dn_psnr_{i}"][0])) < 0.01
            x = load_and_prepare(torch.from_numpy(z[f"sr_counts_lr_{i}"]).cuda()[None], m1d, 416, 0.0022336, "sqrt")
            t = load_and_prepare(torch.from_numpy(z[f"sr_counts_hr_{i}"]).cuda()[None], m2d, 832, 0.0005584, "sqrt")
            y = sr(x)[0, 0].cpu().numpy()
            assert np.abs(y[::9, ::9] - z[f"sr_y_sub_{i}"]).max() < 1e-4
            assert abs(gc.psnr(y, t[0, 0].cpu().numpy()) - float(z[f"sr_psnr_{i}"][0])) < 0.01


def test_infer_file_end_to_end(tmp_path):
    """FITS (big-endian int32 counts) -> mask/pad/normalize -> SR generator -> denormalize -> FITS with the reference's
    WCS bookkeeping; the prediction equals the oracle's forward on the same prepared input."""
    from collections import OrderedDict
    from xmm_superres_denoise.infer import infer_file, read_fits, write_fits
    z = np.load(os.path.join(G, "example_data.npz"))
    m1 = np.unpackbits(z["mask1x_bits"])[: int(np.prod(z["mask1x_shape"]))].reshape(z["mask1x_shape"])
    counts = z["sr_counts_lr_0"]
    src = os.path.join(tmp_path, "P0123_detxy.fits")
    # the writer emits float32; build an int32 BITPIX-32 file by hand to exercise the raw big-endian path
    hdr = OrderedDict(CRPIX1=201.5, CRPIX2=205.5, CDELT1=-0.0011, CDELT2=0.0011, PA_PNT=12.5, EXPOSURE=20000.0)
    write_fits(src, counts.astype(np.float32), hdr)
    state = gc.make_state("sr", 32, 1, 4321, last_bias=0.05)
    m = build_module("sr", 1, 1, state)
    y, out_path = infer_file(src, m, torch.from_numpy(m1).cuda(), os.path.join(tmp_path, "out"))
    x = oracle.normalize(oracle.mask_pad(counts, m1, 416), 0.0022336, "sqrt")[None]
    yo = oracle.forward("sr", 32, 1, oracle.flatten_state(state), x)
    yo = oracle.denormalize(yo, 0.0005584, "sqrt")[0, 0]
    assert y.shape == (832, 832) and np.abs(y - yo).max() < 1e-4 * 0.0005584 + 1e-9
    back, h = read_fits(out_path)
    assert np.array_equal(back.astype(np.float32), y.astype(np.float32))
    assert h["CRPIX1"] == 2 * (201.5 + 6) + 0.5 and h["CDELT2"] == 0.00055 and h["IMG_FILE"] == "P0123_detxy.fits"


def test_train_driver_overfits_fixed_batch_and_checkpoints(tmp_path):
    """The train driver's pieces: loss goes down on a fixed batch under L1 + Adam, and a checkpoint in the reference's
    Lightning layout ('model.' + reference key names) restores the exact weights."""
    from xmm_superres_denoise.config.config import model_cfg
    from xmm_superres_denoise.models import Model
    from xmm_superres_denoise.parallel import DataParallelTrainer
    from xmm_superres_denoise.train import load_checkpoint, save_checkpoint
    torch.manual_seed(0)
    model = Model(model_cfg("rrdb_denoise", batch_size=2, residual_blocks=1), (64, 64), (64, 64), None, None, None, None, None)
    model.configure_model()
    model.cuda()
    tr = DataParallelTrainer(model.model, lr=1e-3)
    x = torch.rand(2, 1, 64, 64, device="cuda")
    t = (x * 0.5 + 0.1).contiguous()
    losses = [float(tr.train_step(x, t)) for _ in range(12)]
    assert losses[-1] < 0.7 * losses[0]
    p = os.path.join(tmp_path, "last.ckpt")
    save_checkpoint(p, model, tr, epoch=0)
    ck = torch.load(p, map_location="cpu", weights_only=True)
    assert "model.rrdb.0.RDB3.conv5.weight" in ck["state_dict"] and "model.conv_first.bias" in ck["state_dict"]
    m2 = Model(model_cfg("rrdb_denoise", batch_size=2, residual_blocks=1), (64, 64), (64, 64), None, None, None, None, None)
    load_checkpoint(p, m2)
    m2.cuda()
    with torch.no_grad():
        assert torch.equal(m2(x), model(x))


def test_full_size_properties():
    """BASELINE-size tiles (1x512x512, full 4-block DN) where the oracle is too slow: size-independent properties.
    (a) determinism: two runs are bit-identical; (b) batch independence: a tile's output does not depend on its batch
    neighbours (bitwise); (c) the three math modes agree within 1e-4 (tolerance 1e-3); (d) the backward pass is linear
    in dy: grads(2*dy) == 2*grads(dy) up to rounding; (e) outputs are clamped to [0,1]."""
    state = gc.make_state("dn", 32, 4, 2024)
    x = torch.from_numpy(gc.make_input((2, 1, 512, 512), 2025)).cuda()
    ys = {}
    for math in MATHS:
        m = build_module("dn", 4, 1, state).set_math(math)
        with torch.no_grad():
            y1 = m(x)
            y2 = m(x)
            y_single = m(x[1:2].contiguous())
        assert torch.equal(y1, y2), math
        assert torch.equal(y1[1:2], y_single), math
        assert float(y1.min()) >= 0.0 and float(y1.max()) <= 1.0
        ys[math] = y1
        if math == "fp32":
            eng = m._get_engine(torch.device("cuda", 0))
            eng.pack(m.flat_parameters())
            y = eng.forward(x, save_for_backward=True)
            dy = torch.from_numpy(gc.make_input((2, 1, 512, 512), 2026) - 0.5).cuda() / x.numel()
            g1 = torch.empty_like(m.flat_parameters())
            g2 = torch.empty_like(g1)
            eng.backward(dy, g1)
            eng.backward((2.0 * dy).contiguous(), g2)
            assert float((g2 - 2.0 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())
    assert float((ys["bf16x6"] - ys["fp32"]).abs().max()) < 5e-6
    assert float((ys["f16x3"] - ys["fp32"]).abs().max()) < 5e-6


@pytest.mark.parametrize("kind", ["dn", "sr", "dn_16_filters_2_channels", "dn_64_filters"])
def test_memory_efficient_recompute_matches_full_batch(kind, monkeypatch):
    """memory_efficient=True (rrdb_blocks.py:39-47: same math, activations recomputed in backward) keeps no activations
    between forward and backward and recomputes XSD_ME_CHUNK tiles at a time.  Outputs are bitwise those of the plain
    path; gradients differ only by the summation order over chunks (5 tiles in chunks of 2 here)."""
    from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
    from xmm_superres_denoise.parallel import DataParallelTrainer
    monkeypatch.setenv("XSD_ME_CHUNK", "2")
    torch.manual_seed(0)
    ch = 2 if kind == "dn_16_filters_2_channels" else 1        # (16 filters zero-padded to 32, two image channels: Builder::build_multi)
    mk = (lambda me: GeneratorRRDB_DN(1, 1, 32, 1, memory_efficient=me)) if kind == "dn" else \
         (lambda me: GeneratorRRDB_SR(1, 1, 32, 1, num_upsample=1, memory_efficient=me)) if kind == "sr" else \
         (lambda me: GeneratorRRDB_DN(1, 1, 64, 1, memory_efficient=me)) if kind == "dn_64_filters" else \
         (lambda me: GeneratorRRDB_DN(2, 2, 16, 1, memory_efficient=me))
    plain, me = mk(False).cuda(), mk(True).cuda()
    me.load_state_dict(plain.state_dict())
    for m in (plain, me):
        m.set_math("fp32")
    s = 2 if kind == "sr" else 1
    x = torch.rand(5, ch, 40, 48, device="cuda")
    t = torch.rand(5, ch, 40 * s, 48 * s, device="cuda")
    outs, grads, dxs = [], [], []
    for m in (plain, me):
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        (y - t).abs().mean().backward()
        outs.append(y.detach()); dxs.append(xi.grad.clone())
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]))
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(dxs[0], dxs[1])                      # per-tile input gradients do not depend on the batching
    assert (grads[0] - grads[1]).abs().max().item() <= 2e-6 * grads[0].abs().max().item()
    # trainer path: identical update (up to that summation order) and both replicas keep training
    ta, tb = DataParallelTrainer(plain, lr=1e-3), DataParallelTrainer(me, lr=1e-3)
    la, lb = float(ta.train_step(x, t)), float(tb.train_step(x, t))
    assert la == lb
    assert (ta.grads - tb.grads).abs().max().item() <= 2e-6 * ta.grads.abs().max().item()


def test_two_forwards_before_backward_are_reentrant():
    """y1 = m(x1); y2 = m(x2); (l1 + l2).backward(): the engine holds ONE saved activation set, so the context of y1 must
    notice that it was displaced and recompute instead of using x2's activations (different shapes on purpose: using
    the wrong set would index out of bounds).  Gradients must equal the sum of two separate passes."""
    state = gc.make_state("dn", 32, 1, 41)
    m = build_module("dn", 1, 1, state).set_math("fp32")
    x1 = torch.from_numpy(gc.make_input((2, 1, 24, 40), 42)).cuda()
    x2 = torch.from_numpy(gc.make_input((1, 1, 33, 19), 43)).cuda()
    t1, t2 = torch.rand_like(x1), torch.rand_like(x2)

    def separate(x, t):
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.nn.functional.l1_loss(m(xi), t).backward()
        return xi.grad.clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    dx1, g1 = separate(x1, t1)
    dx2, g2 = separate(x2, t2)
    for p in m.parameters():
        p.grad = None
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    y1 = m(a)
    y2 = m(b)
    (torch.nn.functional.l1_loss(y1, t1) + torch.nn.functional.l1_loss(y2, t2)).backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.equal(a.grad, dx1) and torch.equal(b.grad, dx2)
    assert (g - (g1 + g2)).abs().max().item() <= 1e-6 * (g1 + g2).abs().max().item()


def test_first_forward_under_inference_mode_leaves_the_module_trainable():
    """Lightning validates -- and sanity-checks before the first training step -- under torch.inference_mode(): the module's first
    forward, the one that lays the parameters into the engine's flat buffer, happens there.  A buffer created in that mode would be an
    inference tensor (no version counter, no in-place update outside the mode): the module could never be trained afterwards.  The
    same steps with and without that first validation forward must give the same parameters."""
    state = gc.make_state("dn", 32, 1, 741)
    x = torch.from_numpy(gc.make_input((2, 1, 24, 40), 742)).cuda()
    t = torch.from_numpy(gc.make_input((2, 1, 24, 40), 743)).cuda()

    def run(validate_first):
        m = build_module("dn", 1, 1, state)
        if validate_first:
            with torch.inference_mode():
                yv = m(x)
            assert yv.is_inference() and not m._flat.is_inference() and not any(p.is_inference() for p in m.parameters())
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        for _ in range(2):
            opt.zero_grad()
            torch.nn.functional.l1_loss(m(x), t).backward()
            opt.step()
        with torch.inference_mode():
            return m(x).clone(), torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()

    ya, pa = run(True)
    yb, pb = run(False)
    assert torch.equal(pa, pb) and torch.equal(ya, yb)


def test_a_used_module_can_be_deep_copied_and_pickled(tmp_path):
    """The reference's generators are plain torch modules: copy.deepcopy (EMA / SWA copies), torch.save(module) and spawn-style launchers
    work on them at any time.  Here a module that has run holds an engine handle (a pointer of this process) and parameters that are views
    of one flat buffer: the copy must come out as an independent, working module with the same parameters -- its own engine, its own
    buffer -- and training the copy must not move the original."""
    import copy
    import pickle
    state = gc.make_state("sr", 32, 1, 751)
    m = build_module("sr", 1, 1, state)
    x = torch.from_numpy(gc.make_input((2, 1, 24, 40), 752)).cuda()
    t = torch.from_numpy(gc.make_input((2, 1, 48, 80), 753)).cuda()
    with torch.no_grad():
        y = m(x)
    c = copy.deepcopy(m)
    u = pickle.loads(pickle.dumps(m))
    torch.save(m, tmp_path / "whole_module.pt")
    for other in (c, u):
        assert other._engine is None
        with torch.no_grad():
            assert torch.equal(other(x), y)
        assert other._engine is not m._engine and other._flat.data_ptr() != m._flat.data_ptr()
    before = m._flat.clone()
    opt = torch.optim.Adam(c.parameters(), lr=1e-3)
    torch.nn.functional.l1_loss(c(x), t).backward()
    opt.step()
    assert torch.equal(m._flat, before) and not torch.equal(c._flat, before)
    with torch.no_grad():
        assert torch.equal(m(x), y) and not torch.equal(c(x), y)
    # the composed loss holds a handle too
    from xmm_superres_denoise.utils.loss_functions import Loss
    loss = Loss({"l1": 0.7, "psnr": 0.3})
    l2 = copy.deepcopy(loss)
    l3 = pickle.loads(pickle.dumps(loss))
    p = y.clone().requires_grad_(True)
    v = loss(p, t)
    assert torch.equal(l2(p, t), v) and torch.equal(l3(p, t), v) and l2.h.value != loss.h.value


def test_non_contiguous_tensors_and_side_streams_like_any_torch_module():
    """torch modules take strided views and run on whatever stream is current.  A transposed input, a transposed incoming gradient and a
    forward + backward issued on a side stream (the engine launches on torch's current stream: engine/_lib.py) give bit for bit what
    contiguous tensors on the default stream give."""
    state = gc.make_state("dn", 32, 1, 791)
    m = build_module("dn", 1, 1, state).set_math("bf16x6")
    base = torch.from_numpy(gc.make_input((2, 1, 40, 40), 792)).cuda()
    dyb = torch.from_numpy(gc.make_input((2, 1, 40, 40), 793) - 0.5).cuda()

    def run(x, dy):
        x = x.detach().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(x)
        y.backward(dy)
        return y.detach().clone(), x.grad.clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    want = run(base.transpose(2, 3).contiguous(), dyb.transpose(2, 3).contiguous())
    xt, dyt = base.transpose(2, 3), dyb.transpose(2, 3)
    assert not xt.is_contiguous() and not dyt.is_contiguous()
    got = run(xt, dyt)
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = run(xt, dyt)
    side.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(want, got))


def test_recomputed_backward_refuses_parameters_updated_since_the_forward():
    """torch refuses a backward whose saved tensors were modified in place since the forward.  Here that matters where the forward has to
    be RE-RUN in the backward (memory_efficient; a context displaced by a later forward): re-running it with other weights would give a
    gradient of a different function, silently.  An optimizer step updates the parameters, not the flat buffer they are views of -- and
    their version counters are their own -- so the guard has to watch both."""
    from xmm_superres_denoise.engine import XsdError
    from xmm_superres_denoise.models import GeneratorRRDB_DN
    state = gc.make_state("dn", 32, 1, 801)
    m = GeneratorRRDB_DN(1, 1, 32, 1, memory_efficient=True)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    m.cuda()
    x = torch.from_numpy(gc.make_input((2, 1, 24, 40), 802)).cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    loss = m(x).sum()
    loss.backward()                                   # untouched parameters: fine
    opt.step()
    loss = m(x).sum()
    opt.step()                                        # parameters change between this forward and its (recomputing) backward
    with pytest.raises(XsdError, match="modified in place"):
        loss.backward()
    loss = m(x).sum()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    with pytest.raises(XsdError, match="modified in place"):
        loss.backward()
    m(x).sum().backward()                             # and the module is fine afterwards


def test_backward_rejects_mismatched_dy_and_stale_generation():
    from xmm_superres_denoise.engine import XsdError
    state = gc.make_state("dn", 32, 1, 41)
    m = build_module("dn", 1, 1, state)
    eng = m._get_engine(torch.device("cuda", 0))
    eng.pack(m.flat_parameters())
    x = torch.rand(2, 1, 24, 40, device="cuda")
    y = eng.forward(x, save_for_backward=True)
    gen = eng.generation
    grads = torch.empty_like(m.flat_parameters())
    with pytest.raises(XsdError, match="dy has shape"):
        eng.backward(torch.zeros(1, 1, 24, 40, device="cuda"), grads)
    eng.forward(torch.rand(1, 1, 16, 32, device="cuda"), save_for_backward=True)
    with pytest.raises(XsdError, match="later forward replaced"):
        eng.backward(torch.zeros_like(y), grads, generation=gen)
    eng.forward(x, save_for_backward=False)
    with pytest.raises(XsdError, match="preceding forward"):
        eng.backward(torch.zeros_like(y), grads)


@pytest.mark.parametrize("kind,nf,in_ch,out_ch,blocks,nup,shape", [
    ("dn", 64, 1, 1, 1, 1, (2, 20, 24)),     # the dense block's own default width (rrdb_blocks.py:23)
    ("sr", 16, 3, 2, 2, 1, (2, 19, 21)),     # RGB in, two channels out, ragged against the 16 x 16 tile
    ("dn", 12, 2, 2, 1, 1, (1, 33, 17)),     # width that is no multiple of 8
    ("dn", 8, 1, 3, 1, 1, (2, 16, 16)),      # `out + x` with x broadcast over the output channels (generator_rrdb.py:134)
    ("sr", 8, 1, 1, 1, 2, (1, 9, 11)),       # two pixel-shuffle stages
    ("dn", 64, 3, 3, 2, 1, (2, 70, 133)),    # several tiles per image, 320-channel dense convs, RGB with the skip
    ("sr", 48, 1, 1, 1, 1, (1, 21, 45)),     # matrix-instruction path with partial 32-channel blocks (48 .. 240 inputs, 192-channel shuffle conv), ragged against its 8 x 32 tile
    ("sr", 64, 1, 1, 1, 1, (2, 19, 37)),     # 64 filters, one image channel: the plane kernels with two planes per tensor (Builder::build_multi), shuffle conv 64 -> 256
    ("sr", 64, 1, 1, 2, 2, (1, 9, 11)),      # ... two pixel-shuffle stages, two blocks
    ("dn", 96, 1, 1, 1, 1, (1, 17, 40)),     # three planes per tensor: K-loops of up to 15 planes in launches of five
    ("sr", 32, 3, 2, 1, 1, (2, 18, 35)),     # the shipped width with RGB in / two channels out: plane kernels, image-side layers per image channel
    ("dn", 32, 2, 2, 2, 1, (1, 33, 20)),     # ... DN with its skip per channel
    ("dn", 64, 4, 4, 1, 1, (1, 12, 34)),
    ("dn", 32, 1, 3, 1, 1, (2, 17, 33)),     # ... and with a one-channel x broadcast over three output channels (generator_rrdb.py:134)
    # what the plane kernels do not take stays on the exact-fp32 kernels of csrc/generic_net.hip:
    ("dn", 8, 9, 9, 1, 1, (1, 19, 21)),      # more than 8 image channels: direct convolutions throughout
    ("sr", 272, 1, 1, 1, 1, (1, 6, 9)),      # more than 256 filters: the fp32 matrix instruction with partial 32-channel blocks, 272 -> 1088 shuffle conv
    ("dn", 20, 10, 10, 1, 1, (1, 17, 33)),   # ... both kinds of conv in one net (20-filter trunk on the matrix instruction, 10-channel image side direct)
])
def test_generic_widths_vs_float64_restatement(kind, nf, in_ch, out_ch, blocks, nup, shape):
    """Widths other than the shipped 32 / 1 / 1 (reference constructors take any: generator_rrdb.py:10-54).  Up to 256 filters
    and 8 image channels run on the plane kernels (widths that are no multiple of 32 zero-padded: the 8-, 12-, 16-, 48-filter
    cases; wider nets with several planes per tensor); beyond that the exact-fp32 kernels of csrc/generic_net.hip (convs with
    >= 16 channels on both sides on the fp32 matrix instruction, narrower ones as direct convolutions).  Forward, dL/dx and every parameter gradient against a float64 evaluation of the reference graph (oracle.torch_forward; the C oracle handles one image channel only), through
    the nn.Module API, L1 loss."""
    _widths_vs_float64(kind, nf, in_ch, out_ch, blocks, nup, shape)


def _widths_vs_float64(kind, nf, in_ch, out_ch, blocks, nup, shape, math=None):
    from collections import OrderedDict
    rng = np.random.default_rng(4242 + nf + in_ch)
    shapes = gc.rrdb_param_shapes(kind, nf, blocks, in_ch=in_ch, out_ch=out_ch, num_upsample=nup)
    state, fan = OrderedDict(), 1
    for n, shp in shapes.items():
        if n.endswith(".weight"):
            fan = shp[1] * 9
        state[n] = rng.uniform(-1, 1, size=shp).astype(np.float32) / np.sqrt(fan)
    if kind == "sr":
        state["conv_last.bias"][:] = 0.4
    B, H, W = shape
    s = 2 ** nup if kind == "sr" else 1
    x = gc.make_input((B, in_ch, H, W), 11)
    t = gc.make_input((B, out_ch, H * s, W * s), 12)
    st64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in state.items()}
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    y64 = oracle.torch_forward(kind, nf, blocks, st64, x64, num_upsample=nup)
    l64 = torch.nn.functional.l1_loss(y64, torch.from_numpy(t).double())
    l64.backward()
    m = build_module(kind, blocks, nup, state, nf=nf, in_ch=in_ch, out_ch=out_ch)
    if math:
        m.set_math(math)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    y = m(xd)
    assert y.shape == tuple(y64.shape)
    assert np.abs(y.detach().cpu().numpy() - y64.detach().numpy()).max() < 2e-5
    loss = torch.nn.functional.l1_loss(y, torch.from_numpy(t).cuda())
    assert abs(loss.item() - l64.item()) < 1e-5
    loss.backward()
    # flip-aware, like the shipped width (round 3 allowed 30 % of a tensor's rows to exceed the tight bar here): only the rows a
    # float64 evaluation names as flip candidates (a pre-activation within 4e-6 rms of zero in that conv's output channel)
    # may exceed 2e-4 of the tensor's largest entry; everything else must explain itself
    cand, n_out = flip_candidates(kind, blocks, state, x, t, nup)
    mark = len(FLIP_LOG)
    strict = _ripple_bar(x)
    assert_grad_close(xd.grad.cpu().numpy().reshape(-1, W), x64.grad.numpy().reshape(-1, W), "dx", tight=4e-4, loose=5e-2, candidates=cand, n_out_candidates=n_out, strict=strict)
    for n, p in m.named_parameters():
        assert_grad_close(p.grad.cpu().numpy(), st64[n].grad.numpy(), n, tight=2e-4, loose=5e-2, candidates=cand, n_out_candidates=n_out, strict=strict)
    _report_flips(f"{kind} nf={nf} {in_ch}->{out_ch} nup={nup} {math or 'default'}", mark)
    # and without autograd (no activations kept: two slabs ping-pong) the same output, bit for bit
    with torch.no_grad():
        assert torch.equal(m(xd.detach()), y.detach())


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("kind,nf,blocks,nup,shape", [("dn", 64, 1, 1, (2, 24, 40)), ("sr", 128, 1, 1, (1, 12, 33)), ("dn", 160, 1, 1, (1, 10, 21)), ("sr", 256, 1, 1, (1, 7, 9))])
def test_wide_plane_nets_in_every_math_mode(kind, nf, blocks, nup, shape, math):
    """64, 96, ... 256 filters with one image channel run on the 32-filter configuration's own kernels, a feature tensor being
    2 - 8 planes of 32 channels and a conv's K-loop cut into launches of <= 5 planes that accumulate (csrc/xsd_engine.hip,
    Builder::build_multi; the dense block's default width is 64, rrdb_blocks.py:23).  Forward, dL/dx and every parameter
    gradient against float64 in each math mode (the 256-filter SR case: eight planes, a 256 -> 1024 shuffle conv, K-loops of
    up to 40 planes)."""
    _widths_vs_float64(kind, nf, 1, 1, blocks, nup, shape, math=math)


@pytest.mark.parametrize("math", ["f16x3", "bf16x6"])
def test_block_weight_gradient_launch_equals_one_launch_per_g(tmp_path, math):
    """The split-precision weight gradients of a dense block run as ONE pair-list launch over its 15 (X, G) pairs;
    XSD_WGRAD_BLOCK=0 restores one launch per G.  Same products, same per-workgroup tile order inside a pair, a different cut
    of the tiles into partial sums (16 chunks instead of 256 / pairs): the two gradients agree to fp32 summation-order level.
    The switch is read once per process, so each variant runs in a child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        "root = sys.argv[1]\n"
        "for p in (root, os.path.join(root, 'xmm-superres-denoise_amd'), os.path.join(root, 'tests'), os.path.join(root, 'tests', 'golden')): sys.path.insert(0, p)\n"
        "from xmm_superres_denoise.models import GeneratorRRDB_DN\n"
        "torch.manual_seed(5)\n"
        "m = GeneratorRRDB_DN(1, 1, 32, 2).cuda().set_math(sys.argv[3])\n"
        "x = torch.rand(3, 1, 72, 100, device='cuda', requires_grad=True); t = torch.rand(3, 1, 72, 100, device='cuda')\n"
        "y = m(x); loss = (y - t).abs().mean(); loss.backward()\n"
        "g = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}\n"
        "g['dx'] = x.grad.detach().cpu().numpy()\n"
        "np.savez(sys.argv[2], **g)\n")
    # variants (include/xsd.h): the default (MI355X: 16 parts + a 17th on the spare CUs), one launch per G, no tail part, and the
    # plans of devices with other CU counts (XSD_TEST_NCU: 128 CUs -> 8 parts, no tail; 140 -> 8 parts + tail; 100 -> fewer than
    # one part per XCD: falls back to one launch per G).  Every cut of the tiles into partial sums must give the same gradient.
    # Round 6: only the test-hooks variant of the library reads these variables (make -C csrc hooks -> lib/libxsd_hip_hooks.so, selected
    # with XSD_LIB; include/xsd.h); the product library ignores them -- "product" runs it with every switch set and must equal "default".
    hooks = os.path.join(root, "xmm-superres-denoise_amd", "lib", "libxsd_hip_hooks.so")
    assert os.path.exists(hooks), "build the hooks variant: make -C xmm-superres-denoise_amd/csrc hooks (__graft_entry__.build() does)"
    variants = {"default": {"XSD_LIB": hooks}, "per_g": {"XSD_LIB": hooks, "XSD_WGRAD_BLOCK": "0"}, "no_tail": {"XSD_LIB": hooks, "XSD_WGRAD_TAIL": "0"},
                "ncu128": {"XSD_LIB": hooks, "XSD_TEST_NCU": "128"}, "ncu140": {"XSD_LIB": hooks, "XSD_TEST_NCU": "140"}, "ncu100": {"XSD_LIB": hooks, "XSD_TEST_NCU": "100"},
                "product": {"XSD_WGRAD_BLOCK": "0", "XSD_WGRAD_TAIL": "0", "XSD_TEST_NCU": "100"}}
    outs = {}
    for name, extra in variants.items():
        out = str(tmp_path / f"g_{name}.npz")
        env = {k: v for k, v in os.environ.items() if k not in ("XSD_WGRAD_BLOCK", "XSD_WGRAD_TAIL", "XSD_TEST_NCU", "XSD_LIB")}
        env.update(extra)
        p = subprocess.run([sys.executable, "-c", code, root, out, math], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, (name, p.stdout.decode(errors="replace")[-3000:])
        outs[name] = np.load(out)
    b = outs["per_g"]
    for name, a in outs.items():
        assert set(a.files) == set(b.files)
        for k in a.files:
            ref = np.abs(b[k]).max() + 1e-30
            assert np.abs(a[k] - b[k]).max() / ref < 2e-6, (name, k, float(np.abs(a[k] - b[k]).max() / ref))
        assert np.array_equal(a["dx"], b["dx"])      # the input gradient does not depend on the weight-gradient launches at all
    # a device with fewer than 120 CUs takes the per-G path: bit-equal to XSD_WGRAD_BLOCK=0; the differently cut launches are not
    for k in b.files:
        assert np.array_equal(outs["ncu100"][k], b[k]), k
    assert any(not np.array_equal(outs["default"][k], b[k]) for k in b.files)
    assert any(not np.array_equal(outs["no_tail"][k], outs["default"][k]) for k in b.files)
    assert any(not np.array_equal(outs["ncu128"][k], outs["default"][k]) for k in b.files)
    for k in b.files:      # the product library does not replan on an environment variable
        assert np.array_equal(outs["product"][k], outs["default"][k]), k


def test_max_abs_slot_array_grows_without_changing_results(tmp_path):
    """f16x3 scales every operand tensor from a max-|x| slot; the slot array grows (synchronize, reallocate, copy the packed
    panels' two slots, rebuild the plan) when the sizing pass counts more slots than are allocated -- 65,536 by default, i.e.
    only for nets like 256 filters x 64 blocks.  XSD_TEST_AMAX_CAP=128 (include/xsd.h) makes the 4-block shipped net take that
    path, twice over (forward-only plan, then the training plan): outputs and gradients bit-equal to the default capacity."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        "root = sys.argv[1]\n"
        "for p in (root, os.path.join(root, 'xmm-superres-denoise_amd'), os.path.join(root, 'tests'), os.path.join(root, 'tests', 'golden')): sys.path.insert(0, p)\n"
        "from xmm_superres_denoise.models import GeneratorRRDB_DN\n"
        "torch.manual_seed(9)\n"
        "m = GeneratorRRDB_DN(1, 1, 32, 4).cuda().set_math('f16x3')\n"
        "x = torch.rand(2, 1, 40, 72, device='cuda'); t = torch.rand(2, 1, 40, 72, device='cuda')\n"
        "with torch.no_grad(): y0 = m(x).cpu().numpy()\n"
        "xg = x.clone().requires_grad_(True)\n"
        "y = m(xg); loss = (y - t).abs().mean(); loss.backward()\n"
        "g = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}\n"
        "g['dx'] = xg.grad.detach().cpu().numpy(); g['y0'] = y0; g['y'] = y.detach().cpu().numpy()\n"
        "np.savez(sys.argv[2], **g)\n")
    hooks = os.path.join(root, "xmm-superres-denoise_amd", "lib", "libxsd_hip_hooks.so")      # the only build that reads the variable (round 6)
    assert os.path.exists(hooks), "build the hooks variant: make -C xmm-superres-denoise_amd/csrc hooks"
    outs = []
    for cap in (None, "128"):
        out = str(tmp_path / f"cap_{cap}.npz")
        env = {k: v for k, v in os.environ.items() if k not in ("XSD_TEST_AMAX_CAP", "XSD_LIB")}
        if cap:
            env["XSD_TEST_AMAX_CAP"] = cap
            env["XSD_LIB"] = hooks
        p = subprocess.run([sys.executable, "-c", code, root, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
        outs.append(np.load(out))
    a, b = outs
    assert np.abs(a["y"]).max() > 0 and np.abs(a["conv_first.weight"]).max() > 0
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_second_backward_after_one_forward_starts_from_clean_max_slots():
    """Two backwards after ONE forward with different dy (include/xsd.h: allowed).  The backward launches publish their planes'
    max |x| by atomic max into slots that stage 0 re-zeroes: the second backward's gradients are bit-equal to those of a fresh
    forward + backward with that dy -- also when the first dy was 2^20 times larger (stale maxima would have cost f16x3
    twenty operand bits, not an overflow)."""
    state = gc.make_state("dn", 32, 2, 31)
    m = build_module("dn", 2, 1, state).set_math("f16x3")
    eng = m._get_engine(torch.device("cuda", 0))
    flat = m.flat_parameters()
    x = torch.from_numpy(gc.make_input((2, 1, 40, 56), 32)).cuda()
    dy_small = torch.from_numpy(gc.make_input((2, 1, 40, 56), 33) - 0.5).cuda().contiguous()
    dy_big = (dy_small * 2.0 ** 20).contiguous()
    eng.pack(flat)
    g_ref = torch.empty_like(flat)
    eng.forward(x, save_for_backward=True)
    dx_ref = eng.backward(dy_small, g_ref, need_dx=True)
    g1, g2 = torch.empty_like(flat), torch.empty_like(flat)
    eng.forward(x, save_for_backward=True)
    eng.backward(dy_big, g1, need_dx=True)
    dx2 = eng.backward(dy_small, g2, need_dx=True)
    assert torch.equal(g2, g_ref) and torch.equal(dx2, dx_ref)
    assert float(g1.abs().max()) > 1e3 * float(g_ref.abs().max())
