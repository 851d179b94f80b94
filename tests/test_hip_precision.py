"""How close to fp32 are the two split math modes?  "bf16x6" (strict: exact 3-term bf16 split, six products, MFMA accumulation)
and "f16x3" (default and benchmark headline: two-term fp16 split of power-of-two-scaled operands, 22-23 significant bits per
operand, three products) are measured against a float64 evaluation of the same network, next to the two things that define
"the reference's precision": torch's own fp32 path on the CPU (the reference's nn.Conv2d arithmetic, rrdb_blocks.py:27-54)
and this engine's exact-fp32 MFMA mode (bitwise an fp32 fma chain).

What the asserts enforce (and include/xsd.h, bench.py's `dtype` label and DESIGN.md section 4 claim no more than this):
  single layers (no non-linearity: the arithmetic alone), K = 1440: forward both split modes <= the exact-fp32 mode and <= 2 x
           torch fp32 (whose CPU kernel keeps 16 partial sums per output); input-gradient <= torch fp32 and <= the exact-fp32
           mode; weight-gradient <= torch fp32 (measured 3.5-4.3 x better) and <= 2 x the exact-fp32 mode.
  whole networks, forward (goldens, 512 x 512 x 4 blocks): both split modes <= torch fp32 and <= the exact-fp32 mode
           (measured 0.55-0.67 x).
  whole networks, backward (4 seeds at 256 x 256, 512 x 512 with batch 2; every parameter-gradient tensor and dL/dx): the four
           fp32 paths -- torch fp32, the exact-fp32 mode, bf16x6, f16x3 -- differ by which LeakyReLU' / clamp decisions
           fall the other way, not by their arithmetic: over the seeds every ordering of the four occurs (e.g. seed 7301:
           f16x3 < torch < fp32 < bf16x6; seed 7101: bf16x6 < torch < f16x3 < fp32), the exact-fp32 mode itself is up to 2.3 x
           torch.  So the bar that IS a fact: both split modes within 2 x of torch fp32 AND within 2 x of the exact-fp32 mode on
           the worst seed and size, for the flat-gradient rms, the dL/dx rms and the worst per-tensor rms (measured: f16x3
           <= 1.57 x torch, <= 1.73 x fp32 mode; bf16x6 <= 1.36 x, <= 1.27 x).  Round 2's "bf16x6 below both yard-sticks
           everywhere" and "f16x3 never above the fp32 mode" were one-seed statements and do not survive four seeds.
  every mode: per-tensor max error of every parameter gradient and of dL/dx at 512 x 512, batch 2, below 2e-4 of the
           tensor's largest entry on every row without a LeakyReLU flip candidate (util_hip.assert_grad_close).
Run with `pytest -s` to get the tables; the log of the round is committed under profiles/."""
import os

import numpy as np
import pytest
import torch

SPLITS = ("bf16x6", "f16x3")   # the two split modes

import gen_common as gc
from oracle import oracle
from util_hip import FLIP_LOG, assert_grad_close, build_module, flip_candidates, load_case, nchw_to_planes, planes_to_nchw, ptr_array

pytestmark = pytest.mark.gpu


def _rms(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.sqrt(np.mean((a - ref) ** 2)) / (np.sqrt(np.mean(ref ** 2)) + 1e-300))


def _state_t(state, dtype):
    return {k: torch.from_numpy(v).to(dtype) for k, v in state.items()}


def test_mfma_accumulation_is_single_rounding():
    """The property the mode rests on (tools/mfma_probe.hip measures the instruction itself): a K = 1440 reduction whose
    exact value needs more than 24 bits comes out closer to float64 from the bf16x6 conv than from an fp32 fma chain."""
    from xmm_superres_denoise.engine import Engine
    from xmm_superres_denoise.engine._lib import check
    rng = np.random.default_rng(5)
    B, H, W, n_in = 1, 32, 64, 5
    x = rng.normal(size=(B, 32 * n_in, H, W)).astype(np.float32)
    w = (rng.normal(size=(32, 32 * n_in, 3, 3)) / np.sqrt(288 * n_in)).astype(np.float32)
    b = rng.normal(size=(32,)).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1).numpy()
    t32 = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1).numpy()
    errs = {"torch_fp32": _rms(t32, ref)}
    xin = nchw_to_planes(x)
    wd, bd = torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()
    for math in ("fp32", "bf16x6", "f16x3"):
        e = Engine("dn", 1, 1, 32, 1)
        e.set_math(math)
        out = [torch.full((B, H, W, 32), float("nan"), device="cuda")]
        check(e.L.xsd_test_conv3x3(e.h, ptr_array(xin), n_in, wd.data_ptr(), bd.data_ptr(), ptr_array(out), 1, 1.0, B, H, W, None))
        errs[math] = _rms(planes_to_nchw(out), ref)
    print("single conv K=1440 rms error vs float64:", errs)
    # fp32 fma chain of 1440 terms (this engine's exact-fp32 MFMA mode, bitwise an fmaf chain): ~8e-7; bf16x6: ~3e-7.
    # torch's CPU kernel (oneDNN) keeps 16 SIMD partial sums per output and lands at ~1.6e-7 on this worst-case layer; on
    # whole networks, where the fp32 rounding of the stored activations dominates, bf16x6 is below it (tests below).
    for m in SPLITS:
        assert errs[m] <= 0.5 * errs["fp32"], m
        assert errs[m] <= 2.0 * errs["torch_fp32"], m


def test_single_layer_backward_error_vs_float64():
    """The arithmetic of the input-gradient and weight-gradient kernels alone (no LeakyReLU, no clamp: nothing can flip):
    K = 1440 channels x taps for dX, 2 x 64 x 96 pixels for dW, against float64, next to torch's fp32 autograd."""
    from xmm_superres_denoise.engine import Engine
    from xmm_superres_denoise.engine._lib import check
    rng = np.random.default_rng(6)
    B, H, W, n_in = 2, 64, 96, 5
    x = rng.normal(size=(B, 32 * n_in, H, W)).astype(np.float32)
    w = (rng.normal(size=(32, 32 * n_in, 3, 3)) / np.sqrt(288 * n_in)).astype(np.float32)
    g = rng.normal(size=(B, 32, H, W)).astype(np.float32)

    def torch_bwd(dtype):
        xt, wt = torch.from_numpy(x).to(dtype).requires_grad_(True), torch.from_numpy(w).to(dtype).requires_grad_(True)
        torch.nn.functional.conv2d(xt, wt, None, padding=1).backward(torch.from_numpy(g).to(dtype))
        return xt.grad.numpy(), wt.grad.numpy()

    dx64, dw64 = torch_bwd(torch.float64)
    dx32, dw32 = torch_bwd(torch.float32)
    errs = {"torch_fp32": {"dx": _rms(dx32, dx64), "dw": _rms(dw32, dw64)}}
    xin, gp, wd = nchw_to_planes(x), nchw_to_planes(g)[0], torch.from_numpy(w).cuda()
    for math in ("fp32", "bf16x6", "f16x3"):
        e = Engine("dn", 1, 1, 32, 1)
        e.set_math(math)
        dxs = [torch.full((B, H, W, 32), float("nan"), device="cuda") for _ in range(n_in)]
        dw, db = torch.full_like(wd, float("nan")), torch.full((32,), float("nan"), device="cuda")
        check(e.L.xsd_test_conv3x3_bwd(e.h, ptr_array(xin), n_in, wd.data_ptr(), gp.data_ptr(), ptr_array(dxs), dw.data_ptr(), db.data_ptr(), B, H, W, None))
        errs[math] = {"dx": _rms(planes_to_nchw(dxs), dx64), "dw": _rms(dw.cpu().numpy(), dw64)}
    print("single conv backward (dX: K = 288; dW: 12,288 pixels), rms error vs float64:", errs)
    # measured: dX -- torch 2.2e-7, exact-fp32 mode 3.0e-7, bf16x6 0.9e-7, f16x3 1.4e-7; dW -- torch 5.7e-7, exact-fp32 mode 1.2e-7
    # (its per-tile fma chains are short and the cross-tile sum is done in double, like the split modes'), bf16x6 1.7e-7, f16x3 1.3e-7
    for m in SPLITS:
        assert errs[m]["dx"] <= errs["fp32"]["dx"] and errs[m]["dx"] <= errs["torch_fp32"]["dx"], m
        assert errs[m]["dw"] <= errs["torch_fp32"]["dw"] and errs[m]["dw"] <= 2.0 * errs["fp32"]["dw"], m


@pytest.mark.parametrize("name,kind", [("dn_nf32_b4_32x32", "dn"), ("sr_nf32_b4_24x40", "sr")])
def test_golden_cases_error_vs_float64(name, kind):
    z, nf, blocks, nup, state, x, t = load_case(name, kind)
    y64 = oracle.torch_forward(kind, 32, blocks, _state_t(state, torch.float64), torch.from_numpy(x).double(), num_upsample=nup).numpy()
    y32 = oracle.torch_forward(kind, 32, blocks, _state_t(state, torch.float32), torch.from_numpy(x), num_upsample=nup).numpy()
    errs = {"torch_fp32": _rms(y32, y64), "golden(reference fp32)": _rms(z["y"], y64)}
    for math in ("fp32", "bf16x6", "f16x3"):
        m = build_module(kind, blocks, nup, state).set_math(math)
        with torch.no_grad():
            errs[math] = _rms(m(torch.from_numpy(x).cuda()).cpu().numpy(), y64)
    print(name, "output rms error vs float64:", errs)
    for m in SPLITS:
        assert errs[m] <= errs["torch_fp32"], m
        assert errs[m] <= errs["golden(reference fp32)"], m
        assert errs[m] <= errs["fp32"], m


def test_f16x3_survives_extreme_ranges():
    """fp16 has five exponent bits; mode f16x3 relies on the per-tensor power-of-two scales (plane / weight max |x|).  A single
    K = 1440 conv must keep its relative accuracy when the activations and the weights sit far outside the fp16 range, and
    when a few outliers are 1e5 times larger than everything else (the scale is set by the outliers; the rest must not
    lose its low term to the fp16 subnormals -- that is what storing l * 2^11 is for)."""
    from xmm_superres_denoise.engine import Engine
    from xmm_superres_denoise.engine._lib import check
    rng = np.random.default_rng(11)
    B, H, W, n_in = 1, 32, 64, 5
    base_x = rng.normal(size=(B, 32 * n_in, H, W)).astype(np.float32)
    base_w = (rng.normal(size=(32, 32 * n_in, 3, 3)) / np.sqrt(288 * n_in)).astype(np.float32)
    heavy = base_x.copy()
    heavy.reshape(-1)[rng.integers(0, heavy.size, 64)] *= 1e5
    cases = {"unit": (base_x, base_w), "x * 2^40, w * 2^-30": (base_x * np.float32(2.0 ** 40), base_w * np.float32(2.0 ** -30)),
             "x * 2^-60, w * 2^20": (base_x * np.float32(2.0 ** -60), base_w * np.float32(2.0 ** 20)), "64 outliers x 1e5": (heavy, base_w)}
    errs = {}
    for name, (x, w) in cases.items():
        ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, padding=1).numpy()
        t32 = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, padding=1).numpy()
        e = Engine("dn", 1, 1, 32, 1)
        e.set_math("f16x3")
        out = [torch.full((B, H, W, 32), float("nan"), device="cuda")]
        xin, wd, zero_b = nchw_to_planes(x), torch.from_numpy(w).cuda(), torch.zeros(32, device="cuda")   # (kept alive over the call)
        check(e.L.xsd_test_conv3x3(e.h, ptr_array(xin), n_in, wd.data_ptr(), zero_b.data_ptr(), ptr_array(out), 1, 1.0, B, H, W, None))
        got = planes_to_nchw(out)
        assert np.isfinite(got).all(), name
        if name == "64 outliers x 1e5":   # judged where the outliers do not reach (elsewhere their own rounding dominates any path)
            quiet = np.abs(ref) < 10 * np.sqrt(np.mean(base_x.astype(np.float64) ** 2)) * 3
            errs[name] = (_rms(got[quiet], ref[quiet]), _rms(t32[quiet], ref[quiet]))
        else:
            errs[name] = (_rms(got, ref), _rms(t32, ref))
    print("f16x3 single conv K=1440, rms error vs float64 (engine, torch fp32):", errs)
    # measured: 2.5e-7 on unit-scale data and EXACTLY the same after the power-of-two rescalings (the scales are exact);
    # with the outliers 2.1e-6 against torch fp32's 2.5e-6 on the same data (their rounding reaches every output they touch)
    for name, (mine, torch32) in errs.items():
        if "2^" in name:
            assert mine <= 1.01 * errs["unit"][0], name     # the same relative accuracy as on unit-scale data
        assert mine <= 1.5 * torch32, name                  # and in torch fp32's class on the same data


MODES = ("fp32", "bf16x6", "f16x3")


def _per_tensor(flat, ref, shapes):
    """(worst per-tensor relative rms, worst per-tensor max |err| / max |ref|, name of that tensor) over the parameter tensors"""
    off, worst_rms, worst_max, who = 0, 0.0, 0.0, ""
    for name, shp in shapes.items():
        k = int(np.prod(shp))
        a, r = np.asarray(flat[off:off + k], np.float64), np.asarray(ref[off:off + k], np.float64)
        off += k
        worst_rms = max(worst_rms, _rms(a, r))
        mx = float(np.abs(a - r).max() / (np.abs(r).max() + 1e-300))
        if mx > worst_max:
            worst_max, who = mx, name
    return worst_rms, worst_max, who


def _net_errors(size, blocks, seed, with_grad, batch=1, flip_aware=False, kind="dn", tight_w=2e-4, strict_w=1e-3, with_torch32=True, dx_loose=5e-2,
                dx_flip_frac=0.1, dx_l2_bar=2.5e-2):
    """DN (or SR 2x) generator, `blocks` RRDB blocks, `batch` tiles of size x size, seeded weights and input; gradient of the linear
    functional <dy, y> (no loss discontinuity).  Returns {mode: {y, g, dx, t_rms, t_max}} relative to float64 torch, with
    torch's own fp32 path as one of the modes.  flip_aware: also hold every mode's every gradient tensor and dL/dx to the
    row-wise tolerance of the golden tests (2e-4 / 4e-4 of the tensor's largest entry off the flip-candidate rows)."""
    sc = 2 if kind == "sr" else 1
    state = gc.make_state(kind, 32, blocks, seed, gain=1.0, last_bias=0.3 if kind == "sr" else None)
    x = gc.make_input((batch, 1, size, size), seed + 1)
    dy = (gc.make_input((batch, 1, size * sc, size * sc), seed + 2) - 0.5).astype(np.float32) / (batch * size * size * sc * sc)
    shapes = gc.rrdb_param_shapes(kind, 32, blocks)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    cand = None
    if flip_aware:
        # LeakyReLU candidates per conv (rows that may deviate), and the output pixels whose pre-clamp value is within 3e-5 of a
        # clamp bound: their dy is set to zero for EVERY path, so that a clamp decision falling the other way (a 100 % change of
        # that pixel's contribution, localised in dL/dx) cannot occur -- the row-wise check then has only LeakyReLU' flips to allow
        cand, near = flip_candidates(kind, blocks, state, x, np.zeros_like(dy), 1, return_clamp_mask=True, device="cuda")
        torch.cuda.empty_cache()
        dy = np.where(near, np.float32(0), dy)

    def torch_path(dtype, device):
        # two tiles at a time: <dy, y> is a sum over tiles, so the parameter gradients accumulate over the chunks (autograd keeps
        # ~20 GB of float64 activations per 512 x 512 tile)
        st = {k: v.to(device).requires_grad_(with_grad) for k, v in _state_t(state, dtype).items()}
        ys, dxs = [], []
        for i in range(0, batch, 2):
            xt = torch.from_numpy(x[i:i + 2]).to(dtype).to(device).requires_grad_(with_grad)
            y = oracle.torch_forward(kind, 32, blocks, st, xt)
            if with_grad:
                y.backward(torch.from_numpy(dy[i:i + 2]).to(dtype).to(device))
                dxs.append(xt.grad.cpu().numpy())
            ys.append(y.detach().cpu().numpy())
            del y, xt
        g = dx = None
        if with_grad:
            g = torch.cat([v.grad.reshape(-1) for v in st.values()]).cpu().numpy()
            dx = np.concatenate(dxs)
        return np.concatenate(ys), g, dx

    # the float64 yard-stick is evaluated by torch on the GPU (im2col + dgemm: seconds instead of minutes on the host cores; any
    # float64 evaluation is exact to ~1e-15 here); the fp32 yard-stick is torch's CPU path, the reference's own arithmetic
    try:
        y64, g64, dx64 = torch_path(torch.float64, "cuda")
    except RuntimeError as err:      # no float64 convolution on this device build: fall back to the host cores
        print("float64 yard-stick on the CPU (GPU path failed: %s)" % str(err).splitlines()[0])
        y64, g64, dx64 = torch_path(torch.float64, "cpu")
    torch.cuda.empty_cache()
    # torch's fp32 path on the CPU = the reference's own arithmetic, the second yard-stick (skipped for the batch-8 case: eight
    # 512 x 512 train passes on the host cores are minutes, and that case is about the row-wise check against float64)
    y32, g32, dx32 = torch_path(torch.float32, "cpu") if with_torch32 else (None, None, None)
    inside = (y64 > 0) & (y64 < 1)          # the clamp hides errors where it saturates: compare where it is the identity

    def record(y, g, dx):
        rec = {"y": _rms(y[inside], y64[inside])}
        if with_grad:
            t_rms, t_max, who = _per_tensor(g, g64, shapes)
            rec.update(g=_rms(g, g64), dx=_rms(dx, dx64), t_rms=t_rms, t_max=t_max, t_max_at=who,
                       dx_max=float(np.abs(dx.astype(np.float64) - dx64).max() / np.abs(dx64).max()))
        return rec

    out = {"torch_fp32": record(y32, g32, dx32)} if with_torch32 else {}
    for math in MODES:
        m = build_module(kind, blocks, 1, state).set_math(math)
        eng = m._get_engine(torch.device("cuda", 0))
        eng.pack(m.flat_parameters())
        y = eng.forward(torch.from_numpy(x).cuda(), save_for_backward=with_grad)
        g = dx = None
        if with_grad:
            grads = torch.empty_like(m.flat_parameters())
            dx = eng.backward(torch.from_numpy(dy).cuda(), grads, need_dx=True).cpu().numpy()
            g = grads.cpu().numpy()
        out[math] = record(y.cpu().numpy(), g, dx)
        if flip_aware:
            mark = len(FLIP_LOG)
            # dL/dx: a LeakyReLU' decision that falls the other way in one of the first layers shows up as a blob of a few pixels
            # (its receptive field back to the input; measured 7 x 7 pixels at up to 3e-2 of max |dL/dx| for the SR net, at
            # different places in different modes, the exact-fp32 mode included: tools/dbg_sr_dx.py) -- no row list can name
            # them, so dL/dx is held to: at most 10 % of the image rows above 4e-4, nothing above 5e-2, relative L2 below 2.5e-2
            e_dx = np.abs(dx.astype(np.float64) - dx64).reshape(-1, size) / np.abs(dx64).max()
            print(f"    {math}: dL/dx rows over 4e-4: {100 * float((e_dx.max(axis=1) > 4e-4).mean()):.1f} %, worst pixel {e_dx.max():.2e}, "
                  f"relative L2 {np.linalg.norm(dx - dx64) / np.linalg.norm(dx64):.2e}", flush=True)
            assert_grad_close(dx.reshape(-1, size), dx64.reshape(-1, size), "dx", tight=4e-4, loose=dx_loose, max_flip_frac=dx_flip_frac, l2_bar=dx_l2_bar)
            off = 0
            for name, shp in shapes.items():
                k = int(np.prod(shp))
                assert_grad_close(g[off:off + k].reshape(shp), g64[off:off + k].reshape(shp), name, tight=tight_w, strict=strict_w, candidates=cand, n_out_candidates=0)
                off += k
            out[math]["rows_needing_flip_allowance"] = sum(r["rows_over_tight"] for r in FLIP_LOG[mark:])
        del m, eng
        torch.cuda.empty_cache()
    return out, float(inside.mean())


def _table(title, errs):
    keys = [k for k in ("y", "g", "dx", "t_rms", "t_max", "dx_max") if k in next(iter(errs.values()))]
    print(title)
    print("    %-11s" % "mode" + "".join("%11s" % k for k in keys) + "   worst tensor")
    for mode, rec in errs.items():
        print("    %-11s" % mode + "".join("%11.3e" % rec[k] for k in keys) + "   " + str(rec.get("t_max_at", "")) +
              ("   (rows needing the flip allowance: %d)" % rec["rows_needing_flip_allowance"] if "rows_needing_flip_allowance" in rec else ""))


def test_full_size_forward_error_vs_float64():
    """BASELINE tile size, BASELINE depth: 1 x 512 x 512, 4 RRDB blocks (36 dense blocks, K up to 1440 per conv)."""
    errs, frac = _net_errors(512, 4, 7001, with_grad=False)
    _table(f"512^2 x 4 blocks, forward rms error vs float64 ({100 * frac:.0f}% of pixels unclamped):", errs)
    for m in SPLITS:
        assert errs[m]["y"] <= errs["torch_fp32"]["y"], m
        assert errs[m]["y"] <= errs["fp32"]["y"], m


# (size, batch, seed, flip-aware per-tensor check): four seeds at 256 x 256, one at the BASELINE tile size with batch 2
# (a second 512 x 512 seed, 7601, was measured once: profiles/r03_precision_first.log)
# and (round 4) one at batch 8: 4096 tiles over the 256 persistent workgroups, i.e. every workgroup walks 16 tiles through several
# batch slices, as at the bench batch of 32 -- every gradient tensor of every mode row-wise against float64 there too
# -- and the bench batch itself (32 tiles, BASELINE configs[2]: 64 tiles per persistent workgroup), the same row-wise check; the
# float64 yard-stick and the flip-candidate pass run on the GPU (chunks of two tiles), so the big cases take seconds
# -- and (round 6) the REFERENCE's own operating points, 416 x 416 tiles (res/baseline_config.toml:36) at batch 1 (its default,
# :13: 338 tiles on 256 persistent workgroups -- 82 of them walk a second tile) and batch 4 (res/configs/runs/*.yaml:26: 1352 tiles,
# 5.28 per workgroup), where the launches are 4 - 60 half-steps long instead of the bench batch's 384
BACKWARD_CASES = [(256, 1, 7101, False), (256, 1, 7201, False), (256, 1, 7301, False), (256, 1, 7401, False),
                  (512, 2, 7501, True), (512, 8, 7701, True), (512, 32, 7801, True), (416, 1, 7911, True), (416, 4, 7921, True)]
_BWD_RESULTS = {}


@pytest.mark.parametrize("size,batch,seed,flip_aware", BACKWARD_CASES)
def test_backward_error_vs_float64(size, batch, seed, flip_aware):
    """4 blocks, gradient of the linear functional <dy, y>: every parameter gradient and dL/dx against float64 autograd, per
    case.  The 512 x 512, batch 2 case with flip_aware additionally checks EVERY gradient tensor row-wise (a wrong tile-edge
    term in one layer's dW would vanish in a flat rms over 1.67 M parameters; LeakyReLU' flips hit every fp32 path alike and
    are allowed only on the rows a float64 evaluation names).  The relative bars are asserted per case here and on the worst
    case over all cases in test_backward_worst_case_summary."""
    errs, frac = _net_errors(size, 4, seed, with_grad=True, batch=batch, flip_aware=flip_aware, with_torch32=batch <= 2)
    _table(f"{size}^2 x {batch} tile(s) x 4 blocks, seed {seed}: errors vs float64 ({100 * frac:.0f}% of pixels unclamped):", errs)
    if "torch_fp32" not in errs:      # the batch-8 case: every tensor of every mode was held row-wise to float64 inside _net_errors
        for mode in MODES:
            assert errs[mode]["t_max"] < 2e-2 and errs[mode]["g"] < 1e-4 and errs[mode]["dx"] < 1e-4, mode
        for m in SPLITS:
            assert errs[m]["y"] <= errs["fp32"]["y"], m
            for key in ("g", "dx", "t_rms"):
                assert errs[m][key] <= 2.0 * errs["fp32"][key], (m, key)
        return
    _BWD_RESULTS[(size, batch, seed)] = errs
    t32, f32 = errs["torch_fp32"], errs["fp32"]
    for m in SPLITS:
        assert errs[m]["y"] <= t32["y"] and errs[m]["y"] <= f32["y"], m       # forward: below both fp32 yard-sticks, every seed
        for key in ("g", "dx", "t_rms"):                                        # backward: flip noise decides the order (module docstring)
            assert errs[m][key] <= 2.0 * t32[key], (m, key)
            assert errs[m][key] <= 2.0 * f32[key], (m, key)
    for mode in MODES:                                        # nobody is anywhere near north_star's 1e-3
        assert errs[mode]["t_max"] < 2e-2 and errs[mode]["g"] < 1e-4 and errs[mode]["dx"] < 1e-4, mode


def test_backward_worst_case_summary():
    """The same bars on the WORST ratio over every seed and size that ran (what DESIGN.md section 4 quotes)."""
    if not _BWD_RESULTS:
        pytest.skip("test_backward_error_vs_float64 did not run in this session")
    worst = {}
    for errs in _BWD_RESULTS.values():
        for mode in SPLITS:
            for key in ("y", "g", "dx", "t_rms"):
                for ref in ("torch_fp32", "fp32"):
                    k = (mode, key, ref)
                    worst[k] = max(worst.get(k, 0.0), errs[mode][key] / errs[ref][key])
    print(f"worst error ratios over {len(_BWD_RESULTS)} cases (mode / yard-stick):")
    for (mode, key, ref), v in sorted(worst.items()):
        print(f"    {mode:7s} {key:6s} vs {ref:10s}: {v:.3f}")
    for m in SPLITS:
        for key in ("y", "g", "dx", "t_rms"):
            bar = 1.0 if key == "y" else 2.0
            assert worst[(m, key, "torch_fp32")] <= bar and worst[(m, key, "fp32")] <= bar, (m, key)


def test_sr_backward_every_tensor_vs_float64():
    """The SR 2x generator (BASELINE configs[1] / [3]: 32 -> 128 conv with the pixel-shuffled store, LeakyReLU(0.01), HRconv at the
    output resolution): 4 blocks, 256 x 256 -> 512 x 512, every parameter-gradient tensor (130 of them, incl. upsampling.0 and
    HRconv) and dL/dx of every mode row-wise against float64, and the same relative bars as the DN cases."""
    # Row-wise bars of this case: 2e-3 of the tensor's largest entry on rows without a flip candidate, 5e-3 anywhere off the
    # candidate rows, 2e-2 on them.  They are 10 x the DN case's because this functional has no input skip: every gradient is a
    # random-sign sum over the output pixels, and ONE LeakyReLU' decision that falls the other way in an early layer moves every
    # row of conv_first's gradient by 1e-3 of its maximum (measured, tools/dbg_sr_dx.py: the exact-fp32 mode included, different
    # pixels in different modes).  A structural error (a missing halo row, a wrong tile-edge tap) is a few percent on every row.
    errs, frac = _net_errors(256, 4, 7701, with_grad=True, batch=1, flip_aware=True, kind="sr", tight_w=2e-3, strict_w=5e-3)
    _table(f"SR 2x, 256^2 -> 512^2 x 4 blocks, seed 7701: errors vs float64 ({100 * frac:.0f}% of output pixels unclamped):", errs)
    t32, f32 = errs["torch_fp32"], errs["fp32"]
    for m in SPLITS:
        assert errs[m]["y"] <= t32["y"] and errs[m]["y"] <= f32["y"], m
    # backward: no ordering between the four fp32 paths is asserted here -- with this functional the flat-gradient rms IS the flip
    # noise (measured 2.4e-4 torch, 6.2e-4 bf16x6 on this seed; a handful of decisions); the bars are the row-wise ones above and
    # an absolute ceiling an order of magnitude under the task's 1e-3 ... of the tensor maxima
    for mode in MODES + ("torch_fp32",):
        assert errs[mode]["g"] < 5e-3 and errs[mode]["dx"] < 5e-3, mode


def test_sr_backward_every_tensor_at_the_configs3_share_vs_float64():
    """BASELINE configs[3]'s per-GPU share at full size: SR 2x train, 16 tiles of 512 x 512 -> 1024 x 1024, 4 blocks -- every one of
    the 130 parameter-gradient tensors and dL/dx of every math mode row-wise against float64 (the float64 pass and the flip-
    candidate pass run on the GPU, two tiles at a time), with the row-wise bars of the 256 x 256 SR case above (flip noise IS
    the gradient error of this functional).  4096 tiles over the persistent workgroups and 16 batch slices per plane: what the
    driver's 4-GPU run executes on every rank (round 4 had properties only at this size).
    dL/dx bars at this size: worst pixel 1e-1 of max |dL/dx| and 20 % of the image rows above 4e-4 (256 x 256: 5e-2 and 10 %).
    Measured: the EXACT-fp32 mode -- an fp32 fma chain on kernels that share no code with the split modes -- has 10.6 % of the rows
    over 4e-4 and a worst pixel of 5.6e-2; bf16x6 4.1 % / 2.3e-2, f16x3 5.9 % / 5.6e-2; relative L2 4 - 6e-4 for all three (bar:
    3e-3 since round 6, five times the measured value: a structural edge error of ~1 % L2 confined to a few rows -- which the two
    row-wise allowances above would let through -- fails it).  That is flip noise (64 x the LeakyReLU' decisions of the small case: the extreme of the same distribution), not a
    tile-edge error: a structural error would be percents of L2 and show in the row-wise check of every parameter gradient."""
    errs, frac = _net_errors(512, 4, 7901, with_grad=True, batch=16, flip_aware=True, kind="sr", tight_w=2e-3, strict_w=5e-3, with_torch32=False,
                             dx_loose=1e-1, dx_flip_frac=0.2, dx_l2_bar=3e-3)
    _table(f"SR 2x, 16 tiles of 512^2 -> 1024^2 x 4 blocks, seed 7901: errors vs float64 ({100 * frac:.0f}% of output pixels unclamped):", errs)
    for m in SPLITS:
        assert errs[m]["y"] <= errs["fp32"]["y"], m
    for mode in MODES:
        assert errs[mode]["g"] < 5e-3 and errs[mode]["dx"] < 5e-3 and errs[mode]["y"] < 1e-6, mode


def test_sr_full_size_forward_error_vs_float64():
    """BASELINE configs[1] at its tile size: SR 2x forward, 512 x 512 -> 1024 x 1024, 4 blocks, batch 2, against float64."""
    errs, frac = _net_errors(512, 4, 7801, with_grad=False, batch=2, kind="sr")
    _table(f"SR 2x forward, 2 tiles of 512^2 -> 1024^2 ({100 * frac:.0f}% of output pixels unclamped):", errs)
    for m in SPLITS:
        assert errs[m]["y"] <= errs["torch_fp32"]["y"] and errs[m]["y"] <= errs["fp32"]["y"], m
    for mode in MODES:
        assert errs[mode]["y"] < 1e-6, mode
