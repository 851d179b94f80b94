"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/xsd.h declares,
the Python surface mirrors the reference's names, and the product path fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from xmm_superres_denoise.engine import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "xsd.h")).read()
    declared = set(re.findall(r"\b(xsd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    L.xsd_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.xsd_version()


def test_python_surface_matches_reference_names():
    import xmm_superres_denoise.models as M
    import xmm_superres_denoise.transforms as T
    from xmm_superres_denoise.config.config import ModelCfg, OptimizerCfg, RrdbCfg, model_cfg
    for n in ("Model", "GeneratorRRDB_DN", "GeneratorRRDB_SR"):
        assert hasattr(M, n)
    for n in ("Crop", "ImageUpsample", "Normalize"):
        assert hasattr(T, n)
    import inspect
    sig = inspect.signature(M.GeneratorRRDB_SR.__init__)
    assert list(sig.parameters)[1:] == ["in_channels", "out_channels", "num_filters", "num_res_blocks", "num_upsample", "memory_efficient"]
    assert sig.parameters["num_upsample"].default == 2
    sig = inspect.signature(M.GeneratorRRDB_DN.__init__)
    assert list(sig.parameters)[1:] == ["in_channels", "out_channels", "num_filters", "num_res_blocks", "memory_efficient"]
    cfg = model_cfg("esr_gen", batch_size=4)
    assert isinstance(cfg, ModelCfg) and isinstance(cfg.model, RrdbCfg) and isinstance(cfg.optimizer, OptimizerCfg)
    m = M.Model(cfg, (416, 416), (832, 832), None, None, None, None, None)
    m.configure_model()
    assert isinstance(m.model, M.GeneratorRRDB_SR) and m.model.num_upsample == 1
    with pytest.raises(ValueError):
        M.Model(cfg, (416, 416), (1248, 1248), None, None, None, None, None).configure_model()
    opt = m.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 1e-4 and opt.defaults["betas"] == (0.9, 0.999)


def test_state_dict_keys_and_default_init_match_reference():
    import numpy as np
    import xmm_superres_denoise.models as M
    z = np.load(os.path.join(ROOT, "tests", "golden", "init_parity.npz"))
    for kind in ("dn", "sr"):
        torch.manual_seed(0)
        m = M.GeneratorRRDB_DN(1, 1, 32, 4) if kind == "dn" else M.GeneratorRRDB_SR(1, 1, 32, 4, num_upsample=1)
        sd = m.state_dict()
        assert list(sd.keys()) == [str(n) for n in z[kind + "_names"]]
        assert sum(p.numel() for p in m.parameters()) == int(z[kind + "_nparams"][0])
        st = z[kind + "_stats"]
        for i, k in enumerate(sd.keys()):
            v = sd[k].double().flatten()
            assert abs(v.sum().item() - st[i, 0]) < 1e-9 and abs(v.abs().sum().item() - st[i, 1]) < 1e-9, k
        flat = m.flatten_parameters()
        assert flat.numel() == int(z[kind + "_nparams"][0])
        assert m.conv_first.weight.data_ptr() == flat.data_ptr()
        # a round trip through state_dict keeps the flat aliasing
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
        assert m.conv_first.weight.data_ptr() == flat.data_ptr()


def test_product_path_fails_loudly_on_cpu():
    import xmm_superres_denoise.models as M
    from xmm_superres_denoise.engine import XsdError
    from xmm_superres_denoise.transforms import ImageUpsample, Normalize
    m = M.GeneratorRRDB_DN(1, 1, 32, 1)
    with pytest.raises(XsdError):
        m(torch.zeros(1, 1, 8, 8))
    with pytest.raises(XsdError):
        Normalize(1.0, 1.0, "sqrt").normalize_lr_image(torch.zeros(1, 4, 4))
    with pytest.raises(XsdError):
        ImageUpsample(2)(torch.zeros(1, 4, 4))


def test_no_product_import_of_oracle():
    """The product package must never import, call or link anything under oracle/."""
    pkg = os.path.join(ROOT, "xmm-superres-denoise_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower() or f == "README.md", os.path.join(dp, f)


def test_constructors_take_any_widths_like_the_reference():
    """reference constructors take any widths (generator_rrdb.py:10-54; rrdb_blocks.py:23 defaults nf = 64): so do these, with
    the reference's parameter names and shapes (checked against the names the reference itself produced for the nf = 8
    goldens); what cannot work is refused at construction, not at the first forward"""
    import numpy as np
    import gen_common as gc
    from xmm_superres_denoise import models as M
    for kind, name, nup in (("dn", "dn_nf8_b1", 1), ("sr", "sr_nf8_b1", 1), ("sr", "sr_nf8_b1_up2", 2)):
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        m = M.GeneratorRRDB_DN(1, 1, 8, 1) if kind == "dn" else M.GeneratorRRDB_SR(1, 1, 8, 1, num_upsample=nup)
        assert [n for n, _ in m.named_parameters()] == [str(n) for n in z["param_names"]]
        shapes = gc.rrdb_param_shapes(kind, 8, 1, num_upsample=nup)
        assert {n: tuple(p.shape) for n, p in m.named_parameters()} == dict(shapes)
    m = M.GeneratorRRDB_SR(3, 2, 64, 2, num_upsample=1)          # the dense block's own default width, RGB in
    assert m.conv_first.weight.shape == (64, 3, 3, 3) and m.rrdb[1].RDB3.conv5.weight.shape == (64, 320, 3, 3) and m.conv_last.weight.shape == (2, 64, 3, 3)
    with pytest.raises(ValueError, match="in_channels must equal"):
        M.GeneratorRRDB_DN(3, 2, 32, 1)                           # `out + x` cannot broadcast (generator_rrdb.py:134)
    with pytest.raises(ValueError, match="num_filters"):
        M.GeneratorRRDB_DN(1, 1, 0, 1)
    with pytest.raises(ValueError, match="num_upsample"):
        M.GeneratorRRDB_SR(1, 1, 32, 1, num_upsample=3)


def test_bench_contract_without_gpu():
    """bench.py's static contract, checked without touching a GPU: the math table agrees with the engine's, the default
    math is an fp32-class mode, and --gpus N > 1 outside torchrun takes the self-launch path (parent starts
    torch.distributed.run with N ranks and never calls torch.cuda)."""
    import importlib.util
    import sys
    from unittest import mock
    from xmm_superres_denoise.engine import Engine
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert set(bench.MATHS) == set(Engine.MATH)
    assert bench.DEFAULT_MATH == "f16x3" and "22-23 significant" in bench.MATHS["f16x3"][0]   # the headline's dtype label says what it is
    calls = {}

    def fake_run(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return mock.Mock(returncode=0)

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    with mock.patch.dict(os.environ, env, clear=True), mock.patch("subprocess.run", fake_run), \
            mock.patch.object(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"]), \
            mock.patch("torch.cuda.device_count", side_effect=AssertionError("parent must not touch the GPU")):
        with pytest.raises(SystemExit) as e:
            bench.main()
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and calls["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    with mock.patch.dict(os.environ, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}), \
            mock.patch.object(sys, "argv", ["bench.py", "--gpus", "4"]):
        with pytest.raises(SystemExit, match="does not match"):
            bench.main()


def test_library_holds_only_the_adopted_conv_kernel_instances():
    """csrc/conv3x3_h2x.hip: X3_KINDS names the epilogue kinds that run on their own kernel instance; the others are not
    instantiated at all (no dead device code in the shipped library)."""
    import re
    import subprocess
    from xmm_superres_denoise.engine import _lib
    src = open(os.path.join(_lib.CSRC_DIR, "conv3x3_h2x.hip")).read()
    kinds_mask = int(re.search(r"#define X3_KINDS (\d+)", src).group(1))
    kind_list = [int(v) for v in re.search(r"X3_KIND_LIST\[7\] = \{([^}]*)\}", src).group(1).split(",")]
    adopted = {k for i, k in enumerate(kind_list) if (kinds_mask >> i) & 1}
    syms = subprocess.run(["nm", "-C", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    built = {int(m) for m in re.findall(r"__device_stub__conv3x3_h2x_kind_kernel<(\d+)>", syms)}
    assert built == adopted, (built, adopted)


def test_product_library_reads_no_test_hook_from_the_environment():
    """Round 6: XSD_WGRAD_BLOCK / XSD_WGRAD_TAIL / XSD_TEST_NCU / XSD_TEST_AMAX_CAP live in the test-hooks variant only
    (make -C csrc hooks; include/xsd.h).  The product library's binary does not even hold their names; the hooks variant holds
    all four and exports the same C ABI."""
    import ctypes
    from xmm_superres_denoise.engine import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prod = os.path.join(root, "xmm-superres-denoise_amd", "lib", "libxsd_hip.so")
    hooks = os.path.join(root, "xmm-superres-denoise_amd", "lib", "libxsd_hip_hooks.so")
    names = (b"XSD_WGRAD_BLOCK", b"XSD_WGRAD_TAIL", b"XSD_TEST_NCU", b"XSD_TEST_AMAX_CAP")
    blob = open(prod, "rb").read()
    assert b"XSD_MATH" in blob
    for n in names:
        assert n not in blob, n
    assert os.path.exists(hooks), "make -C xmm-superres-denoise_amd/csrc hooks (__graft_entry__.build() does)"
    hb = open(hooks, "rb").read()
    for n in names:
        assert n in hb, n
    L = ctypes.CDLL(hooks)
    for s in _lib.ABI_SYMBOLS:
        assert hasattr(L, s), s
    L.xsd_version.restype = ctypes.c_char_p
    assert b"test-hooks" in L.xsd_version()


def test_persistent_conv_grid_rule():
    """csrc/xsd_kernels.h: persistent_grid (round 6; the measured tables: profiles/r06_grid_scan.txt).  A persistent conv launch of at most 8
    rounds uses the BALANCED grid -- the smallest one that needs no more rounds than the full grid -- and the full grid beyond.  Pure host
    arithmetic behind a diagnostic C-ABI entry: no device needed."""
    import ctypes
    from xmm_superres_denoise.engine import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    g = L.xsd_debug_persistent_grid
    g.argtypes = [ctypes.c_int, ctypes.c_int]
    assert g(200, 256) == 200 and g(256, 256) == 256 and g(1, 256) == 1 and g(0, 256) == 0
    assert g(338, 256) == 169            # a 416 x 416 image (the reference's tile), batch 1: 169 x 2 instead of 82 x 2 + 174 x 1
    assert g(1352, 256) == 226           # batch 4 (the reference's training batch): 6 rounds on 226 workgroups
    assert g(676, 256) == 226            # batch 2: 3 rounds
    assert g(512, 256) == 256 and g(1024, 256) == 256 and g(2048, 256) == 256      # full rounds stay full (512 x 512 at batch 1 / 2 / 4)
    assert g(576, 256) == 192
    assert g(2704, 256) == 256           # batch 8: 11 rounds -> the full grid
    assert g(16384, 256) == 256          # the bench batch (32 x 512 x 512): 64 rounds
    for ncu in (64, 120, 256, 304):
        for nt in range(0, 12 * ncu, 7):
            G = g(nt, ncu)
            rounds = -(-nt // ncu) if nt else 0
            assert G <= ncu and (nt == 0 or -(-nt // G) == rounds), (nt, ncu, G)       # never more rounds than the full grid needs
            if 1 < rounds <= 8:
                assert G == -(-nt // rounds), (nt, ncu, G)                              # ... and no workgroup more than those rounds need
    assert g(-1, 256) == -1 and g(10, 0) == -1


def test_normalize_is_picklable_like_the_reference():
    """transforms/normalize.py:35-62 of the reference holds its stretch functions as plain functions: a Normalize can be pickled (spawned
    DataLoader workers, deep copies of whatever holds one).  The mirror's .norm / .denorm are partials of a module-level function."""
    import copy
    import pickle
    from xmm_superres_denoise.transforms import Normalize
    n = Normalize(0.0022336, 0.0005584, "asinh")
    for other in (pickle.loads(pickle.dumps(n)), copy.deepcopy(n)):
        assert other.stretch_mode == "asinh" and float(other.hr_max) == float(n.hr_max)
        assert other.norm.keywords == {"mode": "asinh", "inverse": False} and other.denorm.keywords == {"mode": "asinh", "inverse": True}
