"""ctypes binding of libxsd_hip.so (C ABI: include/xsd.h).  Fails loudly if the library is missing."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))          # .../xmm-superres-denoise_amd
LIB_PATH = os.environ.get("XSD_LIB") or os.path.join(_PKG_ROOT, "lib", "libxsd_hip.so")   # XSD_LIB: A/B builds
CSRC_DIR = os.path.join(_PKG_ROOT, "csrc")

_lib = None


class XsdError(RuntimeError):
    pass


class XsdConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("kind", "in_channels", "out_channels", "num_filters", "num_res_blocks", "num_upsample",
                 "memory_efficient", "reserved")]


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    if force:
        subprocess.check_call(["make", "-C", CSRC_DIR, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC_DIR, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XsdError(f"{LIB_PATH} not found: build it with `make -C {CSRC_DIR}` "
                       "(or __graft_entry__.build()). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, fp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    L.xsd_last_error.restype = ctypes.c_char_p
    L.xsd_version.restype = ctypes.c_char_p
    L.xsd_create.argtypes = [ctypes.POINTER(XsdConfig), ctypes.POINTER(vp)]
    L.xsd_destroy.argtypes = [vp]
    L.xsd_destroy.restype = None
    L.xsd_param_count.argtypes = [vp]
    L.xsd_param_count.restype = i64
    L.xsd_pack_weights.argtypes = [vp, fp, vp]
    L.xsd_set_math.argtypes = [vp, i32]
    L.xsd_get_math.argtypes = [vp]
    L.xsd_forward.argtypes = [vp, fp, fp, i32, i32, i32, i32, vp]
    L.xsd_backward.argtypes = [vp, fp, fp, fp, vp]
    L.xsd_backward_num_stages.argtypes = [vp]
    L.xsd_backward_stage.argtypes = [vp, i32, fp, fp, fp, vp]
    L.xsd_grad_range.argtypes = [vp, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.xsd_l1_loss.argtypes = [vp, fp, fp, fp, fp, i64, vp]
    L.xsd_loss_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(vp)]
    L.xsd_loss_destroy.argtypes = [vp]
    L.xsd_loss_destroy.restype = None
    L.xsd_loss_eval.argtypes = [vp, fp, fp, fp, fp, i32, i32, i32, vp]
    L.xsd_loss_set_channels.argtypes = [vp, i32]
    L.xsd_adam_step.argtypes = [vp, fp, fp, fp, fp, i64, i32, f32, f32, f32, f32, f32, vp]
    L.xsd_mask_pad_normalize.argtypes = [vp, i32, vp, fp, i32, i32, i32, i32, i32, f32, i32, vp]
    L.xsd_compose_input.argtypes = [vp, vp, vp, i32, i32, vp, fp, i32, i32, i32, i32, i32, i32, f32, i32, vp]
    L.xsd_normalize.argtypes = [fp, fp, i64, f32, i32, i32, vp]
    L.xsd_image_upsample.argtypes = [fp, fp, i32, i32, i32, i32, vp]
    L.xsd_profile_enable.argtypes = [vp, i32]
    L.xsd_debug_stamps.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_uint64)]
    L.xsd_profile_read.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64),
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.xsd_probe_mfma_stream.argtypes = [i32, ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), vp]
    L.xsd_test_conv3x3.argtypes = [vp, ctypes.POINTER(vp), i32, fp, fp, ctypes.POINTER(vp), i32, f32, i32, i32, i32, vp]
    L.xsd_test_conv3x3_bwd.argtypes = [vp, ctypes.POINTER(vp), i32, fp, fp, ctypes.POINTER(vp), fp, fp, i32, i32, i32, vp]
    _lib = L
    return L


# every symbol include/xsd.h declares (checked by tests/test_abi.py without a GPU)
ABI_SYMBOLS = [
    "xsd_last_error", "xsd_version", "xsd_create", "xsd_destroy", "xsd_param_count", "xsd_set_math", "xsd_get_math", "xsd_pack_weights",
    "xsd_forward", "xsd_backward", "xsd_backward_num_stages", "xsd_backward_stage", "xsd_grad_range",
    "xsd_l1_loss", "xsd_loss_create", "xsd_loss_destroy", "xsd_loss_eval", "xsd_loss_set_channels", "xsd_adam_step", "xsd_mask_pad_normalize", "xsd_compose_input", "xsd_normalize", "xsd_image_upsample",
    "xsd_profile_enable", "xsd_profile_read", "xsd_probe_mfma_stream", "xsd_debug_stamps", "xsd_debug_persistent_grid", "xsd_debug_occupancy", "xsd_debug_residency_ms", "xsd_test_conv3x3", "xsd_test_conv3x3_bwd",
]


def check(rc: int):
    if rc != 0:
        raise XsdError(f"libxsd_hip error {rc}: {load().xsd_last_error().decode()}")
