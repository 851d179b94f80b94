"""Thin host wrapper over the C ABI: torch supplies device memory and the HIP stream, nothing else."""
from __future__ import annotations

import ctypes
import functools

import torch

from . import _lib
from ._lib import XsdConfig, XsdError, check

STRETCH = {"linear": 0, "sqrt": 1, "asinh": 2, "log": 3}


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda_f32(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise XsdError(f"{name} must be a CUDA(HIP) tensor: the MI355X engine has no CPU fallback (got {t.device})")
    if t.dtype != torch.float32:
        raise XsdError(f"{name} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise XsdError(f"{name} must be contiguous")


def _on_engine_device(fn):
    @functools.wraps(fn)
    def guarded(self, *a, **k):
        with torch.cuda.device(self.device_index):
            return fn(self, *a, **k)
    return guarded


def _on_tensor_device(fn):
    """the stateless entry points launch on the current stream of their first tensor's device: make that device current for the call"""
    @functools.wraps(fn)
    def guarded(t, *a, **k):
        if isinstance(t, torch.Tensor) and t.is_cuda:
            with torch.cuda.device(t.device):
                return fn(t, *a, **k)
        return fn(t, *a, **k)       # (the argument checks of fn say what is wrong with it)
    return guarded


class Engine:
    """One engine per model instance per GPU (xsd_create / xsd_destroy)."""

    def __init__(self, kind: str, in_channels: int, out_channels: int, num_filters: int, num_res_blocks: int,
                 num_upsample: int = 1, memory_efficient: bool = False):
        self.L = _lib.load()
        cfg = XsdConfig(kind={"dn": 0, "sr": 1}[kind], in_channels=in_channels, out_channels=out_channels,
                        num_filters=num_filters, num_res_blocks=num_res_blocks, num_upsample=num_upsample,
                        memory_efficient=int(memory_efficient), reserved=0)
        h = ctypes.c_void_p()
        check(self.L.xsd_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        # The C engine allocates and launches on the CURRENT device (it never switches devices itself): every call below runs under a
        # guard for the device the engine was created on, so a process that holds modules on several GPUs, or whose current device is
        # not the module's, still puts workspace and kernels where the tensors are (one process per GPU -- the normal case -- pays a no-op)
        self.device_index = torch.cuda.current_device()
        self.kind = kind
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.scale = 2 ** num_upsample if kind == "sr" else 1
        self.nparams = int(self.L.xsd_param_count(self.h))
        self.num_stages = int(self.L.xsd_backward_num_stages(self.h))
        # The engine keeps ONE set of saved activations (one plan / workspace).  Every forward that saves them gets a new
        # generation id; a backward must name the generation it belongs to (autograd contexts do) and is refused when a
        # later forward has replaced the activations or when dy does not have the saved output's shape.
        self.generation = 0
        self._saved_gen = None
        self._saved_out_shape = None
        self._x_ref = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.xsd_destroy(self.h)
                self.h = None
        except Exception:
            pass

    MATH = {"fp32": 0, "bf16x6": 3, "f16x3": 4}

    @_on_engine_device
    def set_math(self, mode: str):
        """'fp32' (exact fp32 MFMA), 'bf16x6' (strict: exact 3-term bf16 split, 6 products, single-rounding MFMA accumulation)
        or 'f16x3' (default: 2-term fp16 split of power-of-two-scaled operands, 22-23 significant bits per operand, 3
        products); fp32 planes in all three.  include/xsd.h: xsd_set_math."""
        if mode not in self.MATH:
            raise XsdError(f"unknown math mode {mode!r}: the modes are {sorted(self.MATH)}")
        check(self.L.xsd_set_math(self.h, self.MATH[mode]))

    def get_math(self) -> str:
        return {v: k for k, v in self.MATH.items()}[int(self.L.xsd_get_math(self.h))]

    # ---- weights
    @_on_engine_device
    def pack(self, flat_params: torch.Tensor):
        _require_cuda_f32(flat_params, "flat_params")
        if flat_params.numel() != self.nparams:
            raise XsdError(f"flat_params has {flat_params.numel()} elements, engine expects {self.nparams}")
        self._params_ref = flat_params  # keep alive: the engine reads biases from it
        check(self.L.xsd_pack_weights(self.h, flat_params.data_ptr(), _stream_ptr(flat_params.device)))

    # ---- forward / backward
    @_on_engine_device
    def forward(self, x: torch.Tensor, save_for_backward: bool = False) -> torch.Tensor:
        _require_cuda_f32(x, "x")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise XsdError(f"x must be [B,{self.in_channels},H,W] (got {tuple(x.shape)})")
        B, _, H, W = x.shape
        y = torch.empty((B, self.out_channels, H * self.scale, W * self.scale), device=x.device, dtype=torch.float32)
        check(self.L.xsd_forward(self.h, x.data_ptr(), y.data_ptr(), B, H, W, int(save_for_backward), _stream_ptr(x.device)))
        # any forward rebuilds / reuses the workspace, so previously saved activations are gone either way
        self._x_ref = x if save_for_backward else None  # conv_first's weight gradient re-reads x
        self._saved_out_shape = tuple(y.shape) if save_for_backward else None
        if save_for_backward:
            self.generation += 1
            self._saved_gen = self.generation
        else:
            self._saved_gen = None
        return y

    def has_saved(self, generation: int) -> bool:
        """True while the activations saved by forward number `generation` are still the engine's current ones."""
        return generation is not None and self._saved_gen == generation

    def _check_backward_args(self, dy, flat_grads, dx, generation):
        _require_cuda_f32(dy, "dy")
        _require_cuda_f32(flat_grads, "flat_grads")
        if self._saved_gen is None:
            raise XsdError("backward needs a preceding forward(save_for_backward=True) whose activations are still held")
        if generation is not None and generation != self._saved_gen:
            raise XsdError(f"backward for forward #{generation}, but the engine holds the activations of forward "
                           f"#{self._saved_gen}: a later forward replaced them (one saved activation set per engine)")
        if tuple(dy.shape) != self._saved_out_shape:
            raise XsdError(f"dy has shape {tuple(dy.shape)}, the saved forward produced {self._saved_out_shape}")
        if flat_grads.numel() != self.nparams:
            raise XsdError(f"flat_grads has {flat_grads.numel()} elements, engine expects {self.nparams}")
        if dx is not None:
            _require_cuda_f32(dx, "dx")
            if tuple(dx.shape) != tuple(self._x_ref.shape):
                raise XsdError(f"dx has shape {tuple(dx.shape)}, the saved input has {tuple(self._x_ref.shape)}")

    @_on_engine_device
    def backward(self, dy: torch.Tensor, flat_grads: torch.Tensor, need_dx: bool = False, generation: int | None = None):
        self._check_backward_args(dy, flat_grads, None, generation)
        dx = torch.empty_like(self._x_ref) if need_dx else None
        check(self.L.xsd_backward(self.h, dy.data_ptr(), dx.data_ptr() if need_dx else None, flat_grads.data_ptr(),
                                  _stream_ptr(dy.device)))
        return dx

    @_on_engine_device
    def backward_stage(self, stage: int, dy: torch.Tensor, flat_grads: torch.Tensor, dx: torch.Tensor | None = None,
                       generation: int | None = None):
        self._check_backward_args(dy, flat_grads, dx, generation)
        check(self.L.xsd_backward_stage(self.h, stage, dy.data_ptr(), dx.data_ptr() if dx is not None else None,
                                        flat_grads.data_ptr(), _stream_ptr(dy.device)))

    def grad_range(self, stage: int):
        off, cnt = ctypes.c_int64(), ctypes.c_int64()
        self.L.xsd_grad_range(self.h, stage, 0, ctypes.byref(off), ctypes.byref(cnt))
        return off.value, cnt.value

    # ---- loss / optimizer
    @_on_engine_device
    def l1_loss(self, y: torch.Tensor, target: torch.Tensor, want_grad: bool = True):
        _require_cuda_f32(y, "y")
        _require_cuda_f32(target, "target")
        if y.shape != target.shape:
            raise XsdError(f"shape mismatch {tuple(y.shape)} vs {tuple(target.shape)}")
        dy = torch.empty_like(y) if want_grad else None
        loss = torch.empty((), device=y.device, dtype=torch.float32)
        check(self.L.xsd_l1_loss(self.h, y.data_ptr(), target.data_ptr(), dy.data_ptr() if want_grad else None,
                                 loss.data_ptr(), y.numel(), _stream_ptr(y.device)))
        return loss, dy

    @_on_engine_device
    def adam_step(self, params, grads, m, v, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        for n, t in (("params", params), ("grads", grads), ("m", m), ("v", v)):
            _require_cuda_f32(t, n)
        check(self.L.xsd_adam_step(self.h, params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                                   params.numel(), int(step), lr, betas[0], betas[1], eps, grad_scale,
                                   _stream_ptr(params.device)))

    # ---- measurement
    @_on_engine_device
    def profile_enable(self, on: bool):
        check(self.L.xsd_profile_enable(self.h, int(on)))

    @_on_engine_device
    def profile_read(self, klass: int):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        check(self.L.xsd_profile_read(self.h, klass, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)))
        return {"ms": ms.value, "launches": n.value, "flop": fl.value, "bytes": by.value}

    @_on_engine_device
    def probe_mfma_stream(self, fmt: str = "f16", seconds: float = 2.0):
        """dense 16-bit MFMA TFLOP/s and in-kernel clock the current device sustains on the conv's bare MFMA-wave stream
        (include/xsd.h: xsd_probe_mfma_stream); blocks for about `seconds`"""
        tf, gz = ctypes.c_double(), ctypes.c_double()
        check(self.L.xsd_probe_mfma_stream({"f16": 0, "bf16": 1}[fmt], float(seconds), ctypes.byref(tf), ctypes.byref(gz),
                                           _stream_ptr(torch.device("cuda", torch.cuda.current_device()))))
        return {"mfma_tflops": tf.value, "sclk_ghz": gz.value, "seconds": float(seconds), "fmt": fmt}


# ---- stateless transform entry points --------------------------------------------------------------------------
@_on_tensor_device
def mask_pad_normalize(counts: torch.Tensor, mask: torch.Tensor | None, res: int, max_val: float | None,
                       stretch: str = "linear") -> torch.Tensor:
    """counts [B,Hin,Win] int32|float32 (CUDA), mask [Hin,Win] uint8 -> [B,1,res,res] float32."""
    L = _lib.load()
    if not counts.is_cuda:
        raise XsdError("counts must be a CUDA(HIP) tensor: no CPU fallback")
    if counts.dtype not in (torch.int32, torch.float32) or not counts.is_contiguous():
        raise XsdError("counts must be contiguous int32 or float32")
    if mask is not None and (mask.dtype != torch.uint8 or not mask.is_cuda or tuple(mask.shape) != tuple(counts.shape[-2:])):
        raise XsdError("mask must be a CUDA uint8 tensor [Hin,Win]")
    B, Hin, Win = counts.shape
    out = torch.empty((B, 1, res, res), device=counts.device, dtype=torch.float32)
    check(L.xsd_mask_pad_normalize(counts.data_ptr(), int(counts.dtype == torch.int32),
                                   mask.data_ptr() if mask is not None else None, out.data_ptr(), B, Hin, Win, res,
                                   int(max_val is not None), float(max_val or 0.0), STRETCH[stretch],
                                   _stream_ptr(counts.device)))
    return out


@_on_tensor_device
def compose_input(img: torch.Tensor, agn: torch.Tensor | None, bkg: torch.Tensor | None, mask: torch.Tensor | None,
                  res: int, max_val: float | None, stretch: str = "linear", upsample: int = 1,
                  big_endian: bool = False) -> torch.Tensor:
    """One-kernel sample composition (include/xsd.h: xsd_compose_input): img (+agn) (+bkg) -> * mask -> optional nearest
    upsample / s^2 -> centred pad to res -> optional normalize.  img/agn/bkg: [B,Hin,Win] int32 or float32 (CUDA); with
    big_endian=True they hold raw FITS words (e.g. torch.frombuffer of the HDU data block viewed as int32)."""
    L = _lib.load()
    for n, t in (("img", img), ("agn", agn), ("bkg", bkg)):
        if t is None:
            continue
        if not t.is_cuda or t.dtype != img.dtype or tuple(t.shape) != tuple(img.shape) or not t.is_contiguous():
            raise XsdError(f"{n} must be a contiguous CUDA tensor with img's dtype and shape")
    if img.dtype not in (torch.int32, torch.float32):
        raise XsdError("img must be int32 or float32")
    if mask is not None and (mask.dtype != torch.uint8 or not mask.is_cuda or tuple(mask.shape) != tuple(img.shape[-2:])):
        raise XsdError("mask must be a CUDA uint8 tensor [Hin,Win]")
    B, Hin, Win = img.shape
    out = torch.empty((B, 1, res, res), device=img.device, dtype=torch.float32)
    check(L.xsd_compose_input(img.data_ptr(), agn.data_ptr() if agn is not None else None,
                              bkg.data_ptr() if bkg is not None else None, int(img.dtype == torch.int32), int(big_endian),
                              mask.data_ptr() if mask is not None else None, out.data_ptr(), B, Hin, Win, int(upsample), res,
                              int(max_val is not None), float(max_val or 0.0), STRETCH[stretch], _stream_ptr(img.device)))
    return out


@_on_tensor_device
def normalize(img: torch.Tensor, max_val: float, stretch: str, inverse: bool = False) -> torch.Tensor:
    L = _lib.load()
    _require_cuda_f32(img, "img")
    out = torch.empty_like(img)
    check(L.xsd_normalize(img.data_ptr(), out.data_ptr(), img.numel(), float(max_val), STRETCH[stretch], int(inverse),
                          _stream_ptr(img.device)))
    return out


@_on_tensor_device
def image_upsample(x: torch.Tensor, scale: int) -> torch.Tensor:
    L = _lib.load()
    _require_cuda_f32(x, "x")
    H, W = x.shape[-2:]
    n = x.numel() // (H * W)
    out = torch.empty(tuple(x.shape[:-2]) + (H * scale, W * scale), device=x.device, dtype=torch.float32)
    check(L.xsd_image_upsample(x.data_ptr(), out.data_ptr(), n, H, W, scale, _stream_ptr(x.device)))
    return out
