"""GeneratorRRDB_SR / GeneratorRRDB_DN with the reference constructor signatures and state_dict keys
(reference models/modules/generator_rrdb.py:9-64,72-101,113-121), computing through the MI355X engine.

forward(x[B,1,H,W] fp32, CUDA) -> clamp(...)[B,1,sH,sW]   (reference :103-110, :130-137)
Autograd is supplied by one torch.autograd.Function spanning the whole network: backward calls xsd_backward,
which returns dL/dx and every parameter gradient in one flat buffer laid out like the flat parameter buffer.
"""
from __future__ import annotations

import functools
import math
import os

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from xmm_superres_denoise.engine import Engine, XsdError

from .rrdb_blocks import RRDB, make_layer


def me_chunk() -> int:
    """tiles per recompute chunk of the memory_efficient path (activations kept: 2.1 GB per 512 x 512 tile)"""
    return max(1, int(os.environ.get("XSD_ME_CHUNK", "8")))


def forward_chunked(eng, x, chunk):
    """forward without keeping activations, `chunk` tiles at a time (outputs are bitwise independent of the batching)"""
    return torch.cat([eng.forward(x[i:i + chunk].contiguous(), save_for_backward=False) for i in range(0, x.shape[0], chunk)])


def backward_recompute(eng, x, dy, grads, need_dx, chunk, last_chunk_hook=None):
    """memory_efficient backward (the reference's checkpointing of every dense-block layer, rrdb_blocks.py:17-19,39-47,
    restated at batch granularity): re-run the forward of `chunk` tiles keeping activations, run their backward,
    accumulate the flat parameter gradient.  Peak activation memory = chunk tiles instead of the batch.
    last_chunk_hook(stage): called after each backward stage of the LAST chunk, when grads' range of that stage is final."""
    B = x.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    tmp = torch.empty_like(grads)
    starts = list(range(0, B, chunk))
    for i in starts:
        last = i == starts[-1]
        eng.forward(x[i:i + chunk].contiguous(), save_for_backward=True)
        dyc = dy[i:i + chunk].contiguous()
        dxc = torch.empty_like(x[i:i + chunk]) if need_dx else None
        for st in range(eng.num_stages):
            eng.backward_stage(st, dyc, tmp, dx=dxc)
            if last:
                off, cnt = eng.grad_range(st)
                if i == 0:
                    grads[off:off + cnt].copy_(tmp[off:off + cnt])
                else:
                    grads[off:off + cnt].add_(tmp[off:off + cnt])
                if last_chunk_hook is not None:
                    last_chunk_hook(st)
        if not last:
            if i == 0:
                grads.copy_(tmp)
            else:
                grads.add_(tmp)
        if need_dx:
            dx[i:i + chunk].copy_(dxc)
    return dx


def _split_flat(grads, plist):
    """the flat gradient as one view per parameter (state_dict order), None where none is wanted"""
    outs, off = [], 0
    for p in plist:
        n = p.numel()
        outs.append(grads[off:off + n].view_as(p) if p.requires_grad else None)
        off += n
    return outs


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, need, x, *params):
        eng = module._get_engine(x.device)
        eng.pack(module._flat)
        ctx.module = module
        ctx.need_dx = x.requires_grad
        ctx.recompute = bool(need and module.memory_efficient)
        ctx.flat_version = module._param_version()
        # torch's convs take an empty batch (empty output, zero gradients: the reference's modules do, e.g. for an empty shard);
        # the C ABI refuses B < 1, so there is nothing to launch and the answer is made here
        ctx.empty = x.dim() == 4 and x.shape[0] == 0 and x.shape[1] == eng.in_channels
        if need:
            ctx.save_for_backward(x)   # 1 channel: 1/2000 of the activations; lets a displaced context recompute
        if ctx.empty:
            ctx.generation = None
            return x.new_empty((0, eng.out_channels, x.shape[2] * eng.scale, x.shape[3] * eng.scale))
        if ctx.recompute:
            ctx.generation = None
            return forward_chunked(eng, x.contiguous(), me_chunk())
        y = eng.forward(x.contiguous(), save_for_backward=need)
        ctx.generation = eng.generation if need else None
        return y

    @staticmethod
    @once_differentiable      # the backward is kernels, not torch ops: a second differentiation (create_graph=True) is refused by name
    def backward(ctx, dy):
        m = ctx.module
        eng = m._engine
        grads = torch.empty_like(m._flat)
        (x,) = ctx.saved_tensors
        if tuple(dy.shape) != (x.shape[0], eng.out_channels, x.shape[2] * eng.scale, x.shape[3] * eng.scale):
            raise XsdError(f"dy has shape {tuple(dy.shape)} for an input of shape {tuple(x.shape)}")
        if ctx.empty:
            grads.zero_()
            dx = torch.zeros_like(x) if ctx.need_dx else None
            return (None, None, dx, *_split_flat(grads, m._plist))
        stale = not ctx.recompute and not eng.has_saved(ctx.generation)
        if (ctx.recompute or stale) and m._param_version() != ctx.flat_version:
            # same rule as torch's saved-tensor version check: the forward must be re-run with the weights it saw
            raise XsdError("parameters were modified in place between forward and backward of a pass whose activations "
                           "have to be recomputed (memory_efficient, or a later forward displaced them)")
        if ctx.recompute:
            eng.pack(m._flat)
            dx = backward_recompute(eng, x.contiguous(), dy.contiguous(), grads, ctx.need_dx, me_chunk())
        elif stale:
            # another forward of this module ran in between (y1 = m(x1); y2 = m(x2); (l1 + l2).backward()): the engine
            # holds one activation set, so re-run this context's forward from its saved input, then its backward
            eng.pack(m._flat)
            eng.forward(x.contiguous(), save_for_backward=True)
            dx = eng.backward(dy.contiguous(), grads, need_dx=ctx.need_dx, generation=eng.generation)
        else:
            dx = eng.backward(dy.contiguous(), grads, need_dx=ctx.need_dx, generation=ctx.generation)
        return (None, None, dx, *_split_flat(grads, m._plist))


class _GeneratorRRDB(nn.Module):
    _kind = None

    def __init__(self, in_channels: int, out_channels: int, num_filters: int, num_res_blocks: int,
                 memory_efficient: bool = False):
        super().__init__()
        # Any widths, like the reference (generator_rrdb.py:10-54).  The shipped configuration (res/configs/models.toml: 32
        # filters, one image channel), every width up to 256 filters and up to 8 image channels run on the split-precision MFMA
        # kernels; beyond that the exact-fp32 kernels of csrc/generic_net.hip.  What cannot work is said HERE, not at the first forward.
        for name, v in (("in_channels", in_channels), ("out_channels", out_channels), ("num_filters", num_filters)):
            if not 1 <= int(v) <= 1024:
                raise ValueError(f"{name} must be in [1, 1024] (got {v})")
        if self._kind == "dn" and in_channels != out_channels and in_channels != 1:
            raise ValueError(f"GeneratorRRDB_DN adds its input to the output (generator_rrdb.py:134): in_channels must equal "
                             f"out_channels or be 1 (got {in_channels}, {out_channels})")
        if not 1 <= int(num_res_blocks) <= 64:
            raise ValueError(f"num_res_blocks must be in [1, 64] (got {num_res_blocks})")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_filters = num_filters
        self.num_res_blocks = num_res_blocks
        self.memory_efficient = memory_efficient
        # same construction order as the reference => same default init under the same torch seed
        rrdb = functools.partial(RRDB, nf=num_filters, gc=num_filters, memory_efficient=memory_efficient)
        self.conv_first = nn.Conv2d(in_channels, num_filters, 3, 1, 1)
        self.rrdb = make_layer(rrdb, num_res_blocks)
        self.trunk_conv = nn.Conv2d(num_filters, num_filters, 3, 1, 1)
        self.conv_last = nn.Conv2d(num_filters, out_channels, 3, 1, 1)
        # positive-biased init of conv_last (reference generator_rrdb.py:56-64)
        positive_offset_std = 0.01
        stdv = 1.0 / math.sqrt(self.conv_last.weight.size(1))
        self.conv_last.weight.data.uniform_(-stdv, stdv + positive_offset_std * stdv)
        if self.conv_last.bias is not None:
            self.conv_last.bias.data.uniform_(-stdv, stdv + positive_offset_std * stdv)
        self._engine = None
        self._engine_dev = None
        self._flat = None
        self._plist = None
        self._math = None  # None: engine default (env XSD_MATH, else f16x3)

    # ---- copies and pickles (copy.deepcopy for EMA / SWA copies, torch.save(module), spawn-style launchers) --------------------
    def __getstate__(self):
        """The engine handle belongs to this process and this module: a copy or an unpickled module builds its own (and lays its own
        flat parameter buffer) at its first forward, exactly like a freshly constructed one."""
        st = self.__dict__.copy()
        st["_engine"] = st["_engine_dev"] = st["_flat"] = st["_plist"] = None
        return st

    # ---- flat parameter buffer ---------------------------------------------------------------------------------
    def _num_upsample(self):
        return 1

    def flatten_parameters(self):
        """Make every parameter a view into ONE contiguous fp32 buffer in state_dict order (the engine's flat-params
        layout, include/xsd.h).  Re-run automatically after .to()/.cuda() moved the parameters."""
        plist = list(self.parameters())
        dev = plist[0].device
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            off = 0
            base = self._flat.data_ptr()
            for p in plist:
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            # The first forward of a Lightning run is a validation sanity check under torch.inference_mode(): a buffer made there would be
            # an inference tensor and the parameters views of it -- no optimizer could ever update them.  Make it a normal tensor.
            with torch.inference_mode(False):
                flat = torch.empty(sum(p.numel() for p in plist), dtype=torch.float32, device=dev)
                off = 0
                for p in plist:
                    n = p.numel()
                    flat[off:off + n].copy_(p.data.reshape(-1).float())
                    p.data = flat[off:off + n].view(p.shape)
                    off += n
            self._flat = flat
        self._plist = plist
        return self._flat

    def flat_parameters(self) -> torch.Tensor:
        return self.flatten_parameters()

    def _param_version(self) -> int:
        """Changes with every in-place update torch knows of: the parameters keep their OWN version counters when their .data is
        pointed into the flat buffer (an optimizer step or load_state_dict bumps the parameter's, an update of the flat buffer itself
        -- the fused Adam of parallel.py goes through a raw pointer and is sequenced by its caller -- the buffer's)."""
        return self._flat._version + sum(p._version for p in self._plist)

    def set_math(self, mode: str):
        """Math mode of the conv kernels (Engine.set_math): 'fp32' (exact), 'bf16x6' (strict split), 'f16x3' (default split)."""
        if mode not in Engine.MATH:
            raise ValueError(mode)
        self._math = mode
        if self._engine is not None:
            self._engine.set_math(mode)
        return self

    def _get_engine(self, device):
        if not torch.device(device).type == "cuda":
            raise XsdError("the MI355X engine needs CUDA(HIP) tensors; there is no CPU fallback")
        flat = self.flatten_parameters()
        if flat.device != torch.device(device):
            raise XsdError(f"module parameters are on {flat.device} but the input is on {device}")
        if self._engine is None or self._engine_dev != flat.device:
            with torch.cuda.device(flat.device):
                self._engine = Engine(self._kind, self.in_channels, self.out_channels, self.num_filters,
                                      self.num_res_blocks, self._num_upsample(), self.memory_efficient)
            self._engine_dev = flat.device
            if self._math is not None:
                self._engine.set_math(self._math)
        return self._engine

    def forward(self, x):
        if x.dtype != torch.float32:
            raise XsdError(f"input must be float32 (got {x.dtype})")
        self._get_engine(x.device)
        # grad mode is off inside autograd.Function.forward, so decide here whether activations must be kept
        need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._plist))
        return _EngineFn.apply(self, need, x, *self._plist)


class GeneratorRRDB_SR(_GeneratorRRDB):
    _kind = "sr"

    def __init__(self, in_channels: int, out_channels: int, num_filters: int, num_res_blocks: int,
                 num_upsample: int = 2, memory_efficient: bool = False):
        super().__init__(in_channels=in_channels, out_channels=out_channels, num_filters=num_filters,
                         num_res_blocks=num_res_blocks, memory_efficient=memory_efficient)
        if num_upsample not in (1, 2):
            raise ValueError(f"the MI355X engine supports num_upsample 1 or 2 (got {num_upsample})")
        self.num_upsample = num_upsample
        layers = []
        for _ in range(num_upsample):  # reference generator_rrdb.py:91-99
            layers += [nn.Conv2d(num_filters, num_filters * 4, 3, 1, 1), nn.LeakyReLU(inplace=True), nn.PixelShuffle(2)]
        self.upsampling = nn.Sequential(*layers)
        self.HRconv = nn.Conv2d(num_filters, num_filters, 3, 1, 1, bias=True)

    def _num_upsample(self):
        return self.num_upsample


class GeneratorRRDB_DN(_GeneratorRRDB):
    _kind = "dn"

    def __init__(self, in_channels, out_channels, num_filters, num_res_blocks, memory_efficient=False):
        super().__init__(in_channels=in_channels, out_channels=out_channels, num_filters=num_filters,
                         num_res_blocks=num_res_blocks, memory_efficient=memory_efficient)
