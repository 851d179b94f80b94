"""Data-parallel training of the RRDB generators: one process per GPU, RCCL (torch.distributed backend "nccl")
all-reduce of the flat gradient vector over xGMI, overlapped with the backward pass.

What it replaces in the reference: Lightning's DDP strategy + DistributedSampler + torch.optim.Adam
(train.py:141-155; models/model.py:72-86,239-247).  The reference never touches a collective itself; the semantics
reproduced here are DDP's: every rank runs forward/backward on its shard of the global minibatch, gradients are
averaged over ranks, every rank applies the identical Adam update.

Layout: parameters, gradients and Adam moments are single flat fp32 buffers (6.7 MB), so the whole exchange is
`num_res_blocks + 2` all-reduces on contiguous slices: xsd_backward_stage(s) finishes a slice, its all-reduce is
issued immediately on RCCL's stream while later stages compute.  The 1/world mean is folded into the Adam kernel.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def collectives_on(group=None) -> bool:
    """Does this process take the distributed code path?  Yes with more than one rank -- and with ONE rank when XSD_FORCE_DP=1:
    a one-rank RCCL group on a single MI355X then executes every collective line a real node executes (communicator init
    with `device_id`, the in-place asynchronous all-reduce of each gradient slice on RCCL's stream, `wait()`, the device-tensor
    reductions of bench.py), with results that must be bit-equal to the plain single-process step (tests/test_hip_parallel.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("XSD_FORCE_DP", "0") == "1"


def all_reduce_any(t: torch.Tensor, op=None, group=None) -> torch.Tensor:
    """dist.all_reduce in place, wherever `t` lives: device tensors go through the host under gloo (the rehearsal backend; its
    device-tensor all-reduce stalls behind device work when three or more processes share one MI355X: DataParallelTrainer), stay
    on the device under RCCL."""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


class DataParallelTrainer:
    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, engine=None,
                 process_group=None, loss=None):
        """loss: None = mean L1 (BASELINE configs[2]); or the object `utils.create_loss` returns (the reference's
        composed PSNR / MS-SSIM / ... loss, utils/loss_functions.py:11-47), evaluated by xsd_loss_eval."""
        self.model = model
        self.loss = loss
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        self.flat = model.flat_parameters()
        self.engine = engine if engine is not None else model._get_engine(self.flat.device)
        self.grads = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.step_count = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.distributed = collectives_on(process_group)      # world > 1, or one rank with XSD_FORCE_DP=1 (the RCCL path on one GPU)
        # gloo (the rehearsal backend for boxes with fewer GPUs than ranks): its device-tensor all-reduce is not used.  With three
        # ranks sharing ONE MI355X it took 5 s per batch-2 step (round 3, ~60x too slow), with four it did not come back.
        # tools/gloo_cuda_probe.py reproduces that with no engine code at all -- asynchronous gloo all-reduces of device tensors
        # behind plain torch matmuls: fine with two processes on the device, 0.6-2.9 s per iteration with three, no return with
        # four (profiles/r04_gloo_cuda_probe.txt): gloo's CUDA path against the kernels of >= 3 processes time-slicing one device.
        # It cannot occur with one process per GPU.  Under gloo the gradient slices therefore travel through a pinned host buffer
        # and are reduced as CPU tensors; RCCL ("nccl") reduces in place on the device, asynchronously.
        self._host = None
        self._copy_stream = None
        self._comm_events = None      # measurement (bench.py): [(before, after)] event pairs around the wait for the exchange
        if self.distributed:
            # replicas must start identical (DDP broadcasts rank 0's parameters at construction)
            if dist.get_backend(process_group) == "gloo" and self.flat.is_cuda:
                self._host = torch.empty(self.flat.numel(), dtype=torch.float32).pin_memory()
                self._copy_stream = torch.cuda.Stream(device=self.flat.device)
                self._host.copy_(self.flat)
                dist.broadcast(self._host, src=0, group=process_group)
                self.flat.copy_(self._host)
            else:
                dist.broadcast(self.flat, src=0, group=process_group)

    def shard(self, global_batch: torch.Tensor) -> torch.Tensor:
        """This rank's contiguous slice of a global minibatch (DistributedSampler analogue, no shuffling)."""
        rank = dist.get_rank(self.pg) if self.distributed else 0
        n = global_batch.shape[0]
        if n % self.world:
            raise ValueError(f"global batch {n} is not divisible by world size {self.world}")
        per = n // self.world
        return global_batch[rank * per:(rank + 1) * per].contiguous()

    def train_step(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        eng = self.engine
        eng.pack(self.flat)
        recompute = bool(getattr(self.model, "memory_efficient", False))
        if recompute:   # rrdb_blocks.py:39-47 policy at batch granularity: keep no activations, recompute per chunk
            from xmm_superres_denoise.models.modules.generator_rrdb import backward_recompute, forward_chunked, me_chunk
            y = forward_chunked(eng, x, me_chunk())
        else:
            y = eng.forward(x, save_for_backward=True)
        loss, dy = eng.l1_loss(y, target) if self.loss is None else self.loss.value_and_grad(y, target)
        works = []        # RCCL: (work, None) per stage
        staged = []       # gloo: (copy-done event, offset, count) per stage

        def reduce_stage(st):
            if not self.distributed:
                return
            off, cnt = eng.grad_range(st)
            if self._host is None:
                # RCCL: in place on the device.  torch's NCCL process group makes its own stream wait for everything enqueued on
                # the current stream so far (this stage's kernels) and returns at once; later stages run beside the exchange
                works.append((dist.all_reduce(self.grads[off:off + cnt], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), None))
            else:
                # gloo: device -> pinned host on a COPY stream that waits for an event recorded behind the stage's kernels.  The
                # host does not block here (round 3 synchronised the compute stream at this point, which hid exactly the
                # stage -> exchange ordering the RCCL branch relies on and serialised the rehearsal): later stages are enqueued
                # at once, the host reductions start below as the copies land
                cur = torch.cuda.current_stream(self.grads.device)
                ready = torch.cuda.Event()
                ready.record(cur)
                with torch.cuda.stream(self._copy_stream):
                    self._copy_stream.wait_event(ready)
                    self._host[off:off + cnt].copy_(self.grads[off:off + cnt], non_blocking=True)
                    landed = torch.cuda.Event()
                    landed.record(self._copy_stream)
                staged.append((landed, off, cnt))

        if recompute:
            backward_recompute(eng, x, dy, self.grads, False, me_chunk(), reduce_stage)
        else:
            for st in range(eng.num_stages):
                eng.backward_stage(st, dy, self.grads)
                reduce_stage(st)
        for landed, off, cnt in staged:          # gloo: stage s is reduced on the host while the device still computes stages > s
            landed.synchronize()
            works.append((dist.all_reduce(self._host[off:off + cnt], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), (off, cnt)))
        # Exposed communication = what the compute stream spends between its last backward kernel and Adam.  Under RCCL
        # `wait()` does not block the host: it makes the current stream wait for RCCL's; under gloo the host blocks and the
        # copies back are enqueued afterwards.  Either way an event pair on the compute stream around this loop measures it.
        ev = None
        if self._comm_events is not None and self.distributed:
            cur = torch.cuda.current_stream(self.grads.device)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(cur)
        for w, back in works:
            w.wait()
            if back is not None:
                off, cnt = back
                self.grads[off:off + cnt].copy_(self._host[off:off + cnt], non_blocking=True)
        if ev is not None:
            ev[1].record(cur)
            self._comm_events.append(ev)
        self.step_count += 1
        eng.adam_step(self.flat, self.grads, self.m, self.v, self.step_count, self.lr, self.betas, self.eps,
                      grad_scale=1.0 / self.world)
        return loss

    def comm_events_begin(self):
        """start recording one event pair per train step around the wait for the gradient exchange (bench.py: comm_ms_exposed)"""
        self._comm_events = []

    def comm_events_end(self) -> float:
        """total milliseconds the compute stream spent waiting for the exchange since comm_events_begin (0.0 without a process
        group: there is no exchange); call after a device synchronize"""
        evs, self._comm_events = self._comm_events or [], None
        return float(sum(a.elapsed_time(b) for a, b in evs))

    def global_loss(self, local_loss: torch.Tensor) -> torch.Tensor:
        """Mean of the per-rank mean losses (what `sync_dist=True` logging reports, models/model.py:118)."""
        if not self.distributed:
            return local_loss
        t = local_loss.detach().clone()
        all_reduce_any(t, dist.ReduceOp.SUM, self.pg)
        return t / self.world
