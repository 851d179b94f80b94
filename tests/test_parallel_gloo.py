"""World-size-2 data-parallel test on CPU (gloo): DataParallelTrainer's sharding + staged all-reduce + Adam with
grad_scale = 1/world must reproduce a single-process step on the full batch.  The compute engine is stubbed by the
CPU oracle (the HIP engine needs a GPU); the trainer code under test is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gen_common as gc


class OracleEngine:
    """Implements the engine protocol the trainer uses, on CPU tensors, via oracle/."""

    def __init__(self, kind, blocks):
        from oracle import oracle
        self.o, self.kind, self.blocks = oracle, kind, blocks
        self.num_stages = blocks + 2
        shapes = gc.rrdb_param_shapes(kind, 32, blocks)
        offs, off = {}, 0
        for k, s in shapes.items():
            offs[k] = off
            off += int(np.prod(s))
        self.n = off
        self.first_end = offs["rrdb.0.RDB1.conv1.weight"]
        self.rrdb_begin = [offs[f"rrdb.{i}.RDB1.conv1.weight"] for i in range(blocks)] + [offs["trunk_conv.weight"]]

    def pack(self, flat):
        self.p = flat.numpy().copy()

    def forward(self, x, save_for_backward=False):
        self.x = x.numpy()
        return torch.from_numpy(self.o.forward(self.kind, 32, self.blocks, self.p, self.x))

    def l1_loss(self, y, target):
        self.t = target.numpy()
        d = y.numpy() - self.t
        return torch.tensor(np.abs(d).mean(), dtype=torch.float32), torch.from_numpy(np.sign(d) / d.size)

    def backward_stage(self, st, dy, grads):
        if st == 0:
            _, _, _, g = self.o.l1_train(self.kind, 32, self.blocks, self.p, self.x, self.t)
            self._g = g
        off, cnt = self.grad_range(st)
        grads[off:off + cnt] = torch.from_numpy(self._g[off:off + cnt])

    def grad_range(self, st):
        b = self.blocks
        if st == 0:
            return self.rrdb_begin[b], self.n - self.rrdb_begin[b]
        if st <= b:
            i = b - st
            return self.rrdb_begin[i], self.rrdb_begin[i + 1] - self.rrdb_begin[i]
        return 0, self.first_end

    def adam_step(self, p, g, m, v, step, lr, betas, eps, grad_scale=1.0):
        gs = (g.numpy() * np.float32(grad_scale)).astype(np.float32)
        self.o.adam(p.numpy(), gs, m.numpy(), v.numpy(), step, lr, betas[0], betas[1], eps)


def _make(kind, blocks, seed):
    from xmm_superres_denoise.models import GeneratorRRDB_DN
    m = GeneratorRRDB_DN(1, 1, 32, blocks)
    st = gc.make_state(kind, 32, blocks, seed)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    return m


def _worker(rank, world, port, ret, nglobal=4):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xmm_superres_denoise.parallel import DataParallelTrainer
    torch.set_num_threads(2 if world <= 2 else 1)
    blocks = 1
    # rank 1 starts from different weights on purpose: the constructor's broadcast must fix that
    m = _make("dn", blocks, 300 + (7 if rank == 1 else 0))
    tr = DataParallelTrainer(m, lr=1e-4, engine=OracleEngine("dn", blocks))
    X = torch.from_numpy(gc.make_input((nglobal, 1, 12, 20), 301))
    T = torch.from_numpy(gc.make_input((nglobal, 1, 12, 20), 302))
    x, t = tr.shard(X), tr.shard(T)
    losses = []
    for _ in range(2):
        loss = tr.train_step(x, t)
        losses.append(float(tr.global_loss(loss)))
    ret[rank] = (tr.flat.numpy().copy(), losses)
    dist.destroy_process_group()


def test_dp2_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    # single process, full batch
    from xmm_superres_denoise.parallel import DataParallelTrainer
    m = _make("dn", 1, 300)
    tr = DataParallelTrainer(m, lr=1e-4, engine=OracleEngine("dn", 1))
    X = torch.from_numpy(gc.make_input((4, 1, 12, 20), 301))
    T = torch.from_numpy(gc.make_input((4, 1, 12, 20), 302))
    losses = [float(tr.train_step(X, T)) for _ in range(2)]
    p0, l0 = ret[0]
    p1, l1 = ret[1]
    assert np.array_equal(p0, p1), "replicas diverged"
    assert np.allclose(l0, losses, atol=1e-6) and np.allclose(l1, losses, atol=1e-6)
    # the update after two Adam steps is ~2*lr per weight; the mean-of-shard-gradients equals the full-batch gradient
    # up to fp32 summation order
    assert np.abs(p0 - tr.flat.numpy()).max() < 2e-6
    start = np.concatenate([v.ravel() for v in gc.make_state("dn", 32, 1, 300).values()])
    assert np.abs(p0 - start).max() > 1e-4  # it actually trained


def test_dp8_matches_single_process():
    """BASELINE configs[4]'s rank count: eight ranks (one tile each of a global batch of eight), the staged all-reduce of every
    gradient slice, Adam with grad_scale = 1/8 -- against one process on the full batch.  (On the GPU box the bench rehearses
    at most four ranks on the one card: tests/test_hip_parallel.py.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(8, port, ret, 8), nprocs=8, join=True)
    from xmm_superres_denoise.parallel import DataParallelTrainer
    m = _make("dn", 1, 300)
    tr = DataParallelTrainer(m, lr=1e-4, engine=OracleEngine("dn", 1))
    X = torch.from_numpy(gc.make_input((8, 1, 12, 20), 301))
    T = torch.from_numpy(gc.make_input((8, 1, 12, 20), 302))
    losses = [float(tr.train_step(X, T)) for _ in range(2)]
    for r in range(1, 8):
        assert np.array_equal(ret[0][0], ret[r][0]), f"replica {r} diverged"
        assert np.allclose(ret[r][1], losses, atol=1e-6)
    assert np.abs(ret[0][0] - tr.flat.numpy()).max() < 2e-6
    start = np.concatenate([v.ravel() for v in gc.make_state("dn", 32, 1, 300).values()])
    assert np.abs(ret[0][0] - start).max() > 1e-4


def _metric_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xmm_superres_denoise.utils.loss_functions import EpochState
    st = EpochState()
    # out vector of xsd_loss_eval: [total, l1, poisson, psnr, ssim, ms_ssim, mse, tmin, tmax]
    outs = [[0, 0.10, 0.5, 0, 0.80, 0.70, 0.005, 0.10, 0.60], [0, 0.30, 0.7, 0, 0.60, 0.50, 0.080, 0.05, 0.90]]
    st.add(torch.tensor(outs[rank], dtype=torch.float32), n=1000 * (rank + 1), nimg=2 * (rank + 1))
    st.sync()
    ret[rank] = {k: float(v) for k, v in st.compute().items()}
    dist.destroy_process_group()


def test_epoch_metric_states_are_reduced_not_averaged():
    """Multi-rank validation metrics: the STATES are summed / min-maxed over ranks before compute(), like torchmetrics'
    dist_reduce_fx (reference metrics/metrics.py:16-21).  PSNR of the pooled error differs from the mean of rank PSNRs."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_metric_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] == ret[1]
    sse = 0.005 * 1000 + 0.080 * 2000
    mse = sse / 3000
    dr = 0.90 - 0.0                       # running min starts at 0 (PeakSignalNoiseRatio(data_range=None))
    want = {"psnr": 10 * np.log10(dr * dr / mse), "l2": mse, "l1": (0.10 * 1000 + 0.30 * 2000) / 3000,
            "ssim": (0.80 * 2 + 0.60 * 4) / 6, "ms_ssim": (0.70 * 2 + 0.50 * 4) / 6, "poisson": (0.5 * 2 + 0.7 * 4) / 6}
    for k, v in want.items():
        assert abs(ret[0][k] - v) < 1e-6, (k, ret[0][k], v)
    mean_of_rank_psnr = 0.5 * (10 * np.log10(0.6 ** 2 / 0.005) + 10 * np.log10(0.9 ** 2 / 0.080))
    assert abs(ret[0]["psnr"] - mean_of_rank_psnr) > 0.1


def _uneven_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xmm_superres_denoise.models.model import _mean_over_ranks
    from xmm_superres_denoise.utils.loss_functions import EpochState
    st = EpochState()
    if rank == 0:       # rank 1's validation shard is empty: it must still enter the three collectives
        st.add(torch.tensor([0, 0.10, 0.5, 0, 0.80, 0.70, 0.005, 0.10, 0.60], dtype=torch.float32), n=1000, nimg=2)
    st.sync()
    vals = {k: float(v) for k, v in st.compute().items()}
    empty = EpochState()
    empty.sync()        # nobody saw a batch: the collectives still match and the state stays empty
    fb = _mean_over_ranks({"b": torch.tensor(float(rank)), "a": torch.tensor(10.0 + rank)})
    ret[rank] = (vals, empty.acc is None, {k: float(v) for k, v in fb.items()})
    dist.destroy_process_group()


def test_epoch_state_sync_with_an_empty_rank_does_not_hang():
    """EpochState.sync() used to return early on a rank without batches while the others entered three
    all_reduces (a hang with uneven validation shards).  Now every rank takes part with the identity state; collections
    without a sync() are mean-reduced over the ranks (Model._on_epoch_end's fallback)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_uneven_worker, args=(2, port, ret), nprocs=2, join=True)
    v0, e0, f0 = ret[0]
    v1, e1, f1 = ret[1]
    assert v0 == v1 and e0 and e1
    assert abs(v0["l2"] - 0.005) < 1e-9 and abs(v0["l1"] - 0.10) < 1e-7 and abs(v0["psnr"] - 10 * np.log10(0.6 ** 2 / 0.005)) < 1e-5
    assert f0 == f1 == {"a": 10.5, "b": 0.5}
