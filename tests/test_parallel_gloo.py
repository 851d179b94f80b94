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


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xmm_superres_denoise.parallel import DataParallelTrainer
    torch.set_num_threads(2)
    blocks = 1
    # rank 1 starts from different weights on purpose: the constructor's broadcast must fix that
    m = _make("dn", blocks, 300 + (7 if rank == 1 else 0))
    tr = DataParallelTrainer(m, lr=1e-4, engine=OracleEngine("dn", blocks))
    X = torch.from_numpy(gc.make_input((4, 1, 12, 20), 301))
    T = torch.from_numpy(gc.make_input((4, 1, 12, 20), 302))
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
