"""One data-parallel rank on the REAL HIP engine (not a test module: started by tests/test_hip_parallel.py, one process
per rank).  Ranks may share one GPU (gloo backend) or own one each (nccl = RCCL).  Usage:
    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_worker.py <out_dir> <backend> <steps> <math> [dn|sr]
Writes <out_dir>/rank<r>.npz with the flat parameters after every step, the all-reduced flat gradients of every step and
the per-step local losses."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import gen_common as gc  # noqa: E402

BLOCKS, SHAPE, GLOBAL_B = 2, (40, 72), 4


def build(seed, kind="dn"):
    from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
    m = GeneratorRRDB_DN(1, 1, 32, BLOCKS) if kind == "dn" else GeneratorRRDB_SR(1, 1, 32, BLOCKS, num_upsample=1)
    st = gc.make_state(kind, 32, BLOCKS, seed, last_bias=0.2 if kind == "sr" else None)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    return m


def global_batch(kind="dn"):
    s = 2 if kind == "sr" else 1
    x = gc.make_input((GLOBAL_B, 1) + SHAPE, 501)
    t = gc.make_input((GLOBAL_B, 1, SHAPE[0] * s, SHAPE[1] * s), 502)
    return torch.from_numpy(x), torch.from_numpy(t)


def run(out_dir, backend, steps, math, kind="dn"):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", rank % ndev if backend != "nccl" else rank)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from xmm_superres_denoise.parallel import DataParallelTrainer
    # rank > 0 starts from different weights on purpose: the trainer's construction-time broadcast must fix that
    model = build(300 + rank, kind).to(dev).set_math(math)
    tr = DataParallelTrainer(model, lr=1e-3)
    x, t = global_batch(kind)
    xs, ts = tr.shard(x).to(dev), tr.shard(t).to(dev)
    params, grads, losses = [], [], []
    for _ in range(steps):
        losses.append(float(tr.global_loss(tr.train_step(xs, ts))))
        grads.append(tr.grads.cpu().numpy().copy())       # after the all-reduce: SUM over ranks of the local mean-loss grads
        params.append(tr.flat.cpu().numpy().copy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=np.stack(params), grads=np.stack(grads), losses=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "dn")
