"""Worker of tests/test_hip_parallel.py::test_torch_ddp_two_ranks_matches_the_full_batch_step: one rank of a torch
DistributedDataParallel run over the generator module (what Lightning's DDP strategy does with the reference's Model, train.py:141-155).
Each rank takes its half of a seeded global batch, two SGD steps; rank 0 saves the parameters."""
import os
import sys

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd"), HERE, os.path.join(HERE, "golden")):
    sys.path.insert(0, p)
import gen_common as gc  # noqa: E402
from util_hip import build_module  # noqa: E402


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    backend = "nccl" if ndev >= world else "gloo"          # ranks share cuda:0 over gloo on a one-GPU box
    torch.cuda.set_device(rank % ndev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    state = gc.make_state("dn", 32, 1, 771)
    m = build_module("dn", 1, 1, state, device=f"cuda:{rank % ndev}")
    if rank == 1:      # DDP broadcasts rank 0's parameters at construction: start rank 1 somewhere else on purpose
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(0.5)
    net = DistributedDataParallel(m, device_ids=[rank % ndev])
    per = 2
    x = torch.from_numpy(gc.make_input((per * world, 1, 24, 40), 772))[rank * per:(rank + 1) * per].cuda()
    t = torch.from_numpy(gc.make_input((per * world, 1, 24, 40), 773))[rank * per:(rank + 1) * per].cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.l1_loss(net(x), t).backward()
        opt.step()
    flat = m.flatten_parameters()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    if backend == "gloo":
        host = [g.cpu() for g in gathered]
        dist.all_gather(host, flat.cpu())
        gathered = host
    else:
        dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"flat": gathered[0].cpu(), "identical": all(torch.equal(gathered[0].cpu(), g.cpu()) for g in gathered)}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
