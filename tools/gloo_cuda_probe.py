#!/usr/bin/env python3
"""What does gloo's device-tensor all-reduce cost when N ranks share ONE GPU?  (the round-3 stall -- three ranks
5 s per step, four ranks not back inside the call -- was routed around through a pinned host buffer without a diagnosis.)

N ranks on cuda:0 over gloo, no engine involved:
  phase A  all_reduce of a 6.7 MB CUDA tensor, nothing else in flight                (is the collective itself slow?)
  phase B  the same as six slices issued async_op=True and waited for at the end     (the trainer's call pattern)
  phase C  phase B with ~20 ms of matmul kernels enqueued in front of every slice    (does it serialise behind device work?)
  phase D  the pinned-host detour of parallel.py for comparison
Every rank prints its per-iteration wall times; the process group carries a 60 s timeout so a stuck collective ends the run.

    python tools/gloo_cuda_probe.py N        # starts its own N ranks
"""
import datetime
import os
import socket
import subprocess
import sys
import time


def worker():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    dev = torch.device("cuda", 0)
    n = 1_670_657
    g = torch.full((n,), float(rank + 1), device=dev)
    a = torch.randn(2048, 2048, device=dev)
    cuts = [0, 289, 9537, 424737, 839937, 1255137, n]      # the flat gradient's stage slices of the 4-block DN net

    def timed(tag, fn, iters=5):
        ts = []
        for _ in range(iters):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"[rank {rank}/{world}] {tag}: " + " ".join(f"{t:.1f}" for t in ts) + " ms", flush=True)

    def phase_a():
        dist.all_reduce(g)

    def phase_b():
        ws = [dist.all_reduce(g[cuts[i]:cuts[i + 1]], async_op=True) for i in range(6)]
        for w in ws:
            w.wait()

    def phase_c():
        ws = []
        for i in range(6):
            b = a
            for _ in range(8):
                b = b @ a * 1e-3
            ws.append(dist.all_reduce(g[cuts[i]:cuts[i + 1]], async_op=True))
        for w in ws:
            w.wait()

    host = torch.empty(n).pin_memory()

    def phase_d():
        ws = []
        for i in range(6):
            b = a
            for _ in range(8):
                b = b @ a * 1e-3
            h = host[cuts[i]:cuts[i + 1]]
            h.copy_(g[cuts[i]:cuts[i + 1]], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            ws.append((dist.all_reduce(h, async_op=True), i))
        for w, i in ws:
            w.wait()
            g[cuts[i]:cuts[i + 1]].copy_(host[cuts[i]:cuts[i + 1]], non_blocking=True)

    timed("A device tensor, one call, idle device", phase_a)
    g.fill_(float(rank + 1))
    timed("B device tensor, six async slices", phase_b)
    g.fill_(float(rank + 1))
    timed("C six async slices behind matmuls", phase_c)
    g.fill_(float(rank + 1))
    timed("D pinned-host detour behind matmuls", phase_d)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if "RANK" in os.environ:
        return worker()
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    if world > 3:
        # profiles/r04_gloo_cuda_probe.txt: with four processes on ONE device the probe finished no iteration in 150 s and had to be
        # killed mid-collective on the GPU.  That result is on record; reproducing it means leaving hung GPU work behind.
        sys.exit("gloo_cuda_probe: refusing N > 3 ranks on one GPU (the 4-rank stall is recorded in profiles/r04_gloo_cuda_probe.txt; "
                 "do not re-run it to reproduce the hang)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=200)
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
