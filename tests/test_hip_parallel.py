"""Data-parallel training on the real HIP engine (reference: Lightning DDP, train.py:141-155).  Two ranks are started as
child processes (the pytest process only launches and compares); on a one-GPU box they share cuda:0 over gloo, which
exercises exactly the stream ordering that matters -- xsd_backward_stage's kernels on the compute stream vs the
asynchronous all-reduce of that stage's slice of the flat gradient (parallel.py) -- and with >= 2 GPUs the same test runs
over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(out_dir, world, backend, steps, math, kind="dn"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(out_dir), backend, str(steps), math, kind],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    return [np.load(os.path.join(out_dir, f"rank{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("math,kind", [("f16x3", "dn"), ("bf16x6", "dn"), ("fp32", "dn"), ("f16x3", "sr")])
def test_two_ranks_on_hip_engine_match_single_process_full_batch(tmp_path, math, kind):
    """kind "sr": the per-GPU share of BASELINE configs[3] (SR train, data parallel), two ranks"""
    import dp_worker as W
    from xmm_superres_denoise.parallel import DataParallelTrainer
    steps, world = 2, 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    res = _launch(tmp_path, world, backend, steps, math, kind)
    # (1) replicas stay bit-identical (same all-reduced gradient, same Adam update on every rank)
    assert np.array_equal(res[0]["params"], res[1]["params"])
    assert np.array_equal(res[0]["grads"], res[1]["grads"])
    # (2) and equal a single-process step on the full batch: rank 0's initial weights (broadcast), whole global batch
    model = W.build(300, kind).cuda().set_math(math)
    tr = DataParallelTrainer(model, lr=1e-3)
    x, t = W.global_batch(kind)
    x, t = x.cuda(), t.cuda()
    for s in range(steps):
        loss = float(tr.train_step(x, t))
        g_full = tr.grads.cpu().numpy()
        g_dp = res[0]["grads"][s] / world        # ranks hold SUM of per-rank mean-loss grads; Adam folds the 1/world
        scale = np.abs(g_full).max()
        # forward is bitwise batch-independent, so both sides see the same sign(y - t): only the summation order differs
        assert np.abs(g_dp - g_full).max() <= 2e-6 * scale, (s, np.abs(g_dp - g_full).max() / scale)
        assert abs(res[0]["losses"][s] - loss) <= 1e-6
        assert np.abs(res[0]["params"][s] - tr.flat.cpu().numpy()).max() <= 2e-5   # Adam: |update| <= lr = 1e-3 per step
    # the update moved the weights at all (lr 1e-3, 2 steps)
    assert np.abs(res[0]["params"][-1] - res[0]["params"][0]).max() > 1e-4
