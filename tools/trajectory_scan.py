#!/usr/bin/env python3
"""How chaotic is the 200-step optimisation of tools/trajectory.py, as a function of the data?  Engine legs only (a 200-step run
takes a second): for every (tile size, batch, data seed, noise) the three math modes from the same start; printed: the largest
pairwise |PSNR difference| (dB, worst held-out tile) and |loss difference| among the modes at steps 50 / 100 / 200.  Three
fp32-class arithmetics that stay within 1e-3 dB of each other mark a configuration whose trajectory is NOT chaotic at fp32
rounding level -- the one to hold a math mode to an absolute bar on."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import trajectory as tj  # noqa: E402


def pairs(n, size, seed, noise):
    x, t = tj.denoise_pairs(n, size, seed)
    if noise != 0.12:
        rng = np.random.Generator(np.random.PCG64(seed + 1000))
        x = np.clip(t + noise * rng.standard_normal(t.shape), 0, 1).astype(np.float32)
    return x, t


def main():
    steps, cps = 200, (50, 100, 200)
    state = tj.start_state()
    print("size batch seed noise | max pairwise |dPSNR| dB at 50 / 100 / 200 | max pairwise |dloss| at 50 / 100 / 200 | final loss")
    for size, batch, seed, noise in itertools.product((64, 96, 128), (4, 16), (11, 12), (0.05, 0.12)):
        x, t = pairs(batch, size, seed, noise)
        xh, th = pairs(2, size, seed + 12, noise)
        res = {m: tj.run_engine(m, state, x, t, xh, th, steps, cps) for m in tj.ENGINE_MODES}
        dp, dl = [], []
        for c in cps:
            ps = [res[m][1][c] for m in tj.ENGINE_MODES]
            ls = [res[m][0][c - 1] for m in tj.ENGINE_MODES]
            dp.append(max(abs(a[i] - b[i]) for a in ps for b in ps for i in range(2)))
            dl.append(max(abs(a - b) for a in ls for b in ls))
        print(f"{size:4d} {batch:5d} {seed:4d} {noise:5.2f} | " + "  ".join(f"{v:9.2e}" for v in dp) + " | " + "  ".join(f"{v:9.2e}" for v in dl) +
              f" | {res['f16x3'][0][-1]:.5f}", flush=True)


if __name__ == "__main__":
    main()
