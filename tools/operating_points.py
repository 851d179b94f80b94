#!/usr/bin/env python3
"""operating_points.py -- the engine at the REFERENCE's own operating points (round 6).

The reference trains at batch 1 / 4 / 8 on 416 x 416 (832 x 832 HR) tiles (res/baseline_config.toml:13,36,43;
res/configs/runs/esr_gen_local_test.yaml:26, rrdb_denoise_local_test.yaml:26) and infers one image per call
(xmm_superres_denoise/utils/run_inference_on_file.py:36-39); every kernel decision of rounds 1-5 was taken at batch 32 x 512^2.
This tool runs, in ONE process on one device, the matrix
    {dn_train, sr_train, dn_fwd, sr_fwd} x batch {32, 16, 8, 4, 1} x tile {512, 416}     in f16x3
    and the same workloads at batch {32, 4} in bf16x6 (strict),
no per-launch profile events (bench.py --no-profile), the same step as bench.py (repack -> forward -> L1 -> backward -> Adam for the
train cells; forward for the others; synthetic U[0,1) tiles resident in HBM), and prints each cell as tiles/s AND as a fraction of
the batch-32 / 512^2 PER-PIXEL rate of the same workload and math mode:
    frac = (tiles/s x tile^2) / (tiles/s at batch 32, 512^2  x  512^2).
A cell: 2 warm-up steps, one probe step, then as many steps as fit ~`--seconds` (3 .. 400), bracketed by synchronize().

usage (GPU box):  python3 tools/operating_points.py [--seconds 1.0] [--out gpurun_out/r06_operating_points.txt] [--only dn_train]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

WORKLOADS = ["dn_train", "sr_train", "dn_fwd", "sr_fwd"]
BATCHES = [32, 16, 8, 4, 1]
TILES = [512, 416]


def run_cell(model, trainer, kind, train, B, T, seconds, dev):
    scale = 2 if kind == "sr" else 1
    x = torch.rand((B, 1, T, T), generator=torch.Generator().manual_seed(0)).to(dev)
    tgt = torch.rand((B, 1, T * scale, T * scale), generator=torch.Generator().manual_seed(1)).to(dev) if train else None

    def step():
        if train:
            return trainer.train_step(x, tgt)
        with torch.no_grad():
            return model(x)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    n = max(3, min(400, int(seconds / max(one, 1e-5))))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del x, tgt
    return {"tiles_per_s": B * n / dt, "ms_per_step": 1e3 * dt / n, "steps": n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="comma list of workloads")
    ap.add_argument("--maths", default="f16x3,bf16x6")
    ap.add_argument("--batches", default=None)
    ap.add_argument("--tiles", default=None)
    args = ap.parse_args()
    from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
    from xmm_superres_denoise.parallel import DataParallelTrainer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wls = args.only.split(",") if args.only else WORKLOADS
    batches = [int(b) for b in args.batches.split(",")] if args.batches else BATCHES
    tiles = [int(t) for t in args.tiles.split(",")] if args.tiles else TILES
    rows = []
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit("# engine at the reference's operating points; lib = %s" % os.environ.get("XSD_LIB", "product"))
    emit("# device: %s" % torch.cuda.get_device_name(0))
    emit("%-9s %-7s %5s %5s %10s %10s %7s %6s" % ("workload", "math", "batch", "tile", "tiles/s", "ms/step", "frac", "steps"))
    for wl in wls:
        kind, mode = wl.split("_")
        train = mode == "train"
        torch.manual_seed(0)
        model = (GeneratorRRDB_DN(1, 1, 32, 4) if kind == "dn" else GeneratorRRDB_SR(1, 1, 32, 4, num_upsample=1)).to(dev)
        trainer = DataParallelTrainer(model, lr=1e-4, betas=(0.9, 0.999))
        for math in args.maths.split(","):
            model.set_math(math)
            bs = batches if math == "f16x3" else [b for b in batches if b in (32, 4)]
            base = None
            for T in tiles:
                for B in bs:
                    r = run_cell(model, trainer, kind, train, B, T, args.seconds, dev)
                    px_rate = r["tiles_per_s"] * T * T
                    if T == 512 and B == 32:
                        base = px_rate
                    frac = px_rate / base if base else float("nan")
                    rows.append({"workload": wl, "math": math, "batch": B, "tile": T, **r, "frac_of_b32_512_pixel_rate": frac})
                    emit("%-9s %-7s %5d %5d %10.2f %10.3f %7.3f %6d" % (wl, math, B, T, r["tiles_per_s"], r["ms_per_step"], frac, r["steps"]))
        del trainer, model
        torch.cuda.empty_cache()
    emit("# json: " + json.dumps(rows))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
