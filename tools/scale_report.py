#!/usr/bin/env python3
"""Reads bench.py lines of one workload at several N (files holding the JSON line, or a driver record whose values contain such
lines) and says what a scaling curve is made of:
    efficiency(N) = value(N) / (N x value(1))
                  = [mean rank rate / the N = 1 rate]  x  [slowest rank's share: mean(ms) / max(ms)]  x  [what the barrier adds]
with `comm_ms_exposed` (the compute stream's wait for the gradient exchange) and the per-rank clocks / sustained matrix rates
beside it -- bench.py reports the MAX over ranks, so one slow device costs every rank its time before any byte is exchanged.
usage: python tools/scale_report.py n1.json n2.json n4.json n8.json"""
import json
import sys


def find_lines(obj, out):
    if isinstance(obj, dict):
        if "metric" in obj and "value" in obj and "n_gpus" in obj:
            out.append(obj)
            return
        for v in obj.values():
            find_lines(v, out)
    elif isinstance(obj, list):
        for v in obj:
            find_lines(v, out)
    elif isinstance(obj, str) and '"metric"' in obj:
        for l in obj.splitlines():
            l = l.strip()
            if l.startswith("{") and '"metric"' in l:
                try:
                    find_lines(json.loads(l), out)
                except ValueError:
                    pass


def load(paths):
    lines = []
    for p in paths:
        txt = open(p).read()
        try:
            find_lines(json.loads(txt), lines)
        except ValueError:
            find_lines(txt, lines)
    by_n = {}
    for d in lines:
        by_n[int(d["n_gpus"])] = d
    return by_n


def report(by_n):
    if 1 not in by_n:
        return "no N = 1 line: efficiencies need it"
    base = by_n[1]["value"]
    base_ms = by_n[1]["ms_per_step"]
    rows = [f"{'N':>3s} {'tiles/s':>10s} {'efficiency':>11s} {'mean rank / N=1':>16s} {'slowest share':>14s} {'barrier':>8s} {'comm ms exposed (max)':>22s}  per-rank sclk MHz / sustained TFLOP/s"]
    for n in sorted(by_n):
        d = by_n[n]
        eff = d["value"] / (n * base)
        pr = d.get("per_rank")
        if pr and pr.get("ms_per_step"):
            ms = [float(v) for v in pr["ms_per_step"]]
            mean_rate = base_ms / (sum(ms) / len(ms))                 # per-rank work is the same at every N (weak scaling)
            slow = (sum(ms) / len(ms)) / max(ms)
            barrier = max(ms) / d["ms_per_step"]
            comm = max((v for v in pr.get("comm_ms_exposed", []) if v is not None), default=None)
            extra = " ".join(f"{c or 0:.0f}/{s or 0:.0f}" for c, s in zip(pr.get("sclk_mhz", []), pr.get("sustained_mfma_tflops", [])))
            rows.append(f"{n:3d} {d['value']:10.1f} {eff:11.3f} {mean_rate:16.3f} {slow:14.3f} {barrier:8.3f} {comm if comm is not None else float('nan'):22.3f}  {extra}")
        else:
            rows.append(f"{n:3d} {d['value']:10.1f} {eff:11.3f} {'-':>16s} {'-':>14s} {'-':>8s} {d.get('comm_ms_exposed') if d.get('comm_ms_exposed') is not None else float('nan'):22.3f}")
    return "\n".join(rows)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    print(report(load(sys.argv[1:])))
