#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: XMM 512x512 tiles/sec for a train step.

A "step" = one pass of the hot path over one batch of synthetic tiles (x ~ U[0,1) seed 0, target ~ U[0,1) seed 1,
reference-default weight init under torch.manual_seed(0)):
    weight repack -> forward (activations kept) -> mean-L1 loss -> backward (input + weight gradients)
    -> [RCCL all-reduce of the flat gradient, overlapped with backward, when N > 1] -> fused Adam.
Workload: BASELINE configs[2] "XMM-DeNoise train step, batch 32, 1 MI355X, fwd+bwd HIP kernels".  The per-GPU batch is 32 at
EVERY N (weak scaling of configs[2]: the N = 1 line and the N > 1 lines run the same per-GPU work, so a scaling efficiency
computed from them is like-for-like); `--batch 16` gives the per-GPU share of configs[3]/[4] (64 over 4, 128 over 8).
Other workloads (parity-test configs, not bench lines): --workload sr_fwd (configs[1]), sr_train, dn_fwd.

Math modes (include/xsd.h: xsd_set_math), all with fp32 planes and fp32 MFMA accumulation:
  f16x3  (default, the headline `value`): operands as two-term fp16 splits of power-of-two-scaled tensors, 22-23 significant
         bits per operand (fp32 has 24), three fp16 MFMA products per multiply.  Measured against float64
         (tests/test_hip_precision.py; log under profiles/): forward error below torch's fp32 CPU path; backward within 2x of
         it and at the level of an exact fp32 fma chain.  `dtype` says exactly that, not "f32".
  bf16x6 (strict; the `extra` leg, timed over the same --steps/--warmup with its own roofline block): exact three-term bf16
         split, six products: error below torch fp32 everywhere, forward and backward.
  fp32   exact fp32 MFMA.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (one child process per GPU via
torch.distributed.run; the parent never touches the GPU) and relays rank 0's JSON line; under an external torchrun it
runs as a rank.  XSD_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : the dominant kernel's algorithmic FLOP (or bytes) / its average HIP-event launch time inside the timed
                 region against the peak that binds the mode (MI355X_MICROARCH.md); `sustained_peak` / `frac_of_sustained`: the
                 same against what THIS device sustains on the conv's bare MFMA stream, measured in this process after the timed
                 region (xsd_probe_mfma_stream: the roof the 1400 W package cap leaves of the nominal peak),
  power        : watts, clock, fraction of the cap and joules per tile over the timed region (sysfs, sampled between the two
                 synchronizes),
  psnr_delta_db: the second half of BASELINE's metric -- PSNR(engine in the timed math mode) - PSNR(reference) on the two
                 example_data tiles of tests/golden/example_data.npz, identical seeded weights, outside the timed region,
  comm_ms_exposed / per_rank (N > 1): what the compute stream waited for the gradient exchange; every rank's own step time,
                 device clock, watts and sustained matrix rate -- a < N x curve separates device spread from communication,
  cpu_baseline : the same train step (B=1) through oracle/oracle.py's torch restatement on the host cores (1 warm-up step,
                 then best of 3; every sample recorded).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3
TILE = 512


def host_cores() -> int:
    """CPU threads this process may really use: min(affinity, cgroup cpu.max quota, 16 = a 1-GPU box's CPU share)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(kind: str, train: bool, repeats: int = 3, tile: int = TILE):
    """Reference op graph on the host CPU (torch restatement, pinned to the reference by tests/test_oracle_pinned.py): B=1
    512x512 steps on all host cores -- one untimed warm-up step at full size (thread pool, oneDNN primitives, allocator),
    then `repeats` timed steps; the best is `value`, every sample is recorded."""
    from oracle import oracle
    from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = GeneratorRRDB_DN(1, 1, 32, 4) if kind == "dn" else GeneratorRRDB_SR(1, 1, 32, 4, num_upsample=1)
    state = {k: v.detach().clone().requires_grad_(train) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.rand((1, 1, tile, tile), generator=g)
    s = 2 if kind == "sr" else 1
    t = torch.rand((1, 1, tile * s, tile * s), generator=torch.Generator().manual_seed(1))
    opt = torch.optim.Adam(list(state.values()), lr=1e-4, betas=(0.9, 0.999)) if train else None

    def one_step():
        t0 = time.perf_counter()
        if train:
            opt.zero_grad(set_to_none=True)
            y = oracle.torch_forward(kind, 32, 4, state, x)
            loss = torch.nn.functional.l1_loss(y, t)
            loss.backward()
            opt.step()
        else:
            with torch.no_grad():
                oracle.torch_forward(kind, 32, 4, state, x)
        return time.perf_counter() - t0

    warm = one_step()
    samples = [one_step() for _ in range(repeats)]
    best = min(samples)
    return {"value": 1.0 / best, "unit": "tiles/s", "cores": cores, "kind": "port",
            "samples_s": [round(v, 3) for v in samples], "warmup_s": round(warm, 3),
            "sample": f"{'train step (fwd+L1+bwd+Adam)' if train else 'forward'} of 1 tile 1x{tile}x{tile}, torch-CPU restatement of the "
                      f"reference graph (oracle/oracle.py:torch_forward): 1 warm-up + best of {repeats} ({best:.1f} s)"}


BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0


class PowerWatch:
    """Samples the package power and the shader clock of THIS job's amdgpu devices from sysfs (hwmon power1_average / power1_input,
    freq1_input, power1_cap) while the timed region runs -- a reader thread, no subprocess, nothing on the GPU.  `pci` names the
    devices by PCI address prefix ("0000:05:00", from torch's device properties): a GPU box shows the hwmon files of every card
    of its host, other tenants' included.  Without it (tools, tests) the busiest device is reported and the source says so.
    The train step of this engine runs AT the package power cap: the clock the chip holds, and with it every
    kernel's milliseconds, is set by the energy a tile costs, so the line reports what was drawn beside what was computed.
    Returns None where the files are not readable (no GPU, other driver)."""

    def __init__(self, period=0.1, root="/sys/class/drm", pci=None):
        import glob
        import threading
        self.period = period
        self.devs = []
        self.matched = False
        found = []
        for h in sorted(glob.glob(os.path.join(root, "card*/device/hwmon/hwmon*"))):
            pw = next((os.path.join(h, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, f))), None)
            if pw:
                addr = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(h))))       # .../0000:05:00.0
                found.append({"hwmon": h, "pci": addr, "power": pw, "freq": os.path.join(h, "freq1_input"), "cap": os.path.join(h, "power1_cap"), "w": [], "mhz": []})
        if pci:
            mine = [d for d in found if any(d["pci"].lower().startswith(p.lower()) for p in pci)]
            if mine:
                found, self.matched = mine, True
        self.devs = found
        self._stop = threading.Event()
        self._thr = threading.Thread(target=self._run, daemon=True) if self.devs else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop.is_set():
            for d in self.devs:
                w = self._read(d["power"])
                if w is not None:
                    d["w"].append(w * 1e-6)
                f = self._read(d["freq"])
                if f is not None:
                    d["mhz"].append(f * 1e-6)
            self._stop.wait(self.period)

    def start(self):
        """called right after the synchronize() that closes the warm-up: every sample lies inside the timed region"""
        if self._thr and not self._thr.is_alive():
            self._thr.start()

    def stop(self):
        if self._thr and self._thr.is_alive():
            self._stop.set()
            self._thr.join()

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *a):
        self.stop()

    def summary(self):
        best = None
        avgs = []
        for d in self.devs:
            w = d["w"]              # every sample: the reader runs from the post-warm-up synchronize() to the closing one
            if len(w) < 3:
                continue
            avgs.append(round(sum(w) / len(w), 1))
            if best is None or sum(w) / len(w) > best["avg_w"]:
                mhz = d["mhz"]
                cap = self._read(d["cap"])
                best = {"avg_w": round(sum(w) / len(w), 1), "max_w": round(max(w), 1), "cap_w": round(cap * 1e-6, 1) if cap else None,
                        "sclk_mhz": round(sum(mhz) / len(mhz), 0) if mhz else None, "samples": len(w), "pci": d["pci"],
                        "source": "sysfs hwmon of %s, sampled every %.0f ms between the synchronize() that closes the warm-up and the one that closes the timed steps" %
                                  ("this job's device(s), by PCI address" if self.matched else "the BUSIEST amdgpu device of the host (not matched to this job)", 1e3 * self.period)}
        if best and best["cap_w"]:
            best["frac_of_cap"] = round(best["avg_w"] / best["cap_w"], 3)
        if best and self.matched and len(avgs) > 1:      # a multi-GPU run: every rank's device (rank 0 reads them all), busiest first
            best["devices_avg_w"] = sorted(avgs, reverse=True)
        return best


def device_pci(dev=0):
    """PCI address prefix of a torch device ("0000:05:00"), or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(dev)
        return "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def pmc_traffic(math: str, batch: int, klass: str):
    """HBM bytes per launch from the committed PMC passes (tools/traffic.sh -> profiles/rNN_traffic_<math>.json, newest round
    first), valid only for the workload/batch they were collected on.  Returns (bytes, file) or (None, None): the figure is
    READ FROM THAT FILE, not measured in this run (PMC passes need rocprofv3 around the process)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        rel = os.path.join("profiles", f"{rnd}_traffic_{math}.json")
        try:
            d = json.load(open(os.path.join(ROOT, rel)))
            if d.get("per_gpu_batch") == batch and d.get("workload") == "dn_train":
                return d[klass]["traffic_bytes_per_launch"], rel
        except Exception:
            pass
    return None, None


MATHS = {
    # mode: (dtype label = the arithmetic the path computes in, conv kernel name)
    "fp32": ("f32 (exact fp32 MFMA)", "conv3x3_mfma_kernel"),
    "bf16x6": ("f32 emulated exactly on bf16 MFMA (bf16x6: 3-term bf16 split = all 24 significant bits, 6 products per multiply, "
               "single-rounding f32 accumulate, f32 planes)", "conv3x3_s3x_kernel"),
    "f16x3": ("f16x3 (f32 planes and f32 accumulate; operands as 2-term fp16 splits of power-of-two-scaled tensors = 22-23 significant "
              "bits per operand, 3 fp16 MFMA products per multiply; gradients measured at 1.2-1.6x the float64-error of torch's fp32 path)",
              "conv3x3_h2x_kernel + its per-epilogue-kind instances conv3x3_h2x_kind_kernel<2>, <48> (one kernel body; rocprofv3 lists three names: the "
              "average launch time here is over all of them)"),
}
DEFAULT_MATH = "f16x3"
# include/xsd.h: xsd_profile_read classes.  0 / 1 are MFMA-bound; the others are HBM-bound ("report both GB/s vs 8 TB/s
# and MFMA-util ... per kernel; never substitute one for the other")
PROFILE_CLASSES = ("conv", "wgrad", "edge_expand", "edge_reduce", "edge_wgrad", "l1_loss", "adam", "clamp_bwd", "plane_amax")
MATH_PRODUCTS = {"bf16x6": 6, "f16x3": 3}   # 16-bit MFMAs per fp32 product
STEP_BYTES = {("dn", True): 117020.0, ("dn", False): 33420.0, ("sr", True): 124224.0, ("sr", False): 35476.0}   # SURVEY 8(d), per LR pixel
STEP_FLOP = {("dn", True): 2.62e12, ("dn", False): 8.749e11, ("sr", True): 2.74e12, ("sr", False): 9.140e11}   # per tile


def roofline_block(math, prof, batch, kind, train, world, tiles_per_s, shipped_width=True, sustained=None, tile=TILE):
    """roofline of the dominant kernel (the conv: forward + input-gradient launches) from the HIP-event records of the timed
    region; `prof` = {0: conv totals, 1: weight-gradient totals} (Engine.profile_read)."""
    k = prof[0]
    sec = k["ms"] * 1e-3
    tf = k["flop"] / sec / 1e12
    gbs = k["bytes"] / sec / 1e9
    pmc_ok = train and kind == "dn" and world == 1 and shipped_width and tile == TILE
    traffic, tfile = pmc_traffic(math, batch, "conv") if pmc_ok else (None, None)
    hb = {"achieved": gbs, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBPS}
    if math == "fp32":   # exact fp32 MFMA: compute-bound by 4-5x
        roof = {"bound": "mfma", "kernel": MATHS[math][1], "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / FP32_MFMA_PEAK_TFLOPS}
    else:                # split modes: `nprod` 16-bit MFMAs per fp32 product -> effective matrix peak 2500 / nprod TFLOP/s
        nprod = MATH_PRODUCTS[math]
        eff_peak = BF16_MFMA_PEAK_TFLOPS / nprod
        roof = {"bound": "mfma", "kernel": MATHS[math][1], "achieved": tf, "peak": eff_peak,
                "unit": f"TFLOP/s (algorithmic fp32 FLOP; {nprod} {'fp16' if math == 'f16x3' else 'bf16'} MFMAs each)", "frac": tf / eff_peak}
    if sustained is not None and math in MATH_PRODUCTS:
        sp = sustained["mfma_tflops"] / MATH_PRODUCTS[math]
        roof.update({"sustained_peak": sp, "frac_of_sustained": tf / sp,
                     "sustained_from": "xsd_probe_mfma_stream in this process on this device after the timed region: %.1f s of the conv's bare "
                                       "%s MFMA-wave stream (no staging, no global traffic) reached %.0f dense TFLOP/s at %.0f MHz in-kernel clock = "
                                       "%.3f of the nominal %.0f; / %d products per multiply"
                                       % (sustained["seconds"], "fp16" if math == "f16x3" else "bf16", sustained["mfma_tflops"], 1e3 * sustained["sclk_ghz"],
                                          sustained["mfma_tflops"] / BF16_MFMA_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS, MATH_PRODUCTS[math])})
    roof.update({"traffic": traffic, "traffic_from": (tfile + " (committed PMC passes of this workload; not measured in this run)") if tfile else None,
                 "hbm": hb, "launches": k["launches"], "avg_launch_ms": k["ms"] / k["launches"],
                 "algorithmic_bytes_per_launch": k["bytes"] / k["launches"], "algorithmic_flop_per_launch": k["flop"] / k["launches"]})
    if prof[1]["launches"] > 0:
        w = prof[1]
        wsec = w["ms"] * 1e-3
        wt, wfile = pmc_traffic(math, batch, "wgrad") if pmc_ok else (None, None)
        roof["wgrad_kernel"] = {"achieved_TFLOPs": w["flop"] / wsec / 1e12, "achieved_GBps": w["bytes"] / wsec / 1e9,
                                "launches": w["launches"], "avg_launch_ms": w["ms"] / w["launches"], "traffic": wt, "traffic_from": wfile}
    # the HBM-bound kernels of the step (profile classes 2..): algorithmic bytes / HIP-event time per kernel against 8 TB/s
    edge = {}
    for k in range(2, len(PROFILE_CLASSES)):
        r = prof.get(k)
        if r and r["launches"] > 0 and r["ms"] > 0:
            g = r["bytes"] / (r["ms"] * 1e-3) / 1e9
            edge[PROFILE_CLASSES[k]] = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": g / HBM_PEAK_GBPS,
                                        "launches": r["launches"], "avg_launch_us": 1e3 * r["ms"] / r["launches"],
                                        "algorithmic_bytes_per_launch": r["bytes"] / r["launches"]}
    if edge:
        tot_ms = sum(prof[k]["ms"] for k in prof)
        edge["share_of_profiled_kernel_time"] = sum(prof[k]["ms"] for k in range(2, len(PROFILE_CLASSES)) if k in prof) / tot_ms if tot_ms > 0 else None
        roof["edge"] = edge
    if not shipped_width:
        return roof
    # whole-step algorithmic bytes / flops per tile (fp32 counting rule) x tiles/s against the peaks
    step_bytes = STEP_BYTES[(kind, train)] * tile * tile
    step_flop = STEP_FLOP[(kind, train)] * (tile * tile) / float(TILE * TILE)      # the table is per 512 x 512 tile
    per_gpu = tiles_per_s / world
    roof["whole_step"] = {"algorithmic_GBps": step_bytes * per_gpu / 1e9, "hbm_frac": step_bytes * per_gpu / 1e9 / HBM_PEAK_GBPS,
                          "algorithmic_TFLOPs": step_flop * per_gpu / 1e12, "fp32_mfma_frac": step_flop * per_gpu / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                          "note": "per GPU; SURVEY.md 8(d) bytes/flops per tile x tiles/s"}
    return roof


def psnr_delta_vs_reference(kind, math, dev):
    """PSNR(engine output, target) - PSNR(reference output, target) on the two example_data tiles of tests/golden/example_data.npz
    (416 x 416 LR; SR: 832 x 832 HR), identical seeded weights through both: the reference's figures were written into the
    fixture by tests/golden/make_golden.py from the imported reference modules (models/model.py:48-49 over the prepared tiles of
    data/dataset.py:41-47); the engine runs here in `math`.  No trained weights exist offline, so this is
    the parity of the two implementations on real count tiles, not a quality figure."""
    import numpy as np
    import gen_common as gc
    from xmm_superres_denoise.data.tools import load_and_prepare
    from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
    z = np.load(os.path.join(ROOT, "tests", "golden", "example_data.npz"))
    masks = {}
    for tag in ("1x", "2x"):
        shp = z[f"mask{tag}_shape"]
        masks[tag] = torch.from_numpy(np.unpackbits(z[f"mask{tag}_bits"])[: int(np.prod(shp))].reshape(shp)).to(dev)
    if kind == "dn":
        state = gc.make_state("dn", 32, 4, 1234)
        m = GeneratorRRDB_DN(1, 1, 32, 4)
    else:
        state = gc.make_state("sr", 32, 4, 4321, last_bias=0.05)
        m = GeneratorRRDB_SR(1, 1, 32, 4, num_upsample=1)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    m = m.to(dev)
    m.set_math(math)
    deltas, refs = [], []
    with torch.no_grad():
        for i in range(2):
            if kind == "dn":
                x = load_and_prepare(torch.from_numpy(z[f"dn_counts20_{i}"]).to(dev)[None], masks["1x"], 416, 0.0022336, "sqrt")
                t = load_and_prepare(torch.from_numpy(z[f"dn_counts50_{i}"]).to(dev)[None], masks["1x"], 416, 0.0022336, "sqrt")
            else:
                x = load_and_prepare(torch.from_numpy(z[f"sr_counts_lr_{i}"]).to(dev)[None], masks["1x"], 416, 0.0022336, "sqrt")
                t = load_and_prepare(torch.from_numpy(z[f"sr_counts_hr_{i}"]).to(dev)[None], masks["2x"], 832, 0.0005584, "sqrt")
            y = m(x)
            mse = float(((y.double() - t.double()) ** 2).mean())
            ref = float(z[f"{kind}_psnr_{i}"][0])
            refs.append(ref)
            deltas.append(-10.0 * float(np.log10(mse)) - ref)
    del m
    worst = max(deltas, key=abs)
    return {"psnr_delta_db": worst,
            "psnr": {"delta_db_per_tile": deltas, "reference_db": refs, "math": math, "bar_db": 0.01,
                     "tiles": "2 example_data tiles (real 20 ks counts; %s), detector mask * pad 416 * sqrt-normalize, seeded weights" %
                              ("target 50 ks" if kind == "dn" else "target 100 ks sim at 2x, 832 x 832"),
                     "reference_from": "tests/golden/example_data.npz (written from the imported reference by tests/golden/make_golden.py)"}}


def self_launch(args) -> int:
    """--gpus N > 1 outside torchrun: start the N ranks as children of torch.distributed.run and relay their output.
    This parent never initialises the GPU (no torch.cuda call, no HIP library loaded) and does not exec."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def default_steps(batch, tile, train, math):
    """K when --steps is not given: ~5 s of timed region, from the full-batch step time (255 ms for 32 x 512^2 in f16x3) scaled by the
    work; never fewer than 20 (the bench batch's K).  A timed region of under a second starts before the package has settled at its
    1400 W cap and flatters the line (docs/LAB_NOTEBOOK.md R6.16).  A function of the arguments only: every rank computes the same K."""
    est_ms = 255.0 * (batch * tile * tile) / (32.0 * 512 * 512) * (1.0 if train else 0.35) * {"bf16x6": 1.55, "fp32": 2.7}.get(math, 1.0)
    return max(20, min(5000, int(5000.0 / est_ms)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0,
                    help="timed steps K (default: 20 at the bench batch, more for smaller work so that the timed region is ~5 s: a region of "
                         "under a second starts before the package has settled at its power cap and flatters the line)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps W (default: max(5, K // 4))")
    ap.add_argument("--workload", default="dn_train", choices=["dn_train", "sr_train", "dn_fwd", "sr_fwd"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 32 at every N; sr_fwd: 16)")
    ap.add_argument("--tile", type=int, default=TILE,
                    help="LR tile side in pixels (default 512 = BASELINE's metric).  416 = the reference's own operating point "
                         "(res/baseline_config.toml:36 lr res, 832 HR); any other value is NOT the headline: the metric names the size, "
                         "cpu_baseline is timed at that size, no PMC traffic")
    ap.add_argument("--math", default=os.environ.get("XSD_MATH", DEFAULT_MATH), choices=sorted(MATHS),
                    help="MFMA math mode of the conv kernels (include/xsd.h: xsd_set_math); the headline must be fp32-class")
    ap.add_argument("--input-pipeline", action="store_true",
                    help="configs[4]: each step starts from int32 count tiles (411x403, Poisson, seed 2) and runs the fused "
                         "detector-mask * pad * sqrt-normalize kernel on the GPU instead of reusing resident float tiles")
    ap.add_argument("--loss", default="l1", choices=["l1", "paper"],
                    help="l1 = BASELINE configs[2]; paper = the reference's shipped default 0.5 psnr + 0.5 ms_ssim with the "
                         "'linear' scaling table (res/configs/loss_functions.toml), reported separately (SURVEY 8d config 3)")
    ap.add_argument("--filters", type=int, default=32,
                    help="num_filters of the generator (BASELINE: 32 = res/configs/models.toml).  Other widths are NOT the headline: "
                         "the line then carries no cpu_baseline, no PMC traffic and no whole-step figures (those are defined for 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--extra-math", default="bf16x6", choices=sorted(MATHS) + ["none"],
                    help="second, labelled measurement in another math mode (same --steps/--warmup, own roofline; never the headline)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra-math leg")
    ap.add_argument("--no-sustained", action="store_true", help="skip roofline.sustained_peak (the in-process MFMA-stream probe)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0)
    ap.add_argument("--no-psnr", action="store_true", help="skip psnr_delta_db (example_data tiles through the engine, outside the timed region)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    # one process per GPU; XSD_DIST_BACKEND=gloo lets the multi-process path be rehearsed on a single-GPU box
    backend = os.environ.get("XSD_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank (XSD_DIST_BACKEND=gloo rehearses the "
                         "multi-rank path on fewer GPUs)")
    dev_index = local_rank % ndev if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # XSD_FORCE_DP=1 with one rank: a one-rank process group, every collective line of this file and of parallel.py executes
    # (tests/test_hip_parallel.py runs it over RCCL on the single MI355X -- the code a multi-GPU node takes)
    dp = world > 1 or os.environ.get("XSD_FORCE_DP", "0") == "1"
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
    from xmm_superres_denoise.parallel import DataParallelTrainer

    trace_on = bool(os.environ.get("XSD_BENCH_TRACE"))

    def trace(msg):      # progress markers of every rank on stderr (diagnosing a multi-rank run that does not come back)
        if trace_on:
            print(f"[bench rank {rank}/{world} +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    trace("process group up")
    kind, mode = args.workload.split("_")
    train = mode == "train"
    B = args.batch or (16 if args.workload == "sr_fwd" else 32)
    scale = 2 if kind == "sr" else 1
    T = args.tile
    if T < 16 or T > 2048:
        raise SystemExit("--tile must be in [16, 2048]")
    if args.input_pipeline and T < 411:
        raise SystemExit("--input-pipeline pads 411 x 403 count tiles: --tile must be >= 411")
    if args.steps <= 0:
        args.steps = default_steps(B, T, train, args.math)
    if args.warmup < 0:
        args.warmup = max(5, args.steps // 4)

    torch.manual_seed(0)  # same seeded default init on every rank (DP replicas start identical)
    NF = args.filters
    model = (GeneratorRRDB_DN(1, 1, NF, 4) if kind == "dn" else GeneratorRRDB_SR(1, 1, NF, 4, num_upsample=1)).to(dev)
    # synthetic data: global batch generated from one seed, each rank takes its shard (DistributedSampler analogue)
    gx = torch.Generator().manual_seed(0)
    gt = torch.Generator().manual_seed(1)
    x = torch.rand((B * world, 1, T, T), generator=gx)[rank * B:(rank + 1) * B].contiguous().to(dev)
    tgt = None
    if train:
        tgt = torch.rand((B * world, 1, T * scale, T * scale), generator=gt)[rank * B:(rank + 1) * B].contiguous().to(dev)

    model.set_math(args.math)
    loss_fn = None
    if args.loss == "paper":
        from xmm_superres_denoise.utils import create_loss, load_loss_config
        loss_fn = create_loss(*load_loss_config("linear"))
    trainer = DataParallelTrainer(model, lr=1e-4, betas=(0.9, 0.999), loss=loss_fn)
    eng = trainer.engine
    run_math = eng.get_math()     # what the kernels compute in: widths beyond the plane kernels (> 256 filters) run exact fp32 whatever --math says
    if run_math != args.math:
        print(f"bench.py: --filters {NF} runs on the exact-fp32 kernels; labelling this run '{run_math}', not '{args.math}'", file=sys.stderr)
        args.math = run_math
    trace("trainer built (parameters broadcast)")

    counts = mask = None
    if args.input_pipeline:
        import numpy as np
        from xmm_superres_denoise.engine import compose_input
        rngc = np.random.default_rng(2 + rank)
        counts = torch.from_numpy(rngc.poisson(0.1, size=(B, 411, 403)).astype(np.int32)).to(dev)
        gold = os.path.join(ROOT, "tests", "golden", "example_data.npz")
        z = np.load(gold)
        m1 = np.unpackbits(z["mask1x_bits"])[: int(np.prod(z["mask1x_shape"]))].reshape(z["mask1x_shape"])
        mask = torch.from_numpy(m1).to(dev)

    def step():
        if args.input_pipeline:
            xin = compose_input(counts, None, None, mask, T, 0.0022336, "sqrt")  # 411x403 -> centred pad to the tile (512; 416 = data/tools.py:103-126)
            if train:
                return trainer.train_step(xin, tgt)
            with torch.no_grad():
                return model(xin)
        if train:
            return trainer.train_step(x, tgt)
        with torch.no_grad():
            return model(x)

    def timed(nwarm, nsteps, profile, watch=None):
        """-> (max-over-ranks seconds, this rank's own seconds, profile records, exposed communication ms per step on this rank)"""
        for i in range(nwarm):
            step()
            trace(f"warm-up step {i} enqueued")
        if profile:
            eng.profile_enable(True)
        if dp:
            dist.barrier()
        torch.cuda.synchronize()
        if train:
            trainer.comm_events_begin()       # event pair around the wait for the gradient exchange, read after the timed region
        if watch is not None:
            watch.start()
        trace("timed region starts")
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0       # this rank's own K steps (before the closing barrier): the per-rank figure
        trace("timed region done")
        if dp:
            dist.barrier()
        dt = time.perf_counter() - t0
        if watch is not None:
            watch.stop()
        if dp:
            tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        comm_ms = trainer.comm_events_end() / nsteps if train else None
        prof = None
        if profile:
            prof = {k: eng.profile_read(k) for k in range(len(PROFILE_CLASSES))}
            eng.profile_enable(False)
        return dt, mine, prof, comm_ms

    # every rank watches ITS OWN device (by PCI address): the per-rank block below carries each device's clock and watts
    my_pci = device_pci(dev.index if hasattr(dev, "index") and dev.index is not None else 0)
    pwatch = PowerWatch(pci=[my_pci] if my_pci else None)
    dt, my_dt, prof, comm_ms = timed(args.warmup, args.steps, not args.no_profile, pwatch)
    power = pwatch.summary()
    # The headline's timed region carries the per-launch HIP events the roofline block is computed from (two hipEventRecord per
    # MFMA-kernel launch: ~0.2 % of a batch-32 step, ~10 % of a batch-1 step).  The same K steps once more WITHOUT them, reported
    # beside the headline as `unprofiled` -- what a user's step costs
    dt_plain = None
    if not args.no_profile:
        dt_plain, _, _, _ = timed(0, args.steps, False)

    # the rate the matrix pipes of THIS device sustain on the conv's bare MFMA stream, measured now, in this process
    # (include/xsd.h: xsd_probe_mfma_stream): the roof the package power cap leaves of the nominal 2.5 PFLOP/s
    sustained = None
    if args.math in MATH_PRODUCTS and not args.no_sustained:
        fmt = "f16" if args.math == "f16x3" else "bf16"
        if dp and world > ndev:      # a rehearsal with ranks SHARING a device: one rank at a time, or per_rank.sustained_mfma_tflops is a contended figure that reads as device spread
            for r in range(world):
                if r == rank:
                    sustained = eng.probe_mfma_stream(fmt, args.sustained_seconds)
                dist.barrier()
        else:
            sustained = eng.probe_mfma_stream(fmt, args.sustained_seconds)

    per_rank = None
    if dp:      # what a < N x curve is made of: every rank's own step time, device clock and watts, exposed exchange wait
        mine_rec = {"ms_per_step": round(1e3 * my_dt / args.steps, 3), "sclk_mhz": power["sclk_mhz"] if power else None,
                    "avg_w": power["avg_w"] if power else None, "pci": my_pci,
                    "comm_ms_exposed": None if comm_ms is None else round(comm_ms, 4),
                    "sustained_mfma_tflops": None if sustained is None else round(sustained["mfma_tflops"], 1)}
        recs = [None] * world
        dist.all_gather_object(recs, mine_rec)
        per_rank = {k: [r[k] for r in recs] for k in mine_rec}

    replicas_identical = None
    if dp and train:     # DDP invariant, checked outside the timed region: every rank holds bit-identical parameters
        bits = trainer.flat.view(torch.int32).to(torch.int64)
        h = torch.stack([bits.sum(), (bits * torch.arange(1, bits.numel() + 1, device=dev, dtype=torch.int64)).sum()])   # (wraps: fine, it is a hash)
        hs = [torch.zeros_like(h) for _ in range(world)]
        if backend == "nccl":
            dist.all_gather(hs, h)
        else:
            hc = [t.cpu() for t in hs]
            dist.all_gather(hc, h.cpu())
            hs = hc
        replicas_identical = all(bool(torch.equal(t.cpu(), hs[0].cpu())) for t in hs)

    # ---- the second half of BASELINE's metric ("PSNR delta vs ref"), outside the timed region, in the mode the line is timed in
    psnr_delta = None
    if rank == 0 and NF == 32 and not args.no_psnr:
        psnr_delta = psnr_delta_vs_reference(kind, args.math, dev)
        # north_star names the XMM-SuperRes PSNR: every line carries BOTH generators' figures (psnr.sr_delta_db, psnr.dn_delta_db);
        # the top-level psnr_delta_db stays the timed workload's kind
        other = "sr" if kind == "dn" else "dn"
        o = psnr_delta_vs_reference(other, args.math, dev)
        psnr_delta["psnr"][kind + "_delta_db"] = psnr_delta["psnr_delta_db"]
        psnr_delta["psnr"][other + "_delta_db"] = o["psnr_delta_db"]
        psnr_delta["psnr"][other] = o["psnr"]

    # ---- extra leg (same process, same inputs): another math mode over the same --steps/--warmup, labelled; never the headline
    extra = None
    xm = None if (args.no_extra or world > 1 or args.extra_math in ("none", args.math)) else args.extra_math   # N=1 only
    if xm is not None:
        with torch.no_grad():
            y_head = model(x[:2]).clone()
        model.set_math(xm)
        with torch.no_grad():
            y_x = model(x[:2])
        err = float((y_x - y_head).abs().max())
        xm = eng.get_math()       # what the kernels really compute in (generic widths run exact fp32 whatever was asked for)
        pwx = PowerWatch(pci=[my_pci] if my_pci else None)
        dte, _, profe, _ = timed(args.warmup, args.steps, not args.no_profile, pwx)
        extra = {"math": xm, "dtype": MATHS[xm][0], "value": B * world * args.steps / dte, "unit": "tiles/s",
                 "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dte / args.steps,
                 "max_abs_output_diff_vs_" + args.math: err, "note": "not the headline: the strict mode, reported for comparison"}
        if pwx.summary() is not None:
            extra["power"] = pwx.summary()
            extra["power"]["joules_per_tile"] = round(extra["power"]["avg_w"] * (dte / args.steps) / B, 3)
        if profe is not None and profe[0]["launches"] > 0:
            xs = eng.probe_mfma_stream("f16" if xm == "f16x3" else "bf16", args.sustained_seconds) if (xm in MATH_PRODUCTS and not args.no_sustained) else None
            extra["roofline"] = roofline_block(xm, profe, B, kind, train, world, B * world * args.steps / dte, NF == 32, xs, T)
        model.set_math(args.math)

    if rank == 0:
        tiles = B * world * args.steps
        dtype, kname = MATHS[args.math]
        out = {
            "metric": f"XMM {T}x{T} tiles/sec (train step)" if train else f"XMM {T}x{T} tiles/sec (forward)",
            "value": tiles / dt, "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": {"dn_train": "XMM-DeNoise train step (%s + Adam), fwd+bwd HIP kernels" % ("L1" if args.loss == "l1" else "0.5 PSNR + 0.5 MS-SSIM"),
                                    "sr_train": "XMM-SuperRes 2x train step (%s + Adam)" % ("L1" if args.loss == "l1" else "0.5 PSNR + 0.5 MS-SSIM"),
                                    "dn_fwd": "XMM-DeNoise forward", "sr_fwd": "XMM-SuperRes 2x generator forward"}[args.workload],
                       "tile": f"1x{T}x{T}", "per_gpu_batch": B, "global_batch": B * world,
                       "layers": "RRDB generator, 32 filters x 4 blocks (res/configs/models.toml)" if NF == 32 else f"RRDB generator, {NF} filters x 4 blocks (NOT the BASELINE width)",
                       "math": args.math,
                       "parallelism": f"dp{world}", "dist_backend": backend if dp else None},
        }
        if dt_plain is not None:
            out["unprofiled"] = {"value": tiles / dt_plain, "ms_per_step": 1e3 * dt_plain / args.steps,
                                 "note": "the same K steps again without the per-launch HIP events of the roofline block"}
        if NF % 32:
            out["config"]["runs_zero_padded_to_filters"] = (NF + 31) // 32 * 32      # the kernels' flop / byte counts below are those of the padded width
        if replicas_identical is not None:
            out["replicas_identical"] = replicas_identical
        if prof is not None and prof[0]["launches"] > 0:
            out["roofline"] = roofline_block(args.math, prof, B, kind, train, world, tiles / dt, NF == 32, sustained, T)
        if power is not None:
            # the step runs at the package cap, so the rate IS watts / joules per tile: the figure a kernel change has to move
            power["joules_per_tile"] = round(power["avg_w"] * (dt / args.steps) / B, 3)
            out["power"] = power
        if train:
            out["comm_ms_exposed"] = None if comm_ms is None else round(comm_ms, 4)     # rank 0's; every rank's in per_rank
        if per_rank is not None:
            out["per_rank"] = per_rank
        if psnr_delta is not None:
            out.update(psnr_delta)
        if extra is not None:
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline and NF == 32:
            out["cpu_baseline"] = cpu_baseline(kind, train, tile=T)      # the same step at the same tile size on the host cores (B = 1)
        print(json.dumps(out), flush=True)
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
