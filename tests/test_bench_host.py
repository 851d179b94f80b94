"""Host-side pieces of bench.py that need no GPU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_card(root, name, watts, mhz, cap=1400, pci=None):
    """card<N>/device is a symlink to the PCI device directory, as in sysfs"""
    pci = pci or ("0000:%02x:00.0" % (int(name[4:]) + 5))
    devdir = os.path.join(root, "_pci", pci)
    h = os.path.join(devdir, "hwmon", "hwmon0")
    os.makedirs(h)
    os.makedirs(os.path.join(root, name))
    os.symlink(devdir, os.path.join(root, name, "device"))
    for f, v in (("power1_input", int(watts * 1e6)), ("freq1_input", int(mhz * 1e6)), ("power1_cap", int(cap * 1e6))):
        with open(os.path.join(h, f), "w") as fh:
            fh.write("%d\n" % v)
    return pci


def test_power_watch_reads_the_busiest_device(tmp_path):
    """bench.py's `power` block: sysfs hwmon files in microwatts / hertz, the busiest device wins, every sample between
    start() and stop() counts (bench.py starts the reader after the warm-up's synchronize), the fraction of the cap is reported."""
    import bench
    _fake_card(str(tmp_path), "card0", 240, 100)
    _fake_card(str(tmp_path), "card1", 1395, 1650)
    with bench.PowerWatch(0.01, root=str(tmp_path)) as pw:
        time.sleep(0.3)
    s = pw.summary()
    assert s is not None and s["cap_w"] == 1400.0 and abs(s["avg_w"] - 1395.0) < 0.5 and s["sclk_mhz"] == 1650.0
    assert abs(s["frac_of_cap"] - 1395.0 / 1400.0) < 1e-3 and s["samples"] >= 3
    assert "first fifth" not in s["source"] and "warm-up" in s["source"]


def test_power_watch_counts_only_samples_between_start_and_stop(tmp_path):
    """an idle device before start() leaves no trace in the mean: the reader thread does not run until start()"""
    import bench
    pci = _fake_card(str(tmp_path), "card0", 200, 100)
    pw = bench.PowerWatch(0.01, root=str(tmp_path))
    time.sleep(0.1)                                          # 'engine creation + warm-up': not sampled
    h = os.path.join(str(tmp_path), "_pci", pci, "hwmon", "hwmon0")
    with open(os.path.join(h, "power1_input"), "w") as fh:
        fh.write("%d\n" % int(1398e6))
    pw.start()
    time.sleep(0.2)
    pw.stop()
    n = pw.summary()["samples"]
    time.sleep(0.05)
    s = pw.summary()
    assert s["samples"] == n and abs(s["avg_w"] - 1398.0) < 0.5


def test_power_watch_without_devices_reports_nothing(tmp_path):
    import bench
    with bench.PowerWatch(0.01, root=str(tmp_path)) as pw:
        time.sleep(0.05)
    assert pw.summary() is None


def test_power_watch_reports_this_jobs_devices_by_pci_address(tmp_path):
    """a GPU box shows the hwmon files of every card of its host: the job's own devices are picked by PCI address (another
    tenant's busier card is ignored), and a multi-GPU run lists every rank's device"""
    import bench
    _fake_card(str(tmp_path), "card0", 1399, 1700)                     # somebody else's
    a = _fake_card(str(tmp_path), "card1", 1390, 1640)
    b = _fake_card(str(tmp_path), "card2", 1396, 1650)
    with bench.PowerWatch(0.01, root=str(tmp_path), pci=[a[:10]]) as pw:
        time.sleep(0.3)
    s = pw.summary()
    assert s["pci"] == a and abs(s["avg_w"] - 1390.0) < 0.5 and "this job" in s["source"] and "devices_avg_w" not in s
    with bench.PowerWatch(0.01, root=str(tmp_path), pci=[a[:10], b[:10]]) as pw:
        time.sleep(0.3)
    s = pw.summary()
    assert s["devices_avg_w"] == [1396.0, 1390.0] and s["pci"] == b
    with bench.PowerWatch(0.01, root=str(tmp_path), pci=["0000:7f:00"]) as pw:     # no match: falls back to the busiest, and says so
        time.sleep(0.3)
    s = pw.summary()
    assert abs(s["avg_w"] - 1399.0) < 0.5 and "BUSIEST" in s["source"]


def test_scale_report_decomposes_an_efficiency(tmp_path):
    """tools/scale_report.py: efficiency = mean rank rate x slowest rank's share x barrier, from the per_rank block of bench.py's lines"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_report as sr
    one = {"metric": "m", "value": 128.0, "n_gpus": 1, "ms_per_step": 250.0, "comm_ms_exposed": 0.0}
    ms = [250.0, 255.0, 260.0, 270.0]
    four = {"metric": "m", "value": 4 * 32 / 0.271, "n_gpus": 4, "ms_per_step": 271.0,
            "per_rank": {"ms_per_step": ms, "sclk_mhz": [1700, 1650, 1620, 1580], "sustained_mfma_tflops": [1520, 1500, 1480, 1450],
                         "comm_ms_exposed": [0.1, 0.12, 0.2, 0.05]}}
    p1, p4 = os.path.join(tmp_path, "n1.json"), os.path.join(tmp_path, "n4.json")
    open(p1, "w").write("noise before\n" + json.dumps(one) + "\n")
    open(p4, "w").write(json.dumps({"run": {"stdout_tail": json.dumps(four) + "\n"}}))      # a driver record with the line inside a string
    by_n = sr.load([p1, p4])
    assert sorted(by_n) == [1, 4]
    text = sr.report(by_n)
    row = [l for l in text.splitlines() if l.strip().startswith("4 ")][0].split()
    eff, mean_rate, slow, barrier, comm = (float(v) for v in row[2:7])
    assert abs(eff - four["value"] / (4 * 128.0)) < 1e-3
    assert abs(mean_rate - 250.0 / (sum(ms) / 4)) < 1e-3 and abs(slow - (sum(ms) / 4) / 270.0) < 1e-3 and abs(barrier - 270.0 / 271.0) < 1e-3
    assert abs(eff - mean_rate * slow * barrier) < 2e-3 and comm == 0.2


def test_roofline_block_edge_kernels_and_tile_size():
    """bench.roofline_block (round 6): the HBM-bound kernels of the step -- profile classes 2.. of include/xsd.h -- each as algorithmic
    bytes / event time against 8 TB/s; at a tile size other than BASELINE's 512 the whole-step figures scale with the pixels and no
    committed PMC traffic is quoted (those files were collected at 512 x 512, batch 32)."""
    import bench
    px = 32 * 512 * 512
    prof = {k: {"ms": 0.0, "launches": 0, "flop": 0.0, "bytes": 0.0} for k in range(len(bench.PROFILE_CLASSES))}
    prof[0] = {"ms": 170.0, "launches": 122, "flop": 122 * 458.79e9, "bytes": 122 * 4.26e9}
    prof[1] = {"ms": 76.0, "launches": 13, "flop": 13 * 2.153e12, "bytes": 13 * 10.08e9}
    prof[2] = {"ms": 0.75, "launches": 2, "flop": 0.0, "bytes": 2 * px * 132.0}            # edge_expand: 2.95 TB/s
    prof[6] = {"ms": 0.013, "launches": 1, "flop": 0.0, "bytes": 28.0 * 1670657}            # Adam over the 1.67 M parameters
    r = bench.roofline_block("f16x3", prof, 32, "dn", True, 1, 126.0)
    assert r["bound"] == "mfma" and abs(r["frac"] - (122 * 458.79e9 / 0.170 / 1e12) / (2500.0 / 3)) < 1e-9
    e = r["edge"]
    assert set(e) == {"edge_expand", "adam", "share_of_profiled_kernel_time"}              # classes that did not run are not listed
    assert abs(e["edge_expand"]["achieved"] - 2 * px * 132.0 / 0.75e-3 / 1e9) < 1e-6 and e["edge_expand"]["peak"] == 8000.0
    assert abs(e["edge_expand"]["frac"] - e["edge_expand"]["achieved"] / 8000.0) < 1e-12 and abs(e["edge_expand"]["avg_launch_us"] - 375.0) < 1e-9
    assert abs(e["share_of_profiled_kernel_time"] - 0.763 / (170.0 + 76.0 + 0.763)) < 1e-9
    assert r["traffic_from"] is None or "profiles/" in r["traffic_from"]                    # 512 x 512, batch 32, DN train: the committed PMC file, if any
    w512 = r["whole_step"]
    r416 = bench.roofline_block("f16x3", prof, 32, "dn", True, 1, 126.0, True, None, 416)
    assert r416["traffic"] is None and r416["traffic_from"] is None                         # no PMC figure off the size it was collected at
    s = (416 * 416) / (512 * 512)
    assert abs(r416["whole_step"]["algorithmic_GBps"] - s * w512["algorithmic_GBps"]) < 1e-6
    assert abs(r416["whole_step"]["algorithmic_TFLOPs"] - s * w512["algorithmic_TFLOPs"]) < 1e-9
    other = bench.roofline_block("f16x3", prof, 32, "dn", True, 1, 126.0, False)           # another width: no whole-step figures, the edge block stays
    assert "whole_step" not in other and "edge" in other


def test_default_steps_give_a_settled_timed_region():
    """bench.default_steps (round 6, docs/LAB_NOTEBOOK.md R6.16): without --steps the bench batch runs the driver's 20 steps, smaller work
    runs more so that the timed region stays ~5 s (a region of under a second is timed before the package settles at its power cap);
    the count depends on the arguments only, so every rank of a multi-GPU run computes the same K."""
    import bench
    assert bench.default_steps(32, 512, True, "f16x3") == 20
    assert bench.default_steps(32, 512, True, "bf16x6") == 20 and bench.default_steps(64, 512, True, "f16x3") == 20
    k = bench.default_steps(1, 416, False, "f16x3")          # one 416 x 416 image per call: ~2.4 ms each
    assert 1500 <= k <= 5000
    k4 = bench.default_steps(4, 416, True, "f16x3")          # the reference's training batch of four: ~22 ms per step
    assert 150 <= k4 <= 300 and 3.0 <= k4 * 0.0217 <= 6.5
    assert bench.default_steps(1, 16, False, "f16x3") == 5000
