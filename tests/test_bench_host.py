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
