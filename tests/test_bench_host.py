"""Host-side pieces of bench.py that need no GPU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_card(root, name, watts, mhz, cap=1400):
    h = os.path.join(root, name, "device", "hwmon", "hwmon0")
    os.makedirs(h)
    for f, v in (("power1_input", int(watts * 1e6)), ("freq1_input", int(mhz * 1e6)), ("power1_cap", int(cap * 1e6))):
        with open(os.path.join(h, f), "w") as fh:
            fh.write("%d\n" % v)
    return h


def test_power_watch_reads_the_busiest_device(tmp_path):
    """bench.py's `power` block: sysfs hwmon files in microwatts / hertz, the busiest device wins, the first fifth of the
    samples (ramp from idle) is dropped, the fraction of the cap is reported."""
    import bench
    _fake_card(str(tmp_path), "card0", 240, 100)
    _fake_card(str(tmp_path), "card1", 1395, 1650)
    with bench.PowerWatch(0.01, root=str(tmp_path)) as pw:
        time.sleep(0.3)
    s = pw.summary()
    assert s is not None and s["cap_w"] == 1400.0 and abs(s["avg_w"] - 1395.0) < 0.5 and s["sclk_mhz"] == 1650.0
    assert abs(s["frac_of_cap"] - 1395.0 / 1400.0) < 1e-3 and s["samples"] >= 3


def test_power_watch_without_devices_reports_nothing(tmp_path):
    import bench
    with bench.PowerWatch(0.01, root=str(tmp_path)) as pw:
        time.sleep(0.05)
    assert pw.summary() is None


def test_power_watch_lists_every_working_device_of_a_multi_gpu_run(tmp_path):
    import bench
    _fake_card(str(tmp_path), "card0", 240, 100)
    _fake_card(str(tmp_path), "card1", 1390, 1640)
    _fake_card(str(tmp_path), "card2", 1396, 1650)
    with bench.PowerWatch(0.01, root=str(tmp_path)) as pw:
        time.sleep(0.3)
    s = pw.summary()
    assert s["busy_devices_avg_w"] == [1396.0, 1390.0] and abs(s["avg_w"] - 1396.0) < 0.5
