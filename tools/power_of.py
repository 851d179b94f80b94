"""Package power and shader clock of THIS job's device (sysfs hwmon, bench.py's PowerWatch; the device is named by the PCI address
`rocm-smi --showbus` reports -- no GPU call in this process, the command runs as a child) while a command runs:
python tools/power_of.py <command ...>.  The first fifth of the samples (ramp from idle) is dropped."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PowerWatch


def my_pci():
    try:
        out = subprocess.run(["rocm-smi", "--showbus"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=30).stdout
        return [m.group(1).lower()[:10] for m in re.finditer(r"PCI Bus: ([0-9A-Fa-f]{4}:[0-9A-Fa-f]{2}:[0-9A-Fa-f]{2})", out)] or None
    except Exception:
        return None


if __name__ == "__main__":
    with PowerWatch(0.05, pci=my_pci()) as pw:
        rc = subprocess.run(sys.argv[1:]).returncode
    print("power:", pw.summary())
    sys.exit(rc)
