"""Package power and shader clock (sysfs hwmon of the busiest amdgpu device, bench.py's PowerWatch) while a command runs:
python tools/power_of.py <command ...>.  The first fifth of the samples (ramp from idle) is dropped."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PowerWatch
with PowerWatch(0.05) as pw:
    rc = subprocess.run(sys.argv[1:]).returncode
print("power:", pw.summary())
sys.exit(rc)
