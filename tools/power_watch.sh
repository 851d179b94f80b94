# samples rocm-smi power / clocks while the default train bench runs: is the step power-limited?  usage: bash tools/power_watch.sh [lib]
L=${1:-libxsd_hip.so}
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' '|'; echo; sleep 0.5; done ) > gpurun_out/power_samples.txt &
SP=$!
XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/$L timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/power_bench.log 2>&1
wait $SP
grep "^{" gpurun_out/power_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['ms_per_step'])"
cat gpurun_out/power_samples.txt
rocm-smi --showmaxpower 2>/dev/null | grep -i power
