# bf16x6 conv: parity (kernel / precision / network tests on the default build), phase stamps (diagnostic build), then A/B of the two production builds
set -e
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_precision.py tests/test_hip_network.py -x -q -m gpu > gpurun_out/rs_pytest.log 2>&1 || { tail -30 gpurun_out/rs_pytest.log; exit 1; }
tail -2 gpurun_out/rs_pytest.log
XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_diag.so timeout -k 10 200 python tools/stamps.py bf16x6 32 2>&1 | grep -v amdgpu.ids
bash tools/ab_libs.sh
