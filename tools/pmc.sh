cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  if [ $i = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; fi
  if [ $i = 2 ]; then C="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS"; fi
  rm -rf $R/gpurun_out/pmc_$1_$i
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$1_$i -- python3 $R/bench.py --steps 1 --warmup 1 --batch ${PMC_BATCH:-4} --no-cpu-baseline --no-profile --no-extra --no-sustained --no-psnr --math $1 --workload ${PMC_WORKLOAD:-dn_fwd} > $R/gpurun_out/pmc_$1_$i.log 2>&1 || exit 1
done
echo ok
