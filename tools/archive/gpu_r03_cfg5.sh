#!/bin/bash
# round 3 evidence for cfg 5: rocprofv3 kernel stats of 25 block-GMRES iterations, counter passes of the SpMM kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; TAG=${1:-r03}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/cfg5_only.py > gpurun_out/${TAG}_cfg5_plain.log 2>&1; cat gpurun_out/${TAG}_cfg5_plain.log | tail -2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cfg5 -o c -- python $R/tools/cfg5_only.py > $R/gpurun_out/prof_${TAG}_cfg5.log 2>&1; echo "cfg5 stats exit $?"
cd $R
cut -c1-160 gpurun_out/prof_${TAG}_cfg5/c_kernel_stats.csv | head -14
{
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  t=$(echo $set | cut -d' ' -f1)
  bash tools/archive/prof_spmm_pmc.sh ${TAG}_$t $set 2>&1 | grep -v "^W2\|^E2\|^I2" | grep "spmm_tile_kernel\|spmm p=" | cut -c1-20,60-140
done
} > gpurun_out/${TAG}_spmm_tile_pmc.log 2>&1
cat gpurun_out/${TAG}_spmm_tile_pmc.log
