#!/bin/bash
# A/B of the tile-SpMM group order under rocprofv3 counters (separate --pmc passes, no trace domains)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03e}
{
for opts in "spmm_tile_pencil=2" "spmm_tile_pencil=54,spmm_tile_exp=8" "spmm_tile_pencil=2,spmm_tile_exp=16"; do
  echo "## KHIP_OPTS=$opts"
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
    t=$(echo $set | cut -d' ' -f1)
    KHIP_OPTS=$opts bash tools/archive/prof_spmm_pmc.sh ${TAG}_$t $set 2>&1 | grep -v "^W2\|^E2\|^I2" | grep "spmm_tile_kernel\|spmm p=" | cut -c1-20,60-140
  done
done
} > gpurun_out/${TAG}_spmm_tile_pmc_ab.log 2>&1
cat gpurun_out/${TAG}_spmm_tile_pmc_ab.log
