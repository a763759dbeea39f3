#!/bin/bash
# round 3: tile SpMM experiments + rocprofv3 counter passes (separate --pmc runs, no trace domains)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03c}
SHAPES=0 timeout 400 python tools/archive/sweep_spmm_tile.py 216 > gpurun_out/${TAG}_spmm_tile_sweep.log 2>&1
cut -c1-70 gpurun_out/${TAG}_spmm_tile_sweep.log
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TCP\|TA\|TCC\|TD\)_[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/${TAG}_counters_avail.txt
{
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_LEVEL_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  t=$(echo $set | cut -d' ' -f1)
  bash tools/archive/prof_spmm_pmc.sh ${TAG}_$t $set 2>&1 | grep -v "^W2\|^E2\|^I2" | grep "spmm_tile_kernel\|exit" | cut -c1-20,60-140
done
} > gpurun_out/${TAG}_spmm_tile_pmc.log 2>&1
cat gpurun_out/${TAG}_spmm_tile_pmc.log
