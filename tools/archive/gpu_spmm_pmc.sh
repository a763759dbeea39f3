#!/bin/bash
# rocprofv3 counter passes over the SpMM window kernel (p = 16, 27-point 216^3): separate --pmc runs, no trace domains
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02}
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  t=$(echo $set | cut -d' ' -f1)
  bash tools/archive/prof_spmm_pmc.sh ${TAG}_$t $set 2>&1 | grep -v "^W2\|^E2\|^I2" | tail -12
done
