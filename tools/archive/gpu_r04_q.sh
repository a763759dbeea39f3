#!/bin/bash
# TCC hit / miss / EA requests and LDS bank conflicts of the tile SpMM at cfg 5 with and without sliding windows (separate --pmc passes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for cfg in "spmm_tile_slide=0" "spmm_tile_slide=1" "spmm_tile_slide=14" "spmm_tile_slide=1,spmm_tile_shape=4" "spmm_tile_slide=0,spmm_tile_shape=4"; do
  tag=$(echo $cfg | tr ',=' '__')
  echo "== $cfg"
  KHIP_OPTS=$cfg bash tools/archive/prof_spmm_pmc.sh q_${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum 2>&1 | grep -E "spmm_tile_kernel|spmm p=" | cut -c1-20,60-140
  KHIP_OPTS=$cfg bash tools/archive/prof_spmm_pmc.sh q_${tag}_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS 2>&1 | grep -E "spmm_tile_kernel" | cut -c1-20,60-140
done
