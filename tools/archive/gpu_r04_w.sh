#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 400 python tools/archive/sweep_spmm_pair3.py > gpurun_out/r04u_spmm_pair3.log 2>&1; cat gpurun_out/r04u_spmm_pair3.log
for cfg in "spmm_tile_pair=1,spmm_tile_slide=27,spmm_tile_dbuf=0" "spmm_tile_pair=1,spmm_tile_slide=0,spmm_tile_dbuf=0"; do
  tag=$(echo $cfg | tr ',=' '__')
  echo "== $cfg"
  KHIP_OPTS=$cfg bash tools/archive/prof_spmm_pmc.sh w_${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum 2>&1 | grep -E "spmm_tile|spmm p=" | cut -c1-24,60-140
done
