#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/archive/sweep_headline.py "" "spmv_nty=2" "spmv_nty=1" "spmv_nty=2,spmv_tiles=1" > gpurun_out/r04l_sweep_nty.log 2>&1; cat gpurun_out/r04l_sweep_nty.log
EXPS="0,32,0,32" PENCILS="4" timeout 200 python tools/archive/sweep_spmm_tile.py 216 > gpurun_out/r04l_spmm_sc1.log 2>&1; cut -c1-120 gpurun_out/r04l_spmm_sc1.log
