#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 120 python tools/archive/sweep_spmm_slide.py --small > gpurun_out/r04d_spmm_slide_small.log 2>&1; echo "small exit $?"; cut -c1-250 gpurun_out/r04d_spmm_slide_small.log
timeout 600 python -m pytest tests/test_gpu_block.py -x -q -k "spmm" > gpurun_out/r04d_pytest_spmm.log 2>&1; echo "pytest spmm exit $?"; tail -5 gpurun_out/r04d_pytest_spmm.log
timeout 400 python tools/archive/sweep_spmm_slide.py > gpurun_out/r04d_spmm_slide.log 2>&1; echo "sweep exit $?"; cut -c1-300 gpurun_out/r04d_spmm_slide.log
