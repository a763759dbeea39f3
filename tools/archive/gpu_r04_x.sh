#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_dist.py tests/test_gpu_scale_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 200 python tools/cfg5_only.py 2>&1 | tail -2
timeout 300 python tools/bench_configs.py 2>&1 | grep -E "cfg5|spmm" | cut -c1-200
timeout 300 python tools/cfg5_widths.py 2>&1 | tail -8 | cut -c1-200
