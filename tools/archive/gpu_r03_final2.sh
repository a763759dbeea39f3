#!/bin/bash
# round 3, after the last panel changes (one partial tile per workgroup, pipelined tile reduction, deflating panel QR): the panel /
# parity tests, bench.py with the committed counter file, kernel stats of cfg 5, the configs and the irregular operators.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/bench_configs.jsonl gpurun_out/bench_irregular.jsonl
timeout 400 python -m pytest tests/test_gpu_block.py tests/test_gpu_scale_parity.py tests/test_gpu_dist.py -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python bench.py --steps 100 --warmup 10 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 300 gpurun_out/r03_bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_cfg5 -o c -- python $R/tools/cfg5_only.py > $R/gpurun_out/prof_r03_cfg5.log 2>&1; echo "cfg5 stats exit $?"
cd $R
tail -1 gpurun_out/prof_r03_cfg5.log
timeout 300 python tools/bench_configs.py > gpurun_out/r03_bench_configs.log 2>&1; tail -8 gpurun_out/r03_bench_configs.log | cut -c1-260
timeout 400 python tools/bench_irregular.py > gpurun_out/r03_bench_irregular.log 2>&1; tail -4 gpurun_out/r03_bench_irregular.log | cut -c1-260
