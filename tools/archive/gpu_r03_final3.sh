#!/bin/bash
# round 3, last build: the whole GPU suite, kernel stats of cfg 5, the configs and the irregular operators (bench.py and the
# SpMV counter passes are those of tools/gpu_r03_final.sh / final2.sh: the sources they depend on have not changed since).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/bench_configs.jsonl gpurun_out/bench_irregular.jsonl gpurun_out/parity_log.jsonl
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_cfg5 -o c -- python $R/tools/cfg5_only.py > $R/gpurun_out/prof_r03_cfg5.log 2>&1; echo "cfg5 stats exit $?"
cd $R
grep cfg5 gpurun_out/prof_r03_cfg5.log | tail -1
timeout 300 python tools/bench_configs.py > gpurun_out/r03_bench_configs.log 2>&1; tail -6 gpurun_out/r03_bench_configs.log | cut -c1-200
timeout 400 python tools/bench_irregular.py > gpurun_out/r03_bench_irregular.log 2>&1; grep block_gmres gpurun_out/r03_bench_irregular.log | cut -c1-200
