#!/bin/bash
# round 3 final evidence: the GPU suite, bench.py, rocprofv3 kernel stats of bench.py / cfg 3 / cfg 5, the SpMV counter passes
# (tools/gpu_prof.sh), the configs and the irregular operators.  Every command under its own timeout.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/bench_configs.jsonl gpurun_out/bench_irregular.jsonl
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python bench.py --steps 100 --warmup 10 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 600 gpurun_out/r03_bench.json
bash tools/gpu_prof.sh r03 2>&1 | tail -12
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_cfg3 -o c -- python $R/tools/cfg3_only.py > $R/gpurun_out/prof_r03_cfg3.log 2>&1; echo "cfg3 stats exit $?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_cfg5 -o c -- python $R/tools/cfg5_only.py > $R/gpurun_out/prof_r03_cfg5.log 2>&1; echo "cfg5 stats exit $?"
cd $R
tail -1 gpurun_out/prof_r03_cfg3.log; tail -1 gpurun_out/prof_r03_cfg5.log
timeout 300 python tools/bench_configs.py > gpurun_out/r03_bench_configs.log 2>&1; tail -12 gpurun_out/r03_bench_configs.log | cut -c1-260
timeout 400 python tools/bench_irregular.py > gpurun_out/r03_bench_irregular.log 2>&1; tail -3 gpurun_out/r03_bench_irregular.log | cut -c1-200
