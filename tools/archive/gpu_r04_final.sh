#!/bin/bash
# round 4 final evidence: the GPU suite, bench.py (driver setting and 100 steps), rocprofv3 kernel stats of bench.py / cfg 3 at
# 256^3 and 384^3 / cfg 5, the SpMV counter passes of this build (tools/gpu_prof.sh), the configs, the irregular operators and
# the MatrixMarket fixtures.  Every command under its own timeout.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/bench_configs.jsonl gpurun_out/bench_irregular.jsonl
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/gpu_prof.sh r04 2>&1 | tail -12
cp gpurun_out/prof_r04_pmc.json profiles/r04_spmv_pmc.json 2>/dev/null      # bench.py quotes it only when the kernel-source sha matches
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_setting.json 2> gpurun_out/r04_bench.err; tail -c 400 gpurun_out/r04_bench_driver_setting.json
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r04_bench.json 2>> gpurun_out/r04_bench.err; tail -c 300 gpurun_out/r04_bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_cfg3 -o c -- python $R/tools/cfg3_only.py > $R/gpurun_out/prof_r04_cfg3.log 2>&1; echo "cfg3 stats exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_gmres384 -o c -- python $R/tools/cfg3_only.py 384 > $R/gpurun_out/prof_r04_gmres384.log 2>&1; echo "gmres 384 stats exit $?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_cfg5 -o c -- python $R/tools/cfg5_only.py > $R/gpurun_out/prof_r04_cfg5.log 2>&1; echo "cfg5 stats exit $?"
cd $R
tail -1 gpurun_out/prof_r04_cfg3.log; tail -1 gpurun_out/prof_r04_gmres384.log; tail -1 gpurun_out/prof_r04_cfg5.log
timeout 300 python tools/bench_configs.py > gpurun_out/r04_bench_configs.log 2>&1; tail -8 gpurun_out/r04_bench_configs.log | cut -c1-260
timeout 400 python tools/bench_irregular.py > gpurun_out/r04_bench_irregular.log 2>&1; grep -E "spmv|cg!" gpurun_out/r04_bench_irregular.log | cut -c1-300
timeout 100 python tools/bench_mtx.py --oracle tests/golden > gpurun_out/r04_bench_mtx.log 2>&1; cut -c1-300 gpurun_out/r04_bench_mtx.log
timeout 300 python tools/bench_sizes.py > gpurun_out/r04_bench_sizes.jsonl 2>&1; tail -3 gpurun_out/r04_bench_sizes.jsonl | cut -c1-260
timeout 300 python tools/bench_ilu.py 256 > gpurun_out/r04_bench_ilu.jsonl 2>&1; tail -2 gpurun_out/r04_bench_ilu.jsonl | cut -c1-400
timeout 200 python tools/archive/cg_solve_overhead.py > gpurun_out/r04_cg_solve_overhead.log 2>&1; cat gpurun_out/r04_cg_solve_overhead.log | cut -c1-200
