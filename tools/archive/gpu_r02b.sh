#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02b}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/archive/sweep_pipe.py 512 > gpurun_out/sweep_pipe_$TAG.log 2>&1; echo "sweep exit $?"; cat gpurun_out/sweep_pipe_$TAG.log
rm -f gpurun_out/parity_log.jsonl
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_$TAG.log
tail -15 gpurun_out/pytest_$TAG.log
cp gpurun_out/parity_log.jsonl gpurun_out/parity_log_$TAG.jsonl 2>/dev/null
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
