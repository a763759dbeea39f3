#!/bin/bash
# Quick GPU pass: parity tests + sweep + bench.  Usage: bash tools/gpu_quick.sh [tag]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-q}
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/sweep.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu_$TAG.log
tail -15 gpurun_out/pytest_gpu_$TAG.log
timeout 600 python tools/archive/sweep_kernels_512.py --quick > gpurun_out/sweep_$TAG.log 2>&1; echo "sweep exit $?"; grep -v "^{'kernel': 'dot'\|nrm2" gpurun_out/sweep_$TAG.log | tail -60
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
