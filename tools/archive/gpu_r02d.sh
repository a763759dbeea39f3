#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02d}
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/bench_configs.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_$TAG.log
tail -8 gpurun_out/pytest_$TAG.log
cp gpurun_out/parity_log.jsonl gpurun_out/parity_log_$TAG.jsonl 2>/dev/null
timeout 600 python tools/bench_configs.py > gpurun_out/bench_configs_$TAG.log 2>&1; echo "configs exit $?"; cat gpurun_out/bench_configs.jsonl
