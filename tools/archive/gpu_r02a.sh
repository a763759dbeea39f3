#!/bin/bash
# round 2, first GPU pass: new parity tests (coded columns, BASELINE-size oracle goldens), codes A/B, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02a}
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl
timeout 900 python -m pytest tests/test_gpu_scale_parity.py tests/test_gpu_primitives.py -m gpu -q --timeout 600 > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_$TAG.log
tail -25 gpurun_out/pytest_$TAG.log
cp gpurun_out/parity_log.jsonl gpurun_out/parity_log_$TAG.jsonl 2>/dev/null
timeout 300 python tools/archive/sweep_codes.py 512 > gpurun_out/sweep_codes_$TAG.log 2>&1; echo "sweep exit $?"; cat gpurun_out/sweep_codes_$TAG.log
timeout 300 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --opt spmv_codes=0 > gpurun_out/bench_${TAG}_codes0.json 2> gpurun_out/bench_${TAG}_codes0.err; echo "bench exit $?"; cat gpurun_out/bench_${TAG}_codes0.json
