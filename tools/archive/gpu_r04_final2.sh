#!/bin/bash
# round 4, last build: the whole GPU suite (parity log), bench.py at the driver's setting and at 100 steps with the committed
# counter file (the SpMV kernel sources have not changed since tools/gpu_r04_final.sh), smoke().
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_setting.json 2> gpurun_out/r04_bench.err; tail -c 300 gpurun_out/r04_bench_driver_setting.json
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r04_bench.json 2>> gpurun_out/r04_bench.err; tail -c 200 gpurun_out/r04_bench.json
timeout 100 python -c "import __graft_entry__ as g; g.smoke()"
