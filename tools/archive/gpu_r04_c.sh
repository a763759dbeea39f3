#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04c_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r04c_pytest.log
timeout 400 python tools/archive/sweep_wave.py > gpurun_out/r04c_sweep_wave.log 2>&1; cat gpurun_out/r04c_sweep_wave.log
timeout 300 python tools/archive/sweep_headline.py "" "spmv_tiles=1" "spmv_tiles=1,spmv_blk_pub=0" "spmv_blk_pub=0" > gpurun_out/r04c_sweep_headline.log 2>&1; cat gpurun_out/r04c_sweep_headline.log
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-parity > gpurun_out/r04c_bench_100.json 2>/dev/null; cut -c1-200 gpurun_out/r04c_bench_100.json
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-parity --opt spmv_tiles=1 > gpurun_out/r04c_bench_100_tiles1.json 2>/dev/null; cut -c1-200 gpurun_out/r04c_bench_100_tiles1.json
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04c_bench_20.json 2>/dev/null; cut -c1-200 gpurun_out/r04c_bench_20.json
