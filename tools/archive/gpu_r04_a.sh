#!/bin/bash
# round 4, GPU call A: the whole -m gpu suite, the new stream forms on the non-stencil operators, counter passes on the
# irregular operator, the headline bench + two A/B switches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r04a_pytest.log
timeout 300 python tools/archive/sweep_delta.py > gpurun_out/r04a_sweep_delta.log 2>&1; echo "sweep_delta exit $?"; cat gpurun_out/r04a_sweep_delta.log | cut -c1-330
timeout 600 bash tools/gpu_r04_irregular_pmc.sh > gpurun_out/r04a_irregular_pmc.log 2>&1; echo "pmc exit $?"; tail -25 gpurun_out/r04a_irregular_pmc.log | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err; echo "bench exit $?"; cut -c1-600 gpurun_out/r04a_bench.json
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-parity --opt compensated=0 > gpurun_out/r04a_bench_nocomp.json 2>/dev/null; cut -c1-200 gpurun_out/r04a_bench_nocomp.json
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-parity > gpurun_out/r04a_bench_100.json 2>/dev/null; cut -c1-200 gpurun_out/r04a_bench_100.json
timeout 200 python tools/archive/sweep_plane_order.py 512 > gpurun_out/r04a_sweep7_coded_plane.log 2>&1; cat gpurun_out/r04a_sweep7_coded_plane.log
