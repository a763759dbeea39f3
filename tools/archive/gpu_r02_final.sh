#!/bin/bash
# round-2 evidence pass: rocprofv3 stats + PMC of this build, then the bench line (default steps) with and without coded columns
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02f}
bash tools/gpu_prof.sh $TAG 2>&1 | tail -14
cp gpurun_out/prof_${TAG}_pmc.json profiles/r02_spmv_pmc.json      # so that bench.py below quotes the traffic of THIS build
timeout 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json
timeout 300 python bench.py --no-cpu-baseline --opt spmv_codes=0 > gpurun_out/bench_${TAG}_codes0.json 2> gpurun_out/bench_${TAG}_codes0.err; echo "bench codes0 exit $?"; cut -c1-400 gpurun_out/bench_${TAG}_codes0.json
