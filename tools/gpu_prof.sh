#!/bin/bash
# rocprofv3 passes: kernel stats (csv) and HBM byte counters (separate --pmc passes, no trace domains mixed in)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; TAG=${1:-p}; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -o s -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_stats.log 2>&1; echo "stats exit $?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o f -- python $R/tools/spmv_only.py 512 3 > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1; echo "fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o w -- python $R/tools/spmv_only.py 512 3 > $R/gpurun_out/prof_${TAG}_write.log 2>&1; echo "write exit $?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_${TAG}_tcc -o t -- python $R/tools/spmv_only.py 512 3 > $R/gpurun_out/prof_${TAG}_tcc.log 2>&1; echo "tcc exit $?"
cd $R; find gpurun_out/prof_${TAG}_* -name "*.csv" | head -20
