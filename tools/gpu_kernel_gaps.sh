#!/bin/bash
# Device idle time between kernels of the cfg 5 / cfg 3 solves and of the headline CG loop: rocprofv3 --kernel-trace, then
# tools/kernel_gaps.py over the last milliseconds of each trace (the timed solve).  usage (GPU box): bash tools/gpu_kernel_gaps.sh <tag>
tag=${1:-r06ag}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out
run() {   # name, last-ms, command...
  name=$1; last=$2; shift 2
  rm -rf $out/${tag}_trace; mkdir -p $out/${tag}_trace
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_trace -o $name -- "$@" > $out/${tag}_${name}.log 2>&1
  f=$(find $out/${tag}_trace -name '*kernel_trace.csv' | head -1)
  { grep -E "^cfg|it/s|ms per" $out/${tag}_${name}.log | tail -2; python $root/tools/kernel_gaps.py $f --last-ms $last; } | tee $out/${tag}_${name}_gaps.log
  rm -rf $out/${tag}_trace $out/${tag}_${name}.log
}
run cfg5 130 python $root/tools/cfg5_only.py
run cfg3 140 python $root/tools/cfg3_only.py
run cg512 300 python $root/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs --no-full-parity
