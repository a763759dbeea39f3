#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for i in 1 2; do
  for o in "" "--opt spmv_nty=1"; do
    timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-parity $o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$o', round(d['value'],2), 'it/s spmv', round(d['roofline']['avg_ms'],4), 'self', d['self_consistency']['max_rel_dev'])"
  done
done
