#!/bin/bash
# Headline CG loop (bench.py, 50 steps) with the coded kernel and with the sliced form under a few option sets, on ONE box.
# Prints it/s, ms per iteration and the HIP-event average of the fused SpMV per setting.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
run() {
  out=$(timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --no-full-parity "$@" 2>/dev/null)
  python - "$out" "$*" <<'PY'
import json, sys
o = json.loads(sys.argv[1]); r = o["roofline"]
print(f"{sys.argv[2]:60s} {o['value']:7.2f} it/s  {o['ms_per_step']:.4f} ms/it  spmv {r['avg_ms']:.4f} ms  frac {r['frac']:.3f}  parity {o['parity']['ok']} self {o.get('self_consistency', {}).get('max_rel_dev')}")
PY
}
for rep in 1 2; do
run
run --opt spmv_sell_pair=0
run --opt spmv_sell=0
done
