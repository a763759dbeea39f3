#!/bin/bash
# rocprofv3 passes for the committed evidence: kernel stats of bench.py (csv) and HBM byte counters of the
# SpMV kernels (separate --pmc passes; never mixed with trace domains; every command under its own timeout)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; TAG=${1:-p}; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -o s -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_stats.log 2>&1; echo "stats exit $?"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o f -- python $R/tools/spmv_only.py 512 3 > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1; echo "fetch exit $?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o w -- python $R/tools/spmv_only.py 512 3 > $R/gpurun_out/prof_${TAG}_write.log 2>&1; echo "write exit $?"
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_${TAG}_tcc -o t -- python $R/tools/spmv_only.py 512 3 > $R/gpurun_out/prof_${TAG}_tcc.log 2>&1; echo "tcc exit $?"
# the int32 column stream (spmv_codes = 0) for comparison
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch0 -o f -- python $R/tools/spmv_only.py 512 3 spmv_codes=0 > $R/gpurun_out/prof_${TAG}_fetch0.log 2>&1; echo "fetch0 exit $?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write0 -o w -- python $R/tools/spmv_only.py 512 3 spmv_codes=0 > $R/gpurun_out/prof_${TAG}_write0.log 2>&1; echo "write0 exit $?"
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, json, sys
tag = sys.argv[1]
res = {}
for name, f in (("FETCH_SIZE", f"gpurun_out/prof_{tag}_fetch/f_counter_collection.csv"), ("WRITE_SIZE", f"gpurun_out/prof_{tag}_write/w_counter_collection.csv"),
                ("TCC", f"gpurun_out/prof_{tag}_tcc/t_counter_collection.csv"),
                ("FETCH_SIZE", f"gpurun_out/prof_{tag}_fetch0/f_counter_collection.csv"), ("WRITE_SIZE", f"gpurun_out/prof_{tag}_write0/w_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        if "khip::spmv" in k or "reduce_kernel" in k or "map_kernel<5" in k:
            res.setdefault(k, {})[c] = sum(v) / len(v)
import hashlib, os
h = hashlib.sha256()
for f in ("spmv.hip", "spmv_common.hpp", "colcode.hip", "device_reduce.hpp"):       # = bench.py KERNEL_SOURCES
    h.update(open(os.path.join("krylov.jl_amd", "csrc", f), "rb").read())
res["_kernel_source_sha"] = h.hexdigest()[:16]
json.dump(res, open(f"gpurun_out/prof_{tag}_pmc.json", "w"), indent=1)
for k, v in res.items():
    print(k[:100], v)
# the same passes for the int32 column stream (spmv_codes = 0) are taken by tools/gpu_prof_codes0.sh
PY
cat gpurun_out/prof_${TAG}_stats/s_kernel_stats.csv | cut -c1-200 | head -8
