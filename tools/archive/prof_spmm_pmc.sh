#!/bin/bash
# one rocprofv3 --pmc pass over SpMM p=16 at 216^3 (counter collection only, no trace domains)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; TAG=${1:-w}; shift; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/prof_spmm_${TAG} -o c -- python $R/tools/spmm_only.py 216 3 > $R/gpurun_out/prof_spmm_${TAG}.log 2>&1; echo "exit $?"
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, sys, glob
tag = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/prof_spmm_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm" in r["Kernel_Name"] and "build" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:60s} {c:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
tail -3 gpurun_out/prof_spmm_${TAG}.log
