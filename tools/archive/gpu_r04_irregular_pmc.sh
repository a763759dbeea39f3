#!/bin/bash
# rocprofv3 counter passes of the SpMV kernels on the banded + random operator: separate --pmc runs,
# never mixed with trace domains, each under its own timeout.  -> gpurun_out/r04_spmv_irregular_pmc.json
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
run() {   # tag, counters, options...
  tag=$1; ctr=$2; shift 2
  timeout 240 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/irr_$tag -o p -- python $R/tools/spmv_irregular_only.py "$@" > $R/gpurun_out/irr_$tag.log 2>&1
  echo "$tag exit $?"
}
for form in "old spmv_kernel=1 spmv_delta=0 spmv_wide=0" "stage spmv_kernel=4 spmv_delta=0 spmv_wide=0 spmv_codes=0" "d8 spmv_kernel=1 spmv_delta=8 spmv_wide=1"; do
  set -- $form; f=$1; shift
  run ${f}_fetch FETCH_SIZE "$@"
  run ${f}_write WRITE_SIZE "$@"
  run ${f}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "$@"
  run ${f}_tcp "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "$@"
  run ${f}_sq "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES" "$@"
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json, os
res = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/irr_*/p_counter_collection.csv")):
    form = f.split("/")[1][4:].split("_")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "khip::spmv" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        res[form + " | " + k][c] = sum(v) / len(v)
ALG = 3575390532
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_side_bytes"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024     # MI355X_MICROARCH.md: KiB units, 128-B fetches tallied as 64 B
        v["hbm_side_over_algorithmic"] = v["hbm_side_bytes"] / ALG
    if "TCC_MISS_sum" in v:
        v["l2_miss_bytes"] = v["TCC_MISS_sum"] * 128
        v["l2_miss_over_algorithmic"] = v["l2_miss_bytes"] / ALG
res = dict(res); res["_algorithmic_bytes"] = ALG
json.dump(res, open("gpurun_out/r04_spmv_irregular_pmc.json", "w"), indent=1)
for k, v in res.items(): print(k[:110], v)
PY
