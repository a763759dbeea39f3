#!/bin/bash
# rocprofv3 counter passes of the tile SpMM on the banded + random operator: separate --pmc passes, never mixed
# with trace domains, each under its own timeout -> gpurun_out/<tag>_spmm_irregular_pmc.json; then the synthetic twin of its traffic.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; TAG=${1:-r06}; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_spmmirr_pmc$i -o c -- python $R/tools/spmm_irregular_only.py > $R/gpurun_out/${TAG}_spmmirr_pmc$i.log 2>&1; echo "pmc pass $i ($set) exit $?"
done
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
res = {}
for f in sorted(glob.glob(f"gpurun_out/{tag}_spmmirr_pmc*/**/c_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        if "khip::spmm" in k:
            res.setdefault(k[:110], {})[c] = sum(v) / len(v)
line = open(f"gpurun_out/{tag}_spmmirr_pmc1.log").read().strip().splitlines()[-1]
res["_workload"] = line
for k, v in list(res.items()):
    if isinstance(v, dict) and "FETCH_SIZE" in v:
        # FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM section)
        v["_hbm_read_bytes_2x_fetch_size"] = 2 * 1024 * v["FETCH_SIZE"]
        v["_hbm_write_bytes"] = 1024 * v.get("WRITE_SIZE", 0.0)
        v["_l2_miss_bytes_128B"] = 128 * v.get("TCC_MISS_sum", 0.0)
        v["_l1_to_l2_read_bytes_128B"] = 128 * v.get("TCP_TCC_READ_REQ_sum", 0.0)
json.dump(res, open(f"gpurun_out/{tag}_spmm_irregular_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
timeout 200 tools/streamfloor spmmirr > gpurun_out/${TAG}_spmm_irregular_twin.log 2>&1; cat gpurun_out/${TAG}_spmm_irregular_twin.log
