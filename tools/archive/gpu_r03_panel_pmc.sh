#!/bin/bash
# rocprofv3 counter passes over the block Gram-Schmidt kernels (separate --pmc runs, no trace domains)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"; do
  t=$(echo $set | cut -d' ' -f1)
  cd /tmp
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_panel_$t -o c -- python $R/tools/archive/panel_mgs_only.py > $R/gpurun_out/prof_panel_$t.log 2>&1
  cd $R
  python3 - "$t" <<'PY'
import csv, collections, sys, glob
t = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/prof_panel_{t}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "panel_nn_tn" in r["Kernel_Name"] or "panel_gemm_tn" in r["Kernel_Name"] or "panel_gemm_nn" in r["Kernel_Name"] or "panel_multi" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][12:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:34s} {c:30s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
} > gpurun_out/r03_panel_pmc.log 2>&1
cat gpurun_out/r03_panel_pmc.log
