#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for cfg in "spmv_xcd=0" "spmv_xcd=4" "spmv_xcd=16" "spmv_xcd=64"; do
  tag=$(echo $cfg | tr ' =' '__')
  timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/tools/spmv_only.py 512 2 $cfg > $R/gpurun_out/pmc_$tag.log 2>&1; echo "$cfg exit $?"
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_*/p_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'spmv_stage' in r['Kernel_Name'] and 'false, false, false, false' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[1], {k: sum(v)/len(v) for k,v in agg.items()})
PY
