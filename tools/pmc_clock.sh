#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/pmc_clk -- python3 $R/tools/warp_only.py 20 > $R/gpurun_out/pmc_clk.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pmc_clk/*/*counter_collection.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'warp_fused' in r['Kernel_Name']]
kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob('gpurun_out/pmc_clk/*/*kernel_trace.csv')[0]))}
import collections
acc = collections.defaultdict(dict)
for r in rows: acc[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
for d, c in list(acc.items())[-6:]:
    k = kt[d]; dur = (int(k['End_Timestamp']) - int(k['Start_Timestamp']))
    print(d, c, "dur ns", dur, "GUI_ACTIVE/ns", c.get('GRBM_GUI_ACTIVE', 0) / dur)
PY
