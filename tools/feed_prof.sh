#!/bin/bash
# isolated kernel statistics of the roofline legs (warp, feed, finalise): rocprofv3 --kernel-trace --stats of tools/feed_only.py
R=$GRAFT_REPO_ROOT
tag=${1:-f}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o b -- python3 $R/tools/feed_only.py 20 > $R/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/$tag/b_kernel_stats.csv')))
for r in rows[:16]:
    print("%-64s calls %5s total %9.1f us avg %8.2f us"%(r['Name'][:64], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
PY
