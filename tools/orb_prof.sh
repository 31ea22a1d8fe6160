#!/bin/bash
# kernel statistics of the ORB feature stage alone (tools/orb_only.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/orb_prof -- python3 $R/tools/orb_only.py > $R/gpurun_out/orb_prof.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/orb_prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")[:44]
    if float(r["TotalDurationNs"]) > 1e5 and "synth" not in n:
        print("%-44s calls %4s avg %8.1f us max %8.1f total %8.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -3 gpurun_out/orb_prof.log
