#!/bin/bash
# kernel statistics of the matcher alone (tools/match_time.py: 4 calls of mis_match_all_pairs on config 3)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/match_prof -- python3 $R/tools/match_time.py > $R/gpurun_out/match_prof.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/match_prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")[:44]
    if float(r["TotalDurationNs"]) > 2e5 and "synth" not in n:
        print("%-44s calls %4s avg %8.1f us max %8.1f total %8.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
grep "C call" gpurun_out/match_prof.log
