#!/bin/bash
# kernel trace of a few bench steps -> the timeline of one step (tools/step_timeline.py):   bash tools/timeline3.sh [flags]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tl3
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-cpp-host "$@" > $O/log.txt 2>&1
cd $R
f=$(ls $O/*kernel_trace.csv $O/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/step_timeline.py $f 5 10 > $O/timeline.txt
cat $O/timeline.txt
