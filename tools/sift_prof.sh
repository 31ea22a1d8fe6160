#!/bin/bash
# Kernel-level profile of tools/sift_time.py (run through gpurun): bash tools/sift_prof.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/sift_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sift_prof -o sift -- python3 $R/tools/sift_time.py > $R/gpurun_out/sift_prof.log 2>&1
grep "per frame" $R/gpurun_out/sift_prof.log
