#!/bin/bash
# rocprofv3 kernel statistics of the config-5 job (run through gpurun): bash tools/c5_prof.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c5_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5_prof -o c5 -- python3 $R/bench.py --workload config5 --steps 2 --warmup 1 > $R/gpurun_out/c5_prof.log 2>&1
tail -1 $R/gpurun_out/c5_prof.log | cut -c1-140
