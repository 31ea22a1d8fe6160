#!/bin/bash
# rocprofv3 kernel statistics of the default bench (config 3): bash tools/bench_prof.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/bench_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bench_$tag -o b -- python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/bench_$tag.log 2>&1
tail -1 $R/gpurun_out/bench_$tag.log | cut -c1-200
