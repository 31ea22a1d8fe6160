#!/bin/bash
# The judged profile set of a round (run through gpurun): bench line, rocprofv3 kernel statistics of the same command,
# the roofline leg alone, and the config-5 job.  bash tools/profile_round.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-v6}
O=$R/gpurun_out/prof_$tag
rm -rf $O && mkdir -p $O
cd $R && python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -o b -- python3 $R/bench.py --roofline-only > $O/roofline.log 2>&1
cd $R && python3 bench.py --workload config5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -o b -- python3 $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline > $O/stats_config5.log 2>&1
tail -1 $O/bench.json | cut -c1-400
tail -1 $O/bench_config5.json | cut -c1-200
