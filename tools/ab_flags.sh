#!/bin/bash
# A/B of library build flags on one box: bash tools/ab_flags.sh "<flags A>" "<flags B>" [bench args]  -> ms_per_step of each, twice (A B A B)
R=${GRAFT_REPO_ROOT:-/root/repo}
fa=$1; fb=$2; shift 2
for rep in 1 2; do
  for f in "$fa" "$fb"; do
    bash $R/tools/run_variant.sh "$f" python3 $R/bench.py --no-cpu-baseline --no-cpp-host "$@" 2>$R/gpurun_out/ab_flags.err | python3 -c "import sys, json; j = json.loads(sys.stdin.readlines()[-1]); print('flags [%s]: %.3f ms / step' % ('$f', j['ms_per_step']))"
  done
done
