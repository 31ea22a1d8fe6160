#!/bin/bash
# A/B of an environment setting on one box: bash tools/ab_env.sh "VAR=a" "VAR=b" [bench args] -> ms_per_step of each, twice (A B A B)
R=${GRAFT_REPO_ROOT:-/root/repo}
ea=$1; eb=$2; shift 2
for rep in 1 2; do
  for e in "$ea" "$eb"; do
    env $e python3 $R/bench.py --no-cpu-baseline --no-cpp-host "$@" 2>/dev/null | python3 -c "import sys, json; j = json.loads(sys.stdin.readlines()[-1]); print('[%s]: %.3f ms / step (median %.3f)' % ('$e', j['ms_per_step'], j['ms_per_step_median']))"
  done
done
