#!/bin/bash
# A/B of strip-length plans of the batched warp grid (MIS_WARP_NT_PLAN, read once per process): bash tools/warp_plan_ab.sh "plan1" "plan2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for plan in "$@"; do
  if [ "$plan" = "default" ]; then unset MIS_WARP_NT_PLAN; else export MIS_WARP_NT_PLAN="$plan"; fi
  echo "plan [$plan]: $(python3 tools/warp_only.py 5 2>&1 | grep 'batched' | awk '{printf "%.2f ", $NF}')"
done
