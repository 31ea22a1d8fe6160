#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for plan in "$@"; do
  if [ "$plan" = "default" ]; then unset MIS_WARP_NT_PLAN; else export MIS_WARP_NT_PLAN="$plan"; fi
  echo "plan [$plan]: $(python3 tools/warp_single_via_batch.py 2>&1 | tail -2 | tr '\n' ' ')"
done
