#!/bin/bash
# A/B of warp.hip build variants (batched grid only, default strip plan): bash tools/warp_spec_ab.sh "<flags 1>" "<flags 2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  cd $R/image_stitching_amd/csrc && touch warp.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $spec" > $R/gpurun_out/var_build.log 2>&1 || { tail -5 $R/gpurun_out/var_build.log; continue; }
  cd $R
  echo "spec [$spec]: $(python3 tools/warp_only.py 5 2>&1 | grep 'batched' | awk '{printf "%.2f ", $NF}')"
done
