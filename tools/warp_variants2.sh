#!/bin/bash
# A/B of warp kernel build variants on the GPU box (batched-grid and single-launch times only); see warp_variants.sh
R=$GRAFT_REPO_ROOT
i=0
for spec in "$@"; do
  i=$((i+1))
  cd $R/image_stitching_amd/csrc && touch warp.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $spec" > $R/gpurun_out/var_build$i.log 2>&1
  cd $R
  echo "variant $i: [$spec] $(python3 tools/warp_only.py 20 2>&1 | grep 'back-to-back\|batched' | awk '{printf "%s %.2f | ", $1, $NF}')"
done
