#!/bin/bash
# A/B of warp kernel build variants on the GPU box: each argument is "ENV=val,... -Dflags"; rebuilds libmistitch.so there.
# Timing = back-to-back launches between HIP events (mis_warp_spherical_fused_timed), GPU kept busy.
R=$GRAFT_REPO_ROOT
i=0
for spec in "$@"; do
  i=$((i+1))
  envs=$(echo "$spec" | tr ' ' '\n' | grep '=' | grep -v '^-D' | tr '\n' ' ')
  flags=$(echo "$spec" | tr ' ' '\n' | grep '^-D' | tr '\n' ' ')
  cd $R/image_stitching_amd/csrc && touch warp.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $flags" > $R/gpurun_out/var_build$i.log 2>&1
  cd $R
  echo "variant $i: [$spec]"; env $envs python3 tools/warp_only.py 20 2>&1 | grep "avg us"
done
