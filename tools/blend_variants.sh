#!/bin/bash
# A/B of blend.hip build variants on the GPU box: each argument is a set of -D flags; rebuilds libmistitch.so there.
R=$GRAFT_REPO_ROOT
i=0
for spec in "$@"; do
  i=$((i+1))
  cd $R/image_stitching_amd/csrc && touch blend.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $spec" > $R/gpurun_out/bvar_build$i.log 2>&1
  cd $R
  echo "variant $i: [$spec]"; python3 tools/feed_only.py 20 2>&1 | grep "feed us"
done
