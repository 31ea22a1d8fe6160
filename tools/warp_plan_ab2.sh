#!/bin/bash
# build variant of warp.hip ($1 = extra flags), then A/B of strip plans ($2...): bash tools/warp_plan_ab2.sh "-DWV3_NT_MAX=32" plan1 plan2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
spec="$1"; shift
cd $R/image_stitching_amd/csrc && touch warp.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $spec" > $R/gpurun_out/var_build.log 2>&1 || { tail -5 $R/gpurun_out/var_build.log; exit 1; }
cd $R
echo "build [$spec]"
bash tools/warp_plan_ab.sh "$@"
