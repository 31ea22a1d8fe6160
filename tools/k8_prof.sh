#!/bin/bash
# kernel time of K8 under rocprofv3 for a list of build variants: bash tools/k8_prof.sh "" "-DK8_SUB=2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
i=0
for flags in "$@"; do
  cd $R/image_stitching_amd/csrc && touch match.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $flags" > $R/gpurun_out/variant_build.log 2>&1
  cd /tmp && rm -rf $R/gpurun_out/k8_$i && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k8_$i -o k8 -- python3 $R/tools/k8_time.py 24000 24000 20 > $R/gpurun_out/k8_$i.log 2>&1
  echo "variant [$flags]: $(grep l2_knn2_mfma $R/gpurun_out/k8_$i/k8_kernel_stats.csv | awk -F'","' '{print "calls " $2 " avg_ns " $4}')"
  i=$((i+1))
done
