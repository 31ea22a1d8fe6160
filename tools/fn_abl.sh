#!/bin/bash
# fast_nms_kernel with its later phases cut off (FN_ABL = 1: staging only, 2: + stage A, 3: + arc evaluation, 0: whole): kernel time per batch launch
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in 1 2 3 0; do
  rm -rf $R/gpurun_out/orb_prof
  bash $R/tools/run_variant.sh "-DFN_ABL=$a" bash $R/tools/orb_prof.sh 2>&1 | grep fast_nms | sed "s/^/FN_ABL=$a: /"
done
cd $R/image_stitching_amd/csrc && touch *.hip && make -s > /dev/null 2>&1
