#!/bin/bash
# Round 4: PMC passes over the batched strip kernel for build variants of warp.hip, then the phase stamps of the default build.
#   bash tools/pmc_warp4.sh "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
build() { cd $R/image_stitching_amd/csrc && touch warp.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $1" > $R/gpurun_out/var_build.log 2>&1 || { tail -5 $R/gpurun_out/var_build.log; return 1; }; cd $R; }
run() { # name counters...
  n=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc4_$n -- python3 $R/tools/warp_only.py 3 > $R/gpurun_out/pmc4_$n.log 2>&1 )
}
i=0
for spec in "$@"; do
  i=$((i+1))
  build "$spec" || continue
  rm -rf $R/gpurun_out/pmc4_*
  run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run wait SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run grbm GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU
  echo "=== spec $i [$spec]"
  python3 tools/pmc_summary.py -k=warp_strip_batch gpurun_out/pmc4_inst gpurun_out/pmc4_wait gpurun_out/pmc4_grbm
done
build "-DWV_STAMPS" && echo "=== stamps (batch, middle frame)" && python3 tools/warp_stamps3.py batch
