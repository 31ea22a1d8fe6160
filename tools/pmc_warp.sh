#!/bin/bash
# Kernel time + PMC passes over the warp kernel (run on the GPU box): separate passes per counter group.
# usage: bash tools/pmc_warp.sh [quick]
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_time -- python3 $R/tools/warp_only.py 40 > $R/gpurun_out/pmc_time.log 2>&1
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$n -- python3 $R/tools/warp_only.py 10 > $R/gpurun_out/pmc_$n.log 2>&1
}
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
if [ "$1" != "quick" ]; then
  run fetch FETCH_SIZE
  run write WRITE_SIZE
fi
cd $R
python3 tools/prof_kernels.py gpurun_out/pmc_time warp
python3 tools/pmc_summary.py gpurun_out/pmc_inst gpurun_out/pmc_wait $( [ "$1" != "quick" ] && echo gpurun_out/pmc_fetch gpurun_out/pmc_write )
