#!/bin/bash
# PMC passes over the round-3 strip kernels (run on the GPU box): separate passes per counter group.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc3_$n -- python3 $R/tools/warp_only.py 10 > $R/gpurun_out/pmc3_$n.log 2>&1
}
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sca SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC
run grbm GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU
if [ "$1" == "full" ]; then
  run fetch FETCH_SIZE
  run write WRITE_SIZE
fi
cd $R
python3 tools/pmc_summary.py -k=warp_strip gpurun_out/pmc3_inst gpurun_out/pmc3_wait gpurun_out/pmc3_sca gpurun_out/pmc3_grbm $( [ "$1" == "full" ] && echo gpurun_out/pmc3_fetch gpurun_out/pmc3_write )
