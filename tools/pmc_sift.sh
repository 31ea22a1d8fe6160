#!/bin/bash
# PMC passes over the SIFT kernels of one 8K frame (tools/sift_single.py): bash tools/pmc_sift.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcs_$n -- python3 $R/tools/sift_single.py 2 > $R/gpurun_out/pmcs_$n.log 2>&1; }
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run busy GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
cd $R
python3 tools/pmc_summary.py gpurun_out/pmcs_inst gpurun_out/pmcs_wait gpurun_out/pmcs_busy -k=sift_blur_fused_kernel -k=sift_descriptor -k=sift_refine -k=sift_extrema
