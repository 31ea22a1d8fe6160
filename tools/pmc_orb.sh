#!/bin/bash
# PMC passes over the ORB kernels of one 4K frame at a time (tools/orb_single.py): bash tools/pmc_orb.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmco_$n -- python3 $R/tools/orb_single.py 6 > $R/gpurun_out/pmco_$n.log 2>&1; }
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $R
python3 tools/pmc_summary.py gpurun_out/pmco_inst gpurun_out/pmco_wait -k=fast_nms -k=select_rank -k=assemble_angle -k=describe_direct -k=resize_kernel -k=gray_kernel -k=compact_kernel
