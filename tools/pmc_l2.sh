#!/bin/bash
# issue / matrix-pipe / LDS / L2 counters of the L2 2-NN pass on MFMA (run on the GPU box): bash tools/pmc_l2.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcl_mix -- python3 $R/tools/l2_single.py > $R/gpurun_out/pmcl_mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcl_pipe -- python3 $R/tools/l2_single.py > $R/gpurun_out/pmcl_pipe.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/pmcl_l2 -- python3 $R/tools/l2_single.py > $R/gpurun_out/pmcl_l2.log 2>&1
cd $R
python3 tools/pmc_summary.py -k=l2_knn2_mfma gpurun_out/pmcl_mix gpurun_out/pmcl_pipe gpurun_out/pmcl_l2
