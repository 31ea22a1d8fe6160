#!/bin/bash
# LDS / issue counters of the strip kernels (run on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc3_lds -- python3 $R/tools/warp_only.py 10 > $R/gpurun_out/pmc3_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc3_mix -- python3 $R/tools/warp_only.py 10 > $R/gpurun_out/pmc3_mix.log 2>&1
cd $R
python3 tools/pmc_summary.py -k=warp_strip_batch gpurun_out/pmc3_lds gpurun_out/pmc3_mix
tail -3 gpurun_out/pmc3_lds.log
