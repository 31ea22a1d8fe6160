#!/bin/bash
# PMC pass over the Hamming 2-NN kernel inside the config-3 job: bash tools/pmc_knn.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_knn
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_knn -- python3 $R/tools/step_times.py 3 > $R/gpurun_out/pmc_knn.log 2>&1
cd $R && python3 tools/pmc_summary.py -k=knn2_hamming gpurun_out/pmc_knn
