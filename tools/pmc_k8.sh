#!/bin/bash
# PMC passes over K8 (tools/k8_time.py), separate passes per counter group: bash tools/pmc_k8.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  rm -rf $R/gpurun_out/pmck8_$n
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmck8_$n -- python3 $R/tools/k8_time.py 24000 24000 4 > $R/gpurun_out/pmck8_$n.log 2>&1
}
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
cd $R
python3 tools/pmc_summary.py -k=l2_knn2_mfma gpurun_out/pmck8_inst gpurun_out/pmck8_wait gpurun_out/pmck8_mfma gpurun_out/pmck8_tcc gpurun_out/pmck8_tcp
