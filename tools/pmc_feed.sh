#!/bin/bash
# PMC passes over the blend feed / finalise kernels (run on the GPU box): separate passes per counter group, as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in passes of their own).  usage: bash tools/pmc_feed.sh [quick]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcf_$n -- python3 $R/tools/feed_only.py 5 > $R/gpurun_out/pmcf_$n.log 2>&1
}
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
if [ "$1" != "quick" ]; then
  run fetch FETCH_SIZE
  run write WRITE_SIZE
fi
cd $R
K="-k=pyr_down -k=feed_gather -k=feed_tail -k=collapse2x2 -k=finalize_kernel -k=laplace"
python3 tools/pmc_summary.py gpurun_out/pmcf_inst gpurun_out/pmcf_wait $( [ "$1" != "quick" ] && echo gpurun_out/pmcf_fetch gpurun_out/pmcf_write ) $K
