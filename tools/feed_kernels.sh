#!/bin/bash
# per-call durations of the feed kernels (batched legs = the largest calls) from a kernel trace of bench.py --roofline-only
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/feedk
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/feedk -o b -- python3 $R/bench.py --roofline-only > $R/gpurun_out/feedk.log 2>&1 )
python3 - <<PY
import csv, collections, glob
rows=list(csv.DictReader(open((glob.glob("$R/gpurun_out/feedk/*/b_kernel_trace.csv") + glob.glob("$R/gpurun_out/feedk/b_kernel_trace.csv"))[0])))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]
    agg[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k in ("warp_strip_batch_kernel","pyr_down_l1_batch_kernel","pyr_down_level_batch_kernel","feed_tail_build_kernel","feed_gather_kernel","collapse2x2_final_kernel"):
    v=sorted(agg[k], reverse=True)
    print(k, len(v), "largest:", [round(x,1) for x in v[:6]], "median", round(v[len(v)//2],1))
PY
