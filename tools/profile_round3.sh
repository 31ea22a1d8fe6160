#!/bin/bash
# The judged profile set of round 3 (run through gpurun): bench line, rocprofv3 kernel statistics of the same command, the
# roofline legs alone (warp / feed / finalise isolated), PMC traffic passes (FETCH_SIZE / WRITE_SIZE in passes of their own),
# config-4 (one GPU) and config-5 jobs.    bash tools/profile_round3.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-a}
O=$R/gpurun_out/prof3_$tag
rm -rf $O && mkdir -p $O
cd $R && python3 bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cpp-host > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -o b -- python3 $R/bench.py --roofline-only > $O/roofline.log 2>&1
cd $R
# PMC passes: feed_only.py runs the feed / finalise legs, warp_only.py the warp kernels (batched grid and one frame per launch)
MIS_ROOFLINE_BATCH_ONLY=1 bash tools/pmc_feed.sh > $O/feed_pmc_summary.txt 2>&1
python3 tools/pmc_json.py gpurun_out/pmcf_fetch gpurun_out/pmcf_write $O/feed_pmc.json 32 pyr_down_l1_batch_kernel pyr_down_level_batch_kernel feed_tail_build_kernel feed_gather_kernel > /dev/null
python3 tools/pmc_json.py gpurun_out/pmcf_fetch gpurun_out/pmcf_write $O/finalize_pmc.json 2 normalize_kernel collapse2x2_kernel collapse2x2_final_kernel finalize_kernel > /dev/null
bash tools/pmc_warp3.sh full > $O/warp_pmc_summary.txt 2>&1
python3 tools/pmc_json.py gpurun_out/pmc3_fetch gpurun_out/pmc3_write $O/warp_pmc_raw.json 1 warp_fused_kernel warp_strip_batch_kernel > /dev/null
python3 bench.py --workload config5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err
python3 bench.py --workload config4 --steps 3 --warmup 1 --no-cpu-baseline --no-cpp-host > $O/bench_config4_1gpu.json 2> $O/bench_config4_1gpu.err
tail -1 $O/bench.json | cut -c1-300
tail -1 $O/bench_config5.json | cut -c1-200
tail -1 $O/bench_config4_1gpu.json | cut -c1-200
