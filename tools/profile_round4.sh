#!/bin/bash
# The judged profile set of round 4 (run through gpurun): the bench line, rocprofv3 kernel statistics of the same command, the roofline
# legs alone, the PMC passes (instruction / wait counters and, in runs of their own, FETCH_SIZE / WRITE_SIZE) over warp, feed and
# finalise, the config-5 and config-4 (one GPU) jobs and the reference-default pipeline.   bash tools/profile_round4.sh <tag> [quick]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-a}
O=$R/gpurun_out/prof4_$tag
rm -rf $O && mkdir -p $O
cd $R && BENCH_TRACE=1 python3 bench.py > $O/bench.json 2> $O/bench.err
# (the plain bench runs first: after the PMC passes a box keeps the profiling clock state for a while -- config 5 then measured 160 - 260 ms per step instead of 83)
if [ "$2" != "quick" ]; then
  BENCH_TRACE=1 python3 bench.py --workload config5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err
  python3 bench.py --workload config4 --steps 3 --warmup 1 --no-cpu-baseline --no-cpp-host > $O/bench_config4_1gpu.json 2> $O/bench_config4_1gpu.err
  python3 bench.py --pipeline reference --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_reference.json 2> $O/bench_reference.err
  python3 bench.py --pipeline hot_path_plus_seams --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_plus_seams.json 2> $O/bench_plus_seams.err
fi
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cpp-host > $O/stats.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -o b -- python3 $R/bench.py --roofline-only > $O/roofline.log 2>&1 )
cd $R
MIS_ROOFLINE_BATCH_ONLY=1 bash tools/pmc_feed.sh > $O/feed_pmc_summary.txt 2>&1
python3 tools/pmc_json.py gpurun_out/pmcf_fetch gpurun_out/pmcf_write $O/feed_pmc.json 32 pyr_down_l1_batch_kernel pyr_down_level_batch_kernel feed_tail_build_kernel feed_gather_kernel > /dev/null
python3 tools/pmc_json.py gpurun_out/pmcf_fetch gpurun_out/pmcf_write $O/finalize_pmc.json 2 normalize_kernel collapse2x2_kernel collapse2x2_final_kernel > /dev/null
bash tools/pmc_warp3.sh full > $O/warp_pmc_summary.txt 2>&1
python3 tools/pmc_json.py gpurun_out/pmc3_fetch gpurun_out/pmc3_write $O/warp_pmc_raw.json 1 warp_fused_kernel warp_strip_batch_kernel > /dev/null
if [ "$2" != "quick" ]; then
  # SIFT: kernel timeline of one 8K detect; the latency micro-benchmark; the matcher's chain stamps; LAST: the tails' per-section timers (rebuilds the library with -DMIS_TAIL_PROF)
  bash tools/sift_prof.sh > $O/sift_time.txt 2>&1
  python3 tools/sift_timeline.py > $O/sift_timeline.txt 2>&1
  [ -x tools/micro/_bin/lat_bench ] && ./tools/micro/_bin/lat_bench > $O/lat_bench.txt 2>&1
  MIS_MATCH_TRACE=1 python3 tools/host_timeline.py 10 > $O/host_timeline.txt 2>&1
  bash tools/run_variant.sh "-DMIS_TAIL_PROF" python3 tools/tail_prof.py > $O/tail_prof.txt 2>&1
fi
tail -1 $O/bench.json | cut -c1-300
for f in bench_config5 bench_config4_1gpu bench_reference bench_plus_seams; do [ -f $O/$f.json ] && tail -1 $O/$f.json | cut -c1-160; done
