"""Copies the judged artefacts of a tools/profile_round4.sh run from gpurun_out/ (scratch) into profiles/ (tracked), and writes
profiles/r04_traffic_pmc.json with the hash of the kernel sources the counters were measured on (bench.py quotes `traffic` only when
that hash equals the hash of the sources it runs from):   python tools/collect_profiles4.py <tag> <version>   e.g.  a v1"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
tag, ver = sys.argv[1], sys.argv[2]
src = os.path.join(R, "gpurun_out", "prof4_" + tag)
dst = os.path.join(R, "profiles")
cp = [("bench.json", "r04_bench_%s.json"), ("stats/b_kernel_stats.csv", "r04_kernel_stats_%s.csv"), ("roofline/b_kernel_stats.csv", "r04_kernel_stats_%s_roofline_only.csv"),
      ("feed_pmc_summary.txt", "r04_feed_pmc_summary_%s.txt"), ("warp_pmc_summary.txt", "r04_warp_pmc_summary_%s.txt"), ("bench_config5.json", "r04_bench_%s_config5.json"),
      ("bench_config4_1gpu.json", "r04_bench_%s_config4_1gpu.json"), ("sift_timeline.txt", "r04_sift_timeline_8k_%s.txt"), ("lat_bench.txt", "r04_lat_bench_%s.txt"),
      ("host_timeline.txt", "r04_chain_stamps_%s.txt"), ("tail_prof.txt", "r04_tail_prof_%s.txt"), ("bench_reference.json", "r04_bench_%s_reference.json"), ("bench_plus_seams.json", "r04_bench_%s_hot_path_plus_seams.json")]
for a, b in cp:
    for p in (os.path.join(src, a), os.path.join(src, os.path.dirname(a), "*", os.path.basename(a))):
        import glob
        hits = glob.glob(p)
        if hits:
            data = open(hits[0]).read()
            if a.endswith(".json"):
                data = data.strip().splitlines()[-1] + "\n"
            open(os.path.join(dst, b % ver), "w").write(data)
            break
import bench
feed = json.load(open(os.path.join(src, "feed_pmc.json")))
fin = json.load(open(os.path.join(src, "finalize_pmc.json")))
warp = json.load(open(os.path.join(src, "warp_pmc_raw.json")))
b = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
cfg = b["config"]
batch = warp["kernels"]["warp_strip_batch_kernel"]
single = warp["kernels"]["warp_fused_kernel"]
frames = cfg["frames"]
out = {"version": ver, "round": 4, "frame_size": cfg["frame_size"], "workload": cfg["workload"], "kernels_sha": bench.kernels_sha(),
       "kernels_sha_note": "sha256 (16 hex digits) of csrc/warp.hip resp. csrc/blend.hip + common.h + dev_math.h + Makefile at the time of the PMC passes",
       "correction": feed["correction"] + " -- calibrated for wide (16 B per lane) streaming reads only; the blend kernels read dwords / 8-byte pixels (level 1 of the feed now by 16-byte LDS-DMA pieces), so their doubled FETCH_SIZE is an upper estimate",
       "warp": {"kernel": "warp_strip_batch_kernel (the compose loop's grid: %d frames per dispatch; traffic_bytes_per_launch = one dispatch / %d)" % (frames, frames),
                "traffic_bytes_per_launch": batch["traffic_bytes_per_dispatch"] // frames,
                "FETCH_SIZE_KiB": round(batch["FETCH_SIZE_KiB_per_dispatch"] / frames, 1), "WRITE_SIZE_KiB": round(batch["WRITE_SIZE_KiB_per_dispatch"] / frames, 1),
                "batch_dispatch": batch,
                "single_frame_kernel": dict(single, kernel="warp_fused_kernel (one frame per launch: round 2's tile kernel)")},
       "feed": {"traffic_bytes_per_frame": feed["traffic_bytes_per_unit"], "kernels": feed["kernels"]},
       "finalize": {"traffic_bytes_per_panorama": fin["traffic_bytes_per_unit"], "kernels": fin["kernels"]}}
json.dump(out, open(os.path.join(dst, "r04_traffic_pmc.json"), "w"), indent=1)
print("copied", ver, out["kernels_sha"])
