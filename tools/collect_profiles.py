"""Copies the judged artefacts of a tools/profile_round2.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
python tools/collect_profiles.py <tag> <version>   e.g.  a v1"""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, ver = sys.argv[1], sys.argv[2]
src = os.path.join(R, "gpurun_out", "prof2_" + tag)
dst = os.path.join(R, "profiles")
cp = [("bench.json", "r02_bench_%s.json"), ("stats/b_kernel_stats.csv", "r02_kernel_stats_%s.csv"), ("roofline/b_kernel_stats.csv", "r02_kernel_stats_%s_roofline_only.csv"),
      ("feed_pmc_summary.txt", "r02_feed_pmc_summary_%s.txt"), ("bench_config5.json", "r02_bench_%s_config5.json"), ("bench_config4_1gpu.json", "r02_bench_%s_config4_1gpu.json")]
for a, b in cp:
    p = os.path.join(src, a)
    if os.path.exists(p):
        data = open(p).read()
        if a.endswith(".json"):
            data = data.strip().splitlines()[-1] + "\n"
        open(os.path.join(dst, b % ver), "w").write(data)
# traffic summaries the bench line reads (latest version wins: fixed names)
feed = json.load(open(os.path.join(src, "feed_pmc.json")))
fin = json.load(open(os.path.join(src, "finalize_pmc.json")))
warp = json.load(open(os.path.join(src, "warp_pmc_raw.json")))
b = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
cfg = b["config"]
out = {"version": ver, "frame_size": cfg["frame_size"], "workload": cfg["workload"],
       "correction": feed["correction"] + " -- calibrated for wide (16 B per lane) streaming reads only; the blend kernels read dwords / 8-byte pixels, so their doubled FETCH_SIZE is an upper estimate",
       "warp": {"kernel": "warp_fused_kernel (one frame per launch; warp_fused_batch_kernel moves the same bytes per frame: see batch_16_frames)",
                "traffic_bytes_per_launch": warp["kernels"]["warp_fused_kernel"]["traffic_bytes_per_dispatch"],
                "FETCH_SIZE_KiB": warp["kernels"]["warp_fused_kernel"]["FETCH_SIZE_KiB_per_dispatch"], "WRITE_SIZE_KiB": warp["kernels"]["warp_fused_kernel"]["WRITE_SIZE_KiB_per_dispatch"],
                "batch_16_frames": warp["kernels"].get("warp_fused_batch_kernel")},
       "feed": {"traffic_bytes_per_frame": feed["traffic_bytes_per_unit"], "kernels": feed["kernels"]},
       "finalize": {"traffic_bytes_per_panorama": fin["traffic_bytes_per_unit"], "kernels": fin["kernels"]}}
json.dump(out, open(os.path.join(dst, "r02_traffic_pmc.json"), "w"), indent=1)
print("copied", ver)
