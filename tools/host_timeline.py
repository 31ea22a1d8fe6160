"""Host-side time stamps of StitchJob.run (MIS_JOB_TRACE=1): python tools/host_timeline.py [steps]"""
import os, sys
os.environ["MIS_JOB_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import synth
import image_stitching_amd as isa
from image_stitching_amd import distributed as misdist
wl = os.environ.get("MIS_WORKLOAD", "config3")      # MIS_WORKLOAD=config5: 8 x 8K, SIFT
cams = synth.workload(wl)
ctx = isa.Context(0)
feat = "sift" if wl == "config5" else "orb"
cfg = isa.StitchConfig(features_type=feat) if os.environ.get("MIS_PIPELINE") in ("reference_default", "hot_path_plus_seams") else isa.StitchConfig.hot_path(features_type=feat)
job = misdist.StitchJob(ctx, (cams[0]["width"], cams[0]["height"]), cams, config=cfg)
frames = {i: synth.render_frame_gpu(cams[i]) for i in job.my_frames}
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for it in range(n):
    job.run(frames)
    if it >= n - 3:
        t0 = job.marks[0][1]
        print(" | ".join("%s %.2f" % (k, (t - t0) * 1e3) for k, t in sorted(job.marks, key=lambda m: m[1])))
