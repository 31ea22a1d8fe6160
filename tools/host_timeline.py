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
pl = os.environ.get("MIS_PIPELINE", "hot_path")      # hot_path | hot_path_plus_seams | reference (bench.py's --pipeline)
cfg = {"hot_path": lambda: isa.StitchConfig.hot_path(features_type=feat), "hot_path_plus_seams": lambda: isa.StitchConfig(features_type=feat, compose_megapix=-1),
       "reference_default": lambda: isa.StitchConfig(features_type=feat, compose_megapix=-1), "reference": lambda: isa.StitchConfig.reference(features_type=feat)}[pl]()
job = misdist.StitchJob(ctx, (cams[0]["width"], cams[0]["height"]), cams, config=cfg)
frames = {i: synth.render_frame_gpu(cams[i]) for i in job.my_frames}
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for it in range(n):
    job.run(frames)
    if it >= n - 3:
        t0 = job.marks[0][1]
        print(" | ".join("%s %.2f" % (k, (t - t0) * 1e3) for k, t in sorted(job.marks, key=lambda m: m[1])))
