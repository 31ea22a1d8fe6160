"""Which stage of a config-5 step (8 x 8K, SIFT) holds the occasional +300 ms?  Per-step stage stamps (MIS_JOB_TRACE marks)."""
import os, sys, time
os.environ["MIS_JOB_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc, torch
import image_stitching_amd as isa, synth
from image_stitching_amd import distributed as misdist
cams = synth.workload("config5")
ctx = isa.Context(0)
cfg = isa.StitchConfig.hot_path(features_type="sift")
job = misdist.StitchJob(ctx, (cams[0]["width"], cams[0]["height"]), cams, config=cfg)
frames = {i: synth.render_frame_gpu(cams[i]) for i in job.my_frames}
torch.cuda.synchronize()
for _ in range(2):
    job.run(frames)
gc.collect(); gc.disable()
for s in range(14):
    t0 = time.perf_counter()
    job.run(frames)
    dt = (time.perf_counter() - t0) * 1e3
    m = job.marks
    parts = " ".join("%s=%.1f" % (m[i][0].replace(" ", "_"), (m[i][1] - m[i - 1][1]) * 1e3) for i in range(1, len(m)))
    print("step %2d %7.1f ms | %s | torch reserved %.2f GB" % (s, dt, parts, torch.cuda.memory_reserved() / 1e9), flush=True)
