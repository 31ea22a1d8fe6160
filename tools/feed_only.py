"""Feed leg alone (config 3): the 16 mis_blender_feed calls between HIP events on the compose stream + finalise."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, synth
import image_stitching_amd as isa
from image_stitching_amd import distributed as misdist
cams = synth.workload("config3")
ctx = isa.Context(0)
job = misdist.StitchJob(ctx, (3840, 2160), cams)
frames = {i: synth.render_frame_gpu(cams[i]) for i in job.my_frames}
torch.cuda.synchronize()
r = bench.measure_roofline(ctx, job, frames, cams, int(sys.argv[1]) if len(sys.argv) > 1 else 50)
print("feed us/frame %.1f frac %.4f | finalize us %.1f frac %.4f | warp us %.2f | aggregate frac %.4f" % (
    r["parts"]["feed"]["us_per_frame"], r["parts"]["feed"]["frac"], r["parts"]["finalize"]["us"], r["parts"]["finalize"]["frac"],
    r["parts"]["warp"]["avg_launch_us"], r["frac"]))
