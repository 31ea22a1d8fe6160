"""cProfile of the Python side of one reference-pipeline step (StitchConfig.reference(), config 3): python tools/dbg/ref_prof.py"""
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams, config=isa.StitchConfig.reference())
for _ in range(2):
    out = job.run(frames)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    out = job.run(frames)
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
