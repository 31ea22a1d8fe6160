"""Pass rates of the FAST stages on the bench frames (library built with -DMIS_ORB_STATS): python tools/orb_stats.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
out = (C.c_ulonglong * 8)()
ctx.lib.mis_debug_orb_stats(out, 1)
job.stage_features(frames)
ctx.lib.mis_debug_orb_stats(out, 1)
v = list(out)
print("pyramid pixels scored %d (16 frames, incl. the tiles' halo): opposite-pair test passes %.2f %%, pre-test passes %.2f %%, corners %.2f %%"
      % (v[0], 100.0 * v[1] / v[0], 100.0 * v[2] / v[0], 100.0 * v[3] / v[0]))
