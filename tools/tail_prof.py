"""Section timing of scan_tail_kernel (library built with -DMIS_TAIL_PROF): python tools/tail_prof.py"""
import ctypes as C, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
feats = job.stage_features(frames)
job.stage_match(feats)
out = (C.c_ulonglong * 8)()
jp = (C.c_ulonglong * 8)()
ctx.lib.mis_debug_tail_prof(out, 1)
ctx.lib.mis_debug_jac_prof(jp, 1)
job.stage_match(feats)
ctx.lib.mis_debug_tail_prof(out, 1)
ctx.lib.mis_debug_jac_prof(jp, 1)
v = list(out)
tick = 0.01  # us (100 MHz)
print("tails %d  rotations %d  LM iterations %d" % (v[6], v[1], v[3]))
print("jacobi %.1f us total (%.2f us / rotation)   normal_eq %.1f us   dlt(incl. its jacobi) %.1f us   tail %.1f us  max tail %.1f us"
      % (v[0] * tick, v[0] * tick / max(v[1], 1), v[2] * tick, v[4] * tick, v[5] * tick, v[7] * tick))
j = list(jp)
rot = max(v[1], 1)
print("per rotation (shader cycles): pivot search %.0f  math %.0f  rotation %.0f  index update %.0f  sum %.0f" % (j[0] / rot, j[1] / rot, j[2] / rot, j[3] / rot, sum(j[:4]) / rot))
