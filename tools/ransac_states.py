"""RANSAC states after one matcher call on config 3 (diagnostics): which problems are slow to draw?"""
import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
feats = job.stage_features(frames)
pm = job.engine.match(feats, 0, 1)
for which in (0, 1):
    out = np.zeros((256, 8), np.int32)
    n = ctx.lib.mis_debug_ransac_states(ctx.h, which, out.ctypes.data_as(C.c_void_p), 256)
    st = out[:n]
    print("batch", which, "problems", n, " columns: n mode n_sub iter niters draw_fail done max_good")
    print("  draw_fail:", int(st[:, 5].sum()), " n of failing:", sorted(st[st[:, 5] == 1, 0].tolist()))
    print("  mode counts:", np.bincount(st[:, 1], minlength=3).tolist(), " iter>128:", int((st[:, 3] > 128).sum()), " n_sub hist:", np.histogram(st[:, 2], bins=[0, 1, 64, 128, 129, 1000, 2001])[0].tolist())
    small = st[(st[:, 0] > 0) & (st[:, 0] < 12)]
    print("  problems with n < 12:", small.tolist()[:20])
    if which == 0:      # round 4: who needs the second phase?  (n, iterations run) of every problem, by n
        order = np.argsort(st[:, 0])
        print("  first estimation, (n, iter) sorted by n:", [(int(st[i, 0]), int(st[i, 3])) for i in order])
