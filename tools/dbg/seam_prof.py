import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, "/root/repo")
import image_stitching_amd as isa, synth
from image_stitching_amd import stitching as st
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = [synth.render_frame_gpu(c) for c in cams]
cfg = isa.StitchConfig.reference()
scale = st.Stitcher.warped_image_scale(cams)
for _ in range(2):
    out = [st.seam_scale_warp(ctx, cfg, (3840, 2160), f, c, scale) for f, c in zip(frames, cams)]
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
out = [st.seam_scale_warp(ctx, cfg, (3840, 2160), f, c, scale) for f, c in zip(frames, cams)]
torch.cuda.synchronize()
print("16 seam-scale warps: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
