"""One 4K frame through the batched strip grid (mis_warp_spherical_fused_batch with n = 1) against the one-frame launch
(mis_warp_spherical_fused): which kernel should a single launch run?  MIS_WARP_NT_PLAN sets the strip length."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
cams = synth.workload("config3")
scale = isa.Stitcher.warped_image_scale(cams)
w = isa.SphericalWarper(ctx, scale)
for idx in (8, 0):
    cam = cams[idx]
    frame = synth.render_frame_gpu(cam)
    roi = w.warpRoi((3840, 2160), cam["K"], cam["R"])
    dst, msk = w.alloc_fused(roi)
    w.warp_fused_timed(frame, cam["K"], cam["R"], roi, dst, msk, 50)
    a = [w.warp_fused_timed(frame, cam["K"], cam["R"], roi, dst, msk, 200) for _ in range(3)]
    args = ([frame], [cam], [roi], [dst], [msk])
    w.warp_fused_batch_timed(*args, 50)
    b = [w.warp_fused_batch_timed(*args, 200) for _ in range(3)]
    print("frame %d: one-frame launch %s | batch-of-one %s" % (idx, " ".join("%.2f" % v for v in a), " ".join("%.2f" % v for v in b)))
