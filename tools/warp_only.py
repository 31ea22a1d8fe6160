import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
cams = synth.workload("config3")
cam = cams[8]
frame = synth.render_frame_gpu(cam)
scale = isa.Stitcher.warped_image_scale(cams)
w = isa.SphericalWarper(ctx, scale)
roi = w.warpRoi((3840, 2160), cam["K"], cam["R"])
dst, msk = w.alloc_fused(roi)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3): w.warp_fused_into(frame, cam["K"], cam["R"], roi, dst, msk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): w.warp_fused_into(frame, cam["K"], cam["R"], roi, dst, msk)
e1.record(); torch.cuda.synchronize()
print("roi", roi, "avg us (host-paced launches)", e0.elapsed_time(e1) / n * 1e3)
for _ in range(3):
    print("back-to-back kernel avg us", w.warp_fused_timed(frame, cam["K"], cam["R"], roi, dst, msk, 200))
# the compose loop's form: all 16 frames in one grid (mis_warp_spherical_fused_batch)
rois = isa.stitching.warp_rois(ctx, scale, (3840, 2160), cams)
fr = [synth.render_frame_gpu(c) for c in cams]
outs = [w.alloc_fused(r) for r in rois]
args = (fr, cams, rois, [o[0] for o in outs], [o[1] for o in outs])
w.warp_fused_batch_timed(*args, 5)
for _ in range(3):
    print("batched grid avg us per frame", w.warp_fused_batch_timed(*args, 12) / len(cams))
