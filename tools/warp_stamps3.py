"""Per-tile phase stamps of warp_strip_kernel (library built with -DWV_STAMPS): where a pipeline step's cycles go."""
import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
cams = synth.workload("config3")
cam = cams[8]
frame = synth.render_frame_gpu(cam)
scale = isa.Stitcher.warped_image_scale(cams)
w = isa.SphericalWarper(ctx, scale)
roi = w.warpRoi((3840, 2160), cam["K"], cam["R"])
dst, msk = w.alloc_fused(roi)
if len(sys.argv) > 1 and sys.argv[1] == "batch":     # steady state: the stamps of the middle frame of the 16-frame grid
    rois = isa.stitching.warp_rois(ctx, scale, (3840, 2160), cams)
    fr = [synth.render_frame_gpu(c) for c in cams]
    outs = [w.alloc_fused(r) for r in rois]
    w.warp_fused_batch_timed(fr, cams, rois, [o[0] for o in outs], [o[1] for o in outs], 3)
    roi = rois[8]
else:
    for _ in range(5): w.warp_fused_into(frame, cam["K"], cam["R"], roi, dst, msk)
n = ((roi[2] + 63) // 64) * ((roi[3] + 7) // 8)
buf = np.zeros((n, 8), np.uint64)
ctx.lib.mis_debug_warp_stamps(buf.ctypes.data_as(C.c_void_p), n)
s = buf[:, :6].astype(np.int64)
ok = (s > 0).all(axis=1)
s = s[ok]
print("tiles", n, "with all stamps", ok.sum())
names = ["map(k+1)", "classify+copies issue", "wait copies(k)", "gather(k)", "stage+stores(k)"]
d = np.diff(s, axis=1)
for i, nm in enumerate(names):
    print("%-24s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f" % (nm, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
step = s[:, 5] - s[:, 0]
print("step                     mean %8.0f  p50 %8.0f  p90 %8.0f" % (step.mean(), *np.percentile(step, [50, 90])))
r = buf[ok][:, 6:8].astype(np.int64)
print("kernel wall span %.2f us" % ((r[:, 1].max() - r[:, 0].min()) / 100.0))
