import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import synth
import image_stitching_amd as isa
from image_stitching_amd import stitching as st, _capi as capi
cams = synth.workload("config3")
ctx = isa.Context(0)
f = synth.render_frame_gpu(cams[3])
def timed(name, fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    print("%-30s %8.1f us" % (name, (time.perf_counter() - t0) / reps * 1e6)); return r
s = min(1.0, float(np.sqrt(0.1e6 / (3840 * 2160))))
img = timed("resize 4K -> seam", lambda: st.resize(ctx, f, fx=s, fy=s))
scale = isa.Stitcher.warped_image_scale(cams)
w = isa.SphericalWarper(ctx, np.float32(np.float32(scale) * np.float32(s)))
K = np.array(cams[3]["K"], np.float32).copy(); K[0,0]*=s; K[0,2]*=s; K[1,1]*=s; K[1,2]*=s
R = np.asarray(cams[3]["R"], np.float32)
timed("warp_roi (host)", lambda: st.warp_roi(w.scale, (img.shape[1], img.shape[0]), K, R))
timed("warp linear", lambda: w.warp(img, K, R, capi.INTER_LINEAR, capi.BORDER_REFLECT))
full = timed("torch.full", lambda: torch.full((img.shape[0], img.shape[1]), 255, dtype=torch.uint8, device=ctx.device))
timed("warp nearest mask", lambda: w.warp(full, K, R, capi.INTER_NEAREST, capi.BORDER_CONSTANT))
cfg = isa.StitchConfig()
timed("seam_scale_warp", lambda: st.seam_scale_warp(ctx, cfg, (3840, 2160), f, cams[3], scale))
# the same through a second context on a stream of its own (what the job's engine does)
from image_stitching_amd import distributed as misdist
eng = misdist.HipEngine(ctx, (3840, 2160), cfg)
with torch.cuda.stream(eng.compose_stream):
    timed("seam_scale_warp (compose ctx)", lambda: st.seam_scale_warp(eng.cctx, cfg, (3840, 2160), f, cams[3], scale))
    timed("  resize", lambda: st.resize(eng.cctx, f, fx=s, fy=s))
    w2 = isa.SphericalWarper(eng.cctx, w.scale)
    timed("  warp linear", lambda: w2.warp(img, K, R, capi.INTER_LINEAR, capi.BORDER_REFLECT))
    timed("  torch.full", lambda: torch.full((img.shape[0], img.shape[1]), 255, dtype=torch.uint8, device=ctx.device))
    timed("  torch.empty", lambda: torch.empty((229, 384, 3), dtype=torch.uint8, device=ctx.device))
frames = [synth.render_frame_gpu(c) for c in cams]
torch.cuda.synchronize()
with torch.cuda.stream(eng.compose_stream):
    for rep in range(2):
        ts = []
        keep = []
        for fr, c in zip(frames, cams):
            t0 = time.perf_counter()
            keep.append(st.seam_scale_warp(eng.cctx, cfg, (3840, 2160), fr, c, scale))
            ts.append((time.perf_counter() - t0) * 1e6)
        t0 = time.perf_counter(); torch.cuda.synchronize(); tsync = (time.perf_counter() - t0) * 1e6
        print("per-frame host us:", " ".join("%.0f" % t for t in ts), "| final sync %.0f us" % tsync)
