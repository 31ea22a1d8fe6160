"""Tile classification counts of warp_fused_kernel (library built with -DMIS_WARP_STATS)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
for wl in ("config3", "config4"):
    cams = synth.workload(wl)
    scale = isa.Stitcher.warped_image_scale(cams)
    w = isa.SphericalWarper(ctx, scale)
    for idx in (0, len(cams) // 2):
        cam = cams[idx]
        frame = synth.render_frame_gpu(cam)
        roi = w.warpRoi((3840, 2160), cam["K"], cam["R"])
        dst, msk = w.alloc_fused(roi)
        out = (C.c_uint * 8)()
        ctx.lib.mis_debug_warp_stats(out, 1)
        w.warp_fused_into(frame, cam["K"], cam["R"], roi, dst, msk)
        ctx.lib.mis_debug_warp_stats(out, 1)
        print(wl, idx, "roi", roi, "interior %d folded %d cold/global %d generic-map %d not-overlapped %d" % tuple(out[:5]), "mask frac %.3f" % (msk.float().mean().item() / 255))
