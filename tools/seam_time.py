"""Where the reference-default pipeline's seam-scale step spends its time (config 3): python tools/seam_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import synth
import image_stitching_amd as isa
from image_stitching_amd import stitching as st
from image_stitching_amd import distributed as misdist
cams = synth.workload("config3")
ctx = isa.Context(0)
cfg = isa.StitchConfig()
job = misdist.StitchJob(ctx, (3840, 2160), cams, config=cfg)
frames = {i: synth.render_frame_gpu(cams[i]) for i in job.my_frames}
eng = job.engine
idx = list(range(job.n))


def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print("%-34s %8.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))
    return r


with torch.cuda.stream(eng.compose_stream):
    job.stage_compose_prepare(idx)
    local = timed("seam-scale warps (16)", lambda: eng.seam_local([frames[i] for i in idx], [cams[i] for i in idx], job.scale))
    corners = [it[0] for it in local]; images = [it[1] for it in local]; masks = [it[2] for it in local]
    print("seam-scale image", tuple(images[0].shape))
    comp = None
    if cfg.expos_comp_type != "no":
        def feed():
            c = st.BlocksGainCompensator(eng.cctx)
            c.feed(corners, images, masks)
            return c
        comp = timed("gain-blocks compensator feed", feed)
    finder = st.DpSeamFinder(eng.cctx)
    timed("DpSeamFinder(COLOR).find", lambda: finder.find(images, corners, [m.clone() for m in masks]))
    timed("seam_solve (feed + find)", lambda: eng.seam_global(corners, images, masks))
    rois = job._compose_rois

    def feeds():
        job.stage_compose_prepare(idx)
        for i in idx:
            eng.warp_feed_seam(frames[i], cams[i], rois[i], i)
    timed("16 x (warp, gains, seam mask, feed)", feeds)
    w = eng.warper
    tl, img_s, mask = w.warp_fused(frames[0], cams[0]["K"], cams[0]["R"], rois[0])
    compensator, seam_masks = eng._seam
    timed("  compensator.apply", lambda: compensator.apply(0, tl, img_s, mask))
    timed("  seam_mask_apply", lambda: st.seam_mask_apply(eng.cctx, seam_masks[0], mask))
    timed("  blender.feed", lambda: eng.blender.feed(img_s, mask, tl))
    timed("  warp_fused", lambda: w.warp_fused(frames[0], cams[0]["K"], cams[0]["R"], rois[0]))
