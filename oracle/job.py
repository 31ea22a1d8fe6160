"""ORACLE (test infrastructure): the whole hot-path sequence of main() for one panorama on the CPU.

image_stitching/image_stitching.cpp:567-1228 with compose_megapix = -1 and the cameras supplied by the caller:
features (:613) -> all pairs 2-NN + RANSAC (:653) -> myLeaveBiggestComponent (:661) -> median focal of the kept
cameras (:884-895) -> warpRoi (:1138) -> warp image + mask (:1154-1164) -> multiband feed (:1218) -> blend (:1225).
Used by the end-to-end parity tests and by bench.py's cpu_baseline leg (the timed spans mirror the reference's
print sites plus features and matching).  Never imported by the product.
"""
import time

import numpy as np

from . import bindings as o


def warped_image_scale(cams):
    focals = sorted(float(c["K"][1][1]) for c in cams)
    n = len(focals)
    if n % 2 == 1:
        return float(np.float32(focals[n // 2]))
    return float(np.float32(focals[n // 2 - 1] + focals[n // 2]) * np.float32(0.5))


def stitch_job(frames, cams, conf_thresh=0.95, match_conf=0.32, blend_type=o.BLEND_MULTI_BAND, blend_strength=5.0,
               keep_warped=False):
    """frames: list of (H, W, 3) uint8 arrays -> dict(features, matches, confidence, indices, scale, rois, pano, mask,
    num_bands, pano_size, spans_s)."""
    n = len(frames)
    H, W = frames[0].shape[:2]
    spans = {}
    t0 = time.perf_counter()
    orb = o.Orb(W, H)
    feats = []
    for f in frames:
        k, d = orb.run(np.ascontiguousarray(f))
        feats.append(dict(img_w=W, img_h=H, kps=k, xy=np.stack([k["x"], k["y"]], 1), desc=d))
    spans["features"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pm = o.match_all_pairs(feats, o.match_default_params(match_conf=match_conf))
    conf = np.array([m["confidence"] for m in pm], np.float64).reshape(n, n)
    indices = [int(i) for i in o.leave_biggest_component(conf, conf_thresh)]
    spans["matching"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    kept = [cams[i] for i in indices]
    scale = warped_image_scale(kept)
    rois = [o.warp_roi(scale, W, H, c["K"].astype(np.float32), c["R"].astype(np.float32)) for c in kept]
    corners = [(r[0], r[1]) for r in rois]
    sizes = [(r[2], r[3]) for r in rois]
    x0 = min(c[0] for c in corners); y0 = min(c[1] for c in corners)
    x1 = max(c[0] + s[0] for c, s in zip(corners, sizes)); y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
    btype, bands, sharp = o.blend_config(blend_type, blend_strength, x1 - x0, y1 - y0)
    bl = o.Blender(btype, bands, sharp)
    bl.prepare(corners, sizes)
    warped = []
    full = np.full((H, W), 255, np.uint8)
    for i, c in zip(indices, kept):
        K, R = c["K"].astype(np.float32), c["R"].astype(np.float32)
        img, tl = o.warp_spherical(frames[i], scale, K, R)
        msk, _ = o.warp_spherical(full, scale, K, R, o.INTER_NEAREST, o.BORDER_CONSTANT)
        img_s = img.astype(np.int16)
        bl.feed(img_s, msk, tl)
        if keep_warped:
            warped.append((img_s, msk, tl))
    pano, mask = bl.blend()
    spans["compositing"] = time.perf_counter() - t0
    return {"features": feats, "matches": pm, "confidence": conf, "indices": indices, "scale": scale, "rois": rois,
            "pano": pano, "mask": mask, "num_bands": bl.num_bands, "pano_size": (x1 - x0, y1 - y0), "spans_s": spans,
            "warped": warped}


def stitch_job_reference(frames, cams, conf_thresh=0.95, match_conf=0.32, blend_type=o.BLEND_MULTI_BAND, blend_strength=5.0, seam_megapix=0.1,
                         compose_megapix=0.4, ba_refine_mask="_____", wave_correct="horiz", block_size=64, nr_filtering=2):
    """What the reference's main() runs with its globals untouched (image_stitching/image_stitching.cpp:49-85), on frames and start
    cameras supplied by the caller (the EXIF path of :340-528 stands outside): features at full resolution (:613) -> all pairs
    (:653) -> myLeaveBiggestComponent (:661) -> BundleAdjusterReproj with the refinement mask "_____" on the kept subset
    (:680-713) -> waveCorrect HORIZ (:721-729) -> median focal (:884-895) -> seam-scale resize + warps (:604-622, :973-990) ->
    BlocksGainCompensator feed (:1002-1023) -> DpSeamFinder COLOR (:1056-1065) -> the compositing loop at compose scale
    (:1086-1220: intrinsics and warper scale times compose_work_aspect, frames resized INTER_LINEAR_EXACT, warp, gains, seam mask
    dilate -> resize -> AND, MultiBandBlender feed) -> blend (:1225).
    -> dict(indices, confidence, cameras (refined, kept), scale, seam_masks, gain_maps, rois, pano, mask, num_bands, pano_size)."""
    n = len(frames)
    H, W = frames[0].shape[:2]
    orb = o.Orb(W, H)
    feats = []
    for f in frames:
        k, d = orb.run(np.ascontiguousarray(f))
        feats.append(dict(img_w=W, img_h=H, kps=k, xy=np.stack([k["x"], k["y"]], 1), desc=d))
    pm = o.match_all_pairs(feats, o.match_default_params(match_conf=match_conf))
    conf = np.array([m["confidence"] for m in pm], np.float64).reshape(n, n)
    indices = [int(i) for i in o.leave_biggest_component(conf, conf_thresh)]
    k = len(indices)
    # the kept subset, re-indexed (:215-278), into the adjuster
    sub = []
    for a, i in enumerate(indices):
        for b, j in enumerate(indices):
            m = dict(pm[i * n + j])
            m["src_img_idx"], m["dst_img_idx"] = a, b
            sub.append(m)
    start = [dict(focal=float(cams[i]["K"][0, 0]), aspect=float(cams[i]["K"][1, 1] / cams[i]["K"][0, 0]), ppx=float(cams[i]["K"][0, 2]),
                  ppy=float(cams[i]["K"][1, 2]), R=np.asarray(cams[i]["R"], np.float64)) for i in indices]
    refined, _ = o.bundle_adjust_reproj([feats[i] for i in indices], sub, start, conf_thresh, ba_refine_mask)
    if wave_correct != "no":
        for c, R in zip(refined, o.wave_correct([c["R"] for c in refined], 1 if wave_correct == "vert" else 0)):
            c["R"] = R
    focals = sorted(c["focal"] for c in refined)
    scale = float(np.float32(focals[k // 2])) if k % 2 == 1 else float(np.float32(focals[k // 2 - 1] + focals[k // 2]) * np.float32(0.5))
    # ---- seam scale ----
    seam_scale = min(1.0, float(np.sqrt(seam_megapix * 1e6 / (W * H))))
    swa = np.float32(seam_scale)
    sscale = float(np.float32(np.float32(scale) * swa))
    s_corners, s_imgs, s_masks = [], [], []
    for i, c in zip(indices, refined):
        f = np.ascontiguousarray(frames[i])
        img = o.resize_exact(f, fx=seam_scale, fy=seam_scale) if seam_scale < 1 else f
        K = np.array([[c["focal"], 0, c["ppx"]], [0, c["focal"] * c["aspect"], c["ppy"]], [0, 0, 1]], np.float64).astype(np.float32)
        K[0, 0] *= swa; K[0, 2] *= swa; K[1, 1] *= swa; K[1, 2] *= swa
        R = np.asarray(c["R"], np.float64).astype(np.float32)
        wi, tl = o.warp_spherical(img, sscale, K, R)
        wm, _ = o.warp_spherical(np.full(img.shape[:2], 255, np.uint8), sscale, K, R, o.INTER_NEAREST, o.BORDER_CONSTANT)
        s_corners.append(tl); s_imgs.append(wi); s_masks.append(wm)
    comp = o.Compensator(block_size, block_size, nr_filtering)
    comp.feed(s_corners, s_imgs, s_masks)
    seam_masks = o.dp_seams(s_imgs, s_corners, s_masks)
    # ---- compose scale (:1105-1146) ----
    cs = min(1.0, float(np.sqrt(compose_megapix * 1e6 / (W * H)))) if compose_megapix > 0 else 1.0
    wscale = float(np.float32(scale) * np.float32(cs))
    resized = abs(cs - 1) > 1e-1
    cw, ch = (int(round(W * cs)), int(round(H * cs))) if resized else (W, H)
    Ks, Rs = [], []
    for c in refined:
        f = c["focal"] * cs
        Ks.append(np.array([[f, 0, c["ppx"] * cs], [0, f * c["aspect"], c["ppy"] * cs], [0, 0, 1]], np.float64).astype(np.float32))
        Rs.append(np.asarray(c["R"], np.float64).astype(np.float32))
    rois = [o.warp_roi(wscale, cw, ch, K, R) for K, R in zip(Ks, Rs)]
    corners = [(r[0], r[1]) for r in rois]
    sizes = [(r[2], r[3]) for r in rois]
    x0 = min(c[0] for c in corners); y0 = min(c[1] for c in corners)
    x1 = max(c[0] + s[0] for c, s in zip(corners, sizes)); y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
    btype, bands, sharp = o.blend_config(blend_type, blend_strength, x1 - x0, y1 - y0)
    bl = o.Blender(btype, bands, sharp)
    bl.prepare(corners, sizes)
    for q, i in enumerate(indices):
        f = np.ascontiguousarray(frames[i])
        img = o.resize_exact(f, fx=cs, fy=cs) if resized else f
        wi, tl = o.warp_spherical(img, wscale, Ks[q], Rs[q])
        wm, _ = o.warp_spherical(np.full(img.shape[:2], 255, np.uint8), wscale, Ks[q], Rs[q], o.INTER_NEAREST, o.BORDER_CONSTANT)
        wi = comp.apply(q, wi)
        wm = o.seam_mask_apply(seam_masks[q], wm)
        bl.feed(wi.astype(np.int16), wm, tl)
    pano, mask = bl.blend()
    return {"indices": indices, "confidence": conf, "cameras": refined, "scale": scale, "seam_masks": seam_masks,
            "gain_maps": [comp.gain_map(q).copy() for q in range(k)], "rois": rois, "pano": pano, "mask": mask, "num_bands": bl.num_bands,
            "pano_size": (x1 - x0, y1 - y0), "features": feats}
