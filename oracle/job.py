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
