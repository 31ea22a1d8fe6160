"""CPU engine for the distributed-orchestration tests: the same engine interface as
image_stitching_amd.distributed.HipEngine, implemented with the oracle (test infrastructure).
It exists so that world_size-2 gloo runs can exercise the sharding / collective logic on CPU."""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

import oracle


@dataclass
class _MI:
    confidence: float = 0.0
    src_img_idx: int = -1
    dst_img_idx: int = -1
    num_inliers: int = 0


class OracleEngine:
    CAP = 4000 + 128 * 8

    def __init__(self, frame_size, blend_type=oracle.BLEND_MULTI_BAND, blend_strength=5.0, match_conf=0.32, config=None):
        self.cfg = config
        self.frame_size = frame_size
        self.orb = oracle.Orb(*frame_size)
        self.blend_type, self.blend_strength, self.match_conf = blend_type, blend_strength, match_conf
        self.blender = None
        self.pano_size = None

    def detect(self, frames):
        w, h = self.frame_size
        out = []
        for f in frames:
            k, d = self.orb.run(np.asarray(f))
            out.append(dict(img_w=w, img_h=h, kps=k, xy=np.stack([k["x"], k["y"]], 1), desc=d))
        return out

    def feature_counts(self, feats):
        return torch.tensor([len(f["kps"]) for f in feats], dtype=torch.int32)

    def pack_features(self, feats, cap):
        m = len(feats)
        kps = torch.zeros((m, cap * 24), dtype=torch.uint8)
        desc = torch.zeros((m, cap * 32), dtype=torch.uint8)
        for i, f in enumerate(feats):
            n = len(f["kps"])
            kps[i, : n * 24] = torch.from_numpy(np.frombuffer(f["kps"].tobytes(), np.uint8).copy())
            desc[i, : n * 32] = torch.from_numpy(f["desc"].reshape(-1).copy())
        return kps, desc

    def unpack_features(self, kps_all, desc_all, counts_all):
        w, h = self.frame_size
        out = []
        for i, n in enumerate(counts_all.tolist()):
            k = np.frombuffer(kps_all[i, : n * 24].numpy().tobytes(), oracle.KP_DTYPE).copy()
            d = desc_all[i, : n * 32].numpy().reshape(n, 32).copy()
            out.append(dict(img_w=w, img_h=h, kps=k, xy=np.stack([k["x"], k["y"]], 1), desc=d))
        return out

    def match(self, feats, rank, world):
        n = len(feats)
        out = [_MI() for _ in range(n * n)]
        p = oracle.match_default_params(match_conf=self.match_conf)
        pair = 0
        for i in range(n):
            for j in range(i + 1, n):
                if len(feats[i]["kps"]) == 0 or len(feats[j]["kps"]) == 0:
                    continue
                mine = pair % world == rank
                pair += 1
                if not mine:
                    continue
                r = oracle.match_pair(feats[i], feats[j], p)
                out[i * n + j] = _MI(r["confidence"], i, j, r["num_inliers"])
                out[j * n + i] = _MI(r["confidence"], j, i, r["num_inliers"])
        return out

    def confidence_tensor(self, pm, n):
        return torch.tensor([m.confidence for m in pm], dtype=torch.float64).view(n, n)

    def warp_roi(self, scale, cam, size=None):
        w, h = size or self.frame_size
        return oracle.warp_roi(scale, w, h, cam["K"].astype(np.float32), cam["R"].astype(np.float32))

    def resize_frame(self, frame, f):
        return oracle.resize_exact(np.asarray(frame), fx=f, fy=f)

    def begin_compose(self, scale, corners, sizes):
        x0 = min(c[0] for c in corners); y0 = min(c[1] for c in corners)
        x1 = max(c[0] + s[0] for c, s in zip(corners, sizes)); y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
        btype, bands, sharp = oracle.blend_config(self.blend_type, self.blend_strength, x1 - x0, y1 - y0)
        self.blender = oracle.Blender(btype, bands, sharp)
        self.blender.prepare(corners, sizes)
        self.scale = scale
        self.pano_size = (x1 - x0, y1 - y0)
        return btype, self.blender.num_bands

    def warp_feed(self, frame, cam, roi):
        K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
        f = np.asarray(frame)
        img, tl = oracle.warp_spherical(f, self.scale, K, R)
        msk, _ = oracle.warp_spherical(np.full(f.shape[:2], 255, np.uint8), self.scale, K, R, oracle.INTER_NEAREST, oracle.BORDER_CONSTANT)
        self.blender.feed(img.astype(np.int16), msk, tl)

    # ---- seam-scale step (oracle twins of seam_scale_warp / seam_solve) ----
    def seam_local(self, frames, cams, scale):
        w, h = self.frame_size
        seam_scale = min(1.0, float(np.sqrt(self.cfg.seam_megapix * 1e6 / (w * h))))
        swa = np.float32(seam_scale)
        sscale = float(np.float32(np.float32(scale) * swa))
        out = []
        for f, cam in zip(frames, cams):
            f = np.asarray(f)
            img = oracle.resize_exact(f, fx=seam_scale, fy=seam_scale) if seam_scale < 1 else f
            K = cam["K"].astype(np.float32).copy()
            K[0, 0] *= swa; K[0, 2] *= swa; K[1, 1] *= swa; K[1, 2] *= swa
            R = cam["R"].astype(np.float32)
            wi, tl = oracle.warp_spherical(img, sscale, K, R)
            wm, _ = oracle.warp_spherical(np.full(img.shape[:2], 255, np.uint8), sscale, K, R, oracle.INTER_NEAREST, oracle.BORDER_CONSTANT)
            out.append((tl, torch.from_numpy(wi), torch.from_numpy(wm)))
        return out

    def seam_pack(self, item, cap):
        _, iw, mw = item
        buf = torch.zeros(cap * 4, dtype=torch.uint8)
        n = mw.numel()
        buf[:3 * n] = iw.reshape(-1)
        buf[3 * cap:3 * cap + n] = mw.reshape(-1)
        return buf

    def seam_unpack(self, buf, w, h, cap):
        return buf[:3 * w * h].view(h, w, 3), buf[3 * cap:3 * cap + w * h].view(h, w)

    def seam_global(self, corners, images, masks):
        iw = [np.ascontiguousarray(i.numpy()) for i in images]
        mw = [np.ascontiguousarray(m.numpy()) for m in masks]
        comp = None
        if self.cfg.expos_comp_type == "gain_blocks":
            comp = oracle.Compensator(self.cfg.expos_comp_block_size, self.cfg.expos_comp_block_size, self.cfg.expos_comp_nr_filtering)
            comp.feed(corners, iw, mw)
        if self.cfg.seam_find_type == "voronoi":
            mw = oracle.voronoi_seams(corners, mw)
        elif self.cfg.seam_find_type == "dp_color":
            mw = oracle.dp_seams(iw, corners, mw)
        self._seam = (comp, mw)

    def warp_feed_seam(self, frame, cam, roi, k):
        K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
        f = np.asarray(frame)
        img, tl = oracle.warp_spherical(f, self.scale, K, R)
        msk, _ = oracle.warp_spherical(np.full(f.shape[:2], 255, np.uint8), self.scale, K, R, oracle.INTER_NEAREST, oracle.BORDER_CONSTANT)
        comp, seam_masks = self._seam
        if comp is not None:
            img = comp.apply(k, img)
        msk = oracle.seam_mask_apply(seam_masks[k], msk)
        self.blender.feed(img.astype(np.int16), msk, tl)

    def accumulators(self):
        """In-place tensor views of the oracle blender's pyramids (shared memory with the C arrays)."""
        out = []
        L = oracle.lib()
        for l in range(self.blender.num_bands + 1):
            w, h = C.c_int(), C.c_int()
            lp = L.mo_blender_level_lap(self.blender.h_, l, C.byref(w), C.byref(h))
            wp = L.mo_blender_level_weight(self.blender.h_, l, C.byref(w), C.byref(h))
            lap = np.ctypeslib.as_array(C.cast(lp, C.POINTER(C.c_int16)), shape=(h.value, w.value * 3))
            wgt = np.ctypeslib.as_array(C.cast(wp, C.POINTER(C.c_float)), shape=(h.value, w.value))
            out.append((torch.from_numpy(lap), torch.from_numpy(wgt)))
        return out

    def finalize(self):
        return self.blender.blend()

    # ---- blend exchange (numpy twins of mis_blender_pack_rects / add_rects / zero_rects / blend_columns) ----
    def level_sizes(self):
        return [(w.shape[1], w.shape[0]) for _, w in self.accumulators()]

    def pack_rects(self, rects, nbytes):
        buf = np.zeros(max(int(nbytes), 16), np.uint8)
        acc = self.accumulators()
        for l, x0, y0, x1, y1, off in rects:
            lap, wgt = acc[l][0].numpy(), acc[l][1].numpy()
            m = (x1 - x0) * (y1 - y0)
            buf[off:off + m * 6] = np.ascontiguousarray(lap[y0:y1, 3 * x0:3 * x1]).view(np.uint8).reshape(-1)
            ow = off + (m * 6 + 15) // 16 * 16
            buf[ow:ow + m * 4] = np.ascontiguousarray(wgt[y0:y1, x0:x1]).view(np.uint8).reshape(-1)
        return torch.from_numpy(buf[:int(nbytes)].copy())

    def add_rects(self, rects, buf):
        b = buf.numpy()
        acc = self.accumulators()
        for l, x0, y0, x1, y1, off in rects:
            lap, wgt = acc[l][0].numpy(), acc[l][1].numpy()
            m = (x1 - x0) * (y1 - y0)
            lap[y0:y1, 3 * x0:3 * x1] += b[off:off + m * 6].view(np.int16).reshape(y1 - y0, 3 * (x1 - x0))      # wraps
            ow = off + (m * 6 + 15) // 16 * 16
            wgt[y0:y1, x0:x1] += b[ow:ow + m * 4].view(np.float32).reshape(y1 - y0, x1 - x0)

    def zero_rects(self, rects):
        acc = self.accumulators()
        for l, x0, y0, x1, y1, _ in rects:
            acc[l][0].numpy()[y0:y1, 3 * x0:3 * x1] = 0
            acc[l][1].numpy()[y0:y1, x0:x1] = 0

    def finalize_columns(self, x0, x1):
        # blend() of the whole pyramid: outside this strip's need ranges the sums are partial and the result is junk, the
        # strip's own columns only depend on the need ranges
        pano, mask = self.blender.blend()
        return pano[:, x0:x1].copy(), mask[:, x0:x1].copy()

    def strip_to_bytes(self, img, msk, cap_w):
        h, w = msk.shape
        out = np.zeros(h * cap_w * 7, np.uint8)
        a = out[:h * cap_w * 6].view(np.int16).reshape(h, cap_w * 3)
        a[:, :3 * w] = img.reshape(h, 3 * w)
        out[h * cap_w * 6:].reshape(h, cap_w)[:, :w] = msk
        return torch.from_numpy(out)

    def assemble(self, strips, bounds, cap_w, size):
        w, h = size
        pano, mask = np.zeros((h, w, 3), np.int16), np.zeros((h, w), np.uint8)
        for k, (x0, x1) in enumerate(bounds):
            x1 = min(x1, w)
            if x1 <= x0:
                continue
            b = strips[k].numpy()
            pano[:, x0:x1] = b[:h * cap_w * 6].view(np.int16).reshape(h, cap_w, 3)[:, :x1 - x0]
            mask[:, x0:x1] = b[h * cap_w * 6:].reshape(h, cap_w)[:, :x1 - x0]
        return pano, mask

    def num_bands(self):
        return self.blender.num_bands

    def sync(self):
        pass
