"""Panorama job orchestration: one process per GPU, frames sharded in contiguous blocks.

SURVEY.md section 8(e).  The reference is a single process (image_stitching.cpp main()); this module
spreads the same sequence over N ranks:

  stage                     partition                         exchange (torch.distributed = RCCL over xGMI)
  detect + describe         frames, contiguous block / rank   none
  match + RANSAC            pairs dealt round-robin           all-gather of {keypoints, descriptors, counts}
  component pruning         replicated (n <= 64)              all-reduce (sum) of the n x n confidence matrix
  warp + blend accumulate   frames (same blocks)              none
  blend finalise            rank 0                            region gather of the Laplacian / weight pyramids

Blend exchange.  A rank's frames are a contiguous block of the sweep, so its pyramid contributions are
confined to one rectangle of the panorama (frame ROIs + the blender's 3 * 2^bands margin, aligned to
2^bands so the rectangle maps exactly onto every pyramid level).  Instead of an all-reduce of the whole
pyramid (which would also need int16 -> int32 widening: RCCL has no 16-bit integer type) each rank packs
only its rectangle of every level into one byte buffer, ONE gather brings the buffers to rank 0 over the
point-to-point xGMI links in parallel, and rank 0 adds them in rank order.  16SC3 Laplacian sums are
two's-complement wrap-around additions and therefore exact in any order; the f32 weight sums are added
in a fixed order (deterministic), differing from the single-GPU feed order in the last bit where >= 3
frames overlap -- inside the 1-LSB pixel tolerance of the north star.

The orchestration is engine-agnostic: `HipEngine` (the product) drives libmistitch through the C ABI;
the CPU tests inject an engine of their own to exercise the sharding / collective logic under gloo.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import _capi as capi
from . import stitching as st


def frame_block(n, rank, world):
    """Contiguous block of frame indices owned by `rank` (blocks differ by at most one frame)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


class Comm:
    """Thin wrapper over torch.distributed (nccl = RCCL on GPUs, gloo on CPU); identity when world = 1.
    With the gloo backend device tensors are staged through host memory (rehearsals of the N > 1 path on one GPU)."""

    def __init__(self, rank=0, world=1, group=None):
        self.rank, self.world, self.group = rank, world, group
        self._stage = False
        if world > 1:
            import torch.distributed as dist
            self._stage = dist.get_backend(group) == "gloo"

    def _h(self, t):
        return t.cpu() if (self._stage and t.is_cuda) else t

    def all_gather(self, t):
        if self.world == 1:
            return t.unsqueeze(0)
        import torch.distributed as dist
        src = self._h(t).contiguous()
        out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out.view(-1), src.view(-1), group=self.group)
        return out.to(t.device)

    def all_reduce_sum(self, t):
        if self.world > 1:
            import torch.distributed as dist
            h = self._h(t)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            if h is not t:
                t.copy_(h)
        return t

    def gather_to_root(self, t):
        """Equal-sized 1-D tensors -> list of world tensors on rank 0 (None elsewhere)."""
        if self.world == 1:
            return [t]
        import torch.distributed as dist
        h = self._h(t)
        bufs = [torch.empty_like(h) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(h, bufs, dst=0, group=self.group)
        return [b.to(t.device) for b in bufs] if bufs is not None else None


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy, no ownership)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def dev_tensor(ptr, shape, typestr, device):
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)


class HipEngine:
    """The product engine: every stage runs in libmistitch (HIP) through the C ABI."""

    def __init__(self, ctx, frame_size, config=None):
        self.ctx = ctx
        self.cfg = config or st.StitchConfig()
        self.frame_size = frame_size
        # finder per features_type (image_stitching.cpp:543-563): ORB, or SIFT (float descriptors -> the L2 matcher)
        self.finder = st.SiftFeatureFinder(ctx, frame_size) if self.cfg.features_type == "sift" else st.OrbFeatureFinder(ctx, frame_size)
        self.matcher = st.BestOf2NearestMatcher(ctx, self.cfg.match_conf)
        self.blender = None
        self._keep = []
        # warp + blend run in a context of their own (second stream): StitchJob composes speculatively while the
        # latency-bound RANSAC chains of the matcher leave most of the device idle
        # (normal priority: a low-priority stream was measured -- the compose work then finishes after the matcher, 20.1 ms)
        h = C.c_void_p()
        ctx.check(ctx.lib.mis_stream_create(ctx.device.index, 0, C.byref(h)))
        self._compose_stream_handle = h
        self.compose_stream = torch.cuda.ExternalStream(h.value, device=ctx.device)
        self.cctx = st.Context(ctx.device.index, stream=h.value)
        self.speculative_compose = True

    # ---- features ----
    def detect(self, frames):
        return self.finder.detect_batch(frames)

    def feature_counts(self, feats):
        return torch.tensor([len(f) for f in feats], dtype=torch.int32, device=self.ctx.device)

    def pack_features(self, feats, cap):
        """-> (kps u8 [m, cap*24], desc u8 [m, cap*row_bytes]) device tensors for the all-gather; `cap` = the largest
        keypoint count of any frame of the job (agreed by the ranks beforehand: SIFT has no fixed budget)."""
        m = len(feats)
        dev = self.ctx.device
        rb = self._row_bytes(feats)
        kps = torch.zeros((m, cap * 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((m, cap * rb), dtype=torch.uint8, device=dev)
        for i, f in enumerate(feats):
            n = len(f)
            if n:
                kps[i, : n * 24] = dev_tensor(f.raw.keypoints, (n * 24,), "|u1", dev)
                desc[i, : n * rb] = dev_tensor(f.raw.descriptors, (n * rb,), "|u1", dev)
        return kps, desc

    def _row_bytes(self, feats=None):
        return 512 if self.cfg.features_type == "sift" else 32

    def unpack_features(self, kps_all, desc_all, counts_all):
        """Gathered tensors [n, ...] -> ImageFeatures views (no copies; tensors kept alive)."""
        self._keep = [kps_all, desc_all, counts_all]
        counts = counts_all.cpu().tolist()
        w, h = self.frame_size
        sift = self.cfg.features_type == "sift"
        out = []
        for i, n in enumerate(counts):
            raw = capi.MisFeatures()
            raw.img_idx, raw.img_w, raw.img_h, raw.n = i, w, h, int(n)
            raw.keypoints = kps_all[i].data_ptr()
            raw.descriptors = desc_all[i].data_ptr()
            raw.desc_cols, raw.desc_dtype, raw.owner_ = (128, capi.F32, None) if sift else (32, capi.U8, None)
            out.append(st.ImageFeatures(self.ctx, raw))
        return out

    def match(self, feats, rank, world):
        return self.matcher(feats, rank, world)

    def confidence_tensor(self, pm, n, on_device=True):
        t = torch.from_numpy(pm.confidences()).view(n, n)
        return t.to(self.ctx.device) if on_device else t      # a single rank prunes on the host: no round trip through HBM

    def refine_cameras(self, feats, pm, indices, cams):
        return st.refine_cameras(self.ctx, feats, pm, indices, cams, self.cfg)

    # ---- compose ----
    def warp_roi(self, scale, cam):
        return st.warp_roi(scale, self.frame_size, cam["K"], cam["R"])

    def warp_rois(self, scale, cams):
        """warpRoi of every camera (image_stitching.cpp:1119-1140): one kernel on the compose stream, nothing cached."""
        return st.warp_rois(self.cctx, scale, self.frame_size, cams)

    def begin_compose(self, scale, corners, sizes):
        x, y, pw, ph = st.result_roi(corners, sizes)
        btype, bands, sharp = st.blend_config(self.cfg.blend_type, self.cfg.blend_strength, (pw, ph))
        key = (btype, bands, sharp)
        if self.blender is None or getattr(self, "_blender_key", None) != key:   # band count / sharpness are creation parameters
            self._blender_key = key
            self.blender = {capi.BLEND_MULTI_BAND: lambda: st.MultiBandBlender(self.cctx, bands),
                            capi.BLEND_FEATHER: lambda: st.FeatherBlender(self.cctx, sharp), capi.BLEND_NO: lambda: st.Blender(self.cctx)}[btype]()
        self.blender.prepare(corners, sizes)
        self.warper = st.SphericalWarper(self.cctx, scale)
        self.pano_size = (pw, ph)
        return btype, bands

    def warp_feed(self, frame, cam, roi):
        tl, img_s, mask = self.warper.warp_fused(frame, cam["K"], cam["R"], roi)
        self.blender.feed(img_s, mask, tl)

    def match_fence_target(self):
        """Sequence number the NEXT matcher call of this engine will carry (see compose_after_knn)."""
        return int(self.ctx.lib.mis_match_sequence(self.ctx.h)) + 1

    def compose_after_knn(self, target_seq):
        """Queue the compose stream behind the 2-NN pass of matcher call `target_seq` (made by another thread): that
        pass fills the device, the RANSAC chains after it do not -- composing from there on costs the matcher nothing
        (measured: 0.5 ms per 16 x 4K step against starting at once)."""
        import os
        self.ctx.check(self.ctx.lib.mis_match_knn_fence(self.ctx.h, None if os.environ.get('MIS_COMPOSE_GPU_FENCE') == '0' else self._compose_stream_handle, target_seq, 50))

    def warp_feed_many(self, frames, cams, rois):
        """warp_feed for a list of frames in one library call (no interpreter work between the launches: the thread that
        composes speculatively does not compete for the GIL with the thread that drives the matcher)."""
        self.blender.compose_frames(frames, self.warper.scale, cams, rois)

    def accumulators(self):
        """[(lap int16 [h, w*3], weight f32 [h, w])] tensor views of the blender's panorama pyramids."""
        out = []
        dev = self.ctx.device
        nb = self.ctx.lib.mis_blender_num_bands(self.blender.h)
        for l in range(nb + 1):
            w, h, lp, wp = C.c_int(), C.c_int(), C.c_void_p(), C.c_void_p()
            self.cctx.check(self.cctx.lib.mis_blender_level_info(self.blender.h, l, C.byref(w), C.byref(h), C.byref(lp), C.byref(wp)))
            out.append((dev_tensor(lp.value, (h.value, w.value * 3), "<i2", dev), dev_tensor(wp.value, (h.value, w.value), "<f4", dev)))
        return out

    def finalize(self):
        return self.blender.blend()

    def num_bands(self):
        return self.ctx.lib.mis_blender_num_bands(self.blender.h)

    def sync(self):
        torch.cuda.synchronize(self.ctx.device)


class StitchJob:
    """The hot-path sequence of main() (image_stitching.cpp:567-1228) for one panorama, sharded."""

    def __init__(self, ctx, frame_size, cameras, rank=0, world_size=1, group=None, engine=None, config=None, force_collectives=False):
        self.cfg = config or st.StitchConfig()
        if self.cfg.ba_cost_func != "no" and world_size > 1:
            raise NotImplementedError("bundle adjustment needs every pair's matches on one rank: run it with world_size 1")
        if self.cfg.expos_comp_type != "no" or self.cfg.seam_find_type != "no":
            # refused rather than ignored: the seam-scale step needs every warped image on one rank
            raise NotImplementedError("the sharded job composes without exposure compensation / seam finding; "
                                      "use Stitcher.compose (single GPU) for expos_comp_type / seam_find_type")
        self.engine = engine or HipEngine(ctx, frame_size, self.cfg)
        self.cams = cameras
        self.n = len(cameras)
        self.rank, self.world = rank, world_size
        self.comm = Comm(rank, world_size, group)
        self.my_frames = frame_block(self.n, rank, world_size)
        self.frame_size = frame_size
        self.scale = st.Stitcher.warped_image_scale(cameras)
        self.force_collectives = force_collectives   # run the pack / gather / reduce code even at world size 1 (tests)
        counts = {len(frame_block(self.n, r, world_size)) for r in range(world_size)}
        if len(counts) != 1:
            raise ValueError("the frame count must divide evenly over the ranks")

    # -- stages -------------------------------------------------------------------------------
    def stage_features(self, frames):
        return self.engine.detect([frames[i] for i in self.my_frames])

    def stage_gather(self, local_feats):
        if self.world == 1 and not self.force_collectives:
            return local_feats
        counts_all = self.comm.all_gather(self.engine.feature_counts(local_feats)).flatten(0, 1)
        cap = max(int(counts_all.max()), 1)
        kps, desc = self.engine.pack_features(local_feats, cap)
        kps_all = self.comm.all_gather(kps).flatten(0, 1)
        desc_all = self.comm.all_gather(desc).flatten(0, 1)
        return self.engine.unpack_features(kps_all, desc_all, counts_all)

    def stage_match(self, feats):
        pm = self.engine.match(feats, self.rank, self.world)
        if self.world == 1 and not self.force_collectives and isinstance(self.engine, HipEngine):
            return pm, self.engine.confidence_tensor(pm, self.n, on_device=False)
        conf = self.engine.confidence_tensor(pm, self.n)
        conf = self.comm.all_reduce_sum(conf)          # every pair is owned by exactly one rank
        return pm, conf

    def stage_prune(self, conf):
        return [int(i) for i in st.leaveBiggestComponentConf(conf.cpu().numpy().reshape(self.n, self.n), self.cfg.conf_thresh)]

    def stage_compose_prepare(self, indices):
        """blender sizing + prepare for the kept frames (image_stitching.cpp:1138, :1175-1192): needs the cameras only."""
        eng = self.engine
        # the reference takes the median focal of the KEPT cameras (image_stitching.cpp:746-748, :884-895)
        self.scale = st.Stitcher.warped_image_scale([self.cams[i] for i in indices])
        if hasattr(eng, "warp_rois"):
            rois = dict(zip(indices, eng.warp_rois(self.scale, [self.cams[i] for i in indices])))
        else:
            rois = {i: eng.warp_roi(self.scale, self.cams[i]) for i in indices}
        corners = [(rois[i][0], rois[i][1]) for i in indices]
        sizes = [(rois[i][2], rois[i][3]) for i in indices]
        btype, bands = eng.begin_compose(self.scale, corners, sizes)
        self._compose_indices, self._compose_rois = indices, rois
        return btype, bands

    def stage_compose(self, frames, indices, prepared=None):
        eng = self.engine
        btype, bands = prepared if prepared is not None else self.stage_compose_prepare(indices)
        rois = self._compose_rois
        mine = [i for i in self.my_frames if i in rois]
        if hasattr(eng, "warp_feed_many"):
            eng.warp_feed_many([frames[i] for i in mine], [self.cams[i] for i in mine], [rois[i] for i in mine])
        else:
            for i in mine:
                eng.warp_feed(frames[i], self.cams[i], rois[i])
        return btype, bands

    # -- blend exchange: region gather ---------------------------------------------------------
    def rank_region(self, r, indices, rois, bands, w0, h0):
        """Level-0 rectangle (x0, y0, x1, y1) of the panorama that rank r's frames can touch, or None."""
        mine = [i for i in frame_block(self.n, r, self.world) if i in rois]
        if not mine:
            return None
        px = min(rois[i][0] for i in indices)
        py = min(rois[i][1] for i in indices)
        gap, a = 3 << bands, (1 << bands) - 1
        x0 = max(0, min(rois[i][0] for i in mine) - px - gap) & ~a
        y0 = max(0, min(rois[i][1] for i in mine) - py - gap) & ~a
        x1 = min(w0, (max(rois[i][0] + rois[i][2] for i in mine) - px + gap + a) & ~a)
        y1 = min(h0, (max(rois[i][1] + rois[i][3] for i in mine) - py + gap + a) & ~a)
        return x0, y0, x1, y1

    @staticmethod
    def _level_rects(region, levels):
        """Per level (x0, y0, x1, y1) of a level-0 rectangle, clipped to the level size."""
        x0, y0, x1, y1 = region
        out = []
        for l, (lap, wgt) in enumerate(levels):
            h, w = wgt.shape
            out.append((min(x0 >> l, w), min(y0 >> l, h), min(-((-x1) >> l), w), min(-((-y1) >> l), h)))
        return out

    @staticmethod
    def _packed_size(rects):
        return sum(((x1 - x0) * (y1 - y0) * 10 + 31) // 16 * 16 for x0, y0, x1, y1 in rects)   # 6 B lap + 4 B weight, 16 B aligned parts

    @staticmethod
    def _pack(levels, rects, nbytes):
        dev = levels[0][0].device
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        o = 0
        for (lap, wgt), (x0, y0, x1, y1) in zip(levels, rects):
            m = (x1 - x0) * (y1 - y0)
            if m:
                buf[o:o + m * 6] = lap[y0:y1, 3 * x0:3 * x1].contiguous().view(torch.uint8).reshape(-1)
                ow = o + (m * 6 + 15) // 16 * 16
                buf[ow:ow + m * 4] = wgt[y0:y1, x0:x1].contiguous().view(torch.uint8).reshape(-1)
            o += (m * 10 + 31) // 16 * 16
        return buf

    @staticmethod
    def _add_packed(levels, rects, buf):
        o = 0
        for (lap, wgt), (x0, y0, x1, y1) in zip(levels, rects):
            m = (x1 - x0) * (y1 - y0)
            if m:
                lap[y0:y1, 3 * x0:3 * x1] += buf[o:o + m * 6].view(torch.int16).view(y1 - y0, 3 * (x1 - x0))     # wraps
                ow = o + (m * 6 + 15) // 16 * 16
                wgt[y0:y1, x0:x1] += buf[ow:ow + m * 4].view(torch.float32).view(y1 - y0, x1 - x0)
            o += (m * 10 + 31) // 16 * 16

    def stage_reduce(self):
        if self.world == 1 and not self.force_collectives:
            return
        levels = self.engine.accumulators()
        bands = len(levels) - 1
        h0, w0 = levels[0][1].shape
        indices, rois = self._compose_indices, self._compose_rois
        regions = [self.rank_region(r, indices, rois, bands, w0, h0) for r in range(self.world)]
        rects = [self._level_rects(g, levels) if g else None for g in regions]
        nbytes = max([self._packed_size(rc) for rc in rects[1:] if rc] + [16])
        if self.force_collectives and self.world == 1:
            # test mode: move this rank's own rectangle out, zero it, and add it back through the packed path
            buf = self._pack(levels, rects[0], self._packed_size(rects[0]))
            for (lap, wgt), (x0, y0, x1, y1) in zip(levels, rects[0]):
                lap[y0:y1, 3 * x0:3 * x1] = 0
                wgt[y0:y1, x0:x1] = 0
            self._add_packed(levels, rects[0], buf)
            return
        mine = rects[self.rank]
        if self.rank != 0 and mine:
            buf = self._pack(levels, mine, nbytes)
        else:
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=levels[0][0].device)
        bufs = self.comm.gather_to_root(buf)
        if self.rank == 0:
            for r in range(1, self.world):
                if rects[r]:
                    self._add_packed(levels, rects[r], bufs[r])

    def stage_finalize(self):
        if self.rank == 0:
            return self.engine.finalize()
        return None, None

    # -- whole job ---------------------------------------------------------------------------
    def _compose_on_side_stream(self, frames, indices, prepared=None):
        """stage_compose with the engine's compose stream current (allocations and launches belong to it)."""
        eng = self.engine
        with torch.cuda.stream(eng.compose_stream):
            return self.stage_compose(frames, indices, prepared)

    def run(self, frames):
        refine = self.cfg.ba_cost_func != "no"
        if refine and self.world > 1:
            raise NotImplementedError("bundle adjustment needs every pair's matches on one rank: run it with world_size 1")
        spec = getattr(self.engine, "speculative_compose", False) and not refine   # refined cameras: compose must wait
        prepared = None
        side = getattr(self.engine, "compose_stream", None)
        if side is not None:
            # the compose stream is non-blocking: order it behind whatever produced `frames` on the caller's stream
            side.wait_stream(torch.cuda.current_stream(self.engine.ctx.device))
        if spec:
            # the blender's prepare (sizing + zeroing ~200 MB of panorama pyramids, ~1.2 ms) depends on the cameras only:
            # it goes to the compose stream before anything else and runs under the feature stage
            with torch.cuda.stream(self.engine.compose_stream):
                prepared = self.stage_compose_prepare(list(range(self.n)))
        feats = self.stage_gather(self.stage_features(frames))
        if spec:
            # Speculation: almost always every frame survives the pruning, and warp + blend do not depend on the
            # matches otherwise (the cameras are inputs).  Compose for ALL frames on the second stream from a helper
            # thread (the library calls release the GIL) while the matcher runs; redo it if a frame was dropped.
            import threading
            everyone = list(range(self.n))
            box = {}

            fence = self.engine.match_fence_target() if hasattr(self.engine, "match_fence_target") else None

            # a single rank has no exchange between feed and finalise: the collapse of the pyramid is speculated too
            solo = self.world == 1 and not self.force_collectives

            def work():
                try:
                    if fence is not None:
                        self.engine.compose_after_knn(fence)
                    box["r"] = self._compose_on_side_stream(frames, everyone, prepared)
                    if solo:
                        with torch.cuda.stream(self.engine.compose_stream):
                            box["f"] = self.stage_finalize()
                except BaseException as e:   # re-raised on the caller's thread
                    box["e"] = e
            th = threading.Thread(target=work)
            th.start()
            try:
                pm, conf = self.stage_match(feats)
                indices = self.stage_prune(conf)
            finally:
                th.join()
            if "e" in box:
                raise box["e"]
            if indices == everyone and "f" in box:
                (btype, bands), (pano, mask) = box["r"], box["f"]
            else:
                btype, bands = box["r"] if indices == everyone else self._compose_on_side_stream(frames, indices)
                self.engine.compose_stream.synchronize()     # the accumulators are complete before any exchange
                with torch.cuda.stream(self.engine.compose_stream):
                    self.stage_reduce()
                    pano, mask = self.stage_finalize()
        else:
            pm, conf = self.stage_match(feats)
            indices = self.stage_prune(conf)
            if refine:
                if self.cfg.ba_cost_func != "reproj":
                    raise ValueError("ba_cost_func must be 'no' or 'reproj'")
                refined = self.engine.refine_cameras(feats, pm, indices, self.cams)
                self.cams = list(self.cams)
                for i, c in zip(indices, refined):
                    self.cams[i] = c
                self.scale = st.Stitcher.warped_image_scale([self.cams[i] for i in indices])
            btype, bands = self.stage_compose(frames, indices)
            self.stage_reduce()
            pano, mask = self.stage_finalize()
        self.engine.sync()      # the job's results are complete when run() returns
        return {"pano": pano, "mask": mask, "indices": indices, "confidence": conf, "matches": pm, "features": feats,
                "pano_size": self.engine.pano_size, "num_bands": bands}

    def breakdown(self, frames, reps=3):
        """Per-stage wall time (ms, host synchronised between stages) -- diagnostics, not the metric."""
        out = {}

        def timed(name, fn):
            self.engine.sync()
            t0 = time.perf_counter()
            r = fn()
            self.engine.sync()
            out[name] = out.get(name, 0.0) + (time.perf_counter() - t0) * 1e3 / reps
            return r
        for _ in range(reps):
            lf = timed("features (detect+describe)", lambda: self.stage_features(frames))
            feats = timed("feature all-gather", lambda: self.stage_gather(lf))
            pm, conf = timed("match + RANSAC", lambda: self.stage_match(feats))
            idx = timed("prune (host)", lambda: self.stage_prune(conf))
            side = getattr(self.engine, "compose_stream", None)

            def on_side(fn):
                if side is None:
                    return fn()
                with torch.cuda.stream(side):
                    return fn()
            timed("warp + blend feed", lambda: on_side(lambda: self.stage_compose(frames, idx)))
            timed("pyramid reduce", lambda: on_side(self.stage_reduce))
            timed("blend finalise", lambda: on_side(self.stage_finalize))
        return out
