"""Panorama job orchestration: one process per GPU, frames sharded in contiguous blocks.

SURVEY.md section 8(e).  The reference is a single process (image_stitching.cpp main()); this module
spreads the same sequence over N ranks:

  stage                     partition                         exchange (torch.distributed = RCCL over xGMI)
  detect + describe         frames, contiguous block / rank   none
  match + RANSAC            pairs dealt round-robin           all-gather of {keypoints, descriptors, counts}
  component pruning         replicated (n <= 64)              all-reduce (sum) of the n x n confidence matrix
  warp + blend accumulate   frames (same blocks)              none
  blend exchange            column strips of the panorama     all-to-all of pyramid rectangles (each rank -> each strip owner)
  blend finalise            column strips (owner = rank k)    all-gather of the finished strips

Blend exchange (reduce-scatter by panorama column strips, local finalise, all-gather).  Rank k owns the columns
[c_k, c_k+1) of the padded panorama (boundaries multiples of 2^bands, so a strip maps exactly onto every pyramid level).
Finalising a strip needs the SUMMED accumulators of that strip plus pyrUp's halo (one coarse column either side per level,
accumulated: at most two columns) -- `need_l(k)` below.  A rank's frames are a contiguous block of the sweep, so its own
contributions are confined to one rectangle of the panorama (frame ROIs + the blender's 3 * 2^bands margin): rank r packs, for
every owner k, the intersection of that rectangle with need_l(k) at every level (mis_blender_pack_rects) and ONE all-to-all
moves the buffers over the point-to-point xGMI links, every link in parallel, each carrying ~1/N^2 of the pyramid set (an
all-reduce of the whole 1.4 GB set of config 4 would also need int16 -> int32 widening: RCCL has no 16-bit integer type).
The owner zeroes its need ranges and adds the N buffers in rank order (its own included: it travels through the same path):
16SC3 sums are two's-complement wrap-around additions, exact in any order; the f32 weight sums have the fixed association
((0 + p_0) + p_1) + ... over the per-rank partial sums p_r (each the sum of that rank's frames in feed order), which the CPU
tests reproduce with the oracle bit for bit, and which differs from the single-GPU feed order in the last bit where >= 3 frames
of different ranks overlap (inside the 1-LSB pixel tolerance of the north star).  Every rank then runs normalise + collapse +
crop on its strip only (mis_blender_blend_columns: work per rank ~ 1/N) and the finished strips are all-gathered.  Pack, add,
zero, the strip finalise, the strip copies and the feature packing are kernels / copies of the library: no torch arithmetic.

The orchestration is engine-agnostic: `HipEngine` (the product) drives libmistitch through the C ABI;
the CPU tests inject an engine of their own to exercise the sharding / collective logic under gloo.
"""
import ctypes as C
import os
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

    def __init__(self, rank=0, world=1, group=None, always_collective=False):
        self.rank, self.world, self.group = rank, world, group
        self._stage = False
        # always_collective: issue the collectives even in a group of one rank (a one-GPU rehearsal of the RCCL calls of the N > 1 path:
        # dtypes, split sizes, stream ordering); needs an initialised process group
        self._solo = world == 1 and not always_collective
        if not self._solo:
            import torch.distributed as dist
            self._stage = dist.get_backend(group) == "gloo"

    def _h(self, t):
        return t.cpu() if (self._stage and t.is_cuda) else t

    def all_gather(self, t):
        if self._solo:
            return t.unsqueeze(0)
        import torch.distributed as dist
        src = self._h(t).contiguous()
        out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out.view(-1), src.view(-1), group=self.group)
        return out.to(t.device)

    def all_reduce_sum(self, t):
        if not self._solo:
            import torch.distributed as dist
            h = self._h(t)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            if h is not t:
                t.copy_(h)
        return t

    def all_to_all_bytes(self, send, recv_sizes):
        """send: list of world 1-D uint8 tensors (send[k] goes to rank k) -> list of world tensors (recv[r] came from rank r,
        recv_sizes[r] bytes).  One collective (NCCL / RCCL all_to_all_single with split sizes); isend / irecv pairs under gloo."""
        if self._solo:
            return [send[0]]
        import torch.distributed as dist
        dev = send[0].device
        hs = [self._h(t).contiguous() for t in send]
        hdev = hs[0].device
        recv = [torch.empty(int(n), dtype=torch.uint8, device=hdev) for n in recv_sizes]
        if self._stage:
            ops = []
            for k in range(self.world):
                if k == self.rank:
                    recv[k].copy_(hs[k])
                    continue
                if hs[k].numel():
                    ops.append(dist.P2POp(dist.isend, hs[k], k, group=self.group))
                if recv[k].numel():
                    ops.append(dist.P2POp(dist.irecv, recv[k], k, group=self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        else:
            inp, out = torch.cat(hs), torch.empty(int(sum(recv_sizes)), dtype=torch.uint8, device=hdev)
            dist.all_to_all_single(out, inp, [int(n) for n in recv_sizes], [int(t.numel()) for t in hs], group=self.group)
            o = 0
            for k, n in enumerate(recv_sizes):
                recv[k] = out[o:o + int(n)]
                o += int(n)
        return [r.to(dev) for r in recv]

    def all_gather_objects(self, obj):
        """Small host-side records of every rank (pickled by torch.distributed) -> list in rank order."""
        if self._solo:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def gather_to_root(self, t):
        """Equal-sized 1-D tensors -> list of world tensors on rank 0 (None elsewhere)."""
        if self._solo:
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
        self.cfg = config or st.StitchConfig.hot_path()
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
        ctx.check(ctx.lib.mis_stream_create(ctx.device.index, int(os.environ.get('MIS_COMPOSE_PRIO', '0')), C.byref(h)))      # (MIS_COMPOSE_PRIO > 0: the least urgent priority -- measured in round 4, eight runs of 150 steps: no difference)
        self._compose_stream_handle = h
        self.compose_stream = torch.cuda.ExternalStream(h.value, device=ctx.device)
        self.cctx = st.Context(ctx.device.index, stream=h.value)
        self.exchange_device = ctx.device     # where the buffers of the N > 1 exchanges live (distributed.stage_seam)
        self.speculative_compose = True

    # ---- features ----
    def detect(self, frames):
        return self.finder.detect_batch(frames)

    def feature_counts(self, feats):
        return torch.tensor([len(f) for f in feats], dtype=torch.int32, device=self.ctx.device)

    def pack_features(self, feats, cap):
        """-> (kps u8 [m, cap*24], desc u8 [m, cap*row_bytes]) device tensors for the all-gather; `cap` = the largest
        keypoint count of any frame of the job (agreed by the ranks beforehand: SIFT has no fixed budget).  Device-to-device
        copies inside the library (mis_features_pack)."""
        m = len(feats)
        dev = self.ctx.device
        rb = self._row_bytes(feats)
        kps = torch.empty((m, cap * 24), dtype=torch.uint8, device=dev)
        desc = torch.empty((m, cap * rb), dtype=torch.uint8, device=dev)
        if m:
            arr = (capi.MisFeatures * m)()
            for k, f in enumerate(feats):
                C.memmove(C.byref(arr[k]), C.byref(f.raw), C.sizeof(capi.MisFeatures))
            self.ctx.check(self.ctx.lib.mis_features_pack(self.ctx.h, arr, m, int(cap), int(rb), C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr())))
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

    def match_entries(self, pm, conf_thresh):
        """What bundle adjustment needs of this rank's pairs (i < j): (i, j, matches, inliers_mask, num_inliers, has_H, H,
        confidence); the match arrays only for pairs above the confidence threshold (the others are never read)."""
        out = []
        n = pm.n
        for i in range(n):
            for j in range(i + 1, n):
                raw = pm._mis[i * n + j]
                if raw.src_img_idx < 0:
                    continue                       # a pair of another rank (or without keypoints)
                m = pm[i * n + j]
                strong = m.confidence > conf_thresh
                out.append((i, j, m.matches if strong else m.matches[:0], m.inliers_mask if strong else m.inliers_mask[:0], m.num_inliers, m.H is not None,
                            m.H, m.confidence))
        return out

    def matches_from_entries(self, per_rank, n):
        return st.PairwiseMatches.from_entries(self.ctx, n, [e for entries in per_rank for e in entries])

    # ---- seam-scale step (exposure compensation + seam finder) ----
    def seam_local(self, frames, cams, scale):
        """This rank's frames at seam scale -> [(corner, image 8UC3 tensor, mask 8U tensor)] (compose stream)."""
        return [st.seam_scale_warp(self.cctx, self.cfg, self.frame_size, f, c, scale) for f, c in zip(frames, cams)]

    def seam_pack(self, item, cap):
        """(corner, image, mask) -> one uint8 tensor of cap * 4 bytes (image, then mask) for the all-gather."""
        _, iw, mw = item
        buf = torch.zeros(cap * 4, dtype=torch.uint8, device=self.ctx.device)
        n = mw.numel()
        buf[:3 * n] = iw.reshape(-1)
        buf[3 * cap:3 * cap + n] = mw.reshape(-1)
        return buf

    def seam_unpack(self, buf, w, h, cap):
        return buf[:3 * w * h].view(h, w, 3), buf[3 * cap:3 * cap + w * h].view(h, w)

    def seam_global(self, corners, images, masks):
        """Gains and seam masks from every kept frame's seam-scale image (same inputs on every rank: same result)."""
        self._seam = st.seam_solve(self.cctx, self.cfg, corners, images, masks)

    def warp_feed_seam_many(self, frames, cams, rois, ks):
        """warp_feed_seam for a list of frames (ks: their indices among the kept frames): one batched warp, gains and seam mask per
        frame, one batched feed -- the same accumulators as the per-frame sequence."""
        if not frames:       # every frame of this rank was pruned: nothing to warp or feed (the collectives around still run)
            return
        warped = self.warper.warp_fused_batch(frames, cams, rois)
        compensator, seam_masks = self._seam
        for (tl, img_s, mask), k in zip(warped, ks):
            if compensator is not None:
                compensator.apply(k, tl, img_s, mask)
            st.seam_mask_apply(self.cctx, seam_masks[k], mask)
        self.blender.feed_batch([w[1] for w in warped], [w[2] for w in warped], [w[0] for w in warped])

    def warp_feed_seam(self, frame, cam, roi, k):
        """warp -> gains of image k -> seam mask of image k -> feed (image_stitching.cpp:1154-1171, :1218)."""
        tl, img_s, mask = self.warper.warp_fused(frame, cam["K"], cam["R"], roi)
        compensator, seam_masks = self._seam
        if compensator is not None:
            compensator.apply(k, tl, img_s, mask)
        st.seam_mask_apply(self.cctx, seam_masks[k], mask)
        self.blender.feed(img_s, mask, tl)

    # ---- compose ----
    def warp_roi(self, scale, cam, size=None):
        return st.warp_roi(scale, size or self.frame_size, cam["K"], cam["R"])

    def warp_rois(self, scale, cams, size=None):
        """warpRoi of every camera (image_stitching.cpp:1119-1140): one kernel on the compose stream, nothing cached."""
        return st.warp_rois(self.cctx, scale, size or self.frame_size, cams)

    def resize_frame(self, frame, f):
        """cv::resize(full_img, img, Size(), f, f, INTER_LINEAR_EXACT) (image_stitching.cpp:1143-1146) on the compose stream."""
        return st.resize(self.cctx, frame, fx=f, fy=f)

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

    def on_match_enqueued(self, fn):
        """fn() runs inside this engine's NEXT matcher call, on the calling thread, once that call's device work is enqueued
        (mis_match_on_enqueued): the thread is idle from there until the device finishes.  fn=None clears a pending hook."""
        if fn is None:
            self._enqueued_fn = None
            self.ctx.lib.mis_match_on_enqueued(self.ctx.h, None, None)
            return
        # ONE ctypes callback object per instance, created on first use: a CFUNCTYPE instance is a reference cycle that only the
        # cyclic garbage collector frees, and a fresh one per step kept that step's closure -- and through it the step's panorama --
        # alive while bench.py times with the collector paused (config 5: +1 GB of allocator growth per step, round 4)
        self._enqueued_fn = fn
        if getattr(self, "_enqueued_cb", None) is None:
            self._enqueued_cb = C.CFUNCTYPE(None, C.c_void_p)(self._run_enqueued)
        self.ctx.check(self.ctx.lib.mis_match_on_enqueued(self.ctx.h, C.cast(self._enqueued_cb, C.c_void_p), None))

    def _run_enqueued(self, _user):
        fn, self._enqueued_fn = getattr(self, "_enqueued_fn", None), None
        if fn is not None:
            fn()

    def compose_after_knn(self, target_seq):
        """Queue the compose stream behind the 2-NN pass of matcher call `target_seq` (made by another thread): that
        pass fills the device, the RANSAC chains after it do not -- composing from there on costs the matcher nothing
        (measured: 0.5 ms per 16 x 4K step against starting at once)."""
        import os
        rc = self.ctx.lib.mis_match_knn_fence(self.ctx.h, None if os.environ.get('MIS_COMPOSE_GPU_FENCE') == '0' else self._compose_stream_handle, target_seq, 50)
        if rc == 1:      # MIS_FENCE_TIMEOUT: the matcher call did not show up (a first call that allocates its arenas can take longer)
            # no fence, only less overlap control: the frames' producer was ordered before the compose stream at run() entry
            self.fence_timeouts = getattr(self, "fence_timeouts", 0) + 1
        elif rc != 0:
            self.ctx.check(rc)

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

    # ---- blend exchange (library kernels on the compose stream) ----
    def level_sizes(self):
        out = []
        for l in range(self.num_bands() + 1):
            w, h = C.c_int(), C.c_int()
            self.cctx.check(self.cctx.lib.mis_blender_level_info(self.blender.h, l, C.byref(w), C.byref(h), None, None))
            out.append((w.value, h.value))
        return out

    @staticmethod
    def _rect_array(rects):
        return (capi.MisLevelRect * max(len(rects), 1))(*[capi.MisLevelRect(*r) for r in rects])

    def pack_rects(self, rects, nbytes):
        buf = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.ctx.device)
        if rects:
            self.cctx.check(self.cctx.lib.mis_blender_pack_rects(self.blender.h, self._rect_array(rects), len(rects), C.c_void_p(buf.data_ptr()), buf.numel()))
        return buf[:int(nbytes)]

    def add_rects(self, rects, buf):
        if rects:
            self.cctx.check(self.cctx.lib.mis_blender_add_rects(self.blender.h, self._rect_array(rects), len(rects), C.c_void_p(buf.data_ptr()), buf.numel()))

    def zero_rects(self, rects):
        if rects:
            self.cctx.check(self.cctx.lib.mis_blender_zero_rects(self.blender.h, self._rect_array(rects), len(rects)))

    def finalize_columns(self, x0, x1):
        """blend() for the panorama columns x0 .. x1 - 1 -> (16SC3 strip, mask strip)."""
        w, h = self.blender._size
        x1 = min(x1, w)
        dst = st._empty_image(self.cctx, h, x1 - x0, 3, torch.int16)
        msk = st._empty_image(self.cctx, h, x1 - x0, 1, torch.uint8)
        d, m = st.as_image(dst), st.as_image(msk)
        self.cctx.check(self.cctx.lib.mis_blender_blend_columns(self.blender.h, int(x0), int(x1), C.byref(d), C.byref(m)))
        return dst, msk

    def strip_to_bytes(self, img, msk, cap_w):
        """A finished strip as one flat byte tensor of the common size (strip widths differ by at most 2^bands)."""
        h, w = msk.shape
        out = torch.zeros(h * cap_w * 7, dtype=torch.uint8, device=self.ctx.device)
        lib, cx = self.cctx.lib, self.cctx
        cx.check(lib.mis_copy_2d(cx.h, C.c_void_p(out.data_ptr()), cap_w * 6, C.c_void_p(img.data_ptr()), img.stride(0) * 2, w * 6, h))
        cx.check(lib.mis_copy_2d(cx.h, C.c_void_p(out.data_ptr() + h * cap_w * 6), cap_w, C.c_void_p(msk.data_ptr()), msk.stride(0), w, h))
        return out

    def assemble(self, strips, bounds, cap_w, size):
        """All ranks' strip byte tensors [world, h * cap_w * 7] -> (panorama 16SC3, mask) device tensors."""
        w, h = size
        pano = st._empty_image(self.cctx, h, w, 3, torch.int16)
        mask = st._empty_image(self.cctx, h, w, 1, torch.uint8)
        lib, cx = self.cctx.lib, self.cctx
        for k, (x0, x1) in enumerate(bounds):
            x1 = min(x1, w)
            if x1 <= x0:
                continue
            base = strips[k].data_ptr()
            cx.check(lib.mis_copy_2d(cx.h, C.c_void_p(pano.data_ptr() + x0 * 6), pano.stride(0) * 2, C.c_void_p(base), cap_w * 6, (x1 - x0) * 6, h))
            cx.check(lib.mis_copy_2d(cx.h, C.c_void_p(mask.data_ptr() + x0), mask.stride(0), C.c_void_p(base + h * cap_w * 6), cap_w, x1 - x0, h))
        return pano, mask

    def sync(self):
        torch.cuda.synchronize(self.ctx.device)


class StitchJob:
    """The hot-path sequence of main() (image_stitching.cpp:567-1228) for one panorama, sharded."""

    def __init__(self, ctx, frame_size, cameras, rank=0, world_size=1, group=None, engine=None, config=None, force_collectives=False, always_collective=False):
        self.cfg = config or st.StitchConfig.hot_path()
        st.check_seam_config(self.cfg)
        self.engine = engine or HipEngine(ctx, frame_size, self.cfg)
        self.cams = cameras
        self.cams0 = cameras          # the caller's cameras; self.cams holds the refined ones after a run with bundle adjustment
        self.n = len(cameras)
        self.rank, self.world = rank, world_size
        self.comm = Comm(rank, world_size, group, always_collective)
        self.my_frames = frame_block(self.n, rank, world_size)
        self.frame_size = frame_size
        self.scale = st.Stitcher.warped_image_scale(cameras)
        self.force_collectives = force_collectives   # run the pack / gather / reduce code even at world size 1 (tests)
        self.seam_needed = self.cfg.expos_comp_type != "no" or self.cfg.seam_find_type != "no"
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
        # compose scale (:1105-1140): warper scale, intrinsics and frame size of the compositing loop
        g = st.compose_geometry(self.cfg, self.frame_size, self.scale)
        self._geom = g
        ccams = {i: st.scaled_camera(self.cams[i], g.aspect) for i in indices} if g.aspect != 1.0 else {i: self.cams[i] for i in indices}
        if g.size != tuple(self.frame_size) and not hasattr(eng, "resize_frame"):
            raise NotImplementedError("this engine does not resize frames (compose_megapix > 0 needs resize_frame)")
        if hasattr(eng, "warp_rois"):
            rois = dict(zip(indices, eng.warp_rois(g.warp_scale, [ccams[i] for i in indices], g.size)))
        else:
            rois = {i: eng.warp_roi(g.warp_scale, ccams[i], g.size) for i in indices}
        corners = [(rois[i][0], rois[i][1]) for i in indices]
        sizes = [(rois[i][2], rois[i][3]) for i in indices]
        btype, bands = eng.begin_compose(g.warp_scale, corners, sizes)
        self._compose_indices, self._compose_rois, self._compose_cams = indices, rois, ccams
        return btype, bands

    def stage_compose(self, frames, indices, prepared=None):
        eng = self.engine
        btype, bands = prepared if prepared is not None else self.stage_compose_prepare(indices)
        rois, ccams, g = self._compose_rois, self._compose_cams, self._geom
        mine = [i for i in self.my_frames if i in rois]
        # the loop's frames at compose scale (:1143-1146: INTER_LINEAR_EXACT when |compose_scale - 1| > 0.1)
        cframes = {i: eng.resize_frame(frames[i], g.compose_scale) for i in mine} if g.size != tuple(self.frame_size) else frames
        if self.seam_needed:
            marks = getattr(self, "marks", None)
            if marks is not None:
                marks.append(("compose: sized", time.perf_counter()))
            self.stage_seam(frames, indices)
            if marks is not None:
                marks.append(("compose: seams solved", time.perf_counter()))
            if hasattr(eng, "warp_feed_seam_many"):
                eng.warp_feed_seam_many([cframes[i] for i in mine], [ccams[i] for i in mine], [rois[i] for i in mine], [indices.index(i) for i in mine])
            else:
                for i in mine:
                    eng.warp_feed_seam(cframes[i], ccams[i], rois[i], indices.index(i))
            return btype, bands
        if hasattr(eng, "warp_feed_many"):
            eng.warp_feed_many([cframes[i] for i in mine], [ccams[i] for i in mine], [rois[i] for i in mine])
        else:
            for i in mine:
                eng.warp_feed(cframes[i], ccams[i], rois[i])
        return btype, bands

    def stage_seam(self, frames, indices):
        """The seam-scale step for the kept frames: every rank warps its own frames at seam scale, the small images are
        all-gathered (0.1 MP each), every rank solves the gains / seams of all of them (host + small kernels, deterministic)."""
        eng = self.engine
        mine = [i for i in self.my_frames if i in indices]
        local = eng.seam_local([frames[i] for i in mine], [self.cams[i] for i in mine], self.scale)
        if getattr(self, "marks", None) is not None:
            self.marks.append(("compose: seam-scale warps enqueued", time.perf_counter()))
        if self.world == 1 and not self.force_collectives:
            items = local
        else:
            per = len(self.my_frames)          # equal on every rank; ranks with dropped frames pad with empty records
            meta = torch.zeros((per, 5), dtype=torch.int32)
            for k, (i, it) in enumerate(zip(mine, local)):
                meta[k] = torch.tensor([i, it[0][0], it[0][1], it[2].shape[1], it[2].shape[0]], dtype=torch.int32)
            meta[len(mine):, 0] = -1
            # the device the exchange buffers live on is the engine's, also when this rank has no frame left after the pruning
            # (a CPU tensor handed to an RCCL all-gather raises on this rank and leaves the others in the collective)
            dev = torch.device(getattr(eng, "exchange_device", None) or (local[0][1].device if local else "cpu"))
            meta_all = self.comm.all_gather(meta.to(dev)).flatten(0, 1).cpu()
            cap = max(int((meta_all[:, 3] * meta_all[:, 4]).max()), 1)
            bufs = [eng.seam_pack(it, cap) for it in local] + [torch.zeros(cap * 4, dtype=torch.uint8, device=dev) for _ in range(per - len(mine))]
            all_bufs = self.comm.all_gather(torch.stack(bufs)).flatten(0, 1)
            by_index = {}
            for row, buf in zip(meta_all.tolist(), all_bufs):
                if row[0] >= 0:
                    iw, mw = eng.seam_unpack(buf, row[3], row[4], cap)
                    by_index[row[0]] = ((row[1], row[2]), iw, mw)
            items = [by_index[i] for i in indices]
        eng.seam_global([it[0] for it in items], [it[1] for it in items], [it[2] for it in items])

    # -- blend exchange: column strips ----------------------------------------------------------
    def rank_region(self, r, indices, rois, bands, w0, h0):
        """Level-0 rectangle (x0, y0, x1, y1) of the padded panorama that rank r's frames can touch, or None."""
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
    def strip_bounds(w0, bands, world):
        """Column strips [c_k, c_k+1) of the padded panorama (w0 a multiple of 2^bands), boundaries multiples of 2^bands."""
        q = 1 << bands
        nq = w0 // q
        c = [(k * nq // world) * q for k in range(world)] + [w0]
        return [(c[k], c[k + 1]) for k in range(world)]

    @staticmethod
    def need_ranges(strip, level_sizes):
        """Columns [lo, hi) of every level whose summed accumulators the owner of `strip` needs: the strip itself and, level by
        level, the coarse columns pyrUp reads for the level below (mis_blender_blend_columns uses the same recurrence)."""
        lo, hi = strip
        out = [(lo, hi)]
        for l in range(1, len(level_sizes)):
            lo, hi = max(0, (lo >> 1) - 1), min(level_sizes[l][0], ((hi - 1) >> 1) + 2)
            out.append((lo, hi))
        return out

    @staticmethod
    def exchange_rects(region, need, level_sizes):
        """Rectangles (level, x0, y0, x1, y1, offset) of `region` (a rank's level-0 rectangle) inside the columns `need`, and
        the packed size: per level the 16SC3 block then the f32 block, each rounded up to 16 bytes."""
        rects, off = [], 0
        if region is None:
            return rects, 0
        rx0, ry0, rx1, ry1 = region
        for l, ((lw, lh), (nlo, nhi)) in enumerate(zip(level_sizes, need)):
            x0, x1 = max(min(rx0 >> l, lw), nlo), min(min(-((-rx1) >> l), lw), nhi)
            y0, y1 = min(ry0 >> l, lh), min(-((-ry1) >> l), lh)
            if x1 <= x0 or y1 <= y0:
                continue
            m = (x1 - x0) * (y1 - y0)
            rects.append((l, x0, y0, x1, y1, off))
            off += (m * 6 + 15) // 16 * 16 + (m * 4 + 15) // 16 * 16
        return rects, off

    def stage_reduce_finalize(self):
        """Strip exchange + per-strip finalise + assembly -> (pano, mask) on rank 0 (None elsewhere)."""
        eng = self.engine
        sizes = eng.level_sizes()
        bands = len(sizes) - 1
        w0, h0 = sizes[0]
        indices, rois = self._compose_indices, self._compose_rois
        world = self.world
        bounds = self.strip_bounds(w0, bands, world)
        needs = [self.need_ranges(bnd, sizes) for bnd in bounds]
        regions = [self.rank_region(r, indices, rois, bands, w0, h0) for r in range(world)]
        plan = [[self.exchange_rects(regions[r], needs[k], sizes) for k in range(world)] for r in range(world)]   # plan[src][dst]
        me = self.rank
        send = [eng.pack_rects(*plan[me][k]) for k in range(world)]
        recv = self.comm.all_to_all_bytes(send, [plan[r][me][1] for r in range(world)])
        full_h = [(l, lo, 0, hi, sizes[l][1], 0) for l, (lo, hi) in enumerate(needs[me]) if hi > lo]
        eng.zero_rects(full_h)
        for r in range(world):                      # fixed association of the f32 sums: rank order
            eng.add_rects(plan[r][me][0], recv[r])
        pw, ph = eng.pano_size
        x0, x1 = bounds[me][0], min(bounds[me][1], pw)
        cap_w = max(min(b1, pw) - b0 for b0, b1 in bounds)
        if x1 > x0:
            img, msk = eng.finalize_columns(x0, x1)
            mine = eng.strip_to_bytes(img, msk, cap_w)
        else:                                       # more ranks than 2^bands-wide column groups: this rank owns nothing
            mine = torch.zeros(ph * cap_w * 7, dtype=torch.uint8, device=send[0].device)
        strips = self.comm.all_gather(mine)
        if self.rank != 0:
            return None, None
        return eng.assemble(strips, bounds, cap_w, (pw, ph))

    def stage_finalize(self):
        if self.rank == 0:
            return self.engine.finalize()
        return None, None

    def stage_exchange_finalize(self):
        """Single rank: blend().  N ranks (or force_collectives): strip exchange, per-strip finalise, assembly."""
        if self.world == 1 and not self.force_collectives:
            return self.stage_finalize()
        side = getattr(self.engine, "compose_stream", None)
        if side is None:
            return self.stage_reduce_finalize()
        with torch.cuda.stream(side):        # the library's kernels and torch.distributed's collectives meet on this stream
            return self.stage_reduce_finalize()

    # -- whole job ---------------------------------------------------------------------------
    def _compose_on_side_stream(self, frames, indices, prepared=None):
        """stage_compose with the engine's compose stream current (allocations and launches belong to it)."""
        eng = self.engine
        if getattr(eng, "compose_stream", None) is None:
            return self.stage_compose(frames, indices, prepared)
        with torch.cuda.stream(eng.compose_stream):
            return self.stage_compose(frames, indices, prepared)

    def run(self, frames):
        marks = self.marks = [("start", time.perf_counter())] if os.environ.get("MIS_JOB_TRACE") else None   # host time stamps (diagnostics)

        def mark(name):
            if marks is not None:
                marks.append((name, time.perf_counter()))
        refine = self.cfg.ba_cost_func != "no"
        # refined cameras: compose must wait for the matcher.  A seam-scale step needs the kept set too, but on a single rank it is
        # speculated like the composition itself (all frames kept is the rule; redone when the pruning drops one): its ~30 ms of
        # host work (DP seams) and small kernels then run under the matcher's 5 ms instead of behind them
        solo_rank = self.world == 1 and not self.force_collectives
        spec = getattr(self.engine, "speculative_compose", False) and not refine and (not self.seam_needed or solo_rank)
        prepared = None
        side = getattr(self.engine, "compose_stream", None)
        if side is not None:
            # the compose stream is non-blocking: order it behind whatever produced `frames` on the caller's stream
            side.wait_stream(torch.cuda.current_stream(self.engine.ctx.device))
        if spec:
            # the blender's prepare (warpRoi of all cameras: a kernel + a synchronisation of the compose stream; sizing; zeroing ~200 MB
            # of panorama pyramids) depends on the cameras only.  With an ORB finder it runs from the finder's hook, once the feature
            # batch is enqueued (in front of the features it kept the main stream idle for 0.13 ms of every step); otherwise before them
            finder = getattr(self.engine, "finder", None)
            box0 = {}

            def prep():
                try:
                    with torch.cuda.stream(self.engine.compose_stream):
                        box0["p"] = self.stage_compose_prepare(list(range(self.n)))
                except BaseException as e:   # re-raised on the caller's thread
                    box0["e"] = e
            hooked = hasattr(finder, "on_enqueued") and len(self.my_frames) > 0
            if hooked:
                finder.on_enqueued(prep)
            else:
                prep()
            try:
                local = self.stage_features(frames)
            finally:
                if hooked:
                    finder.on_enqueued(None)
            if "p" not in box0 and "e" not in box0:
                prep()              # the finder returned before its hook
            if "e" in box0:
                raise box0["e"]
            prepared = box0["p"]
            feats = self.stage_gather(local)
        else:
            feats = self.stage_gather(self.stage_features(frames))
        mark("features")
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
                    mark("compose: fence passed")
                    box["r"] = self._compose_on_side_stream(frames, everyone, prepared)
                    mark("compose: enqueued")
                    if solo:
                        with torch.cuda.stream(self.engine.compose_stream):
                            box["f"] = self.stage_finalize()
                    mark("compose: finalise enqueued")
                except BaseException as e:   # re-raised on the caller's thread
                    box["e"] = e
            if hasattr(self.engine, "on_match_enqueued"):
                # no second thread: the composition is enqueued from inside the matcher call, by the thread that would otherwise
                # only wait there for the device (a helper thread cost 0.1 ms to start and contended for the interpreter lock)
                self.engine.on_match_enqueued(work)
                try:
                    pm, conf = self.stage_match(feats)
                finally:
                    self.engine.on_match_enqueued(None)
                mark("match returned")
                if not box:
                    work()              # a matcher call without pairs returns before its hook
                indices = self.stage_prune(conf)
                mark("pruned")
            else:
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
                with torch.cuda.stream(self.engine.compose_stream):
                    pano, mask = self.stage_exchange_finalize()
            box.clear()          # (the hook's closure may outlive this call: it must not hold a panorama)
        else:
            pm, conf = self.stage_match(feats)
            indices = self.stage_prune(conf)
            if refine:
                if self.cfg.ba_cost_func != "reproj":
                    raise ValueError("ba_cost_func must be 'no' or 'reproj'")
                if self.world > 1 or self.force_collectives:
                    # the adjuster needs every connected pair's inlier matches, and the pairs were dealt over the ranks: every
                    # rank contributes the entries it owns, all ranks assemble the same table and run the same (host) solver
                    pm = self.engine.matches_from_entries(self.comm.all_gather_objects(self.engine.match_entries(pm, self.cfg.conf_thresh)), self.n)
                refined = self.engine.refine_cameras(feats, pm, indices, self.cams0)     # every run starts from the cameras the job was given
                self.cams = list(self.cams0)
                for i, c in zip(indices, refined):
                    self.cams[i] = c
                self.scale = st.Stitcher.warped_image_scale([self.cams[i] for i in indices])
            btype, bands = self._compose_on_side_stream(frames, indices)     # (torch allocations of the seam step on the library's stream)
            pano, mask = self.stage_exchange_finalize()
        self.engine.sync()      # the job's results are complete when run() returns
        mark("synchronised")
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
            timed("blend exchange + finalise", lambda: on_side(self.stage_exchange_finalize))
        return out
