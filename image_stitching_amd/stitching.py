"""Host-side mirror of the reference's stitching interface over the libmistitch C ABI.

Names and argument meaning follow the cv::detail objects that image_stitching.cpp's main() drives
(file:line cited per class).  Images are torch CUDA tensors (zero-copy device pointers) or numpy
arrays (host buffers staged by the library).
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _capi as capi

try:  # torch is plumbing: device memory + streams
    import torch
except Exception:  # pragma: no cover
    torch = None


class MisError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libmistitch error %d: %s" % (code, text))
        self.code = code


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4")])
DMATCH_DTYPE = np.dtype([("query_idx", "i4"), ("train_idx", "i4"), ("img_idx", "i4"), ("distance", "f4")])

_TORCH_DT = {}
if torch is not None:
    _TORCH_DT = {torch.uint8: capi.U8, torch.int16: capi.S16, torch.float32: capi.F32}
_NP_DT = {np.dtype(np.uint8): capi.U8, np.dtype(np.int16): capi.S16, np.dtype(np.float32): capi.F32}


def _mat9(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(9))
    return a, a.ctypes.data_as(C.c_void_p)


def as_image(a):
    """torch CUDA tensor / numpy array (H,W) or (H,W,C) -> MisImage describing it in place."""
    img = capi.MisImage()
    if torch is not None and isinstance(a, torch.Tensor):
        if a.dim() == 2:
            h, w = a.shape
            c = 1
            assert a.stride(1) == 1, "rows must be dense"
        else:
            h, w, c = a.shape
            assert a.stride(2) == 1 and a.stride(1) == c, "pixels must be dense within a row"
        img.data = a.data_ptr()
        img.stride = a.stride(0) * a.element_size()
        img.dtype = _TORCH_DT[a.dtype]
        img.mem = capi.MEM_DEVICE if a.is_cuda else capi.MEM_HOST
    else:
        a = np.asarray(a)
        if a.ndim == 2:
            h, w = a.shape
            c = 1
            assert a.strides[1] == a.itemsize
        else:
            h, w, c = a.shape
            assert a.strides[2] == a.itemsize and a.strides[1] == c * a.itemsize
        img.data = a.ctypes.data
        img.stride = a.strides[0]
        img.dtype = _NP_DT[a.dtype]
        img.mem = capi.MEM_HOST
    img.width, img.height, img.channels = w, h, c
    return img


def _empty_image(ctx, h, w, c, tdtype):
    """Device image with 256-byte aligned row pitch, returned as a (possibly strided) tensor view."""
    es = torch.empty((), dtype=tdtype).element_size()
    row = w * c * es
    pitch = (row + 255) // 256 * 256
    buf = torch.empty((h, pitch // es), dtype=tdtype, device=ctx.device)
    v = buf[:, : w * c]
    return v.view(h, w, c) if c > 1 else v


class Context:
    """One HIP device + one stream (include/mistitch.h: MisContext)."""

    def __init__(self, device=0, stream=None):
        self.lib = capi.load()
        if torch is None:
            raise ImportError("torch is required for device memory management")
        self.device = torch.device("cuda", device)
        if stream is None:
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream().cuda_stream
        self.stream = stream
        h = C.c_void_p()
        rc = self.lib.mis_context_create(device, C.c_void_p(stream), C.byref(h))
        if rc != capi.MIS_OK:
            raise MisError(rc, "mis_context_create failed (no HIP device? there is no CPU fallback)")
        self.h = h

    def check(self, rc):
        if rc != capi.MIS_OK:
            raise MisError(rc, self.lib.mis_last_error(self.h).decode())

    def synchronize(self):
        self.check(self.lib.mis_context_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mis_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# warp: cv::detail::SphericalWarper (image_stitching.cpp:973, :985, :988, :1117, :1138, :1154, :1159)
def warp_roi(scale, src_size, K, R):
    """RotationWarper::warpRoi(src_size, K, R) -> (x, y, width, height)."""
    lib = capi.load()
    _, kp = _mat9(K)
    _, rp = _mat9(R)
    r = capi.MisRect()
    rc = lib.mis_warp_roi(float(scale), int(src_size[0]), int(src_size[1]), kp, rp, C.byref(r))
    if rc != capi.MIS_OK:
        raise MisError(rc, "mis_warp_roi: invalid arguments")
    return r.x, r.y, r.width, r.height


def warp_rois(ctx, scale, src_size, cameras):
    """The warpRoi loop of main() (image_stitching.cpp:1119-1140) for all cameras in one library call: the border walks run
    in one small kernel on the context's stream -> [(x, y, width, height)]."""
    n = len(cameras)
    Ks = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float32).reshape(9) for c in cameras]))
    Rs = np.ascontiguousarray(np.stack([np.asarray(c["R"], np.float32).reshape(9) for c in cameras]))
    rr = (capi.MisRect * n)()
    ctx.check(ctx.lib.mis_warp_roi_batch(ctx.h, float(scale), int(src_size[0]), int(src_size[1]), n,
                                         Ks.ctypes.data_as(C.c_void_p), Rs.ctypes.data_as(C.c_void_p), rr))
    return [(r.x, r.y, r.width, r.height) for r in rr]


class SphericalWarper:
    """warper_creator->create(scale) (image_stitching.cpp:973, :1117)."""

    def __init__(self, ctx, scale):
        self.ctx, self.scale = ctx, float(scale)

    def warpRoi(self, src_size, K, R):
        return warp_roi(self.scale, src_size, K, R)

    def warp(self, src, K, R, interp=capi.INTER_LINEAR, border=capi.BORDER_REFLECT):
        """Point warp(src, K, R, interp, border, dst) -> (tl, dst)."""
        simg = as_image(src)
        x, y, w, h = warp_roi(self.scale, (simg.width, simg.height), K, R)
        dst = _empty_image(self.ctx, h, w, simg.channels, torch.uint8)
        dimg = as_image(dst)
        ka, kp = _mat9(K)
        ra, rp = _mat9(R)
        tl = capi.MisPoint()
        self.ctx.check(self.ctx.lib.mis_warp_spherical(self.ctx.h, C.byref(simg), self.scale, kp, rp, interp, border,
                                                       C.byref(dimg), C.byref(tl)))
        return (tl.x, tl.y), dst

    def alloc_fused(self, roi):
        """Output buffers of warp_fused for a roi (16SC3 image, 8U mask), 256-byte aligned rows."""
        x, y, w, h = roi
        return _empty_image(self.ctx, h, w, 3, torch.int16), _empty_image(self.ctx, h, w, 1, torch.uint8)

    def warp_fused_into(self, src_bgr, K, R, roi, dst, msk):
        simg, dimg, mimg = as_image(src_bgr), as_image(dst), as_image(msk)
        ka, kp = _mat9(K)
        ra, rp = _mat9(R)
        tl = capi.MisPoint()
        if roi is not None:      # the roi warpRoi gave for these parameters: the warp does not walk the border again
            rr = capi.MisRect(int(roi[0]), int(roi[1]), int(roi[2]), int(roi[3]))
            self.ctx.check(self.ctx.lib.mis_warp_spherical_fused_roi(self.ctx.h, C.byref(simg), self.scale, kp, rp, C.byref(rr),
                                                                     C.byref(dimg), C.byref(mimg), C.byref(tl)))
        else:
            self.ctx.check(self.ctx.lib.mis_warp_spherical_fused(self.ctx.h, C.byref(simg), self.scale, kp, rp, C.byref(dimg),
                                                                 C.byref(mimg), C.byref(tl)))
        return (tl.x, tl.y)

    def warp_fused_batch(self, imgs, cameras, rois):
        """mis_warp_spherical_fused_batch: the compose-scale step of main() for all frames (one grid per 16 frames)
        -> [(tl, img_warped_s, mask_warped)], the results of warp_fused per frame."""
        n = len(imgs)
        if n == 0:
            return []
        outs = [self.alloc_fused(r) for r in rois]
        im = (capi.MisImage * n)(*[as_image(i) for i in imgs])
        ds = (capi.MisImage * n)(*[as_image(o[0]) for o in outs])
        ms = (capi.MisImage * n)(*[as_image(o[1]) for o in outs])
        Ks = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float32).reshape(9) for c in cameras]))
        Rs = np.ascontiguousarray(np.stack([np.asarray(c["R"], np.float32).reshape(9) for c in cameras]))
        rs = (capi.MisRect * n)(*[capi.MisRect(int(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in rois])
        tls = (capi.MisPoint * n)()
        fp = C.POINTER(C.c_float)
        self.ctx.check(self.ctx.lib.mis_warp_spherical_fused_batch(self.ctx.h, im, n, float(self.scale), Ks.ctypes.data_as(fp), Rs.ctypes.data_as(fp), rs, ds, ms, tls))
        return [((t.x, t.y), o[0], o[1]) for t, o in zip(tls, outs)]

    def warp_fused_batch_timed(self, imgs, cameras, rois, dsts, masks, repeats):
        """mis_warp_spherical_fused_batch_timed: the fused warps of all frames in one grid per 16 frames, launched `repeats` times
        back to back between two HIP events on the context's stream -> average microseconds of one pass over all frames."""
        n = len(imgs)
        im = (capi.MisImage * n)(*[as_image(i) for i in imgs])
        ds = (capi.MisImage * n)(*[as_image(d) for d in dsts])
        ms = (capi.MisImage * n)(*[as_image(m) for m in masks])
        Ks = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float32).reshape(9) for c in cameras]))
        Rs = np.ascontiguousarray(np.stack([np.asarray(c["R"], np.float32).reshape(9) for c in cameras]))
        rs = (capi.MisRect * n)(*[capi.MisRect(int(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in rois])
        tls = (capi.MisPoint * n)()
        us = C.c_float()
        fp = C.POINTER(C.c_float)
        self.ctx.check(self.ctx.lib.mis_warp_spherical_fused_batch_timed(self.ctx.h, im, n, float(self.scale), Ks.ctypes.data_as(fp), Rs.ctypes.data_as(fp), rs, ds, ms, tls,
                                                                        int(repeats), C.byref(us)))
        return us.value

    def warp_fused_timed(self, src_bgr, K, R, roi, dst, msk, repeats):
        """Average duration (us) of the fused warp kernel over `repeats` back-to-back launches (HIP events)."""
        simg, dimg, mimg = as_image(src_bgr), as_image(dst), as_image(msk)
        ka, kp = _mat9(K)
        ra, rp = _mat9(R)
        tl, us = capi.MisPoint(), C.c_float()
        self.ctx.check(self.ctx.lib.mis_warp_spherical_fused_timed(self.ctx.h, C.byref(simg), self.scale, kp, rp, C.byref(dimg),
                                                                   C.byref(mimg), C.byref(tl), int(repeats), C.byref(us)))
        return float(us.value)

    def warp_fused(self, src_bgr, K, R, roi=None):
        """Compose-scale step of main(): warp(img, LINEAR, REFLECT) + warp(mask, NEAREST, CONSTANT) +
        convertTo(CV_16S) (image_stitching.cpp:1154-1164) -> (tl, img_warped_s, mask_warped)."""
        if roi is None:
            simg = as_image(src_bgr)
            roi = warp_roi(self.scale, (simg.width, simg.height), K, R)
        dst, msk = self.alloc_fused(roi)
        tl = self.warp_fused_into(src_bgr, K, R, roi, dst, msk)
        return tl, dst, msk


# ------------------------------------------------------------------------------------------------
# image operators either side of the path (image_stitching.cpp:571-580, :619, :1144, :1169-1171)
ROTATE_90_CLOCKWISE, ROTATE_180, ROTATE_90_COUNTERCLOCKWISE = 0, 1, 2


def resize(ctx, src, dsize=None, fx=0.0, fy=0.0):
    """cv::resize(src, dst, dsize, fx, fy, INTER_LINEAR_EXACT) on the device (8UC1 / 8UC3)."""
    simg = as_image(src)
    if dsize:
        dw, dh = int(dsize[0]), int(dsize[1])
    else:
        dw, dh = int(np.rint(simg.width * fx)), int(np.rint(simg.height * fy))   # cvRound: half to even
    dst = _empty_image(ctx, dh, dw, simg.channels, torch.uint8)
    dimg = as_image(dst)
    ctx.check(ctx.lib.mis_resize_linear_exact(ctx.h, C.byref(simg), dw if dsize else 0, dh if dsize else 0, float(fx), float(fy), C.byref(dimg)))
    return dst


def rotate(ctx, src, code):
    """cv::rotate(src, dst, code) on the device (8UC1 / 8UC3)."""
    simg = as_image(src)
    dw, dh = (simg.width, simg.height) if code == ROTATE_180 else (simg.height, simg.width)
    dst = _empty_image(ctx, dh, dw, simg.channels, torch.uint8)
    dimg = as_image(dst)
    ctx.check(ctx.lib.mis_rotate(ctx.h, C.byref(simg), int(code), C.byref(dimg)))
    return dst


def seam_mask_apply(ctx, seam_mask_warped, mask_warped):
    """mask_warped &= resize(dilate(seam_mask_warped), mask_warped.size(), INTER_LINEAR_EXACT), in place
    (image_stitching.cpp:1169-1171)."""
    simg, mimg = as_image(seam_mask_warped), as_image(mask_warped)
    ctx.check(ctx.lib.mis_seam_mask_apply(ctx.h, C.byref(simg), C.byref(mimg)))
    return mask_warped


class BlocksGainCompensator:
    """cv::detail::BlocksGainCompensator as the reference configures it (image_stitching.cpp:1002-1023): 64x64 blocks,
    one feed, two gain-filtering passes.  feed() takes the seam-scale warped images / masks and their corners;
    apply() multiplies an 8UC3 (or the fused warp's 16SC3) image by the gains, in place."""

    def __init__(self, ctx, bl_width=64, bl_height=64, nr_gain_filtering_iterations=2):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.mis_compensator_create(ctx.h, bl_width, bl_height, nr_gain_filtering_iterations, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:   # a destroyed context already released the device (the C object points into it)
            self.ctx.lib.mis_compensator_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, corners, images, masks):
        n = len(images)
        cs = (capi.MisPoint * n)(*[capi.MisPoint(int(c[0]), int(c[1])) for c in corners])
        im = (capi.MisImage * n)(*[as_image(i) for i in images])
        mk = (capi.MisImage * n)(*[as_image(m) for m in masks])
        self.ctx.check(self.ctx.lib.mis_compensator_feed(self.h, cs, im, mk, n))

    def gain_map(self, index):
        bx, by = C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.mis_compensator_gain_map(self.h, index, None, 0, C.byref(bx), C.byref(by)))
        out = np.zeros((by.value, bx.value), np.float32)
        self.ctx.check(self.ctx.lib.mis_compensator_gain_map(self.h, index, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, None, None))
        return out

    def apply(self, index, corner, image, mask=None):
        """corner and mask are accepted for signature parity (OpenCV ignores them too)."""
        img = as_image(image)
        self.ctx.check(self.ctx.lib.mis_compensator_apply(self.h, index, C.byref(img)))
        return image


class NoSeamFinder:
    """seam_find_type "no" (image_stitching.cpp:1029): masks stay as warped."""

    def find(self, images, corners, masks):
        return masks


class VoronoiSeamFinder:
    """seam_find_type "voronoi" (image_stitching.cpp:1031); images are not looked at."""

    def __init__(self, ctx):
        self.ctx = ctx

    def find(self, images, corners, masks):
        n = len(masks)
        cs = (capi.MisPoint * n)(*[capi.MisPoint(int(c[0]), int(c[1])) for c in corners])
        mk = (capi.MisImage * n)(*[as_image(m) for m in masks])
        self.ctx.check(self.ctx.lib.mis_seam_voronoi(self.ctx.h, cs, mk, n))
        return masks


class DpSeamFinder:
    """seam_find_type "dp_color" -- the reference's default (image_stitching.cpp:77, :1056-1057, :1065): DpSeamFinder(COLOR).
    images: the seam-scale warped 8UC3 images; masks are edited in place."""
    COLOR = 0

    def __init__(self, ctx, cost_func=0):
        self.ctx, self.cost_func = ctx, cost_func

    def find(self, images, corners, masks):
        n = len(masks)
        cs = (capi.MisPoint * n)(*[capi.MisPoint(int(c[0]), int(c[1])) for c in corners])
        im = (capi.MisImage * n)(*[as_image(i) for i in images])
        mk = (capi.MisImage * n)(*[as_image(m) for m in masks])
        self.ctx.check(self.ctx.lib.mis_seam_dp(self.ctx.h, cs, im, mk, n, int(self.cost_func)))
        return masks


# ------------------------------------------------------------------------------------------------
# blend: cv::detail::Blender / MultiBandBlender / FeatherBlender (image_stitching.cpp:1173-1225)
def result_roi(corners, sizes):
    lib = capi.load()
    n = len(corners)
    cs = (capi.MisPoint * n)(*[capi.MisPoint(int(c[0]), int(c[1])) for c in corners])
    ss = (capi.MisSize * n)(*[capi.MisSize(int(s[0]), int(s[1])) for s in sizes])
    r = capi.MisRect()
    rc = lib.mis_result_roi(cs, ss, n, C.byref(r))
    if rc != capi.MIS_OK:
        raise MisError(rc, "mis_result_roi: invalid arguments")
    return r.x, r.y, r.width, r.height


def blend_config(blend_type, blend_strength, pano_size):
    """image_stitching.cpp:1176-1190 -> (type, num_bands, sharpness)."""
    lib = capi.load()
    t, nb, sh = C.c_int(), C.c_int(), C.c_float()
    lib.mis_blend_config(int(blend_type), float(blend_strength), int(pano_size[0]), int(pano_size[1]), C.byref(t),
                         C.byref(nb), C.byref(sh))
    return t.value, nb.value, sh.value


class Blender:
    """Blender::createDefault(type) (image_stitching.cpp:1175)."""

    def __init__(self, ctx, btype=capi.BLEND_NO, num_bands=5, sharpness=0.02):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.mis_blender_create(ctx.h, btype, num_bands, sharpness, C.byref(h)))
        self.h = h
        self.type = btype
        self._size = None

    @staticmethod
    def createDefault(ctx, btype):
        return {capi.BLEND_NO: Blender, capi.BLEND_FEATHER: FeatherBlender, capi.BLEND_MULTI_BAND: MultiBandBlender}[btype](ctx)

    def prepare(self, corners, sizes):
        n = len(corners)
        cs = (capi.MisPoint * n)(*[capi.MisPoint(int(c[0]), int(c[1])) for c in corners])
        ss = (capi.MisSize * n)(*[capi.MisSize(int(s[0]), int(s[1])) for s in sizes])
        self.ctx.check(self.ctx.lib.mis_blender_prepare(self.h, cs, ss, n))
        x, y, w, h = result_roi(corners, sizes)
        self._size = (w, h)

    def compose_frames(self, frames, scale, cameras, rois):
        """The compositing loop's per-frame body for all frames in one library call (fused warp + feed each;
        image_stitching.cpp:1154-1164, :1218): the calling thread stays out of the interpreter between launches."""
        n = len(frames)
        if n == 0:
            return
        arr = (capi.MisImage * n)(*[as_image(f) for f in frames])
        Ks = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float32).reshape(9) for c in cameras]))
        Rs = np.ascontiguousarray(np.stack([np.asarray(c["R"], np.float32).reshape(9) for c in cameras]))
        rr = (capi.MisRect * n)(*[capi.MisRect(int(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in rois])
        self.ctx.check(self.ctx.lib.mis_compose_frames(self.h, arr, n, float(scale), Ks.ctypes.data_as(C.c_void_p), Rs.ctypes.data_as(C.c_void_p), rr))

    def feed(self, img, mask, tl):
        i, m = as_image(img), as_image(mask)
        self.ctx.check(self.ctx.lib.mis_blender_feed(self.h, C.byref(i), C.byref(m), capi.MisPoint(int(tl[0]), int(tl[1]))))

    def feed_batch(self, imgs, masks, tls):
        """n feeds in one call (mis_blender_feed_batch): the result of feed(imgs[0], ...) ... feed(imgs[n - 1], ...) in that order."""
        n = len(imgs)
        im = (capi.MisImage * n)(*[as_image(i) for i in imgs])
        mk = (capi.MisImage * n)(*[as_image(m) for m in masks])
        ts = (capi.MisPoint * n)(*[capi.MisPoint(int(t[0]), int(t[1])) for t in tls])
        self.ctx.check(self.ctx.lib.mis_blender_feed_batch(self.h, im, mk, ts, n))

    def blend(self):
        if self._size is None:
            raise MisError(-5, "blend before prepare")      # MIS_E_STATE, as the library answers
        w, h = self._size
        dst = _empty_image(self.ctx, h, w, 3, torch.int16)
        msk = _empty_image(self.ctx, h, w, 1, torch.uint8)
        d, m = as_image(dst), as_image(msk)
        self.ctx.check(self.ctx.lib.mis_blender_blend(self.h, C.byref(d), C.byref(m)))
        return dst, msk

    def level(self, i):
        """Accumulated pyramid level (host copies) before blend() -- parity tests only."""
        w, h, lp, wp = C.c_int(), C.c_int(), C.c_void_p(), C.c_void_p()
        self.ctx.check(self.ctx.lib.mis_blender_level_info(self.h, i, C.byref(w), C.byref(h), C.byref(lp), C.byref(wp)))
        self.ctx.synchronize()
        n = w.value * h.value
        lap = np.empty((h.value, w.value, 3), np.int16)
        wgt = np.empty((h.value, w.value), np.float32)
        # raw device pointers -> host through torch's runtime binding
        _memcpy_dtoh(lap, lp.value, n * 6)
        _memcpy_dtoh(wgt, wp.value, n * 4)
        return lap, wgt

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:   # a destroyed context already released the device
            self.ctx.lib.mis_blender_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiBandBlender(Blender):
    """MultiBandBlender + setNumBands (image_stitching.cpp:1180-1184)."""

    def __init__(self, ctx, num_bands=5):
        super().__init__(ctx, capi.BLEND_MULTI_BAND, num_bands, 0.0)

    def numBands(self):
        return self.ctx.lib.mis_blender_num_bands(self.h)


class FeatherBlender(Blender):
    """FeatherBlender + setSharpness (image_stitching.cpp:1186-1190)."""

    def __init__(self, ctx, sharpness=0.02):
        super().__init__(ctx, capi.BLEND_FEATHER, 0, sharpness)


_hip = None


def _memcpy_dtoh(dst_np, dev_ptr, nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = _hip.hipMemcpy(dst_np.ctypes.data_as(C.c_void_p), C.c_void_p(dev_ptr), nbytes, 2)
    if rc:
        raise RuntimeError("hipMemcpy D2H failed: %d" % rc)


# ------------------------------------------------------------------------------------------------
# features: ORB::create + computeImageFeatures (image_stitching.cpp:545, :613)
class ImageFeatures:
    """cv::detail::ImageFeatures: device-resident keypoints + descriptors."""

    def __init__(self, ctx, raw):
        self.ctx, self.raw = ctx, raw

    @property
    def img_idx(self):
        return self.raw.img_idx

    @img_idx.setter
    def img_idx(self, v):
        self.raw.img_idx = int(v)

    @property
    def img_size(self):
        return self.raw.img_w, self.raw.img_h

    def __len__(self):
        return self.raw.n

    def download(self):
        """-> (keypoints structured array, descriptors (n, cols))"""
        n = self.raw.n
        kps = np.zeros(n, KP_DTYPE)
        dt = np.uint8 if self.raw.desc_dtype == capi.U8 else np.float32
        desc = np.zeros((n, self.raw.desc_cols), dt)
        if n:
            self.ctx.check(self.ctx.lib.mis_features_download(self.ctx.h, C.byref(self.raw), kps.ctypes.data_as(C.c_void_p),
                                                              desc.ctypes.data_as(C.c_void_p)))
        return kps, desc

    @staticmethod
    def upload(ctx, img_size, kps, desc, img_idx=0):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc)
        raw = capi.MisFeatures()
        ctx.check(ctx.lib.mis_features_upload(ctx.h, int(img_size[0]), int(img_size[1]), len(kps),
                                              kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), desc.shape[1],
                                              _NP_DT[desc.dtype], C.byref(raw)))
        raw.img_idx = img_idx
        return ImageFeatures(ctx, raw)

    def close(self):
        if self.raw is not None and self.ctx.h:
            self.ctx.lib.mis_features_free(self.ctx.h, C.byref(self.raw))
        self.raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def orb_params(**kw):
    p = capi.MisOrbParams()
    capi.load().mis_orb_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class OrbFeatureFinder:
    """ORB::create(4000, 1.2, 8, 1, 0, 2, HARRIS_SCORE, 40, 20) (image_stitching.cpp:545)."""

    def __init__(self, ctx, max_size, params=None):
        self.ctx = ctx
        self.params = params or orb_params()
        h = C.c_void_p()
        ctx.check(ctx.lib.mis_orb_create(ctx.h, C.byref(self.params), int(max_size[0]), int(max_size[1]), C.byref(h)))
        self.h = h

    def detect(self, img):
        raw = capi.MisFeatures()
        i = as_image(img)
        self.ctx.check(self.ctx.lib.mis_orb_detect(self.h, C.byref(i), C.byref(raw)))
        return ImageFeatures(self.ctx, raw)

    def on_enqueued(self, fn):
        """fn() runs inside this finder's NEXT detect_batch call, on the calling thread, once the batch's device work is enqueued
        (mis_orb_on_enqueued).  fn=None clears a pending hook."""
        if fn is None:
            self._enqueued_fn = None
            self.ctx.lib.mis_orb_on_enqueued(self.h, None, None)
            return
        # ONE ctypes callback object per instance, created on first use: a CFUNCTYPE instance is a reference cycle that only the
        # cyclic garbage collector frees, and a fresh one per step kept that step's closure -- and through it the step's panorama --
        # alive while bench.py times with the collector paused (config 5: +1 GB of allocator growth per step, round 4)
        self._enqueued_fn = fn
        if getattr(self, "_enqueued_cb", None) is None:
            self._enqueued_cb = C.CFUNCTYPE(None, C.c_void_p)(self._run_enqueued)
        self.ctx.check(self.ctx.lib.mis_orb_on_enqueued(self.h, C.cast(self._enqueued_cb, C.c_void_p), None))

    def _run_enqueued(self, _user):
        fn, self._enqueued_fn = getattr(self, "_enqueued_fn", None), None
        if fn is not None:
            fn()

    def detect_batch(self, imgs):
        n = len(imgs)
        arr = (capi.MisImage * n)(*[as_image(i) for i in imgs])
        raws = (capi.MisFeatures * n)()
        self.ctx.check(self.ctx.lib.mis_orb_detect_batch(self.h, arr, n, raws))
        out = []
        for k in range(n):
            r = capi.MisFeatures()
            C.memmove(C.byref(r), C.byref(raws[k]), C.sizeof(capi.MisFeatures))
            r.img_idx = k
            out.append(ImageFeatures(self.ctx, r))
        return out

    def debug_level(self, level, which):
        w, h = C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.mis_orb_debug_level(self.h, level, which, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self.ctx.check(self.ctx.lib.mis_orb_debug_level(self.h, level, which, out.ctypes.data_as(C.c_void_p), C.byref(w), C.byref(h)))
        return out

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.mis_orb_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SiftFeatureFinder:
    """SIFT::create() (image_stitching.cpp:559, features_type == "sift"): same detect() interface as the ORB finder;
    descriptors are n x 128 f32 with integer values 0..255 (the L2 matcher's MFMA path, K8, consumes them)."""

    def __init__(self, ctx, max_size, params=None):
        self.ctx = ctx
        p = capi.MisSiftParams()
        ctx.lib.mis_sift_default_params(C.byref(p))
        for k, v in (params or {}).items():
            setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        ctx.check(ctx.lib.mis_sift_create(ctx.h, C.byref(p), int(max_size[0]), int(max_size[1]), C.byref(h)))
        self.h = h

    def detect(self, img):
        raw = capi.MisFeatures()
        i = as_image(img)
        self.ctx.check(self.ctx.lib.mis_sift_detect(self.h, C.byref(i), C.byref(raw)))
        return ImageFeatures(self.ctx, raw)

    def detect_batch(self, imgs):
        n = len(imgs)
        if n == 0:
            return []
        arr = (capi.MisImage * n)(*[as_image(im) for im in imgs])
        raws = (capi.MisFeatures * n)()
        self.ctx.check(self.ctx.lib.mis_sift_detect_batch(self.h, arr, n, raws))
        out = []
        for k in range(n):
            raw = capi.MisFeatures()
            C.memmove(C.byref(raw), C.byref(raws[k]), C.sizeof(capi.MisFeatures))
            out.append(ImageFeatures(self.ctx, raw))
        return out

    def debug_level(self, img, octave, layer, dog=False):
        i = as_image(img)
        w, h = C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.mis_sift_debug_level(self.h, C.byref(i), octave, layer, int(dog), None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.float32)
        self.ctx.check(self.ctx.lib.mis_sift_debug_level(self.h, C.byref(i), octave, layer, int(dog), out.ctypes.data_as(C.c_void_p), C.byref(w), C.byref(h)))
        return out

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.mis_sift_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def computeImageFeatures(finder, img, img_idx=0):
    """cv::detail::computeImageFeatures(finder, img, features) + features.img_idx = i (:613-614)."""
    f = finder.detect(img)
    f.img_idx = img_idx
    return f


# ------------------------------------------------------------------------------------------------
# matching: BestOf2NearestMatcher (image_stitching.cpp:647, :653), myLeaveBiggestComponent (:215-278)
@dataclass
class MatchesInfo:
    src_img_idx: int = -1
    dst_img_idx: int = -1
    matches: np.ndarray = field(default_factory=lambda: np.zeros(0, DMATCH_DTYPE))
    inliers_mask: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint8))
    num_inliers: int = 0
    H: np.ndarray = None
    confidence: float = 0.0


def _unpack_mi(mi):
    n = mi.n_matches
    out = MatchesInfo(mi.src_img_idx, mi.dst_img_idx)
    if n and mi.matches:
        out.matches = np.frombuffer((C.c_char * (n * 16)).from_address(C.addressof(mi.matches.contents)), DMATCH_DTYPE).copy()
    if n and mi.inliers_mask:
        out.inliers_mask = np.frombuffer((C.c_char * n).from_address(C.addressof(mi.inliers_mask.contents)), np.uint8).copy()
    out.num_inliers = mi.num_inliers
    out.H = np.array(list(mi.H), np.float64).reshape(3, 3) if mi.has_H else None
    out.confidence = mi.confidence
    return out


class PairwiseMatches:
    """The n x n cv::detail::MatchesInfo table of one matcher call, kept in the library's C structs and converted to
    MatchesInfo objects only when an entry is read (a job needs the confidences alone; converting 256 entries with their
    match arrays costs more host time than the pruning they feed).  Behaves like a list."""

    def __init__(self, ctx, mis, n):
        self._ctx, self._mis, self.n = ctx, mis, n
        self._cache = {}

    @classmethod
    def from_entries(cls, ctx, n, entries):
        """A table assembled on the host from (i, j, matches, inliers_mask, num_inliers, has_H, H, confidence) records of the
        pairs i < j (a sharded job gathers them from the ranks that matched them); the mirrored entries are filled as
        FeaturesMatcher::operator() fills them.  The arrays live in this object, not in the library."""
        mis = (capi.MisMatchesInfo * (n * n))()
        keep = []
        for k in range(n * n):
            mis[k].src_img_idx = mis[k].dst_img_idx = -1
        for i, j, matches, mask, num_inliers, has_H, H, conf in entries:
            mm = np.ascontiguousarray(matches, DMATCH_DTYPE)
            mk = np.ascontiguousarray(mask, np.uint8)
            sw = mm.copy()
            sw["query_idx"], sw["train_idx"] = mm["train_idx"], mm["query_idx"]
            Hd = np.asarray(H, np.float64).reshape(3, 3) if has_H else None
            for (a, b, arr, Hm) in ((i, j, mm, Hd), (j, i, sw, np.linalg.inv(Hd) if has_H else None)):
                e = mis[a * n + b]
                e.src_img_idx, e.dst_img_idx, e.n_matches = a, b, len(arr)
                e.matches = C.cast(arr.ctypes.data, C.POINTER(capi.MisDMatch)) if len(arr) else None
                e.inliers_mask = C.cast(mk.ctypes.data, C.POINTER(C.c_uint8)) if len(mk) else None
                e.num_inliers, e.has_H, e.confidence = int(num_inliers), 1 if has_H else 0, float(conf)
                if has_H:
                    for q, v in enumerate(Hm.reshape(9)):
                        e.H[q] = v
            keep += [mm, mk, sw]
        obj = cls(ctx, mis, n)
        obj._keep, obj._borrowed = keep, True
        return obj

    def __len__(self):
        return self.n * self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if i not in self._cache:
            self._cache[i] = _unpack_mi(self._mis[i])
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def confidences(self):
        """-> float64 array (n * n), row-major"""
        return np.array([self._mis[i].confidence for i in range(len(self))], np.float64)

    def __del__(self):
        try:
            if self._mis is not None and self._ctx.h and not getattr(self, "_borrowed", False):
                self._ctx.lib.mis_matches_free(self._mis, self.n * self.n)
            self._mis = None
        except Exception:
            pass


def match_params(**kw):
    p = capi.MisMatchParams()
    capi.load().mis_match_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class BestOf2NearestMatcher:
    """makePtr<BestOf2NearestMatcher>(try_cuda, match_conf) (image_stitching.cpp:647)."""

    def __init__(self, ctx, match_conf=0.32, num_matches_thresh1=6, num_matches_thresh2=6):
        self.ctx = ctx
        self.params = match_params(match_conf=match_conf, num_matches_thresh1=num_matches_thresh1,
                                   num_matches_thresh2=num_matches_thresh2)

    def __call__(self, features, rank=0, world_size=1):
        """(*matcher)(features, pairwise_matches) -> list of n*n MatchesInfo (row-major)."""
        n = len(features)
        arr = (capi.MisFeatures * n)()
        for k, f in enumerate(features):
            C.memmove(C.byref(arr[k]), C.byref(f.raw), C.sizeof(capi.MisFeatures))
        mis = (capi.MisMatchesInfo * (n * n))()
        if world_size == 1:
            rc = self.ctx.lib.mis_match_all_pairs(self.ctx.h, arr, n, C.byref(self.params), mis)
        else:
            rc = self.ctx.lib.mis_match_pairs_sharded(self.ctx.h, arr, n, C.byref(self.params), rank, world_size, mis)
        self.ctx.check(rc)
        return PairwiseMatches(self.ctx, mis, n)

    def collectGarbage(self):
        pass


def leaveBiggestComponent(pairwise_matches, n, conf_threshold):
    """myLeaveBiggestComponent (image_stitching.cpp:215-278) on the matcher output -> kept indices."""
    lib = capi.load()
    mis = (capi.MisMatchesInfo * (n * n))()
    for i, m in enumerate(pairwise_matches):
        mis[i].confidence = m.confidence
    idx = np.zeros(n, np.int32)
    k = C.c_int()
    rc = lib.mis_leave_biggest_component(mis, n, float(conf_threshold), idx.ctypes.data_as(C.c_void_p), C.byref(k))
    if rc != capi.MIS_OK:
        raise MisError(rc, "mis_leave_biggest_component")
    return idx[: k.value].copy()


def leaveBiggestComponentConf(confidence, conf_threshold):
    """myLeaveBiggestComponent on the bare n x n confidence matrix -> kept indices."""
    conf = np.ascontiguousarray(confidence, np.float64)
    n = conf.shape[0]
    idx = np.zeros(n, np.int32)
    k = C.c_int()
    rc = capi.load().mis_leave_biggest_component_conf(conf.ctypes.data_as(C.c_void_p), n, float(conf_threshold), idx.ctypes.data_as(C.c_void_p), C.byref(k))
    if rc != capi.MIS_OK:
        raise MisError(rc, "mis_leave_biggest_component_conf")
    return idx[: k.value].copy()


WAVE_CORRECT_HORIZ, WAVE_CORRECT_VERT = 0, 1


def refine_cameras(ctx, features, pairwise_matches, indices, cameras, cfg):
    """The reference's sequence after the pruning (image_stitching.cpp:671-726) on the kept subset: bundle adjustment
    of the subset's cameras (features / matches re-indexed as leaveBiggestComponent does), then wave correction.
    cameras: dicts with K (3x3), R; returns new dicts for the kept frames, in the order of `indices`."""
    n, k = pairwise_matches.n, len(indices)
    sub = (capi.MisMatchesInfo * (k * k))()
    for a, i in enumerate(indices):
        for b, j in enumerate(indices):
            C.memmove(C.byref(sub[a * k + b]), C.byref(pairwise_matches._mis[i * n + j]), C.sizeof(capi.MisMatchesInfo))
            sub[a * k + b].src_img_idx, sub[a * k + b].dst_img_idx = a, b
    view = PairwiseMatches.__new__(PairwiseMatches)
    view._ctx, view._mis, view.n, view._cache = ctx, sub, k, {}
    view._owner = pairwise_matches              # the match arrays belong to the full table
    start = [dict(focal=float(cameras[i]["K"][0, 0]), aspect=float(cameras[i]["K"][1, 1] / cameras[i]["K"][0, 0]), ppx=float(cameras[i]["K"][0, 2]),
                  ppy=float(cameras[i]["K"][1, 2]), R=cameras[i]["R"]) for i in indices]
    try:
        refined = bundle_adjust_reproj(ctx, [features[i] for i in indices], view, start, cfg.conf_thresh, cfg.ba_refine_mask)
    finally:
        view._mis = None                        # borrowed entries: nothing to free
    if cfg.wave_correct != "no":
        Rs = wave_correct([c["R"] for c in refined], WAVE_CORRECT_VERT if cfg.wave_correct == "vert" else WAVE_CORRECT_HORIZ)
        for c, R in zip(refined, Rs):
            c["R"] = R
    out = []
    for i, c in zip(indices, refined):
        d = dict(cameras[i])
        d["K"] = np.array([[c["focal"], 0, c["ppx"]], [0, c["focal"] * c["aspect"], c["ppy"]], [0, 0, 1]], np.float64)
        d["R"] = c["R"]
        d["f"] = c["focal"]
        out.append(d)
    return out


def bundle_adjust_reproj(ctx, features, pairwise_matches, cameras, conf_thresh=0.95, refine_mask="xxxxx"):
    """(*makePtr<detail::BundleAdjusterReproj>())(features, pairwise_matches, cameras) with setConfThresh /
    setRefinementMask (image_stitching.cpp:681-712).  cameras: list of dicts with focal, ppx, ppy, R (3x3) and
    optionally aspect, t -> new list of dicts (refined; rotations relative to the spanning tree's centre image)."""
    n = len(features)
    if not isinstance(pairwise_matches, PairwiseMatches):
        raise TypeError("pairwise_matches must be the PairwiseMatches object a BestOf2NearestMatcher call returned")
    arr = (capi.MisFeatures * n)()
    for k, f in enumerate(features):
        C.memmove(C.byref(arr[k]), C.byref(f.raw), C.sizeof(capi.MisFeatures))
    cams = (capi.MisCameraParams * n)()
    for k, c in enumerate(cameras):
        cams[k].focal, cams[k].aspect, cams[k].ppx, cams[k].ppy = float(c["focal"]), float(c.get("aspect", 1.0)), float(c["ppx"]), float(c["ppy"])
        R = np.asarray(c["R"], np.float64).reshape(9)
        t = np.asarray(c.get("t", np.zeros(3)), np.float64).reshape(3)
        for i in range(9):
            cams[k].R[i] = R[i]
        for i in range(3):
            cams[k].t[i] = t[i]
    ctx.check(ctx.lib.mis_bundle_adjust_reproj(ctx.h, arr, pairwise_matches._mis, n, float(conf_thresh), refine_mask.encode(), cams))
    return [dict(focal=cams[k].focal, aspect=cams[k].aspect, ppx=cams[k].ppx, ppy=cams[k].ppy, R=np.array(list(cams[k].R)).reshape(3, 3),
                 t=np.array(list(cams[k].t))) for k in range(n)]


def wave_correct(rmats, kind=WAVE_CORRECT_HORIZ):
    """detail::waveCorrect(rmats, kind) (image_stitching.cpp:718-726) -> list of corrected 3x3 rotations."""
    a = np.ascontiguousarray(np.stack([np.asarray(r, np.float64).reshape(3, 3) for r in rmats]))
    rc = capi.load().mis_wave_correct(a.ctypes.data_as(C.c_void_p), len(rmats), int(kind))
    if rc != capi.MIS_OK:
        raise MisError(rc, "mis_wave_correct")
    return [a[i].copy() for i in range(len(rmats))]


def find_homography(ctx, src, dst, thresh=3.0, max_iters=2000, confidence=0.995):
    """cv::findHomography(src, dst, mask, RANSAC) on the GPU -> (ok, H, mask)."""
    src = np.ascontiguousarray(src, np.float32)
    dst = np.ascontiguousarray(dst, np.float32)
    n = src.shape[0]
    H = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ok = C.c_int()
    ctx.check(ctx.lib.mis_find_homography(ctx.h, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), n, thresh,
                                          max_iters, confidence, H.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                                          C.byref(ok)))
    return bool(ok.value), H.reshape(3, 3), mask[:n]


# ------------------------------------------------------------------------------------------------
def check_seam_config(cfg):
    if cfg.seam_find_type not in ("no", "voronoi", "dp_color"):
        raise NotImplementedError("seam_find_type %r: 'no', 'voronoi' and 'dp_color' are implemented (dp_colorgrad and the "
                                  "graph-cut finders are outside this library; DESIGN.md section 8)" % (cfg.seam_find_type,))
    if cfg.expos_comp_type not in ("no", "gain_blocks"):
        raise NotImplementedError("expos_comp_type %r: only 'no' and 'gain_blocks' are implemented" % (cfg.expos_comp_type,))


def seam_scale_warp(ctx, cfg, frame_size, frame, camera, warped_image_scale, work_scale=1.0):
    """One frame of the seam-scale loop of main() (image_stitching.cpp:604-622 resize to seam_megapix, :973-990 warp of image and
    mask with K scaled by seam_work_aspect) -> (corner, image_warped 8UC3, mask_warped 8U) on the device."""
    w, h = frame_size
    seam_scale = min(1.0, float(np.sqrt(cfg.seam_megapix * 1e6 / (w * h))))
    swa = np.float32(seam_scale / work_scale)
    warper = SphericalWarper(ctx, np.float32(np.float32(warped_image_scale) * swa))
    img = resize(ctx, frame, fx=seam_scale, fy=seam_scale) if seam_scale < 1.0 else frame
    K = np.array(camera["K"], np.float32).copy()
    K[0, 0] *= swa; K[0, 2] *= swa; K[1, 1] *= swa; K[1, 2] *= swa
    R = np.asarray(camera["R"], np.float32)
    tl, iw = warper.warp(img, K, R, capi.INTER_LINEAR, capi.BORDER_REFLECT)
    full = torch.full((img.shape[0], img.shape[1]), 255, dtype=torch.uint8, device=ctx.device)
    _, mw = warper.warp(full, K, R, capi.INTER_NEAREST, capi.BORDER_CONSTANT)
    return tl, iw, mw


def seam_solve(ctx, cfg, corners, images_warped, masks_warped):
    """The part of the seam-scale pass that needs every image (image_stitching.cpp:1002-1023 exposure compensator feed, :1029-1065
    seam finder) -> (compensator | None, masks_warped edited in place)."""
    compensator = None
    if cfg.expos_comp_type == "gain_blocks":
        compensator = BlocksGainCompensator(ctx, cfg.expos_comp_block_size, cfg.expos_comp_block_size, cfg.expos_comp_nr_filtering)
        compensator.feed(corners, images_warped, masks_warped)
    if cfg.seam_find_type == "voronoi":
        VoronoiSeamFinder(ctx).find(images_warped, corners, masks_warped)
    elif cfg.seam_find_type == "dp_color":
        DpSeamFinder(ctx, DpSeamFinder.COLOR).find(images_warped, corners, masks_warped)
    return compensator, masks_warped


@dataclass
class ComposeGeometry:
    compose_scale: float      # min(1, sqrt(compose_megapix * 1e6 / area)); 1 when compose_megapix <= 0
    aspect: float             # compose_work_aspect = compose_scale / work_scale (work_scale = 1: features at full resolution)
    warp_scale: float         # warped_image_scale * (float)compose_work_aspect
    size: tuple               # frame size inside the compositing loop


def compose_geometry(cfg, frame_size, warped_image_scale):
    """The scales of the compositing loop (image_stitching.cpp:1105-1146): compose_scale from compose_megapix, the warper's scale
    and the intrinsics multiplied by compose_work_aspect, and -- only when |compose_scale - 1| > 0.1, as the reference tests it --
    frames and their sizes resized (cvRound)."""
    w, h = frame_size
    cs = 1.0
    if cfg.compose_megapix > 0:
        cs = min(1.0, float(np.sqrt(cfg.compose_megapix * 1e6 / (w * h))))
    aspect = cs / 1.0
    warp_scale = float(np.float32(warped_image_scale) * np.float32(aspect))
    size = (int(round(w * cs)), int(round(h * cs))) if abs(cs - 1) > 1e-1 else (int(w), int(h))      # cvRound: half to even, as Python's round
    return ComposeGeometry(cs, aspect, warp_scale, size)


def scaled_camera(cam, aspect):
    """cameras[i].focal *= a; ppx *= a; ppy *= a (image_stitching.cpp:1122-1125), in double like CameraParams; K() = [f 0 ppx; 0 f*aspect ppy]"""
    K = np.array(cam["K"], np.float64)
    f = K[0, 0] * aspect
    K2 = K.copy()
    K2[0, 0] = f
    K2[1, 1] = f * (K[1, 1] / K[0, 0])
    K2[0, 2] = K[0, 2] * aspect
    K2[1, 2] = K[1, 2] * aspect
    d = dict(cam)
    d["K"] = K2
    d["f"] = f
    return d


@dataclass
class StitchConfig:
    """The reference's globals-as-config (image_stitching.cpp:49-85), same defaults; compose_megapix
    <= 0 keeps frames at full resolution through warp + blend (the throughput configuration)."""
    work_megapix: float = -1
    seam_megapix: float = 0.1
    compose_megapix: float = 0.4
    conf_thresh: float = 0.95
    features_type: str = "orb"
    match_conf: float = 0.32
    warp_type: str = "spherical"
    blend_type: int = capi.BLEND_MULTI_BAND
    blend_strength: float = 5.0
    # camera refinement between matching and warping (image_stitching.cpp:681-726).  The reference's default is
    # "reproj"; here the default is "no" because the jobs of this package take exact (sensor / ground-truth) cameras.
    ba_cost_func: str = "no"          # "no" | "reproj"
    ba_refine_mask: str = "_____"     # the reference's default (image_stitching.cpp:67): rotations only
    wave_correct: str = "horiz"       # "horiz" | "vert" | "no"; applied after the bundle adjustment only
    # the seam-scale step between warp and blend (image_stitching.cpp:940-1070, :1162-1171), the reference's defaults
    # (:73-77): block gain compensation and the dynamic-programming colour seam finder
    expos_comp_type: str = "gain_blocks"   # "no" | "gain_blocks"
    expos_comp_block_size: int = 64
    expos_comp_nr_filtering: int = 2
    seam_find_type: str = "dp_color"       # "no" | "voronoi" | "dp_color"

    @classmethod
    def reference(cls, **kw):
        """What the reference's main() runs with its globals untouched (image_stitching.cpp:49-85): reprojection bundle adjustment +
        horizontal wave correction, block gain compensation, dp_color seams, seam_megapix 0.1, compose_megapix 0.4."""
        kw.setdefault("ba_cost_func", "reproj")
        return cls(**kw)

    @classmethod
    def hot_path(cls, **kw):
        """The configuration of the north-star hot path (features -> match -> warp -> blend): no exposure compensation and no
        seam finder (the steps between warp and blend); everything else as given."""
        kw.setdefault("expos_comp_type", "no")
        kw.setdefault("seam_find_type", "no")
        kw.setdefault("compose_megapix", -1)      # true-resolution warp + blend: the throughput configuration
        return cls(**kw)


class Stitcher:
    """The hot-path sequence of main() (image_stitching.cpp:567-1228) for frames already in HBM:
    features -> pairwise matches (+RANSAC) -> [cameras supplied by the caller] -> compose-scale
    warp -> blend.  Optional (off by default): reprojection bundle adjustment with wave
    correction (refine_cameras), block gain compensation and the "voronoi" seam finder (seam_step)."""

    def __init__(self, ctx, frame_size, config=None):
        self.ctx = ctx
        self.cfg = config or StitchConfig()
        self.frame_size = frame_size
        self.finder = OrbFeatureFinder(ctx, frame_size)
        self.matcher = BestOf2NearestMatcher(ctx, self.cfg.match_conf)

    def features(self, frames):
        return self.finder.detect_batch(frames)

    def match(self, feats, rank=0, world_size=1):
        return self.matcher(feats, rank, world_size)

    @staticmethod
    def warped_image_scale(cameras):
        """median focal (image_stitching.cpp:884-895)"""
        focals = sorted(float(c["K"][1][1]) for c in cameras)
        n = len(focals)
        if n % 2 == 1:
            return float(np.float32(focals[n // 2]))
        return float(np.float32(focals[n // 2 - 1] + focals[n // 2]) * np.float32(0.5))

    def compose(self, frames, cameras, indices=None, blender=None):
        """Compositing loop (image_stitching.cpp:1086-1225), frames and intrinsics at compose scale (compose_megapix)."""
        indices = list(range(len(frames)) if indices is None else indices)
        # the reference replaces `cameras` by the kept subset (image_stitching.cpp:746-748) before the median focal (:884-895)
        scale = self.warped_image_scale([cameras[i] for i in indices])
        g = compose_geometry(self.cfg, self.frame_size, scale)
        seam = self.seam_step(frames, cameras, indices, scale)      # at seam scale, with the un-scaled intrinsics (:973-1065)
        if g.aspect != 1.0:
            cameras = {i: scaled_camera(cameras[i], g.aspect) for i in indices}
        if g.size != tuple(self.frame_size):
            frames = {i: resize(self.ctx, frames[i], fx=g.compose_scale, fy=g.compose_scale) for i in indices}
        scale = g.warp_scale
        warper = SphericalWarper(self.ctx, scale)
        w, h = g.size
        rois = warp_rois(self.ctx, scale, (w, h), [cameras[i] for i in indices])
        corners = [(r[0], r[1]) for r in rois]
        sizes = [(r[2], r[3]) for r in rois]
        if blender is None:
            x, y, pw, ph = result_roi(corners, sizes)
            btype, bands, sharp = blend_config(self.cfg.blend_type, self.cfg.blend_strength, (pw, ph))
            if btype == capi.BLEND_MULTI_BAND:
                blender = MultiBandBlender(self.ctx, bands)
            elif btype == capi.BLEND_FEATHER:
                blender = FeatherBlender(self.ctx, sharp)
            else:
                blender = Blender(self.ctx)
        blender.prepare(corners, sizes)
        for k, i in enumerate(indices):
            tl, img_s, mask = warper.warp_fused(frames[i], cameras[i]["K"], cameras[i]["R"], rois[k])
            if seam is not None:
                compensator, seam_masks = seam
                if compensator is not None:
                    compensator.apply(k, corners[k], img_s, mask)       # :1162 (on the 16S copy: same values)
                seam_mask_apply(self.ctx, seam_masks[k], mask)           # :1169-1171
            blender.feed(img_s, mask, tl)
        return blender.blend()

    def seam_step(self, frames, cameras, indices, warped_image_scale, work_scale=1.0):
        """The seam-scale pass of main() (image_stitching.cpp:604-622 resize, :973-990 warp, :1002-1023 exposure
        compensator feed, :1029-1065 seam finder) -> (compensator | None, masks_warped) or None when both are off."""
        check_seam_config(self.cfg)
        if self.cfg.expos_comp_type == "no" and self.cfg.seam_find_type == "no":
            return None
        items = [seam_scale_warp(self.ctx, self.cfg, self.frame_size, frames[i], cameras[i], warped_image_scale, work_scale) for i in indices]
        return seam_solve(self.ctx, self.cfg, [it[0] for it in items], [it[1] for it in items], [it[2] for it in items])

    def stitch(self, frames, cameras):
        feats = self.features(frames)
        pm = self.match(feats)
        idx = leaveBiggestComponent(pm, len(frames), self.cfg.conf_thresh)
        result, mask = self.compose(frames, cameras, list(idx))
        return result, mask, feats, pm, idx
