"""ctypes bindings of the CPU oracle (test infrastructure only; see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmisoracle.so")


def build(force=False):
    """Compile oracle/libmisoracle.so with gcc (no-op when up to date)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "libmisoracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


c_u8p = C.POINTER(C.c_uint8)
c_i8p = C.POINTER(C.c_int8)
c_i16p = C.POINTER(C.c_int16)
c_ip = C.POINTER(C.c_int)
c_fp = C.POINTER(C.c_float)
c_dp = C.POINTER(C.c_double)


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("edge_threshold", C.c_int), ("first_level", C.c_int), ("wta_k", C.c_int),
                ("score_type", C.c_int), ("patch_size", C.c_int), ("fast_threshold", C.c_int)]


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4")])


class DMatch(C.Structure):
    _fields_ = [("query_idx", C.c_int), ("train_idx", C.c_int), ("img_idx", C.c_int), ("distance", C.c_float)]


DMATCH_DTYPE = np.dtype([("query_idx", "i4"), ("train_idx", "i4"), ("img_idx", "i4"), ("distance", "f4")])


class Features(C.Structure):
    _fields_ = [("img_w", C.c_int), ("img_h", C.c_int), ("n", C.c_int), ("xy", c_fp), ("desc_u8", c_u8p),
                ("desc_f32", c_fp), ("dim", C.c_int)]


class MatchParams(C.Structure):
    _fields_ = [("match_conf", C.c_float), ("num_matches_thresh1", C.c_int), ("num_matches_thresh2", C.c_int),
                ("ransac_thresh", C.c_double), ("max_iters", C.c_int), ("confidence", C.c_double)]


class MatchesInfo(C.Structure):
    _fields_ = [("src_img_idx", C.c_int), ("dst_img_idx", C.c_int), ("n_matches", C.c_int),
                ("matches", C.POINTER(DMatch)), ("inliers_mask", c_u8p), ("num_inliers", C.c_int),
                ("has_H", C.c_int), ("H", C.c_double * 9), ("confidence", C.c_double), ("ransac_iters", C.c_int * 2)]


class Projector(C.Structure):
    _fields_ = [("scale", C.c_float), ("k", C.c_float * 9), ("rinv", C.c_float * 9), ("r_kinv", C.c_float * 9),
                ("k_rinv", C.c_float * 9)]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


def _declare(L):
    L.mo_fast_atan2.restype = C.c_float
    L.mo_fast_atan2.argtypes = [C.c_float, C.c_float]
    for n in ("mo_sinf", "mo_cosf", "mo_acosf"):
        getattr(L, n).restype = C.c_float
        getattr(L, n).argtypes = [C.c_float]
    L.mo_atan2f.restype = C.c_float
    L.mo_atan2f.argtypes = [C.c_float, C.c_float]
    L.mo_log_d.restype = C.c_double
    L.mo_log_d.argtypes = [C.c_double]
    L.mo_orb_create.restype = C.c_void_p
    L.mo_orb_create.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_int]
    L.mo_orb_destroy.argtypes = [C.c_void_p]
    L.mo_orb_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mo_orb_num_keypoints.argtypes = [C.c_void_p]
    L.mo_orb_keypoints.restype = C.c_void_p
    L.mo_orb_keypoints.argtypes = [C.c_void_p]
    L.mo_orb_descriptors.restype = C.c_void_p
    L.mo_orb_descriptors.argtypes = [C.c_void_p]
    for n in ("mo_orb_level_width", "mo_orb_level_height", "mo_orb_level_nfeatures"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int]
    L.mo_orb_level_scale.restype = C.c_float
    L.mo_orb_level_scale.argtypes = [C.c_void_p, C.c_int]
    for n in ("mo_orb_level_gray", "mo_orb_level_nms", "mo_orb_level_blur"):
        getattr(L, n).restype = C.c_void_p
        getattr(L, n).argtypes = [C.c_void_p, C.c_int]
    L.mo_orb_level_count.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mo_orb_pattern.restype = C.c_void_p
    L.mo_orb_pattern.argtypes = [C.c_void_p]
    L.mo_orb_umax.restype = C.c_void_p
    L.mo_orb_umax.argtypes = [C.c_void_p]
    L.mo_bgr2gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
    L.mo_resize_linear_exact_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_int,
                                            C.c_int, C.c_size_t]
    L.mo_gauss7_kernel_q8.argtypes = [c_ip]
    L.mo_knn2_hamming.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.mo_knn2_l2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.mo_match_pair.argtypes = [C.POINTER(Features), C.POINTER(Features), C.POINTER(MatchParams), C.POINTER(MatchesInfo)]
    L.mo_match_all_pairs.argtypes = [C.POINTER(Features), C.c_int, C.POINTER(MatchParams), C.POINTER(MatchesInfo)]
    L.mo_matches_free.argtypes = [C.POINTER(MatchesInfo), C.c_int]
    L.mo_find_homography_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double,
                                            C.c_void_p, C.c_void_p, c_ip]
    L.mo_homography_dlt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mo_jacobi_eigen.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.mo_homography_refine_lm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.mo_ransac_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    L.mo_leave_biggest_component.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.mo_projector_set.argtypes = [C.POINTER(Projector), C.c_float, C.c_void_p, C.c_void_p]
    L.mo_map_forward.argtypes = [C.POINTER(Projector), C.c_float, C.c_float, c_fp, c_fp]
    L.mo_map_backward.argtypes = [C.POINTER(Projector), C.c_float, C.c_float, c_fp, c_fp]
    L.mo_warp_roi.argtypes = [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Rect)]
    L.mo_warp_spherical.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, c_ip, c_ip]
    L.mo_build_maps.argtypes = [C.POINTER(Projector), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.mo_blend_config.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, c_ip, c_fp]
    L.mo_blender_create.restype = C.c_void_p
    L.mo_blender_create.argtypes = [C.c_int, C.c_int, C.c_float]
    L.mo_blender_destroy.argtypes = [C.c_void_p]
    L.mo_blender_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.mo_blender_num_bands.argtypes = [C.c_void_p]
    L.mo_blender_roi.argtypes = [C.c_void_p, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip]
    L.mo_blender_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                  C.c_int, C.c_int]
    L.mo_blender_blend.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.mo_blender_level_lap.restype = C.c_void_p
    L.mo_blender_level_lap.argtypes = [C.c_void_p, C.c_int, c_ip, c_ip]
    L.mo_blender_level_weight.restype = C.c_void_p
    L.mo_blender_level_weight.argtypes = [C.c_void_p, C.c_int, c_ip, c_ip]
    L.mo_pyr_down_s16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mo_pyr_down_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mo_pyr_up_s16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mo_distance_l1.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    for n, T in (("mo_rot_to_euler_d", "d"), ("mo_rot_to_euler_f", "f"), ("mo_euler_to_rot_d", "d"), ("mo_euler_to_rot_f", "f")):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.mo_quat_from_rot_d.argtypes = [C.c_void_p, C.c_void_p]
    L.mo_quat_to_rot_d.argtypes = [C.c_void_p, C.c_void_p]
    L.mo_camera_rehand_d.argtypes = [C.c_void_p, C.c_int, C.c_void_p]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _from_ptr(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


# ----------------------------------------------------------------------------------------------
def orb_default_params(**kw):
    p = OrbParams(4000, 1.2, 8, 1, 0, 2, 0, 40, 20)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Orb:
    """Oracle ORB; keeps every intermediate of the last run for stage-level parity checks."""

    def __init__(self, width, height, params=None):
        self.params = params or orb_default_params()
        self.w, self.h = width, height
        self.h_ = lib().mo_orb_create(C.byref(self.params), width, height)
        if not self.h_:
            raise ValueError("mo_orb_create rejected the parameters")

    def close(self):
        if self.h_:
            lib().mo_orb_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        self.close()

    def run(self, bgr):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        assert bgr.shape == (self.h, self.w, 3)
        n = lib().mo_orb_run(self.h_, _p(bgr), bgr.strides[0])
        if n < 0:
            raise RuntimeError("mo_orb_run failed: %d" % n)
        kps = _from_ptr(lib().mo_orb_keypoints(self.h_), (n,), KP_DTYPE)
        desc = _from_ptr(lib().mo_orb_descriptors(self.h_), (n, 32), np.uint8)
        return kps, desc

    def level_size(self, l):
        return lib().mo_orb_level_width(self.h_, l), lib().mo_orb_level_height(self.h_, l)

    def level_scale(self, l):
        return lib().mo_orb_level_scale(self.h_, l)

    def level_nfeatures(self, l):
        return lib().mo_orb_level_nfeatures(self.h_, l)

    def level_gray(self, l):
        w, h = self.level_size(l)
        return _from_ptr(lib().mo_orb_level_gray(self.h_, l), (h, w), np.uint8)

    def level_nms(self, l):
        w, h = self.level_size(l)
        return _from_ptr(lib().mo_orb_level_nms(self.h_, l), (h, w), np.uint8)

    def level_blur(self, l, border=32):
        w, h = self.level_size(l)
        return _from_ptr(lib().mo_orb_level_blur(self.h_, l), (h + 2 * border, w + 2 * border), np.uint8)

    def level_count(self, l, which):
        return lib().mo_orb_level_count(self.h_, l, which)

    def pattern(self):
        return _from_ptr(lib().mo_orb_pattern(self.h_), (512, 2), np.int8)

    def umax(self):
        return _from_ptr(lib().mo_orb_umax(self.h_), (self.params.patch_size // 2 + 2,), np.int32)


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    g = np.empty((h, w), np.uint8)
    lib().mo_bgr2gray(_p(bgr), w, h, bgr.strides[0], _p(g), g.strides[0])
    return g


def resize_linear_exact(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    dst = np.empty((dh, dw) if cn == 1 else (dh, dw, cn), np.uint8)
    lib().mo_resize_linear_exact_u8(_p(src), sw, sh, src.strides[0], cn, _p(dst), dw, dh, dst.strides[0])
    return dst


def gauss7_kernel_q8():
    k = (C.c_int * 7)()
    lib().mo_gauss7_kernel_q8(k)
    return list(k)


def knn2_hamming(q, t):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    idx = np.empty((q.shape[0], 2), np.int32)
    dist = np.empty((q.shape[0], 2), np.int32)
    lib().mo_knn2_hamming(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(dist))
    return idx, dist


def knn2_l2(q, t):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    idx = np.empty((q.shape[0], 2), np.int32)
    dist = np.empty((q.shape[0], 2), np.float32)
    lib().mo_knn2_l2(_p(q), q.shape[0], _p(t), t.shape[0], q.shape[1], _p(idx), _p(dist))
    return idx, dist


def match_default_params(**kw):
    p = MatchParams(0.32, 6, 6, 3.0, 2000, 0.995)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _mk_features(f):
    """f: dict(img_w, img_h, xy (n,2) f32, desc (n,32) u8 | (n,dim) f32) -> (Features, keepalive)"""
    xy = np.ascontiguousarray(f["xy"], np.float32)
    d = f["desc"]
    F = Features()
    F.img_w, F.img_h, F.n = int(f["img_w"]), int(f["img_h"]), int(xy.shape[0])
    F.xy = xy.ctypes.data_as(c_fp)
    if d.dtype == np.uint8:
        d = np.ascontiguousarray(d, np.uint8)
        F.desc_u8 = d.ctypes.data_as(c_u8p)
        F.dim = 32
    else:
        d = np.ascontiguousarray(d, np.float32)
        F.desc_f32 = d.ctypes.data_as(c_fp)
        F.dim = d.shape[1]
    return F, (xy, d)


def _unpack_matches(mi):
    n = mi.n_matches
    out = {
        "src_img_idx": mi.src_img_idx, "dst_img_idx": mi.dst_img_idx,
        "matches": (_from_ptr(C.addressof(mi.matches.contents), (n,), DMATCH_DTYPE) if n and mi.matches
                    else np.zeros((0,), DMATCH_DTYPE)),
        "inliers_mask": (_from_ptr(C.addressof(mi.inliers_mask.contents), (n,), np.uint8)
                         if n and mi.inliers_mask else np.zeros((0,), np.uint8)),
        "num_inliers": mi.num_inliers, "has_H": bool(mi.has_H),
        "H": np.array(list(mi.H), np.float64).reshape(3, 3), "confidence": mi.confidence,
        "ransac_iters": (mi.ransac_iters[0], mi.ransac_iters[1]),
    }
    return out


def match_pair(f1, f2, params=None):
    params = params or match_default_params()
    F1, k1 = _mk_features(f1)
    F2, k2 = _mk_features(f2)
    mi = MatchesInfo()
    lib().mo_match_pair(C.byref(F1), C.byref(F2), C.byref(params), C.byref(mi))
    out = _unpack_matches(mi)
    lib().mo_matches_free(C.byref(mi), 1)
    return out


def match_all_pairs(feats, params=None):
    params = params or match_default_params()
    n = len(feats)
    arr = (Features * n)()
    keep = []
    for i, f in enumerate(feats):
        F, k = _mk_features(f)
        arr[i] = F
        keep.append(k)
    mis = (MatchesInfo * (n * n))()
    lib().mo_match_all_pairs(arr, n, C.byref(params), mis)
    out = [_unpack_matches(mis[i]) for i in range(n * n)]
    lib().mo_matches_free(mis, n * n)
    return out


def find_homography_ransac(src, dst, thresh=3.0, max_iters=2000, confidence=0.995):
    src = np.ascontiguousarray(src, np.float32)
    dst = np.ascontiguousarray(dst, np.float32)
    n = src.shape[0]
    H = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    it = C.c_int(0)
    ok = lib().mo_find_homography_ransac(_p(src), _p(dst), n, thresh, max_iters, confidence, _p(H), _p(mask), C.byref(it))
    return bool(ok), H.reshape(3, 3), mask[:n], it.value


def homography_dlt(src, dst):
    src = np.ascontiguousarray(src, np.float32)
    dst = np.ascontiguousarray(dst, np.float32)
    H = np.zeros(9, np.float64)
    ok = lib().mo_homography_dlt(_p(src), _p(dst), src.shape[0], _p(H))
    return bool(ok), H.reshape(3, 3)


def jacobi_eigen(A):
    A = np.array(A, np.float64, copy=True, order="C")
    n = A.shape[0]
    W = np.zeros(n)
    V = np.zeros((n, n))
    lib().mo_jacobi_eigen(_p(A), n, _p(W), _p(V))
    return W, V


def homography_refine_lm(src, dst, H, max_iters=10):
    src = np.ascontiguousarray(src, np.float32)
    dst = np.ascontiguousarray(dst, np.float32)
    H = np.array(H, np.float64).reshape(9).copy()
    it = lib().mo_homography_refine_lm(_p(src), _p(dst), src.shape[0], _p(H), max_iters)
    return H.reshape(3, 3), it


def ransac_update_num_iters(p, ep, max_iters):
    return lib().mo_ransac_update_num_iters(p, ep, 4, max_iters)


def leave_biggest_component(conf, thresh):
    conf = np.ascontiguousarray(conf, np.float64)
    n = conf.shape[0]
    idx = np.zeros(n, np.int32)
    k = lib().mo_leave_biggest_component(_p(conf), n, thresh, _p(idx))
    return idx[:k].copy()


# ----------------------------------------------------------------------------------------------
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REFLECT = 0, 2


def projector(scale, K, R):
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    P = Projector()
    lib().mo_projector_set(C.byref(P), scale, _p(K), _p(R))
    return P


def map_forward(P, x, y):
    u, v = C.c_float(), C.c_float()
    lib().mo_map_forward(C.byref(P), x, y, C.byref(u), C.byref(v))
    return u.value, v.value


def map_backward(P, u, v):
    x, y = C.c_float(), C.c_float()
    lib().mo_map_backward(C.byref(P), u, v, C.byref(x), C.byref(y))
    return x.value, y.value


def warp_roi(scale, w, h, K, R):
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    r = Rect()
    lib().mo_warp_roi(scale, w, h, _p(K), _p(R), C.byref(r))
    return r.x, r.y, r.width, r.height


def warp_spherical(src, scale, K, R, interp=INTER_LINEAR, border=BORDER_REFLECT):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    x, y, rw, rh = warp_roi(scale, w, h, K, R)
    dst = np.zeros((rh, rw) if cn == 1 else (rh, rw, cn), np.uint8)
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    tx, ty = C.c_int(), C.c_int()
    rc = lib().mo_warp_spherical(_p(src), w, h, src.strides[0], cn, scale, _p(K), _p(R), interp, border, _p(dst),
                                 dst.strides[0], rw, rh, C.byref(tx), C.byref(ty))
    if rc:
        raise RuntimeError("mo_warp_spherical failed: %d" % rc)
    return dst, (tx.value, ty.value)


def build_maps(P, tlx, tly, brx, bry):
    xm = np.zeros((bry - tly + 1, brx - tlx + 1), np.float32)
    ym = np.zeros_like(xm)
    lib().mo_build_maps(C.byref(P), tlx, tly, brx, bry, _p(xm), _p(ym))
    return xm, ym


# ----------------------------------------------------------------------------------------------
BLEND_NO, BLEND_FEATHER, BLEND_MULTI_BAND = 0, 1, 2


def blend_config(blend_type, blend_strength, pano_w, pano_h):
    nb, sh = C.c_int(), C.c_float()
    t = lib().mo_blend_config(blend_type, blend_strength, pano_w, pano_h, C.byref(nb), C.byref(sh))
    return t, nb.value, sh.value


class Blender:
    def __init__(self, btype=BLEND_MULTI_BAND, num_bands=5, sharpness=0.02):
        self.h_ = lib().mo_blender_create(btype, num_bands, sharpness)
        self.type = btype

    def close(self):
        if self.h_:
            lib().mo_blender_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        self.close()

    def prepare(self, corners, sizes):
        c = np.ascontiguousarray(corners, np.int32).reshape(-1, 2)
        s = np.ascontiguousarray(sizes, np.int32).reshape(-1, 2)
        rc = lib().mo_blender_prepare(self.h_, _p(c), _p(s), c.shape[0])
        if rc:
            raise RuntimeError("mo_blender_prepare failed: %d" % rc)

    @property
    def num_bands(self):
        return lib().mo_blender_num_bands(self.h_)

    def roi(self):
        v = [C.c_int() for _ in range(6)]
        lib().mo_blender_roi(self.h_, *[C.byref(i) for i in v])
        return tuple(i.value for i in v)

    def feed(self, img_s16, mask, tl):
        img = np.ascontiguousarray(img_s16, np.int16)
        mask = np.ascontiguousarray(mask, np.uint8)
        h, w = mask.shape
        assert img.shape == (h, w, 3)
        rc = lib().mo_blender_feed(self.h_, _p(img), img.strides[0] // 2, _p(mask), mask.strides[0], w, h, int(tl[0]), int(tl[1]))
        if rc:
            raise RuntimeError("mo_blender_feed failed: %d" % rc)

    def level(self, i):
        w, h = C.c_int(), C.c_int()
        lp = lib().mo_blender_level_lap(self.h_, i, C.byref(w), C.byref(h))
        lap = _from_ptr(lp, (h.value, w.value, 3), np.int16)
        wp = lib().mo_blender_level_weight(self.h_, i, C.byref(w), C.byref(h))
        wgt = _from_ptr(wp, (h.value, w.value), np.float32) if wp else None
        return lap, wgt

    def blend(self):
        _, _, _, _, fw, fh = self.roi()
        dst = np.zeros((fh, fw, 3), np.int16)
        m = np.zeros((fh, fw), np.uint8)
        rc = lib().mo_blender_blend(self.h_, _p(dst), dst.strides[0] // 2, _p(m), m.strides[0])
        if rc:
            raise RuntimeError("mo_blender_blend failed: %d" % rc)
        return dst, m


def pyr_down_s16(src):
    src = np.ascontiguousarray(src, np.int16)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.zeros(((h + 1) // 2, (w + 1) // 2) + (() if src.ndim == 2 else (cn,)), np.int16)
    lib().mo_pyr_down_s16(_p(src), w, h, cn, _p(dst))
    return dst


def pyr_down_f32(src):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    dst = np.zeros(((h + 1) // 2, (w + 1) // 2), np.float32)
    lib().mo_pyr_down_f32(_p(src), w, h, _p(dst))
    return dst


def pyr_up_s16(src):
    src = np.ascontiguousarray(src, np.int16)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.zeros((2 * h, 2 * w) + (() if src.ndim == 2 else (cn,)), np.int16)
    lib().mo_pyr_up_s16(_p(src), w, h, cn, _p(dst))
    return dst


def distance_l1(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    d = np.zeros((h, w), np.float32)
    lib().mo_distance_l1(_p(mask), mask.strides[0], w, h, _p(d))
    return d


# ----------------------------------------------------------------------------------------------
EULER_ORDERS = {"XYZ": 0, "YXZ": 1, "ZXY": 2, "ZYX": 3, "YZX": 4, "XZY": 5}


def rot_to_euler(R, order, dtype=np.float64):
    R = np.ascontiguousarray(R, dtype).reshape(9)
    e = np.zeros(3, dtype)
    fn = lib().mo_rot_to_euler_d if dtype == np.float64 else lib().mo_rot_to_euler_f
    fn(_p(R), EULER_ORDERS[order], _p(e))
    return e


def euler_to_rot(e, order, dtype=np.float64):
    e = np.ascontiguousarray(e, dtype).reshape(3)
    R = np.zeros(9, dtype)
    fn = lib().mo_euler_to_rot_d if dtype == np.float64 else lib().mo_euler_to_rot_f
    fn(_p(e), EULER_ORDERS[order], _p(R))
    return R.reshape(3, 3)


def quat_from_rot(R):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    q = np.zeros(4)
    lib().mo_quat_from_rot_d(_p(R), _p(q))
    return q


def quat_to_rot(q):
    q = np.ascontiguousarray(q, np.float64).reshape(4)
    R = np.zeros(9)
    lib().mo_quat_to_rot_d(_p(q), _p(R))
    return R.reshape(3, 3)


def camera_rehand(R, is_portrait=False):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    o = np.zeros(9)
    lib().mo_camera_rehand_d(_p(R), int(is_portrait), _p(o))
    return o.reshape(3, 3)


# ---- image operators (mo_imgops.c) -------------------------------------------------------------
def resize_exact(src, dsize=None, fx=0.0, fy=0.0):
    """cv::resize(src, dst, dsize, fx, fy, INTER_LINEAR_EXACT), u8 with 1 or 3 channels."""
    L = lib()
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = C.c_int(), C.c_int()
    L.mo_resize_dsize(sw, sh, int(dsize[0]) if dsize else 0, int(dsize[1]) if dsize else 0, C.c_double(fx), C.c_double(fy), C.byref(dw), C.byref(dh))
    dst = np.zeros((dh.value, dw.value) if cn == 1 else (dh.value, dw.value, cn), np.uint8)
    L.mo_resize_linear_exact_u8_ex(src.ctypes.data_as(C.c_void_p), sw, sh, C.c_size_t(sw * cn), cn, dst.ctypes.data_as(C.c_void_p), dw.value, dh.value,
                                   C.c_size_t(dw.value * cn), C.c_double(fx), C.c_double(fy), 0 if dsize else 1)
    return dst


def rotate(src, code):
    L = lib()
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = (sw, sh) if code == 1 else (sh, sw)
    dst = np.zeros((dh, dw) if cn == 1 else (dh, dw, cn), np.uint8)
    L.mo_rotate_u8(src.ctypes.data_as(C.c_void_p), sw, sh, C.c_size_t(sw * cn), cn, int(code), dst.ctypes.data_as(C.c_void_p), C.c_size_t(dw * cn))
    return dst


def dilate3x3(src):
    L = lib()
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros_like(src)
    L.mo_dilate3x3_u8(src.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), dst.ctypes.data_as(C.c_void_p), C.c_size_t(w))
    return dst


def seam_mask_apply(seam, mask):
    """returns resize(dilate(seam), mask.shape, INTER_LINEAR_EXACT) & mask"""
    L = lib()
    seam = np.ascontiguousarray(seam, np.uint8)
    out = np.ascontiguousarray(mask, np.uint8).copy()
    sh, sw = seam.shape
    mh, mw = out.shape
    L.mo_seam_mask_apply(seam.ctypes.data_as(C.c_void_p), sw, sh, C.c_size_t(sw), out.ctypes.data_as(C.c_void_p), mw, mh, C.c_size_t(mw))
    return out


# ---- SIFT (mo_sift.c) ----------------------------------------------------------------------------
class SiftParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("n_octave_layers", C.c_int), ("contrast_threshold", C.c_double), ("edge_threshold", C.c_double),
                ("sigma", C.c_double)]


class Sift:
    """cv::SIFT::create() + detectAndCompute on one BGR image (defaults of the reference, image_stitching.cpp:559)."""

    def __init__(self, width, height, params=None):
        L = lib()
        L.mo_sift_create.restype = C.c_void_p
        L.mo_sift_keypoints.restype = C.c_void_p
        L.mo_sift_descriptors.restype = C.c_void_p
        L.mo_sift_gauss.restype = C.c_void_p
        L.mo_sift_dog.restype = C.c_void_p
        p = SiftParams()
        L.mo_sift_default_params(C.byref(p))
        if params:
            for k, v in params.items():
                setattr(p, k, v)
        self.params = p
        self.w, self.h = width, height
        self.h_ = C.c_void_p(L.mo_sift_create(C.byref(p), width, height))

    def __del__(self):
        try:
            lib().mo_sift_destroy(self.h_)
        except Exception:
            pass

    def run(self, bgr):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        assert bgr.shape == (self.h, self.w, 3)
        n = lib().mo_sift_run(self.h_, bgr.ctypes.data_as(C.c_void_p), C.c_size_t(self.w * 3))
        if n < 0:
            raise RuntimeError("mo_sift_run failed")
        kps = _from_ptr(lib().mo_sift_keypoints(self.h_), (n,), KP_DTYPE) if n else np.zeros(0, KP_DTYPE)
        desc = _from_ptr(lib().mo_sift_descriptors(self.h_), (n, 128), np.float32) if n else np.zeros((0, 128), np.float32)
        return kps, desc

    def num_octaves(self):
        return lib().mo_sift_num_octaves(self.h_)

    def num_raw_keypoints(self):
        return lib().mo_sift_num_raw_keypoints(self.h_)

    def gauss(self, o, i):
        w, h = C.c_int(), C.c_int()
        p = lib().mo_sift_gauss(self.h_, o, i, C.byref(w), C.byref(h))
        return _from_ptr(p, (h.value, w.value), np.float32)

    def dog(self, o, i):
        w, h = C.c_int(), C.c_int()
        p = lib().mo_sift_dog(self.h_, o, i, C.byref(w), C.byref(h))
        return _from_ptr(p, (h.value, w.value), np.float32)


def expf(x):
    L = lib()
    L.mo_expf.restype = C.c_float
    return L.mo_expf(C.c_float(x))


def gaussian_taps_f32(sigma):
    buf = (C.c_float * 64)()
    n = lib().mo_gaussian_taps_f32(C.c_double(sigma), buf)
    return np.array(buf[:n], np.float32)


# ----------------------------------------------------------------------------------------------
# exposure compensation (GAIN_BLOCKS) and the Voronoi seam finder (mo_expos.c)
class Compensator:
    """BlocksGainCompensator(block 64x64, 1 feed, 2 gain-filtering passes by default)."""

    def __init__(self, block_w=64, block_h=64, nr_filtering=2):
        L = lib()
        L.mo_compensator_create.restype = C.c_void_p
        self.h_ = C.c_void_p(L.mo_compensator_create(block_w, block_h, nr_filtering))

    def __del__(self):
        if getattr(self, "h_", None):
            lib().mo_compensator_destroy(self.h_)
            self.h_ = None

    def feed(self, corners, images, masks):
        n = len(images)
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        msks = [np.ascontiguousarray(m, np.uint8) for m in masks]
        c = np.ascontiguousarray(corners, np.int32).reshape(n, 2)
        s = np.array([[m.shape[1], m.shape[0]] for m in msks], np.int32)
        ip = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        mp = (C.c_void_p * n)(*[m.ctypes.data for m in msks])
        rc = lib().mo_compensator_feed(self.h_, n, _p(c), _p(s), ip, mp)
        if rc:
            raise RuntimeError("mo_compensator_feed failed: %d" % rc)

    def gain_map(self, index):
        p, w, h = C.c_void_p(), C.c_int(), C.c_int()
        rc = lib().mo_compensator_gain_map(self.h_, index, C.byref(p), C.byref(w), C.byref(h))
        if rc:
            raise IndexError(index)
        return _from_ptr(p.value, (h.value, w.value), np.float32)

    def apply(self, index, image):
        img = np.ascontiguousarray(image, np.uint8).copy()
        h, w = img.shape[:2]
        rc = lib().mo_compensator_apply(self.h_, index, _p(img), w, h)
        if rc:
            raise IndexError(index)
        return img


def solve_lu(A, b):
    A = np.array(A, np.float64, order="C")
    b = np.array(b, np.float64).reshape(-1).copy()
    ok = lib().mo_solve_lu(_p(A), _p(b), A.shape[0])
    return b if ok else None


def voronoi_seams(corners, masks):
    """VoronoiSeamFinder::find; returns the updated masks."""
    n = len(masks)
    msks = [np.ascontiguousarray(m, np.uint8).copy() for m in masks]
    c = np.ascontiguousarray(corners, np.int32).reshape(n, 2)
    s = np.array([[m.shape[1], m.shape[0]] for m in msks], np.int32)
    mp = (C.c_void_p * n)(*[m.ctypes.data for m in msks])
    lib().mo_voronoi_seams(n, _p(c), _p(s), mp)
    return msks


def dp_seams(images, corners, masks):
    """DpSeamFinder(COLOR)::find (image_stitching.cpp:1056-1065) on 8UC3 seam-scale images; returns the updated masks."""
    n = len(masks)
    msks = [np.ascontiguousarray(m, np.uint8).copy() for m in masks]
    imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
    c = np.ascontiguousarray(corners, np.int32).reshape(n, 2)
    s = np.array([[m.shape[1], m.shape[0]] for m in msks], np.int32)
    ip = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
    mp = (C.c_void_p * n)(*[m.ctypes.data for m in msks])
    L = lib()
    L.mo_seam_dp_color.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mo_seam_dp_color(n, _p(c), _p(s), ip, mp)
    return msks


class Camera(C.Structure):
    _fields_ = [("focal", C.c_double), ("aspect", C.c_double), ("ppx", C.c_double), ("ppy", C.c_double), ("R", C.c_double * 9), ("t", C.c_double * 3)]


def bundle_adjust_reproj(feats, matches, cameras, conf_thresh=0.95, refine_mask="xxxxx"):
    """BundleAdjusterReproj (image_stitching.cpp:681-712).  feats: dicts as for match_all_pairs (xy is what matters); matches: the
    n*n list of dicts match_all_pairs returns (or objects with the same fields as attributes); cameras: dicts focal, ppx, ppy,
    aspect, R -> (new list of dicts, LM iterations)."""
    n = len(feats)
    fa = (Features * n)()
    keep = []
    for i, f in enumerate(feats):
        F, k = _mk_features(f)
        fa[i] = F
        keep.append(k)
    mis = (MatchesInfo * (n * n))()

    def g(m, name):
        return m[name] if isinstance(m, dict) else getattr(m, name)
    for k, m in enumerate(matches):
        mm = np.ascontiguousarray(g(m, "matches"), DMATCH_DTYPE)
        mk = np.ascontiguousarray(g(m, "inliers_mask"), np.uint8)
        keep += [mm, mk]
        mis[k].src_img_idx, mis[k].dst_img_idx, mis[k].n_matches = int(g(m, "src_img_idx")), int(g(m, "dst_img_idx")), len(mm)
        mis[k].matches = C.cast(mm.ctypes.data, C.POINTER(DMatch)) if len(mm) else None
        mis[k].inliers_mask = C.cast(mk.ctypes.data, C.POINTER(C.c_uint8)) if len(mk) else None
        mis[k].num_inliers = int(g(m, "num_inliers"))
        H = g(m, "H")
        has = (g(m, "has_H") if isinstance(m, dict) else H is not None)
        mis[k].has_H = 1 if has else 0
        if has:
            for q, v in enumerate(np.asarray(H, np.float64).reshape(9)):
                mis[k].H[q] = v
        mis[k].confidence = float(g(m, "confidence"))
    cams = (Camera * n)()
    for k, c in enumerate(cameras):
        cams[k].focal, cams[k].aspect, cams[k].ppx, cams[k].ppy = float(c["focal"]), float(c.get("aspect", 1.0)), float(c["ppx"]), float(c["ppy"])
        for q, v in enumerate(np.asarray(c["R"], np.float64).reshape(9)):
            cams[k].R[q] = v
    it = C.c_int(0)
    L = lib()
    L.mo_bundle_adjust_reproj.argtypes = [C.c_int, C.POINTER(Features), C.POINTER(MatchesInfo), C.c_float, C.c_char_p, C.POINTER(Camera), C.POINTER(C.c_int)]
    rc = L.mo_bundle_adjust_reproj(n, fa, mis, float(conf_thresh), refine_mask.encode(), cams, C.byref(it))
    if rc:
        raise RuntimeError("mo_bundle_adjust_reproj: %d" % rc)
    return [dict(focal=cams[k].focal, aspect=cams[k].aspect, ppx=cams[k].ppx, ppy=cams[k].ppy, R=np.array(list(cams[k].R)).reshape(3, 3)) for k in range(n)], it.value


def wave_correct(rmats, kind=0):
    """detail::waveCorrect (image_stitching.cpp:718-726): kind 0 = HORIZ, 1 = VERT -> list of 3x3 rotations."""
    a = np.ascontiguousarray(np.stack([np.asarray(r, np.float64).reshape(3, 3) for r in rmats]))
    L = lib()
    L.mo_wave_correct.argtypes = [C.c_void_p, C.c_int, C.c_int]
    if L.mo_wave_correct(a.ctypes.data, len(rmats), int(kind)):
        raise RuntimeError("mo_wave_correct failed")
    return [a[i].copy() for i in range(len(rmats))]
