"""ctypes declarations of include/mistitch.h (the C ABI of libmistitch.so).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C image_stitching_amd/csrc``).
There is no fallback of any kind: if the shared object is missing or was built without a symbol the
import fails loudly.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmistitch.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mistitch.h")

MIS_OK = 0
MIS_FENCE_TIMEOUT = 1
MEM_HOST, MEM_DEVICE = 0, 1
U8, S16, F32 = 0, 1, 2
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REFLECT = 0, 2
BLEND_NO, BLEND_FEATHER, BLEND_MULTI_BAND = 0, 1, 2


class MisPoint(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int)]


class MisSize(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int)]


class MisRect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class MisLevelRect(C.Structure):
    _fields_ = [("level", C.c_int), ("x0", C.c_int), ("y0", C.c_int), ("x1", C.c_int), ("y1", C.c_int), ("offset", C.c_ulonglong)]


class MisImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("channels", C.c_int),
                ("stride", C.c_size_t), ("dtype", C.c_int), ("mem", C.c_int)]


class MisKeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int)]


class MisOrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("edge_threshold", C.c_int), ("first_level", C.c_int), ("wta_k", C.c_int),
                ("score_type", C.c_int), ("patch_size", C.c_int), ("fast_threshold", C.c_int)]


class MisSiftParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("n_octave_layers", C.c_int), ("contrast_threshold", C.c_double),
                ("edge_threshold", C.c_double), ("sigma", C.c_double)]


class MisCameraParams(C.Structure):
    _fields_ = [("focal", C.c_double), ("aspect", C.c_double), ("ppx", C.c_double), ("ppy", C.c_double), ("R", C.c_double * 9),
                ("t", C.c_double * 3)]


class MisFeatures(C.Structure):
    _fields_ = [("img_idx", C.c_int), ("img_w", C.c_int), ("img_h", C.c_int), ("n", C.c_int),
                ("keypoints", C.c_void_p), ("descriptors", C.c_void_p), ("desc_cols", C.c_int),
                ("desc_dtype", C.c_int), ("owner_", C.c_void_p)]


class MisMatchParams(C.Structure):
    _fields_ = [("match_conf", C.c_float), ("num_matches_thresh1", C.c_int), ("num_matches_thresh2", C.c_int),
                ("ransac_thresh", C.c_double), ("max_iters", C.c_int), ("confidence", C.c_double)]


class MisDMatch(C.Structure):
    _fields_ = [("query_idx", C.c_int), ("train_idx", C.c_int), ("img_idx", C.c_int), ("distance", C.c_float)]


class MisMatchesInfo(C.Structure):
    _fields_ = [("src_img_idx", C.c_int), ("dst_img_idx", C.c_int), ("n_matches", C.c_int),
                ("matches", C.POINTER(MisDMatch)), ("inliers_mask", C.POINTER(C.c_uint8)), ("num_inliers", C.c_int),
                ("has_H", C.c_int), ("H", C.c_double * 9), ("confidence", C.c_double)]


_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_P = C.POINTER

# name -> (restype, argtypes); must list every function declared in include/mistitch.h
PROTOTYPES = {
    "mis_context_create": (_i, [_i, _vp, _P(_vp)]),
    "mis_stream_create": (_i, [_i, _i, _P(_vp)]),
    "mis_stream_destroy": (_i, [_vp]),
    "mis_context_destroy": (_i, [_vp]),
    "mis_context_synchronize": (_i, [_vp]),
    "mis_context_wait": (_i, [_vp, _vp]),
    "mis_last_error": (C.c_char_p, [_vp]),
    "mis_version": (C.c_char_p, []),
    "mis_image_free": (_i, [_vp, _P(MisImage)]),
    "mis_orb_default_params": (None, [_P(MisOrbParams)]),
    "mis_orb_create": (_i, [_vp, _P(MisOrbParams), _i, _i, _P(_vp)]),
    "mis_orb_destroy": (_i, [_vp]),
    "mis_orb_detect": (_i, [_vp, _P(MisImage), _P(MisFeatures)]),
    "mis_orb_detect_batch": (_i, [_vp, _P(MisImage), _i, _P(MisFeatures)]),
    "mis_orb_on_enqueued": (_i, [_vp, _vp, _vp]),
    "mis_features_download": (_i, [_vp, _P(MisFeatures), _vp, _vp]),
    "mis_features_upload": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _P(MisFeatures)]),
    "mis_features_free": (_i, [_vp, _P(MisFeatures)]),
    "mis_orb_debug_level": (_i, [_vp, _i, _i, _vp, _P(_i), _P(_i)]),
    "mis_sift_default_params": (None, [_P(MisSiftParams)]),
    "mis_sift_create": (_i, [_vp, _P(MisSiftParams), _i, _i, _P(_vp)]),
    "mis_sift_destroy": (_i, [_vp]),
    "mis_sift_detect": (_i, [_vp, _P(MisImage), _P(MisFeatures)]),
    "mis_sift_detect_batch": (_i, [_vp, _P(MisImage), _i, _P(MisFeatures)]),
    "mis_sift_debug_level": (_i, [_vp, _P(MisImage), _i, _i, _i, _vp, _P(_i), _P(_i)]),
    "mis_match_default_params": (None, [_P(MisMatchParams)]),
    "mis_match_all_pairs": (_i, [_vp, _P(MisFeatures), _i, _P(MisMatchParams), _P(MisMatchesInfo)]),
    "mis_match_pairs_sharded": (_i, [_vp, _P(MisFeatures), _i, _P(MisMatchParams), _i, _i, _P(MisMatchesInfo)]),
    "mis_matches_free": (_i, [_P(MisMatchesInfo), _i]),
    "mis_match_sequence": (C.c_longlong, [_vp]),
    "mis_match_knn_fence": (_i, [_vp, _vp, C.c_longlong, _i]),
    "mis_match_on_enqueued": (_i, [_vp, _vp, _vp]),
    "mis_knn2": (_i, [_vp, _P(MisFeatures), _P(MisFeatures), _vp, _vp]),
    "mis_find_homography": (_i, [_vp, _vp, _vp, _i, _d, _i, _d, _vp, _vp, _P(_i)]),
    "mis_leave_biggest_component": (_i, [_P(MisMatchesInfo), _i, _f, _vp, _P(_i)]),
    "mis_leave_biggest_component_conf": (_i, [_vp, _i, _f, _vp, _P(_i)]),
    "mis_bundle_adjust_reproj": (_i, [_vp, _P(MisFeatures), _P(MisMatchesInfo), _i, _f, C.c_char_p, _P(MisCameraParams)]),
    "mis_wave_correct": (_i, [_vp, _i, _i]),
    "mis_warp_roi": (_i, [_f, _i, _i, _vp, _vp, _P(MisRect)]),
    "mis_warp_spherical": (_i, [_vp, _P(MisImage), _f, _vp, _vp, _i, _i, _P(MisImage), _P(MisPoint)]),
    "mis_warp_roi_batch": (_i, [_vp, _f, _i, _i, _i, _vp, _vp, _P(MisRect)]),
    "mis_warp_spherical_fused_roi": (_i, [_vp, _P(MisImage), _f, _vp, _vp, _P(MisRect), _P(MisImage), _P(MisImage), _P(MisPoint)]),
    "mis_warp_spherical_fused_batch": (_i, [_vp, _P(MisImage), _i, C.c_float, _P(C.c_float), _P(C.c_float), _P(MisRect), _P(MisImage), _P(MisImage), _P(MisPoint)]),
    "mis_warp_spherical_fused_batch_timed": (_i, [_vp, _P(MisImage), _i, C.c_float, _P(C.c_float), _P(C.c_float), _P(MisRect), _P(MisImage), _P(MisImage), _P(MisPoint), _i,
                                               _P(C.c_float)]),
    "mis_warp_spherical_fused": (_i, [_vp, _P(MisImage), _f, _vp, _vp, _P(MisImage), _P(MisImage), _P(MisPoint)]),
    "mis_resize_linear_exact": (_i, [_vp, _P(MisImage), _i, _i, C.c_double, C.c_double, _P(MisImage)]),
    "mis_rotate": (_i, [_vp, _P(MisImage), _i, _P(MisImage)]),
    "mis_seam_mask_apply": (_i, [_vp, _P(MisImage), _P(MisImage)]),
    "mis_compensator_create": (_i, [_vp, _i, _i, _i, _P(_vp)]),
    "mis_compensator_destroy": (_i, [_vp]),
    "mis_compensator_feed": (_i, [_vp, _P(MisPoint), _P(MisImage), _P(MisImage), _i]),
    "mis_compensator_gain_map": (_i, [_vp, _i, _P(C.c_float), _i, _P(_i), _P(_i)]),
    "mis_compensator_apply": (_i, [_vp, _i, _P(MisImage)]),
    "mis_seam_voronoi": (_i, [_vp, _P(MisPoint), _P(MisImage), _i]),
    "mis_seam_dp": (_i, [_vp, _P(MisPoint), _P(MisImage), _P(MisImage), _i, _i]),
    "mis_warp_spherical_fused_timed": (_i, [_vp, _P(MisImage), _f, _vp, _vp, _P(MisImage), _P(MisImage), _P(MisPoint), _i, _P(C.c_float)]),
    "mis_blend_config": (_i, [_i, _f, _i, _i, _P(_i), _P(_i), _P(_f)]),
    "mis_result_roi": (_i, [_P(MisPoint), _P(MisSize), _i, _P(MisRect)]),
    "mis_blender_create": (_i, [_vp, _i, _i, _f, _P(_vp)]),
    "mis_blender_destroy": (_i, [_vp]),
    "mis_blender_prepare": (_i, [_vp, _P(MisPoint), _P(MisSize), _i]),
    "mis_blender_num_bands": (_i, [_vp]),
    "mis_blender_feed": (_i, [_vp, _P(MisImage), _P(MisImage), MisPoint]),
    "mis_blender_feed_batch": (_i, [_vp, _P(MisImage), _P(MisImage), _P(MisPoint), _i]),
    "mis_blender_blend": (_i, [_vp, _P(MisImage), _P(MisImage)]),
    "mis_compose_frames": (_i, [_vp, _P(MisImage), _i, _f, _vp, _vp, _P(MisRect)]),
    "mis_blender_blend_columns": (_i, [_vp, _i, _i, _P(MisImage), _P(MisImage)]),
    "mis_blender_pack_rects": (_i, [_vp, _P(MisLevelRect), _i, _vp, C.c_size_t]),
    "mis_blender_add_rects": (_i, [_vp, _P(MisLevelRect), _i, _vp, C.c_size_t]),
    "mis_blender_zero_rects": (_i, [_vp, _P(MisLevelRect), _i]),
    "mis_features_pack": (_i, [_vp, _P(MisFeatures), _i, _i, _i, _vp, _vp]),
    "mis_copy_2d": (_i, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_size_t]),
    "mis_blender_feed_rect": (_i, [_vp, _i, _i, MisPoint, _P(MisRect)]),
    "mis_blender_level_info": (_i, [_vp, _i, _P(_i), _P(_i), _P(_vp), _P(_vp)]),
}


def header_functions():
    """Names of every function declared in include/mistitch.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mis_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load():
    """Load libmistitch.so and bind every prototype; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmistitch.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C image_stitching_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
