"""image_stitching_amd -- MI355X-native hot path of a1q123456/image_stitching.

Host-side mirror (Python) of the interface the reference drives in ``main()``
(image_stitching/image_stitching.cpp:545-1228): ``computeImageFeatures`` / ``BestOf2NearestMatcher`` /
``SphericalWarper`` / ``MultiBandBlender`` / ``FeatherBlender``, each a thin veneer over the C ABI of
``libmistitch.so`` (include/mistitch.h).  torch is used only to own device memory and streams.

There is no CPU fallback: everything here needs the HIP library and a GPU.
"""
from . import _capi
from ._capi import (BLEND_FEATHER, BLEND_MULTI_BAND, BLEND_NO, BORDER_CONSTANT, BORDER_REFLECT, INTER_LINEAR,
                    INTER_NEAREST)
from .stitching import (BestOf2NearestMatcher, Blender, BlocksGainCompensator, NoSeamFinder, VoronoiSeamFinder, DpSeamFinder, Context, FeatherBlender, ImageFeatures, MatchesInfo,
                        MisError, MultiBandBlender, OrbFeatureFinder, SiftFeatureFinder, SphericalWarper, StitchConfig, Stitcher,
                        blend_config, bundle_adjust_reproj, computeImageFeatures, find_homography, leaveBiggestComponent, resize, result_roi,
                        rotate, seam_mask_apply, warp_roi, wave_correct)

__all__ = [
    "Context", "MisError", "SphericalWarper", "Blender", "MultiBandBlender", "FeatherBlender", "OrbFeatureFinder", "SiftFeatureFinder",
    "computeImageFeatures", "ImageFeatures", "BestOf2NearestMatcher", "MatchesInfo", "leaveBiggestComponent",
    "find_homography", "warp_roi", "result_roi", "blend_config", "StitchConfig", "Stitcher", "resize", "rotate",
    "seam_mask_apply", "bundle_adjust_reproj", "wave_correct", "BlocksGainCompensator", "NoSeamFinder", "VoronoiSeamFinder", "DpSeamFinder",
    "INTER_NEAREST", "INTER_LINEAR", "BORDER_CONSTANT", "BORDER_REFLECT", "BLEND_NO", "BLEND_FEATHER",
    "BLEND_MULTI_BAND",
]
