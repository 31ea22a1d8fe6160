/*
 * mo_blend.h -- ORACLE (test infrastructure): restatement of cv::detail::MultiBandBlender and
 * FeatherBlender as the reference drives them (image_stitching/image_stitching.cpp:1175-1192
 * createDefault / setNumBands / setSharpness / prepare, :1218 feed, :1225 blend).
 * Algorithm notes: SURVEY.md Appendix A.7.  PARITY UNPINNED.
 */
#ifndef MO_BLEND_H
#define MO_BLEND_H
#include "mo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

#define MO_BLEND_NO 0
#define MO_BLEND_FEATHER 1
#define MO_BLEND_MULTI_BAND 2

typedef struct MoBlender MoBlender;

/* reference-side sizing (image_stitching.cpp:1176-1190): returns blend type actually used, fills
 * num_bands / sharpness from the panorama area and blend_strength */
int mo_blend_config(int blend_type, float blend_strength, int pano_w, int pano_h, int* num_bands, float* sharpness);
/* resultRoi(corners, sizes) */
void mo_result_roi(const int* corners_xy, const int* sizes_wh, int n, int* x, int* y, int* w, int* h);

MoBlender* mo_blender_create(int type, int num_bands, float sharpness);
void mo_blender_destroy(MoBlender* b);
int mo_blender_prepare(MoBlender* b, const int* corners_xy, const int* sizes_wh, int n);
int mo_blender_num_bands(const MoBlender* b);
/* dst roi after prepare (padded for multi-band) and the final (un-padded) size */
void mo_blender_roi(const MoBlender* b, int* x, int* y, int* w, int* h, int* final_w, int* final_h);
int mo_blender_feed(MoBlender* b, const int16_t* img, size_t img_stride_elems, const uint8_t* mask, size_t mask_stride,
                    int w, int h, int tlx, int tly);
/* dst: final_h x final_w x 3 s16, dst_mask: final_h x final_w u8 */
int mo_blender_blend(MoBlender* b, int16_t* dst, size_t dst_stride_elems, uint8_t* dst_mask, size_t mask_stride);
/* intermediates for stage tests: accumulated pyramids before blend() */
const int16_t* mo_blender_level_lap(const MoBlender* b, int level, int* w, int* h);
const float* mo_blender_level_weight(const MoBlender* b, int level, int* w, int* h);

/* building blocks */
void mo_pyr_down_s16(const int16_t* src, int w, int h, int cn, int16_t* dst);       /* dst ((w+1)/2)x((h+1)/2) */
void mo_pyr_down_f32(const float* src, int w, int h, float* dst);
void mo_pyr_up_s16(const int16_t* src, int w, int h, int cn, int16_t* dst);         /* dst (2w)x(2h) */
void mo_distance_l1(const uint8_t* mask, size_t stride, int w, int h, float* dist); /* distanceTransform(DIST_L1,3) */

#ifdef __cplusplus
}
#endif
#endif
