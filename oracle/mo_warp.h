/*
 * mo_warp.h -- ORACLE (test infrastructure): restatement of cv::detail::SphericalWarper as the
 * reference drives it (image_stitching/image_stitching.cpp:973, :1117 create(scale); :985, :988,
 * :1154, :1159 warp(img, K, R, interp, border, dst); :1138 warpRoi(sz, K, R); :1164 convertTo
 * CV_16S).  Algorithm notes: SURVEY.md Appendix A.6.  PARITY UNPINNED.
 */
#ifndef MO_WARP_H
#define MO_WARP_H
#include "mo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

#define MO_INTER_NEAREST 0
#define MO_INTER_LINEAR 1
#define MO_BORDER_CONSTANT 0
#define MO_BORDER_REFLECT 2

typedef struct {
    float scale;
    float k[9], rinv[9], r_kinv[9], k_rinv[9];
} MoProjector;

typedef struct { int x, y, width, height; } MoRect;

void mo_projector_set(MoProjector* p, float scale, const float K[9], const float R[9]);
void mo_map_forward(const MoProjector* p, float x, float y, float* u, float* v);
void mo_map_backward(const MoProjector* p, float u, float v, float* x, float* y);
/* detectResultRoi: tl and br (inclusive) of the destination on the sphere */
void mo_detect_result_roi(const MoProjector* p, int src_w, int src_h, int* tlx, int* tly, int* brx, int* bry);
/* RotationWarper::warpRoi */
void mo_warp_roi(float scale, int src_w, int src_h, const float K[9], const float R[9], MoRect* roi);
/* RotationWarper::warp: dst must hold (bry-tly+1) rows of (brx-tlx+1) pixels; returns tl via *tlx,*tly.
 * src/dst are u8 with `cn` interleaved channels. */
int mo_warp_spherical(const uint8_t* src, int w, int h, size_t stride, int cn, float scale, const float K[9],
                      const float R[9], int interp, int border, uint8_t* dst, size_t dstride, int dst_w, int dst_h,
                      int* tlx, int* tly);
/* the f32 maps of buildMaps (for map-level parity tests) */
void mo_build_maps(const MoProjector* p, int tlx, int tly, int brx, int bry, float* xmap, float* ymap);

#ifdef __cplusplus
}
#endif
#endif
