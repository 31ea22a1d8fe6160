/* mo_imgops.h -- CPU restatement (TEST INFRASTRUCTURE, see mo_common.h) of the small image operators either
 * side of the hot path: cv::resize(INTER_LINEAR_EXACT) with scale factors, cv::rotate, and the seam-mask
 * step dilate(3x3) -> resize -> AND.  Reference call sites: image_stitching/image_stitching.cpp:571-580,
 * :602, :619, :1144 (resize / rotate), :1169-1171 (mask step).  PARITY UNPINNED (OpenCV restated from memory). */
#ifndef MO_IMGOPS_H
#define MO_IMGOPS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* cv::resize(src, dst, dsize, fx, fy, INTER_LINEAR_EXACT): dsize used when dw > 0 && dh > 0, otherwise
 * dsize = (cvRound(sw*fx), cvRound(sh*fy)) and the coordinate scale is 1/fx, 1/fy exactly. */
void mo_resize_dsize(int sw, int sh, int dw_in, int dh_in, double fx, double fy, int* dw, int* dh);
void mo_resize_linear_exact_u8_ex(const uint8_t* src, int sw, int sh, size_t sstride, int cn, uint8_t* dst, int dw, int dh,
                                  size_t dstride, double fx, double fy, int by_factor);
/* cv::rotate: code 0 = ROTATE_90_CLOCKWISE, 1 = ROTATE_180, 2 = ROTATE_90_COUNTERCLOCKWISE */
void mo_rotate_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn, int code, uint8_t* dst, size_t dstride);
/* cv::dilate(src, dst, Mat()): 3x3 rectangle, anchor centre, border = lowest value (never wins) */
void mo_dilate3x3_u8(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);
/* mask_warped = resize(dilate(seam_mask), mask.size(), INTER_LINEAR_EXACT) & mask_warped */
void mo_seam_mask_apply(const uint8_t* seam, int sw, int sh, size_t sstride, uint8_t* mask, int mw, int mh, size_t mstride);
#ifdef __cplusplus
}
#endif
#endif
