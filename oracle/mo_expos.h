/* mo_expos.h -- CPU restatement (TEST INFRASTRUCTURE, see mo_common.h) of the exposure compensation and of the
 * simple seam finders of the step between warp and blend:
 *   ExposureCompensator::createDefault(GAIN_BLOCKS), setNrFeeds(1), setNrGainsFilteringIterations(2), setBlockSize(64, 64),
 *   feed(corners, images_warped, masks_warped), apply(i, corner, img_warped, mask_warped)   image_stitching.cpp:1002-1023, :1162
 *   NoSeamFinder / VoronoiSeamFinder::find                                                 image_stitching.cpp:1029-1065
 * OpenCV source restated from memory: stitching/src/exposure_compensate.cpp (BlocksCompensator, GainCompensator::singleFeed),
 * seam_finders.cpp (PairwiseSeamFinder, VoronoiSeamFinder), core LU solve, imgproc sepFilter2D / resize / distanceTransform.
 * PARITY UNPINNED.  The reference's default seam finder, DpSeamFinder(COLOR), is not restated. */
#ifndef MO_EXPOS_H
#define MO_EXPOS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct MoCompensator MoCompensator;
MoCompensator* mo_compensator_create(int block_w, int block_h, int nr_filtering);
void mo_compensator_destroy(MoCompensator* c);
/* images: n pointers to tight 8UC3 images; masks: n pointers to tight 8UC1 masks (255 = valid); corners x,y; sizes w,h */
int mo_compensator_feed(MoCompensator* c, int n, const int* corners_xy, const int* sizes_wh, const uint8_t* const* images, const uint8_t* const* masks);
int mo_compensator_gain_map(const MoCompensator* c, int index, const float** map, int* bw, int* bh);
/* image (w x h x 3, u8, tight) *= gain map resized to w x h, in place */
int mo_compensator_apply(const MoCompensator* c, int index, uint8_t* image, int w, int h);
/* cv::solve(A, b, x) with DECOMP_LU for a square double system (n x n, row-major); returns 0 when singular */
int mo_solve_lu(double* A, double* b, int n);
/* VoronoiSeamFinder::find on n images: masks (tight 8UC1) are updated in place */
void mo_voronoi_seams(int n, const int* corners_xy, const int* sizes_wh, uint8_t* const* masks);
#ifdef __cplusplus
}
#endif
#endif
