/*
 * mo_orb.h -- ORACLE (test infrastructure): restatement of the ORB detect+describe path the
 * reference reaches through cv::detail::computeImageFeatures
 * (image_stitching/image_stitching.cpp:545 ORB::create(4000,1.2,8,1,0,2,HARRIS_SCORE,40,20),
 *  :613 computeImageFeatures).  Algorithm notes: SURVEY.md Appendix A.1-A.2.  PARITY UNPINNED.
 */
#ifndef MO_ORB_H
#define MO_ORB_H
#include "mo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

#define MO_ORB_MAX_LEVELS 16
#define MO_ORB_BORDER 32 /* >= max(edgeThreshold, ceil(halfPatch*sqrt2), 3)+1 = 30 for patch 40 */

typedef struct {
    int nfeatures;      /* 4000 */
    float scale_factor; /* 1.2f  */
    int nlevels;        /* 8 */
    int edge_threshold; /* 1 */
    int first_level;    /* 0 (only 0 supported) */
    int wta_k;          /* 2 (only 2 supported) */
    int score_type;     /* 0 = HARRIS_SCORE */
    int patch_size;     /* 40 */
    int fast_threshold; /* 20 */
} MoOrbParams;

typedef struct {
    float x, y, size, angle, response;
    int octave;
} MoKeyPoint;

typedef struct MoOrb MoOrb;

void mo_orb_default_params(MoOrbParams* p);
MoOrb* mo_orb_create(const MoOrbParams* p, int width, int height);
void mo_orb_destroy(MoOrb* o);
/* run the whole path on one BGR u8 image; returns number of keypoints or <0 on error */
int mo_orb_run(MoOrb* o, const uint8_t* bgr, size_t stride);
int mo_orb_num_keypoints(const MoOrb* o);
const MoKeyPoint* mo_orb_keypoints(const MoOrb* o);
const uint8_t* mo_orb_descriptors(const MoOrb* o); /* n x 32 */

/* intermediates (for stage-by-stage parity tests against the HIP kernels) */
int mo_orb_level_width(const MoOrb* o, int l);
int mo_orb_level_height(const MoOrb* o, int l);
float mo_orb_level_scale(const MoOrb* o, int l);
int mo_orb_level_nfeatures(const MoOrb* o, int l);
const uint8_t* mo_orb_level_gray(const MoOrb* o, int l);    /* tight w*h */
const uint8_t* mo_orb_level_nms(const MoOrb* o, int l);     /* tight w*h, NMS-surviving FAST score or 0 */
const uint8_t* mo_orb_level_blur(const MoOrb* o, int l);    /* padded (w+2B)*(h+2B) */
int mo_orb_level_count(const MoOrb* o, int l, int which);   /* 0: FAST corners after NMS, 1: kept after retainBest(2N), 2: final */
const int8_t* mo_orb_pattern(const MoOrb* o);               /* 512 x (x,y) */
const int* mo_orb_umax(const MoOrb* o);                     /* halfPatch+2 entries */

/* building blocks, also used on their own by tests */
void mo_bgr2gray(const uint8_t* bgr, int w, int h, size_t stride, uint8_t* gray, size_t gstride);
void mo_resize_linear_exact_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn,
                               uint8_t* dst, int dw, int dh, size_t dstride);
void mo_gauss7_kernel_q8(int k[7]);

#ifdef __cplusplus
}
#endif
#endif
