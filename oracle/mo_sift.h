/* mo_sift.h -- CPU restatement (TEST INFRASTRUCTURE, see mo_common.h) of cv::SIFT::create()->detectAndCompute
 * as the reference calls it: image_stitching/image_stitching.cpp:559 (`SIFT::create()`, features_type == "sift")
 * and :613 (`computeImageFeatures`).  OpenCV source restated from memory: features2d/src/sift.dispatch.cpp,
 * sift.simd.hpp (4.5+, float scale space: SIFT_FIXPT_SCALE = 1), imgproc GaussianBlur / resize.
 *
 * PARITY UNPINNED: OpenCV is absent.  Deliberate restatement choices where OpenCV's own result depends
 * on its SIMD build: separable Gaussian taps accumulated in ascending tap order without FMA; exp via the shared
 * Cephes polynomial mo_expf (cv::hal::exp32f is table based); 2^x via mo_expf(x ln 2).
 *
 * Defaults of SIFT::create(): nfeatures 0 (unbounded), nOctaveLayers 3, contrastThreshold 0.04, edgeThreshold 10,
 * sigma 1.6, firstOctave -1 (input doubled), descriptors CV_32F with integer values 0..255. */
#ifndef MO_SIFT_H
#define MO_SIFT_H
#include <stddef.h>
#include <stdint.h>
#include "mo_orb.h" /* MoKeyPoint */
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int nfeatures;             /* 0 */
    int n_octave_layers;       /* 3 */
    double contrast_threshold; /* 0.04 */
    double edge_threshold;     /* 10 */
    double sigma;              /* 1.6 */
} MoSiftParams;

typedef struct MoSift MoSift;

void mo_sift_default_params(MoSiftParams* p);
MoSift* mo_sift_create(const MoSiftParams* p, int width, int height);
void mo_sift_destroy(MoSift* s);
int mo_sift_run(MoSift* s, const uint8_t* bgr, size_t stride); /* -> number of keypoints, < 0 on error */
int mo_sift_num_keypoints(const MoSift* s);
const MoKeyPoint* mo_sift_keypoints(const MoSift* s);
const float* mo_sift_descriptors(const MoSift* s); /* n x 128 */
/* intermediates for stage-by-stage parity tests */
int mo_sift_num_octaves(const MoSift* s);
const float* mo_sift_gauss(const MoSift* s, int octave, int layer, int* w, int* h); /* layer 0 .. nOctaveLayers + 2 */
const float* mo_sift_dog(const MoSift* s, int octave, int layer, int* w, int* h);   /* layer 0 .. nOctaveLayers + 1 */
int mo_sift_num_raw_keypoints(const MoSift* s); /* before duplicate removal */

float mo_expf(float x);
/* Gaussian taps of GaussianBlur(sigma) for CV_32F images: ksize = cvRound(sigma * 8 + 1) | 1; returns ksize */
int mo_gaussian_taps_f32(double sigma, float* taps /* >= 64 */);
#ifdef __cplusplus
}
#endif
#endif
