/* mo_motion.h -- ORACLE (test infrastructure): bundle adjustment (reprojection cost) + wave correction; see mo_motion.c. */
#ifndef MO_MOTION_H
#define MO_MOTION_H
#include "mo_match.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { double focal, aspect, ppx, ppy, R[9], t[3]; } MoCamera;   /* cv::detail::CameraParams */
/* (*adjuster)(features, pairwise_matches, cameras) with BundleAdjusterReproj (image_stitching.cpp:681-712): features[i].xy are the
 * keypoints, pairwise the n x n MatchesInfo table; cameras are refined in place.  refine_mask: the 5-letter ba_refine_mask.
 * Returns 0, or < 0 when there is nothing to adjust / the solver diverged. */
int mo_bundle_adjust_reproj(int n, const MoFeatures* features, const MoMatchesInfo* pairwise, float conf_thresh, const char* refine_mask,
                            MoCamera* cameras, int* iterations);
/* waveCorrect(rmats, kind) (image_stitching.cpp:718-726): n x 9 rotations in place; kind 0 = HORIZ, 1 = VERT */
int mo_wave_correct(double* rmats, int n, int kind);
#ifdef __cplusplus
}
#endif
#endif
