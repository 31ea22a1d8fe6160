/*
 * mo_match.h -- ORACLE (test infrastructure): restatement of the pairwise matching path the
 * reference reaches through cv::detail::BestOf2NearestMatcher
 * (image_stitching/image_stitching.cpp:647 makePtr<BestOf2NearestMatcher>(try_cuda, match_conf),
 *  :653 (*matcher)(features, pairwise_matches)) and of myLeaveBiggestComponent (:215-278).
 * Algorithm notes: SURVEY.md Appendix A.3-A.5.  The 2-NN search is the EXACT brute-force form
 * the north star asks for (OpenCV's CPU path is approximate FLANN).  PARITY UNPINNED.
 */
#ifndef MO_MATCH_H
#define MO_MATCH_H
#include "mo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int query_idx, train_idx, img_idx; float distance; } MoDMatch;

typedef struct {
    int img_w, img_h;
    int n;
    const float* xy;       /* n x 2 keypoint coordinates (level-0 pixels) */
    const uint8_t* desc_u8; /* n x 32 (ORB) or NULL */
    const float* desc_f32;  /* n x dim (SIFT) or NULL */
    int dim;
} MoFeatures;

typedef struct {
    float match_conf;        /* 0.32 for ORB (image_stitching.cpp:62) */
    int num_matches_thresh1; /* 6 */
    int num_matches_thresh2; /* 6 */
    double ransac_thresh;    /* 3.0 */
    int max_iters;           /* 2000 */
    double confidence;       /* 0.995 */
} MoMatchParams;

typedef struct {
    int src_img_idx, dst_img_idx;
    int n_matches;
    MoDMatch* matches;      /* malloc'ed */
    uint8_t* inliers_mask;  /* malloc'ed, n_matches entries (or NULL when RANSAC did not run) */
    int num_inliers;
    int has_H;
    double H[9];
    double confidence;
    int ransac_iters[2];    /* diagnostics: hypotheses evaluated by the two findHomography calls */
} MoMatchesInfo;

void mo_match_default_params(MoMatchParams* p);
void mo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx2, int* dist2);
void mo_knn2_l2(const float* q, int nq, const float* t, int nt, int dim, int* idx2, float* dist2);
int mo_match_pair(const MoFeatures* f1, const MoFeatures* f2, const MoMatchParams* p, MoMatchesInfo* out);
/* out: n*n row-major, mirrored as FeaturesMatcher::operator() does */
int mo_match_all_pairs(const MoFeatures* feats, int n, const MoMatchParams* p, MoMatchesInfo* out);
void mo_matches_free(MoMatchesInfo* m, int count);

/* cv::findHomography(src, dst, mask, RANSAC, thresh, maxIters, confidence); returns 1 when H valid */
int mo_find_homography_ransac(const float* src_xy, const float* dst_xy, int n, double thresh, int max_iters,
                              double confidence, double H[9], uint8_t* mask, int* iters_run);
/* building blocks exposed for stage tests */
int mo_homography_dlt(const float* src_xy, const float* dst_xy, int n, double H[9]);
void mo_jacobi_eigen(double* A, int n, double* W, double* V);
int mo_homography_refine_lm(const float* src_xy, const float* dst_xy, int n, double H[9], int max_iters);
int mo_ransac_update_num_iters(double p, double ep, int model_points, int max_iters);

/* myLeaveBiggestComponent (image_stitching.cpp:215-278): returns number of kept indices */
int mo_leave_biggest_component(const double* confidence /* n*n */, int n, float conf_threshold, int* indices);

#ifdef __cplusplus
}
#endif
#endif
