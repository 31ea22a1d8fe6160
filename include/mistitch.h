/*
 * mistitch.h -- C ABI of libmistitch.so: the MI355X-native (gfx950 / HIP) stitching hot path.
 *
 * The reference (a1q123456/image_stitching) has no FFI of its own: image_stitching.h is an empty
 * stub (image_stitching/image_stitching.h:1-8) and the hot path is the set of cv::detail calls
 * main() makes.  Every entry point below replaces one of those call sites (cited per function);
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++ / torch types.
 *  - every function returns int: MIS_OK (0) or a negative MIS_E_* code; mis_last_error() gives text.
 *    Nothing throws across the ABI (the reference's OpenCV calls throw cv::Exception instead).
 *  - one MisContext = one HIP device + one stream; calls on a context are serialised by the caller.
 *  - image buffers are described by MisImage; `mem` says whether `data` is a host or a device
 *    pointer.  Device pointers are used in place (zero copy); host pointers are staged through HBM.
 *  - outputs whose `data` is NULL on entry are allocated by the library (in HBM) and released with
 *    mis_image_free(); otherwise they must have exactly the required size.
 *  - there is NO CPU fallback: without a HIP device mis_context_create() fails.
 */
#ifndef MISTITCH_H
#define MISTITCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIS_OK 0
#define MIS_E_INVALID (-1)     /* bad argument */
#define MIS_E_HIP (-2)         /* HIP runtime error */
#define MIS_E_NOMEM (-3)
#define MIS_E_OVERFLOW (-4)    /* an internal candidate buffer was too small for the input */
#define MIS_E_STATE (-5)       /* call order violated (e.g. feed before prepare) */
#define MIS_E_UNSUPPORTED (-6)
#define MIS_FENCE_TIMEOUT 1     /* mis_match_knn_fence only, not an error: the matcher call did not show up within the timeout, nothing was queued */

enum { MIS_MEM_HOST = 0, MIS_MEM_DEVICE = 1 };
enum { MIS_U8 = 0, MIS_S16 = 1, MIS_F32 = 2 };
enum { MIS_INTER_NEAREST = 0, MIS_INTER_LINEAR = 1 };          /* cv::INTER_NEAREST / INTER_LINEAR */
enum { MIS_BORDER_CONSTANT = 0, MIS_BORDER_REFLECT = 2 };       /* cv::BORDER_CONSTANT / BORDER_REFLECT */
enum { MIS_BLEND_NO = 0, MIS_BLEND_FEATHER = 1, MIS_BLEND_MULTI_BAND = 2 }; /* cv::detail::Blender::NO/FEATHER/MULTI_BAND */

typedef struct MisContext MisContext;
typedef struct MisOrb MisOrb;
typedef struct MisBlender MisBlender;

typedef struct { int x, y; } MisPoint;
typedef struct { int width, height; } MisSize;
typedef struct { int x, y, width, height; } MisRect;

typedef struct {
    void* data;
    int width, height, channels;
    size_t stride; /* bytes between rows */
    int dtype;     /* MIS_U8 / MIS_S16 / MIS_F32 */
    int mem;       /* MIS_MEM_HOST / MIS_MEM_DEVICE */
} MisImage;

/* ---------------------------------------------------------------- context ------------------- */
/* `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL is the device's
 * default (null) stream.  All work of the context is enqueued on that stream. */
int mis_context_create(int device, void* stream, MisContext** out);
/* A non-blocking HIP stream for a second context on the same device (priority > 0: background work that yields
 * to the default-priority streams, < 0: urgent, 0: normal).  Pass the handle as `stream` to mis_context_create. */
int mis_stream_create(int device, int priority, void** stream);
int mis_stream_destroy(void* stream);
int mis_context_destroy(MisContext* ctx);
int mis_context_synchronize(MisContext* ctx);
/* Device-side ordering between two contexts of one device: work enqueued on `ctx` after this call starts only after everything
 * enqueued on `other` before it has finished (an event on other's stream; the host does not wait).  What the job does at its
 * entry between the stream that produced the frames and its compose stream. */
int mis_context_wait(MisContext* ctx, MisContext* other);
const char* mis_last_error(const MisContext* ctx);
const char* mis_version(void);
int mis_image_free(MisContext* ctx, MisImage* img);

/* ---------------------------------------------------------------- features ------------------ */
/* cv::KeyPoint as ImageFeatures::keypoints carries it (class_id is never read by the reference). */
typedef struct {
    float x, y, size, angle, response;
    int octave;
} MisKeyPoint;

/* ORB::create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType,
 * patchSize, fastThreshold) -- replaces image_stitching/image_stitching.cpp:545.  Supported:
 * first_level 0, wta_k 2, score_type 0 (HARRIS_SCORE), patch_size <= 40, nlevels <= 16. */
typedef struct {
    int nfeatures;
    float scale_factor;
    int nlevels, edge_threshold, first_level, wta_k, score_type, patch_size, fast_threshold;
} MisOrbParams;

/* cv::detail::ImageFeatures (image_stitching.cpp:537): keypoints + descriptors stay in HBM. */
typedef struct {
    int img_idx;
    int img_w, img_h;
    int n;                    /* number of keypoints */
    MisKeyPoint* keypoints;   /* device, n entries */
    void* descriptors;        /* device, n x desc_cols of desc_dtype (ORB: 32 x u8) */
    int desc_cols, desc_dtype;
    void* owner_;             /* internal */
} MisFeatures;

void mis_orb_default_params(MisOrbParams* p); /* the reference's values (image_stitching.cpp:545) */
int mis_orb_create(MisContext* ctx, const MisOrbParams* p, int max_width, int max_height, MisOrb** out);
int mis_orb_destroy(MisOrb* orb);
/* computeImageFeatures(finder, img, features[i]) -- replaces image_stitching.cpp:613.
 * `bgr` is 8UC3 interleaved BGR.  Keypoints are ordered by level, then response (descending), y, x. */
int mis_orb_detect(MisOrb* orb, const MisImage* bgr, MisFeatures* out);
/* same, for a batch of frames of one size with a single host synchronisation at the end */
int mis_orb_detect_batch(MisOrb* orb, const MisImage* bgr, int n_images, MisFeatures* out);
/* One-shot hook of this finder's NEXT mis_orb_detect_batch call: fn(user) runs on the calling thread once the batch's device work
 * is enqueued and before the call waits for it (fn = NULL clears a pending hook).  The job sizes its blender there (warpRoi of all
 * cameras: a kernel and a synchronisation on the compose stream), under the feature stage instead of in front of it. */
int mis_orb_on_enqueued(MisOrb* orb, void (*fn)(void*), void* user);
int mis_features_download(MisContext* ctx, const MisFeatures* f, MisKeyPoint* kps_host, void* desc_host);
/* wrap caller-provided (host) keypoints/descriptors as device-resident features */
int mis_features_upload(MisContext* ctx, int img_w, int img_h, int n, const MisKeyPoint* kps_host,
                        const void* desc_host, int desc_cols, int desc_dtype, MisFeatures* out);
int mis_features_free(MisContext* ctx, MisFeatures* f);
/* m feature sets into two dense device arrays (frame i at kps_dst + i * cap * 24 and desc_dst + i * cap * row_bytes, tails
 * zeroed): the send buffers of the descriptor all-gather of a sharded job; device-to-device, no host copy */
int mis_features_pack(MisContext* ctx, const MisFeatures* feats, int m, int cap, int row_bytes, void* kps_dst, void* desc_dst);
/* 2-D device-to-device copy on the context's stream (a rank's finished column strip into the assembled panorama) */
int mis_copy_2d(MisContext* ctx, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t height);
/* stage intermediates of the last mis_orb_detect (host copies, for parity tests):
 * which = 0 gray level (tight w*h), 1 NMS-surviving FAST score map, 2 blurred bordered level */
int mis_orb_debug_level(MisOrb* orb, int level, int which, uint8_t* host_out, int* width, int* height);

/* ---------------------------------------------------------------- matching ------------------ */
/* BestOf2NearestMatcher(try_cuda, match_conf, num_matches_thresh1 = 6, num_matches_thresh2 = 6)
 * -- replaces image_stitching.cpp:647; findHomography defaults (RANSAC, 3.0, 2000, 0.995). */
typedef struct {
    float match_conf;
    int num_matches_thresh1, num_matches_thresh2;
    double ransac_thresh;
    int max_iters;
    double confidence;
} MisMatchParams;

typedef struct { int query_idx, train_idx, img_idx; float distance; } MisDMatch; /* cv::DMatch */

/* cv::detail::MatchesInfo (image_stitching.cpp:642); arrays are host memory owned by the library */
typedef struct {
    int src_img_idx, dst_img_idx;
    int n_matches;
    MisDMatch* matches;
    uint8_t* inliers_mask; /* n_matches entries, NULL when RANSAC did not run */
    int num_inliers;
    int has_H;
    double H[9];
    double confidence;
} MisMatchesInfo;

void mis_match_default_params(MisMatchParams* p);
/* (*matcher)(features, pairwise_matches) -- replaces image_stitching.cpp:653.  `out` is an n*n
 * row-major array (diagonal entries stay default-initialised, (j,i) mirrors (i,j) with H^-1). */
int mis_match_all_pairs(MisContext* ctx, const MisFeatures* feats, int n, const MisMatchParams* p, MisMatchesInfo* out);
/* sharded form: only the pairs with (pair_index % world_size) == rank are matched (others left default) */
int mis_match_pairs_sharded(MisContext* ctx, const MisFeatures* feats, int n, const MisMatchParams* p, int rank,
                            int world_size, MisMatchesInfo* out);
int mis_matches_free(MisMatchesInfo* m, int count);
/* Ordering aid for a caller that overlaps other device work with a matcher call made by another host thread (the
 * job's speculative composition): mis_match_sequence = number of matcher calls this context has started;
 * mis_match_knn_fence(ctx, stream, mis_match_sequence(ctx) + 1 taken BEFORE the other thread calls the matcher, ms)
 * makes `stream` wait for the point of that call behind which other work shares the device well: by default the first draw of
 * the side RANSAC chain, 0.55 ms behind the 2-NN pass (the chains' kernels are few large workgroups that wait for room once another
 * stream's grids fill the compute units; from that point on they hold theirs).  MIS_COMPOSE_GATE in the environment moves it:
 * 1 = the end of the first RANSAC phase, 0 = the end of the 2-NN pass. */
long long mis_match_sequence(MisContext* ctx);
int mis_match_knn_fence(MisContext* ctx, void* stream, long long target_seq, int timeout_ms);
/* The same without a second host thread: a one-shot hook of this context's NEXT matcher call.  fn(user) runs on the thread that
 * calls mis_match_all_pairs / mis_match_pairs_sharded, once all of the call's device work is enqueued and before the call waits
 * for the device (the ~5 ms in which that thread is idle); inside it mis_match_knn_fence(ctx, stream, mis_match_sequence(ctx), 0)
 * returns at once and queues `stream` behind the 2-NN pass.  The hook is not called when the matcher call fails earlier. */
int mis_match_on_enqueued(MisContext* ctx, void (*fn)(void*), void* user);
/* exact 2-NN (distance, trainIdx) for one direction; results in host memory (stage test hook) */
int mis_knn2(MisContext* ctx, const MisFeatures* query, const MisFeatures* train, int* idx2_host, float* dist2_host);
/* cv::findHomography(src, dst, mask, RANSAC, thresh, max_iters, confidence) on host point lists */
int mis_find_homography(MisContext* ctx, const float* src_xy, const float* dst_xy, int n, double thresh, int max_iters,
                        double confidence, double H[9], uint8_t* mask, int* ok);
/* myLeaveBiggestComponent -- replaces image_stitching.cpp:215-278 (host logic on the matcher output) */
int mis_leave_biggest_component(const MisMatchesInfo* pairwise, int n, float conf_threshold, int* indices, int* n_indices);
/* the same on the bare n x n confidence matrix (row-major; what a sharded job has after its all-reduce) */
int mis_leave_biggest_component_conf(const double* confidence, int n, float conf_threshold, int* indices, int* n_indices);

/* ---------------------------------------------------------------- camera refinement --------- */
/* cv::detail::CameraParams (focal, aspect, ppx, ppy, R, t) as the reference fills it (image_stitching.cpp:485-517) */
typedef struct {
    double focal, aspect, ppx, ppy;
    double R[9];   /* row-major */
    double t[3];
} MisCameraParams;
/* (*adjuster)(features, pairwise_matches, cameras) with makePtr<detail::BundleAdjusterReproj>(), setConfThresh,
 * setRefinementMask -- replaces image_stitching.cpp:681-712.  Host logic (as in OpenCV): Levenberg-Marquardt over 7
 * parameters per camera on the inlier correspondences of the pairs above conf_thresh; refine_mask = the reference's
 * ba_refine_mask string ("xxxxx"); the rotations are normalised to the centre of the maximum spanning tree. */
int mis_bundle_adjust_reproj(MisContext* ctx, const MisFeatures* features, const MisMatchesInfo* pairwise, int n, float conf_thresh,
                             const char* refine_mask, MisCameraParams* cameras);
/* waveCorrect(rmats, wave_correct) -- replaces image_stitching.cpp:718-726; rmats: n x 9 doubles in place; kind 0 = HORIZ, 1 = VERT */
int mis_wave_correct(double* rmats, int n, int kind);

/* ---------------------------------------------------------------- warp ---------------------- */
/* warper->warpRoi(sz, K, R) -- replaces image_stitching.cpp:1138 (K, R: 3x3 f32 row-major) */
int mis_warp_roi(float scale, int src_width, int src_height, const float K[9], const float R[9], MisRect* roi);
/* the loop `for i: sizes[i], corners[i] = warper->warpRoi(sz, K_i, R_i)` of image_stitching.cpp:1119-1140 in one call:
 * the 2(W+H) border projections of every frame run in one small kernel (one workgroup per frame) instead of on the host;
 * Ks, Rs: n x 9 floats.  Same rois as n calls of mis_warp_roi (nothing is cached between calls in either form). */
int mis_warp_roi_batch(MisContext* ctx, float scale, int src_width, int src_height, int n, const float* Ks, const float* Rs, MisRect* rois);
/* warper->warp(src, K, R, interp, border, dst) -- replaces image_stitching.cpp:985, :988, :1154, :1159.
 * u8 source with 1 or 3 channels; (INTER_LINEAR, BORDER_REFLECT) or (INTER_NEAREST, BORDER_CONSTANT). */
int mis_warp_spherical(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9], int interp,
                       int border, MisImage* dst, MisPoint* tl);
/* fused compose-scale warp: image (LINEAR, REFLECT) converted to 16SC3 plus the validity mask
 * (NEAREST, CONSTANT of an all-255 mask) in one pass -- replaces :1154 + :1157-1159 + :1164. */
int mis_warp_spherical_fused(MisContext* ctx, const MisImage* src_bgr, float scale, const float K[9], const float R[9],
                             MisImage* dst_s16x3, MisImage* dst_mask, MisPoint* tl);
/* the same with the roi already known (the value mis_warp_roi / mis_warp_roi_batch returned for exactly these scale, K, R and
 * source size): skips the border walk the reference repeats inside warp() -- the compose loop's form (:1138 then :1154). */
int mis_warp_spherical_fused_roi(MisContext* ctx, const MisImage* src_bgr, float scale, const float K[9], const float R[9],
                                 const MisRect* roi, MisImage* dst_s16x3, MisImage* dst_mask, MisPoint* tl);
/* the fused warps of n frames (the loop :1086-1220 for all of them) in one grid per 16 frames: the results of n
 * mis_warp_spherical_fused_roi calls (dsts[i] / dmasks[i]: caller's buffers or NULL data as there), without a launch of its own per
 * frame -- a 4K frame alone fills and drains the device for a quarter of its launch */
int mis_warp_spherical_fused_batch(MisContext* ctx, const MisImage* srcs_u8x3, int n, float scale, const float* Ks, const float* Rs, const MisRect* rois,
                                   MisImage* dsts_s16x3, MisImage* dmasks_u8, MisPoint* tls);
/* measurement aid (bench.py): the batch launched `repeats` times between two HIP events on the context's stream; *avg_us = one pass */
int mis_warp_spherical_fused_batch_timed(MisContext* ctx, const MisImage* srcs_u8x3, int n, float scale, const float* Ks, const float* Rs,
                                         const MisRect* rois, MisImage* dsts_s16x3, MisImage* dmasks_u8, MisPoint* tls, int repeats, float* avg_us);
/* measurement aid: the same warp with the main kernel launched `repeats` times back to back on the context's
 * stream between two HIP events; *avg_us = average kernel duration (bench.py's roofline leg: no host gaps). */
int mis_warp_spherical_fused_timed(MisContext* ctx, const MisImage* src_bgr, float scale, const float K[9], const float R[9],
                                   MisImage* dst_s16x3, MisImage* dst_mask, MisPoint* tl, int repeats, float* avg_us);

/* ---------------------------------------------------------------- SIFT ---------------------- */
/* SIFT::create() -- replaces image_stitching/image_stitching.cpp:559 (features_type == "sift").  Defaults of the
 * reference: nfeatures 0 (only value supported), nOctaveLayers 3, contrastThreshold 0.04, edgeThreshold 10, sigma 1.6. */
typedef struct {
    int nfeatures, n_octave_layers;
    double contrast_threshold, edge_threshold, sigma;
} MisSiftParams;
typedef struct MisSift MisSift;
void mis_sift_default_params(MisSiftParams* p);
int mis_sift_create(MisContext* ctx, const MisSiftParams* params /* NULL = defaults */, int max_width, int max_height, MisSift** out);
int mis_sift_destroy(MisSift* sift);
/* computeImageFeatures(finder, img, features) -- replaces image_stitching.cpp:613 for the SIFT finder: keypoints in the
 * order of KeyPointsFilter::removeDuplicatedSorted, descriptors n x 128 f32 with integer values 0..255 (feeds K8). */
int mis_sift_detect(MisSift* sift, const MisImage* bgr, MisFeatures* out);
/* n frames of one size, two in flight (second scale space allocated on first use); out[i].img_idx = i as at :614 */
int mis_sift_detect_batch(MisSift* sift, const MisImage* bgr, int n_images, MisFeatures* out);
/* test aid: one image of the Gaussian (dog = 0) or DoG (dog = 1) pyramid of `bgr`, copied to host_out (may be NULL) */
int mis_sift_debug_level(MisSift* sift, const MisImage* bgr, int octave, int layer, int dog, float* host_out, int* width, int* height);

/* ---------------------------------------------------------------- image operators ----------- */
/* cv::resize(src, dst, dsize, fx, fy, INTER_LINEAR_EXACT) -- replaces image_stitching.cpp:580 (work scale), :619 (seam
 * scale), :1144 (compose scale).  8UC1 / 8UC3.  dst_w, dst_h > 0: that size (scale = dsize / ssize); otherwise
 * dsize = (cvRound(w * fx), cvRound(h * fy)) and the coordinate scale is exactly 1/fx, 1/fy, as in resize(). */
int mis_resize_linear_exact(MisContext* ctx, const MisImage* src, int dst_w, int dst_h, double fx, double fy, MisImage* dst);
/* cv::rotate(src, dst, code) -- replaces image_stitching.cpp:571 (ROTATE_90_CLOCKWISE = 0), :576 (ROTATE_180 = 1);
 * 2 = ROTATE_90_COUNTERCLOCKWISE.  8UC1 / 8UC3. */
int mis_rotate(MisContext* ctx, const MisImage* src, int rotate_code, MisImage* dst);
/* dilate(masks_warped[i], 3x3) -> resize(to mask_warped.size(), INTER_LINEAR_EXACT) -> mask_warped &= ... in one
 * pass -- replaces image_stitching.cpp:1169-1171.  Both 8UC1; mask_warped is updated in place. */
int mis_seam_mask_apply(MisContext* ctx, const MisImage* seam_mask_warped, MisImage* mask_warped);

/* ---- exposure compensation and seam finders between warp and blend ----
 * Replaces ExposureCompensator::createDefault(GAIN_BLOCKS) with setNrFeeds(1), setNrGainsFilteringIterations(2),
 * setBlockSize(64, 64) (image_stitching.cpp:1002-1016), compensator->feed(corners, images_warped, masks_warped) (:1023) and
 * compensator->apply(img_idx, corners[img_idx], img_warped, mask_warped) (:1162).  One feed only (the reference's value).
 * Images are 8UC3, masks 8UC1 (255 = valid), host or device. */
typedef struct MisCompensator MisCompensator;
int mis_compensator_create(MisContext* ctx, int block_width, int block_height, int nr_gain_filtering_iterations, MisCompensator** out);
int mis_compensator_destroy(MisCompensator* c);
int mis_compensator_feed(MisCompensator* c, const MisPoint* corners, const MisImage* images, const MisImage* masks, int n);
/* smoothed gain map of one image (one float per block, row-major); map_host may be NULL to query the grid size */
int mis_compensator_gain_map(const MisCompensator* c, int index, float* map_host, int capacity, int* blocks_x, int* blocks_y);
/* image *= gains, in place; 8UC3, or the 16SC3 image the fused warp produces (values 0..255) */
int mis_compensator_apply(MisCompensator* c, int index, MisImage* image);
/* VoronoiSeamFinder::find (seam_find_type "voronoi", image_stitching.cpp:1031): masks (8UC1) are edited in place.
 * "no" (NoSeamFinder) needs no call; the reference's default, DpSeamFinder(COLOR) ("dp_color"), is mis_seam_dp below. */
int mis_seam_voronoi(MisContext* ctx, const MisPoint* corners, MisImage* masks, int n);
/* seam_finder = makePtr<detail::DpSeamFinder>(DpSeamFinder::COLOR); seam_finder->find(images_warped_f, corners, masks_warped)
 * -- replaces image_stitching.cpp:1056-1057, :1065 (the reference's default seam finder, "dp_color").  images: the seam-scale
 * warped 8UC3 images (converted to float inside, as :992-994 does); masks: 8U, edited in place.  Host logic on copies of the
 * (~0.1 MP) images; device or host buffers.  cost_func: MIS_SEAM_DP_COLOR (COLOR_GRAD is not built). */
#define MIS_SEAM_DP_COLOR 0
int mis_seam_dp(MisContext* ctx, const MisPoint* corners, const MisImage* images, MisImage* masks, int n, int cost_func);

/* ---------------------------------------------------------------- blend --------------------- */
/* reference-side blender sizing, image_stitching.cpp:1176-1190: returns the blend type to use in
 * *type_out and fills num_bands (MULTI_BAND) or sharpness (FEATHER) */
int mis_blend_config(int blend_type, float blend_strength, int pano_width, int pano_height, int* type_out, int* num_bands,
                     float* sharpness);
int mis_result_roi(const MisPoint* corners, const MisSize* sizes, int n, MisRect* roi); /* cv::detail::resultRoi */
/* Blender::createDefault(type) + setNumBands / setSharpness -- replaces :1175, :1183, :1189 */
int mis_blender_create(MisContext* ctx, int type, int num_bands, float sharpness, MisBlender** out);
int mis_blender_destroy(MisBlender* b);
int mis_blender_prepare(MisBlender* b, const MisPoint* corners, const MisSize* sizes, int n); /* :1192 */
int mis_blender_num_bands(const MisBlender* b);
/* blender->feed(img_warped_s [16SC3], mask_warped [8U], corners[i]) -- replaces :1218 */
int mis_blender_feed(MisBlender* b, const MisImage* img_s16x3, const MisImage* mask_u8, MisPoint tl);
/* the feeds of n frames of the loop :1086-1220 in one call: the result of mis_blender_feed(imgs[0], ...) ... mis_blender_feed(imgs[n-1], ...)
 * in that order, bit for bit; the multi-band blender builds the frames' Gaussian pyramids together (one launch per level for all
 * frames instead of one per level and frame) before it accumulates the Laplacians frame by frame */
int mis_blender_feed_batch(MisBlender* b, const MisImage* imgs_s16x3, const MisImage* masks_u8, const MisPoint* tls, int n);
/* blender->blend(result, result_mask) -- replaces :1225 */
int mis_blender_blend(MisBlender* b, MisImage* dst_s16x3, MisImage* dst_mask);
/* blend() for the panorama columns x0 .. x1 - 1 only (relative to the result roi; x1 is clipped to its width): the same values
 * as those columns of mis_blender_blend's result, computed from the accumulators of that strip + a halo of two columns per
 * level.  A rank that owns a column strip of the panorama finalises only that strip. */
int mis_blender_blend_columns(MisBlender* b, int x0, int x1, MisImage* dst_s16x3, MisImage* dst_mask);
/* Multi-GPU blend exchange (replaces nothing in the single-process reference: its feed loop :1218 adds every frame into one
 * pyramid; N ranks add theirs into N pyramids and exchange rectangles).  A rectangle of accumulator level `level`,
 * columns x0 .. x1 - 1, rows y0 .. y1 - 1, lives in a byte buffer at `offset`: 16SC3 Laplacian sums row by row (6 B per pixel,
 * the block rounded up to 16 B), then the f32 weight sums (4 B per pixel, rounded up to 16 B).
 *   pack: accumulators -> buffer;   add: accumulators += buffer (16-bit sums wrap: any order gives the same bits; f32 sums are
 *   added in call order);   zero: accumulators = 0.   All on the blender's stream, device buffers. */
typedef struct { int level, x0, y0, x1, y1; unsigned long long offset; } MisLevelRect;
int mis_blender_pack_rects(MisBlender* b, const MisLevelRect* rects, int n, void* dev_buf, size_t bytes);
int mis_blender_add_rects(MisBlender* b, const MisLevelRect* rects, int n, const void* dev_buf, size_t bytes);
int mis_blender_zero_rects(MisBlender* b, const MisLevelRect* rects, int n);
/* The per-frame body of the compositing loop for n frames in one call: fused warp of frames[i] with (Ks + 9 i, Rs + 9 i)
 * at `scale` into library-owned device blocks of size rois[i] (= mis_warp_roi of the frame), then feed -- replaces
 * image_stitching.cpp:1154-1164 and :1218 for every image of the loop at :1086.  Same results as the single calls. */
int mis_compose_frames(MisBlender* b, const MisImage* frames, int n, float scale, const float* Ks, const float* Rs, const MisRect* rois);
/* the rectangle of the (padded) panorama a feed of a width x height frame at `tl` touches: MultiBandBlender::feed's tile
 * (frame + 3 * 2^bands margin, clipped, aligned to 2^bands); P_b of SURVEY 8(d) = tile.width * tile.height */
int mis_blender_feed_rect(const MisBlender* b, int width, int height, MisPoint tl, MisRect* tile);
/* accumulated pyramid level before blend() (host copies, parity tests / multi-GPU reduction hooks) */
int mis_blender_level_info(const MisBlender* b, int level, int* width, int* height, void** lap_dev, void** weight_dev);

#ifdef __cplusplus
}
#endif
#endif
