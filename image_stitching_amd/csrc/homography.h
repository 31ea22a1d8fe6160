// homography.h -- internal interface of the batched cv::findHomography(RANSAC) engine (homography.hip).
#pragma once
#include "common.h"

// One findHomography problem: n correspondences src -> dst (f32 x,y pairs), optional inlier mask.
struct HomoCall {
    const float* src;
    const float* dst;
    uint8_t* mask;   // n bytes or null
    long long pt_off;  // offset of this problem's points in the batch-wide scratch arrays
    int n;
    int active;      // 0: skipped (result: ok = 0)
};

struct HomoResult {
    double H[9];
    int ok;
    int iters;   // RANSAC hypotheses evaluated
    int ninl;    // inliers of the best model
    int pad;
};

// Device workspace for a batch of problems (grow-only, owned by the caller).
struct HomoBatch {
    int count = 0;            // problems
    long long points = 0;     // total points over all problems
    int max_iters = 0;
    void* mem = nullptr;
    size_t bytes = 0;
    // carved pointers (device)
    HomoCall* calls = nullptr;
    HomoResult* results = nullptr;
    void* state = nullptr;
    int* sub_idx = nullptr;
    double* Hc = nullptr;
    int* valid = nullptr;
    int* good = nullptr;
    float* scr = nullptr;     // 4 floats per point: compressed inliers
    double* rec = nullptr;    // 10 doubles per point: per-point terms of the DLT / LM sums
    unsigned* draw_next = nullptr;  // per problem: stream-position tables of the subset drawing
    int* draw_idx = nullptr;
    int* fin = nullptr;       // per problem: RANSAC phase in which it finished (0 / 1), -1 while unfinished
};

int homo_batch_reserve(MisContext* ctx, HomoBatch* b, int count, long long points, int max_iters);
int homo_batch_debug_states(MisContext* ctx, const HomoBatch* b, int* out, int cap);
void homo_batch_release(HomoBatch* b);
// `calls` (device array of b->count entries) must be filled before this is enqueued on ctx->stream.
// phases: 0 = hypotheses [0, PHASE0) + replay + tails of the problems that finish there; 1 = the rest; 2 = both;
// 3 / 6 = like 0 / 1 without the tails (left pending), 4 = the pending tails of phase 0 (any stream, concurrently with a phases = 1 run);
// 10 + 2 w / 11 + 2 w = the pending tails of phase w in two steps: mask + inlier compaction / DLT + LM refinement.
// `stream` = nullptr: the context's stream.
// Optional ordering hooks of a run: rec is recorded behind the second phase's draw_kernel (rec_pos 0), its 4-point solves (1) or the
// FIRST phase's draw (2); rec_hyp0 behind the first phase's solves; the second phase's solves wait for wait_hyp1.
struct HomoSync { hipEvent_t rec = nullptr; int rec_pos = 0; hipEvent_t rec_hyp0 = nullptr; hipEvent_t wait_hyp1 = nullptr; };
int homo_batch_run(MisContext* ctx, HomoBatch* b, double thresh, int max_iters, double confidence, int phases = 2, hipStream_t stream = nullptr,
                   const HomoSync* sync = nullptr);
