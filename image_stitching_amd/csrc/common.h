// common.h -- internal declarations shared by the libmistitch.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/mistitch.h"

// per-context caches that outlive a call (grow-only device workspaces); freed with the context
struct MisWorkspace {
    virtual ~MisWorkspace() {}
};

struct MisContext {
    int device = 0;
    int num_cu = 256;   // compute units of the device (persistent kernels size their grids from it)
    MisWorkspace* match_ws = nullptr;
    // recycled device blocks (size, pointer): feature sets are allocated and released every frame, and a
    // hipFree would synchronise the whole device each time
    std::vector<std::pair<size_t, void*>> pool;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // grow-only scratch for host<->device staging
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // The context's two auxiliary streams (non-blocking, created on first use).  Every stage that forks work off the
    // context's stream takes them from here -- the ORB batch's helper lanes, then the matcher's side chains -- so a job owns
    // four streams in all (this one, the two auxiliaries, the caller's compose stream): one per hardware queue of the
    // runtime's default of four, whichever stage is running.
    hipStream_t aux[2] = {nullptr, nullptr};
    // pinned, device-visible host block of mis_warp_roi_batch (jobs in, extremes out)
    void* roi_pinned = nullptr;
    size_t roi_pinned_bytes = 0;
    void* host_stage = nullptr;      // pinned host scratch of host-logic entries (mis_seam_dp): grow-only, see mis_host_stage
    size_t host_stage_bytes = 0;
};

int mis_set_error(MisContext* ctx, int code, const char* fmt, ...);
int mis_host_stage(MisContext* ctx, size_t bytes, void** out);   // pinned scratch of at least `bytes` (valid until the next call)
int mis_pool_alloc(MisContext* ctx, size_t bytes, void** out, size_t* got);
int mis_aux_stream(MisContext* ctx, int k, hipStream_t* out);   // k = 0, 1
void mis_pool_free(MisContext* ctx, void* p, size_t bytes);
// a device block for a feature set from ctx's pool of recycled blocks, registered so that mis_features_free recycles it (orb.hip)
int mis_feat_block_alloc(MisContext* ctx, size_t bytes, void** out);

#define MIS_HIP(ctx, call)                                                                             \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return mis_set_error((ctx), MIS_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                                  \
    } while (0)

#define MIS_CHECK(ctx, cond, code, ...)                              \
    do {                                                             \
        if (!(cond)) return mis_set_error((ctx), (code), __VA_ARGS__); \
    } while (0)

static inline size_t mis_dtype_size(int dtype) { return dtype == MIS_U8 ? 1 : (dtype == MIS_S16 ? 2 : 4); }
static inline size_t mis_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Device view of an image: either the caller's device pointer or a staged copy of a host buffer.
struct DevImage {
    void* data = nullptr;
    size_t stride = 0;
    bool owned = false;
};
int mis_dev_image_in(MisContext* ctx, const MisImage* img, DevImage* out);   // read access
int mis_dev_image_release(MisContext* ctx, DevImage* d);
// Prepare an output image (allocate when data == NULL) and, for host outputs, a device twin.
int mis_dev_image_out(MisContext* ctx, MisImage* img, int width, int height, int channels, int dtype, DevImage* out);
int mis_dev_image_commit(MisContext* ctx, const MisImage* img, DevImage* d);  // copy back for host outputs, release

