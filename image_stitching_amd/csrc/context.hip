// context.hip -- MisContext: device + stream + error text, image staging helpers.
#include "common.h"
#include <mutex>
#include <stdarg.h>

int mis_set_error(MisContext* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

static std::mutex g_pool_mutex;      // (the SIFT batch's lanes allocate their output blocks from the caller's context on several host threads)
int mis_pool_alloc(MisContext* ctx, size_t bytes, void** out, size_t* got) {
    // best fit among the recycled blocks (at most 2x the request), else a fresh allocation
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); i++)
        if (ctx->pool[i].first >= bytes && ctx->pool[i].first <= 2 * bytes && (best < 0 || ctx->pool[i].first < ctx->pool[best].first)) best = i;
    if (best >= 0) {
        *out = ctx->pool[best].second; *got = ctx->pool[best].first;
        ctx->pool.erase(ctx->pool.begin() + best);
        return MIS_OK;
    }
    MIS_HIP(ctx, hipMalloc(out, bytes));
    *got = bytes;
    return MIS_OK;
}

// (non-coherent = cached on the host: the CPU reads what a copy put there at memory speed -- through a coherent mapping a byte loop over
// 4 MB took 33 ms; visibility is ordered by the stream synchronisation that follows the copies)
int mis_host_stage(MisContext* ctx, size_t bytes, void** out) {
    if (ctx->host_stage_bytes < bytes) {
        if (ctx->host_stage) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipHostFree(ctx->host_stage)); ctx->host_stage = nullptr; ctx->host_stage_bytes = 0; }
        MIS_HIP(ctx, hipHostMalloc(&ctx->host_stage, bytes + bytes / 2 + 4096, hipHostMallocNonCoherent));
        ctx->host_stage_bytes = bytes + bytes / 2 + 4096;
    }
    *out = ctx->host_stage;
    return MIS_OK;
}

int mis_aux_stream(MisContext* ctx, int k, hipStream_t* out) {
    MIS_CHECK(ctx, k == 0 || k == 1, MIS_E_INVALID, "auxiliary stream index %d", k);
    if (!ctx->aux[k]) {
        // MIS_AUX_PRIO=1: the auxiliary streams (the matcher's side chains) at the device's most urgent priority
        static const bool urgent = getenv("MIS_AUX_PRIO") && atoi(getenv("MIS_AUX_PRIO")) > 0;
        int least = 0, greatest = 0;
        if (urgent && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) MIS_HIP(ctx, hipStreamCreateWithPriority(&ctx->aux[k], hipStreamNonBlocking, greatest));
        else MIS_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux[k], hipStreamNonBlocking));
    }
    *out = ctx->aux[k];
    return MIS_OK;
}

void mis_pool_free(MisContext* ctx, void* p, size_t bytes) {
    // stream-ordered reuse: every consumer of the block was enqueued on ctx->stream before this call
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (p) ctx->pool.emplace_back(bytes, p);
}

extern "C" const char* mis_version(void) { return "mistitch 0.1 (gfx950)"; }

extern "C" int mis_context_create(int device, void* stream, MisContext** out) {
    if (!out) return MIS_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MIS_E_HIP;  // no CPU fallback
    if (device < 0 || device >= count) return MIS_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MIS_E_HIP;
    MisContext* ctx = new MisContext();
    ctx->device = device;
    {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) ctx->num_cu = cu;
    }
    // NULL is the device's default (null) stream -- the same stream torch uses unless told otherwise,
    // so library kernels stay ordered with the caller's copies and allocator reuse.
    ctx->stream = (hipStream_t)stream;
    *out = ctx;
    return MIS_OK;
}

extern "C" int mis_context_destroy(MisContext* ctx) {
    if (!ctx) return MIS_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    delete ctx->match_ws;
    for (auto& b : ctx->pool) hipFree(b.second);
    if (ctx->stage) hipFree(ctx->stage);
    if (ctx->roi_pinned) hipHostFree(ctx->roi_pinned);
    if (ctx->host_stage) hipHostFree(ctx->host_stage);
    for (hipStream_t& a : ctx->aux) if (a) { hipStreamSynchronize(a); hipStreamDestroy(a); a = nullptr; }
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return MIS_OK;
}

extern "C" int mis_stream_create(int device, int priority, void** out) {
    if (!out) return MIS_E_INVALID;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return MIS_E_HIP;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return MIS_E_HIP;
    const int pr = priority > 0 ? least : (priority < 0 ? greatest : 0);   // HIP: numerically smaller = more urgent
    hipStream_t s = nullptr;
    // (Round 3 measured two more knobs here -- a compute-unit mask for the composition stream and a priority override -- slower
    // at every setting / no effect; both are gone.)
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr) != hipSuccess) return MIS_E_HIP;
    *out = (void*)s;
    return MIS_OK;
}

extern "C" int mis_stream_destroy(void* stream) {
    if (!stream) return MIS_OK;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? MIS_OK : MIS_E_HIP;
}

extern "C" int mis_context_synchronize(MisContext* ctx) {
    if (!ctx) return MIS_E_INVALID;
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIS_OK;
}

// ctx's stream waits (on the device, the host does not block) for everything enqueued so far on other's stream
extern "C" int mis_context_wait(MisContext* ctx, MisContext* other) {
    if (!ctx || !other) return MIS_E_INVALID;
    MIS_CHECK(ctx, ctx->device == other->device, MIS_E_INVALID, "the two contexts belong to different devices");
    if (ctx->stream == other->stream) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev;
    MIS_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, other->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
    hipEventDestroy(ev);       // released once the wait has consumed it
    MIS_HIP(ctx, e);
    return MIS_OK;
}

extern "C" const char* mis_last_error(const MisContext* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int mis_image_free(MisContext* ctx, MisImage* img) {
    if (!ctx || !img) return MIS_E_INVALID;
    if (img->data && img->mem == MIS_MEM_DEVICE) {
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MIS_HIP(ctx, hipFree(img->data));
    }
    img->data = nullptr;
    return MIS_OK;
}

int mis_dev_image_in(MisContext* ctx, const MisImage* img, DevImage* out) {
    MIS_CHECK(ctx, img && img->data && img->width > 0 && img->height > 0, MIS_E_INVALID, "null or empty input image");
    size_t row = (size_t)img->width * img->channels * mis_dtype_size(img->dtype);
    MIS_CHECK(ctx, img->stride >= row, MIS_E_INVALID, "stride %zu smaller than a row (%zu)", img->stride, row);
    if (img->mem == MIS_MEM_DEVICE) {
        out->data = img->data; out->stride = img->stride; out->owned = false;
        return MIS_OK;
    }
    size_t pitch = mis_align_up(row, 256);
    MIS_HIP(ctx, hipMalloc(&out->data, pitch * img->height));
    out->stride = pitch; out->owned = true;
    MIS_HIP(ctx, hipMemcpy2DAsync(out->data, pitch, img->data, img->stride, row, img->height, hipMemcpyHostToDevice, ctx->stream));
    return MIS_OK;
}

int mis_dev_image_release(MisContext* ctx, DevImage* d) {
    if (d->owned && d->data) {
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MIS_HIP(ctx, hipFree(d->data));
    }
    d->data = nullptr; d->owned = false;
    return MIS_OK;
}

int mis_dev_image_out(MisContext* ctx, MisImage* img, int width, int height, int channels, int dtype, DevImage* out) {
    MIS_CHECK(ctx, img, MIS_E_INVALID, "null output image");
    size_t row = (size_t)width * channels * mis_dtype_size(dtype);
    if (!img->data) {
        size_t pitch = mis_align_up(row, 256);
        void* p = nullptr;
        MIS_HIP(ctx, hipMalloc(&p, pitch * (size_t)height));
        img->data = p; img->width = width; img->height = height; img->channels = channels;
        img->stride = pitch; img->dtype = dtype; img->mem = MIS_MEM_DEVICE;
    }
    MIS_CHECK(ctx, img->width == width && img->height == height && img->channels == channels && img->dtype == dtype,
              MIS_E_INVALID, "output image is %dx%dx%d dtype %d, expected %dx%dx%d dtype %d", img->width, img->height,
              img->channels, img->dtype, width, height, channels, dtype);
    MIS_CHECK(ctx, img->stride >= row, MIS_E_INVALID, "output stride too small");
    if (img->mem == MIS_MEM_DEVICE) {
        out->data = img->data; out->stride = img->stride; out->owned = false;
        return MIS_OK;
    }
    size_t pitch = mis_align_up(row, 256);
    MIS_HIP(ctx, hipMalloc(&out->data, pitch * (size_t)height));
    out->stride = pitch; out->owned = true;
    return MIS_OK;
}

int mis_dev_image_commit(MisContext* ctx, const MisImage* img, DevImage* d) {
    if (d->owned && d->data) {
        size_t row = (size_t)img->width * img->channels * mis_dtype_size(img->dtype);
        MIS_HIP(ctx, hipMemcpy2DAsync(img->data, img->stride, d->data, d->stride, row, img->height, hipMemcpyDeviceToHost, ctx->stream));
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MIS_HIP(ctx, hipFree(d->data));
    }
    d->data = nullptr; d->owned = false;
    return MIS_OK;
}

// Feature sets of m frames into two dense device arrays for the descriptor all-gather: frame i's keypoints at
// kps_dst + i * cap * 24, its descriptors at desc_dst + i * cap * row_bytes, the tails zeroed.  Device-to-device copies on the
// context's stream; no host round trip.
extern "C" int mis_features_pack(MisContext* ctx, const MisFeatures* feats, int m, int cap, int row_bytes, void* kps_dst, void* desc_dst) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, feats && m >= 0 && cap >= 1 && row_bytes >= 1 && kps_dst && desc_dst, MIS_E_INVALID, "invalid argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < m; i++) {
        const int n = feats[i].n;
        MIS_CHECK(ctx, n >= 0 && n <= cap, MIS_E_INVALID, "frame %d has %d keypoints, capacity %d", i, n, cap);
        uint8_t* k = (uint8_t*)kps_dst + (size_t)i * cap * sizeof(MisKeyPoint);
        uint8_t* d = (uint8_t*)desc_dst + (size_t)i * cap * row_bytes;
        if (n) {
            MIS_HIP(ctx, hipMemcpyAsync(k, feats[i].keypoints, (size_t)n * sizeof(MisKeyPoint), hipMemcpyDeviceToDevice, ctx->stream));
            MIS_HIP(ctx, hipMemcpyAsync(d, feats[i].descriptors, (size_t)n * row_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (n < cap) {
            MIS_HIP(ctx, hipMemsetAsync(k + (size_t)n * sizeof(MisKeyPoint), 0, (size_t)(cap - n) * sizeof(MisKeyPoint), ctx->stream));
            MIS_HIP(ctx, hipMemsetAsync(d + (size_t)n * row_bytes, 0, (size_t)(cap - n) * row_bytes, ctx->stream));
        }
    }
    return MIS_OK;
}

// 2-D device-to-device copy on the context's stream (assembling the finished column strips of a panorama)
extern "C" int mis_copy_2d(MisContext* ctx, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t height) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, dst && src && width_bytes <= dst_pitch && width_bytes <= src_pitch, MIS_E_INVALID, "invalid argument");
    if (!width_bytes || !height) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MIS_HIP(ctx, hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, hipMemcpyDeviceToDevice, ctx->stream));
    return MIS_OK;
}
