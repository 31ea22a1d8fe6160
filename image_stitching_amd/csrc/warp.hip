// warp.hip -- spherical rotation warp, replaces the reference's
// cv::detail::SphericalWarper calls: image_stitching/image_stitching.cpp:973/:1117 (create(scale)),
// :985/:988/:1154/:1159 (warp), :1138 (warpRoi), :1164 (convertTo CV_16S, fused here).
//
// One pass per frame: the inverse map (mapBackward) is evaluated in registers -- OpenCV's xmap/ymap
// (8 B per output pixel written and read back) never exist -- and the bilinear gather (INTER_BITS = 5
// coordinates, Q15 weights, BORDER_REFLECT) writes the 16SC3 image and the 8U validity mask directly.
// sin/cos of the column angle u and of the row angle v are separable: a tile computes them once
// into LDS (128 + 16 evaluations per 2048 pixels).
#include "common.h"
#include "dev_math.h"
#include <mutex>
#include <vector>

namespace {

struct Projector {
    float scale;
    float k[9], rinv[9], r_kinv[9], k_rinv[9];
};

// ProjectorBase::setCameraParams (stitching/src/warpers.cpp): float matrices, double intermediates
void projector_set(Projector* p, float scale, const float K[9], const float R[9]) {
    double kinv[9], d;
    float kinv_f[9];
    p->scale = scale;
    for (int i = 0; i < 9; i++) p->k[i] = K[i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p->rinv[i * 3 + j] = R[j * 3 + i];
    auto KD = [&](int r, int c) { return (double)K[r * 3 + c]; };
    d = KD(0, 0) * (KD(1, 1) * KD(2, 2) - KD(1, 2) * KD(2, 1)) - KD(0, 1) * (KD(1, 0) * KD(2, 2) - KD(1, 2) * KD(2, 0)) +
        KD(0, 2) * (KD(1, 0) * KD(2, 1) - KD(1, 1) * KD(2, 0));
    if (d != 0.) d = 1. / d;
    kinv[0] = (KD(1, 1) * KD(2, 2) - KD(1, 2) * KD(2, 1)) * d;
    kinv[1] = (KD(0, 2) * KD(2, 1) - KD(0, 1) * KD(2, 2)) * d;
    kinv[2] = (KD(0, 1) * KD(1, 2) - KD(0, 2) * KD(1, 1)) * d;
    kinv[3] = (KD(1, 2) * KD(2, 0) - KD(1, 0) * KD(2, 2)) * d;
    kinv[4] = (KD(0, 0) * KD(2, 2) - KD(0, 2) * KD(2, 0)) * d;
    kinv[5] = (KD(0, 2) * KD(1, 0) - KD(0, 0) * KD(1, 2)) * d;
    kinv[6] = (KD(1, 0) * KD(2, 1) - KD(1, 1) * KD(2, 0)) * d;
    kinv[7] = (KD(0, 1) * KD(2, 0) - KD(0, 0) * KD(2, 1)) * d;
    kinv[8] = (KD(0, 0) * KD(1, 1) - KD(0, 1) * KD(1, 0)) * d;
    for (int i = 0; i < 9; i++) kinv_f[i] = (float)kinv[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0, t = 0;
            for (int k = 0; k < 3; k++) {
                s += (double)R[i * 3 + k] * (double)kinv_f[k * 3 + j];
                t += (double)K[i * 3 + k] * (double)p->rinv[k * 3 + j];
            }
            p->r_kinv[i * 3 + j] = (float)s;
            p->k_rinv[i * 3 + j] = (float)t;
        }
}

// SphericalProjector::mapForward (warpers_inl.hpp)
void map_forward(const Projector* p, float x, float y, float* u, float* v) {
    const float* m = p->r_kinv;
    float x_ = (m[0] * x + m[1] * y) + m[2];
    float y_ = (m[3] * x + m[4] * y) + m[5];
    float z_ = (m[6] * x + m[7] * y) + m[8];
    *u = p->scale * mis_atan2f(x_, z_);
    float w = y_ / sqrtf((x_ * x_ + y_ * y_) + z_ * z_);
    *v = p->scale * (MIS_PI_F - mis_acosf(w == w ? w : 0));
}

// SphericalWarper::detectResultRoi: border projection + pole tests; 2(W+H) points on the host
void detect_result_roi(const Projector* p, int sw, int sh, int* tlx, int* tly, int* brx, int* bry) {
    float tl_uf = FLT_MAX, tl_vf = FLT_MAX, br_uf = -FLT_MAX, br_vf = -FLT_MAX, u, v;
    auto upd = [&]() {
        if (u < tl_uf) tl_uf = u;
        if (v < tl_vf) tl_vf = v;
        if (u > br_uf) br_uf = u;
        if (v > br_vf) br_vf = v;
    };
    for (int x = 0; x < sw; ++x) {
        map_forward(p, (float)x, 0, &u, &v); upd();
        map_forward(p, (float)x, (float)(sh - 1), &u, &v); upd();
    }
    for (int y = 0; y < sh; ++y) {
        map_forward(p, 0, (float)y, &u, &v); upd();
        map_forward(p, (float)(sw - 1), (float)y, &u, &v); upd();
    }
    tl_uf = (float)(int)tl_uf; tl_vf = (float)(int)tl_vf; br_uf = (float)(int)br_uf; br_vf = (float)(int)br_vf;
    for (int pass = 0; pass < 2; pass++) {
        float x = p->rinv[1], y = pass == 0 ? p->rinv[4] : -p->rinv[4], z = p->rinv[7];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < (float)sw && y_ > 0.f && y_ < (float)sh) {
                float pv = pass == 0 ? (float)(3.14159265358979323846 * (double)p->scale) : 0.f;
                if (0.f < tl_uf) tl_uf = 0.f;
                if (pv < tl_vf) tl_vf = pv;
                if (0.f > br_uf) br_uf = 0.f;
                if (pv > br_vf) br_vf = pv;
            }
        }
    }
    *tlx = (int)tl_uf; *tly = (int)tl_vf; *brx = (int)br_uf; *bry = (int)br_vf;
}

// detectResultRoi walks 2(W+H) border pixels on the host (~0.1 ms at 4K); the compose loop asks for the
// same (scale, K, R, size) twice (warpRoi, then warp), so the last few answers are cached.
struct RoiKey {
    float scale, K[9], R[9];
    int w, h;
};
struct RoiEntry {
    RoiKey key;
    Projector proj;
    int tlx, tly, brx, bry;
};
std::mutex g_roi_mutex;
std::vector<RoiEntry> g_roi_cache;

void projector_and_roi(float scale, const float K[9], const float R[9], int w, int h, Projector* p, int* tlx, int* tly, int* brx, int* bry) {
    RoiKey key;
    memset(&key, 0, sizeof(key));
    key.scale = scale; key.w = w; key.h = h;
    memcpy(key.K, K, sizeof(key.K)); memcpy(key.R, R, sizeof(key.R));
    {
        std::lock_guard<std::mutex> lock(g_roi_mutex);
        for (const RoiEntry& e : g_roi_cache)
            if (memcmp(&e.key, &key, sizeof(key)) == 0) { *p = e.proj; *tlx = e.tlx; *tly = e.tly; *brx = e.brx; *bry = e.bry; return; }
    }
    projector_set(p, scale, K, R);
    detect_result_roi(p, w, h, tlx, tly, brx, bry);
    std::lock_guard<std::mutex> lock(g_roi_mutex);
    if (g_roi_cache.size() >= 256) g_roi_cache.erase(g_roi_cache.begin());
    g_roi_cache.push_back(RoiEntry{key, *p, *tlx, *tly, *brx, *bry});
}

struct WarpArgs {
    float m[9];  // k_rinv
    float scale;
    int tlx, tly, dw, dh, sw, sh, cn;
    const uint8_t* src;
    size_t sstride;
    void* dst;       // s16x3 (fused) or u8 x cn
    size_t dstride;  // bytes
    uint8_t* mask;
    size_t mstride;
};

constexpr int TILE_W = 128, TILE_H = 16;

// SphericalProjector::mapBackward with the separable trig pre-evaluated
__device__ __forceinline__ void map_backward(const float* m, float sinu, float cosu, float sinv, float cosv, float* x, float* y) {
    float x_ = sinv * sinu, y_ = cosv, z_ = sinv * cosu;
    float xx = (m[0] * x_ + m[1] * y_) + m[2] * z_;
    float yy = (m[3] * x_ + m[4] * y_) + m[5] * z_;
    float z = (m[6] * x_ + m[7] * y_) + m[8] * z_;
    if (z > 0) { *x = xx / z; *y = yy / z; }
    else { *x = -1.f; *y = -1.f; }
}

// remap INTER_LINEAR, BORDER_REFLECT on u8: INTER_BITS = 5, Q15 weights, round at bit 14
template <int CN>
__device__ __forceinline__ void sample_linear(const uint8_t* src, size_t stride, int sw, int sh, float x, float y, int* out) {
    int sxq = mis_round_sat_f(x * 32.f), syq = mis_round_sat_f(y * 32.f);
    int fx = sxq & 31, fy = syq & 31;
    int sx = mis_sat_short(sxq >> 5), sy = mis_sat_short(syq >> 5);
    int x0, x1, y0, y1;
    if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) { x0 = sx; x1 = sx + 1; y0 = sy; y1 = sy + 1; }
    else { x0 = mis_reflect(sx, sw); x1 = mis_reflect(sx + 1, sw); y0 = mis_reflect(sy, sh); y1 = mis_reflect(sy + 1, sh); }
    int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const uint8_t* r0 = src + (size_t)y0 * stride;
    const uint8_t* r1 = src + (size_t)y1 * stride;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        int s = r0[x0 * CN + c] * w00 + r0[x1 * CN + c] * w01 + r1[x0 * CN + c] * w10 + r1[x1 * CN + c] * w11;
        out[c] = (s + (1 << 14)) >> 15;  // always within 0..255
    }
}

// remap INTER_NEAREST, BORDER_CONSTANT(0): is the nearest source pixel inside the image?
__device__ __forceinline__ bool nearest_inside(int sw, int sh, float x, float y, int* sx, int* sy) {
    *sx = mis_sat_short(mis_round_sat_f(x));
    *sy = mis_sat_short(mis_round_sat_f(y));
    return (unsigned)*sx < (unsigned)sw && (unsigned)*sy < (unsigned)sh;
}

__device__ __forceinline__ void tile_trig(const WarpArgs& a, int tx0, int ty0, float* su, float* cu, float* sv, float* cv) {
    int t = threadIdx.x;
    if (t < TILE_W) {
        float u = (float)(a.tlx + tx0 + t) / a.scale;
        mis_sincosf(u, &su[t], &cu[t]);
    } else if (t < TILE_W + TILE_H) {
        int r = t - TILE_W;
        float v = (float)(a.tly + ty0 + r) / a.scale;
        mis_sincosf(MIS_PI_F - v, &sv[r], &cv[r]);
    }
    __syncthreads();
}

// Fused compose-scale warp: 8UC3 source -> 16SC3 image + 8U mask.  256 threads = 4 waves; a wave
// owns 4 consecutive rows of the 128x16 tile, a lane owns 2 adjacent columns (12-byte store).
__global__ __launch_bounds__(256) void warp_fused_kernel(WarpArgs a) {
    __shared__ float su[TILE_W], cu[TILE_W], sv[TILE_H], cv[TILE_H];
    const int tx0 = blockIdx.x * TILE_W, ty0 = blockIdx.y * TILE_H;
    tile_trig(a, tx0, ty0, su, cu, sv, cv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cx = 2 * lane, gx = tx0 + cx;
    if (gx >= a.dw) return;
    const bool two = gx + 1 < a.dw;
    const float su0 = su[cx], cu0 = cu[cx], su1 = su[cx + 1], cu1 = cu[cx + 1];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = wave * 4 + i, gy = ty0 + r;
        if (gy >= a.dh) break;
        const float s_v = sv[r], c_v = cv[r];
        float x, y;
        int p0[3], p1[3] = {0, 0, 0}, sx, sy;
        map_backward(a.m, su0, cu0, s_v, c_v, &x, &y);
        sample_linear<3>(a.src, a.sstride, a.sw, a.sh, x, y, p0);
        unsigned m0 = nearest_inside(a.sw, a.sh, x, y, &sx, &sy) ? 255u : 0u, m1 = 0u;
        if (two) {
            map_backward(a.m, su1, cu1, s_v, c_v, &x, &y);
            sample_linear<3>(a.src, a.sstride, a.sw, a.sh, x, y, p1);
            m1 = nearest_inside(a.sw, a.sh, x, y, &sx, &sy) ? 255u : 0u;
        }
        uint8_t* drow = (uint8_t*)a.dst + (size_t)gy * a.dstride + (size_t)gx * 6;
        uint8_t* mrow = a.mask + (size_t)gy * a.mstride + gx;
        if (two) {
            uint3 w;
            w.x = (unsigned)p0[0] | ((unsigned)p0[1] << 16);
            w.y = (unsigned)p0[2] | ((unsigned)p1[0] << 16);
            w.z = (unsigned)p1[1] | ((unsigned)p1[2] << 16);
            *reinterpret_cast<uint3*>(drow) = w;
            *reinterpret_cast<unsigned short*>(mrow) = (unsigned short)(m0 | (m1 << 8));
        } else {
            int16_t* d = reinterpret_cast<int16_t*>(drow);
            d[0] = (int16_t)p0[0]; d[1] = (int16_t)p0[1]; d[2] = (int16_t)p0[2];
            mrow[0] = (uint8_t)m0;
        }
    }
}

// General warp (seam-scale path and plain masks): u8 with CN channels, one column per lane.
template <int CN, bool LINEAR>
__global__ __launch_bounds__(256) void warp_u8_kernel(WarpArgs a) {
    __shared__ float su[TILE_W], cu[TILE_W], sv[TILE_H], cv[TILE_H];
    const int tx0 = blockIdx.x * TILE_W, ty0 = blockIdx.y * TILE_H;
    tile_trig(a, tx0, ty0, su, cu, sv, cv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int half = 0; half < 2; half++) {
        const int cx = lane + 64 * half, gx = tx0 + cx;
        if (gx >= a.dw) continue;
        for (int i = 0; i < 4; i++) {
            const int r = wave * 4 + i, gy = ty0 + r;
            if (gy >= a.dh) break;
            float x, y;
            map_backward(a.m, su[cx], cu[cx], sv[r], cv[r], &x, &y);
            uint8_t* d = (uint8_t*)a.dst + (size_t)gy * a.dstride + (size_t)gx * CN;
            if (LINEAR) {
                int p[CN];
                sample_linear<CN>(a.src, a.sstride, a.sw, a.sh, x, y, p);
#pragma unroll
                for (int c = 0; c < CN; c++) d[c] = (uint8_t)p[c];
            } else {
                int sx, sy;
                bool in = nearest_inside(a.sw, a.sh, x, y, &sx, &sy);
#pragma unroll
                for (int c = 0; c < CN; c++) d[c] = in ? a.src[(size_t)sy * a.sstride + (size_t)sx * CN + c] : (uint8_t)0;
            }
        }
    }
}

int setup(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9], WarpArgs* a, int* brx, int* bry) {
    MIS_CHECK(ctx, src && K && R, MIS_E_INVALID, "null argument");
    MIS_CHECK(ctx, src->dtype == MIS_U8 && (src->channels == 1 || src->channels == 3), MIS_E_UNSUPPORTED,
              "warp source must be 8UC1 or 8UC3");
    MIS_CHECK(ctx, src->width >= 2 && src->height >= 2 && src->width <= 32767 && src->height <= 32767, MIS_E_INVALID,
              "source size %dx%d out of range", src->width, src->height);
    MIS_CHECK(ctx, scale > 0.f, MIS_E_INVALID, "scale must be positive");
    Projector p;
    int tlx, tly;
    projector_and_roi(scale, K, R, src->width, src->height, &p, &tlx, &tly, brx, bry);
    for (int i = 0; i < 9; i++) a->m[i] = p.k_rinv[i];
    a->scale = scale; a->tlx = tlx; a->tly = tly;
    a->dw = *brx - tlx + 1; a->dh = *bry - tly + 1;
    a->sw = src->width; a->sh = src->height; a->cn = src->channels;
    MIS_CHECK(ctx, a->dw > 0 && a->dh > 0 && (long long)a->dw * a->dh < (1ll << 31), MIS_E_INVALID, "degenerate warp roi %dx%d", a->dw, a->dh);
    return MIS_OK;
}

}  // namespace

extern "C" int mis_warp_roi(float scale, int w, int h, const float K[9], const float R[9], MisRect* roi) {
    if (!K || !R || !roi || w < 1 || h < 1 || !(scale > 0.f)) return MIS_E_INVALID;
    Projector p;
    int tlx, tly, brx, bry;
    projector_and_roi(scale, K, R, w, h, &p, &tlx, &tly, &brx, &bry);
    roi->x = tlx; roi->y = tly; roi->width = brx + 1 - tlx; roi->height = bry + 1 - tly;
    return MIS_OK;
}

extern "C" int mis_warp_spherical(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9],
                                  int interp, int border, MisImage* dst, MisPoint* tl) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, (interp == MIS_INTER_LINEAR && border == MIS_BORDER_REFLECT) || (interp == MIS_INTER_NEAREST && border == MIS_BORDER_CONSTANT),
              MIS_E_UNSUPPORTED, "supported: (LINEAR, REFLECT) and (NEAREST, CONSTANT)");
    WarpArgs a;
    int brx, bry, rc;
    if ((rc = setup(ctx, src, scale, K, R, &a, &brx, &bry)) != MIS_OK) return rc;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage din, dout;
    if ((rc = mis_dev_image_in(ctx, src, &din)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, dst, a.dw, a.dh, src->channels, MIS_U8, &dout)) != MIS_OK) { mis_dev_image_release(ctx, &din); return rc; }
    a.src = (const uint8_t*)din.data; a.sstride = din.stride;
    a.dst = dout.data; a.dstride = dout.stride; a.mask = nullptr; a.mstride = 0;
    dim3 grid((a.dw + TILE_W - 1) / TILE_W, (a.dh + TILE_H - 1) / TILE_H), block(256);
    if (src->channels == 3) {
        if (interp == MIS_INTER_LINEAR) hipLaunchKernelGGL((warp_u8_kernel<3, true>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((warp_u8_kernel<3, false>), grid, block, 0, ctx->stream, a);
    } else {
        if (interp == MIS_INTER_LINEAR) hipLaunchKernelGGL((warp_u8_kernel<1, true>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((warp_u8_kernel<1, false>), grid, block, 0, ctx->stream, a);
    }
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dout)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_release(ctx, &din)) != MIS_OK) return rc;
    if (tl) { tl->x = a.tlx; tl->y = a.tly; }
    return MIS_OK;
}

extern "C" int mis_warp_spherical_fused(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9],
                                        MisImage* dst, MisImage* dmask, MisPoint* tl) {
    if (!ctx) return MIS_E_INVALID;
    WarpArgs a;
    int brx, bry, rc;
    if ((rc = setup(ctx, src, scale, K, R, &a, &brx, &bry)) != MIS_OK) return rc;
    MIS_CHECK(ctx, src->channels == 3, MIS_E_UNSUPPORTED, "fused warp needs an 8UC3 source");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage din, dout, dm;
    if ((rc = mis_dev_image_in(ctx, src, &din)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, dst, a.dw, a.dh, 3, MIS_S16, &dout)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, dmask, a.dw, a.dh, 1, MIS_U8, &dm)) != MIS_OK) return rc;
    MIS_CHECK(ctx, dout.stride % 4 == 0 && dm.stride % 2 == 0 && ((uintptr_t)dout.data % 4) == 0 && ((uintptr_t)dm.data % 2) == 0,
              MIS_E_INVALID, "fused warp outputs need 4-byte (image) / 2-byte (mask) aligned rows");
    a.src = (const uint8_t*)din.data; a.sstride = din.stride;
    a.dst = dout.data; a.dstride = dout.stride; a.mask = (uint8_t*)dm.data; a.mstride = dm.stride;
    dim3 grid((a.dw + TILE_W - 1) / TILE_W, (a.dh + TILE_H - 1) / TILE_H), block(256);
    hipLaunchKernelGGL(warp_fused_kernel, grid, block, 0, ctx->stream, a);
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dout)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_commit(ctx, dmask, &dm)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_release(ctx, &din)) != MIS_OK) return rc;
    if (tl) { tl->x = a.tlx; tl->y = a.tly; }
    return MIS_OK;
}
