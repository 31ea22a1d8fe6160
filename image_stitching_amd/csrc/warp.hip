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

// Fused compose-scale warp: 8UC3 source -> 16SC3 image + 8U mask.
//
// sin/cos of the column angle u = (tlx + x) / scale and of the row angle pi - v are separable, so a
// tiny pre-kernel tabulates them (dw + dh entries) and the main kernel has no trigonometry at all.
// One wave per workgroup (no workgroup barriers): the wave owns a 128 x 4 output tile, a lane owns
// 2 adjacent columns x 4 rows (12-byte stores).  The wave evaluates the map of all its pixels (kept
// in registers), all-reduces the bounding box of the interior taps with shuffles, stages exactly that
// box of the source in LDS with coalesced dword loads, and gathers the 12 taps per pixel from LDS.
// Pixels whose taps need BORDER_REFLECT (outside the frame) and tiles whose box does not fit take the
// global-memory gather in a cold fix-up pass.
constexpr int FT_W = 128, FT_H = 4;
constexpr int STAGE_BYTES = 16 * 512;  // 16 rows x 512 bytes

__global__ __launch_bounds__(256) void warp_trig_kernel(WarpArgs a, float* tab) {
    // tab: su[dw] cu[dw] sv[dh] cv[dh]
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.dw) {
        float u = (float)(a.tlx + i) / a.scale;
        mis_sincosf(u, &tab[i], &tab[a.dw + i]);
    } else if (i < a.dw + a.dh) {
        int r = i - a.dw;
        float v = (float)(a.tly + r) / a.scale;
        mis_sincosf(MIS_PI_F - v, &tab[2 * a.dw + r], &tab[2 * a.dw + a.dh + r]);
    }
}

// cvRound(v) in [0, len): round-half-even maps [-0.5, len - 0.5) into range, and the upper end point
// len - 0.5 too when len - 1 is even (ties go to the even neighbour)
__device__ __forceinline__ bool round_in_range(float v, float hi, bool hi_even) { return v >= -0.5f && (v < hi || (hi_even && v == hi)); }

// bilinear gather of 8 pixels from the staged box.  REFLECT = false: every tap is interior, the four
// taps of a pixel are at fixed offsets from the top-left one.  REFLECT = true: taps are folded once
// (BORDER_REFLECT, coordinates known to lie in [-len, 2 len)) and addressed individually.
template <bool REFLECT>
__device__ __forceinline__ void sample8(const uint8_t* stage, int lbase, int pitch, int sw, int sh, const int* sxq, const int* syq, int (*p)[3]) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int fx = sxq[q] & 31, fy = syq[q] & 31;
        const int sx = sxq[q] >> 5, sy = syq[q] >> 5;
        const int wa = 32 - fx, wb = 32 - fy;
        const uint8_t *t00, *t01, *t10, *t11;
        if (!REFLECT) {
            t00 = stage + (lbase + sy * pitch + sx * 3); t01 = t00 + 3; t10 = t00 + pitch; t11 = t10 + 3;
        } else {
            const int x0 = mis_reflect1(sx, sw) * 3, x1 = mis_reflect1(sx + 1, sw) * 3;
            const int y0 = lbase + mis_reflect1(sy, sh) * pitch, y1 = lbase + mis_reflect1(sy + 1, sh) * pitch;
            t00 = stage + (y0 + x0); t01 = stage + (y0 + x1); t10 = stage + (y1 + x0); t11 = stage + (y1 + x1);
        }
        // sum(w_ij * p_ij) with w_ij = 32 * a_i * b_j factors exactly: two horizontal lerps, one vertical
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int h0 = t00[c] * wa + t01[c] * fx, h1 = t10[c] * wa + t11[c] * fx;
            p[q][c] = (h0 * wb + h1 * fy + 512) >> 10;
        }
    }
}

__global__ __launch_bounds__(64, 4) void warp_fused_kernel(WarpArgs a, const float* __restrict__ tab) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGE_BYTES];
    const int tx0 = blockIdx.x * FT_W, ty0 = blockIdx.y * FT_H;
    const int lane = threadIdx.x;
    const int gxr = tx0 + 2 * lane;
    const bool col_ok = gxr < a.dw, two = gxr + 1 < a.dw;
    const int gx = col_ok ? gxr : a.dw - 1;  // out-of-roi lanes shadow the last column (never stored)
    const float xhi = (float)a.sw - 0.5f, yhi = (float)a.sh - 0.5f;  // exact: sizes < 2^15
    const bool xe = ((a.sw - 1) & 1) == 0, ye = ((a.sh - 1) & 1) == 0;
    int sxq[8], syq[8];
    unsigned msk = 0;  // bit q: nearest source pixel of pixel q lies inside the frame
    {
        const int gx1 = two ? gx + 1 : gx;
        const float su0 = tab[gx], cu0 = tab[a.dw + gx], su1 = tab[gx1], cu1 = tab[a.dw + gx1];
#pragma unroll
        for (int i = 0; i < FT_H; i++) {
            const int gy = min(ty0 + i, a.dh - 1);  // wave-uniform: scalar loads; rows past the roi shadow the last one
            const float s_v = tab[2 * a.dw + gy], c_v = tab[2 * a.dw + a.dh + gy];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int q = 2 * i + k;
                float x, y;
                map_backward(a.m, k ? su1 : su0, k ? cu1 : cu0, s_v, c_v, &x, &y);
                // saturate_cast<short> of the integer part, the 5 fraction bits are kept below it
                const int xq = mis_round_sat_f(x * 32.f), yq = mis_round_sat_f(y * 32.f);
                sxq[q] = (mis_sat_short(xq >> 5) << 5) | (xq & 31);
                syq[q] = (mis_sat_short(yq >> 5) << 5) | (yq & 31);
                msk |= (round_in_range(x, xhi, xe) && round_in_range(y, yhi, ye)) ? (1u << q) : 0u;
            }
        }
    }
    // bounding box of the top-left taps; all taps interior <=> 0 <= min and max + 1 <= len - 1
    int xmin = sxq[0] >> 5, xmax = xmin, ymin = syq[0] >> 5, ymax = ymin;
#pragma unroll
    for (int q = 1; q < 8; q++) {
        xmin = min(xmin, sxq[q] >> 5); xmax = max(xmax, sxq[q] >> 5);
        ymin = min(ymin, syq[q] >> 5); ymax = max(ymax, syq[q] >> 5);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = min(xmin, __shfl_xor(xmin, o)); ymin = min(ymin, __shfl_xor(ymin, o));
        xmax = max(xmax, __shfl_xor(xmax, o)); ymax = max(ymax, __shfl_xor(ymax, o));
    }
    // wave-uniform classification
    const bool interior = xmin >= 0 && ymin >= 0 && xmax + 1 <= a.sw - 1 && ymax + 1 <= a.sh - 1;
    const bool foldable = xmin >= -a.sw && xmax + 1 < 2 * a.sw && ymin >= -a.sh && ymax + 1 < 2 * a.sh;
    int bx0 = xmin, bx1 = xmax + 1, by0 = ymin, by1 = ymax + 1;  // box of source columns / rows to stage
    if (!interior && foldable) {
        // fold the tap range once: the box of reflected coordinates of [lo, hi]
        auto fold = [](int lo, int hi, int len, int* o0, int* o1) {
            if (lo >= 0 && hi < len) { *o0 = lo; *o1 = hi; }
            else if (hi < 0) { *o0 = -hi - 1; *o1 = -lo - 1; }
            else if (lo >= len) { *o0 = 2 * len - 1 - hi; *o1 = 2 * len - 1 - lo; }
            else if (lo < 0) { *o0 = 0; *o1 = max(-lo - 1, min(hi, len - 1)); if (hi >= len) *o1 = len - 1; }
            else { *o0 = min(lo, 2 * len - 1 - hi); *o1 = len - 1; }
        };
        fold(xmin, xmax + 1, a.sw, &bx0, &bx1);
        fold(ymin, ymax + 1, a.sh, &by0, &by1);
    }
    // With a 4-byte-multiple stride every row has the same alignment shift, so LDS offsets are 32-bit
    // and affine in the source coordinates.
    const int shift = (bx0 * 3) & 3;
    const int pitch = ((bx1 - bx0 + 1) * 3 + shift + 3) & ~3;
    const int nrows = by1 - by0 + 1;
    constexpr int MAXR = 16;  // rows of the staged box; a row is at most 128 dwords (two per lane)
    const bool staged = (interior || foldable) && (a.sstride & 3) == 0 && nrows <= MAXR && pitch <= 512;
    if (staged) {
        // all loads of the box are issued before the first LDS write: one memory latency per tile
        const int dwords_per_row = pitch >> 2;
        const size_t total_bytes = (size_t)(a.sh - 1) * a.sstride + (size_t)a.sw * 3;  // last valid byte + 1
        const size_t gbase = (size_t)by0 * a.sstride + (size_t)(bx0 * 3 - shift) + 4 * (size_t)lane;
        const bool c0 = lane < dwords_per_row, c1 = lane + 64 < dwords_per_row;
        unsigned v0[MAXR], v1[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            const size_t g0 = gbase + (size_t)r * a.sstride;
            v0[r] = 0; v1[r] = 0;
            if (r < nrows) {
                if (c0) {
                    if (g0 + 4 <= total_bytes) v0[r] = *reinterpret_cast<const unsigned*>(a.src + g0);
                    else for (int k = 0; k < 4; k++) if (g0 + k < total_bytes) v0[r] |= (unsigned)a.src[g0 + k] << (8 * k);
                }
                if (c1) {
                    if (g0 + 256 + 4 <= total_bytes) v1[r] = *reinterpret_cast<const unsigned*>(a.src + g0 + 256);
                    else for (int k = 0; k < 4; k++) if (g0 + 256 + k < total_bytes) v1[r] |= (unsigned)a.src[g0 + 256 + k] << (8 * k);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            if (r < nrows) {
                if (c0) *reinterpret_cast<unsigned*>(stage + r * pitch + 4 * lane) = v0[r];
                if (c1) *reinterpret_cast<unsigned*>(stage + r * pitch + 4 * lane + 256) = v1[r];
            }
        }
    }
    __syncthreads();  // single-wave workgroup: an LDS fence
    if (!col_ok) return;
    const int lbase = shift - by0 * pitch - bx0 * 3;  // LDS byte offset of source pixel (0, 0)
    int p[8][3];
    if (staged && interior) sample8<false>(stage, lbase, pitch, a.sw, a.sh, sxq, syq, p);
    else if (staged) sample8<true>(stage, lbase, pitch, a.sw, a.sh, sxq, syq, p);
    else {
        // box too large for LDS or coordinates far outside the frame: gather from global memory
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int fx = sxq[q] & 31, fy = syq[q] & 31, sx = sxq[q] >> 5, sy = syq[q] >> 5;
            const int x0 = mis_reflect(sx, a.sw), x1 = mis_reflect(sx + 1, a.sw), y0 = mis_reflect(sy, a.sh), y1 = mis_reflect(sy + 1, a.sh);
            const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
            const uint8_t* r0 = a.src + (size_t)y0 * a.sstride;
            const uint8_t* r1 = a.src + (size_t)y1 * a.sstride;
#pragma unroll
            for (int c = 0; c < 3; c++)
                p[q][c] = (r0[x0 * 3 + c] * w00 + r0[x1 * 3 + c] * w01 + r1[x0 * 3 + c] * w10 + r1[x1 * 3 + c] * w11 + (1 << 14)) >> 15;
        }
    }
#pragma unroll
    for (int i = 0; i < FT_H; i++) {
        const int gy = ty0 + i;
        if (gy >= a.dh) break;
        uint8_t* drow = (uint8_t*)a.dst + (size_t)gy * a.dstride + (size_t)gx * 6;
        uint8_t* mrow = a.mask + (size_t)gy * a.mstride + gx;
        const unsigned m0 = (msk >> (2 * i) & 1) ? 255u : 0u, m1 = (msk >> (2 * i + 1) & 1) ? 255u : 0u;
        if (two) {
            uint3 w;
            w.x = (unsigned)p[2 * i][0] | ((unsigned)p[2 * i][1] << 16);
            w.y = (unsigned)p[2 * i][2] | ((unsigned)p[2 * i + 1][0] << 16);
            w.z = (unsigned)p[2 * i + 1][1] | ((unsigned)p[2 * i + 1][2] << 16);
            *reinterpret_cast<uint3*>(drow) = w;
            *reinterpret_cast<unsigned short*>(mrow) = (unsigned short)(m0 | (m1 << 8));
        } else {
            int16_t* d = reinterpret_cast<int16_t*>(drow);
            d[0] = (int16_t)p[2 * i][0]; d[1] = (int16_t)p[2 * i][1]; d[2] = (int16_t)p[2 * i][2];
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
    // separable trig tables live in the context's grow-only scratch
    const size_t tab_bytes = sizeof(float) * 2 * ((size_t)a.dw + a.dh);
    if (ctx->stage_bytes < tab_bytes) {
        if (ctx->stage) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
        MIS_HIP(ctx, hipMalloc(&ctx->stage, tab_bytes * 2 + 4096));
        ctx->stage_bytes = tab_bytes * 2 + 4096;
    }
    float* tab = (float*)ctx->stage;
    hipLaunchKernelGGL(warp_trig_kernel, dim3((a.dw + a.dh + 255) / 256), dim3(256), 0, ctx->stream, a, tab);
    dim3 grid((a.dw + FT_W - 1) / FT_W, (a.dh + FT_H - 1) / FT_H), block(64);
    hipLaunchKernelGGL(warp_fused_kernel, grid, block, 0, ctx->stream, a, (const float*)tab);
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dout)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_commit(ctx, dmask, &dm)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_release(ctx, &din)) != MIS_OK) return rc;
    if (tl) { tl->x = a.tlx; tl->y = a.tly; }
    return MIS_OK;
}
