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
#include <algorithm>
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

// SphericalProjector::mapForward (warpers_inl.hpp); the same float operations on the host and on the device
MIS_HD void map_forward(const float* m, float scale, float x, float y, float* u, float* v) {
    float x_ = (m[0] * x + m[1] * y) + m[2];
    float y_ = (m[3] * x + m[4] * y) + m[5];
    float z_ = (m[6] * x + m[7] * y) + m[8];
    *u = scale * mis_atan2f(x_, z_);
    float w = y_ / sqrtf((x_ * x_ + y_ * y_) + z_ * z_);
    *v = scale * (MIS_PI_F - mis_acosf(w == w ? w : 0));
}

// SphericalWarper::detectResultRoi, second half: the float extremes of the border projection -> integer roi with the pole tests
void roi_from_extremes(const Projector* p, int sw, int sh, float tl_uf, float tl_vf, float br_uf, float br_vf, int* tlx, int* tly, int* brx, int* bry) {
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

// SphericalWarper::detectResultRoi: border projection + pole tests; 2(W+H) points on the host (single-call entry points;
// a job's frames go through mis_warp_roi_batch: one small kernel for all of them).  Nothing is cached: every call pays it.
void detect_result_roi(const Projector* p, int sw, int sh, int* tlx, int* tly, int* brx, int* bry) {
    float tl_uf = FLT_MAX, tl_vf = FLT_MAX, br_uf = -FLT_MAX, br_vf = -FLT_MAX, u, v;
    auto upd = [&]() {
        if (u < tl_uf) tl_uf = u;
        if (v < tl_vf) tl_vf = v;
        if (u > br_uf) br_uf = u;
        if (v > br_vf) br_vf = v;
    };
    for (int x = 0; x < sw; ++x) {
        map_forward(p->r_kinv, p->scale, (float)x, 0, &u, &v); upd();
        map_forward(p->r_kinv, p->scale, (float)x, (float)(sh - 1), &u, &v); upd();
    }
    for (int y = 0; y < sh; ++y) {
        map_forward(p->r_kinv, p->scale, 0, (float)y, &u, &v); upd();
        map_forward(p->r_kinv, p->scale, (float)(sw - 1), (float)y, &u, &v); upd();
    }
    roi_from_extremes(p, sw, sh, tl_uf, tl_vf, br_uf, br_vf, tlx, tly, brx, bry);
}

void projector_and_roi(float scale, const float K[9], const float R[9], int w, int h, Projector* p, int* tlx, int* tly, int* brx, int* bry) {
    projector_set(p, scale, K, R);
    detect_result_roi(p, w, h, tlx, tly, brx, bry);
}

// The border walk of detectResultRoi for a whole job: workgroup f projects the 2(W+H) border pixels of frame f and reduces
// the four extremes (min / max of finite floats: any order gives the host loop's result).  Jobs and results live in pinned,
// device-visible host memory: one launch + one stream synchronisation for all frames of a panorama.
struct RoiJob {
    float r_kinv[9], scale;
    int sw, sh;
    float ext[4];   // out: min u, min v, max u, max v
};
__global__ __launch_bounds__(256) void warp_roi_kernel(RoiJob* jobs) {
    RoiJob* j = jobs + blockIdx.x;
    __shared__ float red[4][4];
    float m[9];
    for (int i = 0; i < 9; i++) m[i] = j->r_kinv[i];
    const float scale = j->scale;
    const int sw = j->sw, sh = j->sh;
    float lo_u = FLT_MAX, lo_v = FLT_MAX, hi_u = -FLT_MAX, hi_v = -FLT_MAX;
    for (int i = threadIdx.x; i < 2 * (sw + sh); i += 256) {
        float x, y, u, v;
        if (i < 2 * sw) { x = (float)(i >> 1); y = (i & 1) ? (float)(sh - 1) : 0.f; }
        else { const int k = i - 2 * sw; y = (float)(k >> 1); x = (k & 1) ? (float)(sw - 1) : 0.f; }
        map_forward(m, scale, x, y, &u, &v);
        lo_u = fminf(lo_u, u); lo_v = fminf(lo_v, v); hi_u = fmaxf(hi_u, u); hi_v = fmaxf(hi_v, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo_u = fminf(lo_u, __shfl_xor(lo_u, o)); lo_v = fminf(lo_v, __shfl_xor(lo_v, o));
        hi_u = fmaxf(hi_u, __shfl_xor(hi_u, o)); hi_v = fmaxf(hi_v, __shfl_xor(hi_v, o));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = lo_u; red[wave][1] = lo_v; red[wave][2] = hi_u; red[wave][3] = hi_v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        j->ext[0] = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
        j->ext[1] = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1]));
        j->ext[2] = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2]));
        j->ext[3] = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
    }
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

// Fused compose-scale warp: 8UC3 source -> 16SC3 image + 8U mask (K10, the roofline kernel).
//
// The kernel is VALU bound (PMC: profiles/), so the design minimises vector instructions per pixel:
//  * sin/cos of the column angle u = (tlx + x) / scale and of the row angle pi - v are separable: a tiny
//    pre-kernel tabulates {sin u, cos u} per column and {sin v, m1 cos v, m4 cos v, m7 cos v} per row
//    (the row products are the same f32 products the per-pixel formula forms): no trigonometry here.
//  * One wave per workgroup (no workgroup barriers) owns a 32 x 16 output tile -- nearly square, because
//    off the optical axis a tile's source footprint is a slanted strip (dy/du = tan v sin u / cos^2 u:
//    ~29 source rows across 128 columns at a 4K frame's corner) and the staged box must stay small.
//    A lane owns 2 adjacent columns x 4 rows; the two columns are the halves of packed-f32 registers.
//  * x / z and y / z share one reciprocal refinement: the divide is the exact FMA sequence of an IEEE f32
//    division (rcp, Newton step, quotient, two residual corrections) without its scaling / fix-up
//    instructions, valid while z is in [2^-30, 2^30] and |x|, |y| < 2^15; a wave-uniform guard sends
//    everything else down the generic path (IEEE division, x86 cvRound overflow, saturate_cast<short>).
//  * The wave all-reduces the bounding box of its taps (DPP), stages exactly that box of the source in
//    LDS with direct global->LDS loads (16-byte pieces, several box rows per instruction, no staging
//    registers), and gathers the 12 taps per pixel from LDS.  Tiles whose taps need BORDER_REFLECT fold
//    the box once; boxes that do not fit take a global-memory gather (cold).
#ifndef WV_LROWS
#define WV_LROWS 4
#endif
#ifndef WV_V3
#define WV_V3 1      // 1: the pipelined strip kernels of round 3; 0: round 2's one-tile-per-wave kernels (kept for A/B runs)
#endif
#ifndef WV_ABL
#define WV_ABL 0     // diagnostics builds only (tools/warp_variants.sh): phases switched off one at a time
#endif
constexpr int LROWS = WV_LROWS;          // rows per lane (4 lane rows per tile)
constexpr int LPX = 2 * LROWS;            // pixels per lane
constexpr int FT_W = 32, FT_H = 4 * LROWS;
constexpr int STAGE_BYTES = 1280 * LROWS;  // LROWS = 4: 5 KB, 32 waves per CU fit the 160 KB of LDS
typedef float v2f __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ int trig_cols(int dw) { return (dw + 3) & ~1; }  // >= dw + 2, even: 16-byte aligned row table
static size_t trig_table_floats(int dw, int dh) { return 2 * (size_t)trig_cols(dw) + 4 * (size_t)dh; }

__global__ __launch_bounds__(256) void warp_trig_kernel(WarpArgs a, float* tab) {
    const int i = blockIdx.x * 256 + threadIdx.x, ncol = trig_cols(a.dw);
    if (i < ncol) {
        float u = (float)(a.tlx + i) / a.scale;
        mis_sincosf(u, &tab[2 * i], &tab[2 * i + 1]);
    } else if (i < ncol + a.dh) {
        const int r = i - ncol;
        float v = (float)(a.tly + r) / a.scale, s, c;
        mis_sincosf(MIS_PI_F - v, &s, &c);
        float* t = tab + 2 * ncol + 4 * r;
        t[0] = s; t[1] = a.m[1] * c; t[2] = a.m[4] * c; t[3] = a.m[7] * c;
    }
}

// cvRound(v) in [0, len): round-half-even maps [-0.5, len - 0.5) into range, and the upper end point
// len - 0.5 too when len - 1 is even (ties go to the even neighbour)
__device__ __forceinline__ bool round_in_range(float v, float hi, bool hi_even) { return (v >= -0.5f) & ((v < hi) | (hi_even & (v == hi))); }

// wave-wide min / max -> uniform result: xor-1, xor-2, half-mirror and mirror inside the 16-lane rows as
// single DPP instructions (s_nop 1: two wait states between a VALU write and its DPP read), then 4 readlanes
#define MIS_DPP_STEP(OP, CTRL) asm volatile("s_nop 1\n\tv_" OP "_i32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(v))
__device__ __forceinline__ int wave_min_i32(int v) {
    MIS_DPP_STEP("min", "quad_perm:[1,0,3,2]"); MIS_DPP_STEP("min", "quad_perm:[2,3,0,1]");
    MIS_DPP_STEP("min", "row_half_mirror"); MIS_DPP_STEP("min", "row_mirror");
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v) {
    MIS_DPP_STEP("max", "quad_perm:[1,0,3,2]"); MIS_DPP_STEP("max", "quad_perm:[2,3,0,1]");
    MIS_DPP_STEP("max", "row_half_mirror"); MIS_DPP_STEP("max", "row_mirror");
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#undef MIS_DPP_STEP

// numerators / denominator of SphericalProjector::mapBackward for a lane's two columns of one row
__device__ __forceinline__ void map_terms(const float* m, v2f su, v2f cu, float4 rt, v2f* xx, v2f* yy, v2f* zz) {
    const v2f x_ = su * rt.x, z_ = cu * rt.x;   // sinv * sinu, sinv * cosu;  y_ = cosv is folded into rt.y/z/w
    *xx = (x_ * m[0] + rt.y) + z_ * m[2];
    *yy = (x_ * m[3] + rt.z) + z_ * m[5];
    *zz = (x_ * m[6] + rt.w) + z_ * m[8];
}

// nx / d and ny / d for both columns: the FMA chain of an IEEE-correct f32 division (see the header comment)
__device__ __forceinline__ void div2_shared(v2f nx, v2f ny, v2f d, v2f* qx, v2f* qy) {
    const v2f r0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const v2f nd = -d, one = {1.f, 1.f};
    const v2f e = __builtin_elementwise_fma(nd, r0, one);
    const v2f r = __builtin_elementwise_fma(e, r0, r0);
    v2f q = nx * r;
    v2f t = __builtin_elementwise_fma(nd, q, nx);
    q = __builtin_elementwise_fma(t, r, q);
    t = __builtin_elementwise_fma(nd, q, nx);
    *qx = __builtin_elementwise_fma(t, r, q);
    q = ny * r;
    t = __builtin_elementwise_fma(nd, q, ny);
    q = __builtin_elementwise_fma(t, r, q);
    t = __builtin_elementwise_fma(nd, q, ny);
    *qy = __builtin_elementwise_fma(t, r, q);
}

// 24-bit multiply-add in one instruction (the compiler prefers mul + add3 chains: 6 instead of 4 per channel)
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// bilinear sample of one pixel from the staged box (the Q15 weights of remap factor exactly as
// 32 * a_i * b_j, so (sum(w_ij p_ij) + 2^14) >> 15 == (sum(a_i b_j p_ij) + 512) >> 10)
template <bool REFLECT>
__device__ __forceinline__ void sample1(const uint8_t* stage, int lbase, int pitch, int sw, int sh, int xq, int yq, int* p) {
    const int fx = xq & 31, fy = yq & 31, sx = xq >> 5, sy = yq >> 5;
    const int wa = 32 - fx, wb = 32 - fy;
    const uint8_t *t00, *t01, *t10, *t11;
    if (!REFLECT) {
        t00 = stage + (lbase + __mul24(sy, pitch) + sx * 3); t01 = t00 + 3; t10 = t00 + pitch; t11 = t10 + 3;
    } else {
        const int x0 = mis_reflect1(sx, sw) * 3, x1 = mis_reflect1(sx + 1, sw) * 3;
        const int y0 = lbase + __mul24(mis_reflect1(sy, sh), pitch), y1 = lbase + __mul24(mis_reflect1(sy + 1, sh), pitch);
        t00 = stage + (y0 + x0); t01 = stage + (y0 + x1); t10 = stage + (y1 + x0); t11 = stage + (y1 + x1);
    }
    const int w00 = wa * wb, w01 = fx * wb, w10 = wa * fy, w11 = fx * fy;
#pragma unroll
    for (int c = 0; c < 3; c++) p[c] = (int)(mad24(t11[c], w11, mad24(t10[c], w10, mad24(t01[c], w01, mad24(t00[c], w00, 512u)))) >> 10);
}

__device__ __forceinline__ void sample1_global(const WarpArgs& a, int xq, int yq, int* p) {
    const int fx = xq & 31, fy = yq & 31, sx = xq >> 5, sy = yq >> 5;
    const int x0 = mis_reflect(sx, a.sw), x1 = mis_reflect(sx + 1, a.sw), y0 = mis_reflect(sy, a.sh), y1 = mis_reflect(sy + 1, a.sh);
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const uint8_t* r0 = a.src + (size_t)y0 * a.sstride;
    const uint8_t* r1 = a.src + (size_t)y1 * a.sstride;
#pragma unroll
    for (int c = 0; c < 3; c++)
        p[c] = (r0[x0 * 3 + c] * w00 + r0[x1 * 3 + c] * w01 + r1[x0 * 3 + c] * w10 + r1[x1 * 3 + c] * w11 + (1 << 14)) >> 15;
}

// global -> LDS copy of `nrows` box rows of `pitch` bytes (PIECE bytes per lane, 64 / (pitch / PIECE) rows per
// instruction); the LDS image is row-major with that pitch.  Wave-uniform arguments except `lane`.
template <int PIECE>
__device__ __forceinline__ void stage_box(const uint8_t* src, size_t sstride, uint8_t* stage, int pitch, int nrows, int lane) {
    const int lpr = pitch / PIECE;                 // lanes per row (<= 64)
    const int k = 64 / lpr;                        // rows per instruction
    const int lr = (lane * (65536 / lpr + 1)) >> 16, lc = lane - lr * lpr;   // lane / lpr, lane % lpr (exact for lane < 64)
    const uint8_t* g = src + (size_t)lr * sstride + (size_t)lc * PIECE;
    const int live = lr < k ? nrows - lr : 0;      // this lane copies rows r0 + lr while r0 < live
    for (int r0 = 0; r0 < nrows; r0 += k) {
        if (r0 < live) {
            const __attribute__((address_space(1))) void* gp = (const __attribute__((address_space(1))) void*)(g + (size_t)r0 * sstride);
            __attribute__((address_space(3))) void* lp = (__attribute__((address_space(3))) void*)(stage + r0 * pitch);
            if constexpr (PIECE == 16) __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
        }
    }
}

#ifdef WV_STAMPS
__device__ unsigned long long g_warp_stamps[16384 * 8];
#endif
#ifdef MIS_WARP_STATS
__device__ unsigned g_warp_stats[8];   // tiles: interior, folded, global gather, generic-map
#define WSTAT(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_warp_stats[i], 1u); } while (0)
#else
#define WSTAT(i)
#endif
// Everything a tile needs between its map phase and its sample phase (one lane's share in registers).
struct TileState {
    int xq[LPX], yq[LPX];   // q = 2 * row + column: 5 fraction bits below the (short-range) integer coordinate
    unsigned msk;       // bit q: the nearest source pixel of pixel q lies inside the frame
    int gx, gy0;        // this lane's first column / row
    bool col_ok, two;
    // wave-uniform
    bool interior, staged, wide;
    int pitch, nrows, lbase;
    size_t gbase;
};

// Phase 1 of a tile: backward map of the lane's 2 x 4 pixels, box reduction, classification.
// The separable trig terms of a lane's 2 columns x 4 rows (prefetched one pipeline stage ahead of the map)
struct TileTrig {
    float4 cs;      // {sin u0, cos u0, sin u1, cos u1}
    float4 rt[LROWS];   // per row {sin v, m1 cos v, m4 cos v, m7 cos v}
};
__device__ __forceinline__ void tile_trig(const WarpArgs& a, const float* __restrict__ tab, int tx0, int ty0, int lane, TileTrig& g) {
    const int gxr = tx0 + 2 * (lane & 15), gy0 = ty0 + LROWS * (lane >> 4);
    const int gx = gxr < a.dw ? gxr : ((a.dw - 1) & ~1);
    g.cs = *reinterpret_cast<const float4*>(tab + 2 * gx);
    const float4* rowtab = reinterpret_cast<const float4*>(tab + 2 * trig_cols(a.dw));
#pragma unroll
    for (int i = 0; i < LROWS; i++) g.rt[i] = rowtab[min(gy0 + i, a.dh - 1)];  // rows past the roi shadow the last one
}

__device__ __forceinline__ void tile_map(const WarpArgs& a, const TileTrig& g, int tx0, int ty0, int lane, TileState& t) {
    const int lx = lane & 15, ly = lane >> 4;
    const int gxr = tx0 + 2 * lx, gy0 = ty0 + LROWS * ly;
    t.col_ok = gxr < a.dw; t.two = gxr + 1 < a.dw;
    const int gx = t.col_ok ? gxr : ((a.dw - 1) & ~1);  // out-of-roi lanes shadow the last column pair (never stored)
    t.gx = gx; t.gy0 = gy0;
    const float xhi = (float)a.sw - 0.5f, yhi = (float)a.sh - 0.5f;  // exact: sizes < 2^15
    const bool xe = ((a.sw - 1) & 1) == 0, ye = ((a.sh - 1) & 1) == 0;
    // v < hi || (even && v == hi)  <=>  v < hi2 with hi2 the next float above hi when the end point is included
    const float xhi2 = xe ? __uint_as_float(__float_as_uint(xhi) + 1u) : xhi, yhi2 = ye ? __uint_as_float(__float_as_uint(yhi) + 1u) : yhi;
    const v2f su = {g.cs.x, g.cs.z}, cu = {g.cs.y, g.cs.w};
    int* xq = t.xq; int* yq = t.yq;
    unsigned msk = (1u << LPX) - 1u;
    // ---- fast map: assumes z >= 2^-30 everywhere in the tile (checked below, wave-uniform) ----
    float zlo = 3.0e38f;
    const v2f k32 = {32.f, 32.f}, magic = {12582912.f, 12582912.f};   // 1.5 * 2^23: float add rounds to nearest-even integer
#pragma unroll
    for (int i = 0; i < LROWS; i++) {
        v2f xx, yy, zz, qx, qy;
        map_terms(a.m, su, cu, g.rt[i], &xx, &yy, &zz);
#if WV_ABL == 4   // diagnostics: no division (approximate coordinates)
        { const v2f r0 = {__builtin_amdgcn_rcpf(zz.x), __builtin_amdgcn_rcpf(zz.y)}; qx = xx * r0; qy = yy * r0; }
#elif WV_ABL == 5  // diagnostics: no map at all
        qx = v2f{(float)gx + 100.25f, (float)gx + 101.25f} + 0.f * su; qy = v2f{(float)(gy0 + i) + 100.5f, (float)(gy0 + i) + 100.5f} + 0.f * g.rt[i].x; zz = v2f{1.f, 1.f};
#else
        div2_shared(xx, yy, zz, &qx, &qy);
#endif
        zlo = fminf(fminf(zlo, zz.x), zz.y);
        // cvRound(32 x) for |32 x| < 2^22; anything larger lands outside +-2^22 as well and is caught by the box check
        const v2f tx = qx * k32 + magic, ty = qy * k32 + magic;
        xq[2 * i] = (int)__float_as_uint(tx.x) - 0x4B400000; xq[2 * i + 1] = (int)__float_as_uint(tx.y) - 0x4B400000;
        yq[2 * i] = (int)__float_as_uint(ty.x) - 0x4B400000; yq[2 * i + 1] = (int)__float_as_uint(ty.y) - 0x4B400000;
    }
    // bounding box of the top-left taps (floor division by 32 is monotonic: reduce the packed values)
    int xmin, xmax, ymin, ymax;
    auto reduce_box = [&]() {
        xmin = xq[0]; xmax = xq[0]; ymin = yq[0]; ymax = yq[0];
#pragma unroll
        for (int q = 1; q < LPX; q++) {
            xmin = min(xmin, xq[q]); xmax = max(xmax, xq[q]);
            ymin = min(ymin, yq[q]); ymax = max(ymax, yq[q]);
        }
        xmin = wave_min_i32(xmin); xmax = wave_max_i32(xmax);
        ymin = wave_min_i32(ymin); ymax = wave_max_i32(ymax);
    };
    reduce_box();
    // |z| <= |m6| + |m7| + |m8| bounds z from above (uniform); z from below per lane; coordinates through the box
    const bool good = (fabsf(a.m[6]) + fabsf(a.m[7]) + fabsf(a.m[8]) <= 1048576.f) && !__any(!(zlo >= 9.31322574615478515625e-10f)) &&
                      xmin > -(1 << 20) && ymin > -(1 << 20) && xmax < (1 << 20) && ymax < (1 << 20);
    const bool all_inside = good && xmin >= 0 && ymin >= 0 && (xmax >> 5) + 1 <= a.sw - 1 && (ymax >> 5) + 1 <= a.sh - 1;
    if (!good) {
        // generic map (cold): IEEE division, x86 cvRound overflow semantics, saturate_cast<short> of the integer part
        WSTAT(3);
        msk = 0;
#pragma unroll
        for (int i = 0; i < LROWS; i++) {
            v2f xx, yy, zz;
            map_terms(a.m, su, cu, g.rt[i], &xx, &yy, &zz);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const bool front = zz[k] > 0;
                const float x = front ? xx[k] / zz[k] : -1.f, y = front ? yy[k] / zz[k] : -1.f;
                const int xr = mis_round_sat_f(x * 32.f), yr = mis_round_sat_f(y * 32.f);
                xq[2 * i + k] = (mis_sat_short(xr >> 5) << 5) | (xr & 31);
                yq[2 * i + k] = (mis_sat_short(yr >> 5) << 5) | (yr & 31);
                msk |= (unsigned)(round_in_range(x, xhi, xe) & round_in_range(y, yhi, ye)) << (2 * i + k);
            }
        }
        reduce_box();
    } else if (!all_inside) {
        // some taps leave the frame: the mask (nearest source pixel inside?) needs the unquantised coordinates again
        msk = 0;
#pragma unroll
        for (int i = 0; i < LROWS; i++) {
            v2f xx, yy, zz, qx, qy;
            map_terms(a.m, su, cu, g.rt[i], &xx, &yy, &zz);
            div2_shared(xx, yy, zz, &qx, &qy);
#pragma unroll
            for (int k = 0; k < 2; k++)
                msk |= (unsigned)((qx[k] >= -0.5f) & (qx[k] < xhi2) & (qy[k] >= -0.5f) & (qy[k] < yhi2)) << (2 * i + k);
        }
    }
    t.msk = msk;
    xmin >>= 5; xmax >>= 5; ymin >>= 5; ymax >>= 5;
    // wave-uniform classification; all taps interior <=> 0 <= min and max + 1 <= len - 1
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
    // Rows are copied in whole 16-byte (or, for frames whose base / stride is not 16-byte aligned, 4-byte)
    // pieces starting at an aligned address: every row has the same alignment shift, so LDS offsets are
    // 32-bit and affine in the source coordinates.  The last piece of the last row must lie inside the frame.
    const bool wide = (((size_t)a.src | a.sstride) & 15) == 0;
    const int amask = wide ? 15 : 3;
    const int shift = (bx0 * 3) & amask;
    const int pitch = ((bx1 - bx0 + 1) * 3 + shift + amask) & ~amask;
    const int nrows = by1 - by0 + 1;
    const size_t total_bytes = (size_t)(a.sh - 1) * a.sstride + (size_t)a.sw * 3;  // last valid byte + 1
    const size_t gbase = (size_t)by0 * a.sstride + (size_t)(bx0 * 3 - shift);
    t.interior = interior; t.wide = wide; t.pitch = pitch; t.nrows = nrows; t.gbase = gbase;
    t.staged = (interior || foldable) && (((size_t)a.src | a.sstride) & 3) == 0 && nrows * pitch <= STAGE_BYTES &&
               pitch <= (wide ? 1024 : 256) && gbase + (size_t)(nrows - 1) * a.sstride + pitch <= total_bytes;
    t.lbase = shift - by0 * pitch - bx0 * 3;  // LDS byte offset of source pixel (0, 0)
    WSTAT(t.staged ? (interior ? 0 : 1) : 2);
}

// Phase 2: issue the global -> LDS copies of the tile's source box (asynchronous; completion = vmcnt)
__device__ __forceinline__ void tile_stage(const WarpArgs& a, const TileState& t, uint8_t* stage, int lane) {
    if (!t.staged) return;
#if WV_ABL == 3   // diagnostics: no global -> LDS copies
    return;
#endif
    if (t.wide) stage_box<16>(a.src + t.gbase, a.sstride, stage, t.pitch, t.nrows, lane);
    else stage_box<4>(a.src + t.gbase, a.sstride, stage, t.pitch, t.nrows, lane);
}

// Phase 3: bilinear gather from the staged box and the 16SC3 / mask stores
#ifndef WV_TILE_LDS_STORE
#define WV_TILE_LDS_STORE 0     // 1: whole tiles leave through an LDS transpose as 16-byte pieces. Measured SLOWER here (25.6 vs 24.4 us per launch,
                                // gpurun_out/r3_v3_var12.txt): this kernel is bound by its LDS (2.1-fold bank conflicts of the 2 x 4 lane blocks); the strip kernel is the form where it pays
#endif
__device__ __forceinline__ void tile_sample_store(const WarpArgs& a, const TileState& t, uint8_t* stage, int tx0, int ty0, int lane) {
    uint3 w[LROWS];
    unsigned mm[LROWS];
#pragma unroll
    for (int i = 0; i < LROWS; i++) {
        int p0[3], p1[3];
#if WV_ABL == 2   // diagnostics: no LDS gather, no bilinear arithmetic
        if (true) {
            p0[0] = t.xq[2 * i] & 255; p0[1] = t.yq[2 * i] & 255; p0[2] = 7; p1[0] = t.xq[2 * i + 1] & 255; p1[1] = t.yq[2 * i + 1] & 255; p1[2] = 9;
        } else
#endif
        if (t.staged && t.interior) {
            sample1<false>(stage, t.lbase, t.pitch, a.sw, a.sh, t.xq[2 * i], t.yq[2 * i], p0);
            sample1<false>(stage, t.lbase, t.pitch, a.sw, a.sh, t.xq[2 * i + 1], t.yq[2 * i + 1], p1);
        } else if (t.staged) {
            sample1<true>(stage, t.lbase, t.pitch, a.sw, a.sh, t.xq[2 * i], t.yq[2 * i], p0);
            sample1<true>(stage, t.lbase, t.pitch, a.sw, a.sh, t.xq[2 * i + 1], t.yq[2 * i + 1], p1);
        } else {
            // box too large for LDS or coordinates far outside the frame: gather from global memory
            sample1_global(a, t.xq[2 * i], t.yq[2 * i], p0);
            sample1_global(a, t.xq[2 * i + 1], t.yq[2 * i + 1], p1);
        }
        w[i].x = (unsigned)p0[0] | ((unsigned)p0[1] << 16);
        w[i].y = (unsigned)p0[2] | ((unsigned)p1[0] << 16);
        w[i].z = (unsigned)p1[1] | ((unsigned)p1[2] << 16);
        mm[i] = ((t.msk >> (2 * i) & 1) ? 255u : 0u) | ((t.msk >> (2 * i + 1) & 1) ? 0xff00u : 0u);
    }
#if WV_TILE_LDS_STORE && WV_LROWS == 4 && WV_ABL != 1
    // whole tile inside the roi, 16-byte aligned outputs (wave-uniform): the 16 rows of 192 bytes are written to the stage (the
    // box is spent: every tap of this wave has been read), read back as 192 pieces of 16 bytes -- piece p = row * 12 + column
    // piece, lane l takes p = l, l + 64, l + 128 -- and stored with global_store_dwordx4: 12 lanes cover a row's 192 bytes
    const bool full = tx0 + FT_W <= a.dw && ty0 + FT_H <= a.dh && ((((size_t)a.dst | a.dstride) & 15) == 0) && ((((size_t)a.mask | a.mstride) & 7) == 0);
    if (full) {
        const int lx = lane & 15, ly = lane >> 4;
        volatile unsigned* so = reinterpret_cast<volatile unsigned*>(stage + (LROWS * ly) * (FT_W * 6) + lx * 12);
        volatile unsigned short* sm = reinterpret_cast<volatile unsigned short*>(stage + FT_W * FT_H * 6 + (LROWS * ly) * FT_W + lx * 2);
#pragma unroll
        for (int i = 0; i < LROWS; i++) {
            so[i * (FT_W * 6 / 4)] = w[i].x; so[i * (FT_W * 6 / 4) + 1] = w[i].y; so[i * (FT_W * 6 / 4) + 2] = w[i].z;
            sm[i * (FT_W / 2)] = (unsigned short)mm[i];
        }
        uint8_t* dbase = (uint8_t*)a.dst + (size_t)ty0 * a.dstride + (size_t)tx0 * 6;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const int p = kk * 64 + lane, row = p / 12, cb = (p - row * 12) * 16;
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 v = *reinterpret_cast<const u4*>(stage + p * 16);
            *reinterpret_cast<u4*>(dbase + (size_t)row * a.dstride + cb) = v;
        }
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 mv = *reinterpret_cast<const u2*>(stage + FT_W * FT_H * 6 + lane * 8);
        *reinterpret_cast<u2*>(a.mask + (size_t)(ty0 + (lane >> 2)) * a.mstride + tx0 + (lane & 3) * 8) = mv;
        return;
    }
#endif
    if (!t.col_ok) return;
    uint8_t* drow = (uint8_t*)a.dst + (size_t)t.gy0 * a.dstride + (size_t)t.gx * 6;
    uint8_t* mrow = a.mask + (size_t)t.gy0 * a.mstride + t.gx;
#pragma unroll
    for (int i = 0; i < LROWS; i++, drow += a.dstride, mrow += a.mstride) {
        if (t.gy0 + i >= a.dh) break;
        if (t.two) {
#if WV_ABL == 1   // diagnostics: everything but the global stores
            asm volatile("" :: "v"(w[i].x), "v"(w[i].y), "v"(w[i].z), "v"(mm[i]), "v"(drow), "v"(mrow));
#else
            *reinterpret_cast<uint3*>(drow) = w[i];
            *reinterpret_cast<unsigned short*>(mrow) = (unsigned short)mm[i];
#endif
        } else {
            int16_t* d = reinterpret_cast<int16_t*>(drow);
            d[0] = (int16_t)(w[i].x & 0xffff); d[1] = (int16_t)(w[i].x >> 16); d[2] = (int16_t)(w[i].y & 0xffff);
            mrow[0] = (uint8_t)mm[i];
        }
    }
}

// One tile per wave, TILE_WAVES independent waves per workgroup (no barriers; fewer, larger workgroups to dispatch).
#ifndef WV_TILE_WAVES
#define WV_TILE_WAVES 2
#endif
constexpr int TILE_WAVES = WV_TILE_WAVES;
#ifndef WV_WAVES_MIN
#define WV_WAVES_MIN 4     // occupancy floor the register allocator works to (6 and 8 measured in the batched grid: no gain)
#endif
#ifndef WV_ORDER
#define WV_ORDER 2
#endif
#ifndef WV_CH_W
#define WV_CH_W 4
#endif
#ifndef WV_CH_H
#define WV_CH_H 4
#endif
constexpr int CH_W = WV_CH_W, CH_H = WV_CH_H;   // chunk of tiles owned by one XCD (CH_W a multiple of TILE_WAVES)
static_assert(CH_W % TILE_WAVES == 0, "chunk width must be a whole number of workgroups");
// bid / nwg: the workgroup's index in its frame's grid and the size of that grid
__device__ __forceinline__ void warp_fused_body(const WarpArgs& a, const float* __restrict__ tab, int ntiles, int bid, int nwg_frame, uint8_t (*stage_all)[STAGE_BYTES]) {
    const int ntx = (a.dw + FT_W - 1) / FT_W, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* stage = stage_all[wave];
    // Tile order.  Workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own L2.  The tile grid is cut into
    // chunks of CH_W x CH_H tiles (CH_W / TILE_WAVES x CH_H workgroups); chunk k belongs to XCD k % 8 and an XCD walks its chunks in
    // raster order: neighbouring tiles, which share source rows and the 16-byte pieces at their box edges, hit the same L2, and
    // every XCD gets the same mix of cheap interior tiles and expensive border tiles (a contiguous eighth per XCD left the two
    // XCDs that own the top and bottom bands 7 % more work: they finished 2.7 us after the others, tools/warp_stamps.py).
    const int nty = (a.dh + FT_H - 1) / FT_H;
#if WV_ORDER == 0
    const int nwg = nwg_frame, xcd = bid & 7;
    const int wg = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (bid >> 3);
    const int tile = wg * TILE_WAVES + wave;
    if (tile >= ntiles) return;
    const int tx = tile % ntx, ty = tile / ntx;
#elif WV_ORDER == 1
    const int tile = bid * TILE_WAVES + wave;
    if (tile >= ntiles) return;
    const int tx = tile % ntx, ty = tile / ntx;
    (void)nwg_frame;
#else
    constexpr int WG_X = CH_W / TILE_WAVES, WG_PER_CHUNK = WG_X * CH_H;
    const int nchx = (ntx + CH_W - 1) / CH_W;
    const int i = bid >> 3;
    const int chunk = (i / WG_PER_CHUNK) * 8 + (bid & 7), within = i % WG_PER_CHUNK;
    (void)nwg_frame;
    const int chy = chunk / nchx, chx = chunk - chy * nchx;
    const int tx = chx * CH_W + (within % WG_X) * TILE_WAVES + wave, ty = chy * CH_H + within / WG_X;
    if (tx >= ntx || ty >= nty) return;
    (void)ntiles;
#endif
    const int tx0 = tx * FT_W, ty0 = ty * FT_H;
    TileState cur;
    TileTrig trig;
#ifdef WV_STAMPS   // diagnostics build (tools/warp_stamps.py): per-wave phase time stamps, shader clock and wall clock
    const unsigned long long r0 = wall_clock64(), s0 = __builtin_readcyclecounter();
    tile_trig(a, tab, tx0, ty0, lane, trig);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long s1 = __builtin_readcyclecounter();
    tile_map(a, trig, tx0, ty0, lane, cur);
    const unsigned long long s2 = __builtin_readcyclecounter();
    tile_stage(a, cur, stage, lane);
    const unsigned long long s3 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const unsigned long long s4 = __builtin_readcyclecounter();
    tile_sample_store(a, cur, stage, tx0, ty0, lane);
    const unsigned long long s5 = __builtin_readcyclecounter();
    if (lane == 0 && ty * ntx + tx < 16384) {
        unsigned long long* o = g_warp_stamps + 8 * (ty * ntx + tx);
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4; o[5] = s5; o[6] = r0; o[7] = wall_clock64();
    }
#else
    tile_trig(a, tab, tx0, ty0, lane, trig);
    tile_map(a, trig, tx0, ty0, lane, cur);
    tile_stage(a, cur, stage, lane);                       // asynchronous global -> LDS copies ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ... have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    tile_sample_store(a, cur, stage, tx0, ty0, lane);
#endif
}
__global__ __launch_bounds__(64 * TILE_WAVES) __attribute__((amdgpu_waves_per_eu(WV_WAVES_MIN, 8))) void warp_fused_kernel(WarpArgs a, const float* __restrict__ tab, int ntiles) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[TILE_WAVES][STAGE_BYTES];
    warp_fused_body(a, tab, ntiles, (int)blockIdx.x, (int)gridDim.x, stage_all);
}
// The frames of a compositing loop in ONE grid (y = frame): a 4K frame is 2.8 generations of resident workgroups, so a launch of
// its own spends a quarter of its time filling and draining the device (tools/warp_stamps.py: the last workgroups start at 19 us of
// 24); in a common grid the next frame's tiles take the slots the previous frame's stragglers free.
constexpr int WB_MAX = 16;
struct WarpBatch {
    WarpArgs a[WB_MAX];
    const float* tab[WB_MAX];
    int ntiles[WB_MAX], nwg[WB_MAX];
};
#if !WV_V3      // round 2's batched tile kernel: only in the A / B build without the strip kernels
__global__ __launch_bounds__(64 * TILE_WAVES) __attribute__((amdgpu_waves_per_eu(WV_WAVES_MIN, 8))) void warp_fused_batch_kernel(WarpBatch b) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[TILE_WAVES][STAGE_BYTES];
    const int f = blockIdx.y;
    if ((int)blockIdx.x >= b.nwg[f]) return;      // the grid covers the largest frame of the batch
    const WarpArgs a = b.a[f];
    warp_fused_body(a, b.tab[f], b.ntiles[f], (int)blockIdx.x, b.nwg[f], stage_all);
}
#endif
__global__ __launch_bounds__(256) void warp_trig_batch_kernel(WarpBatch b) {
    const int f = blockIdx.y;
    const WarpArgs& a = b.a[f];
    float* tab = const_cast<float*>(b.tab[f]);
    const int i = blockIdx.x * 256 + threadIdx.x, ncol = trig_cols(a.dw);
    if (i < ncol) {
        float u = (float)(a.tlx + i) / a.scale;
        mis_sincosf(u, &tab[2 * i], &tab[2 * i + 1]);
    } else if (i < ncol + a.dh) {
        const int r = i - ncol;
        float v = (float)(a.tly + r) / a.scale, s, c;
        mis_sincosf(MIS_PI_F - v, &s, &c);
        float* t = tab + 2 * ncol + 4 * r;
        t[0] = s; t[1] = a.m[1] * c; t[2] = a.m[4] * c; t[3] = a.m[7] * c;
    }
}


// =====================================================================================================================
// K10, round 3: pipelined strips (warp_strip_kernel / warp_strip_batch_kernel).
//
// What the ablations of round 2's kernel said (gpurun_out/r3_abl1.txt; 23.9 us per frame in the batched grid on that box:
// without its stores 17.8, without its global -> LDS copies 16.0, without the division 22.3, without the gather 20.2): the
// one-tile-per-wave body is bound by the latency of its memory operations and by the shape of its stores, not by arithmetic.
//  * A wave owns a vertical STRIP of nt tiles of 64 x V3_TH pixels and runs them as a software pipeline: the backward map of
//    tile k + 1 and the global -> LDS copies of its source box are issued BEFORE tile k is gathered, so a copy's latency is
//    covered by the wave's own arithmetic while the stores of tile k - 1 drain.  The waits are counted: s_waitcnt vmcnt(n)
//    with n = the vector-memory instructions issued after the copies that are waited for (never more: a lower bound is safe).
//  * lane = ONE column x V3_TH rows.  The row terms {sin v, m1 cos v, m4 cos v, m7 cos v} are wave-uniform: scalar loads into
//    SGPRs, no vector loads and no VGPRs; a lane's column terms are loaded once per strip; a byte read of the gather touches
//    32 consecutive pixels per half wave (96 bytes: no LDS bank conflict; the 2 x 4 lane blocks of round 2 conflicted 2.1-fold).
//  * Between map and gather a pixel is ONE dword {LDS address of its top-left tap, fx, fy, tap steps}: the reflect folds and
//    the address arithmetic happen in the map stage and two tiles in flight cost 2 x V3_TH VGPRs.
//  * The 16SC3 tile is transposed through the LDS region of its own (spent) source box, 8 rows at a time, and leaves as whole
//    1 KB runs: 3 global_store_dwordx4 per lane and 8 rows, every 128-byte line of a 384-byte tile row written whole by one
//    instruction (round 2's 12-byte-per-lane stores wrote partial lines at arbitrary 6-byte offsets: WRITE_SIZE 1.22 x the bytes).
//  * The two boxes in flight share a ring per wave: even tiles grow from its bottom, odd tiles from its top; a pair that does
//    not fit (tall boxes at a frame's corners) is simply not overlapped.
//  * Scalar instructions are the scarce kind here (one scalar unit per compute unit): rows come in 64-byte scalar loads, the
//    four box reductions are interleaved DPP chains that end in one v_readlane each, offsets are 32 bits, the two tiles in
//    flight swap roles instead of being copied.
//  * Tiles outside the fast path's guards (z <= 2^-30 somewhere, coordinates beyond +-2^15, a box larger than the ring, a
//    frame that is not dword aligned or larger than 4 GB) are redone after the strip's loop by a compact per-pixel loop: IEEE
//    division, x86 cvRound, global-memory taps.
// The float operations per pixel, the guards and the results are those of round 2's kernel.
#ifndef WV3_RING
#define WV3_RING 10240
#endif
#ifndef WV3_WAVES
#define WV3_WAVES 2
#endif
#ifndef WV3_CHX
#define WV3_CHX 2
#endif
#ifndef WV3_CHY
#define WV3_CHY 1
#endif
#ifndef WV3_WAVES_MIN
#define WV3_WAVES_MIN 4
#endif
#ifndef WV3_TH
#define WV3_TH 8
#endif
#ifndef WV3_GG
#define WV3_GG 4
#endif
#ifndef WV3_MG
#define WV3_MG 8
#endif
#ifndef WV3_PRIO
#define WV3_PRIO 0      // diagnostics: 1 = raised priority during the gather, 2 = during map + copies
#endif
#ifndef WV3_PITCH_ALIGN
#define WV3_PITCH_ALIGN 128
#endif
constexpr int V3_TW = 64, V3_TH = WV3_TH;
constexpr int V3_RING = WV3_RING, V3_WAVES = WV3_WAVES, V3_GG = WV3_GG, V3_MG = WV3_MG, V3_PITCH_ALIGN = WV3_PITCH_ALIGN;
constexpr int V3_OUT_IMG = V3_TW * 8 * 6, V3_OUT_BYTES = V3_OUT_IMG + V3_TW * 8;   // 8 rows of the 16SC3 tile + of the mask tile
static_assert(V3_TH % 8 == 0 && V3_TH <= 16, "tiles are whole groups of 8 rows");
static_assert(V3_OUT_BYTES <= V3_RING / 2 && V3_RING <= 65536, "two output groups must fit the ring; LDS addresses are 16 bits");
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f4v* cf4p;     // constant address space: uniform indices become scalar loads
typedef const __attribute__((address_space(4))) f16v* cf16p;
typedef const __attribute__((address_space(3))) uint8_t* lds_cp;
typedef __attribute__((address_space(3))) uint8_t* lds_p;
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

enum { V3F_WIDE_SRC = 1, V3F_FAST_OK = 2, V3F_WIDE_OUT = 4, V3F_WIDE_MASK = 8, V3F_STAMP = 16 /* diagnostics build: this frame records its tiles' stamps */ };
struct V3Frame {        // everything a frame's strips share (filled on the host: v3_frame_of)
    float m[9];
    float xhi2, yhi2;   // mask test: a coordinate rounds inside the frame iff -0.5 <= v < hi2
    int sw, sh, dw, dh, ncol;
    unsigned sstride, dstride, mstride, total_bytes, flags;
    const uint8_t* src;
    uint8_t* dst;
    uint8_t* mask;
    const float* tab;
};

struct V3Uni {      // wave-uniform description of a tile between its map and its gather
    int fast;       // 1: packed taps + staged box; 0: the per-pixel loop
    int interior;   // every tap inside the frame (mask all 255, tap steps +1 / +pitch)
    int pitch;      // LDS bytes per box row: a multiple of 128, so a tap's bank does not depend on its row (a half wave's 32 taps are
                    // 96 consecutive bytes of the image whatever rows they fall in: conflict free; with the tight pitch the rows'
                    // bank offsets collided: 42 % of the LDS cycles were conflicts, gpurun_out/r3_pmc_v32.txt)
    int rowb;       // bytes copied per row (whole pieces)
    int nrows, bytes;
    unsigned gbase;
};

__device__ __forceinline__ void div1_shared(float nx, float ny, float d, float* qx, float* qy) {
    const float r0 = __builtin_amdgcn_rcpf(d), nd = -d;
    const float e = __builtin_fmaf(nd, r0, 1.f);
    const float r = __builtin_fmaf(e, r0, r0);
    float q = nx * r;
    float t = __builtin_fmaf(nd, q, nx);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(nd, q, nx);
    *qx = __builtin_fmaf(t, r, q);
    q = ny * r;
    t = __builtin_fmaf(nd, q, ny);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(nd, q, ny);
    *qy = __builtin_fmaf(t, r, q);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n vector-memory instructions of this wave are outstanding (n uniform; any smaller immediate is safe)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    if (n >= 8) {
        if (n >= 12) wait_vm<12>(); else if (n >= 10) wait_vm<10>(); else wait_vm<8>();
    } else if (n >= 4) {
        if (n >= 6) { if (n == 7) wait_vm<7>(); else wait_vm<6>(); } else { if (n == 5) wait_vm<5>(); else wait_vm<4>(); }
    } else {
        if (n >= 2) { if (n == 3) wait_vm<3>(); else wait_vm<2>(); } else { if (n == 1) wait_vm<1>(); else wait_vm<0>(); }
    }
}

// One global -> LDS copy instruction outside the compiler's view (it would wait for EVERY outstanding copy before the next
// LDS read; the pipeline waits by count): LDS address = M0 + lane * PIECE, global address = base (SGPR pair) + voff.
// (s_nop 0: one wait state between a scalar write of M0 and an LDS-DMA instruction that reads it -- the hazard recogniser
// does not look into inline assembly.)
template <int PIECE>
__device__ __forceinline__ void dma_piece(const uint8_t* base, unsigned voff, uint32_t lds_off) {
    if constexpr (PIECE == 16) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_off) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(lds_off) : "memory", "m0");
}
// the copies of a box (layout as stage_box: row-major with `pitch`); returns the number of instructions issued (uniform)
template <int PIECE>
__device__ __forceinline__ int v3_stage_box(const uint8_t* src, unsigned gbase, unsigned sstride, uint32_t lds_off, int pitch, int rowb, int nrows, int lane) {
    const int lpr = pitch / PIECE, k = 64 / lpr;         // lanes per LDS row (the first rowb / PIECE of them copy), rows per instruction
    const int lr = (lane * (65536 / lpr + 1)) >> 16, lc = lane - lr * lpr;
    unsigned voff = gbase + (unsigned)lr * sstride + (unsigned)lc * PIECE;
    const unsigned vstep = (unsigned)k * sstride;
    const int lstep = k * pitch;
    int left = (lr < k && lc * PIECE < rowb) ? nrows - lr : 0;      // this lane copies a row while left > 0
    int n = 0;
#if WV_ABL == 3
    return 0;
#else
#pragma unroll 1
    for (int r0 = 0; r0 < nrows; r0 += k, n++) {
        if (left > 0) dma_piece<PIECE>(src, voff, lds_off);
        voff += vstep; lds_off += lstep; left -= k;
    }
    return n;
#endif
}

// four wave-wide integer reductions at once: min(a), max(b), min(c), max(d) -> uniform values.  The chains are interleaved, so
// a DPP read is always three instructions behind the write it depends on (two wait states needed); row_bcast 15 / 31 carry the
// row results up the wave and lane 63 holds the answers: 24 DPP instructions and four v_readlane, no scalar min / max.
__device__ __forceinline__ void wave_box_reduce(int& a, int& b, int& c, int& d) {
#define MIS_STEP4(CTRL, MASKS)                                                   \
    "v_min_i32_dpp %0, %0, %0 " CTRL " " MASKS "\n\t"                               \
    "v_max_i32_dpp %1, %1, %1 " CTRL " " MASKS "\n\t"                               \
    "v_min_i32_dpp %2, %2, %2 " CTRL " " MASKS "\n\t"                               \
    "v_max_i32_dpp %3, %3, %3 " CTRL " " MASKS "\n\t"
    asm volatile("s_nop 1\n\t"
                 MIS_STEP4("quad_perm:[1,0,3,2]", "row_mask:0xf bank_mask:0xf")
                 MIS_STEP4("quad_perm:[2,3,0,1]", "row_mask:0xf bank_mask:0xf")
                 MIS_STEP4("row_half_mirror", "row_mask:0xf bank_mask:0xf")
                 MIS_STEP4("row_mirror", "row_mask:0xf bank_mask:0xf")
                 MIS_STEP4("row_bcast:15", "row_mask:0xa bank_mask:0xf")
                 MIS_STEP4("row_bcast:31", "row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef MIS_STEP4
    a = __builtin_amdgcn_readlane(a, 63); b = __builtin_amdgcn_readlane(b, 63);
    c = __builtin_amdgcn_readlane(c, 63); d = __builtin_amdgcn_readlane(d, 63);
}

// packed pixel between map and gather: bits 0..15 LDS address of tap (0, 0); bits 16..23 = 8 fx + cx; bits 24..31 = 8 fy + cy, with
// cx / cy = the column / row step of the right / lower taps + 1 (steps are -1 / 0 / +1 under BORDER_REFLECT; interior tiles pack 0:
// their steps are +1 and the fields read as the weights 8 fx, 8 fy without a mask).  The weights are scaled by 8 x 8 = 64 so that
// the rounded bilinear sum lands in the upper half of its register: (sum + 512) >> 10 == (64 sum + 32768) >> 16, stored by a
// ds_write_b16_d16_hi without a shift.
__device__ __forceinline__ unsigned v3_pack(int addr, unsigned xq, unsigned yq, int cx, int cy) {
    // xq, yq: 5 fraction bits in bits 0..4 (anything above is shifted out or masked)
    return ((unsigned)addr | (unsigned)(cx << 16) | (unsigned)(cy << 24)) | ((xq << 19) & 0x00F80000u) | (yq << 27);
}

// Map stage: backward map of a lane's column x V3_TH rows of the tile at rows ty0.., box reduction, classification, packing
// (tile_map's logic; the cold decisions only set u.fast = 0: the per-pixel loop redoes such a tile from scratch).
// ring_lds / parity: where the tile's region of the ring will be (even tiles from the bottom, odd tiles from the top).
template <int RING>
__device__ __forceinline__ void v3_map(const V3Frame& f, cf4p rowtab, float su, float cu, int ty0, uint32_t ring_lds, int parity,
                                       unsigned* w, unsigned& msk_out, V3Uni& u) {
    int xq[V3_TH], yq[V3_TH];
    float zlo = 3.0e38f;
    f4v rt[V3_TH];
    if (ty0 + V3_TH <= f.dh) {      // whole tile inside the roi: the rows' terms are V3_TH * 16 consecutive bytes
        cf16p r16 = (cf16p)(rowtab + ty0);
#pragma unroll
        for (int g = 0; g < V3_TH / 4; g++) {
            const f16v r = r16[g];
            rt[4 * g] = r.s0123; rt[4 * g + 1] = r.s4567; rt[4 * g + 2] = r.s89ab; rt[4 * g + 3] = r.scdef;
        }
    } else {
#pragma unroll
        for (int i = 0; i < V3_TH; i++) rt[i] = rowtab[min(ty0 + i, f.dh - 1)];      // rows past the roi shadow the last one
    }
    auto terms = [&](int i, float* xx, float* yy, float* zz) {
        const float x_ = su * rt[i].x, z_ = cu * rt[i].x;
        *xx = (x_ * f.m[0] + rt[i].y) + z_ * f.m[2];
        *yy = (x_ * f.m[3] + rt[i].z) + z_ * f.m[5];
        *zz = (x_ * f.m[6] + rt[i].w) + z_ * f.m[8];
    };
#pragma unroll
    for (int i = 0; i < V3_TH; i++) {
        float xx, yy, zz, qx, qy;
#if WV_ABL == 5
        qx = su * 0.f + 100.25f + (float)(threadIdx.x & 63); qy = (float)(ty0 + i) + 100.5f; zz = 1.f;
#elif WV_ABL == 4
        terms(i, &xx, &yy, &zz);
        { const float r0 = __builtin_amdgcn_rcpf(zz); qx = xx * r0; qy = yy * r0; }
#else
        terms(i, &xx, &yy, &zz);
        div1_shared(xx, yy, zz, &qx, &qy);
#endif
        zlo = fminf(zlo, zz);
        // cvRound(32 x) by the 1.5 * 2^23 magic add; one fma: 32 x is exact, so fma(x, 32, magic) rounds once, like (32 x) + magic.
        // xq, yq keep the float's bits: 0x4B400000 + q with q the rounded coordinate (|q| < 2^22); the bias is a multiple of 2^22,
        // so bits 0..4 are q's fraction bits, bits 5..21 its integer part, and integer min / max order biased values like q.
        xq[i] = (int)__float_as_uint(__builtin_fmaf(qx, 32.f, 12582912.f));
        yq[i] = (int)__float_as_uint(__builtin_fmaf(qy, 32.f, 12582912.f));
        if (V3_MG < V3_TH && i % V3_MG == V3_MG - 1) __builtin_amdgcn_sched_barrier(0);     // bounds the pixels whose chains are in flight (registers)
    }
    int xmin = xq[0], xmax = xq[0], ymin = yq[0], ymax = yq[0];
#pragma unroll
    for (int q = 1; q < V3_TH; q++) {
        xmin = min(xmin, xq[q]); xmax = max(xmax, xq[q]);
        ymin = min(ymin, yq[q]); ymax = max(ymax, yq[q]);
    }
    wave_box_reduce(xmin, xmax, ymin, ymax);
    xmin -= 0x4B400000; xmax -= 0x4B400000; ymin -= 0x4B400000; ymax -= 0x4B400000;     // (uniform) back to q
    const bool good = (f.flags & V3F_FAST_OK) && !__any(!(zlo >= 9.31322574615478515625e-10f)) &&
                      xmin > -(1 << 20) && ymin > -(1 << 20) && xmax < (1 << 20) && ymax < (1 << 20);
    xmin >>= 5; xmax >>= 5; ymin >>= 5; ymax >>= 5;
    const bool interior = xmin >= 0 && ymin >= 0 && xmax + 1 <= f.sw - 1 && ymax + 1 <= f.sh - 1;
    const bool foldable = xmin >= -f.sw && xmax + 1 < 2 * f.sw && ymin >= -f.sh && ymax + 1 < 2 * f.sh;
    int bx0 = xmin, bx1 = xmax + 1, by0 = ymin, by1 = ymax + 1;
    if (!interior && foldable) {
        auto fold = [](int lo, int hi, int len, int* o0, int* o1) {
            if (lo >= 0 && hi < len) { *o0 = lo; *o1 = hi; }
            else if (hi < 0) { *o0 = -hi - 1; *o1 = -lo - 1; }
            else if (lo >= len) { *o0 = 2 * len - 1 - hi; *o1 = 2 * len - 1 - lo; }
            else if (lo < 0) { *o0 = 0; *o1 = max(-lo - 1, min(hi, len - 1)); if (hi >= len) *o1 = len - 1; }
            else { *o0 = min(lo, 2 * len - 1 - hi); *o1 = len - 1; }
        };
        fold(xmin, xmax + 1, f.sw, &bx0, &bx1);
        fold(ymin, ymax + 1, f.sh, &by0, &by1);
    }
    const bool wide = (f.flags & V3F_WIDE_SRC) != 0;
    const int amask = wide ? 15 : 3;
    const int shift = (bx0 * 3) & amask;
    const int rowb = ((bx1 - bx0 + 1) * 3 + shift + amask) & ~amask;
    const int nrows = by1 - by0 + 1;
    const int pitch_cf = (rowb + V3_PITCH_ALIGN - 1) & ~(V3_PITCH_ALIGN - 1);       // conflict free; the tight pitch when that takes more than the tile's share of the ring
    const int pitch = nrows * pitch_cf <= RING / 2 ? pitch_cf : rowb;
    const unsigned gbase = (unsigned)by0 * f.sstride + (unsigned)(bx0 * 3 - shift);    // frames are < 4 GB on the fast path
    const bool fast = good && (interior || foldable) && nrows * pitch <= RING && pitch <= (wide ? 1024 : 256) &&
                      gbase + (unsigned)(nrows - 1) * f.sstride + (unsigned)rowb <= f.total_bytes;
    u.fast = fast; u.interior = interior; u.pitch = pitch; u.rowb = rowb; u.nrows = nrows; u.gbase = gbase;
    u.bytes = fast ? max((nrows * pitch + 15) & ~15, V3_OUT_BYTES) : V3_OUT_BYTES;
    unsigned msk = (1u << V3_TH) - 1u;
    if (fast) {
        // LDS address of source pixel (0, 0) in the tile's region
        const int lbase = (int)ring_lds + (parity ? RING - u.bytes : 0) + shift - by0 * pitch - bx0 * 3;
        // integer parts: sign-extended bits 5..21 of the biased values (one v_bfe_i32 each)
        if (interior) {
#pragma unroll
            for (int i = 0; i < V3_TH; i++) {
                const int sx = __builtin_amdgcn_sbfe(xq[i], 5, 17), sy = __builtin_amdgcn_sbfe(yq[i], 5, 17);
                w[i] = v3_pack(lbase + __mul24(sy, pitch) + sx * 3, (unsigned)xq[i], (unsigned)yq[i], 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < V3_TH; i++) {
                const int sx = __builtin_amdgcn_sbfe(xq[i], 5, 17), sy = __builtin_amdgcn_sbfe(yq[i], 5, 17);
                const int x0 = mis_reflect1(sx, f.sw), x1 = mis_reflect1(sx + 1, f.sw), y0 = mis_reflect1(sy, f.sh), y1 = mis_reflect1(sy + 1, f.sh);
                w[i] = v3_pack(lbase + __mul24(y0, pitch) + x0 * 3, (unsigned)xq[i], (unsigned)yq[i], x1 - x0 + 1, y1 - y0 + 1);
            }
            // some taps leave the frame: the mask (nearest source pixel inside?) needs the unquantised coordinates again
            msk = 0;
#pragma unroll
            for (int i = 0; i < V3_TH; i++) {
                float xx, yy, zz, qx, qy;
                terms(i, &xx, &yy, &zz);
                div1_shared(xx, yy, zz, &qx, &qy);
                msk |= (unsigned)((qx >= -0.5f) & (qx < f.xhi2) & (qy >= -0.5f) & (qy < f.yhi2)) << i;
            }
        }
    }
    msk_out = msk;
}

// bilinear sample of a packed pixel (sample1's arithmetic with both weight factors scaled by 8): p[c] holds the channel in
// its UPPER 16 bits: (64 sum + 32768) with sum < 2^18
template <bool INTERIOR>
__device__ __forceinline__ void v3_sample(int pitch, unsigned w, unsigned* p) {
    unsigned fx8 = (w >> 16) & 0xffu, fy8 = w >> 24;
    lds_cp t00 = (lds_cp)(uintptr_t)(w & 0xffffu);
    lds_cp t01, t10, t11;
    if (INTERIOR) { t01 = t00 + 3; t10 = t00 + pitch; t11 = t10 + 3; }
    else {
        const int dx = ((int)(fx8 & 3) - 1) * 3, dy = __mul24((int)(fy8 & 3) - 1, pitch);
        fx8 &= ~7u; fy8 &= ~7u;
        t01 = t00 + dx; t10 = t00 + dy; t11 = t10 + dx;
    }
    const unsigned wa8 = 256 - fx8, wb8 = 256 - fy8;
    const unsigned w00 = __umul24(wa8, wb8), w01 = __umul24(fx8, wb8), w10 = __umul24(wa8, fy8), w11 = __umul24(fx8, fy8);
#pragma unroll
    for (int c = 0; c < 3; c++) p[c] = mad24(t11[c], w11, mad24(t10[c], w10, mad24(t01[c], w01, mad24(t00[c], w00, 32768u))));
}

__device__ __forceinline__ void v3_sample_global(const V3Frame& f, int xq, int yq, int* p) {
    const int fx = xq & 31, fy = yq & 31, sx = xq >> 5, sy = yq >> 5;
    const int x0 = mis_reflect(sx, f.sw), x1 = mis_reflect(sx + 1, f.sw), y0 = mis_reflect(sy, f.sh), y1 = mis_reflect(sy + 1, f.sh);
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const uint8_t* r0 = f.src + (size_t)y0 * f.sstride;
    const uint8_t* r1 = f.src + (size_t)y1 * f.sstride;
#pragma unroll
    for (int c = 0; c < 3; c++)
        p[c] = (r0[x0 * 3 + c] * w00 + r0[x1 * 3 + c] * w01 + r1[x0 * 3 + c] * w10 + r1[x1 * 3 + c] * w11 + (1 << 14)) >> 15;
}

// A tile outside the fast path, pixel by pixel (cold): generic map (IEEE division, x86 cvRound overflow, saturate_cast<short>
// of the integer part), taps from global memory, 2-byte stores.
__device__ __noinline__ void v3_cold_tile(const V3Frame& f, float su, float cu, int gx, int ty0) {
    if (gx >= f.dw) return;
    const float xhi = (float)f.sw - 0.5f, yhi = (float)f.sh - 0.5f;
    const bool xe = ((f.sw - 1) & 1) == 0, ye = ((f.sh - 1) & 1) == 0;
    const float4* rowtab = reinterpret_cast<const float4*>(f.tab + 2 * f.ncol);
#pragma unroll 1
    for (int i = 0; i < V3_TH; i++) {
        const int gy = ty0 + i;
        if (gy >= f.dh) break;
        const float4 rt = rowtab[gy];
        const float x_ = su * rt.x, z_ = cu * rt.x;
        const float xx = (x_ * f.m[0] + rt.y) + z_ * f.m[2];
        const float yy = (x_ * f.m[3] + rt.z) + z_ * f.m[5];
        const float zz = (x_ * f.m[6] + rt.w) + z_ * f.m[8];
        const bool front = zz > 0;
        const float x = front ? xx / zz : -1.f, y = front ? yy / zz : -1.f;
        const int xr = mis_round_sat_f(x * 32.f), yr = mis_round_sat_f(y * 32.f);
        int p[3];
        v3_sample_global(f, (mis_sat_short(xr >> 5) << 5) | (xr & 31), (mis_sat_short(yr >> 5) << 5) | (yr & 31), p);
        int16_t* d = reinterpret_cast<int16_t*>(f.dst + (size_t)gy * f.dstride + (size_t)gx * 6);
        d[0] = (int16_t)p[0]; d[1] = (int16_t)p[1]; d[2] = (int16_t)p[2];
        f.mask[(size_t)gy * f.mstride + gx] = (round_in_range(x, xhi, xe) & round_in_range(y, yhi, ye)) ? (uint8_t)255 : (uint8_t)0;
    }
}

__device__ __forceinline__ void v3_store_piece(uint8_t* p, u4v v, bool wide_out) {
    if (wide_out) *reinterpret_cast<u4v*>(p) = v;       // one global_store_dwordx4 (the vector type carries the 16-byte alignment)
    else { volatile uint32_t* q = reinterpret_cast<volatile uint32_t*>(p); q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
}

// The strip a wave owns: columns tx0 .. tx0 + 63, tiles ty_first .. ty_first + nt - 1 (V3_TH rows each).
// PIPE = false: the same stages one tile after the other (map, copies, wait, gather, stores) -- no second tile in flight, fewer
// registers, more waves per SIMD: the form for a launch of ONE frame, whose strips are too short to fill a pipeline.
template <bool PIPE, int RING>
__device__ __forceinline__ void warp_strip_body(const V3Frame& f, int tx0, int ty_first, int nt, lds_p ring) {
    const int lane = threadIdx.x & 63;
    const int nty = (f.dh + V3_TH - 1) / V3_TH;
    nt = min(nt, nty - ty_first);
    if (nt <= 0) return;
    const int gx = min(tx0 + lane, f.dw - 1);     // lanes past the roi shadow the last column (never stored)
    const float2 cs = *reinterpret_cast<const float2*>(f.tab + 2 * gx);
    cf4p rowtab = (cf4p)(uintptr_t)(f.tab + 2 * f.ncol);
    const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
    const bool wide_out = (f.flags & V3F_WIDE_OUT) != 0, wide_mask = (f.flags & V3F_WIDE_MASK) != 0;
    const int valid_px = min(f.dw - tx0, V3_TW), valid_bytes = valid_px * 6;
    // a group of 8 rows leaves as 192 pieces of 16 bytes: lane l stores pieces (l & 7) + 8 kk (kk = 0, 1, 2) of row l >> 3, so one
    // 32-bit offset per lane (plus immediates) addresses them, in the LDS image and in memory (unsigned offsets from a uniform
    // base: the saddr + voffset form); a store instruction writes one whole 128-byte line in each of the 8 rows
    const unsigned poff = (unsigned)(lane >> 3) * f.dstride + (unsigned)(lane & 7) * 16;
    const unsigned loff = (unsigned)(lane >> 3) * (V3_TW * 6) + (unsigned)(lane & 7) * 16;
    const unsigned moff = (unsigned)(lane >> 3) * f.mstride + (unsigned)(lane & 7) * 8;
    uint8_t* drow = f.dst + (size_t)(ty_first * V3_TH) * f.dstride + (size_t)tx0 * 6;       // the current tile's first row (bumped per tile)
    uint8_t* mrow = f.mask + (size_t)(ty_first * V3_TH) * f.mstride + tx0;

    unsigned wA[V3_TH], wB[V3_TH];
    unsigned mskA = 0, mskB = 0;
    V3Uni uA, uB;
    uA.fast = 0; uA.interior = 0; uA.pitch = 128; uA.rowb = 16; uA.nrows = 0; uA.bytes = V3_OUT_BYTES; uA.gbase = 0;
    uB = uA;
    int pend = 0;        // vector-memory instructions issued after the copies of the tile that is gathered next (a lower bound)
    unsigned cold = 0;   // bit k: tile k is outside the fast path
    auto issue = [&](const V3Uni& u, int parity) -> int {      // the copies of a tile's box into its region of the ring
        const uint32_t off = ring_lds + (parity ? (uint32_t)(RING - u.bytes) : 0u);
        return (f.flags & V3F_WIDE_SRC) ? v3_stage_box<16>(f.src, u.gbase, f.sstride, off, u.pitch, u.rowb, u.nrows, lane)
                                        : v3_stage_box<4>(f.src, u.gbase, f.sstride, off, u.pitch, u.rowb, u.nrows, lane);
    };
#ifdef WV_STAMPS     // diagnostics build (tools/warp_stamps3.py): shader-clock stamps at the phase boundaries of a step
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long wall0 = wall_clock64();
#define V3_STAMP(i) st[i] = __builtin_readcyclecounter()
#else
#define V3_STAMP(i)
#endif
    // one pipeline step: map + copies of tile k + 1 (into `nw`, `nu`), gather + stores of tile k (from `cw`, `cu`)
    auto step = [&](int k, unsigned* cw, unsigned& cmsk, V3Uni& cu_, unsigned* nw, unsigned& nmsk, V3Uni& nu) {
        const int ty0 = (ty_first + k) * V3_TH, parity = PIPE ? (k & 1) : 0;
        bool deferred = false;
        V3_STAMP(0);
#if WV3_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        if (!PIPE) {        // tile k itself: map, copies, (wait for everything)
            v3_map<RING>(f, rowtab, cs.x, cs.y, ty0, ring_lds, 0, cw, cmsk, cu_);
            V3_STAMP(1);
            if (cu_.fast) issue(cu_, 0);
            pend = 0;
        } else if (k + 1 < nt) {
            v3_map<RING>(f, rowtab, cs.x, cs.y, ty0 + V3_TH, ring_lds, parity ^ 1, nw, nmsk, nu);
            V3_STAMP(1);
            if (nu.fast) {
                if (k < 0) { issue(nu, parity ^ 1); pend = 0; }     // the strip's first tile: nothing is younger than its copies yet
                else if (cu_.bytes + nu.bytes <= RING) pend += issue(nu, parity ^ 1);
                else deferred = true;
            }
        }
        V3_STAMP(2);
#if WV3_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        if (k >= 0) {
            WSTAT(cu_.fast ? (cu_.interior ? 0 : 1) : 2);
            if (deferred) WSTAT(4);
            if (cu_.fast) {
                const uint32_t reg = ring_lds + (parity ? (uint32_t)(RING - cu_.bytes) : 0u);
                unsigned p[V3_TH][3];     // a channel in the upper 16 bits
                wait_vm_dyn(pend);
                V3_STAMP(3);
#if WV3_PRIO == 1
                __builtin_amdgcn_s_setprio(2);      // the gather's LDS reads go out ahead of other waves' arithmetic
#endif
                // (the empty asm makes a pixel's word opaque per branch: otherwise the common decode of all pixels is hoisted above
                // the branch and lives in 5 VGPRs per pixel)
                if (cu_.interior) {
#pragma unroll
                    for (int i = 0; i < V3_TH; i++) {
                        unsigned w = cw[i];
                        asm volatile("" : "+v"(w));
#if WV_ABL == 2
                        p[i][0] = w << 16; p[i][1] = w << 8; p[i][2] = w;
#else
                        v3_sample<true>(cu_.pitch, w, p[i]);
#endif
                        if (i % V3_GG == V3_GG - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < V3_TH; i++) {
                        unsigned w = cw[i];
                        asm volatile("" : "+v"(w));
                        v3_sample<false>(cu_.pitch, w, p[i]);
                        if (i % V3_GG == V3_GG - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
#if WV3_PRIO == 1
                __builtin_amdgcn_s_setprio(0);
#endif
                const bool ones = cu_.interior != 0;     // (uniform) all taps inside the frame: every lane's mask bits are set
                int nst = 0;
                asm volatile("" ::"v"(p[V3_TH - 1][2]));
                V3_STAMP(4);
                // 8 rows at a time through the (spent) region: a lane's column of 8 pixels -> rows of 384 bytes -> 16-byte pieces
#pragma unroll
                for (int g = 0; g < V3_TH / 8; g++) {
                    lds_p ob = (lds_p)(uintptr_t)(reg + lane * 6);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        // three 2-byte writes (volatile: merged into 4-byte writes at lane * 6 they are misaligned in every other lane,
                        // and the LDS then takes ~40 cycles per instruction: SQ_LDS_IDX_ACTIVE, gpurun_out/r3_pmc_v31.txt)
                        volatile __attribute__((address_space(3))) uint16_t* o = (volatile __attribute__((address_space(3))) uint16_t*)(ob + i * (V3_TW * 6));
                        o[0] = (uint16_t)(p[8 * g + i][0] >> 16); o[1] = (uint16_t)(p[8 * g + i][1] >> 16); o[2] = (uint16_t)(p[8 * g + i][2] >> 16);    // ds_write_b16_d16_hi
                    }
                    if (!ones) {
                        lds_p om = (lds_p)(uintptr_t)(reg + V3_OUT_IMG + lane);
#pragma unroll
                        for (int i = 0; i < 8; i++) om[i * V3_TW] = (cmsk >> (8 * g + i) & 1) ? (uint8_t)255 : (uint8_t)0;
                    }
                    uint8_t* dg = drow + (size_t)(8 * g) * f.dstride;
                    uint8_t* mg = mrow + (size_t)(8 * g) * f.mstride;
                    const int gy0 = ty0 + 8 * g;
                    if (valid_px == V3_TW && gy0 + 8 <= f.dh) {
                        u4v v[3];
#pragma unroll
                        for (int kk = 0; kk < 3; kk++) v[kk] = *(const __attribute__((address_space(3))) u4v*)(uintptr_t)(reg + loff + kk * 128);
#if WV_ABL == 1
                        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(dg + poff));
#else
                        if (wide_out) {
                            // scalar base + 32-bit lane offset + immediate: no 64-bit address pair per lane (three of them cost the
                            // kernel its fifth wave per SIMD)
                            // (s_nop 1: a store of more than 64 bits needs two wait states before its data VGPRs are overwritten, and the
                            // hazard recogniser does not see inside the statement -- tests/test_build_checks.py looks for it)
                            asm volatile("global_store_dwordx4 %0, %1, %4\n\tglobal_store_dwordx4 %0, %2, %4 offset:128\n\tglobal_store_dwordx4 %0, %3, %4 offset:256\n\ts_nop 1"
                                         ::"v"(poff), "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(dg) : "memory");
                        } else {
#pragma unroll
                            for (int kk = 0; kk < 3; kk++) v3_store_piece(dg + poff + kk * 128, v[kk], false);
                        }
#endif
                        const u2v mv = ones ? u2v{0xffffffffu, 0xffffffffu} : *(const __attribute__((address_space(3))) u2v*)(uintptr_t)(reg + V3_OUT_IMG + lane * 8);
#if WV_ABL == 1
                        asm volatile("" ::"v"(mv), "v"(mg + moff));
#else
                        if (wide_mask) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(moff), "v"(mv), "s"(mg) : "memory");
                        else { volatile uint16_t* q = reinterpret_cast<volatile uint16_t*>(mg + moff); q[0] = (uint16_t)mv.x; q[1] = (uint16_t)(mv.x >> 16); q[2] = (uint16_t)mv.y; q[3] = (uint16_t)(mv.y >> 16); }
#endif
                        nst += (wide_out ? 3 : 0) + (wide_mask ? 1 : 0);     // (narrow stores: not counted -- a lower bound keeps the wait safe)
                    } else if (gy0 < f.dh) {
                        // edge groups (the roi's last columns / rows): whole pieces where they fit, single shorts / bytes for the rest
                        // (rolled loops straight from the LDS image: this path must not cost the pipeline registers)
                        const int prow = lane >> 3;
#pragma unroll 1
                        for (int kk = 0; kk < 3; kk++) {
                            const int pcb = ((lane & 7) + 8 * kk) * 16;
                            if (gy0 + prow >= f.dh || pcb >= valid_bytes) continue;
                            const uint32_t lsrc = reg + loff + kk * 128;
                            uint8_t* d = dg + poff + kk * 128;
                            if (pcb + 16 <= valid_bytes) v3_store_piece(d, *(const __attribute__((address_space(3))) u4v*)(uintptr_t)lsrc, wide_out);
                            else {
#pragma unroll 1
                                for (int s2 = 0; pcb + 2 * s2 < valid_bytes && s2 < 8; s2++)
                                    reinterpret_cast<uint16_t*>(d)[s2] = *(const __attribute__((address_space(3))) uint16_t*)(uintptr_t)(lsrc + 2 * s2);
                            }
                        }
                        const int mc = (lane & 7) * 8;
                        if (gy0 + prow < f.dh && mc < valid_px) {
#pragma unroll 1
                            for (int b = 0; b < 8 && mc + b < valid_px; b++)
                                mg[moff + b] = ones ? (uint8_t)255 : *(const __attribute__((address_space(3))) uint8_t*)(uintptr_t)(reg + V3_OUT_IMG + lane * 8 + b);
                        }
                    }
                }
                V3_STAMP(5);
                pend = nst;     // issued after the copies of tile k + 1
            } else {
                cold |= 1u << k;            // redone pixel by pixel after the loop (keeps the call out of the pipeline's registers)
                pend = 0;
            }
            if (deferred) {     // the pair did not fit the ring together: tile k + 1's copies start now, into the space tile k left
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // tile k's reads of the ring have their data
                issue(nu, parity ^ 1);
                pend = 0;
            }
            drow += (size_t)V3_TH * f.dstride; mrow += (size_t)V3_TH * f.mstride;
#ifdef WV_STAMPS
            if (lane == 0 && (f.flags & V3F_STAMP)) {      // one frame of a batched grid (the middle one): the steady state
                const int tile = (ty_first + k) * ((f.dw + V3_TW - 1) / V3_TW) + tx0 / V3_TW;
                if (tile < 16384) { unsigned long long* o = g_warp_stamps + 8 * tile; for (int q = 0; q < 6; q++) o[q] = st[q]; o[6] = wall0; o[7] = wall_clock64(); }
            }
#endif
        }
    };
    if (PIPE) {
#pragma unroll 1
        for (int k = -1; k < nt; k += 2) {      // the two tiles in flight swap roles: no copies
            step(k, wB, mskB, uB, wA, mskA, uA);
            if (k + 1 < nt) step(k + 1, wA, mskA, uA, wB, mskB, uB);
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < nt; k++) step(k, wA, mskA, uA, wB, mskB, uB);
    }
    if (cold) {
        const V3Frame fc = f;       // a copy for the call: the kernel's own arguments stay in registers
#pragma unroll 1
        for (int k = 0; k < nt; k++)
            if (cold >> k & 1) v3_cold_tile(fc, cs.x, cs.y, tx0 + lane, (ty_first + k) * V3_TH);
    }
}

__device__ __forceinline__ bool v3_strip_of(int bid, int wave, int dw, int dh, int nt, int* tx0, int* ty_first) {
    const int nsx = (dw + V3_TW - 1) / V3_TW, nty = (dh + V3_TH - 1) / V3_TH, nsy = (nty + nt - 1) / nt;
    const int nwx = (nsx + V3_WAVES - 1) / V3_WAVES;
    constexpr int WG_PER_CHUNK = WV3_CHX * WV3_CHY;
    const int nchx = (nwx + WV3_CHX - 1) / WV3_CHX;
    const int i = bid >> 3, chunk = (i / WG_PER_CHUNK) * 8 + (bid & 7), within = i % WG_PER_CHUNK;
    const int chy = chunk / nchx, chx = chunk - chy * nchx;
    const int wx = chx * WV3_CHX + within % WV3_CHX, wy = chy * WV3_CHY + within / WV3_CHX;
    const int sx = wx * V3_WAVES + wave;
    if (sx >= nsx || wy >= nsy) return false;
    *tx0 = sx * V3_TW; *ty_first = wy * nt;
    return true;
}
static int v3_grid_of(const WarpArgs& a, int nt) {
    const int nsx = (a.dw + V3_TW - 1) / V3_TW, nty = (a.dh + V3_TH - 1) / V3_TH, nsy = (nty + nt - 1) / nt;
    const int nwx = (nsx + V3_WAVES - 1) / V3_WAVES;
    const int nchunks = ((nwx + WV3_CHX - 1) / WV3_CHX) * ((nsy + WV3_CHY - 1) / WV3_CHY);
    return ((nchunks + 7) / 8) * 8 * WV3_CHX * WV3_CHY;
}
#ifndef WV3_NT_MAX
#define WV3_NT_MAX 8       // <= 32: a strip's cold tiles are a bit mask
#endif
#ifndef WV3_SLOTS
#define WV3_SLOTS 16       // strips wanted per compute unit before strips get longer
#endif
// Tiles per strip: long strips amortise the pipeline's fill, but a launch should still put WV3_SLOTS waves on every compute unit
// (frames = the number of frames sharing the grid).
static int v3_strip_tiles(const MisContext* ctx, const WarpArgs& a, int frames) {
    const long long tiles = (long long)((a.dw + V3_TW - 1) / V3_TW) * ((a.dh + V3_TH - 1) / V3_TH) * std::max(frames, 1);
    const long long nt = tiles / ((long long)std::max(ctx->num_cu, 1) * WV3_SLOTS);
    return (int)std::min<long long>(WV3_NT_MAX, std::max<long long>(1, nt));
}
static V3Frame v3_frame_of(const WarpArgs& a, const float* tab) {
    V3Frame f;
    for (int i = 0; i < 9; i++) f.m[i] = a.m[i];
    const float xhi = (float)a.sw - 0.5f, yhi = (float)a.sh - 0.5f;     // exact: sizes < 2^15
    // v < hi || (even && v == hi)  <=>  v < hi2 with hi2 the next float above hi when the end point rounds inside (ties to even)
    f.xhi2 = ((a.sw - 1) & 1) == 0 ? nextafterf(xhi, 2.f * xhi) : xhi;
    f.yhi2 = ((a.sh - 1) & 1) == 0 ? nextafterf(yhi, 2.f * yhi) : yhi;
    f.sw = a.sw; f.sh = a.sh; f.dw = a.dw; f.dh = a.dh; f.ncol = trig_cols(a.dw);
    const unsigned long long total = (unsigned long long)(a.sh - 1) * a.sstride + (unsigned long long)a.sw * 3;      // last valid byte + 1
    const bool small = total < (1ull << 32) && a.sstride < (1ull << 31) && a.dstride < (1ull << 28) && a.mstride < (1ull << 28);
    f.sstride = (unsigned)a.sstride; f.dstride = (unsigned)a.dstride; f.mstride = (unsigned)a.mstride; f.total_bytes = (unsigned)total;
    f.flags = 0;
    if ((((size_t)a.src | a.sstride) & 15) == 0) f.flags |= V3F_WIDE_SRC;
    // |z| <= |m6| + |m7| + |m8| bounds z from above; dword-aligned frames below 4 GB only
    if (small && (((size_t)a.src | a.sstride) & 3) == 0 && fabsf(a.m[6]) + fabsf(a.m[7]) + fabsf(a.m[8]) <= 1048576.f) f.flags |= V3F_FAST_OK;
    if ((((size_t)a.dst | a.dstride) & 15) == 0) f.flags |= V3F_WIDE_OUT;
    if ((((size_t)a.mask | a.mstride) & 7) == 0) f.flags |= V3F_WIDE_MASK;
    f.src = a.src; f.dst = (uint8_t*)a.dst; f.mask = a.mask; f.tab = tab;
    return f;
}
struct V3Batch {
    V3Frame f[WB_MAX];
    int wg_base[WB_MAX + 1];    // frame k owns workgroups wg_base[k] .. wg_base[k + 1] - 1 of the (one-dimensional) grid; multiples of 8
    int nt[WB_MAX];             // tiles per strip of frame k
};

// Tiles per strip for every frame of one launch (frames in grid order).  `base` is what v3_strip_tiles asks for; the frames at the
// end of the grid get shorter strips so that the launch's tail is made of short waves: a frame whose tiles all lie within the last
// `WV3_TAIL2` tile-rounds of the launch (one round = one tile per wave slot of the device) takes 2-tile strips, within the last
// WV3_TAIL4 rounds 4-tile strips.  (Short strips pay the pipeline's fill more often: 3 maps per 2 gathers instead of 9 per 8.)
#ifndef WV3_TAIL2
#define WV3_TAIL2 4
#endif
#ifndef WV3_TAIL4
#define WV3_TAIL4 8
#endif
static void v3_plan_strips(const MisContext* ctx, const WarpArgs* args, int ng, int base, int* nt_out) {
    const long long slots = (long long)std::max(ctx->num_cu, 1) * 4 * WV3_WAVES_MIN;
    long long after = 0;        // tiles of the frames behind frame k in the grid
    static const char* plan_env = getenv("MIS_WARP_NT_PLAN");      // experiments: "8,8,...,4,2" (per frame of a launch, the last entry repeats)
    for (int k = ng - 1; k >= 0; k--) {
        const long long tiles = (long long)((args[k].dw + V3_TW - 1) / V3_TW) * ((args[k].dh + V3_TH - 1) / V3_TH);
        int nt = base;
        if (after + tiles <= slots * WV3_TAIL2) nt = std::min(nt, 2);
        else if (after + tiles <= slots * WV3_TAIL4) nt = std::min(nt, 4);
        nt_out[k] = nt;
        after += tiles;
    }
    if (plan_env) {
        int v = base, k = 0;
        const char* p = plan_env;
        while (k < ng) {
            if (*p) { v = atoi(p); while (*p && *p != ',') p++; if (*p == ',') p++; }
            nt_out[k++] = std::max(1, std::min(v, WV3_NT_MAX));
        }
    }
}

// The compose loop's grid: the strips of all frames of a batch, frame after frame, in ONE one-dimensional grid.  The dispatcher
// hands out workgroups in index order as slots free up, so the strips of the LAST frames are the launch's tail: the host plans
// them shorter (v3_plan_strips) -- with 8-tile strips everywhere a 16-frame launch is 7.25 generations of ~37 us waves and a
// quarter of the slots run an eighth wave while the others idle.
__global__ __launch_bounds__(64 * V3_WAVES) __attribute__((amdgpu_waves_per_eu(WV3_WAVES_MIN, 8))) void warp_strip_batch_kernel(V3Batch b) {
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[V3_WAVES][V3_RING];
    const int bid = (int)blockIdx.x;
    int fi = 0;
#pragma unroll
    for (int k = 1; k < WB_MAX; k++) fi += bid >= b.wg_base[k] ? 1 : 0;       // (uniform: scalar compares on the kernel arguments)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const V3Frame& f = b.f[fi];
    const int nt = b.nt[fi];
    int tx0, ty_first;
    if (!v3_strip_of(bid - b.wg_base[fi], wave, f.dw, f.dh, nt, &tx0, &ty_first)) return;
    warp_strip_body<true, V3_RING>(f, tx0, ty_first, nt, (lds_p)ring_all[wave]);
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

// `known`: the roi warpRoi / mis_warp_roi_batch returned for exactly these (scale, K, R, size) -- skips the border walk
int setup(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9], WarpArgs* a, int* brx, int* bry, const MisRect* known = nullptr) {
    MIS_CHECK(ctx, src && K && R, MIS_E_INVALID, "null argument");
    MIS_CHECK(ctx, src->dtype == MIS_U8 && (src->channels == 1 || src->channels == 3), MIS_E_UNSUPPORTED,
              "warp source must be 8UC1 or 8UC3");
    MIS_CHECK(ctx, src->width >= 2 && src->height >= 2 && src->width <= 32767 && src->height <= 32767, MIS_E_INVALID,
              "source size %dx%d out of range", src->width, src->height);
    MIS_CHECK(ctx, scale > 0.f, MIS_E_INVALID, "scale must be positive");
    Projector p;
    int tlx, tly;
    if (known) {
        projector_set(&p, scale, K, R);
        tlx = known->x; tly = known->y; *brx = known->x + known->width - 1; *bry = known->y + known->height - 1;
    } else {
        projector_and_roi(scale, K, R, src->width, src->height, &p, &tlx, &tly, brx, bry);
    }
    for (int i = 0; i < 9; i++) a->m[i] = p.k_rinv[i];
    a->scale = scale; a->tlx = tlx; a->tly = tly;
    a->dw = *brx - tlx + 1; a->dh = *bry - tly + 1;
    a->sw = src->width; a->sh = src->height; a->cn = src->channels;
    MIS_CHECK(ctx, a->dw > 0 && a->dh > 0 && (long long)a->dw * a->dh < (1ll << 31), MIS_E_INVALID, "degenerate warp roi %dx%d", a->dw, a->dh);
    return MIS_OK;
}

}  // namespace

#ifdef MIS_WARP_STATS
extern "C" int mis_debug_warp_stats(unsigned* out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_warp_stats), sizeof(unsigned) * 8);
    if (reset) { unsigned z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_warp_stats), z, sizeof(z)); }
    return 0;
}
#endif

#ifdef WV_STAMPS
extern "C" int mis_debug_warp_stamps(unsigned long long* out, int n) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_warp_stamps), sizeof(unsigned long long) * 8 * n);
    return 0;
}
#endif

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

static int grid_of(const WarpArgs& a, int* ntiles);
static int warp_fused_impl(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9],
                           MisImage* dst, MisImage* dmask, MisPoint* tl, int repeats, float* avg_us, const MisRect* known_roi = nullptr) {
    if (!ctx) return MIS_E_INVALID;
    WarpArgs a;
    int brx, bry, rc;
    if ((rc = setup(ctx, src, scale, K, R, &a, &brx, &bry, known_roi)) != MIS_OK) return rc;
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
    const size_t tab_bytes = sizeof(float) * trig_table_floats(a.dw, a.dh);
    if (ctx->stage_bytes < tab_bytes) {
        if (ctx->stage) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
        MIS_HIP(ctx, hipMalloc(&ctx->stage, tab_bytes * 2 + 4096));
        ctx->stage_bytes = tab_bytes * 2 + 4096;
    }
    float* tab = (float*)ctx->stage;
    hipLaunchKernelGGL(warp_trig_kernel, dim3((trig_cols(a.dw) + a.dh + 255) / 256), dim3(256), 0, ctx->stream, a, tab);
    int ntiles;
    const int nwg = grid_of(a, &ntiles);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (avg_us) {
        MIS_HIP(ctx, hipEventCreate(&e0));
        MIS_HIP(ctx, hipEventCreate(&e1));
        MIS_HIP(ctx, hipEventRecord(e0, ctx->stream));
    }
    // one frame per launch: round 2's tile kernel (measured faster there than the strip forms: 23.2 us against 29.3 for one-tile
    // strips and more for longer ones -- a frame alone is 3.5 tiles per wave slot, gpurun_out/r4_plan_ab3.txt)
    for (int rep = 0; rep < repeats; rep++)
        hipLaunchKernelGGL(warp_fused_kernel, dim3(nwg), dim3(64 * TILE_WAVES), 0, ctx->stream, a, (const float*)tab, ntiles);
    if (avg_us) {
        float ms = 0.f;
        MIS_HIP(ctx, hipEventRecord(e1, ctx->stream));
        MIS_HIP(ctx, hipEventSynchronize(e1));
        MIS_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        *avg_us = ms * 1000.f / (float)repeats;
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dout)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_commit(ctx, dmask, &dm)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_release(ctx, &din)) != MIS_OK) return rc;
    if (tl) { tl->x = a.tlx; tl->y = a.tly; }
    return MIS_OK;
}

extern "C" int mis_warp_spherical_fused(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9],
                                        MisImage* dst, MisImage* dmask, MisPoint* tl) {
    return warp_fused_impl(ctx, src, scale, K, R, dst, dmask, tl, 1, nullptr);
}

// the compose loop's form: the roi is the one mis_warp_roi / mis_warp_roi_batch gave for these parameters
extern "C" int mis_warp_spherical_fused_roi(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9], const MisRect* roi,
                                            MisImage* dst, MisImage* dmask, MisPoint* tl) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, roi && roi->width > 0 && roi->height > 0, MIS_E_INVALID, "empty roi");
    return warp_fused_impl(ctx, src, scale, K, R, dst, dmask, tl, 1, nullptr, roi);
}

static int grid_of(const WarpArgs& a, int* ntiles) {
    *ntiles = ((a.dw + FT_W - 1) / FT_W) * ((a.dh + FT_H - 1) / FT_H);
#if WV_ORDER == 2
    // whole chunks, a multiple of 8 of them (workgroups past the tile grid return at once)
    const int nchunks = (((a.dw + FT_W - 1) / FT_W + CH_W - 1) / CH_W) * (((a.dh + FT_H - 1) / FT_H + CH_H - 1) / CH_H);
    return ((nchunks + 7) / 8) * 8 * (CH_W / TILE_WAVES) * CH_H;
#else
    return (*ntiles + TILE_WAVES - 1) / TILE_WAVES;
#endif
}

// n fused warps (the loop of image_stitching.cpp:1154-1164 for all frames) in one grid per WB_MAX frames; the results are those of
// n mis_warp_spherical_fused_roi calls.  repeats / avg_us: the batch launched `repeats` times back to back between two HIP events
// (avg_us = the average duration of one pass over all n frames), for the roofline measurement.
static int warp_fused_batch_impl(MisContext* ctx, const MisImage* srcs, int n, float scale, const float* Ks, const float* Rs, const MisRect* rois,
                                 MisImage* dsts, MisImage* dmasks, MisPoint* tls, int repeats, float* avg_us) {
    MIS_CHECK(ctx, srcs && Ks && Rs && rois && dsts && dmasks && n >= 1, MIS_E_INVALID, "null argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<WarpArgs> args((size_t)n);
    std::vector<DevImage> din((size_t)n), dout((size_t)n), dm((size_t)n);
    std::vector<size_t> tab_off((size_t)n);
    size_t tab_total = 0;
    int rc = MIS_OK, got = 0;
    // which outputs this call allocates (data == NULL on entry): an error hands them back and leaves the caller's structs as they were
    std::vector<uint8_t> fresh_d((size_t)n), fresh_m((size_t)n);
    for (int i = 0; i < n; i++) { fresh_d[i] = dsts[i].data == nullptr; fresh_m[i] = dmasks[i].data == nullptr; }
    auto undo = [&](int upto) {      // frames 0 .. upto (inclusive) may hold staged inputs, device twins of host outputs, fresh outputs
        hipStreamSynchronize(ctx->stream);
        for (int i = 0; i <= upto && i < n; i++) {
            if (din[i].owned && din[i].data) hipFree(din[i].data);
            if (dout[i].owned && dout[i].data) hipFree(dout[i].data);
            if (dm[i].owned && dm[i].data) hipFree(dm[i].data);
            if (fresh_d[i] && dsts[i].data) { hipFree(dsts[i].data); dsts[i].data = nullptr; }
            if (fresh_m[i] && dmasks[i].data) { hipFree(dmasks[i].data); dmasks[i].data = nullptr; }
            din[i] = DevImage(); dout[i] = DevImage(); dm[i] = DevImage();
        }
    };
    for (; got < n; got++) {
        const int i = got;
        WarpArgs& a = args[i];
        int brx, bry;
        if (!(rois[i].width > 0 && rois[i].height > 0)) { rc = mis_set_error(ctx, MIS_E_INVALID, "frame %d: empty roi", i); break; }
        if ((rc = setup(ctx, &srcs[i], scale, Ks + 9 * i, Rs + 9 * i, &a, &brx, &bry, &rois[i])) != MIS_OK) break;
        if (srcs[i].channels != 3) { rc = mis_set_error(ctx, MIS_E_UNSUPPORTED, "fused warp needs an 8UC3 source"); break; }
        if ((rc = mis_dev_image_in(ctx, &srcs[i], &din[i])) != MIS_OK) break;
        if ((rc = mis_dev_image_out(ctx, &dsts[i], a.dw, a.dh, 3, MIS_S16, &dout[i])) != MIS_OK) break;
        if ((rc = mis_dev_image_out(ctx, &dmasks[i], a.dw, a.dh, 1, MIS_U8, &dm[i])) != MIS_OK) break;
        if (!(dout[i].stride % 4 == 0 && dm[i].stride % 2 == 0 && ((uintptr_t)dout[i].data % 4) == 0 && ((uintptr_t)dm[i].data % 2) == 0)) {
            rc = mis_set_error(ctx, MIS_E_INVALID, "fused warp outputs need 4-byte (image) / 2-byte (mask) aligned rows"); break;
        }
        a.src = (const uint8_t*)din[i].data; a.sstride = din[i].stride;
        a.dst = dout[i].data; a.dstride = dout[i].stride; a.mask = (uint8_t*)dm[i].data; a.mstride = dm[i].stride;
        tab_off[i] = tab_total;
        tab_total += mis_align_up(sizeof(float) * trig_table_floats(a.dw, a.dh), 256);
    }
    if (rc != MIS_OK) { undo(got); return rc; }
    if (ctx->stage_bytes < tab_total) {
        if (ctx->stage) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
        MIS_HIP(ctx, hipMalloc(&ctx->stage, tab_total * 2 + 4096));
        ctx->stage_bytes = tab_total * 2 + 4096;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<WarpBatch> batches;
    std::vector<dim3> grids;
#if WV_V3
    std::vector<V3Batch> v3batches;
    int nt3 = 1 << 30;      // tiles per strip: one value for the grid (the smallest any frame asks for)
    for (int i = 0; i < n; i++) nt3 = std::min(nt3, v3_strip_tiles(ctx, args[i], n));
#endif
    for (int g0 = 0; g0 < n; g0 += WB_MAX) {
        const int ng = std::min(WB_MAX, n - g0);
        WarpBatch b;
        int max_wg = 0, max_trig = 0;
        for (int k = 0; k < WB_MAX; k++) {
            const int i = g0 + (k < ng ? k : 0);
            b.a[k] = args[i]; b.tab[k] = (const float*)((uint8_t*)ctx->stage + tab_off[i]);
            b.nwg[k] = grid_of(args[i], &b.ntiles[k]);
            if (k < ng) { max_wg = std::max(max_wg, b.nwg[k]); max_trig = std::max(max_trig, (trig_cols(args[i].dw) + args[i].dh + 255) / 256); }
        }
        hipLaunchKernelGGL(warp_trig_batch_kernel, dim3(max_trig, ng), dim3(256), 0, ctx->stream, b);
#if WV_V3
        // one-dimensional grid: frame after frame, per-frame strip lengths (short strips for the frames at the launch's end)
        V3Batch vb;
        int nts[WB_MAX];
        v3_plan_strips(ctx, &args[g0], ng, nt3, nts);
        int wg = 0;
        for (int k = 0; k < WB_MAX; k++) {
            vb.f[k] = v3_frame_of(b.a[k], b.tab[k]);
#ifdef WV_STAMPS
            if (k == ng / 2) vb.f[k].flags |= V3F_STAMP;
#endif
            vb.nt[k] = k < ng ? nts[k] : 1;
            vb.wg_base[k] = wg;
            if (k < ng) wg += v3_grid_of(args[g0 + k], nts[k]);
        }
        vb.wg_base[WB_MAX] = wg;
        v3batches.push_back(vb);
        max_wg = wg; 
        batches.push_back(b); grids.push_back(dim3(max_wg));
#else
        batches.push_back(b); grids.push_back(dim3(max_wg, ng));
#endif
    }
    if (avg_us) {
        MIS_HIP(ctx, hipEventCreate(&e0));
        MIS_HIP(ctx, hipEventCreate(&e1));
        MIS_HIP(ctx, hipEventRecord(e0, ctx->stream));
    }
    for (int rep = 0; rep < repeats; rep++)
        for (size_t g = 0; g < batches.size(); g++)
#if WV_V3
            hipLaunchKernelGGL(warp_strip_batch_kernel, grids[g], dim3(64 * V3_WAVES), 0, ctx->stream, v3batches[g]);
#else
            hipLaunchKernelGGL(warp_fused_batch_kernel, grids[g], dim3(64 * TILE_WAVES), 0, ctx->stream, batches[g]);
#endif
    if (avg_us) {
        float ms = 0.f;
        MIS_HIP(ctx, hipEventRecord(e1, ctx->stream));
        MIS_HIP(ctx, hipEventSynchronize(e1));
        MIS_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        *avg_us = ms * 1000.f / (float)repeats;
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    MIS_HIP(ctx, hipGetLastError());
    for (int i = 0; i < n; i++) {
        int r1 = mis_dev_image_commit(ctx, &dsts[i], &dout[i]), r2 = mis_dev_image_commit(ctx, &dmasks[i], &dm[i]), r3 = mis_dev_image_release(ctx, &din[i]);
        if (rc == MIS_OK) rc = r1 != MIS_OK ? r1 : (r2 != MIS_OK ? r2 : r3);
        if (tls) { tls[i].x = args[i].tlx; tls[i].y = args[i].tly; }
    }
    return rc;
}

extern "C" int mis_warp_spherical_fused_batch(MisContext* ctx, const MisImage* srcs, int n, float scale, const float* Ks, const float* Rs, const MisRect* rois,
                                              MisImage* dsts, MisImage* dmasks, MisPoint* tls) {
    if (!ctx) return MIS_E_INVALID;
    if (n == 0) return MIS_OK;
    return warp_fused_batch_impl(ctx, srcs, n, scale, Ks, Rs, rois, dsts, dmasks, tls, 1, nullptr);
}

extern "C" int mis_warp_spherical_fused_batch_timed(MisContext* ctx, const MisImage* srcs, int n, float scale, const float* Ks, const float* Rs, const MisRect* rois,
                                                    MisImage* dsts, MisImage* dmasks, MisPoint* tls, int repeats, float* avg_us) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, repeats >= 1 && avg_us && n >= 1, MIS_E_INVALID, "repeats must be >= 1, n >= 1 and avg_us non-null");
    return warp_fused_batch_impl(ctx, srcs, n, scale, Ks, Rs, rois, dsts, dmasks, tls, repeats, avg_us);
}

extern "C" int mis_warp_roi_batch(MisContext* ctx, float scale, int w, int h, int n, const float* Ks, const float* Rs, MisRect* rois) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, Ks && Rs && rois && n >= 1 && w >= 1 && h >= 1 && scale > 0.f, MIS_E_INVALID, "invalid argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need = sizeof(RoiJob) * (size_t)n;
    if (ctx->roi_pinned_bytes < need) {
        if (ctx->roi_pinned) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipHostFree(ctx->roi_pinned)); ctx->roi_pinned = nullptr; ctx->roi_pinned_bytes = 0; }
        MIS_HIP(ctx, hipHostMalloc(&ctx->roi_pinned, need * 2, hipHostMallocMapped));
        ctx->roi_pinned_bytes = need * 2;
    }
    RoiJob* jobs = (RoiJob*)ctx->roi_pinned;
    std::vector<Projector> proj((size_t)n);
    for (int i = 0; i < n; i++) {
        projector_set(&proj[i], scale, Ks + 9 * i, Rs + 9 * i);
        memcpy(jobs[i].r_kinv, proj[i].r_kinv, sizeof(jobs[i].r_kinv));
        jobs[i].scale = scale; jobs[i].sw = w; jobs[i].sh = h;
    }
    hipLaunchKernelGGL(warp_roi_kernel, dim3(n), dim3(256), 0, ctx->stream, jobs);
    MIS_HIP(ctx, hipGetLastError());
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; i++) {
        int tlx, tly, brx, bry;
        roi_from_extremes(&proj[i], w, h, jobs[i].ext[0], jobs[i].ext[1], jobs[i].ext[2], jobs[i].ext[3], &tlx, &tly, &brx, &bry);
        rois[i].x = tlx; rois[i].y = tly; rois[i].width = brx + 1 - tlx; rois[i].height = bry + 1 - tly;
    }
    return MIS_OK;
}

extern "C" int mis_warp_spherical_fused_timed(MisContext* ctx, const MisImage* src, float scale, const float K[9], const float R[9],
                                              MisImage* dst, MisImage* dmask, MisPoint* tl, int repeats, float* avg_us) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, repeats >= 1 && avg_us, MIS_E_INVALID, "repeats must be >= 1 and avg_us non-null");
    return warp_fused_impl(ctx, src, scale, K, R, dst, dmask, tl, repeats, avg_us);
}
