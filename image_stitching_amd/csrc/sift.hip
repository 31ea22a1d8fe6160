// sift.hip -- cv::SIFT::create()->detectAndCompute on the GPU (config 5 of BASELINE.json).
// Reference call sites: image_stitching/image_stitching.cpp:559 (`SIFT::create()` when features_type == "sift"),
// :613 (`computeImageFeatures`).  OpenCV source restated: features2d sift.dispatch.cpp / sift.simd.hpp (4.5+, float
// scale space), imgproc GaussianBlur / resize -- see oracle/mo_sift.h for the restatement choices.
//
// Stages (all float arithmetic is IEEE + - * / sqrt in the CPU path's order, no FMA contraction):
//   gray (15-bit coefficients) -> 2x bilinear upsample -> Gaussian pyramid (separable blur, nearest 2:1 decimation between
//   octaves) -> DoG -> 26-neighbour extrema (candidates appended through a counter) -> per candidate: 3-D quadratic
//   refinement, contrast / edge tests, 36-bin orientation histogram (sequential sums: order matters in float) ->
//   keypoints -> canonical order of KeyPointsFilter::removeDuplicatedSorted + duplicate removal (a total order, so
//   the append order of the atomics never shows) -> 4x4x8 descriptor per keypoint.
// The sort / duplicate removal of the (tens of thousands of) keypoints runs on the host inside this library between
// the two device phases; everything that touches pixels runs on the device.
#include "common.h"
#include "dev_math.h"
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace {

constexpr int SIFT_IMG_BORDER = 5, SIFT_MAX_INTERP_STEPS = 5, SIFT_ORI_HIST_BINS = 36;
constexpr float SIFT_INIT_SIGMA = 0.5f, SIFT_ORI_SIG_FCTR = 1.5f, SIFT_ORI_RADIUS = 3 * 1.5f, SIFT_ORI_PEAK_RATIO = 0.8f;
constexpr float SIFT_DESCR_SCL_FCTR = 3.f, SIFT_DESCR_MAG_THR = 0.2f, SIFT_INT_DESCR_FCTR = 512.f;
constexpr int MAX_OCT = 16, MAX_LAYERS = 8, MAX_TAPS = 64;

struct Taps {
    int n;
    float k[MAX_TAPS];
};

// GaussianBlur: ksize from sigma for float images; getGaussianKernel in double, normalised, stored as float
Taps gaussian_taps(double sigma) {
    Taps t;
    int n = mis_round_d(sigma * 8 + 1) | 1;
    if (n > MAX_TAPS - 1) n = MAX_TAPS - 1;
    double v[MAX_TAPS], sum = 0, scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        v[i] = exp(scale2x * x * x);
        sum += v[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) t.k[i] = (float)(v[i] * sum);
    t.n = n;
    return t;
}

struct Pyr {   // device-visible description of the scale space of the current image
    int noct, nl;
    int w[MAX_OCT], h[MAX_OCT];
    float* gauss[MAX_OCT * (MAX_LAYERS + 3)];
    float* dog[MAX_OCT * (MAX_LAYERS + 2)];
};

__global__ __launch_bounds__(256) void sift_gray_kernel(const uint8_t* __restrict__ bgr, size_t stride, int w, int h, uint8_t* __restrict__ gray) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = bgr + (size_t)y * stride + 3 * (size_t)x;
    gray[(size_t)y * w + x] = (uint8_t)((s[0] * 3735 + s[1] * 19235 + s[2] * 9798 + (1 << 14)) >> 15);
}

// resize(float, 2x, INTER_LINEAR): source coordinate (d + 0.5) * 0.5 - 0.5, taps clamped, horizontal then vertical
__global__ __launch_bounds__(256) void sift_upsample_kernel(const uint8_t* __restrict__ g, int w, int h, float* __restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, dw = 2 * w;
    if (x >= dw) return;
    float fy = (float)((y + 0.5) * 0.5 - 0.5);
    int sy = mis_floor_f(fy);
    fy -= sy;
    if (sy < 0) { sy = 0; fy = 0; }
    if (sy >= h - 1) { sy = h - 1; fy = 0; }
    const int sy1 = sy + 1 < h ? sy + 1 : sy;
    float fx = (float)((x + 0.5) * 0.5 - 0.5);
    int sx = mis_floor_f(fx);
    fx -= sx;
    if (sx < 0) { sx = 0; fx = 0; }
    if (sx >= w - 1) { sx = w - 1; fx = 0; }
    const int sx1 = sx + 1 < w ? sx + 1 : sx;
    const uint8_t *r0 = g + (size_t)sy * w, *r1 = g + (size_t)sy1 * w;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float h0 = (float)r0[sx] * a0 + (float)r0[sx1] * a1;
    const float h1 = (float)r1[sx] * a0 + (float)r1[sx1] * a1;
    dst[(size_t)y * dw + x] = h0 * b0 + h1 * b1;
}

// separable Gaussian, BORDER_REFLECT_101, taps accumulated in ascending order (the CPU path's order).  A thread
// produces BLK consecutive outputs along the filtered axis from BLK + N - 1 loads; with N a template parameter the
// tap index tests fold away.  N = 0: generic tap count (t.n).  The column pass optionally writes the DoG image
// (this layer minus the previous one) as well, saving a pass over both.
constexpr int BLK = 8;
// Row pass: a workgroup owns ROW_SPAN consecutive outputs of one row.  The inputs (span + taps - 1, borders reflected)
// are staged in LDS with coalesced loads, each thread then computes its BLK consecutive outputs out of registers
// filled from LDS, and the results go back through LDS so that the stores are coalesced too.  LDS index i is stored at
// i + i / 8: a thread's k-th read is word 9 * tid + k + k / 8, and 9 is odd, so the 64 lanes hit 64 different banks.
constexpr int ROW_SPAN = 256 * BLK;
__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 3); }
template <int N>
__global__ __launch_bounds__(256) void sift_blur_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, Taps t) {
    __shared__ float lds[(ROW_SPAN + MAX_TAPS) + (ROW_SPAN + MAX_TAPS) / 8 + 8];
    const int n = N ? N : t.n, r = n / 2;
    const int xb = blockIdx.x * ROW_SPAN, y = blockIdx.y, tid = threadIdx.x;
    const int outs = min(ROW_SPAN, w - xb), span = outs + n - 1;
    const float* row = src + (size_t)y * w;
    const bool inner = xb - r >= 0 && xb + outs - 1 + r < w;
    if (inner)
        for (int i = tid; i < span; i += 256) lds[lds_pad(i)] = row[xb + i - r];
    else
        for (int i = tid; i < span; i += 256) lds[lds_pad(i)] = row[mis_reflect101(xb + i - r, w)];
    __syncthreads();
    float acc[BLK];
#pragma unroll
    for (int j = 0; j < BLK; j++) acc[j] = 0.f;
    const bool live = tid * BLK < outs;
    if (live) {
        const float* my = lds + 9 * tid;     // lds_pad(8 * tid + k) = 9 * tid + k + k / 8
        if (N) {
#pragma unroll
            for (int k = 0; k < N + BLK - 1; k++) {
                // inputs past the span (a partial last block) are never combined with a live output
                const float v = (tid * BLK + k < span) ? my[k + (k >> 3)] : 0.f;
#pragma unroll
                for (int j = 0; j < BLK; j++)
                    if (k - j >= 0 && k - j < N) acc[j] += t.k[k - j] * v;
            }
        } else {
            for (int k = 0; k < n + BLK - 1; k++) {
                const float v = (tid * BLK + k < span) ? my[k + (k >> 3)] : 0.f;
#pragma unroll
                for (int j = 0; j < BLK; j++)
                    if (k - j >= 0 && k - j < n) acc[j] += t.k[k - j] * v;
            }
        }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int j = 0; j < BLK; j++) lds[9 * tid + j] = acc[j];
    }
    __syncthreads();
    float* orow = dst + (size_t)y * w + xb;
    for (int i = tid; i < outs; i += 256) orow[i] = lds[lds_pad(i)];
}

template <int N>
__global__ __launch_bounds__(256) void sift_blur_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, Taps t,
                                                            const float* __restrict__ prev, float* __restrict__ dog) {
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * BLK;
    if (x >= w) return;
    const int n = N ? N : t.n, r = n / 2;
    float acc[BLK];
#pragma unroll
    for (int j = 0; j < BLK; j++) acc[j] = 0.f;
    const bool inner = y0 - r >= 0 && y0 + BLK - 1 + r < h;
    if (N) {
#pragma unroll
        for (int k = 0; k < N + BLK - 1; k++) {
            const int ys = y0 + k - r;
            const float v = src[(size_t)(inner ? ys : mis_reflect101(ys, h)) * w + x];
#pragma unroll
            for (int j = 0; j < BLK; j++)
                if (k - j >= 0 && k - j < N) acc[j] += t.k[k - j] * v;
        }
    } else {
        for (int k = 0; k < n + BLK - 1; k++) {
            const float v = src[(size_t)mis_reflect101(y0 + k - r, h) * w + x];
#pragma unroll
            for (int j = 0; j < BLK; j++)
                if (k - j >= 0 && k - j < n) acc[j] += t.k[k - j] * v;
        }
    }
#pragma unroll
    for (int j = 0; j < BLK; j++)
        if (y0 + j < h) {
            const size_t o = (size_t)(y0 + j) * w + x;
            dst[o] = acc[j];
            if (dog) dog[o] = acc[j] - prev[o];
        }
}

// Row and column pass of one GaussianBlur in ONE kernel (round 4): the two kernels above move 24 bytes per pixel and layer (source
// read, row results written and read back, the previous layer read again for the DoG, blurred layer and DoG written); fused, a
// layer is one read of the source (which IS the previous layer: the DoG's subtrahend comes with it) and the two writes.  A
// workgroup owns a strip of FB_TW columns and walks `seg` output rows downwards, FB_CH input rows at a time: the rows are
// staged in LDS, row-filtered (a thread: 8 consecutive outputs of one row from 8 + N - 1 LDS reads, indices padded as in the
// row kernel) into a ring of the last N + FB_CH - 1 filtered rows, and the column pass (thread = column, 8 output rows from
// N + 7 ring reads) emits the newest 8 complete rows.  Every sum is `acc = 0; acc += tap[k] * v` over ascending k, as in the
// separate passes (and the CPU path): the results are the same bits.  Vertically a strip costs N - 1 extra input rows per
// segment; the next chunk's global loads are in flight while the current one is filtered.
constexpr int FB_TW = 256, FB_CH = 8;
typedef float fb_v2f __attribute__((ext_vector_type(2)));
// Eight consecutive outputs of an N-tap filter from the N + 7 inputs in(0) .. in(N + 6): out[j] = sum_k tap[k] in(j + k), every sum
// `acc = 0; acc += tap[k] * v` over ascending k.  Outputs 2i and 2i + 1 share the inputs 2i + 1 .. 2i + N - 1 with taps one apart:
// those N - 1 steps are one packed multiply and one packed add for the pair (v_pk_mul_f32 / v_pk_add_f32: the full f32 rate of a
// SIMD needs packed operations -- as separate v_mul / v_add the passes held the vector unit for half of the kernel's time,
// tools/pmc_sift.sh); output 2i's first term and output 2i + 1's last one are scalar.  The additions of every output are the same,
// in the same order.
template <int N, typename In>
__device__ __forceinline__ void fb_filter8(const float (&tk)[N], In in, float (&out)[8]) {
    fb_v2f a2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) a2[i] = fb_v2f{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < N + 7; k++) {
        const float v = in(k);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int d = k - 2 * i;
            if (d == 0) a2[i].x += tk[0] * v;
            else if (d >= 1 && d <= N - 1) a2[i] += fb_v2f{tk[d], tk[d - 1]} * fb_v2f{v, v};
            else if (d == N) a2[i].y += tk[N - 1] * v;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { out[2 * i] = a2[i].x; out[2 * i + 1] = a2[i].y; }
}
template <int N>
__global__ __launch_bounds__(256) void sift_blur_fused_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog, int w, int h, int seg, Taps t) {
    constexpr int R = N / 2, RB = N + FB_CH - 1, SPAN = FB_TW + N - 1;
    constexpr int RPITCH = FB_TW + FB_TW / 8, SPITCH = SPAN + SPAN / 8 + 1;
    __shared__ float ring[RB * RPITCH];
    __shared__ float stage[FB_CH * SPITCH];
    const int tid = threadIdx.x, x0 = blockIdx.x * FB_TW, y0 = blockIdx.y * seg;
    const int rows_out = min(seg, h - y0);
    const int nchunks = (rows_out + N - 1 + FB_CH - 1) / FB_CH;
    const bool xin = x0 - R >= 0 && x0 + FB_TW - 1 + R < w;      // no horizontal reflection in this strip
    float tk[N];
#pragma unroll
    for (int k = 0; k < N; k++) tk[k] = t.k[k];
    // a chunk's FB_CH input rows: thread `tid` takes columns tid and (the first N - 1 threads) FB_TW + tid of every row
    const int xa = xin ? x0 - R + tid : mis_reflect101(x0 - R + tid, w);
    const int xb = tid < N - 1 ? (xin ? x0 - R + FB_TW + tid : mis_reflect101(x0 - R + FB_TW + tid, w)) : xa;
    const int x = x0 + tid, xc = min(x, w - 1);
    float pa[FB_CH], pb[FB_CH], ctr[FB_CH];      // the next chunk's inputs and the DoG's subtrahends of its outputs, in flight during the passes
    auto load_chunk = [&](int c) {
        const int yb = y0 - R + FB_CH * c;
        if (yb >= 0 && yb + FB_CH - 1 < h) {      // (one uniform branch per chunk: the reflected rows' index arithmetic stays out of the common path)
            const float* ra = src + (size_t)yb * w;
#pragma unroll
            for (int j = 0; j < FB_CH; j++) { pa[j] = ra[(size_t)j * w + xa]; pb[j] = tid < N - 1 ? ra[(size_t)j * w + xb] : 0.f; }
        } else {
#pragma unroll
            for (int j = 0; j < FB_CH; j++) {
                const size_t ro = (size_t)mis_reflect101(yb + j, h) * w;
                pa[j] = src[ro + xa];
                pb[j] = tid < N - 1 ? src[ro + xb] : 0.f;
            }
        }
        if (dog) {
            const int m0 = FB_CH * c + 1 - N;
            if (m0 >= 0 && m0 + FB_CH - 1 < rows_out) {
                const float* rc = src + (size_t)(y0 + m0) * w + xc;
#pragma unroll
                for (int j = 0; j < FB_CH; j++) ctr[j] = rc[(size_t)j * w];
            } else {
#pragma unroll
                for (int j = 0; j < FB_CH; j++) { const int m = min(max(m0 + j, 0), rows_out - 1); ctr[j] = src[(size_t)(y0 + m) * w + xc]; }
            }
        }
    };
    load_chunk(0);
    const int rj = tid >> 5, rx = tid & 31;       // row pass: row of the chunk, block of 8 columns
    for (int c = 0; c < nchunks; c++) {
        mis_lds_barrier();     // the previous chunk's passes are done with the staged rows and the ring's oldest rows (LDS-only barriers: the
                               // next chunk's loads and this chunk's stores stay in flight across them)
#pragma unroll
        for (int j = 0; j < FB_CH; j++) {
            stage[j * SPITCH + tid + (tid >> 3)] = pa[j];
            if (tid < N - 1) stage[j * SPITCH + (FB_TW + tid) + ((FB_TW + tid) >> 3)] = pb[j];
        }
        float cc[FB_CH];
#pragma unroll
        for (int j = 0; j < FB_CH; j++) cc[j] = ctr[j];
        mis_lds_barrier();
        if (c + 1 < nchunks) load_chunk(c + 1);
        {   // row pass: filtered row q = FB_CH * c + rj (relative to y0 - R) into its ring slot
            float acc[8];
            const float* my = stage + rj * SPITCH + 9 * rx;
            fb_filter8<N>(tk, [&](int k) { return my[k + (k >> 3)]; }, acc);
            float* o = ring + ((FB_CH * c + rj) % RB) * RPITCH + 9 * rx;
#pragma unroll
            for (int jj = 0; jj < 8; jj++) o[jj] = acc[jj];
        }
        mis_lds_barrier();
        // column pass: output rows m0 .. m0 + 7 (relative to y0), from the filtered rows m0 .. m0 + N + 6
        const int m0 = FB_CH * c + 1 - N;       // the newest complete output row is FB_CH * c + FB_CH - N: the eight rows that end there
        if (m0 + 7 >= 0 && m0 < rows_out) {
            float acc[8];
            const int slot0 = ((m0 % RB) + RB) % RB;
            const int col = tid + (tid >> 3);
            // rows above the segment's first input row only meet outputs that are not emitted
            fb_filter8<N>(tk, [&](int k) { const int sl = slot0 + k < RB ? slot0 + k : slot0 + k - RB; return m0 + k >= 0 ? ring[sl * RPITCH + col] : 0.f; }, acc);
            if (x < w) {
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    const int m = m0 + jj;
                    if (m < 0 || m >= rows_out) continue;
                    const size_t o = (size_t)(y0 + m) * w + x;
                    dst[o] = acc[jj];
                    if (dog) dog[o] = acc[jj] - cc[jj];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void sift_sub_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = b[i] - a[i];
}

__global__ __launch_bounds__(256) void sift_decimate_kernel(const float* __restrict__ src, int sw, int sh, float* __restrict__ dst, int w, int h) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int sy = y * 2 < sh ? y * 2 : sh - 1, sx = x * 2 < sw ? x * 2 : sw - 1;
    dst[(size_t)y * w + x] = src[(size_t)sy * sw + sx];
}

#define AT(img, r, c) ((img)[(size_t)(r) * w + (c)])

// findScaleSpaceExtrema, the scan.  "val >= all 26 neighbours" is "val >= the maximum of the 3 x 3 x 3 block" (the block
// contains val), and that maximum is separable: a thread owns one column and walks EXT_ROWS rows; per DoG layer it
// keeps the horizontal 3-max / 3-min of the previous two rows, so a new row costs three loads and four max3 / min3 per
// layer, and the three scale layers share the per-layer 3 x 3 results (15 loads per pixel for all three layers where the
// direct test needs 81).  Candidates are collected per workgroup in LDS and appended with one global atomic (same-address
// global atomics serialise at ~11 ns each).  The list order is arbitrary: the keypoints are sorted into a total order later.
constexpr int EXT_ROWS = 32, EXT_LIST = 1024;
template <int NL>
__global__ __launch_bounds__(256) void sift_extrema_kernel(Pyr P, int o, int threshold, int4* __restrict__ cand, unsigned* __restrict__ n_cand, unsigned cap) {
    constexpr int ND = NL + 2;
    __shared__ int4 list[EXT_LIST];
    __shared__ unsigned cnt, base;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int w = P.w[o], h = P.h[o];
    const int c = blockIdx.x * 256 + threadIdx.x + SIFT_IMG_BORDER, r0 = blockIdx.y * EXT_ROWS + SIFT_IMG_BORDER, r1 = min(r0 + EXT_ROWS, h - SIFT_IMG_BORDER);
    if (c < w - SIFT_IMG_BORDER) {
        const float thr = (float)threshold;
        float amx[ND], amn[ND], bmx[ND], bmn[ND], cen[ND];
#pragma unroll
        for (int l = 0; l < ND; l++) amx[l] = amn[l] = bmx[l] = bmn[l] = cen[l] = 0.f;
        for (int rr = r0 - 1; rr <= r1; rr++) {
            float nmx[ND], nmn[ND], ncen[ND];
#pragma unroll
            for (int l = 0; l < ND; l++) {
                const float* row = P.dog[o * ND + l] + (size_t)rr * w + c;
                const float a = row[-1], b = row[0], d = row[1];
                nmx[l] = fmaxf(fmaxf(a, b), d); nmn[l] = fminf(fminf(a, b), d); ncen[l] = b;
            }
            if (rr > r0) {   // rows rr - 2, rr - 1, rr are in a / b / n: evaluate row rr - 1
                float vmx[ND], vmn[ND];
#pragma unroll
                for (int l = 0; l < ND; l++) { vmx[l] = fmaxf(fmaxf(amx[l], bmx[l]), nmx[l]); vmn[l] = fminf(fminf(amn[l], bmn[l]), nmn[l]); }
#pragma unroll
                for (int l = 1; l <= NL; l++) {
                    const float val = cen[l];
                    if (!(fabsf(val) > thr)) continue;
                    const bool ext = val > 0 ? val >= fmaxf(fmaxf(vmx[l - 1], vmx[l]), vmx[l + 1]) : val <= fminf(fminf(vmn[l - 1], vmn[l]), vmn[l + 1]);
                    if (!ext) continue;
                    const unsigned slot = atomicAdd(&cnt, 1u);
                    if (slot < EXT_LIST) list[slot] = make_int4(o, l, rr - 1, c);
                    else {   // a pathological tile: append directly
                        const unsigned g = atomicAdd(n_cand, 1u);
                        if (g < cap) cand[g] = make_int4(o, l, rr - 1, c);
                    }
                }
            }
#pragma unroll
            for (int l = 0; l < ND; l++) { amx[l] = bmx[l]; amn[l] = bmn[l]; bmx[l] = nmx[l]; bmn[l] = nmn[l]; cen[l] = ncen[l]; }
        }
    }
    __syncthreads();
    const unsigned m = min(cnt, (unsigned)EXT_LIST);
    if (threadIdx.x == 0 && m) base = atomicAdd(n_cand, m);
    __syncthreads();
    for (unsigned i = threadIdx.x; i < m; i += 256)
        if (base + i < cap) cand[base + i] = list[i];
}

// any other number of octave layers: the direct 26-neighbour test, layer = blockIdx.z + 1, one thread per pixel
__global__ __launch_bounds__(256) void sift_extrema_generic_kernel(Pyr P, int o, int threshold, int4* __restrict__ cand, unsigned* __restrict__ n_cand, unsigned cap) {
    const int w = P.w[o], h = P.h[o], nl = P.nl;
    const int c = blockIdx.x * 256 + threadIdx.x + SIFT_IMG_BORDER, r = blockIdx.y + SIFT_IMG_BORDER, layer = blockIdx.z + 1;
    if (c >= w - SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER) return;
    const float* img = P.dog[o * (nl + 2) + layer];
    const float* prev = P.dog[o * (nl + 2) + layer - 1];
    const float* next = P.dog[o * (nl + 2) + layer + 1];
    const float val = AT(img, r, c);
    if (!(fabsf(val) > (float)threshold)) return;
    bool ext = true;
    if (val > 0) {
        for (int dr = -1; dr <= 1; dr++)
            for (int dc = -1; dc <= 1; dc++)
                ext = ext && val >= AT(img, r + dr, c + dc) && val >= AT(prev, r + dr, c + dc) && val >= AT(next, r + dr, c + dc);
    } else {
        for (int dr = -1; dr <= 1; dr++)
            for (int dc = -1; dc <= 1; dc++)
                ext = ext && val <= AT(img, r + dr, c + dc) && val <= AT(prev, r + dr, c + dc) && val <= AT(next, r + dr, c + dc);
    }
    if (!ext) return;
    const unsigned slot = atomicAdd(n_cand, 1u);
    if (slot < cap) cand[slot] = make_int4(o, layer, r, c);
}

// Matx33f::solve(DECOMP_LU): closed form through the determinant
__device__ __forceinline__ bool solve3(const float* a, const float* b, float* x) {
    float d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (d == 0) return false;
    d = 1 / d;
    x[0] = d * (b[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (b[1] * a[8] - a[5] * b[2]) + a[2] * (b[1] * a[7] - a[4] * b[2]));
    x[1] = d * (a[0] * (b[1] * a[8] - a[5] * b[2]) - b[0] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * b[2] - b[1] * a[6]));
    x[2] = d * (a[0] * (a[4] * b[2] - b[1] * a[7]) - a[1] * (a[3] * b[2] - b[1] * a[6]) + b[0] * (a[3] * a[7] - a[4] * a[6]));
    return true;
}

struct SiftConsts {
    float contrast_threshold, edge_threshold, sigma;
};

// adjustLocalExtrema + calcOrientationHist + the peak loop of findScaleSpaceExtrema: one thread per candidate
// adjustLocalExtrema of one candidate (a thread): false = rejected; otherwise the keypoint without its orientation, and where it sits
struct SiftSurv { MisKeyPoint kpt; int octv, layer, r, c; };
__device__ bool sift_refine_one(const Pyr& P, const SiftConsts& K, const int4 cd, SiftSurv* out) {
    const int octv = cd.x, nl = P.nl, w = P.w[octv], h = P.h[octv];
    int layer = cd.y, r = cd.z, c = cd.w;
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int i = 0;
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const float* img = P.dog[octv * (nl + 2) + layer];
        const float* prev = P.dog[octv * (nl + 2) + layer - 1];
        const float* next = P.dog[octv * (nl + 2) + layer + 1];
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        const float v2 = AT(img, r, c) * 2;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dss = (AT(next, r, c) + AT(prev, r, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float dxs = (AT(next, r, c + 1) - AT(next, r, c - 1) - AT(prev, r, c + 1) + AT(prev, r, c - 1)) * cross_deriv_scale;
        const float dys = (AT(next, r + 1, c) - AT(next, r - 1, c) - AT(prev, r + 1, c) + AT(prev, r - 1, c)) * cross_deriv_scale;
        const float H[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss};
        float X[3] = {0, 0, 0};
        solve3(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT_MAX / 3) || fabsf(xr) > (float)(INT_MAX / 3) || fabsf(xc) > (float)(INT_MAX / 3)) return false;
        c += mis_round_f(xc); r += mis_round_f(xr); layer += mis_round_f(xi);
        if (layer < 1 || layer > nl || c < SIFT_IMG_BORDER || c >= w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER) return false;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return false;
    {
        const float* img = P.dog[octv * (nl + 2) + layer];
        const float* prev = P.dog[octv * (nl + 2) + layer - 1];
        const float* next = P.dog[octv * (nl + 2) + layer + 1];
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        const float t = (dD[0] * xc + dD[1] * xr) + dD[2] * xi;
        contr = AT(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nl < K.contrast_threshold) return false;
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy, et = K.edge_threshold;
        if (det <= 0 || tr * tr * et >= (et + 1) * (et + 1) * det) return false;
    }
    MisKeyPoint kpt;
    kpt.x = (c + xc) * (1 << octv);
    kpt.y = (r + xr) * (1 << octv);
    kpt.octave = octv + (layer << 8) + (mis_round_f((xi + 0.5f) * 255) << 16);
    kpt.size = K.sigma * mis_expf(((layer + xi) / nl) * 0.69314718055994530942f) * (1 << octv) * 2;
    kpt.response = fabsf(contr);
    kpt.angle = 0;
    out->kpt = kpt; out->octv = octv; out->layer = layer; out->r = r; out->c = c;
    return true;
}

// Round 4: the refinement used to go on, per THREAD, into calcOrientationHist -- a (2 radius + 1)^2 window of gradient samples added
// into a 36-bin histogram that lived in scratch memory (a read-modify-write of private memory per sample), run by the few lanes of a
// wave whose candidates survived the contrast and edge tests: 1.1 ms per 8K frame.  Now the survivors are listed and a WAVE takes
// one: the samples are evaluated 64 at a time (exp weight, atan2, magnitude), and lane b < 36 keeps bin b in a register, adding
// the samples of its bin in window order (the sums are order dependent; the order is the CPU loop's).
__global__ __launch_bounds__(64) void sift_refine_kernel(Pyr P, SiftConsts K, const int4* __restrict__ cand, const unsigned* __restrict__ n_cand, unsigned cand_cap,
                                                        SiftSurv* __restrict__ surv, unsigned* __restrict__ n_surv, unsigned surv_cap) {
    const unsigned nc = min(*n_cand, cand_cap);
    const int lane = threadIdx.x & 63;
    for (unsigned q0 = blockIdx.x * 64; q0 < nc; q0 += gridDim.x * 64) {      // (the wave stays together: the list append is a wave operation)
        const unsigned q = q0 + threadIdx.x;
        SiftSurv sv;
        const bool ok = q < nc && sift_refine_one(P, K, cand[q], &sv);
        const unsigned long long m = __ballot(ok);
        if (!m) continue;
        const int leader = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(n_surv, (unsigned)__popcll(m));
        base = __shfl(base, leader);
        if (ok) {
            const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (slot < surv_cap) surv[slot] = sv;
        }
    }
}

// calcOrientationHist + the peaks of the smoothed histogram: one wave per surviving candidate (grid-stride over the list)
__global__ __launch_bounds__(64) void sift_orient_kernel(Pyr P, const SiftSurv* __restrict__ surv, const unsigned* __restrict__ n_surv, unsigned surv_cap,
                                                        MisKeyPoint* __restrict__ kps, unsigned* __restrict__ n_kps, unsigned kp_cap) {
    const unsigned ns = min(*n_surv, surv_cap);
    const int lane = threadIdx.x, n = SIFT_ORI_HIST_BINS, nl = P.nl;
    for (unsigned q = blockIdx.x; q < ns; q += gridDim.x) {
        const SiftSurv sv = surv[q];
        MisKeyPoint kpt = sv.kpt;
        const int octv = sv.octv, r = sv.r, c = sv.c, w = P.w[octv], h = P.h[octv];
        const float scl_octv = kpt.size * 0.5f / (1 << octv);
        const int radius = mis_round_f(SIFT_ORI_RADIUS * scl_octv);
        const float sigma_o = SIFT_ORI_SIG_FCTR * scl_octv, expf_scale = -1.f / (2.f * sigma_o * sigma_o);
        const float* gimg = P.gauss[octv * (nl + 3) + sv.layer];
        float th = 0.f;      // bin `lane` of the raw histogram (lanes >= 36: unused)
        const int side = 2 * radius + 1, total = side * side;
        int wi = lane / side, wj = lane % side;
        const int adv_i = 64 / side, adv_j = 64 % side;
        for (int base = 0; base < total; base += 64) {
            const int di = wi - radius, dj = wj - radius, y = r + di, x = c + dj;
            const bool valid = wi < side && y > 0 && y < h - 1 && x > 0 && x < w - 1;
            float val = 0.f;
            int bin = 0;
            if (valid) {
                const float dx = AT(gimg, y, x + 1) - AT(gimg, y, x - 1), dy = AT(gimg, y - 1, x) - AT(gimg, y + 1, x);
                const float wgt = mis_expf((di * di + dj * dj) * expf_scale);
                const float ori = mis_fast_atan2(dy, dx), mag = sqrtf(dx * dx + dy * dy);
                bin = mis_round_f((n / 360.f) * ori);
                if (bin >= n) bin -= n;
                if (bin < 0) bin += n;
                val = wgt * mag;
            }
            // the samples of this trip in window order: sample s goes to the lane of its bin
            unsigned long long m = __ballot(valid);
            while (m) {
                const int sl = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int b = __builtin_amdgcn_readlane(bin, sl);
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), sl));
                if (lane == b) th += v;
            }
            wi += adv_i; wj += adv_j;
            if (wj >= side) { wj -= side; wi++; }
        }
        // smoothing over the circular histogram, its maximum, the peaks
        const int jl = lane < n ? lane : 0;
        const float tm2 = __shfl(th, (jl + n - 2) % n), tm1 = __shfl(th, (jl + n - 1) % n), tp1 = __shfl(th, (jl + 1) % n), tp2 = __shfl(th, (jl + 2) % n);
        const float hj = (tm2 + tp2) * (1.f / 16.f) + (tm1 + tp1) * (4.f / 16.f) + th * (6.f / 16.f);
        float omax = lane < n ? hj : -FLT_MAX;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(omax, o); omax = omax > ov ? omax : ov; }
        const float mag_thr = omax * SIFT_ORI_PEAK_RATIO;
        const float hl = __shfl(hj, (jl + n - 1) % n), hr = __shfl(hj, (jl + 1) % n);
        const bool peak = lane < n && hj > hl && hj > hr && hj >= mag_thr;
        const unsigned long long pm = __ballot(peak);
        if (!pm) continue;
        const int leader = __ffsll((long long)pm) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(n_kps, (unsigned)__popcll(pm));
        base = __shfl(base, leader);
        if (peak) {
            float bin = lane + 0.5f * (hl - hr) / (hl - 2 * hj + hr);
            bin = bin < 0 ? n + bin : (bin >= n ? bin - n : bin);
            kpt.angle = 360.f - (float)((360.f / n) * bin);
            if (fabsf(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
            const unsigned slot = base + (unsigned)__popcll(pm & ((1ull << lane) - 1ull));
            if (slot < kp_cap) kps[slot] = kpt;
        }
    }
}

// calcSIFTDescriptor: one wave per keypoint.  The float sums into the (d+2)(d+2)(n+2) histogram are order
// dependent, so the CPU's order is kept: the samples of the (2 radius + 1)^2 window are visited 64 at a time in raster order, and
// the ones that fall into the rotated 4 x 4 grid (about a third) are queued in that order; whenever 64 are queued, one lane each
// evaluates them (gradient, exp weight, atan2, trilinear split -- the expensive part now runs on full waves; evaluated in place
// it ran with a third of its lanes, and the window index was a 64-bit division per sample) and then their contributions are added
// one sample after the other.  The histogram lives in REGISTERS during that walk: lane L < 36 owns the spatial cell (L / 6, L % 6)
// and its ten orientation slots; a sample touches 2 x 2 cells x 2 neighbouring slots, its first slot o0 is the same for all
// lanes (a scalar switch picks the two registers), and the lanes of the four cells each read their pair of values.  Every bin
// still receives its contributions in sample order.  (As read-modify-writes of an LDS histogram -- plain, or ds_add_f32 atomics
// -- the walk kept the CU's one LDS pipeline busy for 88 % of the kernel: 8 lane-operations per sample at ~8 cycles each.)
constexpr int HISTLEN = 6 * 6 * 10;
__global__ __launch_bounds__(64) void sift_descriptor_kernel(Pyr P, const MisKeyPoint* __restrict__ kps, const unsigned* __restrict__ nk, float* __restrict__ desc) {
    __shared__ float hist[HISTLEN];
    __shared__ int q_ij[128];          // ring of queued samples: (i << 16) | (j & 0xffff)
    __shared__ __attribute__((aligned(16))) int s_idx[64];
    __shared__ __attribute__((aligned(16))) float s_val[64 * 8 + 4];
    const int q = blockIdx.x, lane = threadIdx.x;
    if ((unsigned)q >= *nk) return;    // the grid covers the raw keypoints; duplicates were removed on the device
    const int d = 4, n = 8, nl = P.nl, firstOctave = -1;
    const MisKeyPoint k = kps[q];
    int octave = k.octave & 255;
    const int layer = (k.octave >> 8) & 255;
    octave = octave < 128 ? octave : (-128 | octave);
    const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
    const float size = k.size * scale, ptx = k.x * scale, pty = k.y * scale;
    const int oi = octave - firstOctave;
    float ori = 360.f - k.angle;
    if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
    const float scl = size * 0.5f;
    const float* img = P.gauss[oi * (nl + 3) + layer];
    const int w = P.w[oi], h = P.h[oi];
    const int px = mis_round_f(ptx), py = mis_round_f(pty);
    float cos_t = mis_cosf(ori * (float)(3.14159265358979323846 / 180)), sin_t = mis_sinf(ori * (float)(3.14159265358979323846 / 180));
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = mis_round_f(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt((double)w * w + (double)h * h);
    if (radius > rmax) radius = rmax;      // (< 2^15 for every image this library accepts: i and j fit 16 bits)
    cos_t /= hist_width; sin_t /= hist_width;
    for (int e = lane; e < HISTLEN; e += 64) hist[e] = 0.f;
    __syncthreads();
    const int side = 2 * radius + 1;
    const long long total = (long long)side * side;
    // lane L < 36 is cell (L / 6, L % 6) of the (d + 2) x (d + 2) grid (lanes >= 36: none)
    typedef float desc_v16 __attribute__((ext_vector_type(16)));
    desc_v16 hv = 0.f;      // its orientation slots (0 .. 8 are used; the tenth stays zero)
    // the first `take` queued samples (from ring position qh): evaluate, then accumulate in order
    auto drain = [&](int qh, int take) __attribute__((always_inline)) {
        if (lane < take) {
            const int e = q_ij[(qh + lane) & 127];
            const int i = e >> 16, j = (int)(short)(e & 0xffff);
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            const float dx = AT(img, r, c + 1) - AT(img, r, c - 1), dy = AT(img, r - 1, c) - AT(img, r + 1, c);
            const float wgt = mis_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (mis_fast_atan2(dy, dx) - ori) * bins_per_rad;
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            const int r0 = mis_floor_f(rbin), c0 = mis_floor_f(cbin);
            int o0 = mis_floor_f(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            float v[8];
            v[7] = v_rc11 * obin; v[6] = v_rc11 - v[7]; v[5] = v_rc10 * obin; v[4] = v_rc10 - v[5];
            v[3] = v_rc01 * obin; v[2] = v_rc01 - v[3]; v[1] = v_rc00 * obin; v[0] = v_rc00 - v[1];
            s_idx[lane] = ((r0 + 1) << 8) | ((c0 + 1) << 4) | o0;      // first cell (row, column) and first orientation slot
#pragma unroll
            for (int c8 = 0; c8 < 8; c8++) s_val[lane * 8 + c8] = v[c8];
        }
        __syncthreads();
        // (e is the sample's packed {row, column, slot}: the same in every lane; v: this lane's pair of values, if its cell is one of the
        // four.  The slots are one vector-typed value indexed by a wave-uniform number: register-relative moves; as a private array, or
        // as named scalars behind a switch -- which the compiler turns back into an array --, the histogram went to scratch memory)
        // (the sample's first cell as a lane number, Lb = 6 row + column, is a scalar: the four cells are lanes Lb + {0, 1, 6, 7})
#define MIS_DESC_LB(e) (((e) >> 8) * (d + 2) + (((e) >> 4) & 15))
#define MIS_DESC_ADD(e, v)                                                                          \
        do {                                                                                        \
            const unsigned dd_ = (unsigned)(lane - MIS_DESC_LB(e));                                 \
            if (dd_ < 8u && ((0xC3u >> dd_) & 1u)) {                                                \
                const int o_ = (e) & 15;                                                            \
                hv[o_] += (v).x;                                                                    \
                hv[o_ + 1] += (v).y;                                                                \
            }                                                                                       \
        } while (0)
        // this lane's two values of sample sI (corner (dr, dc): values 4 dr + 2 dc, + 1); anything in range for the other lanes
#define MIS_DESC_PAIR(sI_, e) (*reinterpret_cast<const float2*>(s_val + (sI_) * 8 + 2 * (((lane - MIS_DESC_LB(e)) & 1) | (((lane - MIS_DESC_LB(e)) >> 1) & 2))))
        int sI = 0;
#pragma unroll 1
        for (; sI + 4 <= take; sI += 4) {      // four samples' reads in flight
            const int4 e4v = *reinterpret_cast<const int4*>(s_idx + sI);
            int4 e4;      // the same in every lane: scalars
            e4.x = __builtin_amdgcn_readfirstlane(e4v.x); e4.y = __builtin_amdgcn_readfirstlane(e4v.y);
            e4.z = __builtin_amdgcn_readfirstlane(e4v.z); e4.w = __builtin_amdgcn_readfirstlane(e4v.w);
            const float2 v0 = MIS_DESC_PAIR(sI, e4.x), v1 = MIS_DESC_PAIR(sI + 1, e4.y), v2 = MIS_DESC_PAIR(sI + 2, e4.z), v3 = MIS_DESC_PAIR(sI + 3, e4.w);
            MIS_DESC_ADD(e4.x, v0); MIS_DESC_ADD(e4.y, v1); MIS_DESC_ADD(e4.z, v2); MIS_DESC_ADD(e4.w, v3);
        }
        for (; sI < take; sI++) {
            const int e = __builtin_amdgcn_readfirstlane(s_idx[sI]);
            const float2 v = MIS_DESC_PAIR(sI, e);
            MIS_DESC_ADD(e, v);
        }
        __syncthreads();
    };
    // lane's window position (i, j) = (kk / side - radius, kk % side - radius) of kk = base + lane, advanced by 64 per trip
    int wi = lane / side, wj = lane % side;
    const int adv_i = 64 / side, adv_j = 64 % side;
    int qh = 0, qn = 0;      // ring head, queued samples (wave-uniform)
    for (long long base = 0; base < total; base += 64) {
        bool valid = false;
        const int i = wi - radius, j = wj - radius;
        if (wi < side) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            const float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            valid = rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1;
        }
        const unsigned long long bal = __ballot(valid);
        if (valid) q_ij[(qh + qn + __popcll(bal & ((1ull << lane) - 1ull))) & 127] = (i << 16) | (j & 0xffff);
        qn += __popcll(bal);
        wi += adv_i; wj += adv_j;
        if (wj >= side) { wj -= side; wi++; }
        if (qn >= 64) {
            __syncthreads();
            drain(qh, 64);
            qh = (qh + 64) & 127; qn -= 64;
        }
    }
    __syncthreads();
    if (qn > 0) drain(qh, qn);
    if (lane < (d + 2) * (d + 2)) {
        float* o = hist + lane * (n + 2);
#pragma unroll
        for (int k8 = 0; k8 < n + 1; k8++) o[k8] = hv[k8];
        o[n + 1] = 0.f;
    }
    __syncthreads();
    // The histogram -> descriptor epilogue.  One lane used to do all of it THROUGH the output row in global memory (written, read back
    // for the threshold pass, written, read back for the scaling: ~400 dependent global round trips per keypoint -- most of the
    // kernel's 1.9 ms per 8K frame).  Now: the circular bins folded by 16 lanes, element e = (i d + j) n + k handled by lane e & 63,
    // and only the two sums of squares -- whose order the result depends on -- are walked by one lane, from LDS.
    if (lane < d * d) {
        const int idx = ((lane / d + 1) * (d + 2) + (lane % d + 1)) * (n + 2);
        hist[idx] += hist[idx + n];
        hist[idx + 1] += hist[idx + n + 1];
    }
    __syncthreads();
    float val[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int e = lane + 64 * u;
        val[u] = hist[((e >> 5) + 1) * (d + 2) * (n + 2) + (((e >> 3) & 3) + 1) * (n + 2) + (e & 7)];
        s_val[e] = val[u] * val[u];
    }
    auto ordered_sum128 = [&]() {      // sum of s_val[0 .. 127] in index order, by lane 0; every lane returns it
        __syncthreads();
        if (lane == 0) {
            float acc = 0;
#pragma unroll 1
            for (int e = 0; e < 128; e += 8) {
                float t8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t8[u] = s_val[e + u];
#pragma unroll
                for (int u = 0; u < 8; u++) acc += t8[u];
            }
            s_val[128] = acc;
        }
        __syncthreads();
        return s_val[128];
    };
    const float thr = sqrtf(ordered_sum128()) * SIFT_DESCR_MAG_THR;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        val[u] = val[u] < thr ? val[u] : thr;
        s_val[lane + 64 * u] = val[u] * val[u];
    }
    const float root = sqrtf(ordered_sum128());
    const float fct = SIFT_INT_DESCR_FCTR / (root > FLT_EPSILON ? root : FLT_EPSILON);
    float* dst = desc + 128 * (size_t)q;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int v = mis_round_f(val[u] * fct);   // saturate_cast<uchar>
        dst[lane + 64 * u] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// ---- KeyPointsFilter::removeDuplicatedSorted on the device ----
// The order of KeyPoint_LessThan (x, y ascending; size descending; angle ascending; response, octave descending) as three
// 64-bit keys whose unsigned lexicographic order is that order (floats through the usual order-preserving bit map; none
// of the fields can be -0 or NaN here).  Rank sort: element i goes to position #{j : j before i}; ties (identical
// keypoints) are broken by the list index.  i is a duplicate when an element with the same (x, y, size, angle) precedes it.
__device__ __forceinline__ unsigned ford(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
struct KpKey { unsigned long long a, b, c; };
__device__ __forceinline__ KpKey kp_key(const MisKeyPoint& k) {
    KpKey q;
    q.a = ((unsigned long long)ford(k.x) << 32) | ford(k.y);
    q.b = ((unsigned long long)(~ford(k.size)) << 32) | ford(k.angle);
    q.c = ((unsigned long long)(~ford(k.response)) << 32) | (~((unsigned)k.octave ^ 0x80000000u));
    return q;
}
constexpr int RANK_SLICES = 8;
__global__ __launch_bounds__(256) void sift_rank_kernel(const MisKeyPoint* __restrict__ raw, const unsigned* __restrict__ n_raw, unsigned cap, unsigned* __restrict__ rank_out,
                                                        unsigned* __restrict__ dup_out) {
    __shared__ unsigned long long sa[256], sb[256], sc[256];
    const unsigned n = min(*n_raw, cap), i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    KpKey me{~0ull, ~0ull, ~0ull};
    if (i < n) me = kp_key(raw[i]);
    unsigned rank = 0;
    bool isdup = false;
    // this workgroup's slice of the list (multiples of 256)
    const unsigned tiles = (n + 255) / 256, per = (tiles + RANK_SLICES - 1) / RANK_SLICES;
    const unsigned jb = blockIdx.y * per * 256, je = min(n, (blockIdx.y + 1) * per * 256);
    for (unsigned base = jb; base < je; base += 256) {
        const unsigned j = base + threadIdx.x;
        KpKey kj{~0ull, ~0ull, ~0ull};
        if (j < n) kj = kp_key(raw[j]);
        __syncthreads();
        sa[threadIdx.x] = kj.a; sb[threadIdx.x] = kj.b; sc[threadIdx.x] = kj.c;
        __syncthreads();
        const unsigned m = min(256u, n - base);
        for (unsigned jj = 0; jj < m; jj++) {
            const unsigned long long a = sa[jj], b = sb[jj], c = sc[jj];
            const bool same4 = a == me.a && b == me.b;
            const bool lt = a < me.a || (a == me.a && (b < me.b || (b == me.b && (c < me.c || (c == me.c && base + jj < i)))));
            rank += lt;
            isdup |= lt && same4;
        }
    }
    if (i < n) {
        if (rank) atomicAdd(&rank_out[i], rank);
        if (isdup) atomicOr(&dup_out[i], 1u);
    }
}
__global__ __launch_bounds__(256) void sift_scatter_kernel(const unsigned* __restrict__ n_raw, unsigned cap, const unsigned* __restrict__ rank, const unsigned* __restrict__ dupi,
                                                           unsigned* __restrict__ perm, uint8_t* __restrict__ dup) {
    const unsigned n = min(*n_raw, cap), i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { perm[rank[i]] = i; dup[rank[i]] = (uint8_t)dupi[i]; }
}
// stable compaction of the sorted list (one workgroup walks it 1024 at a time) + the firstOctave = -1 rescaling of
// SIFT::detectAndCompute; writes the final count
__global__ __launch_bounds__(1024) void sift_compact_kernel(const MisKeyPoint* __restrict__ raw, const unsigned* __restrict__ n_raw, unsigned cap,
                                                            const unsigned* __restrict__ perm, const uint8_t* __restrict__ dup, int firstOctave,
                                                            MisKeyPoint* __restrict__ out, unsigned* __restrict__ nk) {
    __shared__ unsigned wsum[16];
    const unsigned n = min(*n_raw, cap), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned running = 0;
    for (unsigned base = 0; base < n; base += 1024) {
        const unsigned p = base + threadIdx.x;
        const bool keep = p < n && !dup[p];
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) wsum[wave] = __popcll(mask);
        __syncthreads();
        unsigned before = 0, total = 0;
        for (unsigned k = 0; k < 16; k++) { const unsigned v = wsum[k]; total += v; before += k < wave ? v : 0; }
        if (keep) {
            MisKeyPoint k = raw[perm[p]];
            const float scale = 1.f / (float)(1 << -firstOctave);
            k.octave = (k.octave & ~255) | ((k.octave + firstOctave) & 255);
            k.x *= scale; k.y *= scale; k.size *= scale;
            out[running + before + __popcll(mask & ((1ull << lane) - 1))] = k;
        }
        running += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *nk = running;
}

}  // namespace

struct MisSift {
    MisContext* ctx;
    MisSiftParams p;
    int max_w, max_h;
    Pyr pyr;            // for the current image size
    int cur_w = 0, cur_h = 0;
    uint8_t* mem = nullptr;   // one allocation: pyramids, scratch, lists
    size_t bytes = 0;
    uint8_t* gray = nullptr;
    float *tmp = nullptr, *grayf = nullptr;
    int4* cand = nullptr;
    MisKeyPoint* raw = nullptr;
    unsigned* counters = nullptr;   // [0] candidates, [1] raw keypoints, [2] keypoints after duplicate removal, [3] candidates that survived the refinement
    SiftSurv* surv = nullptr;       // ... those candidates (kp_cap entries)
    unsigned* perm = nullptr;       // sorted position -> raw index
    uint8_t* dup = nullptr;         // sorted position -> duplicate of an earlier one
    unsigned* rank = nullptr;       // raw index -> sorted position, and raw index -> duplicate flag (2 x kp_cap, zeroed per frame)
    unsigned cand_cap = 0, kp_cap = 0;
    double sig[MAX_LAYERS + 4];
    // second lane of mis_sift_detect_batch: a finder of its own on its own context / stream, driven by a host thread
    MisContext* pool_ctx = nullptr;       // a helper lane's output blocks come from (and go back to) the parent's context
    static constexpr int MAX_LANES = 4;
    MisSift* helper[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};          // [k], k >= 1: lane k of mis_sift_detect_batch
    MisContext* helper_ctx[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    void* helper_stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};       // streams this finder created (lanes beyond the context's two auxiliary streams)
};

static int sift_plan(MisSift* s, int w, int h) {
    // carve the one allocation for this image size (allocation sized for max_w x max_h)
    const int nl = s->p.n_octave_layers, firstOctave = -1;
    Pyr& P = s->pyr;
    const int bw = 2 * w, bh = 2 * h;
    int noct = mis_round_d(log((double)std::min(bw, bh)) / log(2.) - 2) - firstOctave;
    noct = std::max(1, std::min(noct, MAX_OCT));
    P.nl = nl;
    size_t off = 0;
    auto carve = [&](size_t b) { size_t o = off; off += mis_align_up(b, 256); return s->mem ? s->mem + o : (uint8_t*)nullptr; };
    int cw = bw, ch = bh, built = 0;
    for (int o = 0; o < noct; o++) {
        P.w[o] = cw; P.h[o] = ch;
        for (int i = 0; i < nl + 3; i++) P.gauss[o * (nl + 3) + i] = (float*)carve(sizeof(float) * (size_t)cw * ch);
        for (int i = 0; i < nl + 2; i++) P.dog[o * (nl + 2) + i] = (float*)carve(sizeof(float) * (size_t)cw * ch);
        built = o + 1;
        cw /= 2; ch /= 2;
        if (cw < 1 || ch < 1) break;
    }
    P.noct = built;
    s->gray = carve((size_t)w * h);
    s->grayf = (float*)carve(sizeof(float) * (size_t)bw * bh);
    s->tmp = (float*)carve(sizeof(float) * (size_t)bw * bh);
    s->cand_cap = (unsigned)std::max<size_t>(65536, (size_t)bw * bh / 8);
    s->kp_cap = (unsigned)std::max<size_t>(65536, (size_t)bw * bh / 16);
    s->cand = (int4*)carve(sizeof(int4) * s->cand_cap);
    s->raw = (MisKeyPoint*)carve(sizeof(MisKeyPoint) * s->kp_cap);
    s->surv = (SiftSurv*)carve(sizeof(SiftSurv) * s->kp_cap);
    s->counters = (unsigned*)carve(256);
    s->perm = (unsigned*)carve(sizeof(unsigned) * s->kp_cap);
    s->dup = carve(s->kp_cap);
    s->rank = (unsigned*)carve(2 * sizeof(unsigned) * s->kp_cap);
    s->bytes = off;
    s->cur_w = w; s->cur_h = h;
    return MIS_OK;
}

extern "C" void mis_sift_default_params(MisSiftParams* p) {
    if (!p) return;
    p->nfeatures = 0; p->n_octave_layers = 3; p->contrast_threshold = 0.04; p->edge_threshold = 10; p->sigma = 1.6;
}

extern "C" int mis_sift_create(MisContext* ctx, const MisSiftParams* params, int max_width, int max_height, MisSift** out) {
    if (!ctx || !out) return MIS_E_INVALID;
    *out = nullptr;
    MisSiftParams p;
    mis_sift_default_params(&p);
    if (params) p = *params;
    MIS_CHECK(ctx, p.nfeatures == 0, MIS_E_UNSUPPORTED, "SIFT: only nfeatures = 0 (the reference's SIFT::create()) is supported");
    MIS_CHECK(ctx, p.n_octave_layers >= 1 && p.n_octave_layers <= MAX_LAYERS && p.sigma > 0.5, MIS_E_INVALID, "SIFT: bad parameters");
    MIS_CHECK(ctx, max_width >= 16 && max_height >= 16 && max_width <= 16384 && max_height <= 16384, MIS_E_INVALID, "SIFT: image size out of range");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MisSift* s = new MisSift();
    s->ctx = ctx; s->p = p; s->max_w = max_width; s->max_h = max_height;
    sift_plan(s, max_width, max_height);            // sizes only (mem == nullptr)
    const size_t need = s->bytes;
    if (hipMalloc((void**)&s->mem, need) != hipSuccess) { delete s; return mis_set_error(ctx, MIS_E_HIP, "SIFT: cannot allocate %zu bytes of scale space", need); }
    sift_plan(s, max_width, max_height);
    // buildGaussianPyramid: incremental sigmas
    const int nl = p.n_octave_layers;
    s->sig[0] = p.sigma;
    const double k = pow(2., 1. / nl);
    for (int i = 1; i < nl + 3; i++) {
        const double sig_prev = pow(k, (double)(i - 1)) * p.sigma, sig_total = sig_prev * k;
        s->sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
    }
    *out = s;
    return MIS_OK;
}

extern "C" int mis_sift_destroy(MisSift* s) {
    if (!s) return MIS_OK;
    for (int k = 1; k < MisSift::MAX_LANES; k++) {
        if (s->helper[k]) { mis_sift_destroy(s->helper[k]); s->helper[k] = nullptr; }
        if (s->helper_ctx[k]) { mis_context_destroy(s->helper_ctx[k]); s->helper_ctx[k] = nullptr; }
        if (s->helper_stream[k]) { mis_stream_destroy(s->helper_stream[k]); s->helper_stream[k] = nullptr; }
    }
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    if (s->mem) hipFree(s->mem);
    delete s;
    return MIS_OK;
}

// GaussianBlur(src -> dst); with prev / dog set, dog = dst - prev is written by the column pass too
static void blur(MisSift* s, const float* src, float* dst, int w, int h, double sigma, const float* prev = nullptr, float* dog = nullptr) {
    const Taps t = gaussian_taps(sigma);
    hipStream_t st = s->ctx->stream;
    const dim3 grows(((w + BLK - 1) / BLK + 255) / 256, h), gcols((w + 255) / 256, (h + BLK - 1) / BLK), block(256);
    // the tap counts of SIFT::create()'s sigmas run fused (one read, two writes per layer); `prev` is always the source there.
    // Segments: long enough that the N - 1 extra input rows stay a small share, short enough that a layer is >= ~1500 workgroups
    const int nsx = (w + FB_TW - 1) / FB_TW;
    // (small layers are bound by the serial walk of a segment, not by its N - 1 extra rows: short segments there)
    static const int seg_min = getenv("MIS_SIFT_SEG_MIN") ? atoi(getenv("MIS_SIFT_SEG_MIN")) : 16;
    int seg = (h * nsx + 1499) / 1500;
    seg = std::min(256, std::max(seg_min, (seg + 7) & ~7));
    const dim3 gfused(nsx, (h + seg - 1) / seg);
    const bool fused_ok = (prev == nullptr || prev == src) && (dog != nullptr) == (prev != nullptr);
#define MIS_BLUR_CASE(NN)                                                                                                                        \
    case NN:                                                                                                                                     \
        if (fused_ok) { hipLaunchKernelGGL(sift_blur_fused_kernel<NN>, gfused, block, 0, st, src, dst, dog, w, h, seg, t); break; }              \
        hipLaunchKernelGGL(sift_blur_rows_kernel<NN>, grows, block, 0, st, src, s->tmp, w, h, t);                                                \
        hipLaunchKernelGGL(sift_blur_cols_kernel<NN>, gcols, block, 0, st, (const float*)s->tmp, dst, w, h, t, prev, dog);                       \
        break;
    switch (t.n) {
        MIS_BLUR_CASE(11) MIS_BLUR_CASE(13) MIS_BLUR_CASE(17) MIS_BLUR_CASE(21) MIS_BLUR_CASE(27)   // the tap counts of SIFT::create()'s sigmas
        default:
            hipLaunchKernelGGL(sift_blur_rows_kernel<0>, grows, block, 0, st, src, s->tmp, w, h, t);
            hipLaunchKernelGGL(sift_blur_cols_kernel<0>, gcols, block, 0, st, (const float*)s->tmp, dst, w, h, t, prev, dog);
    }
#undef MIS_BLUR_CASE
}

// scale space of one image (everything up to and including the DoG pyramid)
static int sift_build(MisSift* s, const MisImage* bgr, const DevImage& din) {
    MisContext* ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const int w = bgr->width, h = bgr->height, nl = s->p.n_octave_layers;
    if (w != s->cur_w || h != s->cur_h) sift_plan(s, w, h);
    const Pyr& P = s->pyr;
    hipLaunchKernelGGL(sift_gray_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, (const uint8_t*)din.data, din.stride, w, h, s->gray);
    hipLaunchKernelGGL(sift_upsample_kernel, dim3((2 * w + 255) / 256, 2 * h), dim3(256), 0, st, (const uint8_t*)s->gray, w, h, s->grayf);
    const float sd = sqrtf(fmaxf((float)(s->p.sigma * s->p.sigma) - SIFT_INIT_SIGMA * SIFT_INIT_SIGMA * 4, 0.01f));
    blur(s, s->grayf, P.gauss[0], P.w[0], P.h[0], (double)sd);
    for (int o = 0; o < P.noct; o++) {
        const int ow = P.w[o], oh = P.h[o];
        for (int i = 0; i < nl + 3; i++) {
            if (o == 0 && i == 0) continue;
            float* dst = P.gauss[o * (nl + 3) + i];
            if (i == 0)
                hipLaunchKernelGGL(sift_decimate_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, st, (const float*)P.gauss[(o - 1) * (nl + 3) + nl], P.w[o - 1],
                                   P.h[o - 1], dst, ow, oh);
            else   // the column pass also writes dog[i - 1] = gauss[i] - gauss[i - 1]
                blur(s, P.gauss[o * (nl + 3) + i - 1], dst, ow, oh, s->sig[i], P.gauss[o * (nl + 3) + i - 1], P.dog[o * (nl + 2) + i - 1]);
        }
    }
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

extern "C" int mis_sift_detect(MisSift* s, const MisImage* bgr, MisFeatures* out) {
    if (!s) return MIS_E_INVALID;
    MisContext* ctx = s->ctx;
    MIS_CHECK(ctx, bgr && out && bgr->data, MIS_E_INVALID, "null argument");
    MIS_CHECK(ctx, bgr->dtype == MIS_U8 && bgr->channels == 3, MIS_E_UNSUPPORTED, "SIFT input must be 8UC3 (BGR)");
    MIS_CHECK(ctx, bgr->width >= 16 && bgr->height >= 16 && bgr->width <= s->max_w && bgr->height <= s->max_h, MIS_E_INVALID,
              "image %dx%d outside the detector's range (16x16 .. %dx%d)", bgr->width, bgr->height, s->max_w, s->max_h);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevImage din;
    int rc;
    if ((rc = mis_dev_image_in(ctx, bgr, &din)) != MIS_OK) return rc;
    if ((rc = sift_build(s, bgr, din)) != MIS_OK) return rc;
    const Pyr& P = s->pyr;
    const int nl = s->p.n_octave_layers, firstOctave = -1;
    MIS_HIP(ctx, hipMemsetAsync(s->counters, 0, 256, st));
    const int threshold = (int)floor(0.5 * s->p.contrast_threshold / nl * 255);
    for (int o = 0; o < P.noct; o++) {
        const int iw = P.w[o] - 2 * SIFT_IMG_BORDER, ih = P.h[o] - 2 * SIFT_IMG_BORDER;
        if (iw <= 0 || ih <= 0) continue;
        if (nl == 3)
            hipLaunchKernelGGL(sift_extrema_kernel<3>, dim3((iw + 255) / 256, (ih + EXT_ROWS - 1) / EXT_ROWS), dim3(256), 0, st, P, o, threshold, s->cand, s->counters, s->cand_cap);
        else
            hipLaunchKernelGGL(sift_extrema_generic_kernel, dim3((iw + 255) / 256, ih, nl), dim3(256), 0, st, P, o, threshold, s->cand, s->counters, s->cand_cap);
    }
    const SiftConsts K{(float)s->p.contrast_threshold, (float)s->p.edge_threshold, (float)s->p.sigma};
    hipLaunchKernelGGL(sift_refine_kernel, dim3(32 * ctx->num_cu), dim3(64), 0, st, P, K, (const int4*)s->cand, (const unsigned*)s->counters, s->cand_cap, s->surv,
                       s->counters + 3, s->kp_cap);
    hipLaunchKernelGGL(sift_orient_kernel, dim3(32 * ctx->num_cu), dim3(64), 0, st, P, (const SiftSurv*)s->surv, (const unsigned*)(s->counters + 3), s->kp_cap, s->raw,
                       s->counters + 1, s->kp_cap);
    // the one mid-frame synchronisation: the raw keypoint count sizes the output block and the grids that follow
    unsigned counts[4] = {0, 0, 0, 0};
    MIS_HIP(ctx, hipMemcpyAsync(counts, s->counters, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipStreamSynchronize(st));
    MIS_CHECK(ctx, counts[0] <= s->cand_cap, MIS_E_INVALID, "SIFT: %u extrema candidates exceed the capacity %u", counts[0], s->cand_cap);
    MIS_CHECK(ctx, counts[3] <= s->kp_cap, MIS_E_INVALID, "SIFT: %u refined candidates exceed the capacity %u", counts[3], s->kp_cap);
    MIS_CHECK(ctx, counts[1] <= s->kp_cap, MIS_E_INVALID, "SIFT: %u keypoints exceed the capacity %u", counts[1], s->kp_cap);
    // output block: keypoints + descriptors, sized for the raw count (duplicates only shrink it)
    const int nraw = (int)counts[1];
    memset(out, 0, sizeof(*out));
    out->img_w = bgr->width; out->img_h = bgr->height; out->desc_cols = 128; out->desc_dtype = MIS_F32;
    const size_t kb = mis_align_up(sizeof(MisKeyPoint) * (size_t)std::max(nraw, 1), 256), db = sizeof(float) * 128 * (size_t)std::max(nraw, 1);
    // (from the pool of recycled feature blocks: a hipMalloc / hipFree pair per frame synchronises the device -- with three frames in
    // flight every few steps of config 5 took 160 - 360 ms instead of 78)
    uint8_t* blk = nullptr;
    if ((rc = mis_feat_block_alloc(s->pool_ctx ? s->pool_ctx : ctx, kb + db, (void**)&blk)) != MIS_OK) return rc;
    out->owner_ = blk; out->keypoints = (MisKeyPoint*)blk; out->descriptors = blk + kb;
    if (nraw) {
        // KeyPointsFilter::removeDuplicatedSorted (a total order, so the atomics' append order never shows) + the
        // firstOctave < 0 rescaling, then the descriptors of the survivors
        MIS_HIP(ctx, hipMemsetAsync(s->rank, 0, 2 * sizeof(unsigned) * (size_t)nraw, st));
        unsigned* dupi = s->rank + nraw;
        hipLaunchKernelGGL(sift_rank_kernel, dim3((nraw + 255) / 256, RANK_SLICES), dim3(256), 0, st, (const MisKeyPoint*)s->raw, (const unsigned*)(s->counters + 1), s->kp_cap,
                           s->rank, dupi);
        hipLaunchKernelGGL(sift_scatter_kernel, dim3((nraw + 255) / 256), dim3(256), 0, st, (const unsigned*)(s->counters + 1), s->kp_cap, (const unsigned*)s->rank,
                           (const unsigned*)dupi, s->perm, s->dup);
        hipLaunchKernelGGL(sift_compact_kernel, dim3(1), dim3(1024), 0, st, (const MisKeyPoint*)s->raw, (const unsigned*)(s->counters + 1), s->kp_cap, (const unsigned*)s->perm,
                           (const uint8_t*)s->dup, firstOctave, out->keypoints, s->counters + 2);
        hipLaunchKernelGGL(sift_descriptor_kernel, dim3(nraw), dim3(64), 0, st, P, (const MisKeyPoint*)out->keypoints, (const unsigned*)(s->counters + 2),
                           (float*)out->descriptors);
        MIS_HIP(ctx, hipGetLastError());
        MIS_HIP(ctx, hipMemcpyAsync(&counts[2], s->counters + 2, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        MIS_HIP(ctx, hipStreamSynchronize(st));
    }
    out->n = (int)counts[2];
    return mis_dev_image_release(ctx, &din);
}

extern "C" int mis_sift_debug_level(MisSift* s, const MisImage* bgr, int octave, int layer, int dog, float* host_out, int* width, int* height) {
    if (!s) return MIS_E_INVALID;
    MisContext* ctx = s->ctx;
    MIS_CHECK(ctx, bgr && bgr->data && width && height, MIS_E_INVALID, "null argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage din;
    int rc;
    if ((rc = mis_dev_image_in(ctx, bgr, &din)) != MIS_OK) return rc;
    if ((rc = sift_build(s, bgr, din)) != MIS_OK) return rc;
    const Pyr& P = s->pyr;
    const int nl = s->p.n_octave_layers;
    MIS_CHECK(ctx, octave >= 0 && octave < P.noct && layer >= 0 && layer < (dog ? nl + 2 : nl + 3), MIS_E_INVALID, "bad octave / layer");
    *width = P.w[octave]; *height = P.h[octave];
    if (host_out) {
        const float* src = dog ? P.dog[octave * (nl + 2) + layer] : P.gauss[octave * (nl + 3) + layer];
        MIS_HIP(ctx, hipMemcpyAsync(host_out, src, sizeof(float) * (size_t)P.w[octave] * P.h[octave], hipMemcpyDeviceToHost, ctx->stream));
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return mis_dev_image_release(ctx, &din);
}

// Several frames: up to four lanes (this finder and helpers with their own scale space, context and stream), one host thread
// each, frame i on lane i % lanes.  A frame's kernels are partly bandwidth bound (the blurs) and partly latency bound (the fifty
// small layers of octaves >= 2, one wave per keypoint, two host synchronisations), so several frames in flight keep the device
// busier than one: config 5's eight 8K frames take 54 ms on two lanes (round 3) and (round 4, MIS_SIFT_LANES, default 3) less on three.
extern "C" int mis_sift_detect_batch(MisSift* s, const MisImage* imgs, int n, MisFeatures* out) {
    if (!s) return MIS_E_INVALID;
    MisContext* ctx = s->ctx;
    MIS_CHECK(ctx, imgs && out && n >= 1, MIS_E_INVALID, "null argument");
    if (n == 1) return mis_sift_detect(s, imgs, out);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    static const int want = getenv("MIS_SIFT_LANES") ? atoi(getenv("MIS_SIFT_LANES")) : 3;
    const int nl = std::max(1, std::min({want, n, MisSift::MAX_LANES}));
    for (int k = 1; k < nl; k++) {
        if (s->helper[k]) continue;
        // lanes 1 and 2 run on the context's auxiliary streams (shared with the matcher, which runs later), lane 3 on a stream of its own
        hipStream_t st = nullptr;
        int rc = MIS_OK;
        if (k <= 2) rc = mis_aux_stream(ctx, k - 1, &st);
        else { rc = mis_stream_create(ctx->device, 0, &s->helper_stream[k]); st = (hipStream_t)s->helper_stream[k]; }
        if (rc == MIS_OK) rc = mis_context_create(ctx->device, (void*)st, &s->helper_ctx[k]);
        if (rc == MIS_OK) rc = mis_sift_create(s->helper_ctx[k], &s->p, s->max_w, s->max_h, &s->helper[k]);
        if (rc == MIS_OK) s->helper[k]->pool_ctx = ctx;
        if (rc != MIS_OK) return mis_set_error(ctx, rc, "SIFT batch: cannot create lane %d (%s)", k, s->helper_ctx[k] ? s->helper_ctx[k]->err.c_str() : "stream / context");
    }
    // the frames' producers were enqueued on the context's stream; the helpers' streams do not see them otherwise
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MisSift* lanes[MisSift::MAX_LANES] = {s, s->helper[1], s->helper[2], s->helper[3]};
    int rcs[MisSift::MAX_LANES] = {MIS_OK, MIS_OK, MIS_OK, MIS_OK};
    auto work = [&](int lane) {
        for (int i = lane; i < n; i += nl) {
            const int rc = mis_sift_detect(lanes[lane], &imgs[i], &out[i]);
            if (rc != MIS_OK) { rcs[lane] = rc; return; }
            out[i].img_idx = i;
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nl; k++) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    if (rcs[0] != MIS_OK) return rcs[0];
    for (int k = 1; k < nl; k++)
        if (rcs[k] != MIS_OK) return mis_set_error(ctx, rcs[k], "SIFT batch, lane %d: %s", k, s->helper_ctx[k]->err.c_str());
    return MIS_OK;
}
