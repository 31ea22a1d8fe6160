// blend.hip -- multi-band / feather / plain blender, replaces the reference's
// cv::detail::Blender calls: image_stitching/image_stitching.cpp:1175-1192 (createDefault,
// setNumBands / setSharpness, prepare), :1218 (feed), :1225 (blend).
//
// HBM layout: the panorama accumulators are one 16SC3 Laplacian image and one f32 weight image per
// pyramid level, tight rows (level widths are multiples of 2 by construction).  A feed never
// materialises OpenCV's reflect-padded copy of the frame: level 0 of the frame pyramid is a *view*
// (index maps) of the warped image; only levels >= 1 of the Gaussian pyramids exist in scratch.
// The Laplacian (pyrUp + saturating subtract), the weight multiply and the accumulate are one
// kernel per level, and pixels whose weight is exactly 0 are skipped: `dst += (short)(lap * 0)`,
// `wsum += 0` are no-ops, so the result is bit-identical while the read-modify-write traffic of the
// panorama drops to the footprint of the frame's non-zero weights.
#include "common.h"
#include "dev_math.h"
#include <algorithm>
#include <cmath>

#define MIS_MAX_BANDS 16

struct MisBlender {
    MisContext* ctx = nullptr;
    int type = MIS_BLEND_MULTI_BAND, actual_bands = 5, num_bands = 0;
    float sharpness = 0.02f;
    MisRect roi{0, 0, 0, 0};  // dst_roi_ (padded for multi-band)
    int fw = 0, fh = 0;       // dst_roi_final_ size
    int lw[MIS_MAX_BANDS + 1], lh[MIS_MAX_BANDS + 1];
    int16_t* lap[MIS_MAX_BANDS + 1];
    float* wgt[MIS_MAX_BANDS + 1];
    uint8_t* dst_mask = nullptr;  // plain blender
    void* pano_mem = nullptr;
    size_t pano_bytes = 0;  // capacity kept across prepare() calls (grow-only)
    bool prepared = false;
    // grow-only scratch of one feed: Gaussian pyramids of the frame (levels 1..nb) and of its weights
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
};

namespace {

constexpr float WEIGHT_EPS = 1e-5f;
#ifndef FEED_TAIL_PIXELS_N
#define FEED_TAIL_PIXELS_N 8192     // measured on 16 x 4K: 2048 no gain, 8192 -0.3 ms per step, 49152 +1.1 ms (one workgroup builds the levels)
#endif
constexpr size_t FEED_TAIL_PIXELS = FEED_TAIL_PIXELS_N;   // levels of a frame's pyramid this small are handled by the two tail kernels of a feed

__device__ __forceinline__ int16_t sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// Level-0 view of one frame inside its padded tile: copyMakeBorder(BORDER_REFLECT) of the image,
// copyMakeBorder(BORDER_CONSTANT 0) of mask/255.
struct FrameView {
    const int16_t* img;   // 16SC3
    size_t istride;       // in int16 elements
    const uint8_t* mask;
    size_t mstride;
    int w, h;             // image size
    int left, top;        // margins of the padded tile
    int tw, th;           // padded tile size
};

__device__ __forceinline__ void view_px(const FrameView& v, int tx, int ty, int* c) {
    const int16_t* p = v.img + (size_t)mis_reflect(ty - v.top, v.h) * v.istride + 3 * (size_t)mis_reflect(tx - v.left, v.w);
    c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
}
__device__ __forceinline__ float view_w(const FrameView& v, int tx, int ty) {
    int x = tx - v.left, y = ty - v.top;
    if ((unsigned)x >= (unsigned)v.w || (unsigned)y >= (unsigned)v.h) return 0.f;
    return (float)v.mask[(size_t)y * v.mstride + x] * (float)(1. / 255.);
}

// ---- pyrDown, 5-tap [1 4 6 4 1], BORDER_REFLECT_101 ----
// s16: integer, (v + 128) >> 8, the pass order is free.  f32: row = s[2x]*6 + (s[2x-1]+s[2x+1])*4 + s[2x-2] + s[2x+2], the
// same vertically, * 1/256 -- in that order.  A block produces a 32 x 16 destination tile: the 67 x 35 source footprint is
// staged in LDS once (the reflect index maps are evaluated once per source pixel instead of 25 times per output), then the
// separable filter runs out of LDS.
constexpr int PD_W = 32, PD_H = 16, PD_SW = 2 * PD_W + 3, PD_SH = 2 * PD_H + 3;

// ---- one level of a feed: both reductions in one kernel (same 67 x 35 footprint) ----
// Gaussian level 1 of the image (s16x3) and of the weights (mask / 255, f32) from one staging of the tile.  Tiles whose
// footprint lies inside the frame (no reflection at all: the bulk of a 4K frame) stage their rows with aligned dword
// loads (a row is 402 contiguous bytes of 16SC3) instead of three 2-byte loads per pixel, and their weights with dword
// loads of the mask.
constexpr int PDV_ROW_DW = (PD_SW * 6 + 2 + 3) / 4 + 1;   // dwords that cover a row of 67 pixels from an address rounded down to 4
constexpr int PDV_MROW_DW = (PD_SW + 3 + 3) / 4 + 1;      // same for the 67 mask bytes
// FROM_VIEW = false: the same pair of reductions for a level >= 1 (sources: the frame's Gaussian level `src` and weight level
// `wsrc`, sw x sh, tightly packed), one launch instead of two per level.
template <bool FROM_VIEW>
__global__ __launch_bounds__(256) void pyr_down_view_kernel(FrameView v, const int16_t* __restrict__ src, const float* __restrict__ wsrc, int psw, int psh,
                                                            int16_t* __restrict__ dst, float* __restrict__ wdst, int dw, int dh) {
    __shared__ __attribute__((aligned(4))) int16_t tile[PD_SH * PDV_ROW_DW * 2];   // row pitch PDV_ROW_DW dwords; pixels start `toff` shorts in
    __shared__ float wt[PD_SH * PD_SW];
    __shared__ int hbuf[PD_SH * PD_W * 3];
    __shared__ float hw[PD_SH * PD_W];
    const int x0 = blockIdx.x * PD_W, y0 = blockIdx.y * PD_H, t = threadIdx.x;
    const int sw = FROM_VIEW ? v.tw : psw, sh = FROM_VIEW ? v.th : psh;
    const int tx0 = 2 * x0 - 2, ty0 = 2 * y0 - 2;                    // tile coordinates of the footprint's corner
    const int ix0 = tx0 - v.left, iy0 = ty0 - v.top;                 // image coordinates of the same
    const bool interior = FROM_VIEW && tx0 >= 0 && ty0 >= 0 && tx0 + PD_SW <= sw && ty0 + PD_SH <= sh && ix0 >= 0 && iy0 >= 0 && ix0 + PD_SW + 12 <= v.w && iy0 + PD_SH <= v.h &&   /* + 12: the dword loads may run past the last needed byte */
                          (v.istride & 1) == 0 && ((uintptr_t)v.img & 3) == 0 && (v.mstride & 3) == 0 && ((uintptr_t)v.mask & 3) == 0;
    int toff;   // shorts between the start of an LDS row and its first pixel
    if (interior) {
        const size_t e0 = 3 * (size_t)ix0;                           // first short of a row, relative to the row start
        toff = (int)(e0 & 1);
        const int16_t* base = v.img + (size_t)iy0 * v.istride + (e0 - toff);
        // all loads of a thread are issued before the first LDS store (a load -> store loop serialises the round trips)
        constexpr int NI = (PD_SH * PDV_ROW_DW + 255) / 256, NM = (PD_SH * PDV_MROW_DW + 255) / 256;
        const int moff = ix0 & 3;
        const uint8_t* mbase = v.mask + (size_t)iy0 * v.mstride + (ix0 - moff);
        unsigned ri[NI], rm[NM];
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = t + 256 * k, r = i / PDV_ROW_DW, c = i - r * PDV_ROW_DW;
            ri[k] = i < PD_SH * PDV_ROW_DW ? *reinterpret_cast<const unsigned*>(base + (size_t)r * v.istride + 2 * c) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NM; k++) {
            const int i = t + 256 * k, r = i / PDV_MROW_DW, c = i - r * PDV_MROW_DW;
            rm[k] = i < PD_SH * PDV_MROW_DW ? *reinterpret_cast<const unsigned*>(mbase + (size_t)r * v.mstride + 4 * c) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = t + 256 * k;
            if (i < PD_SH * PDV_ROW_DW) reinterpret_cast<unsigned*>(tile)[i] = ri[k];
        }
#pragma unroll
        for (int k = 0; k < NM; k++) {
            const int i = t + 256 * k, r = i / PDV_MROW_DW, c = i - r * PDV_MROW_DW;
            if (i < PD_SH * PDV_MROW_DW) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int x = 4 * c + q - moff;
                    if (x >= 0 && x < PD_SW) wt[r * PD_SW + x] = (float)((rm[k] >> (8 * q)) & 255u) * (float)(1. / 255.);
                }
            }
        }
    } else if (!FROM_VIEW && tx0 >= 0 && ty0 >= 0 && tx0 + PD_SW + 2 <= sw && ty0 + PD_SH <= sh && (sw & 1) == 0 && ((uintptr_t)src & 3) == 0) {
        // a level >= 1, footprint inside it: rows are 3 sw shorts apart (sw even: every row starts on the same dword phase)
        const size_t e0 = 3 * (size_t)tx0;
        toff = (int)(e0 & 1);
        const int16_t* base = src + (size_t)ty0 * sw * 3 + (e0 - toff);
        const float* wbase = wsrc + (size_t)ty0 * sw + tx0;
        constexpr int NI = (PD_SH * PDV_ROW_DW + 255) / 256, NW = (PD_SH * PD_SW + 255) / 256;
        unsigned ri[NI];
        float rw[NW];
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = t + 256 * k, r = i / PDV_ROW_DW, c = i - r * PDV_ROW_DW;
            ri[k] = i < PD_SH * PDV_ROW_DW ? *reinterpret_cast<const unsigned*>(base + (size_t)r * sw * 3 + 2 * c) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const int i = t + 256 * k, r = i / PD_SW, c = i - r * PD_SW;
            rw[k] = i < PD_SH * PD_SW ? wbase[(size_t)r * sw + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = t + 256 * k;
            if (i < PD_SH * PDV_ROW_DW) reinterpret_cast<unsigned*>(tile)[i] = ri[k];
        }
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const int i = t + 256 * k;
            if (i < PD_SH * PD_SW) wt[i] = rw[k];
        }
    } else {
        toff = 0;
        for (int i = t; i < PD_SH * PD_SW; i += 256) {
            const int r = i / PD_SW, c = i - r * PD_SW;
            const int sy = mis_reflect101(ty0 + r, sh), sx = mis_reflect101(tx0 + c, sw);
            int px[3];
            if (FROM_VIEW) view_px(v, sx, sy, px);
            else { const int16_t* q = src + ((size_t)sy * sw + sx) * 3; px[0] = q[0]; px[1] = q[1]; px[2] = q[2]; }
            int16_t* o = tile + (size_t)r * (PDV_ROW_DW * 2) + 3 * c;
            o[0] = (int16_t)px[0]; o[1] = (int16_t)px[1]; o[2] = (int16_t)px[2];
            wt[i] = FROM_VIEW ? view_w(v, sx, sy) : wsrc[(size_t)sy * sw + sx];
        }
    }
    __syncthreads();
    // horizontal pass: one (row, column) of the half-resolution grid per item, all three channels and the weight.  The five
    // source pixels are 15 contiguous shorts of the LDS row: eight aligned dword reads, halves picked at compile time (toff,
    // the only run-time part of the alignment, is uniform in the block).
    static_assert(PD_W == 32, "item -> (row, column) uses shifts");
    for (int item = t; item < PD_SH * PD_W; item += 256) {
        const int r = item >> 5, x = item & 31;
        const unsigned* q = reinterpret_cast<const unsigned*>(tile) + (size_t)r * PDV_ROW_DW + 3 * x;   // shorts 6 x .. 6 x + 15 of the row
        unsigned wd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) wd[k] = q[k];
        int px[15];
        if (toff) {
#pragma unroll
            for (int k = 0; k < 15; k++) px[k] = (int)(int16_t)(wd[(k + 1) >> 1] >> (16 * ((k + 1) & 1)));
        } else {
#pragma unroll
            for (int k = 0; k < 15; k++) px[k] = (int)(int16_t)(wd[k >> 1] >> (16 * (k & 1)));
        }
        int* hb = hbuf + item * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) hb[c] = px[6 + c] * 6 + (px[3 + c] + px[9 + c]) * 4 + px[c] + px[12 + c];
        const float* sw_ = wt + r * PD_SW + 2 * x;
        hw[item] = ((sw_[2] * 6.f + (sw_[1] + sw_[3]) * 4.f) + sw_[0]) + sw_[4];
    }
    __syncthreads();
    // vertical pass: one output pixel per item
    for (int item = t; item < PD_H * PD_W; item += 256) {
        const int y = item >> 5, x = item & 31;
        if (x0 + x >= dw || y0 + y >= dh) continue;
        const int* p = hbuf + (2 * y) * (PD_W * 3) + 3 * x;
        int16_t* o = dst + ((size_t)(y0 + y) * dw + x0 + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int acc = p[2 * PD_W * 3 + c] * 6 + (p[PD_W * 3 + c] + p[3 * PD_W * 3 + c]) * 4 + p[c] + p[4 * PD_W * 3 + c];
            o[c] = (int16_t)((acc + 128) >> 8);
        }
        const float* pw_ = hw + (2 * y) * PD_W + x;
        const float rr = ((pw_[2 * PD_W] * 6.f + (pw_[PD_W] + pw_[3 * PD_W]) * 4.f) + pw_[0]) + pw_[4 * PD_W];
        wdst[(size_t)(y0 + y) * dw + x0 + x] = rr * (1.f / 256.f);
    }
}

// pyrUp of a coarse 16SC3 level evaluated at one fine pixel (fine = 2 x coarse exactly):
// even: r[x-1] + 6 r[x] + r[x+1], odd: 4 (r[x] + r[x+1]); left/top neighbour of sample 0 is sample 1,
// right/bottom neighbour of the last sample is the last sample; (v + 32) >> 6.
__device__ __forceinline__ void pyr_up_at(const int16_t* c, int cw, int ch, int fx, int fy, int* out) {
    int X = fx >> 1, Y = fy >> 1;
    int xm = X > 0 ? X - 1 : (cw > 1 ? 1 : 0), xp = X + 1 < cw ? X + 1 : cw - 1;
    int ym = Y > 0 ? Y - 1 : (ch > 1 ? 1 : 0), yp = Y + 1 < ch ? Y + 1 : ch - 1;
    const int16_t* r0 = c + (size_t)ym * cw * 3;
    const int16_t* r1 = c + (size_t)Y * cw * 3;
    const int16_t* r2 = c + (size_t)yp * cw * 3;
    const bool ox = fx & 1, oy = fy & 1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int h0, h1, h2;
        if (!ox) {
            h0 = r0[3 * xm + k] + r0[3 * X + k] * 6 + r0[3 * xp + k];
            h1 = r1[3 * xm + k] + r1[3 * X + k] * 6 + r1[3 * xp + k];
            h2 = r2[3 * xm + k] + r2[3 * X + k] * 6 + r2[3 * xp + k];
        } else {
            h0 = (r0[3 * X + k] + r0[3 * xp + k]) * 4;
            h1 = (r1[3 * X + k] + r1[3 * xp + k]) * 4;
            h2 = (r2[3 * X + k] + r2[3 * xp + k]) * 4;
        }
        int v = oy ? (h1 + h2) * 4 : (h0 + h1 * 6 + h2);
        out[k] = (int16_t)((v + 32) >> 6);
    }
}

// pyrUp of a coarse level at the 2 x 2 fine block of coarse pixel (X, Y): up[k][c], k = (fy & 1) * 2 + (fx & 1).  The four pixels share
// the 3 x 3 coarse neighbourhood (27 loads per block instead of up to 27 per pixel); per pixel the sums are those of pyr_up_at.
__device__ __forceinline__ void pyr_up_block(const int16_t* __restrict__ c, int cw, int ch, int X, int Y, int (*up)[3]) {
    const int xm = X > 0 ? X - 1 : (cw > 1 ? 1 : 0), xp = X + 1 < cw ? X + 1 : cw - 1;
    const int ym = Y > 0 ? Y - 1 : (ch > 1 ? 1 : 0), yp = Y + 1 < ch ? Y + 1 : ch - 1;
    const int rows[3] = {ym, Y, yp};
    int he[3][3], ho[3][3];   // [row][channel]: horizontal sums for an even / odd fine column
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int16_t* p = c + (size_t)rows[r] * cw * 3;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int a = p[3 * xm + q], b = p[3 * X + q], d = p[3 * xp + q];
            he[r][q] = a + b * 6 + d;
            ho[r][q] = (b + d) * 4;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
        up[0][q] = (int16_t)((he[0][q] + he[1][q] * 6 + he[2][q] + 32) >> 6);
        up[1][q] = (int16_t)((ho[0][q] + ho[1][q] * 6 + ho[2][q] + 32) >> 6);
        up[2][q] = (int16_t)(((he[1][q] + he[2][q]) * 4 + 32) >> 6);
        up[3][q] = (int16_t)(((ho[1][q] + ho[2][q]) * 4 + 32) >> 6);
    }
}

// ---- one pyramid level of a feed: Laplacian = G_i - pyrUp(G_{i+1}) (saturating), then
// dst += (short)(lap * w), wsum += w over the tile rectangle of the panorama level ----
template <bool FROM_VIEW, bool LAST>
__global__ __launch_bounds__(256) void laplace_accumulate_kernel(FrameView v, const int16_t* g, const float* wl, int tw, int th,
                                                                 const int16_t* coarse, int cw, int ch, int16_t* dlap, float* dwgt,
                                                                 int pw, int x_tl, int y_tl) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= tw || y >= th) return;
    float w = FROM_VIEW ? view_w(v, x, y) : wl[(size_t)y * tw + x];
    if (w == 0.f) return;  // exact no-op contribution
    int c[3];
    if (FROM_VIEW) view_px(v, x, y, c);
    else { const int16_t* p = g + ((size_t)y * tw + x) * 3; c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; }
    if (!LAST) {
        int up[3];
        pyr_up_at(coarse, cw, ch, x, y, up);
        c[0] = sat_s16(c[0] - up[0]); c[1] = sat_s16(c[1] - up[1]); c[2] = sat_s16(c[2] - up[2]);
    }
    size_t o = (size_t)(y_tl + y) * pw + (x_tl + x);
    int16_t* d = dlap + o * 3;
    d[0] = (int16_t)(d[0] + (int16_t)((float)c[0] * w));
    d[1] = (int16_t)(d[1] + (int16_t)((float)c[1] * w));
    d[2] = (int16_t)(d[2] + (int16_t)((float)c[2] * w));
    dwgt[o] += w;
}

// Level 0 of a feed with at least one band: a thread owns a 2 x 2 block of the frame's tile (its size is a multiple of 2^bands).
// The four pixels share the 3 x 3 coarse neighbourhood of pyrUp, so the block costs 27 coarse loads instead of up to 27 per
// pixel; the arithmetic per pixel is that of laplace_accumulate_kernel / pyr_up_at (integers, the same sums).
__global__ __launch_bounds__(256) void laplace_accumulate_view2x2_kernel(FrameView v, int tw, int th, const int16_t* __restrict__ coarse, int cw, int ch,
                                                                         int16_t* __restrict__ dlap, float* __restrict__ dwgt, int pw, int x_tl, int y_tl) {
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);   // block index = coarse pixel
    if (2 * X >= tw || 2 * Y >= th) return;
    float w[4];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; k++) { w[k] = view_w(v, 2 * X + (k & 1), 2 * Y + (k >> 1)); any |= w[k] != 0.f; }
    if (!any) return;   // exact no-op contributions
    int up[4][3];
    pyr_up_block(coarse, cw, ch, X, Y, up);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (w[k] == 0.f) continue;
        const int fx = 2 * X + (k & 1), fy = 2 * Y + (k >> 1);
        int px[3];
        view_px(v, fx, fy, px);
        const size_t o = (size_t)(y_tl + fy) * pw + (x_tl + fx);
        int16_t* d = dlap + o * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int lap = sat_s16(px[c] - up[k][c]);
            d[c] = (int16_t)(d[c] + (int16_t)((float)lap * w[k]));
        }
        dwgt[o] += w[k];
    }
}

// ---- the small levels of a feed in two launches ----
// From some level on a frame's pyramid has a few thousand pixels and every per-level launch costs more in launch-to-launch
// latency than in work (a 4K frame at 8 bands: 15 launches of ~6 us for levels 4..8).  feed_tail_build_kernel builds all
// the remaining Gaussian levels in ONE workgroup (a level depends on the previous one: __syncthreads between them;
// the data goes through global memory, which a workgroup sees coherently), feed_tail_accumulate_kernel adds the
// Laplacians of all those levels to the panorama in one grid.  Arithmetic: that of the per-level kernels.
struct FeedTail {
    int first, nb;                                    // levels first + 1 .. nb are built by feed_tail_build_kernel (first >= 1; first > nb: none)
    int acc_first;                                    // levels acc_first .. nb are accumulated by feed_tail_accumulate_kernel
    int tw[MIS_MAX_BANDS + 1], th[MIS_MAX_BANDS + 1]; // tile size per level
    int16_t* G[MIS_MAX_BANDS + 1];                    // Gaussian levels of the frame (scratch), G[first] already built
    float* W[MIS_MAX_BANDS + 1];
    int16_t* lap[MIS_MAX_BANDS + 1];                  // panorama pyramids
    float* wgt[MIS_MAX_BANDS + 1];
    int pw[MIS_MAX_BANDS + 1], x_tl[MIS_MAX_BANDS + 1], y_tl[MIS_MAX_BANDS + 1];
    int blk_off[MIS_MAX_BANDS + 2];                   // accumulate grid: first block of every level
};
__global__ __launch_bounds__(1024) void feed_tail_build_kernel(FeedTail t) {
    for (int l = t.first; l < t.nb; l++) {
        const int sw = t.tw[l], sh = t.th[l], dw = t.tw[l + 1], dh = t.th[l + 1];
        const int16_t* src = t.G[l];
        const float* wsrc = t.W[l];
        for (int i = threadIdx.x; i < dw * dh; i += 1024) {
            const int y = i / dw, x = i - y * dw;
            int xi[5], yi[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { xi[k] = mis_reflect101(2 * x - 2 + k, sw); yi[k] = mis_reflect101(2 * y - 2 + k, sh); }
            int acc[3] = {0, 0, 0};
            float hr[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int wj = j == 2 ? 6 : ((j == 1 || j == 3) ? 4 : 1);
                const int16_t* r = src + (size_t)yi[j] * sw * 3;
#pragma unroll
                for (int c = 0; c < 3; c++)
                    acc[c] += wj * (r[3 * xi[2] + c] * 6 + (r[3 * xi[1] + c] + r[3 * xi[3] + c]) * 4 + r[3 * xi[0] + c] + r[3 * xi[4] + c]);
                const float* wr = wsrc + (size_t)yi[j] * sw;
                hr[j] = ((wr[xi[2]] * 6.f + (wr[xi[1]] + wr[xi[3]]) * 4.f) + wr[xi[0]]) + wr[xi[4]];
            }
            int16_t* o = t.G[l + 1] + (size_t)i * 3;
            o[0] = (int16_t)((acc[0] + 128) >> 8); o[1] = (int16_t)((acc[1] + 128) >> 8); o[2] = (int16_t)((acc[2] + 128) >> 8);
            t.W[l + 1][i] = (((hr[2] * 6.f + (hr[1] + hr[3]) * 4.f) + hr[0]) + hr[4]) * (1.f / 256.f);
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void feed_tail_accumulate_kernel(FeedTail t) {
    int l = t.acc_first;
    while (l < t.nb && (int)blockIdx.x >= t.blk_off[l + 1]) l++;
    const int tw = t.tw[l], th = t.th[l];
    const int i = ((int)blockIdx.x - t.blk_off[l]) * 256 + threadIdx.x;
    if (l < t.nb) {
        // a thread owns a 2 x 2 block (tile sizes below the last level are even): the four pixels share pyrUp's neighbourhood
        const int cw = t.tw[l + 1], ch = t.th[l + 1];
        if (i >= cw * ch) return;
        const int Y = i / cw, X = i - Y * cw;
        float w[4];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; k++) { w[k] = t.W[l][(size_t)(2 * Y + (k >> 1)) * tw + 2 * X + (k & 1)]; any |= w[k] != 0.f; }
        if (!any) return;   // exact no-op contributions
        int up[4][3];
        pyr_up_block(t.G[l + 1], cw, ch, X, Y, up);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (w[k] == 0.f) continue;
            const int fx = 2 * X + (k & 1), fy = 2 * Y + (k >> 1);
            const int16_t* p = t.G[l] + ((size_t)fy * tw + fx) * 3;
            const size_t o = (size_t)(t.y_tl[l] + fy) * t.pw[l] + (t.x_tl[l] + fx);
            int16_t* d = t.lap[l] + o * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int lap = sat_s16(p[c] - up[k][c]);
                d[c] = (int16_t)(d[c] + (int16_t)((float)lap * w[k]));
            }
            t.wgt[l][o] += w[k];
        }
        return;
    }
    if (i >= tw * th) return;
    const int y = i / tw, x = i - y * tw;
    const float w = t.W[l][i];
    if (w == 0.f) return;  // exact no-op contribution
    const int16_t* p = t.G[l] + (size_t)i * 3;
    const size_t o = (size_t)(t.y_tl[l] + y) * t.pw[l] + (t.x_tl[l] + x);
    int16_t* d = t.lap[l] + o * 3;
    d[0] = (int16_t)(d[0] + (int16_t)((float)p[0] * w));
    d[1] = (int16_t)(d[1] + (int16_t)((float)p[1] * w));
    d[2] = (int16_t)(d[2] + (int16_t)((float)p[2] * w));
    t.wgt[l][o] += w;
}

// ---- blend(): normalise by the weight sum, collapse the pyramid, emit the final image + mask ----
__global__ __launch_bounds__(256) void normalize_kernel(int16_t* lap, const float* wgt, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float w = wgt[i] + WEIGHT_EPS;
    int16_t* d = lap + i * 3;
    d[0] = (int16_t)((float)d[0] / w); d[1] = (int16_t)((float)d[1] / w); d[2] = (int16_t)((float)d[2] / w);
}

// fine level (un-normalised) <- sat(pyrUp(coarse, already final) + normalise(fine))
__global__ __launch_bounds__(256) void collapse_kernel(int16_t* fine, const float* fwgt, int fw, int fh, const int16_t* coarse, int cw, int ch) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= fw || y >= fh) return;
    size_t o = (size_t)y * fw + x;
    float w = fwgt[o] + WEIGHT_EPS;
    int16_t* d = fine + o * 3;
    int up[3];
    pyr_up_at(coarse, cw, ch, x, y, up);
    d[0] = sat_s16(up[0] + (int)(int16_t)((float)d[0] / w));
    d[1] = sat_s16(up[1] + (int)(int16_t)((float)d[1] / w));
    d[2] = sat_s16(up[2] + (int)(int16_t)((float)d[2] / w));
}

// the same per 2 x 2 block of the fine level (fine = 2 x coarse exactly)
__global__ __launch_bounds__(256) void collapse2x2_kernel(int16_t* __restrict__ fine, const float* __restrict__ fwgt, int fw, int fh, const int16_t* __restrict__ coarse,
                                                          int cw, int ch) {
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= cw || Y >= ch) return;
    int up[4][3];
    pyr_up_block(coarse, cw, ch, X, Y, up);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t o = (size_t)(2 * Y + (k >> 1)) * fw + 2 * X + (k & 1);
        const float w = fwgt[o] + WEIGHT_EPS;
        int16_t* d = fine + o * 3;
        d[0] = sat_s16(up[k][0] + (int)(int16_t)((float)d[0] / w));
        d[1] = sat_s16(up[k][1] + (int)(int16_t)((float)d[1] / w));
        d[2] = sat_s16(up[k][2] + (int)(int16_t)((float)d[2] / w));
    }
}

// crop to the un-padded roi, dst_mask = wsum0 > eps (or the or-ed mask), zero outside the mask
__global__ __launch_bounds__(256) void finalize_kernel(const int16_t* lap0, const float* w0, const uint8_t* pmask, int pw, int fw, int fh,
                                                       int16_t* dst, size_t dstride, uint8_t* dmask, size_t mstride) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= fw || y >= fh) return;
    size_t o = (size_t)y * pw + x;
    unsigned m = pmask ? pmask[o] : (w0[o] > WEIGHT_EPS ? 255u : 0u);
    const int16_t* s = lap0 + o * 3;
    int16_t* d = (int16_t*)((uint8_t*)dst + (size_t)y * dstride) + 3 * (size_t)x;
    d[0] = m ? s[0] : (int16_t)0; d[1] = m ? s[1] : (int16_t)0; d[2] = m ? s[2] : (int16_t)0;
    dmask[(size_t)y * mstride + x] = (uint8_t)m;
}

// ---- plain Blender::feed ----
__global__ __launch_bounds__(256) void feed_plain_kernel(const int16_t* img, size_t istride, const uint8_t* mask, size_t mstride, int w, int h,
                                                         int16_t* dst, uint8_t* dmask, int pw, int dx, int dy) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    unsigned m = mask[(size_t)y * mstride + x];
    size_t o = (size_t)(dy + y) * pw + dx + x;
    if (m) {
        const int16_t* s = img + (size_t)y * istride + 3 * (size_t)x;
        dst[o * 3] = s[0]; dst[o * 3 + 1] = s[1]; dst[o * 3 + 2] = s[2];
    }
    dmask[o] |= (uint8_t)m;
}

// ---- FeatherBlender: L1 distance transform (exact city-block distance to the nearest zero pixel,
// clamped at 8192) as two separable min-plus sweeps; rows then columns, one thread per line ----
__global__ void dist_rows_kernel(const uint8_t* mask, size_t mstride, int w, int h, int* d) {
    int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h) return;
    const int INF = 8192;
    int run = INF;
    int* r = d + (size_t)y * w;
    for (int x = 0; x < w; x++) { run = mask[(size_t)y * mstride + x] ? min(run + 1, INF) : 0; r[x] = run; }
    run = INF;
    for (int x = w - 1; x >= 0; x--) { run = mask[(size_t)y * mstride + x] ? min(run + 1, INF) : 0; r[x] = min(r[x], run); }
}
__global__ void dist_cols_weight_kernel(int* d, int w, int h, float sharpness, float* wm) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const int INF = 8192;
    int run = INF;
    for (int y = 0; y < h; y++) { int v = d[(size_t)y * w + x]; run = min(run + 1, v); d[(size_t)y * w + x] = run; }
    run = INF;
    for (int y = h - 1; y >= 0; y--) {
        int v = d[(size_t)y * w + x];
        run = min(run + 1, v);
        int t = min(run, INF);
        float wv = (float)t * sharpness;  // createWeightMap: multiply, then THRESH_TRUNC at 1
        wm[(size_t)y * w + x] = wv > 1.f ? 1.f : wv;
    }
}
__global__ __launch_bounds__(256) void feed_feather_kernel(const int16_t* img, size_t istride, const float* wm, int w, int h, int16_t* dst,
                                                           float* dwgt, int pw, int dx, int dy) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    float wv = wm[(size_t)y * w + x];
    size_t o = (size_t)(dy + y) * pw + dx + x;
    const int16_t* s = img + (size_t)y * istride + 3 * (size_t)x;
    int16_t* d = dst + o * 3;
    d[0] = (int16_t)(d[0] + (int16_t)((float)s[0] * wv));
    d[1] = (int16_t)(d[1] + (int16_t)((float)s[1] * wv));
    d[2] = (int16_t)(d[2] + (int16_t)((float)s[2] * wv));
    dwgt[o] += wv;
}

inline dim3 grid2d(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }

int release(MisBlender* b, bool free_memory) {
    MisContext* ctx = b->ctx;
    if (free_memory && b->pano_mem) {
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MIS_HIP(ctx, hipFree(b->pano_mem));
        b->pano_mem = nullptr; b->pano_bytes = 0;
    }
    b->prepared = false;
    return MIS_OK;
}

int ensure_scratch(MisBlender* b, size_t bytes) {
    MisContext* ctx = b->ctx;
    if (bytes <= b->scratch_bytes) return MIS_OK;
    if (b->scratch) {
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MIS_HIP(ctx, hipFree(b->scratch));
        b->scratch = nullptr; b->scratch_bytes = 0;
    }
    MIS_HIP(ctx, hipMalloc(&b->scratch, bytes));
    b->scratch_bytes = bytes;
    return MIS_OK;
}

// MultiBandBlender::feed's padded tile of a frame at `tl` (w x h): gap 3 * 2^bands, clipped to the padded panorama roi,
// corners snapped to multiples of 2^bands (relative to the roi), shifted back when the rounding overshoots
void feed_tile_rect(const MisBlender* b, int w, int h, MisPoint tl, int* otnx, int* otny, int* owidth, int* oheight) {
    const int nb = b->num_bands, q = 1 << nb, gap = 3 * q;
    const MisRect& R = b->roi;
    int brx_roi = R.x + R.width, bry_roi = R.y + R.height;
    int tnx = std::max(R.x, tl.x - gap), tny = std::max(R.y, tl.y - gap);
    int bnx = std::min(brx_roi, tl.x + w + gap), bny = std::min(bry_roi, tl.y + h + gap);
    tnx = R.x + (((tnx - R.x) >> nb) << nb);
    tny = R.y + (((tny - R.y) >> nb) << nb);
    int width = bnx - tnx, height = bny - tny;
    width += (q - width % q) % q;
    height += (q - height % q) % q;
    bnx = tnx + width; bny = tny + height;
    int dy = std::max(bny - bry_roi, 0), dx = std::max(bnx - brx_roi, 0);
    tnx -= dx; tny -= dy;
    *otnx = tnx; *otny = tny; *owidth = width; *oheight = height;
}

int feed_multiband(MisBlender* b, const DevImage& dimg, const DevImage& dmask, int w, int h, MisPoint tl) {
    MisContext* ctx = b->ctx;
    const int nb = b->num_bands;
    const MisRect& R = b->roi;
    int tnx, tny, width, height;
    feed_tile_rect(b, w, h, tl, &tnx, &tny, &width, &height);
    const int bnx = tnx + width, bny = tny + height;
    FrameView v;
    v.img = (const int16_t*)dimg.data; v.istride = dimg.stride / 2;
    v.mask = (const uint8_t*)dmask.data; v.mstride = dmask.stride;
    v.w = w; v.h = h; v.left = tl.x - tnx; v.top = tl.y - tny; v.tw = width; v.th = height;
    int bottom = bny - tl.y - h, right = bnx - tl.x - w;
    MIS_CHECK(ctx, v.left >= 0 && v.top >= 0 && bottom >= 0 && right >= 0, MIS_E_INVALID, "frame does not fit the prepared panorama roi");

    // scratch: Gaussian levels 1..nb of the frame (16SC3) and of the weights (f32)
    int tw[MIS_MAX_BANDS + 1], th[MIS_MAX_BANDS + 1];
    size_t goff[MIS_MAX_BANDS + 1], woff[MIS_MAX_BANDS + 1], total = 0;
    tw[0] = width; th[0] = height;
    for (int i = 1; i <= nb; i++) {
        tw[i] = (tw[i - 1] + 1) / 2; th[i] = (th[i - 1] + 1) / 2;
        goff[i] = total; total += mis_align_up((size_t)tw[i] * th[i] * 6, 256);
        woff[i] = total; total += mis_align_up((size_t)tw[i] * th[i] * 4, 256);
    }
    int rc = ensure_scratch(b, total ? total : 256);
    if (rc != MIS_OK) return rc;
    auto G = [&](int i) { return (int16_t*)((uint8_t*)b->scratch + goff[i]); };
    auto W = [&](int i) { return (float*)((uint8_t*)b->scratch + woff[i]); };
    dim3 blk(256);
    // levels `first` .. nb (each at most FEED_TAIL_PIXELS pixels) go through the two tail kernels
    int first = nb + 1;
    for (int i = nb; i >= 2 && (size_t)tw[i] * th[i] <= FEED_TAIL_PIXELS; i--) first = i;
    if (first >= nb) first = nb + 1;     // a single level is not worth it
    for (int i = 0; i < nb; i++) {
        if (i >= first) break;           // G(first + 1 ..) are built by feed_tail_build_kernel
        dim3 gp((tw[i + 1] + PD_W - 1) / PD_W, (th[i + 1] + PD_H - 1) / PD_H);
        if (i == 0) {
            hipLaunchKernelGGL((pyr_down_view_kernel<true>), gp, blk, 0, ctx->stream, v, nullptr, nullptr, 0, 0, G(1), W(1), tw[1], th[1]);
        } else {
            hipLaunchKernelGGL((pyr_down_view_kernel<false>), gp, blk, 0, ctx->stream, v, (const int16_t*)G(i), (const float*)W(i), tw[i], th[i], G(i + 1), W(i + 1),
                               tw[i + 1], th[i + 1]);
        }
    }
    int y_tl = tny - R.y, x_tl = tnx - R.x, y_br = bny - R.y, x_br = bnx - R.x;
    // level 0 reads the frame view; every other level goes through one multi-level grid (feed_tail_accumulate_kernel)
    FeedTail ft;
    ft.first = first; ft.nb = nb; ft.acc_first = 1;
    for (int i = 0; i <= nb; i++) {
        if (i >= 1) {
            ft.tw[i] = tw[i]; ft.th[i] = th[i]; ft.G[i] = G(i); ft.W[i] = W(i); ft.lap[i] = b->lap[i]; ft.wgt[i] = b->wgt[i]; ft.pw[i] = b->lw[i];
            ft.x_tl[i] = x_tl; ft.y_tl[i] = y_tl;
            x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
            continue;
        }
        const int rw = x_br - x_tl, rh = y_br - y_tl;  // equals tw[0] x th[0] (tile corners are multiples of 2^nb)
        const dim3 g = grid2d(rw, rh);
        if (nb > 0)
            hipLaunchKernelGGL(laplace_accumulate_view2x2_kernel, grid2d((rw + 1) / 2, (rh + 1) / 2), blk, 0, ctx->stream, v, rw, rh, (const int16_t*)G(1), tw[1], th[1],
                               b->lap[0], b->wgt[0], b->lw[0], x_tl, y_tl);
        else
            hipLaunchKernelGGL((laplace_accumulate_kernel<true, true>), g, blk, 0, ctx->stream, v, nullptr, nullptr, rw, rh, (const int16_t*)nullptr, 0, 0,
                               b->lap[0], b->wgt[0], b->lw[0], x_tl, y_tl);
        x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
    }
    if (first <= nb) hipLaunchKernelGGL(feed_tail_build_kernel, dim3(1), dim3(1024), 0, ctx->stream, ft);
    if (nb >= 1) {
        int nblk = 0;
        for (int i = 1; i <= nb; i++) { ft.blk_off[i] = nblk; nblk += ((i < nb ? ft.tw[i + 1] * ft.th[i + 1] : ft.tw[i] * ft.th[i]) + 255) / 256; }
        ft.blk_off[nb + 1] = nblk;
        hipLaunchKernelGGL(feed_tail_accumulate_kernel, dim3(nblk), blk, 0, ctx->stream, ft);
    }
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

int feed_feather(MisBlender* b, const DevImage& dimg, const DevImage& dmask, int w, int h, MisPoint tl) {
    MisContext* ctx = b->ctx;
    size_t n = (size_t)w * h;
    int rc = ensure_scratch(b, mis_align_up(n * 4, 256) * 2);
    if (rc != MIS_OK) return rc;
    int* d = (int*)b->scratch;
    float* wm = (float*)((uint8_t*)b->scratch + mis_align_up(n * 4, 256));
    hipLaunchKernelGGL(dist_rows_kernel, dim3((h + 63) / 64), dim3(64), 0, ctx->stream, (const uint8_t*)dmask.data, dmask.stride, w, h, d);
    hipLaunchKernelGGL(dist_cols_weight_kernel, dim3((w + 63) / 64), dim3(64), 0, ctx->stream, d, w, h, b->sharpness, wm);
    hipLaunchKernelGGL(feed_feather_kernel, grid2d(w, h), dim3(256), 0, ctx->stream, (const int16_t*)dimg.data, dimg.stride / 2, wm, w, h,
                       b->lap[0], b->wgt[0], b->lw[0], tl.x - b->roi.x, tl.y - b->roi.y);
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

}  // namespace

extern "C" int mis_blend_config(int blend_type, float blend_strength, int pano_w, int pano_h, int* type_out, int* num_bands, float* sharpness) {
    if (!type_out || !num_bands || !sharpness) return MIS_E_INVALID;
    // image_stitching.cpp:1176-1190
    float blend_width = sqrtf((float)(pano_w * pano_h)) * blend_strength / 100.f;
    *num_bands = 0; *sharpness = 0.f; *type_out = blend_type;
    if (blend_width < 1.f) *type_out = MIS_BLEND_NO;
    else if (blend_type == MIS_BLEND_MULTI_BAND) *num_bands = (int)(ceil(log((double)blend_width) / log(2.)) - 1.);
    else if (blend_type == MIS_BLEND_FEATHER) *sharpness = 1.f / blend_width;
    return MIS_OK;
}

extern "C" int mis_result_roi(const MisPoint* c, const MisSize* s, int n, MisRect* roi) {
    if (!c || !s || !roi || n < 1) return MIS_E_INVALID;
    int tlx = INT32_MAX, tly = INT32_MAX, brx = INT32_MIN, bry = INT32_MIN;
    for (int i = 0; i < n; i++) {
        tlx = std::min(tlx, c[i].x); tly = std::min(tly, c[i].y);
        brx = std::max(brx, c[i].x + s[i].width); bry = std::max(bry, c[i].y + s[i].height);
    }
    roi->x = tlx; roi->y = tly; roi->width = brx - tlx; roi->height = bry - tly;
    return MIS_OK;
}

extern "C" int mis_blender_create(MisContext* ctx, int type, int num_bands, float sharpness, MisBlender** out) {
    if (!ctx || !out) return MIS_E_INVALID;
    MIS_CHECK(ctx, type == MIS_BLEND_NO || type == MIS_BLEND_FEATHER || type == MIS_BLEND_MULTI_BAND, MIS_E_INVALID, "unknown blender type %d", type);
    MisBlender* b = new MisBlender();
    b->ctx = ctx; b->type = type; b->actual_bands = num_bands; b->sharpness = sharpness;
    for (int i = 0; i <= MIS_MAX_BANDS; i++) { b->lap[i] = nullptr; b->wgt[i] = nullptr; }
    *out = b;
    return MIS_OK;
}

extern "C" int mis_blender_destroy(MisBlender* b) {
    if (!b) return MIS_OK;
    hipSetDevice(b->ctx->device);
    release(b, true);
    if (b->scratch) { hipStreamSynchronize(b->ctx->stream); hipFree(b->scratch); }
    delete b;
    return MIS_OK;
}

extern "C" int mis_blender_num_bands(const MisBlender* b) { return b ? b->num_bands : MIS_E_INVALID; }

extern "C" int mis_blender_prepare(MisBlender* b, const MisPoint* corners, const MisSize* sizes, int n) {
    if (!b) return MIS_E_INVALID;
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, corners && sizes && n >= 1, MIS_E_INVALID, "prepare needs at least one corner/size");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = release(b, false);
    if (rc != MIS_OK) return rc;
    mis_result_roi(corners, sizes, n, &b->roi);
    MIS_CHECK(ctx, b->roi.width > 0 && b->roi.height > 0, MIS_E_INVALID, "empty panorama roi");
    b->fw = b->roi.width; b->fh = b->roi.height;
    b->num_bands = 0;
    if (b->type == MIS_BLEND_MULTI_BAND) {
        // MultiBandBlender::prepare: crop unnecessary bands, pad to a multiple of 2^bands
        double max_len = (double)std::max(b->roi.width, b->roi.height);
        b->num_bands = std::min(b->actual_bands, (int)ceil(log(max_len) / log(2.0)));
        MIS_CHECK(ctx, b->num_bands >= 0 && b->num_bands <= MIS_MAX_BANDS, MIS_E_INVALID, "number of bands %d out of range", b->num_bands);
        int q = 1 << b->num_bands;
        b->roi.width += (q - b->roi.width % q) % q;
        b->roi.height += (q - b->roi.height % q) % q;
    }
    b->lw[0] = b->roi.width; b->lh[0] = b->roi.height;
    for (int i = 1; i <= b->num_bands; i++) { b->lw[i] = (b->lw[i - 1] + 1) / 2; b->lh[i] = (b->lh[i - 1] + 1) / 2; }
    size_t total = 0, loff[MIS_MAX_BANDS + 1], woff[MIS_MAX_BANDS + 1], moff = 0;
    for (int i = 0; i <= b->num_bands; i++) {
        size_t px = (size_t)b->lw[i] * b->lh[i];
        loff[i] = total; total += mis_align_up(px * 6, 256);
        woff[i] = total; total += mis_align_up(px * 4, 256);
    }
    if (b->type == MIS_BLEND_NO) { moff = total; total += mis_align_up((size_t)b->lw[0] * b->lh[0], 256); }
    if (total > b->pano_bytes) {
        if ((rc = release(b, true)) != MIS_OK) return rc;
        MIS_HIP(ctx, hipMalloc(&b->pano_mem, total));
        b->pano_bytes = total;
    }
    MIS_HIP(ctx, hipMemsetAsync(b->pano_mem, 0, total, ctx->stream));
    for (int i = 0; i <= b->num_bands; i++) {
        b->lap[i] = (int16_t*)((uint8_t*)b->pano_mem + loff[i]);
        b->wgt[i] = (float*)((uint8_t*)b->pano_mem + woff[i]);
    }
    b->dst_mask = b->type == MIS_BLEND_NO ? (uint8_t*)b->pano_mem + moff : nullptr;
    b->prepared = true;
    return MIS_OK;
}

extern "C" int mis_blender_feed(MisBlender* b, const MisImage* img, const MisImage* mask, MisPoint tl) {
    if (!b) return MIS_E_INVALID;
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, b->prepared, MIS_E_STATE, "feed before prepare");
    MIS_CHECK(ctx, img && mask && img->dtype == MIS_S16 && img->channels == 3 && mask->dtype == MIS_U8 && mask->channels == 1,
              MIS_E_INVALID, "feed needs a 16SC3 image and an 8U mask");
    MIS_CHECK(ctx, img->width == mask->width && img->height == mask->height, MIS_E_INVALID, "image / mask size mismatch");
    MIS_CHECK(ctx, tl.x >= b->roi.x && tl.y >= b->roi.y && tl.x + img->width <= b->roi.x + b->fw && tl.y + img->height <= b->roi.y + b->fh,
              MIS_E_INVALID, "frame at (%d,%d) %dx%d lies outside the prepared roi", tl.x, tl.y, img->width, img->height);
    MIS_CHECK(ctx, img->stride % 2 == 0, MIS_E_INVALID, "16SC3 stride must be even");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage di, dm;
    int rc;
    if ((rc = mis_dev_image_in(ctx, img, &di)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_in(ctx, mask, &dm)) != MIS_OK) return rc;
    const int w = img->width, h = img->height;
    if (b->type == MIS_BLEND_MULTI_BAND) rc = feed_multiband(b, di, dm, w, h, tl);
    else if (b->type == MIS_BLEND_FEATHER) rc = feed_feather(b, di, dm, w, h, tl);
    else {
        hipLaunchKernelGGL(feed_plain_kernel, grid2d(w, h), dim3(256), 0, ctx->stream, (const int16_t*)di.data, di.stride / 2,
                           (const uint8_t*)dm.data, dm.stride, w, h, b->lap[0], b->dst_mask, b->lw[0], tl.x - b->roi.x, tl.y - b->roi.y);
        rc = hipGetLastError() == hipSuccess ? MIS_OK : mis_set_error(ctx, MIS_E_HIP, "feed_plain launch failed");
    }
    int r1 = mis_dev_image_release(ctx, &di), r2 = mis_dev_image_release(ctx, &dm);
    return rc != MIS_OK ? rc : (r1 != MIS_OK ? r1 : r2);
}

// The compositing loop of main() for n frames in one call (image_stitching.cpp:1154-1164 + :1218 per frame): fused warp
// into recycled device blocks, feed, next frame.  One library call instead of 2 n keeps a host thread that drives the
// composition (e.g. concurrently with the matcher) out of the interpreter between launches.
extern "C" int mis_compose_frames(MisBlender* b, const MisImage* frames, int n, float scale, const float* Ks, const float* Rs, const MisRect* rois) {
    if (!b) return MIS_E_INVALID;
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, b->prepared, MIS_E_STATE, "compose before prepare");
    MIS_CHECK(ctx, frames && Ks && Rs && rois && n >= 0, MIS_E_INVALID, "null argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n; i++) {
        const MisRect& r = rois[i];
        MIS_CHECK(ctx, r.width > 0 && r.height > 0, MIS_E_INVALID, "frame %d: empty warp roi", i);
        const size_t ipitch = mis_align_up((size_t)r.width * 6, 256), mpitch = mis_align_up((size_t)r.width, 256);
        const size_t ibytes = ipitch * r.height, mbytes = mpitch * r.height;
        void* blk = nullptr; size_t got = 0;
        int rc = mis_pool_alloc(ctx, ibytes + mbytes, &blk, &got);
        if (rc != MIS_OK) return rc;
        MisImage img{blk, r.width, r.height, 3, ipitch, MIS_S16, MIS_MEM_DEVICE};
        MisImage msk{(uint8_t*)blk + ibytes, r.width, r.height, 1, mpitch, MIS_U8, MIS_MEM_DEVICE};
        MisPoint tl;
        rc = mis_warp_spherical_fused_roi(ctx, &frames[i], scale, Ks + 9 * i, Rs + 9 * i, &r, &img, &msk, &tl);
        if (rc == MIS_OK) rc = mis_blender_feed(b, &img, &msk, tl);
        mis_pool_free(ctx, blk, got);   // stream-ordered reuse: the next frame's warp is enqueued behind this feed
        if (rc != MIS_OK) return rc;
    }
    return MIS_OK;
}

extern "C" int mis_blender_blend(MisBlender* b, MisImage* dst, MisImage* dmask) {
    if (!b) return MIS_E_INVALID;
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, b->prepared, MIS_E_STATE, "blend before prepare");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage dd, dm;
    int rc;
    if ((rc = mis_dev_image_out(ctx, dst, b->fw, b->fh, 3, MIS_S16, &dd)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, dmask, b->fw, b->fh, 1, MIS_U8, &dm)) != MIS_OK) return rc;
    MIS_CHECK(ctx, dd.stride % 2 == 0, MIS_E_INVALID, "16SC3 stride must be even");
    const int nb = b->num_bands;
    if (b->type != MIS_BLEND_NO) {
        // coarsest level (or the single level of the feather blender): plain normalise;
        // every finer level: normalise fused with the collapse step
        size_t n = (size_t)b->lw[nb] * b->lh[nb];
        hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, b->lap[nb], b->wgt[nb], n);
        for (int i = nb; i > 0; i--) {
            if (b->lw[i - 1] == 2 * b->lw[i] && b->lh[i - 1] == 2 * b->lh[i])   // always, by the padding of prepare(); the per-pixel kernel is the general form
                hipLaunchKernelGGL(collapse2x2_kernel, grid2d(b->lw[i], b->lh[i]), dim3(256), 0, ctx->stream, b->lap[i - 1], b->wgt[i - 1], b->lw[i - 1], b->lh[i - 1],
                                   (const int16_t*)b->lap[i], b->lw[i], b->lh[i]);
            else
                hipLaunchKernelGGL(collapse_kernel, grid2d(b->lw[i - 1], b->lh[i - 1]), dim3(256), 0, ctx->stream, b->lap[i - 1], b->wgt[i - 1],
                                   b->lw[i - 1], b->lh[i - 1], b->lap[i], b->lw[i], b->lh[i]);
        }
    }
    hipLaunchKernelGGL(finalize_kernel, grid2d(b->fw, b->fh), dim3(256), 0, ctx->stream, b->lap[0], b->wgt[0], b->dst_mask, b->lw[0], b->fw,
                       b->fh, (int16_t*)dd.data, dd.stride, (uint8_t*)dm.data, dm.stride);
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dd)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_commit(ctx, dmask, &dm)) != MIS_OK) return rc;
    b->prepared = false;  // the accumulators are consumed (the reference releases them in blend())
    return MIS_OK;
}

extern "C" int mis_blender_feed_rect(const MisBlender* b, int width, int height, MisPoint tl, MisRect* tile) {
    if (!b || !tile || !b->prepared || width < 1 || height < 1) return MIS_E_INVALID;
    if (b->type != MIS_BLEND_MULTI_BAND) { tile->x = tl.x; tile->y = tl.y; tile->width = width; tile->height = height; return MIS_OK; }
    feed_tile_rect(b, width, height, tl, &tile->x, &tile->y, &tile->width, &tile->height);
    return MIS_OK;
}

extern "C" int mis_blender_level_info(const MisBlender* b, int level, int* width, int* height, void** lap_dev, void** weight_dev) {
    if (!b || level < 0 || level > b->num_bands) return MIS_E_INVALID;
    if (width) *width = b->lw[level];
    if (height) *height = b->lh[level];
    if (lap_dev) *lap_dev = b->lap[level];
    if (weight_dev) *weight_dev = b->wgt[level];
    return MIS_OK;
}
