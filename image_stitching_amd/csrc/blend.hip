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
#include <utility>
#include <vector>

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
    bool fresh = false;       // the pyramids are all zero (nothing fed or added since prepare())
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

// index maps with the common case (at most one fold) branch-free; anything further out takes the general closed form (whose
// modulo by a run-time length costs ~35 instructions: the border tiles of a 4K feed spent most of their time in it)
__device__ __forceinline__ int reflect101_near(int p, int len) {
    if ((unsigned)(p + len - 1) < (unsigned)(3 * len - 2)) { const int a = p < 0 ? -p : p; return a >= len ? 2 * len - 2 - a : a; }
    return mis_reflect101(p, len);
}
__device__ __forceinline__ int reflect_near(int p, int len) {
    if ((unsigned)(p + len) < (unsigned)(3 * len)) return mis_reflect1(p, len);
    return mis_reflect(p, len);
}

__device__ __forceinline__ void view_px(const FrameView& v, int tx, int ty, int* c) {
    const int16_t* p = v.img + (size_t)reflect_near(ty - v.top, v.h) * v.istride + 3 * (size_t)reflect_near(tx - v.left, v.w);
    c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
}
__device__ __forceinline__ float view_w(const FrameView& v, int tx, int ty) {
    int x = tx - v.left, y = ty - v.top;
    if ((unsigned)x >= (unsigned)v.w || (unsigned)y >= (unsigned)v.h) return 0.f;
    return (float)v.mask[(size_t)y * v.mstride + x] * (float)(1. / 255.);
}

// ---- 16SC3 pixel access: one 8-byte load per pixel ----
// Pixel p of a row is shorts 3p .. 3p+2: with a 4-byte aligned row start they lie inside the 8 bytes at short (3p & ~1),
// shifted by one short when p is odd.  `ok` = the row start is 4-byte aligned and reading one short past the pixel stays inside
// the allocation; otherwise three 2-byte loads.
template <bool OK>
__device__ __forceinline__ void load_px3_t(const int16_t* __restrict__ row, int p, int* c) {
    if (OK) {
        const int e = 3 * p;
        const uint2 d = *reinterpret_cast<const uint2*>(row + (e & ~1));
        const unsigned sh = (unsigned)(e & 1) << 4;
        const unsigned lo = __builtin_amdgcn_alignbit(d.y, d.x, sh);     // shorts e, e + 1 (a 64-bit shift costs four times this)
        c[0] = (int)(int16_t)(unsigned short)lo; c[1] = (int)lo >> 16; c[2] = (int)(int16_t)(unsigned short)(d.y >> sh);
    } else {
        c[0] = row[3 * p]; c[1] = row[3 * p + 1]; c[2] = row[3 * p + 2];
    }
}
__device__ __forceinline__ void load_px3(const int16_t* __restrict__ row, int p, bool ok, int* c) {
    if (ok) load_px3_t<true>(row, p, c); else load_px3_t<false>(row, p, c);
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
constexpr int PDV_ROW_DW = 104;                           // dwords of an LDS row: 67 pixels (201 shorts) from an address rounded down to 16 bytes (<= 7 shorts in) = 26 pieces of 16 bytes
static_assert(PDV_ROW_DW * 2 >= 7 + 3 * PD_SW && PDV_ROW_DW % 4 == 0, "an LDS row holds the footprint from any 16-byte phase");
constexpr int PDV_MROW_B = 96;                            // bytes of a staged mask row: 67 from an address rounded down to 16 (<= 15 in) = 6 pieces
constexpr int PDV_MROW_DW = (PD_SW + 3 + 3) / 4 + 1;      // same for the 67 mask bytes
// FROM_VIEW = false: the same pair of reductions for a level >= 1 (sources: the frame's Gaussian level `src` and weight level
// `wsrc`, sw x sh, tightly packed), one launch instead of two per level.
// one global -> LDS copy instruction (LDS address = M0 + lane * 16, global address = base + voff): inline assembly, as in warp.hip
// (s_nop 0: the wait state between the scalar write of M0 and the instruction that reads it)
__device__ __forceinline__ void pd_dma16(const void* base, unsigned voff, uint32_t lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_off) : "memory", "m0");
}

template <bool FROM_VIEW>
__device__ __forceinline__ void pyr_down_view_tile(const FrameView& v, const int16_t* __restrict__ src, const float* __restrict__ wsrc, int psw, int psh,
                                                   int16_t* __restrict__ dst, float* __restrict__ wdst, int dw, int dh, int bx, int by) {
    // LDS.  From a frame view (level 0 -> 1) the weights of the footprint are staged as the mask's BYTES (`mraw`, 96-byte rows) and
    // become floats (`wt`) only behind the image's vertical pass, in the memory of the spent `tile`: 24.6 KB per workgroup, six
    // workgroups per compute unit (with a float array of its own: 30.9 KB, five).  Levels >= 1 carry float weights from the start.
    constexpr int WT_BYTES = (int)sizeof(float) * PD_SH * PD_SW;
    static_assert(WT_BYTES <= (int)sizeof(int16_t) * PD_SH * PDV_ROW_DW * 2, "the weights fit the spent image tile");
    __shared__ __attribute__((aligned(16))) int16_t tile[PD_SH * PDV_ROW_DW * 2];   // row pitch PDV_ROW_DW dwords; pixels start `toff` shorts in
    __shared__ __attribute__((aligned(16))) uint8_t wmem[FROM_VIEW ? PD_SH * PDV_MROW_B : WT_BYTES];
    __shared__ int vbuf[PD_H * PDV_ROW_DW];   // vertical sums: 16 rows of packed u16 pairs (fast path) or 8 rows of one int per short
    __shared__ int sflag;     // DMA staging: bit 0 a mask byte of the footprint is not 255, bit 1 one is not 0, bit 2 a short is outside 0..255
    uint8_t* const mraw = wmem;                                                               // FROM_VIEW: mask bytes, row pitch PDV_MROW_B
    float* const wt = FROM_VIEW ? reinterpret_cast<float*>(tile) : reinterpret_cast<float*>(wmem);   // FROM_VIEW: valid behind convert_weights() only
    int moff = 0;             // FROM_VIEW: column of the footprint's first pixel inside a staged mask row
    bool rowchk = false;      // FROM_VIEW: rows outside the image were staged from some valid row and carry no weight (DMA staging)
    unsigned wide_bits = 0;   // bits of the staged shorts outside 0..255 (0 for a converted 8-bit image: the packed 16-bit path is exact)
    int wconst = 0;           // 1: every weight of the footprint is 1 (mask 255 everywhere), 2: every weight is 0 -- the outputs are constants
    const int x0 = bx * PD_W, y0 = by * PD_H, t = threadIdx.x;
    const int sw = FROM_VIEW ? v.tw : psw, sh = FROM_VIEW ? v.th : psh;
    const int tx0 = 2 * x0 - 2, ty0 = 2 * y0 - 2;                    // tile coordinates of the footprint's corner
    const int ix0 = tx0 - v.left, iy0 = ty0 - v.top;                 // image coordinates of the same
    const bool interior = FROM_VIEW && tx0 >= 0 && ty0 >= 0 && tx0 + PD_SW <= sw && ty0 + PD_SH <= sh && ix0 >= 0 && iy0 >= 0 && ix0 + PD_SW + 12 <= v.w && iy0 + PD_SH <= v.h &&   /* + 12: the dword loads may run past the last needed byte */
                          (v.istride & 1) == 0 && ((uintptr_t)v.img & 3) == 0 && (v.mstride & 3) == 0 && ((uintptr_t)v.mask & 3) == 0;
    int toff;   // shorts between the start of an LDS row and its first pixel
    // Tiles of a frame whose rows start on 16 bytes (the job's warped frames: 256-byte pitches) go global -> LDS by LDS-DMA: 26 + 6
    // pieces of 16 bytes per row of image and mask, no staging registers and no index arithmetic per element (the register paths
    // below spent a third of the kernel's vector instructions there).  Eligible: a footprint inside the padded tile whose 67
    // COLUMNS lie inside the image, or entirely in the left / right margin within one reflection -- there the view is the
    // mirrored image (BORDER_REFLECT): the un-mirrored block [cs, cs + 67) is staged and the tile's outputs are the block's in
    // reverse order (the 5-tap kernel is symmetric and the block's windows start at even columns: output x of the tile = output
    // 31 - x of the block), all weights are 0.  ROWS may be anything: a row's address is the reflected row's.
    bool dma = false, mir = false;
    int cs = ix0;       // image column of the staged block's first pixel
    if (FROM_VIEW && tx0 >= 0 && ty0 >= 0 && tx0 + PD_SW <= sw && ty0 + PD_SH <= sh) {
        bool cols = true;
        if (ix0 >= 0 && ix0 + PD_SW <= v.w) cs = ix0;
        else if (ix0 + PD_SW <= 0 && -ix0 <= v.w) { cs = -ix0 - PD_SW; mir = true; }
        else if (ix0 >= v.w && 2 * v.w - PD_SW - ix0 >= 0) { cs = 2 * v.w - PD_SW - ix0; mir = true; }
        else cols = false;
        const size_t ia = ((size_t)3 * cs * 2) & ~(size_t)15, ma = (size_t)cs & ~(size_t)15;
        dma = cols && (((uintptr_t)v.img | (v.istride * 2)) & 15) == 0 && ia + PDV_ROW_DW * 4 <= v.istride * 2 && (size_t)v.h * v.istride * 2 < (1ull << 32) &&
              (mir || ((((uintptr_t)v.mask | v.mstride) & 15) == 0 && ma + PDV_MROW_B <= v.mstride && (size_t)v.h * v.mstride < (1ull << 32)));
        if (!dma) mir = false;
    }
    if (dma) {
        const size_t ib = (size_t)3 * cs * 2, ia = ib & ~(size_t)15;
        toff = (int)((ib - ia) >> 1);
        moff = cs & 15;
        rowchk = true;
        const uint8_t* ibase = reinterpret_cast<const uint8_t*>(v.img) + ia;
        const uint8_t* mbase = v.mask + ((size_t)cs & ~(size_t)15);
        const uint32_t tile_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)tile;
        const uint32_t mraw_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)mraw;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
        constexpr int IPR = PDV_ROW_DW / 4, IP = PD_SH * IPR, MPR = PDV_MROW_B / 16, MP = PD_SH * MPR;      // pieces per row / in all
#pragma unroll
        for (int k = 0; k < (IP + 255) / 256; k++) {
            const int p0 = (k * 4 + wave) * 64, p = p0 + lane;       // (wave-uniform p0: the instruction's LDS base)
            if (p0 < IP) {
                const int r = min(p / IPR, PD_SH - 1), c = p - r * IPR;
                const int iy = reflect_near(iy0 + r, v.h);
                if (p < IP) pd_dma16(ibase, (unsigned)iy * (unsigned)(v.istride * 2) + (unsigned)c * 16u, tile_lds + (uint32_t)p0 * 16u);
            }
        }
        if (!mir) {
            const int p0 = wave * 64, p = p0 + lane;
            if (p0 < MP) {
                const int r = min(p / MPR, PD_SH - 1), c = p - r * MPR;
                const int iy = min(max(iy0 + r, 0), v.h - 1);         // (rows outside the image carry no weight: any valid row will do)
                if (p < MP) pd_dma16(mbase, (unsigned)iy * (unsigned)v.mstride + (unsigned)c * 16u, mraw_lds + (uint32_t)p0 * 16u);
            }
        }
        if (t == 0) sflag = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the footprint's mask bytes: all 255 (the inside of a warped frame: every weight 1) or all 0?  Dwords of the staged rows, bytes
        // outside columns moff .. moff + 66 ignored.  And the bits of the footprint's shorts outside 0..255 (the register paths collect
        // them while staging).  One LDS word gathers the three answers (most workgroups never touch it: an 8-bit frame's inside).
        unsigned all1 = mir ? 0u : 0xffffffffu, any = 0u, wide = 0u;      // (mirrored columns: outside the image, every weight 0)
        if (!mir) {
            for (int i = t; i < PD_SH * (PDV_MROW_B / 4); i += 256) {
                const int r = i / (PDV_MROW_B / 4), c = i - r * (PDV_MROW_B / 4);
                const bool row_in = (unsigned)(iy0 + r) < (unsigned)v.h;         // a row outside the image: weight 0 whatever was staged
                const unsigned d = row_in ? reinterpret_cast<const unsigned*>(mraw)[i] : 0u;
                unsigned in = 0u;
#pragma unroll
                for (int q = 0; q < 4; q++) in |= ((unsigned)(4 * c + q - moff) < (unsigned)PD_SW) ? (0xffu << (8 * q)) : 0u;
                all1 &= d | ~in;
                any |= d & in;
            }
        }
        const int td0 = toff >> 1;
        constexpr int FDW = (3 * PD_SW + 1) / 2 + 1;      // dwords that hold a row's 201 shorts from either half-dword phase
        for (int i = t; i < PD_SH * FDW; i += 256) {
            const int r = i / FDW, c = i - r * FDW;
            wide |= reinterpret_cast<const unsigned*>(tile)[r * PDV_ROW_DW + td0 + c];
        }
        // (one LDS atomic per wave at most: 256 lanes on one word serialise)
        const int bits = (__any(all1 != 0xffffffffu) ? 1 : 0) | (__any(any != 0u) ? 2 : 0) | (__any((wide & 0xFF00FF00u) != 0u) ? 4 : 0);
        if (bits && lane == 0) atomicOr(&sflag, bits);
        __syncthreads();
        const int fl = sflag;
        wconst = !(fl & 1) ? 1 : (!(fl & 2) ? 2 : 0);
        wide_bits = (fl & 4) ? 0xFF00u : 0u;
    } else if (interior) {
        const size_t e0 = 3 * (size_t)ix0;                           // first short of a row, relative to the row start
        toff = (int)(e0 & 1);
        const int16_t* base = v.img + (size_t)iy0 * v.istride + (e0 - toff);
        // all loads of a thread are issued before the first LDS store (a load -> store loop serialises the round trips)
        constexpr int NI = (PD_SH * PDV_ROW_DW + 255) / 256, NM = (PD_SH * PDV_MROW_DW + 255) / 256;
        moff = ix0 & 3;
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
            if (i < PD_SH * PDV_ROW_DW) { reinterpret_cast<unsigned*>(tile)[i] = ri[k]; wide_bits |= ri[k]; }
        }
#pragma unroll
        for (int k = 0; k < NM; k++) {
            const int i = t + 256 * k, r = i / PDV_MROW_DW, c = i - r * PDV_MROW_DW;
            if (i < PD_SH * PDV_MROW_DW) reinterpret_cast<unsigned*>(mraw)[r * (PDV_MROW_B / 4) + c] = rm[k];
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
            if (i < PD_SH * PDV_ROW_DW) { reinterpret_cast<unsigned*>(tile)[i] = ri[k]; wide_bits |= ri[k]; }
        }
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const int i = t + 256 * k;
            if (i < PD_SH * PD_SW) wt[i] = rw[k];
        }
    } else {
        toff = 0;
        for (int i = t; i < PD_SH * (PDV_ROW_DW * 2 - 3 * PD_SW); i += 256) {   // the slack shorts behind the 67 pixels of a row
            const int r = i / (PDV_ROW_DW * 2 - 3 * PD_SW), c = i - r * (PDV_ROW_DW * 2 - 3 * PD_SW);
            tile[(size_t)r * (PDV_ROW_DW * 2) + 3 * PD_SW + c] = 0;
        }
        // border tiles (40 % of a 4K frame's padded tile): all loads of a thread before its first LDS store, as above
        constexpr int NG = (PD_SH * PD_SW + 255) / 256;
        int16_t gp[NG][3];
        float gw[NG];
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const int i = min(t + 256 * k, PD_SH * PD_SW - 1);
            const int r = i / PD_SW, c = i - r * PD_SW;
            const int sy = reflect101_near(ty0 + r, sh), sx = reflect101_near(tx0 + c, sw);
            if (FROM_VIEW) {
                const int iy = sy - v.top, ix = sx - v.left;
                const int16_t* q = v.img + (size_t)reflect_near(iy, v.h) * v.istride + 3 * (size_t)reflect_near(ix, v.w);
                gp[k][0] = q[0]; gp[k][1] = q[1]; gp[k][2] = q[2];
                const bool in = (unsigned)ix < (unsigned)v.w && (unsigned)iy < (unsigned)v.h;
                gw[k] = in ? (float)v.mask[(size_t)(in ? iy : 0) * v.mstride + (in ? ix : 0)] : -1.f;
            } else {
                const int16_t* q = src + ((size_t)sy * sw + sx) * 3;
                gp[k][0] = q[0]; gp[k][1] = q[1]; gp[k][2] = q[2];
                gw[k] = wsrc[(size_t)sy * sw + sx];
            }
        }
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const int i = t + 256 * k;
            if (i < PD_SH * PD_SW) {
                const int r = i / PD_SW, c = i - r * PD_SW;
                int16_t* o = tile + (size_t)r * (PDV_ROW_DW * 2) + 3 * c;
                o[0] = gp[k][0]; o[1] = gp[k][1]; o[2] = gp[k][2];
                wide_bits |= (unsigned)(unsigned short)gp[k][0] | (unsigned)(unsigned short)gp[k][1] | (unsigned)(unsigned short)gp[k][2];
                if (FROM_VIEW) mraw[r * PDV_MROW_B + c] = gw[k] < 0.f ? (uint8_t)0 : (uint8_t)gw[k];     // (moff = 0; outside the image: weight 0)
                else wt[i] = gw[k];
            }
        }
    }
    // every short of the footprint in 0..255 (block-uniform): vertical sums <= 4080 and the full 5 x 5 sums <= 65280 fit 16 bits,
    // so the image goes through packed 16-bit arithmetic on the raw dwords; any other data takes 32-bit sums
    const bool small = dma ? wide_bits == 0 : __syncthreads_or((int)(wide_bits & 0xFF00FF00u)) == 0;      // (dma: block-uniform already)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const unsigned* tw32 = reinterpret_cast<const unsigned*>(tile);
    // a thread owns two adjacent outputs (x even) of one row: 7 source pixels = 21 shorts of the row of vertical sums
    const int j = t >> 4, x = 2 * (t & 15);
    int gx = x0 + x;
    const int gy = y0 + j;
    int o[2][3];
    if (small) {
        // pass 1 (image, vertical first -- integer sums, the pass order is free): V[j][.] = 6 r[2j+2] + 4 (r[2j+1] + r[2j+3]) + r[2j] + r[2j+4]
        for (int item = t; item < PD_H * PDV_ROW_DW; item += 256) {
            const int jj = item / PDV_ROW_DW, d = item - jj * PDV_ROW_DW;
            const unsigned* q = tw32 + (size_t)(2 * jj) * PDV_ROW_DW + d;
            const us2 r0 = __builtin_bit_cast(us2, q[0]), r1 = __builtin_bit_cast(us2, q[PDV_ROW_DW]), r2 = __builtin_bit_cast(us2, q[2 * PDV_ROW_DW]),
                      r3 = __builtin_bit_cast(us2, q[3 * PDV_ROW_DW]), r4 = __builtin_bit_cast(us2, q[4 * PDV_ROW_DW]);
            const us2 sum = r2 * (unsigned short)6 + (r1 + r3) * (unsigned short)4 + (r0 + r4);
            vbuf[item] = (int)__builtin_bit_cast(unsigned, sum);
        }
        __syncthreads();
        // pass 2 (horizontal) on the packed sums
        const unsigned* q = reinterpret_cast<const unsigned*>(vbuf) + (size_t)j * PDV_ROW_DW + 3 * x + (toff >> 1);   // shorts toff + 6 x ..
        unsigned D[11];
        if (toff & 1) {
            unsigned e[12];
#pragma unroll
            for (int k = 0; k < 12; k++) e[k] = q[k];
#pragma unroll
            for (int k = 0; k < 11; k++) D[k] = __builtin_amdgcn_alignbit(e[k + 1], e[k], 16);
        } else {
#pragma unroll
            for (int k = 0; k < 11; k++) D[k] = q[k];
        }
        // pixel P_k = shorts 3k .. 3k+2 of D: even pixels start on a dword ((B,G) = D[3k/2], R = low half of the next), odd pixels
        // one short later (B = high half of D[(3k-1)/2], (G,R) = the next dword)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int b = 3 * p;     // dword of pixel 2p
            const us2 eBG = __builtin_bit_cast(us2, D[b + 3]) * (unsigned short)6 + (__builtin_bit_cast(us2, D[b]) + __builtin_bit_cast(us2, D[b + 6]));
            const us2 eR = __builtin_bit_cast(us2, D[b + 4]) * (unsigned short)6 + (__builtin_bit_cast(us2, D[b + 1]) + __builtin_bit_cast(us2, D[b + 7]));   // low half
            const us2 oB = __builtin_bit_cast(us2, D[b + 1]) + __builtin_bit_cast(us2, D[b + 4]);      // high half: B of the odd pixels
            const us2 oGR = __builtin_bit_cast(us2, D[b + 2]) + __builtin_bit_cast(us2, D[b + 5]);     // (G, R) of the odd pixels
            const us2 oBG = __builtin_bit_cast(us2, __builtin_amdgcn_alignbit(__builtin_bit_cast(unsigned, oGR), __builtin_bit_cast(unsigned, oB), 16));
            const us2 hBG = ((eBG + oBG * (unsigned short)4) + (unsigned short)128) >> (unsigned short)8;
            const unsigned hR = (((unsigned)eR.x + 4u * (unsigned)oGR.y) + 128u) >> 8;
            o[p][0] = hBG.x; o[p][1] = hBG.y; o[p][2] = (int)(hR & 0xffffu);
        }
    } else {
        // any 16SC3 data: 32-bit vertical sums, one int per short; vbuf holds 8 rows of them, so two rounds
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            if (half) __syncthreads();
            for (int item = t; item < (PD_H / 2) * PDV_ROW_DW * 2; item += 256) {
                const int jj = item / (PDV_ROW_DW * 2), c = item - jj * (PDV_ROW_DW * 2);
                const int16_t* q = tile + (size_t)(2 * (jj + 8 * half)) * (PDV_ROW_DW * 2) + c;
                vbuf[item] = (int)q[2 * PDV_ROW_DW * 2] * 6 + ((int)q[PDV_ROW_DW * 2] + (int)q[3 * PDV_ROW_DW * 2]) * 4 + (int)q[0] + (int)q[4 * PDV_ROW_DW * 2];
            }
            __syncthreads();
            if ((j >> 3) == half) {
                const int* q = vbuf + (size_t)(j & 7) * (PDV_ROW_DW * 2) + toff + 6 * x;
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int* sp = q + 6 * p + c;
                        o[p][c] = (sp[6] * 6 + (sp[3] + sp[9]) * 4 + sp[0] + sp[12] + 128) >> 8;
                    }
            }
        }
    }
    if (FROM_VIEW && !wconst) {
        // the staged mask bytes -> weights (mask / 255), into the memory of the image tile: every thread is behind a barrier that
        // follows its last read of `tile` (the vertical passes above)
        for (int i = t; i < PD_SH * PD_SW; i += 256) {
            const int r = i / PD_SW, c = i - r * PD_SW;
            const bool row_in = !rowchk || (unsigned)(iy0 + r) < (unsigned)v.h;
            wt[i] = row_in ? (float)mraw[r * PDV_MROW_B + moff + c] * (float)(1. / 255.) : 0.f;
        }
        __syncthreads();
    }
    // weights: horizontal sums of the five source rows of the output row, then the vertical sum (f32: the reference's order)
    float ow[2];
    if (wconst) {
        // a footprint of equal weights (all 1: the inside of a warped frame; all 0: outside its mask): the same sums on the one value
        const float wv = wconst == 1 ? (float)255u * (float)(1. / 255.) : 0.f;
        const float hr = ((wv * 6.f + (wv + wv) * 4.f) + wv) + wv;
        ow[0] = ow[1] = (((hr * 6.f + (hr + hr) * 4.f) + hr) + hr) * (1.f / 256.f);
    } else {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            float hr[5];
#pragma unroll
            for (int r = 0; r < 5; r++) {
                const float* sw_ = wt + (2 * j + r) * PD_SW + 2 * (x + p);
                hr[r] = ((sw_[2] * 6.f + (sw_[1] + sw_[3]) * 4.f) + sw_[0]) + sw_[4];
            }
            ow[p] = (((hr[2] * 6.f + (hr[1] + hr[3]) * 4.f) + hr[0]) + hr[4]) * (1.f / 256.f);
        }
    }
    if (mir) {      // the block's outputs in reverse order: this thread's pair (x, x + 1) lands at (31 - x, 30 - x)
        gx = x0 + (PD_W - 2 - x);
#pragma unroll
        for (int c = 0; c < 3; c++) { const int tv = o[0][c]; o[0][c] = o[1][c]; o[1][c] = tv; }
        const float tw = ow[0]; ow[0] = ow[1]; ow[1] = tw;
    }
    if (gx >= dw || gy >= dh) return;
    const size_t e = (size_t)gy * dw + gx;
    int16_t* d = dst + e * 3;
    if (gx + 1 < dw && (dw & 1) == 0) {      // even level width: the six shorts start on a dword
        uint3 pk;
        pk.x = (unsigned)(unsigned short)o[0][0] | ((unsigned)(unsigned short)o[0][1] << 16);
        pk.y = (unsigned)(unsigned short)o[0][2] | ((unsigned)(unsigned short)o[1][0] << 16);
        pk.z = (unsigned)(unsigned short)o[1][1] | ((unsigned)(unsigned short)o[1][2] << 16);
        *reinterpret_cast<uint3*>(d) = pk;
        *reinterpret_cast<float2*>(wdst + e) = make_float2(ow[0], ow[1]);
    } else {
        d[0] = (int16_t)o[0][0]; d[1] = (int16_t)o[0][1]; d[2] = (int16_t)o[0][2];
        wdst[e] = ow[0];
        if (gx + 1 < dw) { d[3] = (int16_t)o[1][0]; d[4] = (int16_t)o[1][1]; d[5] = (int16_t)o[1][2]; wdst[e + 1] = ow[1]; }
    }
}

// ---- a batch of frames through the reductions ----
// The Gaussian pyramids of up to FB_MAX frames are built together: one launch per level for all of them (grid z = frame) instead
// of one per level and frame.  Below level 1 a 4K frame's levels are a few hundred workgroups: alone they leave most of the 256
// CUs idle and cost ~7 us of launch-to-launch latency each (5 launches + the tail per frame: 50 of the 163 us of a feed).
// Every frame owns one scratch region laid out for the largest tile of the batch, so the level offsets are the same for all.
constexpr int FB_MAX = 16;
struct FeedLayout {
    int nb, first;                                               // levels first + 1 .. nb are built by the tail kernel (first > nb: none)
    size_t goff[MIS_MAX_BANDS + 1], woff[MIS_MAX_BANDS + 1];     // byte offsets of G_l / W_l inside a frame's region (l >= 1)
};
struct FeedBatch {
    int n;
    uint8_t* base[FB_MAX];                                       // scratch region of frame k
    FrameView v[FB_MAX];
};
__device__ __forceinline__ int level_dim(int d, int l) { for (int i = 0; i < l; i++) d = (d + 1) >> 1; return d; }

__global__ __launch_bounds__(256) void pyr_down_l1_batch_kernel(FeedBatch fb, FeedLayout lay) {
    const int f = blockIdx.z;
    const FrameView v = fb.v[f];
    const int dw = (v.tw + 1) >> 1, dh = (v.th + 1) >> 1;
    if ((int)blockIdx.x * PD_W >= dw || (int)blockIdx.y * PD_H >= dh) return;   // the grid covers the largest frame of the batch
    pyr_down_view_tile<true>(v, nullptr, nullptr, 0, 0, (int16_t*)(fb.base[f] + lay.goff[1]), (float*)(fb.base[f] + lay.woff[1]), dw, dh, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void pyr_down_level_batch_kernel(FeedBatch fb, FeedLayout lay, int l) {   // level l -> l + 1, l >= 1
    const int f = blockIdx.z;
    const int sw = level_dim(fb.v[f].tw, l), sh = level_dim(fb.v[f].th, l), dw = (sw + 1) >> 1, dh = (sh + 1) >> 1;
    if ((int)blockIdx.x * PD_W >= dw || (int)blockIdx.y * PD_H >= dh) return;
    uint8_t* base = fb.base[f];
    FrameView none{};
    pyr_down_view_tile<false>(none, (const int16_t*)(base + lay.goff[l]), (const float*)(base + lay.woff[l]), sw, sh, (int16_t*)(base + lay.goff[l + 1]),
                              (float*)(base + lay.woff[l + 1]), dw, dh, blockIdx.x, blockIdx.y);
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
// the 3 x 3 coarse neighbourhood (9 pixel loads per block); per pixel the sums are those of pyr_up_at.  `ok`: see load_px3.
template <bool OK>
__device__ __forceinline__ void pyr_up_block(const int16_t* __restrict__ c, int cw, int ch, int X, int Y, int (*up)[3]) {
    const int xm = X > 0 ? X - 1 : (cw > 1 ? 1 : 0), xp = X + 1 < cw ? X + 1 : cw - 1;
    const int ym = Y > 0 ? Y - 1 : (ch > 1 ? 1 : 0), yp = Y + 1 < ch ? Y + 1 : ch - 1;
    const int rows[3] = {ym, Y, yp};
    int he[3][3], ho[3][3];   // [row][channel]: horizontal sums for an even / odd fine column
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int16_t* p = c + (unsigned)(rows[r] * cw) * 3u;   // a level of a frame's pyramid is far below 2^32 bytes
        int a[3], b[3], d[3];
        load_px3_t<OK>(p, xm, a); load_px3_t<OK>(p, X, b); load_px3_t<OK>(p, xp, d);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            he[r][q] = a[q] + b[q] * 6 + d[q];
            ho[r][q] = (b[q] + d[q]) * 4;
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

// ---- the small levels of the frames' pyramids in one launch ----
// From some level on a frame's pyramid has a few thousand pixels and every per-level launch costs more in launch-to-launch
// latency than in work.  feed_tail_build_kernel builds all the remaining Gaussian levels of a frame in ONE workgroup (a level
// depends on the previous one: __syncthreads between them; the data goes through global memory, which a workgroup sees
// coherently), one workgroup per frame of the batch.  Arithmetic: that of the per-level kernels.
__global__ __launch_bounds__(1024) void feed_tail_build_kernel(FeedBatch fb, FeedLayout lay) {   // one workgroup per frame
    const int f = blockIdx.x;
    uint8_t* base = fb.base[f];
    int sw = level_dim(fb.v[f].tw, lay.first), sh = level_dim(fb.v[f].th, lay.first);
    for (int l = lay.first; l < lay.nb; l++) {
        const int dw = (sw + 1) >> 1, dh = (sh + 1) >> 1;
        const int16_t* src = (const int16_t*)(base + lay.goff[l]);
        const float* wsrc = (const float*)(base + lay.woff[l]);
        int16_t* gdst = (int16_t*)(base + lay.goff[l + 1]);
        float* wdst = (float*)(base + lay.woff[l + 1]);
        for (int i = threadIdx.x; i < dw * dh; i += 1024) {
            const int y = i / dw, x = i - y * dw;
            int xi[5], yi[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { xi[k] = reflect101_near(2 * x - 2 + k, sw); yi[k] = reflect101_near(2 * y - 2 + k, sh); }
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
            int16_t* o = gdst + (size_t)i * 3;
            o[0] = (int16_t)((acc[0] + 128) >> 8); o[1] = (int16_t)((acc[1] + 128) >> 8); o[2] = (int16_t)((acc[2] + 128) >> 8);
            wdst[i] = (((hr[2] * 6.f + (hr[1] + hr[3]) * 4.f) + hr[0]) + hr[4]) * (1.f / 256.f);
        }
        sw = dw; sh = dh;
        __syncthreads();
    }
}
// ---- the Laplacians of a batch of frames into the panorama pyramids: gather, not scatter ----
// One grid over the bounding box of the batch's tiles in the PANORAMA pyramids, all levels.  A thread owns a 2 x 2 block of a
// panorama level (levels < nb; tile corners are multiples of 2^nb, so panorama blocks are tile blocks) and walks the frames of
// the batch in feed order: Laplacian = G_l - pyrUp(G_{l+1}) (saturating; the four pixels share pyrUp's 3 x 3 coarse
// neighbourhood), times the weight, added to the block's sums in registers -- 16-bit sums wrap (any order gives the same bits),
// the f32 weight sums are formed in frame order ((p + w_0) + w_1) + ..., exactly as n single feeds would.  The block is then
// written ONCE: per frame-pixel read-modify-write traffic of the panorama (20 B, the bulk of a scatter feed: 146 of 318 MB per
// 4K frame) becomes one 10 B write per panorama pixel -- or one read + one write when the pyramids are not known to be zero
// (`fresh` = nothing was fed since prepare()).  Pixels whose weight is exactly 0 contribute nothing (`x + (short)(v * 0) = x`,
// `w + 0 = w`): a block nobody touches is neither read nor written.  A row of a block is 12 contiguous bytes of 16SC3 + 8 of f32.
struct FeedGather {
    int n, nb, fresh;
    int bx0, by0, bw, bh;                             // bounding box of the batch's tiles, level 0, relative to the padded roi (multiples of 2^nb)
    int blk_off[MIS_MAX_BANDS + 2];                   // first workgroup of every level
    int16_t* lap[MIS_MAX_BANDS + 1];                  // panorama pyramids
    float* wgt[MIS_MAX_BANDS + 1];
    int pw[MIS_MAX_BANDS + 1];
    uint8_t* base[FB_MAX];                            // per frame: scratch region (FeedLayout), level-0 view, tile corner, 8-byte pixel loads allowed
    FrameView v[FB_MAX];
    int x_tl[FB_MAX], y_tl[FB_MAX], view_ok[FB_MAX];
};
// (Round 4 tried this loop in two phases -- the weights of four candidate frames probed at once, then the contributing frames with
// branch-free contributions, so that a lane's ~9 dependent memory round trips become ~5: 589 -> 667 us per 16 frames.  The kernel is
// bound by vector issue (66 % busy, gpurun_out/r4_pmc*), not by that latency chain: the probes and the un-skipped zero-weight lanes cost
// more than the overlap gains; a wave-uniform fast path for blocks whose weights are all exactly 1 (no float conversions: 48 fewer
// instructions per visit) measured 595 us against 589.  Kept as it was.)
// One frame's contribution to a thread's 2 x 2 block (levels < nb): returns whether the block was touched.  VIEW: level 0 (the frame
// view); FAST: the frame's / coarse level's rows allow 8-byte pixel loads.  All loads of a phase are unconditional, so that they
// are in flight together: the four weights, then (if any is non-zero) the 3 x 3 coarse pixels and the four pixels of the block.
template <bool VIEW, bool FAST>
__device__ __forceinline__ bool gather_frame(const FeedGather& a, const FeedLayout& lay, int k, int l, int px0, int py0, bool live, int (*acc)[6], float (*accw)[2]) {
    const int xt = a.x_tl[k] >> l, yt = a.y_tl[k] >> l;
    const int tw = level_dim(a.v[k].tw, l), th = level_dim(a.v[k].th, l);
    const int tx = px0 - xt, ty = py0 - yt;             // the block's corner in the frame's tile (tile sizes are even here)
    if (!live || tx < 0 || tx >= tw || ty < 0 || ty >= th) return false;
    const FrameView& v = a.v[k];
    float w[4];
    // view: image coordinates of the block's corner pixel (the block is even-aligned in the tile, not in the image)
    const int ix0 = tx - v.left, iy0 = ty - v.top;
    int cx[2], cy[2];
    if (VIEW) {
        // weights = mask / 255 inside the image, 0 outside (copyMakeBorder CONSTANT); loads at clamped coordinates, then a select
        cx[0] = min(max(ix0, 0), v.w - 1); cx[1] = min(max(ix0 + 1, 0), v.w - 1);
        cy[0] = min(max(iy0, 0), v.h - 1); cy[1] = min(max(iy0 + 1, 0), v.h - 1);
        unsigned m[4];
#pragma unroll
        for (int q = 0; q < 4; q++) m[q] = v.mask[(unsigned)cy[q >> 1] * (unsigned)v.mstride + (unsigned)cx[q & 1]];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ix = ix0 + (q & 1), iy = iy0 + (q >> 1);
            w[q] = ((unsigned)ix < (unsigned)v.w && (unsigned)iy < (unsigned)v.h) ? (float)m[q] * (float)(1. / 255.) : 0.f;
        }
    } else {
        const float* Wl = (const float*)(a.base[k] + lay.woff[l]);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float2 ww = *reinterpret_cast<const float2*>(Wl + (unsigned)((ty + r) * tw + tx));   // tw and tx even
            w[2 * r] = ww.x; w[2 * r + 1] = ww.y;
        }
    }
    if (w[0] == 0.f && w[1] == 0.f && w[2] == 0.f && w[3] == 0.f) return false;   // exact no-op contributions
    const int cw = (tw + 1) >> 1, ch = (th + 1) >> 1;
    int up[4][3], px[4][3];
    pyr_up_block<FAST>((const int16_t*)(a.base[k] + lay.goff[l + 1]), cw, ch, tx >> 1, ty >> 1, up);
    if (VIEW) {
        // a pixel with a non-zero weight lies inside the image; the others contribute nothing, so their (reflected) values
        // are never needed: clamped coordinates instead of reflected ones
#pragma unroll
        for (int q = 0; q < 4; q++) load_px3_t<FAST>(v.img + (size_t)cy[q >> 1] * v.istride, cx[q & 1], px[q]);
    } else {
        const int16_t* Gl = (const int16_t*)(a.base[k] + lay.goff[l]);
#pragma unroll
        for (int q = 0; q < 4; q++) load_px3_t<true>(Gl + (unsigned)((ty + (q >> 1)) * tw) * 3u, tx + (q & 1), px[q]);   // tw even: rows start on a dword
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float wq = w[2 * r + q];
            if (wq == 0.f) continue;        // `+ (short)(v * 0)` and `+ 0.f`: no-ops
#pragma unroll
            for (int c = 0; c < 3; c++) acc[r][3 * q + c] += (int)(int16_t)((float)sat_s16(px[2 * r + q][c] - up[2 * r + q][c]) * wq);
            accw[r][q] += wq;
        }
    }
    return true;
}
constexpr int FG_BX = 32, FG_BY = 8;                  // blocks per workgroup (64 x 16 pixels of the level)
__global__ __launch_bounds__(256) void feed_gather_kernel(FeedGather a, FeedLayout lay) {
    const int bid = (int)blockIdx.x;
    int l = 0;
    while (l < a.nb && bid >= a.blk_off[l + 1]) l++;
    const int rb = bid - a.blk_off[l];
    const int pw = a.pw[l];
    if (l < a.nb) {
        const int gbw = a.bw >> (l + 1), gbh = a.bh >> (l + 1);                  // the box in blocks
        const int nbx = (gbw + FG_BX - 1) / FG_BX;
        const int wy = rb / nbx, wx = rb - wy * nbx;
        const int X = wx * FG_BX + (int)(threadIdx.x & (FG_BX - 1)), Y = wy * FG_BY + (int)(threadIdx.x / FG_BX);
        const bool live = X < gbw && Y < gbh;
        const int rx0 = (a.bx0 >> l) + 2 * FG_BX * wx, ry0 = (a.by0 >> l) + 2 * FG_BY * wy;   // the workgroup's region of the level
        const int px0 = (a.bx0 >> l) + 2 * X, py0 = (a.by0 >> l) + 2 * Y;                       // the thread's block
        int acc[2][6];
        float accw[2][2];
        bool have = false;
        uint3* dp[2]; float2* wp[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const size_t o = (size_t)(py0 + r) * pw + px0;       // pw and px0 even: the six shorts start on a dword
            dp[r] = reinterpret_cast<uint3*>(a.lap[l] + o * 3); wp[r] = reinterpret_cast<float2*>(a.wgt[l] + o);
        }
        // the frames whose tile meets the workgroup's region: lane k of a wave tests frame k, a ballot collects the (workgroup-uniform)
        // set -- walking all FB_MAX frames with scalar loads of their rectangles cost as many scalar as vector instructions
        unsigned long long todo;
        {
            const int kk = (int)(threadIdx.x & (FB_MAX - 1));
            const int xt = a.x_tl[kk] >> l, yt = a.y_tl[kk] >> l;
            const int tw = level_dim(a.v[kk].tw, l), th = level_dim(a.v[kk].th, l);
            const bool meets = kk < a.n && !(rx0 >= xt + tw || rx0 + 2 * FG_BX <= xt || ry0 >= yt + th || ry0 + 2 * FG_BY <= yt);
            todo = __ballot(meets) & ((1ull << FB_MAX) - 1ull);
        }
        if (!todo) return;
        // the sums start from the panorama's values, or from zero when the pyramids are known to be zero (no read at all)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            uint3 d = make_uint3(0u, 0u, 0u);
            float2 ws = make_float2(0.f, 0.f);
            if (!a.fresh && live) { d = *dp[r]; ws = *wp[r]; }
            acc[r][0] = (int)(d.x & 0xffffu); acc[r][1] = (int)(d.x >> 16); acc[r][2] = (int)(d.y & 0xffffu);
            acc[r][3] = (int)(d.y >> 16); acc[r][4] = (int)(d.z & 0xffffu); acc[r][5] = (int)(d.z >> 16);
            accw[r][0] = ws.x; accw[r][1] = ws.y;
        }
        while (todo) {
            const int k = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
            todo &= todo - 1ull;
            const bool cok = (level_dim(a.v[k].tw, l + 1) & 1) == 0;     // even coarse width: its rows start on a dword
            // one instantiation per combination of (level 0 reads the frame view, 8-byte pixel loads allowed): a run-time flag inside
            // the pixel loads turns every one of them into a branch of its own and the 13 loads of a block into 13 round trips
            if (l == 0) {
                if (a.view_ok[k] && cok) have |= gather_frame<true, true>(a, lay, k, l, px0, py0, live, acc, accw);
                else have |= gather_frame<true, false>(a, lay, k, l, px0, py0, live, acc, accw);
            } else {
                if (cok) have |= gather_frame<false, true>(a, lay, k, l, px0, py0, live, acc, accw);
                else have |= gather_frame<false, false>(a, lay, k, l, px0, py0, live, acc, accw);
            }
        }
        if (!have) return;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            uint3 d;
            d.x = ((unsigned)acc[r][0] & 0xffffu) | ((unsigned)acc[r][1] << 16);
            d.y = ((unsigned)acc[r][2] & 0xffffu) | ((unsigned)acc[r][3] << 16);
            d.z = ((unsigned)acc[r][4] & 0xffffu) | ((unsigned)acc[r][5] << 16);
            *dp[r] = d;
            *wp[r] = make_float2(accw[r][0], accw[r][1]);
        }
        return;
    }
    // the last level: one pixel per thread, no pyrUp
    const int gw = a.bw >> l, gh = a.bh >> l;
    const int i = rb * 256 + (int)threadIdx.x;
    if (i >= gw * gh) return;
    const int y = i / gw, x = i - y * gw;
    const int pxx = (a.bx0 >> l) + x, pyy = (a.by0 >> l) + y;
    const size_t o = (size_t)pyy * pw + pxx;
    int16_t* d = a.lap[l] + o * 3;
    int acc3[3];
    float accw1 = 0.f;
    bool have = false;
    for (int k = 0; k < a.n; k++) {
        const int tw = level_dim(a.v[k].tw, l), th = level_dim(a.v[k].th, l);
        const int tx = pxx - (a.x_tl[k] >> l), ty = pyy - (a.y_tl[k] >> l);
        if (tx < 0 || tx >= tw || ty < 0 || ty >= th) continue;
        const size_t e = (size_t)ty * tw + tx;
        const float w = ((const float*)(a.base[k] + lay.woff[l]))[e];
        if (w == 0.f) continue;  // exact no-op contribution
        const int16_t* p = (const int16_t*)(a.base[k] + lay.goff[l]) + e * 3;
        if (!have) {
            have = true;
            if (a.fresh) { acc3[0] = acc3[1] = acc3[2] = 0; accw1 = 0.f; }
            else { acc3[0] = d[0]; acc3[1] = d[1]; acc3[2] = d[2]; accw1 = a.wgt[l][o]; }
        }
        acc3[0] += (int)(int16_t)((float)p[0] * w); acc3[1] += (int)(int16_t)((float)p[1] * w); acc3[2] += (int)(int16_t)((float)p[2] * w);
        accw1 += w;
    }
    if (!have) return;
    d[0] = (int16_t)acc3[0]; d[1] = (int16_t)acc3[1]; d[2] = (int16_t)acc3[2];
    a.wgt[l][o] = accw1;
}

// ---- blend(): normalise by the weight sum, collapse the pyramid, emit the final image + mask ----
__global__ __launch_bounds__(256) void normalize_kernel(int16_t* lap, const float* wgt, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float w = wgt[i] + WEIGHT_EPS;
    int16_t* d = lap + i * 3;
    d[0] = (int16_t)((float)d[0] / w); d[1] = (int16_t)((float)d[1] / w); d[2] = (int16_t)((float)d[2] / w);
}

// fine level (un-normalised) <- sat(pyrUp(coarse, already final) + normalise(fine)), per 2 x 2 block of the fine level (fine = 2 x coarse exactly)
// (coarse columns cx0 .. cx1 - 1 only: a rank that owns a column strip of the panorama collapses just that strip + halo)
__global__ __launch_bounds__(256) void collapse2x2_kernel(int16_t* __restrict__ fine, const float* __restrict__ fwgt, int fw, int fh, const int16_t* __restrict__ coarse,
                                                          int cw, int ch, int cx0, int cx1) {
    const int X = cx0 + blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= cx1 || Y >= ch) return;
    int up[4][3];
    if ((cw & 1) == 0) pyr_up_block<true>(coarse, cw, ch, X, Y, up); else pyr_up_block<false>(coarse, cw, ch, X, Y, up);   // uniform
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

// The last collapse step (level 1 -> level 0) fused with the crop: the collapsed level-0 pixels go straight to the result image
// (zero outside the mask) and the mask (wsum0 > eps), columns xoff .. xoff + ow - 1 and rows < oh of the padded level -- the
// level-0 Laplacian is not written back and read again (16 bytes per panorama pixel less than collapse + finalize_kernel).
// A row of a thread's 2 x 2 block is 12 contiguous bytes of 16SC3 and 8 of f32: whole dwords when `vec` (even widths / aligned rows).
__global__ __launch_bounds__(256) void collapse2x2_final_kernel(const int16_t* __restrict__ fine, const float* __restrict__ fwgt, int fw, const int16_t* __restrict__ coarse,
                                                                int cw, int ch, int cx0, int cx1, int xoff, int ow, int oh, int16_t* __restrict__ dst, size_t dstride,
                                                                uint8_t* __restrict__ dmask, size_t mstride, int vec) {
    const int X = cx0 + blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= cx1 || Y >= ch) return;
    const int ox = 2 * X - xoff;                      // output column of the block's left pixel
    if (ox + 1 < 0 || ox >= ow || 2 * Y >= oh) return;
    int up[4][3];
    if ((cw & 1) == 0) pyr_up_block<true>(coarse, cw, ch, X, Y, up); else pyr_up_block<false>(coarse, cw, ch, X, Y, up);   // uniform
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = 2 * Y + r;
        if (y >= oh) break;
        const size_t o = (size_t)y * fw + 2 * X;
        int f[6];
        float w[2];
        if (vec) {
            const uint3 d = *reinterpret_cast<const uint3*>(fine + o * 3);
            f[0] = (int16_t)(d.x & 0xffffu); f[1] = (int16_t)(d.x >> 16); f[2] = (int16_t)(d.y & 0xffffu); f[3] = (int16_t)(d.y >> 16); f[4] = (int16_t)(d.z & 0xffffu); f[5] = (int16_t)(d.z >> 16);
            const float2 ww = *reinterpret_cast<const float2*>(fwgt + o);
            w[0] = ww.x; w[1] = ww.y;
        } else {
#pragma unroll
            for (int q = 0; q < 6; q++) f[q] = fine[o * 3 + q];
            w[0] = fwgt[o]; w[1] = fwgt[o + 1];
        }
        int v[6];
        unsigned m[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            m[q] = w[q] > WEIGHT_EPS ? 255u : 0u;
            const float wn = w[q] + WEIGHT_EPS;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int val = sat_s16(up[2 * r + q][c] + (int)(int16_t)((float)f[3 * q + c] / wn));
                v[3 * q + c] = m[q] ? val : 0;
            }
        }
        int16_t* drow = (int16_t*)((uint8_t*)dst + (size_t)y * dstride);
        uint8_t* mrow = dmask + (size_t)y * mstride;
        if (vec && ox >= 0 && ox + 1 < ow) {
            uint3 pk;
            pk.x = (unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
            pk.y = (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
            pk.z = (unsigned)(unsigned short)v[4] | ((unsigned)(unsigned short)v[5] << 16);
            *reinterpret_cast<uint3*>(drow + 3 * (size_t)ox) = pk;
            *reinterpret_cast<unsigned short*>(mrow + ox) = (unsigned short)(m[0] | (m[1] << 8));
        } else {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int x = ox + q;
                if (x < 0 || x >= ow) continue;
                drow[3 * (size_t)x] = (int16_t)v[3 * q]; drow[3 * (size_t)x + 1] = (int16_t)v[3 * q + 1]; drow[3 * (size_t)x + 2] = (int16_t)v[3 * q + 2];
                mrow[x] = (uint8_t)m[q];
            }
        }
    }
}

// crop to the un-padded roi, dst_mask = wsum0 > eps (or the or-ed mask), zero outside the mask
// (columns xoff .. xoff + fw - 1 of the panorama -> columns 0 .. fw - 1 of dst)
__global__ __launch_bounds__(256) void finalize_kernel(const int16_t* lap0, const float* w0, const uint8_t* pmask, int pw, int fw, int fh,
                                                       int16_t* dst, size_t dstride, uint8_t* dmask, size_t mstride, int xoff) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= fw || y >= fh) return;
    size_t o = (size_t)y * pw + x + xoff;
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

// ---- FeatherBlender (K15): createWeightMap = min(1, sharpness * distanceTransform(mask, DIST_L1, 3)), then
// dst += (short)(src * w), wsum += w (image_stitching.cpp:1186-1190, :1218).  The 3 x 3 L1 chamfer transform is the exact city-block
// distance to the nearest zero pixel, clamped at 8192, and separates into two min-plus sweeps: along rows
//     g(x, y) = min(8192, distance to the nearest zero of row y),
// then along columns  d(x, y) = min over y' of g(x, y') + |y - y'|.  Both are run as parallel scans, all in integers (exact):
//   feather_rows_kernel    a workgroup per row: 16 pixels per thread, "last zero at or before" / "first zero at or after" by a
//                          block-wide max / min scan; g as u16 (16-byte mask loads, 32-byte stores per thread);
//   feather_colseg_kernel  columns cut into segments of FS_SEG rows: per (segment, column) the sweep's value at the segment's
//                          last row (forward) and first row (backward) when started from "infinity" -- the only thing a later
//                          segment needs to know about this one;
//   feather_feed_kernel    per (segment, 256 columns): carries from the segments above / below (a min-plus chain over their
//                          end values), the two sweeps of the segment in registers, weight = min(1, d * sharpness), and the
//                          accumulation into the panorama in the same kernel -- the weight map never exists in memory.
// Values are capped at 8192 wherever they are stored: a capped value can only reach pixels whose result is the cap anyway.
constexpr int FT_INF = 8192, FT_PX = 16, FT_TB = 256, FS_SEG = 32, FT_MAXCHUNK = 16;    // rows up to 65536 pixels

__device__ __forceinline__ int wave_scan_max_incl(int v) {     // inclusive max scan over the wave's lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if ((int)(threadIdx.x & 63) >= d) v = max(v, o); }
    return v;
}
__device__ __forceinline__ int wave_scan_min_incl_rev(int v) {  // inclusive min scan from the last lane down
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_down(v, d); if ((int)(threadIdx.x & 63) + d < 64) v = min(v, o); }
    return v;
}

__global__ __launch_bounds__(FT_TB) void feather_rows_kernel(const uint8_t* __restrict__ mask, size_t mstride, int w, int h, uint16_t* __restrict__ g, int gp) {
    __shared__ int s_last[FT_TB / 64], s_first[FT_TB / 64], s_cfirst[FT_MAXCHUNK];
    const int y = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint8_t* row = mask + (size_t)y * mstride;
    uint16_t* grow = g + (size_t)y * gp;
    const bool vec_in = (((uintptr_t)mask | mstride) & 15) == 0;
    constexpr int NONE_L = -(1 << 20), NONE_R = 1 << 20;
    // chunks of 4096 pixels, left to right; what a chunk needs from the others: the last zero before it (carried forward) and the
    // first zero behind it (found first, by a backward walk over the chunks' "first zero" values)
    const int nchunk = (w + FT_TB * FT_PX - 1) / (FT_TB * FT_PX);
    if (nchunk > 1) {       // rows wider than one chunk (8K frames): every chunk's first zero, in a pass of its own
        for (int c = t; c < FT_MAXCHUNK; c += FT_TB) s_cfirst[c] = NONE_R;
        __syncthreads();
        for (int c = 1; c < nchunk; c++) {
            const int x0 = c * FT_TB * FT_PX + t * FT_PX;
            int first = NONE_R;
            for (int i = FT_PX - 1; i >= 0; i--) if (x0 + i < w && row[x0 + i] == 0) first = x0 + i;
            first = wave_scan_min_incl_rev(first);
            if (lane == 0 && first != NONE_R) atomicMin(&s_cfirst[c], first);
        }
        __syncthreads();
    }
    int carry_last = NONE_L;
    for (int c = 0; c < nchunk; c++) {
        const int x0 = c * FT_TB * FT_PX + t * FT_PX;
        unsigned m[FT_PX];
        if (vec_in && x0 + FT_PX <= w) {
            const uint4 q = *reinterpret_cast<const uint4*>(row + x0);
            const unsigned qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < FT_PX; i++) m[i] = (qq[i >> 2] >> (8 * (i & 3))) & 255u;
        } else {
#pragma unroll
            for (int i = 0; i < FT_PX; i++) m[i] = x0 + i < w ? row[x0 + i] : 1u;      // beyond the row: not a zero
        }
        int last = NONE_L, first = NONE_R;
#pragma unroll
        for (int i = 0; i < FT_PX; i++) if (m[i] == 0 && x0 + i < w) { last = x0 + i; first = min(first, x0 + i); }
        // exclusive scans over the block: the last zero before this thread's pixels, the first zero behind them
        int il = wave_scan_max_incl(last), ir = wave_scan_min_incl_rev(first);
        if (lane == 63) s_last[wv] = il;
        if (lane == 0) s_first[wv] = ir;
        __syncthreads();
        int before = carry_last, behind = NONE_R;
        for (int k = 0; k < wv; k++) before = max(before, s_last[k]);
        for (int k = wv + 1; k < FT_TB / 64; k++) behind = min(behind, s_first[k]);
        const int pl = __shfl_up(il, 1), pr = __shfl_down(ir, 1);
        if (lane > 0) before = max(before, pl);
        if (lane < 63) behind = min(behind, pr);
        int chunk_last = carry_last;
        for (int k = 0; k < FT_TB / 64; k++) chunk_last = max(chunk_last, s_last[k]);
        for (int k = c + 1; k < nchunk; k++) behind = min(behind, s_cfirst[k]);      // zeros of the later chunks (rows wider than one chunk)
        __syncthreads();
        carry_last = chunk_last;
        // the thread's pixels: distance to the last zero at or before, the first zero at or behind
        int lz = before;
        int dl[FT_PX];
#pragma unroll
        for (int i = 0; i < FT_PX; i++) { if (m[i] == 0) lz = x0 + i; dl[i] = min(x0 + i - lz, FT_INF); }
        int rz = behind;
        unsigned out[FT_PX];
#pragma unroll
        for (int i = FT_PX - 1; i >= 0; i--) { if (m[i] == 0) rz = x0 + i; out[i] = (unsigned)min(dl[i], min(rz - (x0 + i), FT_INF)); }
        if (x0 + FT_PX <= w && (gp & 7) == 0) {
            uint4 o0, o1;
            o0.x = out[0] | (out[1] << 16); o0.y = out[2] | (out[3] << 16); o0.z = out[4] | (out[5] << 16); o0.w = out[6] | (out[7] << 16);
            o1.x = out[8] | (out[9] << 16); o1.y = out[10] | (out[11] << 16); o1.z = out[12] | (out[13] << 16); o1.w = out[14] | (out[15] << 16);
            reinterpret_cast<uint4*>(grow + x0)[0] = o0; reinterpret_cast<uint4*>(grow + x0)[1] = o1;
        } else {
#pragma unroll
            for (int i = 0; i < FT_PX; i++) if (x0 + i < w) grow[x0 + i] = (uint16_t)out[i];
        }
    }
}

// per (segment, column): F = the forward sweep's value at the segment's last row, B = the backward sweep's value at its first row,
// both started from infinity:  F = min_i g[i] + (n - 1 - i),  B = min_i g[i] + i  (capped)
__global__ __launch_bounds__(FT_TB) void feather_colseg_kernel(const uint16_t* __restrict__ g, int gp, int w, int h, uint16_t* __restrict__ F, uint16_t* __restrict__ B) {
    const int x = blockIdx.x * FT_TB + threadIdx.x, s = blockIdx.y;
    if (x >= w) return;
    const int y0 = s * FS_SEG, n = min(FS_SEG, h - y0);
    int f = FT_INF, b = FT_INF;
    unsigned v[FS_SEG];
#pragma unroll
    for (int i = 0; i < FS_SEG; i++) v[i] = g[(size_t)min(y0 + i, h - 1) * gp + x];     // all loads in flight together
#pragma unroll
    for (int i = 0; i < FS_SEG; i++) if (i < n) { f = min(f + 1, (int)v[i]); b = min(b, (int)v[i] + i); }
    F[(size_t)s * w + x] = (uint16_t)min(f, FT_INF);
    B[(size_t)s * w + x] = (uint16_t)min(b, FT_INF);
}

__global__ __launch_bounds__(FT_TB) void feather_feed_kernel(const uint16_t* __restrict__ g, int gp, const uint16_t* __restrict__ F, const uint16_t* __restrict__ B, int nseg,
                                                            const int16_t* __restrict__ img, size_t istride, int w, int h, float sharpness,
                                                            int16_t* __restrict__ dst, float* __restrict__ dwgt, int pw, int dx, int dy) {
    const int x = blockIdx.x * FT_TB + threadIdx.x, s = blockIdx.y;
    if (x >= w) return;
    const int y0 = s * FS_SEG, n = min(FS_SEG, h - y0);
    // the sweeps' values just outside the segment: forward at row y0 - 1, backward at row y0 + n
    int cf = FT_INF, cb = FT_INF;
    for (int k = 0; k < s; k++) cf = min((int)F[(size_t)k * w + x], cf + FS_SEG);                       // segments above are full
    for (int k = nseg - 1; k > s; k--) cb = min((int)B[(size_t)k * w + x], cb + min(FS_SEG, h - k * FS_SEG));
    cf = min(cf, FT_INF); cb = min(cb, FT_INF);
    int d[FS_SEG];
#pragma unroll
    for (int i = 0; i < FS_SEG; i++) d[i] = g[(size_t)min(y0 + i, h - 1) * gp + x];
    int run = cf;
#pragma unroll
    for (int i = 0; i < FS_SEG; i++) { run = min(run + 1, d[i]); d[i] = run; }
    run = cb;
#pragma unroll
    for (int i = FS_SEG - 1; i >= 0; i--) if (i < n) { run = min(run + 1, d[i]); d[i] = run; }
#pragma unroll
    for (int i = 0; i < FS_SEG; i++) {
        if (i >= n) break;
        const float wv0 = (float)min(d[i], FT_INF) * sharpness;      // createWeightMap: multiply, then THRESH_TRUNC at 1
        const float wv = wv0 > 1.f ? 1.f : wv0;
        const int y = y0 + i;
        const size_t o = (size_t)(dy + y) * pw + dx + x;
        const int16_t* sp = img + (size_t)y * istride + 3 * (size_t)x;
        int16_t* dp = dst + o * 3;
        dp[0] = (int16_t)(dp[0] + (int16_t)((float)sp[0] * wv));
        dp[1] = (int16_t)(dp[1] + (int16_t)((float)sp[1] * wv));
        dp[2] = (int16_t)(dp[2] + (int16_t)((float)sp[2] * wv));
        dwgt[o] += wv;
    }
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

// n frames into the multi-band pyramids, bit-identical to n single feeds in the same order: the Gaussian pyramids of all
// frames are built first (batched launches, see FeedBatch), then the Laplacians are accumulated frame by frame (the f32 weight
// sums keep the feed order).
int feed_multiband_batch(MisBlender* b, const MisImage* imgs, const DevImage* dimg, const DevImage* dmask, const MisPoint* tls, int n) {
    MisContext* ctx = b->ctx;
    const int nb = b->num_bands;
    const MisRect& R = b->roi;
    struct Frame { FrameView v; int x_tl, y_tl, view_ok; };
    std::vector<Frame> fr(n);
    int mtw = 0, mth = 0;
    for (int k = 0; k < n; k++) {
        const int w = imgs[k].width, h = imgs[k].height;
        int tnx, tny, width, height;
        feed_tile_rect(b, w, h, tls[k], &tnx, &tny, &width, &height);
        const int bnx = tnx + width, bny = tny + height;
        FrameView& v = fr[k].v;
        v.img = (const int16_t*)dimg[k].data; v.istride = dimg[k].stride / 2;
        v.mask = (const uint8_t*)dmask[k].data; v.mstride = dmask[k].stride;
        v.w = w; v.h = h; v.left = tls[k].x - tnx; v.top = tls[k].y - tny; v.tw = width; v.th = height;
        const int bottom = bny - tls[k].y - h, right = bnx - tls[k].x - w;
        MIS_CHECK(ctx, v.left >= 0 && v.top >= 0 && bottom >= 0 && right >= 0, MIS_E_INVALID, "frame %d does not fit the prepared panorama roi", k);
        fr[k].x_tl = tnx - R.x; fr[k].y_tl = tny - R.y;
        // rows of the frame start on a dword and one short past the last pixel of a row is readable: 8-byte pixel loads
        fr[k].view_ok = ((uintptr_t)v.img & 3) == 0 && (v.istride & 1) == 0 && v.istride >= (size_t)3 * w + 1;
        mtw = std::max(mtw, width); mth = std::max(mth, height);
    }
    dim3 blk(256);
    if (nb == 0) {
        b->fresh = false;
        for (int k = 0; k < n; k++)
            hipLaunchKernelGGL((laplace_accumulate_kernel<true, true>), grid2d(fr[k].v.tw, fr[k].v.th), blk, 0, ctx->stream, fr[k].v, nullptr, nullptr, fr[k].v.tw, fr[k].v.th,
                               (const int16_t*)nullptr, 0, 0, b->lap[0], b->wgt[0], b->lw[0], fr[k].x_tl, fr[k].y_tl);
        MIS_HIP(ctx, hipGetLastError());
        return MIS_OK;
    }
    // scratch: one region per frame, Gaussian levels 1..nb of the frame (16SC3) and of the weights (f32) for the largest tile
    FeedLayout lay;
    lay.nb = nb;
    size_t region = 0;
    {
        int tw = mtw, th = mth;
        for (int i = 1; i <= nb; i++) {
            tw = (tw + 1) / 2; th = (th + 1) / 2;
            lay.goff[i] = region; region += mis_align_up((size_t)tw * th * 6, 256);
            lay.woff[i] = region; region += mis_align_up((size_t)tw * th * 4, 256);
        }
        lay.goff[0] = lay.woff[0] = 0;
    }
    int rc = ensure_scratch(b, region ? region * n : 256);
    if (rc != MIS_OK) return rc;
    // levels `first` .. nb (each at most FEED_TAIL_PIXELS pixels) are built by the tail kernel, one workgroup per frame; the
    // batch shares one split (the largest of the frames' own: either kernel computes the same values)
    int first = 0;
    for (int k = 0; k < n; k++) {
        int tw[MIS_MAX_BANDS + 1], th[MIS_MAX_BANDS + 1];
        tw[0] = fr[k].v.tw; th[0] = fr[k].v.th;
        for (int i = 1; i <= nb; i++) { tw[i] = (tw[i - 1] + 1) / 2; th[i] = (th[i - 1] + 1) / 2; }
        int f = nb + 1;
        for (int i = nb; i >= 2 && (size_t)tw[i] * th[i] <= FEED_TAIL_PIXELS; i--) f = i;
        if (f >= nb) f = nb + 1;     // a single level is not worth it
        first = std::max(first, f);
    }
    lay.first = first;
    for (int g0 = 0; g0 < n; g0 += FB_MAX) {
        const int ng = std::min(FB_MAX, n - g0);
        FeedBatch fb;
        fb.n = ng;
        int gtw = 0, gth = 0;
        for (int k = 0; k < ng; k++) {
            fb.v[k] = fr[g0 + k].v; fb.base[k] = (uint8_t*)b->scratch + (size_t)(g0 + k) * region;
            gtw = std::max(gtw, fb.v[k].tw); gth = std::max(gth, fb.v[k].th);
        }
        for (int k = ng; k < FB_MAX; k++) { fb.v[k] = fb.v[0]; fb.base[k] = fb.base[0]; }
        int lw = (gtw + 1) / 2, lh = (gth + 1) / 2;
        hipLaunchKernelGGL(pyr_down_l1_batch_kernel, dim3((lw + PD_W - 1) / PD_W, (lh + PD_H - 1) / PD_H, ng), blk, 0, ctx->stream, fb, lay);
        for (int i = 1; i < nb && i < first; i++) {   // G(first + 1 ..) are built by feed_tail_build_kernel
            lw = (lw + 1) / 2; lh = (lh + 1) / 2;
            hipLaunchKernelGGL(pyr_down_level_batch_kernel, dim3((lw + PD_W - 1) / PD_W, (lh + PD_H - 1) / PD_H, ng), blk, 0, ctx->stream, fb, lay, i);
        }
        if (first <= nb) hipLaunchKernelGGL(feed_tail_build_kernel, dim3(ng), dim3(1024), 0, ctx->stream, fb, lay);
    }
    // the Laplacians of every group of frames are gathered into the panorama pyramids by one grid over the group's bounding box
    for (int g0 = 0; g0 < n; g0 += FB_MAX) {
        const int ng = std::min(FB_MAX, n - g0);
        FeedGather ga;
        ga.n = ng; ga.nb = nb; ga.fresh = b->fresh ? 1 : 0;
        int x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
        for (int k = 0; k < FB_MAX; k++) {
            const Frame& f = fr[g0 + (k < ng ? k : 0)];
            ga.base[k] = (uint8_t*)b->scratch + (size_t)(g0 + (k < ng ? k : 0)) * region;
            ga.v[k] = f.v; ga.x_tl[k] = f.x_tl; ga.y_tl[k] = f.y_tl; ga.view_ok[k] = f.view_ok;
            if (k < ng) { x0 = std::min(x0, f.x_tl); y0 = std::min(y0, f.y_tl); x1 = std::max(x1, f.x_tl + f.v.tw); y1 = std::max(y1, f.y_tl + f.v.th); }
        }
        ga.bx0 = x0; ga.by0 = y0; ga.bw = x1 - x0; ga.bh = y1 - y0;
        int nblk = 0;
        for (int i = 0; i <= nb; i++) {
            ga.lap[i] = b->lap[i]; ga.wgt[i] = b->wgt[i]; ga.pw[i] = b->lw[i];
            ga.blk_off[i] = nblk;
            if (i < nb) nblk += (((ga.bw >> (i + 1)) + FG_BX - 1) / FG_BX) * (((ga.bh >> (i + 1)) + FG_BY - 1) / FG_BY);
            else nblk += ((ga.bw >> i) * (ga.bh >> i) + 255) / 256;
        }
        ga.blk_off[nb + 1] = nblk;
        hipLaunchKernelGGL(feed_gather_kernel, dim3(nblk), blk, 0, ctx->stream, ga, lay);
        b->fresh = false;
    }
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

int feed_feather(MisBlender* b, const DevImage& dimg, const DevImage& dmask, int w, int h, MisPoint tl) {
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, w <= FT_MAXCHUNK * FT_TB * FT_PX, MIS_E_UNSUPPORTED, "feather blender: frames wider than %d pixels", FT_MAXCHUNK * FT_TB * FT_PX);
    const int gp = (w + 7) & ~7;                                   // u16 row pitch of g: rows start on 16 bytes
    const int nseg = (h + FS_SEG - 1) / FS_SEG;
    const size_t g_bytes = mis_align_up((size_t)gp * h * 2, 256), e_bytes = mis_align_up((size_t)nseg * w * 2, 256);
    int rc = ensure_scratch(b, g_bytes + 2 * e_bytes);
    if (rc != MIS_OK) return rc;
    uint16_t* g = (uint16_t*)b->scratch;
    uint16_t* F = (uint16_t*)((uint8_t*)b->scratch + g_bytes);
    uint16_t* B = (uint16_t*)((uint8_t*)b->scratch + g_bytes + e_bytes);
    hipLaunchKernelGGL(feather_rows_kernel, dim3(h), dim3(FT_TB), 0, ctx->stream, (const uint8_t*)dmask.data, dmask.stride, w, h, g, gp);
    const dim3 cg((w + FT_TB - 1) / FT_TB, nseg);
    hipLaunchKernelGGL(feather_colseg_kernel, cg, dim3(FT_TB), 0, ctx->stream, (const uint16_t*)g, gp, w, h, F, B);
    hipLaunchKernelGGL(feather_feed_kernel, cg, dim3(FT_TB), 0, ctx->stream, (const uint16_t*)g, gp, (const uint16_t*)F, (const uint16_t*)B, nseg,
                       (const int16_t*)dimg.data, dimg.stride / 2, w, h, b->sharpness, b->lap[0], b->wgt[0], b->lw[0], tl.x - b->roi.x, tl.y - b->roi.y);
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
    b->fresh = true;
    return MIS_OK;
}

static int feed_check(MisBlender* b, const MisImage* img, const MisImage* mask, MisPoint tl) {
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, img && mask && img->dtype == MIS_S16 && img->channels == 3 && mask->dtype == MIS_U8 && mask->channels == 1,
              MIS_E_INVALID, "feed needs a 16SC3 image and an 8U mask");
    MIS_CHECK(ctx, img->width == mask->width && img->height == mask->height, MIS_E_INVALID, "image / mask size mismatch");
    MIS_CHECK(ctx, tl.x >= b->roi.x && tl.y >= b->roi.y && tl.x + img->width <= b->roi.x + b->fw && tl.y + img->height <= b->roi.y + b->fh,
              MIS_E_INVALID, "frame at (%d,%d) %dx%d lies outside the prepared roi", tl.x, tl.y, img->width, img->height);
    MIS_CHECK(ctx, img->stride % 2 == 0, MIS_E_INVALID, "16SC3 stride must be even");
    return MIS_OK;
}

extern "C" int mis_blender_feed(MisBlender* b, const MisImage* img, const MisImage* mask, MisPoint tl) {
    if (!b) return MIS_E_INVALID;
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, b->prepared, MIS_E_STATE, "feed before prepare");
    int rc = feed_check(b, img, mask, tl);
    if (rc != MIS_OK) return rc;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage di, dm;
    if ((rc = mis_dev_image_in(ctx, img, &di)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_in(ctx, mask, &dm)) != MIS_OK) { mis_dev_image_release(ctx, &di); return rc; }
    const int w = img->width, h = img->height;
    if (b->type != MIS_BLEND_MULTI_BAND) b->fresh = false;
    if (b->type == MIS_BLEND_MULTI_BAND) rc = feed_multiband_batch(b, img, &di, &dm, &tl, 1);
    else if (b->type == MIS_BLEND_FEATHER) rc = feed_feather(b, di, dm, w, h, tl);
    else {
        hipLaunchKernelGGL(feed_plain_kernel, grid2d(w, h), dim3(256), 0, ctx->stream, (const int16_t*)di.data, di.stride / 2,
                           (const uint8_t*)dm.data, dm.stride, w, h, b->lap[0], b->dst_mask, b->lw[0], tl.x - b->roi.x, tl.y - b->roi.y);
        rc = hipGetLastError() == hipSuccess ? MIS_OK : mis_set_error(ctx, MIS_E_HIP, "feed_plain launch failed");
    }
    int r1 = mis_dev_image_release(ctx, &di), r2 = mis_dev_image_release(ctx, &dm);
    return rc != MIS_OK ? rc : (r1 != MIS_OK ? r1 : r2);
}

// n feeds in one call: same result as mis_blender_feed(imgs[0]) ... mis_blender_feed(imgs[n - 1]) in that order.  The multi-band
// blender builds the frames' pyramids together (one launch per level for all frames); the other blenders loop.
extern "C" int mis_blender_feed_batch(MisBlender* b, const MisImage* imgs, const MisImage* masks, const MisPoint* tls, int n) {
    if (!b) return MIS_E_INVALID;
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, b->prepared, MIS_E_STATE, "feed before prepare");
    MIS_CHECK(ctx, n >= 0 && (n == 0 || (imgs && masks && tls)), MIS_E_INVALID, "null argument");
    if (n == 0) return MIS_OK;
    if (b->type != MIS_BLEND_MULTI_BAND) {
        for (int k = 0; k < n; k++) { int rc = mis_blender_feed(b, &imgs[k], &masks[k], tls[k]); if (rc != MIS_OK) return rc; }
        return MIS_OK;
    }
    int rc = MIS_OK;
    for (int k = 0; k < n; k++) if ((rc = feed_check(b, &imgs[k], &masks[k], tls[k])) != MIS_OK) return rc;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<DevImage> di(n), dm(n);
    int got = 0;
    for (; got < n && rc == MIS_OK; got++) {
        if ((rc = mis_dev_image_in(ctx, &imgs[got], &di[got])) != MIS_OK) break;
        if ((rc = mis_dev_image_in(ctx, &masks[got], &dm[got])) != MIS_OK) { mis_dev_image_release(ctx, &di[got]); break; }
    }
    if (rc == MIS_OK) rc = feed_multiband_batch(b, imgs, di.data(), dm.data(), tls, n);
    for (int k = 0; k < got; k++) {
        const int r1 = mis_dev_image_release(ctx, &di[k]), r2 = mis_dev_image_release(ctx, &dm[k]);
        if (rc == MIS_OK) rc = r1 != MIS_OK ? r1 : r2;
    }
    return rc;
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
    // all warps into pool blocks of their own (one grid for up to 16 frames), then one batched feed (the frames' pyramids are built together)
    std::vector<MisImage> imgs(n), msks(n);
    std::vector<MisPoint> tls(n);
    std::vector<std::pair<void*, size_t>> blocks;
    int rc = MIS_OK;
    for (int i = 0; i < n && rc == MIS_OK; i++) {
        const MisRect& r = rois[i];
        if (!(r.width > 0 && r.height > 0)) { rc = mis_set_error(ctx, MIS_E_INVALID, "frame %d: empty warp roi", i); break; }
        const size_t ipitch = mis_align_up((size_t)r.width * 6 + 2, 256), mpitch = mis_align_up((size_t)r.width, 256);   // + 2: see view_ok
        const size_t ibytes = ipitch * r.height, mbytes = mpitch * r.height;
        void* blk = nullptr; size_t got = 0;
        if ((rc = mis_pool_alloc(ctx, ibytes + mbytes, &blk, &got)) != MIS_OK) break;
        blocks.emplace_back(blk, got);
        imgs[i] = MisImage{blk, r.width, r.height, 3, ipitch, MIS_S16, MIS_MEM_DEVICE};
        msks[i] = MisImage{(uint8_t*)blk + ibytes, r.width, r.height, 1, mpitch, MIS_U8, MIS_MEM_DEVICE};
    }
    if (rc == MIS_OK && n > 0) rc = mis_warp_spherical_fused_batch(ctx, frames, n, scale, Ks, Rs, rois, imgs.data(), msks.data(), tls.data());
    if (rc == MIS_OK) rc = mis_blender_feed_batch(b, imgs.data(), msks.data(), tls.data(), n);
    for (auto& bl : blocks) mis_pool_free(ctx, bl.first, bl.second);   // stream-ordered reuse
    return rc;
}

// blend() restricted to the panorama columns x0 .. x1 - 1 (level 0, relative to the padded roi): normalise + collapse run on
// that strip plus the halo pyrUp needs (one coarse column either side per level, accumulated: at most two), the result is the
// strip of the final image.  The whole panorama is the strip 0 .. fw.
static int blend_columns(MisBlender* b, int x0, int x1, MisImage* dst, MisImage* dmask) {
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, b->prepared, MIS_E_STATE, "blend before prepare");
    x1 = std::min(x1, b->fw);
    MIS_CHECK(ctx, x0 >= 0 && x0 < x1, MIS_E_INVALID, "empty column range %d..%d (panorama width %d)", x0, x1, b->fw);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage dd, dm;
    int rc;
    if ((rc = mis_dev_image_out(ctx, dst, x1 - x0, b->fh, 3, MIS_S16, &dd)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, dmask, x1 - x0, b->fh, 1, MIS_U8, &dm)) != MIS_OK) return rc;
    MIS_CHECK(ctx, dd.stride % 2 == 0, MIS_E_INVALID, "16SC3 stride must be even");
    const int nb = b->num_bands;
    if (b->type != MIS_BLEND_NO) {
        // columns needed in final form per level: need[0] = the strip; need[l + 1] = the coarse columns pyrUp reads for need[l]
        int lo[MIS_MAX_BANDS + 1], hi[MIS_MAX_BANDS + 1];
        lo[0] = x0; hi[0] = x1;
        for (int l = 1; l <= nb; l++) { lo[l] = std::max(0, (lo[l - 1] >> 1) - 1); hi[l] = std::min(b->lw[l], ((hi[l - 1] - 1) >> 1) + 2); }
        // coarsest level (or the single level of the feather blender): plain normalise;
        // every finer level: normalise fused with the collapse step
        size_t n = (size_t)b->lw[nb] * b->lh[nb];
        hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, b->lap[nb], b->wgt[nb], n);
        for (int i = nb; i > 0; i--) {
            MIS_CHECK(ctx, b->lw[i - 1] == 2 * b->lw[i] && b->lh[i - 1] == 2 * b->lh[i], MIS_E_STATE, "pyramid level %d is not twice level %d", i - 1, i);   // by the padding of prepare()
            const int cx0 = lo[i - 1] >> 1, cx1 = ((hi[i - 1] - 1) >> 1) + 1;    // coarse columns whose 2 x 2 blocks cover need[i - 1]
            if (i == 1) {
                // the last step writes the cropped, masked result directly
                const int vec = (b->lw[0] & 1) == 0 && (x0 & 1) == 0 && dd.stride % 4 == 0 && ((uintptr_t)dd.data & 3) == 0 && dm.stride % 2 == 0 && ((uintptr_t)dm.data & 1) == 0;
                hipLaunchKernelGGL(collapse2x2_final_kernel, grid2d(cx1 - cx0, b->lh[1]), dim3(256), 0, ctx->stream, (const int16_t*)b->lap[0], (const float*)b->wgt[0], b->lw[0],
                                   (const int16_t*)b->lap[1], b->lw[1], b->lh[1], cx0, cx1, x0, x1 - x0, b->fh, (int16_t*)dd.data, dd.stride, (uint8_t*)dm.data, dm.stride, vec);
                break;
            }
            hipLaunchKernelGGL(collapse2x2_kernel, grid2d(cx1 - cx0, b->lh[i]), dim3(256), 0, ctx->stream, b->lap[i - 1], b->wgt[i - 1], b->lw[i - 1], b->lh[i - 1],
                               (const int16_t*)b->lap[i], b->lw[i], b->lh[i], cx0, cx1);
        }
    }
    if (b->type == MIS_BLEND_NO || nb == 0)
    hipLaunchKernelGGL(finalize_kernel, grid2d(x1 - x0, b->fh), dim3(256), 0, ctx->stream, b->lap[0], b->wgt[0], b->dst_mask, b->lw[0], x1 - x0,
                       b->fh, (int16_t*)dd.data, dd.stride, (uint8_t*)dm.data, dm.stride, x0);
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dd)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_commit(ctx, dmask, &dm)) != MIS_OK) return rc;
    b->prepared = false;  // the accumulators are consumed (the reference releases them in blend())
    return MIS_OK;
}

extern "C" int mis_blender_blend(MisBlender* b, MisImage* dst, MisImage* dmask) {
    if (!b) return MIS_E_INVALID;
    return blend_columns(b, 0, b->fw, dst, dmask);
}

extern "C" int mis_blender_blend_columns(MisBlender* b, int x0, int x1, MisImage* dst, MisImage* dmask) {
    if (!b) return MIS_E_INVALID;
    return blend_columns(b, x0, x1, dst, dmask);
}

// ---- multi-GPU blend exchange: rectangles of the accumulator pyramids packed into / added from one byte buffer ----
// A rectangle (level, x0, y0, x1, y1) occupies, from `offset`: its 16SC3 Laplacian sums row after row (6 bytes per pixel, rounded
// up to 16), then its f32 weight sums (4 bytes per pixel, rounded up to 16).
struct RectBatch { int n; MisLevelRect r[MIS_MAX_BANDS + 1]; int16_t* lap[MIS_MAX_BANDS + 1]; float* wgt[MIS_MAX_BANDS + 1]; int pw[MIS_MAX_BANDS + 1]; };
template <int MODE>   // 0: pack (pyramids -> buffer), 1: add (buffer -> pyramids; 16-bit sums wrap, f32 sums in call order), 2: zero
__global__ __launch_bounds__(256) void rect_exchange_kernel(RectBatch rb, uint8_t* buf) {
    const MisLevelRect r = rb.r[blockIdx.z];
    const int w = r.x1 - r.x0, h = r.y1 - r.y0;
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (y >= h) return;
    int16_t* lap = rb.lap[blockIdx.z] + ((size_t)(r.y0 + y) * rb.pw[blockIdx.z] + r.x0) * 3;
    float* wgt = rb.wgt[blockIdx.z] + (size_t)(r.y0 + y) * rb.pw[blockIdx.z] + r.x0;
    int16_t* bl = reinterpret_cast<int16_t*>(buf + r.offset) + (size_t)y * w * 3;
    float* bw = reinterpret_cast<float*>(buf + r.offset + (((size_t)w * h * 6 + 15) & ~(size_t)15)) + (size_t)y * w;
    if (x < 3 * w) {
        if (MODE == 0) bl[x] = lap[x];
        else if (MODE == 1) lap[x] = (int16_t)(lap[x] + bl[x]);
        else lap[x] = 0;
    }
    if (x < w) {
        if (MODE == 0) bw[x] = wgt[x];
        else if (MODE == 1) wgt[x] = wgt[x] + bw[x];
        else wgt[x] = 0.f;
    }
}

static int rect_exchange(MisBlender* b, const MisLevelRect* rects, int n, void* buf, size_t bytes, int mode) {
    if (b && mode == 1) b->fresh = false;
    if (!b) return MIS_E_INVALID;
    MisContext* ctx = b->ctx;
    MIS_CHECK(ctx, b->prepared && b->type != MIS_BLEND_NO, MIS_E_STATE, "no prepared accumulator pyramids");
    MIS_CHECK(ctx, rects && n >= 0 && (buf || mode == 2), MIS_E_INVALID, "null argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i0 = 0; i0 < n; i0 += MIS_MAX_BANDS + 1) {
        RectBatch rb;
        rb.n = std::min(n - i0, MIS_MAX_BANDS + 1);
        int maxw = 0, maxh = 0;
        for (int i = 0; i < rb.n; i++) {
            const MisLevelRect& r = rects[i0 + i];
            MIS_CHECK(ctx, r.level >= 0 && r.level <= b->num_bands && r.x0 >= 0 && r.y0 >= 0 && r.x0 <= r.x1 && r.y0 <= r.y1 && r.x1 <= b->lw[r.level] && r.y1 <= b->lh[r.level],
                      MIS_E_INVALID, "rectangle %d outside level %d", i0 + i, r.level);
            const size_t px = (size_t)(r.x1 - r.x0) * (r.y1 - r.y0);
            MIS_CHECK(ctx, mode == 2 || r.offset + ((px * 6 + 15) & ~(size_t)15) + ((px * 4 + 15) & ~(size_t)15) <= bytes, MIS_E_INVALID, "rectangle %d overruns the buffer", i0 + i);
            rb.r[i] = r; rb.lap[i] = b->lap[r.level]; rb.wgt[i] = b->wgt[r.level]; rb.pw[i] = b->lw[r.level];
            maxw = std::max(maxw, r.x1 - r.x0); maxh = std::max(maxh, r.y1 - r.y0);
        }
        if (maxw == 0 || maxh == 0) continue;
        const dim3 grid((3 * maxw + 255) / 256, maxh, rb.n);
        if (mode == 0) hipLaunchKernelGGL(rect_exchange_kernel<0>, grid, dim3(256), 0, ctx->stream, rb, (uint8_t*)buf);
        else if (mode == 1) hipLaunchKernelGGL(rect_exchange_kernel<1>, grid, dim3(256), 0, ctx->stream, rb, (uint8_t*)buf);
        else hipLaunchKernelGGL(rect_exchange_kernel<2>, grid, dim3(256), 0, ctx->stream, rb, (uint8_t*)buf);
    }
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

extern "C" int mis_blender_pack_rects(MisBlender* b, const MisLevelRect* rects, int n, void* dev_buf, size_t bytes) { return rect_exchange(b, rects, n, dev_buf, bytes, 0); }
extern "C" int mis_blender_add_rects(MisBlender* b, const MisLevelRect* rects, int n, const void* dev_buf, size_t bytes) { return rect_exchange(b, rects, n, (void*)dev_buf, bytes, 1); }
extern "C" int mis_blender_zero_rects(MisBlender* b, const MisLevelRect* rects, int n) { return rect_exchange(b, rects, n, nullptr, 0, 2); }

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
    if (lap_dev || weight_dev) const_cast<MisBlender*>(b)->fresh = false;   // the caller may write through these pointers
    return MIS_OK;
}
