// orb.hip -- ORB detect + describe, replaces the reference's
// ORB::create(4000, 1.2, 8, 1, 0, 2, HARRIS_SCORE, 40, 20) (image_stitching/image_stitching.cpp:545)
// and computeImageFeatures(finder, img, features[i]) (:613).
//
// Everything stays in HBM between stages and every level of the pyramid is processed by the same
// launch (grid.z = level), so a frame costs a fixed, small number of launches:
//   gray -> [resize l = 1..n-1] -> reflect101 borders -> FAST score -> NMS + histogram ->
//   score cut (retainBest 2N) -> compaction -> Harris -> Harris cut (retainBest N) + canonical
//   ranking -> keypoint assembly + intensity-centroid angle -> Gaussian blur -> rBRIEF.
// retainBest() keeps every element >= the N-th best value; as a *set* that is order independent, so
// the cuts are found with histograms / radix selection and the canonical keypoint order
// (level, response desc, y, x) is produced by counting ranks -- no sort, no host round trip.
#include "common.h"
#include "dev_math.h"
#include <algorithm>
#include <cmath>
#include <mutex>
#include <unordered_map>

#define ORB_MAX_LEVELS 16
#define ORB_BORDER 32

namespace {

struct LevelDesc {
    int w, h;          // level size
    int pp;            // pitch of the padded buffers (bytes)
    int sp;            // pitch of the score / nms maps
    size_t pad_off;    // offset of the padded gray level inside `pad` / `blur`
    size_t map_off;    // offset of the level inside `score` / `nms`
    float scale;       // 1.2^l
    int nfeat;         // N_l
    int n2;            // retainBest #1 size (2 N_l with the Harris score)
    int tiles_x, tiles_y, tile_off;  // FAST tile grid of the level and its first slot (FT_SLOTS survivors per tile)
    int cap1, cand_off;  // candidate capacity / offset
    int cap2, fin_off;   // final capacity / offset (per level segment before concatenation)
    int tab_off;       // offset of the resize coefficient tables (x then y) of this level
};

struct Levels {
    int n;
    int fast_t, patch, half_patch, edge;
    LevelDesc d[ORB_MAX_LEVELS];
};

struct Work {  // device pointers of one MisOrb workspace
    uint8_t *pad, *blur, *score, *nms;
    int* hist;      // nlevels x 256
    int* thr;       // nlevels: FAST score cut
    int* tile_cnt;      // per tile: NMS survivors
    uint32_t* surv_xy;  // per tile slot of FT_SLOTS: x | y << 16
    uint8_t* surv_sc;   // survivor FAST score
    int* cnt1;      // nlevels: candidates written by the compaction
    int* cnt2;      // nlevels: final keypoints per level
    int* flags;     // [0] overflow flag
    uint32_t* cand_xy;  // x | y << 16
    float* cand_resp;   // FAST score, then Harris response
    uint32_t* fin_xy;
    float* fin_resp;
    int* tab;       // resize tables: per level [xofs(w) xm1(w) yofs(h) ym1(h)]
    int* umax;      // half_patch + 2
    int8_t* pattern;  // 512 x 2
};

// Frames of a batch run through every stage TOGETHER (round 3): their workspaces are identical blocks `ws` bytes apart, so frame
// f's pointer is the first frame's + f * ws, and a stage is one launch with the frame as an extra grid dimension (18 launches per
// 16 frames instead of 18 per frame on three streams: the feature stage was bound by the device's throughput on small launches).
constexpr int ORB_BATCH = 16;
struct OrbIO {      // per-frame inputs / outputs of a batch group (not at a regular stride)
    const uint8_t* src[ORB_BATCH];
    size_t sstride[ORB_BATCH];
    int aligned[ORB_BATCH];
    MisKeyPoint* kps[ORB_BATCH];
    uint32_t* lxy[ORB_BATCH];
    int* n_dev[ORB_BATCH];
    uint8_t* desc[ORB_BATCH];
};
#define WS_OFF(p, f, ws) p = (decltype(p))((const char*)(p) + (size_t)(f) * (ws))

// ---------------------------------------------------------------- K1 gray --------------------
// cvtColor BGR2GRAY, 15-bit coefficients (RGB2Gray<uchar> of OpenCV 4.x: BY15 3735, GY15 19235, RY15 9798, gray_shift 15): (B*3735 + G*19235 + R*9798 + 16384) >> 15, written into the padded level 0
constexpr int GRAY_ROWS = 4;     // rows per thread (a row per thread was 553 000 waves of 3 loads and a store per 16 frames)
__global__ __launch_bounds__(256) void gray_kernel(OrbIO io, int w, int h, uint8_t* dst, int pp, size_t ws) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = blockIdx.y * GRAY_ROWS, f = blockIdx.z;
    if (x >= w) return;
    const uint8_t* bgr = io.src[f];
    const size_t stride = io.sstride[f];
    const int aligned = io.aligned[f];
    WS_OFF(dst, f, ws);
    const int nrows = min(GRAY_ROWS, h - y0);
    if (aligned && x + 4 <= w) {
        // 12 bytes = 4 BGR pixels in three dwords; the destination is dword aligned (border 32, pitch % 64 == 0)
        unsigned w0[GRAY_ROWS], w1[GRAY_ROWS], w2[GRAY_ROWS];
#pragma unroll
        for (int r = 0; r < GRAY_ROWS; r++) {
            const unsigned* sp = reinterpret_cast<const unsigned*>(bgr + (size_t)min(y0 + r, h - 1) * stride + 3 * (size_t)x);
            w0[r] = sp[0]; w1[r] = sp[1]; w2[r] = sp[2];
        }
#pragma unroll
        for (int r = 0; r < GRAY_ROWS; r++) {
            if (r >= nrows) break;
            const unsigned g0 = ((w0[r] & 255) * 3735 + ((w0[r] >> 8) & 255) * 19235 + ((w0[r] >> 16) & 255) * 9798 + (1 << 14)) >> 15;
            const unsigned g1 = ((w0[r] >> 24) * 3735 + (w1[r] & 255) * 19235 + ((w1[r] >> 8) & 255) * 9798 + (1 << 14)) >> 15;
            const unsigned g2 = (((w1[r] >> 16) & 255) * 3735 + (w1[r] >> 24) * 19235 + (w2[r] & 255) * 9798 + (1 << 14)) >> 15;
            const unsigned g3 = (((w2[r] >> 8) & 255) * 3735 + ((w2[r] >> 16) & 255) * 19235 + (w2[r] >> 24) * 9798 + (1 << 14)) >> 15;
            *reinterpret_cast<unsigned*>(dst + (size_t)(y0 + r + ORB_BORDER) * pp + ORB_BORDER + x) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        }
    } else {
        for (int r = 0; r < nrows; r++) {
            const uint8_t* s = bgr + (size_t)(y0 + r) * stride + 3 * (size_t)x;
            uint8_t* o = dst + (size_t)(y0 + r + ORB_BORDER) * pp + ORB_BORDER + x;
            for (int k = 0; k < 4 && x + k < w; k++) o[k] = (uint8_t)((s[3 * k] * 3735 + s[3 * k + 1] * 19235 + s[3 * k + 2] * 9798 + (1 << 14)) >> 15);
        }
    }
}

// ---------------------------------------------------------------- K2 pyramid -----------------
// resize INTER_LINEAR_EXACT u8 (8.8 coefficients from host tables, single final rounding).  A thread produces four
// consecutive pixels of a row and stores them as one dword: their sources span at most nine bytes from the dword that
// holds the first one (the pyramid shrinks by 1.2 per level; larger steps take the per-pixel path), so a row costs
// three aligned dword loads instead of eight byte loads -- the byte-per-lane version was bound by the number of
// memory instructions (level 1 of a 4K frame: 84 us for 14 MB).  Table layout per level: xofs[dw4], xm1[dw4], yofs[dh],
// ym1[dh] with dw4 = dw rounded up to 4, so that the x entries of a thread are one aligned int4 each.
__device__ __forceinline__ unsigned byte_at(unsigned w0, unsigned w1, unsigned w2, int o) {   // byte o (0..11) of the 12 loaded
    const unsigned long long lo = (unsigned long long)w0 | ((unsigned long long)w1 << 32);
    return (o < 8 ? (unsigned)(lo >> (8 * o)) : (w2 >> (8 * (o - 8)))) & 255u;
}
#ifndef MIS_RS_ROWS
#define MIS_RS_ROWS 4
#endif
constexpr int RS_ROWS = MIS_RS_ROWS;     // output rows per wave
__global__ __launch_bounds__(256) void resize_kernel(const uint8_t* __restrict__ src, int sw, int sh, int spp, uint8_t* __restrict__ dst, int dw, int dh, int dpp,
                                                     const int* __restrict__ tab, size_t ws) {
    // a workgroup = 256 columns x 4 RS_ROWS rows, a wave = 256 columns x RS_ROWS rows: the column tables are loaded once for the rows,
    // every row's source loads are in flight before the first is used, and a level is a quarter of the waves a row per wave made
    // (level 1 of 16 frames was 374 000 waves of ~ 100 instructions each: started more slowly than they ran -- 123 us; 4 rows per
    // wave: 70 us, 8 rows: 68 us at 86 registers; the seven levels 358 -> 240 us)
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int ybase = __builtin_amdgcn_readfirstlane((blockIdx.y * 4 + (threadIdx.x >> 6)) * RS_ROWS);
    if (x >= dw || ybase >= dh) return;
    WS_OFF(src, blockIdx.z, ws); WS_OFF(dst, blockIdx.z, ws);
    const int dw4 = (dw + 3) & ~3;
    const int *xo = tab, *xm = tab + dw4, *yo = tab + 2 * dw4, *ym = tab + 2 * dw4 + dh;
    const int4 o4 = *reinterpret_cast<const int4*>(xo + x), m4 = *reinterpret_cast<const int4*>(xm + x);   // entries past dw are zero
    const int xs[4] = {o4.x, o4.y, o4.z, o4.w}, ms[4] = {m4.x, m4.y, m4.z, m4.w};
    const int nvalid = min(4, dw - x), nrows = min(RS_ROWS, dh - ybase);
    const int base = xs[0] & ~3, last = xs[nvalid - 1] + 1 - base;   // the right neighbour of the last column has weight 0 and lies in the padding
    const uint8_t* r0[RS_ROWS];
    const uint8_t* r1[RS_ROWS];
    int my1[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int y = min(ybase + r, dh - 1);                        // (rows past the level repeat its last row and are not stored)
        const int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : y0;
        my1[r] = ym[y];
        r0[r] = src + (size_t)(y0 + ORB_BORDER) * spp + ORB_BORDER;
        r1[r] = src + (size_t)(y1 + ORB_BORDER) * spp + ORB_BORDER;
    }
    uint8_t* out = dst + (size_t)(ybase + ORB_BORDER) * dpp + ORB_BORDER + x;
    unsigned res[RS_ROWS];
    if (last <= 11) {
        unsigned a[RS_ROWS][3], b[RS_ROWS][3];
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {
            const unsigned* p0 = reinterpret_cast<const unsigned*>(r0[r] + base);
            const unsigned* p1 = reinterpret_cast<const unsigned*>(r1[r] + base);
            a[r][0] = p0[0]; a[r][1] = p0[1]; a[r][2] = p0[2]; b[r][0] = p1[0]; b[r][1] = p1[1]; b[r][2] = p1[2];
        }
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) res[r] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int o = xs[k] - base, mx1 = ms[k], mx0 = 256 - mx1;
            const bool ok = k < nvalid;
            const int oo = ok ? o : 0;
            // bytes oo, oo + 1 of the 12 loaded (oo <= 10): one funnel shift of the dword pair that holds them
            const bool p1 = oo >= 4, p2 = oo >= 8;
#pragma unroll
            for (int r = 0; r < RS_ROWS; r++) {
                const unsigned alo = p2 ? a[r][2] : (p1 ? a[r][1] : a[r][0]), ahi = p2 ? 0u : (p1 ? a[r][2] : a[r][1]);
                const unsigned blo = p2 ? b[r][2] : (p1 ? b[r][1] : b[r][0]), bhi = p2 ? 0u : (p1 ? b[r][2] : b[r][1]);
                const unsigned ta = __builtin_amdgcn_alignbyte(ahi, alo, (unsigned)oo & 3u), tb = __builtin_amdgcn_alignbyte(bhi, blo, (unsigned)oo & 3u);
                const unsigned h0 = (ta & 255u) * mx0 + ((ta >> 8) & 255u) * mx1;
                const unsigned h1 = (tb & 255u) * mx0 + ((tb >> 8) & 255u) * mx1;
                res[r] |= ok ? ((h0 * (256 - my1[r]) + h1 * my1[r] + (1u << 15)) >> 16) << (8 * k) : 0u;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {
            res[r] = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k >= nvalid) break;
                const int x0 = xs[k], x1 = x0 + 1 < sw ? x0 + 1 : x0, mx1 = ms[k], mx0 = 256 - mx1;
                const unsigned h0 = (unsigned)r0[r][x0] * mx0 + (unsigned)r0[r][x1] * mx1;
                const unsigned h1 = (unsigned)r1[r][x0] * mx0 + (unsigned)r1[r][x1] * mx1;
                res[r] |= ((h0 * (256 - my1[r]) + h1 * my1[r] + (1u << 15)) >> 16) << (8 * k);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        if (r >= nrows) break;
        uint8_t* o = out + (size_t)r * dpp;
        if (nvalid == 4) *reinterpret_cast<unsigned*>(o) = res[r];
        else
            for (int k = 0; k < nvalid; k++) o[k] = (uint8_t)(res[r] >> (8 * k));
    }
}

// copyMakeBorder(BORDER_REFLECT_101) of every level (grid.z = level); only the border ring is visited:
// SIDES = false walks the top / bottom strips (2B rows, full padded width), SIDES = true the left /
// right strips (2B columns of the interior rows).
template <bool SIDES>
__global__ __launch_bounds__(256) void border_kernel(Levels L, uint8_t* pad, size_t ws) {
    const int f = blockIdx.z / L.n;
    const LevelDesc& d = L.d[blockIdx.z - f * L.n];
    WS_OFF(pad, f, ws);
    const int B = ORB_BORDER, pw = d.w + 2 * B, ph = d.h + 2 * B;
    uint8_t* p = pad + d.pad_off;
    if (!SIDES) {
        // four columns x four rows per thread: a dword whose columns all lie over the level's interior is the same dword of the
        // mirrored row (the row pitch and the border are multiples of 4); the dwords over the side strips go byte by byte.
        // (one dword per thread was 131 000 waves of a load and a store each: 40 us for 16 MB)
        const int px4 = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (px4 >= pw) return;
        const bool inner = px4 >= B && px4 + 3 < B + d.w;
        unsigned v[4];
        size_t drow[4], srow[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int yy = blockIdx.y * 4 + r, py = yy < B ? yy : ph - 2 * B + yy;
            srow[r] = (size_t)(mis_reflect101(py - B, d.h) + B) * d.pp; drow[r] = (size_t)py * d.pp;
        }
        if (inner) {
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = *reinterpret_cast<const unsigned*>(p + srow[r] + px4);
#pragma unroll
            for (int r = 0; r < 4; r++) *reinterpret_cast<unsigned*>(p + drow[r] + px4) = v[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                for (int k = 0; k < 4 && px4 + k < pw; k++) p[drow[r] + px4 + k] = p[srow[r] + mis_reflect101(px4 + k - B, d.w) + B];
        }
        return;
    }
    // sides: a thread = four columns of one row's left or right strip (16 threads per row; a byte per thread was 276 000 waves: 54 us)
    const int t = blockIdx.x * 256 + threadIdx.x, row = t >> 4, g = t & 15;
    if (row >= d.h) return;
    const int px = g < 8 ? 4 * g : pw - B + 4 * (g - 8);
    const size_t ro = (size_t)(row + B) * d.pp;
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (unsigned)p[ro + mis_reflect101(px + k - B, d.w) + B] << (8 * k);
    if ((px & 3) == 0) *reinterpret_cast<unsigned*>(p + ro + px) = v;      // (the right strip starts at w + B: a dword boundary when 4 | w)
    else
#pragma unroll
        for (int k = 0; k < 4; k++) p[ro + px + k] = (uint8_t)(v >> (8 * k));
}

// ---------------------------------------------------------------- K3 FAST-9/16 ---------------
// score = largest threshold for which the pixel is still a corner = max(best dark arc, best bright
// arc) - 1 (cornerScore<16>), 0 when it is not a corner at `t`.
// Necessary condition for a 9-arc on the 16-pixel circle: every run of 9 consecutive positions holds 4
// consecutive even positions (0,2,..,14), so a corner needs 4 consecutive even-position pixels all
// brighter than v + t or all darker than v - t.  Checked on the 4 compass pixels first (2 adjacent).
__device__ __forceinline__ bool fast_pretest(const uint8_t* p, int pp, int t) {
    const int v = p[0], hi = v + t, lo = v - t;
    const int c0 = p[3 * pp], c4 = p[3], c8 = p[-3 * pp], c12 = p[-3];
    unsigned br = (c0 > hi) | ((c4 > hi) << 2) | ((c8 > hi) << 4) | ((c12 > hi) << 6);   // bit k: even position 2k brighter
    unsigned dk = (c0 < lo) | ((c4 < lo) << 2) | ((c8 < lo) << 4) | ((c12 < lo) << 6);
    // two adjacent compass pixels (positions 0-4, 4-8, 8-12, 12-0) of one kind
    const unsigned adj_b = br & ((br >> 2) | (br << 6)), adj_d = dk & ((dk >> 2) | (dk << 6));
    if (!((adj_b | adj_d) & 0x55)) return false;
    const int c2 = p[2 * pp + 2], c6 = p[-2 * pp + 2], c10 = p[-2 * pp - 2], c14 = p[2 * pp - 2];
    br |= ((c2 > hi) << 1) | ((c6 > hi) << 3) | ((c10 > hi) << 5) | ((c14 > hi) << 7);
    dk |= ((c2 < lo) << 1) | ((c6 < lo) << 3) | ((c10 < lo) << 5) | ((c14 < lo) << 7);
    // 4 consecutive set bits in the circular 8-bit masks
    const unsigned b2 = br | (br << 8), d2 = dk | (dk << 8);
    const unsigned rb = b2 & (b2 >> 1) & (b2 >> 2) & (b2 >> 3), rd = d2 & (d2 >> 1) & (d2 >> 2) & (d2 >> 3);
    return ((rb | rd) & 0xff) != 0;
}
__device__ __forceinline__ int fast_score_full(const uint8_t* p, int pp, int t) {
    const int v = p[0];
    int d[16];
    d[0] = v - p[3 * pp]; d[1] = v - p[3 * pp + 1]; d[2] = v - p[2 * pp + 2]; d[3] = v - p[pp + 3];
    d[4] = v - p[3]; d[5] = v - p[-pp + 3]; d[6] = v - p[-2 * pp + 2]; d[7] = v - p[-3 * pp + 1];
    d[8] = v - p[-3 * pp]; d[9] = v - p[-3 * pp - 1]; d[10] = v - p[-2 * pp - 2]; d[11] = v - p[-pp - 3];
    d[12] = v - p[-3]; d[13] = v - p[pp - 3]; d[14] = v - p[2 * pp - 2]; d[15] = v - p[3 * pp - 1];
    // window-9 minima / maxima of the circular sequence from window-3 ones (min3 / max3 instructions)
    int n3[16], x3[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        n3[i] = min(min(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
        x3[i] = max(max(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
    }
    int A = -512, Bm = 512;  // A = max over arcs of min(d), Bm = min over arcs of max(d)
#pragma unroll
    for (int i = 0; i < 16; i++) {
        A = max(A, min(min(n3[i], n3[(i + 3) & 15]), n3[(i + 6) & 15]));
        Bm = min(Bm, max(max(x3[i], x3[(i + 3) & 15]), x3[(i + 6) & 15]));
    }
    const int best = max(A, -Bm);
    return best > t ? best - 1 : 0;
}
__device__ __forceinline__ int fast_score_px(const uint8_t* p, int pp, int t) { return fast_pretest(p, pp, t) ? fast_score_full(p, pp, t) : 0; }

// FAST-9/16 score + 3x3 strict non-max suppression of a 64 x 32 tile in one pass through LDS.
// The gray tile (+4 halo) is staged with coalesced dword loads issued up front, the scores of the tile
// (+1 halo) are computed from LDS, and the survivors go to a per-level list (x | y << 16, score) and a
// per-level histogram: the score map never exists in HBM.
#ifndef FN_ABL
#define FN_ABL 0
#endif
#ifndef MIS_FT_ROWS
#define MIS_FT_ROWS 64
#endif
constexpr int FT_COLS = 64, FT_ROWS = MIS_FT_ROWS, FG_PITCH = 96, FG_COL0 = 15, HIST_COPIES = 8;
constexpr int FT_WR = (FT_ROWS + 2 + 3) / 4;          // scored rows per wave (the tile's rows + one above and below, over four waves)
constexpr int FT_SLOTS = FT_COLS * FT_ROWS / 4;       // survivors a tile can hold: strict 3 x 3 maxima   // g: 96 columns from x0 - 16; scored column c (x = x0 - 1 + c) at g column c + FG_COL0
#ifdef MIS_ORB_STATS
__device__ unsigned long long g_orb_stats[8];   // pixels tested, opposite-pair test passes, pre-test passes, corners (score > 0), survivors
#endif
__global__ __launch_bounds__(256) void fast_nms_kernel(Levels L, const uint8_t* pad, int* hist, int* tile_cnt, uint32_t* surv_xy, uint8_t* surv_sc, size_t ws) {
    __shared__ __attribute__((aligned(16))) uint8_t g[(FT_ROWS + 10) * FG_PITCH];  // rows y0-4 .. y0+35, cols x0-16 .. x0+79 (+ two rows that are read by the last wave's column window and never used)
    __shared__ __attribute__((aligned(16))) uint8_t sc[(FT_ROWS + 2) * (FT_COLS + 4)];                          // rows y0-1 .. y0+32, cols x0-1 .. x0+64 (pitch 68)
    __shared__ int lh[256];
    __shared__ int lcount;
    __shared__ uint32_t lxy[FT_SLOTS];  // at most a quarter of the tile's pixels are strict local maxima
    __shared__ uint8_t lsc[FT_SLOTS];
    __shared__ unsigned short queue[(FT_ROWS + 2) * (FT_COLS + 2)];  // pixels that pass the cheap pre-test
    __shared__ int qcount;
    const int fz = blockIdx.z / L.n, lz = blockIdx.z - fz * L.n;     // frame, level
    const LevelDesc& d = L.d[lz];
    const int x0 = blockIdx.x * FT_COLS, y0 = blockIdx.y * FT_ROWS, t = threadIdx.x;
    if (x0 >= d.w || y0 >= d.h) return;
    WS_OFF(pad, fz, ws); WS_OFF(hist, fz, ws); WS_OFF(tile_cnt, fz, ws); WS_OFF(surv_xy, fz, ws); WS_OFF(surv_sc, fz, ws);
    lh[t] = 0;
    if (t == 0) { lcount = 0; qcount = 0; }
    {
        // the gray tile: FT_ROWS + 8 rows x 96 bytes from column x0 - 16 -- six 16-byte pieces per row, one or two pieces per thread (x0 is a multiple of
        // 64, the border 32, the pitch a multiple of 64: every piece is 16-byte aligned; dword loads from x0 - 4 were 3 - 4 per thread
        // with their index arithmetic, and the staging alone took 23 of a frame's 58 us)
        const int ph = d.h + 2 * ORB_BORDER;
        const uint8_t* src = pad + d.pad_off;
        const int px0 = x0 - 16 + ORB_BORDER, py0 = y0 - 4 + ORB_BORDER;
        for (int i = t; i < (FT_ROWS + 8) * (FG_PITCH / 16); i += 256) {
            const int r = i / (FG_PITCH / 16), c = i - r * (FG_PITCH / 16);
            const int py = min(py0 + r, ph - 1);
            int px = px0 + 16 * c;
            if (px + 15 >= d.pp) px = d.pp - 16;  // past the row: any in-buffer piece (those pixels are never scored)
            reinterpret_cast<uint4*>(g)[i] = *reinterpret_cast<const uint4*>(src + (size_t)py * d.pp + px);
        }
    }
    __syncthreads();
#if FN_ABL == 1
    if (lcount >= 0) return;      // ablation: stop behind the staging of the gray tile
#endif
    const int SP = FT_COLS + 4;
    // Stage A, every pixel of the scored region (34 rows x 66 columns): a 9-arc of the 16-pixel circle holds one of the positions
    // {0, 8} (three rows below / above) and one of {4, 12} (three columns right / left), so a corner needs
    // max(|p0 - v|, |p8 - v|) > t and max(|p4 - v|, |p12 - v|) > t.  4 % of the bench frames' pyramid pixels pass (2.7 % pass round 2's
    // pre-test of adjacent compass pixels + diagonals, 1.4 % are corners: tools/orb_stats.py), so this is where the kernel's time goes:
    // a wave owns FT_WR scored rows, a lane one column -- its FT_WR + 6 gray values are read once and slide through registers (the vertical
    // taps), the horizontal taps are two byte reads, the test is 4 v_sad + 2 max + 2 compares; the passes of the 9 rows are
    // appended to the queue with ONE LDS atomic per wave (tiles of 64 rows: half the workgroups of 32-row tiles, whose staging latency -- load, LDS write, barrier -- was a third of the kernel).  The full arc evaluation then runs on the queue in dense wavefronts.
    for (int i = t; i < (FT_ROWS + 2) * SP / 4; i += 256) reinterpret_cast<unsigned*>(sc)[i] = 0u;
    {
        const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;       // the wave index in a scalar register: row conditions are scalar
        const unsigned ft = (unsigned)L.fast_t;
        const int r0 = FT_WR * wv;                                   // first scored row of the wave (g row of scored row r: r + 3)
        bool pass[FT_WR + 1];
        unsigned long long ball[FT_WR + 1];
        int total = 0;
        {
            const uint8_t* colp = g + r0 * FG_PITCH + (lane + FG_COL0);      // g rows r0 .. r0 + FT_WR + 5 (the last wave reads two rows past the staged ones: g has FT_ROWS + 10)
            unsigned col[FT_WR + 6];
#pragma unroll
            for (int j = 0; j < FT_WR + 6; j++) col[j] = colp[j * FG_PITCH];
            const int x = x0 - 1 + lane;
            const bool xin = x >= 3 && x < d.w - 3;
            const uint8_t* rowp = g + (r0 + 3) * FG_PITCH + (lane + FG_COL0 - 3);      // the columns three to the left (+ 0) and to the right (+ 6) of scored row r0
#pragma unroll
            for (int k = 0; k < FT_WR; k++) {
                const int r = r0 + k, y = y0 - 1 + r;
                const bool rowok = r < FT_ROWS + 2 && y >= 3 && y < d.h - 3;      // uniform
                const unsigned v = col[k + 3];
                // |a - v| of bytes held in dwords: v_sad_u8 (the three upper byte lanes are zero)
                const unsigned dv = max(__builtin_amdgcn_sad_u8(col[k + 6], v, 0u), __builtin_amdgcn_sad_u8(col[k], v, 0u));
                const unsigned dh = max(__builtin_amdgcn_sad_u8(rowp[k * FG_PITCH + 6], v, 0u), __builtin_amdgcn_sad_u8(rowp[k * FG_PITCH], v, 0u));
                pass[k] = rowok & xin & (min(dv, dh) > ft);
                ball[k] = __ballot(pass[k]);
                total += __popcll(ball[k]);
            }
        }
        {
            // the two halo columns (scored columns 64, 65): lane = 2 k + side
            const int k = lane >> 1, cc = FT_COLS + (lane & 1), r = r0 + k, y = y0 - 1 + r, x = x0 - 1 + cc;
            bool ps = false;
            if (lane < 2 * FT_WR && r < FT_ROWS + 2) {
                const uint8_t* pp = g + (r + 3) * FG_PITCH + (cc + FG_COL0);
                const unsigned v = pp[0];
                const unsigned dv = max(__builtin_amdgcn_sad_u8(pp[3 * FG_PITCH], v, 0u), __builtin_amdgcn_sad_u8(pp[-3 * FG_PITCH], v, 0u));
                const unsigned dh = max(__builtin_amdgcn_sad_u8(pp[3], v, 0u), __builtin_amdgcn_sad_u8(pp[-3], v, 0u));
                ps = x >= 3 && x < d.w - 3 && y >= 3 && y < d.h - 3 && min(dv, dh) > ft;
            }
            pass[FT_WR] = ps;
            ball[FT_WR] = __ballot(ps);
            total += __popcll(ball[FT_WR]);
        }
        int base = 0;
        if (lane == 0 && total) base = atomicAdd(&qcount, total);
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < FT_WR + 1; k++) {
            if (pass[k]) {
                const int r = k < FT_WR ? r0 + k : r0 + (lane >> 1), c = k < FT_WR ? lane : FT_COLS + (lane & 1);
                queue[base + __popcll(ball[k] & below)] = (unsigned short)(r * (FT_COLS + 2) + c);
            }
            base += __popcll(ball[k]);
        }
    }
    __syncthreads();
#if FN_ABL == 2
    if (qcount >= 0) return;      // ablation: stop behind stage A
#endif
    for (int q = t; q < qcount; q += 256) {
        const int i = queue[q], r = i / (FT_COLS + 2), c = i - r * (FT_COLS + 2);
        const int fsv = fast_score_full(g + (r + 3) * FG_PITCH + (c + FG_COL0), FG_PITCH, L.fast_t);
        sc[r * SP + c] = (uint8_t)fsv;
#ifdef MIS_ORB_STATS
        if (fsv) atomicAdd(&g_orb_stats[3], 1ull);
#endif
    }
    __syncthreads();
#if FN_ABL == 3
    if (qcount >= 0) return;      // ablation (tools): stop behind the arc evaluation
#endif
    {
        // 3 x 3 strict non-max suppression of the corners, from the queue (1.4 % of the pixels have a score at all: a raster pass
        // over the tile read 9 scores per pixel for nothing).  Every lane takes part in every ballot: survivors are appended with
        // one LDS atomic per wave.
        const int nq = qcount;
        for (int q0 = 0; q0 < nq; q0 += 256) {
            const int q = q0 + t;
            bool keep = false;
            int x = 0, y = 0, v = 0;
            if (q < nq) {
                const int i = queue[q], r = i / (FT_COLS + 2), c = i - r * (FT_COLS + 2);     // scored row / column: the tile's pixels are 1 .. 32 / 1 .. 64
                const uint8_t* s_ = sc + r * SP + c;
                v = s_[0];
                x = x0 - 1 + c; y = y0 - 1 + r;
                keep = v && r >= 1 && r <= FT_ROWS && c >= 1 && c <= FT_COLS && x >= L.edge && x < d.w - L.edge && y >= L.edge && y < d.h - L.edge &&   // runByImageBorder(edgeThreshold)
                       v > s_[-1] && v > s_[1] && v > s_[-SP - 1] && v > s_[-SP] && v > s_[-SP + 1] && v > s_[SP - 1] && v > s_[SP] && v > s_[SP + 1];
            }
            const unsigned long long m = __ballot(keep);
            int base = 0;
            if ((t & 63) == 0 && m) base = atomicAdd(&lcount, __popcll(m));
            base = __shfl(base, 0);
            if (keep) {
                const int j = base + __popcll(m & ((1ull << (t & 63)) - 1ull));
                lxy[j] = (uint32_t)x | ((uint32_t)y << 16); lsc[j] = (uint8_t)v;
            }
        }
    }
    __syncthreads();
    // every tile owns a fixed slot of the survivor arrays: no global counter to contend on
    const int n = lcount;
    const int tile = d.tile_off + blockIdx.y * d.tiles_x + blockIdx.x;
    if (t == 0) tile_cnt[tile] = n;
    if (n == 0) return;  // uniform
    for (int j = t; j < n; j += 256) {
        atomicAdd(&lh[lsc[j]], 1);  // histogram from the list: spread over bins
        surv_xy[(size_t)tile * FT_SLOTS + j] = lxy[j]; surv_sc[(size_t)tile * FT_SLOTS + j] = lsc[j];
    }
    __syncthreads();
    // HIST_COPIES copies of a level's histogram, picked by the tile: the popular bins would otherwise take one atomic from
    // nearly every tile on a single address (same-address atomics serialise); fast_cut_kernel adds the copies up
    if (lh[t]) atomicAdd(&hist[((lz * HIST_COPIES) + ((blockIdx.x + blockIdx.y) & (HIST_COPIES - 1))) * 256 + t], lh[t]);
}

// retainBest(2 N_l) on the integer FAST score: cut = the n2-th best score (1 = keep everything), found
// from the level histogram by every block, then the survivor list is filtered against it
__global__ __launch_bounds__(256) void fast_cut_kernel(Levels L, const int* hist, int* thr, int* flags, size_t ws) {
    __shared__ int suf[256];
    __shared__ int s_thr;
    const int l = blockIdx.x, t = threadIdx.x;
    WS_OFF(hist, blockIdx.y, ws); WS_OFF(thr, blockIdx.y, ws); WS_OFF(flags, blockIdx.y, ws);
    const LevelDesc& d = L.d[l];
    // suffix sums of the histogram: suf[v] = number of survivors with score >= v
    int hsum = 0;
    for (int c = 0; c < HIST_COPIES; c++) hsum += hist[(l * HIST_COPIES + c) * 256 + t];
    suf[t] = t ? hsum : 0;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int add = t + o < 256 ? suf[t + o] : 0;
        __syncthreads();
        suf[t] += add;
        __syncthreads();
    }
    if (t == 0) s_thr = d.n2 == 0 ? 256 : 1;
    __syncthreads();
    // the cut is the largest v with suf[v] >= n2 (when more than n2 survivors exist)
    if (d.n2 > 0 && suf[1] > d.n2 && t >= 1 && suf[t] >= d.n2 && (t == 255 || suf[t + 1] < d.n2)) s_thr = t;
    __syncthreads();
    if (t == 0) {
        thr[l] = s_thr;
        if (s_thr < 256 && suf[s_thr] > d.cap1) atomicOr(&flags[0], 1);
    }
}

// Survivors of every tile slot are filtered against the level's cut and appended to the level's candidate list.  A
// workgroup owns CT_TILES slots (a wave walks CT_TILES / 4 of them): it counts first, reserves its range with ONE global
// atomic and then writes -- the per-level counter is a single address, and same-address atomics serialise at ~11 ns
// each (one per tile slot used to set this kernel's time: 4050 slots at level 0 of a 4K frame).  The order inside the
// list is arbitrary either way; the ranking kernel imposes the canonical one.
constexpr int CT_TILES = 16;
__global__ __launch_bounds__(256) void compact_kernel(Levels L, const int* tile_cnt, const uint32_t* surv_xy, const uint8_t* surv_sc, const int* thr,
                                                      int* cnt1, uint32_t* cand_xy, float* cand_resp, size_t ws) {
    __shared__ int s_cnt[CT_TILES], s_off[CT_TILES];
    __shared__ int s_base;
    const int l = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    WS_OFF(tile_cnt, blockIdx.z, ws); WS_OFF(surv_xy, blockIdx.z, ws); WS_OFF(surv_sc, blockIdx.z, ws); WS_OFF(thr, blockIdx.z, ws);
    WS_OFF(cnt1, blockIdx.z, ws); WS_OFF(cand_xy, blockIdx.z, ws); WS_OFF(cand_resp, blockIdx.z, ws);
    const LevelDesc& d = L.d[l];
    const int ntiles = d.tiles_x * d.tiles_y, first = blockIdx.x * CT_TILES;
    if (first >= ntiles) return;
    const int th = thr[l];
    for (int k = 0; k < CT_TILES / 4; k++) {
        const int slot = wave * (CT_TILES / 4) + k, til = first + slot;
        int cnt = 0;
        if (til < ntiles) {
            const int tile = d.tile_off + til, n = tile_cnt[tile];
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                const int v = j < n ? surv_sc[(size_t)tile * FT_SLOTS + j] : 0;
                cnt += __popcll(__ballot(v >= th && v > 0));
            }
        }
        if (lane == 0) s_cnt[slot] = cnt;
    }
    __syncthreads();
    if (t == 0) {
        int tot = 0;
        for (int k = 0; k < CT_TILES; k++) { s_off[k] = tot; tot += s_cnt[k]; }
        s_base = tot ? atomicAdd(&cnt1[l], tot) : 0;
    }
    __syncthreads();
    for (int k = 0; k < CT_TILES / 4; k++) {
        const int slot = wave * (CT_TILES / 4) + k, til = first + slot;
        if (til >= ntiles || s_cnt[slot] == 0) continue;
        const int tile = d.tile_off + til, n = tile_cnt[tile];
        int run = s_base + s_off[slot];
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const int v = j < n ? surv_sc[(size_t)tile * FT_SLOTS + j] : 0;
            const bool keep = v >= th && v > 0;
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int i = run + __popcll(m & ((1ull << lane) - 1ull));
                if (i < d.cap1) { cand_xy[d.cand_off + i] = surv_xy[(size_t)tile * FT_SLOTS + j]; cand_resp[d.cand_off + i] = (float)v; }
            }
            run += __popcll(m);
        }
    }
}

// debug view (mis_orb_debug_level which = 1): rasterise the survivor list into the NMS score map
__global__ void nms_raster_kernel(LevelDesc d, const int* tile_cnt, const uint32_t* surv_xy, const uint8_t* surv_sc, uint8_t* nms) {
    const int ntiles = d.tiles_x * d.tiles_y;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tile = d.tile_off + tl, n = tile_cnt[tile];
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            uint32_t xy = surv_xy[(size_t)tile * FT_SLOTS + j];
            nms[d.map_off + (size_t)(xy >> 16) * d.sp + (xy & 0xffff)] = surv_sc[(size_t)tile * FT_SLOTS + j];
        }
    }
}

// ---------------------------------------------------------------- K4 Harris ------------------
// HarrisResponses(blockSize 7, k 0.04) at the candidate (integer gradients, f32 response)
__global__ __launch_bounds__(256) void harris_kernel(Levels L, const uint8_t* pad, const int* cnt1, const uint32_t* cand_xy, float* cand_resp, size_t ws) {
    const LevelDesc& d = L.d[blockIdx.y];
    WS_OFF(pad, blockIdx.z, ws); WS_OFF(cnt1, blockIdx.z, ws); WS_OFF(cand_xy, blockIdx.z, ws); WS_OFF(cand_resp, blockIdx.z, ws);
    int i = blockIdx.x * 256 + threadIdx.x;
    int n = min(cnt1[blockIdx.y], d.cap1);
    if (i >= n) return;
    uint32_t xy = cand_xy[d.cand_off + i];
    int x = xy & 0xffff, y = xy >> 16;
    const int pp = d.pp;
    // the 9 x 9 neighbourhood as three aligned dwords per row (the 9 bytes start at byte 0 .. 3 of them) and funnel shifts: 27 loads
    // per thread where byte loads were 81 -- a wave's loads go to 64 different lines each, and the address unit was what the kernel
    // waited for (89 us per 16 frames)
    const uint8_t* q0 = pad + d.pad_off + (size_t)(y + ORB_BORDER - 4) * pp + (x + ORB_BORDER - 4);
    const unsigned sh = (unsigned)((uintptr_t)q0 & 3u);
    const uint8_t* a0 = q0 - sh;                       // (pad, pad_off and the pitch are multiples of 4)
    int px[9][9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const unsigned* row = reinterpret_cast<const unsigned*>(a0 + (size_t)r * pp);
        const unsigned d0 = row[0], d1 = row[1], d2 = row[2];
        const unsigned w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh), w2 = __builtin_amdgcn_alignbyte(0u, d2, sh);
#pragma unroll
        for (int k = 0; k < 4; k++) { px[r][k] = (int)((w0 >> (8 * k)) & 255u); px[r][4 + k] = (int)((w1 >> (8 * k)) & 255u); }
        px[r][8] = (int)(w2 & 255u);
    }
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int by = 0; by < 7; by++)
#pragma unroll
        for (int bx = 0; bx < 7; bx++) {
            // p = px[by + 1][bx + 1]
            const int Ix = (px[by + 1][bx + 2] - px[by + 1][bx]) * 2 + (px[by][bx + 2] - px[by][bx]) + (px[by + 2][bx + 2] - px[by + 2][bx]);
            const int Iy = (px[by + 2][bx + 1] - px[by][bx + 1]) * 2 + (px[by + 2][bx] - px[by][bx]) + (px[by + 2][bx + 2] - px[by][bx + 2]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    float fa = (float)a, fb = (float)b, fc = (float)c;
    float s1 = fa * fb, s2 = fc * fc;
    float s3 = 0.04f * (fa + fb);
    s3 = s3 * (fa + fb);
    cand_resp[d.cand_off + i] = ((s1 - s2) - s3) * scale_sq_sq;
}

// ascending-orderable key of an f32
__device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// retainBest(N_l) on the Harris response (ties at the cut kept) + canonical order
// (response desc, y, x) by counting ranks.  One 1024-thread block per level.
__global__ __launch_bounds__(1024) void select_rank_kernel(Levels L, const int* cnt1, const uint32_t* cand_xy, const float* cand_resp,
                                                            int* cnt2, uint32_t* fin_xy, float* fin_resp, int* flags, int use_harris, size_t ws) {
    WS_OFF(cnt1, blockIdx.y, ws); WS_OFF(cand_xy, blockIdx.y, ws); WS_OFF(cand_resp, blockIdx.y, ws); WS_OFF(cnt2, blockIdx.y, ws);
    WS_OFF(fin_xy, blockIdx.y, ws); WS_OFF(fin_resp, blockIdx.y, ws); WS_OFF(flags, blockIdx.y, ws);
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix, s_mask;
    __shared__ int s_k, s_m;
    __shared__ uint32_t kkey[4096], kxy[4096];
    const int l = blockIdx.x, t = threadIdx.x;
    const LevelDesc& d = L.d[l];
    const int n = min(cnt1[l], d.cap1);
    const uint32_t* xy = cand_xy + d.cand_off;
    const float* rs = cand_resp + d.cand_off;
    const int N = d.nfeat;
    if (N == 0) { if (t == 0) cnt2[l] = 0; return; }
    if (n <= 4096) {
        // the usual case (~2N + ties candidates): bitonic sort of (key desc, y, x) in LDS; the cut is the key
        // at position N-1, the kept set is the sorted prefix up to the last key >= cut -- already in the
        // canonical order, no ranking pass and no atomics
        int P = 1;
        while (P < n) P <<= 1;
        for (int i = t; i < P; i += 1024) {
            if (i < n) { kkey[i] = fkey(rs[i]); kxy[i] = xy[i]; }
            else { kkey[i] = 0; kxy[i] = 0xffffffffu; }  // sentinels sort last
        }
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = t; i < P; i += 1024) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const uint32_t ka = kkey[i], kb = kkey[ixj], pa = kxy[i], pb = kxy[ixj];
                        const bool a_first = ka > kb || (ka == kb && pa < pb);  // a precedes b in the canonical order
                        const bool descending_block = (i & k) == 0;
                        if (a_first != descending_block) { kkey[i] = kb; kkey[ixj] = ka; kxy[i] = pb; kxy[ixj] = pa; }
                    }
                }
                __syncthreads();
            }
        const uint32_t cutk = (use_harris && n > N) ? kkey[N - 1] : 0u;
        if (t == 0) s_m = 0;
        __syncthreads();
        // m = number of keys >= cut: the boundary index in the sorted array
        for (int i = t; i < n; i += 1024)
            if (kkey[i] >= cutk && (i == n - 1 || kkey[i + 1] < cutk)) s_m = i + 1;
        __syncthreads();
        const int m = s_m;
        if (m > d.cap2) { if (t == 0) { atomicOr(&flags[0], 2); cnt2[l] = 0; } return; }
        for (int i = t; i < m; i += 1024) {
            fin_xy[d.fin_off + i] = kxy[i];
            const uint32_t ki = kkey[i], uu = (ki & 0x80000000u) ? (ki & 0x7fffffffu) : ~ki;
            fin_resp[d.fin_off + i] = __uint_as_float(uu);
        }
        if (t == 0) cnt2[l] = m;
        return;
    }
    uint32_t cut = 0;  // keep keys >= cut
    if (use_harris && n > N) {
        // radix select: the N-th largest key
        if (t == 0) { s_prefix = 0; s_mask = 0; s_k = N; }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (t < 256) hist[t] = 0;
            __syncthreads();
            uint32_t prefix = s_prefix, mask = s_mask;
            for (int i = t; i < n; i += 1024) {
                uint32_t k = fkey(rs[i]);
                if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
            }
            __syncthreads();
            if (t == 0) {
                int k = s_k, b = 255;
                for (; b > 0; b--) { if (hist[b] >= k) break; k -= hist[b]; }
                s_k = k; s_prefix = prefix | ((uint32_t)b << shift); s_mask = mask | (0xffu << shift);
            }
            __syncthreads();
        }
        cut = s_prefix;
    }
    // gather the kept set (unordered)
    if (t == 0) s_m = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        uint32_t k = fkey(rs[i]);
        if (k >= cut) {
            int j = atomicAdd(&s_m, 1);
            if (j < 4096) { kkey[j] = k; kxy[j] = xy[i]; }
        }
    }
    __syncthreads();
    int m = s_m;
    if (m > d.cap2 || m > 4096) { if (t == 0) { atomicOr(&flags[0], 2); cnt2[l] = 0; } return; }
    // rank = number of kept elements that come first: larger key, then smaller y, then smaller x
    for (int i = t; i < m; i += 1024) {
        uint32_t ki = kkey[i], pi = kxy[i];
        uint32_t yi = (pi >> 16), xi = pi & 0xffff;
        uint32_t oi = (yi << 16) | xi;
        int r = 0;
        for (int j = 0; j < m; j++) {
            uint32_t kj = kkey[j], pj = kxy[j];
            uint32_t oj = ((pj >> 16) << 16) | (pj & 0xffff);
            r += (kj > ki) || (kj == ki && oj < oi);
        }
        fin_xy[d.fin_off + r] = pi;
        uint32_t u = (ki & 0x80000000u) ? (ki & 0x7fffffffu) : ~ki;
        fin_resp[d.fin_off + r] = __uint_as_float(u);
    }
    if (t == 0) cnt2[l] = m;
}

// ---------------------------------------------------------------- K5 assembly + IC angle -----
// One wave per keypoint: concatenates the per-level segments, evaluates the intensity centroid on
// the un-blurred level (ICAngles), writes the cv::KeyPoint (pt scaled to level-0 coordinates).
__global__ __launch_bounds__(256) void assemble_angle_kernel(Levels L, const uint8_t* pad, const int* cnt2, const uint32_t* fin_xy,
                                                             const float* fin_resp, const int* umax, OrbIO io, int cap_out, size_t ws, int with_angle) {
    // One wave per OUTPUT keypoint j (grid.x = ceil(cap_out / 4), grid.y = frame): its level is the one whose run of the output
    // holds j.  A grid of (capacity per level / 4) x levels x frames was 65 k workgroups of which three quarters found nothing to
    // do -- 0.18 ms of workgroup dispatch for 12 us of arithmetic.
    const int fr = blockIdx.y;
    WS_OFF(pad, fr, ws); WS_OFF(cnt2, fr, ws); WS_OFF(fin_xy, fr, ws); WS_OFF(fin_resp, fr, ws);
    MisKeyPoint* kps = io.kps[fr];
    uint32_t* kp_lxy = io.lxy[fr];
    int* n_out = io.n_dev[fr];
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int base = 0, l = 0, total = 0;
    {
        int c[ORB_MAX_LEVELS];
#pragma unroll
        for (int k = 0; k < ORB_MAX_LEVELS; k++) c[k] = k < L.n ? cnt2[min(k, L.n - 1)] : 0;
#pragma unroll
        for (int k = 0; k < ORB_MAX_LEVELS; k++) {
            if (j >= total + c[k]) { base = total + c[k]; l = k + 1; }      // (runs are consecutive: the last assignment is the level before j's)
            total += c[k];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = min(total, cap_out);
    if (j >= total || j >= cap_out) return;
    const LevelDesc& d = L.d[l];
    const int i = j - base;
    uint32_t xy = fin_xy[d.fin_off + i];
    int x = xy & 0xffff, y = xy >> 16;
    const int pp = d.pp, hp = L.half_patch;
    const uint8_t* ctr = pad + d.pad_off + (size_t)(y + ORB_BORDER) * pp + (x + ORB_BORDER);
    int m01 = 0, m10 = 0;
    const int u = lane - hp;  // lanes 0..2hp cover u = -hp..hp
    if (with_angle && lane <= 2 * hp) {      // (describe_direct_kernel computes the angle from the patch it stages anyway)
        m10 = u * ctr[u];
        if (hp <= 15) {
            // every load of the patch goes out before the first use: the row limits umax[1 .. hp] and the 2 hp pixels of the lane's
            // column (a pixel outside the circle is read -- it lies inside the level's 32-pixel border -- and not added).  The loop
            // below loaded umax[v], compared, then loaded two pixels, per row: hp dependent memory latencies per keypoint (10 us a wave).
            int um[16], vp[16], vm[16];
#pragma unroll
            for (int v = 1; v <= 15; ++v) um[v] = umax[min(v, hp)];
#pragma unroll
            for (int v = 1; v <= 15; ++v) { const int vv = min(v, hp); vp[v] = ctr[u + vv * pp]; vm[v] = ctr[u - vv * pp]; }
#pragma unroll
            for (int v = 1; v <= 15; ++v) {
                const bool in = v <= hp && abs(u) <= um[v];
                m10 += in ? u * (vp[v] + vm[v]) : 0;
                m01 += in ? v * (vp[v] - vm[v]) : 0;
            }
        } else {
            for (int v = 1; v <= hp; ++v) {
                if (abs(u) <= umax[v]) {
                    int vp = ctr[u + v * pp], vm = ctr[u - v * pp];
                    m10 += u * (vp + vm);
                    m01 += v * (vp - vm);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m01 += __shfl_xor(m01, o); m10 += __shfl_xor(m10, o); }
    if (lane == 0) {
        MisKeyPoint kp;
        kp.x = (float)x * d.scale; kp.y = (float)y * d.scale;
        kp.size = (float)L.patch * d.scale;
        kp.angle = with_angle ? mis_fast_atan2((float)m01, (float)m10) : 0.f;
        kp.response = fin_resp[d.fin_off + i];
        kp.octave = l;
        kps[base + i] = kp;
        kp_lxy[base + i] = xy;
    }
}

// ---------------------------------------------------------------- K6 blur + rBRIEF ------------
// GaussianBlur 7x7 sigma 2 (Q8 kernel 18 34 48 56 48 34 18, one rounding at bit 16) of the level
// interior; the border ring keeps the un-blurred reflected pixels (the reference blurs the ROI of
// the bordered pyramid in place).  64x16 output tile, separable through LDS.
__global__ __launch_bounds__(256) void blur_kernel(Levels L, const uint8_t* pad, uint8_t* blur) {
    constexpr int TR = 32, TP = 80;                     // tile rows; LDS pitch (dword aligned, >= 64 + 6 + 3)
    __shared__ __attribute__((aligned(16))) uint8_t tile[(TR + 6) * TP];  // rows ty0-3 .. ty0+34, cols tx0-4 .. tx0+75
    __shared__ __attribute__((aligned(16))) uint16_t hbuf[(TR + 6) * 64];
    const LevelDesc& d = L.d[blockIdx.z];
    const int pw = d.w + 2 * ORB_BORDER, ph = d.h + 2 * ORB_BORDER;
    // tiles cover the padded extent; tile origin in padded coordinates (multiple of 64 -> dword aligned)
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * TR, t = threadIdx.x;
    if (tx0 >= pw || ty0 >= ph) return;
    const uint8_t* src = pad + d.pad_off;
    uint8_t* dst = blur + d.pad_off;
    for (int i = t; i < (TR + 6) * (TP / 4); i += 256) {
        const int r = i / (TP / 4), c = i - r * (TP / 4);
        const int sy = min(max(ty0 + r - 3, 0), ph - 1);
        int sx = tx0 - 4 + 4 * c;
        sx = min(max(sx, 0), d.pp - 4);  // clamped dwords only feed pixels of the un-blurred border ring
        reinterpret_cast<unsigned*>(tile)[i] = *reinterpret_cast<const unsigned*>(src + (size_t)sy * d.pp + sx);
    }
    __syncthreads();
    const int kq[7] = {18, 34, 48, 56, 48, 34, 18};
    // horizontal pass: 4 adjacent outputs per work item from three aligned dwords (12 bytes)
    for (int i = t; i < (TR + 6) * 16; i += 256) {
        const int r = i >> 4, cg = i & 15;
        const unsigned* p = reinterpret_cast<const unsigned*>(tile + r * TP + 4 * cg);
        const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
        int b[12];
#pragma unroll
        for (int k = 0; k < 4; k++) { b[k] = (w0 >> (8 * k)) & 255; b[4 + k] = (w1 >> (8 * k)) & 255; b[8 + k] = (w2 >> (8 * k)) & 255; }
        unsigned short o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // output column 4 cg + k reads tile columns 4 cg + k + 1 .. + 7
            int acc = 0;
#pragma unroll
            for (int j = 0; j < 7; j++) acc += kq[j] * b[k + 1 + j];
            o[k] = (unsigned short)acc;
        }
        *reinterpret_cast<uint2*>(hbuf + r * 64 + 4 * cg) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    }
    __syncthreads();
    // vertical pass: a thread slides down 8 rows of one column (14 reads for 8 outputs)
    const int c = t & 63, px = tx0 + c;
    if (px >= pw) return;
    const int r0 = (t >> 6) * (TR / 4);
    int hv[TR / 4 + 6];
#pragma unroll
    for (int j = 0; j < TR / 4 + 6; j++) hv[j] = hbuf[(r0 + j) * 64 + c];
#pragma unroll
    for (int k = 0; k < TR / 4; k++) {
        const int r = r0 + k, py = ty0 + r;
        if (py >= ph) break;
        const int x = px - ORB_BORDER, y = py - ORB_BORDER;
        uint8_t o;
        if ((unsigned)x < (unsigned)d.w && (unsigned)y < (unsigned)d.h) {
            int acc = 0;
#pragma unroll
            for (int j = 0; j < 7; j++) acc += kq[j] * hv[k + j];
            o = (uint8_t)((acc + (1 << 15)) >> 16);
        } else o = tile[(r + 3) * TP + c + 4];
        dst[(size_t)py * d.pp + px] = o;
    }
}

// computeOrbDescriptors, WTA_K = 2: lane b of a 32-lane half-wave builds byte b (8 tests)
__global__ __launch_bounds__(256) void describe_kernel(Levels L, const uint8_t* blur, const MisKeyPoint* kps, const uint32_t* kp_lxy,
                                                       const int* n_ptr, const int8_t* pattern, uint8_t* desc) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), b = threadIdx.x & 31;
    if (i >= *n_ptr) return;
    const MisKeyPoint kp = kps[i];
    const LevelDesc& d = L.d[kp.octave];
    // the keypoint is already in level-0 coordinates; descriptors scale it back (1/scale) and round
    const float inv = 1.f / d.scale;
    const int cx = mis_round_f(kp.x * inv), cy = mis_round_f(kp.y * inv);
    float ang = kp.angle * (float)(3.14159265358979323846 / 180.f);
    float sa, ca;
    mis_sincosf(ang, &sa, &ca);
    const int pp = d.pp;
    const uint8_t* ctr = blur + d.pad_off + (size_t)(cy + ORB_BORDER) * pp + (cx + ORB_BORDER);
    const int8_t* pat = pattern + b * 32;  // 16 points x (x, y)
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        float x0 = (float)pat[4 * bit], y0 = (float)pat[4 * bit + 1], x1 = (float)pat[4 * bit + 2], y1 = (float)pat[4 * bit + 3];
        int ix0 = mis_round_f(x0 * ca - y0 * sa), iy0 = mis_round_f(x0 * sa + y0 * ca);
        int ix1 = mis_round_f(x1 * ca - y1 * sa), iy1 = mis_round_f(x1 * sa + y1 * ca);
        int t0 = ctr[iy0 * pp + ix0], t1 = ctr[iy1 * pp + ix1];
        val |= (t0 < t1) << bit;
    }
    desc[(size_t)i * 32 + b] = (uint8_t)val;
}

// The same descriptors without blurring the pyramid: only the 512 sample points of a keypoint are ever read from the blurred
// image, so one wave stages the keypoint's (2 (R + 3) + 1)^2 source patch in LDS and evaluates the 7 x 7 Gaussian at its samples
// directly -- sum_j k[j] (sum_i k[i] p[y+j][x+i]), the separable filter's own integers (row sums <= 65280 fit the 16 bits
// the two-pass form keeps) -- 25 k multiply-adds per keypoint instead of 14 per pixel of every level (84 us of blur_kernel per
// 4K frame for 4000 keypoints' worth of samples).  R = the largest rounded pattern coordinate (28 for patchSize 40); samples
// inside the level are blurred (their taps may reach into the reflected border ring), samples in the ring are not: the reference
// blurs the level's ROI of the bordered pyramid only.
constexpr int DD_R = 28, DD_P = DD_R + 3, DD_N = 2 * DD_P + 1, DD_ROW_DW = (DD_N + 3 + 3) / 4, DD_PITCH = 4 * DD_ROW_DW + 4;
struct UMax16 { int v[16]; };      // the disc's row limits umax[0 .. 15] by value: scalar loads from the kernel arguments
__global__ __launch_bounds__(256) void describe_direct_kernel(Levels L, const uint8_t* pad, OrbIO io, const int8_t* pattern, const int* umax, UMax16 um16, size_t ws) {
    __shared__ __attribute__((aligned(16))) uint8_t patch[4][DD_N * DD_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave, fr = blockIdx.y;
    WS_OFF(pad, fr, ws);
    MisKeyPoint* kps = io.kps[fr];
    const uint32_t* kp_lxy = io.lxy[fr];
    const int* n_ptr = io.n_dev[fr];
    uint8_t* desc = io.desc[fr];
    if (i >= *n_ptr) return;   // wave-uniform: no workgroup barrier below
    const MisKeyPoint kp = kps[i];
    const LevelDesc& d = L.d[kp.octave];
    const float inv = 1.f / d.scale;
    const int cx = mis_round_f(kp.x * inv), cy = mis_round_f(kp.y * inv);
    const int pp = d.pp;
    const uint8_t* org = pad + d.pad_off + (size_t)(cy + ORB_BORDER - DD_P) * pp + (cx + ORB_BORDER - DD_P);
    const int shift = (int)((uintptr_t)org & 3);
    const uint8_t* abase = org - shift;   // rows are copied as aligned dwords; the patch starts `shift` bytes into an LDS row
    uint8_t* P = patch[wave];
    for (int k = lane; k < DD_N * DD_ROW_DW; k += 64) {
        const int r = k / DD_ROW_DW, c = k - r * DD_ROW_DW;
        reinterpret_cast<unsigned*>(P + r * DD_PITCH)[c] = *reinterpret_cast<const unsigned*>(abase + (size_t)r * pp + 4 * c);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the lanes of one wave execute their LDS instructions in order
    __builtin_amdgcn_wave_barrier();
    const int c0 = DD_P * DD_PITCH + DD_P + shift;               // byte offset of the keypoint's pixel
    // IC_Angle on the staged patch (the intensity centroid of the 31-pixel disc): the keypoint kernel read the same pixels from
    // memory a second time -- 31 rows = 31 cache lines per keypoint, 0.18 ms for 64 k keypoints
    float kp_angle;
    {
        const int hp = L.half_patch;
        const uint8_t* ctr = P + c0;
        int m01 = 0, m10 = 0;
        const int u = lane - hp;  // lanes 0..2hp cover u = -hp..hp
        if (lane <= 2 * hp) {
            m10 = u * ctr[u];
            if (hp <= 15) {
#pragma unroll 3
                for (int v = 1; v <= hp; ++v) {      // (three rows in flight: unrolled 15 times the 30 pixels held the kernel at 169 registers, 3 waves per SIMD)
                    const int vp = ctr[u + v * DD_PITCH], vm = ctr[u - v * DD_PITCH];
                    const bool in = abs(u) <= um16.v[v];
                    m10 += in ? u * (vp + vm) : 0;
                    m01 += in ? v * (vp - vm) : 0;
                }
            } else {
                for (int v = 1; v <= hp; ++v) {
                    if (abs(u) <= umax[v]) {
                        const int vp = ctr[u + v * DD_PITCH], vm = ctr[u - v * DD_PITCH];
                        m10 += u * (vp + vm);
                        m01 += v * (vp - vm);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m01 += __shfl_xor(m01, o); m10 += __shfl_xor(m10, o); }
        kp_angle = mis_fast_atan2((float)m01, (float)m10);
        if (lane == 0) kps[i].angle = kp_angle;
    }
    float ang = kp_angle * (float)(3.14159265358979323846 / 180.f);
    float sa, ca;
    mis_sincosf(ang, &sa, &ca);
    const int b = lane & 31, h = lane >> 5;                  // lane b (and b + 32) of the wave builds byte b: four tests each
    const int8_t* pat = pattern + b * 32 + h * 16;
    // a row's seven taps are bytes a .. a + 6 of three aligned LDS dwords: two funnel shifts bring them into two dwords, two
    // v_dot4_u32_u8 against the packed kernel (18 34 48 56 | 48 34 18 0) give the row sum
    const unsigned* P32 = reinterpret_cast<const unsigned*>(P);
    auto blurred = [&](int ix, int iy) {
        const int a = c0 + (iy - 3) * DD_PITCH + (ix - 3);       // first tap of the first row
        // keypoints may sit 3 pixels from the edge (edgeThreshold 1): a sample outside the level reads the un-blurred border ring
        const bool inside = (unsigned)(cx + ix) < (unsigned)d.w && (unsigned)(cy + iy) < (unsigned)d.h;
        const unsigned* q = P32 + (a >> 2);
        const unsigned sh = (unsigned)(a & 3) << 3;
        const int kq[7] = {18, 34, 48, 56, 48, 34, 18};
        unsigned acc = 0, centre = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const unsigned d0 = q[j * (DD_PITCH / 4)], d1 = q[j * (DD_PITCH / 4) + 1], d2 = q[j * (DD_PITCH / 4) + 2];
            const unsigned lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
            const unsigned hs = __builtin_amdgcn_udot4(lo, 0x38302212u, __builtin_amdgcn_udot4(hi, 0x00122230u, 0u, false), false);
            acc += (unsigned)kq[j] * hs;
            if (j == 3) centre = lo >> 24;                       // tap 3 of row 3: the sample's own pixel
        }
        return inside ? (int)((acc + (1u << 15)) >> 16) : (int)centre;
    };
    int val = 0;
#ifndef DD_UNROLL
#define DD_UNROLL 1
#endif
#pragma unroll DD_UNROLL
    for (int bit = 0; bit < 4; bit++) {
        float x0 = (float)pat[4 * bit], y0 = (float)pat[4 * bit + 1], x1 = (float)pat[4 * bit + 2], y1 = (float)pat[4 * bit + 3];
        int ix0 = mis_round_f(x0 * ca - y0 * sa), iy0 = mis_round_f(x0 * sa + y0 * ca);
        int ix1 = mis_round_f(x1 * ca - y1 * sa), iy1 = mis_round_f(x1 * sa + y1 * ca);
        const int t0 = blurred(ix0, iy0), t1 = blurred(ix1, iy1);
        val |= (t0 < t1) << (bit + 4 * h);
    }
    val |= __shfl_xor(val, 32);
    if (h == 0) desc[(size_t)i * 32 + b] = (uint8_t)val;
}

// the keypoint counts of a batch and the lanes' overflow flags, gathered into one host-visible (pinned, mapped) array:
// one kernel instead of one small device-to-host copy per frame at the end of the feature stage
constexpr int ORB_GATHER_MAX = 96;
struct GatherPtrs { const int* p[ORB_GATHER_MAX]; };
__global__ void gather_ints_kernel(GatherPtrs g, int n, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = *g.p[i];
}

}  // namespace

struct MisOrb {
    MisContext* ctx = nullptr;
    void (*enqueued_cb)(void*) = nullptr;      // mis_orb_on_enqueued: one-shot hook of the next batch call
    void* enqueued_user = nullptr;
    MisOrbParams p;
    int max_w = 0, max_h = 0;
    int cur_w = 0, cur_h = 0;
    Levels L;
    Work w;             // pointers of the first frame's workspace; frame f of a batch group: + f * ws_stride bytes
    size_t ws_stride = 0;
    int ws_frames = 0;  // workspaces allocated (1 until the first batch of more than one frame)
    int umax_host[64] = {0};
    int8_t pattern_host[1024] = {0};
    void* mem = nullptr;
    size_t pad_bytes = 0, map_bytes = 0;
    int cand_total = 0, fin_total = 0, tab_total = 0, surv_total = 0, out_cap = 0;
    std::vector<int> tab_host;
    int* host_counts = nullptr;   // pinned, device-visible: gather_ints_kernel writes the batch's counts / flags here
    bool direct_describe = false;   // the pattern's reach fits describe_direct_kernel's patch: no blurred pyramid
};

namespace {

void linear_exact_coeffs(int dlen, int slen, int* ofs, int* m1) {
    double inv = (double)dlen / (double)slen, scale = 1.0 / inv;
    for (int i = 0; i < dlen; i++) {
        double v = ((double)i + 0.5) * scale - 0.5;
        int iv = (int)floor(v);
        if (iv < 0) { ofs[i] = 0; m1[i] = 0; }
        else if (iv >= slen - 1) { ofs[i] = slen - 1; m1[i] = 0; }
        else { ofs[i] = iv; m1[i] = (int)lrint((v - (double)iv) * 256.0); }
    }
}

// level geometry + budgets for an image size (orb.cpp: layer sizes, nfeaturesPerLevel)
void plan_levels(MisOrb* o, int w, int h) {
    Levels& L = o->L;
    const MisOrbParams& p = o->p;
    L.n = p.nlevels; L.fast_t = p.fast_threshold; L.patch = p.patch_size; L.half_patch = p.patch_size / 2; L.edge = p.edge_threshold;
    double sf = (double)p.scale_factor;
    size_t pad_off = 0, map_off = 0;
    int cand_off = 0, fin_off = 0, tab_off = 0, surv_off = 0;
    float factor = (float)(1.0 / sf);
    float nd = (float)p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < L.n; l++) {
        LevelDesc& d = L.d[l];
        float sc = (float)pow(sf, (double)l);
        d.scale = sc;
        d.w = (int)lrint((double)((float)w / sc));
        d.h = (int)lrint((double)((float)h / sc));
        d.pp = (int)mis_align_up((size_t)d.w + 2 * ORB_BORDER, 64);
        d.sp = (int)mis_align_up((size_t)d.w, 64);
        d.pad_off = pad_off; pad_off += (size_t)d.pp * (d.h + 2 * ORB_BORDER);
        d.map_off = map_off; map_off += (size_t)d.sp * d.h;
        if (l < L.n - 1) { d.nfeat = (int)lrintf(nd); sum += d.nfeat; nd *= factor; }
        else d.nfeat = std::max(p.nfeatures - sum, 0);
        d.n2 = p.score_type == 0 ? 2 * d.nfeat : d.nfeat;
        d.tiles_x = (d.w + FT_COLS - 1) / FT_COLS; d.tiles_y = (d.h + FT_ROWS - 1) / FT_ROWS;
        d.tile_off = surv_off; surv_off += d.tiles_x * d.tiles_y;
        d.cap1 = 4 * d.n2 + 4096;
        d.cap2 = std::min(d.nfeat + 128, 2048);
        d.cand_off = cand_off; cand_off += d.cap1;
        d.fin_off = fin_off; fin_off += d.cap2;
        d.tab_off = tab_off; tab_off += (2 * ((d.w + 3) & ~3) + 2 * d.h + 3) & ~3;   // xofs[dw4] xm1[dw4] yofs[dh] ym1[dh], int4 aligned
    }
    o->pad_bytes = pad_off; o->map_bytes = map_off; o->cand_total = cand_off; o->fin_total = fin_off; o->tab_total = tab_off; o->surv_total = surv_off;
    o->tab_host.assign(tab_off, 0);
    for (int l = 1; l < L.n; l++) {
        int* t = o->tab_host.data() + L.d[l].tab_off;
        const int dw4 = (L.d[l].w + 3) & ~3;
        linear_exact_coeffs(L.d[l].w, L.d[l - 1].w, t, t + dw4);
        linear_exact_coeffs(L.d[l].h, L.d[l - 1].h, t + 2 * dw4, t + 2 * dw4 + L.d[l].h);
    }
}

int upload_tables(MisOrb* o) {
    MisContext* ctx = o->ctx;
    if (o->tab_total)
        MIS_HIP(ctx, hipMemcpyAsync(o->w.tab, o->tab_host.data(), sizeof(int) * o->tab_total, hipMemcpyHostToDevice, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // tab_host may be rebuilt by the next plan
    return MIS_OK;
}

std::mutex g_feat_mutex;
std::unordered_map<void*, size_t> g_feat_sizes;  // block sizes of live feature sets (for the pool)

// one device block per feature set: [size header 256 B][keypoints][descriptors][level xy][count]
int alloc_features(MisContext* ctx, int cap, int desc_cols, int desc_dtype, MisFeatures* f) {
    size_t kb = mis_align_up(sizeof(MisKeyPoint) * (size_t)cap, 256), db = mis_align_up((size_t)cap * desc_cols * mis_dtype_size(desc_dtype), 256);
    size_t lb = mis_align_up(sizeof(uint32_t) * (size_t)cap, 256);
    void* mem = nullptr;
    size_t got = 0;
    int rc = mis_pool_alloc(ctx, kb + db + lb + 256, &mem, &got);
    if (rc != MIS_OK) return rc;
    f->keypoints = (MisKeyPoint*)mem;
    f->descriptors = (uint8_t*)mem + kb;
    f->desc_cols = desc_cols; f->desc_dtype = desc_dtype;
    f->owner_ = mem;
    f->n = 0;
    std::lock_guard<std::mutex> lock(g_feat_mutex);
    g_feat_sizes[mem] = got;
    return MIS_OK;
}

inline uint32_t* feat_lxy(const MisFeatures* f, int cap) {
    size_t kb = mis_align_up(sizeof(MisKeyPoint) * (size_t)cap, 256), db = mis_align_up((size_t)cap * f->desc_cols * mis_dtype_size(f->desc_dtype), 256);
    return (uint32_t*)((uint8_t*)f->owner_ + kb + db);
}
inline int* feat_count(const MisFeatures* f, int cap) {
    size_t kb = mis_align_up(sizeof(MisKeyPoint) * (size_t)cap, 256), db = mis_align_up((size_t)cap * f->desc_cols * mis_dtype_size(f->desc_dtype), 256);
    size_t lb = mis_align_up(sizeof(uint32_t) * (size_t)cap, 256);
    return (int*)((uint8_t*)f->owner_ + kb + db + lb);
}

// enqueue the whole detect + describe path of a group of ng <= ORB_BATCH frames (one launch per stage); no host synchronisation
int enqueue_detect_group(MisOrb* o, const DevImage* img, int w, int h, MisFeatures* out, int ng) {
    MisContext* ctx = o->ctx;
    const Levels& L = o->L;
    const Work& W = o->w;
    hipStream_t st = ctx->stream;
    const LevelDesc& d0 = L.d[0];
    const size_t ws = o->ws_stride;
    const unsigned nf = (unsigned)ng;
    OrbIO io;
    for (int k = 0; k < ORB_BATCH; k++) {
        const int q = k < ng ? k : 0;
        io.src[k] = (const uint8_t*)img[q].data; io.sstride[k] = img[q].stride;
        io.aligned[k] = (img[q].stride % 4 == 0) && ((uintptr_t)img[q].data % 4 == 0);
        io.kps[k] = out[q].keypoints; io.lxy[k] = feat_lxy(&out[q], o->out_cap); io.n_dev[k] = feat_count(&out[q], o->out_cap);
        io.desc[k] = (uint8_t*)out[q].descriptors;
    }
    // hist, thr, cnt0, cnt1, cnt2 of every frame of the group (same offset in each workspace)
    MIS_HIP(ctx, hipMemset2DAsync(W.hist, ws ? ws : sizeof(int) * (256 * HIST_COPIES * ORB_MAX_LEVELS + 4 * ORB_MAX_LEVELS), 0,
                                  sizeof(int) * (256 * HIST_COPIES * ORB_MAX_LEVELS + 4 * ORB_MAX_LEVELS), nf, st));
    hipLaunchKernelGGL(gray_kernel, dim3((w + 1023) / 1024, (h + GRAY_ROWS - 1) / GRAY_ROWS, nf), dim3(256), 0, st, io, w, h, W.pad + d0.pad_off, d0.pp, ws);
    for (int l = 1; l < L.n; l++) {
        const LevelDesc &s = L.d[l - 1], &d = L.d[l];
        hipLaunchKernelGGL(resize_kernel, dim3((d.w + 255) / 256, (d.h + 4 * RS_ROWS - 1) / (4 * RS_ROWS), nf), dim3(256), 0, st, W.pad + s.pad_off, s.w, s.h, s.pp, W.pad + d.pad_off, d.w,
                           d.h, d.pp, W.tab + d.tab_off, ws);
    }
    const int pw0 = d0.w + 2 * ORB_BORDER, ph0 = d0.h + 2 * ORB_BORDER;
    hipLaunchKernelGGL((border_kernel<true>), dim3((d0.h * 16 + 255) / 256, 1, L.n * nf), dim3(256), 0, st, L, W.pad, ws);   // sides first: the corners mirror them
    hipLaunchKernelGGL((border_kernel<false>), dim3((pw0 + 1023) / 1024, 2 * ORB_BORDER / 4, L.n * nf), dim3(256), 0, st, L, W.pad, ws);
    dim3 gmap((d0.w + FT_COLS - 1) / FT_COLS, (d0.h + FT_ROWS - 1) / FT_ROWS, L.n * nf);
    hipLaunchKernelGGL(fast_nms_kernel, gmap, dim3(256), 0, st, L, W.pad, W.hist, W.tile_cnt, W.surv_xy, W.surv_sc, ws);
    hipLaunchKernelGGL(fast_cut_kernel, dim3(L.n, nf), dim3(256), 0, st, L, W.hist, W.thr, W.flags, ws);
    hipLaunchKernelGGL(compact_kernel, dim3((d0.tiles_x * d0.tiles_y + CT_TILES - 1) / CT_TILES, L.n, nf), dim3(256), 0, st, L, W.tile_cnt, W.surv_xy, W.surv_sc, W.thr, W.cnt1,
                       W.cand_xy, W.cand_resp, ws);
    const int use_harris = o->p.score_type == 0;
    if (use_harris)
        hipLaunchKernelGGL(harris_kernel, dim3((L.d[0].cap1 + 255) / 256, L.n, nf), dim3(256), 0, st, L, W.pad, W.cnt1, W.cand_xy, W.cand_resp, ws);
    hipLaunchKernelGGL(select_rank_kernel, dim3(L.n, nf), dim3(1024), 0, st, L, W.cnt1, W.cand_xy, W.cand_resp, W.cnt2, W.fin_xy, W.fin_resp, W.flags,
                       use_harris, ws);
    hipLaunchKernelGGL(assemble_angle_kernel, dim3((o->out_cap + 3) / 4, nf), dim3(256), 0, st, L, W.pad, W.cnt2, W.fin_xy, W.fin_resp, W.umax, io, o->out_cap, ws, o->direct_describe ? 0 : 1);
    if (o->direct_describe) {
        UMax16 um16;
        for (int v = 0; v < 16; v++) um16.v[v] = o->umax_host[v];
        hipLaunchKernelGGL(describe_direct_kernel, dim3((o->out_cap + 3) / 4, nf), dim3(256), 0, st, L, W.pad, io, W.pattern, W.umax, um16, ws);
    } else {
        // patterns that reach beyond the direct kernel's patch: the blurred pyramid, frame by frame (not the default parameters)
        for (int k = 0; k < ng; k++) {
            hipLaunchKernelGGL(blur_kernel, dim3((pw0 + 63) / 64, (ph0 + 31) / 32, L.n), dim3(256), 0, st, L, W.pad + k * ws, W.blur + k * ws);
            hipLaunchKernelGGL(describe_kernel, dim3((o->out_cap + 7) / 8), dim3(256), 0, st, L, W.blur + k * ws, out[k].keypoints, io.lxy[k], io.n_dev[k], W.pattern,
                               (uint8_t*)out[k].descriptors);
        }
    }
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

int check_image(MisOrb* o, const MisImage* bgr) {
    MisContext* ctx = o->ctx;
    MIS_CHECK(ctx, bgr && bgr->data, MIS_E_INVALID, "null image");
    MIS_CHECK(ctx, bgr->dtype == MIS_U8 && bgr->channels == 3, MIS_E_UNSUPPORTED, "ORB input must be 8UC3 (BGR)");
    MIS_CHECK(ctx, bgr->width <= o->max_w && bgr->height <= o->max_h && bgr->width >= 64 && bgr->height >= 64, MIS_E_INVALID,
              "image %dx%d outside the finder's range (64x64 .. %dx%d)", bgr->width, bgr->height, o->max_w, o->max_h);
    return MIS_OK;
}

int replan_if_needed(MisOrb* o, int w, int h) {
    if (w == o->cur_w && h == o->cur_h) return MIS_OK;
    MIS_HIP(o->ctx, hipStreamSynchronize(o->ctx->stream));
    plan_levels(o, w, h);
    o->cur_w = w; o->cur_h = h;
    return upload_tables(o);
}

// `frames` identical workspaces in one allocation (frame f at + f * ws_stride); W holds the first frame's pointers
int orb_alloc_workspace(MisOrb* o, int frames) {
    MisContext* ctx = o->ctx;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o_ = off; off += mis_align_up(bytes, 256); return o_; };
    // per frame: the padded gray pyramid and the small arrays; the blurred pyramid only when the descriptors need it (patterns that
    // reach beyond describe_direct_kernel's patch).  The NMS map and, otherwise, the blurred pyramid exist ONCE behind the frames'
    // blocks: only the debug views of the last single detect read them (16 complete workspaces were 2.2 GB at 4K)
    const bool frame_blur = !o->direct_describe;
    size_t o_pad = carve(o->pad_bytes), o_blur = frame_blur ? carve(o->pad_bytes) : 0;
    size_t o_hist = carve(sizeof(int) * (256 * HIST_COPIES * ORB_MAX_LEVELS + 4 * ORB_MAX_LEVELS)), o_flags = carve(256);
    size_t o_sxy = carve(sizeof(uint32_t) * FT_SLOTS * (size_t)o->surv_total), o_ssc = carve(FT_SLOTS * (size_t)o->surv_total), o_tc = carve(sizeof(int) * (size_t)o->surv_total);
    size_t o_cxy = carve(sizeof(uint32_t) * o->cand_total), o_cr = carve(sizeof(float) * o->cand_total);
    size_t o_fxy = carve(sizeof(uint32_t) * o->fin_total), o_fr = carve(sizeof(float) * o->fin_total);
    size_t o_tab = carve(sizeof(int) * (o->tab_total + 4)), o_umax = carve(sizeof(int) * 64), o_pat = carve(1024);
    const size_t tail = mis_align_up(o->map_bytes, 256) + (frame_blur ? 0 : mis_align_up(o->pad_bytes, 256));
    void* mem = nullptr;
    if (hipMalloc(&mem, off * (size_t)frames + tail) != hipSuccess) return mis_set_error(ctx, MIS_E_NOMEM, "hipMalloc of %zu bytes failed", off * (size_t)frames + tail);
    o->mem = mem; o->ws_stride = off; o->ws_frames = frames;
    uint8_t* m = (uint8_t*)o->mem;
    uint8_t* shared = m + off * (size_t)frames;
    Work& W = o->w;
    W.pad = m + o_pad; W.nms = shared; W.blur = frame_blur ? m + o_blur : shared + mis_align_up(o->map_bytes, 256); W.score = nullptr;
    W.hist = (int*)(m + o_hist); W.thr = W.hist + 256 * HIST_COPIES * ORB_MAX_LEVELS; W.cnt1 = W.thr + ORB_MAX_LEVELS; W.cnt2 = W.cnt1 + ORB_MAX_LEVELS;
    W.surv_xy = (uint32_t*)(m + o_sxy); W.surv_sc = m + o_ssc; W.tile_cnt = (int*)(m + o_tc);
    W.flags = (int*)(m + o_flags);
    W.cand_xy = (uint32_t*)(m + o_cxy); W.cand_resp = (float*)(m + o_cr); W.fin_xy = (uint32_t*)(m + o_fxy); W.fin_resp = (float*)(m + o_fr);
    W.tab = (int*)(m + o_tab); W.umax = (int*)(m + o_umax); W.pattern = (int8_t*)(m + o_pat);
    return MIS_OK;
}

// constants of a fresh workspace block: umax, the BRIEF pattern and the resize tables (shared: the first frame's), the overflow
// flags of every frame
int orb_init_workspace(MisOrb* o) {
    MisContext* ctx = o->ctx;
    const Work& W = o->w;
    MIS_HIP(ctx, hipMemcpyAsync(W.umax, o->umax_host, sizeof(o->umax_host), hipMemcpyHostToDevice, ctx->stream));
    MIS_HIP(ctx, hipMemcpyAsync(W.pattern, o->pattern_host, sizeof(o->pattern_host), hipMemcpyHostToDevice, ctx->stream));
    MIS_HIP(ctx, hipMemset2DAsync(W.flags, o->ws_stride, 0, 256, (size_t)o->ws_frames, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return upload_tables(o);
}

// at least `frames` (<= ORB_BATCH) workspaces: the block is re-allocated once, on the first batch that needs more
int orb_ensure_frames(MisOrb* o, int frames) {
    frames = std::min(frames, ORB_BATCH);
    if (o->ws_frames >= frames) return MIS_OK;
    MisContext* ctx = o->ctx;
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    void* old = o->mem;
    const Work oldw = o->w;
    const size_t olds = o->ws_stride;
    const int oldf = o->ws_frames;
    // a batch wants `frames` workspaces; when the device cannot give that many, fewer do (the batch loops over groups of ws_frames)
    int rc = MIS_E_NOMEM;
    for (int f = frames; f > oldf; f = std::max(oldf, f / 2)) {
        if ((rc = orb_alloc_workspace(o, f)) == MIS_OK) break;
    }
    if (rc != MIS_OK) {       // keep what there was: the batch runs in groups of the old size
        o->mem = old; o->w = oldw; o->ws_stride = olds; o->ws_frames = oldf;
        return oldf >= 1 ? MIS_OK : rc;
    }
    if (old) MIS_HIP(ctx, hipFree(old));
    return orb_init_workspace(o);
}

}  // namespace

// (common.h) a device block for a feature set from ctx's pool, registered so that mis_features_free recycles it: the SIFT finders' outputs
int mis_feat_block_alloc(MisContext* ctx, size_t bytes, void** out) {
    size_t got = 0;
    const int rc = mis_pool_alloc(ctx, bytes, out, &got);
    if (rc != MIS_OK) return rc;
    std::lock_guard<std::mutex> lock(g_feat_mutex);
    g_feat_sizes[*out] = got;
    return MIS_OK;
}


extern "C" void mis_orb_default_params(MisOrbParams* p) {
    if (p) *p = MisOrbParams{4000, 1.2f, 8, 1, 0, 2, 0, 40, 20};
}

extern "C" int mis_orb_create(MisContext* ctx, const MisOrbParams* p, int max_w, int max_h, MisOrb** out) {
    if (!ctx || !out) return MIS_E_INVALID;
    MIS_CHECK(ctx, p, MIS_E_INVALID, "null params");
    MIS_CHECK(ctx, p->nlevels >= 1 && p->nlevels <= ORB_MAX_LEVELS && p->first_level == 0 && p->wta_k == 2 && p->patch_size >= 2 &&
                       p->patch_size <= 40 && p->nfeatures >= 1 && p->scale_factor > 1.f && (p->score_type == 0 || p->score_type == 1) &&
                       p->edge_threshold >= 0 && p->fast_threshold >= 1 && p->fast_threshold < 255,
              MIS_E_UNSUPPORTED, "unsupported ORB parameters (need first_level 0, wta_k 2, patch <= 40, nlevels <= 16)");
    MIS_CHECK(ctx, max_w >= 64 && max_h >= 64 && max_w <= 32767 && max_h <= 32767, MIS_E_INVALID, "bad maximum size");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MisOrb* o = new MisOrb();
    o->ctx = ctx; o->p = *p; o->max_w = max_w; o->max_h = max_h;
    plan_levels(o, max_w, max_h);
    o->out_cap = o->fin_total;
    // umax (orb.cpp) and the random BRIEF pattern: patchSize != 31 -> RNG(0x34985739), 512 points
    int umax[64] = {0};
    {
        int hp = p->patch_size / 2, v, v0;
        int vmax = (int)floor((double)((float)hp * sqrtf(2.f) / 2 + 1));
        int vmin = (int)ceil((double)((float)hp * sqrtf(2.f) / 2));
        for (v = 0; v <= vmax; ++v) umax[v] = (int)lrint(sqrt((double)hp * hp - (double)v * v));
        for (v = hp, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }
    int8_t pat[1024];
    {
        uint64_t state = 0x34985739u;
        auto next = [&]() { state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32); return (uint32_t)state; };
        int hp = p->patch_size / 2;
        for (int i = 0; i < 1024; i++) pat[i] = (int8_t)(int)(next() % (uint32_t)(2 * hp + 1) + (uint32_t)(-hp));
    }
    {
        int max_sq = 0;
        for (int i = 0; i < 512; i++) max_sq = std::max(max_sq, (int)pat[2 * i] * pat[2 * i] + (int)pat[2 * i + 1] * pat[2 * i + 1]);
        // |round(x cos - y sin)| <= round(|(x, y)|); the patch must also stay inside the padded level (border ring of ORB_BORDER)
        const int reach = (int)floor(sqrt((double)max_sq) + 0.5);
        // FAST keypoints keep 3 pixels from the edge, the level carries a border ring of ORB_BORDER: the patch stays inside the padded level
        o->direct_describe = reach <= DD_R && 3 + ORB_BORDER >= DD_P && getenv("MIS_ORB_FULL_BLUR") == nullptr;
    }
    int rc0 = orb_alloc_workspace(o, 1);      // (after direct_describe is known: it decides whether a frame's block carries a blurred pyramid)
    if (rc0 != MIS_OK) { delete o; return rc0; }
    memcpy(o->umax_host, umax, sizeof(umax));
    memcpy(o->pattern_host, pat, sizeof(pat));
    o->cur_w = max_w; o->cur_h = max_h;
    int rc = orb_init_workspace(o);
    if (rc != MIS_OK) { hipFree(o->mem); delete o; return rc; }
    *out = o;
    return MIS_OK;
}

extern "C" int mis_orb_destroy(MisOrb* o) {
    if (!o) return MIS_OK;
    hipSetDevice(o->ctx->device);
    hipStreamSynchronize(o->ctx->stream);
    if (o->mem) hipFree(o->mem);
    if (o->host_counts) hipHostFree(o->host_counts);
    delete o;
    return MIS_OK;
}


extern "C" int mis_orb_detect_batch(MisOrb* o, const MisImage* imgs, int n, MisFeatures* out) {
    if (!o) return MIS_E_INVALID;
    MisContext* ctx = o->ctx;
    MIS_CHECK(ctx, imgs && out && n >= 1, MIS_E_INVALID, "null argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    for (int i = 0; i < n; i++) {
        if ((rc = check_image(o, &imgs[i])) != MIS_OK) return rc;
        MIS_CHECK(ctx, imgs[i].width == imgs[0].width && imgs[i].height == imgs[0].height, MIS_E_INVALID, "batch frames must share one size");
    }
    if ((rc = replan_if_needed(o, imgs[0].width, imgs[0].height)) != MIS_OK) return rc;
    if ((rc = orb_ensure_frames(o, n)) != MIS_OK) return rc;
    std::vector<DevImage> dimg(n);
    for (int i = 0; i < n; i++) memset(&out[i], 0, sizeof(MisFeatures));
    // an error must not leak the outputs / staged inputs of the frames already set up
    auto fail = [&](int code) {
        hipStreamSynchronize(ctx->stream);
        for (int i = 0; i < n; i++) {
            if (out[i].keypoints || out[i].descriptors) mis_features_free(ctx, &out[i]);
            if (dimg[i].data) mis_dev_image_release(ctx, &dimg[i]);
        }
        return code;
    };
    for (int i = 0; i < n; i++) {
        out[i].img_idx = i; out[i].img_w = imgs[i].width; out[i].img_h = imgs[i].height;
        if ((rc = alloc_features(ctx, o->out_cap, 32, MIS_U8, &out[i])) != MIS_OK) return fail(rc);
        if ((rc = mis_dev_image_in(ctx, &imgs[i], &dimg[i])) != MIS_OK) return fail(rc);
    }
    const bool trace = getenv("MIS_ORB_TRACE") != nullptr;
    const auto t_enq0 = std::chrono::steady_clock::now();
    // groups of up to ORB_BATCH frames, each stage of a group in one launch; groups follow each other on the stream (they share the workspaces)
    for (int g0 = 0; g0 < n; g0 += o->ws_frames) {
        const int ng = std::min(o->ws_frames, n - g0);
        if ((rc = enqueue_detect_group(o, &dimg[g0], imgs[0].width, imgs[0].height, &out[g0], ng)) != MIS_OK) return fail(rc);
    }
    if (trace) fprintf(stderr, "orb batch: %d frames enqueued in %.0f us\n", n, (double)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_enq0).count());
    // one-shot hook (mis_orb_on_enqueued): the caller's own host work -- e.g. the job's warpRoi + blender sizing, which end in a
    // synchronisation of ANOTHER stream -- runs here, under the batch's 2 ms of device work, on the thread that would only wait
    if (o->enqueued_cb) {
        void (*cb)(void*) = o->enqueued_cb;
        void* user = o->enqueued_user;
        o->enqueued_cb = nullptr; o->enqueued_user = nullptr;
        cb(user);
        MIS_HIP(ctx, hipSetDevice(ctx->device));
    }
    // one synchronisation for the whole batch: counts + the workspaces' overflow flags
    std::vector<int> counts(n);
    const int nl = o->ws_frames;
    std::vector<int> ws_flags(nl, 0);
    if (!o->host_counts) MIS_HIP(ctx, hipHostMalloc((void**)&o->host_counts, sizeof(int) * ORB_GATHER_MAX, hipHostMallocMapped));
    for (int i0 = 0; i0 < n + nl; i0 += ORB_GATHER_MAX) {
        const int m = std::min(ORB_GATHER_MAX, n + nl - i0);
        GatherPtrs gp;
        for (int k = 0; k < m; k++)
            gp.p[k] = i0 + k < n ? feat_count(&out[i0 + k], o->out_cap) : (const int*)((const char*)o->w.flags + (size_t)(i0 + k - n) * o->ws_stride);
        int* dev_view = nullptr;
        MIS_HIP(ctx, hipHostGetDevicePointer((void**)&dev_view, o->host_counts, 0));
        hipLaunchKernelGGL(gather_ints_kernel, dim3(1), dim3(ORB_GATHER_MAX), 0, ctx->stream, gp, m, dev_view);
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < m; k++) (i0 + k < n ? counts[i0 + k] : ws_flags[i0 + k - n]) = o->host_counts[k];
    }
    int flags = 0;
    for (int k = 0; k < nl; k++) flags |= ws_flags[k];
    if (flags) hipMemset2DAsync(o->w.flags, o->ws_stride, 0, sizeof(int), (size_t)nl, ctx->stream);
    for (int i = 0; i < n; i++) { out[i].n = counts[i]; mis_dev_image_release(ctx, &dimg[i]); }
    if (flags) {
        return mis_set_error(ctx, MIS_E_OVERFLOW, "ORB candidate buffers overflowed (flags %d): too many tied scores", flags);
    }
    return MIS_OK;
}

extern "C" int mis_orb_detect(MisOrb* o, const MisImage* bgr, MisFeatures* out) { return mis_orb_detect_batch(o, bgr, 1, out); }

extern "C" int mis_orb_on_enqueued(MisOrb* o, void (*fn)(void*), void* user) {
    if (!o) return MIS_E_INVALID;
    o->enqueued_cb = fn; o->enqueued_user = user;
    return MIS_OK;
}

extern "C" int mis_features_download(MisContext* ctx, const MisFeatures* f, MisKeyPoint* kps, void* desc) {
    if (!ctx || !f) return MIS_E_INVALID;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    if (f->n > 0) {
        if (kps) MIS_HIP(ctx, hipMemcpyAsync(kps, f->keypoints, sizeof(MisKeyPoint) * (size_t)f->n, hipMemcpyDeviceToHost, ctx->stream));
        if (desc) MIS_HIP(ctx, hipMemcpyAsync(desc, f->descriptors, (size_t)f->n * f->desc_cols * mis_dtype_size(f->desc_dtype), hipMemcpyDeviceToHost, ctx->stream));
    }
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIS_OK;
}

extern "C" int mis_features_upload(MisContext* ctx, int img_w, int img_h, int n, const MisKeyPoint* kps, const void* desc, int desc_cols,
                                   int desc_dtype, MisFeatures* out) {
    if (!ctx || !out) return MIS_E_INVALID;
    MIS_CHECK(ctx, n >= 0 && (n == 0 || (kps && desc)) && desc_cols > 0 && (desc_dtype == MIS_U8 || desc_dtype == MIS_F32), MIS_E_INVALID,
              "bad feature arrays");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    out->img_w = img_w; out->img_h = img_h;
    int rc = alloc_features(ctx, std::max(n, 1), desc_cols, desc_dtype, out);
    if (rc != MIS_OK) return rc;
    if (n) {
        MIS_HIP(ctx, hipMemcpyAsync(out->keypoints, kps, sizeof(MisKeyPoint) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        MIS_HIP(ctx, hipMemcpyAsync(out->descriptors, desc, (size_t)n * desc_cols * mis_dtype_size(desc_dtype), hipMemcpyHostToDevice, ctx->stream));
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    out->n = n;
    return MIS_OK;
}

extern "C" int mis_features_free(MisContext* ctx, MisFeatures* f) {
    if (!ctx || !f) return MIS_E_INVALID;
    if (f->owner_) {
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> lock(g_feat_mutex);
            auto it = g_feat_sizes.find(f->owner_);
            if (it != g_feat_sizes.end()) { bytes = it->second; g_feat_sizes.erase(it); }
        }
        if (bytes) mis_pool_free(ctx, f->owner_, bytes);  // recycled without synchronising the device
        else {
            MIS_HIP(ctx, hipSetDevice(ctx->device));
            MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            MIS_HIP(ctx, hipFree(f->owner_));
        }
    }
    f->owner_ = nullptr; f->keypoints = nullptr; f->descriptors = nullptr; f->n = 0;
    return MIS_OK;
}

extern "C" int mis_orb_debug_level(MisOrb* o, int level, int which, uint8_t* host_out, int* width, int* height) {
    if (!o) return MIS_E_INVALID;
    MisContext* ctx = o->ctx;
    MIS_CHECK(ctx, level >= 0 && level < o->L.n && which >= 0 && which <= 2, MIS_E_INVALID, "bad level / selector");
    const LevelDesc& d = o->L.d[level];
    int w = which == 2 ? d.w + 2 * ORB_BORDER : d.w, h = which == 2 ? d.h + 2 * ORB_BORDER : d.h;
    if (width) *width = w;
    if (height) *height = h;
    if (!host_out) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    const uint8_t* src;
    size_t pitch;
    if (which == 0) { src = o->w.pad + d.pad_off + (size_t)ORB_BORDER * d.pp + ORB_BORDER; pitch = d.pp; }
    else if (which == 1) {
        MIS_HIP(ctx, hipMemsetAsync(o->w.nms + d.map_off, 0, (size_t)d.sp * d.h, ctx->stream));
        hipLaunchKernelGGL(nms_raster_kernel, dim3(256), dim3(64), 0, ctx->stream, d, o->w.tile_cnt, o->w.surv_xy, o->w.surv_sc, o->w.nms);
        src = o->w.nms + d.map_off; pitch = d.sp;
    }
    else {
        if (o->direct_describe) {   // the detector no longer blurs the pyramid (describe_direct_kernel): this view computes it on demand
            const int pw0 = o->L.d[0].w + 2 * ORB_BORDER, ph0 = o->L.d[0].h + 2 * ORB_BORDER;
            hipLaunchKernelGGL(blur_kernel, dim3((pw0 + 63) / 64, (ph0 + 31) / 32, o->L.n), dim3(256), 0, ctx->stream, o->L, o->w.pad, o->w.blur);
        }
        src = o->w.blur + d.pad_off; pitch = d.pp;
    }
    MIS_HIP(ctx, hipMemcpy2DAsync(host_out, w, src, pitch, w, h, hipMemcpyDeviceToHost, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIS_OK;
}

#ifdef MIS_ORB_STATS
extern "C" int mis_debug_orb_stats(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_orb_stats), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_orb_stats), z, sizeof(z)); }
    return 0;
}
#endif
