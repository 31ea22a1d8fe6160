// match.hip -- pairwise matching, replaces the reference's
// makePtr<BestOf2NearestMatcher>(try_cuda, match_conf) (image_stitching/image_stitching.cpp:647) and
// (*matcher)(features, pairwise_matches) (:653), plus myLeaveBiggestComponent (:215-278) on its output.
//
// Every stage is batched over ALL image pairs:
//   1. knn2_hamming_kernel   exact 2-NN (distance, trainIdx) of every descriptor, both directions
//   2. ratio_union_kernel    ratio test + de-duplicated union, ordered like the reference, plus the
//                            centre-shifted point lists for findHomography
//   3. homography.hip         batched findHomography(RANSAC) over all pairs, run twice (all matches,
//                            then inliers only) exactly as BestOf2NearestMatcher::match does.
#include "common.h"
#include <thread>
#include <chrono>
#include <atomic>
#include "dev_math.h"
#include "homography.h"
#include <algorithm>
#include <vector>

namespace {

struct FeatDev {
    const uint8_t* desc;
    const MisKeyPoint* kps;
    int n, w, h;
};

struct PairDesc {
    int i, j;            // image indices, i < j
    size_t knn_off12, knn_off21;  // offsets (in queries) into idx2 / dist2
    size_t m_off;        // offset into the per-pair match / point / mask arrays
    int cap;             // n_i + n_j
};

// ---------------------------------------------------------------- K7: exact 2-NN, Hamming-256 --
constexpr int KNN_TILE = 256;
__global__ __launch_bounds__(256) void knn2_hamming_kernel(const FeatDev* feats, const PairDesc* pairs, int* idx2, float* dist2) {
    __shared__ uint4 tr[KNN_TILE * 2];
    const PairDesc pd = pairs[blockIdx.y >> 1];
    const bool fwd = (blockIdx.y & 1) == 0;
    const FeatDev Q = feats[fwd ? pd.i : pd.j], T = feats[fwd ? pd.j : pd.i];
    const size_t off = fwd ? pd.knn_off12 : pd.knn_off21;
    const int q0 = blockIdx.x * 256;
    if (q0 >= Q.n) return;
    const int q = q0 + threadIdx.x;
    const bool active = q < Q.n;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (active) {
        const uint4* p = reinterpret_cast<const uint4*>(Q.desc + (size_t)q * 32);
        a0 = p[0]; a1 = p[1];
    }
    // running top two as keys (distance << 22 | train index): the order of the keys is the (distance, index) order of the
    // scalar loop ("d < d0" keeps the earlier train on ties), and the update is a min and a median-of-three
    int k0 = 0x7fffffff, k1 = 0x7fffffff;
    for (int t0 = 0; t0 < T.n; t0 += KNN_TILE) {
        const int nt = min(KNN_TILE, T.n - t0);
        __syncthreads();
        const uint4* src = reinterpret_cast<const uint4*>(T.desc + (size_t)t0 * 32);
        for (int k = threadIdx.x; k < nt * 2; k += 256) tr[k] = src[k];
        __syncthreads();
        if (active) {
            auto one = [&](int j) {
                const uint4 b0 = tr[2 * j], b1 = tr[2 * j + 1];
                // v_bcnt_u32_b32 adds its count to a third operand: eight xor + eight chained bcnt per distance (the compiler
                // prefers independent counts and add3 trees: 3.5 more instructions per distance in an issue-bound kernel)
                unsigned d = 0;
                auto acc = [&](unsigned x) { asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x)); };
                acc(a0.x ^ b0.x); acc(a0.y ^ b0.y); acc(a0.z ^ b0.z); acc(a0.w ^ b0.w);
                acc(a1.x ^ b1.x); acc(a1.y ^ b1.y); acc(a1.z ^ b1.z); acc(a1.w ^ b1.w);
                const int key = (int)((d << 22) | (unsigned)(t0 + j));
                // k0 <= k1 always: the new second best is the median of (k0, k1, key), the new best the minimum
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(k1) : "v"(k0), "v"(k1), "v"(key));
                k0 = min(k0, key);
            };
            // eight trains per trip: LDS offsets and the index part of the key become immediates / scalar adds
            int j = 0;
            for (; j + 8 <= nt; j += 8) {
#pragma unroll
                for (int u = 0; u < 8; u++) one(j + u);
            }
            for (; j < nt; j++) one(j);
        }
    }
    if (active) {
        const int d0 = k0 == 0x7fffffff ? 1 << 30 : k0 >> 22, d1 = k1 == 0x7fffffff ? 1 << 30 : k1 >> 22;
        idx2[(off + q) * 2] = k0 == 0x7fffffff ? -1 : (k0 & 0x3fffff); idx2[(off + q) * 2 + 1] = k1 == 0x7fffffff ? -1 : (k1 & 0x3fffff);
        dist2[(off + q) * 2] = (float)d0; dist2[(off + q) * 2 + 1] = (float)d1;
    }
}

// ---------------------------------------------------------------- K7 on the matrix cores --------
// Hamming-256 as an exact dense product on the matrix cores (north star: "MFMA only if descriptor distance is cast as a dense
// GEMM").  Every descriptor bit becomes a signed element -- +v where a train bit is set, -v where it is clear, the opposite signs
// for a query -- so a product is -v^2 where the bits agree and +v^2 where they differ, and the 256-term dot product is
// 2 v^2 hamming - 256 v^2.  With v^2 = 4096 and the accumulator preloaded with 2^20 + trainIdx the chain delivers the search key
// 8192 hamming + trainIdx itself: the running best two of a query are a minimum and a median of three per candidate, nothing
// else.  Keys need trainIdx < 8192: larger train sets take the vector-pipe kernel above (56 cycles per wave-distance).
// History: round 2 ran this on v_mfma_i32_32x32x32_i8 with bytes +-64 (0.83 ms per 16 x 4K job), round 3
// paced it by the matrix pipe (VGPR accumulators that ping-pong between tiles, float-rate key selections, an XCD-ordered
// workgroup table: 0.65 ms), staged the train tiles by LDS-DMA (0.58 ms) and moved it to the fp4 operands below (0.41 ms); the
// int8 kernels were removed in round 4 once nothing selected them.
// Layout: accumulator register g of lane (r, h) = query r (B column) x train (g & 3) + 8 (g >> 2) + 4 h (A row) of the tile;
// a lane's two running keys per query set cover half of the trains, the halves are merged at the end.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int HM_ROWPAD = 256;         // expanded blocks are padded with zero rows to a multiple of this
constexpr int HM_MAX_TRAINS = 8192;
constexpr int HM_NONE = 0x7f000000;

struct HmFrame { size_t train_off, query_off; };   // byte offsets of a frame's two expanded forms

struct HmKeys { int k0[2], k1[2]; };
// One workgroup of the pass: 256 queries of a directed pair against all trains.  The table is ordered so that workgroup L runs on
// XCD L mod 8 (the dispatcher deals workgroups round-robin over the eight XCDs) and every XCD only ever sees the TRAIN sets of
// the frames t with t mod 8 = its number: two frames of a 16-frame job, 2 MB of expanded descriptors, resident in its 4 MB L2
// (with the (query block, pair) grid every XCD walked all 16 MB and every tile was a miss to the fabric: the tile loads' latency,
// not the matrix pipe, paced the loop).
struct HmJob { int pair, dir, q0, pad; };
// The keys carry HM_FBIAS = the bits of 1.0f: bit patterns of positive normal floats order like the integers they are, and
// v_min_f32 / v_med3_f32 issue at the full vector rate where v_min_i32 / v_med3_i32 take two issue slots (tools/valu_rate.hip) --
// 64 of them per tile and wave were 512 of the 700 vector-issue cycles against 512 cycles of matrix work (rocprofv3: the
// matrix pipe 59 % busy, the waves stalled on issue).  The selections return one of their operands bit for bit.
constexpr int HM_FBIAS = 0x3F800000;
__device__ __forceinline__ void hm_update(HmKeys& K, int set, int key) {
    // k0 <= k1 always: the new second best is the median of (k0, k1, key), the new best the minimum
    // (volatile: the statements keep their place between the MFMAs -- the scheduler otherwise issues a slot pair's two MFMAs back
    // to back, the second waits 24 cycles for the pipe, and the vector work follows while the pipe idles: 92 cycles per pair for 64)
    asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(K.k1[set]) : "v"(K.k0[set]), "v"(K.k1[set]), "v"(key));
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(K.k0[set]) : "v"(K.k0[set]), "v"(key));
}
// Train tiles are copied global -> LDS by LDS-DMA: no staging registers, no ds_write, the copy of tile t + 2 runs beside tile t's
// MFMAs.  A DMA instruction writes 64 x 16 bytes to consecutive LDS addresses, so the image's 16-byte slots are XOR-swizzled (below)
// and each lane fetches the slot its LDS position holds.  The instructions are inline assembly (the compiler would wait for
// every copy before the next LDS read); the wait is the s_waitcnt vmcnt(0) in front of the tile's barrier.
__device__ __forceinline__ void hm_dma(const int8_t* base, unsigned voff, uint32_t lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_off) : "memory", "m0");
}
// The fp4 path of the matrix cores (knn2_hamming_fp4_kernel): v_mfma_scale_f32_32x32x64_f8f6f4 multiplies 64 E2M1 values per lane
// pair and instruction in the time the int8 instruction takes for 32, so a 32 x 32 tile of distances is 4 instructions per query set.  A descriptor bit becomes one nibble: +4 (0x6) where a train bit is set, -4 (0xE) where
// it is clear, the opposite signs for a query; both operands carry the block scale 2^4, so a product is -4096 where the bits agree
// and +4096 where they differ, and a row of 256 sums to 8192 hamming - 2^20.  The seed of the accumulator is the float 2^20 +
// trainIdx: every partial sum is an integer below 2^24, exact in f32, and the result is the float 8192 hamming + trainIdx -- ordered
// like the int8 kernel's keys, and v_min_f32 / v_med3_f32 take it as it is.  Expanded rows are 128 bytes (a tile of 32 trains =
// 4 KB = one LDS-DMA instruction per wave); slot s of row R sits at position s ^ ((R >> 1) & 7) of its row, which spreads the
// fragment reads of 16 consecutive rows (two rows per 256 bytes) over 16 different 16-byte positions.
__global__ __launch_bounds__(256) void hamming_expand4_kernel(const FeatDev* feats, const HmFrame* fr, int8_t* out) {
    const FeatDev F = feats[blockIdx.y];
    const int npad = (F.n + HM_ROWPAD - 1) / HM_ROWPAD * HM_ROWPAD;
    const int item = blockIdx.x * 256 + threadIdx.x, d = item >> 3, w = item & 7;
    if (d >= npad) return;
    const bool live = d < F.n;
    const unsigned x = live ? reinterpret_cast<const unsigned*>(F.desc)[(size_t)d * 8 + w] : 0u;
    unsigned tq[4], qq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned t = (x >> (8 * k)) & 255u;                 // bit i -> bit 4 i + 3 (the sign of nibble i)
        t = (t | (t << 12)) & 0x000F000Fu;
        t = (t | (t << 6)) & 0x03030303u;
        t = (t | (t << 3)) & 0x11111111u;
        tq[k] = live ? (0xEEEEEEEEu ^ (t << 3)) : 0u;       // +4 (set) / -4 (clear)
        qq[k] = live ? (0x66666666u ^ (t << 3)) : 0u;       // -4 (set) / +4 (clear)
    }
    *reinterpret_cast<uint4*>(out + fr[blockIdx.y].train_off + (size_t)d * 128 + 16 * w) = make_uint4(tq[0], tq[1], tq[2], tq[3]);
    *reinterpret_cast<uint4*>(out + fr[blockIdx.y].query_off + (size_t)d * 128 + 16 * w) = make_uint4(qq[0], qq[1], qq[2], qq[3]);
}
#define HM4_MFMA_FIRST(D, A, B, C, SA, SB) \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "=&v"(D) : "v"(A), "v"(B), "v"(C), "v"(SA), "v"(SB))
#define HM4_MFMA_ACC(D, A, B, SA, SB) \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(D) : "v"(A), "v"(B), "v"(SA), "v"(SB))
typedef float v16f __attribute__((ext_vector_type(16)));
// The key selections of the fp4 pass, three keys of each query set at a time: of (x, y, z) only the minimum m1 and the median m2 can
// enter a query's best two (the keys of a query are distinct: they carry the train index), and merging the sorted pair (m1, m2)
// into the sorted pair (k0, k1) is k1' = med3(k0, m1, min(k1, m2)) -- min(k1, m2) is never below min(k0, m1), so the median is
// min(max(k0, m1), min(k1, m2)), the second smallest of the four -- and k0' = min(k0, m1): 5 instructions for 3 keys where the
// key-by-key form takes 6.  One statement per group of both sets: between two statements of inline assembly of which the second
// reads a register the first wrote the compiler pads a wait state (s_nop) that plain vector instructions do not need.
__device__ __forceinline__ void hm_update_triples(HmKeys& K, int x0, int x1, int x2, int y0, int y1, int y2) {
    int t0, t1, u0, u1;
    asm volatile(
        "v_min3_f32 %4, %8, %9, %10\n\tv_med3_f32 %5, %8, %9, %10\n\tv_min3_f32 %6, %11, %12, %13\n\tv_med3_f32 %7, %11, %12, %13\n\t"
        "v_min_f32 %5, %1, %5\n\tv_min_f32 %7, %3, %7\n\t"
        "v_med3_f32 %1, %0, %4, %5\n\tv_med3_f32 %3, %2, %6, %7\n\t"
        "v_min_f32 %0, %0, %4\n\tv_min_f32 %2, %2, %6"
        : "+v"(K.k0[0]), "+v"(K.k1[0]), "+v"(K.k0[1]), "+v"(K.k1[1]), "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1)
        : "v"(x0), "v"(x1), "v"(x2), "v"(y0), "v"(y1), "v"(y2));
}
// one key of each query set
__device__ __forceinline__ void hm_update_both(HmKeys& K, int a, int b) {
    asm volatile("v_med3_f32 %1, %0, %1, %4\n\tv_min_f32 %0, %0, %4\n\tv_med3_f32 %3, %2, %3, %5\n\tv_min_f32 %2, %2, %5"
                 : "+v"(K.k0[0]), "+v"(K.k1[0]), "+v"(K.k0[1]), "+v"(K.k1[1]) : "v"(a), "v"(b));
}
template <bool UPD>
__device__ __forceinline__ void hm_tile_fp4(v16i& accA, v16i& accB, const v16i& prevA, const v16i& prevB, v16f& cb, v4i* a, const v4i (&bq)[2][4], HmKeys& K,
                                            const int8_t* lds_next, uint32_t lds_store, const int8_t* gnext, unsigned voff, const unsigned (&aoff)[4], int sa, int sb) {
    hm_dma(gnext, voff, lds_store);                 // tile t + 2 into the buffer tile t left
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int slot = 2 * s4 + half;
            if (half == 0) { if (s4 == 0) HM4_MFMA_FIRST(accA, a[0], bq[0][0], cb, sa, sb); else HM4_MFMA_ACC(accA, a[s4], bq[0][s4], sa, sb); }
            else {
                if (s4 == 0) HM4_MFMA_FIRST(accB, a[0], bq[1][0], cb, sa, sb); else HM4_MFMA_ACC(accB, a[s4], bq[1][s4], sa, sb);
                a[s4] = *reinterpret_cast<const v4i*>(lds_next + aoff[s4]);      // the next tile's fragment
            }
            // the previous tile's 32 keys over slots 2 .. 7 (its chains ended in its slots 6 and 7): keys 3 j .. 3 j + 2 of both query
            // sets in slot 2 + j, key 15 in slot 7; the seeds advance in slots 4 .. 7
            if (slot >= 2) {
                if (UPD) {
                    const int j = slot - 2;
                    if (j < 5) hm_update_triples(K, prevA[3 * j], prevA[3 * j + 1], prevA[3 * j + 2], prevB[3 * j], prevB[3 * j + 1], prevB[3 * j + 2]);
                    else hm_update_both(K, prevA[15], prevB[15]);
                }
                if (slot >= 4) {
#pragma unroll
                    for (int g = (slot - 4) * 4; g < (slot - 3) * 4; g++) cb[g] += 32.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__global__ __launch_bounds__(256) void knn2_hamming_fp4_kernel(const FeatDev* feats, const PairDesc* pairs, const HmFrame* fr, const int8_t* __restrict__ xp, int* idx2,
                                                               float* dist2, const HmJob* jobs) {
    __shared__ __attribute__((aligned(1024))) int8_t tr[2][32 * 128];
    const HmJob job = jobs[blockIdx.x];
    if (job.pair < 0) return;                       // padding of the XCD interleave
    const PairDesc pd = pairs[job.pair];
    const bool fwd = job.dir == 0;
    const int qi = fwd ? pd.i : pd.j, ti = fwd ? pd.j : pd.i;
    const int nq = feats[qi].n, nt = feats[ti].n;
    const size_t off = fwd ? pd.knn_off12 : pd.knn_off21;
    const int q0 = job.q0;
    if (q0 >= nq) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    const int8_t* qx = xp + fr[qi].query_off;
    const int8_t* tx = xp + fr[ti].train_off;
    HmKeys K;
    K.k0[0] = K.k0[1] = K.k1[0] = K.k1[1] = HM_NONE;           // (as a float: 1.7e38)
    const int ntiles = (nt + 31) / 32;
    if (ntiles > 0) {
        // the wave's 2 x 32 queries as B fragments (step s, lane (r, h): nibbles 64 s + 32 h .. + 31 of query r)
        v4i bq[2][4];
#pragma unroll
        for (int set = 0; set < 2; set++)
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++)
                bq[set][s4] = *reinterpret_cast<const v4i*>(qx + (size_t)(q0 + wave * 64 + set * 32 + r) * 128 + 32 * s4 + 16 * h);
        v16f cb;
#pragma unroll
        for (int g = 0; g < 16; g++) cb[g] = (float)((1 << 20) + (g & 3) + 8 * (g >> 2) + 4 * h);
        const int sa = 127 + 4, sb = 127 + 4;       // E8M0 block scales 2^4 (byte 0 of the scale operands)
        // the wave's copy instruction of a tile: row 8 wave + (lane >> 3), LDS position lane & 7 <- global slot position ^ ((row >> 1) & 7)
        const int R = 8 * wave + (lane >> 3);
        const unsigned voff = (unsigned)(R * 128 + 16 * ((lane & 7) ^ ((R >> 1) & 7)));
        unsigned aoff[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) aoff[s4] = (unsigned)(r * 128 + 16 * ((2 * s4 + h) ^ ((r >> 1) & 7)));
        const uint32_t lds0 = (uint32_t)(uintptr_t)&tr[0][0] + (uint32_t)wave * 1024, lds1 = (uint32_t)(uintptr_t)&tr[1][0] + (uint32_t)wave * 1024;
        auto gtile = [&](int t) { return tx + (size_t)min(t, ntiles - 1) * 32 * 128; };      // (the blocks are padded: any tile of the set is readable)
        hm_dma(gtile(0), voff, lds0);
        hm_dma(gtile(1), voff, lds1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        v4i a[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) a[s4] = *reinterpret_cast<const v4i*>(&tr[0][aoff[s4]]);
        __syncthreads();                            // (as in knn2_hamming_mfma4_kernel: tile 0 is in every wave's registers before tile 2 is copied over it)
        v16i A0, B0, A1, B1;
        hm_tile_fp4<false>(A0, B0, A1, B1, cb, a, bq, K, tr[1], lds0, gtile(2), voff, aoff, sa, sb);
        int t = 1;
        for (; t + 1 < ntiles; t += 2) {
            hm_tile_fp4<true>(A1, B1, A0, B0, cb, a, bq, K, tr[0], lds1, gtile(t + 2), voff, aoff, sa, sb);
            hm_tile_fp4<true>(A0, B0, A1, B1, cb, a, bq, K, tr[1], lds0, gtile(t + 3), voff, aoff, sa, sb);
        }
        const bool odd_last = t < ntiles;
        if (odd_last) hm_tile_fp4<true>(A1, B1, A0, B0, cb, a, bq, K, tr[0], lds1, gtile(t + 2), voff, aoff, sa, sb);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int ka = odd_last ? A1[g] : A0[g], kb = odd_last ? B1[g] : B0[g];
            hm_update(K, 0, ((int)__int_as_float(ka) & (HM_MAX_TRAINS - 1)) < nt ? ka : HM_NONE);
            hm_update(K, 1, ((int)__int_as_float(kb) & (HM_MAX_TRAINS - 1)) < nt ? kb : HM_NONE);
        }
    }
#pragma unroll
    for (int set = 0; set < 2; set++) {
        // (bit patterns of non-negative floats order like the floats)
        const int o0 = __shfl_xor(K.k0[set], 32), o1 = __shfl_xor(K.k1[set], 32);
        const int b0 = min(K.k0[set], o0), b1 = min(max(K.k0[set], o0), min(K.k1[set], o1));
        const int q = q0 + wave * 64 + set * 32 + r;
        if (h == 0 && q < nq) {
            const bool v0 = b0 < HM_NONE, v1 = b1 < HM_NONE;
            const int i0 = v0 ? (int)__int_as_float(b0) : 0, i1 = v1 ? (int)__int_as_float(b1) : 0;
            idx2[(off + q) * 2] = v0 ? (i0 & (HM_MAX_TRAINS - 1)) : -1; idx2[(off + q) * 2 + 1] = v1 ? (i1 & (HM_MAX_TRAINS - 1)) : -1;
            dist2[(off + q) * 2] = (float)(v0 ? i0 >> 13 : 1 << 30); dist2[(off + q) * 2 + 1] = (float)(v1 ? i1 >> 13 : 1 << 30);
        }
    }
}

// ---------------------------------------------------------------- K8: exact 2-NN, L2 on MFMA ----
// SIFT descriptors are integer valued (0..255, stored as f32): they are exact in fp16, every dot
// product of two 128-D descriptors is an integer < 2^24 and therefore exact in the f32 accumulator of
// v_mfma_f32_32x32x16_f16, so |a-b|^2 = |a|^2 + |b|^2 - 2 a.b comes out bit-exact and the 2-NN order
// (distance, trainIdx) equals the CPU's.  Kernel structure: see l2_knn2_mfma_kernel below.
struct L2Set {
    const _Float16* h;  // n x 128 fp16
    const float* nrm;   // n squared norms (exact integers)
    int n;
};
__global__ __launch_bounds__(256) void l2_prep_kernel(const float* desc, int n, int dim, _Float16* h, float* nrm, int* bad) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float s = 0;
    for (int k = lane; k < 128; k += 64) {
        float v = k < dim ? desc[(size_t)i * dim + k] : 0.f;
        if (!(v >= 0.f && v <= 255.f && v == floorf(v))) atomicOr(bad, 1);  // not a SIFT-style integer descriptor
        h[(size_t)i * 128 + k] = (_Float16)v;
        s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) nrm[i] = s;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void top2_insert(float d, int i, float& d0, int& i0, float& d1, int& i1) {
    if (d < d0 || (d == d0 && i < i0)) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
    else if (d < d1 || (d == d1 && i < i1)) { d1 = d; i1 = i; }
}

typedef int int16v __attribute__((ext_vector_type(16)));
constexpr int L2_MAX_SLICES = 16;


// A workgroup of 4 waves owns 256 queries (two sets of 32 per wave, their fragments in registers) and walks the train set in
// tiles of 32 descriptors that are staged ONCE per workgroup in LDS (double buffered, rows padded to 272 bytes so
// the 16-byte fragment reads of the 32 lanes fall into different banks).  Per tile and query set a wave issues 8 MFMAs
// (32 x 32 x 128) with the TRAIN tile as the A operand and the queries as B: the accumulator of lane (r, hh) then
// holds, for query r of the wave, the dot products with the 16 trains (g & 3) + 8 (g >> 2) + 4 hh of the tile --
// the search dimension lies along the lane's registers, so the running top two of a query are four registers of
// one lane (two lanes per query, merged at the end), the filter "below my current second best" needs no cross-lane
// traffic, and after the first tiles a tile costs the MFMAs plus ~25 vector instructions (16 fused e = |t|^2 - 2 q.t,
// a min3 tree, one compare).  An earlier version kept queries along the registers and one top-two per (register,
// column class): its shared bound was so loose that the insertion ran on nearly every tile (0.72 ms per 24 k x 24 k).
// The query norm is the same for every candidate of a query, so the ordering is decided on e (an exact integer,
// possibly negative); |q|^2 is added at the end.  Within a lane the train index only grows, so strict comparisons
// keep the earlier train on ties, as the CPU's (distance, index) order does.  The train set is split into
// `gridDim.y` slices so that small query sets still fill the device; slice results go to part_* and a second kernel
// merges them by (distance, index).
struct Top2 { float e0, e1; int i0, i1; };
// one tile's 16 candidates of a lane (e = |t|^2 - 2 q.t from the accumulator) against the lane's running top two
__device__ __forceinline__ void l2_tile_update(const float16v& acc, const float (&tn)[16], int ib, Top2& t) {
    float e[16];
#pragma unroll
    for (int g = 0; g < 16; g++) e[g] = __builtin_fmaf(-2.f, acc[g], tn[g]);   // exact integers below 2^24 (fused or not), ~3e38 for a padded row
    float m4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) m4[j] = fminf(fminf(e[4 * j], e[4 * j + 1]), fminf(e[4 * j + 2], e[4 * j + 3]));
    const float m = fminf(fminf(m4[0], m4[1]), fminf(m4[2], m4[3]));
    if (!__any(m < t.e1)) return;
    // some lane improves its top two: visit only the groups of four trains that hold such a candidate
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!__any(m4[j] < t.e1)) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float v = e[4 * j + k];
            const int idx = ib + 8 * j + k;   // ascending in (j, k): ties keep the earlier entry
            const bool b0 = v < t.e0, b1 = v < t.e1;
            t.i1 = b0 ? t.i0 : (b1 ? idx : t.i1);
            t.e1 = b0 ? t.e0 : (b1 ? v : t.e1);
            t.i0 = b0 ? idx : t.i0;
            t.e0 = b0 ? v : t.e0;
        }
    }
}
#ifndef K8_WG_PER_CU
#define K8_WG_PER_CU 1      // round 4: 2 -> 1 (at 24 k x 24 k: 4 train slices / 376 workgroups instead of 8 / 752 -- every slice restarts the search bound, and
                            // 752 workgroups on 512 slots left a quarter of the second round empty: config 5 81.8 -> 80.6 ms per step)
#endif
#ifndef K8_SUB
#define K8_SUB 1
#endif
constexpr int L2_TB = 32, L2_PITCH = 272, L2_QW = 2, L2_SUB = K8_SUB;   // L2_QW sets of 32 queries per wave, L2_SUB tiles per barrier step
__global__ __launch_bounds__(256, 2) void l2_knn2_mfma_kernel(L2Set Q, L2Set T, int tiles_per_slice, int slices, int qblocks, int* part_idx, float* part_e) {
    // two tiles (64 trains) per barrier: half the synchronisations, twice the work to hide the next fetch behind
    __shared__ __attribute__((aligned(16))) uint8_t sb[2][L2_SUB][L2_TB * L2_PITCH];
    __shared__ __attribute__((aligned(16))) float sn[2][L2_SUB][L2_TB];   // squared norms of the staged trains (3e38 for padded rows)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8 and every XCD has its own L2.  All workgroups of an XCD walk the
    // same slice(s) of the train set (slices is a power of two), so a slice is fetched into one L2 once and then
    // served from it; with slices spread over all XCDs every L2 had to hold the whole train set (6 MB at 24 k) and
    // the tile prefetch, not the MFMAs, set the kernel's time.
    const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
    int slice, qb;
    if (slices >= 8) { const int s8 = slices >> 3; slice = xcd * s8 + grp % s8; qb = grp / s8; }
    else { slice = xcd % slices; qb = grp * (8 / slices) + xcd / slices; }
    if (qb >= qblocks) return;
    const int q0 = (qb * 4 + wave) * (32 * L2_QW);   // this wave's queries: q0 + 32 u + r
    // query fragments: lane holds Q[query][k = 16 s + 8 hh + j] for the 8 k-steps; every train fragment read from
    // LDS feeds L2_QW independent MFMA chains
    static_assert(L2_QW == 2, "two query sets per wave are spelled out below");
    half8 a0[8], a1[8];
    {
        const int qa = min(q0 + r, Q.n - 1), qb = min(q0 + 32 + r, Q.n - 1);
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
            a0[s8] = *reinterpret_cast<const half8*>(Q.h + (size_t)qa * 128 + 16 * s8 + 8 * hh);
            a1[s8] = *reinterpret_cast<const half8*>(Q.h + (size_t)qb * 128 + 16 * s8 + 8 * hh);
        }
    }
    Top2 ta{3.0e38f, 3.0e38f, 0x7fffffff, 0x7fffffff}, tb = ta;
    // staging role of a thread: 16-byte chunk (t & 15) of train rows (t >> 4) and (t >> 4) + 16 of both tiles of a step
    const int srow = threadIdx.x >> 4, schunk = threadIdx.x & 15;
    const int ntiles_all = (T.n + L2_TB - 1) / L2_TB;
    // slices are whole steps (tiles_per_slice is even), so only the last step of the train set can hold a tile past
    // the end; its rows read as zeros with norm 3e38 and never enter a top two
    const int tile_lo = slice * tiles_per_slice, tile_hi = min(ntiles_all, tile_lo + tiles_per_slice);
    // The norms travel with the tiles (threads 0 .. 32 L2_SUB - 1 carry one each), so the loop's only global loads are
    // this prefetch: waiting for anything else in the loop would also wait for it (vmcnt counts in order).
    uint4 st[2 * L2_SUB];
    float stn = 3.0e38f;
    auto fetch = [&](int tile) {
#pragma unroll
        for (int k = 0; k < 2 * L2_SUB; k++) {
            const int t = tile * L2_TB + srow + 16 * k;
            st[k] = t < T.n ? *reinterpret_cast<const uint4*>(T.h + (size_t)t * 128 + 8 * schunk) : make_uint4(0, 0, 0, 0);
        }
        const int tnrm = tile * L2_TB + (int)threadIdx.x;
        stn = (threadIdx.x < L2_SUB * L2_TB && tnrm < T.n) ? T.nrm[tnrm] : 3.0e38f;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 2 * L2_SUB; k++)
            *reinterpret_cast<uint4*>(sb[buf][k >> 1] + (srow + 16 * (k & 1)) * L2_PITCH + 16 * schunk) = st[k];
        if (threadIdx.x < L2_SUB * L2_TB) sn[buf][threadIdx.x >> 5][threadIdx.x & 31] = stn;
    };
    // This first, unconditional wait for all loads also covers the query fragments: were it conditional, the compiler
    // would guard every MFMA of the loop with a vmcnt wait for its fragment, and such a wait also waits for the
    // prefetch issued before it.  A slice past the end of a short train set reports "nothing found".
    if (tile_lo >= tile_hi) {
        if (hh == 0)
            for (int u = 0; u < L2_QW; u++) {
                const int q = q0 + 32 * u + r;
                if (q < Q.n) {
                    const size_t o = ((size_t)slice * Q.n + q) * 2;
                    part_idx[o] = part_idx[o + 1] = 0x7fffffff;
                    part_e[o] = part_e[o + 1] = 3.0e38f;
                }
            }
        return;
    }
    fetch(tile_lo);
    stash(0);
    __syncthreads();
    for (int tile = tile_lo, step = 0; tile < tile_hi; tile += L2_SUB, step++) {
        const int buf = step & 1;
        if (tile + L2_SUB < tile_hi) fetch(tile + L2_SUB);      // in flight during the MFMAs of this step
#pragma unroll
        for (int sub = 0; sub < L2_SUB; sub++) {
            // squared norms of this lane's 16 trains: rows 8 j + 4 hh + (0..3), j = 0..3
            float tn[16];
            const int tbase = (tile + sub) * L2_TB + 4 * hh;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 v = *reinterpret_cast<const float4*>(&sn[buf][sub][4 * hh + 8 * j]);
                tn[4 * j] = v.x; tn[4 * j + 1] = v.y; tn[4 * j + 2] = v.z; tn[4 * j + 3] = v.w;
            }
            float16v acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
            const uint8_t* brow = sb[buf][sub] + r * L2_PITCH + 16 * hh;
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) {
                const half8 b = *reinterpret_cast<const half8*>(brow + 32 * s8);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a0[s8], acc0, 0, 0, 0);   // rows: trains, columns: queries
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a1[s8], acc1, 0, 0, 0);
            }
            l2_tile_update(acc0, tn, tbase, ta);
            l2_tile_update(acc1, tn, tbase, tb);
        }
        if (tile + L2_SUB < tile_hi) stash(buf ^ 1);
        __syncthreads();
    }
    // merge the two lanes of a query (hh = 0 / 1) by (e, index); lanes with hh == 0 write
    auto finish = [&](Top2 t, int q) {
        const float od0 = __shfl_xor(t.e0, 32), od1 = __shfl_xor(t.e1, 32);
        const int oi0 = __shfl_xor(t.i0, 32), oi1 = __shfl_xor(t.i1, 32);
        top2_insert(od0, oi0, t.e0, t.i0, t.e1, t.i1);
        top2_insert(od1, oi1, t.e0, t.i0, t.e1, t.i1);
        if (hh == 0 && q < Q.n) {
            const size_t o = ((size_t)slice * Q.n + q) * 2;
            part_idx[o] = t.i0; part_idx[o + 1] = t.i1;
            part_e[o] = t.e0; part_e[o + 1] = t.e1;
        }
    };
    finish(ta, q0 + r);
    finish(tb, q0 + 32 + r);
}

// merge of the train slices of one query set: top two by (e, index), then distance = sqrt(|q|^2 + e)
__global__ __launch_bounds__(256) void l2_merge_kernel(L2Set Q, int slices, const int* __restrict__ part_idx, const float* __restrict__ part_e, int* __restrict__ idx2,
                                                      float* __restrict__ dist2) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q.n) return;
    float d0 = 3.0e38f, d1 = 3.0e38f;
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int s = 0; s < slices; s++) {
        const size_t o = ((size_t)s * Q.n + q) * 2;
        top2_insert(part_e[o], part_idx[o], d0, i0, d1, i1);
        top2_insert(part_e[o + 1], part_idx[o + 1], d0, i0, d1, i1);
    }
    const float qn = Q.nrm[q];
    idx2[2 * (size_t)q] = i0 == 0x7fffffff ? -1 : i0; idx2[2 * (size_t)q + 1] = i1 == 0x7fffffff ? -1 : i1;
    dist2[2 * (size_t)q] = i0 == 0x7fffffff ? sqrtf(3.0e38f) : sqrtf(qn + d0);
    dist2[2 * (size_t)q + 1] = i1 == 0x7fffffff ? sqrtf(3.0e38f) : sqrtf(qn + d1);
}

// launches the sliced distance pass + merge for one directed pair; `part` holds 2 * slices * Q.n (int + float)
static void l2_knn2_launch(hipStream_t st, int num_cu, const L2Set& Q, const L2Set& T, void* part, int* idx2, float* dist2) {
    const int qblocks = (Q.n + 128 * L2_QW - 1) / (128 * L2_QW), ntiles = (T.n + L2_TB - 1) / L2_TB;
    // every slice restarts the search bound, so as few slices as fill the device (K8_WG_PER_CU workgroups per CU), at
    // least 16 tiles each; a power of two (<= 16) for the XCD mapping of the kernel
    int slices = 1;
    while (slices < L2_MAX_SLICES && qblocks * slices < K8_WG_PER_CU * num_cu && ntiles / (2 * slices) >= 16) slices *= 2;
    const int tiles_per_slice = ((ntiles + slices - 1) / slices + L2_SUB - 1) / L2_SUB * L2_SUB;   // whole steps
    const int nwg = slices >= 8 ? qblocks * slices : (qblocks + 8 / slices - 1) / (8 / slices) * 8;
    int* pi = (int*)part;
    float* pe = (float*)(pi + (size_t)2 * L2_MAX_SLICES * Q.n);
    hipLaunchKernelGGL(l2_knn2_mfma_kernel, dim3(nwg), dim3(256), 0, st, Q, T, tiles_per_slice, slices, qblocks, pi, pe);
    hipLaunchKernelGGL(l2_merge_kernel, dim3((Q.n + 255) / 256), dim3(256), 0, st, Q, slices, (const int*)pi, (const float*)pe, idx2, dist2);
}

// ---------------------------------------------------------------- ratio test + union ----------
// block-wide exclusive scan of a 0/1 flag (1024 threads); returns the offset, *total the sum
__device__ __forceinline__ int block_scan_flag(int flag, int* total) {
    __shared__ int wsum[16];
    __shared__ int wtot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long b = __ballot(flag);
    int within = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) { int v = wsum[k]; wsum[k] = s; s += v; }
        wtot = s;
    }
    __syncthreads();
    int off = wsum[wave] + within;
    *total = wtot;
    __syncthreads();
    return off;
}

// CpuMatcher::match: accepted 1->2 matches in query order, then 2->1 matches that are not already
// in the set; BestOf2NearestMatcher::match: point lists shifted by half the image size.
__global__ __launch_bounds__(1024) void ratio_union_kernel(const FeatDev* feats, const PairDesc* pairs, const int* idx2, const float* dist2, float ratio,
                                                           MisDMatch* matches, float* src_xy, float* dst_xy, int* n_matches) {
    const PairDesc pd = pairs[blockIdx.x];
    const FeatDev F1 = feats[pd.i], F2 = feats[pd.j];
    MisDMatch* m = matches + pd.m_off;
    float* sp = src_xy + 2 * pd.m_off;
    float* dp = dst_xy + 2 * pd.m_off;
    const int* i12 = idx2 + pd.knn_off12 * 2;
    const float* d12 = dist2 + pd.knn_off12 * 2;
    const int* i21 = idx2 + pd.knn_off21 * 2;
    const float* d21 = dist2 + pd.knn_off21 * 2;
    const float hw1 = (float)F1.w * 0.5f, hh1 = (float)F1.h * 0.5f, hw2 = (float)F2.w * 0.5f, hh2 = (float)F2.h * 0.5f;
    int base = 0;
    if (F2.n >= 2)
        for (int q0 = 0; q0 < F1.n; q0 += 1024) {
            int q = q0 + threadIdx.x, ok = 0;
            if (q < F1.n) ok = d12[2 * q] < ratio * d12[2 * q + 1];
            int tot, off = block_scan_flag(ok, &tot);
            if (ok) {
                int t = i12[2 * q];
                MisDMatch mm = {q, t, 0, d12[2 * q]};
                m[base + off] = mm;
                sp[2 * (base + off)] = F1.kps[q].x - hw1; sp[2 * (base + off) + 1] = F1.kps[q].y - hh1;
                dp[2 * (base + off)] = F2.kps[t].x - hw2; dp[2 * (base + off) + 1] = F2.kps[t].y - hh2;
            }
            base += tot;
        }
    if (F1.n >= 2)
        for (int q0 = 0; q0 < F2.n; q0 += 1024) {
            int q = q0 + threadIdx.x, ok = 0, t1 = 0;
            if (q < F2.n) {
                ok = d21[2 * q] < ratio * d21[2 * q + 1];
                if (ok) {
                    t1 = i21[2 * q];
                    bool acc12 = F2.n >= 2 && d12[2 * t1] < ratio * d12[2 * t1 + 1];
                    if (acc12 && i12[2 * t1] == q) ok = 0;  // (t1, q) already in the 1->2 set
                }
            }
            int tot, off = block_scan_flag(ok, &tot);
            if (ok) {
                MisDMatch mm = {t1, q, -1, d21[2 * q]};
                m[base + off] = mm;
                sp[2 * (base + off)] = F1.kps[t1].x - hw1; sp[2 * (base + off) + 1] = F1.kps[t1].y - hh1;
                dp[2 * (base + off)] = F2.kps[q].x - hw2; dp[2 * (base + off) + 1] = F2.kps[q].y - hh2;
            }
            base += tot;
        }
    if (threadIdx.x == 0) n_matches[blockIdx.x] = base;
}

// ---------------------------------------------------------------- K9: findHomography ----------
// The estimator itself lives in homography.hip (batched over all pairs).  BestOf2NearestMatcher::match
// calls it twice: on all matches (-> inlier mask, confidence) and again on the inliers only.
struct PairOut {
    int ran_ransac;   // matches >= num_matches_thresh1
    int passed;       // first H non-empty and |det| >= eps: num_inliers / confidence are meaningful
    int second;       // the inlier-only estimation ran
    int pad;
};

__host__ __device__ __forceinline__ double det3(const double* H) {
    return H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
}

__global__ void first_calls_kernel(const PairDesc* pairs, int np, const int* n_matches, const float* src_xy, const float* dst_xy, uint8_t* masks,
                                   int thresh1, HomoCall* calls, PairOut* outs) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= np) return;
    const PairDesc pd = pairs[k];
    const int nm = n_matches[k];
    HomoCall c;
    c.src = src_xy + 2 * pd.m_off; c.dst = dst_xy + 2 * pd.m_off; c.mask = masks + pd.m_off; c.pt_off = (long long)pd.m_off;
    c.n = nm; c.active = nm >= thresh1;
    calls[k] = c;
    outs[k].ran_ransac = c.active; outs[k].passed = 0; outs[k].second = 0;
}

// between the two estimations: determinant check, inlier threshold; the inlier-only point lists are
// the compacted lists the first estimation left in its scratch
// `want` selects the problems whose first estimation finished in that RANSAC phase (fin1[k]); the others get an
// inactive call and their PairOut entry is left alone (the launch for the other phase owns it)

// The match lists of a call, packed pair after pair (pair k at the sum of the counts before it) straight into the pinned host
// staging, with the counts: 16 bytes per match that exists instead of a device-to-host copy of the capacity-sized arrays (15 MB
// for 16 x 4000 features, 5 % of it used).  That copy ran as a blit kernel on the compute units, under the first draw_kernel --
// whose 1024-thread workgroups need a compute unit each to themselves: 120 us of work took 360 us.
__global__ __launch_bounds__(256) void pack_lists_kernel(const PairDesc* pairs, int np, const int* nm, const MisDMatch* matches, MisDMatch* host_m, int* host_nm) {
    __shared__ int red[256];
    const int k = blockIdx.x, t = threadIdx.x;
    int part = 0;
    for (int q = t; q < k; q += 256) part += nm[q];
    red[t] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const int off = red[0], cnt = nm[k];
    if (t == 0) host_nm[k] = cnt;
    const uint4* s = reinterpret_cast<const uint4*>(matches + pairs[k].m_off);
    uint4* d = reinterpret_cast<uint4*>(host_m + off);
    static_assert(sizeof(MisDMatch) == 16, "one 16-byte store per match");
    for (int i = t; i < cnt; i += 256) d[i] = s[i];
}

// The small per-pair results of a matcher call written straight into the pinned (device-visible) host staging: one launch
// instead of six device-to-host copies of a few kilobytes each (~15 us apiece on the tail of the call).
struct CopySegs { const uint8_t* src[8]; uint8_t* dst[8]; unsigned bytes[8]; int n; };
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs c) {
    for (int k = 0; k < c.n; k++) {
        const uint8_t* s = c.src[k];
        uint8_t* d = c.dst[k];
        const unsigned nb = c.bytes[k];
        if ((((uintptr_t)s | (uintptr_t)d) & 3) == 0) {
            const unsigned nw = nb >> 2;
            for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < nw; i += gridDim.x * 256) reinterpret_cast<unsigned*>(d)[i] = reinterpret_cast<const unsigned*>(s)[i];
            for (unsigned i = (nw << 2) + blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) d[i] = s[i];
        } else {
            for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) d[i] = s[i];
        }
    }
}

__global__ void second_calls_kernel(int np, const HomoCall* calls1, const HomoResult* res1, const float* scr1, const int* fin1, int want, int thresh2,
                                    HomoCall* calls2, PairOut* outs, int check_det) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= np) return;
    const HomoCall c1 = calls1[k];
    HomoCall c;
    c.src = nullptr; c.dst = nullptr; c.mask = nullptr; c.pt_off = c1.pt_off; c.n = 0; c.active = 0;
    // check_det == 0: the refined first H is still being computed on another stream; the host applies the reference's
    // |det H| >= eps test when it assembles the results and drops a second estimation that was run in vain
    if (fin1[k] == want && c1.active && res1[k].ok && (!check_det || !(fabs(det3(res1[k].H)) < DBL_EPSILON))) {
        outs[k].passed = 1;
        const int ninl = res1[k].ninl;
        if (ninl >= thresh2) {
            c.src = scr1 + 4 * c1.pt_off; c.dst = c.src + 2 * (size_t)c1.n; c.n = ninl; c.active = 1;
            outs[k].second = 1;
        }
    }
    calls2[k] = c;
}

void invert3(const double* H, double* I) {
    double d = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
    if (d != 0.) d = 1. / d;
    I[0] = (H[4] * H[8] - H[5] * H[7]) * d; I[1] = (H[2] * H[7] - H[1] * H[8]) * d; I[2] = (H[1] * H[5] - H[2] * H[4]) * d;
    I[3] = (H[5] * H[6] - H[3] * H[8]) * d; I[4] = (H[0] * H[8] - H[2] * H[6]) * d; I[5] = (H[2] * H[3] - H[0] * H[5]) * d;
    I[6] = (H[3] * H[7] - H[4] * H[6]) * d; I[7] = (H[1] * H[6] - H[0] * H[7]) * d; I[8] = (H[0] * H[4] - H[1] * H[3]) * d;
}

// grow-only device / pinned-host arenas kept in the context: a match call allocates nothing in steady state
struct Arena {
    void* p = nullptr;
    size_t bytes = 0;
    bool host = false;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) { hipError_t e = host ? hipHostFree(p) : hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        need = need + need / 4 + 4096;
        hipError_t e = host ? hipHostMalloc(&p, need, hipHostMallocDefault) : hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() { if (p) { if (host) hipHostFree(p); else hipFree(p); } p = nullptr; bytes = 0; }
};

struct MatchWorkspace : MisWorkspace {
    Arena dev, pinned, l2;
    // b1: first estimation of every pair; b2 / b3: the inlier-only estimation of the pairs whose first one finished in
    // RANSAC phase 0 / phase 1.  b2 runs on `side` concurrently with phase 1 of b1 (the chains are latency bound).
    HomoBatch b1, b2, b3;
    hipStream_t side = nullptr, third = nullptr;
    hipEvent_t ev_phase0 = nullptr, ev_side_done = nullptr, ev_phase1 = nullptr, ev_third_done = nullptr, ev_matches = nullptr;
    // "the 2-NN pass of matcher call number knn_seq has been enqueued, ev_knn marks its end" (mis_match_knn_fence)
    hipEvent_t ev_knn = nullptr;
    hipEvent_t ev_draw1 = nullptr, ev_side_hyp0 = nullptr, ev_b2_replay = nullptr;
    hipEvent_t tev[8] = {nullptr};   // MIS_MATCH_TRACE: timing events (2-NN end, phase 0 end, main chain's second RANSAC phase end, main chain end, side end, third end, tails of phase 0 end, side chain's first RANSAC phase end)
    hipEvent_t ev_gate = nullptr;    // what mis_match_knn_fence queues a stream behind: ev_knn, or the end of the first RANSAC phase (MIS_COMPOSE_GATE)
    std::atomic<long long> seq{0}, knn_seq{0};
    hipEvent_t ev_lists = nullptr;                       // the early download of the match lists has landed
    void (*enqueued_cb)(void*) = nullptr;                // mis_match_on_enqueued: one-shot hook of the next call
    void* enqueued_user = nullptr;
    MatchWorkspace() { pinned.host = true; }
    ~MatchWorkspace() override {
        dev.release(); pinned.release(); l2.release();
        homo_batch_release(&b1); homo_batch_release(&b2); homo_batch_release(&b3);
        // side / third are the context's auxiliary streams: not owned here
        if (ev_phase1) hipEventDestroy(ev_phase1);
        if (ev_third_done) hipEventDestroy(ev_third_done);
        if (ev_matches) hipEventDestroy(ev_matches);
        if (ev_lists) hipEventDestroy(ev_lists);
        if (ev_phase0) hipEventDestroy(ev_phase0);
        if (ev_side_done) hipEventDestroy(ev_side_done);
        if (ev_knn) hipEventDestroy(ev_knn);
        if (ev_draw1) hipEventDestroy(ev_draw1);
        if (ev_side_hyp0) hipEventDestroy(ev_side_hyp0);
        if (ev_b2_replay) hipEventDestroy(ev_b2_replay);
    }
};

MatchWorkspace* workspace(MisContext* ctx) {
    if (!ctx->match_ws) ctx->match_ws = new MatchWorkspace();
    return static_cast<MatchWorkspace*>(ctx->match_ws);
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += mis_align_up(bytes ? bytes : 1, 256); return o; }
};

void init_info(MisMatchesInfo* m) {
    memset(m, 0, sizeof(*m));
    m->src_img_idx = -1; m->dst_img_idx = -1;
}

int match_impl(MisContext* ctx, const MisFeatures* feats, int n, const MisMatchParams* p, int rank, int world, MisMatchesInfo* out) {
    MIS_CHECK(ctx, feats && p && out && n >= 1, MIS_E_INVALID, "null argument");
    MIS_CHECK(ctx, world >= 1 && rank >= 0 && rank < world, MIS_E_INVALID, "bad rank / world size");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    // whatever happens below, a thread waiting in mis_match_knn_fence for this call is released when it returns
    const auto t_begin = std::chrono::steady_clock::now();
    struct SeqGuard {
        MatchWorkspace* ws; long long seq;
        ~SeqGuard() { if (ws->knn_seq.load() < seq) ws->knn_seq.store(seq); }
    } seq_guard{workspace(ctx), 0};
    seq_guard.seq = ++seq_guard.ws->seq;
    for (int i = 0; i < n * n; i++) init_info(&out[i]);
    // FeaturesMatcher::operator(): all i < j with non-empty keypoint lists, dealt round-robin over ranks
    std::vector<PairDesc> pairs;
    std::vector<FeatDev> fd(n);
    size_t knn_total = 0, m_total = 0;
    bool use_l2 = false, use_bin = false;
    for (int i = 0; i < n; i++) {
        const bool bin_i = feats[i].desc_dtype == MIS_U8 && feats[i].desc_cols == 32;
        const bool l2_i = feats[i].desc_dtype == MIS_F32 && feats[i].desc_cols >= 1 && feats[i].desc_cols <= 128;
        MIS_CHECK(ctx, feats[i].n == 0 || bin_i || l2_i, MIS_E_UNSUPPORTED,
                  "all-pairs matching supports 32-byte binary descriptors (Hamming) or f32 descriptors of <= 128 columns (L2)");
        MIS_CHECK(ctx, feats[i].n < (1 << 22), MIS_E_UNSUPPORTED, "more than 4 M keypoints in one image");   // index bits of the 2-NN keys
        if (feats[i].n > 0) { if (l2_i) use_l2 = true; else use_bin = true; }
        fd[i] = FeatDev{(const uint8_t*)feats[i].descriptors, feats[i].keypoints, feats[i].n, feats[i].img_w, feats[i].img_h};
    }
    int pair_index = 0;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) {
            if (feats[i].n <= 0 || feats[j].n <= 0) continue;
            if ((pair_index++ % world) != rank) continue;
            PairDesc pd;
            pd.i = i; pd.j = j;
            pd.knn_off12 = knn_total; knn_total += feats[i].n;
            pd.knn_off21 = knn_total; knn_total += feats[j].n;
            pd.m_off = m_total; pd.cap = feats[i].n + feats[j].n; m_total += pd.cap;
            pairs.push_back(pd);
        }
    MIS_CHECK(ctx, !(use_l2 && use_bin), MIS_E_INVALID, "binary and float descriptors cannot be mixed in one matcher call");
    const int np = (int)pairs.size();
    if (np == 0) return MIS_OK;
    int maxq = 0;
    for (int i = 0; i < n; i++) maxq = std::max(maxq, feats[i].n);
    // workgroup table of the Hamming pass on the matrix cores (HmJob): eight lists by train frame mod 8, interleaved
    std::vector<HmJob> hm_jobs;
    if (!use_l2 && maxq <= HM_MAX_TRAINS) {
        std::vector<HmJob> lists[8];
        for (int t = 0; t < n; t++)
            for (int k = 0; k < np; k++) {
                const int dir = pairs[k].j == t ? 0 : (pairs[k].i == t ? 1 : -1);      // direction 0: queries of i against the trains of j
                if (dir < 0) continue;
                const int nqf = feats[dir == 0 ? pairs[k].i : pairs[k].j].n;
                for (int q0 = 0; q0 < nqf; q0 += 256) lists[t & 7].push_back(HmJob{k, dir, q0, 0});
            }
        size_t longest = 0;
        for (auto& l : lists) longest = std::max(longest, l.size());
        hm_jobs.assign(longest * 8, HmJob{-1, 0, 0, 0});
        for (int c = 0; c < 8; c++)
            for (size_t sl = 0; sl < lists[c].size(); sl++) hm_jobs[sl * 8 + c] = lists[c][sl];
    }
    MatchWorkspace* ws = workspace(ctx);
    hipStream_t st = ctx->stream;
    MIS_HIP(ctx, hipStreamSynchronize(st));  // the arenas may still be read by a previous call's copies
    Carver dc;
    const size_t o_feats = dc.take(sizeof(FeatDev) * n), o_pairs = dc.take(sizeof(PairDesc) * np), o_idx = dc.take(sizeof(int) * 2 * knn_total),
                 o_dist = dc.take(sizeof(int) * 2 * knn_total), o_matches = dc.take(sizeof(MisDMatch) * m_total), o_src = dc.take(sizeof(float) * 2 * m_total),
                 o_dst = dc.take(sizeof(float) * 2 * m_total), o_nm = dc.take(sizeof(int) * np), o_mask = dc.take(m_total), o_out = dc.take(sizeof(PairOut) * np);
    MIS_HIP(ctx, ws->dev.reserve(dc.off));
    uint8_t* D = (uint8_t*)ws->dev.p;
    FeatDev* d_feats = (FeatDev*)(D + o_feats); PairDesc* d_pairs = (PairDesc*)(D + o_pairs);
    int* d_idx = (int*)(D + o_idx); float* d_dist = (float*)(D + o_dist);
    MisDMatch* d_matches = (MisDMatch*)(D + o_matches); float* d_src = (float*)(D + o_src); float* d_dst = (float*)(D + o_dst);
    int* d_nm = (int*)(D + o_nm); uint8_t* d_mask = D + o_mask; PairOut* d_out = (PairOut*)(D + o_out);
    int rc;
    if ((rc = homo_batch_reserve(ctx, &ws->b1, np, (long long)m_total, p->max_iters)) != MIS_OK) return rc;
    if ((rc = homo_batch_reserve(ctx, &ws->b2, np, (long long)m_total, p->max_iters)) != MIS_OK) return rc;
    if ((rc = homo_batch_reserve(ctx, &ws->b3, np, (long long)m_total, p->max_iters)) != MIS_OK) return rc;
    // pinned host mirror of everything that comes back
    Carver hc;
    const size_t h_in = hc.take(sizeof(FeatDev) * n + sizeof(PairDesc) * np + sizeof(HmFrame) * n + 1024), h_nm = hc.take(sizeof(int) * np), h_out = hc.take(sizeof(PairOut) * np),
                 h_r1 = hc.take(sizeof(HomoResult) * np), h_r2 = hc.take(sizeof(HomoResult) * np), h_r3 = hc.take(sizeof(HomoResult) * np), h_fin = hc.take(sizeof(int) * np), h_m = hc.take(sizeof(MisDMatch) * m_total), h_mask = hc.take(m_total), h_bad = hc.take(256), h_jobs = hc.take(sizeof(HmJob) * std::max<size_t>(hm_jobs.size(), 1));
    MIS_HIP(ctx, ws->pinned.reserve(hc.off));
    uint8_t* Hh = (uint8_t*)ws->pinned.p;
    memcpy(Hh + h_in, fd.data(), sizeof(FeatDev) * n);
    PairDesc* h_pairs = (PairDesc*)(Hh + h_in + mis_align_up(sizeof(FeatDev) * n, 256));
    memcpy(h_pairs, pairs.data(), sizeof(PairDesc) * np);
    MIS_HIP(ctx, hipMemcpyAsync(d_feats, Hh + h_in, sizeof(FeatDev) * n, hipMemcpyHostToDevice, st));
    MIS_HIP(ctx, hipMemcpyAsync(d_pairs, h_pairs, sizeof(PairDesc) * np, hipMemcpyHostToDevice, st));
    // flag of the L2 path (non-integer descriptors), copied into the pinned arena: an early return never leaves a copy aimed at this frame
    volatile int& l2_bad = *reinterpret_cast<volatile int*>(Hh + h_bad);
    l2_bad = 0;
    if (!use_l2 && maxq > HM_MAX_TRAINS) {
        // train sets beyond the key's 13 index bits: the vector-pipe kernel (also what mis_knn2 runs)
        hipLaunchKernelGGL(knn2_hamming_kernel, dim3((maxq + 255) / 256, 2 * np), dim3(256), 0, st, (const FeatDev*)d_feats, (const PairDesc*)d_pairs, d_idx, d_dist);
    } else if (!use_l2 && !hm_jobs.empty()) {
        // every frame's descriptors once as fp4 nibbles (train form and query form), then all directed pairs in one MFMA launch
        std::vector<HmFrame> hf(n);
        Carver lc;
        const size_t o_fr = lc.take(sizeof(HmFrame) * n), o_jobs = lc.take(sizeof(HmJob) * hm_jobs.size());
        for (int i = 0; i < n; i++) {
            const size_t rows = (size_t)(std::max(feats[i].n, 1) + HM_ROWPAD - 1) / HM_ROWPAD * HM_ROWPAD;
            hf[i].train_off = lc.take(rows * 128); hf[i].query_off = lc.take(rows * 128);
        }
        MIS_HIP(ctx, ws->l2.reserve(lc.off));
        uint8_t* L = (uint8_t*)ws->l2.p;
        HmFrame* h_fr = (HmFrame*)(Hh + h_in + mis_align_up(sizeof(FeatDev) * n, 256) + mis_align_up(sizeof(PairDesc) * np, 256));
        memcpy(h_fr, hf.data(), sizeof(HmFrame) * n);
        MIS_HIP(ctx, hipMemcpyAsync(L + o_fr, h_fr, sizeof(HmFrame) * n, hipMemcpyHostToDevice, st));
        const int maxpad = (maxq + HM_ROWPAD - 1) / HM_ROWPAD * HM_ROWPAD;
        hipLaunchKernelGGL(hamming_expand4_kernel, dim3(maxpad * 8 / 256, n), dim3(256), 0, st, (const FeatDev*)d_feats, (const HmFrame*)(L + o_fr), (int8_t*)L);
        memcpy(Hh + h_jobs, hm_jobs.data(), sizeof(HmJob) * hm_jobs.size());
        MIS_HIP(ctx, hipMemcpyAsync(L + o_jobs, Hh + h_jobs, sizeof(HmJob) * hm_jobs.size(), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(knn2_hamming_fp4_kernel, dim3((unsigned)hm_jobs.size()), dim3(256), 0, st, (const FeatDev*)d_feats, (const PairDesc*)d_pairs,
                           (const HmFrame*)(L + o_fr), (const int8_t*)L, d_idx, d_dist, (const HmJob*)(L + o_jobs));
    } else if (!use_l2) {
        // no query anywhere (every frame without features): nothing to search, the lists stay empty
    } else {
        // fp16 copies + squared norms of every image once, then one MFMA distance pass per directed pair
        std::vector<size_t> hoff(n), noff(n);
        Carver lc;
        for (int i = 0; i < n; i++) { hoff[i] = lc.take((size_t)std::max(feats[i].n, 1) * 256); noff[i] = lc.take((size_t)std::max(feats[i].n, 1) * 4); }
        const size_t o_bad = lc.take(4);
        const size_t o_part = lc.take((size_t)16 * L2_MAX_SLICES * std::max(maxq, 1));   // slice results of one directed pair (stream ordered reuse)
        MIS_HIP(ctx, ws->l2.reserve(lc.off));
        uint8_t* L = (uint8_t*)ws->l2.p;
        MIS_HIP(ctx, hipMemsetAsync(L + o_bad, 0, 4, st));
        for (int i = 0; i < n; i++)
            if (feats[i].n > 0)
                hipLaunchKernelGGL(l2_prep_kernel, dim3((feats[i].n + 3) / 4), dim3(256), 0, st, (const float*)feats[i].descriptors, feats[i].n,
                                   feats[i].desc_cols, (_Float16*)(L + hoff[i]), (float*)(L + noff[i]), (int*)(L + o_bad));
        for (const PairDesc& pd : pairs) {
            L2Set A{(const _Float16*)(L + hoff[pd.i]), (const float*)(L + noff[pd.i]), feats[pd.i].n}, B{(const _Float16*)(L + hoff[pd.j]), (const float*)(L + noff[pd.j]), feats[pd.j].n};
            l2_knn2_launch(st, ctx->num_cu, A, B, L + o_part, d_idx + 2 * pd.knn_off12, d_dist + 2 * pd.knn_off12);
            l2_knn2_launch(st, ctx->num_cu, B, A, L + o_part, d_idx + 2 * pd.knn_off21, d_dist + 2 * pd.knn_off21);
        }
        MIS_HIP(ctx, hipMemcpyAsync(Hh + h_bad, L + o_bad, 4, hipMemcpyDeviceToHost, st));
    }
    // the 2-NN pass fills the device on its own; what follows are latency-bound chains.  Work that wants to share the
    // device with the matcher (the job's speculative composition) can queue behind this event: mis_match_knn_fence
    if (!ws->ev_knn) MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_knn, hipEventDisableTiming));
    MIS_HIP(ctx, hipEventRecord(ws->ev_knn, st));
    ws->ev_gate = ws->ev_knn;
    hipLaunchKernelGGL(ratio_union_kernel, dim3(np), dim3(1024), 0, st, (const FeatDev*)d_feats, (const PairDesc*)d_pairs, (const int*)d_idx,
                       (const float*)d_dist, 1.f - p->match_conf, d_matches, d_src, d_dst, d_nm);
    hipLaunchKernelGGL(first_calls_kernel, dim3((np + 127) / 128), dim3(128), 0, st, (const PairDesc*)d_pairs, np, (const int*)d_nm, (const float*)d_src,
                       (const float*)d_dst, d_mask, p->num_matches_thresh1, ws->b1.calls, d_out);
    if (!ws->side) {
        if ((rc = mis_aux_stream(ctx, 0, &ws->side)) != MIS_OK) return rc;
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_phase0, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_side_done, hipEventDisableTiming));
    }
    const double rt = p->ransac_thresh, cf = p->confidence;
    bool early_lists = false, packed_lists = false;     // packed: pair k's matches sit at the sum of the counts before it (pack_lists_kernel)
    // everything between the forks to the auxiliary streams and their joins runs inside one scope: an error in there must not
    // leave those streams with work pending (they are the context's, shared with the feature finders) or a copy in flight
    // MIS_COMPOSE_GATE: 0 = a stream fenced by mis_match_knn_fence (the job's speculative composition) starts behind the 2-NN pass,
    // 1 = behind the first RANSAC phase of the first estimation (draw, 4-point solves, replay, masks: 0.7 ms of large workgroups
    // that wait for room once the composition's grids fill the device)
    static const int compose_gate = getenv("MIS_COMPOSE_GATE") ? atoi(getenv("MIS_COMPOSE_GATE")) : 4;
    static const bool trace_ev = getenv("MIS_MATCH_TRACE") != nullptr;
    auto mark = [&](int i, hipStream_t s_) {      // diagnostics: device time stamps of the chains (printed with the host's when MIS_MATCH_TRACE is set)
        if (!trace_ev) return;
        if (!ws->tev[i]) hipEventCreate(&ws->tev[i]);
        hipEventRecord(ws->tev[i], s_);
    };
    mark(0, st);
    auto enqueue_chains = [&]() -> int {
    static const int chains = getenv("MIS_MATCH_CHAINS") ? atoi(getenv("MIS_MATCH_CHAINS")) : 3;   // 2: the two-chain flow below
    if (chains != 3) {
    // first estimation, phase 0 up to the replay's verdict (pairs with a clear overlap finish here)
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 3, st)) != MIS_OK) return rc;
    MIS_HIP(ctx, hipEventRecord(ws->ev_phase0, st));
    if (compose_gate == 1) ws->ev_gate = ws->ev_phase0;
    // side stream: the tails of those pairs (mask, DLT on the inliers, LM: ~2 ms of latency) and their inlier-only estimation ...
    MIS_HIP(ctx, hipStreamWaitEvent(ws->side, ws->ev_phase0, 0));
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 4, ws->side)) != MIS_OK) return rc;
    hipLaunchKernelGGL(second_calls_kernel, dim3((np + 127) / 128), dim3(128), 0, ws->side, np, (const HomoCall*)ws->b1.calls, (const HomoResult*)ws->b1.results,
                       (const float*)ws->b1.scr, (const int*)ws->b1.fin, 0, p->num_matches_thresh2, ws->b2.calls, d_out, 1);
    if ((rc = homo_batch_run(ctx, &ws->b2, rt, p->max_iters, cf, 2, ws->side)) != MIS_OK) return rc;
    MIS_HIP(ctx, hipEventRecord(ws->ev_side_done, ws->side));
    // ... while the main stream finishes the first estimation of the others (which only needs the verdict) and runs their second one
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 1, st)) != MIS_OK) return rc;
    hipLaunchKernelGGL(second_calls_kernel, dim3((np + 127) / 128), dim3(128), 0, st, np, (const HomoCall*)ws->b1.calls, (const HomoResult*)ws->b1.results,
                       (const float*)ws->b1.scr, (const int*)ws->b1.fin, 1, p->num_matches_thresh2, ws->b3.calls, d_out, 1);
    if ((rc = homo_batch_run(ctx, &ws->b3, rt, p->max_iters, cf, 2, st)) != MIS_OK) return rc;
    } else {
    // Three chains (the default; MIS_MATCH_CHAINS=2 selects the flow above): findHomography returns the RANSAC mask, not one
    // recomputed after its refinement, so the second estimation starts from the mask while the DLT + LM refinement of the first H
    // runs on a third stream; the |det H| test of the reference moves to the host assembly below.  Side and third are the
    // context's two auxiliary streams -- the streams the ORB batch's helper lanes ran on a moment ago -- so that the job keeps to
    // four streams (an earlier version created two more here, one in a priority class of its own to dodge a shared hardware
    // queue: the step then moved by 25 % with GPU_MAX_HW_QUEUES).
    if (!ws->third) {
        if ((rc = mis_aux_stream(ctx, 1, &ws->third)) != MIS_OK) return rc;
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_phase1, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_third_done, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_matches, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_lists, hipEventDisableTiming));
    }
    // the match lists are final once the ratio test has run: their download (megabytes) goes to the third stream now, under the
    // RANSAC chains, instead of behind them (0.3 ms at the end of the call)
    MIS_HIP(ctx, hipEventRecord(ws->ev_matches, st));
    MIS_HIP(ctx, hipStreamWaitEvent(ws->third, ws->ev_matches, 0));
    hipLaunchKernelGGL(pack_lists_kernel, dim3(np), dim3(256), 0, ws->third, (const PairDesc*)d_pairs, np, (const int*)d_nm, (const MisDMatch*)d_matches,
                       (MisDMatch*)(Hh + h_m), (int*)(Hh + h_nm));
    MIS_HIP(ctx, hipEventRecord(ws->ev_lists, ws->third));
    early_lists = true; packed_lists = true;
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 3, st)) != MIS_OK) return rc;
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 10, st)) != MIS_OK) return rc;
    MIS_HIP(ctx, hipEventRecord(ws->ev_phase0, st));
    mark(1, st);
    if (compose_gate == 1) ws->ev_gate = ws->ev_phase0;
    MIS_HIP(ctx, hipStreamWaitEvent(ws->side, ws->ev_phase0, 0));
    hipLaunchKernelGGL(second_calls_kernel, dim3((np + 127) / 128), dim3(128), 0, ws->side, np, (const HomoCall*)ws->b1.calls, (const HomoResult*)ws->b1.results,
                       (const float*)ws->b1.scr, (const int*)ws->b1.fin, 0, p->num_matches_thresh2, ws->b2.calls, d_out, 0);
    // MIS_HYP_ORDER=1 (experiment): the main chain's second-phase solves wait for the side chain's first solves.  Measured: the side
    // chain ends where it did (+ 2.65 ms behind the 2-NN pass: its solves are not slowed by the others'), the main chain 0.55 ms later.
    static const int hyp_order = getenv("MIS_HYP_ORDER") ? atoi(getenv("MIS_HYP_ORDER")) : 0;
    // MIS_B2_SPLIT (experiment, default 0): the side chain's second phase -- a handful of problems with few points that run all 2000
    // iterations, 0.5 ms -- taken off it (the replay alone first) and run behind the main chain (1) or behind the first estimation's
    // tails on the third stream (2).  Measured: the side chain then ends at + 2.27 ms instead of + 2.66, and the chain that took the
    // second phase at + 2.65: the matcher ends where it did.
    static const int b2_split = getenv("MIS_B2_SPLIT") ? atoi(getenv("MIS_B2_SPLIT")) : 0;
    if (!ws->ev_draw1) MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_draw1, hipEventDisableTiming));
    if (!ws->ev_side_hyp0) MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_side_hyp0, hipEventDisableTiming));
    {
        HomoSync sy;
        if (compose_gate == 4) { sy.rec = ws->ev_draw1; sy.rec_pos = 2; ws->ev_gate = ws->ev_draw1; }      // gate 4 (the default): behind the side chain's first draw, 0.05 - 0.1 ms behind the first phase -- the tails and that draw hold their compute units by then
        if (hyp_order) sy.rec_hyp0 = ws->ev_side_hyp0;
        if (b2_split) {
            // the side chain was the longest (+ 2.66 ms behind the 2-NN pass): its second phase -- a handful of problems with few
            // points that run all 2000 iterations -- waited behind the 1.3 ms tails of the finishers although it only needs the
            // replay's verdict.  The replay alone first; the second phase goes behind the main chain, which ends first
            if ((rc = homo_batch_run(ctx, &ws->b2, rt, p->max_iters, cf, 3, ws->side, &sy)) != MIS_OK) return rc;
            if (!ws->ev_b2_replay) MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_b2_replay, hipEventDisableTiming));
            MIS_HIP(ctx, hipEventRecord(ws->ev_b2_replay, ws->side));
            if ((rc = homo_batch_run(ctx, &ws->b2, rt, p->max_iters, cf, 4, ws->side)) != MIS_OK) return rc;
        } else
        if ((rc = homo_batch_run(ctx, &ws->b2, rt, p->max_iters, cf, 2, ws->side, &sy)) != MIS_OK) return rc;
    }
    MIS_HIP(ctx, hipEventRecord(ws->ev_side_done, ws->side));
    mark(4, ws->side);
    MIS_HIP(ctx, hipStreamWaitEvent(ws->third, ws->ev_phase0, 0));
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 11, ws->third)) != MIS_OK) return rc;
    mark(6, ws->third);
    if (b2_split == 2) {      // (experiment: behind the first estimation's tails on the third stream -- that chain then ends at + 2.65 ms)
        MIS_HIP(ctx, hipStreamWaitEvent(ws->third, ws->ev_b2_replay, 0));
        if ((rc = homo_batch_run(ctx, &ws->b2, rt, p->max_iters, cf, 1, ws->third)) != MIS_OK) return rc;
    }
    {
        // gate 2: behind the second phase's draw of the main chain -- by then the tails (third stream) and the second estimations
        // (side stream), released together with it, hold their compute units
        HomoSync sy;
        if (compose_gate == 2 || compose_gate == 3) { sy.rec = ws->ev_draw1; sy.rec_pos = compose_gate - 2; ws->ev_gate = ws->ev_draw1; }
        if (hyp_order) sy.wait_hyp1 = ws->ev_side_hyp0;
        if (trace_ev && !sy.rec) {      // diagnostics: the end of the main chain's second draw
            if (!ws->tev[7]) hipEventCreate(&ws->tev[7]);
            sy.rec = ws->tev[7]; sy.rec_pos = 0;
        }
        if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 6, st, &sy)) != MIS_OK) return rc;
    }
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 12, st)) != MIS_OK) return rc;
    MIS_HIP(ctx, hipEventRecord(ws->ev_phase1, st));
    mark(2, st);
    // the refinement of the phase-1 finishers' first H stays on this stream (1.7 ms of latency-bound work: behind the 2 ms
    // refinement of the phase-0 finishers on the third stream it ended the matcher 0.6 ms later); their inlier-only second
    // estimation goes to the third stream instead
    if ((rc = homo_batch_run(ctx, &ws->b1, rt, p->max_iters, cf, 13, st)) != MIS_OK) return rc;
    if (b2_split == 1) {      // the side chain's second phase, behind the main chain (which ends first: + 2.17 ms)
        MIS_HIP(ctx, hipStreamWaitEvent(st, ws->ev_b2_replay, 0));
        if ((rc = homo_batch_run(ctx, &ws->b2, rt, p->max_iters, cf, 1, st)) != MIS_OK) return rc;
    }
    mark(3, st);
    MIS_HIP(ctx, hipStreamWaitEvent(ws->third, ws->ev_phase1, 0));
    hipLaunchKernelGGL(second_calls_kernel, dim3((np + 127) / 128), dim3(128), 0, ws->third, np, (const HomoCall*)ws->b1.calls, (const HomoResult*)ws->b1.results,
                       (const float*)ws->b1.scr, (const int*)ws->b1.fin, 1, p->num_matches_thresh2, ws->b3.calls, d_out, 0);
    if ((rc = homo_batch_run(ctx, &ws->b3, rt, p->max_iters, cf, 2, ws->third)) != MIS_OK) return rc;
    MIS_HIP(ctx, hipEventRecord(ws->ev_third_done, ws->third));
    mark(5, ws->third);
    MIS_HIP(ctx, hipStreamWaitEvent(st, ws->ev_third_done, 0));
    }
    MIS_HIP(ctx, hipStreamWaitEvent(st, ws->ev_side_done, 0));
    return MIS_OK;
    };
    if ((rc = enqueue_chains()) != MIS_OK) {
        if (ws->side) hipStreamSynchronize(ws->side);
        if (ws->third) hipStreamSynchronize(ws->third);
        hipStreamSynchronize(st);
        return rc;
    }
    MIS_HIP(ctx, hipGetLastError());
    int* nm = (int*)(Hh + h_nm);
    PairOut* po = (PairOut*)(Hh + h_out);
    HomoResult* r1 = (HomoResult*)(Hh + h_r1);
    HomoResult* r2 = (HomoResult*)(Hh + h_r2);
    HomoResult* r3 = (HomoResult*)(Hh + h_r3);
    int* fin = (int*)(Hh + h_fin);
    MisDMatch* hm = (MisDMatch*)(Hh + h_m);
    uint8_t* hmask = Hh + h_mask;
    if (!early_lists) {
        MIS_HIP(ctx, hipMemcpyAsync(nm, d_nm, sizeof(int) * np, hipMemcpyDeviceToHost, st));
        MIS_HIP(ctx, hipMemcpyAsync(hm, d_matches, sizeof(MisDMatch) * m_total, hipMemcpyDeviceToHost, st));
    }
    {
        CopySegs cs;
        const void* srcs[6] = {d_out, ws->b1.results, ws->b2.results, ws->b3.results, ws->b1.fin, d_mask};
        void* dsts[6] = {po, r1, r2, r3, fin, hmask};
        const size_t sizes[6] = {sizeof(PairOut) * np, sizeof(HomoResult) * np, sizeof(HomoResult) * np, sizeof(HomoResult) * np, sizeof(int) * np, m_total};
        cs.n = 6;
        for (int k = 0; k < 6; k++) { cs.src[k] = (const uint8_t*)srcs[k]; cs.dst[k] = (uint8_t*)dsts[k]; cs.bytes[k] = (unsigned)sizes[k]; }
        for (int k = 6; k < 8; k++) { cs.src[k] = nullptr; cs.dst[k] = nullptr; cs.bytes[k] = 0; }
        hipLaunchKernelGGL(copy_segments_kernel, dim3(64), dim3(256), 0, st, cs);
    }
    // everything of this call is enqueued: a thread waiting in mis_match_knn_fence may start launching now without
    // competing with this one for the runtime's launch path
    ws->knn_seq.store(seq_guard.seq);
    // one-shot hook (mis_match_on_enqueued): the caller's own enqueue work -- e.g. the job's speculative composition on another
    // stream -- runs here on this thread, which would otherwise only wait for the device; no second host thread, no hand-over
    if (ws->enqueued_cb) {
        void (*cb)(void*) = ws->enqueued_cb;
        void* user = ws->enqueued_user;
        ws->enqueued_cb = nullptr; ws->enqueued_user = nullptr;
        cb(user);
        MIS_HIP(ctx, hipSetDevice(ctx->device));
    }
    const bool trace = getenv("MIS_MATCH_TRACE") != nullptr;
    const auto tq = std::chrono::steady_clock::now();
    // MatchesInfo (host), part 1 under the RANSAC chains: the match lists and their mirrors
    auto lists = [&]() {
        size_t packed_off = 0;
        for (int k = 0; k < np; k++) {
            const PairDesc& pd = pairs[k];
            MisMatchesInfo* a = &out[pd.i * n + pd.j];
            MisMatchesInfo* b = &out[pd.j * n + pd.i];
            a->src_img_idx = pd.i; a->dst_img_idx = pd.j;
            a->n_matches = nm[k];
            a->matches = (MisDMatch*)malloc(sizeof(MisDMatch) * (size_t)(nm[k] + 1));
            memcpy(a->matches, hm + (packed_lists ? packed_off : pd.m_off), sizeof(MisDMatch) * (size_t)nm[k]);
            packed_off += (size_t)nm[k];
            b->matches = (MisDMatch*)malloc(sizeof(MisDMatch) * (size_t)(nm[k] + 1));
            for (int q = 0; q < nm[k]; q++) {
                b->matches[q] = a->matches[q];
                b->matches[q].query_idx = a->matches[q].train_idx;
                b->matches[q].train_idx = a->matches[q].query_idx;
            }
        }
    };
    // an error from here on must not leave half-built entries behind: the lists are released and `out` is back to its zeroed state
    auto drop_lists = [&]() {
        for (int k = 0; k < np; k++) {
            const PairDesc& pd = pairs[k];
            MisMatchesInfo* e[2] = {&out[pd.i * n + pd.j], &out[pd.j * n + pd.i]};
            for (MisMatchesInfo* m : e) { free(m->matches); free(m->inliers_mask); init_info(m); }
        }
    };
    bool have_lists = false;
    if (early_lists) {
        MIS_HIP(ctx, hipEventSynchronize(ws->ev_lists));
        lists();
        have_lists = true;
    }
    {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            if (have_lists) drop_lists();
            return mis_set_error(ctx, MIS_E_HIP, "hipStreamSynchronize failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
        }
    }
    const auto ts = std::chrono::steady_clock::now();
    if (l2_bad) {
        if (have_lists) drop_lists();
        return mis_set_error(ctx, MIS_E_UNSUPPORTED, "L2 matching needs integer-valued descriptors in 0..255 (SIFT style)");
    }
    if (!early_lists) lists();
    // part 2: masks, H, confidence; the mirror entry gets H^-1 and swapped indices
    for (int k = 0; k < np; k++) {
        const PairDesc& pd = pairs[k];
        MisMatchesInfo* a = &out[pd.i * n + pd.j];
        MisMatchesInfo* b = &out[pd.j * n + pd.i];
        MisDMatch* bm = b->matches;
        if (po[k].ran_ransac) {
            a->inliers_mask = (uint8_t*)malloc((size_t)nm[k] + 1);
            memcpy(a->inliers_mask, hmask + pd.m_off, (size_t)nm[k]);
        }
        // matchers.cpp: "if (H.empty() || |det H| < eps) return" after the first estimation.  The two-chain flow tests it on
        // the device before the second estimation; the three-chain flow runs that estimation without waiting for the
        // refined H, and a degenerate first H drops it here (same expression, same rounding: no FMA contraction)
        const bool det_ok = !(fabs(det3(r1[k].H)) < DBL_EPSILON);
        const bool passed = po[k].passed && det_ok, second = po[k].second && det_ok;
        // H of the inlier-only estimation when it ran (it may come back empty), else of the first one
        const HomoResult& hr = second ? (fin[k] == 0 ? r2[k] : r3[k]) : r1[k];
        a->has_H = po[k].ran_ransac ? hr.ok : 0;
        if (a->has_H) memcpy(a->H, hr.H, sizeof(a->H));
        a->num_inliers = passed ? r1[k].ninl : 0;
        if (passed) {
            // Brown & Lowe confidence; > 3 means near-duplicate images and is zeroed (matchers.cpp)
            double c = a->num_inliers / (8 + 0.3 * nm[k]);
            a->confidence = c > 3. ? 0. : c;
        }
        *b = *a;
        b->src_img_idx = pd.j; b->dst_img_idx = pd.i;
        b->matches = bm;
        if (a->inliers_mask) {
            b->inliers_mask = (uint8_t*)malloc((size_t)nm[k] + 1);
            memcpy(b->inliers_mask, a->inliers_mask, (size_t)nm[k]);
        }
        if (a->has_H) invert3(a->H, b->H);
    }
    if (trace) {
        const auto te = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "match: enqueue %.0f us, device wait %.0f us, host assembly %.0f us\n", us(t_begin, tq), us(tq, ts), us(ts, te));
        if (ws->tev[0] && ws->tev[1] && ws->tev[3] && ws->tev[4] && ws->tev[5] && ws->tev[6]) {
            float e1 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0;
            hipEventElapsedTime(&e1, ws->tev[0], ws->tev[1]); hipEventElapsedTime(&e3, ws->tev[0], ws->tev[3]); hipEventElapsedTime(&e4, ws->tev[0], ws->tev[4]);
            hipEventElapsedTime(&e5, ws->tev[0], ws->tev[5]); hipEventElapsedTime(&e6, ws->tev[0], ws->tev[6]);
            float e2 = 0, e7 = 0;
            if (ws->tev[2]) hipEventElapsedTime(&e2, ws->tev[0], ws->tev[2]);
            if (ws->tev[7]) { hipEventElapsedTime(&e7, ws->tev[0], ws->tev[7]); fprintf(stderr, "match chains: main chain's second draw done %.2f\n", e7); }
            fprintf(stderr, "match chains, ms after the 2-NN pass was enqueued-behind (device events): first phase done %.2f | tails of its finishers done %.2f | main chain: second RANSAC phase done %.2f, done %.2f | side chain done %.2f | third chain done %.2f\n", e1, e6, e2, e3, e4, e5);
        }
    }
    return MIS_OK;
}

}  // namespace

// One-shot hook of this context's NEXT matcher call: fn(user) runs on the calling thread of mis_match_all_pairs /
// mis_match_pairs_sharded once all of the call's device work is enqueued, before the call waits for the device.
extern "C" int mis_match_on_enqueued(MisContext* ctx, void (*fn)(void*), void* user) {
    if (!ctx) return MIS_E_INVALID;
    MatchWorkspace* ws = workspace(ctx);
    ws->enqueued_cb = fn; ws->enqueued_user = user;
    return MIS_OK;
}

// Number of matcher calls this context has started (the next one will be this + 1).
extern "C" long long mis_match_sequence(MisContext* ctx) {
    if (!ctx) return -1;
    return workspace(ctx)->seq.load();
}

// Makes `stream` (any stream of the device, e.g. another context's) wait for the end of the 2-NN pass of this context's
// matcher call number `target_seq`, which another host thread is making: blocks the calling thread (at most timeout_ms)
// until that call has enqueued the pass, then enqueues the wait.  Returns MIS_FENCE_TIMEOUT (> 0, not an error) when the time
// ran out before that call showed up: nothing was queued, the stream simply does not wait; MIS_OK also when the call ended
// without a 2-NN pass.
extern "C" int mis_match_knn_fence(MisContext* ctx, void* stream, long long target_seq, int timeout_ms) {
    if (!ctx) return MIS_E_INVALID;
    MatchWorkspace* ws = (MatchWorkspace*)ctx->match_ws;
    MIS_CHECK(ctx, ws, MIS_E_STATE, "mis_match_sequence must have been called on this context first");
    const auto t0 = std::chrono::steady_clock::now();
    while (ws->knn_seq.load() < target_seq) {
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= timeout_ms) return MIS_FENCE_TIMEOUT;   // not an error: `stream` was not queued behind anything
        std::this_thread::yield();
    }
    if (stream && ws->ev_gate && hipStreamWaitEvent((hipStream_t)stream, ws->ev_gate, 0) != hipSuccess) return mis_set_error(ctx, MIS_E_HIP, "hipStreamWaitEvent failed");
    return MIS_OK;
}

// diagnostics: RANSAC states of the last matcher call (which = 0: first estimation of every pair, 1: the inlier-only one)
extern "C" int mis_debug_ransac_states(MisContext* ctx, int which, int* out, int cap) {
    if (!ctx || !ctx->match_ws || !out) return -1;
    MatchWorkspace* ws = (MatchWorkspace*)ctx->match_ws;
    return homo_batch_debug_states(ctx, which ? &ws->b2 : &ws->b1, out, cap);
}

extern "C" void mis_match_default_params(MisMatchParams* p) {
    if (p) *p = MisMatchParams{0.32f, 6, 6, 3.0, 2000, 0.995};
}

extern "C" int mis_match_all_pairs(MisContext* ctx, const MisFeatures* feats, int n, const MisMatchParams* p, MisMatchesInfo* out) {
    if (!ctx) return MIS_E_INVALID;
    return match_impl(ctx, feats, n, p, 0, 1, out);
}

extern "C" int mis_match_pairs_sharded(MisContext* ctx, const MisFeatures* feats, int n, const MisMatchParams* p, int rank, int world,
                                       MisMatchesInfo* out) {
    if (!ctx) return MIS_E_INVALID;
    return match_impl(ctx, feats, n, p, rank, world, out);
}

extern "C" int mis_matches_free(MisMatchesInfo* m, int count) {
    if (!m) return MIS_E_INVALID;
    for (int i = 0; i < count; i++) {
        free(m[i].matches); free(m[i].inliers_mask);
        m[i].matches = nullptr; m[i].inliers_mask = nullptr;
    }
    return MIS_OK;
}

extern "C" int mis_knn2(MisContext* ctx, const MisFeatures* q, const MisFeatures* t, int* idx2, float* dist2) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, q && t && idx2 && dist2, MIS_E_INVALID, "null argument");
    const bool binary = q->desc_dtype == MIS_U8 && q->desc_cols == 32 && t->desc_dtype == MIS_U8 && t->desc_cols == 32;
    const bool l2 = q->desc_dtype == MIS_F32 && t->desc_dtype == MIS_F32 && q->desc_cols == t->desc_cols && q->desc_cols >= 1 && q->desc_cols <= 128;
    MIS_CHECK(ctx, binary || l2, MIS_E_UNSUPPORTED, "mis_knn2 supports 32-byte binary descriptors (Hamming) and f32 descriptors of <= 128 columns (L2)");
    if (q->n <= 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MatchWorkspace* ws = workspace(ctx);
    hipStream_t st = ctx->stream;
    MIS_HIP(ctx, hipStreamSynchronize(st));
    const size_t nq = (size_t)q->n, nt = (size_t)std::max(t->n, 0);
    Carver dc;
    const size_t o_feats = dc.take(2 * sizeof(FeatDev)), o_pairs = dc.take(sizeof(PairDesc)), o_idx = dc.take(sizeof(int) * 2 * nq), o_dist = dc.take(sizeof(float) * 2 * nq);
    const size_t o_qh = dc.take(nq * 256), o_th = dc.take(nt * 256), o_qn = dc.take(nq * 4), o_tn = dc.take(nt * 4), o_bad = dc.take(4);
    const size_t o_part = dc.take((size_t)16 * L2_MAX_SLICES * nq);
    MIS_HIP(ctx, ws->dev.reserve(dc.off));
    uint8_t* D = (uint8_t*)ws->dev.p;
    int bad = 0;
    if (binary) {
        FeatDev fd[2] = {{(const uint8_t*)q->descriptors, q->keypoints, q->n, q->img_w, q->img_h},
                         {(const uint8_t*)t->descriptors, t->keypoints, t->n, t->img_w, t->img_h}};
        PairDesc pd;
        pd.i = 0; pd.j = 1; pd.knn_off12 = 0; pd.knn_off21 = q->n; pd.m_off = 0; pd.cap = q->n + t->n;
        MIS_HIP(ctx, hipMemcpyAsync(D + o_feats, fd, sizeof(fd), hipMemcpyHostToDevice, st));
        MIS_HIP(ctx, hipMemcpyAsync(D + o_pairs, &pd, sizeof(pd), hipMemcpyHostToDevice, st));
        MIS_HIP(ctx, hipStreamSynchronize(st));  // fd / pd live on this stack frame
        hipLaunchKernelGGL(knn2_hamming_kernel, dim3((q->n + 255) / 256, 1), dim3(256), 0, st, (const FeatDev*)(D + o_feats), (const PairDesc*)(D + o_pairs),
                           (int*)(D + o_idx), (float*)(D + o_dist));
    } else {
        MIS_HIP(ctx, hipMemsetAsync(D + o_bad, 0, 4, st));
        hipLaunchKernelGGL(l2_prep_kernel, dim3((q->n + 3) / 4), dim3(256), 0, st, (const float*)q->descriptors, q->n, q->desc_cols, (_Float16*)(D + o_qh),
                           (float*)(D + o_qn), (int*)(D + o_bad));
        if (t->n > 0)
            hipLaunchKernelGGL(l2_prep_kernel, dim3((t->n + 3) / 4), dim3(256), 0, st, (const float*)t->descriptors, t->n, t->desc_cols, (_Float16*)(D + o_th),
                               (float*)(D + o_tn), (int*)(D + o_bad));
        L2Set Q{(const _Float16*)(D + o_qh), (const float*)(D + o_qn), q->n}, T{(const _Float16*)(D + o_th), (const float*)(D + o_tn), std::max(t->n, 0)};
        l2_knn2_launch(st, ctx->num_cu, Q, T, D + o_part, (int*)(D + o_idx), (float*)(D + o_dist));
        MIS_HIP(ctx, hipMemcpyAsync(&bad, D + o_bad, 4, hipMemcpyDeviceToHost, st));
    }
    MIS_HIP(ctx, hipGetLastError());
    MIS_HIP(ctx, hipMemcpyAsync(idx2, D + o_idx, sizeof(int) * 2 * nq, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipMemcpyAsync(dist2, D + o_dist, sizeof(float) * 2 * nq, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipStreamSynchronize(st));
    MIS_CHECK(ctx, !bad, MIS_E_UNSUPPORTED, "L2 matching needs integer-valued descriptors in 0..255 (SIFT style): exactness of the fp16 MFMA path");
    return MIS_OK;
}

extern "C" int mis_find_homography(MisContext* ctx, const float* src, const float* dst, int n, double thresh, int max_iters, double confidence,
                                   double H[9], uint8_t* mask, int* ok) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, src && dst && H && ok && n >= 0, MIS_E_INVALID, "null argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MatchWorkspace* ws = workspace(ctx);
    hipStream_t st = ctx->stream;
    MIS_HIP(ctx, hipStreamSynchronize(st));
    const size_t nn = (size_t)std::max(n, 1);
    Carver dc;
    const size_t o_src = dc.take(sizeof(float) * 2 * nn), o_dst = dc.take(sizeof(float) * 2 * nn), o_mask = dc.take(nn);
    MIS_HIP(ctx, ws->dev.reserve(dc.off));
    uint8_t* D = (uint8_t*)ws->dev.p;
    int rc;
    if ((rc = homo_batch_reserve(ctx, &ws->b1, 1, (long long)nn, max_iters)) != MIS_OK) return rc;
    if (n) {
        MIS_HIP(ctx, hipMemcpyAsync(D + o_src, src, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
        MIS_HIP(ctx, hipMemcpyAsync(D + o_dst, dst, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    }
    HomoCall c;
    c.src = (const float*)(D + o_src); c.dst = (const float*)(D + o_dst); c.mask = D + o_mask; c.pt_off = 0; c.n = n; c.active = 1;
    MIS_HIP(ctx, hipMemcpyAsync(ws->b1.calls, &c, sizeof(c), hipMemcpyHostToDevice, st));
    MIS_HIP(ctx, hipStreamSynchronize(st));  // `c` lives on this stack frame
    if ((rc = homo_batch_run(ctx, &ws->b1, thresh, max_iters, confidence)) != MIS_OK) return rc;
    HomoResult r;
    MIS_HIP(ctx, hipMemcpyAsync(&r, ws->b1.results, sizeof(r), hipMemcpyDeviceToHost, st));
    if (mask && n) MIS_HIP(ctx, hipMemcpyAsync(mask, D + o_mask, n, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipStreamSynchronize(st));
    *ok = r.ok;
    if (r.ok) memcpy(H, r.H, sizeof(r.H)); else memset(H, 0, sizeof(r.H));
    return MIS_OK;
}

// myLeaveBiggestComponent (image_stitching.cpp:215-278): union-find over pairs with
// confidence >= threshold (cv::detail::DisjointSets semantics), indices of the biggest component
// myLeaveBiggestComponent's graph part: union-find over the pairs with confidence >= threshold, indices of the largest set
static int biggest_component(const double* conf, int n, float conf_threshold, int* indices, int* n_indices) {
    std::vector<int> parent(n), rank_(n, 0), size(n, 1);
    for (int i = 0; i < n; i++) parent[i] = i;
    auto find = [&](int elem) {
        int set = elem;
        while (set != parent[set]) set = parent[set];
        while (elem != parent[elem]) { int next = parent[elem]; parent[elem] = set; elem = next; }
        return set;
    };
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if (conf[(size_t)i * n + j] < conf_threshold) continue;
            int c1 = find(i), c2 = find(j);
            if (c1 == c2) continue;
            if (rank_[c1] < rank_[c2]) { parent[c1] = c2; size[c2] += size[c1]; }
            else if (rank_[c2] < rank_[c1]) { parent[c2] = c1; size[c1] += size[c2]; }
            else { parent[c1] = c2; rank_[c2]++; size[c2] += size[c1]; }
        }
    int max_comp = (int)(std::max_element(size.begin(), size.end()) - size.begin());
    int k = 0;
    for (int i = 0; i < n; i++) if (find(i) == max_comp) indices[k++] = i;
    *n_indices = k;
    return MIS_OK;
}

extern "C" int mis_leave_biggest_component(const MisMatchesInfo* pm, int n, float conf_threshold, int* indices, int* n_indices) {
    if (!pm || !indices || !n_indices || n < 1) return MIS_E_INVALID;
    std::vector<double> conf((size_t)n * n);
    for (size_t i = 0; i < conf.size(); i++) conf[i] = pm[i].confidence;
    return biggest_component(conf.data(), n, conf_threshold, indices, n_indices);
}

extern "C" int mis_leave_biggest_component_conf(const double* confidence, int n, float conf_threshold, int* indices, int* n_indices) {
    if (!confidence || !indices || !n_indices || n < 1) return MIS_E_INVALID;
    return biggest_component(confidence, n, conf_threshold, indices, n_indices);
}
