// homography.hip -- batched cv::findHomography(src, dst, mask, RANSAC, thresh, maxIters, confidence)
// (SURVEY K9), the estimator behind BestOf2NearestMatcher::match
// (image_stitching/image_stitching.cpp:653).  OpenCV sources restated: calib3d fundam.cpp
// (HomographyEstimatorCallback, HomographyRefineCallback, findHomography), ptsetreg.cpp
// (RANSACPointSetRegistrator), core lapack.cpp (Jacobi), LMSolver.
//
// A batch = many independent problems (one per image pair).  Four kernels, all exact:
//   draw_kernel       per problem: the RANSAC subset sequence.  cv::RNG(-1)'s output stream is data
//                     independent; what depends on the data is how many draws each getSubset attempt
//                     consumes (duplicate redraws) and whether it passes checkSubset.  Every stream
//                     position is simulated as the start of ONE attempt in parallel, then one thread
//                     chases start -> end -> ... through the table: the exact sequential sequence.
//   hyp_quad_kernel   four lanes per hypothesis: 4-point normalised DLT, 9x9 Jacobi on the upper triangle;
//   hyp_count_kernel  one wave per hypothesis: its inlier count.
//   scan_tail_kernel  sequential replay of the adaptive loop (niters update) over the scored
//                     hypotheses; when the loop ends: inlier mask, ordered compaction, DLT on all
//                     inliers, 10-iteration LM.  f64 sums keep the CPU's order: per-point terms in
//                     parallel, then ONE thread per accumulator; Jacobi rotations spread over n threads.
// Phase 0 covers hypotheses [0,128) (overlapping pairs converge there), phase 1 the rest.
#include "homography.h"
#include "dev_math.h"
#include <algorithm>
#include <mutex>
#include <vector>

namespace {

#ifndef MIS_PHASE0
#define MIS_PHASE0 128
#endif
constexpr int PHASE0 = MIS_PHASE0;       // hypotheses evaluated before the first replay (a single phase over all 2000 was measured: 26 ms instead of 15 ms per step)
constexpr int RNG_TABLE = 1 << 17;
constexpr int TB = 256;           // threads of draw / scan_tail workgroups

struct RansacState {
    int mode;      // 0 skip, 1 exactly 4 points, 2 RANSAC
    int n_sub;     // subsets drawn so far (the sequential getSubset sequence)
    int iter, niters, max_good, done, best_k, result;
    int tail_pending;   // the loop ended in a scan-only launch: mask / DLT / LM refinement still to run (scan_tail_kernel part 2)
    int draw_k, draw_fail;   // next iteration to draw; getSubset exhausted its 10000 attempts
    long long draw_pos;      // RNG stream position after the last drawn subset
};

// ---------------------------------------------------------------- shared scalar helpers --------
// IEEE f64 division and square root as the compiler's own FMA chains (v_rcp_f64 / v_rsq_f64, two refinement steps, one residual
// correction) WITHOUT the operand scaling (v_div_scale, v_ldexp) and the special-case selection (v_div_fmas / v_div_fixup,
// v_cmp_class) around them: 8 dependent operations instead of 10 (division) and 13 (square root).  A Jacobi rotation is one
// dependency chain of three divisions and two square roots, so this is a quarter of its arithmetic latency
// (tools/micro/lat_bench.hip: 904 cycles for the generic chain).  Bit-identical under ONE guard per rotation: the pivot p and
// y = (W[l] - W[k]) / 2 normal with exponents in [-370, 370] -- then every operand of the chain (|p|, |y|, t <= 2.5 max, s <= 1.5 t)
// stays within [-370, 373]: v_div_scale leaves such operands alone (|exponent difference| < 768, no denormal anywhere), the
// square roots' arguments are 1 + (b/a)^2 in [1, 2].  A per-division guard costs more than it saves (a divergent branch per
// division: 177 cycles against the generic division's 104).
__device__ __forceinline__ bool jd_mid(int hi_word) { return (((unsigned)hi_word >> 20) & 0x7ffu) - 653u <= 740u; }
__device__ __forceinline__ double jd_div(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}
__device__ __forceinline__ double jd_sqrt_1to2(double x) {      // x in [1, 2]
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double cv_hypot(double a, double b) {
    a = fabs(a); b = fabs(b);
    if (a > b) { b /= a; return a * sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
    return 0;
}
// the rotation of JacobiImpl_ for the pivot p and y = (W[l] - W[k]) / 2: c, s and the diagonal shift t
template <bool LEAN>
__device__ __forceinline__ void jacobi_rotation(double p, double y, double* c_, double* s_, double* t_) {
    double tt, sn, c;
    if (LEAN) {
        // cv_hypot(a, b) = max * sqrt(1 + (min / max)^2) in both of its branches (a == b takes the second: 1 + 1)
        const double ap = fabs(p), ay = fabs(y);
        const double big = fmax(ap, ay), q1 = jd_div(fmin(ap, ay), big);
        tt = ay + big * jd_sqrt_1to2(1 + q1 * q1);
        const double q2 = jd_div(ap, tt);              // tt >= |p|
        sn = tt * jd_sqrt_1to2(1 + q2 * q2);
        c = jd_div(tt, sn);
        sn = jd_div(p, sn); tt = jd_div(p, tt) * p;
    } else {
        tt = fabs(y) + cv_hypot(p, y);
        sn = cv_hypot(p, tt);
        c = tt / sn;
        sn = p / sn; tt = (p / tt) * p;
    }
    if (y < 0) sn = -sn, tt = -tt;
    *c_ = c; *s_ = sn; *t_ = tt;
}

// fundam.cpp haveCollinearPoints (4.5.x: only the last point is tested) + the orientation test
__device__ bool have_collinear4(const float* p) {
    const int i = 3;
    for (int j = 0; j < i; j++) {
        double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}
__device__ double det3_pts(const float* p, int t0, int t1, int t2) {
    double a00 = p[2 * t0], a01 = p[2 * t0 + 1], a10 = p[2 * t1], a11 = p[2 * t1 + 1], a20 = p[2 * t2], a21 = p[2 * t2 + 1];
    return a00 * (a11 * 1. - 1. * a21) - a01 * (a10 * 1. - 1. * a20) + 1. * (a10 * a21 - a11 * a20);
}
__device__ bool check_subset(const float* s, const float* d) {
    if (have_collinear4(s) || have_collinear4(d)) return false;
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; i++) negative += det3_pts(s, tt[i][0], tt[i][1], tt[i][2]) * det3_pts(d, tt[i][0], tt[i][1], tt[i][2]) < 0;
    return negative == 0 || negative == 4;
}

__device__ int ransac_update_num_iters(double p, double ep, int max_iters) {
    if (p < 0.) p = 0.; if (p > 1.) p = 1.;
    if (ep < 0.) ep = 0.; if (ep > 1.) ep = 1.;
    double num = 1. - p; if (num < DBL_MIN) num = DBL_MIN;
    double w = 1. - ep, w2 = w * w;
    double denom = 1. - w2 * w2;
    if (denom < DBL_MIN) return 0;
    num = mis_log_d(num);
    denom = mis_log_d(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : mis_round_d(num / denom);
}

__device__ __forceinline__ int is_inlier(const float* Hf, float Mx, float My, float mx, float my, float t) {
    float ww = 1.f / ((Hf[6] * Mx + Hf[7] * My) + 1.f);
    float dx = ((Hf[0] * Mx + Hf[1] * My) + Hf[2]) * ww - mx;
    float dy = ((Hf[3] * Mx + Hf[4] * My) + Hf[5]) * ww - my;
    float e = dx * dx + dy * dy;
    return e <= t;
}

// eigenvector of the smallest eigenvalue (H0) -> de-normalised, scaled homography
__device__ void dlt_denormalise(const double* H0, const double* nrm /* cmx cmy cMx cMy smx smy sMx sMy */, double* H) {
    double T[9], R[9];
    const double invHnorm[9] = {1. / nrm[4], 0, nrm[0], 0, 1. / nrm[5], nrm[1], 0, 0, 1};
    const double Hnorm2[9] = {nrm[6], 0, -nrm[2] * nrm[6], 0, nrm[7], -nrm[3] * nrm[7], 0, 0, 1};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double acc = 0;
        for (int k = 0; k < 3; k++) acc += invHnorm[i * 3 + k] * H0[k * 3 + j];
        T[i * 3 + j] = acc;
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double acc = 0;
        for (int k = 0; k < 3; k++) acc += T[i * 3 + k] * Hnorm2[k * 3 + j];
        R[i * 3 + j] = acc;
    }
    double sc = 1. / R[8];
    for (int i = 0; i < 9; i++) H[i] = R[i] * sc;
}

#ifndef MIS_CHAIN_PRIO
#define MIS_CHAIN_PRIO 3
#endif
#ifdef MIS_TAIL_PROF
__device__ unsigned long long g_draw_prof[12];   // per phase p (0, 1): [4p] chunks (all problems), [4p+1] most chunks of a problem, [4p+2] longest problem (wall ticks), [4p+3] problems with work
#endif
// ---------------------------------------------------------------- draw_kernel ------------------
// cv::RNG multiply-with-carry stream: U[s] is the (s+1)-th output from seed (uint64)-1
struct DrawCtx {
    const unsigned* U;
    const unsigned* lds_u = nullptr;   // stream positions lds_base .. lds_base + lds_n - 1 staged in LDS (the current chunk of draw_kernel + the reach of an attempt)
    long long lds_base = 0;
    int lds_n = 0;
    bool lds_mod = false;              // the staged values are draw % n
    unsigned long long state_T;  // generator state after RNG_TABLE draws
    // serial continuation beyond the table (positions are visited in increasing order)
    unsigned long long cur_state;
    long long cur_pos;
};
__device__ __forceinline__ unsigned draw_at(DrawCtx& d, long long pos) {
    const long long rel = pos - d.lds_base;
    if (rel >= 0 && rel < d.lds_n) return d.lds_u[rel];
    if (pos < RNG_TABLE) return d.U[pos];
    unsigned v = 0;
    while (d.cur_pos <= pos) {
        d.cur_state = (unsigned long long)(unsigned)d.cur_state * 4164903690u + (unsigned)(d.cur_state >> 32);
        v = (unsigned)d.cur_state;
        d.cur_pos++;
    }
    return v;
}

// draw % n at a stream position: the staged positions hold the remainder already (draw_kernel reduces a chunk's numbers once, as it
// stages them -- an attempt reads four or more consecutive positions and every position is read by the attempts of four starts)
__device__ __forceinline__ int draw_mod_at(DrawCtx& d, long long pos, int n) {
    const long long rel = pos - d.lds_base;
    if (d.lds_mod && rel >= 0 && rel < d.lds_n) return (int)d.lds_u[rel];
    return (int)(draw_at(d, pos) % (unsigned)n);
}

// The draws of ONE getSubset attempt starting at stream position pos: four distinct indices (redraw on duplicates).  Returns the
// end position.  Every stream position is a possible start, one in ~4.5 is a real one: the chunk loop of draw_kernel finds the
// real ones from the lengths alone and runs checkSubset (150 f64 operations) on those only.
__device__ __forceinline__ long long attempt_len(DrawCtx& d, long long pos, int n, int* idx) {
    for (int i = 0; i < 4; i++) {
        int idx_i;
        bool dup;
        do {
            idx_i = draw_mod_at(d, pos++, n);
            dup = false;
            for (int q = 0; q < i; q++) dup |= idx[q] == idx_i;
        } while (dup);
        idx[i] = idx_i;
    }
    return pos;
}
__device__ __forceinline__ bool subset_passes(const float* src, const float* dst, const int* idx) {
    float ms1[8], ms2[8];
    for (int i = 0; i < 4; i++) {
        const float2 ps = reinterpret_cast<const float2*>(src)[idx[i]], pd = reinterpret_cast<const float2*>(dst)[idx[i]];
        ms1[2 * i] = ps.x; ms1[2 * i + 1] = ps.y;
        ms2[2 * i] = pd.x; ms2[2 * i + 1] = pd.y;
    }
    return check_subset(ms1, ms2);
}

// ONE getSubset attempt starting at stream position pos: four distinct indices (redraw on duplicates),
// then checkSubset.  Returns the end position; *pass / idx describe the attempt.
__device__ long long attempt_at(DrawCtx& d, long long pos, const float* src, const float* dst, int n, int* idx, bool* pass) {
    float ms1[8], ms2[8];
    for (int i = 0; i < 4; i++) {
        int idx_i;
        bool dup;
        do {
            idx_i = (int)(draw_at(d, pos++) % (unsigned)n);
            dup = false;
            for (int q = 0; q < i; q++) dup |= idx[q] == idx_i;
        } while (dup);
        idx[i] = idx_i;
        const float2 ps = reinterpret_cast<const float2*>(src)[idx_i], pd = reinterpret_cast<const float2*>(dst)[idx_i];
        ms1[2 * i] = ps.x; ms1[2 * i + 1] = ps.y;
        ms2[2 * i] = pd.x; ms2[2 * i + 1] = pd.y;
    }
    *pass = check_subset(ms1, ms2);
    return pos;
}

// Subsets of iterations [st->draw_k, k_hi) of every problem.  Attempt start positions are simulated in
// chunks of DRAW_CHUNK stream positions by all threads, then thread 0 chases through the chunk; a new
// chunk starts exactly where the chase left the previous one.
constexpr int DRAW_CHUNK = 4096;
#ifndef MIS_DRAW_TB
#define MIS_DRAW_TB 1024
#endif
constexpr int DRAW_TB = MIS_DRAW_TB;   // threads of a problem's workgroup: a chunk's positions are simulated DRAW_CHUNK / DRAW_TB per thread (256 threads: 0.54 + 1.08 ms for the two draw launches on the matcher's critical path)
constexpr int DRAW_PTS = 2048;
__global__ __launch_bounds__(DRAW_TB) void draw_kernel(const HomoCall* calls, RansacState* states, int* sub_idx, int* draw_idx, const unsigned* U,
                                                  unsigned long long state_T, int max_iters, int phase, int k_hi_arg) {
#if MIS_CHAIN_PRIO
    __builtin_amdgcn_s_setprio(MIS_CHAIN_PRIO);      // a latency-bound chain beside the composition's bandwidth-bound kernels: its few waves issue first
#endif
    __shared__ unsigned char tab[DRAW_CHUNK];  // per position: min(end - start, 127) | pass << 7
    __shared__ float2 pts[2 * DRAW_PTS];       // src then dst of small problems: the random gathers stay on chip
    __shared__ long long s_pos;
    __shared__ int s_k, s_attempts, s_more, s_kfirst, s_nacc;
    __shared__ unsigned short acc_o[DRAW_CHUNK / 4];  // chunk offsets of the accepted attempts (serial path)
    __shared__ unsigned short nxtA[DRAW_CHUNK + 1], nxtB[DRAW_CHUNK + 1];  // J^(2^r): start of the attempt 2^r hops ahead
    __shared__ unsigned char reach[DRAW_CHUNK];
    __shared__ unsigned su[DRAW_CHUNK + 128];
    __shared__ int scan[DRAW_TB];
    __shared__ int s_big, s_firstvis, s_lastaccvis;
    __shared__ long long s_endpos;
    const int b = blockIdx.x, t = threadIdx.x;
#ifdef MIS_TAIL_PROF
    const unsigned long long dpk = wall_clock64();
#endif
    const HomoCall c = calls[b];
    RansacState* st = states + b;
    if (phase == 0) {
        if (t == 0) {
            st->mode = (!c.active || c.n < 4) ? 0 : (c.n == 4 ? 1 : 2);
            st->n_sub = 0; st->iter = 0; st->niters = max_iters > 1 ? max_iters : 1; st->max_good = 0; st->done = 0; st->best_k = -1; st->result = 0; st->tail_pending = 0;
            st->draw_k = 0; st->draw_fail = 0; st->draw_pos = 0;
        }
        if (!c.active || c.n <= 4) return;
    } else {
        if (st->mode != 2 || st->done || st->draw_fail) return;
    }
    __syncthreads();
    const int k_hi = min(k_hi_arg, phase == 0 ? max_iters : st->niters);
    if (t == 0) { s_pos = st->draw_pos; s_k = st->draw_k; s_attempts = 0; s_more = s_k < k_hi; }
    __syncthreads();
    DrawCtx d;
    d.U = U; d.state_T = state_T; d.cur_state = state_T; d.cur_pos = RNG_TABLE;
    const float* psrc = c.src;
    const float* pdst = c.dst;
    if (c.n <= DRAW_PTS) {
        for (int i = t; i < c.n; i += DRAW_TB) { pts[i] = reinterpret_cast<const float2*>(c.src)[i]; pts[DRAW_PTS + i] = reinterpret_cast<const float2*>(c.dst)[i]; }
        psrc = reinterpret_cast<const float*>(pts); pdst = reinterpret_cast<const float*>(pts + DRAW_PTS);
        __syncthreads();
    }
    int* didx = draw_idx + (size_t)b * DRAW_CHUNK * 4;
    int* sidx = sub_idx + (size_t)b * max_iters * 4;
    if (phase == 0 && c.n <= 10) {
        // Tiny problems (a handful of spurious matches) are often infeasible outright: every ordered 4-tuple of distinct
        // points fails checkSubset, so getSubset burns its 10000 attempts and the RANSAC loop ends without a model.
        // All n (n-1)(n-2)(n-3) <= 5040 tuples are tested here; if none passes, that outcome is known without simulating
        // ~150k stream positions (two such problems set the duration of the whole launch in a 16-frame job).
        const int n = c.n, total = n * (n - 1) * (n - 2) * (n - 3);
        int any = 0;
        for (int q = t; q < total && !any; q += DRAW_TB) {
            int r = q, id[4];
            id[0] = r % n; r /= n;
            int a1 = r % (n - 1); r /= (n - 1);
            int a2 = r % (n - 2); r /= (n - 2);
            int a3 = r;
            // k-th unused index, in increasing order
            auto pick = [&](int kth, int cnt) {
                for (int v = 0; v < n; v++) {
                    bool used = false;
                    for (int u = 0; u < cnt; u++) used |= id[u] == v;
                    if (!used && kth-- == 0) return v;
                }
                return 0;
            };
            id[1] = pick(a1, 1); id[2] = pick(a2, 2); id[3] = pick(a3, 3);
            float ms1[8], ms2[8];
            for (int i = 0; i < 4; i++) {
                const float2 ps = reinterpret_cast<const float2*>(psrc)[id[i]], pd = reinterpret_cast<const float2*>(pdst)[id[i]];
                ms1[2 * i] = ps.x; ms1[2 * i + 1] = ps.y; ms2[2 * i] = pd.x; ms2[2 * i + 1] = pd.y;
            }
            any |= check_subset(ms1, ms2) ? 1 : 0;
        }
        const int found = __syncthreads_or(any);
#ifdef MIS_TAIL_PROF
        (void)dpk;
#endif
        if (!found) {
            if (t == 0) { st->draw_fail = 1; st->n_sub = 0; st->draw_k = 0; }
            return;
        }
    }
#ifdef MIS_TAIL_PROF
    const unsigned long long dp0 = wall_clock64();
    unsigned long long dp_last = dp0;
    int dp_chunks = 0;

#endif
    while (s_more) {
#ifdef MIS_TAIL_PROF
        dp_chunks++;
#endif
        const long long base = s_pos;
        // the chunk's random numbers (+ the 127 positions an attempt can reach past it) staged in LDS: an attempt is a chain of
        // draw -> modulo -> duplicate test -> next draw, and from the table in global memory every link was a memory latency
        // (17 of a chunk's 31 us)
        __syncthreads();        // (the previous chunk's serial path may still read the staging)
        for (int o = t; o < DRAW_CHUNK + 128; o += DRAW_TB) su[o] = (base + o < RNG_TABLE ? U[base + o] : 0u) % (unsigned)c.n;
        d.lds_u = su; d.lds_base = base; d.lds_n = (int)max(0ll, min((long long)(DRAW_CHUNK + 128), (long long)RNG_TABLE - base)); d.lds_mod = true;
        __syncthreads();
        for (int o = t; o < DRAW_CHUNK; o += DRAW_TB) {      // lengths only; the verdicts of the visited attempts follow the chase
            int idx[4];
            const long long delta = attempt_len(d, base + o, c.n, idx) - (base + o);
            tab[o] = (unsigned char)(delta > 127 ? 127 : delta);
        }
        if (t == 0) s_big = 0;
        __syncthreads();
#ifdef MIS_TAIL_PROF
        const unsigned long long cs1 = wall_clock64();
#endif
        // ---- which attempts does the sequential chain visit?  start -> end -> ... by pointer doubling ----
        constexpr int C = DRAW_CHUNK;
        for (int q = t; q < C; q += DRAW_TB) {
            const unsigned char e = tab[q];
            if ((e & 0x7f) == 127) s_big = 1;  // an attempt longer than 126 draws: take the serial path for this chunk
            nxtA[q] = (unsigned short)(q + (e & 0x7f));
            reach[q] = q == 0;
        }
        if (t == 0) { nxtA[C] = C; nxtB[C] = C; }
        __syncthreads();
        if (!s_big) {
            unsigned short* cur = nxtA;
            unsigned short* oth = nxtB;
            for (int r = 0; r < 11; r++) {  // 2^11 hops > C / 4 attempts
                for (int q = t; q < C; q += DRAW_TB) if (reach[q]) { const int j = cur[q]; if (j < C) reach[j] = 1; }
                __syncthreads();
                for (int q = t; q < C; q += DRAW_TB) { const int j = min((int)cur[q], C); oth[q] = j < C ? cur[j] : (unsigned short)C; }
                __syncthreads();
                unsigned short* tmp = cur; cur = oth; oth = tmp;
            }
#ifdef MIS_TAIL_PROF
            const unsigned long long cs2 = wall_clock64();
#endif
            // checkSubset of the visited attempts: hops are >= 4 positions, so a thread's C / DRAW_TB = 4 consecutive positions hold at
            // most one (every thread has at most one attempt to test)
            const int q0 = t * (C / DRAW_TB);
            int qv = -1;
            for (int q = q0; q < q0 + C / DRAW_TB; q++) qv = reach[q] ? q : qv;
            if (qv >= 0) {       // outside the loop: inside it the wave would run the test once per loop trip, a quarter of its lanes each time
                int idx[4];
                attempt_len(d, base + qv, c.n, idx);
                if (subset_passes(psrc, pdst, idx)) tab[qv] |= 0x80;
                *reinterpret_cast<int4*>(didx + 4 * qv) = make_int4(idx[0], idx[1], idx[2], idx[3]);
            }
            // ranks of the visited / accepted attempts in stream order (each thread owns C / DRAW_TB consecutive positions)
            int lv = 0, la = 0;
            for (int q = q0; q < q0 + C / DRAW_TB; q++) { const int rv = reach[q]; lv += rv; la += rv && (tab[q] & 0x80); }
            scan[t] = (lv << 16) | la;
            __syncthreads();
            for (int o = 1; o < DRAW_TB; o <<= 1) {
                const int add = t >= o ? scan[t - o] : 0;
                __syncthreads();
                scan[t] += add;
                __syncthreads();
            }
            const int incl = scan[t], total = scan[DRAW_TB - 1];
            const int total_v = total >> 16, total_a = total & 0xffff;
            int vr = (incl >> 16) - lv, ar = (incl & 0xffff) - la;  // exclusive ranks at q0
            const int need = k_hi - s_k, cut = min(total_a, need);
            if (t == 0) { s_firstvis = -1; s_lastaccvis = -1; s_endpos = -1; }
            __syncthreads();
            for (int q = q0; q < q0 + C / DRAW_TB; q++) {
                if (!reach[q]) continue;
                const unsigned char e = tab[q];
                if (e & 0x80) {
                    if (ar == 0) s_firstvis = vr;
                    if (ar < cut) *reinterpret_cast<int4*>(sidx + 4 * (s_k + ar)) = *reinterpret_cast<const int4*>(didx + 4 * q);
                    if (ar == cut - 1) { s_lastaccvis = vr; if (cut == need) s_endpos = base + q + (e & 0x7f); }
                    ar++;
                }
                if (vr == total_v - 1 && cut < need) s_endpos = base + q + (e & 0x7f);  // the chain leaves the chunk here
                vr++;
            }
            __syncthreads();
            if (t == 0) {
                int attempts = s_attempts;
                bool fail = false;
                if (total_a == 0 || cut == 0) { attempts += total_v; fail = attempts >= 10000; }
                else {
                    fail = attempts + s_firstvis >= 10000;                 // getSubset gave up before the first accept
                    attempts = cut == need ? 0 : total_v - (s_lastaccvis + 1);  // failures trailing the last accepted attempt
                }
                const int k = fail ? s_k : s_k + cut;
                s_pos = s_endpos; s_k = k; s_attempts = attempts;
                s_more = !fail && k < k_hi;
                if (!s_more) { st->draw_pos = s_pos; st->draw_k = k; st->draw_fail = fail; st->n_sub = k; }
#ifdef MIS_TAIL_PROF
                const unsigned long long cs3 = wall_clock64();
                atomicAdd(&g_draw_prof[8], cs1 - (dp_last)); atomicAdd(&g_draw_prof[9], cs2 - cs1); atomicAdd(&g_draw_prof[10], cs3 - cs2); atomicAdd(&g_draw_prof[11], 1ull);
#endif
            }
            __syncthreads();
#ifdef MIS_TAIL_PROF
            dp_last = wall_clock64();
#endif
            continue;
        }
        __syncthreads();
        // serial path (an attempt of the chunk consumed > 126 draws: tiny n): the verdict of every position, then one thread walks
        for (int o = t; o < DRAW_CHUNK; o += DRAW_TB) {
            if ((tab[o] & 0x7f) == 127) continue;
            int idx[4];
            attempt_len(d, base + o, c.n, idx);
            if (subset_passes(psrc, pdst, idx)) tab[o] |= 0x80;
            *reinterpret_cast<int4*>(didx + 4 * o) = make_int4(idx[0], idx[1], idx[2], idx[3]);
        }
        __syncthreads();
        if (t == 0) {
            long long pos = base;
            int k = s_k, attempts = s_attempts, nacc = 0;
            bool fail = false;
            while (k < k_hi && pos < base + DRAW_CHUNK) {
                unsigned char e = tab[pos - base];
                bool pass;
                if ((e & 0x7f) != 127) {
                    pass = (e & 0x80) != 0;
                    if (pass) acc_o[nacc++] = (unsigned short)(pos - base);  // indices copied in parallel after the chase
                    pos += e & 0x7f;
                } else {  // an attempt that consumed > 126 draws (tiny n): redo it serially
                    int idx[4];
                    pos = attempt_at(d, pos, psrc, pdst, c.n, idx, &pass);
                    if (pass) { acc_o[nacc++] = 0xffff; sidx[4 * k] = idx[0]; sidx[4 * k + 1] = idx[1]; sidx[4 * k + 2] = idx[2]; sidx[4 * k + 3] = idx[3]; }
                }
                attempts++;
                if (pass) { k++; attempts = 0; }
                else if (attempts >= 10000) { fail = true; break; }  // getSubset gave up: the RANSAC loop ends here
            }
            s_kfirst = s_k; s_nacc = nacc;
            s_pos = pos; s_k = k; s_attempts = attempts;
            s_more = !fail && k < k_hi;
            if (!s_more) { st->draw_pos = pos; st->draw_k = k; st->draw_fail = fail; st->n_sub = k; }
        }
        __syncthreads();
        for (int j = t; j < s_nacc; j += DRAW_TB) {
            const unsigned o = acc_o[j];
            if (o != 0xffff) *reinterpret_cast<int4*>(sidx + 4 * (s_kfirst + j)) = *reinterpret_cast<const int4*>(didx + 4 * o);
        }
        __syncthreads();
    }
#ifdef MIS_TAIL_PROF
    if (t == 0 && dp_chunks) {
        const int ph = phase ? 4 : 0;
        atomicAdd(&g_draw_prof[ph], (unsigned long long)dp_chunks); atomicMax(&g_draw_prof[ph + 1], (unsigned long long)dp_chunks);
        atomicMax(&g_draw_prof[ph + 2], wall_clock64() - dp0); atomicAdd(&g_draw_prof[ph + 3], 1ull);
    }
#endif
}

#ifdef MIS_TAIL_PROF
__device__ unsigned long long g_hyp_prof[8];    // [6], [7]: summed / longest replay of a second phase (scan_tail_kernel)
#endif
// ---------------------------------------------------------------- hyp_quad_kernel --------------
// The 4-point solves (normalised DLT + 9 x 9 Jacobi of every hypothesis) with FOUR lanes per hypothesis (a DPP quad).  One thread
// per solve (rounds 1-2's hyp_kernel, removed in round 4) ran ~1300 instructions per rotation on a wave that is alone on its SIMD (126 doubles of LDS per solve allow two waves per compute
// unit): 9500 cycles per rotation, 140 rotations, 0.6 ms per launch whatever its size, and three such launches sit on the
// matcher's critical path.  Here lane q of a quad owns the indices i = q, q + 4, q + 8 (< 9): its element pairs of A, its
// columns of V, the candidates of rows / columns i (index + |value| in registers).  A rotation is: quad arg-max of the
// candidates (two DPP steps), ONE LDS round trip for p, W[k], W[l] and the lane's <= 3 + 3 pairs, the arithmetic chain (every
// lane computes c, s, t), the lane's share of the rotation, four quad arg-max re-scans over the rotated values in registers.
// A wave holds 16 solves and a workgroup 64 (66 KB of LDS: the same 128 solves per compute unit, on eight waves instead of
// two).  Arithmetic per element and visiting order of every scan are those of the serial loops.
#ifndef MIS_HQ_HYPS
#define MIS_HQ_HYPS 64
#endif
constexpr int HQ_HYPS = MIS_HQ_HYPS;  // solves per workgroup (4 lanes each)
constexpr int HQ_STRIDE = HQ_HYPS + 1;         // doubles between consecutive elements of a solve (solve h at + h): the quad's lanes, same h, different elements, fall in different banks
constexpr int HQ_ELEMS = 127;         // 36 (strict upper triangle) + 9 (W) + 81 (V) + 1 (a dummy element: the target of the pairs that do not exist)
struct QSlot {
    double* base;
    __device__ __forceinline__ double& e(int idx) const { return base[idx * HQ_STRIDE]; }
};
__device__ __forceinline__ int hq_T(int r) { return ((r * (17 - r)) >> 1) - r - 1; }       // element (r, c), r < c, at T(r) + c
constexpr int HQ_W = 36, HQ_V = 45, HQ_DUMMY = 126;

struct QCand { double a; int pk; };     // |value| (or -1: none) and a packed payload whose upper bits order ties (smaller first)
template <int CTRL>
__device__ __forceinline__ QCand hq_step(const QCand& c) {
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(c.a), CTRL, 0xf, 0xf, false);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(c.a), CTRL, 0xf, 0xf, false);
    const int pk = __builtin_amdgcn_update_dpp(0, c.pk, CTRL, 0xf, 0xf, false);
    const double oa = __hiloint2double(hi, lo);
    const bool take = (c.a < oa) | ((c.a == oa) & (pk < c.pk));
    QCand r;
    r.a = take ? oa : c.a; r.pk = take ? pk : c.pk;
    return r;
}
__device__ __forceinline__ QCand hq_quad_best(QCand c) {       // all four lanes end with the quad's first maximum
    c = hq_step<0xB1>(c);      // quad_perm [1,0,3,2]
    c = hq_step<0x4E>(c);      // quad_perm [2,3,0,1]
    return c;
}

__device__ void jacobi9_quad(const QSlot s, const int q) {
    constexpr int n = 9;
    const double eps = DBL_EPSILON;
    int ii[3];
    bool own[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { ii[j] = q + 4 * j; own[j] = ii[j] < n; }
    for (int e = q; e < n * n; e += 4) s.e(HQ_V + e) = (e % (n + 1) == 0) ? 1. : 0.;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // candidates of the owned rows / columns: index and |value|
    int ir[3], ic[3];
    double arv[3], acv[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int i = ii[j];
        ir[j] = 0; ic[j] = 0; arv[j] = -1.; acv[j] = -1.;
        if (own[j] && i < n - 1) {
            int m = i + 1;
            double mv = fabs(s.e(hq_T(i) + m));
            for (int c = i + 2; c < n; c++) { const double v = fabs(s.e(hq_T(i) + c)); if (mv < v) mv = v, m = c; }
            ir[j] = m; arv[j] = mv;
        }
        if (own[j] && i > 0) {
            int m = 0;
            double mv = fabs(s.e(hq_T(0) + i));
            for (int r = 1; r < i; r++) { const double v = fabs(s.e(hq_T(r) + i)); if (mv < v) mv = v, m = r; }
            ic[j] = m; acv[j] = mv;
        }
    }
    for (int iters = 0; iters < n * n * 30; iters++) {
        // pivot: first maximum over the row candidates 0 .. n-2, then the column candidates 1 .. n-1 (order in bits 8.., k, l below)
        QCand best; best.a = -1.; best.pk = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = ii[j];
            const bool ok = own[j] & (i < n - 1);
            const bool up = ok & (best.a < arv[j]);
            best.a = up ? arv[j] : best.a; best.pk = up ? ((i << 8) | (i << 4) | ir[j]) : best.pk;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = ii[j];
            const bool ok = own[j] & (i > 0);
            const bool up = ok & (best.a < acv[j]);
            best.a = up ? acv[j] : best.a; best.pk = up ? (((n - 1 + i) << 8) | (ic[j] << 4) | i) : best.pk;
        }
        best = hq_quad_best(best);
        const int k = (best.pk >> 4) & 15, l = best.pk & 15;
        if (!(best.a > eps)) break;          // |p| <= eps (uniform over the quad)
        // every operand of the rotation, at addresses that depend on k and l only
        const int Tk = hq_T(k), Tl = hq_T(l);
        double* Xp[3]; double* Yp[3]; double* VXp[3]; double* VYp[3];
        bool pair[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = ii[j];
            pair[j] = own[j] & (i != k) & (i != l);
            const int ox = i < k ? hq_T(i) + k : Tk + i, oy = i < l ? hq_T(i) + l : Tl + i;
            Xp[j] = &s.e(pair[j] ? ox : HQ_DUMMY); Yp[j] = &s.e(pair[j] ? oy : HQ_DUMMY);
            VXp[j] = &s.e(own[j] ? HQ_V + n * k + i : HQ_DUMMY); VYp[j] = &s.e(own[j] ? HQ_V + n * l + i : HQ_DUMMY);
        }
        double* Pp = &s.e(Tk + l);
        const double p = *Pp, wk = s.e(HQ_W + k), wl = s.e(HQ_W + l);
        double a0[3], b0[3], va[3], vb[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { a0[j] = *Xp[j]; b0[j] = *Yp[j]; va[j] = *VXp[j]; vb[j] = *VYp[j]; }
        const double y = (wl - wk) * 0.5;
        double t, sn, c;
        if (__all(jd_mid(__double2hiint(p)) & jd_mid(__double2hiint(y)))) jacobi_rotation<true>(p, y, &c, &sn, &t);     // uniform branch
        else jacobi_rotation<false>(p, y, &c, &sn, &t);
        *Pp = 0;
        s.e(HQ_W + k) = wk - t; s.e(HQ_W + l) = wl + t;      // the four lanes write the same values
        double xa[3], ya[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            xa[j] = a0[j] * c - b0[j] * sn;
            ya[j] = a0[j] * sn + b0[j] * c;
            *Xp[j] = xa[j]; *Yp[j] = ya[j];
            *VXp[j] = va[j] * c - vb[j] * sn;
            *VYp[j] = va[j] * sn + vb[j] * c;
        }
        // the candidates of the rows / columns other than k, l keep their (stale) indices; their VALUES follow the rotation
        QCand rk, ck, rl, cl;       // the four re-scans (first maxima; payload: scan position << 4 | position, so the smaller position wins ties)
        rk.a = ck.a = rl.a = cl.a = -1.; rk.pk = ck.pk = rl.pk = cl.pk = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = ii[j];
            const double ax = fabs(xa[j]), ay = fabs(ya[j]);
            arv[j] = (pair[j] & (i < k) & (ir[j] == k)) ? ax : arv[j];
            arv[j] = (pair[j] & (i < l) & (ir[j] == l)) ? ay : arv[j];
            acv[j] = (pair[j] & (i > k) & (ic[j] == k)) ? ax : acv[j];
            acv[j] = (pair[j] & (i > l) & (ic[j] == l)) ? ay : acv[j];
            // element i of row / column k is xa (zero at i = l: the pivot), of row / column l ya (zero at i = k)
            const double ek = i == l ? 0. : ax, el = i == k ? 0. : ay;
            const bool vk = own[j] & (i != k), vl = own[j] & (i != l);
            const bool u0 = vk & (i > k) & (rk.a < ek), u1 = vk & (i < k) & (ck.a < ek), u2 = vl & (i > l) & (rl.a < el), u3 = vl & (i < l) & (cl.a < el);
            const int tag = (i << 4) | i;
            rk.a = u0 ? ek : rk.a; rk.pk = u0 ? tag : rk.pk;
            ck.a = u1 ? ek : ck.a; ck.pk = u1 ? tag : ck.pk;
            rl.a = u2 ? el : rl.a; rl.pk = u2 ? tag : rl.pk;
            cl.a = u3 ? el : cl.a; cl.pk = u3 ? tag : cl.pk;
        }
        rk = hq_quad_best(rk); ck = hq_quad_best(ck); rl = hq_quad_best(rl); cl = hq_quad_best(cl);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = ii[j];
            const bool isk = own[j] & (i == k), isl = own[j] & (i == l);
            ir[j] = isk ? (rk.pk & 15) : (isl ? (rl.pk & 15) : ir[j]);
            arv[j] = isk ? rk.a : (isl ? rl.a : arv[j]);
            ic[j] = isk ? (ck.pk & 15) : (isl ? (cl.pk & 15) : ic[j]);
            acv[j] = isk ? ck.a : (isl ? cl.a : acv[j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the next rotation's reads come after these writes (one wave: LDS executes in order)
    }
    // sort the eigenvalues (and vectors) in descending order: selection sort of the serial code, by all four lanes in lockstep
    // (same reads, same writes)
    for (int k = 0; k < n - 1; k++) {
        int m = k;
        for (int i = k + 1; i < n; i++) if (s.e(HQ_W + m) < s.e(HQ_W + i)) m = i;
        if (k != m) {
            double tw = s.e(HQ_W + m); s.e(HQ_W + m) = s.e(HQ_W + k); s.e(HQ_W + k) = tw;
            for (int i = 0; i < n; i++) { double tv = s.e(HQ_V + n * m + i); s.e(HQ_V + n * m + i) = s.e(HQ_V + n * k + i); s.e(HQ_V + n * k + i) = tv; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// HomographyEstimatorCallback::runKernel on 4 correspondences by one quad (every lane computes the normalisation and L^T L)
__device__ int dlt4_quad(const float* M, const float* m, const QSlot s, const int q, double* H) {
    const int count = 4;
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) { cmx += m[2 * i]; cmy += m[2 * i + 1]; cMx += M[2 * i]; cMy += M[2 * i + 1]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[2 * i] - cmx); smy += fabs(m[2 * i + 1] - cmy);
        sMx += fabs(M[2 * i] - cMx); sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;      // uniform over the quad
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double LtL[45];  // upper triangle incl. diagonal, row-major; static indices only -> registers
#pragma unroll
    for (int e = 0; e < 45; e++) LtL[e] = 0;
#pragma unroll
    for (int i = 0; i < count; i++) {
        const double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        const double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        int e = 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++, e++) LtL[e] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    {
        int e = 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++, e++) {
                if ((e & 3) != q) continue;            // a quarter of the stores per lane
                if (k == j) s.e(HQ_W + j) = LtL[e];
                else s.e(hq_T(j) + k) = LtL[e];
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    jacobi9_quad(s, q);
    double H0[9];
    for (int i = 0; i < 9; i++) H0[i] = s.e(HQ_V + 72 + i);
    const double nrm[8] = {cmx, cmy, cMx, cMy, smx, smy, sMx, sMy};
    dlt_denormalise(H0, nrm, H);
    return 1;
}

__global__ __launch_bounds__(4 * HQ_HYPS) void hyp_quad_kernel(const HomoCall* calls, const RansacState* states, const int* sub_idx, double* Hc, int* valid, int lo,
                                                               int max_iters) {
#if MIS_CHAIN_PRIO
    __builtin_amdgcn_s_setprio(MIS_CHAIN_PRIO);
#endif
    extern __shared__ double sl[];
    const int b = blockIdx.y, t = threadIdx.x, h = t >> 2, q = t & 3;
    const RansacState st = states[b];
    if (st.mode != 2 || st.done) return;
    const int limit = min(st.n_sub, st.niters);  // hypotheses at or beyond niters can never be replayed
    const int k = lo + blockIdx.x * HQ_HYPS + h;
    if (k >= limit) return;                      // whole quads leave together
    const HomoCall c = calls[b];
    const int* id = sub_idx + ((size_t)b * max_iters + k) * 4;
    float ms1[8], ms2[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int qq = id[i];
        ms1[2 * i] = c.src[2 * qq]; ms1[2 * i + 1] = c.src[2 * qq + 1];
        ms2[2 * i] = c.dst[2 * qq]; ms2[2 * i + 1] = c.dst[2 * qq + 1];
    }
    double H[9];
    const QSlot s{sl + h};
    const int ok = dlt4_quad(ms1, ms2, s, q, H);
    if (q == 0) {
        if (ok) {
            double* o = Hc + ((size_t)b * max_iters + k) * 9;
#pragma unroll
            for (int i = 0; i < 9; i++) o[i] = H[i];
        }
        valid[(size_t)b * max_iters + k] = ok;
    }
}

// findInliers of every hypothesis of hyp_kernel: one wave per hypothesis, lanes over the points (the per-thread loop over all
// matches was a third of hyp_kernel's latency for a well-matched pair: 1500 points x 30 instructions behind the Jacobi solve)
__global__ __launch_bounds__(256) void hyp_count_kernel(const HomoCall* calls, const RansacState* states, const double* Hc, const int* valid, int* good, int lo,
                                                        int max_iters, float thr) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const RansacState st = states[b];
    if (st.mode != 2 || st.done) return;
    const int limit = min(st.n_sub, st.niters);
    const int k = lo + blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (k >= limit) return;      // wave-uniform
    const size_t e = (size_t)b * max_iters + k;
    int cnt = 0;
    if (valid[e]) {
        const HomoCall c = calls[b];
        float Hf[9];
#pragma unroll
        for (int i = 0; i < 9; i++) Hf[i] = (float)Hc[e * 9 + i];
        for (int i0 = 0; i0 < c.n; i0 += 64) {
            const int i = i0 + lane;
            const int f = i < c.n ? is_inlier(Hf, c.src[2 * i], c.src[2 * i + 1], c.dst[2 * i], c.dst[2 * i + 1], thr) : 0;
            cnt += __popcll(__ballot(f));
        }
    }
    if (lane == 0) good[e] = cnt;
}

// ---------------------------------------------------------------- scan_tail_kernel -------------
#ifdef MIS_TAIL_PROF
__device__ unsigned long long g_jac_prof[8];    // shader cycles: pivot search, math, rotation, index update (summed over rotations)
#ifdef MIS_JAC_PROF     // per-rotation section timers: they serialise the rotation (s_memtime waits for the LDS queue), use the coarse ones for totals
#define JP_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define JP_ADD(i, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_jac_prof[i], (b) - (a)); } while (0)
#else
#define JP_T(v)
#define JP_ADD(i, a, b)
#endif
__device__ unsigned long long g_tail_log[6 * 1024];
__device__ unsigned g_tail_log_n;
__device__ unsigned long long g_tail_prof[12];  // [8..11], part 4 launches: first entry (wall clock) + 1, last exit, longest workgroup, workgroups with work
//   // jacobi ticks, rotations, normal_eq ticks, LM iterations, dlt ticks, tail ticks, tails, max tail ticks
#define PROF_T0(v) unsigned long long v = wall_clock64()
#define PROF_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_tail_prof[i], wall_clock64() - (v)); } while (0)
#define PROF_INC(i, n) do { if (threadIdx.x == 0) atomicAdd(&g_tail_prof[i], (unsigned long long)(n)); } while (0)
#else
#define PROF_T0(v)
#define PROF_ADD(i, v)
#define PROF_INC(i, n)
#define JP_T(v)
#define JP_ADD(i, a, b)
#endif
#ifndef MIS_PS_PTS
#define MIS_PS_PTS 32
#endif
constexpr int PS_PTS = MIS_PS_PTS;           // points per stage of ordered_sums (a multiple of 8, at most 64)
constexpr int PS_TERMS = 42;                 // non-zero terms of a point: 6 accumulators with two (LM) + 30 with one, or 36 with one (L^T L)
constexpr int PS_PITCH = 43;                 // doubles between consecutive points of a stage: 86 dwords, so that 32 lanes (one point each) write 32 different bank pairs;
                                             // a point's 43rd double is 0.0: the second term of the accumulators that have one
constexpr int PS_STAGE = PS_PTS * PS_PITCH;  // doubles of one stage
constexpr size_t TAIL_DYN_LDS = sizeof(double) * 2 * PS_STAGE;     // dynamic LDS of the launches that run a DLT / LM refinement
struct TailShared {
    double A[81], V[81], W[9];
    double best[9], nrm[8];
    double lm[8 + 8 + 64 + 64 + 8 + 8 + 8 + 8];  // x, xd, A, Ap, v, d, D, scalars
    float Hf[9];
    int indR[9], indC[9];
    int np, go;
    int bad;                // = the number of the ordered_sums call whose chunk holds a non-finite record: those sums take the plain loop
    double chunk[TB * 10];  // per-point records of the current 256-point chunk of a sequential sum
    int staged;             // the launch carries ordered_sums' term stages (PS_PTS points, double-buffered) in its dynamic LDS (none in the
                            // launches that only replay or mask: 23 KB workgroups find a compute unit beside the composition's kernels, larger ones wait)
#ifdef MIS_TAIL_PROF
    int prof_rot, prof_lm;  // this tail's rotations / LM iterations (g_tail_log)
#endif
};
// The DLT (two passes for the normalisation, one for L^T L) and every normal-equations pass of the LM refinement (1 + up to 20)
// walk the same inlier set 256 points at a time, thread t taking point base + t.  From global memory each chunk would start with an
// exposed memory latency (1 - 3 us beside the composition's kernels); round 3 staged the whole set in 32 KB of LDS instead, which
// made the workgroup wait for room beside the composition's grids (a step moves by ~0.1 ms per 16 KB of the tails' LDS, round 4).
// Now a thread keeps its point of chunk 0 in registers for the whole tail and loads its point of chunk c + 1 while chunk c is being
// summed; the barriers inside a pass wait for LDS traffic only (lds_barrier), so the load stays in flight across them.
__device__ __forceinline__ float4 tail_load_point(const float* s1, const float* d1, int i) {
    const float2 dd = reinterpret_cast<const float2*>(d1)[i], ss = reinterpret_cast<const float2*>(s1)[i];
    return make_float4(dd.x, dd.y, ss.x, ss.y);
}
__device__ __forceinline__ void lds_barrier() { mis_lds_barrier(); }      // every wave's LDS writes are done and visible; global loads stay in flight (dev_math.h)

// Jacobi with the n independent plane rotations of a step spread over n lanes and the four
// index-table scans over four lanes; the arithmetic of every element is that of the serial loop.
// Runs on wave 0 only: the lanes of one wave execute LDS instructions in order, so a wavefront-scope
// fence (no s_barrier) is all the synchronisation the steps need; the other waves wait at the end.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// ---- round 3: a rotation without an LDS round trip on its dependency chain ----
// The wave's four DPP rows (16 lanes each) all run the rotation: lane (g, q) of row g does the serial loop's work for index q
// (its pair of A, its pair of V, W[q]) and keeps the serial algorithm's candidates of row q (indR[q] and the signed value
// A[q][indR[q]]) and of column q (indC[q], A[indC[q]][q]) in registers -- four identical copies, so nothing crosses the rows.
//  * After a rotation the new elements of rows / columns k and l ARE the values xa / ya the lanes have just computed (lane q
//    holds A[q|k] and A[q|l]): the four re-scans (indR[k], indC[k], indR[l], indC[l]) need no LDS read; DPP row g does one of them.
//  * "First maximum" = maximum, then the first lane that holds it: |value| is max-reduced by v_max_f64 over DPP moves (four steps
//    in a 16-lane row, one more for 32 lanes), every lane compares its own value with the maximum and s_ff1 of the ballot is the
//    serial loop's answer (lane order = scan order; candidate order = rows in lanes 0 .. 15, columns in lanes 16 .. 31).  An
//    arg-max that carries (value, order, payload) through every step -- this round's first form -- costs 64-bit compares, a
//    VALU -> SALU -> VALU mask round trip and four selects per step: 705 cycles for a wave (tools/micro/lat_bench.hip) against
//    904 for the rotation's divisions and square roots.
//  * W[k], W[l] are lane reads; the LDS copies of A and V are read before the arithmetic chain (they do not depend on it) and
//    written behind it: the lanes of one wave execute LDS instructions in order, the next rotation's reads see them.
// Values and results are those of the serial loop (and of rounds 2's and this round's earlier forms).
__device__ __forceinline__ double jc_readlane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double jc_max_step(double v) {      // max(v, v of the DPP source lane); lanes outside ROW_MASK keep v
    const int hi = __double2hiint(v), lo = __double2loint(v);
    // every lane is written when ROW_MASK is 0xf (old = 0: no copy of v on the chain); a partial step keeps v in the other rows
    const int ohi = ROW_MASK == 0xf ? 0 : hi, olo = ROW_MASK == 0xf ? 0 : lo;
    const double o = __hiloint2double(__builtin_amdgcn_update_dpp(ohi, hi, CTRL, ROW_MASK, 0xf, false), __builtin_amdgcn_update_dpp(olo, lo, CTRL, ROW_MASK, 0xf, false));
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(o));      // the keys are never NaN-signalling: no canonicalising v_max of each operand (fmax adds two)
    return r;
}
__device__ __forceinline__ double jc_row_max(double v) {       // every lane of a 16-lane DPP row ends with the row's maximum
    v = jc_max_step<0xB1>(v);      // quad_perm [1,0,3,2]
    v = jc_max_step<0x4E>(v);      // quad_perm [2,3,0,1]
    v = jc_max_step<0x141>(v);     // row_half_mirror
    v = jc_max_step<0x140>(v);     // row_mirror
    return v;
}

template <int n>
__device__ __forceinline__ void jacobi_eigen_coop(TailShared& S) {
    static_assert(n <= 15, "one DPP row per scan");
    double* A = S.A; double* V = S.V; double* W = S.W;
    const int t = threadIdx.x;
    PROF_T0(pj);
    if (t < 64) {
        JP_T(j_in);
        const double eps = DBL_EPSILON;
        const int q = t & 15, g = t >> 4;
        const bool live = q < n;
        const int qc = live ? q : 0;
        int i, m;
        double mv;
        double Wt = 0;               // W[q]: W[k], W[l] of a step are two lane reads, not an LDS round trip
        int ir = 0, ic = 0;          // indR[q], indC[q]
        double rv = 0, cv = 0;       // A[q][ir], A[ic][q] (signed)
        if (live) {
            if (g == 0) for (int j = 0; j < n; j++) V[q * n + j] = (j == q) ? 1. : 0.;
            Wt = A[(n + 1) * q];
            if (q < n - 1) {
                for (m = q + 1, mv = fabs(A[n * q + m]), i = q + 2; i < n; i++) {
                    double val = fabs(A[n * q + i]);
                    if (mv < val) mv = val, m = i;
                }
                ir = m; rv = A[n * q + m];
            }
            if (q > 0) {
                for (m = 0, mv = fabs(A[q]), i = 1; i < q; i++) {
                    double val = fabs(A[n * i + q]);
                    if (mv < val) mv = val, m = i;
                }
                ic = m; cv = A[n * m + q];
            }
        }
        wave_sync();
        // pivot: first maximum over the row candidates 0 .. n-2 (lanes 0 .. of DPP row 0), then the column candidates 1 .. n-1 (row 1)
        const bool isr = g == 0 && q < n - 1, isc = g == 1 && q >= 1 && live;
        int k, l;
        double p;
        auto pivot = [&]() {
            const double key = isr ? fabs(rv) : (isc ? fabs(cv) : -1.);
            const double m32 = jc_max_step<0x142, 0x2>(jc_row_max(key));       // row_bcast:15 into row 1: lane 31 holds the maximum of lanes 0 .. 31
            const double mx = jc_readlane(m32, 31);
            const unsigned long long eq = __ballot(key == mx);
            const int L = __builtin_ctzll(eq | (1ull << 63));
            const int pk = isr ? (q << 8) | ir : (ic << 8) | q;
            const int wk = __builtin_amdgcn_readlane(pk, L);
            k = wk >> 8; l = wk & 0xff;
            p = jc_readlane(isr ? rv : cv, L);
        };
        pivot();
        JP_T(j_loop);
        JP_ADD(0, j_in, j_loop);
        const int maxIters = n * n * 30;
        int iters = 0;
        if (n > 1) for (; iters < maxIters; iters++) {
            if (fabs(p) <= eps) break;
            JP_T(j1);
            // this lane's pair of A (q < k: (A[q][k], A[q][l]); k < q < l: (A[k][q], A[q][l]); l < q: (A[k][q], A[l][q])) and of V
            const bool aa = live & (q != k) & (q != l);
            double* X = A + (n * min(qc, k) + max(qc, k));
            double* Y = A + (n * min(qc, l) + max(qc, l));
            double* VX = V + n * k + qc;
            double* VY = V + n * l + qc;
            const double a0 = *X, b0 = *Y, va0 = *VX, vb0 = *VY;
            const double Wk = jc_readlane(Wt, k), Wl = jc_readlane(Wt, l);
            const double y = (Wl - Wk) * 0.5;
            double c, sn, tt;
            if (jd_mid(__builtin_amdgcn_readfirstlane(__double2hiint(p))) && jd_mid(__builtin_amdgcn_readfirstlane(__double2hiint(y))))      // scalar branch
                jacobi_rotation<true>(p, y, &c, &sn, &tt);
            else
                jacobi_rotation<false>(p, y, &c, &sn, &tt);
#ifdef MIS_JAC_PROF
            asm volatile("" :: "v"(c), "v"(sn), "v"(tt));
#endif
            JP_T(j2);
            // everything below is selects, not branches: a divergent region costs a VALU -> SALU -> exec round trip each
            const bool uk = q == k, ul = q == l;
            Wt = uk ? Wt - tt : (ul ? Wt + tt : Wt);
            const double xa = a0 * c - b0 * sn, ya = a0 * sn + b0 * c, xv = va0 * c - vb0 * sn, yv = va0 * sn + vb0 * c;
            if (g == 0 && live) {
                // lanes k and l hold the pivot element (lane k as its Y, lane l as its X: zero) and a diagonal element (never read again: W is in registers)
                *X = aa ? xa : (ul ? 0. : a0);
                *Y = aa ? ya : (uk ? 0. : b0);
                *VX = xv; *VY = yv;
            }
            // the candidates of the rows / columns other than k, l keep their (stale) indices; their VALUES follow the rotation
            // (q < k: row candidate in column k or l; k < q < l: column candidate in row k, row candidate in column l; l < q: column candidate in row k or l)
            // (bitwise & on the conditions: && chains come out as exec-mask branches)
            rv = (aa & (q < k) & (ir == k)) ? xa : rv;
            rv = (aa & (q < l) & (ir == l)) ? ya : rv;
            cv = (aa & (q > k) & (ic == k)) ? xa : cv;
            cv = (aa & (q > l) & (ic == l)) ? ya : cv;
            {
                // the scans: row k (DPP row 0), column k (1), row l (2), column l (3); element q of each is this lane's xa / ya, the
                // pivot element A[k][l] is zero.  An empty scan (column 0, row n - 1) leaves its candidate alone.
                const int idx = g < 2 ? k : l;
                const bool rowscan = (g & 1) == 0;
                const bool valid = (rowscan & (q > idx) & live) | (!rowscan & (q < idx));
                const double ev = g < 2 ? (ul ? 0. : xa) : (uk ? 0. : ya);
                const double key = valid ? fabs(ev) : -1.;
                const double mx = jc_row_max(key);
                const unsigned long long eq = __ballot(valid & (key == mx));
                int w[4];
                double wv[4];
                bool has[4];
#pragma unroll
                for (int gg = 0; gg < 4; gg++) {
                    const unsigned bits = (unsigned)(eq >> (16 * gg)) & 0xffffu;
                    has[gg] = bits != 0;
                    w[gg] = __builtin_ctz(bits | 0x10000u) & 15;
                    wv[gg] = jc_readlane(ev, 16 * gg + w[gg]);
                }
                const bool r_k = uk & has[0], c_k = uk & has[1], r_l = ul & has[2], c_l = ul & has[3];
                ir = r_k ? w[0] : (r_l ? w[2] : ir); rv = r_k ? wv[0] : (r_l ? wv[2] : rv);
                ic = c_k ? w[1] : (c_l ? w[3] : ic); cv = c_k ? wv[1] : (c_l ? wv[3] : cv);
            }
            JP_T(j3);
            pivot();
            wave_sync();
            JP_T(j4);
            JP_ADD(1, j1, j2); JP_ADD(2, j2, j3); JP_ADD(3, j3, j4);
        }
        PROF_INC(1, iters);
#ifdef MIS_TAIL_PROF
        if (t == 0) S.prof_rot += iters;
#endif
        JP_T(j_sort);
        if (t < n) W[t] = Wt;
        wave_sync();
        if (t == 0) {
            int kk;
            for (kk = 0; kk < n - 1; kk++) {
                m = kk;
                for (i = kk + 1; i < n; i++) if (W[m] < W[i]) m = i;
                if (kk != m) {
                    double tw = W[m]; W[m] = W[kk]; W[kk] = tw;
                    for (i = 0; i < n; i++) { double tv = V[n * m + i]; V[n * m + i] = V[n * kk + i]; V[n * kk + i] = tv; }
                }
            }
        }
        JP_T(j_end);
        JP_ADD(4, j_sort, j_end);
#ifdef MIS_JAC_PROF
        if (t == 0) atomicAdd(&g_jac_prof[5], 1ull);
#endif
    }
    __syncthreads();
    PROF_ADD(0, pj);
}


// ---- round 4: the ordered sums of the tails as a pipeline ----
// Every pass of the DLT / LM refinement over the inlier set is a set of 45 sums that must add their terms in point order (the CPU
// loop's order: the results are compared bit for bit).  Round 3 gave the 45 sums to 45 lanes of one wave, which per point read four
// operands from LDS, multiplied and added: 6 f64 operations per point at ~9.5 issue cycles each for a wave alone on its SIMD
// (tools/micro/lat_bench.hip) = ~60 cycles per point, 38 us per pass at 1500 points, nine to twelve passes per tail -- 45 % of the
// longest tails.  Only the additions have to be serial.  Now waves 1 .. 3 turn PS_PTS points at a time into the accumulators' terms
// -- a lane takes ONE point: its record (10 doubles) in five 16-byte reads, then a third of the terms, whose operand indices are
// compile-time constants of the wave's number (a first form with the indices in registers read every operand from LDS again: 16
// conflicting 8-byte reads per lane, no faster than round 3) --, double-buffered in S.prod, and lane a < 45 of wave 0 reads its
// term(s) of a point and adds: the additions -- two per point for six of the LM sums (acc += u0 v0; acc += u1 v1), one for the
// others and for L^T L (acc += (u0 v0 + u1 v1)) -- are all that is left on its chain (11 cycles per dependent add: 22 per point).
//  * Terms with a structural zero are not formed: a record's zero entry (LM: index 9, L^T L: index 3) makes 48 of the 90 products
//    +-0, and adding +-0 leaves an accumulator that is not -0 as it is (ours start at +0 and x + y is -0 only for x = y = -0).  That
//    holds for finite records; a chunk with a non-finite one (0 x inf = NaN in the CPU loop) takes round 3's plain loop instead
//    (S.bad).  A point's 42 terms are 344 bytes of LDS instead of 720.
struct PsEntry { int a0, b0, a1, b1; };
constexpr int ps_nib(unsigned long long tab, int i) { return (int)((tab >> (4 * i)) & 15ull); }
// operand indices of LM accumulator a: 36 entries of J^T J (upper triangle, row-major), 8 of J^T r, |r|^2; a per-point record is
// {a b ww c0 c1 c2 c3 e0 e1 0}: row 2q of J is (a b ww 0 0 0 c0 c1), row 2q + 1 is (0 0 0 a b ww c2 c3)
constexpr PsEntry lm_entry(int a) {
    const unsigned long long j0 = 0x43999210ull, j1 = 0x65210999ull;
    if (a == 44) return PsEntry{7, 7, 8, 8};
    if (a >= 36) return PsEntry{ps_nib(j0, a - 36), 7, ps_nib(j1, a - 36), 8};
    int ai = 0, aj = a;
    while (aj >= 8 - ai) { aj -= 8 - ai; ai++; }
    aj += ai;
    return PsEntry{ps_nib(j0, ai), ps_nib(j0, aj), ps_nib(j1, ai), ps_nib(j1, aj)};
}
// ... of L^T L entry a (upper triangle of 9 x 9, row-major); a per-point record is {X Y 1 0 -xX -xY -x -yX -yY -y}
constexpr PsEntry dlt_entry(int a) {
    const unsigned long long lx = 0x654333210ull, ly = 0x987210333ull;
    int j = 0, k = a;
    while (k >= 9 - j) { k -= 9 - j; j++; }
    k += j;
    return PsEntry{ps_nib(lx, j), ps_nib(lx, k), ps_nib(ly, j), ps_nib(ly, k)};
}
// ADDS: 2 = LM (two separately added terms), 1 = L^T L (the two products are added to each other first)
template <int ADDS> constexpr PsEntry ps_entry(int a) { return ADDS == 2 ? lm_entry(a) : dlt_entry(a); }
template <int ADDS> constexpr bool ps_z1(int a) { const PsEntry e = ps_entry<ADDS>(a); const int z = ADDS == 2 ? 9 : 3; return e.a0 == z || e.b0 == z; }
template <int ADDS> constexpr bool ps_z2(int a) { const PsEntry e = ps_entry<ADDS>(a); const int z = ADDS == 2 ? 9 : 3; return e.a1 == z || e.b1 == z; }
// terms accumulator a adds per point: 0 (both products structurally zero), 1, or 2 (LM only)
template <int ADDS> constexpr int ps_kind(int a) { return (ps_z1<ADDS>(a) && ps_z2<ADDS>(a)) ? 0 : (ADDS == 2 && !ps_z1<ADDS>(a) && !ps_z2<ADDS>(a)) ? 2 : 1; }
// where they sit in a point's block of PS_TERMS doubles: the two-term accumulators first (term pairs), then the one-term ones
template <int ADDS> constexpr int ps_off(int a) {
    int pairs = 0;
    for (int i = 0; i < 45; i++) pairs += ps_kind<ADDS>(i) == 2;
    int p = 0, o = 0;
    for (int i = 0; i < a; i++) { p += ps_kind<ADDS>(i) == 2; o += ps_kind<ADDS>(i) == 1; }
    return ps_kind<ADDS>(a) == 2 ? 2 * p : ps_kind<ADDS>(a) == 1 ? 2 * pairs + o : -1;
}
static_assert(ps_off<2>(44) == 10 && ps_kind<2>(44) == 2 && ps_kind<2>(3) == 0 && ps_off<2>(43) == 8 && ps_off<2>(41) == PS_TERMS - 1, "LM term layout");
static_assert(ps_kind<1>(44) == 1 && ps_off<1>(44) == 35 && ps_kind<1>(3) == 0, "L^T L term layout");
// the terms of accumulators A .. 44 whose offsets fall in wave W's third of the block (W = 0 .. 2; W = 3: accumulator 44 alone)
template <int ADDS, int W, int A> struct PsEmit {
    static __device__ __forceinline__ void put(const double (&r)[10], double* out) {
        constexpr PsEntry e = ps_entry<ADDS>(A);
        constexpr int kind = ps_kind<ADDS>(A), off = ps_off<ADDS>(A);
        constexpr bool mine = kind != 0 && (W == 3 ? A == 44 : off / (PS_TERMS / 3) == W);
        if constexpr (mine) {
            if constexpr (ADDS == 1) {
                if constexpr (ps_z2<1>(A)) out[off] = r[e.a0] * r[e.b0];
                else if constexpr (ps_z1<1>(A)) out[off] = r[e.a1] * r[e.b1];
                else out[off] = r[e.a0] * r[e.b0] + r[e.a1] * r[e.b1];
            } else if constexpr (kind == 2) {
                out[off] = r[e.a0] * r[e.b0];
                out[off + 1] = r[e.a1] * r[e.b1];
            } else {
                out[off] = ps_z2<2>(A) ? r[e.a0] * r[e.b0] : r[e.a1] * r[e.b1];
            }
        }
        PsEmit<ADDS, W, A + 1>::put(r, out);
    }
};
template <int ADDS, int W> struct PsEmit<ADDS, W, 45> { static __device__ __forceinline__ void put(const double (&)[10], double*) {} };

#if defined(MIS_TAIL_PROF) && defined(MIS_PS_PROF)
#define PS_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define PS_ADD(i, who, a, b) do { if (threadIdx.x == (who)) atomicAdd(&g_jac_prof[i], (b) - (a)); } while (0)
#else
#define PS_T(v)
#define PS_ADD(i, who, a, b)
#endif
// The sums of one chunk: cnt <= TB records in S.chunk, written by all threads, not yet fenced.  `all`: every accumulator; otherwise only
// the 45th (|r|^2 of an LM trial step).  kind / off: this lane's accumulator (lane t < 45 of wave 0; ps_kind / ps_off of t), ent:
// its packed operand indices for the plain loop.  Barriers wait for LDS only: the caller's load of the next chunk's point is in flight.
extern __shared__ double tail_dyn[];      // the term stages (TAIL_DYN_LDS bytes in the launches that have them)
template <int ADDS>
__device__ __forceinline__ void ordered_sums(TailShared& S, int cnt, bool all, int kind, int off, int ent, int& gen, double& acc) {
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), L = t & 63;
    lds_barrier();
    // a thread with a non-finite record stamped the flag with this call's number (never reset: the numbers do not repeat); a launch
    // without term stages (a four-correspondence problem met by a replay-only launch) takes the plain loop as well
    const bool bad = S.bad == gen || !S.staged;
    gen++;
    if (bad) {
        // round 3's loop: 45 lanes, operands from the records, every product formed
        if (all ? t < 45 : t == 44) {
            const int a0 = ent & 15, b0 = (ent >> 4) & 15, a1 = (ent >> 8) & 15, b1 = (ent >> 12) & 15;
            for (int q = 0; q < cnt; q++) {
                const double* r = S.chunk + 10 * q;
                if (ADDS == 2) { acc += r[a0] * r[b0]; acc += r[a1] * r[b1]; }
                else acc += r[a0] * r[b0] + r[a1] * r[b1];
            }
        }
        lds_barrier();
        return;
    }
    const int nsub = (cnt + PS_PTS - 1) / PS_PTS;
    const bool accumulate = (all ? t < 45 : t == 44) && kind != 0;
    const int off2 = kind == 2 ? off + 1 : PS_TERMS;
    auto produce = [&](int sb) {
        const int p = sb * PS_PTS + L;
        if (wave == 0 || L >= PS_PTS || p >= cnt) return;
        double r[10];
        const double2* rp = reinterpret_cast<const double2*>(S.chunk + 10 * p);
#pragma unroll
        for (int i = 0; i < 5; i++) { const double2 v = rp[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
        double* out = tail_dyn + (sb & 1) * PS_STAGE + L * PS_PITCH;
        if (!all) { if (wave == 1) PsEmit<ADDS, 3, 0>::put(r, out); }
        else if (wave == 1) PsEmit<ADDS, 0, 0>::put(r, out);
        else if (wave == 2) PsEmit<ADDS, 1, 0>::put(r, out);
        else PsEmit<ADDS, 2, 0>::put(r, out);
    };
    PS_T(ps0);
#if defined(MIS_TAIL_PROF) && defined(MIS_PS_PROF)
    unsigned long long ps_acc = 0, ps_prod = 0, ps_bar = 0;
#endif
    produce(0);
    lds_barrier();
    for (int sb = 0; sb < nsub; sb++) {
        PS_T(ps1);
        if (sb + 1 < nsub) produce(sb + 1);
        PS_T(ps2);
#if defined(MIS_TAIL_PROF) && defined(MIS_PS_PROF)
        ps_prod += ps2 - ps1;
#endif
        if (accumulate) {
            // a lane's second term: the slot behind its first (two-term accumulators) or the point's zero (x + 0 = x: no select on the chain)
            const double* P = tail_dyn + (sb & 1) * PS_STAGE;
            const int n_s = min(PS_PTS, cnt - sb * PS_PTS);
            if (n_s == PS_PTS) {
#pragma unroll
                for (int u0 = 0; u0 < PS_PTS; u0 += 8) {      // eight points in flight
                    double v0[8], v1[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { v0[u] = P[(u0 + u) * PS_PITCH + off]; if (ADDS == 2) v1[u] = P[(u0 + u) * PS_PITCH + off2]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc += v0[u]; if (ADDS == 2) acc += v1[u]; }
                }
            } else {
                for (int u = 0; u < n_s; u++) {
                    acc += P[u * PS_PITCH + off];
                    if (ADDS == 2) acc += P[u * PS_PITCH + off2];
                }
            }
        }
#if defined(MIS_TAIL_PROF) && defined(MIS_PS_PROF)
        asm volatile("" :: "v"(acc));
#endif
        PS_T(ps3);
        lds_barrier();
#if defined(MIS_TAIL_PROF) && defined(MIS_PS_PROF)
        ps_acc += ps3 - ps2;
        ps_bar += __builtin_readcyclecounter() - ps3;
#endif
    }
    PS_T(ps4);
#if defined(MIS_TAIL_PROF) && defined(MIS_PS_PROF)
    PS_ADD(0, 44, 0ull, ps_acc);
    PS_ADD(1, 64, 0ull, ps_prod);
    PS_ADD(2, 44, ps0, ps4);
    PS_ADD(3, 44, 0ull, (unsigned long long)nsub);
    PS_ADD(4, 44, 0ull, ps_bar);
    PS_ADD(5, 64, 0ull, ps_bar);
#endif
}
// a record's entries are all finite (their sum of magnitudes is: an overflowing sum only sends the chunk to the plain loop)
__device__ __forceinline__ bool ps_record_bad(const double* r) {
    double m = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) m += fabs(r[i]);
    return !(m <= DBL_MAX);
}

// LM callback of the homography refinement (fundam.cpp HomographyRefineCallback)
__device__ __forceinline__ void lm_point(const double* h, double Mx, double My, double* ww, double* xi, double* yi) {
    double w = (h[6] * Mx + h[7] * My) + 1.;
    w = fabs(w) > DBL_EPSILON ? 1. / w : 0;
    *ww = w;
    *xi = ((h[0] * Mx + h[1] * My) + h[2]) * w;
    *yi = ((h[3] * Mx + h[4] * My) + h[5]) * w;
}

// HomographyEstimatorCallback::runKernel on np points (s1 -> d1) by the whole workgroup; S.best is
// overwritten unless the configuration is degenerate.  Sums keep the sequential order.
__device__ void dlt_coop(TailShared& S, const float* s1, const float* d1, int np, float4 pt0, double* rec) {
    const int t = threadIdx.x;
    {
        // the four centroids and the four mean absolute deviations: sequential sums over the points by four accumulator lanes.  All
        // threads turn their point into the four double terms of the pass (the conversion, |x - c|) in LDS; the lanes only add.
        double* stage = S.chunk;      // 4 doubles per point: d.x d.y s.x s.y
        for (int pass = 0; pass < 2; pass++) {
            double acc = 0;
            const double c0 = pass ? S.nrm[0] : 0., c1 = pass ? S.nrm[1] : 0., c2 = pass ? S.nrm[2] : 0., c3 = pass ? S.nrm[3] : 0.;
            float4 cur = pt0;
            for (int base = 0; base < np; base += TB) {
                const int i = base + t, cnt = min(TB, np - base);
                float4 nxt = cur;
                if (i + TB < np) nxt = tail_load_point(s1, d1, i + TB);
                if (i < np) {
                    double* o = stage + 4 * t;
                    if (pass == 0) { o[0] = cur.x; o[1] = cur.y; o[2] = cur.z; o[3] = cur.w; }
                    else { o[0] = fabs(cur.x - c0); o[1] = fabs(cur.y - c1); o[2] = fabs(cur.z - c2); o[3] = fabs(cur.w - c3); }
                }
                lds_barrier();
                if (t < 4) {
                    const double* q = stage + t;
                    int u = 0;
#pragma unroll 1
                    for (; u + 8 <= cnt; u += 8) {
                        double v[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) v[k] = q[4 * (u + k)];
#pragma unroll
                        for (int k = 0; k < 8; k++) acc += v[k];
                    }
                    for (; u < cnt; u++) acc += q[4 * u];
                }
                lds_barrier();
                cur = nxt;
            }
            if (t < 4) { if (pass == 0) S.nrm[t] = acc / np; else S.nrm[4 + t] = acc; }      // cmx cmy cMx cMy, then smx smy sMx sMy
            __syncthreads();
        }
    }
    const bool degenerate = fabs(S.nrm[4]) < DBL_EPSILON || fabs(S.nrm[5]) < DBL_EPSILON || fabs(S.nrm[6]) < DBL_EPSILON || fabs(S.nrm[7]) < DBL_EPSILON;
    __syncthreads();
    if (t == 0) S.go = !degenerate;
    if (degenerate) { __syncthreads(); return; }
    if (t < 4) S.nrm[4 + t] = np / S.nrm[4 + t];
    __syncthreads();
    {
        // L^T L: 256 points at a time -- every thread produces the record of one point into LDS, then the 45 sums take their terms
        // in point order (the sequential order of the CPU loop) through ordered_sums
        const double cmx = S.nrm[0], cmy = S.nrm[1], cMx = S.nrm[2], cMy = S.nrm[3], smx = S.nrm[4], smy = S.nrm[5], sMx = S.nrm[6], sMy = S.nrm[7];
        int j = 0, k = t;  // t-th entry of the upper triangle, row-major (t < 45)
        while (t < 45 && k >= 9 - j) { k -= 9 - j; j++; }
        k += j;
        const int ta = t < 45 ? t : 0;
        const PsEntry e = dlt_entry(ta);
        const int kind = ps_kind<1>(ta), off = ps_off<1>(ta), ent = e.a0 | (e.b0 << 4) | (e.a1 << 8) | (e.b1 << 12);
        int gen = 1;
        double acc = 0;
        float4 cur = pt0;
        for (int base = 0; base < np; base += TB) {
            const int i = base + t, cnt = min(TB, np - base);
            float4 nxt = cur;
            if (i + TB < np) nxt = tail_load_point(s1, d1, i + TB);
            if (i < np) {
                double x = (cur.x - cmx) * smx, y = (cur.y - cmy) * smy;
                double X = (cur.z - cMx) * sMx, Y = (cur.w - cMy) * sMy;
                const double rv[10] = {X, Y, 1., 0., -x * X, -x * Y, -x, -y * X, -y * Y, -y};      // X Y 1 0 -xX -xY -x -yX -yY -y
                double* r = S.chunk + 10 * t;
#pragma unroll
                for (int q = 0; q < 10; q++) r[q] = rv[q];
                if (ps_record_bad(rv)) S.bad = gen;
            }
            ordered_sums<1>(S, cnt, true, kind, off, ent, gen, acc);      // acc += r[xj] r[xk] + r[yj] r[yk], in point order
            cur = nxt;
        }
        if (t < 45) S.A[j * 9 + k] = acc;
    }
    __syncthreads();
    if (t < 81) { int j = t / 9, k = t % 9; if (k < j) S.A[j * 9 + k] = S.A[k * 9 + j]; }
    __syncthreads();
    jacobi_eigen_coop<9>(S);
    if (t == 0) dlt_denormalise(S.V + 72, S.nrm, S.best);
    __syncthreads();
}

// createLMSolver(HomographyRefineCallback, 10)->run(H8): OpenCV <= 4.5 LMSolverImpl::run
__device__ void lm_refine_coop(TailShared& S, const float* s1, const float* d1, int np, float4 pt0, double* rec) {
    const int t = threadIdx.x;
    double* x = S.lm;            double* xd = x + 8;   double* A = xd + 8;  double* Ap = A + 64;
    double* v = Ap + 64;         double* d = v + 8;    double* D = d + 8;   double* sc = D + 8;  // S, Sd, rmax, accepted, lambda, lc, need_invert, nu
    int ai = 0, aj = t;
    while (t < 36 && aj >= 8 - ai) { aj -= 8 - ai; ai++; }
    aj += ai;
    // this lane's accumulator in ordered_sums (lane t < 45 of wave 0)
    const int ta = t < 45 ? t : 0;
    const PsEntry le = lm_entry(ta);
    const int kind = ps_kind<2>(ta), off = ps_off<2>(ta), ent = le.a0 | (le.b0 << 4) | (le.a1 << 8) | (le.b1 << 12);
    int gen = 1 << 20;
    auto normal_eq = [&](const double* h, bool with_J) {
        // the sequential sums, all of one form, acc += r[a0] r[b0]; acc += r[a1] r[b1] per point: 36 entries of J^T J (lane t < 36 of
        // wave 0), 8 of J^T r (36 .. 43), |r|^2 (44).  |r|_inf ("if (fabs(e) > mx) mx = fabs(e)" is a maximum: order-free) is kept by
        // the threads that produce the points' records and reduced at the end.
        double acc = 0, mx = 0;
        PROF_T0(pn);
        float4 cur = pt0;
        for (int base = 0; base < np; base += TB) {
            const int p = base + t, cnt = min(TB, np - base);
            float4 nxt = cur;
            if (p + TB < np) nxt = tail_load_point(s1, d1, p + TB);
            if (p < np) {
                double Mx = (double)cur.z, My = (double)cur.w, ww, xi, yi;
                lm_point(h, Mx, My, &ww, &xi, &yi);
                const double e0 = xi - (double)cur.x, e1 = yi - (double)cur.y;
                const double rv[10] = {Mx * ww, My * ww, ww, -Mx * ww * xi, -My * ww * xi, -Mx * ww * yi, -My * ww * yi, e0, e1, 0.};      // a b ww c0 c1 c2 c3 e0 e1 0
                double* r = S.chunk + 10 * t;
#pragma unroll
                for (int i = 0; i < 10; i++) r[i] = rv[i];
                asm("v_max_f64 %0, %1, |%2|" : "=v"(mx) : "v"(mx), "v"(e0));
                asm("v_max_f64 %0, %1, |%2|" : "=v"(mx) : "v"(mx), "v"(e1));
                if (ps_record_bad(rv)) S.bad = gen;
            }
            ordered_sums<2>(S, cnt, with_J, kind, off, ent, gen, acc);
            cur = nxt;
        }
        if (with_J && t < 36) { A[ai * 8 + aj] = acc; A[aj * 8 + ai] = acc; }
        else if (with_J && t >= 36 && t < 44) v[t - 36] = acc;
        else if (t == 44) sc[with_J ? 0 : 1] = acc;
        if (with_J) {
            // |r|_inf over all points: per-thread maxima -> 16 partial maxima -> one
            S.chunk[t] = mx;
            __syncthreads();
            if (t < 16) {
                double m = 0;
                for (int i = 0; i < TB / 16; i++) { const double o = S.chunk[t * (TB / 16) + i]; asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o)); }
                S.chunk[TB + t] = m;
            }
            __syncthreads();
            if (t == 0) {
                double m = 0;
                for (int i = 0; i < 16; i++) { const double o = S.chunk[TB + i]; asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o)); }
                sc[2] = m;
            }
        }
        __syncthreads();
        PROF_ADD(2, pn);
#ifdef MIS_TAIL_PROF
        if (t == 0) { atomicAdd(&g_jac_prof[6], (unsigned long long)np); atomicAdd(&g_jac_prof[7], 1ull); }
#endif
    };
    if (t < 8) x[t] = S.best[t];
    __syncthreads();
    normal_eq(x, true);
    if (t < 8) D[t] = A[t * 8 + t];
    if (t == 0) { sc[4] = 1; sc[5] = 0.75; }  // lambda, lc
    __syncthreads();
    for (int iter = 0;;) {
        if (t < 64) S.A[t] = (t / 8 == t % 8) ? A[t] + sc[4] * D[t / 8] : A[t];
        __syncthreads();
        jacobi_eigen_coop<8>(S);
        if (t == 0) {
            // solve(Ap, v, d, DECOMP_EIG): Jacobi + SVBkSb back substitution
            double thrw = 0;
            for (int i = 0; i < 8; i++) thrw += S.W[i];
            thrw *= DBL_EPSILON * 2;
            for (int j = 0; j < 8; j++) d[j] = 0;
            for (int i = 0; i < 8; i++) {
                double wi = S.W[i];
                if (fabs(wi) <= thrw) continue;
                wi = 1 / wi;
                double s = 0;
                for (int j = 0; j < 8; j++) s += S.V[i * 8 + j] * v[j];
                s *= wi;
                for (int j = 0; j < 8; j++) d[j] = d[j] + s * S.V[i * 8 + j];
            }
            for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
        }
        __syncthreads();
        normal_eq(xd, false);  // Sd
        if (t == 0) {
            double Sv = sc[0], Sd = sc[1], lambda = sc[4], lc = sc[5];
            double temp_d[8], dS = 0;
            for (int i = 0; i < 8; i++) {
                double s = 0;
                for (int j = 0; j < 8; j++) s += A[i * 8 + j] * d[j];
                temp_d[i] = s * -1. + v[i] * 2.;
            }
            for (int i = 0; i < 8; i++) dS += d[i] * temp_d[i];
            double R = (Sv - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
            sc[6] = 0;
            if (R > 0.75) {
                lambda *= 0.5;
                if (lambda < lc) lambda = 0;
            } else if (R < 0.25) {
                double tt = 0;
                for (int i = 0; i < 8; i++) tt += d[i] * v[i];
                double nu = (Sd - Sv) / (fabs(tt) > DBL_EPSILON ? tt : 1) + 2;
                nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
                if (lambda == 0) sc[6] = 1;  // needs invert(A): done by the whole workgroup below
                else lambda *= nu;
                sc[7] = nu;
            }
            sc[4] = lambda; sc[5] = lc;
        }
        __syncthreads();
        if (sc[6] != 0.) {
            // invert(A, Ap, DECOMP_EIG) -> lambda = lc = 1 / max |diag|, nu halved
            if (t < 64) S.A[t] = A[t];
            __syncthreads();
            jacobi_eigen_coop<8>(S);
            if (t == 0) {
                double thrw = 0;
                for (int i = 0; i < 8; i++) thrw += S.W[i];
                thrw *= DBL_EPSILON * 2;
                for (int e = 0; e < 64; e++) Ap[e] = 0;
                for (int i = 0; i < 8; i++) {
                    double wi = S.W[i];
                    if (fabs(wi) <= thrw) continue;
                    wi = 1 / wi;
                    for (int r = 0; r < 8; r++)
                        for (int c = 0; c < 8; c++) Ap[r * 8 + c] = Ap[r * 8 + c] + S.V[i * 8 + r] * (S.V[i * 8 + c] * wi);
                }
                double maxval = DBL_EPSILON;
                for (int i = 0; i < 8; i++) { double a = fabs(Ap[i * 8 + i]); if (a > maxval) maxval = a; }
                double lambda = 1. / maxval, nu = sc[7] * 0.5;
                sc[5] = lambda;
                sc[4] = lambda * nu;
            }
            __syncthreads();
        }
        if (t == 0) {
            double Sv = sc[0], Sd = sc[1];
            sc[3] = Sd < Sv ? 1. : 0.;
            if (Sd < Sv) { sc[0] = Sd; for (int i = 0; i < 8; i++) x[i] = xd[i]; }
        }
        __syncthreads();
        if (sc[3] != 0.) normal_eq(x, true);
        iter++;
        PROF_INC(3, 1);
#ifdef MIS_TAIL_PROF
        if (t == 0) S.prof_lm++;
#endif
        double dmax = 0;
        for (int i = 0; i < 8; i++) { double a = fabs(d[i]); if (a > dmax) dmax = a; }
        bool proceed = iter < 10 && dmax >= (double)FLT_EPSILON && sc[2] >= (double)FLT_EPSILON;
        __syncthreads();
        if (!proceed) break;
    }
    if (t < 8) S.best[t] = x[t];
    __syncthreads();
}

// part 0: replay + tail (the finished problems' mask, DLT on the inliers, LM refinement) in one launch;
// part 1: replay only -- a problem that ends here is marked tail_pending; part 2: the tail of the pending problems whose
// fin equals `want`.  The split lets the tails of the problems that end in RANSAC phase 0 (latency bound, ~2 ms) run on
// another stream while phase 1 of the others -- which only needs the replay's verdict -- goes on.
#ifndef MIS_TAIL_WAVES
#define MIS_TAIL_WAVES 3      // <= 168 registers: a 200-register workgroup waits longer for room beside the composition's grids (6.65 vs 6.9 ms per step; 4 waves = 128 registers spill into the Jacobi loop)
#endif
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(MIS_TAIL_WAVES, 8))) void scan_tail_kernel(const HomoCall* calls, RansacState* states, const double* Hc, const int* valid, const int* good,
                                                       float* scr_all, double* rec_all, HomoResult* results, int lo, int hi, int max_iters,
                                                       double confidence, float thr, int* fin, int part, int want, int staged) {
#if MIS_CHAIN_PRIO
    __builtin_amdgcn_s_setprio(MIS_CHAIN_PRIO);      // a latency-bound chain beside the composition's bandwidth-bound kernels: its few waves issue first
#endif
#ifdef MIS_TAIL_PROF
    const unsigned long long wg_in = wall_clock64();
    if (part == 4 && want == 0 && threadIdx.x == 0) atomicMin(&g_tail_prof[8], wg_in);
#endif
    __shared__ TailShared S;
    __shared__ int s_done_now;
    if (threadIdx.x == 0) { S.staged = staged; S.bad = 0; }
    if (staged)
        for (int i = threadIdx.x; i < 2 * PS_PTS; i += TB) tail_dyn[i * PS_PITCH + PS_TERMS] = 0.;      // (stage 1 follows stage 0 at PS_PTS * PS_PITCH)
    __syncthreads();
    __shared__ int wcnt[TB / 64];
    __shared__ int s_base;
    const int b = blockIdx.x, t = threadIdx.x;
    const HomoCall c = calls[b];
    RansacState* st = states + b;
    HomoResult* res = results + b;
    const int n = c.n;
    float* s1 = scr_all + 4 * c.pt_off;
    float* d1 = s1 + 2 * (size_t)(n > 0 ? n : 0);
    double* rec = rec_all + 10 * c.pt_off;
    if (part >= 2) {
        // part 2 runs the whole tail of a pending problem (flag 1 -> 0); parts 3 / 4 split it: mask + compaction (1 -> 2), DLT + LM (2 -> 0)
        if (st->tail_pending != (part == 4 ? 2 : 1) || fin[b] != want) return;   // uniform
        __syncthreads();                                    // every thread has read the flag before it changes
        if (part == 2 && t == 0) st->tail_pending = 0;
    } else {
    if (st->done) return;  // finished in an earlier phase (uniform)
    const int mode = st->mode;
    if (t == 0) s_done_now = 0;
    __syncthreads();
    if (mode == 0) {
        for (int i = t; c.mask && c.active && i < n; i += TB) c.mask[i] = 0;
        if (t == 0) { res->ok = 0; res->iters = 0; res->ninl = 0; st->done = 1; fin[b] = lo == 0 ? 0 : 1; }
        return;
    }
    if (mode == 1) {
        // exactly four correspondences: runKernel directly, mask all ones, no refinement
        if (t == 0) for (int i = 0; i < 9; i++) S.best[i] = 0;
        __syncthreads();
        dlt_coop(S, c.src, c.dst, 4, t < 4 ? tail_load_point(c.src, c.dst, t) : make_float4(0.f, 0.f, 0.f, 0.f), rec);
        const int ok = S.go;
        for (int i = t; c.mask && i < n; i += TB) c.mask[i] = ok ? 1 : 0;
        if (t == 0) { res->ok = ok; res->iters = 0; res->ninl = ok ? 4 : 0; if (ok) for (int i = 0; i < 9; i++) res->H[i] = S.best[i]; st->done = 1; fin[b] = lo == 0 ? 0 : 1; }
        return;
    }
    // ---- replay of RANSACPointSetRegistrator::run over hypotheses [lo, hi) ----
    // The loop only changes state at a hypothesis that beats every earlier one (a new maximum of `good`), so the workgroup stages
    // good[] (-1 for a degenerate sample) in LDS with the maximum of every run of `seg` entries beside it, and the replaying thread
    // steps over the runs that hold no new maximum.  A serial walk over global memory paid a load latency per hypothesis: 0.43 ms
    // for the 1872 hypotheses of phase 1, on the matcher's critical path.
    {
#ifdef MIS_TAIL_PROF
        const unsigned long long rp0 = wall_clock64();
#endif
        // (niters never grows: nothing at or beyond the current limit is visited -- or has been computed)
        const int nsub = st->n_sub, kmax = min(hi, nsub), cnt = max(min(kmax - lo, st->niters - st->iter), 0);
        int* gl = reinterpret_cast<int*>(S.chunk);            // cnt entries (<= 4096), then the run maxima at + 4096
        int* segmax = gl + 4096;
        const bool staged = cnt <= 4096;
        const int seg = (cnt + TB - 1) / TB > 0 ? (cnt + TB - 1) / TB : 1;
        if (staged) {
            for (int i = t; i < cnt; i += TB) gl[i] = valid[(size_t)b * max_iters + lo + i] ? good[(size_t)b * max_iters + lo + i] : -1;
            __syncthreads();
            int mx = -1;
            for (int i = t * seg; i < min((t + 1) * seg, cnt); i++) mx = max(mx, gl[i]);
            segmax[t] = mx;
            __syncthreads();
        }
        if (t == 0) {
            int iter = st->iter, niters = st->niters, max_good = st->max_good, best_k = st->best_k;
            int k = lo;
            while (k < kmax && iter < niters) {
                const int off = k - lo;
                if (staged && off % seg == 0 && segmax[off / seg] <= (max_good > 3 ? max_good : 3)) {
                    // nothing in this run changes the state: the loop walks to its end, or to the iteration limit
                    const int adv = min(min(seg, kmax - k), niters - iter);
                    k += adv; iter += adv;
                    continue;
                }
                iter++;
                const int g = staged ? gl[off] : (valid[(size_t)b * max_iters + k] ? good[(size_t)b * max_iters + k] : -1);
                if (g >= 0 && g > (max_good > 3 ? max_good : 3)) {
                    best_k = k; max_good = g;
                    niters = ransac_update_num_iters(confidence, (double)(n - g) / n, niters);
                }
                k++;
            }
            st->iter = iter; st->niters = niters; st->max_good = max_good; st->best_k = best_k;
            // the loop ends when iter reaches niters, when getSubset failed (subsets exhausted) or at maxIters
            if (iter >= niters || (k >= nsub && st->draw_fail) || hi >= max_iters) { s_done_now = 1; st->done = 1; fin[b] = lo == 0 ? 0 : 1; }
#ifdef MIS_TAIL_PROF
            if (lo > 0) { const unsigned long long d = wall_clock64() - rp0; atomicAdd(&g_hyp_prof[6], d); atomicMax(&g_hyp_prof[7], d); }
#endif
        }
    }
    __syncthreads();
    if (!s_done_now) return;
    if (part == 1) {
        if (t == 0) st->tail_pending = 1;
        return;
    }
    }   // part != 2
    const int result = st->max_good > 0;
    if (part != 4 && t == 0) { res->iters = st->iter; res->ok = result; res->ninl = 0; }
    if (!result) {
        for (int i = t; c.mask && i < n; i += TB) c.mask[i] = 0;
        if (t == 0) st->tail_pending = 0;
        return;
    }
    // best model -> mask, ordered compaction of the inliers (compressElems)
    if (t == 0) {
        const double* hb = Hc + ((size_t)b * max_iters + st->best_k) * 9;
        for (int i = 0; i < 9; i++) { S.best[i] = hb[i]; S.Hf[i] = (float)hb[i]; }
        s_base = part == 4 ? res->ninl : 0;
    }
    __syncthreads();
    if (part != 4)
    for (int i0 = 0; i0 < n; i0 += TB) {
        int i = i0 + t, f = 0;
        if (i < n) f = is_inlier(S.Hf, c.src[2 * i], c.src[2 * i + 1], c.dst[2 * i], c.dst[2 * i + 1], thr);
        if (c.mask && i < n) c.mask[i] = (uint8_t)f;
        unsigned long long bal = __ballot(f);
        int within = __popcll(bal & ((1ull << (t & 63)) - 1ull));
        if ((t & 63) == 0) wcnt[t >> 6] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int k = 0; k < (t >> 6); k++) off += wcnt[k];
        if (f) {
            s1[2 * (off + within)] = c.src[2 * i]; s1[2 * (off + within) + 1] = c.src[2 * i + 1];
            d1[2 * (off + within)] = c.dst[2 * i]; d1[2 * (off + within) + 1] = c.dst[2 * i + 1];
        }
        __syncthreads();
        if (t == 0) { int s = 0; for (int k = 0; k < TB / 64; k++) s += wcnt[k]; s_base += s; }
        __syncthreads();
    }
    const int np = s_base;
    if (part == 3) {   // the inlier set is final here; DLT + LM on it (part 4) only changes H
        if (t == 0) { res->ninl = np; for (int i = 0; i < 9; i++) res->H[i] = S.best[i]; st->tail_pending = 2; }
        return;
    }
    PROF_T0(pt);
#ifdef MIS_TAIL_PROF
    if (t == 0) { S.prof_rot = 0; S.prof_lm = 0; }
#endif
    if (np > 0) {
        // this thread's point of every pass's first chunk: loaded once (the compaction above wrote it: its stores are fenced by the loop's barriers)
        const float4 pt0 = t < np ? tail_load_point(s1, d1, t) : make_float4(0.f, 0.f, 0.f, 0.f);
        dlt_coop(S, s1, d1, np, pt0, rec);   // runKernel on all inliers (keeps the RANSAC model if degenerate)
        PROF_ADD(4, pt);
        lm_refine_coop(S, s1, d1, np, pt0, rec);
    }
    PROF_ADD(5, pt);
    PROF_INC(6, 1);
#ifdef MIS_TAIL_PROF
    if (t == 0) atomicMax(&g_tail_prof[7], wall_clock64() - pt);
    if (t == 0) {      // one line per tail: points, LM iterations, rotations, ticks, launch kind, entry tick
        const unsigned k = atomicAdd(&g_tail_log_n, 1u);
        if (k < 1024) { unsigned long long* e = g_tail_log + 6 * k; e[0] = np; e[1] = S.prof_lm; e[2] = S.prof_rot; e[3] = wall_clock64() - pt; e[4] = part * 10 + want; e[5] = pt; }
    }
    if (part == 4 && want == 0 && t == 0) { const unsigned long long o = wall_clock64(); atomicMax(&g_tail_prof[9], o); atomicMax(&g_tail_prof[10], o - wg_in); atomicAdd(&g_tail_prof[11], 1ull); }
#endif
    if (t == 0) { for (int i = 0; i < 9; i++) res->H[i] = S.best[i]; res->ninl = np; if (part == 4) st->tail_pending = 0; }
}

// ---------------------------------------------------------------- host side --------------------
struct RngTable {
    unsigned* U = nullptr;
    unsigned long long state_T = 0;
};
std::mutex g_rng_mutex;
RngTable g_rng[64];

int rng_table(MisContext* ctx, RngTable* out) {
    std::lock_guard<std::mutex> lock(g_rng_mutex);
    RngTable& r = g_rng[ctx->device & 63];
    if (!r.U) {
        std::vector<unsigned> h(RNG_TABLE);
        unsigned long long st = ~0ull;  // RNG rng((uint64)-1)
        for (int i = 0; i < RNG_TABLE; i++) {
            st = (unsigned long long)(unsigned)st * 4164903690u + (unsigned)(st >> 32);
            h[i] = (unsigned)st;
        }
        MIS_HIP(ctx, hipMalloc((void**)&r.U, sizeof(unsigned) * RNG_TABLE));
        MIS_HIP(ctx, hipMemcpy(r.U, h.data(), sizeof(unsigned) * RNG_TABLE, hipMemcpyHostToDevice));
        r.state_T = st;
    }
    *out = r;
    return MIS_OK;
}

}  // namespace

int homo_batch_reserve(MisContext* ctx, HomoBatch* b, int count, long long points, int max_iters) {
    count = std::max(count, 1); points = std::max(points, 1ll); max_iters = std::max(max_iters, 1);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += mis_align_up(bytes, 256); return o; };
    size_t o_calls = carve(sizeof(HomoCall) * count), o_res = carve(sizeof(HomoResult) * count), o_state = carve(sizeof(RansacState) * count);
    size_t o_sub = carve(sizeof(int) * 4 * (size_t)count * max_iters), o_hc = carve(sizeof(double) * 9 * (size_t)count * max_iters);
    size_t o_valid = carve(sizeof(int) * (size_t)count * max_iters), o_good = carve(sizeof(int) * (size_t)count * max_iters);
    size_t o_scr = carve(sizeof(float) * 4 * (size_t)points), o_rec = carve(sizeof(double) * 10 * (size_t)points);
    size_t o_dn = carve(256), o_di = carve(sizeof(int) * 4 * (size_t)count * DRAW_CHUNK), o_fin = carve(sizeof(int) * (size_t)count);
    if (off > b->bytes) {
        if (b->mem) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipFree(b->mem)); b->mem = nullptr; b->bytes = 0; }
        MIS_HIP(ctx, hipMalloc(&b->mem, off));
        b->bytes = off;
    }
    uint8_t* m = (uint8_t*)b->mem;
    b->calls = (HomoCall*)(m + o_calls); b->results = (HomoResult*)(m + o_res); b->state = m + o_state;
    b->sub_idx = (int*)(m + o_sub); b->Hc = (double*)(m + o_hc); b->valid = (int*)(m + o_valid); b->good = (int*)(m + o_good);
    b->scr = (float*)(m + o_scr); b->rec = (double*)(m + o_rec); b->draw_next = (unsigned*)(m + o_dn); b->draw_idx = (int*)(m + o_di); b->fin = (int*)(m + o_fin);
    b->count = count; b->points = points; b->max_iters = max_iters;
    return MIS_OK;
}

#ifdef MIS_TAIL_PROF
extern "C" int mis_debug_tail_prof(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_prof), sizeof(unsigned long long) * 12);
    if (reset) { unsigned long long z[12] = {0}; z[8] = ~0ull; hipMemcpyToSymbol(HIP_SYMBOL(g_tail_prof), z, sizeof(z)); }
    return 0;
}
extern "C" int mis_debug_tail_log(unsigned long long* out, int cap, int reset) {      // -> entries copied (6 values each)
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tail_log_n), sizeof(n)) != hipSuccess) return -1;
    if (n > 1024) n = 1024;
    if ((int)n > cap) n = cap;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_log), sizeof(unsigned long long) * 6 * n) != hipSuccess) return -1;
    if (reset) { unsigned z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tail_log_n), &z, sizeof(z)) != hipSuccess) return -1; }
    return (int)n;
}
extern "C" int mis_debug_draw_prof(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_draw_prof), sizeof(unsigned long long) * 12);
    if (reset) { unsigned long long z[12] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_draw_prof), z, sizeof(z)); }
    return 0;
}
extern "C" int mis_debug_hyp_prof(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hyp_prof), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_hyp_prof), z, sizeof(z)); }
    return 0;
}
extern "C" int mis_debug_jac_prof(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jac_prof), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_jac_prof), z, sizeof(z)); }
    return 0;
}
#endif

// diagnostics (tools/ransac_states.py): per-problem {n, mode, n_sub, iter, niters, draw_fail, done, max_good} of a finished batch
int homo_batch_debug_states(MisContext* ctx, const HomoBatch* b, int* out, int cap) {
    const int n = std::min(cap, b->count);
    std::vector<RansacState> st(n);
    std::vector<HomoCall> calls(n);
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MIS_HIP(ctx, hipMemcpy(st.data(), b->state, sizeof(RansacState) * n, hipMemcpyDeviceToHost));
    MIS_HIP(ctx, hipMemcpy(calls.data(), b->calls, sizeof(HomoCall) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        int* o = out + 8 * i;
        o[0] = calls[i].n; o[1] = st[i].mode; o[2] = st[i].n_sub; o[3] = st[i].iter; o[4] = st[i].niters; o[5] = st[i].draw_fail; o[6] = st[i].done; o[7] = st[i].max_good;
    }
    return n;
}

void homo_batch_release(HomoBatch* b) {
    if (b->mem) hipFree(b->mem);
    b->mem = nullptr; b->bytes = 0;
}

int homo_batch_run(MisContext* ctx, HomoBatch* b, double thresh, int max_iters, double confidence, int phases, hipStream_t stream, const HomoSync* sync) {
    const HomoSync none;
    const HomoSync& sy = sync ? *sync : none;
    MIS_CHECK(ctx, max_iters >= 1 && max_iters <= b->max_iters, MIS_E_INVALID, "max_iters %d outside the reserved range", max_iters);
    MIS_CHECK(ctx, confidence > 0 && confidence < 1, MIS_E_INVALID, "confidence must be in (0,1)");
    RngTable rt;
    int rc = rng_table(ctx, &rt);
    if (rc != MIS_OK) return rc;
    if (thresh <= 0) thresh = 3;
    const float thr = (float)(thresh * thresh);
    hipStream_t st = stream ? stream : ctx->stream;
    static bool attr_set[64] = {false};
    const size_t hq_lds = sizeof(double) * HQ_ELEMS * HQ_STRIDE;
    if (!attr_set[ctx->device & 63]) {
        MIS_HIP(ctx, hipFuncSetAttribute((const void*)hyp_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hq_lds));
        attr_set[ctx->device & 63] = true;
    }
    RansacState* states = (RansacState*)b->state;
    const int p0 = std::min(PHASE0, max_iters);
    // the launches that run a DLT / LM refinement (parts 0, 2, 4) carry the term stages of ordered_sums in dynamic LDS; replay-only and mask-only ones do not
    // (MIS_TAIL_PLAIN=1: none anywhere -- every ordered sum takes round 3's plain loop; the parity tests run both)
    static const bool plain = getenv("MIS_TAIL_PLAIN") != nullptr && atoi(getenv("MIS_TAIL_PLAIN")) != 0;
    auto tail_staged = [&](int part) { return (part == 1 || part == 3 || plain) ? 0 : 1; };
    auto tail_lds = [&](int part) { return tail_staged(part) ? TAIL_DYN_LDS : (size_t)0; };
    if (phases == 0 || phases == 2 || phases == 3) {
        MIS_HIP(ctx, hipMemsetAsync(b->fin, 0xff, sizeof(int) * (size_t)b->count, st));
        hipLaunchKernelGGL(draw_kernel, dim3(b->count), dim3(DRAW_TB), 0, st, b->calls, states, b->sub_idx, b->draw_idx, rt.U, rt.state_T, max_iters, 0, p0);
        if (sy.rec && sy.rec_pos == 2) MIS_HIP(ctx, hipEventRecord(sy.rec, st));
        hipLaunchKernelGGL(hyp_quad_kernel, dim3((p0 + HQ_HYPS - 1) / HQ_HYPS, b->count), dim3(4 * HQ_HYPS), hq_lds, st, b->calls, states, b->sub_idx, b->Hc, b->valid, 0, max_iters);
        if (sy.rec_hyp0) MIS_HIP(ctx, hipEventRecord(sy.rec_hyp0, st));
        hipLaunchKernelGGL(hyp_count_kernel, dim3((p0 + 3) / 4, b->count), dim3(256), 0, st, b->calls, states, (const double*)b->Hc, (const int*)b->valid, b->good, 0, max_iters, thr);
        hipLaunchKernelGGL(scan_tail_kernel, dim3(b->count), dim3(TB), tail_lds(phases == 3 ? 1 : 0), st, b->calls, states, b->Hc, b->valid, b->good, b->scr, b->rec, b->results, 0, p0,
                           max_iters, confidence, thr, b->fin, phases == 3 ? 1 : 0, 0, tail_staged(phases == 3 ? 1 : 0));
    }
    if (phases == 4)   // the tails a phases == 3 run left pending (fin == 0)
        hipLaunchKernelGGL(scan_tail_kernel, dim3(b->count), dim3(TB), tail_lds(2), st, b->calls, states, b->Hc, b->valid, b->good, b->scr, b->rec, b->results, 0, p0,
                           max_iters, confidence, thr, b->fin, 2, 0, tail_staged(2));
    if (phases >= 10) {   // 10 + 2 w: mask + compaction, 11 + 2 w: DLT + LM, of the problems a replay-only run left pending with fin == w
        const int want = (phases - 10) >> 1, part = 3 + ((phases - 10) & 1);
        hipLaunchKernelGGL(scan_tail_kernel, dim3(b->count), dim3(TB), tail_lds(part), st, b->calls, states, b->Hc, b->valid, b->good, b->scr, b->rec, b->results, 0, p0,
                           max_iters, confidence, thr, b->fin, part, want, tail_staged(part));
    }
    if ((phases == 1 || phases == 2 || phases == 6) && max_iters > p0) {
        hipLaunchKernelGGL(draw_kernel, dim3(b->count), dim3(DRAW_TB), 0, st, b->calls, states, b->sub_idx, b->draw_idx, rt.U, rt.state_T, max_iters, 1, max_iters);
        if (sy.rec && sy.rec_pos == 0) MIS_HIP(ctx, hipEventRecord(sy.rec, st));
        if (sy.wait_hyp1) MIS_HIP(ctx, hipStreamWaitEvent(st, sy.wait_hyp1, 0));
        hipLaunchKernelGGL(hyp_quad_kernel, dim3((max_iters - p0 + HQ_HYPS - 1) / HQ_HYPS, b->count), dim3(4 * HQ_HYPS), hq_lds, st, b->calls, states, b->sub_idx, b->Hc, b->valid, p0, max_iters);
        if (sy.rec && sy.rec_pos == 1) MIS_HIP(ctx, hipEventRecord(sy.rec, st));
        hipLaunchKernelGGL(hyp_count_kernel, dim3((max_iters - p0 + 3) / 4, b->count), dim3(256), 0, st, b->calls, states, (const double*)b->Hc, (const int*)b->valid, b->good, p0,
                           max_iters, thr);
        hipLaunchKernelGGL(scan_tail_kernel, dim3(b->count), dim3(TB), tail_lds(phases == 6 ? 1 : 0), st, b->calls, states, b->Hc, b->valid, b->good, b->scr, b->rec, b->results, p0,
                           max_iters, max_iters, confidence, thr, b->fin, phases == 6 ? 1 : 0, 0, tail_staged(phases == 6 ? 1 : 0));
    }
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}
