// match.hip -- pairwise matching, replaces the reference's
// makePtr<BestOf2NearestMatcher>(try_cuda, match_conf) (image_stitching/image_stitching.cpp:647) and
// (*matcher)(features, pairwise_matches) (:653), plus myLeaveBiggestComponent (:215-278) on its output.
//
// Three launches for ALL image pairs:
//   1. knn2_hamming_kernel   exact 2-NN (distance, trainIdx) of every descriptor, both directions
//   2. ratio_union_kernel    ratio test + de-duplicated union, ordered like the reference, plus the
//                            centre-shifted point lists for findHomography
//   3. pair_homography_kernel  one workgroup per pair: RANSAC (subsets drawn sequentially from
//                            cv::RNG(-1) so the data-dependent RNG consumption and the adaptive
//                            iteration count are reproduced exactly; hypotheses of a chunk solved and
//                            scored in parallel), inlier mask, re-fit on inliers, 10-iteration LM.
// Every f64 reduction that the CPU path does sequentially is kept sequential per accumulator (one
// thread per matrix entry), so H comes out bit-identical; the library is built with
// -ffp-contract=off.
#include "common.h"
#include "dev_math.h"
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
__global__ __launch_bounds__(256) void knn2_hamming_kernel(const FeatDev* feats, const PairDesc* pairs, int* idx2, int* dist2) {
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
    int d0 = 1 << 30, d1 = 1 << 30, i0 = -1, i1 = -1;
    for (int t0 = 0; t0 < T.n; t0 += KNN_TILE) {
        const int nt = min(KNN_TILE, T.n - t0);
        __syncthreads();
        const uint4* src = reinterpret_cast<const uint4*>(T.desc + (size_t)t0 * 32);
        for (int k = threadIdx.x; k < nt * 2; k += 256) tr[k] = src[k];
        __syncthreads();
        if (active) {
            for (int j = 0; j < nt; j++) {
                uint4 b0 = tr[2 * j], b1 = tr[2 * j + 1];
                int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                        __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = t0 + j; }
                else if (d < d1) { d1 = d; i1 = t0 + j; }
            }
        }
    }
    if (active) {
        idx2[(off + q) * 2] = i0; idx2[(off + q) * 2 + 1] = i1;
        dist2[(off + q) * 2] = d0; dist2[(off + q) * 2 + 1] = d1;
    }
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
__global__ __launch_bounds__(1024) void ratio_union_kernel(const FeatDev* feats, const PairDesc* pairs, const int* idx2, const int* dist2, float ratio,
                                                           MisDMatch* matches, float* src_xy, float* dst_xy, int* n_matches) {
    const PairDesc pd = pairs[blockIdx.x];
    const FeatDev F1 = feats[pd.i], F2 = feats[pd.j];
    MisDMatch* m = matches + pd.m_off;
    float* sp = src_xy + 2 * pd.m_off;
    float* dp = dst_xy + 2 * pd.m_off;
    const int* i12 = idx2 + pd.knn_off12 * 2;
    const int* d12 = dist2 + pd.knn_off12 * 2;
    const int* i21 = idx2 + pd.knn_off21 * 2;
    const int* d21 = dist2 + pd.knn_off21 * 2;
    const float hw1 = (float)F1.w * 0.5f, hh1 = (float)F1.h * 0.5f, hw2 = (float)F2.w * 0.5f, hh2 = (float)F2.h * 0.5f;
    int base = 0;
    if (F2.n >= 2)
        for (int q0 = 0; q0 < F1.n; q0 += 1024) {
            int q = q0 + threadIdx.x, ok = 0;
            if (q < F1.n) ok = (float)d12[2 * q] < ratio * (float)d12[2 * q + 1];
            int tot, off = block_scan_flag(ok, &tot);
            if (ok) {
                int t = i12[2 * q];
                MisDMatch mm = {q, t, 0, (float)d12[2 * q]};
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
                ok = (float)d21[2 * q] < ratio * (float)d21[2 * q + 1];
                if (ok) {
                    t1 = i21[2 * q];
                    bool acc12 = F2.n >= 2 && (float)d12[2 * t1] < ratio * (float)d12[2 * t1 + 1];
                    if (acc12 && i12[2 * t1] == q) ok = 0;  // (t1, q) already in the 1->2 set
                }
            }
            int tot, off = block_scan_flag(ok, &tot);
            if (ok) {
                MisDMatch mm = {t1, q, -1, (float)d21[2 * q]};
                m[base + off] = mm;
                sp[2 * (base + off)] = F1.kps[t1].x - hw1; sp[2 * (base + off) + 1] = F1.kps[t1].y - hh1;
                dp[2 * (base + off)] = F2.kps[q].x - hw2; dp[2 * (base + off) + 1] = F2.kps[q].y - hh2;
            }
            base += tot;
        }
    if (threadIdx.x == 0) n_matches[blockIdx.x] = base;
}

// ---------------------------------------------------------------- K9: findHomography ----------
constexpr int HB = 256;    // threads of the homography workgroup
constexpr int CHUNK = 32;  // RANSAC hypotheses solved in parallel per round

// per-thread (slot) matrices in LDS: element e of slot s lives at base[e * CHUNK + s]
struct Slot {
    double *A, *V, *W;
    __device__ __forceinline__ double& a(int e) const { return A[e * CHUNK]; }
    __device__ __forceinline__ double& v(int e) const { return V[e * CHUNK]; }
    __device__ __forceinline__ double& w(int e) const { return W[e * CHUNK]; }
};

__device__ __forceinline__ double cv_hypot(double a, double b) {
    a = fabs(a); b = fabs(b);
    if (a > b) { b /= a; return a * sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
    return 0;
}

// core/src/lapack.cpp JacobiImpl_<double>: eigen-decomposition of the symmetric n x n matrix in s.A;
// eigenvalues (descending) in s.W, eigenvectors as rows of s.V
__device__ void jacobi_eigen(const Slot s, const int n) {
    const double eps = DBL_EPSILON;
    int i, j, k, m, indR[9], indC[9];
    double mv;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) s.v(i * n + j) = 0; s.v(i * n + i) = 1; }
    const int maxIters = n * n * 30;
    for (k = 0; k < n; k++) {
        s.w(k) = s.a((n + 1) * k);
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(s.a(n * k + m)), i = k + 2; i < n; i++) {
                double val = fabs(s.a(n * k + i));
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(s.a(k)), i = 1; i < k; i++) {
                double val = fabs(s.a(n * i + k));
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1) for (int iters = 0; iters < maxIters; iters++) {
        for (k = 0, mv = fabs(s.a(indR[0])), i = 1; i < n - 1; i++) {
            double val = fabs(s.a(n * i + indR[i]));
            if (mv < val) mv = val, k = i;
        }
        int l = indR[k];
        for (i = 1; i < n; i++) {
            double val = fabs(s.a(n * indC[i] + i));
            if (mv < val) mv = val, k = indC[i], l = i;
        }
        double p = s.a(n * k + l);
        if (fabs(p) <= eps) break;
        double y = (s.w(l) - s.w(k)) * 0.5;
        double t = fabs(y) + cv_hypot(p, y);
        double sn = cv_hypot(p, t);
        double c = t / sn;
        sn = p / sn; t = (p / t) * p;
        if (y < 0) sn = -sn, t = -t;
        s.a(n * k + l) = 0;
        s.w(k) -= t; s.w(l) += t;
        double a0, b0;
#define MIS_ROT(X, Y) a0 = X, b0 = Y, X = a0 * c - b0 * sn, Y = a0 * sn + b0 * c
        for (i = 0; i < k; i++) MIS_ROT(s.a(n * i + k), s.a(n * i + l));
        for (i = k + 1; i < l; i++) MIS_ROT(s.a(n * k + i), s.a(n * i + l));
        for (i = l + 1; i < n; i++) MIS_ROT(s.a(n * k + i), s.a(n * l + i));
        for (i = 0; i < n; i++) MIS_ROT(s.v(n * k + i), s.v(n * l + i));
#undef MIS_ROT
        for (j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = fabs(s.a(n * idx + m)), i = idx + 2; i < n; i++) {
                    double val = fabs(s.a(n * idx + i));
                    if (mv < val) mv = val, m = i;
                }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabs(s.a(idx)), i = 1; i < idx; i++) {
                    double val = fabs(s.a(n * i + idx));
                    if (mv < val) mv = val, m = i;
                }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (s.w(m) < s.w(i)) m = i;
        if (k != m) {
            double tw = s.w(m); s.w(m) = s.w(k); s.w(k) = tw;
            for (i = 0; i < n; i++) { double tv = s.v(n * m + i); s.v(n * m + i) = s.v(n * k + i); s.v(n * k + i) = tv; }
        }
    }
}

// rows of the DLT design matrix (HomographyEstimatorCallback::runKernel)
__device__ __forceinline__ double dlt_lx(int j, double X, double Y, double x) {
    switch (j) { case 0: return X; case 1: return Y; case 2: return 1; case 6: return -x * X; case 7: return -x * Y; case 8: return -x; default: return 0; }
}
__device__ __forceinline__ double dlt_ly(int j, double X, double Y, double y) {
    switch (j) { case 3: return X; case 4: return Y; case 5: return 1; case 6: return -y * X; case 7: return -y * Y; case 8: return -y; default: return 0; }
}

// eigenvector of the smallest eigenvalue -> de-normalised, scaled homography
__device__ void dlt_denormalise(const Slot s, const double* nrm /* cmx cmy cMx cMy smx smy sMx sMy */, double* H) {
    double H0[9], T[9], R[9];
    for (int i = 0; i < 9; i++) H0[i] = s.v(72 + i);
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
__device__ void dlt_finish(const Slot s, const double* nrm, double* H) {
    for (int j = 0; j < 9; j++) for (int k = 0; k < j; k++) s.a(j * 9 + k) = s.a(k * 9 + j);
    jacobi_eigen(s, 9);
    dlt_denormalise(s, nrm, H);
}

// serial DLT of `count` points by ONE thread (the 4-point RANSAC hypotheses)
__device__ int dlt_serial(const float* M, const float* m, int count, const Slot s, double* H) {
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) { cmx += m[2 * i]; cmy += m[2 * i + 1]; cMx += M[2 * i]; cMy += M[2 * i + 1]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[2 * i] - cmx); smy += fabs(m[2 * i + 1] - cmy);
        sMx += fabs(M[2 * i] - cMx); sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    for (int e = 0; e < 81; e++) s.a(e) = 0;
    for (int i = 0; i < count; i++) {
        double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        for (int j = 0; j < 9; j++) {
            double lxj = dlt_lx(j, X, Y, x), lyj = dlt_ly(j, X, Y, y);
            for (int k = j; k < 9; k++) s.a(j * 9 + k) += lxj * dlt_lx(k, X, Y, x) + lyj * dlt_ly(k, X, Y, y);
        }
    }
    const double nrm[8] = {cmx, cmy, cMx, cMy, smx, smy, sMx, sMy};
    dlt_finish(s, nrm, H);
    return 1;
}

// fundam.cpp haveCollinearPoints (only the last point is tested) + the 4-point orientation test
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

struct HomoShared {
    double A[81 * CHUNK], V[81 * CHUNK], W[9 * CHUNK];
    double Hc[CHUNK][9];
    float ms1[CHUNK][8], ms2[CHUNK][8];
    int valid[CHUNK], good[CHUNK];
    double best[9];
    double nrm[8];
    double lm[8 + 8 + 64 + 64 + 8 + 8 + 8 + 8];  // x, xd, A, Ap, v, d, D, scalars
    int indR[9], indC[9];
    float Hf[9];
    unsigned long long rng;
    int niters, iter, max_good, n_gen, stop, result, np;
};


// The same Jacobi iteration with the n independent plane rotations of a step spread over n threads
// and the four index-table scans over four threads; the arithmetic of every element is unchanged.
// Called by the whole workgroup on slot 0 (threads >= n only take part in the barriers).
__device__ void jacobi_eigen_coop(HomoShared& S, const int n) {
    const Slot s = {S.A, S.V, S.W};
    const int t = threadIdx.x;
    const double eps = DBL_EPSILON;
    int i, k, l, m;
    double mv;
    if (t < n) {
        for (int j = 0; j < n; j++) s.v(t * n + j) = (j == t) ? 1. : 0.;
        s.w(t) = s.a((n + 1) * t);
        k = t;
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(s.a(n * k + m)), i = k + 2; i < n; i++) {
                double val = fabs(s.a(n * k + i));
                if (mv < val) mv = val, m = i;
            }
            S.indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(s.a(k)), i = 1; i < k; i++) {
                double val = fabs(s.a(n * i + k));
                if (mv < val) mv = val, m = i;
            }
            S.indC[k] = m;
        }
    }
    __syncthreads();
    const int maxIters = n * n * 30;
    if (n > 1) for (int iters = 0; iters < maxIters; iters++) {
        for (k = 0, mv = fabs(s.a(S.indR[0])), i = 1; i < n - 1; i++) {
            double val = fabs(s.a(n * i + S.indR[i]));
            if (mv < val) mv = val, k = i;
        }
        l = S.indR[k];
        for (i = 1; i < n; i++) {
            double val = fabs(s.a(n * S.indC[i] + i));
            if (mv < val) mv = val, k = S.indC[i], l = i;
        }
        const double p = s.a(n * k + l);
        if (fabs(p) <= eps) break;  // uniform: every thread reads the same LDS words
        const double y = (s.w(l) - s.w(k)) * 0.5;
        double tt = fabs(y) + cv_hypot(p, y);
        double sn = cv_hypot(p, tt);
        const double c = tt / sn;
        sn = p / sn; tt = (p / tt) * p;
        if (y < 0) sn = -sn, tt = -tt;
        __syncthreads();  // all pivot inputs read before anything is rewritten
        if (t == 0) { s.a(n * k + l) = 0; s.w(k) -= tt; s.w(l) += tt; }
        if (t < n) {
            double a0, b0;
#define MIS_ROT(X, Y) a0 = X, b0 = Y, X = a0 * c - b0 * sn, Y = a0 * sn + b0 * c
            if (t < k) MIS_ROT(s.a(n * t + k), s.a(n * t + l));
            else if (t > k && t < l) MIS_ROT(s.a(n * k + t), s.a(n * t + l));
            else if (t > l) MIS_ROT(s.a(n * k + t), s.a(n * l + t));
            MIS_ROT(s.v(n * k + t), s.v(n * l + t));
#undef MIS_ROT
        }
        __syncthreads();
        if (t < 4) {
            const int idx = t < 2 ? k : l;
            if ((t & 1) == 0) {
                if (idx < n - 1) {
                    for (m = idx + 1, mv = fabs(s.a(n * idx + m)), i = idx + 2; i < n; i++) {
                        double val = fabs(s.a(n * idx + i));
                        if (mv < val) mv = val, m = i;
                    }
                    S.indR[idx] = m;
                }
            } else if (idx > 0) {
                for (m = 0, mv = fabs(s.a(idx)), i = 1; i < idx; i++) {
                    double val = fabs(s.a(n * i + idx));
                    if (mv < val) mv = val, m = i;
                }
                S.indC[idx] = m;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (t == 0) {
        for (k = 0; k < n - 1; k++) {
            m = k;
            for (i = k + 1; i < n; i++) if (s.w(m) < s.w(i)) m = i;
            if (k != m) {
                double tw = s.w(m); s.w(m) = s.w(k); s.w(k) = tw;
                for (i = 0; i < n; i++) { double tv = s.v(n * m + i); s.v(n * m + i) = s.v(n * k + i); s.v(n * k + i) = tv; }
            }
        }
    }
    __syncthreads();
}

// LM callback of the homography refinement (fundam.cpp HomographyRefineCallback)
__device__ __forceinline__ void lm_point(const double* h, double Mx, double My, double* ww, double* xi, double* yi) {
    double w = (h[6] * Mx + h[7] * My) + 1.;
    w = fabs(w) > DBL_EPSILON ? 1. / w : 0;
    *ww = w;
    *xi = ((h[0] * Mx + h[1] * My) + h[2]) * w;
    *yi = ((h[3] * Mx + h[4] * My) + h[5]) * w;
}
__device__ __forceinline__ void lm_jrow(int row, double Mx, double My, double ww, double xi, double yi, double* J) {
    if (row == 0) { J[0] = Mx * ww; J[1] = My * ww; J[2] = ww; J[3] = J[4] = J[5] = 0.; J[6] = -Mx * ww * xi; J[7] = -My * ww * xi; }
    else { J[0] = J[1] = J[2] = 0.; J[3] = Mx * ww; J[4] = My * ww; J[5] = ww; J[6] = -Mx * ww * yi; J[7] = -My * ww * yi; }
}

// cv::findHomography(src, dst, mask, RANSAC, thresh, maxIters, confidence) by one workgroup.
// src/dst: n points; mask (may be null); scratch: 4*n floats for the compressed inliers.
__device__ int find_homography_block(HomoShared& S, const float* src, const float* dst, int n, double thresh, int max_iters, double confidence,
                                     double* Hout, uint8_t* mask, float* scratch, double* rec, int* iters_out) {
    const int t = threadIdx.x;
    const Slot slot = {S.A + (t % CHUNK), S.V + (t % CHUNK), S.W + (t % CHUNK)};
    const Slot slot0 = {S.A, S.V, S.W};
    if (thresh <= 0) thresh = 3;
    const float thr = (float)(thresh * thresh);
    if (t == 0) { S.result = 0; S.max_good = 0; S.iter = 0; S.stop = 0; S.niters = max_iters > 1 ? max_iters : 1; S.rng = ~0ull; }
    __syncthreads();
    if (n < 4) {
        for (int i = t; mask && i < n; i += HB) mask[i] = 0;
        if (iters_out && t == 0) *iters_out = 0;
        __syncthreads();
        return 0;
    }
    if (n == 4) {
        if (t == 0) { S.result = dlt_serial(src, dst, 4, slot0, S.best) > 0; }
        __syncthreads();
        for (int i = t; mask && i < n; i += HB) mask[i] = S.result ? 1 : 0;
        if (t == 0) { if (S.result) for (int i = 0; i < 9; i++) Hout[i] = S.best[i]; if (iters_out) *iters_out = 0; }
        int r = S.result;
        __syncthreads();
        return r;
    }
    // ---- RANSACPointSetRegistrator::run ----
    while (true) {
        if (t == 0) {
            // draw the subsets of the next round sequentially: RNG consumption depends on the data only
            // through checkSubset, never on the models, so drawing ahead of the adaptive exit is exact
            int g = 0;
            unsigned long long st = S.rng;
            const int budget = min(CHUNK, S.niters - S.iter);
            for (; g < budget; g++) {
                bool found = false;
                for (int attempts = 0; attempts < 10000 && !found; attempts++) {
                    int idx[4];
                    for (int i = 0; i < 4; i++) {
                        int idx_i;
                        bool dup;
                        do {
                            st = (unsigned long long)(unsigned)st * 4164903690u + (unsigned)(st >> 32);
                            idx_i = (int)((unsigned)st % (unsigned)n);
                            dup = false;
                            for (int q = 0; q < i; q++) dup |= idx[q] == idx_i;
                        } while (dup);
                        idx[i] = idx_i;
                        S.ms1[g][2 * i] = src[2 * idx_i]; S.ms1[g][2 * i + 1] = src[2 * idx_i + 1];
                        S.ms2[g][2 * i] = dst[2 * idx_i]; S.ms2[g][2 * i + 1] = dst[2 * idx_i + 1];
                    }
                    found = check_subset(S.ms1[g], S.ms2[g]);
                }
                if (!found) { S.stop = 1; break; }
            }
            S.rng = st; S.n_gen = g;
        }
        __syncthreads();
        const int ngen = S.n_gen;
        if (t < ngen) S.valid[t] = dlt_serial(S.ms1[t], S.ms2[t], 4, slot, S.Hc[t]);
        __syncthreads();
        // findInliers of every hypothesis: one wave per hypothesis, lanes stride over the points
        for (int c = t >> 6; c < ngen; c += HB / 64) {
            int cnt = 0;
            if (S.valid[c]) {
                float Hf[9];
                for (int i = 0; i < 9; i++) Hf[i] = (float)S.Hc[c][i];
                for (int i = t & 63; i < n; i += 64) cnt += is_inlier(Hf, src[2 * i], src[2 * i + 1], dst[2 * i], dst[2 * i + 1], thr);
                for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            }
            if ((t & 63) == 0) S.good[c] = cnt;
        }
        __syncthreads();
        if (t == 0) {
            // replay the sequential loop over this round's hypotheses
            for (int c = 0; c < ngen && S.iter < S.niters; c++) {
                S.iter++;
                if (!S.valid[c]) continue;
                int good = S.good[c];
                if (good > (S.max_good > 3 ? S.max_good : 3)) {
                    for (int i = 0; i < 9; i++) S.best[i] = S.Hc[c][i];
                    S.max_good = good;
                    S.niters = ransac_update_num_iters(confidence, (double)(n - good) / n, S.niters);
                }
            }
            if (S.stop || S.iter >= S.niters) S.stop = 2;
        }
        __syncthreads();
        if (S.stop == 2) break;
    }
    if (iters_out && t == 0) *iters_out = S.iter;
    const int result = S.max_good > 0;
    if (!result) {
        for (int i = t; mask && i < n; i += HB) mask[i] = 0;
        __syncthreads();
        return 0;
    }
    // best mask + ordered compaction of the inliers (compressElems)
    if (t == 0) { for (int i = 0; i < 9; i++) S.Hf[i] = (float)S.best[i]; S.np = 0; }
    __syncthreads();
    float* s1 = scratch;
    float* d1 = scratch + 2 * (size_t)n;
    {
        __shared__ int wcnt[HB / 64];
        __shared__ int base;
        if (t == 0) base = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += HB) {
            int i = i0 + t, f = 0;
            if (i < n) f = is_inlier(S.Hf, src[2 * i], src[2 * i + 1], dst[2 * i], dst[2 * i + 1], thr);
            if (mask && i < n) mask[i] = (uint8_t)f;
            unsigned long long b = __ballot(f);
            int within = __popcll(b & ((1ull << (t & 63)) - 1ull));
            if ((t & 63) == 0) wcnt[t >> 6] = __popcll(b);
            __syncthreads();
            int off = base;
            for (int k = 0; k < (t >> 6); k++) off += wcnt[k];
            if (f) {
                s1[2 * (off + within)] = src[2 * i]; s1[2 * (off + within) + 1] = src[2 * i + 1];
                d1[2 * (off + within)] = dst[2 * i]; d1[2 * (off + within) + 1] = dst[2 * i + 1];
            }
            __syncthreads();
            if (t == 0) { int s = 0; for (int k = 0; k < HB / 64; k++) s += wcnt[k]; base += s; }
            __syncthreads();
        }
        if (t == 0) S.np = base;
        __syncthreads();
    }
    const int np = S.np;
    if (np > 0) {
        // ---- runKernel on all inliers.  Every f64 accumulator is summed in point order by ONE thread
        // (bit-identical to the sequential CPU loop); the per-point terms are produced in parallel. ----
        if (t < 4) {
            double acc = 0;
            const float* p = t < 2 ? d1 : s1;  // cmx cmy cMx cMy
            for (int i = 0; i < np; i++) acc += p[2 * i + (t & 1)];
            S.nrm[t] = acc / np;
        }
        __syncthreads();
        if (t < 4) {
            double acc = 0, c = S.nrm[t];
            const float* p = t < 2 ? d1 : s1;  // smx smy sMx sMy
            for (int i = 0; i < np; i++) acc += fabs(p[2 * i + (t & 1)] - c);
            S.nrm[4 + t] = acc;
        }
        __syncthreads();
        const bool degenerate = fabs(S.nrm[4]) < DBL_EPSILON || fabs(S.nrm[5]) < DBL_EPSILON || fabs(S.nrm[6]) < DBL_EPSILON || fabs(S.nrm[7]) < DBL_EPSILON;
        __syncthreads();
        if (!degenerate) {
            if (t < 4) S.nrm[4 + t] = np / S.nrm[4 + t];
            __syncthreads();
            {
                const double cmx = S.nrm[0], cmy = S.nrm[1], cMx = S.nrm[2], cMy = S.nrm[3], smx = S.nrm[4], smy = S.nrm[5], sMx = S.nrm[6], sMy = S.nrm[7];
                for (int i = t; i < np; i += HB) {
                    double x = (d1[2 * i] - cmx) * smx, y = (d1[2 * i + 1] - cmy) * smy;
                    double X = (s1[2 * i] - cMx) * sMx, Y = (s1[2 * i + 1] - cMy) * sMy;
                    double* r = rec + 10 * (size_t)i;  // X Y 1 0 -xX -xY -x -yX -yY -y
                    r[0] = X; r[1] = Y; r[2] = 1; r[3] = 0; r[4] = -x * X; r[5] = -x * Y; r[6] = -x; r[7] = -y * X; r[8] = -y * Y; r[9] = -y;
                }
            }
            __syncthreads();
            if (t < 45) {
                int j = 0, k = t;  // t-th entry of the upper triangle, row-major
                while (k >= 9 - j) { k -= 9 - j; j++; }
                k += j;
                const int lxi[9] = {0, 1, 2, 3, 3, 3, 4, 5, 6}, lyi[9] = {3, 3, 3, 0, 1, 2, 7, 8, 9};
                const int xj = lxi[j], xk = lxi[k], yj = lyi[j], yk = lyi[k];
                double acc = 0;
                for (int i = 0; i < np; i++) {
                    const double* r = rec + 10 * (size_t)i;
                    acc += r[xj] * r[xk] + r[yj] * r[yk];
                }
                slot0.a(j * 9 + k) = acc;
            }
            __syncthreads();
            if (t < 81) { int j = t / 9, k = t % 9; if (k < j) slot0.a(j * 9 + k) = slot0.a(k * 9 + j); }
            __syncthreads();
            jacobi_eigen_coop(S, 9);
            if (t == 0) dlt_denormalise(slot0, S.nrm, S.best);
            __syncthreads();
        }
        // ---- LMSolver, 10 iterations, on the 8 free parameters ----
        double* x = S.lm;            double* xd = x + 8;   double* A = xd + 8;  double* Ap = A + 64;
        double* v = Ap + 64;         double* d = v + 8;    double* D = d + 8;   double* sc = D + 8;  // sc: S, Sd, rmax, flag, lambda, lc
        // normal equations at h: A = J^T J, v = J^T r, S = |r|^2, rmax = |r|_inf
        auto normal_eq = [&](const double* h, bool with_J) {
            for (int p = t; p < np; p += HB) {
                double Mx = (double)s1[2 * p], My = (double)s1[2 * p + 1], ww, xi, yi;
                lm_point(h, Mx, My, &ww, &xi, &yi);
                double* r = rec + 10 * (size_t)p;  // a b ww c0 c1 c2 c3 e0 e1 0
                r[7] = xi - (double)d1[2 * p]; r[8] = yi - (double)d1[2 * p + 1];
                if (with_J) {
                    r[0] = Mx * ww; r[1] = My * ww; r[2] = ww;
                    r[3] = -Mx * ww * xi; r[4] = -My * ww * xi; r[5] = -Mx * ww * yi; r[6] = -My * ww * yi; r[9] = 0;
                }
            }
            __syncthreads();
            const int j0[8] = {0, 1, 2, 9, 9, 9, 3, 4}, j1[8] = {9, 9, 9, 0, 1, 2, 5, 6};
            if (with_J && t < 36) {
                int i = 0, j = t;
                while (j >= 8 - i) { j -= 8 - i; i++; }
                j += i;
                const int a0 = j0[i], b0 = j0[j], a1 = j1[i], b1 = j1[j];
                double acc = 0;
                for (int p = 0; p < np; p++) {
                    const double* r = rec + 10 * (size_t)p;
                    acc += r[a0] * r[b0];
                    acc += r[a1] * r[b1];
                }
                A[i * 8 + j] = acc; A[j * 8 + i] = acc;
            } else if (with_J && t >= 64 && t < 72) {
                const int i = t - 64, a0 = j0[i], a1 = j1[i];
                double acc = 0;
                for (int p = 0; p < np; p++) {
                    const double* r = rec + 10 * (size_t)p;
                    acc += r[a0] * r[7];
                    acc += r[a1] * r[8];
                }
                v[i] = acc;
            } else if (t == 128) {
                double acc = 0, mx = 0;
                for (int p = 0; p < np; p++) {
                    const double* r = rec + 10 * (size_t)p;
                    double e0 = r[7], e1 = r[8];
                    acc += e0 * e0; acc += e1 * e1;
                    if (fabs(e0) > mx) mx = fabs(e0);
                    if (fabs(e1) > mx) mx = fabs(e1);
                }
                sc[with_J ? 0 : 1] = acc;
                if (with_J) sc[2] = mx;
            }
            __syncthreads();
        };
        // SVBkSb thresholding + back substitution pieces shared by solve() and invert() (DECOMP_EIG)
        if (t < 8) x[t] = S.best[t];
        __syncthreads();
        normal_eq(x, true);
        if (t < 8) D[t] = A[t * 8 + t];
        if (t == 0) { sc[4] = 1; sc[5] = 0.75; }  // lambda, lc
        __syncthreads();
        for (int iter = 0;;) {
            if (t < 64) slot0.a(t) = (t / 8 == t % 8) ? A[t] + sc[4] * D[t / 8] : A[t];
            __syncthreads();
            jacobi_eigen_coop(S, 8);
            if (t == 0) {
                // solve(Ap, v, d, DECOMP_EIG): SVBkSb
                double thrw = 0;
                for (int i = 0; i < 8; i++) thrw += slot0.w(i);
                thrw *= DBL_EPSILON * 2;
                for (int j = 0; j < 8; j++) d[j] = 0;
                for (int i = 0; i < 8; i++) {
                    double wi = slot0.w(i);
                    if (fabs(wi) <= thrw) continue;
                    wi = 1 / wi;
                    double s = 0;
                    for (int j = 0; j < 8; j++) s += slot0.v(i * 8 + j) * v[j];
                    s *= wi;
                    for (int j = 0; j < 8; j++) d[j] = d[j] + s * slot0.v(i * 8 + j);
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
                sc[6] = 0;  // needs invert()
                if (R > 0.75) {
                    lambda *= 0.5;
                    if (lambda < lc) lambda = 0;
                } else if (R < 0.25) {
                    double tt = 0;
                    for (int i = 0; i < 8; i++) tt += d[i] * v[i];
                    double nu = (Sd - Sv) / (fabs(tt) > DBL_EPSILON ? tt : 1) + 2;
                    nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
                    if (lambda == 0) sc[6] = 1;
                    else lambda *= nu;
                    sc[7] = nu;
                }
                sc[4] = lambda; sc[5] = lc;
            }
            __syncthreads();
            if (sc[6] != 0.) {
                // invert(A, Ap, DECOMP_EIG) -> lambda = lc = 1 / max |diag|, nu halved
                if (t < 64) slot0.a(t) = A[t];
                __syncthreads();
                jacobi_eigen_coop(S, 8);
                if (t == 0) {
                    double thrw = 0;
                    for (int i = 0; i < 8; i++) thrw += slot0.w(i);
                    thrw *= DBL_EPSILON * 2;
                    for (int e = 0; e < 64; e++) Ap[e] = 0;
                    for (int i = 0; i < 8; i++) {
                        double wi = slot0.w(i);
                        if (fabs(wi) <= thrw) continue;
                        wi = 1 / wi;
                        for (int r = 0; r < 8; r++)
                            for (int c = 0; c < 8; c++) Ap[r * 8 + c] = Ap[r * 8 + c] + slot0.v(i * 8 + r) * (slot0.v(i * 8 + c) * wi);
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
            double dmax = 0;
            for (int i = 0; i < 8; i++) { double a = fabs(d[i]); if (a > dmax) dmax = a; }
            bool proceed = iter < 10 && dmax >= (double)FLT_EPSILON && sc[2] >= (double)FLT_EPSILON;
            __syncthreads();
            if (!proceed) break;
        }
        if (t < 8) S.best[t] = x[t];
        __syncthreads();
    }
    if (t == 0) for (int i = 0; i < 9; i++) Hout[i] = S.best[i];
    __syncthreads();
    return 1;
}

struct PairOut {
    double H[9];
    int has_H, num_inliers, ran_ransac, iters0, iters1;
    int passed;  // survived the determinant check: num_inliers / confidence are meaningful
};

__device__ __forceinline__ double det3(const double* H) {
    return H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
}

// BestOf2NearestMatcher::match after the 2-NN stage, one workgroup per pair
__global__ __launch_bounds__(HB) void pair_homography_kernel(const PairDesc* pairs, const int* n_matches, const float* src_xy, const float* dst_xy,
                                                             uint8_t* masks, float* scratch, double* recs, PairOut* outs, int thresh1, int thresh2,
                                                             double ransac_thresh, int max_iters, double confidence) {
    __shared__ HomoShared S;
    __shared__ int s_ninl;
    const PairDesc pd = pairs[blockIdx.x];
    const int nm = n_matches[blockIdx.x], t = threadIdx.x;
    PairOut* o = outs + blockIdx.x;
    const float* sp = src_xy + 2 * pd.m_off;
    const float* dp = dst_xy + 2 * pd.m_off;
    uint8_t* mask = masks + pd.m_off;
    float* scr = scratch + 8 * pd.m_off;  // 8 floats per potential match: two compressions
    double* rec = recs + 10 * pd.m_off;   // 10 doubles per potential match: per-point terms of the DLT / LM sums
    if (t == 0) { o->has_H = 0; o->num_inliers = 0; o->ran_ransac = 0; o->iters0 = o->iters1 = 0; o->passed = 0; s_ninl = 0; }
    __syncthreads();
    if (nm < thresh1) return;
    int ok = find_homography_block(S, sp, dp, nm, ransac_thresh, max_iters, confidence, o->H, mask, scr, rec, &o->iters0);
    if (t == 0) { o->ran_ransac = 1; o->has_H = ok; }
    __syncthreads();
    if (!ok || fabs(det3(o->H)) < DBL_EPSILON) return;
    // inliers only -> second estimation (points compressed in match order)
    float* s2 = scr + 4 * (size_t)nm;
    float* d2 = s2 + 2 * (size_t)nm;
    {
        __shared__ int wcnt[HB / 64];
        __shared__ int base;
        if (t == 0) base = 0;
        __syncthreads();
        for (int i0 = 0; i0 < nm; i0 += HB) {
            int i = i0 + t, f = i < nm ? mask[i] : 0;
            unsigned long long b = __ballot(f);
            int within = __popcll(b & ((1ull << (t & 63)) - 1ull));
            if ((t & 63) == 0) wcnt[t >> 6] = __popcll(b);
            __syncthreads();
            int off = base;
            for (int k = 0; k < (t >> 6); k++) off += wcnt[k];
            if (f) {
                s2[2 * (off + within)] = sp[2 * i]; s2[2 * (off + within) + 1] = sp[2 * i + 1];
                d2[2 * (off + within)] = dp[2 * i]; d2[2 * (off + within) + 1] = dp[2 * i + 1];
            }
            __syncthreads();
            if (t == 0) { int s = 0; for (int k = 0; k < HB / 64; k++) s += wcnt[k]; base += s; }
            __syncthreads();
        }
        if (t == 0) { s_ninl = base; o->num_inliers = base; o->passed = 1; }
        __syncthreads();
    }
    const int ninl = s_ninl;
    if (ninl < thresh2) return;
    ok = find_homography_block(S, s2, d2, ninl, ransac_thresh, max_iters, confidence, o->H, nullptr, scr, rec, &o->iters1);
    if (t == 0) o->has_H = ok;
}

// stand-alone findHomography on caller-supplied point lists (stage-test hook, mis_find_homography)
__global__ __launch_bounds__(HB) void find_homography_kernel(const float* src, const float* dst, int n, double thresh, int max_iters, double confidence,
                                                             double* H, uint8_t* mask, float* scratch, double* rec, int* ok_iters) {
    __shared__ HomoShared S;
    int ok = find_homography_block(S, src, dst, n, thresh, max_iters, confidence, H, mask, scratch, rec, &ok_iters[1]);
    if (threadIdx.x == 0) ok_iters[0] = ok;
}

void invert3(const double* H, double* I) {
    double d = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
    if (d != 0.) d = 1. / d;
    I[0] = (H[4] * H[8] - H[5] * H[7]) * d; I[1] = (H[2] * H[7] - H[1] * H[8]) * d; I[2] = (H[1] * H[5] - H[2] * H[4]) * d;
    I[3] = (H[5] * H[6] - H[3] * H[8]) * d; I[4] = (H[0] * H[8] - H[2] * H[6]) * d; I[5] = (H[2] * H[3] - H[0] * H[5]) * d;
    I[6] = (H[3] * H[7] - H[4] * H[6]) * d; I[7] = (H[1] * H[6] - H[0] * H[7]) * d; I[8] = (H[0] * H[4] - H[1] * H[3]) * d;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 256); }
};

void init_info(MisMatchesInfo* m) {
    memset(m, 0, sizeof(*m));
    m->src_img_idx = -1; m->dst_img_idx = -1;
}

int match_impl(MisContext* ctx, const MisFeatures* feats, int n, const MisMatchParams* p, int rank, int world, MisMatchesInfo* out) {
    MIS_CHECK(ctx, feats && p && out && n >= 1, MIS_E_INVALID, "null argument");
    MIS_CHECK(ctx, world >= 1 && rank >= 0 && rank < world, MIS_E_INVALID, "bad rank / world size");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n * n; i++) init_info(&out[i]);
    // FeaturesMatcher::operator(): all i < j with non-empty keypoint lists, dealt round-robin over ranks
    std::vector<PairDesc> pairs;
    std::vector<FeatDev> fd(n);
    size_t knn_total = 0, m_total = 0;
    for (int i = 0; i < n; i++) {
        MIS_CHECK(ctx, feats[i].n == 0 || (feats[i].desc_dtype == MIS_U8 && feats[i].desc_cols == 32), MIS_E_UNSUPPORTED,
                  "all-pairs matching supports 32-byte binary descriptors");
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
    const int np = (int)pairs.size();
    if (np == 0) return MIS_OK;
    int maxq = 0;
    for (int i = 0; i < n; i++) maxq = std::max(maxq, feats[i].n);
    DevBuf d_feats, d_pairs, d_idx, d_dist, d_matches, d_src, d_dst, d_nm, d_mask, d_scr, d_rec, d_out;
    MIS_HIP(ctx, d_feats.alloc(sizeof(FeatDev) * n));
    MIS_HIP(ctx, d_pairs.alloc(sizeof(PairDesc) * np));
    MIS_HIP(ctx, d_idx.alloc(sizeof(int) * 2 * knn_total));
    MIS_HIP(ctx, d_dist.alloc(sizeof(int) * 2 * knn_total));
    MIS_HIP(ctx, d_matches.alloc(sizeof(MisDMatch) * m_total));
    MIS_HIP(ctx, d_src.alloc(sizeof(float) * 2 * m_total));
    MIS_HIP(ctx, d_dst.alloc(sizeof(float) * 2 * m_total));
    MIS_HIP(ctx, d_nm.alloc(sizeof(int) * np));
    MIS_HIP(ctx, d_mask.alloc(m_total));
    MIS_HIP(ctx, d_scr.alloc(sizeof(float) * 8 * m_total));
    MIS_HIP(ctx, d_rec.alloc(sizeof(double) * 10 * m_total));
    MIS_HIP(ctx, d_out.alloc(sizeof(PairOut) * np));
    hipStream_t st = ctx->stream;
    MIS_HIP(ctx, hipMemcpyAsync(d_feats.p, fd.data(), sizeof(FeatDev) * n, hipMemcpyHostToDevice, st));
    MIS_HIP(ctx, hipMemcpyAsync(d_pairs.p, pairs.data(), sizeof(PairDesc) * np, hipMemcpyHostToDevice, st));
    MIS_HIP(ctx, hipMemsetAsync(d_mask.p, 0, m_total, st));
    hipLaunchKernelGGL(knn2_hamming_kernel, dim3((maxq + 255) / 256, 2 * np), dim3(256), 0, st, (const FeatDev*)d_feats.p, (const PairDesc*)d_pairs.p,
                       (int*)d_idx.p, (int*)d_dist.p);
    hipLaunchKernelGGL(ratio_union_kernel, dim3(np), dim3(1024), 0, st, (const FeatDev*)d_feats.p, (const PairDesc*)d_pairs.p, (const int*)d_idx.p,
                       (const int*)d_dist.p, 1.f - p->match_conf, (MisDMatch*)d_matches.p, (float*)d_src.p, (float*)d_dst.p, (int*)d_nm.p);
    hipLaunchKernelGGL(pair_homography_kernel, dim3(np), dim3(HB), 0, st, (const PairDesc*)d_pairs.p, (const int*)d_nm.p, (const float*)d_src.p,
                       (const float*)d_dst.p, (uint8_t*)d_mask.p, (float*)d_scr.p, (double*)d_rec.p, (PairOut*)d_out.p, p->num_matches_thresh1, p->num_matches_thresh2,
                       p->ransac_thresh, p->max_iters, p->confidence);
    MIS_HIP(ctx, hipGetLastError());
    std::vector<int> nm(np);
    std::vector<PairOut> po(np);
    std::vector<MisDMatch> hm(m_total);
    std::vector<uint8_t> hmask(m_total);
    MIS_HIP(ctx, hipMemcpyAsync(nm.data(), d_nm.p, sizeof(int) * np, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipMemcpyAsync(po.data(), d_out.p, sizeof(PairOut) * np, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipMemcpyAsync(hm.data(), d_matches.p, sizeof(MisDMatch) * m_total, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipMemcpyAsync(hmask.data(), d_mask.p, m_total, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipStreamSynchronize(st));
    // assemble MatchesInfo (host): confidence, mirror entry with H^-1 and swapped indices
    for (int k = 0; k < np; k++) {
        const PairDesc& pd = pairs[k];
        MisMatchesInfo* a = &out[pd.i * n + pd.j];
        MisMatchesInfo* b = &out[pd.j * n + pd.i];
        a->src_img_idx = pd.i; a->dst_img_idx = pd.j;
        a->n_matches = nm[k];
        a->matches = (MisDMatch*)malloc(sizeof(MisDMatch) * (size_t)(nm[k] + 1));
        memcpy(a->matches, hm.data() + pd.m_off, sizeof(MisDMatch) * (size_t)nm[k]);
        if (po[k].ran_ransac) {
            a->inliers_mask = (uint8_t*)malloc((size_t)nm[k] + 1);
            memcpy(a->inliers_mask, hmask.data() + pd.m_off, (size_t)nm[k]);
        }
        a->has_H = po[k].has_H;
        memcpy(a->H, po[k].H, sizeof(a->H));
        a->num_inliers = po[k].num_inliers;
        if (po[k].passed) {
            // Brown & Lowe confidence; > 3 means near-duplicate images and is zeroed (matchers.cpp)
            double c = a->num_inliers / (8 + 0.3 * nm[k]);
            a->confidence = c > 3. ? 0. : c;
        }
        *b = *a;
        b->src_img_idx = pd.j; b->dst_img_idx = pd.i;
        b->matches = (MisDMatch*)malloc(sizeof(MisDMatch) * (size_t)(nm[k] + 1));
        for (int q = 0; q < nm[k]; q++) {
            b->matches[q] = a->matches[q];
            b->matches[q].query_idx = a->matches[q].train_idx;
            b->matches[q].train_idx = a->matches[q].query_idx;
        }
        if (a->inliers_mask) {
            b->inliers_mask = (uint8_t*)malloc((size_t)nm[k] + 1);
            memcpy(b->inliers_mask, a->inliers_mask, (size_t)nm[k]);
        }
        if (a->has_H) invert3(a->H, b->H);
    }
    return MIS_OK;
}

}  // namespace

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
    MIS_CHECK(ctx, q->desc_dtype == MIS_U8 && q->desc_cols == 32 && t->desc_dtype == MIS_U8 && t->desc_cols == 32, MIS_E_UNSUPPORTED,
              "mis_knn2 supports 32-byte binary descriptors");
    if (q->n <= 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    FeatDev fd[2] = {{(const uint8_t*)q->descriptors, q->keypoints, q->n, q->img_w, q->img_h},
                     {(const uint8_t*)t->descriptors, t->keypoints, t->n, t->img_w, t->img_h}};
    PairDesc pd;
    pd.i = 0; pd.j = 1; pd.knn_off12 = 0; pd.knn_off21 = q->n; pd.m_off = 0; pd.cap = q->n + t->n;
    DevBuf d_feats, d_pairs, d_idx, d_dist;
    size_t tot = (size_t)q->n + (size_t)std::max(t->n, 0);
    MIS_HIP(ctx, d_feats.alloc(sizeof(fd)));
    MIS_HIP(ctx, d_pairs.alloc(sizeof(pd)));
    MIS_HIP(ctx, d_idx.alloc(sizeof(int) * 2 * tot));
    MIS_HIP(ctx, d_dist.alloc(sizeof(int) * 2 * tot));
    hipStream_t st = ctx->stream;
    MIS_HIP(ctx, hipMemcpyAsync(d_feats.p, fd, sizeof(fd), hipMemcpyHostToDevice, st));
    MIS_HIP(ctx, hipMemcpyAsync(d_pairs.p, &pd, sizeof(pd), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(knn2_hamming_kernel, dim3((q->n + 255) / 256, 1), dim3(256), 0, st, (const FeatDev*)d_feats.p, (const PairDesc*)d_pairs.p,
                       (int*)d_idx.p, (int*)d_dist.p);
    MIS_HIP(ctx, hipGetLastError());
    std::vector<int> hd(2 * (size_t)q->n);
    MIS_HIP(ctx, hipMemcpyAsync(idx2, d_idx.p, sizeof(int) * 2 * q->n, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipMemcpyAsync(hd.data(), d_dist.p, sizeof(int) * 2 * q->n, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < hd.size(); i++) dist2[i] = (float)hd[i];
    return MIS_OK;
}

extern "C" int mis_find_homography(MisContext* ctx, const float* src, const float* dst, int n, double thresh, int max_iters, double confidence,
                                   double H[9], uint8_t* mask, int* ok) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, src && dst && H && ok && n >= 0, MIS_E_INVALID, "null argument");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d_src, d_dst, d_H, d_mask, d_scr, d_rec, d_ok;
    size_t nn = (size_t)std::max(n, 1);
    MIS_HIP(ctx, d_src.alloc(sizeof(float) * 2 * nn));
    MIS_HIP(ctx, d_dst.alloc(sizeof(float) * 2 * nn));
    MIS_HIP(ctx, d_H.alloc(sizeof(double) * 9));
    MIS_HIP(ctx, d_mask.alloc(nn));
    MIS_HIP(ctx, d_scr.alloc(sizeof(float) * 4 * nn));
    MIS_HIP(ctx, d_rec.alloc(sizeof(double) * 10 * nn));
    MIS_HIP(ctx, d_ok.alloc(sizeof(int) * 2));
    hipStream_t st = ctx->stream;
    if (n) {
        MIS_HIP(ctx, hipMemcpyAsync(d_src.p, src, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
        MIS_HIP(ctx, hipMemcpyAsync(d_dst.p, dst, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    }
    MIS_HIP(ctx, hipMemsetAsync(d_H.p, 0, sizeof(double) * 9, st));
    hipLaunchKernelGGL(find_homography_kernel, dim3(1), dim3(HB), 0, st, (const float*)d_src.p, (const float*)d_dst.p, n, thresh, max_iters, confidence,
                       (double*)d_H.p, (uint8_t*)d_mask.p, (float*)d_scr.p, (double*)d_rec.p, (int*)d_ok.p);
    MIS_HIP(ctx, hipGetLastError());
    int oki[2] = {0, 0};
    MIS_HIP(ctx, hipMemcpyAsync(H, d_H.p, sizeof(double) * 9, hipMemcpyDeviceToHost, st));
    if (mask && n) MIS_HIP(ctx, hipMemcpyAsync(mask, d_mask.p, n, hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipMemcpyAsync(oki, d_ok.p, sizeof(oki), hipMemcpyDeviceToHost, st));
    MIS_HIP(ctx, hipStreamSynchronize(st));
    *ok = oki[0];
    return MIS_OK;
}

// myLeaveBiggestComponent (image_stitching.cpp:215-278): union-find over pairs with
// confidence >= threshold (cv::detail::DisjointSets semantics), indices of the biggest component
extern "C" int mis_leave_biggest_component(const MisMatchesInfo* pm, int n, float conf_threshold, int* indices, int* n_indices) {
    if (!pm || !indices || !n_indices || n < 1) return MIS_E_INVALID;
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
            if (pm[i * n + j].confidence < conf_threshold) continue;
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
