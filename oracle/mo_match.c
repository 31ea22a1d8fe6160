/*
 * mo_match.c -- ORACLE (test infrastructure): 2-NN + ratio test + RANSAC homography + LM re-fit.
 * Reference call sites: image_stitching/image_stitching.cpp:647, :653 (BestOf2NearestMatcher),
 * :215-278 (myLeaveBiggestComponent).  OpenCV sources restated:
 * stitching/src/matchers.cpp, calib3d/src/fundam.cpp, ptsetreg.cpp, core/src/lapack.cpp (Jacobi),
 * calib3d/src/lmsolver? (core LMSolver).  PARITY UNPINNED.  Never linked into the product.
 */
#include "mo_match.h"
#include <stdlib.h>
#include <string.h>

void mo_match_default_params(MoMatchParams* p) {
    p->match_conf = 0.32f; p->num_matches_thresh1 = 6; p->num_matches_thresh2 = 6;
    p->ransac_thresh = 3.0; p->max_iters = 2000; p->confidence = 0.995;
}

/* ---- exact 2-NN, order (distance, trainIdx) ---- */
void mo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx2, int* dist2) {
    int i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < nq; i++) {
        uint64_t a[4];
        memcpy(a, q + (size_t)i * 32, 32);
        int d0 = 1 << 30, d1 = 1 << 30, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            uint64_t b[4];
            memcpy(b, t + (size_t)j * 32, 32);
            int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
                    __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else if (d < d1) { d1 = d; i1 = j; }
        }
        idx2[2 * i] = i0; idx2[2 * i + 1] = i1; dist2[2 * i] = d0; dist2[2 * i + 1] = d1;
    }
}

void mo_knn2_l2(const float* q, int nq, const float* t, int nt, int dim, int* idx2, float* dist2) {
    int i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < nq; i++) {
        const float* a = q + (size_t)i * dim;
        float d0 = FLT_MAX, d1 = FLT_MAX; int i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            const float* b = t + (size_t)j * dim;
            float s = 0;
            for (int k = 0; k < dim; k++) { float df = a[k] - b[k]; s += df * df; }
            if (s < d0) { d1 = d0; i1 = i0; d0 = s; i0 = j; }
            else if (s < d1) { d1 = s; i1 = j; }
        }
        idx2[2 * i] = i0; idx2[2 * i + 1] = i1; dist2[2 * i] = sqrtf(d0); dist2[2 * i + 1] = sqrtf(d1);
    }
}

/* ---- core/src/lapack.cpp JacobiImpl_<double>: cv::eigen of a symmetric matrix ---- */
static double cv_hypot(double a, double b) {
    a = fabs(a); b = fabs(b);
    if (a > b) { b /= a; return a * sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
    return 0;
}

void mo_jacobi_eigen(double* A, int n, double* W, double* V) {
    const double eps = DBL_EPSILON;
    int i, j, k, m, indR[16], indC[16];
    double mv;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i * n + j] = 0; V[i * n + i] = 1; }
    int iters, maxIters = n * n * 30;
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; i++) {
                double val = fabs(A[n * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
                double val = fabs(A[n * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1) for (iters = 0; iters < maxIters; iters++) {
        for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
            double val = fabs(A[n * i + indR[i]]);
            if (mv < val) mv = val, k = i;
        }
        int l = indR[k];
        for (i = 1; i < n; i++) {
            double val = fabs(A[n * indC[i] + i]);
            if (mv < val) mv = val, k = indC[i], l = i;
        }
        double p = A[n * k + l];
        if (fabs(p) <= eps) break;
        double y = (W[l] - W[k]) * 0.5;
        double t = fabs(y) + cv_hypot(p, y);
        double s = cv_hypot(p, t);
        double c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t; W[l] += t;
        double a0, b0;
#define MO_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; i++) MO_ROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) MO_ROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) MO_ROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++) MO_ROT(V[n * k + i], V[n * l + i]);
#undef MO_ROT
        for (j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
                    double val = fabs(A[n * idx + i]);
                    if (mv < val) mv = val, m = i;
                }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
                    double val = fabs(A[n * i + idx]);
                    if (mv < val) mv = val, m = i;
                }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) {
            double tw = W[m]; W[m] = W[k]; W[k] = tw;
            for (i = 0; i < n; i++) { double tv = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tv; }
        }
    }
}

/* ---- fundam.cpp HomographyEstimatorCallback::runKernel: normalised DLT ---- */
int mo_homography_dlt(const float* M, const float* m, int count, double Hout[9]) {
    double LtL[81], W[9], V[81];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    int i, j, k;
    for (i = 0; i < count; i++) {
        cmx += m[2 * i]; cmy += m[2 * i + 1];
        cMx += M[2 * i]; cMy += M[2 * i + 1];
    }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (i = 0; i < count; i++) {
        smx += fabs(m[2 * i] - cmx); smy += fabs(m[2 * i + 1] - cmy);
        sMx += fabs(M[2 * i] - cMx); sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    for (i = 0; i < 81; i++) LtL[i] = 0;
    for (i = 0; i < count; i++) {
        double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (j = 0; j < 9; j++)
            for (k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (j = 0; j < 9; j++) for (k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    mo_jacobi_eigen(LtL, 9, W, V);
    const double* H0 = V + 72;
    double T[9], R[9];
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) {
        double s = 0;
        for (k = 0; k < 3; k++) s += invHnorm[i * 3 + k] * H0[k * 3 + j];
        T[i * 3 + j] = s;
    }
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) {
        double s = 0;
        for (k = 0; k < 3; k++) s += T[i * 3 + k] * Hnorm2[k * 3 + j];
        R[i * 3 + j] = s;
    }
    double sc = 1. / R[8];
    for (i = 0; i < 9; i++) Hout[i] = R[i] * sc;
    return 1;
}

/* ---- fundam.cpp checkSubset / haveCollinearPoints (4.5.x: only the last point is tested) ---- */
static int have_collinear(const float* p, int count) {
    int j, k, i = count - 1;
    for (j = 0; j < i; j++) {
        double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (k = 0; k < j; k++) {
            double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}
static double det3_pts(const float* p, const int* t) {
    /* determinant of [x0 y0 1; x1 y1 1; x2 y2 1] as Matx33d determinant() expands it */
    double a00 = p[2 * t[0]], a01 = p[2 * t[0] + 1], a10 = p[2 * t[1]], a11 = p[2 * t[1] + 1],
           a20 = p[2 * t[2]], a21 = p[2 * t[2] + 1];
    return a00 * (a11 * 1. - 1. * a21) - a01 * (a10 * 1. - 1. * a20) + 1. * (a10 * a21 - a11 * a20);
}
static int check_subset(const float* s, const float* d) {
    static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    if (have_collinear(s, 4) || have_collinear(d, 4)) return 0;
    int negative = 0;
    for (int i = 0; i < 4; i++) negative += det3_pts(s, tt[i]) * det3_pts(d, tt[i]) < 0;
    if (negative != 0 && negative != 4) return 0;
    return 1;
}

int mo_ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    (void)model_points; /* 4 */
    if (p < 0.) p = 0.; if (p > 1.) p = 1.;
    if (ep < 0.) ep = 0.; if (ep > 1.) ep = 1.;
    double num = 1. - p; if (num < DBL_MIN) num = DBL_MIN;
    double w = 1. - ep, w2 = w * w;
    double denom = 1. - w2 * w2;
    if (denom < DBL_MIN) return 0;
    num = mo_log_d(num);
    denom = mo_log_d(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : mo_round_d(num / denom);
}

static int find_inliers(const float* M, const float* m, int count, const double H[9], float t, uint8_t* mask) {
    float Hf[9];
    int nz = 0;
    for (int i = 0; i < 9; i++) Hf[i] = (float)H[i];
    for (int i = 0; i < count; i++) {
        float Mx = M[2 * i], My = M[2 * i + 1];
        float ww = 1.f / ((Hf[6] * Mx + Hf[7] * My) + 1.f);
        float dx = ((Hf[0] * Mx + Hf[1] * My) + Hf[2]) * ww - m[2 * i];
        float dy = ((Hf[3] * Mx + Hf[4] * My) + Hf[5]) * ww - m[2 * i + 1];
        float e = dx * dx + dy * dy;
        int f = e <= t;
        mask[i] = (uint8_t)f; nz += f;
    }
    return nz;
}

/* ---- LMSolver (OpenCV <= 4.5 calib3d/src/levmarq.cpp LMSolverImpl::run) on the 8 free
 * homography parameters, callback fundam.cpp HomographyRefineCallback ---- */
static void lm_compute(const float* M, const float* m, int count, const double* h, double* err, double* J) {
    for (int i = 0; i < count; i++) {
        double Mx = M[2 * i], My = M[2 * i + 1];
        double ww = (h[6] * Mx + h[7] * My) + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        double xi = ((h[0] * Mx + h[1] * My) + h[2]) * ww;
        double yi = ((h[3] * Mx + h[4] * My) + h[5]) * ww;
        err[2 * i] = xi - m[2 * i];
        err[2 * i + 1] = yi - m[2 * i + 1];
        if (J) {
            double* Jp = J + (size_t)i * 16;
            Jp[0] = Mx * ww; Jp[1] = My * ww; Jp[2] = ww; Jp[3] = Jp[4] = Jp[5] = 0.;
            Jp[6] = -Mx * ww * xi; Jp[7] = -My * ww * xi;
            Jp[8] = Jp[9] = Jp[10] = 0.;
            Jp[11] = Mx * ww; Jp[12] = My * ww; Jp[13] = ww;
            Jp[14] = -Mx * ww * yi; Jp[15] = -My * ww * yi;
        }
    }
}
static void lm_normal_eq(const double* J, const double* r, int rows, double* A, double* v) {
    /* A = J^T J (upper then mirrored), v = J^T r; sums run over rows in ascending order */
    for (int i = 0; i < 8; i++) {
        for (int j = i; j < 8; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) s += J[(size_t)k * 8 + i] * J[(size_t)k * 8 + j];
            A[i * 8 + j] = s; A[j * 8 + i] = s;
        }
        double s = 0;
        for (int k = 0; k < rows; k++) s += J[(size_t)k * 8 + i] * r[k];
        v[i] = s;
    }
}
static double sumsq(const double* r, int n) { double s = 0; for (int i = 0; i < n; i++) s += r[i] * r[i]; return s; }
static double maxabs(const double* r, int n) { double s = 0; for (int i = 0; i < n; i++) { double a = fabs(r[i]); if (a > s) s = a; } return s; }

/* solve / invert with DECOMP_EIG: Jacobi eigen-decomposition + SVBkSb back substitution */
static void eig_solve8(const double* Ap, const double* b, double* x) {
    double a[64], w[8], V[64];
    memcpy(a, Ap, sizeof(a));
    mo_jacobi_eigen(a, 8, w, V);
    double thr = 0;
    for (int i = 0; i < 8; i++) thr += w[i];
    thr *= DBL_EPSILON * 2;
    for (int j = 0; j < 8; j++) x[j] = 0;
    for (int i = 0; i < 8; i++) {
        double wi = w[i];
        if (fabs(wi) <= thr) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < 8; j++) s += V[i * 8 + j] * b[j];
        s *= wi;
        for (int j = 0; j < 8; j++) x[j] = x[j] + s * V[i * 8 + j];
    }
}
static void eig_invert8(const double* A, double* inv) {
    double a[64], w[8], V[64];
    memcpy(a, A, sizeof(a));
    mo_jacobi_eigen(a, 8, w, V);
    double thr = 0;
    for (int i = 0; i < 8; i++) thr += w[i];
    thr *= DBL_EPSILON * 2;
    for (int j = 0; j < 64; j++) inv[j] = 0;
    for (int i = 0; i < 8; i++) {
        double wi = w[i];
        if (fabs(wi) <= thr) continue;
        wi = 1 / wi;
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++) inv[r * 8 + c] = inv[r * 8 + c] + V[i * 8 + r] * (V[i * 8 + c] * wi);
    }
}

int mo_homography_refine_lm(const float* M, const float* m, int count, double H[9], int max_iters) {
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    int rows = 2 * count, i, iter = 0;
    double *r = (double*)malloc(sizeof(double) * rows), *rd = (double*)malloc(sizeof(double) * rows);
    double* J = (double*)malloc(sizeof(double) * (size_t)rows * 8);
    double x[8], xd[8], A[64], Ap[64], v[8], d[8], D[8], temp_d[8];
    memcpy(x, H, sizeof(x));
    lm_compute(M, m, count, x, r, J);
    double S = sumsq(r, rows);
    lm_normal_eq(J, r, rows, A, v);
    for (i = 0; i < 8; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    for (;;) {
        memcpy(Ap, A, sizeof(Ap));
        for (i = 0; i < 8; i++) Ap[i * 8 + i] += lambda * D[i];
        eig_solve8(Ap, v, d);
        for (i = 0; i < 8; i++) xd[i] = x[i] - d[i];
        lm_compute(M, m, count, xd, rd, NULL);
        double Sd = sumsq(rd, rows);
        for (i = 0; i < 8; i++) {
            double s = 0;
            for (int j = 0; j < 8; j++) s += A[i * 8 + j] * d[j];
            temp_d[i] = s * -1. + v[i] * 2.;
        }
        double dS = 0;
        for (i = 0; i < 8; i++) dS += d[i] * temp_d[i];
        double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (i = 0; i < 8; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
            if (lambda == 0) {
                eig_invert8(A, Ap);
                double maxval = DBL_EPSILON;
                for (i = 0; i < 8; i++) { double a = fabs(Ap[i * 8 + i]); if (a > maxval) maxval = a; }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof(x));
            lm_compute(M, m, count, x, r, J);
            lm_normal_eq(J, r, rows, A, v);
        }
        iter++;
        int proceed = iter < max_iters && maxabs(d, 8) >= epsx && maxabs(r, rows) >= epsf;
        if (!proceed) break;
    }
    memcpy(H, x, sizeof(x));
    free(r); free(rd); free(J);
    return iter;
}

/* ---- ptsetreg.cpp RANSACPointSetRegistrator::run + fundam.cpp findHomography tail ---- */
int mo_find_homography_ransac(const float* src, const float* dst, int n, double thresh, int max_iters,
                              double confidence, double H[9], uint8_t* mask_out, int* iters_run) {
    int result = 0, iters_done = 0;
    uint8_t* best_mask = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    uint8_t* mask = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    double best[9];
    if (thresh <= 0) thresh = 3;
    if (n < 4) goto done;
    if (n == 4) {
        memset(best_mask, 1, 4);
        result = mo_homography_dlt(src, dst, 4, best) > 0;
        goto tail;
    }
    {
        int niters = max_iters > 1 ? max_iters : 1, max_good = 0, iter;
        MoRng rng; mo_rng_init(&rng, (uint64_t)-1);
        const float t = (float)(thresh * thresh);
        for (iter = 0; iter < niters; iter++) {
            int idx[4], found = 0, attempts;
            float ms1[8], ms2[8];
            double model[9];
            for (attempts = 0; attempts < 10000; attempts++) {
                for (int i = 0; i < 4; i++) {
                    int idx_i, dup;
                    do {
                        idx_i = mo_rng_uniform(&rng, 0, n);
                        dup = 0;
                        for (int q = 0; q < i; q++) if (idx[q] == idx_i) dup = 1;
                    } while (dup);
                    idx[i] = idx_i;
                    ms1[2 * i] = src[2 * idx_i]; ms1[2 * i + 1] = src[2 * idx_i + 1];
                    ms2[2 * i] = dst[2 * idx_i]; ms2[2 * i + 1] = dst[2 * idx_i + 1];
                }
                if (!check_subset(ms1, ms2)) continue;
                found = 1;
                break;
            }
            if (!found) { if (iter == 0) { iters_done = 0; goto done; } break; }
            iters_done = iter + 1;
            if (mo_homography_dlt(ms1, ms2, 4, model) <= 0) continue;
            int good = find_inliers(src, dst, n, model, t, mask);
            if (good > (max_good > 3 ? max_good : 3)) {
                uint8_t* tmp = mask; mask = best_mask; best_mask = tmp;
                memcpy(best, model, sizeof(best));
                max_good = good;
                niters = mo_ransac_update_num_iters(confidence, (double)(n - good) / n, 4, niters);
            }
        }
        result = max_good > 0;
    }
tail:
    if (result && n > 4) {
        /* compress inliers, re-run the kernel on all of them, then LM-refine 10 iterations */
        float* s1 = (float*)malloc(sizeof(float) * 2 * (size_t)n);
        float* d1 = (float*)malloc(sizeof(float) * 2 * (size_t)n);
        int np = 0;
        for (int i = 0; i < n; i++) if (best_mask[i]) {
            s1[2 * np] = src[2 * i]; s1[2 * np + 1] = src[2 * i + 1];
            d1[2 * np] = dst[2 * i]; d1[2 * np + 1] = dst[2 * i + 1]; np++;
        }
        if (np > 0) {
            mo_homography_dlt(s1, d1, np, best);
            mo_homography_refine_lm(s1, d1, np, best, 10);
        }
        free(s1); free(d1);
    }
done:
    if (result) {
        memcpy(H, best, sizeof(best));
        if (mask_out) memcpy(mask_out, best_mask, (size_t)n);
    } else if (mask_out && n > 0) memset(mask_out, 0, (size_t)n);
    if (iters_run) *iters_run = iters_done;
    free(best_mask); free(mask);
    return result;
}

/* ---- stitching/src/matchers.cpp: CpuMatcher::match (exact 2-NN form) + BestOf2NearestMatcher::match ---- */
static void features_knn2(const MoFeatures* q, const MoFeatures* t, int* idx2, float* dist2) {
    if (q->desc_u8) {
        int* d = (int*)malloc(sizeof(int) * 2 * (size_t)(q->n > 0 ? q->n : 1));
        mo_knn2_hamming(q->desc_u8, q->n, t->desc_u8, t->n, idx2, d);
        for (int i = 0; i < 2 * q->n; i++) dist2[i] = (float)d[i];
        free(d);
    } else {
        mo_knn2_l2(q->desc_f32, q->n, t->desc_f32, t->n, q->dim, idx2, dist2);
    }
}

static double det3(const double* H) {
    return H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
}

int mo_match_pair(const MoFeatures* f1, const MoFeatures* f2, const MoMatchParams* p, MoMatchesInfo* out) {
    memset(out, 0, sizeof(*out));
    out->src_img_idx = -1; out->dst_img_idx = -1;
    int n1 = f1->n, n2 = f2->n;
    int* i12 = (int*)malloc(sizeof(int) * 2 * (size_t)(n1 + 1)); float* d12 = (float*)malloc(sizeof(float) * 2 * (size_t)(n1 + 1));
    int* i21 = (int*)malloc(sizeof(int) * 2 * (size_t)(n2 + 1)); float* d21 = (float*)malloc(sizeof(float) * 2 * (size_t)(n2 + 1));
    uint8_t* acc12 = (uint8_t*)calloc((size_t)(n1 + 1), 1);
    out->matches = (MoDMatch*)malloc(sizeof(MoDMatch) * (size_t)(n1 + n2 + 1));
    int nm = 0;
    const float ratio = 1.f - p->match_conf;
    if (n2 >= 2) {
        features_knn2(f1, f2, i12, d12);
        for (int i = 0; i < n1; i++)
            if (d12[2 * i] < ratio * d12[2 * i + 1]) {
                MoDMatch m = {i, i12[2 * i], 0, d12[2 * i]};
                out->matches[nm++] = m; acc12[i] = 1;
            }
    }
    if (n1 >= 2) {
        features_knn2(f2, f1, i21, d21);
        for (int i = 0; i < n2; i++)
            if (d21[2 * i] < ratio * d21[2 * i + 1]) {
                int t1 = i21[2 * i]; /* index into image 1 */
                if (!(acc12[t1] && i12[2 * t1] == i)) {
                    MoDMatch m = {t1, i, -1, d21[2 * i]};
                    out->matches[nm++] = m;
                }
            }
    }
    out->n_matches = nm;
    free(i12); free(d12); free(i21); free(d21); free(acc12);
    if (nm < p->num_matches_thresh1) return 0;

    float* sp = (float*)malloc(sizeof(float) * 2 * (size_t)nm);
    float* dp = (float*)malloc(sizeof(float) * 2 * (size_t)nm);
    for (int i = 0; i < nm; i++) {
        const MoDMatch* m = &out->matches[i];
        sp[2 * i] = f1->xy[2 * m->query_idx] - (float)f1->img_w * 0.5f;
        sp[2 * i + 1] = f1->xy[2 * m->query_idx + 1] - (float)f1->img_h * 0.5f;
        dp[2 * i] = f2->xy[2 * m->train_idx] - (float)f2->img_w * 0.5f;
        dp[2 * i + 1] = f2->xy[2 * m->train_idx + 1] - (float)f2->img_h * 0.5f;
    }
    out->inliers_mask = (uint8_t*)calloc((size_t)nm, 1);
    out->has_H = mo_find_homography_ransac(sp, dp, nm, p->ransac_thresh, p->max_iters, p->confidence, out->H,
                                           out->inliers_mask, &out->ransac_iters[0]);
    if (!out->has_H || fabs(det3(out->H)) < DBL_EPSILON) { free(sp); free(dp); return 0; }
    out->num_inliers = 0;
    for (int i = 0; i < nm; i++) if (out->inliers_mask[i]) out->num_inliers++;
    out->confidence = out->num_inliers / (8 + 0.3 * nm);
    out->confidence = out->confidence > 3. ? 0. : out->confidence;
    if (out->num_inliers < p->num_matches_thresh2) { free(sp); free(dp); return 0; }
    int k = 0;
    for (int i = 0; i < nm; i++) if (out->inliers_mask[i]) {
        sp[2 * k] = sp[2 * i]; sp[2 * k + 1] = sp[2 * i + 1];
        dp[2 * k] = dp[2 * i]; dp[2 * k + 1] = dp[2 * i + 1]; k++;
    }
    out->has_H = mo_find_homography_ransac(sp, dp, k, p->ransac_thresh, p->max_iters, p->confidence, out->H, NULL,
                                           &out->ransac_iters[1]);
    free(sp); free(dp);
    return 0;
}

static void invert3(const double* H, double* I) {
    /* Mat::inv() (DECOMP_LU) of a 3x3 is evaluated in closed form by OpenCV: adjugate / det */
    double d = det3(H);
    if (d != 0.) d = 1. / d;
    I[0] = (H[4] * H[8] - H[5] * H[7]) * d; I[1] = (H[2] * H[7] - H[1] * H[8]) * d; I[2] = (H[1] * H[5] - H[2] * H[4]) * d;
    I[3] = (H[5] * H[6] - H[3] * H[8]) * d; I[4] = (H[0] * H[8] - H[2] * H[6]) * d; I[5] = (H[2] * H[3] - H[0] * H[5]) * d;
    I[6] = (H[3] * H[7] - H[4] * H[6]) * d; I[7] = (H[1] * H[6] - H[0] * H[7]) * d; I[8] = (H[0] * H[4] - H[1] * H[3]) * d;
}

int mo_match_all_pairs(const MoFeatures* feats, int n, const MoMatchParams* p, MoMatchesInfo* out) {
    for (int i = 0; i < n * n; i++) { memset(&out[i], 0, sizeof(out[i])); out[i].src_img_idx = out[i].dst_img_idx = -1; }
    /* pairs are independent: they run under one parallel loop, as FeaturesMatcher::operator() runs them under
     * parallel_for_ (matchers.cpp MatchPairsBody); the 2-NN loops inside a pair then run serially (no nested teams) */
    int npairs = 0;
    int* pi = (int*)malloc(sizeof(int) * (size_t)(n * n + 1) * 2);
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (feats[i].n > 0 && feats[j].n > 0) { pi[2 * npairs] = i; pi[2 * npairs + 1] = j; npairs++; }
    int k;
#pragma omp parallel for schedule(dynamic, 1)
    for (k = 0; k < npairs; k++) {
        const int i = pi[2 * k], j = pi[2 * k + 1];
        MoMatchesInfo* a = &out[i * n + j];
        mo_match_pair(&feats[i], &feats[j], p, a);
        a->src_img_idx = i; a->dst_img_idx = j;
        MoMatchesInfo* b = &out[j * n + i];
        *b = *a;
        b->src_img_idx = j; b->dst_img_idx = i;
        b->matches = (MoDMatch*)malloc(sizeof(MoDMatch) * (size_t)(a->n_matches + 1));
        for (int q = 0; q < a->n_matches; q++) {
            b->matches[q] = a->matches[q];
            b->matches[q].query_idx = a->matches[q].train_idx;
            b->matches[q].train_idx = a->matches[q].query_idx;
        }
        if (a->inliers_mask) {
            b->inliers_mask = (uint8_t*)malloc((size_t)(a->n_matches + 1));
            memcpy(b->inliers_mask, a->inliers_mask, (size_t)a->n_matches);
        }
        if (a->has_H) invert3(a->H, b->H);
    }
    free(pi);
    return 0;
}

void mo_matches_free(MoMatchesInfo* m, int count) {
    for (int i = 0; i < count; i++) { free(m[i].matches); free(m[i].inliers_mask); m[i].matches = NULL; m[i].inliers_mask = NULL; }
}

/* ---- image_stitching.cpp:215-278 myLeaveBiggestComponent (DisjointSets from stitching/util.cpp) ---- */
static int ds_find(int* parent, int elem) {
    int set = elem;
    while (set != parent[set]) set = parent[set];
    while (elem != parent[elem]) { int next = parent[elem]; parent[elem] = set; elem = next; }
    return set;
}
int mo_leave_biggest_component(const double* confidence, int n, float conf_threshold, int* indices) {
    int* parent = (int*)malloc(sizeof(int) * 3 * (size_t)n);
    int *rank = parent + n, *size = parent + 2 * n;
    for (int i = 0; i < n; i++) { parent[i] = i; rank[i] = 0; size[i] = 1; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if (confidence[i * n + j] < conf_threshold) continue;
            int c1 = ds_find(parent, i), c2 = ds_find(parent, j);
            if (c1 != c2) {
                if (rank[c1] < rank[c2]) { parent[c1] = c2; size[c2] += size[c1]; }
                else if (rank[c2] < rank[c1]) { parent[c2] = c1; size[c1] += size[c2]; }
                else { parent[c1] = c2; rank[c2]++; size[c2] += size[c1]; }
            }
        }
    int max_comp = 0;
    for (int i = 1; i < n; i++) if (size[i] > size[max_comp]) max_comp = i;
    int k = 0;
    for (int i = 0; i < n; i++) if (ds_find(parent, i) == max_comp) indices[k++] = i;
    free(parent);
    return k;
}
