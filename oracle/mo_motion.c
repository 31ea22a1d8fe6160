/*
 * mo_motion.c -- ORACLE (test infrastructure): camera refinement between matching and warping.
 * Reference call sites: image_stitching/image_stitching.cpp:626-634 (camera R, t converted to CV_32F), :680-713
 * (BundleAdjusterReproj, setConfThresh, setRefinementMask, (*adjuster)(features, pairwise_matches, cameras)), :718-726
 * (waveCorrect(rmats, WAVE_CORRECT_HORIZ) on the CV_32F rotations).
 * OpenCV sources restated (recalled; OpenCV is absent offline -- PARITY UNPINNED, [uncertain] marks the weaker memories):
 *   stitching/src/motion_estimators.cpp  BundleAdjusterBase::estimate, BundleAdjusterReproj::{setUpInitialCameraParams,
 *                                        obtainRefinedCameraParams, calcError, calcJacobian}, findMaxSpanningTree, waveCorrect
 *   calib3d/src/compat_ptsetreg.cpp      CvLevMarq::update / step (solve with DECOMP_SVD)
 *   calib3d/src/calibration.cpp          cvRodrigues2 (both directions; the matrix is projected on SO(3) by an SVD first)
 *   core/src/lapack.cpp                  JacobiSVDImpl_<float|double>, SVBkSb, JacobiImpl_<float> (cv::eigen), 3x3 invert / det
 *   core/src/matmul                      the small-matrix product (len 2..4: written-out sums in the matrix type)
 * Precision follows the reference: the cameras enter with CV_32F rotations, the solver state is CV_64F, the refined rotations
 * leave as CV_32F and wave correction runs in CV_32F.  Transcendentals are libm's (cos, sin, acos, exp, log, hypot).
 * Never linked into the product.
 */
#define _GNU_SOURCE      /* sincos */
#include "mo_motion.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* convertTo(CV_32F) of one value: through a volatile, so that no optimisation level may skip the rounding */
static double round_f32(double v) { volatile float f = (float)v; return (double)f; }

/* ------------------------------------------------------------------ core/src/lapack.cpp ---------- */
/* JacobiSVDImpl_<T>: one-sided Jacobi on the n rows (length m) of At.  W = singular values (descending); the rows of At
 * become the left singular vectors, Vt the right ones.  Directions with a zero singular value are zeroed (OpenCV completes
 * them with a seeded random basis: never reached by rotation matrices or by the solver's systems).  */
#define MO_DEFINE_JACOBI_SVD(NAME, T, EPS, MINVAL)                                                                        \
    static void NAME(T* At, int astep, T* Wout, T* Vt, int vstep, int m, int n) {                                           \
        double* W = (double*)malloc(sizeof(double) * (size_t)n);                                                             \
        const int max_iter = m > 30 ? m : 30;                                                                                \
        const T eps = (T)(EPS);                                                                                              \
        for (int i = 0; i < n; i++) {                                                                                        \
            double sd = 0;                                                                                                   \
            for (int k = 0; k < m; k++) { const T t = At[i * astep + k]; sd += (double)t * t; }                              \
            W[i] = sd;                                                                                                       \
            for (int k = 0; k < n; k++) Vt[i * vstep + k] = 0;                                                               \
            Vt[i * vstep + i] = 1;                                                                                           \
        }                                                                                                                    \
        for (int iter = 0; iter < max_iter; iter++) {                                                                        \
            int changed = 0;                                                                                                 \
            for (int i = 0; i < n - 1; i++)                                                                                  \
                for (int j = i + 1; j < n; j++) {                                                                            \
                    T *Ai = At + i * astep, *Aj = At + j * astep;                                                            \
                    double a = W[i], p = 0, b = W[j];                                                                        \
                    for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];                                                  \
                    if (fabs(p) <= eps * sqrt((double)a * b)) continue;                                                      \
                    p *= 2;                                                                                                  \
                    const double beta = a - b, gamma = hypot((double)p, beta);                                               \
                    T c, s;                                                                                                  \
                    if (beta < 0) {                                                                                          \
                        const double delta = (gamma - beta) * 0.5;                                                           \
                        s = (T)sqrt(delta / gamma);                                                                          \
                        c = (T)(p / (gamma * s * 2));                                                                        \
                    } else {                                                                                                 \
                        c = (T)sqrt((gamma + beta) / (gamma * 2));                                                           \
                        s = (T)(p / (gamma * c * 2));                                                                        \
                    }                                                                                                        \
                    a = b = 0;                                                                                               \
                    for (int k = 0; k < m; k++) {                                                                            \
                        const T t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];                                     \
                        Ai[k] = t0; Aj[k] = t1;                                                                              \
                        a += (double)t0 * t0; b += (double)t1 * t1;                                                          \
                    }                                                                                                        \
                    W[i] = a; W[j] = b;                                                                                      \
                    changed = 1;                                                                                             \
                    T *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;                                                            \
                    for (int k = 0; k < n; k++) {                                                                            \
                        const T t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k];                                     \
                        Vi[k] = t0; Vj[k] = t1;                                                                              \
                    }                                                                                                        \
                }                                                                                                            \
            if (!changed) break;                                                                                             \
        }                                                                                                                    \
        for (int i = 0; i < n; i++) {                                                                                        \
            double sd = 0;                                                                                                   \
            for (int k = 0; k < m; k++) { const T t = At[i * astep + k]; sd += (double)t * t; }                              \
            W[i] = sqrt(sd);                                                                                                 \
        }                                                                                                                    \
        for (int i = 0; i < n - 1; i++) {                                                                                    \
            int j = i;                                                                                                       \
            for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;                                                          \
            if (i != j) {                                                                                                    \
                double tw = W[i]; W[i] = W[j]; W[j] = tw;                                                                    \
                for (int k = 0; k < m; k++) { const T t = At[i * astep + k]; At[i * astep + k] = At[j * astep + k]; At[j * astep + k] = t; } \
                for (int k = 0; k < n; k++) { const T t = Vt[i * vstep + k]; Vt[i * vstep + k] = Vt[j * vstep + k]; Vt[j * vstep + k] = t; } \
            }                                                                                                                \
        }                                                                                                                    \
        for (int i = 0; i < n; i++) {                                                                                        \
            Wout[i] = (T)W[i];                                                                                               \
            const T s = (T)(W[i] > (MINVAL) ? 1 / W[i] : 0.);                                                                \
            for (int k = 0; k < m; k++) At[i * astep + k] *= s;                                                              \
        }                                                                                                                    \
        free(W);                                                                                                             \
    }
MO_DEFINE_JACOBI_SVD(jacobi_svd_f64, double, DBL_EPSILON * 10, DBL_MIN)
MO_DEFINE_JACOBI_SVD(jacobi_svd_f32, float, FLT_EPSILON * 2, FLT_MIN)

/* SVD::compute(R, w, u, vt) of a 3x3 matrix: the routine works on the transpose (its rows = the columns of R) */
#define MO_DEFINE_SVD3(NAME, T, JSVD)                                                                  \
    static void NAME(const T* R, T* u, T* w, T* vt) {                                                  \
        T At[9];                                                                                       \
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i * 3 + j] = R[j * 3 + i];          \
        JSVD(At, 3, w, vt, 3, 3, 3);                                                                   \
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) u[j * 3 + i] = At[i * 3 + j];          \
    }
MO_DEFINE_SVD3(svd3_f64, double, jacobi_svd_f64)
MO_DEFINE_SVD3(svd3_f32, float, jacobi_svd_f32)

/* 3x3 product: the written-out sums of the small-matrix path, in the matrix type ([uncertain] for CV_32F: float sums) */
#define MO_DEFINE_MUL3(NAME, T)                                                                                           \
    static void NAME(const T* a, const T* b, T* o) {                                                                        \
        T t[9];                                                                                                             \
        for (int i = 0; i < 3; i++)                                                                                         \
            for (int j = 0; j < 3; j++) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];  \
        memcpy(o, t, sizeof(t));                                                                                            \
    }
MO_DEFINE_MUL3(mul3_f64, double)
MO_DEFINE_MUL3(mul3_f32, float)

/* determinant / inverse of a 3x3 matrix: cofactors in double for either type, the result stored in the matrix type */
#define MO_DEFINE_INV3(DET, INV, T)                                                                                        \
    static double DET(const T* a) {                                                                                          \
        return a[0] * ((double)a[4] * a[8] - (double)a[5] * a[7]) - a[1] * ((double)a[3] * a[8] - (double)a[5] * a[6]) +    \
               a[2] * ((double)a[3] * a[7] - (double)a[4] * a[6]);                                                          \
    }                                                                                                                        \
    static int INV(const T* a, T* o) {                                                                                       \
        double d = DET(a);                                                                                                   \
        if (d == 0) return 0;                                                                                                \
        d = 1. / d;                                                                                                          \
        const double t[9] = {((double)a[4] * a[8] - (double)a[5] * a[7]) * d, ((double)a[2] * a[7] - (double)a[1] * a[8]) * d, \
                             ((double)a[1] * a[5] - (double)a[2] * a[4]) * d, ((double)a[5] * a[6] - (double)a[3] * a[8]) * d, \
                             ((double)a[0] * a[8] - (double)a[2] * a[6]) * d, ((double)a[2] * a[3] - (double)a[0] * a[5]) * d, \
                             ((double)a[3] * a[7] - (double)a[4] * a[6]) * d, ((double)a[1] * a[6] - (double)a[0] * a[7]) * d, \
                             ((double)a[0] * a[4] - (double)a[1] * a[3]) * d};                                              \
        for (int i = 0; i < 9; i++) o[i] = (T)t[i];                                                                          \
        return 1;                                                                                                            \
    }
MO_DEFINE_INV3(det3_f64, inv3_f64, double)
MO_DEFINE_INV3(det3_f32, inv3_f32, float)

/* cv::solve(A, b, x, DECOMP_SVD), square: SVD of A, back substitution with the threshold 2 eps sum(w) (SVBkSb) */
static void solve_svd(const double* A, const double* b, double* x, int n) {
    double* At = (double*)malloc(sizeof(double) * (size_t)n * n * 2 + sizeof(double) * (size_t)n);
    double *Vt = At + (size_t)n * n, *w = Vt + (size_t)n * n;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) At[i * n + j] = A[j * n + i];
    jacobi_svd_f64(At, n, w, Vt, n, n, n);
    double thr = 0;
    for (int i = 0; i < n; i++) thr += w[i];
    thr *= 2 * DBL_EPSILON;
    for (int k = 0; k < n; k++) x[k] = 0;
    for (int i = 0; i < n; i++) {
        if (!(w[i] > thr)) continue;
        double s = 0;
        for (int k = 0; k < n; k++) s += At[i * n + k] * b[k];
        s /= w[i];
        for (int k = 0; k < n; k++) x[k] += s * Vt[i * n + k];
    }
    free(At);
}

/* normL2Sqr<double, double>: groups of four, as the unrolled loop of core's stat code adds them */
static double l2sqr(const double* a, int n) {
    double s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) s += a[i] * a[i] + a[i + 1] * a[i + 1] + a[i + 2] * a[i + 2] + a[i + 3] * a[i + 3];
    for (; i < n; i++) s += a[i] * a[i];
    return s;
}
static double l2sqr_diff(const double* a, const double* b, int n) {
    double s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        const double v0 = a[i] - b[i], v1 = a[i + 1] - b[i + 1], v2 = a[i + 2] - b[i + 2], v3 = a[i + 3] - b[i + 3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; i++) { const double v = a[i] - b[i]; s += v * v; }
    return s;
}

/* ------------------------------------------------------------------ cvRodrigues2 ------------------ */
static void rodrigues_vec_to_mat(const double* r, double* R) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; memcpy(R, I, sizeof(I)); return; }
    /* cos and sin of one angle: ONE sincos call on both sides (gcc merges the two calls into sincos at -O2, clang keeps cos and sin,
     * and glibc's sincos is not bit-identical to its cos / sin for every argument: a refined camera moved by one float ulp) */
    double c, s;
    sincos(theta, &s, &c);
    const double c1 = 1. - c, it = 1 / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx[k];
}
static void rodrigues_mat_to_vec(const double* Rin, double* r) {
    double U[9], W[3], Vt[9], R[9];
    svd3_f64(Rin, U, W, Vt);                      /* the matrix is projected on SO(3) first */
    mul3_f64(U, Vt, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
        t = (R[4] + 1) * 0.5; ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* ------------------------------------------------------------------ BundleAdjusterReproj ---------- */
typedef struct { int i, j, first, count; } BaEdge;
typedef struct {
    int n, nedges, total;
    BaEdge* edges;
    float* obs;            /* total x 4: x1, y1 (image i), x2, y2 (image j) */
    double* cam;           /* 7 per camera: focal, ppx, ppy, aspect, rvec */
    int on[7];             /* which of the 7 parameters are differentiated */
} Ba;

static void ba_calc_error(const Ba* ba, double* err) {
    for (int e = 0; e < ba->nedges; e++) {
        const int i = ba->edges[e].i, j = ba->edges[e].j;
        const double* ci = ba->cam + 7 * i;
        const double* cj = ba->cam + 7 * j;
        double R1[9], R2[9], R2i[9], K1i[9], H[9];
        rodrigues_vec_to_mat(ci + 4, R1);
        rodrigues_vec_to_mat(cj + 4, R2);
        const double K1[9] = {ci[0], 0, ci[1], 0, ci[0] * ci[3], ci[2], 0, 0, 1}, K2[9] = {cj[0], 0, cj[1], 0, cj[0] * cj[3], cj[2], 0, 0, 1};
        inv3_f64(K1, K1i);
        inv3_f64(R2, R2i);
        mul3_f64(K2, R2i, H); mul3_f64(H, R1, H); mul3_f64(H, K1i, H);        /* H = K2 * R2^-1 * R1 * K1^-1 */
        for (int k = 0; k < ba->edges[e].count; k++) {
            const int m = ba->edges[e].first + k;
            const float* o = ba->obs + 4 * (size_t)m;
            const double x = H[0] * o[0] + H[1] * o[1] + H[2], y = H[3] * o[0] + H[4] * o[1] + H[5], z = H[6] * o[0] + H[7] * o[1] + H[8];
            err[2 * m] = o[2] - x / z;
            err[2 * m + 1] = o[3] - y / z;
        }
    }
}

static void ba_calc_jacobian(Ba* ba, double* J, int ncols, double* e1, double* e2) {
    const double step = 1e-4;
    const int nerr = 2 * ba->total;
    for (int i = 0; i < ba->n; i++)
        for (int j = 0; j < 7; j++) {
            if (!ba->on[j]) continue;
            const double val = ba->cam[7 * i + j];
            ba->cam[7 * i + j] = val - step; ba_calc_error(ba, e1);
            ba->cam[7 * i + j] = val + step; ba_calc_error(ba, e2);
            for (int k = 0; k < nerr; k++) J[(size_t)k * ncols + 7 * i + j] = (e2[k] - e1[k]) / (2 * step);
            ba->cam[7 * i + j] = val;
        }
}

/* findMaxSpanningTree: Kruskal on num_inliers (heaviest first; [uncertain] std::sort there is unstable: ties keep the order of
 * generation here), the centre = the vertex with the smallest greatest distance to a leaf */
static int spanning_tree_center(int n, const MoMatchesInfo* pm) {
    typedef struct { int from, to; float w; } GE;
    GE* all = (GE*)malloc(sizeof(GE) * (size_t)(n * n + 1));
    int ne = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (pm[i * n + j].has_H) { all[ne].from = i; all[ne].to = j; all[ne].w = (float)pm[i * n + j].num_inliers; ne++; }
    for (int a = 1; a < ne; a++) {          /* stable insertion sort, descending weight */
        const GE g = all[a];
        int b = a - 1;
        while (b >= 0 && all[b].w < g.w) { all[b + 1] = all[b]; b--; }
        all[b + 1] = g;
    }
    int* parent = (int*)malloc(sizeof(int) * (size_t)n * 4);
    int *rnk = parent + n, *power = parent + 2 * n, *maxd = parent + 3 * n;
    uint8_t* adj = (uint8_t*)calloc((size_t)n * n, 1);
    for (int i = 0; i < n; i++) { parent[i] = i; rnk[i] = 0; power[i] = 0; maxd[i] = 0; }
    for (int k = 0; k < ne; k++) {
        int c1 = all[k].from, c2 = all[k].to;
        while (parent[c1] != c1) c1 = parent[c1];
        while (parent[c2] != c2) c2 = parent[c2];
        if (c1 == c2) continue;
        if (rnk[c1] < rnk[c2]) parent[c1] = c2; else if (rnk[c2] < rnk[c1]) parent[c2] = c1; else { parent[c1] = c2; rnk[c2]++; }
        adj[all[k].from * n + all[k].to] = adj[all[k].to * n + all[k].from] = 1;
        power[all[k].from]++; power[all[k].to]++;
    }
    int* dist = (int*)malloc(sizeof(int) * (size_t)n * 2);
    int* q = dist + n;
    for (int leaf = 0; leaf < n; leaf++) {
        if (power[leaf] != 1) continue;
        for (int i = 0; i < n; i++) dist[i] = -1;
        int qh = 0, qt = 0;
        q[qt++] = leaf; dist[leaf] = 0;
        while (qh < qt) {
            const int v = q[qh++];
            for (int u = 0; u < n; u++) if (adj[v * n + u] && dist[u] < 0) { dist[u] = dist[v] + 1; q[qt++] = u; }
        }
        for (int i = 0; i < n; i++) if (dist[i] > maxd[i]) maxd[i] = dist[i];
    }
    int best = 0;
    for (int i = 1; i < n; i++) if (maxd[i] < maxd[best]) best = i;
    free(dist); free(adj); free(parent); free(all);
    return best;
}

int mo_bundle_adjust_reproj(int n, const MoFeatures* features, const MoMatchesInfo* pairwise, float conf_thresh, const char* refine_mask,
                            MoCamera* cameras, int* iterations) {
    if (n < 2) return -1;
    Ba ba;
    memset(&ba, 0, sizeof(ba));
    ba.n = n;
    const int r00 = !refine_mask || refine_mask[0] == 'x', r02 = !refine_mask || refine_mask[2] == 'x', r11 = !refine_mask || refine_mask[3] == 'x',
              r12 = !refine_mask || refine_mask[4] == 'x';
    /* parameter order: focal (0,0), ppx (0,2), ppy (1,2), aspect (1,1), rotation vector (always) */
    ba.on[0] = r00; ba.on[1] = r02; ba.on[2] = r12; ba.on[3] = r11; ba.on[4] = ba.on[5] = ba.on[6] = 1;
    ba.cam = (double*)malloc(sizeof(double) * (size_t)n * 7);
    /* setUpInitialCameraParams: the CV_32F rotation through its SVD (u * vt, sign fixed), then Rodrigues -> CV_32F vector */
    for (int i = 0; i < n; i++) {
        double* c = ba.cam + 7 * i;
        c[0] = cameras[i].focal; c[1] = cameras[i].ppx; c[2] = cameras[i].ppy; c[3] = cameras[i].aspect;
        float Rf[9], u[9], w[3], vt[9], R[9];
        for (int k = 0; k < 9; k++) Rf[k] = (float)cameras[i].R[k];
        svd3_f32(Rf, u, w, vt);
        mul3_f32(u, vt, R);
        if (det3_f32(R) < 0) for (int k = 0; k < 9; k++) R[k] *= -1;
        double Rd[9], rv[3];
        for (int k = 0; k < 9; k++) Rd[k] = R[k];
        rodrigues_mat_to_vec(Rd, rv);
        for (int k = 0; k < 3; k++) c[4 + k] = round_f32(rv[k]);
    }
    /* the consistent pairs and their inlier correspondences */
    ba.edges = (BaEdge*)malloc(sizeof(BaEdge) * (size_t)(n * n));
    size_t cap = 0;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) if (pairwise[i * n + j].confidence > conf_thresh) cap += (size_t)pairwise[i * n + j].n_matches;
    ba.obs = (float*)malloc(sizeof(float) * 4 * (cap + 1));
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            const MoMatchesInfo* mi = &pairwise[i * n + j];
            if (!(mi->confidence > conf_thresh)) continue;
            BaEdge* e = &ba.edges[ba.nedges++];
            e->i = i; e->j = j; e->first = ba.total; e->count = 0;
            for (int k = 0; k < mi->n_matches; k++) {
                if (!mi->inliers_mask || !mi->inliers_mask[k]) continue;
                const MoDMatch* m = &mi->matches[k];
                float* o = ba.obs + 4 * (size_t)ba.total;
                o[0] = features[i].xy[2 * m->query_idx]; o[1] = features[i].xy[2 * m->query_idx + 1];
                o[2] = features[j].xy[2 * m->train_idx]; o[3] = features[j].xy[2 * m->train_idx + 1];
                ba.total++; e->count++;
            }
        }
    if (ba.total == 0) { free(ba.cam); free(ba.edges); free(ba.obs); return -2; }
    /* CvLevMarq(n * 7, total * 2, TermCriteria(EPS + COUNT, 1000, DBL_EPSILON)) */
    const int np = n * 7, ne = ba.total * 2, max_iter = 1000;
    const double epsilon = DBL_EPSILON;
    double* param = (double*)malloc(sizeof(double) * (size_t)np * 4);
    double *prev = param + np, *JtErr = param + 2 * np, *delta = param + 3 * np;
    double* err = (double*)malloc(sizeof(double) * (size_t)ne * 3);
    double *e1 = err + ne, *e2 = err + 2 * ne;
    double* J = (double*)calloc((size_t)ne * np, sizeof(double));
    double* JtJ = (double*)malloc(sizeof(double) * (size_t)np * np * 2);
    double* A = JtJ + (size_t)np * np;
    memcpy(param, ba.cam, sizeof(double) * (size_t)np);
    enum { DONE, STARTED, CALC_J, CHECK_ERR } state = STARTED;
    int iters = 0, lambdaLg10 = -3, evals = 0;
    double prevErrNorm = DBL_MAX, errNorm = 0;
    for (;;) {
        int want_J = 0, want_err = 0, proceed = 1, do_step = 0;
        if (state == DONE) proceed = 0;
        else if (state == STARTED) { want_J = want_err = 1; state = CALC_J; }
        else if (state == CALC_J) {
            /* JtJ = J^T J, JtErr = J^T err ([uncertain] summation order of mulTransposed / gemm: plain ascending rows) */
            for (int a = 0; a < np; a++)
                for (int b = a; b < np; b++) {
                    double s = 0;
                    for (int k = 0; k < ne; k++) s += J[(size_t)k * np + a] * J[(size_t)k * np + b];
                    JtJ[(size_t)a * np + b] = JtJ[(size_t)b * np + a] = s;
                }
            for (int a = 0; a < np; a++) { double s = 0; for (int k = 0; k < ne; k++) s += J[(size_t)k * np + a] * err[k]; JtErr[a] = s; }
            memcpy(prev, param, sizeof(double) * (size_t)np);
            do_step = 1;
            if (iters == 0) prevErrNorm = sqrt(l2sqr(err, ne));
            want_err = 1; state = CHECK_ERR;
        } else {
            errNorm = sqrt(l2sqr(err, ne));
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) { do_step = 1; want_err = 1; state = CHECK_ERR; }
            else {
                lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
                const double change = sqrt(l2sqr_diff(param, prev, np)) / (sqrt(l2sqr(prev, np)) + DBL_EPSILON);
                if (++iters >= max_iter || change < epsilon) state = DONE;
                else { prevErrNorm = errNorm; memset(J, 0, sizeof(double) * (size_t)ne * np); want_J = want_err = 1; state = CALC_J; }
            }
        }
        if (do_step) {                 /* CvLevMarq::step: (JtJ with its diagonal scaled by 1 + lambda) delta = JtErr, by SVD */
            const double lambda = exp(lambdaLg10 * log(10.));
            memcpy(A, JtJ, sizeof(double) * (size_t)np * np);
            for (int a = 0; a < np; a++) A[(size_t)a * np + a] *= 1. + lambda;
            solve_svd(A, JtErr, delta, np);
            if (getenv("MO_BA_TRACE")) {      /* diagnostics: bit checksums of the step's system and its solution */
                unsigned long long ha = 0, hb = 0, hd = 0, t;
                for (size_t q = 0; q < (size_t)np * np; q++) { memcpy(&t, &A[q], 8); ha += t * (q + 1); }
                for (int q = 0; q < np; q++) { memcpy(&t, &JtErr[q], 8); hb += t * (unsigned long long)(q + 1); memcpy(&t, &delta[q], 8); hd += t * (unsigned long long)(q + 1); }
                fprintf(stderr, "[mo-ba] step lambdaLg10 %d A %016llx JtErr %016llx delta %016llx\n", lambdaLg10, ha, hb, hd);
            }
            for (int a = 0; a < np; a++) param[a] = prev[a] - delta[a];
        }
        memcpy(ba.cam, param, sizeof(double) * (size_t)np);
        if (getenv("MO_BA_TRACE")) fprintf(stderr, "[mo-ba] state %d iters %d lambdaLg10 %d prevErr %.17g err %.17g p0 %.17g p4 %.17g\n", (int)state, iters, lambdaLg10, prevErrNorm, errNorm, param[0], param[4]);
        if (!proceed || !want_err) break;
        if (want_J) ba_calc_jacobian(&ba, J, np, e1, e2);
        ba_calc_error(&ba, err);
        evals++;
    }
    if (iterations) *iterations = iters;
    int ok = 1;
    for (int a = 0; a < np; a++) if (ba.cam[a] != ba.cam[a]) ok = 0;
    if (ok) {
        /* obtainRefinedCameraParams: Rodrigues in CV_64F, the rotation stored as CV_32F */
        for (int i = 0; i < n; i++) {
            const double* c = ba.cam + 7 * i;
            cameras[i].focal = c[0]; cameras[i].ppx = c[1]; cameras[i].ppy = c[2]; cameras[i].aspect = c[3];
            double R[9];
            rodrigues_vec_to_mat(c + 4, R);
            for (int k = 0; k < 9; k++) cameras[i].R[k] = round_f32(R[k]);
        }
        /* the motion normalised to the centre image of the maximum spanning tree: R_i = R_c^-1 * R_i in CV_32F */
        const int c0 = spanning_tree_center(n, pairwise);
        float Rc[9], Rinv[9];
        for (int k = 0; k < 9; k++) Rc[k] = (float)cameras[c0].R[k];
        if (inv3_f32(Rc, Rinv))
            for (int i = 0; i < n; i++) {
                float Ri[9], Ro[9];
                for (int k = 0; k < 9; k++) Ri[k] = (float)cameras[i].R[k];
                mul3_f32(Rinv, Ri, Ro);
                for (int k = 0; k < 9; k++) cameras[i].R[k] = Ro[k];
            }
    }
    free(param); free(err); free(J); free(JtJ); free(ba.cam); free(ba.edges); free(ba.obs);
    return ok ? 0 : -3;
}

/* ------------------------------------------------------------------ waveCorrect (CV_32F) ---------- */
static float hypot_f32(float a, float b) {       /* cv::hypot(float, float) */
    a = fabsf(a); b = fabsf(b);
    if (a > b) { b /= a; return a * sqrtf(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrtf(1 + a * a); }
    return 0;
}

/* cv::eigen of a symmetric 3x3 CV_32F matrix = JacobiImpl_<float>: eigenvalues descending, eigenvectors as rows */
static void jacobi_eigen3_f32(float* A, float* W, float* V) {
    const int n = 3;
    const float eps = FLT_EPSILON;
    int i, j, k, m, indR[3], indC[3];
    float mv;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i * n + j] = 0; V[i * n + i] = 1; }
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabsf(A[n * k + m]), i = k + 2; i < n; i++) { const float val = fabsf(A[n * k + i]); if (mv < val) mv = val, m = i; }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabsf(A[k]), i = 1; i < k; i++) { const float val = fabsf(A[n * i + k]); if (mv < val) mv = val, m = i; }
            indC[k] = m;
        }
    }
    for (int iters = 0; iters < n * n * 30; iters++) {
        for (k = 0, mv = fabsf(A[indR[0]]), i = 1; i < n - 1; i++) { const float val = fabsf(A[n * i + indR[i]]); if (mv < val) mv = val, k = i; }
        int l = indR[k];
        for (i = 1; i < n; i++) { const float val = fabsf(A[n * indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
        const float p = A[n * k + l];
        if (fabsf(p) <= eps) break;
        const float y = (float)((W[l] - W[k]) * 0.5);
        float t = fabsf(y) + hypot_f32(p, y);
        float s = hypot_f32(p, t);
        const float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t; W[l] += t;
        float a0, b0;
#define MO_ROTF(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; i++) MO_ROTF(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) MO_ROTF(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) MO_ROTF(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++) MO_ROTF(V[n * k + i], V[n * l + i]);
#undef MO_ROTF
        for (j = 0; j < 2; j++) {
            const int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = fabsf(A[n * idx + m]), i = idx + 2; i < n; i++) { const float val = fabsf(A[n * idx + i]); if (mv < val) mv = val, m = i; }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabsf(A[idx]), i = 1; i < idx; i++) { const float val = fabsf(A[n * i + idx]); if (mv < val) mv = val, m = i; }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) {
            const float tw = W[m]; W[m] = W[k]; W[k] = tw;
            for (i = 0; i < n; i++) { const float tv = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tv; }
        }
    }
}

int mo_wave_correct(double* rmats, int n, int kind) {
    if (!rmats || n < 1 || (kind != 0 && kind != 1)) return -1;
    if (n <= 1) return 0;
    float* R = (float*)malloc(sizeof(float) * 9 * (size_t)n);
    for (int i = 0; i < 9 * n; i++) R[i] = (float)rmats[i];
    float moment[9] = {0};
    for (int i = 0; i < n; i++) {
        const float col[3] = {R[9 * i], R[9 * i + 3], R[9 * i + 6]};            /* col(0): the camera's x axis */
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) moment[a * 3 + b] += col[a] * col[b];
    }
    float evals[3], evecs[9];
    jacobi_eigen3_f32(moment, evals, evecs);
    float rg1[3], img_k[3] = {0, 0, 0};
    memcpy(rg1, kind == 0 ? evecs + 6 : evecs, sizeof(rg1));                   /* HORIZ: smallest eigenvalue; VERT: largest */
    for (int i = 0; i < n; i++) { img_k[0] += R[9 * i + 2]; img_k[1] += R[9 * i + 5]; img_k[2] += R[9 * i + 8]; }
    float rg0[3] = {rg1[1] * img_k[2] - rg1[2] * img_k[1], rg1[2] * img_k[0] - rg1[0] * img_k[2], rg1[0] * img_k[1] - rg1[1] * img_k[0]};
    const double rg0_norm = sqrt((double)rg0[0] * rg0[0] + (double)rg0[1] * rg0[1] + (double)rg0[2] * rg0[2]);
    if (rg0_norm <= DBL_MIN) { free(R); return 0; }
    for (int k = 0; k < 3; k++) rg0[k] = (float)(rg0[k] * (1. / rg0_norm));     /* rg0 /= norm: scaled by the reciprocal, in double */
    float rg2[3] = {rg0[1] * rg1[2] - rg0[2] * rg1[1], rg0[2] * rg1[0] - rg0[0] * rg1[2], rg0[0] * rg1[1] - rg0[1] * rg1[0]};
    double conf = 0;
    for (int i = 0; i < n; i++) {
        /* Mat::dot on CV_32F: products and sum in double */
        const float* g = kind == 0 ? rg0 : rg1;
        const double d = (double)g[0] * R[9 * i] + (double)g[1] * R[9 * i + 3] + (double)g[2] * R[9 * i + 6];
        conf += kind == 0 ? d : -d;
    }
    if (conf < 0) for (int k = 0; k < 3; k++) { rg0[k] *= -1; rg1[k] *= -1; }
    const float Rg[9] = {rg0[0], rg0[1], rg0[2], rg1[0], rg1[1], rg1[2], rg2[0], rg2[1], rg2[2]};
    for (int i = 0; i < n; i++) {
        float o[9];
        mul3_f32(Rg, R + 9 * i, o);
        for (int k = 0; k < 9; k++) rmats[9 * i + k] = o[k];
    }
    free(R);
    return 0;
}
