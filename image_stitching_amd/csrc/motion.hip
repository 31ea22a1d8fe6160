// motion.hip -- camera refinement between matching and warping, host logic of the library:
//   (*adjuster)(features, pairwise_matches, cameras) with BundleAdjusterReproj   image_stitching/image_stitching.cpp:681-712
//   waveCorrect(rmats, WAVE_CORRECT_HORIZ)                                       image_stitching/image_stitching.cpp:718-726
// The reference runs both on the CPU inside OpenCV (stitching/src/motion_estimators.cpp, calib3d CvLevMarq, core
// JacobiSVD / Jacobi eigen, calib3d Rodrigues); restated here from the published algorithms with the reference's precisions: the
// cameras' rotations are CV_32F (image_stitching.cpp:626-634), the solver state is CV_64F, the refined rotations leave as CV_32F
// and wave correction runs in CV_32F.  PARITY UNPINNED (OpenCV absent offline); bit-exact against the oracle's independent
// restatement (oracle/mo_motion.c, tests/test_motion_gpu.py).
// No kernels: the data is a few thousand inlier correspondences and 7 parameters per camera.
#include "common.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- small dense linear algebra ---
struct Mat {
    int r = 0, c = 0;
    std::vector<double> v;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.) {}
    double& operator()(int i, int j) { return v[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return v[(size_t)i * c + j]; }
};

// 3x3 product: the written-out sums of OpenCV's small-matrix path, in the matrix type
template <typename T>
void mul3(const T* a, const T* b, T* o) {   // o = a * b (3x3)
    T t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    memcpy(o, t, sizeof(t));
}
// determinant / inverse: cofactors in double for either matrix type, the result stored in the matrix type (cv::determinant / cv::invert, 3x3)
template <typename T>
double det3(const T* a) {
    return a[0] * ((double)a[4] * a[8] - (double)a[5] * a[7]) - a[1] * ((double)a[3] * a[8] - (double)a[5] * a[6]) + a[2] * ((double)a[3] * a[7] - (double)a[4] * a[6]);
}
template <typename T>
bool inv3(const T* a, T* o) {
    double d = det3(a);
    if (d == 0) return false;
    d = 1. / d;
    const double t[9] = {((double)a[4] * a[8] - (double)a[5] * a[7]) * d, ((double)a[2] * a[7] - (double)a[1] * a[8]) * d, ((double)a[1] * a[5] - (double)a[2] * a[4]) * d,
                         ((double)a[5] * a[6] - (double)a[3] * a[8]) * d, ((double)a[0] * a[8] - (double)a[2] * a[6]) * d, ((double)a[2] * a[3] - (double)a[0] * a[5]) * d,
                         ((double)a[3] * a[7] - (double)a[4] * a[6]) * d, ((double)a[1] * a[6] - (double)a[0] * a[7]) * d, ((double)a[0] * a[4] - (double)a[1] * a[3]) * d};
    for (int i = 0; i < 9; i++) o[i] = (T)t[i];
    return true;
}

template <typename T> struct SvdTraits;
template <> struct SvdTraits<double> { static double eps() { return DBL_EPSILON * 10; } static double minval() { return DBL_MIN; } };
template <> struct SvdTraits<float> { static float eps() { return FLT_EPSILON * 2; } static double minval() { return FLT_MIN; } };

// core/src/lapack.cpp JacobiSVDImpl_<T>: one-sided Jacobi on the rows of At (n rows of length m); W = singular
// values (descending), rows of At become the left vectors scaled to unit length, Vt the right vectors.  The squared norms and
// the rotation angle are computed in double for either T, the rotations themselves in T.
template <typename T>
void jacobi_svd(T* At, int astep, T* Wout, T* Vt, int vstep, int m, int n) {
    const T eps = SvdTraits<T>::eps();
    const double minval = SvdTraits<T>::minval();
    std::vector<double> W(n);
    const int max_iter = std::max(m, 30);
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const T t = At[i * astep + k]; sd += (double)t * t; }
        W[i] = sd;
        for (int k = 0; k < n; k++) Vt[i * vstep + k] = 0;
        Vt[i * vstep + i] = 1;
    }
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                T *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot((double)p, beta);
                T c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (T)std::sqrt(delta / gamma);
                    c = (T)(p / (gamma * s * 2));
                } else {
                    c = (T)std::sqrt((gamma + beta) / (gamma * 2));
                    s = (T)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const T t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                T *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                for (int k = 0; k < n; k++) {
                    const T t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const T t = At[i * astep + k]; sd += (double)t * t; }
        W[i] = std::sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            for (int k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
            for (int k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
        }
    }
    for (int i = 0; i < n; i++) {
        Wout[i] = (T)W[i];
        const T s = (T)(W[i] > minval ? 1 / W[i] : 0.);   // (degenerate directions: zeroed; the random completion of
        for (int k = 0; k < m; k++) At[i * astep + k] *= s;   //  OpenCV only matters for full bases, not for solve)
    }
}

// cv::solve(A, b, x, DECOMP_SVD) for a square system: SVD of A, then SVBkSb with the threshold 2 eps sum(w)
bool solve_svd(const Mat& A, const std::vector<double>& b, std::vector<double>& x) {
    const int n = A.r;
    // JacobiSVD works on the transposed matrix: rows of At = columns of A; A = U diag(w) Vt with At rows -> u_i
    Mat At(n, n), Vt(n, n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) At(i, j) = A(j, i);
    std::vector<double> w(n);
    jacobi_svd(At.v.data(), n, w.data(), Vt.v.data(), n, n, n);
    // after the call: At(i, :) = i-th left singular vector (as a row), Vt(i, :) = i-th right singular vector
    double thr = 0;
    for (int i = 0; i < n; i++) thr += w[i];
    thr *= 2 * DBL_EPSILON;
    x.assign(n, 0.);
    for (int i = 0; i < n; i++) {
        if (!(w[i] > thr)) continue;
        double s = 0;
        for (int k = 0; k < n; k++) s += At(i, k) * b[k];
        s /= w[i];
        for (int k = 0; k < n; k++) x[k] += s * Vt(i, k);
    }
    return true;
}

// 3x3 SVD through the same routine: R = U diag(w) Vt  (u, vt as 3x3 row-major)
template <typename T>
void svd3(const T* R, T* u, T* w, T* vt) {
    T At[9], Vt[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i * 3 + j] = R[j * 3 + i];
    jacobi_svd(At, 3, w, Vt, 3, 3, 3);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { u[j * 3 + i] = At[i * 3 + j]; vt[i * 3 + j] = Vt[i * 3 + j]; }
}

// calib3d Rodrigues: rotation vector -> matrix
void rodrigues_to_mat(const double* r, double* R) {
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) { double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; memcpy(R, I, sizeof(I)); return; }
    // cos and sin of one angle through ONE sincos call, as the oracle does: gcc merges separate calls into sincos, clang does not,
    // and glibc's sincos differs from its cos / sin in the last place for some arguments (found on a 6-camera scene: one float ulp in R)
    double c, s;
    ::sincos(theta, &s, &c);
    const double c1 = 1. - c, it = 1 / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx[k];
}
// matrix -> rotation vector (cvRodrigues2 projects the matrix on SO(3) through its SVD first)
void rodrigues_to_vec(const double* Rin, double* r) {
    double U[9], W[3], Vt[9], R[9];
    svd3(Rin, U, W, Vt);
    mul3(U, Vt, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = std::sqrt(std::max(t, 0.));
        t = (R[4] + 1) * 0.5; ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

// Host threads for the two loops that make bundle adjustment slow at 16 x 4K (round 4: 3.7 s per job on one thread; every item is
// computed as before -- each sum walks its terms in the same order -- so the results are the same bits)
template <class F>
static void ba_parallel_for(int n, F f) {
    const int nt = std::max(1, std::min<int>({n, 16, (int)std::thread::hardware_concurrency()}));
    if (nt == 1) { for (int i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([=]() { for (int i = t; i < n; i += nt) f(i); });
    for (auto& x : th) x.join();
}

// ---------------------------------------------------------------- CvLevMarq ---------------------
// calib3d compat_ptsetreg.cpp: the state machine of the Levenberg-Marquardt solver BundleAdjusterBase drives
struct LevMarq {
    enum { DONE = 0, STARTED = 1, CALC_J = 2, CHECK_ERR = 3 };
    int nparams, nerrs, max_iter, state = STARTED, iters = 0, lambdaLg10 = -3;
    double epsilon, prevErrNorm = DBL_MAX, errNorm = 0;
    std::vector<double> param, prevParam, err, JtErr;
    Mat J, JtJ;
    std::vector<std::vector<std::pair<int, int>>> col_rows;      // optional: per column of J, the row ranges [first, second) that can be non-zero
    LevMarq(int np, int ne, int maxit, double eps) : nparams(np), nerrs(ne), max_iter(maxit), epsilon(eps), param(np, 0.), prevParam(np, 0.), err(ne, 0.), JtErr(np, 0.), J(ne, np), JtJ(np, np) {}
    void step() {
        const double lambda = std::exp(lambdaLg10 * std::log(10.));
        Mat A = JtJ;
        for (int i = 0; i < nparams; i++) for (int j = 0; j < i; j++) A(i, j) = A(j, i);   // completeSymm (upper -> lower)
        for (int i = 0; i < nparams; i++) A(i, i) *= 1. + lambda;
        std::vector<double> d;
        solve_svd(A, JtErr, d);
        if (getenv("MIS_BA_TRACE")) {      // diagnostics: bit checksums of the step's system and its solution (the oracle prints the same)
            unsigned long long ha = 0, hb = 0, hd = 0, t;
            for (size_t q = 0; q < A.v.size(); q++) { memcpy(&t, &A.v[q], 8); ha += t * (q + 1); }
            for (int q = 0; q < nparams; q++) { memcpy(&t, &JtErr[q], 8); hb += t * (unsigned long long)(q + 1); memcpy(&t, &d[q], 8); hd += t * (unsigned long long)(q + 1); }
            fprintf(stderr, "[ba] step lambdaLg10 %d A %016llx JtErr %016llx delta %016llx\n", lambdaLg10, ha, hb, hd);
        }
        for (int i = 0; i < nparams; i++) param[i] = prevParam[i] - d[i];
    }
    // normL2Sqr<double, double>: groups of four, as the unrolled loop of core's stat code adds them
    static double l2sqr(const double* a, const double* b, int n) {
        double s = 0;
        int i = 0;
        for (; i <= n - 4; i += 4) {
            const double v0 = a[i] - (b ? b[i] : 0.), v1 = a[i + 1] - (b ? b[i + 1] : 0.), v2 = a[i + 2] - (b ? b[i + 2] : 0.), v3 = a[i + 3] - (b ? b[i + 3] : 0.);
            s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
        }
        for (; i < n; i++) { const double v = a[i] - (b ? b[i] : 0.); s += v * v; }
        return s;
    }
    static double norm2(const std::vector<double>& a) { return std::sqrt(l2sqr(a.data(), nullptr, (int)a.size())); }
    // returns proceed; want_J / want_err tell the caller what to compute for `param`
    bool update(bool& want_J, bool& want_err) {
        want_J = want_err = false;
        if (state == DONE) return false;
        if (state == STARTED) {
            std::fill(J.v.begin(), J.v.end(), 0.); std::fill(err.begin(), err.end(), 0.);
            want_J = want_err = true; state = CALC_J;
            return true;
        }
        if (state == CALC_J) {
            // JtJ = J^T J (upper triangle), JtErr = J^T err: every entry the same sequential sum over the rows of J as before, one row of
            // JtJ per host thread.  col_rows (set by the caller): per column the row ranges outside which the column is exactly zero -- a
            // camera's parameters only move the errors of the pairs the camera is part of, and a central difference of two identical
            // evaluations is +0 --; a product with such a zero is +-0 and adding it leaves a sum that is not -0 unchanged, so the sums
            // walk the intersection of the two columns' ranges only (16 cameras: 60 x fewer terms).  Without col_rows: all rows, read
            // from a transposed copy (J's columns are 8 nparams bytes apart).
            if (!col_rows.empty()) {
                ba_parallel_for(nparams, [&](int i) {
                    const auto& ri = col_rows[i];
                    for (int j = i; j < nparams; j++) {
                        const auto& rj = col_rows[j];
                        double s = 0;
                        size_t a = 0, b = 0;
                        while (a < ri.size() && b < rj.size()) {      // sorted, disjoint ranges: their intersection in row order
                            const int lo = std::max(ri[a].first, rj[b].first), hi = std::min(ri[a].second, rj[b].second);
                            for (int k = lo; k < hi; k++) s += J(k, i) * J(k, j);
                            if (ri[a].second < rj[b].second) a++; else b++;
                        }
                        JtJ(i, j) = s;
                    }
                    double s = 0;
                    for (const auto& r : ri)
                        for (int k = r.first; k < r.second; k++) s += J(k, i) * err[k];
                    JtErr[i] = s;
                });
            } else {
                const size_t ne = (size_t)nerrs;
                std::vector<double> Jt((size_t)nparams * ne);
                ba_parallel_for(nparams, [&](int i) { double* o = Jt.data() + (size_t)i * ne; for (int k = 0; k < nerrs; k++) o[k] = J(k, i); });
                ba_parallel_for(nparams, [&](int i) {
                    const double* a = Jt.data() + (size_t)i * ne;
                    for (int j = i; j < nparams; j++) {
                        const double* b = Jt.data() + (size_t)j * ne;
                        double s = 0;
                        for (int k = 0; k < nerrs; k++) s += a[k] * b[k];
                        JtJ(i, j) = s;
                    }
                    double s = 0;
                    for (int k = 0; k < nerrs; k++) s += a[k] * err[k];
                    JtErr[i] = s;
                });
            }
            prevParam = param;
            step();
            if (iters == 0) prevErrNorm = norm2(err);
            std::fill(err.begin(), err.end(), 0.);
            want_err = true; state = CHECK_ERR;
            return true;
        }
        errNorm = norm2(err);
        if (errNorm > prevErrNorm) {
            if (++lambdaLg10 <= 16) {
                step();
                std::fill(err.begin(), err.end(), 0.);
                want_err = true; state = CHECK_ERR;
                return true;
            }
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        // cvNorm(param, prevParam, CV_RELATIVE_L2)
        const double change = std::sqrt(l2sqr(param.data(), prevParam.data(), nparams)) / (std::sqrt(l2sqr(prevParam.data(), nullptr, nparams)) + DBL_EPSILON);
        if (++iters >= max_iter || change < epsilon) { state = DONE; return true; }
        prevErrNorm = errNorm;
        std::fill(J.v.begin(), J.v.end(), 0.);
        want_J = want_err = true; state = CALC_J;
        return true;
    }
};

// ---------------------------------------------------------------- BundleAdjusterReproj ----------
struct Edge { int i, j; };
struct Obs { float x1, y1, x2, y2; };   // inlier correspondence of an edge: keypoint of image i, keypoint of image j

struct Adjuster {
    int n = 0;
    std::vector<Edge> edges;
    std::vector<std::vector<Obs>> obs;    // per edge
    int total = 0;
    std::vector<double> cam;              // 7 per camera: focal, ppx, ppy, aspect, rvec
    uint8_t refine[5] = {1, 1, 1, 1, 1};   // focal, skew (unused), ppx, aspect, ppy  -- ba_refine_mask "xxxxx"

    std::vector<int> edge_m;                  // first observation of every edge (set by index_edges)
    std::vector<std::vector<int>> cam_edges;  // per camera: the edges it is part of, ascending
    void index_edges() {
        edge_m.assign(edges.size(), 0);
        cam_edges.assign(n, {});
        int m = 0;
        for (size_t e = 0; e < edges.size(); e++) { edge_m[e] = m; m += (int)obs[e].size(); cam_edges[edges[e].i].push_back((int)e); cam_edges[edges[e].j].push_back((int)e); }
    }
    void calc_error(std::vector<double>& err) const { calc_error_at(cam, err, nullptr); }
    // only: the edges to evaluate (the other rows of err are left as they are); nullptr: all, err is resized and zeroed first
    void calc_error_at(const std::vector<double>& cam, std::vector<double>& err, const std::vector<int>* only) const {
        if (!only) err.assign((size_t)total * 2, 0.);
        const size_t ne = only ? only->size() : edges.size();
        for (size_t q = 0; q < ne; q++) {
            const size_t e = only ? (size_t)(*only)[q] : q;
            int m = edge_m[e];
            const int i = edges[e].i, j = edges[e].j;
            const double f1 = cam[i * 7], f2 = cam[j * 7], ppx1 = cam[i * 7 + 1], ppx2 = cam[j * 7 + 1], ppy1 = cam[i * 7 + 2], ppy2 = cam[j * 7 + 2];
            const double a1 = cam[i * 7 + 3], a2 = cam[j * 7 + 3];
            double R1[9], R2[9], R2i[9];
            rodrigues_to_mat(&cam[i * 7 + 4], R1);
            rodrigues_to_mat(&cam[j * 7 + 4], R2);
            const double K1[9] = {f1, 0, ppx1, 0, f1 * a1, ppy1, 0, 0, 1}, K2[9] = {f2, 0, ppx2, 0, f2 * a2, ppy2, 0, 0, 1};
            double K1i[9], H[9];
            inv3(K1, K1i);
            inv3(R2, R2i);
            mul3(K2, R2i, H); mul3(H, R1, H); mul3(H, K1i, H);      // H1to2 = K2 * R2^-1 * R1 * K1^-1
            for (const Obs& o : obs[e]) {
                const double x = H[0] * o.x1 + H[1] * o.y1 + H[2], y = H[3] * o.x1 + H[4] * o.y1 + H[5], z = H[6] * o.x1 + H[7] * o.y1 + H[8];
                err[2 * m] = o.x2 - x / z;
                err[2 * m + 1] = o.y2 - y / z;
                m++;
            }
        }
    }
    void calc_jacobian(Mat& J) {
        const double step = 1e-4;
        // columns: 0 focal, 1 ppx, 2 ppy, 3 aspect, 4..6 rotation; refinement mask positions (0,0) (0,2) (1,2) (1,1).  A column is a pair
        // of error evaluations at its own perturbed copy of the cameras: one column per host thread at a time
        const bool on[7] = {(bool)refine[0], (bool)refine[2], (bool)refine[4], (bool)refine[3], true, true, true};
        // (only the rows of the pairs the column's camera is part of are evaluated and written: elsewhere the two evaluations are
        // identical, their difference is +0, and J was zeroed by the solver before this call)
        ba_parallel_for(7 * n, [&](int col) {
            if (!on[col % 7]) return;
            const std::vector<int>& mine = cam_edges[col / 7];
            std::vector<double> c = cam, e1((size_t)total * 2), e2((size_t)total * 2);
            const double val = c[col];
            c[col] = val - step; calc_error_at(c, e1, &mine);
            c[col] = val + step; calc_error_at(c, e2, &mine);
            for (int e : mine)
                for (int k = 2 * edge_m[e]; k < 2 * (edge_m[e] + (int)obs[e].size()); k++) J(k, col) = (e2[k] - e1[k]) / (2 * step);
        });
    }
};

// stitching/src/motion_estimators.cpp findMaxSpanningTree: Kruskal on num_inliers (descending), then the tree's centres
void max_spanning_tree_centers(int n, const MisMatchesInfo* pm, std::vector<int>& centers) {
    struct GE { int from, to; float w; };
    std::vector<GE> all;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if (pm[i * n + j].has_H == 0) continue;
            const float w = (float)pm[i * n + j].num_inliers;
            all.push_back({i, j, w});
        }
    std::stable_sort(all.begin(), all.end(), [](const GE& a, const GE& b) { return a.w > b.w; });
    std::vector<int> parent(n), rnk(n, 0);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int e) { int s = e; while (s != parent[s]) s = parent[s]; while (e != parent[e]) { int nx = parent[e]; parent[e] = s; e = nx; } return s; };
    std::vector<std::vector<int>> adj(n);
    std::vector<int> power(n, 0);
    for (const GE& e : all) {
        int c1 = find(e.from), c2 = find(e.to);
        if (c1 == c2) continue;
        if (rnk[c1] < rnk[c2]) parent[c1] = c2; else if (rnk[c2] < rnk[c1]) parent[c2] = c1; else { parent[c1] = c2; rnk[c2]++; }
        adj[e.from].push_back(e.to); adj[e.to].push_back(e.from);
        power[e.from]++; power[e.to]++;
    }
    std::vector<int> leafs;
    for (int i = 0; i < n; i++) if (power[i] == 1) leafs.push_back(i);
    // distance from every leaf (BFS); the centres minimise the maximum distance to a leaf
    std::vector<int> max_dists(n, 0);
    for (int leaf : leafs) {
        std::vector<int> dist(n, -1), q{leaf};
        dist[leaf] = 0;
        for (size_t h = 0; h < q.size(); h++) for (int v : adj[q[h]]) if (dist[v] < 0) { dist[v] = dist[q[h]] + 1; q.push_back(v); }
        for (int i = 0; i < n; i++) if (dist[i] >= 0) max_dists[i] = std::max(max_dists[i], dist[i]);
    }
    int mn = max_dists.empty() ? 0 : *std::min_element(max_dists.begin(), max_dists.end());
    centers.clear();
    for (int i = 0; i < n; i++) if (max_dists[i] == mn) centers.push_back(i);
}

inline float hypot_cv(float a, float b) {   // cv::hypot(float, float)
    a = std::fabs(a); b = std::fabs(b);
    if (a > b) { b /= a; return a * std::sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * std::sqrt(1 + a * a); }
    return 0;
}

// cv::eigen of a symmetric n x n CV_32F matrix (core/src/lapack.cpp JacobiImpl_<float>): cyclic pivoting on the largest
// off-diagonal element of the upper triangle (row / column maxima kept in indR / indC), eigenvalues descending, eigenvectors as rows
void jacobi_eigen_f32(float* A, int n, float* W, float* V) {
    const float eps = FLT_EPSILON;
    std::vector<int> indR(n), indC(n);
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) V[i * n + j] = 0; V[i * n + i] = 1; }
    auto row_max = [&](int k) { int m = k + 1; float mv = std::fabs(A[n * k + m]); for (int i = k + 2; i < n; i++) { const float val = std::fabs(A[n * k + i]); if (mv < val) mv = val, m = i; } return m; };
    auto col_max = [&](int k) { int m = 0; float mv = std::fabs(A[k]); for (int i = 1; i < k; i++) { const float val = std::fabs(A[n * i + k]); if (mv < val) mv = val, m = i; } return m; };
    for (int k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) indR[k] = row_max(k);
        if (k > 0) indC[k] = col_max(k);
    }
    if (n > 1)
        for (int iters = 0; iters < n * n * 30; iters++) {
            int k = 0;
            float mv = std::fabs(A[indR[0]]);
            for (int i = 1; i < n - 1; i++) { const float val = std::fabs(A[n * i + indR[i]]); if (mv < val) mv = val, k = i; }
            int l = indR[k];
            for (int i = 1; i < n; i++) { const float val = std::fabs(A[n * indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
            const float p = A[n * k + l];
            if (std::fabs(p) <= eps) break;
            const float y = (float)((W[l] - W[k]) * 0.5);
            float t = std::fabs(y) + hypot_cv(p, y);
            float sn = hypot_cv(p, t);
            const float c = t / sn;
            sn = p / sn; t = (p / t) * p;
            if (y < 0) sn = -sn, t = -t;
            A[n * k + l] = 0;
            W[k] -= t; W[l] += t;
            auto rot = [&](float& v0, float& v1) { const float a0 = v0, b0 = v1; v0 = a0 * c - b0 * sn; v1 = a0 * sn + b0 * c; };
            for (int i = 0; i < k; i++) rot(A[n * i + k], A[n * i + l]);
            for (int i = k + 1; i < l; i++) rot(A[n * k + i], A[n * i + l]);
            for (int i = l + 1; i < n; i++) rot(A[n * k + i], A[n * l + i]);
            for (int i = 0; i < n; i++) rot(V[n * k + i], V[n * l + i]);
            for (int j = 0; j < 2; j++) {
                const int idx = j == 0 ? k : l;
                if (idx < n - 1) indR[idx] = row_max(idx);
                if (idx > 0) indC[idx] = col_max(idx);
            }
        }
    for (int k = 0; k < n - 1; k++) {
        int m = k;
        for (int i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) { std::swap(W[m], W[k]); for (int i = 0; i < n; i++) std::swap(V[n * m + i], V[n * k + i]); }
    }
}

}  // namespace

extern "C" int mis_bundle_adjust_reproj(MisContext* ctx, const MisFeatures* features, const MisMatchesInfo* pairwise, int n, float conf_thresh,
                                        const char* refine_mask, MisCameraParams* cameras) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, features && pairwise && cameras && n >= 2, MIS_E_INVALID, "null argument or fewer than two cameras");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    Adjuster ad;
    ad.n = n;
    if (refine_mask && strlen(refine_mask) >= 5)
        for (int k = 0; k < 5; k++) ad.refine[k] = refine_mask[k] == 'x';
    // keypoints of every image to the host (a few hundred KB)
    std::vector<std::vector<MisKeyPoint>> kps(n);
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; i++) {
        kps[i].resize((size_t)std::max(features[i].n, 0));
        if (features[i].n > 0) MIS_HIP(ctx, hipMemcpy(kps[i].data(), features[i].keypoints, sizeof(MisKeyPoint) * (size_t)features[i].n, hipMemcpyDeviceToHost));
    }
    // setUpInitialCameraParams: R projected on SO(3) through its SVD, then Rodrigues
    ad.cam.assign((size_t)n * 7, 0.);
    for (int i = 0; i < n; i++) {
        ad.cam[i * 7] = cameras[i].focal; ad.cam[i * 7 + 1] = cameras[i].ppx; ad.cam[i * 7 + 2] = cameras[i].ppy; ad.cam[i * 7 + 3] = cameras[i].aspect;
        // the reference's cameras carry CV_32F rotations (image_stitching.cpp:626-634): SVD, u * vt and the sign fix in float,
        // Rodrigues in double on the float matrix, the rotation vector stored as CV_32F
        float Rf[9], u[9], w[3], vt[9], R[9];
        for (int k = 0; k < 9; k++) Rf[k] = (float)cameras[i].R[k];
        svd3(Rf, u, w, vt);
        mul3(u, vt, R);
        if (det3(R) < 0) for (float& v : R) v *= -1;
        double Rd[9], rv[3];
        for (int k = 0; k < 9; k++) Rd[k] = R[k];
        rodrigues_to_vec(Rd, rv);
        for (int k = 0; k < 3; k++) ad.cam[i * 7 + 4 + k] = (double)(float)rv[k];
    }
    // leave only consistent image pairs
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            const MisMatchesInfo& mi = pairwise[i * n + j];
            if (!(mi.confidence > conf_thresh)) continue;
            std::vector<Obs> ob;
            for (int k = 0; k < mi.n_matches; k++) {
                if (!mi.inliers_mask || !mi.inliers_mask[k]) continue;
                const MisDMatch& m = mi.matches[k];
                MIS_CHECK(ctx, m.query_idx >= 0 && m.query_idx < (int)kps[i].size() && m.train_idx >= 0 && m.train_idx < (int)kps[j].size(), MIS_E_INVALID,
                          "match indices of pair (%d, %d) exceed the feature counts", i, j);
                ob.push_back({kps[i][m.query_idx].x, kps[i][m.query_idx].y, kps[j][m.train_idx].x, kps[j][m.train_idx].y});
            }
            ad.edges.push_back({i, j});
            ad.total += (int)ob.size();
            ad.obs.push_back(std::move(ob));
        }
    MIS_CHECK(ctx, ad.total > 0, MIS_E_INVALID, "bundle adjustment: no image pair above the confidence threshold");
    ad.index_edges();
    LevMarq solver(n * 7, ad.total * 2, 1000, DBL_EPSILON);   // TermCriteria(EPS + COUNT, 1000, DBL_EPSILON)
    solver.param = ad.cam;
    {
        // the rows of an edge (two per observation, in edge order) depend on the parameters of its two cameras only
        std::vector<std::vector<std::pair<int, int>>> cam_rows(n);
        int row = 0;
        for (size_t e = 0; e < ad.edges.size(); e++) {
            const int r1 = row + 2 * (int)ad.obs[e].size();
            for (int c : {ad.edges[e].i, ad.edges[e].j}) {
                auto& v = cam_rows[c];
                if (!v.empty() && v.back().second == row) v.back().second = r1; else v.emplace_back(row, r1);
            }
            row = r1;
        }
        solver.col_rows.resize((size_t)n * 7);
        for (int c = 0; c < n; c++) for (int j = 0; j < 7; j++) solver.col_rows[c * 7 + j] = cam_rows[c];
    }
    const bool trace = getenv("MIS_BA_TRACE") != nullptr;
    for (;;) {
        bool want_J, want_err;
        const bool proceed = solver.update(want_J, want_err);
        if (trace) fprintf(stderr, "[ba] state %d iters %d lambdaLg10 %d prevErr %.17g err %.17g p0 %.17g p4 %.17g\n", solver.state, solver.iters, solver.lambdaLg10, solver.prevErrNorm, solver.errNorm, solver.param[0], solver.param[4]);
        ad.cam = solver.param;
        if (!proceed || !want_err) break;
        if (want_J) ad.calc_jacobian(solver.J);
        if (want_err) ad.calc_error(solver.err);
    }
    for (double v : ad.cam) MIS_CHECK(ctx, std::isfinite(v), MIS_E_INVALID, "bundle adjustment diverged (non-finite camera parameter)");
    // obtainRefinedCameraParams
    for (int i = 0; i < n; i++) {
        cameras[i].focal = ad.cam[i * 7]; cameras[i].ppx = ad.cam[i * 7 + 1]; cameras[i].ppy = ad.cam[i * 7 + 2]; cameras[i].aspect = ad.cam[i * 7 + 3];
        double R[9];
        rodrigues_to_mat(&ad.cam[i * 7 + 4], R);
        for (int k = 0; k < 9; k++) cameras[i].R[k] = (double)(float)R[k];      // R.convertTo(tmp, CV_32F)
    }
    // normalise the motion to the centre image of the maximum spanning tree: R_i = R_c^-1 * R_i on the CV_32F matrices
    std::vector<int> centers;
    max_spanning_tree_centers(n, pairwise, centers);
    if (!centers.empty()) {
        float Rc[9], Rinv[9];
        for (int k = 0; k < 9; k++) Rc[k] = (float)cameras[centers[0]].R[k];
        if (inv3(Rc, Rinv))
            for (int i = 0; i < n; i++) {
                float Ri[9], Ro[9];
                for (int k = 0; k < 9; k++) Ri[k] = (float)cameras[i].R[k];
                mul3(Rinv, Ri, Ro);
                for (int k = 0; k < 9; k++) cameras[i].R[k] = Ro[k];
            }
    }
    return MIS_OK;
}

// detail::waveCorrect (motion_estimators.cpp): kind 0 = WAVE_CORRECT_HORIZ, 1 = WAVE_CORRECT_VERT; rmats: n x 9 doubles holding the
// CV_32F rotations the reference passes (image_stitching.cpp:720-722); everything in float as there, results returned in place
extern "C" int mis_wave_correct(double* rmats, int n, int kind) {
    if (!rmats || n < 1 || (kind != 0 && kind != 1)) return MIS_E_INVALID;
    if (n <= 1) return MIS_OK;
    std::vector<float> R((size_t)n * 9);
    for (int i = 0; i < 9 * n; i++) R[i] = (float)rmats[i];
    float moment[9] = {0};
    for (int i = 0; i < n; i++) {
        const float col[3] = {R[9 * i], R[9 * i + 3], R[9 * i + 6]};   // col(0): the camera's x axis
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) moment[a * 3 + b] += col[a] * col[b];
    }
    float evals[3], evecs[9];
    jacobi_eigen_f32(moment, 3, evals, evecs);
    float rg1[3], img_k[3] = {0, 0, 0};
    memcpy(rg1, kind == 0 ? evecs + 6 : evecs, sizeof(rg1));   // HORIZ: the eigenvector of the smallest eigenvalue; VERT: of the largest
    for (int i = 0; i < n; i++) { img_k[0] += R[9 * i + 2]; img_k[1] += R[9 * i + 5]; img_k[2] += R[9 * i + 8]; }   // sum of the z axes
    float rg0[3] = {rg1[1] * img_k[2] - rg1[2] * img_k[1], rg1[2] * img_k[0] - rg1[0] * img_k[2], rg1[0] * img_k[1] - rg1[1] * img_k[0]};
    const double rg0n = std::sqrt((double)rg0[0] * rg0[0] + (double)rg0[1] * rg0[1] + (double)rg0[2] * rg0[2]);
    if (rg0n <= DBL_MIN) return MIS_OK;
    for (float& v : rg0) v = (float)(v * (1. / rg0n));    // rg0 /= norm: scaled by the reciprocal, in double
    const float rg2[3] = {rg0[1] * rg1[2] - rg0[2] * rg1[1], rg0[2] * rg1[0] - rg0[0] * rg1[2], rg0[0] * rg1[1] - rg0[1] * rg1[0]};
    double conf = 0;
    for (int i = 0; i < n; i++) {
        const float* g = kind == 0 ? rg0 : rg1;     // Mat::dot on CV_32F: products and sum in double
        const double d = (double)g[0] * R[9 * i] + (double)g[1] * R[9 * i + 3] + (double)g[2] * R[9 * i + 6];
        conf += kind == 0 ? d : -d;
    }
    if (conf < 0) { for (float& v : rg0) v *= -1; for (float& v : rg1) v *= -1; }
    const float Rg[9] = {rg0[0], rg0[1], rg0[2], rg1[0], rg1[1], rg1[2], rg2[0], rg2[1], rg2[2]};
    for (int i = 0; i < n; i++) {
        float o[9];
        mul3(Rg, &R[9 * i], o);
        for (int k = 0; k < 9; k++) rmats[9 * i + k] = o[k];
    }
    return MIS_OK;
}
