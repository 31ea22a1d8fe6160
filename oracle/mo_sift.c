/* mo_sift.c -- see mo_sift.h.  TEST INFRASTRUCTURE ONLY. */
#include "mo_sift.h"
#include "mo_common.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_INIT_SIGMA 0.5f
#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_ORI_SIG_FCTR 1.5f
#define SIFT_ORI_RADIUS (3 * SIFT_ORI_SIG_FCTR)
#define SIFT_ORI_PEAK_RATIO 0.8f
#define SIFT_DESCR_SCL_FCTR 3.f
#define SIFT_DESCR_MAG_THR 0.2f
#define SIFT_INT_DESCR_FCTR 512.f
#define MO_SIFT_MAX_OCT 16

/* Cephes expf: the shared exponential of both sides (sift.simd.hpp uses cv::hal::exp32f) */
float mo_expf(float xx) {
    float x = xx, z;
    int n;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.f;
    z = floorf(1.44269504088896341f * x + 0.5f);
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    n = (int)z;
    z = x * x;
    z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x + 4.1665795894E-2f) * x + 1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
    /* ldexpf(z, n) by exact power-of-two multiplications (two steps below the normal range) */
    if (n < -126) {
        union { uint32_t u; float f; } a, b;
        a.u = (uint32_t)(1) << 23; /* 2^-126 */
        b.u = (uint32_t)(n + 126 + 127) << 23;
        return (z * b.f) * a.f;
    } else {
        union { uint32_t u; float f; } a;
        a.u = (uint32_t)(n + 127) << 23;
        return z * a.f;
    }
}
static float pow2f(float x) { return mo_expf(x * 0.69314718055994530942f); }

int mo_gaussian_taps_f32(double sigma, float* taps) {
    /* GaussianBlur: ksize from sigma for float images; getGaussianKernel in double, normalised, stored as float */
    int n = mo_round_d(sigma * 8 + 1) | 1;
    double scale2x = -0.5 / (sigma * sigma), sum = 0, t[64];
    if (n > 63) n = 63;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        t[i] = exp(scale2x * x * x);
        sum += t[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) taps[i] = (float)(t[i] * sum);
    return n;
}

struct MoSift {
    MoSiftParams p;
    int w, h, noct, nl;
    int ow[MO_SIFT_MAX_OCT], oh[MO_SIFT_MAX_OCT];
    float** gauss; /* noct * (nl + 3) */
    float** dog;   /* noct * (nl + 2) */
    uint8_t* gray;
    float* grayf;
    float* tmp;
    MoKeyPoint* kps;
    int nk, capk, nraw;
    float* desc;
};

void mo_sift_default_params(MoSiftParams* p) {
    p->nfeatures = 0; p->n_octave_layers = 3; p->contrast_threshold = 0.04; p->edge_threshold = 10; p->sigma = 1.6;
}

MoSift* mo_sift_create(const MoSiftParams* pp, int width, int height) {
    MoSift* s = (MoSift*)calloc(1, sizeof(MoSift));
    mo_sift_default_params(&s->p);
    if (pp) s->p = *pp;
    s->w = width; s->h = height; s->nl = s->p.n_octave_layers;
    int bw = width * 2, bh = height * 2, firstOctave = -1;
    s->noct = mo_round_d(log((double)(bw < bh ? bw : bh)) / log(2.) - 2) - firstOctave;
    if (s->noct > MO_SIFT_MAX_OCT) s->noct = MO_SIFT_MAX_OCT;
    if (s->noct < 1) s->noct = 1;
    s->gauss = (float**)calloc((size_t)s->noct * (s->nl + 3), sizeof(float*));
    s->dog = (float**)calloc((size_t)s->noct * (s->nl + 2), sizeof(float*));
    int cw = bw, ch = bh;
    for (int o = 0; o < s->noct; o++) {
        s->ow[o] = cw; s->oh[o] = ch;
        for (int i = 0; i < s->nl + 3; i++) s->gauss[o * (s->nl + 3) + i] = (float*)malloc(sizeof(float) * (size_t)cw * ch);
        for (int i = 0; i < s->nl + 2; i++) s->dog[o * (s->nl + 2) + i] = (float*)malloc(sizeof(float) * (size_t)cw * ch);
        cw /= 2; ch /= 2;
        if (cw < 1 || ch < 1) { s->noct = o + 1; break; }
    }
    s->gray = (uint8_t*)malloc((size_t)width * height);
    s->grayf = (float*)malloc(sizeof(float) * (size_t)bw * bh);
    s->tmp = (float*)malloc(sizeof(float) * (size_t)bw * bh);
    return s;
}

void mo_sift_destroy(MoSift* s) {
    if (!s) return;
    for (int i = 0; i < s->noct * (s->nl + 3); i++) free(s->gauss[i]);
    for (int i = 0; i < s->noct * (s->nl + 2); i++) free(s->dog[i]);
    free(s->gauss); free(s->dog); free(s->gray); free(s->grayf); free(s->tmp); free(s->kps); free(s->desc); free(s);
}

/* resize(float, 2x, INTER_LINEAR): source coordinate (d + 0.5) * 0.5 - 0.5, taps clamped, horizontal then vertical */
static void upsample2x(const uint8_t* g, int w, int h, float* dst) {
    int dw = 2 * w, dh = 2 * h, y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < dh; y++) {
        float fy = (float)((y + 0.5) * 0.5 - 0.5);
        int sy = mo_floor_d(fy);
        fy -= sy;
        if (sy < 0) { sy = 0; fy = 0; }
        if (sy >= h - 1) { sy = h - 1; fy = 0; }
        int sy1 = sy + 1 < h ? sy + 1 : sy;
        const uint8_t *r0 = g + (size_t)sy * w, *r1 = g + (size_t)sy1 * w;
        float b0 = 1.f - fy, b1 = fy;
        for (int x = 0; x < dw; x++) {
            float fx = (float)((x + 0.5) * 0.5 - 0.5);
            int sx = mo_floor_d(fx);
            fx -= sx;
            if (sx < 0) { sx = 0; fx = 0; }
            if (sx >= w - 1) { sx = w - 1; fx = 0; }
            int sx1 = sx + 1 < w ? sx + 1 : sx;
            float a0 = 1.f - fx, a1 = fx;
            float h0 = (float)r0[sx] * a0 + (float)r0[sx1] * a1;
            float h1 = (float)r1[sx] * a0 + (float)r1[sx1] * a1;
            dst[(size_t)y * dw + x] = h0 * b0 + h1 * b1;
        }
    }
}

/* GaussianBlur(src, dst, Size(), sigma, sigma), BORDER_REFLECT_101: rows then columns, taps in ascending order */
static void gaussian_blur(const float* src, int w, int h, double sigma, float* tmp, float* dst) {
    float k[64];
    int n = mo_gaussian_taps_f32(sigma, k), r = n / 2, y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0;
            for (int t = 0; t < n; t++) acc += k[t] * src[(size_t)y * w + mo_reflect101(x + t - r, w)];
            tmp[(size_t)y * w + x] = acc;
        }
#pragma omp parallel for schedule(static)
    for (y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0;
            for (int t = 0; t < n; t++) acc += k[t] * tmp[(size_t)mo_reflect101(y + t - r, h) * w + x];
            dst[(size_t)y * w + x] = acc;
        }
}

static int solve3(const float* a /* 3x3 */, const float* b, float* x) {
    /* Matx33f::solve(DECOMP_LU): closed form through the determinant (Matx_FastSolveOp<_Tp, 3, 3, 1>) */
    float d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (d == 0) return 0;
    d = 1 / d;
    x[0] = d * (b[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (b[1] * a[8] - a[5] * b[2]) + a[2] * (b[1] * a[7] - a[4] * b[2]));
    x[1] = d * (a[0] * (b[1] * a[8] - a[5] * b[2]) - b[0] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * b[2] - b[1] * a[6]));
    x[2] = d * (a[0] * (a[4] * b[2] - b[1] * a[7]) - a[1] * (a[3] * b[2] - b[1] * a[6]) + b[0] * (a[3] * a[7] - a[4] * a[6]));
    return 1;
}

#define AT(img, r, c) ((img)[(size_t)(r) * w + (c)])

static int adjust_local_extrema(const MoSift* s, int octv, int* layer_io, int* r_io, int* c_io, MoKeyPoint* kpt) {
    const int nl = s->nl, w = s->ow[octv], h = s->oh[octv];
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int i = 0, layer = *layer_io, r = *r_io, c = *c_io;
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const float* img = s->dog[octv * (nl + 2) + layer];
        const float* prev = s->dog[octv * (nl + 2) + layer - 1];
        const float* next = s->dog[octv * (nl + 2) + layer + 1];
        float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                       (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        float v2 = AT(img, r, c) * 2;
        float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        float dss = (AT(next, r, c) + AT(prev, r, c) - v2) * second_deriv_scale;
        float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        float dxs = (AT(next, r, c + 1) - AT(next, r, c - 1) - AT(prev, r, c + 1) + AT(prev, r, c - 1)) * cross_deriv_scale;
        float dys = (AT(next, r + 1, c) - AT(next, r - 1, c) - AT(prev, r + 1, c) + AT(prev, r - 1, c)) * cross_deriv_scale;
        float H[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss}, X[3] = {0, 0, 0};
        solve3(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT_MAX / 3) || fabsf(xr) > (float)(INT_MAX / 3) || fabsf(xc) > (float)(INT_MAX / 3)) return 0;
        c += mo_round_f(xc); r += mo_round_f(xr); layer += mo_round_f(xi);
        if (layer < 1 || layer > nl || c < SIFT_IMG_BORDER || c >= w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER) return 0;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return 0;
    {
        const float* img = s->dog[octv * (nl + 2) + layer];
        const float* prev = s->dog[octv * (nl + 2) + layer - 1];
        const float* next = s->dog[octv * (nl + 2) + layer + 1];
        float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                       (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        float t = (dD[0] * xc + dD[1] * xr) + dD[2] * xi;
        contr = AT(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nl < (float)s->p.contrast_threshold) return 0;
        float v2 = AT(img, r, c) * 2.f;
        float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        float et = (float)s->p.edge_threshold;
        if (det <= 0 || tr * tr * et >= (et + 1) * (et + 1) * det) return 0;
    }
    kpt->x = (c + xc) * (1 << octv);
    kpt->y = (r + xr) * (1 << octv);
    kpt->octave = octv + (layer << 8) + (mo_round_f((xi + 0.5f) * 255) << 16);
    kpt->size = (float)s->p.sigma * pow2f((layer + xi) / nl) * (1 << octv) * 2;
    kpt->response = fabsf(contr);
    *layer_io = layer; *r_io = r; *c_io = c;
    return 1;
}

static float calc_orientation_hist(const float* img, int w, int h, int px, int py, int radius, float sigma, float* hist, int n) {
    float temphist[SIFT_ORI_HIST_BINS + 4];
    float* th = temphist + 2;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    for (int i = 0; i < n + 4; i++) temphist[i] = 0.f;
    for (int i = -radius; i <= radius; i++) {
        int y = py + i;
        if (y <= 0 || y >= h - 1) continue;
        for (int j = -radius; j <= radius; j++) {
            int x = px + j;
            if (x <= 0 || x >= w - 1) continue;
            float dx = AT(img, y, x + 1) - AT(img, y, x - 1), dy = AT(img, y - 1, x) - AT(img, y + 1, x);
            float wgt = mo_expf((i * i + j * j) * expf_scale);
            float ori = mo_fast_atan2(dy, dx), mag = sqrtf(dx * dx + dy * dy);
            int bin = mo_round_f((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            th[bin] += wgt * mag;
        }
    }
    th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
    for (int i = 0; i < n; i++)
        hist[i] = (th[i - 2] + th[i + 2]) * (1.f / 16.f) + (th[i - 1] + th[i + 1]) * (4.f / 16.f) + th[i] * (6.f / 16.f);
    float maxval = hist[0];
    for (int i = 1; i < n; i++) maxval = maxval > hist[i] ? maxval : hist[i];
    return maxval;
}

static void calc_descriptor(const float* img, int w, int h, float ptx, float pty, float ori, float scl, float* dst) {
    const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    int px = mo_round_f(ptx), py = mo_round_f(pty);
    float cos_t = mo_cosf(ori * (float)(3.14159265358979323846 / 180)), sin_t = mo_sinf(ori * (float)(3.14159265358979323846 / 180));
    float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = mo_round_f(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    int rmax = (int)sqrt((double)w * w + (double)h * h);
    if (radius > rmax) radius = rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    float hist[(SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_HIST_BINS + 2)];
    const int histlen = (d + 2) * (d + 2) * (n + 2);
    for (int i = 0; i < histlen; i++) hist[i] = 0.f;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1)) continue;
            float dx = AT(img, r, c + 1) - AT(img, r, c - 1), dy = AT(img, r - 1, c) - AT(img, r + 1, c);
            float wgt = mo_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (mo_fast_atan2(dy, dx) - ori) * bins_per_rad, mag = sqrtf(dx * dx + dy * dy) * wgt;
            int r0 = mo_floor_f(rbin), c0 = mo_floor_f(cbin), o0 = mo_floor_f(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx] += v_rco000; hist[idx + 1] += v_rco001;
            hist[idx + (n + 2)] += v_rco010; hist[idx + (n + 3)] += v_rco011;
            hist[idx + (d + 2) * (n + 2)] += v_rco100; hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
            hist[idx + (d + 3) * (n + 2)] += v_rco110; hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
        }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; k++) dst[(i * d + j) * n + k] = hist[idx + k];
        }
    const int len = d * d * n;
    float nrm2 = 0;
    for (int k = 0; k < len; k++) nrm2 += dst[k] * dst[k];
    float thr = sqrtf(nrm2) * SIFT_DESCR_MAG_THR;
    nrm2 = 0;
    for (int k = 0; k < len; k++) {
        float val = dst[k] < thr ? dst[k] : thr;
        dst[k] = val;
        nrm2 += val * val;
    }
    float root = sqrtf(nrm2);
    nrm2 = SIFT_INT_DESCR_FCTR / (root > FLT_EPSILON ? root : FLT_EPSILON);
    for (int k = 0; k < len; k++) {
        int v = mo_round_f(dst[k] * nrm2);   /* saturate_cast<uchar> */
        dst[k] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static int kp_less(const void* pa, const void* pb) {
    /* KeyPoint_LessThan of KeyPointsFilter::removeDuplicatedSorted */
    const MoKeyPoint *a = (const MoKeyPoint*)pa, *b = (const MoKeyPoint*)pb;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    if (a->angle != b->angle) return a->angle < b->angle ? -1 : 1;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    if (a->octave != b->octave) return a->octave > b->octave ? -1 : 1;
    return 0;
}

static void push_kp(MoSift* s, const MoKeyPoint* k) {
    if (s->nk == s->capk) {
        s->capk = s->capk ? s->capk * 2 : 4096;
        s->kps = (MoKeyPoint*)realloc(s->kps, sizeof(MoKeyPoint) * (size_t)s->capk);
    }
    s->kps[s->nk++] = *k;
}

int mo_sift_run(MoSift* s, const uint8_t* bgr, size_t stride) {
    const int nl = s->nl, firstOctave = -1;
    const double sigma = s->p.sigma;
    /* createInitialImage: gray -> float, doubled with INTER_LINEAR, blurred to sigma */
    mo_bgr2gray(bgr, s->w, s->h, stride, s->gray, (size_t)s->w);
    upsample2x(s->gray, s->w, s->h, s->grayf);
    {
        float sd = sqrtf(fmaxf((float)(sigma * sigma) - SIFT_INIT_SIGMA * SIFT_INIT_SIGMA * 4, 0.01f));
        gaussian_blur(s->grayf, s->ow[0], s->oh[0], (double)sd, s->tmp, s->gauss[0]);
    }
    /* buildGaussianPyramid */
    double sig[16];
    sig[0] = sigma;
    {
        double k = pow(2., 1. / nl);
        for (int i = 1; i < nl + 3; i++) {
            double sig_prev = pow(k, (double)(i - 1)) * sigma, sig_total = sig_prev * k;
            sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
        }
    }
    for (int o = 0; o < s->noct; o++) {
        const int w = s->ow[o], h = s->oh[o];
        for (int i = 0; i < nl + 3; i++) {
            float* dst = s->gauss[o * (nl + 3) + i];
            if (o == 0 && i == 0) continue;
            if (i == 0) {
                /* resize(src, Size(src.cols / 2, src.rows / 2), INTER_NEAREST) of layer nOctaveLayers of the previous octave */
                const float* src = s->gauss[(o - 1) * (nl + 3) + nl];
                const int sw = s->ow[o - 1], sh = s->oh[o - 1];
                for (int y = 0; y < h; y++) {
                    int sy = y * 2 < sh ? y * 2 : sh - 1;
                    for (int x = 0; x < w; x++) dst[(size_t)y * w + x] = src[(size_t)sy * sw + (x * 2 < sw ? x * 2 : sw - 1)];
                }
            } else {
                gaussian_blur(s->gauss[o * (nl + 3) + i - 1], w, h, sig[i], s->tmp, dst);
            }
        }
        /* buildDoGPyramid */
        for (int i = 0; i < nl + 2; i++) {
            const float *a = s->gauss[o * (nl + 3) + i], *b = s->gauss[o * (nl + 3) + i + 1];
            float* d = s->dog[o * (nl + 2) + i];
            for (size_t q = 0; q < (size_t)w * h; q++) d[q] = b[q] - a[q];
        }
    }
    /* findScaleSpaceExtrema */
    s->nk = 0;
    const int threshold = mo_floor_d(0.5 * s->p.contrast_threshold / nl * 255);
    const int n = SIFT_ORI_HIST_BINS;
    for (int o = 0; o < s->noct; o++) {
        const int w = s->ow[o], h = s->oh[o];
        for (int i = 1; i <= nl; i++) {
            const float* img = s->dog[o * (nl + 2) + i];
            const float* prev = s->dog[o * (nl + 2) + i - 1];
            const float* next = s->dog[o * (nl + 2) + i + 1];
            for (int r = SIFT_IMG_BORDER; r < h - SIFT_IMG_BORDER; r++)
                for (int c = SIFT_IMG_BORDER; c < w - SIFT_IMG_BORDER; c++) {
                    float val = AT(img, r, c);
                    if (!(fabsf(val) > (float)threshold)) continue;
                    int is_ext = 1;
                    if (val > 0) {
                        for (int dr = -1; dr <= 1 && is_ext; dr++)
                            for (int dc = -1; dc <= 1; dc++)
                                if (!(val >= AT(img, r + dr, c + dc) && val >= AT(prev, r + dr, c + dc) && val >= AT(next, r + dr, c + dc))) { is_ext = 0; break; }
                    } else {
                        for (int dr = -1; dr <= 1 && is_ext; dr++)
                            for (int dc = -1; dc <= 1; dc++)
                                if (!(val <= AT(img, r + dr, c + dc) && val <= AT(prev, r + dr, c + dc) && val <= AT(next, r + dr, c + dc))) { is_ext = 0; break; }
                    }
                    if (!is_ext) continue;
                    MoKeyPoint kpt;
                    int r1 = r, c1 = c, layer = i;
                    memset(&kpt, 0, sizeof(kpt));
                    if (!adjust_local_extrema(s, o, &layer, &r1, &c1, &kpt)) continue;
                    float scl_octv = kpt.size * 0.5f / (1 << o);
                    float hist[SIFT_ORI_HIST_BINS];
                    float omax = calc_orientation_hist(s->gauss[o * (nl + 3) + layer], w, h, c1, r1, mo_round_f(SIFT_ORI_RADIUS * scl_octv),
                                                       SIFT_ORI_SIG_FCTR * scl_octv, hist, n);
                    float mag_thr = omax * SIFT_ORI_PEAK_RATIO;
                    for (int j = 0; j < n; j++) {
                        int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? n + bin : (bin >= n ? bin - n : bin);
                            kpt.angle = 360.f - (float)((360.f / n) * bin);
                            if (fabsf(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
                            push_kp(s, &kpt);
                        }
                    }
                }
        }
    }
    s->nraw = s->nk;
    /* KeyPointsFilter::removeDuplicatedSorted */
    qsort(s->kps, (size_t)s->nk, sizeof(MoKeyPoint), kp_less);
    {
        int m = 0;
        for (int i = 0; i < s->nk; i++) {
            if (m > 0) {
                const MoKeyPoint *a = &s->kps[m - 1], *b = &s->kps[i];
                if (a->x == b->x && a->y == b->y && a->size == b->size && a->angle == b->angle) continue;
            }
            s->kps[m++] = s->kps[i];
        }
        s->nk = m;
    }
    /* (nfeatures == 0: no retainBest) ; firstOctave < 0: back to input-image coordinates */
    for (int i = 0; i < s->nk; i++) {
        MoKeyPoint* k = &s->kps[i];
        float scale = 1.f / (float)(1 << -firstOctave);
        k->octave = (k->octave & ~255) | ((k->octave + firstOctave) & 255);
        k->x *= scale; k->y *= scale; k->size *= scale;
    }
    /* calcDescriptors */
    free(s->desc);
    s->desc = (float*)malloc(sizeof(float) * 128 * (size_t)(s->nk > 0 ? s->nk : 1));
    int q;
#pragma omp parallel for schedule(dynamic, 16)
    for (q = 0; q < s->nk; q++) {
        const MoKeyPoint* k = &s->kps[q];
        int octave = k->octave & 255, layer = (k->octave >> 8) & 255;
        octave = octave < 128 ? octave : (-128 | octave);
        float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
        float size = k->size * scale, ptx = k->x * scale, pty = k->y * scale;
        int oi = octave - firstOctave;
        float angle = 360.f - k->angle;
        if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        calc_descriptor(s->gauss[oi * (nl + 3) + layer], s->ow[oi], s->oh[oi], ptx, pty, angle, size * 0.5f, s->desc + 128 * (size_t)q);
    }
    return s->nk;
}

int mo_sift_num_keypoints(const MoSift* s) { return s->nk; }
const MoKeyPoint* mo_sift_keypoints(const MoSift* s) { return s->kps; }
const float* mo_sift_descriptors(const MoSift* s) { return s->desc; }
int mo_sift_num_octaves(const MoSift* s) { return s->noct; }
const float* mo_sift_gauss(const MoSift* s, int o, int i, int* w, int* h) { *w = s->ow[o]; *h = s->oh[o]; return s->gauss[o * (s->nl + 3) + i]; }
const float* mo_sift_dog(const MoSift* s, int o, int i, int* w, int* h) { *w = s->ow[o]; *h = s->oh[o]; return s->dog[o * (s->nl + 2) + i]; }
int mo_sift_num_raw_keypoints(const MoSift* s) { return s->nraw; }
