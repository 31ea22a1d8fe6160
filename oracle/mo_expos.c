/* mo_expos.c -- see mo_expos.h.  TEST INFRASTRUCTURE ONLY. */
#include "mo_expos.h"
#include "mo_common.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct MoCompensator {
    int bw, bh, nfilt, n;
    float** maps;
    int* mw;
    int* mh;
};

MoCompensator* mo_compensator_create(int block_w, int block_h, int nr_filtering) {
    MoCompensator* c = (MoCompensator*)calloc(1, sizeof(MoCompensator));
    c->bw = block_w; c->bh = block_h; c->nfilt = nr_filtering;
    return c;
}
static void free_maps(MoCompensator* c) {
    for (int i = 0; i < c->n; i++) free(c->maps[i]);
    free(c->maps); free(c->mw); free(c->mh);
    c->maps = NULL; c->mw = c->mh = NULL; c->n = 0;
}
void mo_compensator_destroy(MoCompensator* c) { if (c) { free_maps(c); free(c); } }

/* core/src/lapack.cpp LUImpl<double>: partial pivoting, eps = DBL_EPSILON * 100 */
int mo_solve_lu(double* A, double* b, int m) {
    const double eps = DBL_EPSILON * 100;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++) if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
        if (fabs(A[k * m + i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) { double t = A[i * m + j]; A[i * m + j] = A[k * m + j]; A[k * m + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / A[i * m + i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j * m + i] * d;
            for (k = i + 1; k < m; k++) A[j * m + k] += alpha * A[i * m + k];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < m; k++) s -= A[i * m + k] * b[k];
        b[i] = s / A[i * m + i];
    }
    return 1;
}

typedef struct { int x, y, w, h, img; int ox, oy; } Blk;   /* pano position, size, owning image, offset inside it */

int mo_compensator_feed(MoCompensator* c, int n, const int* cxy, const int* swh, const uint8_t* const* images, const uint8_t* const* masks) {
    free_maps(c);
    c->n = n;
    c->maps = (float**)calloc((size_t)n, sizeof(float*));
    c->mw = (int*)calloc((size_t)n, sizeof(int)); c->mh = (int*)calloc((size_t)n, sizeof(int));
    /* blocks of every image become the "images" of a GainCompensator */
    int nb = 0;
    for (int i = 0; i < n; i++) {
        c->mw[i] = (swh[2 * i] + c->bw - 1) / c->bw; c->mh[i] = (swh[2 * i + 1] + c->bh - 1) / c->bh;
        nb += c->mw[i] * c->mh[i];
    }
    Blk* B = (Blk*)malloc(sizeof(Blk) * (size_t)nb);
    int q = 0;
    for (int i = 0; i < n; i++) {
        const int W = swh[2 * i], H = swh[2 * i + 1];
        const int bw = (W + c->mw[i] - 1) / c->mw[i], bh = (H + c->mh[i] - 1) / c->mh[i];
        for (int by = 0; by < c->mh[i]; by++)
            for (int bx = 0; bx < c->mw[i]; bx++) {
                Blk* b = &B[q++];
                b->ox = bx * bw; b->oy = by * bh;
                int brx = b->ox + bw < W ? b->ox + bw : W, bry = b->oy + bh < H ? b->oy + bh : H;
                b->w = brx - b->ox; b->h = bry - b->oy; b->img = i;
                b->x = cxy[2 * i] + b->ox; b->y = cxy[2 * i + 1] + b->oy;
            }
    }
    /* GainCompensator::singleFeed over the blocks */
    int* N = (int*)calloc((size_t)nb * nb, sizeof(int));
    double* I = (double*)calloc((size_t)nb * nb, sizeof(double));
    uint8_t* skip = (uint8_t*)malloc((size_t)nb);
    memset(skip, 1, (size_t)nb);
    for (int i = 0; i < nb; i++)
        for (int j = i; j < nb; j++) {
            const Blk *a = &B[i], *b = &B[j];
            int x0 = a->x > b->x ? a->x : b->x, y0 = a->y > b->y ? a->y : b->y;
            int x1 = (a->x + a->w < b->x + b->w) ? a->x + a->w : b->x + b->w, y1 = (a->y + a->h < b->y + b->h) ? a->y + a->h : b->y + b->h;
            if (!(x0 < x1 && y0 < y1)) continue;
            const int Wa = swh[2 * a->img], Wb = swh[2 * b->img];
            int cnt = 0;
            double s1 = 0, s2 = 0;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    /* pixel of the pano at (x, y) inside image a / image b */
                    size_t pa = (size_t)(y - cxy[2 * a->img + 1]) * Wa + (x - cxy[2 * a->img]);
                    size_t pb = (size_t)(y - cxy[2 * b->img + 1]) * Wb + (x - cxy[2 * b->img]);
                    if (masks[a->img][pa] != 255 || masks[b->img][pb] != 255) continue;
                    const uint8_t *u = images[a->img] + 3 * pa, *v = images[b->img] + 3 * pb;
                    cnt++;
                    s1 += sqrt((double)(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
                    s2 += sqrt((double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
                }
            if (cnt < 1) cnt = 1;
            N[i * nb + j] = N[j * nb + i] = cnt;
            if (i != j) { skip[i] = 0; skip[j] = 0; }
            I[i * nb + j] = s1 / cnt; I[j * nb + i] = s2 / cnt;
        }
    double* gains = (double*)malloc(sizeof(double) * (size_t)nb);
    for (int i = 0; i < nb; i++) gains[i] = 1;
    int neq = 0;
    for (int i = 0; i < nb; i++) neq += !skip[i];
    if (neq > 0) {
        const double alpha = 0.01, beta = 100;
        double* A = (double*)calloc((size_t)neq * neq, sizeof(double));
        double* b = (double*)calloc((size_t)neq, sizeof(double));
        for (int i = 0, ki = 0; i < nb; i++) {
            if (skip[i]) continue;
            for (int j = 0, kj = 0; j < nb; j++) {
                if (skip[j]) continue;
                b[ki] += beta * N[i * nb + j];
                A[ki * neq + ki] += beta * N[i * nb + j];
                if (j != i) {
                    A[ki * neq + ki] += 2 * alpha * I[i * nb + j] * I[i * nb + j] * N[i * nb + j];
                    A[ki * neq + kj] -= 2 * alpha * I[i * nb + j] * I[j * nb + i] * N[i * nb + j];
                }
                kj++;
            }
            ki++;
        }
        if (mo_solve_lu(A, b, neq))
            for (int i = 0, j = 0; i < nb; i++) if (!skip[i]) gains[i] = b[j++];
        free(A); free(b);
    }
    /* gain maps: one float per block, smoothed nfilt times with the separable [0.25 0.5 0.25] (BORDER_REFLECT_101) */
    q = 0;
    for (int i = 0; i < n; i++) {
        const int mw = c->mw[i], mh = c->mh[i];
        float* m = (float*)malloc(sizeof(float) * (size_t)mw * mh);
        float* t = (float*)malloc(sizeof(float) * (size_t)mw * mh);
        for (int k = 0; k < mw * mh; k++) m[k] = (float)gains[q++];
        for (int it = 0; it < c->nfilt; it++) {
            for (int y = 0; y < mh; y++)
                for (int x = 0; x < mw; x++)
                    t[y * mw + x] = (m[y * mw + mo_reflect101(x - 1, mw)] + m[y * mw + mo_reflect101(x + 1, mw)]) * 0.25f + m[y * mw + x] * 0.5f;
            for (int y = 0; y < mh; y++)
                for (int x = 0; x < mw; x++)
                    m[y * mw + x] = (t[mo_reflect101(y - 1, mh) * mw + x] + t[mo_reflect101(y + 1, mh) * mw + x]) * 0.25f + t[y * mw + x] * 0.5f;
        }
        free(t);
        c->maps[i] = m;
    }
    free(B); free(N); free(I); free(skip); free(gains);
    return 0;
}

int mo_compensator_gain_map(const MoCompensator* c, int index, const float** map, int* bw, int* bh) {
    if (index < 0 || index >= c->n) return -1;
    *map = c->maps[index]; *bw = c->mw[index]; *bh = c->mh[index];
    return 0;
}

/* resize(gain_map, image.size(), INTER_LINEAR) (float) then multiply(image, gains, image): saturate_cast<uchar>(v * g) */
int mo_compensator_apply(const MoCompensator* c, int index, uint8_t* image, int w, int h) {
    if (index < 0 || index >= c->n) return -1;
    const float* m = c->maps[index];
    const int mw = c->mw[index], mh = c->mh[index];
    const double sx = 1.0 / ((double)w / (double)mw), sy = 1.0 / ((double)h / (double)mh);   /* resize(): scale = 1 / (dsize / ssize) */
    for (int y = 0; y < h; y++) {
        float fy = (float)((y + 0.5) * sy - 0.5);
        int iy = mo_floor_f(fy);
        fy -= iy;
        if (iy < 0) { iy = 0; fy = 0; }
        if (iy >= mh - 1) { iy = mh - 1; fy = 0; }
        const int iy1 = iy + 1 < mh ? iy + 1 : iy;
        for (int x = 0; x < w; x++) {
            float fx = (float)((x + 0.5) * sx - 0.5);
            int ix = mo_floor_f(fx);
            fx -= ix;
            if (ix < 0) { ix = 0; fx = 0; }
            if (ix >= mw - 1) { ix = mw - 1; fx = 0; }
            const int ix1 = ix + 1 < mw ? ix + 1 : ix;
            const float h0 = m[iy * mw + ix] * (1.f - fx) + m[iy * mw + ix1] * fx;
            const float h1 = m[iy1 * mw + ix] * (1.f - fx) + m[iy1 * mw + ix1] * fx;
            const float g = h0 * (1.f - fy) + h1 * fy;
            uint8_t* p = image + 3 * ((size_t)y * w + x);
            for (int k = 0; k < 3; k++) {
                int v = mo_round_f((float)p[k] * g);
                p[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
    return 0;
}

/* distanceTransform(src == 0 ? ..., DIST_L1, 3): L1 distance to the nearest zero pixel of `zero_is_feature` (two-pass chamfer
 * with weights 1 / 2 is exact for L1) */
static void dist_l1(const uint8_t* nonzero, int w, int h, float* d) {
    const float INF = 1e9f;
    for (int i = 0; i < w * h; i++) d[i] = nonzero[i] ? INF : 0.f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float v = d[y * w + x];
            if (x > 0 && d[y * w + x - 1] + 1 < v) v = d[y * w + x - 1] + 1;
            if (y > 0 && d[(y - 1) * w + x] + 1 < v) v = d[(y - 1) * w + x] + 1;
            d[y * w + x] = v;
        }
    for (int y = h - 1; y >= 0; y--)
        for (int x = w - 1; x >= 0; x--) {
            float v = d[y * w + x];
            if (x < w - 1 && d[y * w + x + 1] + 1 < v) v = d[y * w + x + 1] + 1;
            if (y < h - 1 && d[(y + 1) * w + x] + 1 < v) v = d[(y + 1) * w + x] + 1;
            d[y * w + x] = v;
        }
}

void mo_voronoi_seams(int n, const int* cxy, const int* swh, uint8_t* const* masks) {
    const int gap = 10;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            int x0 = cxy[2 * i] > cxy[2 * j] ? cxy[2 * i] : cxy[2 * j], y0 = cxy[2 * i + 1] > cxy[2 * j + 1] ? cxy[2 * i + 1] : cxy[2 * j + 1];
            int x1a = cxy[2 * i] + swh[2 * i], x1b = cxy[2 * j] + swh[2 * j], y1a = cxy[2 * i + 1] + swh[2 * i + 1], y1b = cxy[2 * j + 1] + swh[2 * j + 1];
            int x1 = x1a < x1b ? x1a : x1b, y1 = y1a < y1b ? y1a : y1b;
            if (!(x0 < x1 && y0 < y1)) continue;
            const int rw = x1 - x0, rh = y1 - y0, W = rw + 2 * gap, H = rh + 2 * gap;
            uint8_t* u1 = (uint8_t*)calloc((size_t)W * H, 1);
            uint8_t* u2 = (uint8_t*)calloc((size_t)W * H, 1);
            for (int y = -gap; y < rh + gap; y++)
                for (int x = -gap; x < rw + gap; x++) {
                    int ya = y0 - cxy[2 * i + 1] + y, xa = x0 - cxy[2 * i] + x, yb = y0 - cxy[2 * j + 1] + y, xb = x0 - cxy[2 * j] + x;
                    uint8_t m1 = (ya >= 0 && xa >= 0 && ya < swh[2 * i + 1] && xa < swh[2 * i]) ? masks[i][(size_t)ya * swh[2 * i] + xa] : 0;
                    uint8_t m2 = (yb >= 0 && xb >= 0 && yb < swh[2 * j + 1] && xb < swh[2 * j]) ? masks[j][(size_t)yb * swh[2 * j] + xb] : 0;
                    const int collision = m1 && m2;
                    /* unique = submask with the collision cleared; the transform measures the distance to unique != 0 */
                    u1[(y + gap) * W + x + gap] = (m1 && !collision) ? 0 : 1;   /* 1 = "unique1 == 0" is non-zero -> not a feature */
                    u2[(y + gap) * W + x + gap] = (m2 && !collision) ? 0 : 1;
                }
            float* d1 = (float*)malloc(sizeof(float) * (size_t)W * H);
            float* d2 = (float*)malloc(sizeof(float) * (size_t)W * H);
            dist_l1(u1, W, H, d1);
            dist_l1(u2, W, H, d2);
            for (int y = 0; y < rh; y++)
                for (int x = 0; x < rw; x++) {
                    const int k = (y + gap) * W + x + gap;
                    if (d1[k] < d2[k]) masks[j][(size_t)(y0 - cxy[2 * j + 1] + y) * swh[2 * j] + (x0 - cxy[2 * j] + x)] = 0;
                    else masks[i][(size_t)(y0 - cxy[2 * i + 1] + y) * swh[2 * i] + (x0 - cxy[2 * i] + x)] = 0;
                }
            free(u1); free(u2); free(d1); free(d2);
        }
}
