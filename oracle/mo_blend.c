/*
 * mo_blend.c -- ORACLE (test infrastructure): multi-band / feather / plain blender restatement.
 * Reference call sites: image_stitching/image_stitching.cpp:1175-1192, :1218, :1225.
 * OpenCV sources restated: stitching/src/blenders.cpp, imgproc/src/pyramids.cpp,
 * imgproc/src/distransform.cpp.  PARITY UNPINNED.  Never linked into the product.
 */
#include "mo_blend.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define MO_MAX_BANDS 16
static const float WEIGHT_EPS = 1e-5f;

struct MoBlender {
    int type, actual_bands, num_bands;
    float sharpness;
    int rx, ry, rw, rh;  /* dst_roi_ (padded for multi-band) */
    int fw, fh;          /* dst_roi_final_ size */
    int lw[MO_MAX_BANDS + 1], lh[MO_MAX_BANDS + 1];
    int16_t* lap[MO_MAX_BANDS + 1]; /* level 0 doubles as dst_ for feather / no blending */
    float* wgt[MO_MAX_BANDS + 1];
    uint8_t* dst_mask;              /* plain blender */
    int prepared;
};

int mo_blend_config(int blend_type, float blend_strength, int pano_w, int pano_h, int* num_bands, float* sharpness) {
    float blend_width = sqrtf((float)(pano_w * pano_h)) * blend_strength / 100.f;
    *num_bands = 0; *sharpness = 0.f;
    if (blend_width < 1.f) return MO_BLEND_NO;
    if (blend_type == MO_BLEND_MULTI_BAND) *num_bands = (int)(ceil(log((double)blend_width) / log(2.)) - 1.);
    else if (blend_type == MO_BLEND_FEATHER) *sharpness = 1.f / blend_width;
    return blend_type;
}

void mo_result_roi(const int* c, const int* s, int n, int* x, int* y, int* w, int* h) {
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; i++) {
        if (c[2 * i] < tlx) tlx = c[2 * i];
        if (c[2 * i + 1] < tly) tly = c[2 * i + 1];
        if (c[2 * i] + s[2 * i] > brx) brx = c[2 * i] + s[2 * i];
        if (c[2 * i + 1] + s[2 * i + 1] > bry) bry = c[2 * i + 1] + s[2 * i + 1];
    }
    *x = tlx; *y = tly; *w = brx - tlx; *h = bry - tly;
}

MoBlender* mo_blender_create(int type, int num_bands, float sharpness) {
    MoBlender* b = (MoBlender*)calloc(1, sizeof(MoBlender));
    b->type = type; b->actual_bands = num_bands; b->sharpness = sharpness;
    return b;
}

static void blender_release(MoBlender* b) {
    for (int i = 0; i <= MO_MAX_BANDS; i++) { free(b->lap[i]); free(b->wgt[i]); b->lap[i] = NULL; b->wgt[i] = NULL; }
    free(b->dst_mask); b->dst_mask = NULL; b->prepared = 0;
}
void mo_blender_destroy(MoBlender* b) { if (b) { blender_release(b); free(b); } }
int mo_blender_num_bands(const MoBlender* b) { return b->num_bands; }
void mo_blender_roi(const MoBlender* b, int* x, int* y, int* w, int* h, int* fw, int* fh) {
    *x = b->rx; *y = b->ry; *w = b->rw; *h = b->rh; *fw = b->fw; *fh = b->fh;
}

int mo_blender_prepare(MoBlender* b, const int* corners, const int* sizes, int n) {
    blender_release(b);
    mo_result_roi(corners, sizes, n, &b->rx, &b->ry, &b->rw, &b->rh);
    b->fw = b->rw; b->fh = b->rh;
    b->num_bands = 0;
    if (b->type == MO_BLEND_MULTI_BAND) {
        double max_len = (double)(b->rw > b->rh ? b->rw : b->rh);
        int lim = (int)ceil(log(max_len) / log(2.0));
        b->num_bands = b->actual_bands < lim ? b->actual_bands : lim;
        if (b->num_bands > MO_MAX_BANDS || b->num_bands < 0) return -1;
        int q = 1 << b->num_bands;
        b->rw += (q - b->rw % q) % q;
        b->rh += (q - b->rh % q) % q;
    }
    b->lw[0] = b->rw; b->lh[0] = b->rh;
    for (int i = 1; i <= b->num_bands; i++) { b->lw[i] = (b->lw[i - 1] + 1) / 2; b->lh[i] = (b->lh[i - 1] + 1) / 2; }
    for (int i = 0; i <= b->num_bands; i++) {
        b->lap[i] = (int16_t*)calloc((size_t)b->lw[i] * b->lh[i] * 3, sizeof(int16_t));
        if (b->type != MO_BLEND_NO) b->wgt[i] = (float*)calloc((size_t)b->lw[i] * b->lh[i], sizeof(float));
    }
    if (b->type == MO_BLEND_NO) b->dst_mask = (uint8_t*)calloc((size_t)b->rw * b->rh, 1);
    b->prepared = 1;
    return 0;
}

/* ---- imgproc pyramids.cpp: pyrDown / pyrUp, 5-tap [1 4 6 4 1] ---- */
void mo_pyr_down_s16(const int16_t* src, int w, int h, int cn, int16_t* dst) {
    int dw = (w + 1) / 2, dh = (h + 1) / 2, y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < dh; y++) {
        const int16_t* r[5];
        for (int k = 0; k < 5; k++) r[k] = src + (size_t)mo_reflect101(2 * y - 2 + k, h) * w * cn;
        for (int x = 0; x < dw; x++) {
            int xi[5];
            for (int k = 0; k < 5; k++) xi[k] = mo_reflect101(2 * x - 2 + k, w) * cn;
            for (int c = 0; c < cn; c++) {
                int hr[5];
                for (int k = 0; k < 5; k++)
                    hr[k] = r[k][xi[2] + c] * 6 + (r[k][xi[1] + c] + r[k][xi[3] + c]) * 4 + r[k][xi[0] + c] + r[k][xi[4] + c];
                int v = hr[2] * 6 + (hr[1] + hr[3]) * 4 + hr[0] + hr[4];
                dst[((size_t)y * dw + x) * cn + c] = (int16_t)((v + 128) >> 8);
            }
        }
    }
}

void mo_pyr_down_f32(const float* src, int w, int h, float* dst) {
    int dw = (w + 1) / 2, dh = (h + 1) / 2, y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < dh; y++) {
        const float* r[5];
        for (int k = 0; k < 5; k++) r[k] = src + (size_t)mo_reflect101(2 * y - 2 + k, h) * w;
        for (int x = 0; x < dw; x++) {
            int xi[5];
            float hr[5];
            for (int k = 0; k < 5; k++) xi[k] = mo_reflect101(2 * x - 2 + k, w);
            for (int k = 0; k < 5; k++)
                hr[k] = ((r[k][xi[2]] * 6.f + (r[k][xi[1]] + r[k][xi[3]]) * 4.f) + r[k][xi[0]]) + r[k][xi[4]];
            float v = ((hr[2] * 6.f + (hr[1] + hr[3]) * 4.f) + hr[0]) + hr[4];
            dst[(size_t)y * dw + x] = v * (1.f / 256.f);
        }
    }
}

/* pyrUp to exactly twice the size: left/top neighbour of sample 0 is sample 1 (REFLECT_101),
 * right/bottom neighbour of the last sample is the last sample itself */
void mo_pyr_up_s16(const int16_t* src, int w, int h, int cn, int16_t* dst) {
    int dw = 2 * w, y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < h; y++) {
        int ym = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yp = y + 1 < h ? y + 1 : h - 1;
        const int16_t* r0 = src + (size_t)ym * w * cn;
        const int16_t* r1 = src + (size_t)y * w * cn;
        const int16_t* r2 = src + (size_t)yp * w * cn;
        int16_t* d0 = dst + (size_t)(2 * y) * dw * cn;
        int16_t* d1 = dst + (size_t)(2 * y + 1) * dw * cn;
        for (int x = 0; x < w; x++) {
            int xm = (x > 0 ? x - 1 : (w > 1 ? 1 : 0)) * cn, xc = x * cn, xp = (x + 1 < w ? x + 1 : w - 1) * cn;
            for (int c = 0; c < cn; c++) {
                int e0 = r0[xm + c] + r0[xc + c] * 6 + r0[xp + c], o0 = (r0[xc + c] + r0[xp + c]) * 4;
                int e1 = r1[xm + c] + r1[xc + c] * 6 + r1[xp + c], o1 = (r1[xc + c] + r1[xp + c]) * 4;
                int e2 = r2[xm + c] + r2[xc + c] * 6 + r2[xp + c], o2 = (r2[xc + c] + r2[xp + c]) * 4;
                d0[(2 * x) * cn + c] = (int16_t)((e0 + e1 * 6 + e2 + 32) >> 6);
                d0[(2 * x + 1) * cn + c] = (int16_t)((o0 + o1 * 6 + o2 + 32) >> 6);
                d1[(2 * x) * cn + c] = (int16_t)(((e1 + e2) * 4 + 32) >> 6);
                d1[(2 * x + 1) * cn + c] = (int16_t)(((o1 + o2) * 4 + 32) >> 6);
            }
        }
    }
}

/* distanceTransform(mask, DIST_L1, 3) -> f32: exact city-block distance to the nearest zero pixel,
 * clamped at 8192 (DIST_MAX * 2^-16) */
void mo_distance_l1(const uint8_t* mask, size_t stride, int w, int h, float* dist) {
    const int INF = 8192;
    int* t = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = mask[(size_t)y * stride + x] ? INF : 0;
            if (v) {
                if (x > 0 && t[(size_t)y * w + x - 1] + 1 < v) v = t[(size_t)y * w + x - 1] + 1;
                if (y > 0 && t[(size_t)(y - 1) * w + x] + 1 < v) v = t[(size_t)(y - 1) * w + x] + 1;
            }
            t[(size_t)y * w + x] = v;
        }
    for (int y = h - 1; y >= 0; y--)
        for (int x = w - 1; x >= 0; x--) {
            int v = t[(size_t)y * w + x];
            if (x + 1 < w && t[(size_t)y * w + x + 1] + 1 < v) v = t[(size_t)y * w + x + 1] + 1;
            if (y + 1 < h && t[(size_t)(y + 1) * w + x] + 1 < v) v = t[(size_t)(y + 1) * w + x] + 1;
            t[(size_t)y * w + x] = v;
            dist[(size_t)y * w + x] = (float)(v > INF ? INF : v);
        }
    free(t);
}

static int16_t* pad_reflect_s16x3(const int16_t* img, size_t stride, int w, int h, int top, int bottom, int left, int right) {
    int W = w + left + right, H = h + top + bottom, y;
    int16_t* p = (int16_t*)malloc(sizeof(int16_t) * 3 * (size_t)W * H);
#pragma omp parallel for schedule(static)
    for (y = 0; y < H; y++) {
        const int16_t* s = img + (size_t)mo_reflect(y - top, h) * stride;
        int16_t* d = p + (size_t)y * W * 3;
        for (int x = 0; x < W; x++) {
            int sx = mo_reflect(x - left, w);
            d[3 * x] = s[3 * sx]; d[3 * x + 1] = s[3 * sx + 1]; d[3 * x + 2] = s[3 * sx + 2];
        }
    }
    return p;
}

static int feed_plain(MoBlender* b, const int16_t* img, size_t is, const uint8_t* mask, size_t ms, int w, int h, int tlx, int tly) {
    int dx = tlx - b->rx, dy = tly - b->ry;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t o = (size_t)(dy + y) * b->rw + dx + x;
            if (mask[(size_t)y * ms + x]) for (int c = 0; c < 3; c++) b->lap[0][o * 3 + c] = img[(size_t)y * is + 3 * x + c];
            b->dst_mask[o] |= mask[(size_t)y * ms + x];
        }
    return 0;
}

static int feed_feather(MoBlender* b, const int16_t* img, size_t is, const uint8_t* mask, size_t ms, int w, int h, int tlx, int tly) {
    int dx = tlx - b->rx, dy = tly - b->ry;
    float* wm = (float*)malloc(sizeof(float) * (size_t)w * h);
    mo_distance_l1(mask, ms, w, h, wm);
    for (size_t i = 0; i < (size_t)w * h; i++) { float v = wm[i] * b->sharpness; wm[i] = v > 1.f ? 1.f : v; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t o = (size_t)(dy + y) * b->rw + dx + x;
            float wv = wm[(size_t)y * w + x];
            for (int c = 0; c < 3; c++)
                b->lap[0][o * 3 + c] = (int16_t)(b->lap[0][o * 3 + c] + (int16_t)((float)img[(size_t)y * is + 3 * x + c] * wv));
            b->wgt[0][o] += wv;
        }
    free(wm);
    return 0;
}

static int feed_multiband(MoBlender* b, const int16_t* img, size_t is, const uint8_t* mask, size_t ms, int w, int h, int tlx, int tly) {
    const int nb = b->num_bands, q = 1 << nb;
    int gap = 3 * q;
    int brx_roi = b->rx + b->rw, bry_roi = b->ry + b->rh;
    int tnx = b->rx > tlx - gap ? b->rx : tlx - gap, tny = b->ry > tly - gap ? b->ry : tly - gap;
    int bnx = brx_roi < tlx + w + gap ? brx_roi : tlx + w + gap, bny = bry_roi < tly + h + gap ? bry_roi : tly + h + gap;
    tnx = b->rx + (((tnx - b->rx) >> nb) << nb);
    tny = b->ry + (((tny - b->ry) >> nb) << nb);
    int width = bnx - tnx, height = bny - tny;
    width += (q - width % q) % q;
    height += (q - height % q) % q;
    bnx = tnx + width; bny = tny + height;
    int dy = bny - bry_roi > 0 ? bny - bry_roi : 0, dx = bnx - brx_roi > 0 ? bnx - brx_roi : 0;
    tnx -= dx; bnx -= dx; tny -= dy; bny -= dy;
    int top = tly - tny, left = tlx - tnx, bottom = bny - tly - h, right = bnx - tlx - w;
    if (top < 0 || left < 0 || bottom < 0 || right < 0) return -3;

    int pw[MO_MAX_BANDS + 1], ph[MO_MAX_BANDS + 1];
    int16_t* pyr[MO_MAX_BANDS + 1];
    float* wp[MO_MAX_BANDS + 1];
    pw[0] = width; ph[0] = height;
    pyr[0] = pad_reflect_s16x3(img, is, w, h, top, bottom, left, right);
    for (int i = 0; i < nb; i++) {
        pw[i + 1] = (pw[i] + 1) / 2; ph[i + 1] = (ph[i] + 1) / 2;
        pyr[i + 1] = (int16_t*)malloc(sizeof(int16_t) * 3 * (size_t)pw[i + 1] * ph[i + 1]);
        mo_pyr_down_s16(pyr[i], pw[i], ph[i], 3, pyr[i + 1]);
    }
    for (int i = 0; i < nb; i++) {
        int16_t* up = (int16_t*)malloc(sizeof(int16_t) * 3 * (size_t)pw[i] * ph[i]);
        mo_pyr_up_s16(pyr[i + 1], pw[i + 1], ph[i + 1], 3, up);
        size_t n = (size_t)pw[i] * ph[i] * 3;
        for (size_t k = 0; k < n; k++) pyr[i][k] = mo_sat_s16((int)pyr[i][k] - (int)up[k]);
        free(up);
    }
    /* weight pyramid: mask * (1/255) in f32, zero-padded, pyrDown chain */
    wp[0] = (float*)calloc((size_t)width * height, sizeof(float));
    {
        const float a = (float)(1. / 255.);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) wp[0][(size_t)(y + top) * width + x + left] = (float)mask[(size_t)y * ms + x] * a;
    }
    for (int i = 0; i < nb; i++) {
        wp[i + 1] = (float*)malloc(sizeof(float) * (size_t)pw[i + 1] * ph[i + 1]);
        mo_pyr_down_f32(wp[i], pw[i], ph[i], wp[i + 1]);
    }
    int y_tl = tny - b->ry, y_br = bny - b->ry, x_tl = tnx - b->rx, x_br = bnx - b->rx;
    for (int i = 0; i <= nb; i++) {
        int rw_ = x_br - x_tl, rh_ = y_br - y_tl, y;
#pragma omp parallel for schedule(static)
        for (y = 0; y < rh_; y++) {
            const int16_t* s = pyr[i] + (size_t)y * pw[i] * 3;
            const float* wr = wp[i] + (size_t)y * pw[i];
            int16_t* d = b->lap[i] + ((size_t)(y_tl + y) * b->lw[i] + x_tl) * 3;
            float* dw = b->wgt[i] + (size_t)(y_tl + y) * b->lw[i] + x_tl;
            for (int x = 0; x < rw_; x++) {
                for (int c = 0; c < 3; c++) d[3 * x + c] = (int16_t)(d[3 * x + c] + (int16_t)((float)s[3 * x + c] * wr[x]));
                dw[x] += wr[x];
            }
        }
        x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
    }
    for (int i = 0; i <= nb; i++) { free(pyr[i]); free(wp[i]); }
    return 0;
}

int mo_blender_feed(MoBlender* b, const int16_t* img, size_t is, const uint8_t* mask, size_t ms, int w, int h, int tlx, int tly) {
    if (!b->prepared) return -1;
    if (tlx < b->rx || tly < b->ry || tlx + w > b->rx + b->fw || tly + h > b->ry + b->fh) return -2;
    if (b->type == MO_BLEND_MULTI_BAND) return feed_multiband(b, img, is, mask, ms, w, h, tlx, tly);
    if (b->type == MO_BLEND_FEATHER) return feed_feather(b, img, is, mask, ms, w, h, tlx, tly);
    return feed_plain(b, img, is, mask, ms, w, h, tlx, tly);
}

int mo_blender_blend(MoBlender* b, int16_t* dst, size_t ds, uint8_t* dmask, size_t dms) {
    if (!b->prepared) return -1;
    if (b->type != MO_BLEND_NO) {
        /* normalizeUsingWeightMap on every level */
        for (int i = 0; i <= b->num_bands; i++) {
            size_t n = (size_t)b->lw[i] * b->lh[i];
            for (size_t k = 0; k < n; k++) {
                float wv = b->wgt[i][k] + WEIGHT_EPS;
                for (int c = 0; c < 3; c++) b->lap[i][3 * k + c] = (int16_t)((float)b->lap[i][3 * k + c] / wv);
            }
        }
        /* restoreImageFromLaplacePyr */
        for (int i = b->num_bands; i > 0; i--) {
            size_t n = (size_t)b->lw[i - 1] * b->lh[i - 1] * 3;
            int16_t* up = (int16_t*)malloc(sizeof(int16_t) * n);
            mo_pyr_up_s16(b->lap[i], b->lw[i], b->lh[i], 3, up);
            for (size_t k = 0; k < n; k++) b->lap[i - 1][k] = mo_sat_s16((int)up[k] + (int)b->lap[i - 1][k]);
            free(up);
        }
    }
    for (int y = 0; y < b->fh; y++)
        for (int x = 0; x < b->fw; x++) {
            size_t o = (size_t)y * b->rw + x;
            int m = b->type == MO_BLEND_NO ? (b->dst_mask[o] != 0) : (b->wgt[0][o] > WEIGHT_EPS);
            dmask[(size_t)y * dms + x] = b->type == MO_BLEND_NO ? b->dst_mask[o] : (m ? 255 : 0);
            for (int c = 0; c < 3; c++) dst[(size_t)y * ds + 3 * x + c] = m ? b->lap[0][o * 3 + c] : 0;
        }
    return 0;
}

const int16_t* mo_blender_level_lap(const MoBlender* b, int level, int* w, int* h) { *w = b->lw[level]; *h = b->lh[level]; return b->lap[level]; }
const float* mo_blender_level_weight(const MoBlender* b, int level, int* w, int* h) { *w = b->lw[level]; *h = b->lh[level]; return b->wgt[level]; }
