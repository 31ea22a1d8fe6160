/*
 * mo_warp.c -- ORACLE (test infrastructure): spherical rotation warper + remap restatement.
 * Reference call sites: image_stitching/image_stitching.cpp:973,:985,:988,:1117,:1138,:1154,:1159.
 * OpenCV sources restated: stitching/detail/warpers_inl.hpp (SphericalProjector,
 * RotationWarperBase), stitching/src/warpers.cpp (setCameraParams), imgproc/src/imgwarp.cpp
 * (remap, INTER_BITS = 5, Q15 weights).  PARITY UNPINNED.  Never linked into the product.
 */
#include "mo_warp.h"
#include <stdlib.h>
#include <string.h>

/* ProjectorBase::setCameraParams: Mat_<float> algebra; OpenCV evaluates the 3x3 float inverse and
 * float gemm with double intermediates, results are stored as float. */
void mo_projector_set(MoProjector* p, float scale, const float K[9], const float R[9]) {
    double kinv[9], d;
    float kinv_f[9];
    int i, j, k;
    p->scale = scale;
    for (i = 0; i < 9; i++) p->k[i] = K[i];
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) p->rinv[i * 3 + j] = R[j * 3 + i];
#define KD(r, c) ((double)K[(r) * 3 + (c)])
    d = KD(0, 0) * (KD(1, 1) * KD(2, 2) - KD(1, 2) * KD(2, 1)) - KD(0, 1) * (KD(1, 0) * KD(2, 2) - KD(1, 2) * KD(2, 0)) +
        KD(0, 2) * (KD(1, 0) * KD(2, 1) - KD(1, 1) * KD(2, 0));
    if (d != 0.) d = 1. / d;
    kinv[0] = (KD(1, 1) * KD(2, 2) - KD(1, 2) * KD(2, 1)) * d;
    kinv[1] = (KD(0, 2) * KD(2, 1) - KD(0, 1) * KD(2, 2)) * d;
    kinv[2] = (KD(0, 1) * KD(1, 2) - KD(0, 2) * KD(1, 1)) * d;
    kinv[3] = (KD(1, 2) * KD(2, 0) - KD(1, 0) * KD(2, 2)) * d;
    kinv[4] = (KD(0, 0) * KD(2, 2) - KD(0, 2) * KD(2, 0)) * d;
    kinv[5] = (KD(0, 2) * KD(1, 0) - KD(0, 0) * KD(1, 2)) * d;
    kinv[6] = (KD(1, 0) * KD(2, 1) - KD(1, 1) * KD(2, 0)) * d;
    kinv[7] = (KD(0, 1) * KD(2, 0) - KD(0, 0) * KD(2, 1)) * d;
    kinv[8] = (KD(0, 0) * KD(1, 1) - KD(0, 1) * KD(1, 0)) * d;
#undef KD
    for (i = 0; i < 9; i++) kinv_f[i] = (float)kinv[i];
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) {
        double s = 0, t = 0;
        for (k = 0; k < 3; k++) {
            s += (double)R[i * 3 + k] * (double)kinv_f[k * 3 + j];
            t += (double)K[i * 3 + k] * (double)p->rinv[k * 3 + j];
        }
        p->r_kinv[i * 3 + j] = (float)s;
        p->k_rinv[i * 3 + j] = (float)t;
    }
}

void mo_map_forward(const MoProjector* p, float x, float y, float* u, float* v) {
    const float* m = p->r_kinv;
    float x_ = (m[0] * x + m[1] * y) + m[2];
    float y_ = (m[3] * x + m[4] * y) + m[5];
    float z_ = (m[6] * x + m[7] * y) + m[8];
    *u = p->scale * mo_atan2f(x_, z_);
    float w = y_ / sqrtf((x_ * x_ + y_ * y_) + z_ * z_);
    *v = p->scale * (MO_PI_F - mo_acosf(w == w ? w : 0));
}

void mo_map_backward(const MoProjector* p, float u, float v, float* x, float* y) {
    const float* m = p->k_rinv;
    u /= p->scale; v /= p->scale;
    float sinv = mo_sinf(MO_PI_F - v);
    float x_ = sinv * mo_sinf(u);
    float y_ = mo_cosf(MO_PI_F - v);
    float z_ = sinv * mo_cosf(u);
    float xx = (m[0] * x_ + m[1] * y_) + m[2] * z_;
    float yy = (m[3] * x_ + m[4] * y_) + m[5] * z_;
    float z = (m[6] * x_ + m[7] * y_) + m[8] * z_;
    if (z > 0) { *x = xx / z; *y = yy / z; }
    else { *x = -1; *y = -1; }
}

void mo_detect_result_roi(const MoProjector* p, int sw, int sh, int* tlx, int* tly, int* brx, int* bry) {
    float tl_uf = FLT_MAX, tl_vf = FLT_MAX, br_uf = -FLT_MAX, br_vf = -FLT_MAX, u, v;
#define MO_UPD() do { if (u < tl_uf) tl_uf = u; if (v < tl_vf) tl_vf = v; if (u > br_uf) br_uf = u; if (v > br_vf) br_vf = v; } while (0)
    for (int x = 0; x < sw; ++x) {
        mo_map_forward(p, (float)x, 0, &u, &v); MO_UPD();
        mo_map_forward(p, (float)x, (float)(sh - 1), &u, &v); MO_UPD();
    }
    for (int y = 0; y < sh; ++y) {
        mo_map_forward(p, 0, (float)y, &u, &v); MO_UPD();
        mo_map_forward(p, (float)(sw - 1), (float)y, &u, &v); MO_UPD();
    }
#undef MO_UPD
    /* detectResultRoiByBorder truncates to int, the spherical override continues in float */
    tl_uf = (float)(int)tl_uf; tl_vf = (float)(int)tl_vf; br_uf = (float)(int)br_uf; br_vf = (float)(int)br_vf;
    float x = p->rinv[1], y = p->rinv[4], z = p->rinv[7];
    if (y > 0.f) {
        float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
        float y_ = p->k[4] * y / z + p->k[5];
        if (x_ > 0.f && x_ < (float)sw && y_ > 0.f && y_ < (float)sh) {
            float pv = (float)(3.14159265358979323846 * (double)p->scale);
            if (0.f < tl_uf) tl_uf = 0.f; if (pv < tl_vf) tl_vf = pv;
            if (0.f > br_uf) br_uf = 0.f; if (pv > br_vf) br_vf = pv;
        }
    }
    x = p->rinv[1]; y = -p->rinv[4]; z = p->rinv[7];
    if (y > 0.f) {
        float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
        float y_ = p->k[4] * y / z + p->k[5];
        if (x_ > 0.f && x_ < (float)sw && y_ > 0.f && y_ < (float)sh) {
            if (0.f < tl_uf) tl_uf = 0.f; if (0.f < tl_vf) tl_vf = 0.f;
            if (0.f > br_uf) br_uf = 0.f; if (0.f > br_vf) br_vf = 0.f;
        }
    }
    *tlx = (int)tl_uf; *tly = (int)tl_vf; *brx = (int)br_uf; *bry = (int)br_vf;
}

void mo_warp_roi(float scale, int sw, int sh, const float K[9], const float R[9], MoRect* roi) {
    MoProjector p;
    int tlx, tly, brx, bry;
    mo_projector_set(&p, scale, K, R);
    mo_detect_result_roi(&p, sw, sh, &tlx, &tly, &brx, &bry);
    roi->x = tlx; roi->y = tly; roi->width = brx + 1 - tlx; roi->height = bry + 1 - tly;
}

void mo_build_maps(const MoProjector* p, int tlx, int tly, int brx, int bry, float* xmap, float* ymap) {
    int W = brx - tlx + 1, v;
#pragma omp parallel for schedule(static)
    for (v = tly; v <= bry; ++v)
        for (int u = tlx; u <= brx; ++u)
            mo_map_backward(p, (float)u, (float)v, &xmap[(size_t)(v - tly) * W + (u - tlx)], &ymap[(size_t)(v - tly) * W + (u - tlx)]);
}

/* x86 cvRound (cvtss2si): out-of-range and NaN give INT_MIN */
static inline int cvround_sat(float v) {
    if (!(fabsf(v) < 2147483648.f)) return (int)0x80000000;
    return (int)lrintf(v);
}
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

int mo_warp_spherical(const uint8_t* src, int w, int h, size_t stride, int cn, float scale, const float K[9],
                      const float R[9], int interp, int border, uint8_t* dst, size_t dstride, int dst_w, int dst_h,
                      int* otlx, int* otly) {
    MoProjector p;
    int tlx, tly, brx, bry, v;
    mo_projector_set(&p, scale, K, R);
    mo_detect_result_roi(&p, w, h, &tlx, &tly, &brx, &bry);
    if (otlx) *otlx = tlx; if (otly) *otly = tly;
    if (dst_w != brx - tlx + 1 || dst_h != bry - tly + 1) return -1;
    if (!((interp == MO_INTER_LINEAR && border == MO_BORDER_REFLECT) || (interp == MO_INTER_NEAREST && border == MO_BORDER_CONSTANT)))
        return -2;
#pragma omp parallel for schedule(static)
    for (v = tly; v <= bry; ++v) {
        uint8_t* d = dst + (size_t)(v - tly) * dstride;
        for (int u = tlx; u <= brx; ++u) {
            float x, y;
            mo_map_backward(&p, (float)u, (float)v, &x, &y);
            uint8_t* o = d + (size_t)(u - tlx) * cn;
            if (interp == MO_INTER_NEAREST) {
                int sx = sat_short(cvround_sat(x)), sy = sat_short(cvround_sat(y));
                if ((unsigned)sx < (unsigned)w && (unsigned)sy < (unsigned)h)
                    for (int c = 0; c < cn; c++) o[c] = src[(size_t)sy * stride + (size_t)sx * cn + c];
                else
                    for (int c = 0; c < cn; c++) o[c] = 0;
            } else {
                int sxq = cvround_sat(x * 32.f), syq = cvround_sat(y * 32.f);
                int fx = sxq & 31, fy = syq & 31;
                int sx = sat_short(sxq >> 5), sy = sat_short(syq >> 5);
                int x0 = mo_reflect(sx, w), x1 = mo_reflect(sx + 1, w);
                int y0 = mo_reflect(sy, h), y1 = mo_reflect(sy + 1, h);
                int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
                const uint8_t* r0 = src + (size_t)y0 * stride;
                const uint8_t* r1 = src + (size_t)y1 * stride;
                for (int c = 0; c < cn; c++) {
                    int s = r0[x0 * cn + c] * w00 + r0[x1 * cn + c] * w01 + r1[x0 * cn + c] * w10 + r1[x1 * cn + c] * w11;
                    o[c] = mo_sat_u8((s + (1 << 14)) >> 15);
                }
            }
        }
    }
    return 0;
}
