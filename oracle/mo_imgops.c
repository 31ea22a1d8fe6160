/* mo_imgops.c -- see mo_imgops.h.  TEST INFRASTRUCTURE ONLY. */
#include "mo_imgops.h"
#include "mo_common.h"
#include <stdlib.h>
#include <string.h>

/* resize.cpp: coordinate of destination index i is (i + 0.5) * scale - 0.5; 8.8 fixed-point weight of the
 * right / lower tap; clamped to the first / last sample */
static void coeffs(int dlen, int slen, double scale, int* ofs, int* m1) {
    for (int i = 0; i < dlen; i++) {
        double v = ((double)i + 0.5) * scale - 0.5;
        int iv = mo_floor_d(v);
        if (iv < 0) { ofs[i] = 0; m1[i] = 0; }
        else if (iv >= slen - 1) { ofs[i] = slen - 1; m1[i] = 0; }
        else { ofs[i] = iv; m1[i] = mo_round_d((v - (double)iv) * 256.0); }
    }
}

void mo_resize_dsize(int sw, int sh, int dw_in, int dh_in, double fx, double fy, int* dw, int* dh) {
    if (dw_in > 0 && dh_in > 0) { *dw = dw_in; *dh = dh_in; }
    else { *dw = mo_round_d((double)sw * fx); *dh = mo_round_d((double)sh * fy); }
}

void mo_resize_linear_exact_u8_ex(const uint8_t* src, int sw, int sh, size_t sstride, int cn, uint8_t* dst, int dw, int dh,
                                  size_t dstride, double fx, double fy, int by_factor) {
    /* resize(): inv_scale = the given factor, or dsize / ssize when dsize is given; scale = 1 / inv_scale */
    double sx = by_factor ? 1.0 / fx : 1.0 / ((double)dw / (double)sw);
    double sy = by_factor ? 1.0 / fy : 1.0 / ((double)dh / (double)sh);
    int* xo = (int*)malloc(sizeof(int) * (size_t)dw * 2);
    int* yo = (int*)malloc(sizeof(int) * (size_t)dh * 2);
    int *xm = xo + dw, *ym = yo + dh;
    coeffs(dw, sw, sx, xo, xm);
    coeffs(dh, sh, sy, yo, ym);
    for (int y = 0; y < dh; y++) {
        int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : y0, my1 = ym[y], my0 = 256 - my1;
        const uint8_t* r0 = src + (size_t)y0 * sstride;
        const uint8_t* r1 = src + (size_t)y1 * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : x0, mx1 = xm[x], mx0 = 256 - mx1;
            for (int c = 0; c < cn; c++) {
                unsigned h0 = (unsigned)r0[x0 * cn + c] * mx0 + (unsigned)r0[x1 * cn + c] * mx1;
                unsigned h1 = (unsigned)r1[x0 * cn + c] * mx0 + (unsigned)r1[x1 * cn + c] * mx1;
                d[x * cn + c] = (uint8_t)((h0 * my0 + h1 * my1 + (1u << 15)) >> 16);
            }
        }
    }
    free(xo); free(yo);
}

void mo_rotate_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn, int code, uint8_t* dst, size_t dstride) {
    /* ROTATE_90_CLOCKWISE = transpose + flip around y; ROTATE_180 = flip both; 90 CCW = transpose + flip around x */
    int dw = code == 1 ? sw : sh, dh = code == 1 ? sh : sw;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int sx, sy;
            if (code == 0) { sx = y; sy = sh - 1 - x; }
            else if (code == 1) { sx = sw - 1 - x; sy = sh - 1 - y; }
            else { sx = sw - 1 - y; sy = x; }
            memcpy(dst + (size_t)y * dstride + (size_t)x * cn, src + (size_t)sy * sstride + (size_t)sx * cn, (size_t)cn);
        }
}

void mo_dilate3x3_u8(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t m = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    uint8_t v = src[(size_t)yy * sstride + xx];
                    if (v > m) m = v;
                }
            dst[(size_t)y * dstride + x] = m;
        }
}

void mo_seam_mask_apply(const uint8_t* seam, int sw, int sh, size_t sstride, uint8_t* mask, int mw, int mh, size_t mstride) {
    uint8_t* dil = (uint8_t*)malloc((size_t)sw * sh);
    uint8_t* big = (uint8_t*)malloc((size_t)mw * mh);
    mo_dilate3x3_u8(seam, sw, sh, sstride, dil, (size_t)sw);
    mo_resize_linear_exact_u8_ex(dil, sw, sh, (size_t)sw, 1, big, mw, mh, (size_t)mw, 0, 0, 0);
    for (int y = 0; y < mh; y++)
        for (int x = 0; x < mw; x++) mask[(size_t)y * mstride + x] &= big[(size_t)y * mw + x];
    free(dil); free(big);
}
