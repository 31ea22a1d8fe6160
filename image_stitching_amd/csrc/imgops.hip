// imgops.hip -- the small image operators either side of the hot path:
//   cv::resize(..., INTER_LINEAR_EXACT)   image_stitching/image_stitching.cpp:573-580 (work scale), :619 (seam scale),
//                                         :1144 (compose scale), :1170 (seam mask -> compose size)
//   cv::rotate(90 CW / 180 / 90 CCW)      image_stitching/image_stitching.cpp:571, :576
//   dilate(3x3) -> resize -> AND          image_stitching/image_stitching.cpp:1169-1171, fused into one kernel
// All of it is byte arithmetic bound by HBM: one thread per destination pixel (16 B of output per thread for
// the mask kernel), coalesced along rows, coefficient tables built on the host exactly as resize() does.
#include "common.h"
#include "dev_math.h"
#include <vector>

namespace {

// resize.cpp: destination index i samples (i + 0.5) * scale - 0.5; 8.8 fixed-point weight of the right / lower
// tap; clamped to the first / last sample
void coeffs(int dlen, int slen, double scale, int* ofs, int* m1) {
    for (int i = 0; i < dlen; i++) {
        double v = ((double)i + 0.5) * scale - 0.5;
        int iv = (int)v; iv -= (iv > v);
        if (iv < 0) { ofs[i] = 0; m1[i] = 0; }
        else if (iv >= slen - 1) { ofs[i] = slen - 1; m1[i] = 0; }
        else { ofs[i] = iv; m1[i] = mis_round_d((v - (double)iv) * 256.0); }
    }
}

// device copy of {xofs[dw], xm1[dw], yofs[dh], ym1[dh]} in the context's grow-only scratch
int upload_tables(MisContext* ctx, int sw, int sh, int dw, int dh, double sx, double sy, const int** tab) {
    std::vector<int> h(2 * (size_t)dw + 2 * (size_t)dh);
    coeffs(dw, sw, sx, h.data(), h.data() + dw);
    coeffs(dh, sh, sy, h.data() + 2 * dw, h.data() + 2 * dw + dh);
    const size_t bytes = h.size() * sizeof(int);
    if (ctx->stage_bytes < bytes) {
        if (ctx->stage) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
        MIS_HIP(ctx, hipMalloc(&ctx->stage, bytes * 2 + 4096));
        ctx->stage_bytes = bytes * 2 + 4096;
    }
    // pageable source: the copy is complete for the host when the call returns, so `h` may go out of scope
    MIS_HIP(ctx, hipMemcpyAsync(ctx->stage, h.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *tab = (const int*)ctx->stage;
    return MIS_OK;
}

template <int CN>
__global__ __launch_bounds__(256) void resize_exact_kernel(const uint8_t* __restrict__ src, int sw, int sh, size_t ss, uint8_t* __restrict__ dst, int dw,
                                                          int dh, size_t ds, const int* __restrict__ tab) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int *xo = tab, *xm = tab + dw, *yo = tab + 2 * dw, *ym = tab + 2 * dw + dh;
    const int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : x0, mx1 = xm[x], mx0 = 256 - mx1;
    const int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : y0, my1 = ym[y], my0 = 256 - my1;
    const uint8_t* r0 = src + (size_t)y0 * ss;
    const uint8_t* r1 = src + (size_t)y1 * ss;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const unsigned h0 = (unsigned)r0[x0 * CN + c] * mx0 + (unsigned)r0[x1 * CN + c] * mx1;
        const unsigned h1 = (unsigned)r1[x0 * CN + c] * mx0 + (unsigned)r1[x1 * CN + c] * mx1;
        dst[(size_t)y * ds + (size_t)x * CN + c] = (uint8_t)((h0 * my0 + h1 * my1 + (1u << 15)) >> 16);
    }
}

template <int CN>
__global__ __launch_bounds__(256) void rotate_kernel(const uint8_t* __restrict__ src, int sw, int sh, size_t ss, int code, uint8_t* __restrict__ dst,
                                                    int dw, int dh, size_t ds) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    int sx, sy;
    if (code == 0) { sx = y; sy = sh - 1 - x; }
    else if (code == 1) { sx = sw - 1 - x; sy = sh - 1 - y; }
    else { sx = sw - 1 - y; sy = x; }
    const uint8_t* s = src + (size_t)sy * ss + (size_t)sx * CN;
    uint8_t* d = dst + (size_t)y * ds + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) d[c] = s[c];
}

// max over the 3x3 neighbourhood inside the image (the dilate border value never wins)
__device__ __forceinline__ unsigned dil3(const uint8_t* __restrict__ s, int w, int h, size_t ss, int x, int y) {
    unsigned m = 0;
    const int ya = y > 0 ? y - 1 : 0, yb = y + 1 < h ? y + 1 : h - 1, xa = x > 0 ? x - 1 : 0, xb = x + 1 < w ? x + 1 : w - 1;
    for (int yy = ya; yy <= yb; yy++)
        for (int xx = xa; xx <= xb; xx++) m = max(m, (unsigned)s[(size_t)yy * ss + xx]);
    return m;
}

// mask &= resize(dilate(seam)): 4 destination pixels per thread; the seam mask is ~0.1 MP and stays in L2
__global__ __launch_bounds__(256) void seam_mask_kernel(const uint8_t* __restrict__ seam, int sw, int sh, size_t ss, uint8_t* __restrict__ mask, int mw,
                                                       int mh, size_t ms, const int* __restrict__ tab) {
    const int xb = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (xb >= mw) return;
    const int *xo = tab, *xm = tab + mw, *yo = tab + 2 * mw, *ym = tab + 2 * mw + mh;
    const int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : y0, my1 = ym[y], my0 = 256 - my1;
    uint8_t* m = mask + (size_t)y * ms;
    for (int k = 0; k < 4 && xb + k < mw; k++) {
        const int x = xb + k;
        if (m[x] == 0) continue;   // x & 0 = 0
        const int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : x0, mx1 = xm[x], mx0 = 256 - mx1;
        const unsigned h0 = dil3(seam, sw, sh, ss, x0, y0) * mx0 + dil3(seam, sw, sh, ss, x1, y0) * mx1;
        const unsigned h1 = dil3(seam, sw, sh, ss, x0, y1) * mx0 + dil3(seam, sw, sh, ss, x1, y1) * mx1;
        m[x] &= (uint8_t)((h0 * my0 + h1 * my1 + (1u << 15)) >> 16);
    }
}

}  // namespace

extern "C" int mis_resize_linear_exact(MisContext* ctx, const MisImage* src, int dst_w, int dst_h, double fx, double fy, MisImage* dst) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, src && dst && src->data, MIS_E_INVALID, "null image");
    MIS_CHECK(ctx, src->dtype == MIS_U8 && (src->channels == 1 || src->channels == 3), MIS_E_UNSUPPORTED, "resize: 8UC1 / 8UC3 only");
    const bool by_factor = !(dst_w > 0 && dst_h > 0);
    MIS_CHECK(ctx, !by_factor || (fx > 0 && fy > 0), MIS_E_INVALID, "resize needs a destination size or positive scale factors");
    // resize(): dsize = saturate_cast<int>(ssize * inv_scale) when empty; inv_scale = dsize / ssize when given
    const int dw = by_factor ? mis_round_d((double)src->width * fx) : dst_w, dh = by_factor ? mis_round_d((double)src->height * fy) : dst_h;
    MIS_CHECK(ctx, dw > 0 && dh > 0 && dw <= 65535 && dh <= 65535, MIS_E_INVALID, "resize destination %dx%d out of range", dw, dh);
    const double sx = by_factor ? 1.0 / fx : 1.0 / ((double)dw / (double)src->width);
    const double sy = by_factor ? 1.0 / fy : 1.0 / ((double)dh / (double)src->height);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage din, dout;
    int rc;
    if ((rc = mis_dev_image_in(ctx, src, &din)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, dst, dw, dh, src->channels, MIS_U8, &dout)) != MIS_OK) { mis_dev_image_release(ctx, &din); return rc; }
    const int* tab;
    if ((rc = upload_tables(ctx, src->width, src->height, dw, dh, sx, sy, &tab)) != MIS_OK) return rc;
    dim3 grid((dw + 255) / 256, dh), block(256);
    if (src->channels == 3)
        hipLaunchKernelGGL(resize_exact_kernel<3>, grid, block, 0, ctx->stream, (const uint8_t*)din.data, src->width, src->height, din.stride, (uint8_t*)dout.data, dw, dh, dout.stride, tab);
    else
        hipLaunchKernelGGL(resize_exact_kernel<1>, grid, block, 0, ctx->stream, (const uint8_t*)din.data, src->width, src->height, din.stride, (uint8_t*)dout.data, dw, dh, dout.stride, tab);
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dout)) != MIS_OK) return rc;
    return mis_dev_image_release(ctx, &din);
}

extern "C" int mis_rotate(MisContext* ctx, const MisImage* src, int rotate_code, MisImage* dst) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, src && dst && src->data, MIS_E_INVALID, "null image");
    MIS_CHECK(ctx, src->dtype == MIS_U8 && (src->channels == 1 || src->channels == 3), MIS_E_UNSUPPORTED, "rotate: 8UC1 / 8UC3 only");
    MIS_CHECK(ctx, rotate_code >= 0 && rotate_code <= 2, MIS_E_INVALID, "rotate code must be 0 (90 CW), 1 (180) or 2 (90 CCW)");
    const int dw = rotate_code == 1 ? src->width : src->height, dh = rotate_code == 1 ? src->height : src->width;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage din, dout;
    int rc;
    if ((rc = mis_dev_image_in(ctx, src, &din)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, dst, dw, dh, src->channels, MIS_U8, &dout)) != MIS_OK) { mis_dev_image_release(ctx, &din); return rc; }
    dim3 grid((dw + 255) / 256, dh), block(256);
    if (src->channels == 3)
        hipLaunchKernelGGL(rotate_kernel<3>, grid, block, 0, ctx->stream, (const uint8_t*)din.data, src->width, src->height, din.stride, rotate_code, (uint8_t*)dout.data, dw, dh, dout.stride);
    else
        hipLaunchKernelGGL(rotate_kernel<1>, grid, block, 0, ctx->stream, (const uint8_t*)din.data, src->width, src->height, din.stride, rotate_code, (uint8_t*)dout.data, dw, dh, dout.stride);
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, dst, &dout)) != MIS_OK) return rc;
    return mis_dev_image_release(ctx, &din);
}

extern "C" int mis_seam_mask_apply(MisContext* ctx, const MisImage* seam, MisImage* mask) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, seam && mask && seam->data && mask->data, MIS_E_INVALID, "null image");
    MIS_CHECK(ctx, seam->dtype == MIS_U8 && seam->channels == 1 && mask->dtype == MIS_U8 && mask->channels == 1, MIS_E_UNSUPPORTED, "masks must be 8UC1");
    MIS_CHECK(ctx, seam->width > 0 && seam->height > 0 && mask->width > 0 && mask->height > 0, MIS_E_INVALID, "empty mask");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage ds, dm;
    int rc;
    if ((rc = mis_dev_image_in(ctx, seam, &ds)) != MIS_OK) return rc;
    if ((rc = mis_dev_image_out(ctx, mask, mask->width, mask->height, 1, MIS_U8, &dm)) != MIS_OK) { mis_dev_image_release(ctx, &ds); return rc; }
    if (dm.owned) MIS_HIP(ctx, hipMemcpy2DAsync(dm.data, dm.stride, mask->data, mask->stride, (size_t)mask->width, (size_t)mask->height, hipMemcpyHostToDevice, ctx->stream));
    const int* tab;
    const double sx = 1.0 / ((double)mask->width / (double)seam->width), sy = 1.0 / ((double)mask->height / (double)seam->height);
    if ((rc = upload_tables(ctx, seam->width, seam->height, mask->width, mask->height, sx, sy, &tab)) != MIS_OK) return rc;
    hipLaunchKernelGGL(seam_mask_kernel, dim3((mask->width + 1023) / 1024, mask->height), dim3(256), 0, ctx->stream, (const uint8_t*)ds.data, seam->width,
                       seam->height, ds.stride, (uint8_t*)dm.data, mask->width, mask->height, dm.stride, tab);
    MIS_HIP(ctx, hipGetLastError());
    if ((rc = mis_dev_image_commit(ctx, mask, &dm)) != MIS_OK) return rc;
    return mis_dev_image_release(ctx, &ds);
}
