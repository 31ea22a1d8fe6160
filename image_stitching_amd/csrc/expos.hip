// expos.hip -- exposure compensation and the simple seam finders of the step between warp and blend:
//   ExposureCompensator::createDefault(GAIN_BLOCKS) + setNrFeeds(1) / setNrGainsFilteringIterations(2) / setBlockSize(64, 64),
//   feed(corners, images_warped, masks_warped)                    image_stitching/image_stitching.cpp:1002-1023
//   compensator->apply(img_idx, corners[img_idx], img_warped, mask_warped)                       image_stitching.cpp:1162
//   SeamFinder "no" / "voronoi" (find(images_warped_f, corners, masks_warped))                   image_stitching.cpp:1029-1065
// The reference's default seam finder, DpSeamFinder(COLOR) ("dp_color", :1040), is NOT built.
//
// Division of labour.  The overlap statistics (one wave per pair of overlapping 64x64 blocks: pixel count and the two
// sums of BGR norms, accumulated in the reference's row-major order so that the doubles agree bit for bit), the
// per-pixel gain application and the distance transforms of the Voronoi finder run on the device; the normal
// equations of the gains (a dense LU solve with partial pivoting over a few hundred blocks) and the 3-tap
// smoothing of the tiny gain maps are host work, as in the reference.
#include "common.h"
#include "dev_math.h"
#include <float.h>
#include <math.h>
#include <algorithm>
#include <vector>

namespace {

struct ImgDesc {
    const uint8_t* img; size_t istride;
    const uint8_t* msk; size_t mstride;
    int cx, cy, w, h;
};
struct PairDesc { int a, b, x0, y0, x1, y1; };   // images a, b and the pano rectangle the two blocks share
struct PairStat { double s1, s2; int cnt, pad; };

// GainCompensator::singleFeed inner loop: intersect = both masks 255; N = count; Isum += norm(BGR).
// One wave per pair.  The 64 lanes fetch 64 consecutive pixels of the row-major scan, take the (correctly rounded)
// double square roots in parallel and then the wave adds them in lane order, which is the order of the scalar loop
// (a pixel outside the intersection contributes +0.0, which leaves a non-negative sum unchanged).
__global__ __launch_bounds__(256) void overlap_stats_kernel(const ImgDesc* __restrict__ imgs, const PairDesc* __restrict__ pairs, int npairs,
                                                            PairStat* __restrict__ out) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= npairs) return;
    const PairDesc pd = pairs[p];
    const ImgDesc A = imgs[pd.a], B = imgs[pd.b];
    const int rw = pd.x1 - pd.x0, total = rw * (pd.y1 - pd.y0);
    double s1 = 0, s2 = 0;
    int cnt = 0;
    for (int base = 0; base < total; base += 64) {
        const int q = base + lane;
        double v1 = 0, v2 = 0;
        bool ok = false;
        if (q < total) {
            const int y = pd.y0 + q / rw, x = pd.x0 + q % rw;
            const size_t ya = (size_t)(y - A.cy), xa = (size_t)(x - A.cx), yb = (size_t)(y - B.cy), xb = (size_t)(x - B.cx);
            ok = A.msk[ya * A.mstride + xa] == 255 && B.msk[yb * B.mstride + xb] == 255;
            if (ok) {
                const uint8_t* u = A.img + ya * A.istride + 3 * xa;
                const uint8_t* v = B.img + yb * B.istride + 3 * xb;
                v1 = sqrt((double)(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
                v2 = sqrt((double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
            }
        }
        cnt += __popcll(__ballot(ok));
        for (int l = 0; l < 64; l++) {
            s1 += __shfl(v1, l);
            s2 += __shfl(v2, l);
        }
    }
    if (lane == 0) { out[p].s1 = s1; out[p].s2 = s2; out[p].cnt = cnt; out[p].pad = 0; }
}

// BlocksCompensator::apply: gain map -> resize(INTER_LINEAR, float) -> multiply(image, gains, image): one thread per pixel
template <typename T>
__global__ __launch_bounds__(256) void gain_apply_kernel(T* __restrict__ img, size_t stride_elems, int w, int h, const float* __restrict__ map, int mw, int mh,
                                                         double sx, double sy) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    float fy = (float)(((double)y + 0.5) * sy - 0.5);
    int iy = mis_floor_f(fy);
    fy -= (float)iy;
    if (iy < 0) { iy = 0; fy = 0.f; }
    if (iy >= mh - 1) { iy = mh - 1; fy = 0.f; }
    const int iy1 = iy + 1 < mh ? iy + 1 : iy;
    float fx = (float)(((double)x + 0.5) * sx - 0.5);
    int ix = mis_floor_f(fx);
    fx -= (float)ix;
    if (ix < 0) { ix = 0; fx = 0.f; }
    if (ix >= mw - 1) { ix = mw - 1; fx = 0.f; }
    const int ix1 = ix + 1 < mw ? ix + 1 : ix;
    const float h0 = map[iy * mw + ix] * (1.f - fx) + map[iy * mw + ix1] * fx;
    const float h1 = map[iy1 * mw + ix] * (1.f - fx) + map[iy1 * mw + ix1] * fx;
    const float g = h0 * (1.f - fy) + h1 * fy;
    T* p = img + (size_t)y * stride_elems + 3 * (size_t)x;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int v = mis_round_f((float)p[k] * g);
        p[k] = (T)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// core/src/lapack.cpp LUImpl<double> as cv::solve(A, b, x, DECOMP_LU) drives it: partial pivoting, eps = 100 * DBL_EPSILON
bool solve_lu(std::vector<double>& A, std::vector<double>& b, int m) {
    const double eps = DBL_EPSILON * 100;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++) if (fabs(A[(size_t)j * m + i]) > fabs(A[(size_t)k * m + i])) k = j;
        if (fabs(A[(size_t)k * m + i]) < eps) return false;
        if (k != i) {
            for (int j = i; j < m; j++) std::swap(A[(size_t)i * m + j], A[(size_t)k * m + j]);
            std::swap(b[i], b[k]);
        }
        const double d = -1 / A[(size_t)i * m + i];
        const double* ri = &A[(size_t)i * m];
        for (int j = i + 1; j < m; j++) {
            double* rj = &A[(size_t)j * m];
            const double alpha = rj[i] * d;
            for (int c = i + 1; c < m; c++) rj[c] += alpha * ri[c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < m; k++) s -= A[(size_t)i * m + k] * b[k];
        b[i] = s / A[(size_t)i * m + i];
    }
    return true;
}

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

struct Block { int x, y, w, h, img; };

// ---- Voronoi seam finder ----
// VoronoiSeamFinder::findInPair: cut both masks over the shared rectangle grown by `gap`; unique = mask minus the collision
__global__ __launch_bounds__(256) void vor_cut_kernel(ImgDesc A, ImgDesc B, int x0, int y0, int W, int H, int gap, uint8_t* __restrict__ u1, uint8_t* __restrict__ u2) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int px = x0 - gap + x, py = y0 - gap + y;
    const int xa = px - A.cx, ya = py - A.cy, xb = px - B.cx, yb = py - B.cy;
    const uint8_t m1 = (xa >= 0 && ya >= 0 && xa < A.w && ya < A.h) ? A.msk[(size_t)ya * A.mstride + xa] : 0;
    const uint8_t m2 = (xb >= 0 && yb >= 0 && xb < B.w && yb < B.h) ? B.msk[(size_t)yb * B.mstride + xb] : 0;
    const bool collision = m1 && m2;
    // the transform below measures the distance to the nearest ZERO byte: zero where the unique mask is set
    u1[(size_t)y * W + x] = (m1 && !collision) ? 0 : 1;
    u2[(size_t)y * W + x] = (m2 && !collision) ? 0 : 1;
}
// exact L1 distance to the nearest zero byte: rows (both directions), then columns; one thread per line, both maps at once
constexpr int VINF = 1 << 29;
__global__ void vor_rows_kernel(const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, int W, int H, int* __restrict__ d1, int* __restrict__ d2) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * H) return;
    const uint8_t* u = (t < H ? u1 : u2) + (size_t)(t % H) * W;
    int* d = (t < H ? d1 : d2) + (size_t)(t % H) * W;
    int run = VINF;
    for (int x = 0; x < W; x++) { run = u[x] ? min(run + 1, VINF) : 0; d[x] = run; }
    run = VINF;
    for (int x = W - 1; x >= 0; x--) { run = u[x] ? min(run + 1, VINF) : 0; d[x] = min(d[x], run); }
}
__global__ void vor_cols_kernel(int W, int H, int* __restrict__ d1, int* __restrict__ d2) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * W) return;
    int* d = (t < W ? d1 : d2) + (t % W);
    int run = VINF;
    for (int y = 0; y < H; y++) { run = min(min(run + 1, VINF), d[(size_t)y * W]); d[(size_t)y * W] = run; }
    run = VINF;
    for (int y = H - 1; y >= 0; y--) { run = min(min(run + 1, VINF), d[(size_t)y * W]); d[(size_t)y * W] = run; }
}
// seam = dist1 < dist2: the second image loses the pixel, otherwise the first does
__global__ __launch_bounds__(256) void vor_apply_kernel(ImgDesc A, ImgDesc B, uint8_t* __restrict__ ma, uint8_t* __restrict__ mb, int x0, int y0, int rw, int rh, int W,
                                                        int gap, const int* __restrict__ d1, const int* __restrict__ d2) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= rw) return;
    const size_t k = (size_t)(y + gap) * W + x + gap;
    if (d1[k] < d2[k]) mb[(size_t)(y0 - B.cy + y) * B.mstride + (x0 - B.cx + x)] = 0;
    else ma[(size_t)(y0 - A.cy + y) * A.mstride + (x0 - A.cx + x)] = 0;
}

}  // namespace

struct MisCompensator {
    MisContext* ctx = nullptr;
    int bw = 64, bh = 64, nfilt = 2;
    int n = 0;
    std::vector<std::vector<float>> maps;
    std::vector<int> mw, mh;
    float* dev_maps = nullptr;         // all maps back to back
    size_t dev_maps_cap = 0;           // floats allocated
    std::vector<size_t> dev_ofs;
};

extern "C" int mis_compensator_create(MisContext* ctx, int block_w, int block_h, int nr_filtering, MisCompensator** out) {
    if (!ctx || !out) return MIS_E_INVALID;
    MIS_CHECK(ctx, block_w > 0 && block_h > 0 && nr_filtering >= 0, MIS_E_INVALID, "compensator: block %dx%d, %d filtering passes", block_w, block_h, nr_filtering);
    MisCompensator* c = new MisCompensator();
    c->ctx = ctx; c->bw = block_w; c->bh = block_h; c->nfilt = nr_filtering;
    *out = c;
    return MIS_OK;
}

extern "C" int mis_compensator_destroy(MisCompensator* c) {
    if (!c) return MIS_OK;
    if (c->dev_maps) { hipSetDevice(c->ctx->device); hipStreamSynchronize(c->ctx->stream); hipFree(c->dev_maps); }
    delete c;
    return MIS_OK;
}

extern "C" int mis_compensator_feed(MisCompensator* c, const MisPoint* corners, const MisImage* images, const MisImage* masks, int n) {
    if (!c) return MIS_E_INVALID;
    MisContext* ctx = c->ctx;
    MIS_CHECK(ctx, corners && images && masks && n > 0, MIS_E_INVALID, "compensator feed: null argument or no images");
    for (int i = 0; i < n; i++) {
        MIS_CHECK(ctx, images[i].data && images[i].dtype == MIS_U8 && images[i].channels == 3, MIS_E_UNSUPPORTED, "compensator feed: image %d is not 8UC3", i);
        MIS_CHECK(ctx, masks[i].data && masks[i].dtype == MIS_U8 && masks[i].channels == 1 && masks[i].width == images[i].width && masks[i].height == images[i].height,
                  MIS_E_INVALID, "compensator feed: mask %d does not match its image", i);
    }
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    // the blocks of every image become the "images" of one GainCompensator (BlocksCompensator::feed)
    c->n = n; c->mw.assign(n, 0); c->mh.assign(n, 0); c->maps.assign(n, {});
    std::vector<Block> B;
    for (int i = 0; i < n; i++) {
        const int W = images[i].width, H = images[i].height;
        c->mw[i] = (W + c->bw - 1) / c->bw; c->mh[i] = (H + c->bh - 1) / c->bh;
        const int bw = (W + c->mw[i] - 1) / c->mw[i], bh = (H + c->mh[i] - 1) / c->mh[i];
        for (int by = 0; by < c->mh[i]; by++)
            for (int bx = 0; bx < c->mw[i]; bx++) {
                const int ox = bx * bw, oy = by * bh;
                B.push_back({corners[i].x + ox, corners[i].y + oy, std::min(ox + bw, W) - ox, std::min(oy + bh, H) - oy, i});
            }
    }
    const int nb = (int)B.size();
    std::vector<PairDesc> pairs;
    std::vector<std::pair<int, int>> pair_ij;
    for (int i = 0; i < nb; i++)
        for (int j = i; j < nb; j++) {
            const int x0 = std::max(B[i].x, B[j].x), y0 = std::max(B[i].y, B[j].y);
            const int x1 = std::min(B[i].x + B[i].w, B[j].x + B[j].w), y1 = std::min(B[i].y + B[i].h, B[j].y + B[j].h);
            if (x0 < x1 && y0 < y1) { pairs.push_back({B[i].img, B[j].img, x0, y0, x1, y1}); pair_ij.emplace_back(i, j); }
        }
    const int np = (int)pairs.size();

    std::vector<DevImage> di(n), dm(n);
    std::vector<ImgDesc> desc(n);
    int rc = MIS_OK;
    for (int i = 0; i < n && rc == MIS_OK; i++) {
        if ((rc = mis_dev_image_in(ctx, &images[i], &di[i])) != MIS_OK) break;
        if ((rc = mis_dev_image_in(ctx, &masks[i], &dm[i])) != MIS_OK) break;
        desc[i] = {(const uint8_t*)di[i].data, di[i].stride, (const uint8_t*)dm[i].data, dm[i].stride, corners[i].x, corners[i].y, images[i].width, images[i].height};
    }
    std::vector<PairStat> stats(np);
    if (rc == MIS_OK) {
        const size_t b_desc = mis_align_up(sizeof(ImgDesc) * n, 256), b_pairs = mis_align_up(sizeof(PairDesc) * np, 256), b_stats = sizeof(PairStat) * np;
        void* buf = nullptr; size_t got = 0;
        if ((rc = mis_pool_alloc(ctx, b_desc + b_pairs + b_stats, &buf, &got)) == MIS_OK) {
            char* p = (char*)buf;
            hipError_t e = hipMemcpyAsync(p, desc.data(), sizeof(ImgDesc) * n, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(p + b_desc, pairs.data(), sizeof(PairDesc) * np, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(overlap_stats_kernel, dim3((np + 3) / 4), dim3(256), 0, ctx->stream, (const ImgDesc*)p, (const PairDesc*)(p + b_desc), np,
                                   (PairStat*)(p + b_desc + b_pairs));
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(stats.data(), p + b_desc + b_pairs, b_stats, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            mis_pool_free(ctx, buf, got);
            if (e != hipSuccess) rc = mis_set_error(ctx, MIS_E_HIP, "compensator feed: %s", hipGetErrorString(e));
        }
    }
    for (int i = 0; i < n; i++) { mis_dev_image_release(ctx, &di[i]); mis_dev_image_release(ctx, &dm[i]); }
    if (rc != MIS_OK) return rc;

    // GainCompensator::singleFeed: N, I, then the normal equations over the blocks that meet another block
    std::vector<int> N((size_t)nb * nb, 0);
    std::vector<double> I((size_t)nb * nb, 0.0);
    std::vector<char> skip(nb, 1);
    for (int p = 0; p < np; p++) {
        const int i = pair_ij[p].first, j = pair_ij[p].second, cnt = std::max(1, stats[p].cnt);
        N[(size_t)i * nb + j] = N[(size_t)j * nb + i] = cnt;
        if (i != j) skip[i] = skip[j] = 0;
        I[(size_t)i * nb + j] = stats[p].s1 / cnt;
        I[(size_t)j * nb + i] = stats[p].s2 / cnt;
    }
    std::vector<double> gains(nb, 1.0);
    int neq = 0;
    for (int i = 0; i < nb; i++) neq += !skip[i];
    if (neq > 0) {
        const double alpha = 0.01, beta = 100;
        std::vector<double> A((size_t)neq * neq, 0.0), b(neq, 0.0);
        for (int i = 0, ki = 0; i < nb; i++) {
            if (skip[i]) continue;
            for (int j = 0, kj = 0; j < nb; j++) {
                if (skip[j]) continue;
                const double nij = N[(size_t)i * nb + j], iij = I[(size_t)i * nb + j], iji = I[(size_t)j * nb + i];
                b[ki] += beta * nij;
                A[(size_t)ki * neq + ki] += beta * nij;
                if (j != i) {
                    A[(size_t)ki * neq + ki] += 2 * alpha * iij * iij * nij;
                    A[(size_t)ki * neq + kj] -= 2 * alpha * iij * iji * nij;
                }
                kj++;
            }
            ki++;
        }
        if (solve_lu(A, b, neq))
            for (int i = 0, j = 0; i < nb; i++) if (!skip[i]) gains[i] = b[j++];
    }
    // one gain per block -> a small float map per image, smoothed by the separable [1 2 1] / 4 (BORDER_REFLECT_101)
    size_t total = 0;
    c->dev_ofs.assign(n, 0);
    for (int i = 0, q = 0; i < n; i++) {
        const int mw = c->mw[i], mh = c->mh[i];
        std::vector<float> m((size_t)mw * mh), t((size_t)mw * mh);
        for (int k = 0; k < mw * mh; k++) m[k] = (float)gains[q++];
        for (int it = 0; it < c->nfilt; it++) {
            for (int y = 0; y < mh; y++)
                for (int x = 0; x < mw; x++) t[y * mw + x] = (m[y * mw + reflect101(x - 1, mw)] + m[y * mw + reflect101(x + 1, mw)]) * 0.25f + m[y * mw + x] * 0.5f;
            for (int y = 0; y < mh; y++)
                for (int x = 0; x < mw; x++) m[y * mw + x] = (t[reflect101(y - 1, mh) * mw + x] + t[reflect101(y + 1, mh) * mw + x]) * 0.25f + t[y * mw + x] * 0.5f;
        }
        c->dev_ofs[i] = total;
        total += m.size();
        c->maps[i] = std::move(m);
    }
    // the device copy of the maps is grow-only (a free + malloc per feed synchronises the device twice) and filled by ONE copy
    // from the context's pinned staging
    if (c->dev_maps_cap < total) {
        if (c->dev_maps) { MIS_HIP(ctx, hipStreamSynchronize(ctx->stream)); MIS_HIP(ctx, hipFree(c->dev_maps)); c->dev_maps = nullptr; c->dev_maps_cap = 0; }
        MIS_HIP(ctx, hipMalloc(&c->dev_maps, (total + total / 2) * sizeof(float)));
        c->dev_maps_cap = total + total / 2;
    }
    void* hs = nullptr;
    int rcs = mis_host_stage(ctx, total * sizeof(float), &hs);
    if (rcs != MIS_OK) return rcs;
    for (int i = 0; i < n; i++) memcpy((float*)hs + c->dev_ofs[i], c->maps[i].data(), c->maps[i].size() * sizeof(float));
    MIS_HIP(ctx, hipMemcpyAsync(c->dev_maps, hs, total * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the staging buffer is reusable when this returns
    return MIS_OK;
}

extern "C" int mis_compensator_gain_map(const MisCompensator* c, int index, float* map_host, int capacity, int* blocks_x, int* blocks_y) {
    if (!c) return MIS_E_INVALID;
    MIS_CHECK(c->ctx, index >= 0 && index < c->n, MIS_E_INVALID, "compensator: image index %d out of range (fed %d)", index, c->n);
    if (blocks_x) *blocks_x = c->mw[index];
    if (blocks_y) *blocks_y = c->mh[index];
    if (map_host) {
        MIS_CHECK(c->ctx, capacity >= (int)c->maps[index].size(), MIS_E_INVALID, "compensator: gain map needs %zu floats", c->maps[index].size());
        memcpy(map_host, c->maps[index].data(), c->maps[index].size() * sizeof(float));
    }
    return MIS_OK;
}

extern "C" int mis_compensator_apply(MisCompensator* c, int index, MisImage* image) {
    if (!c) return MIS_E_INVALID;
    MisContext* ctx = c->ctx;
    MIS_CHECK(ctx, index >= 0 && index < c->n, MIS_E_INVALID, "compensator: image index %d out of range (fed %d)", index, c->n);
    MIS_CHECK(ctx, image && image->data && image->channels == 3 && (image->dtype == MIS_U8 || image->dtype == MIS_S16), MIS_E_UNSUPPORTED,
              "compensator apply: 8UC3 or 16SC3 (values 0..255) only");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    DevImage d;
    int rc;
    if ((rc = mis_dev_image_in(ctx, image, &d)) != MIS_OK) return rc;
    const int w = image->width, h = image->height, mw = c->mw[index], mh = c->mh[index];
    // resize(): inv_scale = dsize / ssize, scale = 1 / inv_scale
    const double sx = 1.0 / ((double)w / (double)mw), sy = 1.0 / ((double)h / (double)mh);
    dim3 grid((w + 255) / 256, h), block(256);
    if (image->dtype == MIS_U8)
        hipLaunchKernelGGL(gain_apply_kernel<uint8_t>, grid, block, 0, ctx->stream, (uint8_t*)d.data, d.stride, w, h, c->dev_maps + c->dev_ofs[index], mw, mh, sx, sy);
    else
        hipLaunchKernelGGL(gain_apply_kernel<int16_t>, grid, block, 0, ctx->stream, (int16_t*)d.data, d.stride / 2, w, h, c->dev_maps + c->dev_ofs[index], mw, mh, sx, sy);
    MIS_HIP(ctx, hipGetLastError());
    if (d.owned) {
        const size_t row = (size_t)w * 3 * mis_dtype_size(image->dtype);
        MIS_HIP(ctx, hipMemcpy2DAsync(image->data, image->stride, d.data, d.stride, row, h, hipMemcpyDeviceToHost, ctx->stream));
    }
    return mis_dev_image_release(ctx, &d);
}

extern "C" int mis_seam_voronoi(MisContext* ctx, const MisPoint* corners, MisImage* masks, int n) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, corners && masks && n > 0, MIS_E_INVALID, "voronoi seams: null argument or no masks");
    for (int i = 0; i < n; i++)
        MIS_CHECK(ctx, masks[i].data && masks[i].dtype == MIS_U8 && masks[i].channels == 1, MIS_E_UNSUPPORTED, "voronoi seams: mask %d is not 8UC1", i);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<DevImage> dm(n);
    std::vector<ImgDesc> desc(n);
    int rc = MIS_OK;
    for (int i = 0; i < n; i++) {
        if ((rc = mis_dev_image_in(ctx, &masks[i], &dm[i])) != MIS_OK) return rc;
        desc[i] = {nullptr, 0, (const uint8_t*)dm[i].data, dm[i].stride, corners[i].x, corners[i].y, masks[i].width, masks[i].height};
    }
    const int gap = 10;
    // PairwiseSeamFinder::run: every overlapping pair in order; a mask edited by one pair is the input of the next
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            const int x0 = std::max(corners[i].x, corners[j].x), y0 = std::max(corners[i].y, corners[j].y);
            const int x1 = std::min(corners[i].x + masks[i].width, corners[j].x + masks[j].width);
            const int y1 = std::min(corners[i].y + masks[i].height, corners[j].y + masks[j].height);
            if (!(x0 < x1 && y0 < y1)) continue;
            const int rw = x1 - x0, rh = y1 - y0, W = rw + 2 * gap, H = rh + 2 * gap;
            const size_t px = (size_t)W * H, bytes = mis_align_up(px, 256) * 2 + px * 2 * sizeof(int);
            void* buf = nullptr; size_t got = 0;
            if ((rc = mis_pool_alloc(ctx, bytes, &buf, &got)) != MIS_OK) return rc;
            uint8_t* u1 = (uint8_t*)buf; uint8_t* u2 = u1 + mis_align_up(px, 256);
            int* d1 = (int*)(u2 + mis_align_up(px, 256)); int* d2 = d1 + px;
            hipLaunchKernelGGL(vor_cut_kernel, dim3((W + 255) / 256, H), dim3(256), 0, ctx->stream, desc[i], desc[j], x0, y0, W, H, gap, u1, u2);
            hipLaunchKernelGGL(vor_rows_kernel, dim3((2 * H + 63) / 64), dim3(64), 0, ctx->stream, u1, u2, W, H, d1, d2);
            hipLaunchKernelGGL(vor_cols_kernel, dim3((2 * W + 63) / 64), dim3(64), 0, ctx->stream, W, H, d1, d2);
            hipLaunchKernelGGL(vor_apply_kernel, dim3((rw + 255) / 256, rh), dim3(256), 0, ctx->stream, desc[i], desc[j], (uint8_t*)dm[i].data, (uint8_t*)dm[j].data, x0, y0,
                               rw, rh, W, gap, d1, d2);
            mis_pool_free(ctx, buf, got);
            MIS_HIP(ctx, hipGetLastError());
        }
    for (int i = 0; i < n; i++) {
        if (dm[i].owned)
            MIS_HIP(ctx, hipMemcpy2DAsync(masks[i].data, masks[i].stride, dm[i].data, dm[i].stride, (size_t)masks[i].width, masks[i].height, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = mis_dev_image_release(ctx, &dm[i])) != MIS_OK) return rc;
    }
    return MIS_OK;
}
