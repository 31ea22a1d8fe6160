/*
 * mo_orb.c -- ORACLE (test infrastructure): ORB detect+describe restatement.
 * Reference call sites: image_stitching/image_stitching.cpp:545 (ORB::create(4000,1.2,8,1,0,2,
 * HARRIS_SCORE,40,20)) and :613 (computeImageFeatures).  The arithmetic is OpenCV's
 * (features2d/src/orb.cpp, fast.cpp, imgproc color/resize/smooth) restated from SURVEY.md A.1-A.2.
 * PARITY UNPINNED (no OpenCV in the container).  Never linked into the product.
 */
#include "mo_orb.h"
#include <stdlib.h>
#include <string.h>

struct MoOrb {
    MoOrbParams p;
    int w, h, nlevels;
    int lw[MO_ORB_MAX_LEVELS], lh[MO_ORB_MAX_LEVELS];
    float lscale[MO_ORB_MAX_LEVELS];
    int nfeat[MO_ORB_MAX_LEVELS];
    uint8_t* gray[MO_ORB_MAX_LEVELS];
    uint8_t* pad[MO_ORB_MAX_LEVELS];
    uint8_t* blur[MO_ORB_MAX_LEVELS];
    uint8_t* score[MO_ORB_MAX_LEVELS];
    uint8_t* nms[MO_ORB_MAX_LEVELS];
    int cnt[MO_ORB_MAX_LEVELS][3];
    int umax[64];
    int8_t pattern[1024];
    MoKeyPoint* kps;
    uint8_t* desc;
    int n, cap;
};

void mo_orb_default_params(MoOrbParams* p) {
    p->nfeatures = 4000; p->scale_factor = 1.2f; p->nlevels = 8; p->edge_threshold = 1;
    p->first_level = 0; p->wta_k = 2; p->score_type = 0; p->patch_size = 40; p->fast_threshold = 20;
}

/* ---- cvtColor BGR2GRAY u8 (imgproc color_rgb.simd.hpp RGB2Gray<uchar>): the 15-bit coefficients BY15 3735, GY15 19235,
 * RY15 9798 with gray_shift 15 of OpenCV 4.x (SURVEY A.1 lists the older Q14 triple as the alternative; recalled, unpinned) ---- */
void mo_bgr2gray(const uint8_t* bgr, int w, int h, size_t stride, uint8_t* gray, size_t gstride) {
    int y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < h; y++) {
        const uint8_t* s = bgr + (size_t)y * stride;
        uint8_t* d = gray + (size_t)y * gstride;
        for (int x = 0; x < w; x++)
            d[x] = (uint8_t)((s[3 * x] * 3735 + s[3 * x + 1] * 19235 + s[3 * x + 2] * 9798 + (1 << 14)) >> 15);
    }
}

/* ---- resize INTER_LINEAR_EXACT u8: 8.8 fixed-point coefficients, no rounding between the
 * horizontal and the vertical pass, one round-to-nearest at the end ---- */
static void linear_exact_coeffs(int dlen, int slen, int* ofs, int* m1) {
    double inv = (double)dlen / (double)slen;
    double scale = 1.0 / inv;
    for (int i = 0; i < dlen; i++) {
        double v = ((double)i + 0.5) * scale - 0.5;
        int iv = mo_floor_d(v);
        if (iv < 0) { ofs[i] = 0; m1[i] = 0; }
        else if (iv >= slen - 1) { ofs[i] = slen - 1; m1[i] = 0; }
        else { ofs[i] = iv; m1[i] = mo_round_d((v - (double)iv) * 256.0); }
    }
}

void mo_resize_linear_exact_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn,
                               uint8_t* dst, int dw, int dh, size_t dstride) {
    int* xo = (int*)malloc(sizeof(int) * (size_t)dw * 2);
    int* yo = (int*)malloc(sizeof(int) * (size_t)dh * 2);
    int *xm = xo + dw, *ym = yo + dh;
    int y;
    linear_exact_coeffs(dw, sw, xo, xm);
    linear_exact_coeffs(dh, sh, yo, ym);
#pragma omp parallel for schedule(static)
    for (y = 0; y < dh; y++) {
        int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : y0;
        int my1 = ym[y], my0 = 256 - my1;
        const uint8_t* r0 = src + (size_t)y0 * sstride;
        const uint8_t* r1 = src + (size_t)y1 * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : x0;
            int mx1 = xm[x], mx0 = 256 - mx1;
            for (int c = 0; c < cn; c++) {
                unsigned h0 = (unsigned)r0[x0 * cn + c] * mx0 + (unsigned)r0[x1 * cn + c] * mx1;
                unsigned h1 = (unsigned)r1[x0 * cn + c] * mx0 + (unsigned)r1[x1 * cn + c] * mx1;
                d[x * cn + c] = (uint8_t)((h0 * my0 + h1 * my1 + (1u << 15)) >> 16);
            }
        }
    }
    free(xo); free(yo);
}

/* ---- Gaussian 7 taps sigma 2 in Q8 with error diffusion (getGaussianKernelFixedPoint_ED) ---- */
void mo_gauss7_kernel_q8(int k[7]) {
    double g[7], sum = 0, err = 0;
    int isum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; g[i] = exp(-0.125 * x * x); sum += g[i]; }
    for (int i = 0; i < 3; i++) {
        double adj = g[i] / sum * 256.0 + err;
        int v = mo_round_d(adj);
        err = adj - (double)v;
        k[i] = k[6 - i] = v;
        isum += v;
    }
    k[3] = 256 - 2 * isum;
}

/* ---- helpers ---- */
static void pad_reflect101(const uint8_t* g, int w, int h, uint8_t* p) {
    const int B = MO_ORB_BORDER;
    int pw = w + 2 * B, y;
#pragma omp parallel for schedule(static)
    for (y = -B; y < h + B; y++) {
        const uint8_t* s = g + (size_t)mo_reflect101(y, h) * w;
        uint8_t* d = p + (size_t)(y + B) * pw;
        for (int x = -B; x < w + B; x++) d[x + B] = s[mo_reflect101(x, w)];
    }
}

/* FAST-9/16 corner score (fast_score.cpp cornerScore<16>): max threshold for which the pixel is
 * still a corner = max(t, best dark arc, best bright arc) - 1; 0 when it is not a corner at t. */
static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_score_px(const uint8_t* p, int pw, int t) {
    int v = p[0], d[25], k;
    for (k = 0; k < 16; k++) d[k] = v - p[fast_dy[k] * pw + fast_dx[k]];
    for (k = 16; k < 25; k++) d[k] = d[k - 16];
    int a0 = t;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int q = 4; q <= 8; q++) if (d[k + q] < a) a = d[k + q];
        { int m = a < d[k] ? a : d[k]; if (m > a0) a0 = m; }
        { int m = a < d[k + 9] ? a : d[k + 9]; if (m > a0) a0 = m; }
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int q = 3; q <= 5; q++) if (d[k + q] > b) b = d[k + q];
        if (b >= b0) continue;
        for (int q = 6; q <= 8; q++) if (d[k + q] > b) b = d[k + q];
        { int m = b > d[k] ? b : d[k]; if (m < b0) b0 = m; }
        { int m = b > d[k + 9] ? b : d[k + 9]; if (m < b0) b0 = m; }
    }
    int s = -b0 - 1; /* = max(t, A, B) - 1; equals t-1 when no arc passes */
    return s >= t ? s : 0;
}

typedef struct { int x, y; float resp; } Cand;

static int cand_cmp(const void* a, const void* b) {
    const Cand* p = (const Cand*)a; const Cand* q = (const Cand*)b;
    if (p->resp > q->resp) return -1;
    if (p->resp < q->resp) return 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return 0;
}
static int float_desc_cmp(const void* a, const void* b) {
    float p = *(const float*)a, q = *(const float*)b;
    return p > q ? -1 : (p < q ? 1 : 0);
}

MoOrb* mo_orb_create(const MoOrbParams* p, int width, int height) {
    if (p->nlevels < 1 || p->nlevels > MO_ORB_MAX_LEVELS || p->first_level != 0 || p->wta_k != 2 ||
        p->patch_size < 2 || p->patch_size > 40 || width < 8 || height < 8)
        return NULL;
    MoOrb* o = (MoOrb*)calloc(1, sizeof(MoOrb));
    const int B = MO_ORB_BORDER;
    o->p = *p; o->w = width; o->h = height; o->nlevels = p->nlevels;
    double sf = (double)p->scale_factor;
    for (int l = 0; l < o->nlevels; l++) {
        float sc = (float)pow(sf, (double)l);
        o->lscale[l] = sc;
        o->lw[l] = mo_round_d((double)((float)width / sc));
        o->lh[l] = mo_round_d((double)((float)height / sc));
        size_t n = (size_t)o->lw[l] * o->lh[l], np = (size_t)(o->lw[l] + 2 * B) * (o->lh[l] + 2 * B);
        o->gray[l] = (uint8_t*)malloc(n);
        o->score[l] = (uint8_t*)malloc(n);
        o->nms[l] = (uint8_t*)malloc(n);
        o->pad[l] = (uint8_t*)malloc(np);
        o->blur[l] = (uint8_t*)malloc(np);
    }
    /* per-level budgets (orb.cpp detectAndCompute / computeKeyPoints) */
    {
        float factor = (float)(1.0 / sf);
        float nd = (float)p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)o->nlevels));
        int sum = 0;
        for (int l = 0; l < o->nlevels - 1; l++) {
            o->nfeat[l] = mo_round_f(nd);
            sum += o->nfeat[l];
            nd *= factor;
        }
        o->nfeat[o->nlevels - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    }
    /* umax (orb.cpp): quarter-disc extents of the intensity-centroid patch */
    {
        int hp = p->patch_size / 2, v, v0;
        int vmax = mo_floor_d((double)((float)hp * sqrtf(2.f) / 2 + 1));
        int vmin = mo_ceil_d((double)((float)hp * sqrtf(2.f) / 2));
        for (v = 0; v <= vmax; ++v) o->umax[v] = mo_round_d(sqrt((double)hp * hp - (double)v * v));
        for (v = hp, v0 = 0; v >= vmin; --v) {
            while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
            o->umax[v] = v0;
            ++v0;
        }
    }
    /* random BRIEF pattern: patchSize != 31 -> makeRandomPattern(patchSize, ., 512), RNG(0x34985739) */
    {
        MoRng r; mo_rng_init(&r, 0x34985739u);
        int hp = p->patch_size / 2;
        for (int i = 0; i < 512; i++) {
            o->pattern[2 * i] = (int8_t)mo_rng_uniform(&r, -hp, hp + 1);
            o->pattern[2 * i + 1] = (int8_t)mo_rng_uniform(&r, -hp, hp + 1);
        }
    }
    o->cap = p->nfeatures * 2 + 4096;
    o->kps = (MoKeyPoint*)malloc(sizeof(MoKeyPoint) * (size_t)o->cap);
    o->desc = (uint8_t*)malloc((size_t)o->cap * 32);
    return o;
}

void mo_orb_destroy(MoOrb* o) {
    if (!o) return;
    for (int l = 0; l < o->nlevels; l++) { free(o->gray[l]); free(o->score[l]); free(o->nms[l]); free(o->pad[l]); free(o->blur[l]); }
    free(o->kps); free(o->desc); free(o);
}

int mo_orb_run(MoOrb* o, const uint8_t* bgr, size_t stride) {
    const int B = MO_ORB_BORDER;
    const int t = o->p.fast_threshold;
    const int hp = o->p.patch_size / 2;
    int gk[7];
    mo_gauss7_kernel_q8(gk);
    o->n = 0;
    /* 1. gray + pyramid (each level resized from the previous one) */
    mo_bgr2gray(bgr, o->w, o->h, stride, o->gray[0], (size_t)o->w);
    for (int l = 1; l < o->nlevels; l++)
        mo_resize_linear_exact_u8(o->gray[l - 1], o->lw[l - 1], o->lh[l - 1], (size_t)o->lw[l - 1], 1,
                                  o->gray[l], o->lw[l], o->lh[l], (size_t)o->lw[l]);
    for (int l = 0; l < o->nlevels; l++) {
        const int w = o->lw[l], h = o->lh[l], pw = w + 2 * B;
        const uint8_t* P = o->pad[l];
        int y;
        pad_reflect101(o->gray[l], w, h, o->pad[l]);
        /* 2. FAST score map, then 3x3 strict non-max suppression (fast.cpp FAST_t<16>) */
        memset(o->score[l], 0, (size_t)w * h);
#pragma omp parallel for schedule(static)
        for (y = 3; y < h - 3; y++)
            for (int x = 3; x < w - 3; x++)
                o->score[l][(size_t)y * w + x] = (uint8_t)fast_score_px(P + (size_t)(y + B) * pw + x + B, pw, t);
        memset(o->nms[l], 0, (size_t)w * h);
        long nfast = 0;
        long hist[256];
        memset(hist, 0, sizeof(hist));
        for (y = 3; y < h - 3; y++) {
            const uint8_t* s = o->score[l] + (size_t)y * w;
            for (int x = 3; x < w - 3; x++) {
                int v = s[x];
                if (!v) continue;
                if (v > s[x - 1] && v > s[x + 1] && v > s[x - w - 1] && v > s[x - w] && v > s[x - w + 1] &&
                    v > s[x + w - 1] && v > s[x + w] && v > s[x + w + 1]) {
                    /* runByImageBorder(edgeThreshold): keep edge <= x < w-edge (no-op for edge <= 3) */
                    int e = o->p.edge_threshold;
                    if (x < e || x >= w - e || y < e || y >= h - e) continue;
                    o->nms[l][(size_t)y * w + x] = (uint8_t)v;
                    hist[v]++; nfast++;
                }
            }
        }
        o->cnt[l][0] = (int)nfast;
        /* 3. retainBest(2*N_l) on the FAST score: keep everything >= the (2N)-th best score */
        int N = o->nfeat[l], N2 = o->p.score_type == 0 ? 2 * N : N;
        int thr = 1;
        if (N2 == 0) { o->cnt[l][1] = o->cnt[l][2] = 0; goto blur_level; }
        if (nfast > N2) {
            long acc = 0;
            for (int v = 255; v >= 1; v--) { acc += hist[v]; if (acc >= N2) { thr = v; break; } }
        }
        {
            long nk = 0;
            for (int v = thr; v < 256; v++) nk += hist[v];
            Cand* c = (Cand*)malloc(sizeof(Cand) * (size_t)(nk > 0 ? nk : 1));
            long k = 0;
            for (y = 3; y < h - 3; y++)
                for (int x = 3; x < w - 3; x++) {
                    int v = o->nms[l][(size_t)y * w + x];
                    if (v >= thr && v > 0) { c[k].x = x; c[k].y = y; c[k].resp = (float)v; k++; }
                }
            o->cnt[l][1] = (int)k;
            /* 4. Harris response, block 7, k 0.04 (orb.cpp HarrisResponses) */
            if (o->p.score_type == 0) {
                const float scale = 1.f / ((1 << 2) * 7 * 255.f);
                const float scale_sq_sq = scale * scale * scale * scale;
                long i;
#pragma omp parallel for schedule(static)
                for (i = 0; i < k; i++) {
                    const uint8_t* p0 = P + (size_t)(c[i].y + B - 3) * pw + (c[i].x + B - 3);
                    int a = 0, b = 0, cc = 0;
                    for (int by = 0; by < 7; by++)
                        for (int bx = 0; bx < 7; bx++) {
                            const uint8_t* p = p0 + by * pw + bx;
                            int Ix = (p[1] - p[-1]) * 2 + (p[-pw + 1] - p[-pw - 1]) + (p[pw + 1] - p[pw - 1]);
                            int Iy = (p[pw] - p[-pw]) * 2 + (p[pw - 1] - p[-pw - 1]) + (p[pw + 1] - p[-pw + 1]);
                            a += Ix * Ix; b += Iy * Iy; cc += Ix * Iy;
                        }
                    float fa = (float)a, fb = (float)b, fc = (float)cc;
                    float s1 = fa * fb;
                    float s2 = fc * fc;
                    float s3 = 0.04f * (fa + fb);
                    s3 = s3 * (fa + fb);
                    c[i].resp = ((s1 - s2) - s3) * scale_sq_sq;
                }
                /* 5. retainBest(N_l) on the Harris response, ties at the cut retained */
                if (k > N) {
                    float* r = (float*)malloc(sizeof(float) * (size_t)k);
                    for (i = 0; i < k; i++) r[i] = c[i].resp;
                    qsort(r, (size_t)k, sizeof(float), float_desc_cmp);
                    float cut = r[N - 1];
                    free(r);
                    long m = 0;
                    for (i = 0; i < k; i++) if (c[i].resp >= cut) c[m++] = c[i];
                    k = m;
                }
            }
            /* canonical order: response desc, y, x (OpenCV's nth_element order is STL-defined) */
            qsort(c, (size_t)k, sizeof(Cand), cand_cmp);
            o->cnt[l][2] = (int)k;
            if (o->n + k > o->cap) { free(c); return -2; }
            /* 6. intensity-centroid angle on the un-blurred level (orb.cpp ICAngles) */
            for (long i = 0; i < k; i++) {
                const uint8_t* ctr = P + (size_t)(c[i].y + B) * pw + (c[i].x + B);
                int m01 = 0, m10 = 0;
                for (int u = -hp; u <= hp; ++u) m10 += u * ctr[u];
                for (int v = 1; v <= hp; ++v) {
                    int vsum = 0, d = o->umax[v];
                    for (int u = -d; u <= d; ++u) {
                        int vp = ctr[u + v * pw], vm = ctr[u - v * pw];
                        vsum += (vp - vm);
                        m10 += u * (vp + vm);
                    }
                    m01 += v * vsum;
                }
                MoKeyPoint* kp = &o->kps[o->n + i];
                kp->x = (float)c[i].x; kp->y = (float)c[i].y;
                kp->response = c[i].resp; kp->octave = l;
                kp->size = (float)o->p.patch_size * o->lscale[l];
                kp->angle = mo_fast_atan2((float)m01, (float)m10);
            }
            free(c);
        }
    blur_level:
        /* 7. GaussianBlur 7x7 sigma 2 in place on the level ROI of the bordered buffer: interior is
         * blurred (taps beyond the ROI read the REFLECT_101 border), the border itself stays un-blurred */
        memcpy(o->blur[l], o->pad[l], (size_t)pw * (h + 2 * B));
#pragma omp parallel for schedule(static)
        for (y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const uint8_t* p = P + (size_t)(y + B) * pw + (x + B);
                int acc = 0;
                for (int j = -3; j <= 3; j++) {
                    const uint8_t* r = p + j * pw;
                    int hs = 0;
                    for (int i = -3; i <= 3; i++) hs += gk[i + 3] * r[i];
                    acc += gk[j + 3] * hs;
                }
                o->blur[l][(size_t)(y + B) * pw + (x + B)] = (uint8_t)((acc + (1 << 15)) >> 16);
            }
        /* 8. rotated BRIEF, WTA_K = 2 (orb.cpp computeOrbDescriptors) */
        {
            int n0 = o->n, k = o->cnt[l][2];
            float sc = o->lscale[l], inv = 1.f / sc;
            long i;
#pragma omp parallel for schedule(static)
            for (i = 0; i < k; i++) {
                MoKeyPoint* kp = &o->kps[n0 + i];
                /* keypoint goes to level-0 coordinates first, descriptors scale it back */
                float X = l ? kp->x * sc : kp->x, Y = l ? kp->y * sc : kp->y;
                int cx = mo_round_f(X * inv), cy = mo_round_f(Y * inv);
                float ang = kp->angle * (float)(3.14159265358979323846 / 180.f);
                float a = mo_cosf(ang), b = mo_sinf(ang);
                const uint8_t* ctr = o->blur[l] + (size_t)(cy + B) * pw + (cx + B);
                uint8_t* dsc = o->desc + (size_t)(n0 + i) * 32;
                const int8_t* pat = o->pattern;
                for (int by = 0; by < 32; by++) {
                    int val = 0;
                    for (int bit = 0; bit < 8; bit++) {
                        int i0 = (by * 8 + bit) * 2, i1 = i0 + 1;
                        float x0 = (float)pat[2 * i0], y0 = (float)pat[2 * i0 + 1];
                        float x1 = (float)pat[2 * i1], y1 = (float)pat[2 * i1 + 1];
                        int ix0 = mo_round_f(x0 * a - y0 * b), iy0 = mo_round_f(x0 * b + y0 * a);
                        int ix1 = mo_round_f(x1 * a - y1 * b), iy1 = mo_round_f(x1 * b + y1 * a);
                        int t0 = ctr[iy0 * pw + ix0], t1 = ctr[iy1 * pw + ix1];
                        val |= (t0 < t1) << bit;
                    }
                    dsc[by] = (uint8_t)val;
                }
                kp->x = X; kp->y = Y;
            }
            o->n += k;
        }
    }
    return o->n;
}

int mo_orb_num_keypoints(const MoOrb* o) { return o->n; }
const MoKeyPoint* mo_orb_keypoints(const MoOrb* o) { return o->kps; }
const uint8_t* mo_orb_descriptors(const MoOrb* o) { return o->desc; }
int mo_orb_level_width(const MoOrb* o, int l) { return o->lw[l]; }
int mo_orb_level_height(const MoOrb* o, int l) { return o->lh[l]; }
float mo_orb_level_scale(const MoOrb* o, int l) { return o->lscale[l]; }
int mo_orb_level_nfeatures(const MoOrb* o, int l) { return o->nfeat[l]; }
const uint8_t* mo_orb_level_gray(const MoOrb* o, int l) { return o->gray[l]; }
const uint8_t* mo_orb_level_nms(const MoOrb* o, int l) { return o->nms[l]; }
const uint8_t* mo_orb_level_blur(const MoOrb* o, int l) { return o->blur[l]; }
int mo_orb_level_count(const MoOrb* o, int l, int which) { return o->cnt[l][which]; }
const int8_t* mo_orb_pattern(const MoOrb* o) { return o->pattern; }
const int* mo_orb_umax(const MoOrb* o) { return o->umax; }
