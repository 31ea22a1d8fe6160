/*
 * scene.h -- synthetic panorama scene (test / bench infrastructure, not product code).
 * SURVEY.md section 8(d) "master panorama": a seeded, counter-hash-based procedural equirectangular
 * texture (value noise for smooth content + a jittered grid of rectangles and discs for
 * corner-rich content).  It is evaluated per texel on demand (never materialised); frames are
 * rendered by exact pinhole inverse mapping with bilinear sampling in double precision.
 * The same source is compiled by gcc (synth_cpu.c) and by hipcc (synth_gpu.hip).
 */
#ifndef SYNTH_SCENE_H
#define SYNTH_SCENE_H
#include <stdint.h>
#include <math.h>

#ifdef __HIPCC__
#define SY_HD __host__ __device__ static inline
#else
#define SY_HD static inline
#endif

#define SY_MW 32768 /* virtual master width  (360 deg) */
#define SY_MH 16384 /* virtual master height (180 deg) */
#define SY_SEED_LO 0x5717C4u
#define SY_SEED_HI 0x5EEDu
#define SY_CELL 96

typedef struct {
    int width, height;
    double f, cx, cy;
    double R[9];  /* camera-to-world rotation (world ray = R * K^-1 * p), as cv::detail::CameraParams::R */
    double gain;
} SyCamera;

SY_HD uint32_t sy_hash(uint32_t x, uint32_t y, uint32_t s) {
    uint32_t h = (x * 0x9E3779B1u) ^ (y * 0x85EBCA77u) ^ (s * 0xC2B2AE3Du) ^ SY_SEED_LO;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    h += SY_SEED_HI * 0x632BE5ABu;
    h ^= h >> 13; h *= 0x85EBCA6Bu; h ^= h >> 16;
    return h;
}

/* bilinear value noise, lattice spacing 2^k texels, result 0..255 (x wraps at the master width) */
SY_HD int sy_noise(int ix, int iy, int k, uint32_t salt) {
    int S = 1 << k, lx = ix >> k, ly = iy >> k, fx = ix & (S - 1), fy = iy & (S - 1);
    int nx = SY_MW >> k;
    int lx1 = (lx + 1) % nx;
    int a = (int)(sy_hash((uint32_t)lx, (uint32_t)ly, salt) & 255), b = (int)(sy_hash((uint32_t)lx1, (uint32_t)ly, salt) & 255);
    int c = (int)(sy_hash((uint32_t)lx, (uint32_t)(ly + 1), salt) & 255), d = (int)(sy_hash((uint32_t)lx1, (uint32_t)(ly + 1), salt) & 255);
    long long top = (long long)a * (S - fx) + (long long)b * fx, bot = (long long)c * (S - fx) + (long long)d * fx;
    return (int)((top * (S - fy) + bot * fy) >> (2 * k));
}

/* one master texel, BGR */
SY_HD void sy_texel(int ix, int iy, int* bgr) {
    ix = ((ix % SY_MW) + SY_MW) % SY_MW;
    if (iy < 0) iy = 0;
    if (iy > SY_MH - 1) iy = SY_MH - 1;
    for (int c = 0; c < 3; c++) {
        int v = sy_noise(ix, iy, 10, 100u + (uint32_t)c) * 4 + sy_noise(ix, iy, 7, 200u + (uint32_t)c) * 2 +
                sy_noise(ix, iy, 4, 300u) + sy_noise(ix, iy, 2, 400u);
        bgr[c] = 32 + (v * 3) / 32; /* 32 .. 223 */
    }
    /* shapes: each grid cell owns one rectangle and one disc; later cells / discs paint over */
    int cx0 = ix / SY_CELL, cy0 = iy / SY_CELL, ncx = SY_MW / SY_CELL;
    uint32_t best = 0;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            int gx = cx0 + dx, gy = cy0 + dy;
            if (gy < 0) continue;
            int wx = ((gx % ncx) + ncx) % ncx;
            for (int s = 0; s < 2; s++) {
                uint32_t h0 = sy_hash((uint32_t)wx, (uint32_t)gy, 1000u + (uint32_t)s);
                uint32_t h1 = sy_hash((uint32_t)wx, (uint32_t)gy, 2000u + (uint32_t)s);
                uint32_t prio = (h1 >> 8) | 1u;
                if (prio <= best) continue;
                int px = gx * SY_CELL + (int)(h0 % SY_CELL), py = gy * SY_CELL + (int)((h0 >> 8) % SY_CELL);
                int hit;
                if (s == 0) {
                    int hw = 4 + (int)((h0 >> 16) % 27), hh = 4 + (int)((h0 >> 24) % 27);
                    int ax = ix - px, ay = iy - py;
                    /* near the +-180 wrap gx is unwrapped, so compare in unwrapped texel units */
                    hit = ax >= -hw && ax <= hw && ay >= -hh && ay <= hh;
                } else {
                    int r = 4 + (int)((h0 >> 16) % 21);
                    int ax = ix - px, ay = iy - py;
                    hit = ax * ax + ay * ay <= r * r;
                }
                if (hit) {
                    best = prio;
                    bgr[0] = (int)(h1 & 255); bgr[1] = (int)((h1 >> 8) & 255); bgr[2] = (int)((h1 >> 16) & 255);
                }
            }
        }
}

/* one frame pixel: pinhole ray -> equirect master coordinate -> bilinear, u8 BGR */
SY_HD void sy_render_pixel(const SyCamera* cam, int x, int y, uint8_t* out) {
    double px = ((double)x - cam->cx) / cam->f, py = ((double)y - cam->cy) / cam->f, pz = 1.0;
    const double* R = cam->R;
    double X = R[0] * px + R[1] * py + R[2] * pz;
    double Y = R[3] * px + R[4] * py + R[5] * pz;
    double Z = R[6] * px + R[7] * py + R[8] * pz;
    double lon = atan2(X, Z), lat = asin(Y / sqrt(X * X + Y * Y + Z * Z));
    const double PI = 3.14159265358979323846;
    double mu = (lon / (2.0 * PI) + 0.5) * (double)SY_MW - 0.5;
    double mv = (lat / PI + 0.5) * (double)SY_MH - 0.5;
    double fu = floor(mu), fv = floor(mv);
    int iu = (int)fu, iv = (int)fv;
    double ax = mu - fu, ay = mv - fv;
    int t00[3], t01[3], t10[3], t11[3];
    sy_texel(iu, iv, t00); sy_texel(iu + 1, iv, t01); sy_texel(iu, iv + 1, t10); sy_texel(iu + 1, iv + 1, t11);
    for (int c = 0; c < 3; c++) {
        double v = ((double)t00[c] * (1.0 - ax) + (double)t01[c] * ax) * (1.0 - ay) +
                   ((double)t10[c] * (1.0 - ax) + (double)t11[c] * ax) * ay;
        v = v * cam->gain;
        int iv8 = (int)floor(v + 0.5);
        out[c] = (uint8_t)(iv8 < 0 ? 0 : (iv8 > 255 ? 255 : iv8));
    }
}
#endif
