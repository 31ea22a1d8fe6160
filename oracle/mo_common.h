/*
 * mo_common.h -- shared primitives of the CPU ORACLE (test infrastructure only).
 *
 * THIS DIRECTORY IS TEST INFRASTRUCTURE.  Nothing under oracle/ is linked, imported or
 * executed by the product library (image_stitching_amd/csrc, libmistitch.so).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, as the checker.
 *
 * PARITY UNPINNED: the arithmetic of the reference's hot path lives in OpenCV 4.x (vcpkg
 * opencv4[world], baseline 7bc5b8cd..., reference vcpkg.json:5-11) which is absent from the
 * reference tree and from this container.  This file restates the published OpenCV algorithms
 *  keyed to the reference call sites in
 * image_stitching/image_stitching.cpp:545,613,647,653,973-988,1117-1159,1164-1192,1218,1225.
 * Only the header-only rotation math (quaternion.h / euler.h) is pinned by known-answer
 * vectors captured from the reference headers.
 *
 * All float arithmetic here is written so that it means the same on the GPU: no FMA contraction
 * (build with -ffp-contract=off), only + - * / sqrt and explicit polynomials, round-half-even
 * conversions.
 */
#ifndef MO_COMMON_H
#define MO_COMMON_H

#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <float.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- OpenCV rounding helpers (core/fast_math.hpp): cvRound = round-half-even ---- */
static inline int mo_round_f(float v) { return (int)lrintf(v); }
static inline int mo_round_d(double v) { return (int)lrint(v); }
static inline int mo_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int mo_ceil_d(double v) { int i = (int)v; return i + (i < v); }
static inline int mo_floor_f(float v) { int i = (int)v; return i - ((float)i > v); }

static inline int mo_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline uint8_t mo_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline int16_t mo_sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* ---- border index maps (core/copy.cpp borderInterpolate) ---- */
/* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
static inline int mo_reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}
/* BORDER_REFLECT: fedcba|abcdefgh|hgfedcb */
static inline int mo_reflect(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p - 1;
        else p = 2 * len - 1 - p;
    }
    return p;
}

/* ---- cv::RNG (core.hpp): multiply-with-carry, CV_RNG_COEFF 4164903690 ---- */
typedef struct { uint64_t state; } MoRng;
static inline void mo_rng_init(MoRng* r, uint64_t seed) { r->state = seed ? seed : 0xffffffffu; }
static inline uint32_t mo_rng_next(MoRng* r) {
    r->state = (uint64_t)(uint32_t)r->state * 4164903690u + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}
static inline int mo_rng_uniform(MoRng* r, int a, int b) {
    return a == b ? a : (int)(mo_rng_next(r) % (uint32_t)(b - a) + (uint32_t)a);
}

/* ---- cv::fastAtan2 (core/mathfuncs_core): degrees, f32 polynomial ---- */
float mo_fast_atan2(float y, float x);

/* ---- deterministic f32 transcendentals (Cephes-style; identical op sequence on the GPU) ---- */
float mo_sinf(float x);
float mo_cosf(float x);
float mo_atan2f(float y, float x);
float mo_acosf(float x);
/* deterministic f64 natural log (only + * / and exponent extraction) used by the RANSAC
 * iteration-count update so the early exit is reproduced bit-for-bit on the GPU */
double mo_log_d(double x);

#define MO_PI_F 3.14159265358979323846f

#ifdef __cplusplus
}
#endif
#endif
