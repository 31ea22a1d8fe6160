// dev_math.h -- deterministic f32/f64 device math for the stitching kernels.
//
// The stitched result must match the CPU path of the reference bit for bit wherever integers are
// produced (keypoint indices, inlier masks, quantised remap coordinates), so every transcendental
// is an explicit Cephes-style polynomial evaluated with plain IEEE + - * / (the library is built
// with -ffp-contract=off: no FMA contraction; HIP's f32 division and sqrt are correctly rounded).
// These functions are usable from host code of this library too (mis_warp_roi runs on the host).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#define MIS_HD __host__ __device__ __forceinline__
#define MIS_PI_F 3.14159265358979323846f

// A workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and on gfx9 that fence
// is `s_waitcnt vmcnt(0) lgkmcnt(0)`: every global load the wave has in flight -- a software prefetch of the next tile -- is waited
// for at the barrier, and every global store has to drain.  Kernels whose barriers only hand LDS data between waves (the global
// buffers are read-only or written once, by their owner) use this one and keep their loads in flight across it; the compiler still
// waits for a loaded register where it is first used.
__device__ __forceinline__ void mis_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// cvRound: round half to even
MIS_HD int mis_round_f(float v) {
#ifdef __HIP_DEVICE_COMPILE__
    return __float2int_rn(v);
#else
    return (int)lrintf(v);
#endif
}
MIS_HD int mis_round_d(double v) {
#ifdef __HIP_DEVICE_COMPILE__
    return __double2int_rn(v);
#else
    return (int)lrint(v);
#endif
}
// x86 cvRound semantics for out-of-range / NaN inputs (cvtss2si gives INT_MIN)
MIS_HD int mis_round_sat_f(float v) {
    if (!(fabsf(v) < 2147483648.f)) return (int)0x80000000;
    return mis_round_f(v);
}
MIS_HD int mis_sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// BORDER_REFLECT_101 and BORDER_REFLECT index maps (closed forms of cv::borderInterpolate)
MIS_HD int mis_reflect101(int p, int len) {
    if (len == 1) return 0;
    int period = 2 * len - 2;
    int q = p % period;
    if (q < 0) q += period;
    return q < len ? q : period - q;
}
MIS_HD int mis_reflect(int p, int len) {
    if (len == 1) return 0;
    int period = 2 * len;
    int q = p % period;
    if (q < 0) q += period;
    return q < len ? q : period - 1 - q;
}

// BORDER_REFLECT for p in [-len, 2 len): one fold, branch free (caller guarantees the range)
MIS_HD int mis_reflect1(int p, int len) {
    int r = p ^ (p >> 31);  // p < 0 -> -p - 1
    return r >= len ? 2 * len - 1 - r : r;
}

MIS_HD float mis_sin_poly(float x, float z) {
    float y = -1.9515295891E-4f * z + 8.3321608736E-3f;
    y = y * z - 1.6666654611E-1f;
    y = y * z;
    y = y * x;
    return y + x;
}
MIS_HD float mis_cos_poly(float z) {
    float y = 2.443315711809948E-005f * z - 1.388731625493765E-003f;
    y = y * z + 4.166664568298827E-002f;
    y = y * z;
    y = y * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

// sin and cos of one argument share the octant reduction (|x| < 8192)
MIS_HD void mis_sincosf(float xx, float* s, float* c) {
    float x = fabsf(xx);
    unsigned j = (unsigned)(1.27323954473516f * x);
    float y = (float)j;
    if (j & 1u) { j += 1; y += 1.0f; }
    j &= 7u;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = x * x;
    float ps = mis_sin_poly(x, z), pc = mis_cos_poly(z);
    int ssign = xx < 0 ? -1 : 1, csign = 1;
    if (j > 3) { ssign = -ssign; csign = -csign; j -= 4; }
    if (j > 1) csign = -csign;
    bool swap = (j == 1 || j == 2);
    float sv = swap ? pc : ps, cv = swap ? ps : pc;
    *s = ssign < 0 ? -sv : sv;
    *c = csign < 0 ? -cv : cv;
}
MIS_HD float mis_sinf(float x) { float s, c; mis_sincosf(x, &s, &c); return s; }
MIS_HD float mis_cosf(float x) { float s, c; mis_sincosf(x, &s, &c); return c; }

MIS_HD float mis_atanf(float xx) {
    float x = fabsf(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = 8.05374449538e-2f * z - 1.38776856032E-1f;
    p = p * z + 1.99777106478E-1f;
    p = p * z - 3.33329491539E-1f;
    p = p * z;
    p = p * x;
    p = p + x;
    y = y + p;
    return xx < 0 ? -y : y;
}
MIS_HD float mis_atan2f(float y, float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return 1.5707963267948966192f;
        if (y < 0.0f) return -1.5707963267948966192f;
        return 0.0f;
    }
    float z = mis_atanf(y / x);
    if (x < 0.0f) z = (y >= 0.0f) ? z + MIS_PI_F : z - MIS_PI_F;
    return z;
}
MIS_HD float mis_asinf(float xx) {
    float a = fabsf(xx), x, z;
    bool flag = false;
    if (a > 1.0f) return 0.0f;
    if (a < 1.0e-4f) return xx;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = true; }
    else { x = a; z = x * x; }
    float p = 4.2163199048E-2f * z + 2.4181311049E-2f;
    p = p * z + 4.5470025998E-2f;
    p = p * z + 7.4953002686E-2f;
    p = p * z + 1.6666752422E-1f;
    p = p * z;
    p = p * x;
    z = p + x;
    if (flag) { z = z + z; z = 1.5707963267948966192f - z; }
    return xx < 0 ? -z : z;
}
MIS_HD float mis_acosf(float x) {
    if (x < -1.0f || x > 1.0f) return 0.0f;
    if (x < -0.5f) return MIS_PI_F - 2.0f * mis_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * mis_asinf(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - mis_asinf(x);
}

// cv::fastAtan2 (degrees)
MIS_HD float mis_fast_atan2(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Cephes expf with the final scaling by exact power-of-two products (the SIFT weights; same code as the oracle's mo_expf)
MIS_HD float mis_expf(float xx) {
    float x = xx, z;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.f;
    z = floorf(1.44269504088896341f * x + 0.5f);
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    const int n = (int)z;
    z = x * x;
    z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x + 4.1665795894E-2f) * x + 1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
    union { uint32_t u; float f; } a, b;
    if (n < -126) {
        a.u = (uint32_t)1 << 23;  // 2^-126
        b.u = (uint32_t)(n + 126 + 127) << 23;
        return (z * b.f) * a.f;
    }
    a.u = (uint32_t)(n + 127) << 23;
    return z * a.f;
}
MIS_HD int mis_floor_f(float v) { int i = (int)v; return i - ((float)i > v); }

// natural log in f64 from + * / and exponent extraction (RANSAC iteration-count update)
MIS_HD double mis_log_d(double x) {
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff) - 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    for (int k = 21; k >= 1; k -= 2) p = p * z + 1.0 / (double)k;
    return (double)e * 0.6931471805599453094 + 2.0 * s * p;
}
