/*
 * mo_math.c -- ORACLE (test infrastructure): scalar math shared by the restated stages.
 * See mo_common.h for the scope statement.  Build with -ffp-contract=off.
 */
#include "mo_common.h"
#include <string.h>

/* cv::fastAtan2 [OpenCV core/src/mathfuncs_core.simd.hpp, atanImpl]; SURVEY.md A.0.
 * Used by ICAngles for the ORB orientation (reference call site image_stitching.cpp:613). */
float mo_fast_atan2(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
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

/* Cephes single-precision sin/cos: Cody-Waite reduction by pi/4 octants and degree-7/8
 * polynomials.  Valid for |x| < 8192.  Every operation is an IEEE f32 + - * so the GPU
 * kernel (csrc/dev_math.h) reproduces it bit for bit. */
#define MO_FOPI 1.27323954473516f
#define MO_DP1 0.78515625f
#define MO_DP2 2.4187564849853515625e-4f
#define MO_DP3 3.77489497744594108e-8f

static inline float mo_sin_poly(float x, float z) {
    float y = -1.9515295891E-4f * z + 8.3321608736E-3f;
    y = y * z - 1.6666654611E-1f;
    y = y * z;
    y = y * x;
    return y + x;
}
static inline float mo_cos_poly(float z) {
    float y = 2.443315711809948E-005f * z - 1.388731625493765E-003f;
    y = y * z + 4.166664568298827E-002f;
    y = y * z;
    y = y * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

float mo_sinf(float xx) {
    float x = xx, y, z;
    int sign = 1;
    unsigned j;
    if (x < 0) { sign = -1; x = -x; }
    j = (unsigned)(MO_FOPI * x);
    y = (float)j;
    if (j & 1u) { j += 1; y += 1.0f; }
    j &= 7u;
    if (j > 3) { sign = -sign; j -= 4; }
    x = ((x - y * MO_DP1) - y * MO_DP2) - y * MO_DP3;
    z = x * x;
    if (j == 1 || j == 2) y = mo_cos_poly(z);
    else y = mo_sin_poly(x, z);
    return sign < 0 ? -y : y;
}

float mo_cosf(float xx) {
    float x = xx, y, z;
    int sign = 1;
    unsigned j;
    if (x < 0) x = -x;
    j = (unsigned)(MO_FOPI * x);
    y = (float)j;
    if (j & 1u) { j += 1; y += 1.0f; }
    j &= 7u;
    if (j > 3) { j -= 4; sign = -sign; }
    if (j > 1) sign = -sign;
    x = ((x - y * MO_DP1) - y * MO_DP2) - y * MO_DP3;
    z = x * x;
    if (j == 1 || j == 2) y = mo_sin_poly(x, z);
    else y = mo_cos_poly(z);
    return sign < 0 ? -y : y;
}

/* Cephes atanf / atan2f. */
static float mo_atanf(float xx) {
    float x = xx, y, z;
    int sign = 1;
    if (x < 0) { sign = -1; x = -x; }
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    z = x * x;
    {
        float p = 8.05374449538e-2f * z - 1.38776856032E-1f;
        p = p * z + 1.99777106478E-1f;
        p = p * z - 3.33329491539E-1f;
        p = p * z;
        p = p * x;
        p = p + x;
        y = y + p;
    }
    return sign < 0 ? -y : y;
}

float mo_atan2f(float y, float x) {
    float z;
    if (x == 0.0f) {
        if (y > 0.0f) return 1.5707963267948966192f;
        if (y < 0.0f) return -1.5707963267948966192f;
        return 0.0f;
    }
    z = mo_atanf(y / x);
    if (x < 0.0f) {
        if (y >= 0.0f) z = z + MO_PI_F;
        else z = z - MO_PI_F;
    }
    return z;
}

static float mo_asinf(float xx) {
    float a = xx, x, z;
    int sign = 1, flag = 0;
    if (a < 0) { sign = -1; a = -a; }
    if (a > 1.0f) return 0.0f;
    if (a < 1.0e-4f) return xx;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = 1; }
    else { x = a; z = x * x; }
    {
        float p = 4.2163199048E-2f * z + 2.4181311049E-2f;
        p = p * z + 4.5470025998E-2f;
        p = p * z + 7.4953002686E-2f;
        p = p * z + 1.6666752422E-1f;
        p = p * z;
        p = p * x;
        z = p + x;
    }
    if (flag) { z = z + z; z = 1.5707963267948966192f - z; }
    return sign < 0 ? -z : z;
}

float mo_acosf(float x) {
    if (x < -1.0f || x > 1.0f) return 0.0f;
    if (x < -0.5f) return MO_PI_F - 2.0f * mo_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * mo_asinf(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - mo_asinf(x);
}

/* log(x), x > 0 finite normal: x = m * 2^e with m in [sqrt(1/2), sqrt(2));
 * log(m) = 2*atanh(s), s = (m-1)/(m+1), odd series to s^23. */
double mo_log_d(double x) {
    uint64_t bits;
    int e, k;
    double m, f, s, z, p;
    memcpy(&bits, &x, 8);
    e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    memcpy(&m, &bits, 8); /* m in [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    f = m - 1.0;
    s = f / (2.0 + f);
    z = s * s;
    p = 1.0 / 23.0;
    for (k = 21; k >= 1; k -= 2) p = p * z + 1.0 / (double)k;
    return (double)e * 0.6931471805599453094 + 2.0 * s * p;
}
