/*
 * mo_rotation.c -- ORACLE (test infrastructure): see mo_rotation.h for the reference sites.
 */
#include "mo_rotation.h"
#include <math.h>

#define CLAMP1(v, T) ((v) < (T)-1 ? (T)-1 : ((v) > (T)1 ? (T)1 : (v)))

#define DEF_ROT_TO_EULER(NAME, T, ASIN, ATAN2, FABS)                                              \
    int NAME(const T R[9], int order, T e[3]) {                                                   \
        T x, y, z;                                                                                \
        T m11 = R[0], m12 = R[1], m13 = R[2], m21 = R[3], m22 = R[4], m23 = R[5], m31 = R[6],     \
          m32 = R[7], m33 = R[8];                                                                 \
        switch (order) {                                                                          \
        case MO_EULER_XYZ:                                                                        \
            y = ASIN(CLAMP1(m13, T));                                                             \
            if (FABS(m13) < 0.9999999) { x = ATAN2(-m23, m33); z = ATAN2(-m12, m11); }            \
            else { x = ATAN2(m32, m22); z = 0; }                                                  \
            break;                                                                                \
        case MO_EULER_YXZ:                                                                        \
            x = ASIN(-CLAMP1(m23, T));                                                            \
            if (FABS(m23) < 0.9999999) { y = ATAN2(m13, m33); z = ATAN2(m21, m22); }              \
            else { y = ATAN2(-m31, m11); z = 0; }                                                 \
            break;                                                                                \
        case MO_EULER_ZXY:                                                                        \
            x = ASIN(CLAMP1(m32, T));                                                             \
            if (FABS(m32) < 0.9999999) { y = ATAN2(-m31, m33); z = ATAN2(-m12, m22); }            \
            else { y = 0; z = ATAN2(m21, m11); }                                                  \
            break;                                                                                \
        case MO_EULER_ZYX:                                                                        \
            y = ASIN(-CLAMP1(m31, T));                                                            \
            if (FABS(m31) < 0.9999999) { x = ATAN2(m32, m33); z = ATAN2(m21, m11); }              \
            else { x = 0; z = ATAN2(-m12, m22); }                                                 \
            break;                                                                                \
        case MO_EULER_YZX:                                                                        \
            z = ASIN(CLAMP1(m21, T));                                                             \
            if (FABS(m21) < 0.9999999) { x = ATAN2(-m23, m22); y = ATAN2(-m31, m11); }            \
            else { x = 0; y = ATAN2(m13, m33); }                                                  \
            break;                                                                                \
        case MO_EULER_XZY:                                                                        \
            z = ASIN(-CLAMP1(m12, T));                                                            \
            if (FABS(m12) < 0.9999999) { x = ATAN2(m32, m22); y = ATAN2(m13, m11); }              \
            else { x = ATAN2(-m23, m33); y = 0; }                                                 \
            break;                                                                                \
        default: return -1;                                                                       \
        }                                                                                         \
        e[0] = x; e[1] = y; e[2] = z;                                                             \
        return 0;                                                                                 \
    }

DEF_ROT_TO_EULER(mo_rot_to_euler_d, double, asin, atan2, fabs)
DEF_ROT_TO_EULER(mo_rot_to_euler_f, float, asinf, atan2f, (double)fabsf)

#define DEF_EULER_TO_ROT(NAME, T, COS, SIN)                                                        \
    int NAME(const T eu[3], int order, T R[9]) {                                                   \
        T te[11];                                                                                  \
        T x = eu[0], y = eu[1], z = eu[2];                                                         \
        T a = COS(x), b = SIN(x), c = COS(y), d = SIN(y), e = COS(z), f = SIN(z);                  \
        switch (order) {                                                                           \
        case MO_EULER_XYZ: { T ae = a * e, af = a * f, be = b * e, bf = b * f;                     \
            te[0] = c * e; te[4] = -c * f; te[8] = d;                                              \
            te[1] = af + be * d; te[5] = ae - bf * d; te[9] = -b * c;                              \
            te[2] = bf - ae * d; te[6] = be + af * d; te[10] = a * c; break; }                     \
        case MO_EULER_YXZ: { T ce = c * e, cf = c * f, de = d * e, df = d * f;                     \
            te[0] = ce + df * b; te[4] = de * b - cf; te[8] = a * d;                               \
            te[1] = a * f; te[5] = a * e; te[9] = -b;                                              \
            te[2] = cf * b - de; te[6] = df + ce * b; te[10] = a * c; break; }                     \
        case MO_EULER_ZXY: { T ce = c * e, cf = c * f, de = d * e, df = d * f;                     \
            te[0] = ce - df * b; te[4] = -a * f; te[8] = de + cf * b;                              \
            te[1] = cf + de * b; te[5] = a * e; te[9] = df - ce * b;                               \
            te[2] = -a * d; te[6] = b; te[10] = a * c; break; }                                    \
        case MO_EULER_ZYX: { T ae = a * e, af = a * f, be = b * e, bf = b * f;                     \
            te[0] = c * e; te[4] = be * d - af; te[8] = ae * d + bf;                               \
            te[1] = c * f; te[5] = bf * d + ae; te[9] = af * d - be;                               \
            te[2] = -d; te[6] = b * c; te[10] = a * c; break; }                                    \
        case MO_EULER_YZX: { T ac = a * c, ad = a * d, bc = b * c, bd = b * d;                     \
            te[0] = c * e; te[4] = bd - ac * f; te[8] = bc * f + ad;                               \
            te[1] = f; te[5] = a * e; te[9] = -b * e;                                              \
            te[2] = -d * e; te[6] = ad * f + bc; te[10] = ac - bd * f; break; }                    \
        case MO_EULER_XZY: { T ac = a * c, ad = a * d, bc = b * c, bd = b * d;                     \
            te[0] = c * e; te[4] = -f; te[8] = d * e;                                              \
            te[1] = ac * f + bd; te[5] = a * e; te[9] = ad * f - bc;                               \
            te[2] = bc * f - ad; te[6] = b * e; te[10] = bd * f + ac; break; }                     \
        default: return -1;                                                                        \
        }                                                                                          \
        R[0] = te[0]; R[3] = te[1]; R[6] = te[2];                                                  \
        R[1] = te[4]; R[4] = te[5]; R[7] = te[6];                                                  \
        R[2] = te[8]; R[5] = te[9]; R[8] = te[10];                                                 \
        return 0;                                                                                  \
    }

DEF_EULER_TO_ROT(mo_euler_to_rot_d, double, cos, sin)
DEF_EULER_TO_ROT(mo_euler_to_rot_f, float, cosf, sinf)

void mo_quat_from_rot_d(const double R[9], double q[4]) {
    double m11 = R[0], m12 = R[1], m13 = R[2], m21 = R[3], m22 = R[4], m23 = R[5], m31 = R[6], m32 = R[7], m33 = R[8];
    double trace = m11 + m22 + m33, x, y, z, w;
    if (trace > 0) {
        double s = 0.5 / sqrt(trace + 1.0);
        w = 0.25 / s; x = (m32 - m23) * s; y = (m13 - m31) * s; z = (m21 - m12) * s;
    } else if (m11 > m22 && m11 > m33) {
        double s = 2.0 * sqrt(1.0 + m11 - m22 - m33);
        w = (m32 - m23) / s; x = 0.25 * s; y = (m12 + m21) / s; z = (m13 + m31) / s;
    } else if (m22 > m33) {
        double s = 2.0 * sqrt(1.0 + m22 - m11 - m33);
        w = (m13 - m31) / s; x = (m12 + m21) / s; y = 0.25 * s; z = (m23 + m32) / s;
    } else {
        double s = 2.0 * sqrt(1.0 + m33 - m11 - m22);
        w = (m21 - m12) / s; x = (m13 + m31) / s; y = (m23 + m32) / s; z = 0.25 * s;
    }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

void mo_quat_to_rot_d(const double q[4], double R[9]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double x2 = x + x, y2 = y + y, z2 = z + z;
    double xx = x * x2, xy = x * y2, xz = x * z2;
    double yy = y * y2, yz = y * z2, zz = z * z2;
    double wx = w * x2, wy = w * y2, wz = w * z2;
    R[0] = (1 - (yy + zz)); R[3] = (xy + wz); R[6] = (xz - wy);
    R[1] = (xy - wz); R[4] = (1 - (xx + zz)); R[7] = (yz + wx);
    R[2] = (xz + wy); R[5] = (yz - wx); R[8] = (1 - (xx + yy));
}

void mo_camera_rehand_d(const double R[9], int is_portrait, double Rout[9]) {
    double q[4], q2[4];
    mo_quat_from_rot_d(R, q);
    if (is_portrait) { q2[0] = q[1]; q2[1] = q[0]; q2[2] = -q[2]; q2[3] = q[3]; }
    else { q2[0] = -q[0]; q2[1] = q[1]; q2[2] = -q[2]; q2[3] = q[3]; }
    mo_quat_to_rot_d(q2, Rout);
}
