// rotation.hpp -- host-side camera-rotation math of the stitcher (C++17, no OpenCV).
//
// Mirrors the interface of the reference's header-only helpers so that the driver reads like main():
//   Quaternion<T>::set / x,y,z,w / setFromRotationMatrix / toRotationMatrix   (image_stitching/quaternion.h:147, :260-322, :564-596)
//   rotationMatrixToEulerAngles<T>(R, order), eulerAnglesToRotationMatrix<T>(e, order), EulerOrder
//                                                                               (image_stitching/euler.h:4-300, euler_order.h:3-11)
//   rehandCameraRotation(R, isPortrait): the quaternion sign flip of image_stitching.cpp:485-517
// over a plain row-major Mat3<T> instead of cv::Mat.  Only the members the live path instantiates exist
// (SURVEY row a7/a8).  Checked against the reference's known-answer vectors by host/rotation_kat.cpp.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <ostream>

namespace mis {

template <typename T>
struct Mat3 {
    std::array<T, 9> m{};
    T& operator()(int r, int c) { return m[r * 3 + c]; }
    const T& operator()(int r, int c) const { return m[r * 3 + c]; }
    template <typename U>
    Mat3<U> cast() const { Mat3<U> o; for (int i = 0; i < 9; i++) o.m[i] = static_cast<U>(m[i]); return o; }
};

enum class EulerOrder { XYZ, YXZ, ZXY, ZYX, YZX, XZY };

template <typename T>
std::array<T, 3> rotationMatrixToEulerAngles(const Mat3<T>& R, EulerOrder order) {
    const T m11 = R(0, 0), m12 = R(0, 1), m13 = R(0, 2), m21 = R(1, 0), m22 = R(1, 1), m23 = R(1, 2), m31 = R(2, 0), m32 = R(2, 1), m33 = R(2, 2);
    auto unit = [](T v) { return std::clamp(v, T(-1), T(1)); };
    auto regular = [](T v) { return std::abs(v) < 0.9999999; };  // away from gimbal lock
    T x = 0, y = 0, z = 0;
    switch (order) {
    case EulerOrder::XYZ:
        y = std::asin(unit(m13));
        if (regular(m13)) { x = std::atan2(-m23, m33); z = std::atan2(-m12, m11); } else { x = std::atan2(m32, m22); z = 0; }
        break;
    case EulerOrder::YXZ:
        x = std::asin(-unit(m23));
        if (regular(m23)) { y = std::atan2(m13, m33); z = std::atan2(m21, m22); } else { y = std::atan2(-m31, m11); z = 0; }
        break;
    case EulerOrder::ZXY:
        x = std::asin(unit(m32));
        if (regular(m32)) { y = std::atan2(-m31, m33); z = std::atan2(-m12, m22); } else { y = 0; z = std::atan2(m21, m11); }
        break;
    case EulerOrder::ZYX:
        y = std::asin(-unit(m31));
        if (regular(m31)) { x = std::atan2(m32, m33); z = std::atan2(m21, m11); } else { x = 0; z = std::atan2(-m12, m22); }
        break;
    case EulerOrder::YZX:
        z = std::asin(unit(m21));
        if (regular(m21)) { x = std::atan2(-m23, m22); y = std::atan2(-m31, m11); } else { x = 0; y = std::atan2(m13, m33); }
        break;
    case EulerOrder::XZY:
        z = std::asin(-unit(m12));
        if (regular(m12)) { x = std::atan2(m32, m22); y = std::atan2(m13, m11); } else { x = std::atan2(-m23, m33); y = 0; }
        break;
    }
    return {x, y, z};
}

template <typename T>
Mat3<T> eulerAnglesToRotationMatrix(const std::array<T, 3>& euler, EulerOrder order) {
    const T a = std::cos(euler[0]), b = std::sin(euler[0]), c = std::cos(euler[1]), d = std::sin(euler[1]), e = std::cos(euler[2]), f = std::sin(euler[2]);
    Mat3<T> R;
    switch (order) {
    case EulerOrder::XYZ: {
        const T ae = a * e, af = a * f, be = b * e, bf = b * f;
        R(0, 0) = c * e; R(0, 1) = -c * f; R(0, 2) = d;
        R(1, 0) = af + be * d; R(1, 1) = ae - bf * d; R(1, 2) = -b * c;
        R(2, 0) = bf - ae * d; R(2, 1) = be + af * d; R(2, 2) = a * c;
    } break;
    case EulerOrder::YXZ: {
        const T ce = c * e, cf = c * f, de = d * e, df = d * f;
        R(0, 0) = ce + df * b; R(0, 1) = de * b - cf; R(0, 2) = a * d;
        R(1, 0) = a * f; R(1, 1) = a * e; R(1, 2) = -b;
        R(2, 0) = cf * b - de; R(2, 1) = df + ce * b; R(2, 2) = a * c;
    } break;
    case EulerOrder::ZXY: {
        const T ce = c * e, cf = c * f, de = d * e, df = d * f;
        R(0, 0) = ce - df * b; R(0, 1) = -a * f; R(0, 2) = de + cf * b;
        R(1, 0) = cf + de * b; R(1, 1) = a * e; R(1, 2) = df - ce * b;
        R(2, 0) = -a * d; R(2, 1) = b; R(2, 2) = a * c;
    } break;
    case EulerOrder::ZYX: {
        const T ae = a * e, af = a * f, be = b * e, bf = b * f;
        R(0, 0) = c * e; R(0, 1) = be * d - af; R(0, 2) = ae * d + bf;
        R(1, 0) = c * f; R(1, 1) = bf * d + ae; R(1, 2) = af * d - be;
        R(2, 0) = -d; R(2, 1) = b * c; R(2, 2) = a * c;
    } break;
    case EulerOrder::YZX: {
        const T ac = a * c, ad = a * d, bc = b * c, bd = b * d;
        R(0, 0) = c * e; R(0, 1) = bd - ac * f; R(0, 2) = bc * f + ad;
        R(1, 0) = f; R(1, 1) = a * e; R(1, 2) = -b * e;
        R(2, 0) = -d * e; R(2, 1) = ad * f + bc; R(2, 2) = ac - bd * f;
    } break;
    case EulerOrder::XZY: {
        const T ac = a * c, ad = a * d, bc = b * c, bd = b * d;
        R(0, 0) = c * e; R(0, 1) = -f; R(0, 2) = d * e;
        R(1, 0) = ac * f + bd; R(1, 1) = a * e; R(1, 2) = ad * f - bc;
        R(2, 0) = bc * f - ad; R(2, 1) = b * e; R(2, 2) = bd * f + ac;
    } break;
    }
    return R;
}

template <typename T>
class Quaternion {
    T _x, _y, _z, _w;

public:
    Quaternion(T x = 0, T y = 0, T z = 0, T w = 1) : _x(x), _y(y), _z(z), _w(w) {}
    T x() const { return _x; }
    T y() const { return _y; }
    T z() const { return _z; }
    T w() const { return _w; }
    Quaternion& set(T x, T y, T z, T w) { _x = x; _y = y; _z = z; _w = w; return *this; }

    // assumes a pure rotation matrix; branch on the largest diagonal term
    template <typename M>
    Quaternion& setFromRotationMatrix(const Mat3<M>& R) {
        const auto m11 = R(0, 0), m12 = R(0, 1), m13 = R(0, 2), m21 = R(1, 0), m22 = R(1, 1), m23 = R(1, 2), m31 = R(2, 0), m32 = R(2, 1), m33 = R(2, 2);
        const auto trace = m11 + m22 + m33;
        if (trace > 0) {
            const auto s = 0.5 / std::sqrt(trace + 1.0);
            _w = 0.25 / s; _x = (m32 - m23) * s; _y = (m13 - m31) * s; _z = (m21 - m12) * s;
        } else if (m11 > m22 && m11 > m33) {
            const auto s = 2.0 * std::sqrt(1.0 + m11 - m22 - m33);
            _w = (m32 - m23) / s; _x = 0.25 * s; _y = (m12 + m21) / s; _z = (m13 + m31) / s;
        } else if (m22 > m33) {
            const auto s = 2.0 * std::sqrt(1.0 + m22 - m11 - m33);
            _w = (m13 - m31) / s; _x = (m12 + m21) / s; _y = 0.25 * s; _z = (m23 + m32) / s;
        } else {
            const auto s = 2.0 * std::sqrt(1.0 + m33 - m11 - m22);
            _w = (m21 - m12) / s; _x = (m13 + m31) / s; _y = (m23 + m32) / s; _z = 0.25 * s;
        }
        return *this;
    }

    Mat3<T> toRotationMatrix() const {
        const T x2 = _x + _x, y2 = _y + _y, z2 = _z + _z;
        const T xx = _x * x2, xy = _x * y2, xz = _x * z2, yy = _y * y2, yz = _y * z2, zz = _z * z2, wx = _w * x2, wy = _w * y2, wz = _w * z2;
        Mat3<T> R;
        R(0, 0) = 1 - (yy + zz); R(1, 0) = xy + wz; R(2, 0) = xz - wy;
        R(0, 1) = xy - wz; R(1, 1) = 1 - (xx + zz); R(2, 1) = yz + wx;
        R(0, 2) = xz + wy; R(1, 2) = yz - wx; R(2, 2) = 1 - (xx + yy);
        return R;
    }
};

template <typename T>
std::ostream& operator<<(std::ostream& s, const Quaternion<T>& q) {
    return s << "[" << q.x() << "," << q.y() << "," << q.z() << "," << q.w() << "]";
}

// image_stitching.cpp:485-517: sensor rotation -> quaternion -> handedness flip -> camera R
inline Mat3<double> rehandCameraRotation(const Mat3<double>& R, bool isPortrait) {
    Quaternion<double> q, q2;
    q.setFromRotationMatrix<double>(R);
    if (isPortrait) q2.set(q.y(), q.x(), -q.z(), q.w());
    else q2.set(-q.x(), q.y(), -q.z(), q.w());
    return q2.toRotationMatrix();
}

}  // namespace mis
