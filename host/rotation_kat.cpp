// rotation_kat.cpp -- checks host/rotation.hpp against the reference's known-answer vectors
// (tests/golden/rotation_kat.json).  Exit code 0 = all bit-exact.
#include "rotation.hpp"
#include <cstdio>
using namespace mis;
static int fails = 0;
static void eq(double a, double b, const char* what) { if (a != b) { std::printf("MISMATCH %s: %.17g vs %.17g\n", what, a, b); fails++; } }
int main() {
    const std::array<double, 3> e{0.17453292519943295, 0.5235987755982988, 0.08726646259971647};
    Mat3<double> R = eulerAnglesToRotationMatrix<double>(e, EulerOrder::YXZ);
    auto back = rotationMatrixToEulerAngles<double>(R, EulerOrder::YXZ);
    eq(back[0], 0.17453292519943295, "euler x"); eq(back[1], 0.52359877559829882, "euler y"); eq(back[2], 0.087266462599716474, "euler z");
    Quaternion<double> q;
    q.setFromRotationMatrix<double>(R);
    eq(q.x(), 0.095352424550506396, "q.x"); eq(q.y(), 0.25391661851111352, "q.y"); eq(q.z(), 0.019436667336159463, "q.z"); eq(q.w(), 0.96231828515262308, "q.w");
    Mat3<double> Rp = rehandCameraRotation(R, false);
    const double exp[9] = {0.87029713361349026, -0.011014609657371381, 0.49240387650610401, -0.085831651177431301, 0.98106026219040687,
                           0.17364817766693033, -0.48499054308336625, -0.19338934904742242, 0.85286853195244328};
    for (int i = 0; i < 9; i++) eq(Rp.m[i], exp[i], "R'");
    auto ef = rotationMatrixToEulerAngles<float>(Rp.cast<float>(), EulerOrder::YXZ);
    if (std::fabs(ef[0] + 0.17453292f) > 1e-7f || std::fabs(ef[1] - 0.52359879f) > 1e-7f || std::fabs(ef[2] + 0.0872664601f) > 1e-7f) { std::printf("MISMATCH float euler\n"); fails++; }
    std::printf(fails ? "rotation KAT FAILED\n" : "rotation KAT OK\n");
    return fails;
}
