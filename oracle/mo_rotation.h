/*
 * mo_rotation.h -- ORACLE (test infrastructure): restatement of the reference-owned rotation
 * math: image_stitching/euler.h:4-133 (rotationMatrixToEulerAngles), :135-300
 * (eulerAnglesToRotationMatrix), image_stitching/euler_order.h:3-11, image_stitching/quaternion.h
 * :147 (set), :260-322 (setFromRotationMatrix), :564-596 (toRotationMatrix) and the camera
 * re-handing at image_stitching/image_stitching.cpp:485-517.
 * PINNED by the known-answer vectors of SURVEY.md section 8(c) (tests/golden/rotation_kat.json).
 */
#ifndef MO_ROTATION_H
#define MO_ROTATION_H
#ifdef __cplusplus
extern "C" {
#endif
enum { MO_EULER_XYZ = 0, MO_EULER_YXZ, MO_EULER_ZXY, MO_EULER_ZYX, MO_EULER_YZX, MO_EULER_XZY };
int mo_rot_to_euler_d(const double R[9], int order, double e[3]);
int mo_rot_to_euler_f(const float R[9], int order, float e[3]);
int mo_euler_to_rot_d(const double e[3], int order, double R[9]);
int mo_euler_to_rot_f(const float e[3], int order, float R[9]);
void mo_quat_from_rot_d(const double R[9], double q[4]); /* q = (x,y,z,w) */
void mo_quat_to_rot_d(const double q[4], double R[9]);
/* image_stitching.cpp:485-517: R -> q -> flipped q2 -> R' */
void mo_camera_rehand_d(const double R[9], int is_portrait, double Rout[9]);
#ifdef __cplusplus
}
#endif
#endif
