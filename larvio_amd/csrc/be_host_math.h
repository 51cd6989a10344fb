// be_host_math.h — host-side fixed-size math of the back-end (state injection, clone bookkeeping, re-anchoring Jacobian).
// Quaternions are stored [x y z w] (Hamilton) as include/larvio/math_utils.hpp:54-102; conversions restate Eigen's
// Quaternion::toRotationMatrix / Quaternion(Matrix3) which is what larvio.cpp evaluates.  Row-major double.
#pragma once
#include <math.h>
#include <string.h>

static inline void m3_mul(const double* A, const double* B, double* C)
{   /* C = A*B (3x3), C may not alias */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0.; for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
        C[i * 3 + j] = s;
    }
}
static inline void m3_t(const double* A, double* T) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[j * 3 + i]; }
static inline void m3_v(const double* A, const double* v, double* o)
{   double t[3]; for (int i = 0; i < 3; ++i) t[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2]; o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; }
static inline void m3t_v(const double* A, const double* v, double* o)
{   double t[3]; for (int i = 0; i < 3; ++i) t[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2]; o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; }
static inline void skew3(const double* w, double* S)
{   /* math_utils.hpp:26-38 */
    S[0] = 0; S[1] = -w[2]; S[2] = w[1]; S[3] = w[2]; S[4] = 0; S[5] = -w[0]; S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
static inline double v3_norm(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* Eigen Quaterniond(w,x,y,z).toRotationMatrix(), q = [x y z w] */
static inline void quat_to_rot(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
/* Eigen Quaterniond(Matrix3d), result [x y z w] */
static inline void rot_to_quat(const double* m, double* q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}
/* Eigen quaternion product a*b (no normalisation), [x y z w] */
static inline void quat_mul(const double* a, const double* b, double* o)
{
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    double r[4];
    r[3] = aw * bw - ax * bx - ay * by - az * bz;
    r[0] = aw * bx + ax * bw + ay * bz - az * by;
    r[1] = aw * by + ay * bw + az * bx - ax * bz;
    r[2] = aw * bz + az * bw + ax * by - ay * bx;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
}
/* smallAngleQuaternion (math_utils.hpp:85-102), [x y z w] */
static inline void small_angle_quat(const double* dtheta, double* q)
{
    double d[3] = {dtheta[0] / 2.0, dtheta[1] / 2.0, dtheta[2] / 2.0};
    double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 <= 1) { q[0] = d[0]; q[1] = d[1]; q[2] = d[2]; q[3] = sqrt(1 - n2); }
    else {
        double s = sqrt(1 + n2);
        q[0] = d[0] / s; q[1] = d[1] / s; q[2] = d[2] / s; q[3] = 1 / s;
    }
}
