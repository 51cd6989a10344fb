// be_init.h — the moving-start ("dynamic") initialiser of the filter: host code, cold path (runs until the filter has a state, then never again).
// Replaces /root/reference/src/DynamicInitializer.cpp (+ include/Initializer/DynamicInitializer.h), src/initial_sfm.cpp, src/solve_5pts.cpp,
// src/feature_manager.cpp, src/initial_alignment.cpp, include/Initializer/ImuPreintegration.h - LARVIO's adoption of VINS-Mono's
// initialisation: a window of WINDOW_SIZE + 1 = 11 feature messages with IMU pre-integration between them; once the window is full,
//   relativePose        a frame l with enough parallax against the newest one; their relative pose from the essential matrix
//   GlobalSFM           up-to-scale structure from motion over the window: PnP + two-view triangulation chained from (l, newest), then a
//                       bundle adjustment with the gauge fixed as the reference fixes it (rotation of l; translations of l and the newest)
//   VisualIMUAlignment  gyroscope bias from the rotation residuals of the pre-integrations, then the linear system for the frame
//                       velocities, gravity and metric scale, then the gravity refinement on its tangent plane
// and the state handed to the filter: attitude that puts gravity on -z, velocity of the newest frame, gyro bias, zero position.
// FlexibleInitializer.cpp:11-25 tries the static initialiser first and this one when that says no, with the same message.
//
// What comes from OpenCV / Ceres in the reference is restated from the published algorithms, on the host, in plain C++:
//   cv::findFundamentalMat(FM_RANSAC, 0.3/460, 0.99)   the library's own RANSAC kernel (lvk_find_fundamental, fe_track.hip: OpenCV's draws, OpenCV's
//                                                       mask, and the matrix OpenCV returns: the best minimal-sample model, NOT refitted)
//   cv::recoverPose                                     SVD of E, four (R, t) candidates, DLT triangulation, cheirality + 50-unit depth vote
//   cv::solvePnP(..., useExtrinsicGuess = true)         Levenberg-Marquardt on the reprojection error from the given pose (CvLevMarq's job)
//   ceres::Solve (DENSE_SCHUR, quaternion poses)        Levenberg-Marquardt with a Schur complement on the points; accepted like the
//                                                       reference accepts Ceres' answer: converged, or final cost < 5e-3 (initial_sfm.cpp:292)
// None of this is bit-comparable with the reference (different minimisers reach the same minimum to their tolerances); it is pinned by
// an independent numpy / scipy restatement (oracle/dyn_init.py; every intermediate and final result to 1e-13 noise-free / 1e-8 noisy,
// tests/test_oracle_dynamic_init.py through tests/host/init_replay.hip), by closed-form cases (tests/host/init_check.hip) and by
// moving-start runs against ground truth through lvk_ekf_process (tests/test_gpu_dynamic_init.py).
#pragma once
#include <array>
#include <map>
#include <vector>
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "../../include/lvk_c.h"
#include "be_host_math.h"

namespace lvk_init {

constexpr int WIN = 10;                                  // feature_manager.h:24
constexpr double GRAV_NORM = 9.81;                       // ImuPreintegration.h:19 (Gravity = (0, 0, 9.81))

// ------------------------------------------------------------------------------------------------ small dense algebra
// symmetric eigen-decomposition by cyclic Jacobi rotations (n <= 9 here): A (n x n, row-major, destroyed) -> eigenvalues w, eigenvectors
// in the COLUMNS of V
static inline void jacobi_eig(int n, double* A, double* V, double* w)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.;
        for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) {
            const double apq = A[p * n + q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (A[q * n + q] - A[p * n + p]) / (2. * apq);
            const double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
            const double c = 1. / sqrt(t * t + 1.), s = t * c;
            for (int k = 0; k < n; ++k) { const double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
            for (int k = 0; k < n; ++k) { const double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
            for (int k = 0; k < n; ++k) { const double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq; }
        }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}
// the eigenvector of the smallest eigenvalue of M^T M (M: m x n, row-major): the right singular vector DLT problems ask for
static inline void smallest_right_singular(const double* M, int m, int n, double* x)
{
    double A[81], V[81], w[9];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.; for (int k = 0; k < m; ++k) s += M[k * n + i] * M[k * n + j]; A[i * n + j] = s; }
    jacobi_eig(n, A, V, w);
    int best = 0; for (int i = 1; i < n; ++i) if (w[i] < w[best]) best = i;
    for (int i = 0; i < n; ++i) x[i] = V[i * n + best];
}
// SVD of a 3x3 matrix: A = U diag(S) V^T, singular values descending, U and V orthonormal (a zero singular value's U column is completed)
static inline void svd3(const double* A, double* U, double* S, double* V)
{
    double AtA[9], Vv[9], w[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0.; for (int k = 0; k < 3; ++k) s += A[k * 3 + i] * A[k * 3 + j]; AtA[i * 3 + j] = s; }
    jacobi_eig(3, AtA, Vv, w);
    int o[3] = {0, 1, 2};
    std::sort(o, o + 3, [&](int a, int b) { return w[a] > w[b]; });
    for (int c = 0; c < 3; ++c) { S[c] = sqrt(std::max(w[o[c]], 0.0)); for (int r = 0; r < 3; ++r) V[r * 3 + c] = Vv[r * 3 + o[c]]; }
    for (int c = 0; c < 3; ++c) {
        double u[3]; for (int r = 0; r < 3; ++r) u[r] = A[r * 3] * V[c] + A[r * 3 + 1] * V[3 + c] + A[r * 3 + 2] * V[6 + c];
        const double n = v3_norm(u);
        if (n > 1e-12 * (S[0] + 1e-300) && c < 2) for (int r = 0; r < 3; ++r) U[r * 3 + c] = u[r] / n;
        else if (c == 2) {     // third column: orthogonal complement of the first two (right-handed up to the sign of the singular value)
            const double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
            double cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            if (n > 1e-12 * (S[0] + 1e-300) && cr[0] * u[0] + cr[1] * u[1] + cr[2] * u[2] < 0.) { cr[0] = -cr[0]; cr[1] = -cr[1]; cr[2] = -cr[2]; }
            for (int r = 0; r < 3; ++r) U[r * 3 + 2] = cr[r];
        } else {               // a rank-0/1 matrix: any unit vector orthogonal to what is there
            double e[3] = {1, 0, 0};
            if (c == 1) { const double a[3] = {U[0], U[3], U[6]}; if (fabs(a[0]) > 0.9) { e[0] = 0; e[1] = 1; }
                          const double d = e[0] * a[0] + e[1] * a[1] + e[2] * a[2]; for (int r = 0; r < 3; ++r) e[r] -= d * a[r]; }
            const double ne = v3_norm(e); for (int r = 0; r < 3; ++r) U[r * 3 + c] = e[r] / ne;
        }
    }
}
static inline double det3(const double* M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
// solve the symmetric positive (semi-)definite system A x = b (n x n row-major, both destroyed; result in b): LDL^T without pivoting,
// a vanishing pivot drops its unknown (x_i = 0) - what Eigen's pivoted LDLT does with a singular direction, near enough for a gauge
static inline void sym_solve(int n, double* A, double* b)
{
    std::vector<double> d((size_t)n);
    for (int j = 0; j < n; ++j) {
        double dj = A[j * n + j];
        for (int k = 0; k < j; ++k) dj -= A[j * n + k] * A[j * n + k] * d[(size_t)k];
        const bool dead = !(fabs(dj) > 1e-14 * (fabs(A[j * n + j]) + 1e-300));
        d[(size_t)j] = dead ? 0. : dj;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k] * d[(size_t)k];
            A[i * n + j] = dead ? 0. : s / dj;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k]; b[i] = s; }
    for (int i = 0; i < n; ++i) b[i] = d[(size_t)i] != 0. ? b[i] / d[(size_t)i] : 0.;
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k]; b[i] = s; }
}
// Rodrigues: rotation vector -> matrix, and the left-multiplied small-rotation update R <- exp(w) R
static inline void rodrigues(const double* w, double* R)
{
    const double th = v3_norm(w);
    double K[9]; skew3(w, K);
    const double a = th < 1e-8 ? 1. - th * th / 6. : sin(th) / th, b = th < 1e-8 ? 0.5 - th * th / 24. : (1. - cos(th)) / (th * th);
    double K2[9]; m3_mul(K, K, K2);
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1. : 0.) + a * K[i] + b * K2[i];
}
// Quaterniond::FromTwoVectors(a, b) as a rotation matrix (a, b need not be unit)
static inline void rot_from_two_vectors(const double* a, const double* b, double* R)
{
    double v0[3], v1[3]; const double na = v3_norm(a), nb = v3_norm(b);
    for (int i = 0; i < 3; ++i) { v0[i] = a[i] / na; v1[i] = b[i] / nb; }
    const double c = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
    double q[4];
    if (c < -1. + 1e-12) {             // opposite vectors: half a turn about any axis orthogonal to a
        double e[3] = {1, 0, 0}; if (fabs(v0[0]) > 0.9) { e[0] = 0; e[1] = 1; }
        double ax[3] = {v0[1] * e[2] - v0[2] * e[1], v0[2] * e[0] - v0[0] * e[2], v0[0] * e[1] - v0[1] * e[0]};
        const double n = v3_norm(ax); q[0] = ax[0] / n; q[1] = ax[1] / n; q[2] = ax[2] / n; q[3] = 0.;
    } else {
        const double ax[3] = {v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]};
        const double s = sqrt((1. + c) * 2.), inv = 1. / s;
        q[0] = ax[0] * inv; q[1] = ax[1] * inv; q[2] = ax[2] * inv; q[3] = s * 0.5;
    }
    quat_to_rot(q, R);
}

// ------------------------------------------------------------------------------------------------ IMU pre-integration
// IntegrationBase (ImuPreintegration.h:27-230), mid-point rule.  Of its 15 x 15 Jacobian only d(rotation)/d(gyro bias) is ever read
// (solveGyroscopeBias), so that 3 x 3 block is what is carried: J <- (I - [w]x dt) J - I dt  (rows O_R of F, :104-105); the covariance
// is never read by the initialiser and is not carried at all.
struct PreInt {
    double acc0[3], gyr0[3], lin_acc[3], lin_gyr[3], ba[3], bg[3];
    double sum_dt = 0, dp[3], dq[4], dv[3], J_R_bg[9];
    std::vector<double> dt_buf; std::vector<std::array<double, 3>> acc_buf, gyr_buf;
    void start(const double* a0, const double* g0, const double* ba_, const double* bg_)
    {
        memcpy(acc0, a0, 24); memcpy(gyr0, g0, 24); memcpy(lin_acc, a0, 24); memcpy(lin_gyr, g0, 24); memcpy(ba, ba_, 24); memcpy(bg, bg_, 24);
        dt_buf.clear(); acc_buf.clear(); gyr_buf.clear(); zero();
    }
    void zero() { sum_dt = 0; for (int i = 0; i < 3; ++i) { dp[i] = 0; dv[i] = 0; dq[i] = 0; } dq[3] = 1; for (int i = 0; i < 9; ++i) J_R_bg[i] = 0; }
    // Eigen's Quaternion * Vector3 (QuaternionBase::_transformVector): v + w uv + q x uv with uv = 2 q x v.  For a unit quaternion this is
    // the rotation; result_delta_q below is NOT unit (|q|^2 = 1 + |w dt / 2|^2) when it rotates acc_1 (:77-78), and this formula - not
    // toRotationMatrix() - is what the reference then evaluates
    static void quat_times_vec(const double* q, const double* v, double* o)
    {
        double uv[3] = {2 * (q[1] * v[2] - q[2] * v[1]), 2 * (q[2] * v[0] - q[0] * v[2]), 2 * (q[0] * v[1] - q[1] * v[0])};
        const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
        for (int i = 0; i < 3; ++i) o[i] = v[i] + q[3] * uv[i] + c[i];
    }
    void propagate(double dt, const double* a1, const double* g1)
    {   // midPointIntegration (:62-142) + propagate (:144-167)
        double una0[3], t[3], w[3];
        for (int i = 0; i < 3; ++i) t[i] = acc0[i] - ba[i];
        quat_times_vec(dq, t, una0);
        for (int i = 0; i < 3; ++i) w[i] = 0.5 * (gyr0[i] + g1[i]) - bg[i];
        const double dqs[4] = {w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0};
        double q1[4]; quat_mul(dq, dqs, q1);
        double una1[3];
        for (int i = 0; i < 3; ++i) t[i] = a1[i] - ba[i];
        quat_times_vec(q1, t, una1);
        for (int i = 0; i < 3; ++i) { const double ua = 0.5 * (una0[i] + una1[i]); dp[i] = dp[i] + dv[i] * dt + 0.5 * ua * dt * dt; dv[i] = dv[i] + ua * dt; }
        double Wx[9], M[9], Jn[9]; skew3(w, Wx);
        for (int i = 0; i < 9; ++i) M[i] = (i % 4 == 0 ? 1. : 0.) - Wx[i] * dt;
        m3_mul(M, J_R_bg, Jn);
        for (int i = 0; i < 9; ++i) J_R_bg[i] = Jn[i] - (i % 4 == 0 ? dt : 0.);
        const double n = sqrt(q1[0] * q1[0] + q1[1] * q1[1] + q1[2] * q1[2] + q1[3] * q1[3]);
        for (int i = 0; i < 4; ++i) dq[i] = q1[i] / n;
        sum_dt += dt; memcpy(acc0, a1, 24); memcpy(gyr0, g1, 24);
    }
    void push_back(double dt, const double* a, const double* g)
    {
        dt_buf.push_back(dt); acc_buf.push_back({a[0], a[1], a[2]}); gyr_buf.push_back({g[0], g[1], g[2]});
        propagate(dt, a, g);
    }
    void repropagate(const double* ba_, const double* bg_)
    {   // :48-61
        memcpy(acc0, lin_acc, 24); memcpy(gyr0, lin_gyr, 24); memcpy(ba, ba_, 24); memcpy(bg, bg_, 24); zero();
        for (size_t i = 0; i < dt_buf.size(); ++i) propagate(dt_buf[i], acc_buf[i].data(), gyr_buf[i].data());
    }
};

// ------------------------------------------------------------------------------------------------ window bookkeeping
struct Pt2 { double x, y; };
struct FeatTrack { long long id; int start_frame; std::vector<Pt2> per_frame; int end_frame() const { return start_frame + (int)per_frame.size() - 1; } };
struct Frame {                                           // ImageFrame (initial_alignment.h:26-61)
    double t = 0; std::map<long long, Pt2> points; double R[9], T[3]; PreInt pre; bool has_pre = false, key = false;
};
struct SfmFeature { bool state = false; long long id = 0; std::vector<std::pair<int, Pt2>> obs; double position[3] = {0, 0, 0}; };

// ------------------------------------------------------------------------------------------------ two-view geometry
// normalised 8-point algorithm on n >= 8 correspondences (OpenCV's run8Point): F with x2^T F x1 = 0, F[8] = 1 when it is not ~0.
// NOT on the product's path (findFundamentalMat's RANSAC result is not refitted): the stand-in for the RANSAC stage in the host-only tests.
static inline bool eight_point(const std::vector<Pt2>& p1, const std::vector<Pt2>& p2, double* F)
{
    const int n = (int)p1.size();
    if (n < 8) return false;
    double c1[2] = {0, 0}, c2[2] = {0, 0};
    for (int i = 0; i < n; ++i) { c1[0] += p1[i].x; c1[1] += p1[i].y; c2[0] += p2[i].x; c2[1] += p2[i].y; }
    for (int k = 0; k < 2; ++k) { c1[k] /= n; c2[k] /= n; }
    double s1 = 0, s2 = 0;
    for (int i = 0; i < n; ++i) { s1 += sqrt((p1[i].x - c1[0]) * (p1[i].x - c1[0]) + (p1[i].y - c1[1]) * (p1[i].y - c1[1]));
                                  s2 += sqrt((p2[i].x - c2[0]) * (p2[i].x - c2[0]) + (p2[i].y - c2[1]) * (p2[i].y - c2[1])); }
    if (s1 < 1e-300 || s2 < 1e-300) return false;
    s1 = sqrt(2.) * n / s1; s2 = sqrt(2.) * n / s2;
    double A[81]; memset(A, 0, sizeof A);
    for (int i = 0; i < n; ++i) {
        const double x1 = (p1[i].x - c1[0]) * s1, y1 = (p1[i].y - c1[1]) * s1, x2 = (p2[i].x - c2[0]) * s2, y2 = (p2[i].y - c2[1]) * s2;
        const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.};
        for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) A[a * 9 + b] += r[a] * r[b];
    }
    double V[81], w[9]; jacobi_eig(9, A, V, w);
    int best = 0; for (int i = 1; i < 9; ++i) if (w[i] < w[best]) best = i;
    double F0[9]; for (int i = 0; i < 9; ++i) F0[i] = V[i * 9 + best];
    double U[9], S[3], Vv[9]; svd3(F0, U, S, Vv);
    S[2] = 0.;
    double US[9], Vt[9], Fr[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) US[r * 3 + c] = U[r * 3 + c] * S[c];
    m3_t(Vv, Vt); m3_mul(US, Vt, Fr);
    const double T1[9] = {s1, 0, -c1[0] * s1, 0, s1, -c1[1] * s1, 0, 0, 1}, T2[9] = {s2, 0, -c2[0] * s2, 0, s2, -c2[1] * s2, 0, 0, 1};
    double T2t[9], tmp[9]; m3_t(T2, T2t); m3_mul(T2t, Fr, tmp); m3_mul(tmp, T1, F);
    if (fabs(F[8]) > 2.220446049250313e-16) { const double s = 1. / F[8]; for (int i = 0; i < 9; ++i) F[i] *= s; }
    return true;
}
// DLT triangulation of one correspondence from two 3x4 camera matrices (GlobalSFM::triangulatePoint, initial_sfm.cpp:14-29; cv::triangulatePoints)
static inline void triangulate_dlt(const double* P0, const double* P1, Pt2 a, Pt2 b, double* X4)
{
    double M[16];
    for (int c = 0; c < 4; ++c) { M[c] = a.x * P0[8 + c] - P0[c]; M[4 + c] = a.y * P0[8 + c] - P0[4 + c]; M[8 + c] = b.x * P1[8 + c] - P1[c]; M[12 + c] = b.y * P1[8 + c] - P1[4 + c]; }
    smallest_right_singular(M, 4, 4, X4);
}
// cv::recoverPose(E, p1, p2, I, R, t, mask): the (R, t) of the four decompositions that puts the most masked-in points in front of both
// cameras (and nearer than 50 units); mask is updated to those points; returns their number
static inline int recover_pose(const double* E, const std::vector<Pt2>& p1, const std::vector<Pt2>& p2, std::vector<unsigned char>& mask, double* R, double* t)
{
    double U[9], S[3], V[9]; svd3(E, U, S, V);
    if (det3(U) < 0) for (int i = 0; i < 9; ++i) U[i] = -U[i];
    double Vt[9]; m3_t(V, Vt);
    if (det3(Vt) < 0) for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double UW[9], R1[9], R2[9], tt[3] = {U[2], U[5], U[8]};
    m3_mul(U, W, UW); m3_mul(UW, Vt, R1); m3_mul(U, Wt, UW); m3_mul(UW, Vt, R2);
    const int n = (int)p1.size();
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int good[4]; std::vector<unsigned char> m[4];
    for (int k = 0; k < 4; ++k) {
        const double* Rk = (k == 0 || k == 2) ? R1 : R2; const double sg = k < 2 ? 1. : -1.;
        double P1[12];
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) P1[r * 4 + c] = Rk[r * 3 + c]; P1[r * 4 + 3] = sg * tt[r]; }
        m[k].assign((size_t)n, 0); good[k] = 0;
        for (int i = 0; i < n; ++i) {
            double Q[4]; triangulate_dlt(P0, P1, p1[i], p2[i], Q);
            bool ok = Q[2] * Q[3] > 0;
            const double X[3] = {Q[0] / Q[3], Q[1] / Q[3], Q[2] / Q[3]};
            ok = ok && X[2] < 50.;
            const double z2 = P1[8] * X[0] + P1[9] * X[1] + P1[10] * X[2] + P1[11];
            ok = ok && z2 > 0 && z2 < 50.;
            ok = ok && (mask.empty() || mask[(size_t)i]);
            m[k][(size_t)i] = ok; good[k] += ok;
        }
    }
    int best = 0;                                         // the order of preference of solve_5pts.cpp:150-180
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
    else best = 3;
    memcpy(R, (best == 0 || best == 2) ? R1 : R2, 72);
    for (int r = 0; r < 3; ++r) t[r] = (best < 2 ? 1. : -1.) * tt[r];
    mask = m[best];
    return good[best];
}

// ------------------------------------------------------------------------------------------------ PnP and bundle adjustment
// cv::solvePnP(obj, img, I, no distortion, rvec, t, useExtrinsicGuess = true): Levenberg-Marquardt on the reprojection error from the
// given pose; R (3x3), t in place.  x_cam = R X + t, observation = (x/z, y/z).
static inline bool solve_pnp(const std::vector<std::array<double, 3>>& X, const std::vector<Pt2>& z, double* R, double* t)
{
    const int n = (int)X.size();
    auto cost = [&](const double* Rr, const double* tr) {
        double c = 0.;
        for (int i = 0; i < n; ++i) { double p[3]; m3_v(Rr, X[(size_t)i].data(), p); p[0] += tr[0]; p[1] += tr[1]; p[2] += tr[2];
                                      const double ex = p[0] / p[2] - z[(size_t)i].x, ey = p[1] / p[2] - z[(size_t)i].y; c += ex * ex + ey * ey; }
        return c;
    };
    double lambda = 1e-3, c0 = cost(R, t);
    for (int it = 0; it < 30; ++it) {
        double A[36], b[6]; memset(A, 0, sizeof A); memset(b, 0, sizeof b);
        for (int i = 0; i < n; ++i) {
            double q[3]; m3_v(R, X[(size_t)i].data(), q);             // R X (the rotated point: d(exp(w) R X)/dw = -[R X]x)
            const double p[3] = {q[0] + t[0], q[1] + t[1], q[2] + t[2]};
            const double iz = 1. / p[2], ex = p[0] * iz - z[(size_t)i].x, ey = p[1] * iz - z[(size_t)i].y;
            const double Jp[6] = {iz, 0, -p[0] * iz * iz, 0, iz, -p[1] * iz * iz};
            double Sq[9]; skew3(q, Sq);
            double J[12];
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
                double s = 0.; for (int k = 0; k < 3; ++k) s += Jp[r * 3 + k] * -Sq[k * 3 + c];
                J[r * 6 + c] = s; J[r * 6 + 3 + c] = Jp[r * 3 + c];
            }
            for (int a = 0; a < 6; ++a) { for (int c = 0; c < 6; ++c) A[a * 6 + c] += J[a] * J[c] + J[6 + a] * J[6 + c]; b[a] -= J[a] * ex + J[6 + a] * ey; }
        }
        bool improved = false;
        for (int tries = 0; tries < 8 && !improved; ++tries) {
            double Ad[36], d[6]; memcpy(Ad, A, sizeof A); memcpy(d, b, sizeof b);
            for (int a = 0; a < 6; ++a) Ad[a * 6 + a] *= 1. + lambda;
            sym_solve(6, Ad, d);
            double dR[9], Rn[9], tn[3] = {t[0] + d[3], t[1] + d[4], t[2] + d[5]};
            rodrigues(d, dR); m3_mul(dR, R, Rn);
            const double c1 = cost(Rn, tn);
            if (c1 < c0) {
                const double step = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
                const bool done = (c0 - c1) <= 1e-14 * (c0 + 1e-300) || step < 1e-12;
                memcpy(R, Rn, 72); memcpy(t, tn, 24); c0 = c1; lambda = std::max(lambda * 0.1, 1e-12); improved = true;
                if (done) return true;
            } else lambda *= 10.;
        }
        if (!improved) break;
    }
    return std::isfinite(c0);
}

// the window's bundle adjustment (initial_sfm.cpp:232-316): poses as (R_i, t_i) with x_cam = R_i X + t_i; the rotation of frame l and
// the translations of frames l and n-1 are held; Levenberg-Marquardt with the points eliminated (Schur complement), which is what
// Ceres' DENSE_SCHUR does.  Returns true as the reference accepts Ceres' result: converged, or final cost (1/2 sum r^2) < 5e-3.
static inline bool bundle_adjust(int nf, int l, std::vector<std::array<double, 9>>& Rc, std::vector<std::array<double, 3>>& tc, std::vector<SfmFeature>& feats)
{
    std::vector<int> pidx; for (size_t j = 0; j < feats.size(); ++j) if (feats[j].state) pidx.push_back((int)j);
    const int np = (int)pidx.size(), nc = 6 * nf;
    if (np == 0) return false;
    auto total_cost = [&](const std::vector<std::array<double, 9>>& Rr, const std::vector<std::array<double, 3>>& tr, const std::vector<std::array<double, 3>>& Xp) {
        double c = 0.;
        for (int a = 0; a < np; ++a) for (auto& ob : feats[(size_t)pidx[(size_t)a]].obs) {
            double p[3]; m3_v(Rr[(size_t)ob.first].data(), Xp[(size_t)a].data(), p); for (int k = 0; k < 3; ++k) p[k] += tr[(size_t)ob.first][(size_t)k];
            const double ex = p[0] / p[2] - ob.second.x, ey = p[1] / p[2] - ob.second.y; c += ex * ex + ey * ey;
        }
        return 0.5 * c;
    };
    std::vector<std::array<double, 3>> Xp((size_t)np);
    for (int a = 0; a < np; ++a) memcpy(Xp[(size_t)a].data(), feats[(size_t)pidx[(size_t)a]].position, 24);
    double lambda = 1e-4, c0 = total_cost(Rc, tc, Xp);
    bool converged = false;
    // What the reference asks of the solver is its termination TYPE (initial_sfm.cpp:292), and Ceres reports CONVERGENCE as soon as a
    // successful step changes the cost by less than function_tolerance (1e-6, the default: the reference sets no tolerance) times the
    // cost.  The iteration here goes on to the minimum, but the verdict is Ceres': in the flat valley of a window that has barely moved
    // the cost can still be falling in the 7th digit after 50 iterations - Ceres would have stopped, converged, long before (whole-program
    // fuzz case 20: a 10 Hz start from rest, the reference through at message 19, this code - refusing such windows - at message 24)
    bool ceres_stop = false;
    std::vector<double> Hcc((size_t)nc * nc), bc((size_t)nc), Hpp((size_t)np * 9), bp((size_t)np * 3), Hcp;      // Hcp: per observation blocks
    for (int it = 0; it < 50 && !converged; ++it) {
        std::fill(Hcc.begin(), Hcc.end(), 0.); std::fill(bc.begin(), bc.end(), 0.); std::fill(Hpp.begin(), Hpp.end(), 0.); std::fill(bp.begin(), bp.end(), 0.);
        struct ObsBlk { int cam, pt; double Jc[12], Jx[6], e[2]; };
        std::vector<ObsBlk> blks;
        for (int a = 0; a < np; ++a) for (auto& ob : feats[(size_t)pidx[(size_t)a]].obs) {
            const int f = ob.first; const double* R = Rc[(size_t)f].data();
            double q[3]; m3_v(R, Xp[(size_t)a].data(), q);
            const double p[3] = {q[0] + tc[(size_t)f][0], q[1] + tc[(size_t)f][1], q[2] + tc[(size_t)f][2]};
            const double iz = 1. / p[2];
            ObsBlk B; B.cam = f; B.pt = a; B.e[0] = p[0] * iz - ob.second.x; B.e[1] = p[1] * iz - ob.second.y;
            const double Jp[6] = {iz, 0, -p[0] * iz * iz, 0, iz, -p[1] * iz * iz};
            double Sq[9]; skew3(q, Sq);
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
                double s = 0., sx = 0.;
                for (int k = 0; k < 3; ++k) { s += Jp[r * 3 + k] * -Sq[k * 3 + c]; sx += Jp[r * 3 + k] * R[k * 3 + c]; }
                B.Jc[r * 6 + c] = (f == l) ? 0. : s;                                   // rotation of l held
                B.Jc[r * 6 + 3 + c] = (f == l || f == nf - 1) ? 0. : Jp[r * 3 + c];    // translations of l and the newest held
                B.Jx[r * 3 + c] = sx;
            }
            blks.push_back(B);
        }
        for (auto& B : blks) {
            double* Hc = &Hcc[(size_t)(6 * B.cam) * nc + 6 * B.cam];
            for (int a = 0; a < 6; ++a) { for (int c = 0; c < 6; ++c) Hc[a * nc + c] += B.Jc[a] * B.Jc[c] + B.Jc[6 + a] * B.Jc[6 + c]; bc[(size_t)(6 * B.cam + a)] -= B.Jc[a] * B.e[0] + B.Jc[6 + a] * B.e[1]; }
            double* Hp = &Hpp[(size_t)B.pt * 9];
            for (int a = 0; a < 3; ++a) { for (int c = 0; c < 3; ++c) Hp[a * 3 + c] += B.Jx[a] * B.Jx[c] + B.Jx[3 + a] * B.Jx[3 + c]; bp[(size_t)(3 * B.pt + a)] -= B.Jx[a] * B.e[0] + B.Jx[3 + a] * B.e[1]; }
        }
        bool improved = false;
        for (int tries = 0; tries < 10 && !improved; ++tries) {
            // damped point blocks inverted, reduced camera system S = Hcc' - sum_obs Hcp Hpp^-1 Hpc
            std::vector<double> S(Hcc), g(bc), Hpi((size_t)np * 9);
            for (int a = 0; a < nc; ++a) S[(size_t)a * nc + a] = Hcc[(size_t)a * nc + a] * (1. + lambda) + 1e-12;
            for (int a = 0; a < np; ++a) {
                double M[9]; memcpy(M, &Hpp[(size_t)a * 9], 72); { const double tr = M[0] + M[4] + M[8]; for (int k = 0; k < 3; ++k) M[k * 4] = M[k * 4] * (1. + lambda) + 1e-16 * tr; }     // (relative: an absolute 1e-12 outweighs the depth curvature of a point beyond ~1000 baselines and the valley is crawled along, case 20)
                const double d = det3(M);
                double* I = &Hpi[(size_t)a * 9];
                I[0] = (M[4] * M[8] - M[5] * M[7]) / d; I[1] = (M[2] * M[7] - M[1] * M[8]) / d; I[2] = (M[1] * M[5] - M[2] * M[4]) / d;
                I[3] = (M[5] * M[6] - M[3] * M[8]) / d; I[4] = (M[0] * M[8] - M[2] * M[6]) / d; I[5] = (M[2] * M[3] - M[0] * M[5]) / d;
                I[6] = (M[3] * M[7] - M[4] * M[6]) / d; I[7] = (M[1] * M[6] - M[0] * M[7]) / d; I[8] = (M[0] * M[4] - M[1] * M[3]) / d;
            }
            // W_{cam,pt} = sum over that pair's observation (one per pair) Jc^T Jx  (6 x 3)
            struct Wb { int cam, pt; double w[18]; };
            std::vector<Wb> Ws; Ws.reserve(blks.size());
            for (auto& B : blks) { Wb W; W.cam = B.cam; W.pt = B.pt; for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) W.w[a * 3 + c] = B.Jc[a] * B.Jx[c] + B.Jc[6 + a] * B.Jx[3 + c]; Ws.push_back(W); }
            // group by point: for every pair of observations of one point, S -= W_i Hpp^-1 W_j^T ; g -= W_i Hpp^-1 bp
            size_t s0 = 0;
            while (s0 < Ws.size()) {
                size_t s1 = s0; while (s1 < Ws.size() && Ws[s1].pt == Ws[s0].pt) ++s1;
                const double* I = &Hpi[(size_t)Ws[s0].pt * 9]; const double* bpp = &bp[(size_t)Ws[s0].pt * 3];
                for (size_t i = s0; i < s1; ++i) {
                    double WI[18];
                    for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) { double s = 0.; for (int k = 0; k < 3; ++k) s += Ws[i].w[a * 3 + k] * I[k * 3 + c]; WI[a * 3 + c] = s; }
                    for (int a = 0; a < 6; ++a) { double s = 0.; for (int k = 0; k < 3; ++k) s += WI[a * 3 + k] * bpp[k]; g[(size_t)(6 * Ws[i].cam + a)] -= s; }
                    for (size_t j = s0; j < s1; ++j)
                        for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) { double s = 0.; for (int k = 0; k < 3; ++k) s += WI[a * 3 + k] * Ws[j].w[c * 3 + k]; S[(size_t)(6 * Ws[i].cam + a) * nc + 6 * Ws[j].cam + c] -= s; }
                }
                s0 = s1;
            }
            sym_solve(nc, S.data(), g.data());            // held parameters have zero rows / columns: their steps come out as 0
            std::vector<std::array<double, 9>> Rn(Rc); std::vector<std::array<double, 3>> tn(tc), Xn(Xp);
            for (int f = 0; f < nf; ++f) {
                double dR[9], R2[9]; rodrigues(&g[(size_t)(6 * f)], dR); m3_mul(dR, Rc[(size_t)f].data(), R2); memcpy(Rn[(size_t)f].data(), R2, 72);
                for (int k = 0; k < 3; ++k) tn[(size_t)f][(size_t)k] = tc[(size_t)f][(size_t)k] + g[(size_t)(6 * f + 3 + k)];
            }
            // back-substitution for the points: dx = Hpp^-1 (bp - sum_obs W^T dc)
            std::vector<double> rhs(bp);
            for (auto& W : Ws) for (int c = 0; c < 3; ++c) { double s = 0.; for (int a = 0; a < 6; ++a) s += W.w[a * 3 + c] * g[(size_t)(6 * W.cam + a)]; rhs[(size_t)(3 * W.pt + c)] -= s; }
            for (int a = 0; a < np; ++a) for (int c = 0; c < 3; ++c) { double s = 0.; for (int k = 0; k < 3; ++k) s += Hpi[(size_t)a * 9 + c * 3 + k] * rhs[(size_t)(3 * a + k)]; Xn[(size_t)a][(size_t)c] = Xp[(size_t)a][(size_t)c] + s; }
            const double c1 = total_cost(Rn, tn, Xn);
#ifdef LVK_INIT_DEBUG
            fprintf(stderr, "[init] BA it %d try %d lambda %.2e cost %.6e -> %.6e\n", it, tries, lambda, c0, c1);
#endif
            if (std::isfinite(c1) && c1 < c0) {
                converged = (c0 - c1) <= 1e-13 * c0;      // (Ceres stops at 1e-6 or after 0.2 s, initial_sfm.cpp:288-290; this runs to the minimum - a cold path, and a defined target)
                if ((c0 - c1) <= 1e-6 * c0) ceres_stop = true;
                Rc.swap(Rn); tc.swap(tn); Xp.swap(Xn); c0 = c1; lambda = std::max(lambda / 3., 1e-12); improved = true;
            } else lambda *= 4.;
        }
        if (!improved) { converged = true; break; }       // no step lowers the cost any more: a minimum to working precision
    }
    for (int a = 0; a < np; ++a) memcpy(feats[(size_t)pidx[(size_t)a]].position, Xp[(size_t)a].data(), 24);
    return converged || ceres_stop || c0 < 5e-3;
}

// GlobalSFM::construct (initial_sfm.cpp:131-330).  q[i] (as a matrix here) / T[i]: pose of camera i in the frame of camera l, up to scale
static inline bool global_sfm(int nf, int l, const double* relR, const double* relT, std::vector<SfmFeature>& feats,
                              std::vector<std::array<double, 9>>& Rwc, std::vector<std::array<double, 3>>& Twc)
{
    std::vector<std::array<double, 9>> cR((size_t)nf); std::vector<std::array<double, 3>> cT((size_t)nf);     // cam-from-reference
    std::vector<std::array<double, 12>> Pose((size_t)nf); std::vector<char> have((size_t)nf, 0);
    auto set_pose = [&](int i) { for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Pose[(size_t)i][(size_t)(r * 4 + c)] = cR[(size_t)i][(size_t)(r * 3 + c)]; Pose[(size_t)i][(size_t)(r * 4 + 3)] = cT[(size_t)i][(size_t)r]; } have[(size_t)i] = 1; };
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(cR[(size_t)l].data(), I3, 72); cT[(size_t)l] = {0, 0, 0}; set_pose(l);
    {   // q[n-1] = relative_R, T[n-1] = relative_T  ->  camera-from-reference: R^T, -R^T T
        double Rt[9], t[3]; m3_t(relR, Rt); m3_v(Rt, relT, t);
        memcpy(cR[(size_t)(nf - 1)].data(), Rt, 72); cT[(size_t)(nf - 1)] = {-t[0], -t[1], -t[2]}; set_pose(nf - 1);
    }
    auto tri_two = [&](int f0, int f1) {                  // triangulateTwoFrames (:91-128)
        for (auto& sf : feats) {
            if (sf.state) continue;
            bool h0 = false, h1 = false; Pt2 p0{0, 0}, p1{0, 0};
            for (auto& ob : sf.obs) { if (ob.first == f0) { p0 = ob.second; h0 = true; } if (ob.first == f1) { p1 = ob.second; h1 = true; } }
            if (h0 && h1) { double X[4]; triangulate_dlt(Pose[(size_t)f0].data(), Pose[(size_t)f1].data(), p0, p1, X); sf.state = true; for (int k = 0; k < 3; ++k) sf.position[k] = X[k] / X[3]; }
        }
    };
    auto pnp = [&](int i, const double* Rinit, const double* tinit) -> bool {      // solveFrameByPnP (:32-88)
        std::vector<std::array<double, 3>> X; std::vector<Pt2> z;
        for (auto& sf : feats) { if (!sf.state) continue; for (auto& ob : sf.obs) if (ob.first == i) { z.push_back(ob.second); X.push_back({sf.position[0], sf.position[1], sf.position[2]}); break; } }
        if ((int)z.size() < 10) return false;            // (< 15 only warns in the reference)
        double R[9], t[3]; memcpy(R, Rinit, 72); memcpy(t, tinit, 24);
        // cv::solvePnP takes float points (cv::Point2f / Point3f): the reference's inputs are rounded to float there
        for (auto& x : X) for (auto& v : x) v = (double)(float)v;
        for (auto& p : z) { p.x = (double)(float)p.x; p.y = (double)(float)p.y; }
        if (!solve_pnp(X, z, R, t)) return false;
        memcpy(cR[(size_t)i].data(), R, 72); memcpy(cT[(size_t)i].data(), t, 24); set_pose(i);
        return true;
    };
    for (int i = l; i < nf - 1; ++i) {
        if (i > l && !pnp(i, cR[(size_t)(i - 1)].data(), cT[(size_t)(i - 1)].data())) return false;
        tri_two(i, nf - 1);
    }
    for (int i = l + 1; i < nf - 1; ++i) tri_two(l, i);
    for (int i = l - 1; i >= 0; --i) {
        if (!pnp(i, cR[(size_t)(i + 1)].data(), cT[(size_t)(i + 1)].data())) return false;
        tri_two(i, l);
    }
    for (auto& sf : feats) {
        if (sf.state || sf.obs.size() < 2) continue;
        double X[4]; triangulate_dlt(Pose[(size_t)sf.obs.front().first].data(), Pose[(size_t)sf.obs.back().first].data(), sf.obs.front().second, sf.obs.back().second, X);
        sf.state = true; for (int k = 0; k < 3; ++k) sf.position[k] = X[k] / X[3];
    }
    if (!bundle_adjust(nf, l, cR, cT, feats)) return false;
    Rwc.assign((size_t)nf, {}); Twc.assign((size_t)nf, {});
    for (int i = 0; i < nf; ++i) { double Rt[9], t[3]; m3_t(cR[(size_t)i].data(), Rt); m3_v(Rt, cT[(size_t)i].data(), t); memcpy(Rwc[(size_t)i].data(), Rt, 72); Twc[(size_t)i] = {-t[0], -t[1], -t[2]}; }
    return true;
}

// ------------------------------------------------------------------------------------------------ visual-inertial alignment
// (initial_alignment.cpp).  frames: the window's frames in time order, R = body-to-reference-camera rotation, T = camera position in the
// reference camera's frame (up to scale); frames[1..] carry the pre-integration from their predecessor.
static inline void solve_gyro_bias(std::vector<Frame*>& fr, double (*Bgs)[3])
{   // :12-46
    double A[9] = {0}, b[3] = {0};
    for (size_t i = 0; i + 1 < fr.size(); ++i) {
        Frame *fi = fr[i], *fj = fr[i + 1];
        double Rit[9], Rij[9], qij[4], dqi[4], qe[4];
        m3_t(fi->R, Rit); m3_mul(Rit, fj->R, Rij); rot_to_quat(Rij, qij);
        dqi[0] = -fj->pre.dq[0]; dqi[1] = -fj->pre.dq[1]; dqi[2] = -fj->pre.dq[2]; dqi[3] = fj->pre.dq[3];
        quat_mul(dqi, qij, qe);
        const double* J = fj->pre.J_R_bg; const double tb[3] = {2 * qe[0], 2 * qe[1], 2 * qe[2]};
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) { double s = 0.; for (int k = 0; k < 3; ++k) s += J[k * 3 + r] * J[k * 3 + c]; A[r * 3 + c] += s; }
                                      double s = 0.; for (int k = 0; k < 3; ++k) s += J[k * 3 + r] * tb[k]; b[r] += s; }
    }
    sym_solve(3, A, b);
#ifdef LVK_INIT_DEBUG
    fprintf(stderr, "[init] delta bg = %.5f %.5f %.5f\n", b[0], b[1], b[2]);
#endif
    for (int i = 0; i <= WIN; ++i) for (int k = 0; k < 3; ++k) Bgs[i][k] += b[k];
    const double zero[3] = {0, 0, 0};
    for (size_t i = 1; i < fr.size(); ++i) fr[i]->pre.repropagate(zero, Bgs[0]);
}
static inline void tangent_basis(const double* g0, double* b, double* c)
{   // :49-62
    double a[3]; const double n = v3_norm(g0); for (int i = 0; i < 3; ++i) a[i] = g0[i] / n;
    double tmp[3] = {0, 0, 1}; if (a[0] == 0 && a[1] == 0 && a[2] == 1) { tmp[0] = 1; tmp[2] = 0; }
    const double d = a[0] * tmp[0] + a[1] * tmp[1] + a[2] * tmp[2];
    for (int i = 0; i < 3; ++i) b[i] = tmp[i] - a[i] * d;
    const double nb = v3_norm(b); for (int i = 0; i < 3; ++i) b[i] /= nb;
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
// one pass of the normal equations shared by LinearAlignment (ng = 3: free gravity) and RefineGravity (ng = 2: on the tangent plane of g0)
static inline void alignment_system(std::vector<Frame*>& fr, const double* TIC, int ng, const double* g0, const double* lxly /* 3 x 2 or null */, std::vector<double>& A, std::vector<double>& b)
{
    const int nfr = (int)fr.size(), n = nfr * 3 + ng + 1, w = 6 + ng + 1;
    A.assign((size_t)n * n, 0.); b.assign((size_t)n, 0.);
    for (int i = 0; i + 1 < nfr; ++i) {
        Frame *fi = fr[(size_t)i], *fj = fr[(size_t)i + 1];
        const double dt = fj->pre.sum_dt;
        double tA[6 * 10], tb[6]; memset(tA, 0, sizeof tA); memset(tb, 0, sizeof tb);
        double Rit[9]; m3_t(fi->R, Rit);
        double RiRj[9]; m3_mul(Rit, fj->R, RiRj);
        for (int r = 0; r < 3; ++r) {
            tA[r * w + r] = -dt;
            tA[(3 + r) * w + r] = -1.; for (int c = 0; c < 3; ++c) tA[(3 + r) * w + 3 + c] = RiRj[r * 3 + c];
            for (int c = 0; c < ng; ++c) {
                double s = 0.;
                if (ng == 3) s = Rit[r * 3 + c]; else for (int k = 0; k < 3; ++k) s += Rit[r * 3 + k] * lxly[k * 2 + c];
                tA[r * w + 6 + c] = s * dt * dt / 2; tA[(3 + r) * w + 6 + c] = s * dt;
            }
            double dT[3] = {fj->T[0] - fi->T[0], fj->T[1] - fi->T[1], fj->T[2] - fi->T[2]}, v[3]; m3_v(Rit, dT, v);
            tA[r * w + 6 + ng] = v[r] / 100.0;
        }
        double RT[3], Rg[3] = {0, 0, 0}; m3_v(RiRj, TIC, RT);
        if (ng == 2) m3_v(Rit, g0, Rg);
        for (int r = 0; r < 3; ++r) { tb[r] = fj->pre.dp[r] + RT[r] - TIC[r] - Rg[r] * dt * dt / 2; tb[3 + r] = fj->pre.dv[r] - Rg[r] * dt; }
        // r_A = tmp_A^T tmp_A scattered: the two velocity blocks at 3 i, the (gravity, scale) block at the end
        auto col = [&](int c) { return c < 6 ? i * 3 + c : n - (ng + 1) + (c - 6); };
        for (int a = 0; a < w; ++a) {
            for (int c = 0; c < w; ++c) { double s = 0.; for (int k = 0; k < 6; ++k) s += tA[k * w + a] * tA[k * w + c]; A[(size_t)col(a) * n + col(c)] += s; }
            double s = 0.; for (int k = 0; k < 6; ++k) s += tA[k * w + a] * tb[k]; b[(size_t)col(a)] += s;
        }
    }
}
static inline bool visual_imu_alignment(std::vector<Frame*>& fr, double (*Bgs)[3], const double* TIC, double* g, std::vector<double>& x)
{   // VisualIMUAlignment (:204-212): solveGyroscopeBias, LinearAlignment (:130-201), RefineGravity (:65-127)
    solve_gyro_bias(fr, Bgs);
    const int nfr = (int)fr.size();
    std::vector<double> A, b;
    alignment_system(fr, TIC, 3, nullptr, nullptr, A, b);
    int n = nfr * 3 + 4;
    for (auto& v : A) v *= 1000.0;
    for (auto& v : b) v *= 1000.0;
    sym_solve(n, A.data(), b.data());
    double s = b[(size_t)n - 1] / 100.0;
    for (int k = 0; k < 3; ++k) g[k] = b[(size_t)(n - 4 + k)];
#ifdef LVK_INIT_DEBUG
    fprintf(stderr, "[init] linear alignment: g = %.4f %.4f %.4f (|g| %.4f) s = %.5f\n", g[0], g[1], g[2], v3_norm(g), s);
#endif
    if (fabs(v3_norm(g) - GRAV_NORM) > 1.0 || s < 0) return false;
    double g0[3]; { const double ng = v3_norm(g); for (int k = 0; k < 3; ++k) g0[k] = g[k] / ng * GRAV_NORM; }
    n = nfr * 3 + 3;
    // RefineGravity (:65-127) declares its normal equations OUTSIDE the four passes and never clears them: pass k solves
    // 1000 (A_{k-1} + fresh_k), i.e. the first pass's system (in the first pass's tangent basis) outweighs every later one a
    // thousandfold.  Kept as the reference has it (so is VINS-Mono's original).
    std::vector<double> Aacc((size_t)n * n, 0.), bacc((size_t)n, 0.);
    for (int k = 0; k < 4; ++k) {
        double bb[3], cc[3]; tangent_basis(g0, bb, cc);
        const double lxly[6] = {bb[0], cc[0], bb[1], cc[1], bb[2], cc[2]};
        alignment_system(fr, TIC, 2, g0, lxly, A, b);
        for (size_t q = 0; q < Aacc.size(); ++q) Aacc[q] = (Aacc[q] + A[q]) * 1000.0;
        for (size_t q = 0; q < bacc.size(); ++q) bacc[q] = (bacc[q] + b[q]) * 1000.0;
        A = Aacc; b = bacc;
        sym_solve(n, A.data(), b.data());
        double gn[3]; for (int r = 0; r < 3; ++r) gn[r] = g0[r] + lxly[r * 2] * b[(size_t)n - 3] + lxly[r * 2 + 1] * b[(size_t)n - 2];
        const double nn = v3_norm(gn); for (int r = 0; r < 3; ++r) g0[r] = gn[r] / nn * GRAV_NORM;
    }
    memcpy(g, g0, 24);
    x.assign(b.begin(), b.begin() + n);
    s = x[(size_t)n - 1] / 100.0; x[(size_t)n - 1] = s;
    return !(s < 0.0);
}

// ------------------------------------------------------------------------------------------------ the initialiser object
struct Result { double state_time, q[4], p[3], v[3], bg[3], ba[3], last_gyro[3], last_acc[3]; };
// intermediate results of the successful attempt (read by the replay harness of the CPU suite, tests/host/init_replay.hip)
struct Diag { int l = -1; double relR[9], relT[3]; std::vector<std::array<double, 9>> sfm_R; std::vector<std::array<double, 3>> sfm_T; double g[3] = {0, 0, 0}, scale = 0; int n_points = 0; int attempts = 0; };   // attempts: initial_structure() calls so far (full windows tried)
// callback for E = cv::findFundamentalMat(ll, rr, FM_RANSAC, thresh, conf, mask): float correspondences -> inlier mask and the matrix (9 doubles;
// false / all zeros = OpenCV's empty Mat)
typedef bool (*ransac_fn)(void* user, const std::vector<Pt2>& ll, const std::vector<Pt2>& rr, double thresh, double conf, std::vector<unsigned char>& mask, double* F);

struct DynInit {
    // configuration (DynamicInitializer.h:40-75)
    double td = 0, imu_img_time_th = 0, RIC[9], TIC[3], Ma[9], Tg[9], As[9];
    ransac_fn ransac = nullptr; void* ransac_user = nullptr;
    // state
    bool inited = false, first_imu = false; int frame_count = 0;
    double lower_time_bound = 0, ddt = 0, curr_time = -1, initial_timestamp = 0, acc_0[3] = {0, 0, 0}, gyr_0[3] = {0, 0, 0}, last_gyro[3] = {0, 0, 0}, last_acc[3] = {0, 0, 0};
    double Rs[WIN + 1][9], Ps[WIN + 1][3], Vs[WIN + 1][3], Bas[WIN + 1][3], Bgs[WIN + 1][3], Times[WIN + 1];
    PreInt pre[WIN + 1]; bool have_pre[WIN + 1]; PreInt tmp_pre; bool have_tmp = false;
    std::map<double, Frame> frames;                      // all_image_frame
    std::vector<FeatTrack> tracks;                       // FeatureManager::feature (std::list there; erase order is what matters, not the container)
    double g[3] = {0, 0, 0};
    Result out; Diag diag;

    void reset()
    {
        inited = false; first_imu = false; frame_count = 0; lower_time_bound = 0; ddt = 0; curr_time = -1; initial_timestamp = 0;
        for (int i = 0; i <= WIN; ++i) { for (int k = 0; k < 9; ++k) Rs[i][k] = (k % 4 == 0) ? 1. : 0.; for (int k = 0; k < 3; ++k) { Ps[i][k] = Vs[i][k] = Bas[i][k] = Bgs[i][k] = 0.; } have_pre[i] = false; Times[i] = 0; }
        have_tmp = false; frames.clear(); tracks.clear();
    }
    void process_imu(double t, const double* gyro_m, const double* acc_m)
    {   // DynamicInitializer.cpp:47-97
        double la[3], t3[3], w[3], ga[3];
        m3_v(Ma, acc_m, la); m3_v(As, la, t3); for (int k = 0; k < 3; ++k) w[k] = gyro_m[k] - t3[k]; m3_v(Tg, w, ga);
        if (!first_imu) { first_imu = true; memcpy(acc_0, la, 24); memcpy(gyr_0, ga, 24); curr_time = t; }
        const double dt = t - curr_time;
        if (!have_pre[frame_count]) { pre[frame_count].start(acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]); have_pre[frame_count] = true; }
        if (!have_tmp) { tmp_pre.start(acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]); have_tmp = true; }      // (the reference creates it after the first image; nothing is pushed before)
        if (frame_count != 0) {
            pre[frame_count].push_back(dt, la, ga); tmp_pre.push_back(dt, la, ga);
            const int j = frame_count;
            const double grav[3] = {g[0], g[1], g[2]};
            double a0[3], un0[3], ug[3], un1[3];
            for (int k = 0; k < 3; ++k) a0[k] = acc_0[k] - Bas[j][k];
            m3_v(Rs[j], a0, un0); for (int k = 0; k < 3; ++k) un0[k] -= grav[k];
            for (int k = 0; k < 3; ++k) ug[k] = (0.5 * (gyr_0[k] + ga[k]) - Bgs[j][k]) * dt;
            double dqv[4], dR[9], Rn[9]; small_angle_quat(ug, dqv); quat_to_rot(dqv, dR); m3_mul(Rs[j], dR, Rn); memcpy(Rs[j], Rn, 72);
            for (int k = 0; k < 3; ++k) a0[k] = la[k] - Bas[j][k];
            m3_v(Rs[j], a0, un1); for (int k = 0; k < 3; ++k) un1[k] -= grav[k];
            for (int k = 0; k < 3; ++k) { const double ua = 0.5 * (un0[k] + un1[k]); Ps[j][k] += dt * Vs[j][k] + 0.5 * dt * dt * ua; Vs[j][k] += dt * ua; }
        }
        memcpy(acc_0, la, 24); memcpy(gyr_0, ga, 24); curr_time = t; memcpy(last_gyro, ga, 24); memcpy(last_acc, la, 24);
    }
    // FeatureManager::addFeatureCheckParallax (feature_manager.cpp:44-96).  MIN_PARALLAX = 10/460 is an INTEGER division in the reference
    // (feature_manager.h:25): zero, so the function returns true on every path and the window always drops its OLDEST frame.
    void add_features(const lvk_feature_obs* f, int n, double tdd)
    {
        for (int i = 0; i < n; ++i) {
            const long long id = (long long)(int)f[i].id;        // (feature_id is an int in the reference: feature_manager.h:57)
            const Pt2 p{f[i].u + f[i].u_vel * tdd, f[i].v + f[i].v_vel * tdd};
            auto it = std::find_if(tracks.begin(), tracks.end(), [id](const FeatTrack& tr) { return tr.id == id; });
            if (it == tracks.end()) { FeatTrack tr; tr.id = id; tr.start_frame = frame_count; tr.per_frame.push_back(p); tracks.push_back(tr); }
            else it->per_frame.push_back(p);
        }
    }
    void corresponding(int fl, int fr_, std::vector<Pt2>& a, std::vector<Pt2>& b) const
    {   // getCorresponding (:99-120)
        a.clear(); b.clear();
        for (auto& tr : tracks) if (tr.start_frame <= fl && tr.end_frame() >= fr_) { a.push_back(tr.per_frame[(size_t)(fl - tr.start_frame)]); b.push_back(tr.per_frame[(size_t)(fr_ - tr.start_frame)]); }
    }
    bool solve_relative_rt(const std::vector<Pt2>& a, const std::vector<Pt2>& b, double* Rot, double* Trans) const
    {   // MotionEstimator::solveRelativeRT (solve_5pts.cpp:193-230)
        if (a.size() < 15) return false;
        std::vector<Pt2> ll(a), rr(b);
        for (auto& p : ll) { p.x = (double)(float)p.x; p.y = (double)(float)p.y; }      // cv::Point2f
        for (auto& p : rr) { p.x = (double)(float)p.x; p.y = (double)(float)p.y; }
        std::vector<unsigned char> mask;
        double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (!ransac || !ransac(ransac_user, ll, rr, 0.3 / 460, 0.99, mask, E) || mask.size() != ll.size()) return false;
        bool any = false; for (double v : E) any = any || v != 0.;
        if (!any) return false;                          // (the reference would hand an empty Mat to recoverPose: an OpenCV assertion)
        double R[9], t[3];
        const int inl = recover_pose(E, ll, rr, mask, R, t);
        double Rt[9], v[3]; m3_t(R, Rt); m3_v(Rt, t, v);
        memcpy(Rot, Rt, 72); for (int k = 0; k < 3; ++k) Trans[k] = -v[k];
        return inl > 12;
    }
    bool relative_pose(double* relR, double* relT, int* l) const
    {   // DynamicInitializer.cpp:330-359
        std::vector<Pt2> a, b;
        for (int i = 0; i < WIN; ++i) {
            corresponding(i, WIN, a, b);
            if (a.size() > 20) {
                double sum = 0.; for (size_t j = 0; j < a.size(); ++j) sum += sqrt((a[j].x - b[j].x) * (a[j].x - b[j].x) + (a[j].y - b[j].y) * (a[j].y - b[j].y));
                const double avg = sum / (double)(int)a.size();
                if (avg * 460 > 30 && solve_relative_rt(a, b, relR, relT)) { *l = i; return true; }
            }
        }
        return false;
    }
    bool initial_structure()
    {   // :136-275
        std::vector<SfmFeature> sfm;
        for (auto& tr : tracks) { SfmFeature sf; sf.id = tr.id; int j = tr.start_frame - 1; for (auto& p : tr.per_frame) sf.obs.push_back({++j, p}); sfm.push_back(sf); }
        double relR[9], relT[3]; int l = 0;
        diag.attempts += 1;
        if (!relative_pose(relR, relT, &l)) return false;
        std::vector<std::array<double, 9>> Q; std::vector<std::array<double, 3>> T;
        if (!global_sfm(frame_count + 1, l, relR, relT, sfm, Q, T)) return false;      // (marginalization_flag = MARGIN_OLD: it always is, see add_features)
        diag.l = l; memcpy(diag.relR, relR, 72); memcpy(diag.relT, relT, 24); diag.sfm_R = Q; diag.sfm_T = T;
        diag.n_points = 0; for (auto& sf : sfm) diag.n_points += sf.state ? 1 : 0;
#ifdef LVK_INIT_DEBUG
        fprintf(stderr, "[init] l = %d relT = %.4f %.4f %.4f\n", l, relT[0], relT[1], relT[2]);
        for (int i = 0; i <= frame_count; ++i) fprintf(stderr, "[init] frame %d T = %.4f %.4f %.4f  R row0 = %.4f %.4f %.4f\n", i, T[(size_t)i][0], T[(size_t)i][1], T[(size_t)i][2], Q[(size_t)i][0], Q[(size_t)i][1], Q[(size_t)i][2]);
#endif
        // every frame of all_image_frame is a window frame here (the map is trimmed to the window in slide_window), so each takes the
        // structure-from-motion pose: R = Q[i] RIC^T, T = T[i]  (:203-209; the PnP branch for frames between key frames cannot occur)
        int i = 0;
        double RICt[9]; m3_t(RIC, RICt);
        for (auto& kv : frames) {
            if (i > frame_count || kv.first != Times[i] + td) return false;
            kv.second.key = true; m3_mul(Q[(size_t)i].data(), RICt, kv.second.R); memcpy(kv.second.T, T[(size_t)i].data(), 24); ++i;
        }
        return visual_initial_align();
    }
    bool visual_initial_align()
    {   // :278-327
        std::vector<Frame*> fr; for (auto& kv : frames) fr.push_back(&kv.second);
        std::vector<double> x;
        if (!visual_imu_alignment(fr, Bgs, TIC, g, x)) return false;
        memcpy(diag.g, g, 24); diag.scale = x.back();
        for (int i = 0; i <= frame_count; ++i) { Frame& f = frames[Times[i] + td]; memcpy(Ps[i], f.T, 24); memcpy(Rs[i], f.R, 72); f.key = true; }
        int kv = -1;
        for (auto& f : frames) if (f.second.key) { ++kv; m3_v(f.second.R, &x[(size_t)kv * 3], Vs[kv]); }
        const double gw[3] = {0, 0, v3_norm(g)};
        double Rc0w[9], Rbl[9]; rot_from_two_vectors(g, gw, Rc0w); m3_mul(Rc0w, Rs[frame_count], Rbl);
        rot_to_quat(Rbl, out.q);
        out.state_time = Times[frame_count] + td + ddt;
        for (int k = 0; k < 3; ++k) { out.p[k] = 0; out.ba[k] = 0; out.bg[k] = Bgs[frame_count][k]; }
        m3_v(Rc0w, Vs[frame_count], out.v);
        return true;
    }
    void slide_window()
    {   // :362-440, MARGIN_OLD branch (the only one that can occur)
        if (frame_count != WIN) return;
        for (int i = 0; i < WIN; ++i) {
            std::swap_ranges(Rs[i], Rs[i] + 9, Rs[i + 1]); std::swap(pre[i], pre[i + 1]); std::swap(have_pre[i], have_pre[i + 1]);
            Times[i] = Times[i + 1];
            std::swap_ranges(Ps[i], Ps[i] + 3, Ps[i + 1]); std::swap_ranges(Vs[i], Vs[i] + 3, Vs[i + 1]);
            std::swap_ranges(Bas[i], Bas[i] + 3, Bas[i + 1]); std::swap_ranges(Bgs[i], Bgs[i] + 3, Bgs[i + 1]);
        }
        Times[WIN] = Times[WIN - 1];
        memcpy(Ps[WIN], Ps[WIN - 1], 24); memcpy(Vs[WIN], Vs[WIN - 1], 24); memcpy(Rs[WIN], Rs[WIN - 1], 72); memcpy(Bas[WIN], Bas[WIN - 1], 24); memcpy(Bgs[WIN], Bgs[WIN - 1], 24);
        pre[WIN].start(acc_0, gyr_0, Bas[WIN], Bgs[WIN]); have_pre[WIN] = true;
        auto it0 = frames.find(Times[0] + td);
        if (it0 != frames.end()) { it0->second.has_pre = false; frames.erase(frames.begin(), it0); }
        // FeatureManager::removeBack (feature_manager.cpp:205-222)
        for (size_t k = 0; k < tracks.size();) {
            FeatTrack& tr = tracks[k];
            if (tr.start_frame != 0) { tr.start_frame--; ++k; }
            else { tr.per_frame.erase(tr.per_frame.begin()); if (tr.per_frame.empty()) tracks.erase(tracks.begin() + (long)k); else ++k; }
        }
    }
    void process_image(double ts, const lvk_feature_obs* f, int n)
    {   // :100-133
        add_features(f, n, td + ddt);
        Times[frame_count] = ts;
        Frame fr; fr.t = ts + td + ddt;
        for (int i = 0; i < n; ++i) fr.points[(long long)(int)f[i].id] = Pt2{f[i].u + f[i].u_vel * (td + ddt), f[i].v + f[i].v_vel * (td + ddt)};
        if (have_tmp) { fr.pre = tmp_pre; fr.has_pre = true; }
        frames[ts + td] = fr;
        tmp_pre.start(acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]); have_tmp = true;
        if (frame_count == WIN) {
            bool ok = false;
            if (ts - initial_timestamp > 0.1) { ok = initial_structure(); initial_timestamp = ts; }
            if (ok) inited = true; else slide_window();
        } else frame_count++;
    }
    // tryDynInit (:23-44) + assignInitialState (:443-479).  *n_erase = IMU samples the caller erases when this returns true
    bool try_init(double ts, const lvk_feature_obs* f, int n, const lvk_imu* imu, int n_imu, int* n_erase)
    {
        *n_erase = 0;
        const double time_bound = ts + td;
        for (int i = 0; i < n_imu; ++i) {
            if (imu[i].t <= lower_time_bound) continue;
            if (imu[i].t - time_bound > imu_img_time_th) break;
            ddt = imu[i].t - time_bound;
            process_imu(imu[i].t, imu[i].gyro, imu[i].acc);
        }
        lower_time_bound = time_bound + imu_img_time_th;
        process_image(ts, f, n);
        if (!inited) return false;
        int useful = 0; for (int i = 0; i < n_imu; ++i) { if (imu[i].t > out.state_time) break; ++useful; }
        *n_erase = useful;
        memcpy(out.last_gyro, last_gyro, 24); memcpy(out.last_acc, last_acc, 24);
        return true;
    }
};

}  // namespace lvk_init
