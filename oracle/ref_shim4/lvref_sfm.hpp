// lvref_sfm.hpp - TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  The OpenCV names /root/reference/src/DynamicInitializer.cpp and
// src/initial_sfm.cpp use (cv::Mat of doubles, Mat_<double> <<, Point2f / Point3f, eigen2cv / cv2eigen, Rodrigues, solvePnP), so that
// those files can be compiled where they lie (oracle/Makefile, target `ref` -> oracle/_ref/liblvref_dyninit.so).  OpenCV is not
// installed.  solvePnP here is a Levenberg-Marquardt on the reprojection error in (rotation vector, translation) from the caller's
// guess, run to convergence (OpenCV's iterative method minimises the same cost, at most 20 iterations to FLT_EPSILON); Rodrigues is
// the exponential / logarithm of SO(3) in double.
#pragma once
#include "../ref_shim/lvref_cv.hpp"
#include "../ref_shim2/lvref_eigen2.hpp"
#include "lvref_cvalg.hpp"
#include <vector>
#include <cmath>
namespace cv {
enum { FM_7POINT = 1, FM_8POINT = 2, FM_LMEDS = 4, FM_RANSAC = 8, RANSAC = 8, LMEDS = 4 };
template <typename T> struct Point3_ { T x, y, z; Point3_() : x(0), y(0), z(0) {} Point3_(T a, T b, T c) : x(a), y(b), z(c) {} };
typedef Point3_<float> Point3f; typedef Point3_<double> Point3d;
template <typename T> class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, CV_64F) { static_assert(sizeof(T) == 8, "Mat_<double> only"); }
    struct Comma { Mat_& m; int n; Comma& operator,(T v) { m.template at<T>(n / m.cols, n % m.cols) = v; ++n; return *this; } operator Mat() const { return m; } };
    Comma operator<<(T v) { this->template at<T>(0, 0) = v; return Comma{*this, 1}; }
};
inline void eigen2cv(const Eigen::XMat& s, Mat& d) { d = Mat(s.rows(), s.cols(), CV_64F); for (int i = 0; i < s.rows(); ++i) for (int j = 0; j < s.cols(); ++j) d.at<double>(i, j) = s(i, j); }
inline void cv2eigen(const Mat& s, Eigen::XMat& d) { d = Eigen::XMat(s.rows, s.cols); for (int i = 0; i < s.rows; ++i) for (int j = 0; j < s.cols; ++j) d(i, j) = s.at<double>(i, j); }

inline void lv_exp_so3(const double r[3], double R[9])
{
    const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 1e-300) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0); return; }
    const double c = std::cos(th), s = std::sin(th), c1 = 1 - c, x = r[0] / th, y = r[1] / th, z = r[2] / th;
    const double K[9] = {0, -z, y, z, 0, -x, -y, x, 0}, kk[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    for (int i = 0; i < 9; ++i) R[i] = c * (i % 4 == 0) + c1 * kk[i] + s * K[i];
}
inline void lv_log_so3(const double R[9], double r[3])
{   // through the unit quaternion (robust near 0 and near pi)
    double q[4]; const double tr = R[0] + R[4] + R[8];
    if (tr > 0) { double t = std::sqrt(tr + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t; }
    else { int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2; const int j = (i + 1) % 3, k = (j + 1) % 3;
           double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t; q[3] = (R[3 * k + j] - R[3 * j + k]) * t; q[j] = (R[3 * j + i] + R[3 * i + j]) * t; q[k] = (R[3 * k + i] + R[3 * i + k]) * t; }
    if (q[3] < 0) for (double& v : q) v = -v;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const double ang = 2.0 * std::atan2(n, q[3]), s = n > 1e-300 ? ang / n : 2.0;
    for (int i = 0; i < 3; ++i) r[i] = q[i] * s;
}
inline void Rodrigues(const Mat& src, Mat& dst)
{
    if (src.rows == 3 && src.cols == 3) { double R[9], r[3]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = src.at<double>(i, j); lv_log_so3(R, r); dst = Mat(3, 1, CV_64F); for (int i = 0; i < 3; ++i) dst.at<double>(i, 0) = r[i]; }
    else { const double r[3] = {src.at<double>(0, 0), src.rows == 3 ? src.at<double>(1, 0) : src.at<double>(0, 1), src.rows == 3 ? src.at<double>(2, 0) : src.at<double>(0, 2)}; double R[9]; lv_exp_so3(r, R); dst = Mat(3, 3, CV_64F); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dst.at<double>(i, j) = R[3 * i + j]; }
}
// x_cam = R(rvec) X + t, identity camera matrix (both call sites), no distortion; rvec / tvec in (guess) and out
inline bool solvePnP(const std::vector<Point3f>& obj, const std::vector<Point2f>& img, const Mat& K, const Mat&, Mat& rvec, Mat& tvec, bool useGuess = false, int = 0)
{
    if (obj.size() != img.size() || obj.size() < 4 || !useGuess || rvec.empty() || tvec.empty()) return false;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (K.at<double>(i, j) != (i == j ? 1.0 : 0.0)) return false;
    double p[6] = {rvec.at<double>(0, 0), rvec.at<double>(1, 0), rvec.at<double>(2, 0), tvec.at<double>(0, 0), tvec.at<double>(1, 0), tvec.at<double>(2, 0)};
    const int n = (int)obj.size();
    auto residual = [&](const double* q, std::vector<double>& r) {
        double R[9]; lv_exp_so3(q, R); double c = 0; r.resize((size_t)2 * n);
        for (int k = 0; k < n; ++k) {
            const double X = obj[(size_t)k].x, Y = obj[(size_t)k].y, Z = obj[(size_t)k].z;
            const double x = R[0] * X + R[1] * Y + R[2] * Z + q[3], y = R[3] * X + R[4] * Y + R[5] * Z + q[4], z = R[6] * X + R[7] * Y + R[8] * Z + q[5];
            r[(size_t)2 * k] = x / z - img[(size_t)k].x; r[(size_t)2 * k + 1] = y / z - img[(size_t)k].y; c += r[(size_t)2 * k] * r[(size_t)2 * k] + r[(size_t)2 * k + 1] * r[(size_t)2 * k + 1];
        }
        return c;
    };
    std::vector<double> r0, rp, rm; double cost = residual(p, r0), lambda = 1e-4;
    for (int it = 0; it < 100; ++it) {
        std::vector<double> J((size_t)2 * n * 6);
        for (int j = 0; j < 6; ++j) { double q[6]; for (int i = 0; i < 6; ++i) q[i] = p[i]; const double h = 1e-6; q[j] = p[j] + h; residual(q, rp); q[j] = p[j] - h; residual(q, rm); for (int i = 0; i < 2 * n; ++i) J[(size_t)i * 6 + j] = (rp[(size_t)i] - rm[(size_t)i]) / (2 * h); }
        double H[36] = {0}, g[6] = {0};
        for (int i = 0; i < 2 * n; ++i) for (int a = 0; a < 6; ++a) { g[a] += J[(size_t)i * 6 + a] * r0[(size_t)i]; for (int b = 0; b < 6; ++b) H[a * 6 + b] += J[(size_t)i * 6 + a] * J[(size_t)i * 6 + b]; }
        bool stepped = false;
        for (int tries = 0; tries < 12 && !stepped; ++tries) {
            Eigen::XMat A(6, 6), b(6, 1);
            for (int a = 0; a < 6; ++a) { b(a) = -g[a]; for (int c = 0; c < 6; ++c) A(a, c) = H[a * 6 + c] + (a == c ? lambda * std::max(H[a * 6 + a], 1e-12) : 0.0); }
            Eigen::XMat d = A.ldlt().solve(b);
            double q[6]; for (int i = 0; i < 6; ++i) q[i] = p[i] + d(i);
            std::vector<double> r1; const double c1 = residual(q, r1);
            if (c1 <= cost) { double dn = 0; for (int i = 0; i < 6; ++i) { dn = std::max(dn, std::fabs(d(i))); p[i] = q[i]; } const bool tiny = dn < 1e-14 || cost - c1 <= 1e-18 * std::max(cost, 1e-30); cost = c1; r0 = r1; lambda = std::max(lambda * 0.3, 1e-12); stepped = true; if (tiny) it = 100; }
            else lambda *= 10;
        }
        if (!stepped) break;
    }
    rvec = Mat(3, 1, CV_64F); tvec = Mat(3, 1, CV_64F);
    for (int i = 0; i < 3; ++i) { rvec.at<double>(i, 0) = p[i]; tvec.at<double>(i, 0) = p[3 + i]; }
    return true;
}
}  // namespace cv
