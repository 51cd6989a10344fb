// lvref_cvalg.hpp - TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  The slice of OpenCV's Mat algebra that /root/reference/src/
// solve_5pts.cpp is written in - that file carries OpenCV's own decomposeEssentialMat and recoverPose as an excerpt, plus
// MotionEstimator::solveRelativeRT - so that it, too, can be compiled where it lies (oracle/_ref/liblvref_dyninit.so).  Eager: every
// operator evaluates its operands into a MatExpr (a Mat with expression-assignment semantics: assigning it to a view writes THROUGH the
// view, which is what `P1.col(3) = t * 1.0` relies on).  Library calls the excerpt makes are served here: SVD::compute by a one-sided
// Jacobi SVD, triangulatePoints by the per-point DLT (smallest right singular vector of the 4 x 4 system), findFundamentalMat by the
// oracle's RANSAC restatement (mask and matrix).
#pragma once
#include "../ref_shim/lvref_cv.hpp"
#include "../ref_shim2/lvref_eigen2.hpp"
extern "C" {
#include "../lvo.h"
}
#include <vector>
#define CV_Assert(c) do { if (!(c)) { std::fprintf(stderr, "CV_Assert failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); std::abort(); } } while (0)
namespace cv {
typedef Point_<double> Point2d;
struct MatExpr { Mat m; operator Mat() const { return m; } };
inline Mat::Mat(const MatExpr& e) { *this = e.m; }
inline Mat& Mat::operator=(const MatExpr& e)
{
    if (data && rows == e.m.rows && cols == e.m.cols && type() == e.m.type()) { for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) set(y, x, e.m.get(y, x)); }
    else *this = e.m;
    return *this;
}
inline Mat lv_like(const Mat& a, int type = -1) { return Mat(a.rows, a.cols, type < 0 ? a.type() : type); }
template <typename F> inline MatExpr lv_map(const Mat& a, F f, int type = -1) { Mat o = lv_like(a, type); for (int y = 0; y < a.rows; ++y) for (int x = 0; x < a.cols; ++x) o.set(y, x, f(a.get(y, x))); return MatExpr{o}; }
template <typename F> inline MatExpr lv_zip(const Mat& a, const Mat& b, F f, int type = -1) { CV_Assert(a.rows == b.rows && a.cols == b.cols); Mat o = lv_like(a, type); for (int y = 0; y < a.rows; ++y) for (int x = 0; x < a.cols; ++x) o.set(y, x, f(a.get(y, x), b.get(y, x))); return MatExpr{o}; }
inline MatExpr Mat::t() const { Mat o(cols, rows, type()); for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) o.set(x, y, get(y, x)); return MatExpr{o}; }
inline MatExpr Mat::mul(const Mat& b) const { return lv_zip(*this, b, [](double p, double q) { return p * q; }); }
inline Mat& Mat::operator*=(double a) { for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) set(y, x, get(y, x) * a); return *this; }
inline Mat& Mat::operator/=(const Mat& b) { CV_Assert(rows == b.rows && cols == b.cols); for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) set(y, x, get(y, x) / b.get(y, x)); return *this; }
inline void Mat::convertTo(Mat& dst, int type) const { Mat o(rows, cols, type); for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) o.set(y, x, get(y, x)); dst = o; }
inline MatExpr operator-(const Mat& a, double s) { return lv_map(a, [s](double v) { return v - s; }); }
inline MatExpr operator/(const Mat& a, double s) { return lv_map(a, [s](double v) { return v / s; }); }
inline MatExpr operator*(const Mat& a, double s) { return lv_map(a, [s](double v) { return v * s; }); }
inline MatExpr operator-(const Mat& a) { return lv_map(a, [](double v) { return -v; }); }
inline MatExpr operator*(const Mat& a, const Mat& b)
{
    CV_Assert(a.cols == b.rows); Mat o(a.rows, b.cols, CV_64F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < b.cols; ++j) { double s = 0; for (int k = 0; k < a.cols; ++k) s += a.get(i, k) * b.get(k, j); o.at<double>(i, j) = s; }
    return MatExpr{o};
}
inline MatExpr operator>(const Mat& a, double s) { return lv_map(a, [s](double v) { return v > s ? 255.0 : 0.0; }, CV_8U); }
inline MatExpr operator<(const Mat& a, double s) { return lv_map(a, [s](double v) { return v < s ? 255.0 : 0.0; }, CV_8U); }
inline MatExpr operator&(const Mat& a, const Mat& b) { return lv_zip(a, b, [](double p, double q) { return (double)((int)p & (int)q); }, CV_8U); }
inline void bitwise_and(const Mat& a, const Mat& b, Mat& dst) { Mat r = lv_zip(a, b, [](double p, double q) { return (double)((int)p & (int)q); }, CV_8U); dst = r; }
inline int countNonZero(const Mat& a) { int n = 0; for (int y = 0; y < a.rows; ++y) for (int x = 0; x < a.cols; ++x) n += a.get(y, x) != 0; return n; }
inline double determinant(const Mat& a)
{
    CV_Assert(a.rows == 3 && a.cols == 3);
    return a.get(0, 0) * (a.get(1, 1) * a.get(2, 2) - a.get(1, 2) * a.get(2, 1)) - a.get(0, 1) * (a.get(1, 0) * a.get(2, 2) - a.get(1, 2) * a.get(2, 0)) + a.get(0, 2) * (a.get(1, 0) * a.get(2, 1) - a.get(1, 1) * a.get(2, 0));
}
inline Eigen::XMat lv_to_eigen(const Mat& a) { Eigen::XMat o(a.rows, a.cols); for (int y = 0; y < a.rows; ++y) for (int x = 0; x < a.cols; ++x) o(y, x) = a.get(y, x); return o; }
inline Mat lv_from_eigen(const Eigen::XMat& a) { Mat o(a.rows(), a.cols(), CV_64F); for (int y = 0; y < a.rows(); ++y) for (int x = 0; x < a.cols(); ++x) o.at<double>(y, x) = a(y, x); return o; }
struct SVD {
    static void compute(const Mat& A, Mat& w, Mat& u, Mat& vt)
    {
        Eigen::JacobiSVDx s(lv_to_eigen(A));
        Eigen::XMat U = s.matrixU();
        if (U.cols() == 3 && s.singularValues()(2) <= 1e-14 * std::max(s.singularValues()(0), 1e-300)) {      // a rank-2 essential matrix: complete the third left singular vector
            Eigen::XMat c = U.col(0).eval().cross(U.col(1).eval()); for (int i = 0; i < 3; ++i) U(i, 2) = c(i);
        }
        w = lv_from_eigen(s.singularValues()); u = lv_from_eigen(U); vt = lv_from_eigen(s.matrixV().transpose());
    }
};
// cv::triangulatePoints: projection matrices 3 x 4, points 2 x N (any depth), out 4 x N homogeneous (the DLT's null vector per point)
inline void triangulatePoints(const Mat& P0, const Mat& P1, const Mat& x0, const Mat& x1, Mat& out)
{
    CV_Assert(x0.rows == 2 && x1.rows == 2 && x0.cols == x1.cols);
    const int n = x0.cols; Mat Q(4, n, CV_64F);
    for (int k = 0; k < n; ++k) {
        Eigen::XMat A(4, 4);
        for (int c = 0; c < 4; ++c) {
            A(0, c) = x0.get(0, k) * P0.get(2, c) - P0.get(0, c); A(1, c) = x0.get(1, k) * P0.get(2, c) - P0.get(1, c);
            A(2, c) = x1.get(0, k) * P1.get(2, c) - P1.get(0, c); A(3, c) = x1.get(1, k) * P1.get(2, c) - P1.get(1, c);
        }
        Eigen::XMat v = Eigen::JacobiSVDx(A).matrixV().rightCols(1);
        for (int c = 0; c < 4; ++c) Q.at<double>(c, k) = v(c);
    }
    out = Q;
}
// ---- the array proxies of OpenCV's function signatures
class _InputArray {
protected:
    mutable Mat m_; Mat* ext_ = nullptr;
public:
    _InputArray() {}
    _InputArray(const Mat& m) : m_(m) {}
    _InputArray(const MatExpr& e) : m_(e.m) {}
    _InputArray(const std::vector<Point2f>& v) { m_ = Mat((int)v.size(), 2, CV_32F); for (size_t i = 0; i < v.size(); ++i) { m_.at<float>((int)i, 0) = v[i].x; m_.at<float>((int)i, 1) = v[i].y; } }
    Mat getMat() const { return ext_ ? *ext_ : m_; }
    bool empty() const { return getMat().empty(); }
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& m) { ext_ = &m; }
    bool needed() const { return ext_ != nullptr; }
    void create(int r, int c, int type) const { if (ext_ && !(ext_->rows == r && ext_->cols == c && ext_->type() == type && ext_->data)) *ext_ = Mat(r, c, type); }
    void create(Size s, int type) const { create(s.height, s.width, type); }
    void assign(const Mat& m) const { if (ext_) *ext_ = m.clone(); }
};
typedef const _InputArray& InputArray; typedef const _OutputArray& OutputArray; typedef const _OutputArray& InputOutputArray;
inline void Mat::copyTo(const _OutputArray& o) const { o.assign(*this); }
inline Mat findFundamentalMat(const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, int method, double thresh, double conf, Mat& mask)
{
    CV_Assert(method == 8 && p1.size() == p2.size());
    const int n = (int)p1.size(); std::vector<uint8_t> mk((size_t)n, 0); double F[9] = {0};
    if (!lvo_find_fundamental((const lvo_pt2f*)p1.data(), (const lvo_pt2f*)p2.data(), n, thresh, conf, mk.data(), F)) return Mat();
    mask = Mat(n, 1, CV_8U); for (int i = 0; i < n; ++i) mask.at<uchar>(i, 0) = mk[(size_t)i];
    Mat Fm(3, 3, CV_64F); bool any = false; for (int i = 0; i < 9; ++i) { Fm.at<double>(i / 3, i % 3) = F[i]; any |= F[i] != 0; }
    return any ? Fm : Mat();
}
}  // namespace cv
