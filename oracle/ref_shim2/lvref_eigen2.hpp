// lvref_eigen2.hpp - TEST INFRASTRUCTURE ONLY (see oracle/lvo.h): a second, wider stand-in for the few Eigen TYPES the reference's
// filter (src/larvio.cpp and the headers it includes) is written against, so that those sources can be compiled where they lie under
// /root/reference (Eigen itself is not installed in this image; oracle/Makefile target `ref`).  Everything is one concrete, dynamically
// sized, column-major, EAGER matrix of doubles (`XMat`); `Matrix<double, R, C>` are thin shape-checked derivations of it, blocks are
// writable views (`Blk`) that convert to `XMat`.  No expression templates: every operator evaluates its operands and returns a new XMat,
// which is also what Eigen's aliasing rules make the reference's statements mean.  What this does NOT reproduce is Eigen's rounding
// (operation order inside products, the pivoting of LDLT / inverse, JacobiSVD's U, SPQR's Q): the pinned thing is the reference's
// algorithm text, and the comparisons that use this header state tolerances, not bit-equality.
#ifndef LVREF_EIGEN2_HPP
#define LVREF_EIGEN2_HPP
#include <vector>
#include <cmath>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <algorithm>
#include <memory>
#include <limits>
#include <type_traits>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_DEFINE_STL_VECTOR_SPECIALIZATION(...)
#include <execinfo.h>
#define LVREF_CHECK(c, what) do { if (!(c)) { std::fprintf(stderr, "lvref_eigen2: %s (%s:%d)\n", what, __FILE__, __LINE__); void* bt_[24]; backtrace_symbols_fd(bt_, backtrace(bt_, 24), 2); std::abort(); } } while (0)

namespace Eigen {
enum { Dynamic = -1 };
enum { ComputeFullU = 1, ComputeThinU = 2, ComputeFullV = 4, ComputeThinV = 8 };
enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
typedef long Index;
template <typename T> struct aligned_allocator : std::allocator<T> {
    aligned_allocator() {}
    template <typename U> aligned_allocator(const aligned_allocator<U>&) {}
    template <typename U> struct rebind { typedef aligned_allocator<U> other; };
};

class XMat;
class Blk;
struct LDLTx;

// ------------------------------------------------------------------ the one matrix
class XMat {
public:
    int r_ = 0, c_ = 0;
    std::vector<double> d_;
    XMat() {}
    XMat(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
    XMat(const Blk& b);
    typedef double Scalar;
    int rows() const { return r_; }
    int cols() const { return c_; }
    int size() const { return r_ * c_; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    double& operator()(int i, int j) { LVREF_CHECK(i >= 0 && i < r_ && j >= 0 && j < c_, "index out of range"); return d_[(size_t)j * r_ + i]; }
    double operator()(int i, int j) const { LVREF_CHECK(i >= 0 && i < r_ && j >= 0 && j < c_, "index out of range"); return d_[(size_t)j * r_ + i]; }
    double& operator()(int i) { LVREF_CHECK((r_ == 1 || c_ == 1) && i >= 0 && i < r_ * c_, "vector index out of range"); return d_[i]; }
    double operator()(int i) const { LVREF_CHECK((r_ == 1 || c_ == 1) && i >= 0 && i < r_ * c_, "vector index out of range"); return d_[i]; }
    double& operator[](int i) { return (*this)(i); }
    double operator[](int i) const { return (*this)(i); }
    double& x() { return (*this)(0); } double x() const { return (*this)(0); }
    double& y() { return (*this)(1); } double y() const { return (*this)(1); }
    double& z() { return (*this)(2); } double z() const { return (*this)(2); }
    double& w() { return (*this)(3); } double w() const { return (*this)(3); }
    double& coeffRef(int i, int j) { return (*this)(i, j); }
    double coeff(int i, int j) const { return (*this)(i, j); }
    operator double() const { LVREF_CHECK(r_ == 1 && c_ == 1, "1x1 expected"); return d_[0]; }      // `double g = r.transpose() * x;`

    // ---- views (non-const: writable; const: a copy)
    inline Blk block(int i, int j, int r, int c);
    XMat block(int i, int j, int r, int c) const { LVREF_CHECK(i >= 0 && j >= 0 && r >= 0 && c >= 0 && i + r <= r_ && j + c <= c_, "block out of range"); XMat o(r, c); for (int b = 0; b < c; ++b) for (int a = 0; a < r; ++a) o.d_[(size_t)b * r + a] = d_[(size_t)(j + b) * r_ + i + a]; return o; }
    template <int BR, int BC> inline Blk block(int i, int j);
    template <int BR, int BC> XMat block(int i, int j) const { return block(i, j, BR, BC); }
    inline Blk segment(int i, int n);
    XMat segment(int i, int n) const { return c_ == 1 ? block(i, 0, n, 1) : block(0, i, 1, n); }
    template <int N> inline Blk segment(int i);
    template <int N> XMat segment(int i) const { return segment(i, N); }
    inline Blk head(int n); XMat head(int n) const { return segment(0, n); }
    template <int N> inline Blk head(); template <int N> XMat head() const { return segment(0, N); }
    inline Blk tail(int n); XMat tail(int n) const { return segment(size() - n, n); }
    template <int N> inline Blk tail(); template <int N> XMat tail() const { return segment(size() - N, N); }
    inline Blk leftCols(int n); XMat leftCols(int n) const { return block(0, 0, r_, n); }
    template <int N> inline Blk leftCols(); template <int N> XMat leftCols() const { return block(0, 0, r_, N); }
    inline Blk rightCols(int n); XMat rightCols(int n) const { return block(0, c_ - n, r_, n); }
    template <int N> inline Blk rightCols(); template <int N> XMat rightCols() const { return block(0, c_ - N, r_, N); }
    inline Blk topRows(int n); XMat topRows(int n) const { return block(0, 0, n, c_); }
    template <int N> inline Blk topRows(); template <int N> XMat topRows() const { return block(0, 0, N, c_); }
    inline Blk bottomRows(int n); XMat bottomRows(int n) const { return block(r_ - n, 0, n, c_); }
    template <int N> inline Blk bottomRows(); template <int N> XMat bottomRows() const { return block(r_ - N, 0, N, c_); }
    inline Blk middleRows(int i, int n); XMat middleRows(int i, int n) const { return block(i, 0, n, c_); }
    inline Blk middleCols(int j, int n); XMat middleCols(int j, int n) const { return block(0, j, r_, n); }
    inline Blk col(int j); XMat col(int j) const { return block(0, j, r_, 1); }
    inline Blk row(int i); XMat row(int i) const { return block(i, 0, 1, c_); }
    inline Blk topLeftCorner(int r, int c); XMat topLeftCorner(int r, int c) const { return block(0, 0, r, c); }
    inline Blk topRightCorner(int r, int c); XMat topRightCorner(int r, int c) const { return block(0, c_ - c, r, c); }
    inline Blk bottomLeftCorner(int r, int c); XMat bottomLeftCorner(int r, int c) const { return block(r_ - r, 0, r, c); }
    inline Blk bottomRightCorner(int r, int c); XMat bottomRightCorner(int r, int c) const { return block(r_ - r, c_ - c, r, c); }
    template <int R, int C> inline Blk topLeftCorner(); template <int R, int C> XMat topLeftCorner() const { return block(0, 0, R, C); }
    template <int R, int C> inline Blk topRightCorner(); template <int R, int C> XMat topRightCorner() const { return block(0, c_ - C, R, C); }
    template <int R, int C> inline Blk bottomLeftCorner(); template <int R, int C> XMat bottomLeftCorner() const { return block(r_ - R, 0, R, C); }
    template <int R, int C> inline Blk bottomRightCorner(); template <int R, int C> XMat bottomRightCorner() const { return block(r_ - R, c_ - C, R, C); }
    XMat diagonal() const { const int n = std::min(r_, c_); XMat o(n, 1); for (int i = 0; i < n; ++i) o.d_[i] = (*this)(i, i); return o; }

    // ---- whole-matrix operations
    XMat transpose() const { XMat o(c_, r_); for (int j = 0; j < c_; ++j) for (int i = 0; i < r_; ++i) o.d_[(size_t)i * c_ + j] = d_[(size_t)j * r_ + i]; return o; }
    XMat adjoint() const { return transpose(); }
    XMat eval() const { return *this; }
    XMat& noalias() { return *this; }
    const XMat& matrix() const { return *this; }
    XMat array() const { return *this; }
    XMat cwiseAbs() const { XMat o = *this; for (double& v : o.d_) v = std::fabs(v); return o; }
    XMat cwiseSqrt() const { XMat o = *this; for (double& v : o.d_) v = std::sqrt(v); return o; }
    XMat cwiseProduct(const XMat& b) const { LVREF_CHECK(r_ == b.r_ && c_ == b.c_, "cwiseProduct: shape"); XMat o = *this; for (size_t k = 0; k < d_.size(); ++k) o.d_[k] *= b.d_[k]; return o; }
    XMat cwiseQuotient(const XMat& b) const { LVREF_CHECK(r_ == b.r_ && c_ == b.c_, "cwiseQuotient: shape"); XMat o = *this; for (size_t k = 0; k < d_.size(); ++k) o.d_[k] /= b.d_[k]; return o; }
    double squaredNorm() const { double s = 0; for (double v : d_) s += v * v; return s; }
    double norm() const { return std::sqrt(squaredNorm()); }
    double stableNorm() const { return norm(); }
    double sum() const { double s = 0; for (double v : d_) s += v; return s; }
    double mean() const { return sum() / (double)d_.size(); }
    double trace() const { double s = 0; for (int i = 0; i < std::min(r_, c_); ++i) s += (*this)(i, i); return s; }
    double maxCoeff() const { LVREF_CHECK(!d_.empty(), "maxCoeff of an empty matrix"); return *std::max_element(d_.begin(), d_.end()); }
    double minCoeff() const { LVREF_CHECK(!d_.empty(), "minCoeff of an empty matrix"); return *std::min_element(d_.begin(), d_.end()); }
    template <typename I> double maxCoeff(I* ri, I* ci) const { size_t k = std::max_element(d_.begin(), d_.end()) - d_.begin(); *ri = (I)(k % r_); *ci = (I)(k / r_); return d_[k]; }
    template <typename I> double maxCoeff(I* idx) const { size_t k = std::max_element(d_.begin(), d_.end()) - d_.begin(); *idx = (I)k; return d_[k]; }
    template <typename I> double minCoeff(I* idx) const { size_t k = std::min_element(d_.begin(), d_.end()) - d_.begin(); *idx = (I)k; return d_[k]; }
    bool hasNaN() const { for (double v : d_) if (v != v) return true; return false; }
    bool allFinite() const { for (double v : d_) if (!std::isfinite(v)) return false; return true; }
    bool isZero(double prec = 1e-12) const { for (double v : d_) if (std::fabs(v) > prec) return false; return true; }
    XMat normalized() const { const double n = norm(); XMat o = *this; if (n > 0) for (double& v : o.d_) v /= n; return o; }
    void normalize() { const double n = norm(); if (n > 0) for (double& v : d_) v /= n; }
    double dot(const XMat& b) const { LVREF_CHECK(size() == b.size(), "dot: size"); double s = 0; for (size_t k = 0; k < d_.size(); ++k) s += d_[k] * b.d_[k]; return s; }
    XMat cross(const XMat& b) const { LVREF_CHECK(size() == 3 && b.size() == 3, "cross: 3-vectors"); XMat o(3, 1); o.d_[0] = d_[1] * b.d_[2] - d_[2] * b.d_[1]; o.d_[1] = d_[2] * b.d_[0] - d_[0] * b.d_[2]; o.d_[2] = d_[0] * b.d_[1] - d_[1] * b.d_[0]; return o; }
    XMat asDiagonal() const { const int n = size(); XMat o(n, n); for (int i = 0; i < n; ++i) o(i, i) = d_[i]; return o; }
    template <typename T> XMat cast() const { return *this; }
    XMat sparseView() const { return *this; }
    inline XMat inverse() const;
    inline double determinant() const;
    inline LDLTx ldlt() const;
    inline LDLTx llt() const;
    inline struct JacobiSVDx jacobiSvd(unsigned opts = 0) const;
    void swap(XMat& o) { std::swap(r_, o.r_); std::swap(c_, o.c_); d_.swap(o.d_); }
    XMat& setZero() { std::fill(d_.begin(), d_.end(), 0.0); return *this; }
    XMat& setOnes() { std::fill(d_.begin(), d_.end(), 1.0); return *this; }
    XMat& setConstant(double v) { std::fill(d_.begin(), d_.end(), v); return *this; }
    XMat& fill(double v) { return setConstant(v); }
    XMat& setIdentity() { setZero(); for (int i = 0; i < std::min(r_, c_); ++i) (*this)(i, i) = 1.0; return *this; }
    XMat& setZero(int r, int c) { r_ = r; c_ = c; d_.assign((size_t)r * c, 0.0); return *this; }
    XMat& setZero(int n) { return setZero(n, 1); }
    XMat& setIdentity(int r, int c) { setZero(r, c); return setIdentity(); }
    void resize(int r, int c) { if (r != r_ || c != c_) { r_ = r; c_ = c; d_.assign((size_t)r * c, 0.0); } }
    void resize(int n) { if (c_ == 1 || (r_ == 0 && c_ == 0)) resize(n, 1); else if (r_ == 1) resize(1, n); else LVREF_CHECK(false, "resize(n) of a matrix"); }
    void conservativeResize(int r, int c)
    {   // (new entries are uninitialised in Eigen; zero here)
        XMat o(r, c);
        for (int j = 0; j < std::min(c, c_); ++j) for (int i = 0; i < std::min(r, r_); ++i) o.d_[(size_t)j * r + i] = d_[(size_t)j * r_ + i];
        *this = o;
    }
    void conservativeResize(int n) { if (r_ == 1 && c_ != 1) conservativeResize(1, n); else conservativeResize(n, 1); }
    XMat& operator+=(const XMat& b) { LVREF_CHECK(r_ == b.r_ && c_ == b.c_, "+=: shape"); for (size_t k = 0; k < d_.size(); ++k) d_[k] += b.d_[k]; return *this; }
    XMat& operator-=(const XMat& b) { LVREF_CHECK(r_ == b.r_ && c_ == b.c_, "-=: shape"); for (size_t k = 0; k < d_.size(); ++k) d_[k] -= b.d_[k]; return *this; }
    XMat& operator*=(double s) { for (double& v : d_) v *= s; return *this; }
    XMat& operator/=(double s) { for (double& v : d_) v /= s; return *this; }
    inline XMat& operator*=(const XMat& b);

    // ---- comma initialiser (scalars and blocks, row by row, as Eigen's CommaInitializer)
    struct Comma {
        XMat& m; int row, col, cur_rows;
        Comma(XMat& m_, double v) : m(m_), row(0), col(1), cur_rows(1) { LVREF_CHECK(m.r_ > 0 && m.c_ > 0, "<< into an empty matrix"); m(0, 0) = v; }
        Comma(XMat& m_, const XMat& b) : m(m_), row(0), col(b.c_), cur_rows(b.r_) { put(0, 0, b); }
        void put(int i, int j, const XMat& b) { LVREF_CHECK(i + b.r_ <= m.r_ && j + b.c_ <= m.c_, "<<: too many coefficients"); for (int q = 0; q < b.c_; ++q) for (int p = 0; p < b.r_; ++p) m(i + p, j + q) = b(p, q); }
        Comma& operator,(double v) { if (col == m.c_) { row += cur_rows; col = 0; cur_rows = 1; } LVREF_CHECK(row < m.r_ && col < m.c_, "<<: too many coefficients"); m(row, col++) = v; return *this; }
        Comma& operator,(const XMat& b) { if (col == m.c_) { row += cur_rows; col = 0; cur_rows = b.r_; } put(row, col, b); col += b.c_; return *this; }
        XMat& finished() { return m; }
    };
    Comma operator<<(double v) { return Comma(*this, v); }
    Comma operator<<(const XMat& b) { return Comma(*this, b); }
};

inline std::ostream& operator<<(std::ostream& os, const XMat& m)
{
    for (int i = 0; i < m.rows(); ++i) { for (int j = 0; j < m.cols(); ++j) os << (j ? " " : "") << m(i, j); if (i + 1 < m.rows()) os << "\n"; }
    return os;
}
inline bool operator==(const XMat& a, const XMat& b) { return a.r_ == b.r_ && a.c_ == b.c_ && a.d_ == b.d_; }
inline bool operator!=(const XMat& a, const XMat& b) { return !(a == b); }
inline XMat operator+(const XMat& a, const XMat& b) { XMat o = a; o += b; return o; }
inline XMat operator-(const XMat& a, const XMat& b) { XMat o = a; o -= b; return o; }
inline XMat operator-(const XMat& a) { XMat o = a; for (double& v : o.d_) v = -v; return o; }
inline XMat operator*(const XMat& a, double s) { XMat o = a; o *= s; return o; }
inline XMat operator*(double s, const XMat& a) { XMat o = a; o *= s; return o; }
inline XMat operator/(const XMat& a, double s) { XMat o = a; o /= s; return o; }
inline XMat operator*(const XMat& a, const XMat& b)
{
    LVREF_CHECK(a.c_ == b.r_, "product: inner dimensions");
    XMat o(a.r_, b.c_);
    for (int j = 0; j < b.c_; ++j)
        for (int k = 0; k < a.c_; ++k) {
            const double bkj = b.d_[(size_t)j * b.r_ + k];
            if (bkj == 0.0) continue;
            const double* ac = &a.d_[(size_t)k * a.r_]; double* oc = &o.d_[(size_t)j * o.r_];
            for (int i = 0; i < a.r_; ++i) oc[i] += ac[i] * bkj;
        }
    return o;
}
inline XMat& XMat::operator*=(const XMat& b) { *this = *this * b; return *this; }

// ------------------------------------------------------------------ writable view
class Blk {
public:
    XMat& m; int i0, j0, r_, c_;
    Blk(XMat& m_, int i, int j, int r, int c) : m(m_), i0(i), j0(j), r_(r), c_(c) { LVREF_CHECK(i >= 0 && j >= 0 && r >= 0 && c >= 0 && i + r <= m.r_ && j + c <= m.c_, "block out of range"); }
    int rows() const { return r_; } int cols() const { return c_; } int size() const { return r_ * c_; }
    double& operator()(int i, int j) { LVREF_CHECK(i >= 0 && i < r_ && j >= 0 && j < c_, "block index out of range"); return m(i0 + i, j0 + j); }
    double operator()(int i, int j) const { LVREF_CHECK(i >= 0 && i < r_ && j >= 0 && j < c_, "block index out of range"); return ((const XMat&)m)(i0 + i, j0 + j); }
    double& operator()(int i) { return c_ == 1 ? (*this)(i, 0) : (*this)(0, i); }
    double operator()(int i) const { return c_ == 1 ? (*this)(i, 0) : (*this)(0, i); }
    double& operator[](int i) { return (*this)(i); } double operator[](int i) const { return (*this)(i); }
    double& x() { return (*this)(0); } double& y() { return (*this)(1); } double& z() { return (*this)(2); } double& w() { return (*this)(3); }
    double x() const { return (*this)(0); } double y() const { return (*this)(1); } double z() const { return (*this)(2); } double w() const { return (*this)(3); }
    XMat eval() const { XMat o(r_, c_); for (int j = 0; j < c_; ++j) for (int i = 0; i < r_; ++i) o.d_[(size_t)j * r_ + i] = ((const XMat&)m)(i0 + i, j0 + j); return o; }
    Blk& assign(const XMat& b)
    {
        if ((r_ == 1 || c_ == 1) && b.r_ == c_ && b.c_ == r_ && r_ != c_) { for (int k = 0; k < r_ * c_; ++k) (*this)(k) = b.d_[k]; return *this; }     // vector <-> row vector, as Eigen allows
        LVREF_CHECK(b.r_ == r_ && b.c_ == c_, "block assignment: shape"); for (int j = 0; j < c_; ++j) for (int i = 0; i < r_; ++i) m(i0 + i, j0 + j) = b.d_[(size_t)j * r_ + i]; return *this;
    }
    Blk& operator=(const XMat& b) { return assign(b); }
    Blk& operator=(const Blk& b) { return assign(b.eval()); }                     // (evaluated first: overlapping source and destination)
    Blk& operator+=(const XMat& b) { return assign(eval() + b); }
    Blk& operator-=(const XMat& b) { return assign(eval() - b); }
    Blk& operator*=(double s) { return assign(eval() * s); }
    Blk& operator/=(double s) { return assign(eval() / s); }
    Blk& setZero() { for (int j = 0; j < c_; ++j) for (int i = 0; i < r_; ++i) m(i0 + i, j0 + j) = 0.0; return *this; }
    Blk& setConstant(double v) { for (int j = 0; j < c_; ++j) for (int i = 0; i < r_; ++i) m(i0 + i, j0 + j) = v; return *this; }
    Blk& setOnes() { return setConstant(1.0); }
    Blk& setIdentity() { setZero(); for (int i = 0; i < std::min(r_, c_); ++i) m(i0 + i, j0 + i) = 1.0; return *this; }
    Blk& noalias() { return *this; }
    Blk block(int i, int j, int r, int c) { LVREF_CHECK(i + r <= r_ && j + c <= c_, "nested block out of range"); return Blk(m, i0 + i, j0 + j, r, c); }
    template <int BR, int BC> Blk block(int i, int j) { return block(i, j, BR, BC); }
    Blk segment(int i, int n) { return c_ == 1 ? block(i, 0, n, 1) : block(0, i, 1, n); }
    template <int N> Blk segment(int i) { return segment(i, N); }
    Blk head(int n) { return segment(0, n); } template <int N> Blk head() { return segment(0, N); }
    Blk tail(int n) { return segment(size() - n, n); } template <int N> Blk tail() { return segment(size() - N, N); }
    Blk col(int j) { return block(0, j, r_, 1); } Blk row(int i) { return block(i, 0, 1, c_); }
    Blk leftCols(int n) { return block(0, 0, r_, n); } Blk rightCols(int n) { return block(0, c_ - n, r_, n); }
    Blk topRows(int n) { return block(0, 0, n, c_); } Blk bottomRows(int n) { return block(r_ - n, 0, n, c_); }
    XMat transpose() const { return eval().transpose(); }
    XMat inverse() const { return eval().inverse(); }
    XMat cwiseAbs() const { return eval().cwiseAbs(); }
    XMat normalized() const { return eval().normalized(); }
    XMat diagonal() const { return eval().diagonal(); }
    XMat asDiagonal() const { return eval().asDiagonal(); }
    XMat array() const { return eval(); }
    double norm() const { return eval().norm(); } double squaredNorm() const { return eval().squaredNorm(); }
    double sum() const { return eval().sum(); } double trace() const { return eval().trace(); }
    double maxCoeff() const { return eval().maxCoeff(); } double minCoeff() const { return eval().minCoeff(); }
    double dot(const XMat& b) const { return eval().dot(b); } XMat cross(const XMat& b) const { return eval().cross(b); }
    inline LDLTx ldlt() const;
    XMat::Comma operator<<(double v);
    XMat::Comma operator<<(const XMat& b);
};
inline XMat::XMat(const Blk& b) { *this = b.eval(); }
inline Blk XMat::block(int i, int j, int r, int c) { return Blk(*this, i, j, r, c); }
template <int BR, int BC> inline Blk XMat::block(int i, int j) { return Blk(*this, i, j, BR, BC); }
inline Blk XMat::segment(int i, int n) { return c_ == 1 ? Blk(*this, i, 0, n, 1) : Blk(*this, 0, i, 1, n); }
template <int N> inline Blk XMat::segment(int i) { return segment(i, N); }
inline Blk XMat::head(int n) { return segment(0, n); } template <int N> inline Blk XMat::head() { return segment(0, N); }
inline Blk XMat::tail(int n) { return segment(size() - n, n); } template <int N> inline Blk XMat::tail() { return segment(size() - N, N); }
inline Blk XMat::leftCols(int n) { return Blk(*this, 0, 0, r_, n); } template <int N> inline Blk XMat::leftCols() { return leftCols(N); }
inline Blk XMat::rightCols(int n) { return Blk(*this, 0, c_ - n, r_, n); } template <int N> inline Blk XMat::rightCols() { return rightCols(N); }
inline Blk XMat::topRows(int n) { return Blk(*this, 0, 0, n, c_); } template <int N> inline Blk XMat::topRows() { return topRows(N); }
inline Blk XMat::bottomRows(int n) { return Blk(*this, r_ - n, 0, n, c_); } template <int N> inline Blk XMat::bottomRows() { return bottomRows(N); }
inline Blk XMat::middleRows(int i, int n) { return Blk(*this, i, 0, n, c_); } inline Blk XMat::middleCols(int j, int n) { return Blk(*this, 0, j, r_, n); }
inline Blk XMat::col(int j) { return Blk(*this, 0, j, r_, 1); } inline Blk XMat::row(int i) { return Blk(*this, i, 0, 1, c_); }
inline Blk XMat::topLeftCorner(int r, int c) { return Blk(*this, 0, 0, r, c); } inline Blk XMat::topRightCorner(int r, int c) { return Blk(*this, 0, c_ - c, r, c); }
inline Blk XMat::bottomLeftCorner(int r, int c) { return Blk(*this, r_ - r, 0, r, c); } inline Blk XMat::bottomRightCorner(int r, int c) { return Blk(*this, r_ - r, c_ - c, r, c); }
template <int R, int C> inline Blk XMat::topLeftCorner() { return topLeftCorner(R, C); } template <int R, int C> inline Blk XMat::topRightCorner() { return topRightCorner(R, C); }
template <int R, int C> inline Blk XMat::bottomLeftCorner() { return bottomLeftCorner(R, C); } template <int R, int C> inline Blk XMat::bottomRightCorner() { return bottomRightCorner(R, C); }

// ------------------------------------------------------------------ dense kernels the filter calls
// LU with partial pivoting: inverse and determinant
inline bool lvref_lu(XMat& a, std::vector<int>& piv, int& sign)
{
    const int n = a.r_; piv.resize(n); sign = 1;
    for (int k = 0; k < n; ++k) {
        int p = k; double best = std::fabs(a(k, k));
        for (int i = k + 1; i < n; ++i) if (std::fabs(a(i, k)) > best) { best = std::fabs(a(i, k)); p = i; }
        piv[k] = p;
        if (best == 0.0) return false;
        if (p != k) { for (int j = 0; j < n; ++j) std::swap(a(k, j), a(p, j)); sign = -sign; }
        for (int i = k + 1; i < n; ++i) { const double l = a(i, k) / a(k, k); a(i, k) = l; if (l != 0.0) for (int j = k + 1; j < n; ++j) a(i, j) -= l * a(k, j); }
    }
    return true;
}
inline XMat XMat::inverse() const
{
    LVREF_CHECK(r_ == c_, "inverse of a non-square matrix");
    const int n = r_; XMat a = *this; std::vector<int> piv; int sign;
    if (!lvref_lu(a, piv, sign)) { XMat o(n, n); o.setConstant(std::numeric_limits<double>::infinity()); return o; }
    XMat x(n, n); x.setIdentity();
    for (int k = 0; k < n; ++k) if (piv[k] != k) for (int j = 0; j < n; ++j) std::swap(x(k, j), x(piv[k], j));
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) { double s = x(i, j); for (int k = 0; k < i; ++k) s -= a(i, k) * x(k, j); x(i, j) = s; }
        for (int i = n - 1; i >= 0; --i) { double s = x(i, j); for (int k = i + 1; k < n; ++k) s -= a(i, k) * x(k, j); x(i, j) = s / a(i, i); }
    }
    return x;
}
inline double XMat::determinant() const
{
    LVREF_CHECK(r_ == c_, "determinant of a non-square matrix");
    XMat a = *this; std::vector<int> piv; int sign;
    if (!lvref_lu(a, piv, sign)) return 0.0;
    double d = sign; for (int i = 0; i < r_; ++i) d *= a(i, i); return d;
}
// LDL^T with diagonal pivoting (Eigen's LDLT pivots the same way: largest remaining diagonal entry first)
struct LDLTx {
    XMat L; std::vector<double> D; std::vector<int> perm; bool ok = true;
    explicit LDLTx(const XMat& A)
    {
        LVREF_CHECK(A.r_ == A.c_, "ldlt of a non-square matrix");
        const int n = A.r_; L = A; D.assign(n, 0.0); perm.resize(n); for (int i = 0; i < n; ++i) perm[i] = i;
        for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) L(i, j) = L(j, i);      // Eigen's LDLT<_, Lower> reads the lower triangle only (larvio.cpp:1665 hands it a triangular block in the 3-D mode)
        for (int k = 0; k < n; ++k) {
            int p = k; double best = std::fabs(L(k, k));
            for (int i = k + 1; i < n; ++i) if (std::fabs(L(i, i)) > best) { best = std::fabs(L(i, i)); p = i; }
            if (p != k) { for (int j = 0; j < n; ++j) std::swap(L(k, j), L(p, j)); for (int i = 0; i < n; ++i) std::swap(L(i, k), L(i, p)); std::swap(perm[k], perm[p]); }
            const double d = L(k, k); D[k] = d;
            if (d == 0.0) { ok = false; for (int i = k + 1; i < n; ++i) L(i, k) = 0.0; continue; }
            for (int i = k + 1; i < n; ++i) L(i, k) /= d;
            for (int j = k + 1; j < n; ++j) { const double ljk = L(j, k) * d; if (ljk != 0.0) for (int i = j; i < n; ++i) L(i, j) -= L(i, k) * ljk; }
            for (int j = k + 1; j < n; ++j) for (int i = k + 1; i < j; ++i) L(i, j) = L(j, i);       // keep the trailing block symmetric for the pivot swaps
        }
    }
    XMat solve(const XMat& B) const
    {
        const int n = L.r_; LVREF_CHECK(B.r_ == n, "ldlt.solve: rows");
        XMat X(n, B.c_);
        for (int j = 0; j < B.c_; ++j) {
            std::vector<double> y(n);
            for (int i = 0; i < n; ++i) y[i] = B(perm[i], j);
            for (int i = 0; i < n; ++i) { double s = y[i]; for (int k = 0; k < i; ++k) s -= L(i, k) * y[k]; y[i] = s; }
            for (int i = 0; i < n; ++i) y[i] = D[i] != 0.0 ? y[i] / D[i] : 0.0;
            for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L(k, i) * y[k]; y[i] = s; }
            for (int i = 0; i < n; ++i) X(perm[i], j) = y[i];
        }
        return X;
    }
    int info() const { return ok ? 0 : 1; }
    bool isPositive() const { for (double d : D) if (d <= 0) return false; return true; }
};
inline LDLTx XMat::ldlt() const { return LDLTx(*this); }
inline LDLTx XMat::llt() const { return LDLTx(*this); }
inline LDLTx Blk::ldlt() const { return LDLTx(eval()); }
enum { Success = 0, NumericalIssue = 1 };

// Householder QR with the full Q (rows x rows): Q^T A = R.  The stand-in behind JacobiSVD::matrixU (range basis first, null space last)
// and SPQR (natural ordering, no column permutation).
inline void lvref_householder_qr(const XMat& A, XMat& Q, XMat& R)
{
    const int m = A.r_, n = A.c_; R = A; Q = XMat(m, m); Q.setIdentity();
    std::vector<double> v(m);
    for (int k = 0; k < std::min(m - 1, n); ++k) {
        double s = 0; for (int i = k; i < m; ++i) s += R(i, k) * R(i, k);
        const double nrm = std::sqrt(s);
        if (nrm == 0.0) continue;
        const double alpha = R(k, k) >= 0 ? -nrm : nrm;
        for (int i = k; i < m; ++i) v[i] = R(i, k);
        v[k] -= alpha;
        double vn2 = 0; for (int i = k; i < m; ++i) vn2 += v[i] * v[i];
        if (vn2 == 0.0) continue;
        const double beta = 2.0 / vn2;
        for (int j = k; j < n; ++j) { double d = 0; for (int i = k; i < m; ++i) d += v[i] * R(i, j); d *= beta; if (d != 0.0) for (int i = k; i < m; ++i) R(i, j) -= d * v[i]; }
        for (int i = k + 1; i < m; ++i) R(i, k) = 0.0;
        for (int j = 0; j < m; ++j) { double d = 0; for (int i = k; i < m; ++i) d += Q(j, i) * v[i]; d *= beta; if (d != 0.0) for (int i = k; i < m; ++i) Q(j, i) -= d * v[i]; }     // Q <- Q H_k
    }
}

// A real singular value decomposition for the SMALL matrices that ask for singular vectors (initial_sfm.cpp's 4x4 triangulation takes
// matrixV().rightCols<1>()): one-sided Jacobi (Hestenes) on the columns, singular values sorted descending as Eigen's are.
struct JacobiSVDx {
    XMat U_, V_, S_;
    explicit JacobiSVDx(const XMat& A)
    {
        const int m = A.r_, n = A.c_; XMat B = A; V_ = XMat(n, n); V_.setIdentity();
        for (int sweep = 0; sweep < 60; ++sweep) {
            double off = 0;
            for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < m; ++i) { a += B(i, p) * B(i, p); b += B(i, q) * B(i, q); c += B(i, p) * B(i, q); }
                if (c == 0.0) continue;
                off = std::max(off, std::fabs(c) / std::sqrt(std::max(a * b, 1e-300)));
                const double zeta = (b - a) / (2.0 * c), t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta)), cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < m; ++i) { const double x = B(i, p), y = B(i, q); B(i, p) = cs * x - sn * y; B(i, q) = sn * x + cs * y; }
                for (int i = 0; i < n; ++i) { const double x = V_(i, p), y = V_(i, q); V_(i, p) = cs * x - sn * y; V_(i, q) = sn * x + cs * y; }
            }
            if (off < 1e-15) break;
        }
        std::vector<double> sv(n); std::vector<int> order(n);
        for (int j = 0; j < n; ++j) { double s = 0; for (int i = 0; i < m; ++i) s += B(i, j) * B(i, j); sv[j] = std::sqrt(s); order[j] = j; }
        std::sort(order.begin(), order.end(), [&](int a, int b) { return sv[a] > sv[b]; });
        XMat Vs(n, n), Us(m, n); S_ = XMat(n, 1);
        for (int k = 0; k < n; ++k) {
            const int j = order[k]; S_(k) = sv[j];
            for (int i = 0; i < n; ++i) Vs(i, k) = V_(i, j);
            for (int i = 0; i < m; ++i) Us(i, k) = sv[j] > 0 ? B(i, j) / sv[j] : 0.0;
        }
        V_ = Vs; U_ = Us;
    }
    const XMat& matrixV() const { return V_; } const XMat& matrixU() const { return U_; } const XMat& singularValues() const { return S_; }
};
inline JacobiSVDx XMat::jacobiSvd(unsigned) const { return JacobiSVDx(*this); }

// ------------------------------------------------------------------ the shaped derivations
template <typename T, int R, int C, int Opt = 0, int MR = R, int MC = C> class Matrix : public XMat {
    static_assert(std::is_same<T, double>::value, "lvref_eigen2: double matrices only");
    void shape_check() const { LVREF_CHECK((R == Dynamic || r_ == R) && (C == Dynamic || c_ == C), "assignment to a fixed-size matrix: shape"); }
public:
    enum { RowsAtCompileTime = R, ColsAtCompileTime = C };
    Matrix() : XMat(R == Dynamic ? 0 : R, C == Dynamic ? (R == Dynamic ? 0 : 0) : C) { if (R == Dynamic && C != Dynamic) { r_ = 0; c_ = C; } if (C == Dynamic && R != Dynamic) { r_ = R; c_ = 0; } }
    Matrix(const XMat& m) : XMat(m) { fix_vector_shape(); shape_check(); }
    Matrix(const Blk& b) : XMat(b) { fix_vector_shape(); shape_check(); }
    // (rows, cols) of a dynamic matrix, or the two coefficients of a fixed 2-vector
    Matrix(double a, double b) : XMat()
    {
        if (R == Dynamic && C == Dynamic) { r_ = (int)a; c_ = (int)b; d_.assign((size_t)r_ * c_, 0.0); }
        else if (R != Dynamic && C != Dynamic && R * C == 2) { r_ = R; c_ = C; d_ = {a, b}; }
        else if (R == Dynamic && C != Dynamic) { r_ = (int)a; c_ = C; LVREF_CHECK((int)b == C, "cols"); d_.assign((size_t)r_ * c_, 0.0); }
        else if (C == Dynamic && R != Dynamic) { r_ = R; c_ = (int)b; LVREF_CHECK((int)a == R, "rows"); d_.assign((size_t)r_ * c_, 0.0); }
        else LVREF_CHECK(false, "two-argument constructor of this shape");
    }
    explicit Matrix(double a) : XMat()
    {   // size of a dynamic vector, or the coefficient of a 1x1
        if (R == Dynamic && C == 1) { r_ = (int)a; c_ = 1; d_.assign((size_t)r_, 0.0); }
        else if (R == 1 && C == Dynamic) { r_ = 1; c_ = (int)a; d_.assign((size_t)c_, 0.0); }
        else if (R == 1 && C == 1) { r_ = c_ = 1; d_ = {a}; }
        else if (R == Dynamic && C == Dynamic) { r_ = (int)a; c_ = 1; d_.assign((size_t)r_, 0.0); }
        else LVREF_CHECK(false, "one-argument constructor of this shape");
    }
    Matrix(double a, double b, double c) : XMat(R, C) { LVREF_CHECK(R * C == 3, "three coefficients"); d_ = {a, b, c}; }
    Matrix(double a, double b, double c, double d) : XMat(R, C) { LVREF_CHECK(R * C == 4, "four coefficients"); d_ = {a, b, c, d}; }
    Matrix& operator=(const XMat& m) { XMat::operator=(m); fix_vector_shape(); shape_check(); return *this; }
    Matrix& operator=(const Blk& b) { XMat::operator=(b.eval()); fix_vector_shape(); shape_check(); return *this; }
    void fix_vector_shape()
    {   // a row assigned to a column type of the same length (and the reverse) is accepted where Eigen would transpose implicitly: vectors only
        if (C == 1 && R != 1 && c_ != 1 && r_ == 1) std::swap(r_, c_);
        else if (R == 1 && C != 1 && r_ != 1 && c_ == 1) std::swap(r_, c_);
    }
    static Matrix Zero() { Matrix m; m.setZero(); return m; }
    static Matrix Zero(int r, int c) { Matrix m; m.XMat::setZero(r, c); return m; }
    static Matrix Zero(int n) { Matrix m; if (R == 1 && C != 1) m.XMat::setZero(1, n); else m.XMat::setZero(n, 1); return m; }
    static Matrix Ones() { Matrix m; m.setOnes(); return m; }
    static Matrix Ones(int r, int c) { Matrix m; m.XMat::setZero(r, c); m.setOnes(); return m; }
    static Matrix Ones(int n) { Matrix m = Zero(n); m.setOnes(); return m; }
    static Matrix Constant(double v) { Matrix m; m.setConstant(v); return m; }
    static Matrix Constant(int r, int c, double v) { Matrix m; m.XMat::setZero(r, c); m.setConstant(v); return m; }
    static Matrix Constant(int n, double v) { Matrix m = Zero(n); m.setConstant(v); return m; }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Identity(int r, int c) { Matrix m; m.XMat::setZero(r, c); m.setIdentity(); return m; }
    static Matrix Random() { Matrix m; for (double& v : m.d_) v = 2.0 * std::rand() / RAND_MAX - 1.0; return m; }
    static Matrix Random(int r, int c) { Matrix m; m.XMat::setZero(r, c); for (double& v : m.d_) v = 2.0 * std::rand() / RAND_MAX - 1.0; return m; }
    static Matrix UnitX() { Matrix m; m.setZero(); m(0) = 1; return m; }
    static Matrix UnitY() { Matrix m; m.setZero(); m(1) = 1; return m; }
    static Matrix UnitZ() { Matrix m; m.setZero(); m(2) = 1; return m; }
};
inline XMat::Comma Blk::operator<<(double v) { LVREF_CHECK(false, "<< into a block is not provided"); static XMat dummy(1, 1); (void)v; return XMat::Comma(dummy, 0.0); }
inline XMat::Comma Blk::operator<<(const XMat& b) { LVREF_CHECK(false, "<< into a block is not provided"); static XMat dummy(1, 1); (void)b; return XMat::Comma(dummy, 0.0); }

typedef Matrix<double, 2, 2> Matrix2d; typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 1, 2> RowVector2d; typedef Matrix<double, 1, 3> RowVector3d; typedef Matrix<double, 1, 4> RowVector4d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd; typedef Matrix<double, Dynamic, 1> VectorXd; typedef Matrix<double, 1, Dynamic> RowVectorXd;
typedef Matrix<double, 6, 6> Matrix6d; typedef Matrix<double, 6, 1> Vector6d;

// ------------------------------------------------------------------ geometry (Eigen's conventions and formulas: Hamilton, w first in the constructor, x y z w in coeffs())
class AngleAxisd;
class Quaterniond {
    Vector4d q_;                                                         // x y z w
public:
    Quaterniond() { q_ = Vector4d(0, 0, 0, 1); }
    Quaterniond(double w, double x, double y, double z) { q_ = Vector4d(x, y, z, w); }
    explicit Quaterniond(const XMat& m)
    {
        if (m.rows() == 3 && m.cols() == 3) from_rotation(m);
        else if (m.size() == 4) q_ = Vector4d(m(0), m(1), m(2), m(3));
        else LVREF_CHECK(false, "Quaterniond from a matrix of this shape");
    }
    explicit Quaterniond(const Blk& b) : Quaterniond(b.eval()) {}
    inline Quaterniond(const AngleAxisd& aa);
    explicit Quaterniond(const double* p) { q_ = Vector4d(p[0], p[1], p[2], p[3]); }
    void from_rotation(const XMat& mat)
    {   // Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>
        double t = mat.trace();
        if (t > 0) {
            t = std::sqrt(t + 1.0);
            q_(3) = 0.5 * t; t = 0.5 / t;
            q_(0) = (mat(2, 1) - mat(1, 2)) * t; q_(1) = (mat(0, 2) - mat(2, 0)) * t; q_(2) = (mat(1, 0) - mat(0, 1)) * t;
        } else {
            int i = 0;
            if (mat(1, 1) > mat(0, 0)) i = 1;
            if (mat(2, 2) > mat(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + 1.0);
            q_(i) = 0.5 * t; t = 0.5 / t;
            q_(3) = (mat(k, j) - mat(j, k)) * t; q_(j) = (mat(j, i) + mat(i, j)) * t; q_(k) = (mat(k, i) + mat(i, k)) * t;
        }
    }
    Quaterniond& operator=(const XMat& m) { *this = Quaterniond(m); return *this; }
    double w() const { return q_(3); } double x() const { return q_(0); } double y() const { return q_(1); } double z() const { return q_(2); }
    double& w() { return q_(3); } double& x() { return q_(0); } double& y() { return q_(1); } double& z() { return q_(2); }
    const Vector4d& coeffs() const { return q_; } Vector4d& coeffs() { return q_; }
    Vector3d vec() const { return Vector3d(q_(0), q_(1), q_(2)); }
    static Quaterniond Identity() { return Quaterniond(1, 0, 0, 0); }
    Quaterniond& setIdentity() { q_ = Vector4d(0, 0, 0, 1); return *this; }
    double norm() const { return q_.norm(); } double squaredNorm() const { return q_.squaredNorm(); }
    void normalize() { q_.normalize(); }
    Quaterniond normalized() const { Quaterniond o = *this; o.normalize(); return o; }
    Quaterniond conjugate() const { return Quaterniond(w(), -x(), -y(), -z()); }
    Quaterniond inverse() const { const double n2 = squaredNorm(); if (n2 > 0) return Quaterniond(w() / n2, -x() / n2, -y() / n2, -z() / n2); return Quaterniond(0, 0, 0, 0); }
    Quaterniond operator*(const Quaterniond& b) const
    {
        const Quaterniond& a = *this;
        return Quaterniond(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                           a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                           a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                           a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
    }
    Quaterniond& operator*=(const Quaterniond& b) { *this = *this * b; return *this; }
    Matrix3d toRotationMatrix() const
    {   // Eigen's QuaternionBase::toRotationMatrix
        Matrix3d res;
        const double tx = 2 * x(), ty = 2 * y(), tz = 2 * z();
        const double twx = tx * w(), twy = ty * w(), twz = tz * w();
        const double txx = tx * x(), txy = ty * x(), txz = tz * x();
        const double tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        res(0, 0) = 1 - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = 1 - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = 1 - (txx + tyy);
        return res;
    }
    Matrix3d matrix() const { return toRotationMatrix(); }
    Vector3d operator*(const XMat& v) const { LVREF_CHECK(v.size() == 3, "quaternion * 3-vector"); Vector3d p(v(0), v(1), v(2)); Vector3d u = vec(); Vector3d uv = u.cross(p); uv = uv + uv; return Vector3d(p + w() * uv + u.cross(uv)); }
    Vector3d _transformVector(const XMat& v) const { return (*this) * v; }
    static Quaterniond FromTwoVectors(const XMat& a, const XMat& b)
    {   // Eigen's QuaternionBase::setFromTwoVectors (the branch for vectors that are not nearly opposite)
        Vector3d v0 = a.normalized(), v1 = b.normalized();
        double c = v1.dot(v0);
        LVREF_CHECK(c > -1.0 + 1e-12, "FromTwoVectors: opposite vectors are not provided");
        Vector3d axis = v0.cross(v1);
        const double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
        return Quaterniond(s * 0.5, axis(0) * invs, axis(1) * invs, axis(2) * invs);
    }
    Quaterniond& setFromTwoVectors(const XMat& a, const XMat& b) { *this = FromTwoVectors(a, b); return *this; }
    double angularDistance(const Quaterniond& o) const { Quaterniond d = (*this) * o.conjugate(); return 2.0 * std::atan2(d.vec().norm(), std::fabs(d.w())); }
};
typedef Quaterniond Quaternion_d;
template <typename T> using Quaternion = Quaterniond;
class AngleAxisd {
    double angle_ = 0; Vector3d axis_;
public:
    AngleAxisd() { axis_ = Vector3d(1, 0, 0); }
    AngleAxisd(double a, const XMat& ax) : angle_(a) { axis_ = ax; }
    explicit AngleAxisd(const Quaterniond& q) { from_q(q); }
    explicit AngleAxisd(const XMat& R) { from_q(Quaterniond(R)); }
    void from_q(const Quaterniond& q)
    {   // Eigen's AngleAxis::operator=(QuaternionBase)
        double n = q.vec().norm();
        if (n != 0.0) { angle_ = 2.0 * std::atan2(n, std::fabs(q.w())); if (q.w() < 0) n = -n; axis_ = q.vec() / n; }
        else { angle_ = 0; axis_ = Vector3d(1, 0, 0); }
    }
    double angle() const { return angle_; } double& angle() { return angle_; }
    const Vector3d& axis() const { return axis_; } Vector3d& axis() { return axis_; }
    Matrix3d toRotationMatrix() const
    {   // Eigen's AngleAxis::toRotationMatrix
        Matrix3d res; const double s = std::sin(angle_), c = std::cos(angle_);
        Vector3d sin_axis = s * axis_, cos1_axis = (1.0 - c) * axis_;
        double tmp;
        tmp = cos1_axis.x() * axis_.y(); res(0, 1) = tmp - sin_axis.z(); res(1, 0) = tmp + sin_axis.z();
        tmp = cos1_axis.x() * axis_.z(); res(0, 2) = tmp + sin_axis.y(); res(2, 0) = tmp - sin_axis.y();
        tmp = cos1_axis.y() * axis_.z(); res(1, 2) = tmp - sin_axis.x(); res(2, 1) = tmp + sin_axis.x();
        for (int i = 0; i < 3; ++i) res(i, i) = cos1_axis(i) * axis_(i) + c;
        return res;
    }
    Matrix3d matrix() const { return toRotationMatrix(); }
};
template <typename T> using AngleAxis = AngleAxisd;
inline Quaterniond::Quaterniond(const AngleAxisd& aa) { const double h = 0.5 * aa.angle(), s = std::sin(h); q_ = Vector4d(s * aa.axis()(0), s * aa.axis()(1), s * aa.axis()(2), std::cos(h)); }

class Isometry3d {
    Matrix4d m_;
public:
    Isometry3d() { m_.setIdentity(); }
    static Isometry3d Identity() { return Isometry3d(); }
    Isometry3d& setIdentity() { m_.setIdentity(); return *this; }
    Blk linear() { return m_.block(0, 0, 3, 3); } XMat linear() const { return m_.block(0, 0, 3, 3); }
    Blk rotation_ref() { return linear(); } XMat rotation() const { return linear(); }
    Blk translation() { return m_.block(0, 3, 3, 1); } XMat translation() const { return m_.block(0, 3, 3, 1); }
    Matrix4d& matrix() { return m_; } const Matrix4d& matrix() const { return m_; }
    double& operator()(int i, int j) { return m_(i, j); } double operator()(int i, int j) const { return m_(i, j); }
    Isometry3d inverse() const
    {   // (Isometry: R^T, -R^T t)
        Isometry3d o; XMat Rt = linear().transpose(); o.linear() = Rt; o.translation() = -(Rt * translation()); return o;
    }
    Isometry3d operator*(const Isometry3d& b) const { Isometry3d o; o.linear() = linear() * b.linear(); o.translation() = linear() * b.translation() + translation(); return o; }
    Vector3d operator*(const XMat& p) const { LVREF_CHECK(p.size() == 3, "Isometry3d * 3-vector"); return Vector3d(linear() * p + translation()); }
    Isometry3d& operator=(const XMat& m) { LVREF_CHECK(m.rows() == 4 && m.cols() == 4, "Isometry3d from a 4x4"); m_ = m; return *this; }
};
typedef Isometry3d Affine3d;

// ------------------------------------------------------------------ decompositions as the filter names them
template <typename M> class JacobiSVD {
    XMat U_, R_;
public:
    JacobiSVD() {}
    JacobiSVD(const XMat& A, unsigned = 0) { compute(A); }
    JacobiSVD& compute(const XMat& A, unsigned = 0) { lvref_householder_qr(A, U_, R_); return *this; }
    // an orthogonal U whose first rank(A) columns span range(A) and whose last rows - rank columns span its left null space: all the
    // filter takes from it (`matrixU().rightCols(rows - rank)`); NOT the singular vectors themselves
    const XMat& matrixU() const { return U_; }
};
template <typename M> class HouseholderQR {
    XMat Q_, R_;
public:
    HouseholderQR() {}
    explicit HouseholderQR(const XMat& A) { compute(A); }
    HouseholderQR& compute(const XMat& A) { lvref_householder_qr(A, Q_, R_); return *this; }
    const XMat& householderQ() const { return Q_; } const XMat& matrixQR() const { return R_; }
};
template <typename T> using SparseMatrix = XMat;
enum { SPQR_ORDERING_NATURAL = 1 };
// SPQR with natural ordering = a Householder QR without column permutation.  Q is kept as its reflectors (a tall measurement stack has
// tens of thousands of rows: an explicit rows x rows Q would not fit) and applied to whatever it multiplies.
struct SPQRFactors {
    int m = 0, n = 0; XMat R; std::vector<std::vector<double>> v; std::vector<double> beta; std::vector<int> k0;
    void compute(const XMat& A)
    {
        m = A.r_; n = A.c_; R = A; v.clear(); beta.clear(); k0.clear();
        for (int k = 0; k < std::min(m - 1, n); ++k) {
            double s = 0; for (int i = k; i < m; ++i) s += R(i, k) * R(i, k);
            const double nrm = std::sqrt(s);
            if (nrm == 0.0) continue;
            const double alpha = R(k, k) >= 0 ? -nrm : nrm;
            std::vector<double> w((size_t)(m - k));
            for (int i = k; i < m; ++i) w[(size_t)(i - k)] = R(i, k);
            w[0] -= alpha;
            double vn2 = 0; for (double x : w) vn2 += x * x;
            if (vn2 == 0.0) continue;
            const double b = 2.0 / vn2;
            for (int j = k; j < n; ++j) { double d = 0; for (int i = k; i < m; ++i) d += w[(size_t)(i - k)] * R(i, j); d *= b; if (d != 0.0) for (int i = k; i < m; ++i) R(i, j) -= d * w[(size_t)(i - k)]; }
            for (int i = k + 1; i < m; ++i) R(i, k) = 0.0;
            v.push_back(std::move(w)); beta.push_back(b); k0.push_back(k);
        }
    }
    XMat apply(const XMat& B, bool transpose) const
    {   // Q = H_1 H_2 ... H_p:  Q^T B applies H_1 first, Q B applies H_p first
        LVREF_CHECK(B.r_ == m, "SPQR Q times a matrix of another height");
        XMat X = B; const int p = (int)v.size();
        for (int t = 0; t < p; ++t) {
            const int q = transpose ? t : p - 1 - t; const int k = k0[(size_t)q]; const std::vector<double>& w = v[(size_t)q];
            for (int j = 0; j < X.c_; ++j) { double d = 0; for (int i = k; i < m; ++i) d += w[(size_t)(i - k)] * X(i, j); d *= beta[(size_t)q]; if (d != 0.0) for (int i = k; i < m; ++i) X(i, j) -= d * w[(size_t)(i - k)]; }
        }
        return X;
    }
};
struct SPQRProduct { XMat v; void evalTo(XMat& out) const { out = v; } operator XMat() const { return v; } };
struct SPQRQt { const SPQRFactors* f; };
struct SPQRQ { const SPQRFactors* f; SPQRQt transpose() const { return SPQRQt{f}; } };
inline SPQRProduct operator*(const SPQRQt& q, const XMat& b) { return SPQRProduct{q.f->apply(b, true)}; }
inline SPQRProduct operator*(const SPQRQ& q, const XMat& b) { return SPQRProduct{q.f->apply(b, false)}; }
template <typename M> class SPQR {
    SPQRFactors f_;
public:
    void setSPQROrdering(int) {}
    void compute(const XMat& A) { f_.compute(A); }
    SPQRQ matrixQ() const { return SPQRQ{&f_}; }
    const XMat& matrixR() const { return f_.R; }
    int rank() const { int rk = 0; for (int i = 0; i < std::min(f_.R.rows(), f_.R.cols()); ++i) if (f_.R(i, i) != 0.0) ++rk; return rk; }
    int info() const { return 0; }
};
}  // namespace Eigen
#endif
