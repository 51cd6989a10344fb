// lvref_eigen.hpp — ORACLE / TEST INFRASTRUCTURE ONLY.  A stand-in for the slice of Eigen 3 that /root/reference/include/larvio/feature.hpp,
// imu_state.h and math_utils.hpp use, so that those reference headers can be compiled WHERE THEY LIE (oracle/Makefile, target `ref`) on a
// machine without Eigen.  Fixed-size dense matrices with eager evaluation, the Isometry3d / Quaterniond members the headers call, and a
// pivoted LDL^T for the 3x3 solve.  Inner products run over the summation index in ascending order, as Eigen's coefficient-wise lazy
// products of small fixed-size matrices do - but no claim is made about Eigen's rounding: what the compiled reference pins is its
// ALGORITHM TEXT (initial guess, Levenberg-Marquardt schedule, Huber weights, validity tests, frame and quaternion conventions), and the
// tests compare at 1e-9, not bit for bit.  Nothing in the product includes this file.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <utility>
#include <type_traits>
#include <memory>
#include <vector>
#include <map>
#include <algorithm>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

template <typename T> using aligned_allocator = std::allocator<T>;

template <typename T, int R, int C> class Matrix;
class MatrixXd;

// assignable view of a block of a parent matrix (row(), leftCols<>(), rightCols<>(), head<>())
template <typename T, int PR, int PC, int BR, int BC> class BlockRef {
public:
    BlockRef(T* base, int r0, int c0) : b_(base), r0_(r0), c0_(c0) {}
    T& at(int i, int j) const { return b_[(size_t)(c0_ + j) * PR + (r0_ + i)]; }
    Matrix<T, BR, BC> eval() const;
    operator Matrix<T, BR, BC>() const { return eval(); }
    BlockRef& operator=(const Matrix<T, BR, BC>& m);
    BlockRef& operator=(const BlockRef& o) { return *this = o.eval(); }
    BlockRef& operator=(const MatrixXd& m);
    template <int QR, int QC> BlockRef& operator=(const BlockRef<T, QR, QC, BR, BC>& o) { return *this = o.eval(); }
    T& operator()(int i) const { return BR == 1 ? at(0, i) : at(i, 0); }
    T& operator()(int i, int j) const { return at(i, j); }
    friend Matrix<T, BR, BC> operator*(T s, const BlockRef& b) { return s * b.eval(); }
    friend Matrix<T, BR, BC> operator*(const BlockRef& b, T s) { return b.eval() * s; }
    friend Matrix<T, BR, BC> operator/(const BlockRef& b, T s) { return b.eval() / s; }
    friend Matrix<T, BR, BC> operator+(const BlockRef& a, const BlockRef& b) { return a.eval() + b.eval(); }
    friend Matrix<T, BR, BC> operator-(const BlockRef& a, const BlockRef& b) { return a.eval() - b.eval(); }
    friend Matrix<T, BR, BC> operator+(const Matrix<T, BR, BC>& a, const BlockRef& b) { return a + b.eval(); }
    friend Matrix<T, BR, BC> operator-(const Matrix<T, BR, BC>& a, const BlockRef& b) { return a - b.eval(); }
    friend Matrix<T, BR, BC> operator+(const BlockRef& a, const Matrix<T, BR, BC>& b) { return a.eval() + b; }
    friend Matrix<T, BR, BC> operator-(const BlockRef& a, const Matrix<T, BR, BC>& b) { return a.eval() - b; }
private:
    T* b_; int r0_, c0_;
};

template <typename T, int N> struct LDLT3 {
    // symmetric pivoting as Eigen's LDLT (largest remaining |diagonal| first); N x N, N <= 4
    T L[N][N]; T D[N]; int perm[N];
    explicit LDLT3(const Matrix<T, N, N>& A);
    Matrix<T, N, 1> solve(const Matrix<T, N, 1>& b) const;
};

template <typename T, int R, int C> class Matrix {
public:
    Matrix() { for (int i = 0; i < R * C; ++i) d_[i] = T(0); }
    Matrix(T x, T y) { static_assert(R * C == 2, "2-vector"); d_[0] = x; d_[1] = y; }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, "3-vector"); d_[0] = x; d_[1] = y; d_[2] = z; }
    Matrix(T x, T y, T z, T w) { static_assert(R * C == 4, "4-vector"); d_[0] = x; d_[1] = y; d_[2] = z; d_[3] = w; }
    template <int PR, int PC> Matrix(const BlockRef<T, PR, PC, R, C>& b) { *this = b.eval(); }
    explicit Matrix(const MatrixXd& m);                 // sizes must agree (NaN otherwise)
    Matrix& operator=(const MatrixXd& m) { *this = Matrix(m); return *this; }
    template <class XB, typename = decltype(std::declval<const XB&>().eval())> Matrix& operator=(const XB& b) { *this = Matrix(b.eval()); return *this; }
    Matrix& operator+=(const MatrixXd& m) { *this += Matrix(m); return *this; }
    Matrix& operator-=(const MatrixXd& m) { *this -= Matrix(m); return *this; }
    bool operator==(const Matrix& o) const { for (int i = 0; i < R * C; ++i) if (d_[i] != o.d_[i]) return false; return true; }
    template <int N> Matrix<T, N, 1> tail() const { Matrix<T, N, 1> m; for (int i = 0; i < N; ++i) m(i) = d_[R * C - N + i]; return m; }
    template <int N> Matrix<T, N, 1> segment(int i0) const { Matrix<T, N, 1> m; for (int i = 0; i < N; ++i) m(i) = d_[i0 + i]; return m; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
    T& operator()(int i, int j) { return d_[(size_t)j * R + i]; }
    const T& operator()(int i, int j) const { return d_[(size_t)j * R + i]; }
    T& operator()(int i) { return d_[i]; }
    const T& operator()(int i) const { return d_[i]; }
    T& operator[](int i) { return d_[i]; }
    const T& operator[](int i) const { return d_[i]; }
    T& x() { return d_[0]; } T& y() { return d_[1]; } T& z() { return d_[2]; }
    const T& x() const { return d_[0]; } const T& y() const { return d_[1]; } const T& z() const { return d_[2]; }
    T* data() { return d_; }
    const T* data() const { return d_; }
    int rows() const { return R; }
    int cols() const { return C; }
    int size() const { return R * C; }
    // a 1x1 result is a scalar (double depth = (A^T A)^-1 A^T b)
    template <int RR = R, int CC = C, typename = typename std::enable_if<RR == 1 && CC == 1>::type> operator T() const { return d_[0]; }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> m; for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) m(j, i) = (*this)(i, j); return m; }
    T squaredNorm() const { T s = d_[0] * d_[0]; for (int i = 1; i < R * C; ++i) s += d_[i] * d_[i]; return s; }
    T norm() const { return std::sqrt(squaredNorm()); }
    void normalize() { const T n = norm(); for (int i = 0; i < R * C; ++i) d_[i] /= n; }
    Matrix normalized() const { Matrix m = *this; m.normalize(); return m; }
    T trace() const { T s = (*this)(0, 0); for (int i = 1; i < (R < C ? R : C); ++i) s += (*this)(i, i); return s; }
    T dot(const Matrix& o) const { T s = d_[0] * o.d_[0]; for (int i = 1; i < R * C; ++i) s += d_[i] * o.d_[i]; return s; }
    T maxCoeff(int* idx) const { int b = 0; for (int i = 1; i < R * C; ++i) if (d_[i] > d_[b]) b = i; *idx = b; return d_[b]; }
    T maxCoeff() const { int i; return maxCoeff(&i); }
    T maxCoeff(int* row, int* col) const { int br = 0, bc = 0; for (int j = 0; j < C; ++j) for (int i = 0; i < R; ++i) if ((*this)(i, j) > (*this)(br, bc)) { br = i; bc = j; } *row = br; *col = bc; return (*this)(br, bc); }
    Matrix inverse() const { static_assert(R == 1 && C == 1, "only the 1x1 inverse is needed"); Matrix m; m.d_[0] = T(1) / d_[0]; return m; }
    Matrix<T, 3, 1> cross(const Matrix<T, 3, 1>& o) const
    { return Matrix<T, 3, 1>(d_[1] * o(2) - d_[2] * o(1), d_[2] * o(0) - d_[0] * o(2), d_[0] * o(1) - d_[1] * o(0)); }
    void setZero() { for (int i = 0; i < R * C; ++i) d_[i] = T(0); }
    void setIdentity() { *this = Identity(); }
    LDLT3<T, R> ldlt() const { static_assert(R == C, "square"); return LDLT3<T, R>(*this); }
    // blocks
    BlockRef<T, R, C, 1, C> row(int i) { return BlockRef<T, R, C, 1, C>(d_, i, 0); }
    Matrix<T, 1, C> row(int i) const { Matrix<T, 1, C> m; for (int j = 0; j < C; ++j) m(0, j) = (*this)(i, j); return m; }
    BlockRef<T, R, C, R, 1> col(int j) { return BlockRef<T, R, C, R, 1>(d_, 0, j); }
    Matrix<T, R, 1> col(int j) const { Matrix<T, R, 1> m; for (int i = 0; i < R; ++i) m(i) = (*this)(i, j); return m; }
    template <int N> BlockRef<T, R, C, R, N> leftCols() { return BlockRef<T, R, C, R, N>(d_, 0, 0); }
    template <int N> Matrix<T, R, N> leftCols() const { Matrix<T, R, N> m; for (int i = 0; i < R; ++i) for (int j = 0; j < N; ++j) m(i, j) = (*this)(i, j); return m; }
    template <int N> BlockRef<T, R, C, R, N> rightCols() { return BlockRef<T, R, C, R, N>(d_, 0, C - N); }
    template <int N> Matrix<T, R, N> rightCols() const { Matrix<T, R, N> m; for (int i = 0; i < R; ++i) for (int j = 0; j < N; ++j) m(i, j) = (*this)(i, C - N + j); return m; }
    template <int N> BlockRef<T, R, C, N, 1> head() { static_assert(C == 1, "vector"); return BlockRef<T, R, C, N, 1>(d_, 0, 0); }
    template <int N> Matrix<T, N, 1> head() const { Matrix<T, N, 1> m; for (int i = 0; i < N; ++i) m(i) = d_[i]; return m; }
    template <int BR_, int BC_> BlockRef<T, R, C, BR_, BC_> block(int r0, int c0) { return BlockRef<T, R, C, BR_, BC_>(d_, r0, c0); }
    template <int BR_, int BC_> Matrix<T, BR_, BC_> block(int r0, int c0) const { Matrix<T, BR_, BC_> m; for (int i = 0; i < BR_; ++i) for (int j = 0; j < BC_; ++j) m(i, j) = (*this)(r0 + i, c0 + j); return m; }
    // arithmetic
    Matrix operator-() const { Matrix m; for (int i = 0; i < R * C; ++i) m.d_[i] = -d_[i]; return m; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d_[i] += o.d_[i]; return *this; }
    Matrix& operator-=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d_[i] -= o.d_[i]; return *this; }
    Matrix& operator*=(T s) { for (int i = 0; i < R * C; ++i) d_[i] *= s; return *this; }
    Matrix& operator/=(T s) { for (int i = 0; i < R * C; ++i) d_[i] /= s; return *this; }
    friend Matrix operator+(const Matrix& a, const Matrix& b) { Matrix m; for (int i = 0; i < R * C; ++i) m.d_[i] = a.d_[i] + b.d_[i]; return m; }
    friend Matrix operator-(const Matrix& a, const Matrix& b) { Matrix m; for (int i = 0; i < R * C; ++i) m.d_[i] = a.d_[i] - b.d_[i]; return m; }
    friend Matrix operator*(T s, const Matrix& a) { Matrix m; for (int i = 0; i < R * C; ++i) m.d_[i] = s * a.d_[i]; return m; }
    friend Matrix operator*(const Matrix& a, T s) { Matrix m; for (int i = 0; i < R * C; ++i) m.d_[i] = a.d_[i] * s; return m; }
    friend Matrix operator/(const Matrix& a, T s) { Matrix m; for (int i = 0; i < R * C; ++i) m.d_[i] = a.d_[i] / s; return m; }
    template <int K> Matrix<T, R, K> operator*(const Matrix<T, C, K>& o) const
    {
        Matrix<T, R, K> m;
        for (int i = 0; i < R; ++i) for (int k = 0; k < K; ++k) { T s = (*this)(i, 0) * o(0, k); for (int j = 1; j < C; ++j) s += (*this)(i, j) * o(j, k); m(i, k) = s; }
        return m;
    }
    // comma initialiser (row-major fill, as Eigen's)
    struct Comma {
        Matrix& m; int n;
        Comma& operator,(T v) { m((n / C), (n % C)) = v; ++n; return *this; }
    };
    Comma operator<<(T v) { (*this)(0, 0) = v; return Comma{*this, 1}; }
private:
    T d_[R * C];
};

template <typename T, int PR, int PC, int BR, int BC> Matrix<T, BR, BC> BlockRef<T, PR, PC, BR, BC>::eval() const
{ Matrix<T, BR, BC> m; for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) m(i, j) = at(i, j); return m; }
template <typename T, int PR, int PC, int BR, int BC> BlockRef<T, PR, PC, BR, BC>& BlockRef<T, PR, PC, BR, BC>::operator=(const Matrix<T, BR, BC>& m)
{ for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) at(i, j) = m(i, j); return *this; }

template <typename T, int N> LDLT3<T, N>::LDLT3(const Matrix<T, N, N>& A0)
{
    T A[N][N];
    for (int i = 0; i < N; ++i) { perm[i] = i; for (int j = 0; j < N; ++j) { A[i][j] = A0(i, j); L[i][j] = T(0); } }
    for (int k = 0; k < N; ++k) {
        int p = k; for (int i = k + 1; i < N; ++i) if (std::fabs(A[i][i]) > std::fabs(A[p][p])) p = i;
        if (p != k) {                                   // symmetric row/column swap of the trailing matrix and of the rows of L built so far
            for (int j = 0; j < N; ++j) std::swap(A[k][j], A[p][j]);
            for (int i = 0; i < N; ++i) std::swap(A[i][k], A[i][p]);
            for (int j = 0; j < k; ++j) std::swap(L[k][j], L[p][j]);
            std::swap(perm[k], perm[p]);
        }
        D[k] = A[k][k]; L[k][k] = T(1);
        for (int i = k + 1; i < N; ++i) L[i][k] = A[i][k] / D[k];
        for (int i = k + 1; i < N; ++i) for (int j = k + 1; j < N; ++j) A[i][j] -= L[i][k] * D[k] * L[j][k];
    }
}
template <typename T, int N> Matrix<T, N, 1> LDLT3<T, N>::solve(const Matrix<T, N, 1>& b) const
{
    T y[N], x[N];
    for (int i = 0; i < N; ++i) { T s = b(perm[i]); for (int j = 0; j < i; ++j) s -= L[i][j] * y[j]; y[i] = s; }
    for (int i = 0; i < N; ++i) y[i] /= D[i];
    for (int i = N - 1; i >= 0; --i) { T s = y[i]; for (int j = i + 1; j < N; ++j) s -= L[j][i] * x[j]; x[i] = s; }
    Matrix<T, N, 1> out; for (int i = 0; i < N; ++i) out(perm[i]) = x[i];
    return out;
}

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;

// Eigen::Quaterniond: Hamilton convention, constructor order (w, x, y, z); toRotationMatrix as Eigen's QuaternionBase::toRotationMatrix
class Quaterniond {
public:
    Quaterniond() : w_(1), x_(0), y_(0), z_(0) {}
    Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
    explicit Quaterniond(const Matrix3d& m)
    {   // Eigen's quaternionbase_assign_impl<Matrix3> (Shoemake): the branch on the trace, then on the largest diagonal entry
        double q[4];                                        // x y z w
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (m(2, 1) - m(1, 2)) * t; q[1] = (m(0, 2) - m(2, 0)) * t; q[2] = (m(1, 0) - m(0, 1)) * t; }
        else {
            int i = 0; if (m(1, 1) > m(0, 0)) i = 1; if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
            q[i] = 0.5 * t; t = 0.5 / t;
            q[3] = (m(k, j) - m(j, k)) * t; q[j] = (m(j, i) + m(i, j)) * t; q[k] = (m(k, i) + m(i, k)) * t;
        }
        x_ = q[0]; y_ = q[1]; z_ = q[2]; w_ = q[3];
    }
    double& w() { return w_; } double& x() { return x_; } double& y() { return y_; } double& z() { return z_; }
    double w() const { return w_; } double x() const { return x_; } double y() const { return y_; } double z() const { return z_; }
    void normalize() { const double n = std::sqrt(w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_); w_ /= n; x_ /= n; y_ /= n; z_ /= n; }
    Quaterniond normalized() const { Quaterniond q = *this; q.normalize(); return q; }
    static Quaterniond Identity() { return Quaterniond(1, 0, 0, 0); }
    Vector4d coeffs() const { return Vector4d(x_, y_, z_, w_); }
    // Eigen's QuaternionBase::setFromTwoVectors for vectors that are not opposite: axis = v0 x v1, s = sqrt(2 (1 + v0.v1)), (axis / s, s / 2)
    static Quaterniond FromTwoVectors(const Vector3d& a, const Vector3d& b)
    {
        const Vector3d v0 = a.normalized(), v1 = b.normalized();
        const double c = v1.dot(v0);
        if (c < -1.0 + 1e-12) return Quaterniond(0, 1, 0, 0);           // (opposite vectors: Eigen takes an SVD branch; not reached by the callers compiled here)
        const Vector3d axis = v0.cross(v1);
        const double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
        return Quaterniond(s * 0.5, axis(0) * invs, axis(1) * invs, axis(2) * invs);
    }
    void setIdentity() { w_ = 1; x_ = y_ = z_ = 0; }
    Vector3d vec() const { return Vector3d(x_, y_, z_); }
    double squaredNorm() const { return w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_; }
    Quaterniond conjugate() const { return Quaterniond(w_, -x_, -y_, -z_); }
    Quaterniond inverse() const { const double n2 = squaredNorm(); return Quaterniond(w_ / n2, -x_ / n2, -y_ / n2, -z_ / n2); }
    // Hamilton product, as Eigen's internal::quat_product
    Quaterniond operator*(const Quaterniond& b) const
    {
        return Quaterniond(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_, w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                           w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_, w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
    }
    // Eigen's QuaternionBase::_transformVector: v + w uv + vec x uv with uv = 2 vec x v (the rotation only for a unit quaternion)
    Vector3d operator*(const Vector3d& v) const
    {
        Vector3d uv = vec().cross(v); uv += uv;
        return v + w_ * uv + vec().cross(uv);
    }
    Matrix3d toRotationMatrix() const
    {
        Matrix3d res;
        const double tx = 2 * x_, ty = 2 * y_, tz = 2 * z_;
        const double twx = tx * w_, twy = ty * w_, twz = tz * w_;
        const double txx = tx * x_, txy = ty * x_, txz = tz * x_;
        const double tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
        res(0, 0) = 1 - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = 1 - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = 1 - (txx + tyy);
        return res;
    }
private:
    double w_, x_, y_, z_;
};

// Eigen::Isometry3d: rotation + translation, p -> R p + t
class Isometry3d {
public:
    Isometry3d() : R_(Matrix3d::Identity()), t_() {}
    static Isometry3d Identity() { return Isometry3d(); }
    Matrix3d& linear() { return R_; }
    const Matrix3d& linear() const { return R_; }
    Vector3d& translation() { return t_; }
    const Vector3d& translation() const { return t_; }
    Isometry3d inverse() const { Isometry3d o; o.R_ = R_.transpose(); o.t_ = -(o.R_ * t_); return o; }
    Isometry3d operator*(const Isometry3d& o) const { Isometry3d r; r.R_ = R_ * o.R_; r.t_ = R_ * o.t_ + t_; return r; }
    Vector3d operator*(const Vector3d& p) const { return R_ * p + t_; }
private:
    Matrix3d R_; Vector3d t_;
};

// dynamic-size matrix / vector (ImuPreintegration.h builds its 15x15 / 15x18 step matrices as MatrixXd, initial_alignment.cpp its normal
// equations as MatrixXd / VectorXd); column-major, eager.  VectorXd is the same class with one column.
class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
    explicit MatrixXd(int n) : r_(n), c_(1), d_((size_t)n, 0.0) {}
    template <int R, int C> MatrixXd(const Matrix<double, R, C>& m) : r_(R), c_(C), d_((size_t)R * C) { for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) (*this)(i, j) = m(i, j); }
    template <int PR, int PC, int BR, int BC> MatrixXd(const BlockRef<double, PR, PC, BR, BC>& b) : MatrixXd(b.eval()) {}
    static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
    static MatrixXd Identity(int r, int c) { MatrixXd m(r, c); for (int i = 0; i < (r < c ? r : c); ++i) m(i, i) = 1.0; return m; }
    int rows() const { return r_; }
    int cols() const { return c_; }
    int size() const { return r_ * c_; }
    void setZero() { std::fill(d_.begin(), d_.end(), 0.0); }
    double& operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
    const double& operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
    double& operator()(int i) { return d_[(size_t)i]; }
    const double& operator()(int i) const { return d_[(size_t)i]; }
    MatrixXd transpose() const { MatrixXd m(c_, r_); for (int i = 0; i < r_; ++i) for (int j = 0; j < c_; ++j) m(j, i) = (*this)(i, j); return m; }
    double norm() const { double s = 0; for (double v : d_) s += v * v; return std::sqrt(s); }
    // fixed-size block of a dynamic matrix, assignable
    template <int BR, int BC> struct XBlock {
        MatrixXd& m; int r0, c0;
        Matrix<double, BR, BC> eval() const { Matrix<double, BR, BC> o; for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) o(i, j) = m(r0 + i, c0 + j); return o; }
        operator Matrix<double, BR, BC>() const { return eval(); }
        XBlock& operator=(const Matrix<double, BR, BC>& v) { for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) m(r0 + i, c0 + j) = v(i, j); return *this; }
        XBlock& operator=(const MatrixXd& v) { for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) m(r0 + i, c0 + j) = v(i, j); return *this; }
        XBlock& operator=(const XBlock& v) { return *this = v.eval(); }
        XBlock& operator+=(const Matrix<double, BR, BC>& v) { for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) m(r0 + i, c0 + j) += v(i, j); return *this; }
        double& operator()(int i) const { return BR == 1 ? m(r0, c0 + i) : m(r0 + i, c0); }
        double& operator()(int i, int j) const { return m(r0 + i, c0 + j); }
    };
    template <int BR, int BC> XBlock<BR, BC> block(int r0, int c0) { return XBlock<BR, BC>{*this, r0, c0}; }
    template <int BR, int BC> XBlock<BR, BC> topLeftCorner() { return XBlock<BR, BC>{*this, 0, 0}; }
    template <int BR, int BC> XBlock<BR, BC> topRightCorner() { return XBlock<BR, BC>{*this, 0, c_ - BC}; }
    template <int BR, int BC> XBlock<BR, BC> bottomLeftCorner() { return XBlock<BR, BC>{*this, r_ - BR, 0}; }
    template <int BR, int BC> XBlock<BR, BC> bottomRightCorner() { return XBlock<BR, BC>{*this, r_ - BR, c_ - BC}; }
    template <int N> XBlock<N, 1> segment(int i0) { return XBlock<N, 1>{*this, i0, 0}; }
    template <int N> XBlock<N, 1> head() { return XBlock<N, 1>{*this, 0, 0}; }
    template <int N> XBlock<N, 1> tail() { return XBlock<N, 1>{*this, r_ - N, 0}; }
    template <int BR, int BC> MatrixXd(const XBlock<BR, BC>& b) : MatrixXd(b.eval()) {}
    // symmetric pivoting LDL^T, as Eigen's LDLT
    struct XLDLT {
        int n; std::vector<double> L, D; std::vector<int> perm;
        MatrixXd solve(const MatrixXd& b) const
        {
            std::vector<double> y((size_t)n), x((size_t)n);
            for (int i = 0; i < n; ++i) { double s = b(perm[(size_t)i]); for (int j = 0; j < i; ++j) s -= L[(size_t)i * n + j] * y[(size_t)j]; y[(size_t)i] = s; }
            for (int i = 0; i < n; ++i) y[(size_t)i] /= D[(size_t)i];
            for (int i = n - 1; i >= 0; --i) { double s = y[(size_t)i]; for (int j = i + 1; j < n; ++j) s -= L[(size_t)j * n + i] * x[(size_t)j]; x[(size_t)i] = s; }
            MatrixXd out(n); for (int i = 0; i < n; ++i) out(perm[(size_t)i]) = x[(size_t)i];
            return out;
        }
    };
    XLDLT ldlt() const
    {
        const int n = r_; XLDLT f; f.n = n; f.L.assign((size_t)n * n, 0.0); f.D.assign((size_t)n, 0.0); f.perm.resize((size_t)n);
        std::vector<double> A((size_t)n * n);
        for (int i = 0; i < n; ++i) { f.perm[(size_t)i] = i; for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = (*this)(i, j); }
        for (int k = 0; k < n; ++k) {
            int p = k; for (int i = k + 1; i < n; ++i) if (std::fabs(A[(size_t)i * n + i]) > std::fabs(A[(size_t)p * n + p])) p = i;
            if (p != k) {
                for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
                for (int i = 0; i < n; ++i) std::swap(A[(size_t)i * n + k], A[(size_t)i * n + p]);
                for (int j = 0; j < k; ++j) std::swap(f.L[(size_t)k * n + j], f.L[(size_t)p * n + j]);
                std::swap(f.perm[(size_t)k], f.perm[(size_t)p]);
            }
            f.D[(size_t)k] = A[(size_t)k * n + k]; f.L[(size_t)k * n + k] = 1.0;
            for (int i = k + 1; i < n; ++i) f.L[(size_t)i * n + k] = A[(size_t)i * n + k] / f.D[(size_t)k];
            for (int i = k + 1; i < n; ++i) for (int j = k + 1; j < n; ++j) A[(size_t)i * n + j] -= f.L[(size_t)i * n + k] * f.D[(size_t)k] * f.L[(size_t)j * n + k];
        }
        return f;
    }
    friend MatrixXd operator*(const MatrixXd& a, const MatrixXd& b)
    {
        MatrixXd m(a.r_, b.c_);
        for (int i = 0; i < a.r_; ++i) for (int k = 0; k < b.c_; ++k) { double s = a(i, 0) * b(0, k); for (int j = 1; j < a.c_; ++j) s += a(i, j) * b(j, k); m(i, k) = s; }
        return m;
    }
    friend MatrixXd operator+(const MatrixXd& a, const MatrixXd& b) { MatrixXd m(a.r_, a.c_); for (size_t i = 0; i < m.d_.size(); ++i) m.d_[i] = a.d_[i] + b.d_[i]; return m; }
    friend MatrixXd operator-(const MatrixXd& a, const MatrixXd& b) { MatrixXd m(a.r_, a.c_); for (size_t i = 0; i < m.d_.size(); ++i) m.d_[i] = a.d_[i] - b.d_[i]; return m; }
    friend MatrixXd operator*(double s, const MatrixXd& a) { MatrixXd m(a.r_, a.c_); for (size_t i = 0; i < m.d_.size(); ++i) m.d_[i] = s * a.d_[i]; return m; }
    friend MatrixXd operator*(const MatrixXd& a, double s) { MatrixXd m(a.r_, a.c_); for (size_t i = 0; i < m.d_.size(); ++i) m.d_[i] = a.d_[i] * s; return m; }
    friend MatrixXd operator/(const MatrixXd& a, double s) { MatrixXd m(a.r_, a.c_); for (size_t i = 0; i < m.d_.size(); ++i) m.d_[i] = a.d_[i] / s; return m; }
private:
    int r_, c_; std::vector<double> d_;
};
typedef MatrixXd VectorXd;
// mixed fixed / dynamic arithmetic (exact-match templates: no implicit conversions between the two kinds, so no ambiguities)
template <int R, int C> MatrixXd operator*(const MatrixXd& a, const Matrix<double, R, C>& b) { return a * MatrixXd(b); }
template <int R, int C> MatrixXd operator*(const Matrix<double, R, C>& a, const MatrixXd& b) { return MatrixXd(a) * b; }
template <int R, int C> Matrix<double, R, C> operator+(const Matrix<double, R, C>& a, const MatrixXd& b) { return a + Matrix<double, R, C>(b); }
template <int R, int C> Matrix<double, R, C> operator-(const Matrix<double, R, C>& a, const MatrixXd& b) { return a - Matrix<double, R, C>(b); }
template <typename T, int R, int C> Matrix<T, R, C>::Matrix(const MatrixXd& m) { for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) (*this)(i, j) = (m.rows() == R && m.cols() == C) ? m(i, j) : std::nan(""); }
template <typename T, int PR, int PC, int BR, int BC> BlockRef<T, PR, PC, BR, BC>& BlockRef<T, PR, PC, BR, BC>::operator=(const MatrixXd& m)
{ for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) at(i, j) = m(i, j); return *this; }

}  // namespace Eigen
