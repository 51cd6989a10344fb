// lvref_cv.hpp — stand-ins for the few OpenCV types and functions /root/reference/src/ORBDescriptor.cpp touches, so that the
// reference's OWN source file compiles here, unmodified and in place (oracle/Makefile, target _ref/liblvref_orb.so).
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (larvio_amd/, adapter/, include/, examples/) includes or links this.
//
// What is the reference's text in the resulting library: the ORB sampling pattern, the umax table, the mosaic layout, IC_Angle,
// the rotated-BRIEF descriptor and the Hamming distance (ORBDescriptor.cpp:27-514, ORBDescriptor.h:43-59).
// What is NOT the reference's text, because OpenCV is not in this image: the functions below.  They are independent restatements of
// OpenCV's published behaviour [upstream], written without looking at oracle/*.c:
//   cvRound / cvFloor / cvCeil   round-half-to-even (cvtsd2si), floor, ceil                        [core/fast_math.hpp]
//   fastAtan2                     degree-valued 7th-order odd polynomial of min/max, scalar path    [core/mathfuncs_core]
//   copyMakeBorder                REFLECT_101; without BORDER_ISOLATED a ROI's border is first taken from its parent buffer [core/copy.cpp]
//   GaussianBlur (8-bit)          separable filter, kernel = cvRound(256 * float Gaussian), (sum + 2^15) >> 16, border pixels of a ROI
//                                 taken from the parent buffer (OpenCV 3.2, the docker image's version)  [imgproc/smooth.cpp, filter.cpp]
//   resize (INTER_LINEAR)         plain bilinear; only ever fills mosaic levels > 0, which no call site samples (levels are all 0,
//                                 image_processor.cpp:442,677,910)
//   RNG                           the multiply-with-carry generator; only reached for patch sizes != 31 (never)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <climits>
#include <cassert>
#include <memory>
#include <vector>
#include <algorithm>
#include <iostream>

typedef unsigned char uchar;
#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
#define CV_32FC1 5
#define CV_64F 6
#define CV_64FC1 6
inline size_t lvref_elem_size(int type) { return type == CV_8UC3 ? 3 : type == CV_32F ? 4 : type == CV_64F ? 8 : 1; }

inline int cvRound(double v) { return (int)std::nearbyint(v); }            // default rounding mode: half to even
inline int cvRound(float v) { return (int)std::nearbyintf(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { return (int)std::floor(v); }
inline int cvCeil(double v) { return (int)std::ceil(v); }

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    T dot(const Point_& o) const { return (T)(x * o.x + y * o.y); }                       // saturate_cast<_Tp>(x*pt.x + y*pt.y)
    Point_& operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }
    Point_& operator*=(double s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
template <typename T> inline Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>((T)(a.x - b.x), (T)(a.y - b.y)); }
template <typename T> inline Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>((T)(a.x + b.x), (T)(a.y + b.y)); }
template <typename T> inline double norm(const Point_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }
typedef Point_<int> Point;
typedef Point_<float> Point2f;
inline Point2f operator*(const Point2f& p, float s) { return Point2f(p.x * s, p.y * s); }     // saturate_cast<float>(x * s) in OpenCV: the same product

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};

class RNG {
  public:
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffu) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
  private:
    uint64_t state;
};

struct MatExpr;
class _OutputArray;
struct Range { int start, end; Range() : start(0), end(0) {} Range(int s, int e) : start(s), end(e) {} static Range all() { return Range(INT32_MIN, INT32_MAX); } };
inline bool operator==(const Size& a, const Size& b) { return a.width == b.width && a.height == b.height; }
struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
    double operator[](int i) const { return val[i]; }
};
// matrix with OpenCV's header/buffer split: copies share the buffer, operator()(Rect) is a view that remembers where it sits in the
// buffer it was cut from (locateROI), which copyMakeBorder and the filters use for non-isolated borders.  8-bit single channel is what
// the ORB code and the image stages use; CV_64F / CV_32F / CV_8UC3 exist for the calibration matrices and the viewer's picture.
// `aux`: an owner the stand-in functions may hang on a header (the LK pyramid a level belongs to) - shared by copies and views.
class Mat {
  public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    size_t step = 0;
    std::shared_ptr<void> aux;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, const Scalar& v) { create(r, c, type); setTo(v); }
    Mat(int r, int c, int type, void* ext, size_t st) : rows(r), cols(c), data((uchar*)ext), step(st), type_(type), esz(lvref_elem_size(type)), base((uchar*)ext), brows(r), bcols(c) {}
    Mat(const Mat& m, const Rect& r) { *this = m(r); }
    void create(int r, int c, int type)
    {
        type_ = type; esz = lvref_elem_size(type);
        buf = std::make_shared<std::vector<uchar>>((size_t)r * (size_t)c * esz);
        rows = r; cols = c; step = (size_t)c * esz; data = buf->data(); base = data; brows = r; bcols = c;
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    static Mat zeros(int r, int c, int type) { Mat m; m.create(r, c, type); return m; }       // the vector is value-initialised
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    void release() { buf.reset(); aux.reset(); data = base = nullptr; rows = cols = brows = bcols = 0; step = 0; }
    size_t step1() const { return step; }
    int type() const { return type_; }
    size_t elemSize() const { return esz; }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return step == (size_t)cols * esz; }
    Mat clone() const
    {
        Mat m; m.create(rows, cols, type_);
        for (int y = 0; y < rows; ++y) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * esz);
        return m;
    }
    void copyTo(Mat& dst) const { dst = clone(); }
    Mat operator()(const Rect& r) const
    {
        assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
        Mat m = *this;
        m.data = data + (size_t)r.y * step + (size_t)r.x * esz; m.rows = r.height; m.cols = r.width;
        return m;
    }
    Mat row(int y) const { return (*this)(Rect(0, y, cols, 1)); }
    Mat col(int x) const { return (*this)(Rect(x, 0, 1, rows)); }
    Mat operator()(const Range& r, const Range& c) const
    {
        const int r0 = r.start == INT32_MIN ? 0 : r.start, r1 = r.end == INT32_MAX ? rows : r.end, c0 = c.start == INT32_MIN ? 0 : c.start, c1 = c.end == INT32_MAX ? cols : c.end;
        return (*this)(Rect(c0, r0, c1 - c0, r1 - r0));
    }
    // ---- the little dense algebra OpenCV's own recoverPose excerpt (the reference's src/solve_5pts.cpp) is written in: see ref_shim4/lvref_cvalg.hpp
    inline Mat(const MatExpr& e);
    inline Mat& operator=(const MatExpr& e);            // evaluates INTO this header's data when shape and type agree (a view keeps pointing into its parent), else re-allocates
    inline MatExpr t() const;
    inline MatExpr mul(const Mat& o) const;
    inline Mat& operator*=(double a);
    inline Mat& operator/=(const Mat& o);
    inline void convertTo(Mat& dst, int type) const;
    inline void copyTo(const _OutputArray& o) const;
    static Mat eye(int r, int c, int type) { Mat m(r, c, type); for (int i = 0; i < std::min(r, c); ++i) { if (type == CV_64F) m.at<double>(i, i) = 1.0; else if (type == CV_32F) m.at<float>(i, i) = 1.f; else m.at<uchar>(i, i) = 1; } return m; }
    int checkVector(int elemChannels) const { return (channels() == 1 && cols == elemChannels) ? rows : ((rows == 1 || cols == 1) && channels() == elemChannels ? rows * cols : -1); }
    Mat reshape(int cn, int new_rows) const { assert(cn == 1 && channels() == 1 && new_rows == rows); return *this; }
    double get(int y, int x) const { return type_ == CV_64F ? at<double>(y, x) : type_ == CV_32F ? (double)at<float>(y, x) : (double)at<uchar>(y, x); }
    void set(int y, int x, double v) { if (type_ == CV_64F) at<double>(y, x) = v; else if (type_ == CV_32F) at<float>(y, x) = (float)v; else at<uchar>(y, x) = (uchar)std::min(255.0, std::max(0.0, std::nearbyint(v))); }
    Mat& setTo(const Scalar& v)
    {
        for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) {
            uchar* p = data + (size_t)y * step + (size_t)x * esz;
            if (type_ == CV_64F) *(double*)p = v.val[0]; else if (type_ == CV_32F) *(float*)p = (float)v.val[0];
            else for (size_t k = 0; k < esz; ++k) p[k] = (uchar)std::min(255.0, std::max(0.0, std::nearbyint(v.val[k])));
        }
        return *this;
    }
    template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step); }
    void locateROI(Size& whole, Point& ofs) const
    {
        const size_t d = (size_t)(data - base);
        ofs.y = (int)(d / step); ofs.x = (int)((d % step) / esz);
        whole.width = bcols; whole.height = brows;
    }
    // a header over memory somebody else owns, as a view at (ox, oy) of a (bw x bh) buffer that starts at `b` (the LK pyramid's levels)
    static Mat view_in(uchar* b, int bw, int bh, size_t st, int ox, int oy, int w, int h, std::shared_ptr<void> owner)
    {
        Mat m; m.rows = h; m.cols = w; m.step = st; m.data = b + (size_t)oy * st + ox; m.base = b; m.brows = bh; m.bcols = bw; m.aux = owner; return m;
    }
  private:
    int type_ = CV_8U; size_t esz = 1;
    std::shared_ptr<std::vector<uchar>> buf;
    uchar* base = nullptr; int brows = 0, bcols = 0;      // the buffer this header (or the view it was cut from) lives in
};

enum { INTER_LINEAR = 1 };
enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16 };

inline int lvref_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

inline float fastAtan2(float y, float x)
{
    const float scale = (float)(180.0 / CV_PI);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// dst (already allocated by the caller here: a view into the mosaic) <- src framed by REFLECT_101.  Without BORDER_ISOLATED a src that
// is a view first widens into its parent buffer by as much as is there, and only the remainder is reflected (about the widened image).
inline void copyMakeBorder(const Mat& src, Mat& dst, int top, int bottom, int left, int right, int borderType)
{
    assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    assert(dst.rows == src.rows + top + bottom && dst.cols == src.cols + left + right);
    int gl = 0, gt = 0, gr = 0, gb = 0;                     // pixels available in the parent on each side
    if (!(borderType & BORDER_ISOLATED)) {
        Size whole; Point ofs; src.locateROI(whole, ofs);
        gl = std::min(ofs.x, left); gt = std::min(ofs.y, top);
        gr = std::min(whole.width - src.cols - ofs.x, right); gb = std::min(whole.height - src.rows - ofs.y, bottom);
    }
    const int gw = src.cols + gl + gr, gh = src.rows + gt + gb;
    const uchar* g0 = src.data - (ptrdiff_t)gt * (ptrdiff_t)src.step - gl;        // the widened source
    std::vector<uchar> out((size_t)dst.rows * (size_t)dst.cols);                 // src and dst may overlap (they do not here, but keep it safe)
    for (int y = 0; y < dst.rows; ++y) {
        const int sy = lvref_reflect101(y - top + gt, gh);
        for (int x = 0; x < dst.cols; ++x) out[(size_t)y * dst.cols + x] = g0[(size_t)sy * src.step + lvref_reflect101(x - left + gl, gw)];
    }
    for (int y = 0; y < dst.rows; ++y) memcpy(dst.data + (size_t)y * dst.step, &out[(size_t)y * dst.cols], (size_t)dst.cols);
}

inline void resize(const Mat& src, Mat& dst, Size sz, double, double, int)
{   // centre-aligned bilinear; fills mosaic levels that are never sampled
    assert(dst.rows == sz.height && dst.cols == sz.width);
    const double fx = (double)src.cols / sz.width, fy = (double)src.rows / sz.height;
    for (int y = 0; y < sz.height; ++y) {
        double sy = (y + 0.5) * fy - 0.5; int y0 = (int)std::floor(sy); const double wy = sy - y0;
        const int ya = std::min(std::max(y0, 0), src.rows - 1), yb = std::min(std::max(y0 + 1, 0), src.rows - 1);
        for (int x = 0; x < sz.width; ++x) {
            double sx = (x + 0.5) * fx - 0.5; int x0 = (int)std::floor(sx); const double wx = sx - x0;
            const int xa = std::min(std::max(x0, 0), src.cols - 1), xb = std::min(std::max(x0 + 1, 0), src.cols - 1);
            const double v = (1 - wy) * ((1 - wx) * src.at<uchar>(ya, xa) + wx * src.at<uchar>(ya, xb)) + wy * ((1 - wx) * src.at<uchar>(yb, xa) + wx * src.at<uchar>(yb, xb));
            dst.at<uchar>(y, x) = (uchar)std::min(255, std::max(0, cvRound(v)));
        }
    }
}

inline void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sigmaX, double sigmaY, int borderType)
{
    assert(ksize.width == ksize.height && (ksize.width & 1) && sigmaX == sigmaY && borderType == BORDER_REFLECT_101);
    assert(src.rows == dst.rows && src.cols == dst.cols);
    const int n = ksize.width, r = n / 2;
    // getGaussianKernel(n, sigma, CV_32F): exp(-x^2 / (2 sigma^2)) in double, stored as float, normalised by the float entries' double sum
    std::vector<float> kf((size_t)n); double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; kf[(size_t)i] = (float)std::exp(-0.5 * x * x / (sigmaX * sigmaX)); sum += kf[(size_t)i]; }
    std::vector<int> ki((size_t)n);
    for (int i = 0; i < n; ++i) { kf[(size_t)i] = (float)(kf[(size_t)i] * (1.0 / sum)); ki[(size_t)i] = cvRound(kf[(size_t)i] * 256.0f); }
    // rows/columns outside the view: the parent's pixels where it has them, reflection about the widened image otherwise
    Size whole; Point ofs; src.locateROI(whole, ofs);
    const int gl = std::min(ofs.x, r), gt = std::min(ofs.y, r), gr = std::min(whole.width - src.cols - ofs.x, r), gb = std::min(whole.height - src.rows - ofs.y, r);
    const int gw = src.cols + gl + gr, gh = src.rows + gt + gb;
    const uchar* g0 = src.data - (ptrdiff_t)gt * (ptrdiff_t)src.step - gl;
    const int W = src.cols, H = src.rows;
    std::vector<int> hor((size_t)(H + 2 * r) * (size_t)W);
    for (int y = -r; y < H + r; ++y) {
        const uchar* row = g0 + (size_t)lvref_reflect101(y + gt, gh) * src.step;
        for (int x = 0; x < W; ++x) {
            int acc = 0;
            for (int k = 0; k < n; ++k) acc += ki[(size_t)k] * row[lvref_reflect101(x + k - r + gl, gw)];
            hor[(size_t)(y + r) * W + x] = acc;
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int acc = 0;
            for (int k = 0; k < n; ++k) acc += ki[(size_t)k] * hor[(size_t)(y + k) * W + x];
            const int v = (acc + (1 << 15)) >> 16;
            dst.at<uchar>(y, x) = (uchar)std::min(255, std::max(0, v));
        }
}

}   // namespace cv
