// lvref_cv3.hpp - TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  The OpenCV names the reference's FRONT-END is written against, so that
// /root/reference/src/image_processor.cpp (with src/ORBDescriptor.cpp) can be compiled where it lies (oracle/Makefile, target `ref` ->
// oracle/_ref/liblvref_imgproc.so).  OpenCV is not installed.  What stands behind the names here:
//   * the IMAGE ALGORITHMS - cv::createCLAHE()->apply, buildOpticalFlowPyramid, calcOpticalFlowPyrLK, goodFeaturesToTrack,
//     findFundamentalMat, undistortPoints (+ fisheye) - are the ORACLE's restatements of those OpenCV functions (oracle/fe_image.c,
//     fe_track.c: lvo_clahe_u8, lvo_pyramid_build, lvo_lk_track, lvo_good_features, lvo_find_fundamental_mask, lvo_undistort_points),
//     called through oracle/lvo.h.  So the library this header helps to build does NOT pin those algorithms to OpenCV - nothing in this
//     image can - it pins everything AROUND them to the reference's own text: ImageProcessor::processImage's state machine and publish
//     cadence, the forward / reverse / descriptor / RANSAC gates of trackFeatures and trackNewFeatures in their order, every
//     removeUnmarkedElements, ids / lifetimes / init points, the detection mask, getFeatureMsg's undistortion and velocities;
//   * the small fixed-size algebra (Matx33f/d, Vec, Point2f, Rodrigues) is written here from OpenCV's documented semantics
//     (saturate_cast on every assignment, Matx33 inverse by cofactors, Rodrigues in double) - independently of oracle/*.c;
//   * the ORB code is the reference's own (ORBDescriptor.cpp) over the stand-ins of ../ref_shim/lvref_cv.hpp, as in liblvref_orb.so;
//   * drawing and window calls do nothing.
#pragma once
#include "../ref_shim/lvref_cv.hpp"
#include <string>
#include <map>
#include <fstream>
#include <sstream>
#include <cstdio>
#include <cstdlib>
#include <cctype>
extern "C" {
#include "../lvo.h"
}

#define LVCV_CHECK(c, what) do { if (!(c)) { std::fprintf(stderr, "lvref_cv3: %s (%s:%d)\n", what, __FILE__, __LINE__); std::abort(); } } while (0)

namespace cv {

// ------------------------------------------------------------------ small algebra
template <typename T, int N> struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
    Vec(T a, T b) { static_assert(N == 2, ""); val[0] = a; val[1] = b; }
    Vec(T a, T b, T c) { static_assert(N == 3, ""); val[0] = a; val[1] = b; val[2] = c; }
    Vec(T a, T b, T c, T d) { static_assert(N == 4, ""); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
    template <typename U> Vec(const Vec<U, N>& o) { for (int i = 0; i < N; ++i) val[i] = (T)o.val[i]; }          // saturate_cast
    Vec(const Mat& m) { LVCV_CHECK(m.rows * m.cols == N && m.type() == CV_64F, "Vec from a Mat of another shape"); for (int i = 0; i < N; ++i) val[i] = (T)m.at<double>(m.cols == 1 ? i : 0, m.cols == 1 ? 0 : i); }
    T& operator[](int i) { return val[i]; } const T& operator[](int i) const { return val[i]; }
    T& operator()(int i) { return val[i]; } const T& operator()(int i) const { return val[i]; }
    Vec& operator+=(const Vec& o) { for (int i = 0; i < N; ++i) val[i] = (T)(val[i] + o.val[i]); return *this; }
    Vec& operator-=(const Vec& o) { for (int i = 0; i < N; ++i) val[i] = (T)(val[i] - o.val[i]); return *this; }
    Vec& operator*=(float a) { for (int i = 0; i < N; ++i) val[i] = (T)(val[i] * a); return *this; }
    Vec& operator*=(double a) { for (int i = 0; i < N; ++i) val[i] = (T)(val[i] * a); return *this; }
    Vec& operator*=(int a) { for (int i = 0; i < N; ++i) val[i] = (T)(val[i] * a); return *this; }
};
template <typename T, int N> inline Vec<T, N> operator*(const Vec<T, N>& v, double a) { Vec<T, N> o; for (int i = 0; i < N; ++i) o.val[i] = (T)(v.val[i] * a); return o; }
template <typename T, int N> inline Vec<T, N> operator*(const Vec<T, N>& v, float a) { Vec<T, N> o; for (int i = 0; i < N; ++i) o.val[i] = (T)(v.val[i] * a); return o; }
template <typename T, int N> inline Vec<T, N> operator*(double a, const Vec<T, N>& v) { return v * a; }
template <typename T, int N> inline Vec<T, N> operator-(const Vec<T, N>& v) { Vec<T, N> o; for (int i = 0; i < N; ++i) o.val[i] = -v.val[i]; return o; }
template <typename T, int N> inline Vec<T, N> operator+(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> o = a; o += b; return o; }
template <typename T, int N> inline Vec<T, N> operator-(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> o = a; o -= b; return o; }
typedef Vec<float, 3> Vec3f; typedef Vec<double, 3> Vec3d; typedef Vec<double, 4> Vec4d; typedef Vec<int, 2> Vec2i; typedef Vec<float, 2> Vec2f; typedef Vec<double, 2> Vec2d;

template <typename T> struct Matx33 {
    T val[9];
    Matx33() { for (T& v : val) v = T(0); }
    Matx33(T a, T b, T c, T d, T e, T f, T g, T h, T i) : val{a, b, c, d, e, f, g, h, i} {}
    Matx33(const Mat& m) { LVCV_CHECK(m.rows == 3 && m.cols == 3 && m.type() == CV_64F, "Matx33 from a Mat of another shape"); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) val[3 * i + j] = (T)m.at<double>(i, j); }
    template <typename U> Matx33(const Matx33<U>& o) { for (int i = 0; i < 9; ++i) val[i] = (T)o.val[i]; }
    static Matx33 eye() { return Matx33(1, 0, 0, 0, 1, 0, 0, 0, 1); }
    T& operator()(int i, int j) { return val[3 * i + j]; } const T& operator()(int i, int j) const { return val[3 * i + j]; }
    Matx33 t() const { Matx33 o; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.val[3 * i + j] = val[3 * j + i]; return o; }
    Matx33 inv() const
    {   // Matx_FastInvOp<_Tp, 3, 3>: determinant and cofactors in _Tp
        const T* a = val; Matx33 b;
        T d = a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[6] * a[5]) + a[2] * (a[3] * a[7] - a[6] * a[4]);
        if (d == 0) return b;
        d = 1 / d;
        b.val[0] = (a[4] * a[8] - a[5] * a[7]) * d; b.val[1] = (a[2] * a[7] - a[1] * a[8]) * d; b.val[2] = (a[1] * a[5] - a[2] * a[4]) * d;
        b.val[3] = (a[5] * a[6] - a[3] * a[8]) * d; b.val[4] = (a[0] * a[8] - a[2] * a[6]) * d; b.val[5] = (a[2] * a[3] - a[0] * a[5]) * d;
        b.val[6] = (a[3] * a[7] - a[4] * a[6]) * d; b.val[7] = (a[1] * a[6] - a[0] * a[7]) * d; b.val[8] = (a[0] * a[4] - a[1] * a[3]) * d;
        return b;
    }
};
typedef Matx33<float> Matx33f; typedef Matx33<double> Matx33d;
template <typename T> inline Matx33<T> operator*(const Matx33<T>& a, const Matx33<T>& b)
{   // Matx_MatMulOp: s accumulates in _Tp, k ascending
    Matx33<T> o;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T s = 0; for (int k = 0; k < 3; ++k) s += a.val[3 * i + k] * b.val[3 * k + j]; o.val[3 * i + j] = s; }
    return o;
}
template <typename T> inline Matx33<T> operator-(const Matx33<T>& a) { Matx33<T> o; for (int i = 0; i < 9; ++i) o.val[i] = -a.val[i]; return o; }
template <typename T> inline Vec<T, 3> operator*(const Matx33<T>& a, const Vec<T, 3>& v)
{
    Vec<T, 3> o; for (int i = 0; i < 3; ++i) { T s = 0; for (int k = 0; k < 3; ++k) s += a.val[3 * i + k] * v.val[k]; o.val[i] = s; } return o;
}
// a double matrix times a float vector (image_processor.cpp:256: the vector widens, the product is taken in double, the caller's
// Vec3f narrows it again)
inline Vec3d operator*(const Matx33d& a, const Vec3f& v) { return a * Vec3d(v); }

template <typename T> class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, CV_64F) { static_assert(sizeof(T) == 8, "Mat_<double> only"); }
    struct Comma { Mat_& m; int n; Comma& operator,(T v) { m.template at<T>(n / m.cols, n % m.cols) = v; ++n; return *this; } operator Mat() const { return m; } };
    Comma operator<<(T v) { this->template at<T>(0, 0) = v; return Comma{*this, 1}; }
};

template <typename T> using Ptr = std::shared_ptr<T>;
struct TermCriteria { enum { COUNT = 1, MAX_ITER = 1, EPS = 2 }; int type, maxCount; double epsilon; TermCriteria(int t, int c, double e) : type(t), maxCount(c), epsilon(e) {} };
struct NoArray {}; inline NoArray noArray() { return NoArray(); }
enum { OPTFLOW_USE_INITIAL_FLOW = 4, OPTFLOW_LK_GET_MIN_EIGENVALS = 8 };
enum { FM_7POINT = 1, FM_8POINT = 2, FM_LMEDS = 4, FM_RANSAC = 8, RANSAC = 8, LMEDS = 4 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3 };
enum { COLOR_GRAY2RGB = 8, COLOR_GRAY2BGR = 8 };

// ------------------------------------------------------------------ the image algorithms: the oracle's restatements behind OpenCV's names
struct CLAHE {
    double clip; Size tiles;
    void apply(const Mat& src, Mat& dst) const
    {
        LVCV_CHECK(src.type() == CV_8U && !src.empty(), "CLAHE: 8-bit image");
        Mat out(src.rows, src.cols, CV_8UC1);
        lvo_clahe_u8(src.data, src.cols, src.rows, (int)src.step, out.data, (int)out.step, clip, tiles.width, tiles.height);
        dst = out;
    }
};
inline Ptr<CLAHE> createCLAHE(double clip = 40.0, Size tiles = Size(8, 8)) { auto p = std::make_shared<CLAHE>(); p->clip = clip; p->tiles = tiles; return p; }

struct LvPyramidOwner { lvo_pyramid p; LvPyramidOwner() { std::memset(&p, 0, sizeof p); } ~LvPyramidOwner() { lvo_pyramid_free(&p); } };
inline const lvo_pyramid* lv_pyramid_of(const Mat& m) { LVCV_CHECK(m.aux != nullptr, "this Mat is not a level of a pyramid built by buildOpticalFlowPyramid"); return &((LvPyramidOwner*)m.aux.get())->p; }

// pyramid = {image 0, derivative 0, image 1, derivative 1, ...}; every image is a view of its level inside the padded buffer (so that
// ORBdescriptor's copyMakeBorder sees the LK border, image_processor.cpp:150), every derivative header only carries the owner
inline int buildOpticalFlowPyramid(const Mat& img, std::vector<Mat>& pyramid, Size win, int maxLevel, bool withDerivatives = true,
                                   int pyrBorder = BORDER_REFLECT_101, int derivBorder = BORDER_CONSTANT, bool tryReuse = true)
{
    LVCV_CHECK(img.type() == CV_8U && win.width == win.height && withDerivatives && pyrBorder == BORDER_REFLECT_101 && derivBorder == BORDER_CONSTANT, "buildOpticalFlowPyramid: the reference's arguments only");
    (void)tryReuse;
    auto own = std::make_shared<LvPyramidOwner>();
    lvo_pyramid_build(img.data, img.cols, img.rows, (int)img.step, win.width, maxLevel, &own->p);
    const lvo_pyramid& p = own->p;
    pyramid.clear();
    for (int l = 0; l < p.n_levels; ++l) {
        pyramid.push_back(Mat::view_in(p.img[l], p.w[l] + 2 * p.pad, p.h[l] + 2 * p.pad, (size_t)p.istride[l], p.pad, p.pad, p.w[l], p.h[l], own));
        Mat d; d.aux = own; d.rows = p.h[l]; d.cols = p.w[l]; pyramid.push_back(d);
    }
    return p.n_levels - 1;
}
inline void calcOpticalFlowPyrLK(const std::vector<Mat>& prevPyr, const std::vector<Mat>& nextPyr, const std::vector<Point2f>& prevPts, std::vector<Point2f>& nextPts,
                                 std::vector<uchar>& status, NoArray, Size win, int maxLevel, TermCriteria crit, int flags = 0, double minEig = 1e-4)
{
    LVCV_CHECK((flags & OPTFLOW_USE_INITIAL_FLOW) && nextPts.size() == prevPts.size(), "calcOpticalFlowPyrLK: OPTFLOW_USE_INITIAL_FLOW with an initial guess per point");
    LVCV_CHECK(crit.type == (TermCriteria::COUNT + TermCriteria::EPS) && minEig == 1e-4, "calcOpticalFlowPyrLK: the reference's criteria only");
    const lvo_pyramid* a = lv_pyramid_of(prevPyr.at(0)); const lvo_pyramid* b = lv_pyramid_of(nextPyr.at(0));
    LVCV_CHECK(a->pad == win.width && a->n_levels == b->n_levels && a->n_levels <= maxLevel + 1, "calcOpticalFlowPyrLK: pyramids built for this window / level count");
    status.assign(prevPts.size(), 0);
    if (prevPts.empty()) return;
    static_assert(sizeof(Point2f) == sizeof(lvo_pt2f), "");
    lvo_lk_track(a, b, (const lvo_pt2f*)prevPts.data(), (lvo_pt2f*)nextPts.data(), status.data(), (int)prevPts.size(), crit.maxCount, crit.epsilon, nullptr);
}
inline void goodFeaturesToTrack(const Mat& img, std::vector<Point2f>& corners, int maxCorners, double quality, double minDistance, const Mat& mask = Mat(),
                                int blockSize = 3, bool harris = false, double k = 0.04)
{
    LVCV_CHECK(blockSize == 3 && !harris, "goodFeaturesToTrack: the reference's arguments only"); (void)k;
    const lvo_pyramid* p = lv_pyramid_of(img);
    LVCV_CHECK(img.rows == p->h[0] && img.cols == p->w[0], "goodFeaturesToTrack: level 0 of a pyramid");
    const uint8_t* mk = nullptr;
    if (!mask.empty()) { LVCV_CHECK(mask.type() == CV_8U && mask.rows == img.rows && mask.cols == img.cols && mask.isContinuous(), "goodFeaturesToTrack: a continuous 8-bit mask of the image's size"); mk = mask.data; }
    const int cap = img.rows * img.cols;
    std::vector<lvo_pt2f> out((size_t)cap);
    const int n = lvo_good_features(p, mk, maxCorners, quality, minDistance, out.data(), cap);
    corners.resize((size_t)n);
    for (int i = 0; i < n; ++i) corners[(size_t)i] = Point2f(out[(size_t)i].x, out[(size_t)i].y);
}
inline Mat findFundamentalMat(const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, int method, double thresh, double conf, std::vector<uchar>& mask)
{
    LVCV_CHECK(method == FM_RANSAC && p1.size() == p2.size(), "findFundamentalMat: FM_RANSAC");
    std::vector<uchar> m(p1.size());
    if (lvo_find_fundamental_mask((const lvo_pt2f*)p1.data(), (const lvo_pt2f*)p2.data(), (int)p1.size(), thresh, conf, m.data())) mask = m;      // fewer than 7 points: OpenCV leaves the mask alone
    return Mat();
}
inline void lv_undistort(const std::vector<Point2f>& in, std::vector<Point2f>& out, const Matx33d& K, const Vec4d& dist, const Matx33d& R, const Matx33d& P, int model)
{
    for (int i = 0; i < 9; ++i) LVCV_CHECK(R.val[i] == (i % 4 == 0 ? 1.0 : 0.0), "undistortPoints: identity rectification (every call site)");
    const double intr[4] = {K(0, 0), K(1, 1), K(0, 2), K(1, 2)}, nintr[4] = {P(0, 0), P(1, 1), P(0, 2), P(1, 2)}, d[4] = {dist[0], dist[1], dist[2], dist[3]};
    out.resize(in.size());
    if (!in.empty()) lvo_undistort_points((const lvo_pt2f*)in.data(), (int)in.size(), intr, model, d, nintr, (lvo_pt2f*)out.data());
}
inline void undistortPoints(const std::vector<Point2f>& in, std::vector<Point2f>& out, const Matx33d& K, const Vec4d& dist, const Matx33d& R = Matx33d::eye(), const Matx33d& P = Matx33d::eye()) { lv_undistort(in, out, K, dist, R, P, 0); }
namespace fisheye { inline void undistortPoints(const std::vector<Point2f>& in, std::vector<Point2f>& out, const Matx33d& K, const Vec4d& dist, const Matx33d& R = Matx33d::eye(), const Matx33d& P = Matx33d::eye()) { lv_undistort(in, out, K, dist, R, P, 1); } }

// cv::Rodrigues, rotation vector -> matrix: computed in double whatever the argument types, narrowed on the way out [upstream calib3d]
inline void Rodrigues(const Vec3f& rv, Matx33f& Rout)
{
    double rx = rv[0], ry = rv[1], rz = rv[2];
    const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    double R[9];
    if (theta < DBL_EPSILON) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1. : 0.; }
    else {
        const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
        rx *= it; ry *= it; rz *= it;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int i = 0; i < 9; ++i) R[i] = c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i] + s * r_x[i];
    }
    for (int i = 0; i < 9; ++i) Rout.val[i] = (float)R[i];
}

// ------------------------------------------------------------------ the viewer's picture: nothing is drawn
inline void cvtColor(const Mat& src, Mat& dst, int) { if (dst.empty() || dst.rows != src.rows || dst.cols != src.cols) dst = Mat(src.rows, src.cols, CV_8UC3); }
inline void circle(Mat&, Point2f, int, const Scalar&, int = 1) {}
inline void line(Mat&, Point2f, Point2f, const Scalar&, int = 1) {}
inline void imshow(const std::string&, const Mat&) {}
inline int waitKey(int = 0) { return -1; }
inline void destroyAllWindows() {}

// ------------------------------------------------------------------ cv::FileStorage over the YAML dialect the reference ships (see ref_shim2/opencv2/lvref_cv_fs.hpp)
class FileNode {
public:
    bool present = false; std::string text; std::map<std::string, FileNode> kids; Mat mat;
    operator double() const { return present ? std::atof(text.c_str()) : 0.0; }
    operator float() const { return (float)(double)*this; }
    operator int() const { return present ? (int)std::lround(std::atof(text.c_str())) : 0; }
    operator std::string() const { return text; }
    FileNode operator[](const std::string& k) const { auto it = kids.find(k); return it == kids.end() ? FileNode() : it->second; }
    FileNode operator[](const char* k) const { return (*this)[std::string(k)]; }
    bool empty() const { return !present; }
};
inline void operator>>(const FileNode& n, Mat& m) { m = n.mat; }
inline void operator>>(const FileNode& n, std::string& s) { s = n.text; }
class FileStorage {
    FileNode root; bool ok = false;
    static std::string trim(const std::string& s) { size_t a = 0, b = s.size(); while (a < b && std::isspace((unsigned char)s[a])) ++a; while (b > a && std::isspace((unsigned char)s[b - 1])) --b; return s.substr(a, b - a); }
    static std::string unquote(const std::string& s) { if (s.size() >= 2 && (s[0] == '"' || s[0] == '\'') && s.back() == s[0]) return s.substr(1, s.size() - 2); return s; }
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string& path, int) { open(path, READ); }
    bool isOpened() const { return ok; }
    void release() {}
    bool open(const std::string& path, int)
    {
        std::ifstream f(path); if (!f) return false;
        std::vector<std::string> lines; std::string l;
        while (std::getline(f, l)) {      // a '#' outside quotes starts a comment (config/euroc.yaml:14: a quoted value followed by one that quotes again)
            char q = 0; for (size_t k = 0; k < l.size(); ++k) { if (q) { if (l[k] == q) q = 0; } else if (l[k] == '"' || l[k] == '\'') q = l[k]; else if (l[k] == '#') { l = l.substr(0, k); break; } }
            lines.push_back(l);
        }
        std::string parent;
        for (size_t i = 0; i < lines.size(); ++i) {
            const std::string& raw = lines[i];
            if (trim(raw).empty() || raw[0] == '%' || trim(raw) == "---") continue;
            const bool indented = std::isspace((unsigned char)raw[0]);
            const size_t colon = raw.find(':');
            if (colon == std::string::npos) continue;
            const std::string key = trim(raw.substr(0, colon)); std::string val = trim(raw.substr(colon + 1));
            if (!indented) {
                parent.clear();
                FileNode n; n.present = true;
                if (val.rfind("!!opencv-matrix", 0) == 0) { parent = key; val.clear(); }
                else if (val.empty()) parent = key;
                n.text = unquote(val);
                root.kids[key] = n;
            } else if (!parent.empty()) {
                FileNode& p = root.kids[parent];
                if (key == "data") {
                    std::string all = val;
                    while (all.find(']') == std::string::npos && i + 1 < lines.size()) all += " " + trim(lines[++i]);
                    for (char& ch : all) if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
                    std::istringstream ss(all); double v; std::vector<double> vals; while (ss >> v) vals.push_back(v);
                    const int r = (int)p.kids["rows"], c = (int)p.kids["cols"];
                    LVCV_CHECK((int)vals.size() == r * c, "opencv-matrix: data length");
                    p.mat = Mat(r, c, CV_64F);
                    for (int a = 0; a < r; ++a) for (int b = 0; b < c; ++b) p.mat.at<double>(a, b) = vals[(size_t)a * c + b];
                } else { FileNode n; n.present = true; n.text = unquote(val); p.kids[key] = n; }
            }
        }
        ok = true; return true;
    }
    FileNode operator[](const std::string& k) const { return root[k]; }
    FileNode operator[](const char* k) const { return root[std::string(k)]; }
};
}  // namespace cv
