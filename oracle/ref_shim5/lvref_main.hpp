// ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  What the REFERENCE's own driver - /root/reference/app/larvioMain.cpp, compiled
// where it lies (oracle/Makefile, target `ref_main`) - needs besides the two classes it drives: cv::imread(path, 0) (the PNG reader of
// examples/lvk_png.hpp: the job cv::imread does in larvioMain.cpp:95), the tick counters, the two text functions of its overlay
// (no-ops: nothing looks at the overlay), and a headless pangolin / OpenGL (ref_shim5/pangolin/pangolin.h).  cv::Mat is the typed
// stand-in of oracle/ref_shim/lvref_cv.hpp, Eigen the eager one of oracle/ref_shim2 - the headers every src/*.cpp of the reference
// compiles against here, NOT the minimal stubs the adapter was written against (adapter/stubs/): the adapter's two translation units
// are compiled against these as well, which is their second, independent API check.
#pragma once
#include <chrono>
#include <string>
#include "../ref_shim/lvref_cv.hpp"
#include "../../examples/lvk_png.hpp"
typedef long long int64;
namespace cv {
inline Mat imread(const std::string& path, int /*flags: 0 = grey*/)
{
    lvk::GreyImage g; std::string err;
    if (!lvk::read_png_grey(path, &g, &err)) return Mat();
    Mat m(g.height, g.width, CV_8UC1);
    for (int y = 0; y < g.height; ++y) std::memcpy(m.data + (size_t)y * m.step, g.data.data() + (size_t)y * g.width, (size_t)g.width);
    return m;
}
inline int64 getTickCount() { return (int64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline double getTickFrequency() { return 1e9; }
inline Size getTextSize(const std::string& text, int, double scale, int thickness, int* baseLine) { if (baseLine) *baseLine = thickness; return Size((int)(text.size() * 10 * scale) + 1, (int)(20 * scale) + 1); }
inline void putText(Mat&, const std::string&, Point, int, double, Scalar, int = 1) {}
}
