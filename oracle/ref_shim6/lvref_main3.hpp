// ORACLE / TEST INFRASTRUCTURE ONLY.  The driver-side stand-ins of ref_shim5/lvref_main.hpp for the build in which the reference's
// main() drives the reference's OWN classes (oracle/Makefile, _ref/larvio_ref_full): the same cv::imread / tick counters / text no-ops,
// on top of the cv stand-ins src/image_processor.cpp is compiled against (ref_shim3/), since ImageProcessor's interface carries cv::Mat.
#pragma once
#include <chrono>
#include <string>
#include "../ref_shim3/lvref_cv3.hpp"
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
