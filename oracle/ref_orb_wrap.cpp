// ref_orb_wrap.cpp — C entry points over the REFERENCE's own larvio::ORBdescriptor, compiled from
// /root/reference/src/ORBDescriptor.cpp + /root/reference/include/ORB/ORBDescriptor.h where they lie (oracle/Makefile, _ref/liblvref_orb.so)
// against the OpenCV stand-ins of ref_shim/lvref_cv.hpp.  TEST INFRASTRUCTURE: tests/ use it to pin oracle/fe_track.c's ORB block
// (pattern, umax, IC angle, rotated BRIEF, Hamming) and the level-0 mosaic of oracle/fe_image.c against the reference's text.
#include <cstdint>
#include <new>
#include <vector>
#include <memory>
#include <iostream>
#include "lvref_cv.hpp"
// the planes, the pattern and the umax table are private members; this translation unit (the wrapper, not the reference's own .cpp,
// which is compiled untouched) reads them for the tests.  Access specifiers do not change the class layout.
#define private public
#include "ORB/ORBDescriptor.h"
#undef private

struct lvref_orb {
    std::vector<uchar> frame;       // the LK pyramid's level-0 buffer (image + reflect-101 frame) the image view is cut from
    cv::Mat parent, image;
    larvio::ORBdescriptor* d = nullptr;
    int w = 0, h = 0;
};

extern "C" {

// `padded`: (h + 2 pad) x (w + 2 pad) bytes, row stride `stride`: level 0 of cv::buildOpticalFlowPyramid (the image the reference's
// front-end hands to the constructor is a view into that buffer, image_processor.cpp:150,329-333).  pad = 0: an isolated image.
lvref_orb* lvref_orb_create(const uint8_t* padded, int w, int h, int pad, int stride, int nlevels)
{
    lvref_orb* o = new (std::nothrow) lvref_orb();
    if (!o) return nullptr;
    const int fw = w + 2 * pad, fh = h + 2 * pad;
    o->frame.resize((size_t)fw * (size_t)fh);
    for (int y = 0; y < fh; ++y) memcpy(&o->frame[(size_t)y * fw], padded + (size_t)y * stride, (size_t)fw);
    o->parent = cv::Mat(fh, fw, CV_8U, o->frame.data(), (size_t)fw);
    o->image = o->parent(cv::Rect(pad, pad, w, h));
    o->w = w; o->h = h;
    o->d = new larvio::ORBdescriptor(o->image, 2, nlevels);          // image_processor.cpp:150: scale factor 2, pyramid_levels levels
    return o;
}
void lvref_orb_destroy(lvref_orb* o) { if (o) { delete o->d; delete o; } }

// descriptors (n x 32 bytes) and IC angles of n level-0 points (computeDescriptors with levels all 0, image_processor.cpp:442-445)
int lvref_orb_describe(lvref_orb* o, const float* xy, int n, uint8_t* desc, float* angle)
{
    std::vector<cv::Point2f> pts; std::vector<int> levels((size_t)n, 0);
    for (int i = 0; i < n; ++i) pts.push_back(cv::Point2f(xy[2 * i], xy[2 * i + 1]));
    cv::Mat out;
    if (!o->d->computeDescriptors(pts, levels, out)) return -1;
    for (int i = 0; i < n; ++i) { memcpy(desc + (size_t)32 * i, out.ptr<uchar>(i), 32); if (angle) angle[i] = o->d->IC_Angle(0, pts[(size_t)i]); }
    return 0;
}

int lvref_orb_hamming(const uint8_t* a, const uint8_t* b)
{
    uint32_t wa[8], wb[8]; memcpy(wa, a, 32); memcpy(wb, b, 32);     // the reference reads int32 words
    cv::Mat ma(1, 32, CV_8U, wa, 32), mb(1, 32, CV_8U, wb, 32);
    return larvio::ORBdescriptor::computeDescriptorDistance(ma, mb);
}

// level 0 of the mosaic with its border (h + 64) x (w + 64), plain and blurred; border = max(edgeThreshold 31, ...) + 1 = 32
int lvref_orb_planes(lvref_orb* o, uint8_t* ext, uint8_t* blur)
{
    const larvio::ORBdescriptor* L = o->d;
    const cv::Rect li = L->mvLayerInfo[0];
    const int B = li.x;                                                // level 0 sits at (border, border)
    if (li.y != B || li.width != o->w || li.height != o->h) return -1;
    const int ew = o->w + 2 * B, eh = o->h + 2 * B;
    for (int y = 0; y < eh; ++y) {
        memcpy(ext + (size_t)y * ew, L->mImagePyramid.ptr<uchar>(y), (size_t)ew);
        memcpy(blur + (size_t)y * ew, L->mBluredImagePyramid.ptr<uchar>(y), (size_t)ew);
    }
    return B;
}

int lvref_orb_umax(lvref_orb* o, int* out16)
{
    const larvio::ORBdescriptor* L = o->d;
    for (size_t i = 0; i < L->umax.size() && i < 16; ++i) out16[i] = L->umax[i];
    return (int)L->umax.size();
}
int lvref_orb_pattern(lvref_orb* o, int* out1024)
{
    const larvio::ORBdescriptor* L = o->d;
    for (size_t i = 0; i < L->pattern.size() && i < 512; ++i) { out1024[2 * i] = L->pattern[i].x; out1024[2 * i + 1] = L->pattern[i].y; }
    return (int)L->pattern.size();
}

}
