// ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  C entry points around the REFERENCE's own front-end class - /root/reference/src/
// image_processor.cpp (ImageProcessor::processImage and everything under it) and src/ORBDescriptor.cpp, compiled where they lie
// (oracle/Makefile, target `ref` -> oracle/_ref/liblvref_imgproc.so; never copied) against the stand-ins of oracle/ref_shim3/.
// Behind OpenCV's image-algorithm names stand the ORACLE's restatements of them (lvref_cv3.hpp says which), so what this library pins
// is the reference's orchestration around those calls - the state machine, the gates in their order, the bookkeeping, the message -
// not the algorithms themselves.  Used by oracle/lvref.py (RefImageProcessor) to hold the oracle's front-end object (fe_pipeline.c),
// frame by frame, to ImageProcessor::processImage itself.
#include <string>
#include <vector>
#include <map>
#include <set>
#include <sstream>
#include <iostream>
#include <cstring>
#include "lvref_cv3.hpp"
#include <boost/shared_ptr.hpp>
#define private public
#define protected public
#include <larvio/image_processor.h>
#undef private
#undef protected

using namespace larvio;

struct RefFe { ImageProcessor* ip = nullptr; MonoCameraMeasurement msg; };

extern "C" {

void* lvref_imgproc_create(const char* yaml_path)
{
    std::string p(yaml_path);
    RefFe* r = new RefFe(); r->ip = new ImageProcessor(p);
    if (!r->ip->initialize()) { delete r->ip; delete r; return nullptr; }
    return r;
}
// what ImageProcessor::loadParameters (image_processor.cpp:44-113) made of the configuration file, in lvk_fe_config's terms:
// 0 width 1 height 2 pyramid_levels 3 patch_size 4 max_iteration 5 track_precision 6 max_features_num 7 min_distance 8 flag_equalize
// 9 pub_frequency 10 distortion_model (0 radtan, 1 equidistant, -1 other) 11-14 intrinsics 15-18 distortion 19-27 R_cam_imu (row-major)
// 28 ransac_threshold 29 img_rate
void lvref_imgproc_params(void* h, double* o)
{
    ImageProcessor& I = *((RefFe*)h)->ip;
    o[0] = I.cam_resolution[0]; o[1] = I.cam_resolution[1]; o[2] = I.processor_config.pyramid_levels; o[3] = I.processor_config.patch_size;
    o[4] = I.processor_config.max_iteration; o[5] = I.processor_config.track_precision; o[6] = I.processor_config.max_features_num;
    o[7] = I.processor_config.min_distance; o[8] = I.processor_config.flag_equalize ? 1 : 0; o[9] = I.processor_config.pub_frequency;
    o[10] = I.cam_distortion_model == "radtan" ? 0 : I.cam_distortion_model == "equidistant" ? 1 : -1;
    for (int k = 0; k < 4; ++k) { o[11 + k] = I.cam_intrinsics[k]; o[15 + k] = I.cam_distortion_coeffs[k]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[19 + 3 * i + j] = I.R_cam_imu(i, j);
    o[28] = I.processor_config.ransac_threshold; o[29] = I.processor_config.img_rate;
}
void lvref_imgproc_destroy(void* h) { RefFe* r = (RefFe*)h; if (r) { delete r->ip; delete r; } }

// one ImageProcessor::processImage call.  img: h x w bytes (stride bytes per row); imu: m x 7 doubles (t, gyro, acc) = the driver's
// whole buffer as it stands (the front-end only reads it).  feats_out: up to cap x 9 doubles (id, u, v, u_init, v_init, u_vel, v_vel,
// u_init_vel, v_init_vel).  Returns processImage's own answer; *n_out = features in the message.
int lvref_imgproc_process(void* h, double stamp, const unsigned char* img, int w, int hgt, int stride, int m, const double* imu, double* feats_out, int cap, int* n_out)
{
    RefFe* r = (RefFe*)h;
    boost::shared_ptr<ImgData> d(new ImgData());
    d->timeStampToSec = stamp; d->image = cv::Mat(hgt, w, CV_8UC1);
    for (int y = 0; y < hgt; ++y) std::memcpy(d->image.data + (size_t)y * d->image.step, img + (size_t)y * stride, (size_t)w);
    std::vector<ImuData> buf; buf.reserve((size_t)m);
    for (int i = 0; i < m; ++i) buf.push_back(ImuData(imu[7 * i], imu[7 * i + 1], imu[7 * i + 2], imu[7 * i + 3], imu[7 * i + 4], imu[7 * i + 5], imu[7 * i + 6]));
    r->msg.features.clear();
    std::streambuf* keep = std::cout.rdbuf(); std::ostringstream sink; std::cout.rdbuf(sink.rdbuf());
    const bool have = r->ip->processImage(d, buf, &r->msg);
    std::cout.rdbuf(keep);
    const int n = have ? (int)r->msg.features.size() : 0;
    for (int i = 0; i < n && i < cap; ++i) {
        const MonoFeatureMeasurement& f = r->msg.features[(size_t)i]; double* o = feats_out + 9 * i;
        o[0] = (double)f.id; o[1] = f.u; o[2] = f.v; o[3] = f.u_init; o[4] = f.v_init; o[5] = f.u_vel; o[6] = f.v_vel; o[7] = f.u_init_vel; o[8] = f.v_init_vel;
    }
    if (n_out) *n_out = n;
    return have ? 1 : 0;
}
int lvref_imgproc_state(void* h) { return (int)((RefFe*)h)->ip->image_state; }
// the tracks as the next frame will find them (prev_pts_ after the swap at the end of processImage) with ids, lifetimes, init points, descriptors
int lvref_imgproc_tracks(void* h, unsigned long long* ids, float* pts, int* life, float* init, unsigned char* desc, int cap)
{
    ImageProcessor& p = *((RefFe*)h)->ip;
    const int n = (int)p.prev_pts_.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (ids) ids[i] = i < (int)p.pts_ids_.size() ? p.pts_ids_[(size_t)i] : 0;
        if (pts) { pts[2 * i] = p.prev_pts_[(size_t)i].x; pts[2 * i + 1] = p.prev_pts_[(size_t)i].y; }
        if (life) life[i] = i < (int)p.pts_lifetime_.size() ? p.pts_lifetime_[(size_t)i] : 0;
        if (init && i < (int)p.init_pts_.size()) { init[2 * i] = p.init_pts_[(size_t)i].x; init[2 * i + 1] = p.init_pts_[(size_t)i].y; }
        if (desc && i < (int)p.vOrbDescriptors.size()) std::memcpy(desc + 32 * i, p.vOrbDescriptors[(size_t)i].data, 32);
    }
    return n;
}
int lvref_imgproc_new_pts(void* h, float* pts, int cap)
{
    ImageProcessor& p = *((RefFe*)h)->ip;
    const int n = (int)p.new_pts_.size();
    for (int i = 0; i < n && i < cap; ++i) { pts[2 * i] = p.new_pts_[(size_t)i].x; pts[2 * i + 1] = p.new_pts_[(size_t)i].y; }
    return n;
}
int lvref_imgproc_counts(void* h, int* out3) { ImageProcessor& p = *((RefFe*)h)->ip; out3[0] = p.before_tracking; out3[1] = p.after_tracking; out3[2] = p.after_ransac; return (int)p.pts_ids_.size(); }

}  // extern "C"
