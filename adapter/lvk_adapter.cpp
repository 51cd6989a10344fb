// adapter/lvk_adapter.cpp — larvio::ImageProcessor with the reference's exact signatures (adapter/larvio/image_processor.h) on top
// of the C ABI of liblvk_hip.so.  Replaces /root/reference/src/image_processor.cpp + src/ORBDescriptor.cpp in the `image_processor`
// library target (CMakeLists.txt:31-38).  cv::Mat / boost::shared_ptr / Eigen are whatever the tree provides (stubs in the compile
// check of this repo).
#include <larvio/image_processor.h>
#include "lvk_config.hpp"

#include <cstdio>
#include <cstring>

namespace larvio {

ImageProcessor::ImageProcessor(std::string& config_file_) : config_file(config_file_), cfg(), ctx(nullptr), fe(nullptr) {}

ImageProcessor::~ImageProcessor()
{
    if (fe) lvk_frontend_destroy(fe);
    if (ctx) lvk_context_destroy(ctx);
}

bool ImageProcessor::initialize()
{   // loadParameters (image_processor.cpp:44-113) + the object set-up of :116-126
    lvk::ConfigFile f; std::string err;
    if (!f.open(config_file)) { std::printf("config_file error: cannot open %s\n", config_file.c_str()); return false; }    // :39-42
    if (!lvk::load_fe_config(f, &cfg, &err)) { std::printf("config_file error: %s\n", err.c_str()); return false; }
    if (lvk_context_create(0, &ctx) != LVK_OK) { std::printf("ImageProcessor: no usable gfx950 device (there is no CPU fallback)\n"); return false; }
    if (lvk_frontend_create(ctx, &cfg, &fe) != LVK_OK) { std::printf("ImageProcessor: %s\n", lvk_last_error(ctx)); return false; }
    out.resize((size_t)cfg.max_features_num);
    return true;
}

bool ImageProcessor::processImage(const ImageDataPtr& msg, const std::vector<ImuData>& imu_msg_buffer, MonoCameraMeasurementPtr features)
{
    if (!fe || !msg || !features) return false;
    // ImuData = { double t; Vector3d w; Vector3d a } is laid out as lvk_imu (7 doubles); anything else is converted
    static_assert(sizeof(lvk_imu) == 7 * sizeof(double), "lvk_imu layout");
    std::vector<lvk_imu> conv;
    const lvk_imu* imu = nullptr;
    if (!imu_msg_buffer.empty()) {
        if (sizeof(ImuData) == sizeof(lvk_imu)) imu = reinterpret_cast<const lvk_imu*>(imu_msg_buffer.data());
        else {
            conv.resize(imu_msg_buffer.size());
            for (size_t i = 0; i < conv.size(); ++i) {
                conv[i].t = imu_msg_buffer[i].timeStampToSec;
                for (int k = 0; k < 3; ++k) { conv[i].gyro[k] = imu_msg_buffer[i].angular_velocity[k]; conv[i].acc[k] = imu_msg_buffer[i].linear_acceleration[k]; }
            }
            imu = conv.data();
        }
    }
    const cv::Mat& im = msg->image;
    if (im.empty() || im.channels() != 1) { std::printf("ImageProcessor::processImage: an 8-bit single-channel image is expected\n"); return false; }
    const lvk_image li = {im.data, im.cols, im.rows, (int)im.step, /*is_device=*/0};
    int n = 0, has = 0;
    ++frames_seen;
    if (lvk_frontend_process(fe, &li, msg->timeStampToSec, imu, (int)imu_msg_buffer.size(), out.data(), (int)out.size(), &n, &has) != LVK_OK) {
        std::printf("ImageProcessor::processImage: %s\n", lvk_last_error(ctx));
        return false;
    }
    if (!has) return false;
    features->timeStampToSec = msg->timeStampToSec;                   // getFeatureMsg (image_processor.cpp:1076-1128)
    features->features.resize((size_t)n);
    static_assert(sizeof(MonoFeatureMeasurement) == sizeof(lvk_feature_obs), "MonoFeatureMeasurement is the 72-byte wire record");
    if (n) std::memcpy(static_cast<void*>(features->features.data()), out.data(), sizeof(lvk_feature_obs) * (size_t)n);
    // publish() (:1131-1175) draws the frame on every publish; here the picture is built when somebody asks for it (getVisualImg):
    // a whole-image colour conversion and a read-back of the track table do not belong on the path to the filter
    // Once a driver has asked for a picture (getVisualImg), the pixels are SNAPSHOT here (one copy into a buffer that is reused): `im` may
    // wrap memory the caller frees or overwrites before getVisualImg is called (cv_bridge's toCvShare, a camera ring buffer); the reference
    // draws at publish time, from the live image.  A driver that never asks never pays the copy; the very first request is served from the
    // caller's own cv::Mat, and only if no later frame has been processed since (publishVisual).
    if (vis_wanted) {
        if (!vis_shared && vis_src.rows == im.rows && vis_src.cols == im.cols && vis_src.type() == im.type() && vis_src.data != im.data && !vis_src.empty()) {
            for (int y = 0; y < im.rows; ++y) std::memcpy(vis_src.ptr(y), im.ptr(y), (size_t)im.cols);
        } else vis_src = im.clone();
        vis_shared = false;
    } else { vis_src = im; vis_shared = true; }
    vis_frame = frames_seen; vis_pending = true;
    return true;
}

// publish() draws on a COLOR_GRAY2RGB copy of the current image (image_processor.cpp:1136-1168); without cv::circle in reach the
// tracked features are marked as 5x5 squares shaded by lifetime (blue = young, red = old, as the reference's colour ramp :1160-1161)
void ImageProcessor::publishVisual()
{
    vis_pending = false;
    if (vis_shared && frames_seen != vis_frame) { vis_src = cv::Mat(); vis_shared = false; }     // the caller's pixels of an older frame: not ours to read any more
    const cv::Mat& gray = vis_src;
    if (gray.empty()) return;
    cv::Mat rgb(gray.rows, gray.cols, CV_8UC3);
    for (int y = 0; y < gray.rows; ++y) {
        const unsigned char* s = gray.ptr(y); unsigned char* d = rgb.ptr(y);
        for (int x = 0; x < gray.cols; ++x) { d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x]; }
    }
    const int cap = cfg.max_features_num;
    std::vector<lvk_pt2f> pts((size_t)cap); std::vector<int> life((size_t)cap); int n = 0;
    // the track table is the front-end's CURRENT one: the published frame's as long as no later frame has been processed (the drivers ask
    // right after processImage / processFeatures: app/larvioMain.cpp:139-170, the nodelet's callback); otherwise the picture stays unmarked
    if (frames_seen == vis_frame && lvk_frontend_tracks(fe, nullptr, pts.data(), life.data(), nullptr, nullptr, cap, &n) == LVK_OK) {
        for (int i = 0; i < n; ++i) {
            const double len = life[i] >= 50 ? 1.0 : life[i] / 50.0;
            const unsigned char r = (unsigned char)(255 * (1 - len)), b = (unsigned char)(255 * len);
            const int cx = (int)(pts[i].x + 0.5f), cy = (int)(pts[i].y + 0.5f);
            for (int y = cy - 2; y <= cy + 2; ++y) for (int x = cx - 2; x <= cx + 2; ++x)
                if (y >= 0 && y < rgb.rows && x >= 0 && x < rgb.cols) { unsigned char* p = rgb.ptr(y) + 3 * x; p[0] = r; p[1] = 0; p[2] = b; }
        }
    }
    visual_img = rgb;
}

} // end namespace larvio
