// lvk_larvio.hpp — the C++ host side above the C ABI (include/lvk_c.h): the two classes LARVIO's drivers construct, with the
// reference's method names, argument meaning and error convention (bool returns; "false" from processImage = no message this
// frame), minus OpenCV / Eigen / Boost in the signatures (those libraries are not installed here; INTEGRATION.md shows the
// adapter with the reference's exact signatures for a tree that has them).  Header-only; link with liblvk_hip.so.
//
//   reference                                                   here
//   larvio::ImageProcessor (image_processor.h:36-68)            lvk::ImageProcessor
//   larvio::LarVio (larvio.h:39-90)                             lvk::LarVio
//   ImuData (sensors/ImuData.hpp:17-43)                         lvk::ImuData      (same three fields)
//   ImgData{timeStampToSec, cv::Mat} (sensors/ImageData.hpp)    lvk::ImageData    (plain 8-bit view instead of cv::Mat)
//   MonoFeatureMeasurement / MonoCameraMeasurement              lvk::MonoFeatureMeasurement (= lvk_feature_obs, the same 72-byte record) /
//   (feature_msg.h:15-56)                                       lvk::MonoCameraMeasurement
#ifndef LVK_LARVIO_HPP
#define LVK_LARVIO_HPP
#include "lvk_c.h"
#include <cstdio>
#include <cstring>
#include <vector>

namespace lvk {

struct ImuData { double timeStampToSec; double angular_velocity[3]; double linear_acceleration[3]; };
static_assert(sizeof(ImuData) == sizeof(lvk_imu), "ImuData is laid out as lvk_imu");
struct ImageData { double timeStampToSec; const uint8_t* data; int width, height, step; };
typedef lvk_feature_obs MonoFeatureMeasurement;
struct MonoCameraMeasurement { double timeStampToSec; std::vector<MonoFeatureMeasurement> features; };

// one context (= one HIP stream) shared by the two halves unless the caller passes its own
class Context {
public:
    explicit Context(int device = 0) : ctx_(nullptr) { status_ = lvk_context_create(device, &ctx_); }
    ~Context() { if (ctx_) lvk_context_destroy(ctx_); }
    bool ok() const { return status_ == LVK_OK; }
    lvk_context* get() const { return ctx_; }
    const char* error() const { return ctx_ ? lvk_last_error(ctx_) : "no usable gfx950 device (liblvk_hip has no CPU fallback)"; }
private:
    Context(const Context&); Context& operator=(const Context&);
    lvk_context* ctx_; lvk_status status_;
};

class ImageProcessor {
public:
    // the reference's constructor takes the YAML path and loadParameters() reads it (image_processor.cpp:44-113); here the caller
    // hands over the same parameters as a plain struct (field names = YAML keys)
    ImageProcessor(const lvk_fe_config& cfg, lvk_context* ctx) : cfg_(cfg), ctx_(ctx), fe_(nullptr) {}
    ~ImageProcessor() { if (fe_) lvk_frontend_destroy(fe_); }
    bool initialize()                                                                   // image_processor.cpp:116-126
    {
        if (!ctx_) { std::fprintf(stderr, "ImageProcessor: no device context\n"); return false; }
        if (lvk_frontend_create(ctx_, &cfg_, &fe_) != LVK_OK) { std::fprintf(stderr, "ImageProcessor: %s\n", lvk_last_error(ctx_)); return false; }
        out_.resize((size_t)cfg_.max_features_num);
        return true;
    }
    // image_processor.cpp:130-219.  true = `features` holds this frame's message.
    bool processImage(const ImageData& msg, const std::vector<ImuData>& imu_msg_buffer, MonoCameraMeasurement* features)
    {
        if (!fe_ || !features) return false;
        int n = 0, has = 0;
        const lvk_imu* imu = imu_msg_buffer.empty() ? nullptr : reinterpret_cast<const lvk_imu*>(imu_msg_buffer.data());
        if (lvk_frontend_process(fe_, msg.data, msg.step, /*img_is_device=*/0, msg.timeStampToSec, imu, (int)imu_msg_buffer.size(),
                                 out_.data(), (int)out_.size(), &n, &has) != LVK_OK) {
            std::fprintf(stderr, "ImageProcessor::processImage: %s\n", lvk_last_error(ctx_));
            return false;
        }
        if (!has) return false;
        features->timeStampToSec = msg.timeStampToSec;
        features->features.assign(out_.begin(), out_.begin() + n);
        return true;
    }
    lvk_frontend* handle() const { return fe_; }
private:
    ImageProcessor(const ImageProcessor&); ImageProcessor& operator=(const ImageProcessor&);
    lvk_fe_config cfg_; lvk_context* ctx_; lvk_frontend* fe_;
    std::vector<MonoFeatureMeasurement> out_;
};

class LarVio {
public:
    LarVio(const lvk_ekf_config& cfg, lvk_context* ctx) : cfg_(cfg), ctx_(ctx), ekf_(nullptr) {}
    ~LarVio() { if (ekf_) lvk_ekf_destroy(ekf_); }
    bool initialize()                                                                   // larvio.cpp:314-360
    {
        if (!ctx_) { std::fprintf(stderr, "LarVio: no device context\n"); return false; }
        if (lvk_ekf_create(ctx_, &cfg_, &ekf_) != LVK_OK) { std::fprintf(stderr, "LarVio: %s\n", lvk_last_error(ctx_)); return false; }
        return true;
    }
    // larvio.cpp:363-461.  Erases the IMU samples it consumed from the caller's vector (:511-512).  true = the filter was updated.
    bool processFeatures(MonoCameraMeasurement* msg, std::vector<ImuData>& imu_msg_buffer)
    {
        if (!ekf_ || !msg) return false;
        int used = 0, updated = 0;
        const lvk_imu* imu = imu_msg_buffer.empty() ? nullptr : reinterpret_cast<const lvk_imu*>(imu_msg_buffer.data());
        if (lvk_ekf_process(ekf_, msg->timeStampToSec, msg->features.empty() ? nullptr : msg->features.data(), (int)msg->features.size(),
                            imu, (int)imu_msg_buffer.size(), &used, &updated) != LVK_OK) {
            std::fprintf(stderr, "LarVio::processFeatures: %s\n", lvk_last_error(ctx_));
            return false;
        }
        imu_msg_buffer.erase(imu_msg_buffer.begin(), imu_msg_buffer.begin() + used);
        return updated != 0;
    }
    // getTbw (larvio.cpp:2644-2655): body-to-world pose as a row-major 4x4
    void getTbw(double T[16]) const
    {
        double s[30]; lvk_ekf_get_state(ekf_, s);
        const double x = s[1], y = s[2], z = s[3], w = s[4];                            // stored [x y z w]
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                             2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                             2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = s[8 + i]; }
        T[12] = T[13] = T[14] = 0; T[15] = 1;
    }
    void getVel(double v[3]) const { double s[30]; lvk_ekf_get_state(ekf_, s); std::memcpy(v, s + 5, 24); }        // :2658-2660
    // getPpose / getPvel (:2663-2700): covariance blocks of (theta, p) and v
    void getPpose(double P66[36]) const
    {
        const int N = lvk_ekf_dim(ekf_); std::vector<double> P((size_t)N * N); lvk_ekf_get_cov(ekf_, P.data());
        static const int idx[6] = {0, 1, 2, 6, 7, 8};
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) P66[6 * a + b] = P[(size_t)idx[a] * N + idx[b]];
    }
    void getPvel(double P33[9]) const
    {
        const int N = lvk_ekf_dim(ekf_); std::vector<double> P((size_t)N * N); lvk_ekf_get_cov(ekf_, P.data());
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) P33[3 * a + b] = P[(size_t)(3 + a) * N + 3 + b];
    }
    void getSwPoses(std::vector<lvk_clone>& out) const { out.resize(64); out.resize((size_t)lvk_ekf_get_clones(ekf_, out.data(), 64)); }     // :2703-2717
    void getActiveeMapPointPositions(std::vector<int64_t>& ids, std::vector<double>& xyz) const                                          // :2727-2735
    {
        ids.resize(1024); xyz.resize(3 * 1024); std::vector<double> idp(1024);
        const int n = lvk_ekf_get_features(ekf_, ids.data(), idp.data(), xyz.data(), 1024);
        ids.resize((size_t)n); xyz.resize((size_t)3 * n);
    }
    lvk_ekf* handle() const { return ekf_; }
private:
    LarVio(const LarVio&); LarVio& operator=(const LarVio&);
    lvk_ekf_config cfg_; lvk_context* ctx_; lvk_ekf* ekf_;
};

}  // namespace lvk
#endif
