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
#include "lvk_config.hpp"
#include <cstdio>
#include <cstring>
#include <string>
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
    // the reference's own constructor (image_processor.cpp:28-33): the configuration file is read by initialize()
    ImageProcessor(const std::string& config_file, lvk_context* ctx) : cfg_(), config_file_(config_file), ctx_(ctx), fe_(nullptr) {}
    ~ImageProcessor() { if (fe_) lvk_frontend_destroy(fe_); }
    bool initialize()                                                                   // image_processor.cpp:116-126
    {
        if (!config_file_.empty()) {                                                    // loadParameters (:44-113)
            ConfigFile f; std::string err;
            if (!f.open(config_file_)) { std::fprintf(stderr, "config_file error: %s\n", f.error().c_str()); return false; }
            if (!load_fe_config(f, &cfg_, &err)) { std::fprintf(stderr, "config_file error: %s\n", err.c_str()); return false; }
        }
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
        const lvk_image im = {msg.data, msg.width, msg.height, msg.step, /*is_device=*/0};     // a cv::Mat's own size and step
        if (lvk_frontend_process(fe_, &im, msg.timeStampToSec, imu, (int)imu_msg_buffer.size(),
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
    lvk_fe_config cfg_; std::string config_file_; lvk_context* ctx_; lvk_frontend* fe_;
    std::vector<MonoFeatureMeasurement> out_;
};

class LarVio {
public:
    LarVio(const lvk_ekf_config& cfg, lvk_context* ctx) : cfg_(cfg), ctx_(ctx), ekf_(nullptr), f_state_(nullptr), f_takeoff_(nullptr), takeoff_written_(false) {}
    // the reference's own constructor (larvio.cpp:40-44): initialize() reads the file and opens the two debug logs in output_dir
    LarVio(const std::string& config_file, lvk_context* ctx) : cfg_(), config_file_(config_file), ctx_(ctx), ekf_(nullptr), f_state_(nullptr), f_takeoff_(nullptr), takeoff_written_(false) {}
    ~LarVio() { if (ekf_) lvk_ekf_destroy(ekf_); if (f_state_) std::fclose(f_state_); if (f_takeoff_) std::fclose(f_takeoff_); }   // larvio.cpp:47-55
    bool initialize()                                                                   // larvio.cpp:314-360
    {
        if (!config_file_.empty()) {                                                    // loadParameters (:58-311)
            ConfigFile f; std::string err;
            if (!f.open(config_file_)) { std::fprintf(stderr, "config_file error: %s\n", f.error().c_str()); return false; }
            if (!load_ekf_config(f, &cfg_, &err)) { std::fprintf(stderr, "config_file error: %s\n", err.c_str()); return false; }
            const std::string dir = f.str("output_dir");                               // :220, :318-319; a directory that does not exist => no logs, as with ofstream
            if (!dir.empty()) { f_state_ = std::fopen((dir + "msckf_2_state.txt").c_str(), "w"); f_takeoff_ = std::fopen((dir + "msckf_2_takeoff.txt").c_str(), "w"); }
        }
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
        if (updated) write_logs();
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
        const int N = 9; double P[81]; if (lvk_ekf_get_cov_imu(ekf_, N, P) != LVK_OK) std::memset(P, 0, sizeof P);
        static const int idx[6] = {0, 1, 2, 6, 7, 8};
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) P66[6 * a + b] = P[(size_t)idx[a] * N + idx[b]];
    }
    void getPvel(double P33[9]) const
    {
        const int N = 9; double P[81]; if (lvk_ekf_get_cov_imu(ekf_, N, P) != LVK_OK) std::memset(P, 0, sizeof P);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) P33[3 * a + b] = P[(size_t)(3 + a) * N + 3 + b];
    }
    void getSwPoses(std::vector<lvk_clone>& out) const { out.resize(64); out.resize((size_t)lvk_ekf_get_clones(ekf_, out.data(), 64)); }     // :2703-2717
    void getActiveeMapPointPositions(std::vector<int64_t>& ids, std::vector<double>& xyz) const                                          // :2727-2735
    {
        ids.resize(1024); xyz.resize(3 * 1024); std::vector<double> idp(1024);
        const int n = lvk_ekf_get_features(ekf_, ids.data(), idp.data(), xyz.data(), 1024);
        ids.resize((size_t)n); xyz.resize((size_t)3 * n);
    }
    void getStableMapPointPositions(std::vector<int64_t>& ids, std::vector<double>& xyz)                                                 // :2717-2722
    {
        ids.resize(4096); xyz.resize(3 * 4096);
        const int n = lvk_ekf_take_lost_features(ekf_, ids.data(), xyz.data(), 4096);
        ids.resize((size_t)n); xyz.resize((size_t)3 * n);
    }
    lvk_ekf* handle() const { return ekf_; }
private:
    friend class VioPipeline;
    LarVio(const LarVio&); LarVio& operator=(const LarVio&);
    // the debug logs of larvio.cpp:388 and :446-453: take-off stamp once; then per update
    //   t-take_off  qw qx qy qz  vx vy vz  px py pz  bgx bgy bgz  bax bay baz  qbc(w x y z)  t_cam0_imu      ("%g" = the ostream default)
    void write_logs()
    {
        if (!f_state_ && !f_takeoff_) return;
        double s[30]; lvk_ekf_get_state(ekf_, s);
        const double t0 = lvk_ekf_take_off_stamp(ekf_);
        if (f_takeoff_ && !takeoff_written_) { std::fprintf(f_takeoff_, "%.9f\n", t0); std::fflush(f_takeoff_); takeoff_written_ = true; }
        if (!f_state_) return;
        double qbc[4]; rot_to_quat_wxyz(s + 17, qbc);
        std::fprintf(f_state_, "%g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g\n", s[0] - t0, s[4], s[1], s[2], s[3],
                     s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15], s[16], qbc[0], qbc[1], qbc[2], qbc[3], s[26], s[27], s[28]);
    }
    // Eigen::Quaterniond(Matrix3d) (larvio.cpp:436): Shepperd's branch on the trace, w >= 0 in the first branch only
    static void rot_to_quat_wxyz(const double* R, double q[4])
    {
        const double t = R[0] + R[4] + R[8];
        if (t > 0) {
            double r = std::sqrt(t + 1.0); q[0] = 0.5 * r; r = 0.5 / r;
            q[1] = (R[7] - R[5]) * r; q[2] = (R[2] - R[6]) * r; q[3] = (R[3] - R[1]) * r;
        } else {
            int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            double r = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
            q[1 + i] = 0.5 * r; r = 0.5 / r;
            q[0] = (R[3 * k + j] - R[3 * j + k]) * r; q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * r; q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * r;
        }
    }
    lvk_ekf_config cfg_; std::string config_file_; lvk_context* ctx_; lvk_ekf* ekf_;
    FILE* f_state_; FILE* f_takeoff_; bool takeoff_written_;
};

// The driver loop with the filter update of message k overlapping the front-end of the following frames (lvk_vio_pipe_*): same
// results as calling processImage / processFeatures in turn, but processFeatures' answer arrives through a callback on the filter's
// thread — the place to publish odometry.  The two halves must have been created on DIFFERENT Contexts (= HIP streams).
//     pipe.pushImu(...)            samples with t < t_img + 0.05, as larvioMain.cpp:98-102 fills imu_msg_buffer
//     pipe.processImage(img)       true = this frame produced a feature message (its update runs in the background)
//     pipe.drain()                 wait for the updates in flight before reading the estimator
class VioPipeline {
public:
    typedef void (*OdometryFn)(void* user, double ts, const LarVio& estimator);
    VioPipeline(ImageProcessor& fe, LarVio& be) : be_(be), pipe_(nullptr), fn_(nullptr), user_(nullptr)
    {
        if (lvk_vio_pipe_create(fe.handle(), be.handle(), &pipe_) != LVK_OK) pipe_ = nullptr;
        else lvk_vio_pipe_on_update(pipe_, &VioPipeline::trampoline, this);
    }
    ~VioPipeline() { if (pipe_) lvk_vio_pipe_destroy(pipe_); }
    bool ok() const { return pipe_ != nullptr; }
    void onOdometry(OdometryFn fn, void* user) { drain(); fn_ = fn; user_ = user; }
    bool pushImu(const ImuData* v, size_t n) { return pipe_ && lvk_vio_pipe_push_imu(pipe_, reinterpret_cast<const lvk_imu*>(v), (int)n) == LVK_OK; }
    bool processImage(const ImageData& msg)
    {
        int has = 0;
        const lvk_image im = {msg.data, msg.width, msg.height, msg.step, /*is_device=*/0};
        if (!pipe_ || lvk_vio_pipe_submit(pipe_, &im, msg.timeStampToSec, &has) != LVK_OK) return false;
        return has != 0;
    }
    bool drain(long* n_updates = nullptr, long* n_msgs = nullptr) { return pipe_ && lvk_vio_pipe_drain(pipe_, n_updates, n_msgs) == LVK_OK; }
    // erase counts submit() took early / how many of them the filter's thread found different (expected 0; lvk_c.h)
    bool earlyCounts(long* n_early, long* n_wrong) { return pipe_ && lvk_vio_pipe_early_counts(pipe_, n_early, n_wrong) == LVK_OK; }
    lvk_vio_pipe* handle() const { return pipe_; }
private:
    VioPipeline(const VioPipeline&); VioPipeline& operator=(const VioPipeline&);
    static void trampoline(void* self, double ts, const double*)
    {
        VioPipeline* p = static_cast<VioPipeline*>(self);
        p->be_.write_logs();                                      // the reference's state log, as processFeatures writes it
        if (p->fn_) p->fn_(p->user_, ts, p->be_);
    }
    LarVio& be_; lvk_vio_pipe* pipe_; OdometryFn fn_; void* user_;
};

}  // namespace lvk
#endif
