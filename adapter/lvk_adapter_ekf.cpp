// adapter/lvk_adapter_ekf.cpp — larvio::LarVio with the reference's exact signatures (adapter/larvio/larvio.h) on top of the C ABI
// of liblvk_hip.so.  Replaces /root/reference/src/larvio.cpp (+ the initializer sources it drags in) in the `estimator` library
// target (CMakeLists.txt:68-83).
#include <larvio/larvio.h>
#include "lvk_config.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace larvio {

// processFeatures is DEFERRED by default: it returns as soon as its return value and the IMU samples to erase are known (both are
// final before any arithmetic: lvk_ekf_process_async) and the update runs on the filter's worker thread and stream; every getter,
// the next processFeatures and the destructor wait for it first, so callers see exactly the values the blocking call gives.  With the
// reference's drivers this lets processImage of frame k+1 run while the update of frame k is still in flight whenever the driver
// does not ask for the pose in between (the ROS nodelet with no odometry subscriber, a logger that reads the pose one frame late);
// a driver that calls getTbw() right after processFeatures (app/larvioMain.cpp:139) simply waits there instead.
// LVK_ADAPTER_BLOCKING=1 restores the blocking call.
LarVio::LarVio(std::string& config_file_) : config_file(config_file_), cfg(), ctx(nullptr), ekf(nullptr), f_state(nullptr), f_takeoff(nullptr), takeoff_written(false),
                                            failed(false), good_state(), blocking(false), pending(false)
{
    good_state[4] = 1.0;                                                              // identity quaternion (x y z w at [1..4]) until the first update
    good_state[17] = good_state[21] = good_state[25] = 1.0;
    const char* b = std::getenv("LVK_ADAPTER_BLOCKING");
    blocking = b && std::atoi(b) != 0;
}

LarVio::~LarVio()
{   // larvio.cpp:47-55
    finish();
    if (ekf) lvk_ekf_destroy(ekf);
    if (ctx) lvk_context_destroy(ctx);
    if (f_state) std::fclose(f_state);
    if (f_takeoff) std::fclose(f_takeoff);
}

bool LarVio::initialize()
{   // loadParameters (larvio.cpp:58-311) + :314-360
    lvk::ConfigFile f; std::string err;
    if (!f.open(config_file)) { std::printf("config_file error: cannot open %s\n", config_file.c_str()); return false; }
    if (!lvk::load_ekf_config(f, &cfg, &err)) { std::printf("config_file error: %s\n", err.c_str()); return false; }
    const std::string dir = f.str("output_dir");                      // :220, :318-319 (a directory that does not exist: no logs, as with ofstream)
    if (!dir.empty() && !f_state) { f_state = std::fopen((dir + "msckf_2_state.txt").c_str(), "w"); f_takeoff = std::fopen((dir + "msckf_2_takeoff.txt").c_str(), "w"); }
    if (!ctx && lvk_context_create(0, &ctx) != LVK_OK) { std::printf("LarVio: no usable gfx950 device (there is no CPU fallback)\n"); return false; }
    if (lvk_ekf_create(ctx, &cfg, &ekf) != LVK_OK) { std::printf("LarVio: %s\n", lvk_last_error(ctx)); return false; }
    return true;
}

void LarVio::reset()
{
    finish();
    if (ekf) { lvk_ekf_destroy(ekf); ekf = nullptr; }
    active_slam_features.clear(); takeoff_written = false; failed = false;
    initialize();
}

bool LarVio::processFeatures(MonoCameraMeasurementPtr msg, std::vector<ImuData>& imu_msg_buffer)
{
    if (!ekf || !msg || failed) return false;                                        // a failed filter stays failed (the library says so too): the driver sees `false` from now on
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
    static_assert(sizeof(MonoFeatureMeasurement) == sizeof(lvk_feature_obs), "MonoFeatureMeasurement is the 72-byte wire record");
    const lvk_feature_obs* feats = msg->features.empty() ? nullptr : reinterpret_cast<const lvk_feature_obs*>(msg->features.data());
    finish();                                                                         // the previous update's logs and map points, before this one changes the state
    int used = 0, updated = 0;
    const lvk_status st = blocking ? lvk_ekf_process(ekf, msg->timeStampToSec, feats, (int)msg->features.size(), imu, (int)imu_msg_buffer.size(), &used, &updated)
                                   : lvk_ekf_process_async(ekf, msg->timeStampToSec, feats, (int)msg->features.size(), imu, (int)imu_msg_buffer.size(), &used, &updated);
    if (st != LVK_OK) {
        std::printf("LarVio::processFeatures: %s\n", lvk_last_error(ctx));
        return false;
    }
    imu_msg_buffer.erase(imu_msg_buffer.begin(), imu_msg_buffer.begin() + used);      // larvio.cpp:511-512, StaticInitializer.cpp:146-147
    if (!updated) return false;
    pending = true;                                                                   // logs + active_slam_features follow the update: finish()
    if (blocking) finish();
    return true;
}

// what the reference does at the end of processFeatures (:388, :446-458), once the (possibly deferred) update is done
void LarVio::finish()
{
    if (!pending || !ekf) return;
    pending = false;
    int upd = 0;
    if (lvk_ekf_wait(ekf, &upd) != LVK_OK) {
        // the frame's processFeatures already answered `true`; what can still be done: nothing of the failed update is published or
        // logged, the getters keep answering from the last good state, and every later processFeatures returns false
        std::printf("LarVio::processFeatures (deferred): %s\n", lvk_last_error(ctx)); failed = true; return;
    }
    lvk_ekf_get_state(ekf, good_state);
    writeLogs();                                                                      // :388, :446-453
    // active_slam_features (:455-458): the in-state features after this update
    std::vector<int64_t> ids(4096); std::vector<double> idp(4096), pos(3 * 4096);
    const int n = lvk_ekf_get_features(ekf, ids.data(), idp.data(), pos.data(), 4096);
    for (int i = 0; i < n; ++i) active_slam_features[(FeatureIDType)ids[(size_t)i]] = Eigen::Vector3d(pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]);
}

static Eigen::Matrix3d quat_to_rot(const double* q /* x y z w */)
{   // Quaterniond(w, x, y, z).toRotationMatrix()
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    Eigen::Matrix3d R;
    R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - w * z); R(0, 2) = 2 * (x * z + w * y);
    R(1, 0) = 2 * (x * y + w * z); R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - w * x);
    R(2, 0) = 2 * (x * z - w * y); R(2, 1) = 2 * (y * z + w * x); R(2, 2) = 1 - 2 * (x * x + y * y);
    return R;
}

// IMUState::T_imu_body is the identity in the reference (larvio.cpp:35) and nothing sets it: T_b_w = T_i_w, H_pose = H_vel = I.
Eigen::Isometry3d LarVio::getTbw()
{
    finish();
    double s[30]; if (failed || lvk_ekf_get_state(ekf, s) != LVK_OK) std::memcpy(s, good_state, sizeof s);
    Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
    T.linear() = quat_to_rot(s + 1);
    T.translation() = Eigen::Vector3d(s[8], s[9], s[10]);
    return T;
}

Eigen::Vector3d LarVio::getVel()
{
    finish();
    double s[30]; if (failed || lvk_ekf_get_state(ekf, s) != LVK_OK) std::memcpy(s, good_state, sizeof s);
    return Eigen::Vector3d(s[5], s[6], s[7]);
}

Eigen::Matrix<double, 6, 6> LarVio::getPpose()
{
    finish();   // P_imu_pose << P_pp, P_po, P_op, P_oo  (position block first), larvio.cpp:2673-2679
    const int N = 9;
    double P[81]; if (lvk_ekf_get_cov_imu(ekf, N, P) != LVK_OK) std::memset(P, 0, sizeof P);
    static const int idx[6] = {6, 7, 8, 0, 1, 2};
    Eigen::Matrix<double, 6, 6> out;
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out(a, b) = P[(size_t)idx[a] * N + idx[b]];
    return out;
}

Eigen::Matrix3d LarVio::getPvel()
{
    finish();
    const int N = 9;
    double P[81]; if (lvk_ekf_get_cov_imu(ekf, N, P) != LVK_OK) std::memset(P, 0, sizeof P);
    Eigen::Matrix3d out;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out(a, b) = P[(size_t)(3 + a) * N + 3 + b];
    return out;
}

void LarVio::getSwPoses(std::vector<Eigen::Isometry3d>& swPoses)
{
    finish();
    swPoses.clear();
    std::vector<lvk_clone> c(128);
    const int n = lvk_ekf_get_clones(ekf, c.data(), 128);
    for (int i = 0; i < n; ++i) {
        Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
        T.linear() = quat_to_rot(c[(size_t)i].q);
        T.translation() = Eigen::Vector3d(c[(size_t)i].p[0], c[(size_t)i].p[1], c[(size_t)i].p[2]);
        swPoses.push_back(T);
    }
}

void LarVio::getStableMapPointPositions(std::map<larvio::FeatureIDType, Eigen::Vector3d>& mMapPoints)
{
    finish();
    std::vector<int64_t> ids(4096); std::vector<double> pos(3 * 4096);
    for (;;) {                                                       // the library hands them out in chunks and forgets them, as the reference clears its map
        const int n = lvk_ekf_take_lost_features(ekf, ids.data(), pos.data(), 4096);
        for (int i = 0; i < n; ++i) mMapPoints[(FeatureIDType)ids[(size_t)i]] = Eigen::Vector3d(pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]);
        if (n < 4096) break;
    }
}

void LarVio::getActiveeMapPointPositions(std::map<larvio::FeatureIDType, Eigen::Vector3d>& mMapPoints)
{
    finish();
    for (const auto& item : active_slam_features) mMapPoints[item.first] = item.second;
    active_slam_features.clear();
}

// Eigen::Quaterniond(Matrix3d) (larvio.cpp:436): Shepperd's branches, w x y z
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

void LarVio::writeLogs()
{   // take-off stamp once (larvio.cpp:388); then t-take_off, q(w x y z), v, p, bg, ba, q_bc(w x y z), t_cam0_imu per update (:446-453)
    if (!f_state && !f_takeoff) return;
    double s[30]; lvk_ekf_get_state(ekf, s);
    const double t0 = lvk_ekf_take_off_stamp(ekf);
    if (f_takeoff && !takeoff_written) { std::fprintf(f_takeoff, "%.9f\n", t0); std::fflush(f_takeoff); takeoff_written = true; }
    if (!f_state) return;
    double qbc[4]; rot_to_quat_wxyz(s + 17, qbc);
    std::fprintf(f_state, "%g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g %g\n", s[0] - t0, s[4], s[1], s[2], s[3],
                 s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15], s[16], qbc[0], qbc[1], qbc[2], qbc[3], s[26], s[27], s[28]);
}

} // namespace larvio
