// ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  C entry points around the REFERENCE's own filter - /root/reference/src/larvio.cpp,
// src/FlexibleInitializer.cpp, src/StaticInitializer.cpp and src/feature_manager.cpp are compiled where they lie (oracle/Makefile, target
// `ref` -> oracle/_ref/liblvref_larvio.so; never copied) against the stand-in headers of oracle/ref_shim2/ (Eigen, OpenCV's FileStorage,
// boost's chi-squared quantile and shared_ptr, the Ceres names initial_sfm.h mentions).  What is NOT in THIS library: the moving-start
// initialiser's body (DynamicInitializer.cpp with solve_5pts.cpp / initial_sfm.cpp is compiled against the other stand-ins of
// oracle/ref_shim4/ - into liblvref_dyninit.so, and together with these sources into the whole program _ref/larvio_ref_full) - its two
// entry points are defined below as "never succeeds", so a stream must start at rest (the reference's StaticInitializer fires) or from
// a state handed in through lvref_larvio_set_state (+ lvref_larvio_set_last_zupt_time for the start as an initialiser leaves it).  Used by oracle/lvref.py (RefLarVio) to hold the oracle's filter, update by update,
// to LarVio::processFeatures itself.
#include <string>
#include <vector>
#include <map>
#include <set>
#include <list>
#include <fstream>
#include <iostream>
#include <sstream>
#include <cstring>
#include <cmath>
#include "lvref_eigen2.hpp"
#include <boost/shared_ptr.hpp>
#define private public
#define protected public
#include <larvio/larvio.h>
#undef private
#undef protected

namespace larvio {
// the moving-start initialiser is not part of this build (see the header comment)
bool DynamicInitializer::tryDynInit(const std::vector<ImuData>&, MonoCameraMeasurementPtr) { return false; }
void DynamicInitializer::assignInitialState(std::vector<ImuData>&, Eigen::Vector3d&, Eigen::Vector3d&, IMUState&) {}
GlobalSFM::GlobalSFM() {}
}

using namespace larvio;

struct RefVio {
    LarVio* vio = nullptr;
    std::vector<ImuData> imu;
    MonoCameraMeasurement msg;
};

extern "C" {

void* lvref_larvio_create(const char* yaml_path)
{
    IMUState::next_id = 0; Feature::next_id = 0;
    std::string p(yaml_path);
    std::streambuf* keep = std::cout.rdbuf(); std::ostringstream sink; std::cout.rdbuf(sink.rdbuf());      // (the reference prints its set-up)
    RefVio* r = new RefVio(); r->vio = new LarVio(p);
    const bool ok = r->vio->initialize();
    std::cout.rdbuf(keep);
    if (!ok) { delete r->vio; delete r; return nullptr; }
    return r;
}
void lvref_larvio_destroy(void* h) { RefVio* r = (RefVio*)h; if (r) { delete r->vio; delete r; } }

// start from a given state instead of an initialiser: what LarVio::processFeatures does when tryIncInit succeeds (larvio.cpp:375-391),
// with the state handed in (the oracle's lvo_ekf_set_state is the same bypass; both let in-state features in at once)
void lvref_larvio_set_state(void* h, double t, const double* q, const double* p, const double* v, const double* bg, const double* ba,
                            const double* gyro_old, const double* acc_old)
{
    LarVio& L = *((RefVio*)h)->vio;
    IMUState& s = L.state_server.imu_state;
    s.time = t; s.orientation = Eigen::Vector4d(q[0], q[1], q[2], q[3]);
    s.position = Eigen::Vector3d(p[0], p[1], p[2]); s.velocity = Eigen::Vector3d(v[0], v[1], v[2]);
    s.gyro_bias = Eigen::Vector3d(bg[0], bg[1], bg[2]); s.acc_bias = Eigen::Vector3d(ba[0], ba[1], ba[2]);
    L.m_gyro_old = Eigen::Vector3d(gyro_old[0], gyro_old[1], gyro_old[2]); L.m_acc_old = Eigen::Vector3d(acc_old[0], acc_old[1], acc_old[2]);
    L.is_gravity_set = true; L.bFirstFeatures = true;
    L.take_off_stamp = t; L.last_ZUPT_time = t - 10.0; L.last_update_time = t;
    L.state_server.imu_state_FEJ_now = s;
}

// the start as an initialiser leaves it (larvio.cpp:384: last_ZUPT_time = the state time, so in-state features wait 5 s)
void lvref_larvio_set_last_zupt_time(void* h, double t) { ((RefVio*)h)->vio->last_ZUPT_time = t; }

// one LarVio::processFeatures call.  feats: n x 9 doubles (id, u, v, u_init, v_init, u_vel, v_vel, u_init_vel, v_init_vel);
// imu: m x 7 doubles (t, gyro, acc) APPENDED to the driver's buffer (the call erases what it consumes, as in app/larvioMain.cpp).
// Returns processFeatures' own answer; *n_left = samples left in the buffer.
int lvref_larvio_process(void* h, double stamp, int n, const double* feats, int m, const double* imu, int* n_left)
{
    RefVio* r = (RefVio*)h;
    for (int i = 0; i < m; ++i) r->imu.push_back(ImuData(imu[7 * i], imu[7 * i + 1], imu[7 * i + 2], imu[7 * i + 3], imu[7 * i + 4], imu[7 * i + 5], imu[7 * i + 6]));
    r->msg.timeStampToSec = stamp; r->msg.features.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        MonoFeatureMeasurement& f = r->msg.features[(size_t)i]; const double* s = feats + 9 * i;
        f.id = (unsigned long long)s[0]; f.u = s[1]; f.v = s[2]; f.u_init = s[3]; f.v_init = s[4]; f.u_vel = s[5]; f.v_vel = s[6]; f.u_init_vel = s[7]; f.v_init_vel = s[8];
    }
    std::streambuf* keep = std::cout.rdbuf(); std::ostringstream sink; std::cout.rdbuf(sink.rdbuf());
    const bool ok = r->vio->processFeatures(&r->msg, r->imu);
    std::cout.rdbuf(keep);
    if (n_left) *n_left = (int)r->imu.size();
    return ok ? 1 : 0;
}

int lvref_larvio_dim(void* h) { return ((RefVio*)h)->vio->state_server.state_cov.rows(); }
int lvref_larvio_initialized(void* h) { return ((RefVio*)h)->vio->is_gravity_set ? 1 : 0; }
// out30: t, q[4], v, p, bg, ba, R_imu_cam0 (row-major 9), t_cam0_imu, td   (the oracle's lvo_ekf_get_state layout)
void lvref_larvio_get_state(void* h, double* o)
{
    LarVio& L = *((RefVio*)h)->vio; const IMUState& s = L.state_server.imu_state;
    o[0] = s.time; for (int k = 0; k < 4; ++k) o[1 + k] = s.orientation(k);
    for (int k = 0; k < 3; ++k) { o[5 + k] = s.velocity(k); o[8 + k] = s.position(k); o[11 + k] = s.gyro_bias(k); o[14 + k] = s.acc_bias(k); o[26 + k] = s.t_cam0_imu(k); }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[17 + 3 * i + j] = s.R_imu_cam0(i, j);
    o[29] = L.state_server.td;
}
void lvref_larvio_get_cov(void* h, double* P)
{   // row-major N x N
    const Eigen::MatrixXd& C = ((RefVio*)h)->vio->state_server.state_cov; const int N = C.rows();
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) P[(size_t)i * N + j] = C(i, j);
}
// clones in window order: per clone 16 doubles (id, time, q[4], p[3], p_fej[3], q_cam... not kept: 0)
int lvref_larvio_get_clones(void* h, double* out, int cap)
{
    LarVio& L = *((RefVio*)h)->vio; int n = 0;
    for (const auto& kv : L.state_server.imu_states_augment) {
        if (n >= cap) break;
        double* o = out + 12 * n; const IMUState_Aug& c = kv.second;
        o[0] = (double)c.id; o[1] = c.time; for (int k = 0; k < 4; ++k) o[2 + k] = c.orientation(k);
        for (int k = 0; k < 3; ++k) { o[6 + k] = c.position(k); o[9 + k] = c.position_FEJ(k); }
        ++n;
    }
    return n;
}
// the clones' camera poses as the triangulation reads them: per clone 8 doubles (id, orientation_cam[4], position_cam[3])
int lvref_larvio_get_clone_cams(void* h, double* out, int cap)
{
    LarVio& L = *((RefVio*)h)->vio; int n = 0;
    for (const auto& kv : L.state_server.imu_states_augment) {
        if (n >= cap) break;
        double* o = out + 8 * n; const IMUState_Aug& c = kv.second;
        o[0] = (double)c.id; for (int k = 0; k < 4; ++k) o[1 + k] = c.orientation_cam(k);
        for (int k = 0; k < 3; ++k) o[5 + k] = c.position_cam(k);
        ++n;
    }
    return n;
}
// in-state features in state order: ids, inverse depth, world position
int lvref_larvio_get_features(void* h, long long* ids, double* idp, double* pos, int cap)
{
    LarVio& L = *((RefVio*)h)->vio; int n = 0;
    for (auto fid : L.state_server.feature_states) {
        if (n >= cap) break;
        const Feature& f = L.map_server[fid];
        ids[n] = (long long)fid; idp[n] = f.invDepth; for (int k = 0; k < 3; ++k) pos[3 * n + k] = f.position(k);
        ++n;
    }
    return n;
}
// the public getters (larvio.cpp:2645-2700): T_b_w (row-major 4 x 4), velocity, P_pose (row-major 6 x 6), P_vel (row-major 3 x 3)
void lvref_larvio_getters(void* h, double* T16, double* v3, double* Ppose36, double* Pvel9)
{
    LarVio& L = *((RefVio*)h)->vio;
    const Eigen::Isometry3d T = L.getTbw(); const Eigen::Vector3d v = L.getVel(); const Eigen::Matrix<double, 6, 6> Pp = L.getPpose(); const Eigen::Matrix3d Pv = L.getPvel();
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T16[4 * i + j] = T.matrix()(i, j);
    for (int i = 0; i < 3; ++i) v3[i] = v(i);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ppose36[6 * i + j] = Pp(i, j);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Pvel9[3 * i + j] = Pv(i, j);
}
// the map server's features: id, observations, is_initialized, in_state (for looking into a disagreement)
int lvref_larvio_map(void* h, long long* ids, int* n_obs, int* flags, int cap)
{
    LarVio& L = *((RefVio*)h)->vio; int n = 0;
    for (const auto& kv : L.map_server) { if (n >= cap) break; ids[n] = (long long)kv.first; n_obs[n] = (int)kv.second.observations.size(); flags[n] = (kv.second.is_initialized ? 1 : 0) | (kv.second.in_state ? 2 : 0) | (kv.second.ekf_feature ? 4 : 0); ++n; }
    return n;
}
// what LarVio::loadParameters (larvio.cpp:58-311) made of the configuration file, in lvk_ekf_config's terms (noises as standard
// deviations: the reference squares them on loading; initial covariances read back from the diagonal it fills):
// 0 if_FEJ 1 estimate_extrin 2 estimate_td 3 if_ZUPT_valid 4 sw_size 5 max_track_len 6 least_observation_number 7 max_features_in_one_grid
// 8 aug_grid_rows 9 aug_grid_cols 10 pub_frequency 11 imu_rate 12 width 13 height 14-17 intrinsics 18 td 19-23 noise gyro / acc / gyro bias /
// acc bias / feature 24-30 initial covariance orientation / velocity / position / gyro bias / acc bias / extrinsic rotation / translation
// 31-33 rotation / translation / tracking-rate threshold 34 feature_translation_threshold 35 zupt_max_feature_dis 36-38 zupt noise v / p / q
// 39 static_duration 40 feature_idp_dim 41 use_schmidt 42 calib_imu 43-51 R_imu_cam0 (row-major) 52-54 t_cam0_imu
void lvref_larvio_params(void* h, double* o)
{
    LarVio& L = *((RefVio*)h)->vio; const Eigen::MatrixXd& P = L.state_server.state_cov;
    o[0] = L.if_FEJ_config; o[1] = L.estimate_extrin; o[2] = L.estimate_td; o[3] = L.if_ZUPT_valid; o[4] = L.sw_size; o[5] = L.max_track_len;
    o[6] = L.least_Obs_Num; o[7] = L.max_features; o[8] = L.grid_rows; o[9] = L.grid_cols; o[10] = L.features_rate; o[11] = L.imu_rate;
    o[12] = L.cam_resolution[0]; o[13] = L.cam_resolution[1]; for (int k = 0; k < 4; ++k) o[14 + k] = L.cam_intrinsics[k];
    o[18] = L.td_input;
    o[19] = std::sqrt(L.imu_gyro_noise); o[20] = std::sqrt(L.imu_acc_noise); o[21] = std::sqrt(L.imu_gyro_bias_noise); o[22] = std::sqrt(L.imu_acc_bias_noise);
    o[23] = std::sqrt(L.feature_observation_noise);
    o[24] = P(0, 0); o[25] = P(3, 3); o[26] = P(6, 6); o[27] = P(9, 9); o[28] = P(12, 12); o[29] = P(15, 15); o[30] = P(18, 18);
    o[31] = L.rotation_threshold; o[32] = L.translation_threshold; o[33] = L.tracking_rate_threshold; o[34] = Feature::optimization_config.translation_threshold;
    o[35] = L.zupt_max_feature_dis; o[36] = std::sqrt(L.zupt_noise_v); o[37] = std::sqrt(L.zupt_noise_p); o[38] = std::sqrt(L.zupt_noise_q);
    o[39] = L.Static_Duration; o[40] = L.feature_idp_dim; o[41] = L.use_schmidt; o[42] = L.calib_imu;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[43 + 3 * i + j] = L.state_server.imu_state.R_imu_cam0(i, j); o[52 + i] = L.state_server.imu_state.t_cam0_imu(i); }
}
int lvref_larvio_map_size(void* h) { return (int)((RefVio*)h)->vio->map_server.size(); }
double lvref_larvio_chi2(void* h, int dof) { return ((RefVio*)h)->vio->chi_squared_test_table[dof]; }

}  // extern "C"
