// ref_static_wrap.cpp — ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE's own static initialiser
// (/root/reference/src/StaticInitializer.cpp: tryIncInit :12-58, initializeGravityAndBias :61-109, assignInitialState :112-145), compiled
// where it lies against the Eigen stand-in of ref_shim/ (oracle/Makefile, target `ref`).  The wrapper keeps one StaticInitializer
// object per handle, feeds it messages and IMU buffers from flat arrays and reads the state it assigns.
#include <Initializer/StaticInitializer.h>
#include <vector>

using namespace larvio;

namespace larvio {                                    // static members the reference defines in src/larvio.cpp:33-35
IMUState::StateIDType IMUState::next_id = 0;
Eigen::Vector3d IMUState::gravity = Eigen::Vector3d(0, 0, -GRAVITY_ACCELERATION);
Eigen::Isometry3d IMUState::T_imu_body = Eigen::Isometry3d::Identity();
}

extern "C" {

void* lvref_static_create(double max_feature_dis, int static_num, double td, const double* Ma, const double* Tg, const double* As)
{
    Eigen::Matrix3d ma, tg, as;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { ma(i, j) = Ma[3 * i + j]; tg(i, j) = Tg[3 * i + j]; as(i, j) = As[3 * i + j]; }
    return new StaticInitializer(max_feature_dis, static_num, td, ma, tg, as);
}
void lvref_static_destroy(void* h) { delete (StaticInitializer*)h; }

// one message: ids / uv (n features), the driver's IMU buffer (n_imu samples: t, gyro[3], acc[3]).  Returns tryIncInit's answer; on
// success out[0] = state time, [1..4] orientation (x y z w), [5..7] gyro bias, [8] IMU samples erased by assignInitialState,
// [9..11] m_gyro_old, [12..14] m_acc_old
int lvref_static_try(void* h, double ts, int n, const long long* ids, const double* uv, int n_imu, const double* imu7, double* out)
{
    StaticInitializer* si = (StaticInitializer*)h;
    MonoCameraMeasurement msg; msg.timeStampToSec = ts;
    for (int i = 0; i < n; ++i) { MonoFeatureMeasurement f; f.id = ids[i]; f.u = uv[2 * i]; f.v = uv[2 * i + 1]; f.u_init = f.v_init = -1; f.u_vel = f.v_vel = f.u_init_vel = f.v_init_vel = 0; msg.features.push_back(f); }
    std::vector<ImuData> buf;
    for (int i = 0; i < n_imu; ++i) buf.push_back(ImuData(imu7[7 * i], imu7[7 * i + 1], imu7[7 * i + 2], imu7[7 * i + 3], imu7[7 * i + 4], imu7[7 * i + 5], imu7[7 * i + 6]));
    if (!si->tryIncInit(buf, &msg)) return 0;
    Eigen::Vector3d g_old(0, 0, 0), a_old(0, 0, 0); IMUState st;
    const size_t before = buf.size();
    si->assignInitialState(buf, g_old, a_old, st);
    out[0] = st.time; for (int k = 0; k < 4; ++k) out[1 + k] = st.orientation(k); for (int k = 0; k < 3; ++k) { out[5 + k] = st.gyro_bias(k); out[9 + k] = g_old(k); out[12 + k] = a_old(k); }
    out[8] = (double)(before - buf.size());
    return 1;
}

}  // extern "C"
