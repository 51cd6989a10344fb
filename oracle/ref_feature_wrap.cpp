// ref_feature_wrap.cpp — ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE's own Feature code
// (/root/reference/include/larvio/feature.hpp: checkMotion :334-381, initializePosition :383-552, initializePosition_AssignAnchor :554-721,
// initializeInvParamPosition :723-890, with cost / jacobian / generateInitialGuess :252-332), compiled where it lies against the Eigen
// stand-in of ref_shim/ (oracle/Makefile, target `ref`).  The static members the reference defines in src/larvio.cpp:33-39 are defined
// here with the same initial values.  The wrapper only fills the reference's own containers (IMUStateServer, Feature::observations)
// from flat arrays and reads its members back.
#include <larvio/feature.hpp>
#include <cstring>

namespace larvio {
StateIDType IMUState::next_id = 0;                                                   // larvio.cpp:33
Eigen::Vector3d IMUState::gravity = Eigen::Vector3d(0, 0, -GRAVITY_ACCELERATION);     // :34
Eigen::Isometry3d IMUState::T_imu_body = Eigen::Isometry3d::Identity();               // :35
FeatureIDType Feature::next_id = 0;                                                   // :38
Feature::OptimizationConfig Feature::optimization_config;                             // :39
}

using namespace larvio;

static void fill(IMUStateServer& srv, Feature& f, int n_states, const long long* state_ids, const double* q_cam, const double* p_cam,
                 int n_obs, const long long* obs_ids, const double* obs_uv)
{
    for (int i = 0; i < n_states; ++i) {
        IMUState_Aug s(state_ids[i]);
        s.orientation_cam = Eigen::Vector4d(q_cam[4 * i], q_cam[4 * i + 1], q_cam[4 * i + 2], q_cam[4 * i + 3]);
        s.position_cam = Eigen::Vector3d(p_cam[3 * i], p_cam[3 * i + 1], p_cam[3 * i + 2]);
        srv[state_ids[i]] = s;
    }
    for (int i = 0; i < n_obs; ++i) f.observations[obs_ids[i]] = Eigen::Vector2d(obs_uv[2 * i], obs_uv[2 * i + 1]);
}

extern "C" {

// translation_threshold: config feature_translation_threshold (larvio.cpp:77); returns checkMotion's answer
int lvref_feature_check_motion(int n_states, const long long* state_ids, const double* q_cam, const double* p_cam,
                               int n_obs, const long long* obs_ids, const double* obs_uv, int if_tracked, double translation_threshold)
{
    IMUStateServer srv; Feature f(1);
    fill(srv, f, n_states, state_ids, q_cam, p_cam, n_obs, obs_ids, obs_uv);
    Feature::optimization_config.translation_threshold = translation_threshold;
    return f.checkMotion(srv, if_tracked != 0) ? 1 : 0;
}

// mode 0 initializePosition(curr_id), 1 initializePosition_AssignAnchor, 2 initializeInvParamPosition(curr_id).
// is_initialized / position_in: the feature's state before the call (the is_initialized branch starts from position).
// out[0..2] position, [3..5] position_FEJ, [6] invDepth, [7..9] obs_anchor, [10] id_anchor, [11..13] invParam, [14] is_initialized after
int lvref_feature_initialize(int mode, int n_states, const long long* state_ids, const double* q_cam, const double* p_cam,
                             int n_obs, const long long* obs_ids, const double* obs_uv, long long curr_id,
                             int is_initialized, const double* position_in, double* out)
{
    IMUStateServer srv; Feature f(1);
    fill(srv, f, n_states, state_ids, q_cam, p_cam, n_obs, obs_ids, obs_uv);
    f.is_initialized = is_initialized != 0;
    f.position = Eigen::Vector3d(position_in[0], position_in[1], position_in[2]);
    f.position_FEJ = Eigen::Vector3d(0, 0, 0);
    f.invDepth = 0; f.obs_anchor = Eigen::Vector3d(0, 0, 0);
    bool ok;
    if (mode == 0) ok = f.initializePosition(srv, curr_id);
    else if (mode == 1) ok = f.initializePosition_AssignAnchor(srv);
    else ok = f.initializeInvParamPosition(srv, curr_id);
    for (int k = 0; k < 3; ++k) { out[k] = f.position(k); out[3 + k] = f.position_FEJ(k); out[7 + k] = f.obs_anchor(k); out[11 + k] = f.invParam(k); }
    out[6] = f.invDepth; out[10] = (double)f.id_anchor; out[14] = f.is_initialized ? 1.0 : 0.0;
    return ok ? 1 : 0;
}

// the reference's own quaternion / skew helpers (include/larvio/math_utils.hpp:26-38, 85-102, 113-133), for the product's host math
// (larvio_amd/csrc/be_host_math.h: skew3, small_angle_quat): out9 = skewSymmetric(w) row-major, out4a = smallAngleQuaternion(w) [x y z w],
// out4b = getSmallAngleQuaternion(w) [x y z w]
void lvref_math_small_angle(const double* w, double* out9, double* out4a, double* out4b)
{
    const Eigen::Vector3d v(w[0], w[1], w[2]);
    const Eigen::Matrix3d S = skewSymmetric(v);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out9[3 * i + j] = S(i, j);
    const Eigen::Vector4d q = smallAngleQuaternion(v);
    for (int k = 0; k < 4; ++k) out4a[k] = q(k);
    const Eigen::Quaterniond g = getSmallAngleQuaternion(v);
    out4b[0] = g.x(); out4b[1] = g.y(); out4b[2] = g.z(); out4b[3] = g.w();
}

// ... and the conversions (math_utils.hpp:54-83 quaternionMultiplication - normalises its result -, :145-160 quaternionToRotation, :169-232
// rotationToQuaternion): R9 = quaternionToRotation(q) row-major, q2 = rotationToQuaternion(R9), qp = quaternionMultiplication(q, p)
void lvref_math_quat(const double* q, const double* p, double* R9, double* q2, double* qp)
{
    const Eigen::Vector4d a(q[0], q[1], q[2], q[3]), b(p[0], p[1], p[2], p[3]);
    const Eigen::Matrix3d R = quaternionToRotation(a);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R9[3 * i + j] = R(i, j);
    const Eigen::Vector4d c = rotationToQuaternion(R);
    const Eigen::Vector4d d = quaternionMultiplication(a, b);
    for (int k = 0; k < 4; ++k) { q2[k] = c(k); qp[k] = d(k); }
}

}  // extern "C"
