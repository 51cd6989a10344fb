// ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  C entry points around the REFERENCE's own moving-start initialiser -
// /root/reference/src/DynamicInitializer.cpp (tryDynInit / processIMU / processImage / initialStructure / relativePose /
// visualInitialAlign / slideWindow / assignInitialState), src/initial_sfm.cpp (GlobalSFM::construct: the PnP / triangulation chain and the
// bundle adjustment's problem set-up), src/solve_5pts.cpp (MotionEstimator::solveRelativeRT and the excerpt of OpenCV's own
// decomposeEssentialMat / recoverPose it carries), src/initial_alignment.cpp, src/feature_manager.cpp and include/Initializer/
// ImuPreintegration.h, compiled where they lie (oracle/Makefile, target `ref` -> oracle/_ref/liblvref_dyninit.so; never copied) against
// the stand-ins of oracle/ref_shim4/ (Eigen as in ref_shim2; the slice of OpenCV's Mat algebra solve_5pts.cpp is written in; cv::solvePnP
// / Rodrigues / SVD::compute / triangulatePoints and the Ceres names served by small routines written there; cv::findFundamentalMat by
// the oracle's RANSAC restatement).
#include <string>
#include <vector>
#include <map>
#include <list>
#include <iostream>
#include <sstream>
#include <cstring>
#include "lvref_sfm.hpp"
#include "lvref_ceres.hpp"
#include <boost/shared_ptr.hpp>
extern "C" {
#include "lvo.h"
}
#define private public
#define protected public
#include "Initializer/DynamicInitializer.h"
#undef private
#undef protected

using namespace larvio;
struct RefDyn { DynamicInitializer* d = nullptr; std::vector<ImuData> imu; MonoCameraMeasurement msg; };

extern "C" {
// R_c2b: rotation taking camera vectors to the body (= R_imu_cam0^T of the filter), t_bc_b: camera position in the body, as
// FlexibleInitializer hands them over (larvio.cpp:343-349); noise arguments are standard deviations
void* lvref_dyninit_create(double td, const double* R_c2b, const double* t_bc_b, double acc_n, double acc_w, double gyr_n, double gyr_w, double imu_img_time_th)
{
    Eigen::Matrix3d I = Eigen::Matrix3d::Identity(), Z = Eigen::Matrix3d::Zero(), R; Eigen::Vector3d t(t_bc_b[0], t_bc_b[1], t_bc_b[2]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = R_c2b[3 * i + j];
    RefDyn* r = new RefDyn(); r->d = new DynamicInitializer(td, I, I, Z, acc_n, acc_w, gyr_n, gyr_w, R, t, imu_img_time_th);
    return r;
}
void lvref_dyninit_destroy(void* h) { RefDyn* r = (RefDyn*)h; if (r) { delete r->d; delete r; } }
// one FlexibleInitializer step for the moving-start half: tryDynInit with the message and the driver's buffer (new samples APPENDED; nothing
// is erased until it succeeds), then assignInitialState.  out: state_time, orientation[4] (x y z w), velocity[3], gyro_bias[3], last
// gyro[3], last acc[3], gravity in the reference camera frame[3], samples erased.  Returns 1 on success.
int lvref_dyninit_try(void* h, double stamp, int n, const double* feats, int m, const double* imu, double* out)
{
    RefDyn* r = (RefDyn*)h;
    for (int i = 0; i < m; ++i) r->imu.push_back(ImuData(imu[7 * i], imu[7 * i + 1], imu[7 * i + 2], imu[7 * i + 3], imu[7 * i + 4], imu[7 * i + 5], imu[7 * i + 6]));
    r->msg.timeStampToSec = stamp; r->msg.features.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        MonoFeatureMeasurement& f = r->msg.features[(size_t)i]; const double* s = feats + 9 * i;
        f.id = (unsigned long long)s[0]; f.u = s[1]; f.v = s[2]; f.u_init = s[3]; f.v_init = s[4]; f.u_vel = s[5]; f.v_vel = s[6]; f.u_init_vel = s[7]; f.v_init_vel = s[8];
    }
    if (!r->d->tryDynInit(r->imu, &r->msg)) return 0;
    IMUState st; Eigen::Vector3d go, ao; const size_t before = r->imu.size();
    r->d->assignInitialState(r->imu, go, ao, st);
    out[0] = st.time; for (int k = 0; k < 4; ++k) out[1 + k] = st.orientation(k);
    for (int k = 0; k < 3; ++k) { out[5 + k] = st.velocity(k); out[8 + k] = st.gyro_bias(k); out[11 + k] = go(k); out[14 + k] = ao(k); out[17 + k] = r->d->g(k); }
    out[20] = (double)(before - r->imu.size());
    return 1;
}
int lvref_dyninit_frame_count(void* h) { return ((RefDyn*)h)->d->frame_count; }
}  // extern "C"
