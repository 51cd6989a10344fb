// ref_preint_wrap.cpp — ORACLE / TEST INFRASTRUCTURE ONLY.  C entry point around the REFERENCE's own IMU pre-integration
// (/root/reference/include/Initializer/ImuPreintegration.h: IntegrationBase - midPointIntegration :62-142, propagate :144-167,
// repropagate :48-61), compiled where it lies against the Eigen stand-in of ref_shim/ (oracle/Makefile, target `ref`).
#include <Initializer/ImuPreintegration.h>

extern "C" {

// acc0/gyr0: the first sample (the linearisation point); n further samples (dt, acc, gyr) are pushed; when rebias != 0 the buffer is
// then re-propagated about (ba2, bg2).  out: [0..2] delta_p, [3..6] delta_q (x y z w), [7..9] delta_v, [10] sum_dt,
// [11..19] d(delta_q)/d(bg) = jacobian.block<3,3>(O_R, O_BG) row-major, [20..28] dp/dbg, [29..37] dv/dbg, [38..46] dp/dba, [47..55] dv/dba
int lvref_preintegrate(const double* acc0, const double* gyr0, const double* ba, const double* bg, int n, const double* dt, const double* acc, const double* gyr,
                       int rebias, const double* ba2, const double* bg2, double* out)
{
    using namespace larvio;
    IntegrationBase b(Eigen::Vector3d(acc0[0], acc0[1], acc0[2]), Eigen::Vector3d(gyr0[0], gyr0[1], gyr0[2]), Eigen::Vector3d(ba[0], ba[1], ba[2]),
                      Eigen::Vector3d(bg[0], bg[1], bg[2]), 0.08, 0.00004, 0.004, 2.0e-6);
    for (int i = 0; i < n; ++i) b.push_back(dt[i], Eigen::Vector3d(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]), Eigen::Vector3d(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]));
    if (rebias) b.repropagate(Eigen::Vector3d(ba2[0], ba2[1], ba2[2]), Eigen::Vector3d(bg2[0], bg2[1], bg2[2]));
    for (int k = 0; k < 3; ++k) { out[k] = b.delta_p(k); out[7 + k] = b.delta_v(k); }
    out[3] = b.delta_q.x(); out[4] = b.delta_q.y(); out[5] = b.delta_q.z(); out[6] = b.delta_q.w();
    out[10] = b.sum_dt;
    const int blocks[5][2] = {{O_R, O_BG}, {O_P, O_BG}, {O_V, O_BG}, {O_P, O_BA}, {O_V, O_BA}};
    for (int q = 0; q < 5; ++q) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[11 + 9 * q + 3 * i + j] = b.jacobian(blocks[q][0] + i, blocks[q][1] + j);
    return 0;
}

}  // extern "C"
