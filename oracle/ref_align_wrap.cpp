// ref_align_wrap.cpp — ORACLE / TEST INFRASTRUCTURE ONLY.  C entry point around the REFERENCE's own visual-inertial alignment
// (/root/reference/src/initial_alignment.cpp: solveGyroscopeBias :10-46, TangentBasis :49-62, RefineGravity :65-128, LinearAlignment
// :131-201, VisualIMUAlignment :204-212), compiled where it lies - together with include/Initializer/ImuPreintegration.h - against the
// Eigen / boost stand-ins of ref_shim/ (oracle/Makefile, target `ref`).  The wrapper fills the reference's own map<double, ImageFrame>
// (rotation, position, pre-integration of every window frame) from flat arrays and returns what the call leaves behind.
#include <Initializer/initial_alignment.h>

extern "C" {

// n_frames window frames with keys t[], rotations R[] (row-major 3x3: body-to-c0 as all_image_frame holds them) and positions T[].
// Frame j >= 1 carries the pre-integration from frame j-1: linearised at head[6 j .. 6 j + 5] = (acc0, gyr0), zero accelerometer bias and
// bg0, then its n_samples[j] samples (dt, acc[3], gyr[3]) pushed, taken consecutively from samples[].
// out: [0] VisualIMUAlignment's answer, [1..3] Bgs[0] after the call, [4..6] g, [7] n_state, [8..] x (n_state values; the last one is the scale)
int lvref_visual_imu_alignment(int n_frames, const double* t, const double* R, const double* T, const int* n_samples, const double* head, const double* samples,
                               const double* bg0, const double* tic, double* out)
{
    using namespace larvio;
    std::map<double, ImageFrame> frames;
    Eigen::Vector3d Bgs[WINDOW_SIZE + 1];
    for (int i = 0; i <= WINDOW_SIZE; ++i) Bgs[i] = Eigen::Vector3d(bg0[0], bg0[1], bg0[2]);
    size_t s = 0;
    for (int j = 0; j < n_frames; ++j) {
        ImageFrame f;
        for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) f.R(a, b) = R[9 * j + 3 * a + b]; f.T(a) = T[3 * j + a]; }
        f.t = t[j];
        if (j > 0) {
            const double* h = head + 6 * j;
            f.pre_integration.reset(new IntegrationBase(Eigen::Vector3d(h[0], h[1], h[2]), Eigen::Vector3d(h[3], h[4], h[5]), Eigen::Vector3d(0, 0, 0), Bgs[0],
                                                        0.08, 0.00004, 0.004, 2.0e-6));
            for (int k = 0; k < n_samples[j]; ++k, ++s) {
                const double* q = samples + 7 * s;
                f.pre_integration->push_back(q[0], Eigen::Vector3d(q[1], q[2], q[3]), Eigen::Vector3d(q[4], q[5], q[6]));
            }
        }
        frames[t[j]] = f;
    }
    Eigen::Vector3d g(0, 0, 0); Eigen::VectorXd x;
    const bool ok = VisualIMUAlignment(frames, Bgs, g, x, Eigen::Vector3d(tic[0], tic[1], tic[2]));
    out[0] = ok ? 1.0 : 0.0;
    for (int k = 0; k < 3; ++k) { out[1 + k] = Bgs[0](k); out[4 + k] = g(k); }
    out[7] = (double)x.size();
    for (int k = 0; k < x.size(); ++k) out[8 + k] = x(k);
    return ok ? 1 : 0;
}

}  // extern "C"
