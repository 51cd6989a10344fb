/*
 * lvo.h — CPU ORACLE for the LARVIO per-frame hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain-C restatement of the reference's algorithm for the path
 * SURVEY.md §8 scopes (front-end: src/image_processor.cpp, src/ORBDescriptor.cpp;
 * back-end update: src/larvio.cpp, include/larvio/feature.hpp).  It exists so the HIP
 * path can be checked against something.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (larvio_amd/, include/) never
 * includes, links or calls anything here.
 *
 * PINNING.  The reference has no tests, golden vectors or fixtures and
 * cannot be built as a whole here (OpenCV, Eigen, SuiteSparse, Boost, Ceres are absent).
 * BACK-END: PINNED to the reference's own code compiled in place - src/larvio.cpp
 * (LarVio::processFeatures) with FlexibleInitializer / StaticInitializer / feature_manager
 * against stand-in headers (oracle/ref_shim2/, oracle/Makefile target `ref`): the filter in
 * be_filter.c / be_core.c agrees with it after every update (tests/test_oracle_ref_larvio.py,
 * tests/golden/ref_larvio.npz written by the reference); likewise, in smaller pieces, the ORB
 * descriptor, the triangulation, the static initialiser, the moving-start initialiser's window
 * bookkeeping, pre-integration and alignment (PARITY.md).
 * FRONT-END: the ORCHESTRATION in fe_pipeline.c is PINNED to src/image_processor.cpp compiled in
 * place (ImageProcessor::processImage, byte for byte after every frame: tests/
 * test_oracle_ref_imgproc.py), and the ORB block to src/ORBDescriptor.cpp; the IMAGE ALGORITHMS
 * stay PARITY UNPINNED against reference outputs - that arithmetic lives in OpenCV
 * (un-vendored, unpinned: README.md:58 names 3.4.6 / 4.1.2), and in that comparison the
 * reference's cv:: calls are served by these very restatements; the functions
 * below restate the published algorithms of those OpenCV calls ("[upstream]" in comments) and
 * follow the reference's own call sites for parameters.  Where OpenCV's own result depends on
 * its SIMD dispatch (float summation order in LK / boxFilter), the oracle fixes ONE order and
 * says so.
 *
 * All floating-point code here must be built with -ffp-contract=off (see Makefile): the
 * HIP kernels are built the same way so that float32 stages agree bit-for-bit.
 */
#ifndef LVO_H
#define LVO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y; } lvo_pt2f;

/* include/sensors/ImuData.hpp:17-43 */
typedef struct { double t; double gyro[3]; double acc[3]; } lvo_imu;

/* include/larvio/feature_msg.h:15-44 (72 bytes) */
typedef struct {
    uint64_t id;
    double u, v, u_init, v_init, u_vel, v_vel, u_init_vel, v_init_vel;
} lvo_feature_obs;

/* ------------------------------------------------------------------ image passes */

/* cv::createCLAHE(clip, Size(tiles_x,tiles_y))->apply  [upstream clahe.cpp];
 * call site image_processor.cpp:322-325 (clip 3.0, 8x8). */
/* host threads for the loops with independent iterations (rows, tiles, tracks, key points, features); 1 = the reference's
 * single-threaded design.  Results do not depend on the count.  Used by bench.py's all-core cpu_baseline leg. */
void lvo_set_threads(int n);
int lvo_get_threads(void);
void lvo_clahe_u8(const uint8_t* src, int w, int h, int sstride,
                  uint8_t* dst, int dstride, double clip, int tiles_x, int tiles_y);

/* cv::pyrDown u8: separable [1 4 6 4 1], (s+128)>>8, BORDER_REFLECT_101,
 * dst = ((w+1)/2,(h+1)/2)  [upstream pyramids.cpp]. */
void lvo_pyr_down_u8(const uint8_t* src, int w, int h, int sstride,
                     uint8_t* dst, int dstride);

/* calcSharrDeriv [upstream lkpyramid.cpp]: dst interleaved int16 (Ix,Iy), dstride in int16 units. */
void lvo_scharr_deriv(const uint8_t* src, int w, int h, int sstride,
                      int16_t* dst, int dstride);

/* One LK pyramid = what cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), max_level,
 * withDerivatives=true, BORDER_REFLECT_101, BORDER_CONSTANT, false) returns
 * (image_processor.cpp:329-333).  Level l image is padded by `pad`(=win) pixels of
 * reflect-101; derivative planes are padded with zeros. */
#define LVO_MAX_LEVELS 8
typedef struct {
    int n_levels;          /* number of levels actually built (max_level+1 or fewer) */
    int pad;               /* = win */
    int w[LVO_MAX_LEVELS], h[LVO_MAX_LEVELS];
    int istride[LVO_MAX_LEVELS];   /* bytes per padded image row  = w+2*pad */
    int dstride[LVO_MAX_LEVELS];   /* int16 per padded deriv row = 2*(w+2*pad) */
    uint8_t* img[LVO_MAX_LEVELS];  /* padded buffers; pixel (x,y) at img[(y+pad)*istride + x+pad] */
    int16_t* der[LVO_MAX_LEVELS];  /* (Ix,Iy) of (x,y) at der[(y+pad)*dstride + 2*(x+pad)] */
} lvo_pyramid;

void lvo_pyramid_build(const uint8_t* img, int w, int h, int stride, int win, int max_level,
                       lvo_pyramid* out);
void lvo_pyramid_free(lvo_pyramid* p);

/* ORBdescriptor::initializeLayerAndPyramid, level 0 only (ORBDescriptor.cpp:418-484).
 * `pyr` level 0 is the source (the reference passes curr_pyramid_[0], a ROI inside the
 * 21-padded LK buffer, so copyMakeBorder without BORDER_ISOLATED sees that padding).
 * Outputs: ext = (h+64)x(w+64) raw mosaic, blur = same with the interior w x h region
 * replaced by GaussianBlur 7x7 sigma 2 (border stays raw). */
#define LVO_ORB_BORDER 32
void lvo_orb_prepare(const lvo_pyramid* pyr, uint8_t* ext, uint8_t* blur);

/* cv::goodFeaturesToTrack(img, maxCorners, 0.01, min_distance, mask, 3, false) [upstream].
 * img = level-0 of pyr.  mask may be NULL (all 255); mask stride = w.
 * Returns number of corners written (<= cap).  max_corners <= 0 = unlimited. */
int lvo_good_features(const lvo_pyramid* pyr, const uint8_t* mask, int max_corners,
                      double quality, double min_distance, lvo_pt2f* out, int cap);
/* the min-eigenvalue response map alone (w*h floats), for stage-level parity tests */
void lvo_min_eigen_map(const lvo_pyramid* pyr, float* eig);

/* ------------------------------------------------------------------ per-point stages */

/* cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, prev, next, status, noArray(), Size(win,win),
 * maxLevel, TermCriteria(COUNT+EPS, max_iter, eps), OPTFLOW_USE_INITIAL_FLOW, 1e-4)
 * [upstream lkpyramid.cpp LKTrackerInvoker]; call sites image_processor.cpp:368,405,558,618,830,870.
 * next_pts is in/out (initial flow in, result out).  iters_out (optional, n*levels ints) =
 * executed iterations per point and level (for the algorithmic-bytes figure). */
void lvo_set_lk_float_accum(int on);   /* sensitivity probe: OpenCV's scalar float32 running sums instead of the exact ones (never on in parity tests) */
void lvo_lk_track(const lvo_pyramid* prev, const lvo_pyramid* next,
                  const lvo_pt2f* prev_pts, lvo_pt2f* next_pts, uint8_t* status, int n,
                  int max_iter, double eps, int* iters_out);

/* ORBdescriptor::computeDescriptors(pts, levels=0) (ORBDescriptor.cpp:386-416): IC_Angle on
 * ext, 256 rotated-BRIEF tests on blur.  desc = n*32 bytes; angle_out optional. */
void lvo_orb_describe(const uint8_t* ext, const uint8_t* blur, int w, int h,
                      const lvo_pt2f* pts, int n, uint8_t* desc, float* angle_out);
/* ORBdescriptor::computeDescriptorDistance (ORBDescriptor.h:43-59) */
int lvo_hamming256(const uint8_t* a, const uint8_t* b);
/* test hook: the restated cosf / sinf (fe_track.c, ORBDescriptor.cpp:343) against this host's libm on the floats with bit patterns
 * lo_bits..hi_bits: number of inputs where either differs, *first_bad = the smallest such pattern */
long lvo_sincosf_sweep(uint32_t lo_bits, uint32_t hi_bits, uint32_t* first_bad);
/* cv::fastAtan2 [upstream mathfuncs_core] (degrees) */
float lvo_fast_atan2(float y, float x);

/* ImageProcessor::undistortPoints (image_processor.cpp:1040-1072).
 * model 0 = radtan (cv::undistortPoints, 5 iterations), 1 = equidistant (cv::fisheye).
 * K_new given as (fx,fy,cx,cy); rectification = identity (every call site). */
void lvo_undistort_points(const lvo_pt2f* in, int n, const double intr[4], int model,
                          const double dist[4], const double new_intr[4], lvo_pt2f* out);

/* cv::findFundamentalMat(p1, p2, FM_RANSAC, thresh, conf, mask) [upstream fundam.cpp, ptsetreg.cpp]
 * (image_processor.cpp:498,755,968).  Returns 1 when a mask of n bytes was written, 0 when
 * OpenCV would leave the mask untouched (n < 7).  n==7: all ones; 8..14: LMedS; >=15: RANSAC. */
int lvo_find_fundamental_mask(const lvo_pt2f* p1, const lvo_pt2f* p2, int n,
                              double thresh, double conf, uint8_t* mask);
/* the same with the matrix findFundamentalMat returns (the best minimal-sample model; zeros = the empty Mat) */
int lvo_find_fundamental(const lvo_pt2f* p1, const lvo_pt2f* p2, int n,
                         double thresh, double conf, uint8_t* mask, double* F);
/* the RANSAC branch alone, any n >= 8 (stage-level parity with the HIP kernel).
 * iters_out = number of hypotheses drawn. */
int lvo_ransac_fundamental(const lvo_pt2f* p1, const lvo_pt2f* p2, int n,
                           double thresh, double conf, int max_iters,
                           uint8_t* mask, int* iters_out);
/* FMEstimatorCallback::runKernel 7-point: returns number of models (0..3), F row-major 9 each */
int lvo_fundamental_7pt(const lvo_pt2f* m1, const lvo_pt2f* m2, double* F);

/* integrateImuData + predictFeatureTracking (image_processor.cpp:222-293):
 * mean gyro over [t_prev-0.0049, t_curr+0.0049) -> Rodrigues -> R^T ; H = K R K^-1 (float32). */
void lvo_predict_homography(const lvo_imu* imu, int n_imu, double t_prev, double t_curr,
                            const double R_cam_imu[9], const double intr[4], float H[9]);
void lvo_apply_homography(const float H[9], const lvo_pt2f* in, int n, lvo_pt2f* out);

/* ------------------------------------------------------------------ the front-end object */

typedef struct {
    int width, height;
    int pyramid_levels;      /* euroc.yaml:43  (maxLevel; levels built = +1) */
    int patch_size;          /* :44 */
    int max_iteration;       /* :46 */
    double track_precision;  /* :47 */
    int max_features_num;    /* :49 */
    int min_distance;        /* :50 */
    int flag_equalize;       /* :51 */
    int pub_frequency;       /* :52 */
    int distortion_model;    /* 0 radtan, 1 equidistant */
    double intrinsics[4];    /* fx fy cx cy */
    double distortion[4];
    double R_cam_imu[9];     /* row-major; = R_imu_cam^T as image_processor.cpp:93 */
} lvo_fe_config;

typedef struct lvo_frontend lvo_frontend;

lvo_frontend* lvo_frontend_create(const lvo_fe_config* cfg);
void lvo_frontend_destroy(lvo_frontend* fe);
/* ImageProcessor::processImage (image_processor.cpp:130-219).  Returns 1 when a feature
 * message was produced (n_out features in out), else 0. */
int lvo_frontend_process(lvo_frontend* fe, const uint8_t* img, int stride, double ts,
                         const lvo_imu* imu, int n_imu,
                         lvo_feature_obs* out, int cap, int* n_out);
/* introspection for parity tests: live tracks after the call (vectors already rotated,
 * so these are the reference's prev_pts_/pts_ids_/pts_lifetime_/init_pts_/new_pts_) */
int lvo_frontend_tracks(const lvo_frontend* fe, uint64_t* ids, lvo_pt2f* pts, int* lifetime,
                        lvo_pt2f* init_pts, uint8_t* desc, int cap);
int lvo_frontend_new_pts(const lvo_frontend* fe, lvo_pt2f* pts, int cap);
int lvo_frontend_state(const lvo_frontend* fe);   /* 1 FIRST_IMAGE 2 SECOND_IMAGE 3 OTHER_IMAGES */
/* cumulative LK work counters: point-levels run and iterations executed */
void lvo_frontend_lk_stats(const lvo_frontend* fe, uint64_t* point_levels, uint64_t* iterations);

/* ==================================================================== back-end (EKF update)
 * Restates /root/reference/src/larvio.cpp (processFeatures :363-461 and everything it calls) and
 * include/larvio/feature.hpp:252-890 for feature_idp_dim = 1, use_schmidt = 0, calib_imu = 0 or 1
 * (the settings of config/euroc.yaml:8-10,105,108).  Dense algebra that the reference delegates to
 * Eigen / SuiteSparse SPQR is restated with Householder QR and Cholesky: the quantities that reach
 * the state (gate value, K r, (I-KH)P) are invariant to the choice of orthonormal basis / factorisation. */

typedef struct { double R[9]; double t[3]; } lvo_pose;      /* camera-to-world rotation (row-major) + position */

/* one sliding-window clone (IMUState_Aug, imu_state.h:72-117) */
typedef struct {
    int64_t id;
    double time, dt;
    double q[4], p[3], p_fej[3];          /* orientation [x y z w], position, position_FEJ */
    double R_b2c[9], t_c_b[3];            /* extrinsics copied at augmentation time */
    double q_cam[4], p_cam[3];            /* orientation_cam, position_cam */
} lvo_clone;

/* Feature::initializePosition family (feature.hpp:383-890): LM on (alpha,beta,rho) in the LAST pose's frame.
 * poses/obs are the already-selected views in observation order.  use_position: start from position_in
 * (is_initialized branch, :45-48) instead of the two-view guess.  Returns is_valid_solution. */
int lvo_triangulate(const lvo_pose* poses, const double* obs, int n, int use_position, const double* position_in,
                    double* position_out, double* solution_out, double* inv_depth_out, double* obs_anchor_out);
/* Feature::checkMotion (feature.hpp:334-381) on the first and last selected poses */
int lvo_check_motion(const lvo_pose* first, const lvo_pose* last, const double* first_obs, double translation_threshold);

/* featureJacobian_msckf (larvio.cpp:924-981): H ((2M-3) x N, row-major, ld = N) and r from M observations.
 * clone_rank[i] = rank of the observing clone in the window.  Returns the number of rows 2M-3. */
int lvo_msckf_feature_jacobian(const lvo_clone* clones, const int* clone_rank, const double* obs, const double* obs_vel, int M,
                               const double p_w[3], int N, int leg_dim, int if_fej, int estimate_td, double* H, double* r);
/* gatingTest (larvio.cpp:1865-1880): gamma = r^T (H P H^T + sigma2 I)^-1 r */
double lvo_gating_gamma(const double* H, const double* r, int k, int N, const double* P, int ldp, double sigma2);
/* the chi-square table of larvio.cpp:353-357: boost::math::quantile(chi_squared(dof), 0.05), dof 1..99 (0 otherwise) */
double lvo_chi2_table(int dof);
/* QR compression (larvio.cpp:1430-1445): top `cols` rows of Q^T [H r]; H is rows x cols (ld = cols), in place. */
void lvo_qr_compress(double* H, double* r, int rows, int cols);
/* measurement update core (larvio.cpp:1453-1460,1578-1594): dx = K r, P <- (I-KH)P symmetrised. */
void lvo_ekf_update(double* P, int N, int ldp, const double* H, int m, const double* r, double sigma2, double* dx);

typedef struct {
    /* config/euroc.yaml, names as larvio.cpp:58-311 reads them */
    int if_fej, estimate_extrin, estimate_td, if_zupt_valid;
    int sw_size, max_track_len, least_observation_number;
    int max_features_in_one_grid, aug_grid_rows, aug_grid_cols;
    int width, height;
    double intrinsics[4];
    double T_cam_imu[16];
    double td;
    double pub_frequency, imu_rate;   /* features_rate / imu_rate are doubles in the reference (larvio.h:256-259, larvio.cpp:65-67,224) */
    double noise_gyro, noise_acc, noise_gyro_bias, noise_acc_bias, noise_feature;      /* standard deviations */
    double initial_covariance_orientation, initial_covariance_velocity, initial_covariance_position,
           initial_covariance_gyro_bias, initial_covariance_acc_bias, initial_covariance_extrin_rot, initial_covariance_extrin_trans;
    double rotation_threshold, translation_threshold, tracking_rate_threshold, feature_translation_threshold;
    double zupt_max_feature_dis, zupt_noise_v, zupt_noise_p, zupt_noise_q;
    double static_duration;
    int calib_imu_instrinsic;           /* 1: online IMU intrinsics (larvio.cpp:127-186), LEG_DIM 46 instead of 22 */
    int reference_grid;                 /* 1 (the default of lvo_be.py, and the product's): the reference's own bookkeeping for features whose
                                           grid code falls outside the rows x cols cells (undistorted coordinates beyond the image bounds):
                                           grid_map is a std::map<int, vector> (larvio.h:383), so such a code gets a cell of its own that
                                           updateGridMap never clears (larvio.cpp:3356-3366) - it only fills up (:1969-1975).
                                           0: such codes are not counted at all - what this oracle and the product did before the reference's
                                           filter could be run here (the product's opt-out: lvk_ekf_config.legacy_grid / LVK_GRID_REFERENCE=0) */
} lvo_ekf_config;

typedef struct lvo_ekf lvo_ekf;
lvo_ekf* lvo_ekf_create(const lvo_ekf_config* cfg);
void lvo_ekf_destroy(lvo_ekf* e);
/* the 24 IMU-intrinsic parameters T1 T2 T3 A1 A2 A3 M1 M2 (larvio.cpp:129-154; state columns 22..45 when calibrated) */
void lvo_ekf_get_imu_intrinsics(const lvo_ekf* e, double* out24);
void lvo_ekf_set_imu_intrinsics(lvo_ekf* e, const double* in24);
/* LarVio::processFeatures (larvio.cpp:363-461).  imu: the caller's buffer; *n_consumed = samples the reference would erase. */
int lvo_ekf_process(lvo_ekf* e, double ts, const lvo_feature_obs* feats, int n_feats, const lvo_imu* imu, int n_imu, int* n_consumed);
/* bypass the initializer (tests): IMU state at time t; gyro/acc = last IMU sample (m_gyro_old/m_acc_old) */
void lvo_ekf_set_state(lvo_ekf* e, double t, const double q[4], const double p[3], const double v[3], const double bg[3], const double ba[3],
                       const double gyro_old[3], const double acc_old[3]);
/* ... and the last ZUPT time as an initialiser leaves it (= the state time, larvio.cpp:384: in-state features wait 5 s) */
void lvo_ekf_set_last_zupt_time(lvo_ekf* e, double t);
/* stage-level views of two first-party formulas for the pins (tests only) */
int lvo_stage_ekf1d_obs_jacobian(const lvo_clone* k, const lvo_clone* a, const double* p_w, double inv_depth, const double* obs_anchor,
                                 const double* z, double* Hf2, double* Ha12, double* Hx12, double* He12, double* r2);
int lvo_stage_hybrid_update_with_new(double* P, int N, const double* Ho, int m, const double* ro, const double* H1, const double* H2,
                                     const double* r1, int n_acc, double sigma2, double* P_out, double* dx_out);
int lvo_stage_reanchor_row(const lvo_clone* c_old, const lvo_clone* c_new, const double* R_b2c, const double* t_c_b, const double* p_w,
                           double inv_depth_new, double* J19);
int lvo_ekf_dim(const lvo_ekf* e);                                  /* N */
int lvo_ekf_is_initialized(const lvo_ekf* e);
/* IMU state block: t, q[4], v[3], p[3], bg[3], ba[3], R_b2c[9], t_c_b[3], td  (27 doubles after t) */
void lvo_ekf_get_state(const lvo_ekf* e, double* out28);
void lvo_ekf_get_cov(const lvo_ekf* e, double* P /* N*N row-major */);
int lvo_ekf_get_clones(const lvo_ekf* e, lvo_clone* out, int cap);
/* in-state features: ids, inverse depths, world positions */
int lvo_ekf_get_features(const lvo_ekf* e, int64_t* ids, double* inv_depth, double* pos_w, int cap);
/* counters: [0] hybrid updates, [1] msckf updates, [2] rows of last H_o, [3] zupt updates, [4] features gated in, [5] gated out, [6] map size */
void lvo_ekf_counters(const lvo_ekf* e, long* out7);
/* one step of the static initialiser alone (stage-level parity with the reference's StaticInitializer compiled in place) */
int lvo_ekf_static_try_init(lvo_ekf* e, double ts, const lvo_feature_obs* f, int n, const lvo_imu* imu, int n_imu, int* n_erased, double* out8);

#ifdef __cplusplus
}
#endif
#endif
