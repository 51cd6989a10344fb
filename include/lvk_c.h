/*
 * lvk_c.h — C ABI of liblvk_hip.so: the MI355X (gfx950) implementation of LARVIO's per-frame
 * hot path (visual front-end + EKF measurement update), the drop-in boundary under the
 * reference's two C++ classes.
 *
 * The reference has no FFI layer; its boundary is larvio::ImageProcessor
 * (/root/reference/include/larvio/image_processor.h:36-68) and larvio::LarVio
 * (include/larvio/larvio.h:37-90), called from app/larvioMain.cpp:107,114.  Each entry point
 * below cites the reference function it replaces.  INTEGRATION.md shows the adapter classes a
 * maintainer compiles against this header so that larvioMain.cpp / System.cpp link unchanged.
 *
 * Conventions
 *  - plain C types only; every function returns lvk_status; no exceptions cross the ABI;
 *  - handle-scoped state, no process globals; lvk_last_error(ctx) gives a message;
 *  - pointers named d_* are DEVICE pointers (hipMalloc / torch data_ptr()); h_* are host;
 *  - stage-level functions are asynchronous on the context's stream unless stated;
 *  - there is NO CPU fallback: without a gfx950 device lvk_context_create fails.
 */
#ifndef LVK_C_H
#define LVK_C_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int lvk_status;
#define LVK_OK              0
#define LVK_ERR_ARG         1   /* bad argument */
#define LVK_ERR_DEVICE      2   /* HIP runtime error / no device */
#define LVK_ERR_CAPACITY    3   /* a fixed capacity was exceeded */
#define LVK_ERR_UNSUPPORTED 4   /* configuration outside what the kernels are built for */
#define LVK_ERR_NUMERIC     5   /* the measurement update met an innovation covariance that is not positive definite: the filter has diverged */

typedef struct lvk_context  lvk_context;
typedef struct lvk_pyramid  lvk_pyramid;
typedef struct lvk_frontend lvk_frontend;

typedef struct { float x, y; } lvk_pt2f;
/* One 8-bit grey image as the reference's ImageData carries it (include/sensors/ImageData.hpp: a cv::Mat knows its own size and
 * step).  stride = bytes between rows (cv::Mat::step); is_device: 0 = data is a host pointer (copied in through the front-end's
 * pinned staging, the caller may reuse the buffer as soon as the call returns), 1 = data is a device pointer. */
typedef struct { const uint8_t* data; int width, height, stride; int is_device; } lvk_image;
/* include/sensors/ImuData.hpp:17-43 */
typedef struct { double t; double gyro[3]; double acc[3]; } lvk_imu;
/* include/larvio/feature_msg.h:15-44 (MonoFeatureMeasurement, 72 bytes) */
typedef struct {
    uint64_t id;
    double u, v, u_init, v_init, u_vel, v_vel, u_init_vel, v_init_vel;
} lvk_feature_obs;

/* ------------------------------------------------------------------ runtime environment
 * What a deployment should set before the HIP runtime initialises in its process, shipped with the library instead of living in a
 * launcher script: GPU_MAX_HW_QUEUES=8 (one hardware queue per stream of the library and of the application; HIP's default of 4 makes
 * streams share queues: -17 % frames/s), HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory), and - opt-in - the calling
 * thread and every thread started after it bound to the physical cores of one L3 group of its socket (local_rank picks the group).
 * Variables already set by the user are respected.  Call it FIRST in main(), before any HIP call of the process; returns the flags
 * that took effect (a variable the user had set, or a call after this library has already initialised the runtime, reports nothing
 * for that flag; the environment is process-wide state: call it before other threads exist).  lvk_context_create applies LVK_RT_DEFAULT by itself (unless LVK_RUNTIME_ENV=0), which suffices when the library
 * is what makes the process's first HIP call.  No reference counterpart: the reference has no device. */
#define LVK_RT_HW_QUEUES   1u
#define LVK_RT_DEV_KERNARG 2u
#define LVK_RT_BIND_L3     4u
#define LVK_RT_DEFAULT     (LVK_RT_HW_QUEUES | LVK_RT_DEV_KERNARG)
unsigned    lvk_runtime_env(unsigned flags, int local_rank);

/* ------------------------------------------------------------------ context / memory */
lvk_status  lvk_context_create(int device, lvk_context** out);
void        lvk_context_destroy(lvk_context* ctx);
/* use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
lvk_status  lvk_context_set_stream(lvk_context* ctx, void* hip_stream);
void*       lvk_context_get_stream(const lvk_context* ctx);   /* the hipStream_t every call on this context is ordered on */
lvk_status  lvk_sync(lvk_context* ctx);
const char* lvk_last_error(const lvk_context* ctx);
const char* lvk_version(void);
/* thin helpers so a host without torch can drive the stage-level API */
lvk_status  lvk_malloc(lvk_context* ctx, size_t bytes, void** d_out);
lvk_status  lvk_free(lvk_context* ctx, void* d_ptr);
lvk_status  lvk_memcpy_h2d(lvk_context* ctx, void* d_dst, const void* h_src, size_t bytes);  /* async */
lvk_status  lvk_memcpy_d2h(lvk_context* ctx, void* h_dst, const void* d_src, size_t bytes);  /* sync  */
lvk_status  lvk_memset(lvk_context* ctx, void* d_dst, int value, size_t bytes);

/* ------------------------------------------------------------------ image passes
 * replaces createImagePyramids (image_processor.cpp:318-334):
 *   cv::createCLAHE(3.0,(8,8))->apply  +  cv::buildOpticalFlowPyramid(win, levels, derivs)
 * and ORBdescriptor::initializeLayerAndPyramid level 0 (ORBDescriptor.cpp:418-484). */
lvk_status lvk_clahe_u8(lvk_context* ctx, const uint8_t* d_src, int w, int h, int sstride,
                        uint8_t* d_dst, int dstride, double clip, int tiles_x, int tiles_y);

lvk_status lvk_pyramid_create(lvk_context* ctx, int w, int h, int win, int max_level, lvk_pyramid** out);
void       lvk_pyramid_destroy(lvk_pyramid* p);
/* level-0 source is d_img (already equalised or raw); builds all levels, padding, Scharr planes */
lvk_status lvk_pyramid_build(lvk_context* ctx, lvk_pyramid* p, const uint8_t* d_img, int stride);
/* fused CLAHE + build (what the frame-level path runs) */
lvk_status lvk_pyramid_build_clahe(lvk_context* ctx, lvk_pyramid* p, const uint8_t* d_img, int stride,
                                   double clip, int tiles_x, int tiles_y);
int        lvk_pyramid_levels(const lvk_pyramid* p);
/* geometry of level l: interior w,h; padded buffers: image row stride (bytes), deriv row stride
 * (int16 units); device base pointers of the PADDED buffers; pad = win */
lvk_status lvk_pyramid_level(const lvk_pyramid* p, int level, int* w, int* h, int* pad,
                             int* istride, int* dstride, const uint8_t** d_img, const int16_t** d_der);
/* ORB level-0 mosaic (border 32) and its 7x7 sigma-2 blurred copy; both (h+64) x (w+64), stride w+64 */
lvk_status lvk_orb_prepare(lvk_context* ctx, const lvk_pyramid* p, uint8_t* d_ext, uint8_t* d_blur);

/* replaces cv::goodFeaturesToTrack(img, maxCorners, quality, minDistance, mask, 3)
 * (image_processor.cpp:343, 1035-1036).  d_mask may be NULL; d_n_out = device int. */
lvk_status lvk_min_eigen_map(lvk_context* ctx, const lvk_pyramid* p, float* d_eig);
lvk_status lvk_good_features(lvk_context* ctx, const lvk_pyramid* p, const uint8_t* d_mask,
                             int max_corners, double quality, double min_distance,
                             lvk_pt2f* d_out, int cap, int* d_n_out);

/* ------------------------------------------------------------------ per-point stages
 * replaces cv::calcOpticalFlowPyrLK(..., OPTFLOW_USE_INITIAL_FLOW) at
 * image_processor.cpp:368,405,558,618,830,870.  d_next_pts in/out; d_iters optional (n*levels). */
lvk_status lvk_lk_track(lvk_context* ctx, const lvk_pyramid* prev, const lvk_pyramid* next,
                        const lvk_pt2f* d_prev_pts, lvk_pt2f* d_next_pts, uint8_t* d_status, int n,
                        int max_iter, double eps, int* d_iters);
/* replaces ORBdescriptor::computeDescriptors (ORBDescriptor.cpp:386-416); d_desc n*32 bytes */
lvk_status lvk_orb_describe(lvk_context* ctx, const uint8_t* d_ext, const uint8_t* d_blur, int w, int h,
                            const lvk_pt2f* d_pts, int n, uint8_t* d_desc, float* d_angle);
/* replaces ORBdescriptor::computeDescriptorDistance row-wise (ORBDescriptor.h:43-59) */
lvk_status lvk_hamming256_rows(lvk_context* ctx, const uint8_t* d_a, const uint8_t* d_b, int n, int* d_dist);
/* replaces ImageProcessor::undistortPoints (image_processor.cpp:1040-1072); model 0 radtan, 1 equidistant */
lvk_status lvk_undistort_points(lvk_context* ctx, const lvk_pt2f* d_in, int n, const double intr[4], int model,
                                const double dist[4], const double new_intr[4], lvk_pt2f* d_out);
/* replaces cv::findFundamentalMat(p1,p2,FM_RANSAC,thresh,conf,mask) (image_processor.cpp:498,755,968).
 * d_mask n bytes; d_info[0] = 1 if a mask was written (n>=7) else 0, d_info[1] = hypotheses drawn. */
lvk_status lvk_find_fundamental_mask(lvk_context* ctx, const lvk_pt2f* d_p1, const lvk_pt2f* d_p2, int n,
                                     double thresh, double conf, uint8_t* d_mask, int* d_info);
/* the same call where the reference also uses the MATRIX it returns (solve_5pts.cpp:206, the moving-start initialiser's relative pose):
 * d_F (9 doubles, row-major) = the registrator's best minimal-sample model - cv::findFundamentalMat does not refit on the inliers -,
 * all zeros when it has none (OpenCV's empty Mat); n == 7: the first of the solver's up to three models. */
lvk_status lvk_find_fundamental(lvk_context* ctx, const lvk_pt2f* d_p1, const lvk_pt2f* d_p2, int n,
                                double thresh, double conf, uint8_t* d_mask, int* d_info, double* d_F);
/* the RANSAC branch alone for any n >= 8 (stage parity) */
lvk_status lvk_ransac_fundamental(lvk_context* ctx, const lvk_pt2f* d_p1, const lvk_pt2f* d_p2, int n,
                                  double thresh, double conf, int max_iters, uint8_t* d_mask, int* d_info);
/* replaces integrateImuData + predictFeatureTracking's matrix (image_processor.cpp:222-293): host math */
lvk_status lvk_predict_homography(const lvk_imu* h_imu, int n_imu, double t_prev, double t_curr,
                                  const double R_cam_imu[9], const double intr[4], float H[9]);

/* ------------------------------------------------------------------ the front-end object
 * replaces larvio::ImageProcessor (image_processor.h:36-68): lvk_frontend_create = ctor+initialize()
 * (image_processor.cpp:28-126, parameters as loadParameters reads them), lvk_frontend_process =
 * processImage (image_processor.cpp:130-219). */
typedef struct {
    int width, height;
    int pyramid_levels;      /* config/euroc.yaml:43 */
    int patch_size;          /* :44 */
    int max_iteration;       /* :46 */
    double track_precision;  /* :47 */
    int max_features_num;    /* :49 */
    int min_distance;        /* :50 */
    int flag_equalize;       /* :51 */
    int pub_frequency;       /* :52 */
    int distortion_model;    /* 0 radtan, 1 equidistant (:14) */
    double intrinsics[4];    /* fx fy cx cy (:17-21) */
    double distortion[4];    /* (:22-26) */
    double R_cam_imu[9];     /* row-major, = (T_cam_imu rotation)^T as image_processor.cpp:93 */
} lvk_fe_config;

lvk_status lvk_frontend_create(lvk_context* ctx, const lvk_fe_config* cfg, lvk_frontend** out);
void       lvk_frontend_destroy(lvk_frontend* fe);
/* img->width/height must equal the configured resolution and img->stride >= width (LVK_ERR_ARG otherwise: the reference cannot
 * read past a cv::Mat, neither does this).  h_out receives up to cap features when *has_msg = 1 (the reference's
 * `return haveFeatures`). */
lvk_status lvk_frontend_process(lvk_frontend* fe, const lvk_image* img,
                                double ts, const lvk_imu* h_imu, int n_imu,
                                lvk_feature_obs* h_out, int cap, int* n_out, int* has_msg);
/* introspection (synchronises): live tracks after the last call = the reference's
 * pts_ids_/prev_pts_/pts_lifetime_/init_pts_/vOrbDescriptors after the rotation at :207-216 */
lvk_status lvk_frontend_tracks(lvk_frontend* fe, uint64_t* h_ids, lvk_pt2f* h_pts, int* h_lifetime,
                               lvk_pt2f* h_init, uint8_t* h_desc, int cap, int* n_out);
lvk_status lvk_frontend_new_pts(lvk_frontend* fe, lvk_pt2f* h_pts, int cap, int* n_out);
int        lvk_frontend_state(const lvk_frontend* fe);   /* 1 FIRST_IMAGE 2 SECOND_IMAGE 3 OTHER_IMAGES */
/* cumulative LK work: point-levels processed and iterations executed (SURVEY §8d byte model) */
lvk_status lvk_frontend_lk_stats(lvk_frontend* fe, uint64_t* point_levels, uint64_t* iterations);
/* feature messages published so far (getFeatureMsg, image_processor.cpp:1076-1128) and the features they carried in total:
   features / messages = the tracks the tracker holds per published frame (the "~150 tracks" of BASELINE.json's metric) */
lvk_status lvk_frontend_msg_stats(lvk_frontend* fe, uint64_t* messages, uint64_t* features);

/* Per-stage GPU time measured with HIP events on the context's stream (the reference only times the
 * whole call, app/larvioMain.cpp:106-109).  stage_mask bit i enables stage i; reading synchronises.
 * Stages: 0 pyramid(+CLAHE) 1 orb_prepare 2 lk_fwd 3 lk_rev 4 orb_gate 5 ransac_commit 6 min_eigen
 * 7 gftt_select(mask,max,candidates,select) 8 feature_msg.  LK/ORB/RANSAC stages sum old+new launches.
 * Bits 16..23 of stage_mask: bracket every n-th frame only (0 or 1: every frame) - an event record is a barrier packet on the frame's
 * dependent chain, so a measurement that must not slow what it measures samples (bench.py: every 5th frame). */
#define LVK_FE_STAGES 9
lvk_status lvk_frontend_profile_enable(lvk_frontend* fe, unsigned stage_mask);
lvk_status lvk_frontend_profile_read(lvk_frontend* fe, double ms_sum[LVK_FE_STAGES], uint64_t launches[LVK_FE_STAGES], int reset);
const char* lvk_frontend_stage_name(int stage);

/* ==================================================================== back-end: EKF measurement update
 * replaces larvio::LarVio (include/larvio/larvio.h:37-90).  feature_idp_dim = 1, use_schmidt = 0,
 * calib_imu_instrinsic = 0 or 1 (config/euroc.yaml:8-10,105,108) are implemented; other values are refused.
 * Matrices crossing the ABI are ROW-MAJOR doubles with an explicit leading dimension (P is symmetric, so the
 * reference's column-major Eigen::MatrixXd state_cov maps onto it unchanged). */
typedef struct lvk_ekf lvk_ekf;

/* stage level (device pointers): the dense algebra of measurementUpdate_msckf / _hybrid (larvio.cpp:1430-1460,1578-1594) */
/* [H | r] (rows x cols) -> top `cols` rows of Q^T [H | r], in place (SPQR replacement, larvio.cpp:1430-1445). */
lvk_status lvk_ekf_compress_qr(lvk_context* ctx, double* d_H, int ld, int rows, int cols, double* d_r, int* rows_out);
/* The same compression when the caller knows the block structure of H (as the filter does: a feature's rows touch the extrinsics / td
 * columns and the 6-column blocks of the clones that observed it - the sparsity SPQR exploits in the reference).  The rows come in
 * n_groups consecutive groups; group i has h_rows[i] rows that are zero outside the ascending columns
 * h_cols[h_col_off[i] .. h_col_off[i+1]).  A TSQR tree over consecutive groups whose column union fits one workgroup's LDS is planned on
 * the host and run level by level (work ~ 2 r c^2 with c = columns per node instead of 2 r cols^2).  Result in place at the top of
 * d_H / d_r, *rows_out rows (>= cols possible: finish with lvk_ekf_compress_qr).  lvk_ekf_qr_plan is the host-only planning step
 * (no device needed): blocks as 8 ints {in_start, in_rows, out_start, out_rows, ncols, col_off, copy, 0} level after level. */
lvk_status lvk_ekf_compress_qr_groups(lvk_context* ctx, double* d_H, int ld, int rows, int cols, double* d_r, int n_groups,
                                      const int* h_rows, const int* h_col_off, const int* h_cols, int* rows_out);
int        lvk_ekf_qr_plan(int N, int n_groups, const int* h_rows, const int* h_col_off, const int* h_cols, int* h_blocks, int cap_blocks,
                           int* h_block_cols, int cap_cols, int* h_level_blocks, int* h_level_cols, int cap_levels, int* final_rows);
/* S = H P H^T + sigma2 I ; dx = P H^T S^-1 r ; P <- (I - K H) P symmetrised.  H is m x n (ldh), P n x n (ldp).
 * Replaces larvio.cpp:1453-1460, 1578-1594.  Waits for the update: LVK_ERR_NUMERIC when S is not positive definite (d_P is then
 * whatever the factorisation with that pivot replaced by 1 leaves; the reference's pivoted LDLT goes on silently). */
lvk_status lvk_ekf_update(lvk_context* ctx, double* d_P, int ldp, int n, const double* d_H, int ldh, int m,
                          const double* d_r, double sigma2, double* d_dx);
/* C = alpha op(A) op(B) + beta C on the FP64 matrix cores (v_mfma_f64_16x16x4_f64) — the P H^T-class contraction */
lvk_status lvk_dgemm(lvk_context* ctx, int transa, int transb, int M, int N, int K, double alpha, const double* d_A, int lda,
                     const double* d_B, int ldb, double beta, double* d_C, int ldc);

typedef struct {
    /* names and meaning as LarVio::loadParameters reads them (larvio.cpp:58-311, config/euroc.yaml) */
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
    int feature_idp_dim, use_schmidt, calib_imu_instrinsic;   /* must be 1, 0, 0|1 (1: LEG_DIM 46, IMU intrinsics in the state) */
    int max_features;                                          /* capacity hint: features per message (0 = 1024) */
    int legacy_grid;                                           /* 0 (default): grid_map as the reference keeps it - a std::map (larvio.h:383), so a grid code
                                                                  beyond the rows x cols cells (undistorted coordinates outside the image bounds) gets a cell
                                                                  of its own that updateGridMap never clears (larvio.cpp:1969-1975, 3351-3370).
                                                                  1: such codes are not counted (what this library did before round 6; opt-out only,
                                                                  also LVK_GRID_REFERENCE=0 in the environment) */
    int reserved0;                                             /* must be 0 */
} lvk_ekf_config;

/* one sliding-window clone (IMUState_Aug, include/larvio/imu_state.h:72-117) */
typedef struct {
    int64_t id;
    double time, dt;
    double q[4], p[3], p_fej[3];
    double R_b2c[9], t_c_b[3];
    double q_cam[4], p_cam[3];
} lvk_clone;

/* ctor + initialize() (larvio.cpp:40-360) */
lvk_status lvk_ekf_create(lvk_context* ctx, const lvk_ekf_config* cfg, lvk_ekf** out);
void       lvk_ekf_destroy(lvk_ekf* e);
/* LarVio::processFeatures (larvio.cpp:363-461).  h_imu is the caller's buffer; *n_consumed = how many leading samples the
 * reference would erase from it (larvio.cpp:511-512, StaticInitializer.cpp:146-147); *updated = the bool it returns.
 * The call returns when the HOST state (lvk_ekf_get_state, clones, features) is final; the covariance's last launches (the pruning's
 * gather) may still be queued on the context's stream - every entry point that reads the covariance synchronises first, and a device
 * fault in those tail launches is reported by the next call on the handle. */
lvk_status lvk_ekf_process(lvk_ekf* e, double ts, const lvk_feature_obs* h_feats, int n_feats,
                           const lvk_imu* h_imu, int n_imu, int* n_consumed, int* updated);
/* The same update, deferred: the call returns as soon as what the caller needs from processFeatures is known - *n_consumed (the
 * samples to erase from the driver's buffer: a function of time stamps, the state time and td only) and *will_update (the bool
 * processFeatures will return: always true once the filter is initialized; the cold paths - first IMU sample, static initializer,
 * a failed filter - run synchronously inside this call and report their real result) - and the update itself runs on a worker
 * thread of the filter, on the filter's stream.  The buffers are copied before the call returns.  Every other entry point of the
 * filter (all getters, the next update, set_state, destroy) first waits for the queued update, so callers always see the state
 * AFTER it: results are identical to lvk_ekf_process.  This is what lets the reference's blocking drivers
 * (app/larvioMain.cpp:104-116: processImage; processFeatures; getters) overlap the next frame's front-end with this frame's update.
 * lvk_ekf_wait blocks until the queued update is done and returns ITS status (*updated as lvk_ekf_process would have set it). */
lvk_status lvk_ekf_process_async(lvk_ekf* e, double ts, const lvk_feature_obs* h_feats, int n_feats,
                                 const lvk_imu* h_imu, int n_imu, int* n_consumed, int* will_update);
lvk_status lvk_ekf_wait(lvk_ekf* e, int* updated);
/* bypass the initializer (tests, benchmarks): IMU state at time t, last IMU sample (m_gyro_old / m_acc_old) */
lvk_status lvk_ekf_set_state(lvk_ekf* e, double t, const double q[4], const double p[3], const double v[3],
                             const double bg[3], const double ba[3], const double gyro_old[3], const double acc_old[3]);
int        lvk_ekf_dim(const lvk_ekf* e);                      /* state_cov.rows() */
int        lvk_ekf_is_initialized(const lvk_ekf* e);
/* take_off_stamp (larvio.cpp:380): the state time at which the initializer succeeded; the state log's time origin (:446) */
double     lvk_ekf_take_off_stamp(const lvk_ekf* e);
/* 30 doubles: t, q[4] (x y z w), v[3], p[3], bg[3], ba[3], R_imu_cam0[9], t_cam0_imu[3], td  (getTbw/getVel, larvio.cpp:2644-2700) */
lvk_status lvk_ekf_get_state(const lvk_ekf* e, double* h_out30);
/* the 24 IMU-intrinsic parameters T1 T2 T3 A1 A2 A3 M1 M2 (larvio.cpp:129-154; state columns 22..45 when calibrated) */
lvk_status lvk_ekf_get_imu_intrinsics(const lvk_ekf* e, double* h_out24);
lvk_status lvk_ekf_set_imu_intrinsics(lvk_ekf* e, const double* h_in24);
lvk_status lvk_ekf_get_cov(lvk_ekf* e, double* h_P);           /* N*N row-major, synchronises (getPpose/getPvel read blocks of it) */
/* the leading n x n block of the covariance (n <= 16: orientation 0..2, velocity 3..5, position 6..8, gyro bias 9..11, ...), row-major -
 * all that getPpose / getPvel read (larvio.cpp:2673-2690); served from a host-side mirror the update keeps, no transfer of the matrix */
lvk_status lvk_ekf_get_cov_imu(lvk_ekf* e, int n, double* h_out);
int        lvk_ekf_get_clones(const lvk_ekf* e, lvk_clone* h_out, int cap);       /* getSwPoses */
int        lvk_ekf_get_features(const lvk_ekf* e, int64_t* h_ids, double* h_inv_depth, double* h_pos_w, int cap);  /* getActiveeMapPointPositions */
/* getStableMapPointPositions (larvio.cpp:2717-2722): in-state features that were lost since the last call, with their last world
 * position; the entries handed out are removed, as the reference clears lost_slam_features on read.  Returns the count (<= cap). */
int        lvk_ekf_take_lost_features(lvk_ekf* e, int64_t* h_ids, double* h_pos_w, int cap);
/* What the moving-start initialiser (FlexibleInitializer.cpp:11-25 -> DynamicInitializer.cpp) handed to the filter, with the intermediate
 * results of the successful attempt - for parity tests against an independent restatement fed the same messages:
 *   valid        1 once the dynamic initialiser has succeeded on this handle (0: never ran, or the static one fired)
 *   message      0-based index of the lvk_ekf_process call (counted from the first one the initialisers saw) that succeeded
 *   attempts     relative-pose attempts made up to and including the successful one (= calls of the RANSAC stage's window scan that found a frame)
 *   ransac_calls launches of the RANSAC kernel (cv::findFundamentalMat, solve_5pts.cpp:206) over all attempts
 *   l, rel_R/rel_T   relativePose's frame and pose (DynamicInitializer.cpp:331-360), n_points = landmarks the SfM triangulated (initial_sfm.cpp)
 *   sfm_R/sfm_T  the window's 11 structure-from-motion poses (camera-to-c0 rotation row-major, position)
 *   bg, g, scale solveGyroscopeBias / LinearAlignment + RefineGravity (initial_alignment.cpp:74-122 ...)
 *   state_time, q (x y z w), v, erase     the state the filter starts from and the IMU samples erased */
typedef struct lvk_init_report {
    int valid, message, attempts, ransac_calls, l, n_points, erase, pad;
    double state_time, scale, rel_R[9], rel_T[3], sfm_R[11 * 9], sfm_T[11 * 3], bg[3], g[3], q[4], v[3];
} lvk_init_report;
lvk_status lvk_ekf_init_report(const lvk_ekf* e, lvk_init_report* h_out);
/* [0] hybrid updates [1] msckf updates [2] rows of the last update [3] zupt updates [4] gated in [5] gated out [6] map size [7] triangulations */
void       lvk_ekf_counters(const lvk_ekf* e, long* h_out8);
/* HIP-event bracket around the H P GEMM (the P H^T contraction, FP64 MFMA) of every update: enable/disable; h_out3 (optional) receives
 * [milliseconds, flops = sum 2 m N^2, launches] accumulated since the previous call */
lvk_status lvk_ekf_profile(lvk_ekf* e, int enable, double* h_out3);
/* the same switch brackets every level of the structure-aware TSQR compression (k_qr_sparse; replaces the SPQR calls at
 * larvio.cpp:1430-1445, 2209-2229): h_out4 = [milliseconds, Householder flops on the structure actually factored (sum over nodes
 * and reflectors j of 4 (r - j)(c + 1 - j)), launches, rows entering the levels] since the previous call */
lvk_status lvk_ekf_profile_qr(lvk_ekf* e, double* h_out4);

/* ---- sharded measurement update (BASELINE.json configs[4]).  Every rank runs the same filter on the same messages; the
 * per-feature device work of an update (Jacobian rows, null-space projection, chi-square gate, stacking, first compression) is split
 * into `world` contiguous feature ranges in the reference's stacking order (larvio.cpp:2185-2201), rank `rank` doing its own; one
 * all-gather per update hands every rank all compressed blocks and all gate results, in rank order, and the rest of the update is
 * replicated - identical bits on every rank.  The collective is a callback so that the transport is the caller's choice:
 * lvk_shard_allgather_rccl (below) runs ncclAllGather on the filter's stream, device buffers in place; a test may move the bytes
 * through the host.  fn must deliver, in d_recv, the `world` send buffers of bytes_per_rank bytes each in rank order, ordered on
 * hip_stream.  fn = NULL (world 1) switches sharding off; with a transport the sharded path runs at any world size, world 1 included
 * (a loop-back through pack -> all-gather -> unpack -> second stage: validates a transport on one GPU).  The exchange buffers are
 * allocated by this call.  Every capacity test of the sharded update has the same outcome on all ranks (each rank plans every rank's
 * share), so a capacity error is raised everywhere before anybody enters the collective; a rank that fails locally (a launch error
 * while queueing its rows, or later) still enters it, with a poisoned block header, and its peers return LVK_ERR_DEVICE at their
 * next sync instead of waiting.  One case cannot post a block - an update that admits new in-state features sizes its exchange from
 * gate results the failing rank could not read - and calls fn(user, NULL, NULL, 0, stream) instead: bytes_per_rank == 0 means
 * ABORT, the transport must make the peers' pending exchange fail (lvk_shard_allgather_rccl: ncclCommAbort). */
typedef int /* lvk_status */ (*lvk_exchange_fn)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);
lvk_status lvk_ekf_set_shard(lvk_ekf* e, int rank, int world, lvk_exchange_fn fn, void* user);
/* [0] exchanges [1] bytes sent by this rank [2] sharded updates [3] rows this rank stacked; [4] updates the structure-aware
 * compression ran in [5] its levels [6] rows in [7] rows out */
void       lvk_ekf_shard_stats(const lvk_ekf* e, long* h_out8);
/* RCCL transport: one communicator per rank (ncclCommInitRank with the 128-byte id rank 0 obtained from lvk_shard_unique_id and
 * distributed out of band, e.g. torch.distributed.broadcast); pass lvk_shard_allgather_rccl + the communicator to lvk_ekf_set_shard. */
typedef struct lvk_shard_comm lvk_shard_comm;
lvk_status lvk_shard_unique_id(char* h_out128);
lvk_status lvk_shard_comm_create(lvk_context* ctx, const char* h_uid128, int rank, int world, lvk_shard_comm** out);
void       lvk_shard_comm_destroy(lvk_shard_comm* c);
const char* lvk_shard_rccl_path(void);                        /* which librccl the transport bound (the one next to the HIP runtime in use) */
const char* lvk_shard_comm_error(const lvk_shard_comm* c);    /* text of the last RCCL error lvk_shard_allgather_rccl returned on it */
lvk_status lvk_shard_allgather_rccl(void* comm, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);

/* ---- per-feature stages of the update, one call each (parity tests; callers that want a single stage).  Host buffers.
 * lvk_triangulate: Feature::initializePosition (use_position 0) / the LM refinement from a given position (1), feature.hpp:383-890,
 *   n views = camera-to-world poses + normalised observations; *ok_out = the reference's return value.
 * lvk_ekf_gate_and_stack: for a batch of MSCKF features, featureJacobian_msckf (larvio.cpp:924-981: H_x blocks, null-space projection),
 *   gatingTest (:1865-1880) against P, and the stacking of the accepted features' rows (:2185-2201) into H (rows x N, row-major,
 *   ld = N) and r.  h_clone_rank / h_obs / h_obs_vel are indexed by obs_off + k.  gamma / accept are per feature. */
typedef struct { double R[9]; double t[3]; } lvk_cam_pose;
typedef struct { double p_w[3]; int n_obs, obs_off; } lvk_msckf_feature;
lvk_status lvk_triangulate(lvk_context* ctx, const lvk_cam_pose* h_poses, const double* h_obs, int n, int use_position,
                           const double* h_position_in, int* ok_out, double* h_position, double* h_solution, double* h_inv_depth,
                           double* h_obs_anchor);
lvk_status lvk_ekf_gate_and_stack(lvk_context* ctx, const lvk_clone* h_clones, int n_clones, const lvk_msckf_feature* h_feats, int n_feats,
                                  const int* h_clone_rank, const double* h_obs, const double* h_obs_vel, const double* h_P, int N,
                                  int if_fej, int estimate_td, double sigma2, double* h_H, double* h_r, int rows_cap, int* rows_out,
                                  double* h_gamma, int* h_accept);

/* ==================================================================== the driver step
 * One camera frame through both halves, exactly the two calls the reference's drivers make per image
 * (app/larvioMain.cpp:104-116: processImage, then processFeatures when it returned true), with the driver's IMU buffer
 * semantics (samples with t < t_img + 0.05 are visible, processFeatures erases what it consumed, :98-102 and larvio.cpp:511-512).
 * h_imu[0..n_imu) is the CURRENT buffer; *n_consumed tells the caller how many leading samples to drop. */
lvk_status lvk_vio_process(lvk_frontend* fe, lvk_ekf* ekf, const lvk_image* img, double ts,
                           const lvk_imu* h_imu, int n_imu, int* n_consumed, int* has_msg, int* updated);
/* the same step with the update deferred (lvk_frontend_process, then lvk_ekf_process_async): returns when the front-end is done and
 * the update is queued; any getter of the filter (or the next step) waits for it.  This is the schedule the adapter classes
 * (adapter/: larvio::ImageProcessor / larvio::LarVio) give an unchanged blocking driver. */
lvk_status lvk_vio_process_deferred(lvk_frontend* fe, lvk_ekf* ekf, const lvk_image* img, double ts,
                                    const lvk_imu* h_imu, int n_imu, int* n_consumed, int* has_msg, int* will_update);

/* The same loop, pipelined across two HIP streams: the filter update of frame k (worker thread, ekf's context) overlaps the
 * front-end of frame k+1 (caller's thread, fe's context).  fe and ekf must have been created on DIFFERENT lvk_contexts.
 * The pipe owns the driver's IMU vector: push samples as larvioMain.cpp:98-102 does (t < t_img + 0.05) before each submit;
 * erasure happens inside, in the order the sequential loop would do it, so every result is identical to lvk_vio_process.
 * push/submit/drain are called from one thread.  Read the filter (lvk_ekf_get_*) only after lvk_vio_pipe_drain. */
typedef struct lvk_vio_pipe lvk_vio_pipe;
lvk_status lvk_vio_pipe_create(lvk_frontend* fe, lvk_ekf* ekf, lvk_vio_pipe** out);
void       lvk_vio_pipe_destroy(lvk_vio_pipe* p);
lvk_status lvk_vio_pipe_push_imu(lvk_vio_pipe* p, const lvk_imu* h_imu, int n);
lvk_status lvk_vio_pipe_submit(lvk_vio_pipe* p, const lvk_image* img, double ts, int* has_msg);
lvk_status lvk_vio_pipe_drain(lvk_vio_pipe* p, long* n_updates, long* n_msgs);
/* Called on the filter's thread after every update that processFeatures would have answered with true — the point at which the
 * reference's drivers publish odometry (app/larvioMain.cpp:117-, ros_wrapper System.cpp:177-193).  ts = the message's stamp,
 * state30 as lvk_ekf_get_state.  The filter is quiescent for the duration of the call: lvk_ekf_get_* are allowed inside it.
 * Set before the first submit (or after a drain); fn = NULL removes it. */
typedef void (*lvk_odometry_fn)(void* user, double ts, const double* state30);
lvk_status lvk_vio_pipe_on_update(lvk_vio_pipe* p, lvk_odometry_fn fn, void* user);
/* host wall time in microseconds since the last reset: [0] caller thread inside the front-end, [1] caller waiting for an erase
 * count, [2] worker inside filter updates, [3] worker waiting for a message */
lvk_status lvk_vio_pipe_stats(lvk_vio_pipe* p, double* h_out4, int reset);
/* image-in -> state-out latency (microseconds of host wall time, entry of lvk_vio_pipe_submit to the end of the update it triggered)
 * of every frame that produced a feature message since the last reset, in order; *n_out <= cap entries are written.  For a VIO this
 * is the latency that matters (the reference's timing window app/larvioMain.cpp:106-116 spans both calls). */
lvk_status lvk_vio_pipe_latency(lvk_vio_pipe* p, float* h_out_us, int cap, int* n_out, int reset);
/* Erase counts lvk_vio_pipe_submit took EARLY (from the last published camera-IMU time offset, because no IMU sample lies within
 * LVK_PIPE_TD_MARGIN [0.5 ms] of the bound the count depends on - the caller's thread then does not wait for the running update), and
 * how many of them the filter's thread found different when the update started (expected: 0; the reference's sequential schedule
 * - larvio.cpp:464-517 erasing before the next processImage reads the vector - is reproduced exactly whenever it is 0).
 * LVK_PIPE_EARLY_COUNT=0 switches the early count off (every frame after a message frame then waits for the running update). */
lvk_status lvk_vio_pipe_early_counts(lvk_vio_pipe* p, long* n_early, long* n_wrong);

#ifdef __cplusplus
}
#endif
#endif
