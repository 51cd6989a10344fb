/*
 * be_filter.c — ORACLE (test infrastructure, see lvo.h): the LarVio estimator state machine, a plain-C
 * restatement of /root/reference/src/larvio.cpp: processFeatures :363-461, batchImuProcessing :464-517,
 * processModel :520-578, predictNewState :581-649, stateAugmentation :720-801, addFeatureObservations :804-856,
 * featureJacobian_ekf_new/_ekf :1247-1417, measurementUpdate_msckf :1420-1602, measurementUpdate_hybrid
 * :1605-1862, removeLostFeatures :1883-2256, findRedundantImuStates :2259-2307, pruneImuStateBuffer :2310-2641,
 * checkZUPT / measurementUpdate_ZUPT_vpq :2751-2962, updateFeatureCov_1didp :3125-3293, rmLostFeaturesCov
 * :3296-3348, updateGridMap :3351-3370, getNewAnchorId :3412-3472, calPhi :3475-3530 (calib_imu = 0 part) and
 * src/StaticInitializer.cpp:12-163.  feature_idp_dim = 1, use_schmidt = 0, calib_imu = 0 (LEG_DIM 22).
 * PINNED to the reference compiled in place (lvo.h, "PINNING"): LarVio::processFeatures itself, state / covariance /
 * clones / map after every call on simulated and tracker-made streams, 330 random configurations, a fixture written
 * by the reference (tests/test_oracle_ref_larvio.py), and the reference's whole program on files
 * (tests/test_oracle_ref_main.py).  Two deviations are the reference's own (PARITY.md section 2): sw_size 5, and the
 * grid cells of features beyond the image bounds (reference_grid = 0 keeps the older bookkeeping, 1 follows the reference).
 */
#include "lvo.h"
#include "be_math.h"
#include <stdlib.h>
#include <stdio.h>
#include <float.h>

#define LEG (e->leg)          /* LEG_DIM: 22, or 46 with IMU-intrinsics calibration (larvio.cpp:158-161) */
#define LEG_MAX 46
#define MAX_OBS 192
#define GRAV 9.81

typedef struct {
    int64_t id;
    int n_obs;
    int64_t sid[MAX_OBS];
    double z[MAX_OBS][2], zv[MAX_OBS][2];
    double position[3], position_fej[3];
    int is_initialized;
    int64_t id_anchor;
    double inv_depth, obs_anchor[3];
    int in_state, total_obs, ekf_feature;
} feat_t;

typedef struct { double t; double q[4], p[3], v[3], bg[3], ba[3]; } imu_state_t;

struct lvo_ekf {
    lvo_ekf_config cfg;
    /* state_server */
    int64_t imu_id; double imu_dt;
    imu_state_t s, s_old, s_fej_now, s_fej_old;
    double R_b2c[9], t_c_b[3], td;
    int leg;                                /* LEG_DIM */
    double imx[24];                         /* T1 T2 T3 A1 A2 A3 M1 M2 (larvio.cpp:129-154) */
    double Tg[9], As[9], Ma[9];             /* gyro misalignment/scale, g-sensitivity, accel misalignment/scale (updateImuMx :3803-3846) */
    lvo_clone* clones; int n_clones, cap_clones;
    int64_t* feature_states; int n_fs, cap_fs;
    double* P; int N;
    double Qc[12];                         /* diagonal of continuous_noise_cov */
    feat_t** map; int n_map, cap_map;      /* map_server, ascending id */
    int64_t next_state_id;
    int is_gravity_set, b_first_features, if_fej, if_zupt;
    double m_gyro_old[3], m_acc_old[3];
    double take_off_stamp, last_update_time, last_zupt_time, tracking_rate;
    double sigma2, zupt_v2, zupt_p2, zupt_q2, imu_img_time_th;
    double x_min, y_min, grid_w, grid_h;
    int* grid_count;                       /* grid_map sizes */
    int n_phantom, phantom_code[512], phantom_count[512];   /* reference_grid: the cells std::map makes for out-of-range codes (never cleared) */
    double* coarse_dis; int n_coarse, cap_coarse;
    /* static initializer */
    int static_counter, static_num; double lower_time_bound;
    int64_t* init_ids; double* init_uv; int n_init;
    long counters[7];
    FILE* trace;                           /* LVO_TRACE=<file>: decision trace for the pins in tests/test_oracle_decisions.py */
};
#define TR(e, ...) do { if ((e)->trace) fprintf((e)->trace, __VA_ARGS__); } while (0)

/* ------------------------------------------------------------------------ small utilities */
static feat_t* map_find(lvo_ekf* e, int64_t id)
{
    int lo = 0, hi = e->n_map - 1;
    while (lo <= hi) { int mid = (lo + hi) / 2; if (e->map[mid]->id == id) return e->map[mid]; if (e->map[mid]->id < id) lo = mid + 1; else hi = mid - 1; }
    return NULL;
}
static feat_t* map_insert(lvo_ekf* e, int64_t id)
{
    if (e->n_map == e->cap_map) { e->cap_map = e->cap_map ? 2 * e->cap_map : 256; e->map = (feat_t**)realloc(e->map, sizeof(feat_t*) * (size_t)e->cap_map); }
    int pos = e->n_map;
    while (pos > 0 && e->map[pos - 1]->id > id) { e->map[pos] = e->map[pos - 1]; --pos; }
    feat_t* f = (feat_t*)calloc(1, sizeof(feat_t));
    f->id = id; f->id_anchor = -1;
    e->map[pos] = f; e->n_map++;
    return f;
}
static void map_erase(lvo_ekf* e, int64_t id)
{
    for (int i = 0; i < e->n_map; ++i) if (e->map[i]->id == id) {
        free(e->map[i]);
        memmove(e->map + i, e->map + i + 1, sizeof(feat_t*) * (size_t)(e->n_map - i - 1));
        e->n_map--; return;
    }
}
static int feat_obs_find(const feat_t* f, int64_t sid) { for (int i = 0; i < f->n_obs; ++i) if (f->sid[i] == sid) return i; return -1; }
static void feat_obs_set(feat_t* f, int64_t sid, double u, double v, double uv, double vv)
{   /* std::map operator[]: insert sorted or overwrite */
    int i = feat_obs_find(f, sid);
    if (i < 0) {
        if (f->n_obs >= MAX_OBS) return;
        i = f->n_obs++;
        while (i > 0 && f->sid[i - 1] > sid) { f->sid[i] = f->sid[i - 1]; memcpy(f->z[i], f->z[i - 1], 16); memcpy(f->zv[i], f->zv[i - 1], 16); --i; }
        f->sid[i] = sid;
    }
    f->z[i][0] = u; f->z[i][1] = v; f->zv[i][0] = uv; f->zv[i][1] = vv;
}
static void feat_obs_erase(feat_t* f, int64_t sid)
{
    int i = feat_obs_find(f, sid);
    if (i < 0) return;
    for (int k = i; k + 1 < f->n_obs; ++k) { f->sid[k] = f->sid[k + 1]; memcpy(f->z[k], f->z[k + 1], 16); memcpy(f->zv[k], f->zv[k + 1], 16); }
    f->n_obs--;
}
static int clone_rank(const lvo_ekf* e, int64_t id) { for (int i = 0; i < e->n_clones; ++i) if (e->clones[i].id == id) return i; return -1; }
static int fs_rank(const lvo_ekf* e, int64_t id) { for (int i = 0; i < e->n_fs; ++i) if (e->feature_states[i] == id) return i; return -1; }

static void P_symmetrize(double* P, int N)
{
    for (int a = 0; a < N; ++a) for (int b = a + 1; b < N; ++b) {
        double s = (P[(size_t)a * N + b] + P[(size_t)b * N + a]) / 2.0;
        P[(size_t)a * N + b] = P[(size_t)b * N + a] = s;
    }
}
/* P_new[a][b] = P[map[a]][map[b]] */
static void P_gather(lvo_ekf* e, const int* idx, int newN)
{
    double* Q = (double*)malloc(sizeof(double) * (size_t)newN * newN);
    for (int a = 0; a < newN; ++a) for (int b = 0; b < newN; ++b) Q[(size_t)a * newN + b] = e->P[(size_t)idx[a] * e->N + idx[b]];
    free(e->P); e->P = Q; e->N = newN;
}
static void P_delete(lvo_ekf* e, int start, int len)
{
    int* idx = (int*)malloc(sizeof(int) * (size_t)e->N);
    int k = 0;
    for (int i = 0; i < e->N; ++i) if (i < start || i >= start + len) idx[k++] = i;
    P_gather(e, idx, k);
    free(idx);
}
static void clone_cam_pose(const lvo_clone* c, lvo_pose* o) { quat_to_rot(c->q_cam, o->R); o->t[0] = c->p_cam[0]; o->t[1] = c->p_cam[1]; o->t[2] = c->p_cam[2]; }

/* refresh orientation_cam / position_cam of a clone from the CURRENT extrinsics (larvio.cpp:1529-1541) */
static void clone_refresh_cam(const lvo_ekf* e, lvo_clone* c)
{
    double R_c2b[9], R_b2w[9], R_c2w[9], t[3];
    m3_t(e->R_b2c, R_c2b); quat_to_rot(c->q, R_b2w); m3_mul(R_b2w, R_c2b, R_c2w);
    rot_to_quat(R_c2w, c->q_cam);
    m3_v(R_b2w, e->t_c_b, t);
    c->p_cam[0] = c->p[0] + t[0]; c->p_cam[1] = c->p[1] + t[1]; c->p_cam[2] = c->p[2] + t[2];
}

/* ------------------------------------------------------------------------ create / config */
static void update_imu_mx(lvo_ekf* e)
{   /* larvio.cpp:3803-3846 */
    const double *T1 = e->imx, *T2 = e->imx + 3, *T3 = e->imx + 6, *A1 = e->imx + 9, *A2 = e->imx + 12, *A3 = e->imx + 15, *M1 = e->imx + 18, *M2 = e->imx + 21;
    double* Tg = e->Tg; double* As = e->As; double* Ma = e->Ma;
    Tg[0] = T2[0]; Tg[1] = T3[0]; Tg[2] = T3[1]; Tg[3] = T1[0]; Tg[4] = T2[1]; Tg[5] = T3[2]; Tg[6] = T1[1]; Tg[7] = T1[2]; Tg[8] = T2[2];
    As[0] = A2[0]; As[1] = A3[0]; As[2] = A3[1]; As[3] = A1[0]; As[4] = A2[1]; As[5] = A3[2]; As[6] = A1[1]; As[7] = A1[2]; As[8] = A2[2];
    Ma[0] = M2[0]; Ma[1] = 0; Ma[2] = 0; Ma[3] = M1[0]; Ma[4] = M2[1]; Ma[5] = 0; Ma[6] = M1[1]; Ma[7] = M1[2]; Ma[8] = M2[2];
}
void lvo_ekf_get_imu_intrinsics(const lvo_ekf* e, double* out24) { memcpy(out24, e->imx, sizeof e->imx); }
void lvo_ekf_set_imu_intrinsics(lvo_ekf* e, const double* in24) { memcpy(e->imx, in24, sizeof e->imx); update_imu_mx(e); }

lvo_ekf* lvo_ekf_create(const lvo_ekf_config* cfg)
{
    lvo_ekf* e = (lvo_ekf*)calloc(1, sizeof *e);
    e->cfg = *cfg;
    const lvo_ekf_config* c = &e->cfg;
    e->td = c->td;
    e->sigma2 = c->noise_feature * c->noise_feature;
    e->zupt_v2 = c->zupt_noise_v * c->zupt_noise_v; e->zupt_p2 = c->zupt_noise_p * c->zupt_noise_p; e->zupt_q2 = c->zupt_noise_q * c->zupt_noise_q;
    e->imu_img_time_th = 1.0 / (2 * c->imu_rate);
    for (int i = 0; i < 3; ++i) {
        e->Qc[i] = c->noise_gyro * c->noise_gyro; e->Qc[3 + i] = c->noise_acc * c->noise_acc;
        e->Qc[6 + i] = c->noise_gyro_bias * c->noise_gyro_bias; e->Qc[9 + i] = c->noise_acc_bias * c->noise_acc_bias;
    }
    e->leg = c->calib_imu_instrinsic ? 46 : 22;
    memset(e->imx, 0, sizeof e->imx);
    for (int i = 0; i < 3; ++i) { e->imx[3 + i] = 1.0; e->imx[21 + i] = 1.0; }     /* T2 = diag(Tg) = 1, M2 = diag(Ma) = 1 */
    update_imu_mx(e);
    e->N = LEG;
    e->P = (double*)calloc((size_t)LEG * LEG, sizeof(double));
    for (int i = 0; i < 3; ++i) {
        e->P[(size_t)i * LEG + i] = c->initial_covariance_orientation;
        e->P[(size_t)(3 + i) * LEG + 3 + i] = c->initial_covariance_velocity;
        e->P[(size_t)(6 + i) * LEG + 6 + i] = c->initial_covariance_position;
        e->P[(size_t)(9 + i) * LEG + 9 + i] = c->initial_covariance_gyro_bias;
        e->P[(size_t)(12 + i) * LEG + 12 + i] = c->initial_covariance_acc_bias;
        if (c->estimate_extrin) {
            e->P[(size_t)(15 + i) * LEG + 15 + i] = c->initial_covariance_extrin_rot;
            e->P[(size_t)(18 + i) * LEG + 18 + i] = c->initial_covariance_extrin_trans;
        }
    }
    if (c->estimate_td) e->P[(size_t)21 * LEG + 21] = 4e-6;
    if (c->calib_imu_instrinsic) for (int i = 22; i < 46; ++i) e->P[(size_t)i * LEG + i] = 1e-4;      /* :183-186 */
    /* extrinsics (larvio.cpp:189-203): R_imu_cam0 = R of T_cam_imu, t_cam0_imu = -R^T t */
    double R[9], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = c->T_cam_imu[i * 4 + j]; t[i] = c->T_cam_imu[i * 4 + 3]; }
    memcpy(e->R_b2c, R, sizeof R);
    double a[3]; m3t_v(R, t, a);
    e->t_c_b[0] = -a[0]; e->t_c_b[1] = -a[1]; e->t_c_b[2] = -a[2];
    e->s.q[3] = 1.0;
    /* grid (larvio.cpp:232-262) */
    const double fx = c->intrinsics[0], fy = c->intrinsics[1], cx = c->intrinsics[2], cy = c->intrinsics[3];
    e->x_min = -cx / fx; e->y_min = -cy / fy;
    double x_max = (c->width - cx) / fx, y_max = (c->height - cy) / fy;
    if (c->aug_grid_rows * c->aug_grid_cols != 0) { e->grid_w = (x_max - e->x_min) / c->aug_grid_cols; e->grid_h = (y_max - e->y_min) / c->aug_grid_rows; }
    else { e->grid_w = x_max - e->x_min; e->grid_h = y_max - e->y_min; }
    e->grid_count = (int*)calloc((size_t)(c->aug_grid_rows * c->aug_grid_cols + 1), sizeof(int));
    { const char* tp = getenv("LVO_TRACE"); e->trace = (tp && tp[0]) ? fopen(tp, "w") : NULL; }
    e->static_num = (int)((float)c->static_duration * (double)c->pub_frequency);
    return e;
}

void lvo_ekf_destroy(lvo_ekf* e)
{
    if (!e) return;
    for (int i = 0; i < e->n_map; ++i) free(e->map[i]);
    if (e->trace) fclose(e->trace);
    free(e->map); free(e->clones); free(e->feature_states); free(e->P); free(e->grid_count); free(e->coarse_dis);
    free(e->init_ids); free(e->init_uv);
    free(e);
}

void lvo_ekf_set_state(lvo_ekf* e, double t, const double q[4], const double p[3], const double v[3], const double bg[3], const double ba[3],
                       const double gyro_old[3], const double acc_old[3])
{
    e->s.t = t;
    memcpy(e->s.q, q, 32); memcpy(e->s.p, p, 24); memcpy(e->s.v, v, 24); memcpy(e->s.bg, bg, 24); memcpy(e->s.ba, ba, 24);
    memcpy(e->m_gyro_old, gyro_old, 24); memcpy(e->m_acc_old, acc_old, 24);
    e->is_gravity_set = 1; e->b_first_features = 1;
    e->take_off_stamp = t; e->last_zupt_time = t - 10.0; e->last_update_time = t;   /* test bypass: EKF-SLAM features allowed at once (larvio.cpp:1974 needs 5 s since the last ZUPT) */
    e->s_fej_now = e->s;
}

/* the start as an initialiser leaves it (larvio.cpp:379-386): the last ZUPT is "now", so in-state features wait 5 s (:1974) */
void lvo_ekf_set_last_zupt_time(lvo_ekf* e, double t) { e->last_zupt_time = t; }

/* ------------------------------------------------------------------------ propagation */
static void predict_new_state(lvo_ekf* e, double dt, const double* gyro, const double* acc)
{   /* larvio.cpp:581-649 */
    double gn = v3_norm(gyro);
    double Om[16] = {0};
    double S[9]; skew3(gyro, S);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Om[i * 4 + j] = -S[i * 3 + j]; Om[i * 4 + 3] = gyro[i]; Om[12 + i] = -gyro[i]; }
    e->s_old = e->s;
    double* q = e->s.q; double* v = e->s.v; double* p = e->s.p;
    double dq[4], dq2[4];
    for (int half = 0; half < 2; ++half) {
        double* o = half ? dq2 : dq;
        double ang = half ? gn * dt * 0.25 : gn * dt * 0.5;
        double M[16];
        if (gn > 1e-5) {
            double c = cos(ang), s = 1 / gn * sin(ang);
            for (int i = 0; i < 16; ++i) M[i] = c * ((i % 5 == 0) ? 1.0 : 0.0) + s * Om[i];
            for (int i = 0; i < 4; ++i) { double a = 0; for (int k = 0; k < 4; ++k) a += M[i * 4 + k] * q[k]; o[i] = a; }
        } else {
            double f = half ? 0.25 * dt : 0.5 * dt, c = cos(ang);
            for (int i = 0; i < 16; ++i) M[i] = (((i % 5 == 0) ? 1.0 : 0.0) + f * Om[i]) * c;
            for (int i = 0; i < 4; ++i) { double a = 0; for (int k = 0; k < 4; ++k) a += M[i * 4 + k] * q[k]; o[i] = a; }
        }
    }
    double Rdt[9], Rdt2[9], R0[9];
    quat_to_rot(dq, Rdt); quat_to_rot(dq2, Rdt2); quat_to_rot(q, R0);
    const double g[3] = {0, 0, -GRAV};
    double k1v[3], k2v[3], k3v[3], k4v[3], k1p[3], k2p[3], k3p[3], k4p[3], t1[3], t2[3];
    m3_v(R0, acc, t1); for (int i = 0; i < 3; ++i) { k1v[i] = t1[i] + g[i]; k1p[i] = v[i]; }
    double k1_v[3]; for (int i = 0; i < 3; ++i) k1_v[i] = v[i] + k1v[i] * dt / 2;
    m3_v(Rdt2, acc, t2); for (int i = 0; i < 3; ++i) { k2v[i] = t2[i] + g[i]; k2p[i] = k1_v[i]; }
    double k2_v[3]; for (int i = 0; i < 3; ++i) k2_v[i] = v[i] + k2v[i] * dt / 2;
    for (int i = 0; i < 3; ++i) { k3v[i] = t2[i] + g[i]; k3p[i] = k2_v[i]; }
    double k3_v[3]; for (int i = 0; i < 3; ++i) k3_v[i] = v[i] + k3v[i] * dt;
    m3_v(Rdt, acc, t1); for (int i = 0; i < 3; ++i) { k4v[i] = t1[i] + g[i]; k4p[i] = k3_v[i]; }
    double n = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
    for (int i = 0; i < 4; ++i) q[i] = dq[i] / n;
    for (int i = 0; i < 3; ++i) {
        double nv = v[i] + dt / 6 * (k1v[i] + 2 * k2v[i] + 2 * k3v[i] + k4v[i]);
        double np = p[i] + dt / 6 * (k1p[i] + 2 * k2p[i] + 2 * k3p[i] + k4p[i]);
        v[i] = nv; p[i] = np;
    }
    e->s_fej_old = e->s_fej_now;
    e->s_fej_now = e->s;
}

static void cal_phi(const lvo_ekf* e, double* Phi /*22x22*/, double dt, const double* f, const double* w, const double* acc, const double* gyro,
                    const double* f_old, const double* w_old, const double* acc_old, const double* gyro_old)
{   /* larvio.cpp:3475-3530 with Ma = Tg = I, As = 0 */
    (void)f; (void)w; (void)acc; (void)f_old; (void)w_old; (void)acc_old;
    double cr[3] = {gyro_old[1] * gyro[2] - gyro_old[2] * gyro[1], gyro_old[2] * gyro[0] - gyro_old[0] * gyro[2], gyro_old[0] * gyro[1] - gyro_old[1] * gyro[0]};
    double aa[3];
    for (int i = 0; i < 3; ++i) aa[i] = dt * (gyro_old[i] + gyro[i]) / 2 + dt * dt * cr[i] / 12;
    double Ah[9]; skew3(aa, Ah);
    double C[9]; quat_to_rot(e->s_old.q, C);
    for (int i = 0; i < LEG * LEG; ++i) Phi[i] = (i % (LEG + 1) == 0) ? 1.0 : 0.0;
    const imu_state_t* so = e->if_fej ? &e->s_fej_old : &e->s_old;
    const imu_state_t* sn = e->if_fej ? &e->s_fej_now : &e->s;
    const double* vk = so->v; const double* pk = so->p; const double* vk1 = sn->v; const double* pk1 = sn->p;
    const double g[3] = {0, 0, -GRAV};
    double I2A[9]; for (int i = 0; i < 9; ++i) I2A[i] = 2 * ((i % 4 == 0) ? 1.0 : 0.0) + Ah[i];
    double CI2A[9]; m3_mul(C, I2A, CI2A);
    #define BLK(r, c, M, sc) for (int i_ = 0; i_ < 3; ++i_) for (int j_ = 0; j_ < 3; ++j_) Phi[((r) + i_) * LEG + (c) + j_] = (sc) * (M)[i_ * 3 + j_]
    /* Phi_q_bg = -0.5 C (2I+A^) dt Tg ; Phi_q_ba = 0.5 C (2I+A^) dt TA Ma = 0 (As = 0) */
    { double M[9]; for (int i = 0; i < 9; ++i) M[i] = -0.5 * CI2A[i] * dt; BLK(0, 9, M, 1.0); }
    { double Z[9] = {0}; BLK(0, 12, Z, 1.0); }
    /* Phi_v_q */
    { double a[3], S[9]; for (int i = 0; i < 3; ++i) a[i] = vk1[i] - vk[i] - g[i] * dt; skew3(a, S); BLK(3, 0, S, -1.0); }
    /* Phi_v_bg */
    double Pvbg[9];
    { double a[3], b[3], S1[9], S2[9], T1[9], T2[9], T3[9];
      for (int i = 0; i < 3; ++i) { a[i] = -pk1[i] + pk[i] + vk1[i] * dt - 0.5 * g[i] * dt * dt; b[i] = -0.5 * pk1[i] + 0.5 * pk[i] + 0.5 * vk1[i] * dt - g[i] * dt * dt / 6; }
      skew3(a, S1); skew3(b, S2); m3_mul(S1, C, T1); m3_mul(S2, C, T2); m3_mul(T2, Ah, T3);
      for (int i = 0; i < 9; ++i) Pvbg[i] = T1[i] + T3[i];
      BLK(3, 9, Pvbg, 1.0); }
    /* Phi_v_ba = -0.5 C (2I+A^) dt Ma - Phi_v_bg TA Ma = -0.5 C(2I+A^) dt */
    { double M[9]; for (int i = 0; i < 9; ++i) M[i] = -0.5 * CI2A[i] * dt - 0.0; BLK(3, 12, M, 1.0); }
    /* Phi_p_q */
    { double a[3], S[9]; for (int i = 0; i < 3; ++i) a[i] = pk1[i] - pk[i] - vk[i] * dt - 0.5 * g[i] * dt * dt; skew3(a, S); BLK(6, 0, S, -1.0); }
    /* Phi_p_v */
    { double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; BLK(6, 3, I, dt); }
    /* Phi_p_bg */
    { double Sg[9], T1[9], a[3], S2[9], T2[9], T3[9], M[9];
      skew3(g, Sg); m3_mul(Sg, C, T1);
      for (int i = 0; i < 3; ++i) a[i] = pk1[i] - pk[i] - g[i] * dt * dt / 6;
      skew3(a, S2); m3_mul(S2, C, T2); m3_mul(T2, Ah, T3);
      for (int i = 0; i < 9; ++i) M[i] = -dt * dt * dt * T1[i] / 6 + dt * T3[i] / 4;
      BLK(6, 9, M, 1.0); }
    /* Phi_p_ba = -C (3I+A^) dt^2/6 Ma - Phi_p_bg TA Ma */
    { double I3A[9], T[9], M[9]; for (int i = 0; i < 9; ++i) I3A[i] = 3 * ((i % 4 == 0) ? 1.0 : 0.0) + Ah[i];
      m3_mul(C, I3A, T); for (int i = 0; i < 9; ++i) M[i] = -T[i] * dt * dt / 6; BLK(6, 12, M, 1.0); }
    #undef BLK
}

/* calPhi with the IMU-intrinsic matrices in the legacy blocks and the 24 extra columns (larvio.cpp:3475-3800, calib_imu = 1).
 * The extra blocks share one pattern per parameter group X in {T1,T2,T3 | A1,A2,A3 | M1,M2}:
 *   kq1 = Pm X_k, kq2 = R_mid Pm X_kh, kq4 = R_kp1 Pm X_kp1, RX = dt (kq1 + 4 kq2 + kq4)/6               Phi_q = sq C RX
 *   kv1 = F_k, kv2 = R_mid F_kh + [R_mid acc_mid]x dt kq1/2, kv3 = R_mid F_kh + [R_mid acc_mid]x dt kq2/2,
 *   kv4 = R_kp1 F_kp1 + [R_kp1 acc]x RX, fRX = dt (kv1 + 2 kv2 + 2 kv3 + kv4)/6                            Phi_v = sv C fRX
 *   Phi_p = sv C dt (2 (dt kv1/2) + 2 (dt kv2/2) + fRX)/6
 * with Pm = I / Tg / Tg As, X built from w / acc / f as a strictly-lower (L), diagonal (D) or strictly-upper (U) pattern, the
 * F terms present for the M groups only, sq = +,-,- and sv = -,+,+ for the T, A, M groups. */
static void pat3(int kind, const double* v, double* M)
{   /* L: (1,0)=v0 (2,1)=v0 (2,2)=v1 ; D: diag(v) ; U: (0,0)=v1 (0,1)=v2 (1,2)=v2   (larvio.cpp:3534-3630, as written there) */
    for (int i = 0; i < 9; ++i) M[i] = 0;
    if (kind == 0) { M[3] = v[0]; M[7] = v[0]; M[8] = v[1]; }
    else if (kind == 1) { M[0] = v[0]; M[4] = v[1]; M[8] = v[2]; }
    else { M[0] = v[1]; M[1] = v[2]; M[5] = v[2]; }
}
static void cal_phi_calib(const lvo_ekf* e, double* Phi, double dt, const double* f, const double* w, const double* acc, const double* gyro,
                          const double* f_old, const double* w_old, const double* acc_old, const double* gyro_old)
{
    const int L = e->leg;
    double f_mid[3], acc_mid[3], w_mid[3], cw[3] = {w_old[1] * w[2] - w_old[2] * w[1], w_old[2] * w[0] - w_old[0] * w[2], w_old[0] * w[1] - w_old[1] * w[0]};
    for (int i = 0; i < 3; ++i) { f_mid[i] = (f[i] + f_old[i]) / 2; acc_mid[i] = (acc[i] + acc_old[i]) / 2; w_mid[i] = (w_old[i] + w[i]) / 2 + dt * cw[i] / 12; }
    double cr[3] = {gyro_old[1] * gyro[2] - gyro_old[2] * gyro[1], gyro_old[2] * gyro[0] - gyro_old[0] * gyro[2], gyro_old[0] * gyro[1] - gyro_old[1] * gyro[0]};
    double aa[3];
    for (int i = 0; i < 3; ++i) aa[i] = dt * (gyro_old[i] + gyro[i]) / 2 + dt * dt * cr[i] / 12;
    double Ah[9]; skew3(aa, Ah);
    double C[9]; quat_to_rot(e->s_old.q, C);
    for (int i = 0; i < L * L; ++i) Phi[i] = (i % (L + 1) == 0) ? 1.0 : 0.0;
    double TA[9]; m3_mul(e->Tg, e->As, TA);
    const imu_state_t* so = e->if_fej ? &e->s_fej_old : &e->s_old;
    const imu_state_t* sn = e->if_fej ? &e->s_fej_now : &e->s;
    const double* vk = so->v; const double* pk = so->p; const double* vk1 = sn->v; const double* pk1 = sn->p;
    const double g[3] = {0, 0, -GRAV};
    double I2A[9]; for (int i = 0; i < 9; ++i) I2A[i] = 2 * ((i % 4 == 0) ? 1.0 : 0.0) + Ah[i];
    double CI2A[9]; m3_mul(C, I2A, CI2A);
    #define BLK(r, c, M, sc) for (int i_ = 0; i_ < 3; ++i_) for (int j_ = 0; j_ < 3; ++j_) Phi[((r) + i_) * L + (c) + j_] = (sc) * (M)[i_ * 3 + j_]
    double TAMa[9]; m3_mul(TA, e->Ma, TAMa);
    { double M[9], T[9]; for (int i = 0; i < 9; ++i) M[i] = -0.5 * CI2A[i] * dt; m3_mul(M, e->Tg, T); BLK(0, 9, T, 1.0); }          /* Phi_q_bg */
    { double M[9], T[9]; for (int i = 0; i < 9; ++i) M[i] = 0.5 * CI2A[i] * dt; m3_mul(M, TAMa, T); BLK(0, 12, T, 1.0); }           /* Phi_q_ba */
    { double a[3], S[9]; for (int i = 0; i < 3; ++i) a[i] = vk1[i] - vk[i] - g[i] * dt; skew3(a, S); BLK(3, 0, S, -1.0); }          /* Phi_v_q */
    double Pvbg[9], Ppbg[9];
    { double a[3], b[3], S1[9], S2[9], T1[9], T2[9], T3[9];
      for (int i = 0; i < 3; ++i) { a[i] = -pk1[i] + pk[i] + vk1[i] * dt - 0.5 * g[i] * dt * dt; b[i] = -0.5 * pk1[i] + 0.5 * pk[i] + 0.5 * vk1[i] * dt - g[i] * dt * dt / 6; }
      skew3(a, S1); skew3(b, S2); m3_mul(S1, C, T1); m3_mul(S2, C, T2); m3_mul(T2, Ah, T3);
      for (int i = 0; i < 9; ++i) Pvbg[i] = T1[i] + T3[i];
      BLK(3, 9, Pvbg, 1.0); }                                                                                                       /* Phi_v_bg */
    { double M[9], T1[9], T2[9]; for (int i = 0; i < 9; ++i) M[i] = -0.5 * CI2A[i] * dt; m3_mul(M, e->Ma, T1); m3_mul(Pvbg, TAMa, T2);
      for (int i = 0; i < 9; ++i) M[i] = T1[i] - T2[i]; BLK(3, 12, M, 1.0); }                                                       /* Phi_v_ba */
    { double a[3], S[9]; for (int i = 0; i < 3; ++i) a[i] = pk1[i] - pk[i] - vk[i] * dt - 0.5 * g[i] * dt * dt; skew3(a, S); BLK(6, 0, S, -1.0); }   /* Phi_p_q */
    { double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; BLK(6, 3, I, dt); }                                                                /* Phi_p_v */
    { double Sg[9], T1[9], a[3], S2[9], T2[9], T3[9];
      skew3(g, Sg); m3_mul(Sg, C, T1);
      for (int i = 0; i < 3; ++i) a[i] = pk1[i] - pk[i] - g[i] * dt * dt / 6;
      skew3(a, S2); m3_mul(S2, C, T2); m3_mul(T2, Ah, T3);
      for (int i = 0; i < 9; ++i) Ppbg[i] = -dt * dt * dt * T1[i] / 6 + dt * T3[i] / 4;
      BLK(6, 9, Ppbg, 1.0); }                                                                                                       /* Phi_p_bg */
    { double I3A[9], T[9], M[9], T1[9], T2[9]; for (int i = 0; i < 9; ++i) I3A[i] = 3 * ((i % 4 == 0) ? 1.0 : 0.0) + Ah[i];
      m3_mul(C, I3A, T); for (int i = 0; i < 9; ++i) M[i] = -T[i] * dt * dt / 6; m3_mul(M, e->Ma, T1); m3_mul(Ppbg, TAMa, T2);
      for (int i = 0; i < 9; ++i) M[i] = T1[i] - T2[i]; BLK(6, 12, M, 1.0); }                                                       /* Phi_p_ba */
    /* ---- the 24 intrinsic columns */
    double R_mid[9], R_kp1[9];
    for (int i = 0; i < 9; ++i) { const double id = (i % 4 == 0) ? 1.0 : 0.0; R_mid[i] = id + 0.5 * Ah[i]; R_kp1[i] = id + Ah[i]; }
    double ram[3], rka[3], Sm[9], Sk[9];
    m3_v(R_mid, acc_mid, ram); m3_v(R_kp1, acc, rka); skew3(ram, Sm); skew3(rka, Sk);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    /* group table: column, pattern kind, source vectors (k, mid, k+1), pre-multiplier, has F terms, signs */
    struct { int col, kind; const double *vk_, *vm_, *vp_; const double* Pm; int has_f; double sq, sv; } grp[8] = {
        {22, 0, w_old, w_mid, w, I3, 0, 1.0, -1.0}, {25, 1, w_old, w_mid, w, I3, 0, 1.0, -1.0}, {28, 2, w_old, w_mid, w, I3, 0, 1.0, -1.0},
        {31, 0, acc_old, acc_mid, acc, e->Tg, 0, -1.0, 1.0}, {34, 1, acc_old, acc_mid, acc, e->Tg, 0, -1.0, 1.0}, {37, 2, acc_old, acc_mid, acc, e->Tg, 0, -1.0, 1.0},
        {40, 0, f_old, f_mid, f, TA, 1, -1.0, 1.0}, {43, 1, f_old, f_mid, f, TA, 1, -1.0, 1.0}};
    for (int gi = 0; gi < 8; ++gi) {
        double Xk[9], Xm[9], Xp[9], kq1[9], kq2[9], kq4[9], T[9], RX[9];
        pat3(grp[gi].kind, grp[gi].vk_, Xk); pat3(grp[gi].kind, grp[gi].vm_, Xm); pat3(grp[gi].kind, grp[gi].vp_, Xp);
        m3_mul(grp[gi].Pm, Xk, kq1);
        m3_mul(grp[gi].Pm, Xm, T); m3_mul(R_mid, T, kq2);
        m3_mul(grp[gi].Pm, Xp, T); m3_mul(R_kp1, T, kq4);
        for (int i = 0; i < 9; ++i) RX[i] = dt * (kq1[i] + 4 * kq2[i] + kq4[i]) / 6;
        m3_mul(C, RX, T); BLK(0, grp[gi].col, T, grp[gi].sq);
        double kv1[9], kv2[9], kv3[9], kv4[9], A[9], fRX[9];
        for (int i = 0; i < 9; ++i) kv1[i] = grp[gi].has_f ? Xk[i] : 0.0;
        m3_mul(Sm, kq1, A); for (int i = 0; i < 9; ++i) kv2[i] = A[i] * dt / 2;
        m3_mul(Sm, kq2, A); for (int i = 0; i < 9; ++i) kv3[i] = A[i] * dt / 2;
        m3_mul(Sk, RX, kv4);
        if (grp[gi].has_f) {
            double RF[9];
            m3_mul(R_mid, Xm, RF); for (int i = 0; i < 9; ++i) { kv2[i] += RF[i]; kv3[i] += RF[i]; }
            m3_mul(R_kp1, Xp, RF); for (int i = 0; i < 9; ++i) kv4[i] += RF[i];
        }
        for (int i = 0; i < 9; ++i) fRX[i] = dt * (kv1[i] + 2 * kv2[i] + 2 * kv3[i] + kv4[i]) / 6;
        m3_mul(C, fRX, T); BLK(3, grp[gi].col, T, grp[gi].sv);
        double kp[9];
        for (int i = 0; i < 9; ++i) kp[i] = dt * (2 * (dt * kv1[i] / 2) + 2 * (dt * kv2[i] / 2) + fRX[i]) / 6;
        m3_mul(C, kp, T); BLK(6, grp[gi].col, T, grp[gi].sv);
    }
    #undef BLK
}

static void process_model(lvo_ekf* e, double time, const double* m_gyro, const double* m_acc)
{   /* larvio.cpp:520-578 */
    double f[3], w[3], f_old[3], w_old[3], acc[3], gyro[3], acc_old[3], gyro_old[3];
    for (int i = 0; i < 3; ++i) { f[i] = m_acc[i] - e->s.ba[i]; f_old[i] = e->m_acc_old[i] - e->s.ba[i]; }
    if (e->cfg.calib_imu_instrinsic) {
        double t[3];
        m3_v(e->Ma, f, acc); m3_v(e->As, acc, t); for (int i = 0; i < 3; ++i) w[i] = m_gyro[i] - t[i] - e->s.bg[i]; m3_v(e->Tg, w, gyro);
        m3_v(e->Ma, f_old, acc_old); m3_v(e->As, acc_old, t); for (int i = 0; i < 3; ++i) w_old[i] = e->m_gyro_old[i] - t[i] - e->s.bg[i]; m3_v(e->Tg, w_old, gyro_old);
    } else {
        for (int i = 0; i < 3; ++i) { w[i] = m_gyro[i] - e->s.bg[i]; w_old[i] = e->m_gyro_old[i] - e->s.bg[i]; acc[i] = f[i]; gyro[i] = w[i]; acc_old[i] = f_old[i]; gyro_old[i] = w_old[i]; }
    }
    double dtime = time - e->s.t;
    predict_new_state(e, dtime, gyro, acc);
    double Phi[LEG_MAX * LEG_MAX];
    if (e->cfg.calib_imu_instrinsic) cal_phi_calib(e, Phi, dtime, f, w, acc, gyro, f_old, w_old, acc_old, gyro_old);
    else cal_phi(e, Phi, dtime, f, w, acc, gyro, f_old, w_old, acc_old, gyro_old);
    double C[9]; quat_to_rot(e->s_old.q, C);
    double G[LEG_MAX * 12]; memset(G, 0, sizeof G);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { G[i * 12 + j] = -C[i * 3 + j]; G[(3 + i) * 12 + 3 + j] = -C[i * 3 + j]; }
    for (int i = 0; i < 3; ++i) { G[(9 + i) * 12 + 6 + i] = 1.0; G[(12 + i) * 12 + 9 + i] = 1.0; }
    double PG[LEG_MAX * 12], Q[LEG_MAX * LEG_MAX];
    for (int i = 0; i < LEG; ++i) for (int j = 0; j < 12; ++j) { double s = 0; for (int k = 0; k < LEG; ++k) s += Phi[i * LEG + k] * G[k * 12 + j]; PG[i * 12 + j] = s; }
    for (int i = 0; i < LEG; ++i) for (int j = 0; j < LEG; ++j) { double s = 0; for (int k = 0; k < 12; ++k) s += PG[i * 12 + k] * e->Qc[k] * PG[j * 12 + k]; Q[i * LEG + j] = s * dtime; }
    const int N = e->N;
    double* P = e->P;
    /* P_II <- Phi P_II Phi^T + Q */
    double T[LEG_MAX * LEG_MAX], PII[LEG_MAX * LEG_MAX];
    for (int i = 0; i < LEG; ++i) for (int j = 0; j < LEG; ++j) { double s = 0; for (int k = 0; k < LEG; ++k) s += Phi[i * LEG + k] * P[(size_t)k * N + j]; T[i * LEG + j] = s; }
    for (int i = 0; i < LEG; ++i) for (int j = 0; j < LEG; ++j) { double s = 0; for (int k = 0; k < LEG; ++k) s += T[i * LEG + k] * Phi[j * LEG + k]; PII[i * LEG + j] = s + Q[i * LEG + j]; }
    if (N > LEG) {
        double* PIC = (double*)malloc(sizeof(double) * (size_t)LEG * (N - LEG));
        double* PCI = (double*)malloc(sizeof(double) * (size_t)(N - LEG) * LEG);
        for (int i = 0; i < LEG; ++i) for (int j = LEG; j < N; ++j) { double s = 0; for (int k = 0; k < LEG; ++k) s += Phi[i * LEG + k] * P[(size_t)k * N + j]; PIC[(size_t)i * (N - LEG) + j - LEG] = s; }
        for (int i = LEG; i < N; ++i) for (int j = 0; j < LEG; ++j) { double s = 0; for (int k = 0; k < LEG; ++k) s += P[(size_t)i * N + k] * Phi[j * LEG + k]; PCI[(size_t)(i - LEG) * LEG + j] = s; }
        for (int i = 0; i < LEG; ++i) for (int j = LEG; j < N; ++j) P[(size_t)i * N + j] = PIC[(size_t)i * (N - LEG) + j - LEG];
        for (int i = LEG; i < N; ++i) for (int j = 0; j < LEG; ++j) P[(size_t)i * N + j] = PCI[(size_t)(i - LEG) * LEG + j];
        free(PIC); free(PCI);
    }
    for (int i = 0; i < LEG; ++i) for (int j = 0; j < LEG; ++j) P[(size_t)i * N + j] = PII[i * LEG + j];
    P_symmetrize(P, N);
    e->s.t = time; e->s_fej_now.t = time;
}

static int batch_imu(lvo_ekf* e, double time_bound, const lvo_imu* imu, int n_imu)
{   /* larvio.cpp:464-517; returns used_imu_msg_cntr */
    int used = 0; double dt = 0.0;
    for (int i = 0; i < n_imu; ++i) {
        double imu_time = imu[i].t;
        if (imu_time <= e->s.t) { ++used; continue; }
        if (imu_time - time_bound > e->imu_img_time_th) break;
        dt = imu_time - time_bound;
        process_model(e, imu_time, imu[i].gyro, imu[i].acc);
        ++used;
        memcpy(e->m_gyro_old, imu[i].gyro, 24); memcpy(e->m_acc_old, imu[i].acc, 24);
    }
    e->imu_id = e->next_state_id++;
    e->imu_dt = dt;
    return used;
}

static void state_augmentation(lvo_ekf* e)
{   /* larvio.cpp:720-801 */
    if (e->n_clones == e->cap_clones) { e->cap_clones = e->cap_clones ? 2 * e->cap_clones : 64; e->clones = (lvo_clone*)realloc(e->clones, sizeof(lvo_clone) * (size_t)e->cap_clones); }
    lvo_clone* c = &e->clones[e->n_clones];
    memset(c, 0, sizeof *c);
    c->id = e->imu_id; c->time = e->s.t; c->dt = e->imu_dt;
    memcpy(c->q, e->s.q, 32); memcpy(c->p, e->s.p, 24); memcpy(c->p_fej, e->s_fej_now.p, 24);
    memcpy(c->R_b2c, e->R_b2c, 72); memcpy(c->t_c_b, e->t_c_b, 24);
    {   /* q_w_c = Quaterniond((R_b2c R_b2w^T)^T), t_c_w = p + R_b2w t_c_b */
        double R_b2w[9], R_w2b[9], R_w2c[9], R_c2w[9], t[3];
        quat_to_rot(c->q, R_b2w); m3_t(R_b2w, R_w2b); m3_mul(e->R_b2c, R_w2b, R_w2c); m3_t(R_w2c, R_c2w);
        rot_to_quat(R_c2w, c->q_cam);
        m3_v(R_b2w, e->t_c_b, t);
        for (int i = 0; i < 3; ++i) c->p_cam[i] = e->s.p[i] + t[i];
    }
    const int pose_rows = LEG + 6 * e->n_clones;
    e->n_clones++;
    /* covariance: six new rows/cols = copies of rows/cols {0,1,2,6,7,8}, inserted before the feature block */
    const int newN = e->N + 6;
    int* idx = (int*)malloc(sizeof(int) * (size_t)newN);
    static const int sel[6] = {0, 1, 2, 6, 7, 8};
    int k = 0;
    for (int i = 0; i < pose_rows; ++i) idx[k++] = i;
    for (int i = 0; i < 6; ++i) idx[k++] = sel[i];
    for (int i = pose_rows; i < e->N; ++i) idx[k++] = i;
    P_gather(e, idx, newN);
    free(idx);
    P_symmetrize(e->P, e->N);
}

static void add_observations(lvo_ekf* e, const lvo_feature_obs* f, int n)
{   /* larvio.cpp:804-856 */
    const int64_t sid = e->imu_id;
    const int curr_num = e->n_map;
    int tracked = 0;
    const double dt = e->imu_dt;
    const int prev_rank = clone_rank(e, sid - 1);
    for (int i = 0; i < n; ++i) {
        const int64_t id = (int64_t)f[i].id;
        feat_t* ft = map_find(e, id);
        if (!ft) {
            ft = map_insert(e, id);
            feat_obs_set(ft, sid, f[i].u + f[i].u_vel * dt, f[i].v + f[i].v_vel * dt, f[i].u_vel, f[i].v_vel);
            ft->total_obs++;
            if (!(f[i].u_init == -1 && f[i].v_init == -1) && prev_rank >= 0) {
                double dt_ = e->clones[prev_rank].dt;
                feat_obs_set(ft, sid - 1, f[i].u_init + f[i].u_init_vel * dt_, f[i].v_init + f[i].v_init_vel * dt_, f[i].u_init_vel, f[i].v_init_vel);
                ft->total_obs++;
            }
        } else {
            feat_obs_set(ft, sid, f[i].u + f[i].u_vel * dt, f[i].v + f[i].v_vel * dt, f[i].u_vel, f[i].v_vel);
            ft->total_obs++;
            ++tracked;
            int pi;
            if (e->cfg.if_zupt_valid && (pi = feat_obs_find(ft, sid - 1)) >= 0) {
                double dx = f[i].u - ft->z[pi][0], dy = f[i].v - ft->z[pi][1];
                if (e->n_coarse == e->cap_coarse) { e->cap_coarse = e->cap_coarse ? 2 * e->cap_coarse : 256; e->coarse_dis = (double*)realloc(e->coarse_dis, sizeof(double) * (size_t)e->cap_coarse); }
                e->coarse_dis[e->n_coarse++] = sqrt(dx * dx + dy * dy);
            }
        }
    }
    e->tracking_rate = (double)tracked / (double)curr_num;
}

/* ------------------------------------------------------------------------ state injection (shared by the three updates) */
static void inject(lvo_ekf* e, const double* dx, int n_old_features /* features whose dx index is base+i */)
{   /* larvio.cpp:1476-1575 / 1692-1801 / 2836-2936 */
    (void)n_old_features;
    double dq[4], q[4];
    small_angle_quat(dx, dq); quat_mul(dq, e->s.q, q); memcpy(e->s.q, q, 32);
    for (int i = 0; i < 3; ++i) { e->s.v[i] += dx[3 + i]; e->s.p[i] += dx[6 + i]; e->s.bg[i] += dx[9 + i]; e->s.ba[i] += dx[12 + i]; }
    double dqe[4], Re[9], Ret[9], Rn[9];
    small_angle_quat(dx + 15, dqe); quat_to_rot(dqe, Re); m3_t(Re, Ret); m3_mul(e->R_b2c, Ret, Rn); memcpy(e->R_b2c, Rn, 72);
    for (int i = 0; i < 3; ++i) e->t_c_b[i] += dx[18 + i];
    e->td += dx[21];
    if (e->cfg.calib_imu_instrinsic) { for (int i = 0; i < 24; ++i) e->imx[i] += dx[22 + i]; update_imu_mx(e); }      /* :1497-1507 */
    for (int c = 0; c < e->n_clones; ++c) {
        lvo_clone* cl = &e->clones[c];
        const double* d = dx + LEG + 6 * c;
        double dqc[4], qc[4];
        small_angle_quat(d, dqc); quat_mul(dqc, cl->q, qc); memcpy(cl->q, qc, 32);
        for (int i = 0; i < 3; ++i) cl->p[i] += d[3 + i];
        clone_refresh_cam(e, cl);
    }
    const int base = LEG + 6 * e->n_clones;
    for (int i = 0; i < e->n_fs; ++i) {
        feat_t* f = map_find(e, e->feature_states[i]);
        int ar = clone_rank(e, f->id_anchor);
        if (ar < 0) continue;
        const lvo_clone* a = &e->clones[ar];
        double R_c2w[9]; quat_to_rot(a->q_cam, R_c2w);
        f->inv_depth += dx[base + i];
        double pc[3] = {f->obs_anchor[0] / f->inv_depth, f->obs_anchor[1] / f->inv_depth, 1 / f->inv_depth}, pw[3];
        m3_v(R_c2w, pc, pw);
        for (int k = 0; k < 3; ++k) f->position[k] = pw[k] + a->p_cam[k];
    }
}

static int gating_test(lvo_ekf* e, const double* H, const double* r, int k, int dof)
{
    double gamma = lvo_gating_gamma(H, r, k, e->N, e->P, e->N, e->sigma2);
    int ok = gamma < lvo_chi2_table(dof);
    e->counters[ok ? 4 : 5]++;
    return ok;
}

/* featureJacobian_msckf over a set of observing states (all obs of the feature, or the involved ones) */
static int feature_jacobian_msckf(lvo_ekf* e, const feat_t* f, const int64_t* sids, int ns, double* H, double* r)
{
    int ranks[MAX_OBS]; double z[2 * MAX_OBS], zv[2 * MAX_OBS]; int M = 0;
    for (int i = 0; i < ns; ++i) {
        int oi = feat_obs_find(f, sids[i]);
        if (oi < 0) continue;
        ranks[M] = clone_rank(e, sids[i]);
        z[2 * M] = f->z[oi][0]; z[2 * M + 1] = f->z[oi][1]; zv[2 * M] = f->zv[oi][0]; zv[2 * M + 1] = f->zv[oi][1];
        ++M;
    }
    return lvo_msckf_feature_jacobian(e->clones, ranks, z, zv, M, f->position, e->N, LEG, e->if_fej, e->cfg.estimate_td, H, r);
}

/* measurementJacobian_ekf_1didp (larvio.cpp:1117-1244); returns 0 for the anchor's own observation */
static int ekf_obs_jacobian(const lvo_ekf* e, const feat_t* f, const lvo_clone* k, const lvo_clone* a, const double* z,
                            double* Hf /*2*/, double* Ha /*2x6*/, double* Hx /*2x6*/, double* He /*2x6*/, double* r)
{
    const double* R_b2c = k->R_b2c; const double* t_c_b = k->t_c_b; const double* f_an = f->obs_anchor;
    double R_bk2w[9], R_w2bk[9], R_w2ck[9], R_ba2w[9], R_w2ba[9], R_w2ca[9], tmp[3];
    quat_to_rot(k->q, R_bk2w); m3_t(R_bk2w, R_w2bk); m3_mul(R_b2c, R_w2bk, R_w2ck);
    m3_v(R_bk2w, t_c_b, tmp);
    double t_ck_w[3] = {k->p[0] + tmp[0], k->p[1] + tmp[1], k->p[2] + tmp[2]};
    quat_to_rot(a->q, R_ba2w); m3_t(R_ba2w, R_w2ba); m3_mul(R_b2c, R_w2ba, R_w2ca);
    double p_ca[3];
    if (e->if_fej) {
        double d[3] = {f->position_fej[0] - a->p_fej[0], f->position_fej[1] - a->p_fej[1], f->position_fej[2] - a->p_fej[2]}, q[3];
        m3_v(R_w2ba, d, q); q[0] -= t_c_b[0]; q[1] -= t_c_b[1]; q[2] -= t_c_b[2];
        m3_v(R_b2c, q, p_ca);
    } else { p_ca[0] = f_an[0] / f->inv_depth; p_ca[1] = f_an[1] / f->inv_depth; p_ca[2] = 1 / f->inv_depth; }
    const double* p_w = f->position;
    double d[3] = {p_w[0] - t_ck_w[0], p_w[1] - t_ck_w[1], p_w[2] - t_ck_w[2]}, p_ck[3];
    m3_v(R_w2ck, d, p_ck);
    r[0] = z[0] - p_ck[0] / p_ck[2]; r[1] = z[1] - p_ck[1] / p_ck[2];
    if (k->id == f->id_anchor) return 0;
    double Jk[6] = {1 / p_ck[2], 0, -p_ck[0] / (p_ck[2] * p_ck[2]), 0, 1 / p_ck[2], -p_ck[1] / (p_ck[2] * p_ck[2])};
    double R_ca2w[9], M1[9], Jd[3];
    m3_t(R_w2ca, R_ca2w); m3_mul(R_w2ck, R_ca2w, M1); m3_v(M1, f_an, Jd);
    double p_baf[3], p_bkf[3];
    for (int i = 0; i < 3; ++i) {
        p_baf[i] = e->if_fej ? f->position_fej[i] - a->p_fej[i] : p_w[i] - a->p[i];
        p_bkf[i] = e->if_fej ? f->position_fej[i] - k->p_fej[i] : p_w[i] - k->p[i];
    }
    double Sa[9], Sk[9], A1[9], K1[9];
    skew3(p_baf, Sa); skew3(p_bkf, Sk); m3_mul(R_w2ck, Sa, A1); m3_mul(R_w2ck, Sk, K1);
    double Jxa[18], Jxk[18], Je[18];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        Jxa[i * 6 + j] = -A1[i * 3 + j]; Jxa[i * 6 + 3 + j] = R_w2ck[i * 3 + j];
        Jxk[i * 6 + j] = K1[i * 3 + j]; Jxk[i * 6 + 3 + j] = -R_w2ck[i * 3 + j];
    }
    double v1[3], SkewMx[9], RR[9], R_c2b[9], v2[3], S2[9], Mx[9], D[9], E[9];
    m3_v(R_w2bk, p_bkf, v1); v1[0] -= t_c_b[0]; v1[1] -= t_c_b[1]; v1[2] -= t_c_b[2];
    skew3(v1, SkewMx);
    m3_mul(R_w2bk, R_ba2w, RR);                       /* R_w2bk * R_w2ba^T */
    m3_t(R_b2c, R_c2b); m3_v(R_c2b, p_ca, v2); skew3(v2, S2); m3_mul(RR, S2, Mx);
    for (int i = 0; i < 9; ++i) D[i] = SkewMx[i] - Mx[i];
    double JeL[9], JeR[9];
    m3_mul(R_b2c, D, JeL);
    for (int i = 0; i < 9; ++i) E[i] = RR[i] - ((i % 4 == 0) ? 1.0 : 0.0);
    m3_mul(R_b2c, E, JeR);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Je[i * 6 + j] = JeL[i * 3 + j]; Je[i * 6 + 3 + j] = JeR[i * 3 + j]; }
    const double J_rho = -1 / (f->inv_depth * f->inv_depth);
    for (int i = 0; i < 2; ++i) {
        double s = 0; for (int c = 0; c < 3; ++c) s += Jk[i * 3 + c] * Jd[c];
        Hf[i] = s * J_rho;
        for (int j = 0; j < 6; ++j) {
            double s1 = 0, s2 = 0, s3 = 0;
            for (int c = 0; c < 3; ++c) { s1 += Jk[i * 3 + c] * Jxa[c * 6 + j]; s2 += Jk[i * 3 + c] * Jxk[c * 6 + j]; s3 += Jk[i * 3 + c] * Je[c * 6 + j]; }
            Ha[i * 6 + j] = s1; Hx[i * 6 + j] = s2; He[i * 6 + j] = s3;
        }
    }
    return 1;
}

/* rows of an in-state (or about-to-be) feature for a list of observing states; cols = ncols; feature column = fcol */
static int feature_jacobian_ekf(lvo_ekf* e, const feat_t* f, const int64_t* sids, int ns, int skip_anchor, int ncols, int fcol, double* H, double* r)
{
    const int ar = clone_rank(e, f->id_anchor);
    int rows = 0;
    for (int i = 0; i < ns; ++i) {
        int oi = feat_obs_find(f, sids[i]);
        if (oi < 0) continue;
        if (skip_anchor && sids[i] == f->id_anchor) continue;
        int kr = clone_rank(e, sids[i]);
        double Hf[2], Ha[12], Hx[12], He[12], ri[2];
        ekf_obs_jacobian(e, f, &e->clones[kr], &e->clones[ar], f->z[oi], Hf, Ha, Hx, He, ri);
        for (int a = 0; a < 2; ++a) {
            double* row = H + (size_t)(rows + a) * ncols;
            memset(row, 0, sizeof(double) * (size_t)ncols);
            row[fcol] = Hf[a];
            for (int j = 0; j < 6; ++j) row[LEG + 6 * ar + j] = Ha[a * 6 + j];
            for (int j = 0; j < 6; ++j) row[LEG + 6 * kr + j] = Hx[a * 6 + j];
            for (int j = 0; j < 6; ++j) row[15 + j] = He[a * 6 + j];
            if (e->cfg.estimate_td) row[21] = f->zv[oi][a];
            r[rows + a] = ri[a];
        }
        rows += 2;
    }
    return rows;
}

/* ------------------------------------------------------------------------ triangulation wrappers (feature.hpp) */
static int feat_check_motion(const lvo_ekf* e, const feat_t* f, int if_tracked)
{
    int first = 0, last = if_tracked ? f->n_obs - 2 : f->n_obs - 1;
    int r0 = clone_rank(e, f->sid[first]), r1 = clone_rank(e, f->sid[last]);
    lvo_pose a, b; clone_cam_pose(&e->clones[r0], &a); clone_cam_pose(&e->clones[r1], &b);
    return lvo_check_motion(&a, &b, f->z[first], e->cfg.feature_translation_threshold);
}
/* mode 0 initializePosition(curr_id), 1 initializePosition_AssignAnchor, 2 initializeInvParamPosition(curr_id) */
static int feat_initialize(lvo_ekf* e, feat_t* f, int mode)
{
    lvo_pose poses[MAX_OBS]; double obs[2 * MAX_OBS]; int64_t ids[MAX_OBS]; int n = 0;
    for (int i = 0; i < f->n_obs; ++i) {
        int r = clone_rank(e, f->sid[i]);
        if (r < 0) continue;
        if (mode != 1 && f->sid[i] == e->imu_id) continue;
        clone_cam_pose(&e->clones[r], &poses[n]);
        obs[2 * n] = f->z[i][0]; obs[2 * n + 1] = f->z[i][1]; ids[n] = f->sid[i];
        ++n;
    }
    double pos[3], sol[3], idp, oa[3];
    int use_pos = (mode != 2) && f->is_initialized;
    int ok = lvo_triangulate(poses, obs, n, use_pos, f->position, pos, sol, &idp, oa);
    if (ok) {
        if (!f->is_initialized) memcpy(f->position_fej, f->position, 24);      /* feature.hpp:538-539: BEFORE position is overwritten */
        f->is_initialized = 1;
        memcpy(f->position, pos, 24);
        f->id_anchor = ids[n - 1];
        f->inv_depth = idp; memcpy(f->obs_anchor, oa, 24);
        if (mode == 2) f->ekf_feature = 1;
    }
    return ok;
}

/* ------------------------------------------------------------------------ updates */
static void update_msckf(lvo_ekf* e, double* H, double* r, int rows)
{   /* measurementUpdate_msckf (larvio.cpp:1420-1602) */
    if (rows == 0) return;
    const int N = e->N;
    int m = rows;
    if (rows > N) { lvo_qr_compress(H, r, rows, N); m = LEG + 6 * e->n_clones; }
    double* dx = (double*)malloc(sizeof(double) * (size_t)N);
    lvo_ekf_update(e->P, N, N, H, m, r, e->sigma2, dx);
    inject(e, dx, e->n_fs);
    free(dx);
    e->last_update_time = e->s.t;
    e->counters[1]++; e->counters[2] = m;
}

static void rm_lost_features_cov(lvo_ekf* e, const int64_t* ids, int n)
{   /* larvio.cpp:3296-3348 */
    for (int i = 0; i < n; ++i) {
        int seq = fs_rank(e, ids[i]);
        P_delete(e, LEG + 6 * e->n_clones + seq, 1);
        memmove(e->feature_states + seq, e->feature_states + seq + 1, sizeof(int64_t) * (size_t)(e->n_fs - seq - 1));
        e->n_fs--;
        map_erase(e, ids[i]);
    }
}
static int grid_code(const lvo_ekf* e, const double* xy)
{
    int row = (int)((xy[1] - e->y_min) / e->grid_h), col = (int)((xy[0] - e->x_min) / e->grid_w);
    return row * e->cfg.aug_grid_cols + col;
}
static int* phantom_cell(lvo_ekf* e, int code)
{
    for (int i = 0; i < e->n_phantom; ++i) if (e->phantom_code[i] == code) return &e->phantom_count[i];
    if (e->n_phantom < 512) { e->phantom_code[e->n_phantom] = code; e->phantom_count[e->n_phantom] = 0; return &e->phantom_count[e->n_phantom++]; }
    static int full; full = 1 << 30; return &full;
}
static int grid_occupancy(lvo_ekf* e, int code, int cells)
{   /* grid_map[code].size() (larvio.cpp:1974) */
    if (code >= 0 && code < cells) return e->grid_count[code];
    return e->cfg.reference_grid ? *phantom_cell(e, code) : 0;
}
static void grid_add(lvo_ekf* e, int code, int cells)
{   /* grid_map[code].push_back(id) (:1990, :3366) */
    if (code >= 0 && code < cells) e->grid_count[code]++;
    else if (e->cfg.reference_grid) (*phantom_cell(e, code))++;
}
static void update_grid_map(lvo_ekf* e)
{   /* larvio.cpp:3351-3370 */
    const int cells = e->cfg.aug_grid_rows * e->cfg.aug_grid_cols;
    if (cells == 0) return;
    for (int i = 0; i < cells; ++i) e->grid_count[i] = 0;
    for (int i = 0; i < e->n_fs; ++i) {
        feat_t* f = map_find(e, e->feature_states[i]);
        int oi = feat_obs_find(f, e->imu_id);
        double xy[2] = {0, 0};
        if (oi >= 0) { xy[0] = f->z[oi][0]; xy[1] = f->z[oi][1]; }
        int code = grid_code(e, xy);
        grid_add(e, code, cells);
        TR(e, "GRIDF %lld %.17g %.17g\n", (long long)f->id, xy[0], xy[1]);
    }
    if (e->trace) { TR(e, "GRID %d %d %.17g %.17g %.17g %.17g", e->cfg.aug_grid_rows, e->cfg.aug_grid_cols, e->x_min, e->y_min, e->grid_w, e->grid_h);
                    for (int i = 0; i < cells; ++i) TR(e, " %d", e->grid_count[i]); TR(e, "\n"); }
}

/* delayed initialisation of new in-state features (larvio.cpp:1821-1854), 1-D: HH = H_2^-1 H_1 (H_2 diagonal) */
static void delayed_init_dx(int N, int n_acc, const double* H1, const double* H2, const double* r1, double* dx /* N + n_acc */)
{   /* dx_new = -HH dx_leg + H_2^-1 r_1 */
    for (int j = 0; j < n_acc; ++j) {
        double s = 0;
        for (int cc = 0; cc < N; ++cc) s += (H1[(size_t)j * N + cc] / H2[j]) * dx[cc];
        dx[N + j] = -s + r1[j] / H2[j];
    }
}
static double* delayed_init_cov(const double* P, int N, int n_acc, const double* H1, const double* H2, double sigma2)
{   /* nHHP = -HH P ; P22 = -nHHP HH^T + sigma2 (H_2^T H_2)^-1 ; append ; symmetrise.  Returns the (N + n_acc)^2 matrix (malloc) */
    double* HH = (double*)malloc(sizeof(double) * (size_t)(n_acc + 1) * N);
    for (int j = 0; j < n_acc; ++j) for (int cc = 0; cc < N; ++cc) HH[(size_t)j * N + cc] = H1[(size_t)j * N + cc] / H2[j];
    double* nHHP = (double*)malloc(sizeof(double) * (size_t)(n_acc + 1) * N);
    for (int j = 0; j < n_acc; ++j) for (int b = 0; b < N; ++b) { double s = 0; for (int k = 0; k < N; ++k) s += HH[(size_t)j * N + k] * P[(size_t)k * N + b]; nHHP[(size_t)j * N + b] = -s; }
    const int newN = N + n_acc;
    double* Q = (double*)calloc((size_t)newN * newN, sizeof(double));
    for (int a = 0; a < N; ++a) memcpy(Q + (size_t)a * newN, P + (size_t)a * N, sizeof(double) * (size_t)N);
    for (int j = 0; j < n_acc; ++j) {
        for (int b = 0; b < N; ++b) { Q[(size_t)(N + j) * newN + b] = nHHP[(size_t)j * N + b]; Q[(size_t)b * newN + N + j] = nHHP[(size_t)j * N + b]; }
        for (int l = 0; l < n_acc; ++l) {
            double s = 0; for (int k = 0; k < N; ++k) s += nHHP[(size_t)j * N + k] * HH[(size_t)l * N + k];
            Q[(size_t)(N + j) * newN + N + l] = -s + (j == l ? sigma2 * (1.0 / (H2[j] * H2[j])) : 0.0);
        }
    }
    P_symmetrize(Q, newN);
    free(nHHP); free(HH);
    return Q;
}

static void remove_lost_features(lvo_ekf* e)
{   /* larvio.cpp:1883-2256 */
    const lvo_ekf_config* c = &e->cfg;
    int n_ekf = 0, n_ekf_lost = 0;
    int64_t* ekf_ids = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e->n_map + 1));
    int64_t* ekf_lost = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e->n_map + 1));
    for (int i = 0; i < e->n_map; ++i) {
        feat_t* f = e->map[i];
        int tracked = feat_obs_find(f, e->imu_id) >= 0;
        if (f->in_state) { if (tracked) ekf_ids[n_ekf++] = f->id; else ekf_lost[n_ekf_lost++] = f->id; }
    }
    rm_lost_features_cov(e, ekf_lost, n_ekf_lost);
    update_grid_map(e);
    int64_t* invalid = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e->n_map + 1)); int n_invalid = 0;
    int64_t* msckf = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e->n_map + 1)); int n_msckf = 0;
    int64_t* ekf_new = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e->n_map + 1)); int n_new = 0;
    int rows_msckf = 0, rows_new = 0;
    const int cells = c->aug_grid_rows * c->aug_grid_cols;
    /* trace record per feature not in state (tests/test_oracle_decisions.py): TRI id tracked n_obs init_before ekf_feature x y motion
     * tri_ok -> category (0 untouched, 1 invalid, 2 msckf, 3 ekf_new, 4 failed initialisation) ; motion / tri_ok = -1 when the
     * reference's control flow does not evaluate them */
    TR(e, "TRIAGE %d %d %d %d %d %.17g\n", c->least_observation_number, c->max_track_len, c->max_features_in_one_grid, cells, e->n_fs, e->s.t - e->last_zupt_time);
    for (int i = 0; i < e->n_map; ++i) {
        feat_t* f = e->map[i];
        if (f->in_state) continue;
        int tracked = feat_obs_find(f, e->imu_id) >= 0;
        const int init0 = f->is_initialized, ekf0 = f->ekf_feature; int mot = -1, tri = -1, cat = 0; double tx = 0, ty = 0;
        if (!tracked) {
            if (f->n_obs < c->least_observation_number) { invalid[n_invalid++] = f->id; cat = 1; goto rec; }
            if (!f->is_initialized) {
                mot = feat_check_motion(e, f, tracked);
                if (!mot) { invalid[n_invalid++] = f->id; cat = 1; goto rec; }
                tri = feat_initialize(e, f, 0);
                if (!tri) { invalid[n_invalid++] = f->id; cat = 1; goto rec; }
            }
            rows_msckf += 2 * f->n_obs - 3;
            msckf[n_msckf++] = f->id; cat = 2;
        } else {
            if (!(f->n_obs >= c->max_track_len)) goto rec;
            int oi = feat_obs_find(f, e->imu_id);
            int code = grid_code(e, f->z[oi]);
            tx = f->z[oi][0]; ty = f->z[oi][1];
            int gcount = grid_occupancy(e, code, cells);
            if (gcount < c->max_features_in_one_grid && e->s.t - e->last_zupt_time > 5 &&
                (e->n_fs + n_new) < c->max_features_in_one_grid * cells) {
                if (!f->ekf_feature) {
                    f->is_initialized = 0;
                    mot = feat_check_motion(e, f, tracked);
                    if (mot) tri = feat_initialize(e, f, 2);
                }
                if (!f->is_initialized) { cat = 4; goto rec; }
                rows_new += 2 * (f->n_obs - 1);
                ekf_new[n_new++] = f->id; cat = 3;
                grid_add(e, code, cells);
            } else {
                if (!f->is_initialized) { mot = feat_check_motion(e, f, tracked); if (mot) tri = feat_initialize(e, f, 0); }
                if (!f->is_initialized) { cat = 4; goto rec; }
                rows_msckf += 2 * f->n_obs - 3;
                msckf[n_msckf++] = f->id; cat = 2;
            }
        }
    rec:
        TR(e, "TRI %lld %d %d %d %d %.17g %.17g %d %d %d %d\n", (long long)f->id, tracked, f->n_obs, init0, ekf0, tx, ty, mot, tri, cat, f->is_initialized);
    }
    TR(e, "TRIEND\n");
    for (int i = 0; i < n_invalid; ++i) map_erase(e, invalid[i]);
    if (n_msckf == 0 && n_new == 0 && n_ekf == 0) goto done;
    if (!e->if_zupt) {
        const int N = e->N;
        /* ---- new in-state features */
        for (int i = 0; i < n_new; ++i) {
            map_find(e, ekf_new[i])->in_state = 1;
            if (e->n_fs == e->cap_fs) { e->cap_fs = e->cap_fs ? 2 * e->cap_fs : 64; e->feature_states = (int64_t*)realloc(e->feature_states, sizeof(int64_t) * (size_t)e->cap_fs); }
            e->feature_states[e->n_fs++] = ekf_new[i];
        }
        const int n_fs_old = e->n_fs - n_new;
        /* per feature: ekf rows (N + n_new cols) rotated by W_j = [V_j U_j] (larvio.cpp:2095-2119; H_f is block-diagonal
         * in 1-D mode so the rotation factorises per feature).  Null rows -> H_o, the U row -> (H_1, H_2, r_1). */
        double* Hn_top = (double*)calloc((size_t)(rows_new + 1) * N, sizeof(double)); double* rn_top = (double*)calloc((size_t)rows_new + 1, sizeof(double)); int top = 0;
        double* H1 = (double*)calloc((size_t)(n_new + 1) * N, sizeof(double)); double* H2 = (double*)calloc((size_t)n_new + 1, sizeof(double)); double* r1 = (double*)calloc((size_t)n_new + 1, sizeof(double));
        int n_acc = 0;
        int64_t* acc_ids = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_new + 1));
        for (int i = 0; i < n_new; ++i) {
            feat_t* f = map_find(e, ekf_new[i]);
            const int ncols = N + 1;
            double* Hj = (double*)calloc((size_t)(2 * f->n_obs) * ncols, sizeof(double)); double* rj = (double*)calloc((size_t)2 * f->n_obs, sizeof(double));
            int rows = feature_jacobian_ekf(e, f, f->sid, f->n_obs, 1, ncols, N, Hj, rj);
            double* Hm = (double*)calloc((size_t)(2 * f->n_obs) * N, sizeof(double)); double* rm = (double*)calloc((size_t)2 * f->n_obs, sizeof(double));
            int km = feature_jacobian_msckf(e, f, f->sid, f->n_obs, Hm, rm);
            if (gating_test(e, Hm, rm, km, 2 * f->n_obs - 3)) {
                /* Householder on the single feature column, applied to [H_x | r] */
                double* v = (double*)malloc(sizeof(double) * (size_t)rows);
                double nrm2 = 0; for (int a = 0; a < rows; ++a) nrm2 += Hj[(size_t)a * ncols + N] * Hj[(size_t)a * ncols + N];
                double nrm = sqrt(nrm2), alpha = Hj[N] >= 0. ? -nrm : nrm;
                for (int a = 0; a < rows; ++a) v[a] = Hj[(size_t)a * ncols + N];
                v[0] -= alpha;
                double vn2 = 0; for (int a = 0; a < rows; ++a) vn2 += v[a] * v[a];
                double beta = vn2 > 0 ? 2. / vn2 : 0.;
                for (int cc = 0; cc < N; ++cc) {
                    double s = 0; for (int a = 0; a < rows; ++a) s += v[a] * Hj[(size_t)a * ncols + cc];
                    s *= beta; if (s == 0.) continue;
                    for (int a = 0; a < rows; ++a) Hj[(size_t)a * ncols + cc] -= s * v[a];
                }
                { double s = 0; for (int a = 0; a < rows; ++a) s += v[a] * rj[a]; s *= beta; for (int a = 0; a < rows; ++a) rj[a] -= s * v[a]; }
                free(v);
                /* row 0 = range (U) row, H_2 = alpha; rows 1.. = null rows */
                memcpy(H1 + (size_t)n_acc * N, Hj, sizeof(double) * (size_t)N); H2[n_acc] = alpha; r1[n_acc] = rj[0];
                for (int a = 1; a < rows; ++a) { memcpy(Hn_top + (size_t)top * N, Hj + (size_t)a * ncols, sizeof(double) * (size_t)N); rn_top[top++] = rj[a]; }
                acc_ids[n_acc++] = f->id;
            } else {
                f->in_state = 0;
            }
            free(Hj); free(rj); free(Hm); free(rm);
        }
        /* feature_states keeps only the accepted new features, in order */
        e->n_fs = n_fs_old;
        for (int i = 0; i < n_acc; ++i) e->feature_states[e->n_fs++] = acc_ids[i];
        /* ---- tracked in-state features (2 rows each) */
        double* He = (double*)calloc((size_t)(2 * n_ekf + 1) * N, sizeof(double)); double* re = (double*)calloc((size_t)2 * n_ekf + 1, sizeof(double)); int rows_e = 0;
        for (int i = 0; i < n_ekf; ++i) {
            feat_t* f = map_find(e, ekf_ids[i]);
            double Hj[2 * 1024]; double* Hd = (N <= 1024) ? Hj : (double*)malloc(sizeof(double) * 2 * (size_t)N); double rj[2];
            int fcol = LEG + 6 * e->n_clones + fs_rank(e, f->id);
            int64_t sid = e->imu_id;
            feature_jacobian_ekf(e, f, &sid, 1, 0, N, fcol, Hd, rj);
            if (gating_test(e, Hd, rj, 2, 2)) { memcpy(He + (size_t)rows_e * N, Hd, sizeof(double) * 2 * (size_t)N); re[rows_e] = rj[0]; re[rows_e + 1] = rj[1]; rows_e += 2; }
            if (Hd != Hj) free(Hd);
        }
        if (rows_e > N) { lvo_qr_compress(He, re, rows_e, N); rows_e = N; }
        /* ---- MSCKF features */
        double* Hm = (double*)calloc((size_t)(rows_msckf + 1) * N, sizeof(double)); double* rm = (double*)calloc((size_t)rows_msckf + 1, sizeof(double)); int rows_m = 0;
        for (int i = 0; i < n_msckf; ++i) {
            feat_t* f = map_find(e, msckf[i]);
            double* Hj = (double*)calloc((size_t)(2 * f->n_obs) * N, sizeof(double)); double* rj = (double*)calloc((size_t)2 * f->n_obs, sizeof(double));
            int k = feature_jacobian_msckf(e, f, f->sid, f->n_obs, Hj, rj);
            if (gating_test(e, Hj, rj, k, 2 * f->n_obs - 3)) { memcpy(Hm + (size_t)rows_m * N, Hj, sizeof(double) * (size_t)k * N); memcpy(rm + rows_m, rj, sizeof(double) * (size_t)k); rows_m += k; }
            free(Hj); free(rj);
        }
        const int nc = LEG + 6 * e->n_clones;
        if (rows_m > nc) {
            /* compress the (rows x nc) left part; the feature columns of MSCKF rows are zero (larvio.cpp:2185-2229) */
            double* Hc = (double*)malloc(sizeof(double) * (size_t)rows_m * nc);
            for (int a = 0; a < rows_m; ++a) memcpy(Hc + (size_t)a * nc, Hm + (size_t)a * N, sizeof(double) * (size_t)nc);
            lvo_qr_compress(Hc, rm, rows_m, nc);
            memset(Hm, 0, sizeof(double) * (size_t)nc * N);
            for (int a = 0; a < nc; ++a) memcpy(Hm + (size_t)a * N, Hc + (size_t)a * nc, sizeof(double) * (size_t)nc);
            rows_m = nc;
            free(Hc);
        }
        /* ---- measurementUpdate_hybrid (larvio.cpp:1605-1862) */
        const int m = rows_m + rows_e + top;
        if (m + n_acc > 0) {
            double* Ho = (double*)malloc(sizeof(double) * (size_t)(m + 1) * N); double* ro = (double*)malloc(sizeof(double) * (size_t)(m + 1));
            memcpy(Ho, Hm, sizeof(double) * (size_t)rows_m * N); memcpy(ro, rm, sizeof(double) * (size_t)rows_m);
            memcpy(Ho + (size_t)rows_m * N, He, sizeof(double) * (size_t)rows_e * N); memcpy(ro + rows_m, re, sizeof(double) * (size_t)rows_e);
            memcpy(Ho + (size_t)(rows_m + rows_e) * N, Hn_top, sizeof(double) * (size_t)top * N); memcpy(ro + rows_m + rows_e, rn_top, sizeof(double) * (size_t)top);
            double* dx = (double*)calloc((size_t)N + n_acc + 1, sizeof(double));
            /* the pre-update P is needed for nothing else: K, dx_leg and (I-KH)P all come from lvo_ekf_update */
            lvo_ekf_update(e->P, N, N, Ho, m, ro, e->sigma2, dx);
            /* delayed initialisation: HH = H_2^-1 H_1 (diag), dx_new = -HH dx_leg + H_2^-1 r_1 */
            delayed_init_dx(N, n_acc, H1, H2, r1, dx);
            {   /* inject with the new features already in feature_states (their dx index is N + j == base + i) */
                inject(e, dx, n_fs_old);
            }
            if (n_acc > 0) {
                double* Q = delayed_init_cov(e->P, N, n_acc, H1, H2, e->sigma2);
                free(e->P); e->P = Q; e->N = N + n_acc;
            }
            e->last_update_time = e->s.t;
            e->counters[0]++; e->counters[2] = m;
            free(Ho); free(ro); free(dx);
        }
        free(Hn_top); free(rn_top); free(H1); free(H2); free(r1); free(acc_ids); free(He); free(re); free(Hm); free(rm);
    } else {
        for (int i = 0; i < n_msckf; ++i) { feat_t* f = map_find(e, msckf[i]); if (f) f->is_initialized = 0; }
    }
    for (int i = 0; i < n_msckf; ++i) map_erase(e, msckf[i]);
done:
    free(ekf_ids); free(ekf_lost); free(invalid); free(msckf); free(ekf_new);
}

/* ------------------------------------------------------------------------ pruning */
static void find_redundant(lvo_ekf* e, int64_t* rm)
{   /* larvio.cpp:2259-2307 */
    int key = e->n_clones - 4, si = key + 1, fi = 0, n = 0;
    double Rk[9]; quat_to_rot(e->clones[key].q_cam, Rk);
    for (int i = 0; i < 2; ++i) {
        const lvo_clone* c = &e->clones[si];
        double R[9], Rt[9], M[9], q[4];
        quat_to_rot(c->q_cam, R); m3_t(R, Rt); m3_mul(Rt, Rk, M);
        double d[3] = {c->p_cam[0] - e->clones[key].p_cam[0], c->p_cam[1] - e->clones[key].p_cam[1], c->p_cam[2] - e->clones[key].p_cam[2]};
        double distance = v3_norm(d);
        rot_to_quat(M, q);
        double angle = 2 * atan2(sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), fabs(q[3]));     /* AngleAxisd(R).angle() */
        if (angle < e->cfg.rotation_threshold && distance < e->cfg.translation_threshold && e->tracking_rate > e->cfg.tracking_rate_threshold) {
            rm[n++] = c->id; ++si;
        } else {
            rm[n++] = e->clones[fi].id; ++fi; si -= 2;
        }
    }
    if (rm[0] > rm[1]) { int64_t t = rm[0]; rm[0] = rm[1]; rm[1] = t; }
    if (e->trace) {
        TR(e, "REDUNDANT %d %.17g %.17g %.17g %.17g", e->n_clones, e->tracking_rate, e->cfg.rotation_threshold, e->cfg.translation_threshold, e->cfg.tracking_rate_threshold);
        for (int i = 0; i < e->n_clones; ++i) { const lvo_clone* c = &e->clones[i];
            TR(e, " %lld %.17g %.17g %.17g %.17g %.17g %.17g %.17g", (long long)c->id, c->q_cam[0], c->q_cam[1], c->q_cam[2], c->q_cam[3], c->p_cam[0], c->p_cam[1], c->p_cam[2]); }
        TR(e, " -> %lld %lld\n", (long long)rm[0], (long long)rm[1]);
    }
}

static int64_t get_new_anchor_id(lvo_ekf* e, feat_t* f, const int64_t* rm, int nrm)
{   /* larvio.cpp:3412-3472 */
    const int size = e->n_clones;
    if (size <= 2) return e->clones[size - 1].id;
    int valid = 0; double min_dis = 99999; int64_t id_min = 0;
    for (int i = 0; i < size - 2; ++i) {
        const lvo_clone* c = &e->clones[i];
        int oi = feat_obs_find(f, c->id);
        if (oi < 0) continue;
        int removed = 0; for (int k = 0; k < nrm; ++k) if (rm[k] == c->id) removed = 1;
        if (removed) continue;
        double R[9], d[3] = {f->position[0] - c->p_cam[0], f->position[1] - c->p_cam[1], f->position[2] - c->p_cam[2]}, pn[3];
        quat_to_rot(c->q_cam, R); m3t_v(R, d, pn);
        double a = pn[0] / pn[2] - f->z[oi][0], b = pn[1] / pn[2] - f->z[oi][1];
        double dis = sqrt(a * a + b * b);
        if (min_dis > dis) { min_dis = dis; id_min = c->id; valid = 1; }
    }
    if (e->trace) {
        TR(e, "ANCHOR %lld %.17g %.17g %.17g %d", (long long)f->id, f->position[0], f->position[1], f->position[2], nrm);
        for (int k = 0; k < nrm; ++k) TR(e, " %lld", (long long)rm[k]);
        TR(e, " %d", size);
        for (int i = 0; i < size; ++i) { const lvo_clone* c = &e->clones[i]; int oi = feat_obs_find(f, c->id);
            TR(e, " %lld %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %.17g %.17g", (long long)c->id, c->q_cam[0], c->q_cam[1], c->q_cam[2], c->q_cam[3], c->p_cam[0], c->p_cam[1], c->p_cam[2],
               oi >= 0, oi >= 0 ? f->z[oi][0] : 0.0, oi >= 0 ? f->z[oi][1] : 0.0); }
        TR(e, " -> %lld\n", (long long)(valid ? id_min : e->clones[size - 1].id));
    }
    return valid ? id_min : e->clones[size - 1].id;
}

static void update_feature_cov_1d(lvo_ekf* e, const feat_t* f, int64_t old_id, int64_t new_id)
{   /* larvio.cpp:3125-3293: row/col of the feature replaced by J P, J P J^T */
    const int N = e->N;
    const lvo_clone* co = &e->clones[clone_rank(e, old_id)];
    const lvo_clone* cn = &e->clones[clone_rank(e, new_id)];
    const double* R_b2c = e->R_b2c; const double* t_c_b = e->t_c_b; const double* p_w = f->position;
    double R_b2w_old[9], R_c2w_old[9], R_w2c_old[9];
    quat_to_rot(co->q, R_b2w_old); quat_to_rot(co->q_cam, R_c2w_old); m3_t(R_c2w_old, R_w2c_old);
    double d[3] = {p_w[0] - co->p_cam[0], p_w[1] - co->p_cam[1], p_w[2] - co->p_cam[2]}, p_old_[3], p_old[3];
    m3_v(R_w2c_old, d, p_old_);
    if (e->if_fej) {
        double dd[3] = {f->position_fej[0] - co->p_fej[0], f->position_fej[1] - co->p_fej[1], f->position_fej[2] - co->p_fej[2]}, q[3];
        m3t_v(R_b2w_old, dd, q); q[0] -= t_c_b[0]; q[1] -= t_c_b[1]; q[2] -= t_c_b[2];
        m3_v(R_b2c, q, p_old);
    } else memcpy(p_old, p_old_, 24);
    const double inv_old = 1 / p_old_[2];
    const double f_old[3] = {p_old_[0] / p_old_[2], p_old_[1] / p_old_[2], 1};
    double R_b2w_new[9], R_w2b_new[9], R_c2w_new[9], R_w2c_new[9];
    quat_to_rot(cn->q, R_b2w_new); m3_t(R_b2w_new, R_w2b_new); quat_to_rot(cn->q_cam, R_c2w_new); m3_t(R_c2w_new, R_w2c_new);
    const double inv_new = f->inv_depth;
    double pbo[3], pbn[3];
    for (int i = 0; i < 3; ++i) {
        pbo[i] = e->if_fej ? f->position_fej[i] - co->p_fej[i] : p_w[i] - co->p[i];
        pbn[i] = e->if_fej ? f->position_fej[i] - cn->p_fej[i] : p_w[i] - cn->p[i];
    }
    const double J_rho_d_new = -inv_new * inv_new;
    double M[9], Jd_[3]; m3_mul(R_w2c_new, R_c2w_old, M); m3_v(M, f_old, Jd_);
    double So[9], Sn[9], Jto[9], Jtn[9];
    skew3(pbo, So); skew3(pbn, Sn); m3_mul(R_w2c_new, So, Jto); m3_mul(R_w2c_new, Sn, Jtn);
    double v1[3], SkewMx[9], RR[9], R_c2b[9], v2[3], S2[9], Mx[9], D[9], JeT[9], E[9], JeP[9];
    m3_v(R_w2b_new, pbn, v1); v1[0] -= t_c_b[0]; v1[1] -= t_c_b[1]; v1[2] -= t_c_b[2]; skew3(v1, SkewMx);
    m3_mul(R_w2b_new, R_b2w_old, RR);
    m3_t(R_b2c, R_c2b); m3_v(R_c2b, p_old, v2); skew3(v2, S2); m3_mul(RR, S2, Mx);
    for (int i = 0; i < 9; ++i) D[i] = SkewMx[i] - Mx[i];
    m3_mul(R_b2c, D, JeT);
    for (int i = 0; i < 9; ++i) E[i] = RR[i] - ((i % 4 == 0) ? 1.0 : 0.0);
    m3_mul(R_b2c, E, JeP);
    const double J_d_rho_old = -1 / (inv_old * inv_old);
    double* J = (double*)calloc((size_t)N, sizeof(double));
    const int fc = LEG + 6 * e->n_clones + fs_rank(e, f->id);
    const int oc = LEG + 6 * clone_rank(e, old_id), ncn = LEG + 6 * clone_rank(e, new_id);
    J[fc] = J_rho_d_new * Jd_[2] * J_d_rho_old;
    for (int j = 0; j < 3; ++j) {
        J[oc + j] = J_rho_d_new * (-Jto[6 + j]); J[oc + 3 + j] = J_rho_d_new * R_w2c_new[6 + j];
    }
    for (int j = 0; j < 3; ++j) {   /* assignment order as the reference: old, then new (they may coincide? never: old != new) */
        J[ncn + j] = J_rho_d_new * Jtn[6 + j]; J[ncn + 3 + j] = J_rho_d_new * (-R_w2c_new[6 + j]);
    }
    for (int j = 0; j < 3; ++j) { J[15 + j] = J_rho_d_new * JeT[6 + j]; J[18 + j] = J_rho_d_new * JeP[6 + j]; }
    double* Pf = (double*)malloc(sizeof(double) * (size_t)N);
    for (int b = 0; b < N; ++b) { double s = 0; for (int k = 0; k < N; ++k) s += J[k] * e->P[(size_t)k * N + b]; Pf[b] = s; }
    double Pff = 0; for (int k = 0; k < N; ++k) Pff += Pf[k] * J[k];
    for (int b = 0; b < N; ++b) if (b != fc) { e->P[(size_t)fc * N + b] = Pf[b]; e->P[(size_t)b * N + fc] = Pf[b]; }
    e->P[(size_t)fc * N + fc] = Pff;
    P_symmetrize(e->P, N);
    free(J); free(Pf);
}

static void prune_imu_state_buffer(lvo_ekf* e)
{   /* larvio.cpp:2310-2641 */
    int64_t rm[2]; int nrm = 0;
    if (!e->if_zupt) {
        if (e->n_clones < e->cfg.sw_size) return;
        find_redundant(e, rm); nrm = 2;
    } else { rm[0] = e->imu_id - 1; nrm = 1; }
    int rows = 0; int64_t* used = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e->n_map + 1)); int n_used = 0;
    for (int i = 0; i < e->n_map; ++i) {
        feat_t* f = e->map[i];
        int64_t inv[2]; int ninv = 0;
        for (int k = 0; k < nrm; ++k) if (feat_obs_find(f, rm[k]) >= 0) inv[ninv++] = rm[k];
        if (ninv == 0) continue;
        int anchor_involved = 0; for (int k = 0; k < ninv; ++k) if (inv[k] == f->id_anchor) anchor_involved = 1;
        if (f->in_state) {
            if (anchor_involved) {
                int64_t new_id = get_new_anchor_id(e, f, inv, ninv);
                const lvo_clone* cn = &e->clones[clone_rank(e, new_id)];
                double R[9], d[3] = {f->position[0] - cn->p_cam[0], f->position[1] - cn->p_cam[1], f->position[2] - cn->p_cam[2]}, pn[3];
                quat_to_rot(cn->q_cam, R); m3t_v(R, d, pn);
                f->inv_depth = 1 / pn[2];
                f->obs_anchor[0] = pn[0] / pn[2]; f->obs_anchor[1] = pn[1] / pn[2];
                update_feature_cov_1d(e, f, f->id_anchor, new_id);
                f->id_anchor = new_id;
            }
        } else {
            if (f->is_initialized && anchor_involved) {
                int64_t new_id = get_new_anchor_id(e, f, inv, ninv);
                const lvo_clone* cn = &e->clones[clone_rank(e, new_id)];
                double R[9], d[3] = {f->position[0] - cn->p_cam[0], f->position[1] - cn->p_cam[1], f->position[2] - cn->p_cam[2]}, pn[3];
                quat_to_rot(cn->q_cam, R); m3t_v(R, d, pn);
                f->inv_depth = 1 / pn[2];
                int oi = feat_obs_find(f, new_id);
                if (oi >= 0) { f->obs_anchor[0] = f->z[oi][0]; f->obs_anchor[1] = f->z[oi][1]; }
                else { f->obs_anchor[0] = 0; f->obs_anchor[1] = 0; }        /* std::map operator[] default-inserts (0,0) */
                f->id_anchor = new_id;
            }
            if (!e->if_zupt && !f->ekf_feature && ninv > 1) {
                int tracked = feat_obs_find(f, e->imu_id) >= 0;
                if (!f->is_initialized) {
                    if (!feat_check_motion(e, f, tracked)) continue;
                    if (!feat_initialize(e, f, 1)) continue;
                }
                used[n_used++] = f->id;
                rows += 2 * ninv - 3;
            }
        }
    }
    if (!e->if_zupt && n_used != 0) {
        const int N = e->N;
        double* H = (double*)calloc((size_t)(rows + 1) * N, sizeof(double)); double* r = (double*)calloc((size_t)rows + 1, sizeof(double)); int stack = 0;
        for (int i = 0; i < e->n_map; ++i) {
            feat_t* f = e->map[i];
            int64_t inv[2]; int ninv = 0;
            for (int k = 0; k < nrm; ++k) if (feat_obs_find(f, rm[k]) >= 0) inv[ninv++] = rm[k];
            int is_used = 0; for (int k = 0; k < n_used; ++k) if (used[k] == f->id) is_used = 1;
            if (is_used) {
                double Hj[4 * 2048]; double rj[4];
                double* Hd = (N <= 2048) ? Hj : (double*)malloc(sizeof(double) * 4 * (size_t)N);
                int k = feature_jacobian_msckf(e, f, inv, ninv, Hd, rj);
                if (gating_test(e, Hd, rj, k, 2 * ninv - 3)) { memcpy(H + (size_t)stack * N, Hd, sizeof(double) * (size_t)k * N); memcpy(r + stack, rj, sizeof(double) * (size_t)k); stack += k; }
                if (Hd != Hj) free(Hd);
            }
            for (int k = 0; k < ninv; ++k) feat_obs_erase(f, inv[k]);
        }
        update_msckf(e, H, r, stack);
        free(H); free(r);
    } else {
        for (int i = 0; i < e->n_map; ++i) for (int k = 0; k < nrm; ++k) feat_obs_erase(e->map[i], rm[k]);
    }
    for (int k = 0; k < nrm; ++k) {
        int seq = clone_rank(e, rm[k]);
        if (seq < 0) continue;
        P_delete(e, LEG + 6 * seq, 6);
        memmove(e->clones + seq, e->clones + seq + 1, sizeof(lvo_clone) * (size_t)(e->n_clones - seq - 1));
        e->n_clones--;
    }
    free(used);
}

/* ------------------------------------------------------------------------ ZUPT */
static int cmp_dbl(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }
static void update_zupt(lvo_ekf* e)
{   /* measurementUpdate_ZUPT_vpq (larvio.cpp:2791-2962): 9 rows, R = diag(v,p,q noise) */
    const int N = e->N, n = e->n_clones;
    double* H = (double*)calloc((size_t)9 * N, sizeof(double)); double r[9];
    for (int i = 0; i < 3; ++i) {
        H[(size_t)i * N + 3 + i] = 1.0;
        H[(size_t)(3 + i) * N + LEG + 6 * n - 3 + i] = 1.0; H[(size_t)(3 + i) * N + LEG + 6 * n - 9 + i] = -1.0;
        H[(size_t)(6 + i) * N + LEG + 6 * n - 6 + i] = -0.5; H[(size_t)(6 + i) * N + LEG + 6 * n - 12 + i] = 0.5;
    }
    const lvo_clone* cc = &e->clones[clone_rank(e, e->imu_id)];
    const lvo_clone* cp = &e->clones[clone_rank(e, e->imu_id - 1)];
    for (int i = 0; i < 3; ++i) { r[i] = -e->s.v[i]; r[3 + i] = -(cc->p[i] - cp->p[i]); }
    double qpc[4] = {-cp->q[0], -cp->q[1], -cp->q[2], cp->q[3]}, dq[4];
    quat_mul(cc->q, qpc, dq);
    r[6] = dq[0]; r[7] = dq[1]; r[8] = dq[2];
    /* whiten rows so the shared isotropic update applies: scale row i by sigma/sqrt(R_ii) */
    const double Rd[9] = {e->zupt_v2, e->zupt_v2, e->zupt_v2, e->zupt_p2, e->zupt_p2, e->zupt_p2, e->zupt_q2, e->zupt_q2, e->zupt_q2};
    for (int i = 0; i < 9; ++i) { double s = sqrt(e->sigma2 / Rd[i]); for (int j = 0; j < N; ++j) H[(size_t)i * N + j] *= s; r[i] *= s; }
    double* dx = (double*)malloc(sizeof(double) * (size_t)N);
    lvo_ekf_update(e->P, N, N, H, 9, r, e->sigma2, dx);
    inject(e, dx, e->n_fs);
    free(dx); free(H);
    e->last_update_time = e->s.t; e->last_zupt_time = e->s.t;
    e->counters[3]++;
}
static int check_zupt(lvo_ekf* e)
{   /* larvio.cpp:2751-2788 */
    if (e->n_coarse < 20) { e->n_coarse = 0; return 0; }
    qsort(e->coarse_dis, (size_t)e->n_coarse, sizeof(double), cmp_dbl);
    double max_dis = e->coarse_dis[e->n_coarse - 9];
    e->n_coarse = 0;
    if (max_dis < e->cfg.zupt_max_feature_dis) {
        if (e->n_fs > 0) {
            P_delete(e, e->N - e->n_fs, e->n_fs);
            for (int i = 0; i < e->n_fs; ++i) { feat_t* f = map_find(e, e->feature_states[i]); f->is_initialized = 0; f->ekf_feature = 0; f->in_state = 0; }
            e->n_fs = 0;
        }
        update_zupt(e);
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------ static initializer (StaticInitializer.cpp) */
static int static_try_init(lvo_ekf* e, double ts, const lvo_feature_obs* f, int n, const lvo_imu* imu, int n_imu, int* n_erased)
{
    *n_erased = 0;
    if (e->static_counter == 0) {
        e->static_counter++;
        e->init_ids = (int64_t*)realloc(e->init_ids, sizeof(int64_t) * (size_t)(n + 1)); e->init_uv = (double*)realloc(e->init_uv, sizeof(double) * 2 * (size_t)(n + 1));
        for (int i = 0; i < n; ++i) { e->init_ids[i] = (int64_t)f[i].id; e->init_uv[2 * i] = f[i].u; e->init_uv[2 * i + 1] = f[i].v; }
        e->n_init = n;
        e->lower_time_bound = ts + e->td;
        return 0;
    }
    double* dis = (double*)malloc(sizeof(double) * (size_t)(n + 1)); int nd = 0;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < e->n_init; ++k) if (e->init_ids[k] == (int64_t)f[i].id) {
            double dx = f[i].u - e->init_uv[2 * k], dy = f[i].v - e->init_uv[2 * k + 1];
            dis[nd++] = sqrt(dx * dx + dy * dy); break;
        }
    if (nd < 20) { e->static_counter = 0; free(dis); return 0; }
    qsort(dis, (size_t)nd, sizeof(double), cmp_dbl);
    double max_dis = dis[nd - 19];
    free(dis);
    if (max_dis < e->cfg.zupt_max_feature_dis) {
        e->static_counter++;
        e->init_ids = (int64_t*)realloc(e->init_ids, sizeof(int64_t) * (size_t)(n + 1)); e->init_uv = (double*)realloc(e->init_uv, sizeof(double) * 2 * (size_t)(n + 1));
        for (int i = 0; i < n; ++i) { e->init_ids[i] = (int64_t)f[i].id; e->init_uv[2 * i] = f[i].u; e->init_uv[2 * i + 1] = f[i].v; }
        e->n_init = n;
        if (e->static_counter < e->static_num) return 0;
    } else { e->static_counter = 0; return 0; }
    /* initializeGravityAndBias */
    const double time_bound = ts + e->td;
    double sw[3] = {0, 0, 0}, sa[3] = {0, 0, 0}; int cnt = 0; double last_t = 0;
    for (int i = 0; i < n_imu; ++i) {
        if (imu[i].t < e->lower_time_bound) continue;
        if (imu[i].t > time_bound) break;
        {   /* Tg (w - As Ma a) and Ma a (StaticInitializer.cpp:84-85): the identity / zero matrices of a filter that does not calibrate them change no bit */
            double la[3], t3[3], w[3], ga[3];
            m3_v(e->Ma, imu[i].acc, la); m3_v(e->As, la, t3);
            for (int k = 0; k < 3; ++k) w[k] = imu[i].gyro[k] - t3[k];
            m3_v(e->Tg, w, ga);
            for (int k = 0; k < 3; ++k) { sw[k] += ga[k]; sa[k] += la[k]; }
        }
        cnt++; last_t = imu[i].t;
    }
    double gi[3];
    for (int k = 0; k < 3; ++k) { e->s.bg[k] = sw[k] / cnt; gi[k] = sa[k] / cnt; }
    const double gn = v3_norm(gi);
    {   /* Quaterniond::FromTwoVectors(gravity_imu, (0,0,gn)) [upstream Eigen] */
        double v0[3] = {gi[0] / gn, gi[1] / gn, gi[2] / gn}, v1[3] = {0, 0, 1.0};
        double cdot = v1[0] * v0[0] + v1[1] * v0[1] + v1[2] * v0[2];
        double ax[3] = {v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]};
        double s = sqrt((1 + cdot) * 2), invs = 1 / s;
        e->s.q[0] = ax[0] * invs; e->s.q[1] = ax[1] * invs; e->s.q[2] = ax[2] * invs; e->s.q[3] = s * 0.5;
    }
    e->s.t = last_t;
    memset(e->s.p, 0, 24); memset(e->s.v, 0, 24); memset(e->s.ba, 0, 24);
    /* assignInitialState */
    int useful = 0;
    for (int i = 0; i < n_imu; ++i) { if (imu[i].t > last_t) break; useful++; }
    if (useful >= n_imu) useful--;
    memcpy(e->m_gyro_old, imu[useful].gyro, 24); memcpy(e->m_acc_old, imu[useful].acc, 24);
    *n_erased = useful;
    return 1;
}

/* ------------------------------------------------------------------------ processFeatures */
int lvo_ekf_process(lvo_ekf* e, double ts, const lvo_feature_obs* feats, int n_feats, const lvo_imu* imu, int n_imu, int* n_consumed)
{   /* larvio.cpp:363-461 */
    *n_consumed = 0;
    if (!e->b_first_features) {
        if (n_imu > 0 && imu[0].t - ts - e->td <= 0.0) e->b_first_features = 1;
        else return 0;
    }
    int off = 0;
    if (!e->is_gravity_set) {
        int erased = 0;
        if (static_try_init(e, ts, feats, n_feats, imu, n_imu, &erased)) {
            e->is_gravity_set = 1;
            e->take_off_stamp = e->s.t; e->last_zupt_time = e->s.t; e->last_update_time = e->s.t;
            e->s_fej_now = e->s;
            off = erased;
        } else return 0;
    }
    int used = batch_imu(e, ts + e->td, imu + off, n_imu - off);
    *n_consumed = off + used;
    add_observations(e, feats, n_feats);
    state_augmentation(e);
    if (e->cfg.if_zupt_valid) e->if_zupt = check_zupt(e);
    remove_lost_features(e);
    prune_imu_state_buffer(e);
    if (e->cfg.if_fej && !e->if_fej && e->s.t - e->take_off_stamp >= 0) e->if_fej = 1;
    e->counters[6] = e->n_map;
    return 1;
}

/* ------------------------------------------------------------------------ getters */
/* ------------------------------------------------------------------------ stage-level views for the pins in tests/test_oracle_backend.py
 * (no FEJ: the linearisation point is the estimate, so numeric differentiation of the measurement / re-parametrisation applies) */
int lvo_stage_ekf1d_obs_jacobian(const lvo_clone* k, const lvo_clone* a, const double* p_w, double inv_depth, const double* obs_anchor,
                                 const double* z, double* Hf2, double* Ha12, double* Hx12, double* He12, double* r2)
{   /* measurementJacobian_ekf_1didp (larvio.cpp:1117-1244) for ONE observation of an in-state feature anchored in clone a */
    lvo_ekf e; memset(&e, 0, sizeof e); e.leg = 22; e.if_fej = 0;
    feat_t* f = (feat_t*)calloc(1, sizeof(feat_t));
    memcpy(f->position, p_w, 24); memcpy(f->position_fej, p_w, 24);
    f->inv_depth = inv_depth; f->obs_anchor[0] = obs_anchor[0]; f->obs_anchor[1] = obs_anchor[1]; f->obs_anchor[2] = 1.0;
    f->id_anchor = a->id;
    const int ok = ekf_obs_jacobian(&e, f, k, a, z, Hf2, Ha12, Hx12, He12, r2);
    free(f);
    return ok;
}

int lvo_stage_hybrid_update_with_new(double* P /* N x N in, updated in place */, int N, const double* Ho, int m, const double* ro,
                                     const double* H1 /* n_acc x N */, const double* H2 /* n_acc */, const double* r1, int n_acc, double sigma2,
                                     double* P_out /* (N + n_acc)^2 */, double* dx_out /* N + n_acc */)
{   /* measurementUpdate_hybrid with delayed initialisation (larvio.cpp:1605-1862) on given matrices: the update with H_o, then the new
     * features' correction and covariance blocks - the same three calls remove_lost_features makes */
    double* dx = (double*)calloc((size_t)N + n_acc + 1, sizeof(double));
    lvo_ekf_update(P, N, N, Ho, m, ro, sigma2, dx);
    delayed_init_dx(N, n_acc, H1, H2, r1, dx);
    double* Q = delayed_init_cov(P, N, n_acc, H1, H2, sigma2);
    memcpy(P_out, Q, sizeof(double) * (size_t)(N + n_acc) * (N + n_acc)); memcpy(dx_out, dx, sizeof(double) * (size_t)(N + n_acc));
    free(Q); free(dx);
    return 1;
}

int lvo_stage_reanchor_row(const lvo_clone* c_old, const lvo_clone* c_new, const double* R_b2c, const double* t_c_b, const double* p_w,
                           double inv_depth_new, double* J19)
{   /* updateFeatureCov_1didp (larvio.cpp:3125-3293): the row J that maps the error state to the error of the new inverse depth, read
     * back through the covariance it produces.  Layout of J19: [0] old rho, [1..6] old anchor clone, [7..12] new anchor clone,
     * [13..18] extrinsics (rotation, translation).  P = I gives every entry but the feature's own; a second run with one
     * off-diagonal entry set gives that one. */
    lvo_ekf e; memset(&e, 0, sizeof e); e.leg = 22; e.if_fej = 0;
    memcpy(e.R_b2c, R_b2c, 72); memcpy(e.t_c_b, t_c_b, 24);
    lvo_clone cl[2]; cl[0] = *c_old; cl[1] = *c_new;
    if (cl[0].id == cl[1].id) return 0;
    e.clones = cl; e.n_clones = 2;
    int64_t fs[1] = {7}; e.feature_states = fs; e.n_fs = 1;
    const int N = 22 + 12 + 1, fc = N - 1;
    e.N = N; e.P = (double*)malloc(sizeof(double) * (size_t)N * N);
    feat_t* f = (feat_t*)calloc(1, sizeof(feat_t));
    f->id = 7; memcpy(f->position, p_w, 24); memcpy(f->position_fej, p_w, 24); f->inv_depth = inv_depth_new;
    double row0[64], row1[64];
    const double c = 0.125; const int b0 = 22;                      /* couples the feature with the first old-clone column in the second run */
    for (int run = 0; run < 2; ++run) {
        for (int i = 0; i < N * N; ++i) e.P[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
        if (run) { e.P[(size_t)fc * N + b0] = c; e.P[(size_t)b0 * N + fc] = c; }
        update_feature_cov_1d(&e, f, cl[0].id, cl[1].id);
        memcpy(run ? row1 : row0, e.P + (size_t)fc * N, sizeof(double) * (size_t)N);
    }
    J19[0] = (row1[b0] - row0[b0]) / c;                              /* (J P)_b0 = J_b0 + c J_fc */
    for (int j = 0; j < 6; ++j) { J19[1 + j] = row0[22 + j]; J19[7 + j] = row0[28 + j]; J19[13 + j] = row0[15 + j]; }
    free(f); free(e.P);
    return 1;
}

int lvo_ekf_dim(const lvo_ekf* e) { return e->N; }
int lvo_ekf_is_initialized(const lvo_ekf* e) { return e->is_gravity_set; }
void lvo_ekf_get_state(const lvo_ekf* e, double* o)
{
    o[0] = e->s.t; memcpy(o + 1, e->s.q, 32); memcpy(o + 5, e->s.v, 24); memcpy(o + 8, e->s.p, 24); memcpy(o + 11, e->s.bg, 24); memcpy(o + 14, e->s.ba, 24);
    memcpy(o + 17, e->R_b2c, 72); memcpy(o + 26, e->t_c_b, 24); o[29] = e->td;
}
void lvo_ekf_get_cov(const lvo_ekf* e, double* P) { memcpy(P, e->P, sizeof(double) * (size_t)e->N * e->N); }
int lvo_ekf_get_clones(const lvo_ekf* e, lvo_clone* out, int cap) { int n = e->n_clones < cap ? e->n_clones : cap; memcpy(out, e->clones, sizeof(lvo_clone) * (size_t)n); return n; }
int lvo_ekf_get_features(const lvo_ekf* e, int64_t* ids, double* inv_depth, double* pos_w, int cap)
{
    int n = e->n_fs < cap ? e->n_fs : cap;
    for (int i = 0; i < n; ++i) {
        feat_t* f = map_find((lvo_ekf*)e, e->feature_states[i]);
        ids[i] = f->id; inv_depth[i] = f->inv_depth; memcpy(pos_w + 3 * i, f->position, 24);
    }
    return n;
}
void lvo_ekf_counters(const lvo_ekf* e, long* out7) { memcpy(out7, e->counters, sizeof e->counters); }
/* stage-level entry for the tests: ONE step of the static initialiser (StaticInitializer::tryIncInit + assignInitialState) on this handle's
 * counters; on success out8 = state time, q[4], b_g[3] and *n_erased = the IMU samples assignInitialState erases */
int lvo_ekf_static_try_init(lvo_ekf* e, double ts, const lvo_feature_obs* f, int n, const lvo_imu* imu, int n_imu, int* n_erased, double* out8)
{
    const int ok = static_try_init(e, ts, f, n, imu, n_imu, n_erased);
    if (ok) { out8[0] = e->s.t; memcpy(out8 + 1, e->s.q, 32); memcpy(out8 + 5, e->s.bg, 24); }
    return ok;
}
