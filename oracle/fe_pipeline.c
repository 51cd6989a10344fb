/*
 * fe_pipeline.c — ORACLE (test infrastructure, see lvo.h): the ImageProcessor state machine,
 * a plain-C restatement of /root/reference/src/image_processor.cpp:130-219 and the functions it
 * calls (initializeFirstFrame :337, initializeFirstFeatures :355, trackFeatures :540,
 * trackNewFeatures :813, findNewFeaturesToBeTracked :1005, getFeatureMsg :1076, publish :1131).
 * The ORCHESTRATION is PINNED to src/image_processor.cpp compiled in place (byte for byte after every frame,
 * tests/test_oracle_ref_imgproc.py; 400 fuzzed streams, one open case: PARITY.md section 2); the image algorithms it calls are not (lvo.h, "PINNING").
 */
#include "lvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, cap;
    lvo_pt2f* prev; lvo_pt2f* curr; lvo_pt2f* init;
    uint64_t* id; int* life; uint8_t* desc;   /* desc: 32 B per track */
} track_set;

static void ts_reserve(track_set* t, int cap)
{
    if (cap <= t->cap) return;
    t->prev = (lvo_pt2f*)realloc(t->prev, sizeof(lvo_pt2f) * (size_t)cap);
    t->curr = (lvo_pt2f*)realloc(t->curr, sizeof(lvo_pt2f) * (size_t)cap);
    t->init = (lvo_pt2f*)realloc(t->init, sizeof(lvo_pt2f) * (size_t)cap);
    t->id = (uint64_t*)realloc(t->id, sizeof(uint64_t) * (size_t)cap);
    t->life = (int*)realloc(t->life, sizeof(int) * (size_t)cap);
    t->desc = (uint8_t*)realloc(t->desc, (size_t)32 * cap);
    t->cap = cap;
}
static void ts_free(track_set* t)
{
    free(t->prev); free(t->curr); free(t->init); free(t->id); free(t->life); free(t->desc);
    memset(t, 0, sizeof *t);
}
/* removeUnmarkedElements (image_processor.h:215-230) applied to the parallel vectors at once */
static void ts_compact(track_set* t, const uint8_t* mask, int mask_valid)
{
    if (!mask_valid) return;                 /* size mismatch => everything is copied */
    int k = 0;
    for (int i = 0; i < t->n; ++i) {
        if (!mask[i]) continue;
        if (k != i) {
            t->prev[k] = t->prev[i]; t->curr[k] = t->curr[i]; t->init[k] = t->init[i];
            t->id[k] = t->id[i]; t->life[k] = t->life[i];
            memcpy(t->desc + (size_t)32 * k, t->desc + (size_t)32 * i, 32);
        }
        ++k;
    }
    t->n = k;
}

struct lvo_frontend {
    lvo_fe_config cfg;
    int image_state;              /* 1 FIRST_IMAGE, 2 SECOND_IMAGE, 3 OTHER_IMAGES */
    uint64_t next_feature_id;
    int b_first_img;
    long pub_counter;
    double last_pub_time, curr_img_time, prev_img_time;
    lvo_pyramid prev_pyr, curr_pyr;
    uint8_t *prev_ext, *prev_blur, *curr_ext, *curr_blur;
    uint8_t* eq_img;
    track_set tr;                 /* pts_ids_/prev_pts_/curr_pts_/pts_lifetime_/init_pts_/vOrbDescriptors */
    int curr_valid;               /* curr_pts_ non-empty this frame */
    lvo_pt2f* new_pts; int n_new, cap_new;
    float H[9];                   /* K * R_Prev2Curr * K^-1 */
    uint64_t lk_point_levels, lk_iterations;
};

lvo_frontend* lvo_frontend_create(const lvo_fe_config* cfg)
{
    lvo_frontend* fe = (lvo_frontend*)calloc(1, sizeof *fe);
    fe->cfg = *cfg;
    fe->image_state = 1;
    const size_t esz = (size_t)(cfg->width + 64) * (cfg->height + 64);
    fe->prev_ext = (uint8_t*)malloc(esz); fe->prev_blur = (uint8_t*)malloc(esz);
    fe->curr_ext = (uint8_t*)malloc(esz); fe->curr_blur = (uint8_t*)malloc(esz);
    fe->eq_img = (uint8_t*)malloc((size_t)cfg->width * cfg->height);
    return fe;
}

void lvo_frontend_destroy(lvo_frontend* fe)
{
    if (!fe) return;
    lvo_pyramid_free(&fe->prev_pyr); lvo_pyramid_free(&fe->curr_pyr);
    free(fe->prev_ext); free(fe->prev_blur); free(fe->curr_ext); free(fe->curr_blur); free(fe->eq_img);
    ts_free(&fe->tr); free(fe->new_pts);
    free(fe);
}

static void lk(lvo_frontend* fe, const lvo_pyramid* a, const lvo_pyramid* b, const lvo_pt2f* p0, lvo_pt2f* p1,
               uint8_t* status, int n)
{
    int nl = a->n_levels < b->n_levels ? a->n_levels : b->n_levels;
    int* it = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1) * nl);
    lvo_lk_track(a, b, p0, p1, status, n, fe->cfg.max_iteration, fe->cfg.track_precision, it);
    for (int i = 0; i < n * nl; ++i) fe->lk_iterations += (uint64_t)it[i];
    fe->lk_point_levels += (uint64_t)n * nl;
    free(it);
}

static void mark_out_of_image(const lvo_frontend* fe, const lvo_pt2f* p, uint8_t* status, int n)
{   /* image_processor.cpp:381-388, 571-578 */
    for (int i = 0; i < n; ++i) {
        if (!status[i]) continue;
        if (p[i].y < 0 || p[i].y > fe->cfg.height - 1 || p[i].x < 0 || p[i].x > fe->cfg.width - 1) status[i] = 0;
    }
}
static void mark_reverse(const lvo_frontend* fe, const lvo_pt2f* back, const lvo_pt2f* orig, uint8_t* status, int n)
{   /* image_processor.cpp:417-429, 630-642: cv::norm(Point2f) is computed in double */
    for (int i = 0; i < n; ++i) {
        if (!status[i]) continue;
        if (back[i].y < 0 || back[i].y > fe->cfg.height - 1 || back[i].x < 0 || back[i].x > fe->cfg.width - 1) { status[i] = 0; continue; }
        float dx = back[i].x - orig[i].x, dy = back[i].y - orig[i].y;
        float dis = (float)sqrt((double)dx * dx + (double)dy * dy);
        if (dis > 1) status[i] = 0;
    }
}

/* The shared body of initializeFirstFeatures (:355-537) and trackNewFeatures (:813-1002):
 * fwd LK -> in-image -> rev LK -> <=1px -> ORB(prev) vs ORB(curr) <= 58 -> undistort -> F-RANSAC.
 * Survivors are returned in `t` (prev, curr, desc = descriptor in the PREVIOUS image).
 * min_stage: initializeFirstFeatures requires >=20 after each stage, trackNewFeatures requires
 * >0 after LK stages and >=20 before RANSAC.  Returns 0 on an early return. */
static int track_fresh(lvo_frontend* fe, const lvo_pt2f* pts, int n, int bootstrap, track_set* t)
{
    ts_reserve(t, n > 0 ? n : 1);
    t->n = n;
    memcpy(t->prev, pts, sizeof(lvo_pt2f) * (size_t)n);
    lvo_apply_homography(fe->H, t->prev, n, t->curr);
    uint8_t* st = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    lk(fe, &fe->prev_pyr, &fe->curr_pyr, t->prev, t->curr, st, n);
    mark_out_of_image(fe, t->curr, st, n);
    ts_compact(t, st, 1);
    if (bootstrap ? t->n < 20 : t->n <= 0) { free(st); return 0; }
    lvo_pt2f* back = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)t->n);
    memcpy(back, t->prev, sizeof(lvo_pt2f) * (size_t)t->n);
    lk(fe, &fe->curr_pyr, &fe->prev_pyr, t->curr, back, st, t->n);
    mark_reverse(fe, back, t->prev, st, t->n);
    free(back);
    ts_compact(t, st, 1);
    if (bootstrap ? t->n < 20 : t->n <= 0) { free(st); return 0; }
    /* descriptors in both images (levels all 0) */
    uint8_t* dcur = (uint8_t*)malloc((size_t)32 * t->n);
    lvo_orb_describe(fe->prev_ext, fe->prev_blur, fe->cfg.width, fe->cfg.height, t->prev, t->n, t->desc, NULL);
    lvo_orb_describe(fe->curr_ext, fe->curr_blur, fe->cfg.width, fe->cfg.height, t->curr, t->n, dcur, NULL);
    for (int i = 0; i < t->n; ++i) st[i] = lvo_hamming256(t->desc + (size_t)32 * i, dcur + (size_t)32 * i) <= 58;
    free(dcur);
    ts_compact(t, st, 1);
    if (t->n < 20) { free(st); return 0; }
    lvo_pt2f* u0 = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)t->n);
    lvo_pt2f* u1 = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)t->n);
    lvo_undistort_points(t->prev, t->n, fe->cfg.intrinsics, fe->cfg.distortion_model, fe->cfg.distortion, fe->cfg.intrinsics, u0);
    lvo_undistort_points(t->curr, t->n, fe->cfg.intrinsics, fe->cfg.distortion_model, fe->cfg.distortion, fe->cfg.intrinsics, u1);
    int mv = lvo_find_fundamental_mask(u0, u1, t->n, 1.0, 0.99, st);
    free(u0); free(u1);
    ts_compact(t, st, mv);
    free(st);
    if (bootstrap ? t->n < 20 : t->n <= 0) return 0;
    return 1;
}

static void clear_tracks(lvo_frontend* fe) { fe->tr.n = 0; fe->curr_valid = 0; }

static void track_features(lvo_frontend* fe)
{   /* image_processor.cpp:540-811.  On entry tr.prev = prev_pts_ (already rotated). */
    track_set* t = &fe->tr;
    if (t->n == 0) return;
    const int n0 = t->n;
    lvo_apply_homography(fe->H, t->prev, n0, t->curr);
    uint8_t* st = (uint8_t*)malloc((size_t)n0);
    lk(fe, &fe->prev_pyr, &fe->curr_pyr, t->prev, t->curr, st, n0);
    mark_out_of_image(fe, t->curr, st, n0);
    ts_compact(t, st, 1);
    if (t->n == 0) { clear_tracks(fe); free(st); return; }
    lvo_pt2f* back = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)t->n);
    memcpy(back, t->prev, sizeof(lvo_pt2f) * (size_t)t->n);
    lk(fe, &fe->curr_pyr, &fe->prev_pyr, t->curr, back, st, t->n);
    mark_reverse(fe, back, t->prev, st, t->n);
    free(back);
    ts_compact(t, st, 1);
    if (t->n == 0) { clear_tracks(fe); free(st); return; }
    /* ORB at the current points against the STORED first-seen descriptor (:677-699) */
    uint8_t* dcur = (uint8_t*)malloc((size_t)32 * t->n);
    lvo_orb_describe(fe->curr_ext, fe->curr_blur, fe->cfg.width, fe->cfg.height, t->curr, t->n, dcur, NULL);
    for (int i = 0; i < t->n; ++i) st[i] = lvo_hamming256(t->desc + (size_t)32 * i, dcur + (size_t)32 * i) <= 58;
    free(dcur);
    ts_compact(t, st, 1);
    if (t->n == 0) { clear_tracks(fe); free(st); return; }
    lvo_pt2f* u0 = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)t->n);
    lvo_pt2f* u1 = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)t->n);
    lvo_undistort_points(t->prev, t->n, fe->cfg.intrinsics, fe->cfg.distortion_model, fe->cfg.distortion, fe->cfg.intrinsics, u0);
    lvo_undistort_points(t->curr, t->n, fe->cfg.intrinsics, fe->cfg.distortion_model, fe->cfg.distortion, fe->cfg.intrinsics, u1);
    int mv = lvo_find_fundamental_mask(u0, u1, t->n, 1.0, 0.99, st);
    free(u0); free(u1);
    ts_compact(t, st, mv);
    free(st);
    if (t->n == 0) { clear_tracks(fe); return; }
    for (int i = 0; i < t->n; ++i) ++t->life[i];        /* :805 */
    fe->curr_valid = 1;
}

static void track_new_features(lvo_frontend* fe)
{   /* image_processor.cpp:813-1002 */
    if (fe->n_new <= 0) return;
    track_set s; memset(&s, 0, sizeof s);
    if (track_fresh(fe, fe->new_pts, fe->n_new, 0, &s)) {
        track_set* t = &fe->tr;
        if (!fe->curr_valid) { /* curr_pts_ was cleared together with every other vector */ }
        ts_reserve(t, t->n + s.n);
        for (int i = 0; i < s.n; ++i) {
            int k = t->n++;
            t->prev[k] = s.prev[i]; t->curr[k] = s.curr[i];
            t->id[k] = fe->next_feature_id++;
            t->life[k] = 2;
            t->init[k] = s.prev[i];
            memcpy(t->desc + (size_t)32 * k, s.desc + (size_t)32 * i, 32);
        }
        fe->curr_valid = 1;
        fe->n_new = 0;                      /* :1001 — only on the success path */
    }
    ts_free(&s);
}

static void find_new_features(lvo_frontend* fe)
{   /* image_processor.cpp:1005-1037 */
    const int w = fe->cfg.width, h = fe->cfg.height, md = fe->cfg.min_distance;
    uint8_t* mask = (uint8_t*)malloc((size_t)w * h);
    memset(mask, 255, (size_t)w * h);
    const int n_curr = fe->curr_valid ? fe->tr.n : 0;
    for (int i = 0; i < n_curr; ++i) {
        /* round() = C round-half-away on the promoted double; int - int */
        int ry = (int)round((double)fe->tr.curr[i].y), rx = (int)round((double)fe->tr.curr[i].x);
        int r0 = ry - md; if (r0 < 0) r0 = 0;
        int r1 = ry + md; if (r1 > h - 1) r1 = h - 1;
        int c0 = rx - md; if (c0 < 0) c0 = 0;
        int c1 = rx + md; if (c1 > w - 1) c1 = w - 1;
        for (int y = r0; y <= r1; ++y) memset(mask + (size_t)y * w + c0, 0, (size_t)(c1 - c0 + 1));
    }
    fe->n_new = 0;
    /* `max_features_num - curr_pts_.size() > 0` is an unsigned comparison (:1034): true unless equal;
     * a negative difference reaches goodFeaturesToTrack as maxCorners <= 0 = unlimited. */
    if (fe->cfg.max_features_num != n_curr) {
        int want = fe->cfg.max_features_num - n_curr;
        int cap = want > 0 ? want : w * h;
        if (cap > fe->cap_new) { fe->new_pts = (lvo_pt2f*)realloc(fe->new_pts, sizeof(lvo_pt2f) * (size_t)cap); fe->cap_new = cap; }
        fe->n_new = lvo_good_features(&fe->curr_pyr, mask, want, 0.01, (double)md, fe->new_pts, cap);
    }
    free(mask);
}

static int get_feature_msg(lvo_frontend* fe, lvo_feature_obs* out, int cap)
{   /* image_processor.cpp:1076-1128 */
    track_set* t = &fe->tr;
    const int n = fe->curr_valid ? t->n : 0;
    if (n == 0) return 0;
    const double unit[4] = {1, 1, 0, 0};
    lvo_pt2f* uc = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)n * 3);
    lvo_pt2f* ui = uc + n; lvo_pt2f* up = ui + n;
    lvo_undistort_points(t->curr, n, fe->cfg.intrinsics, fe->cfg.distortion_model, fe->cfg.distortion, unit, uc);
    lvo_undistort_points(t->init, n, fe->cfg.intrinsics, fe->cfg.distortion_model, fe->cfg.distortion, unit, ui);
    lvo_undistort_points(t->prev, n, fe->cfg.intrinsics, fe->cfg.distortion_model, fe->cfg.distortion, unit, up);
    const double dt_1 = fe->curr_img_time - fe->prev_img_time;
    const int prev_is_last = fe->prev_img_time == fe->last_pub_time;
    const double dt_2 = prev_is_last ? dt_1 : fe->prev_img_time - fe->last_pub_time;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        lvo_feature_obs f; memset(&f, 0, sizeof f);
        f.id = t->id[i];
        f.u = uc[i].x; f.v = uc[i].y;
        f.u_vel = (uc[i].x - up[i].x) / dt_1;
        f.v_vel = (uc[i].y - up[i].y) / dt_1;
        if (t->init[i].x == -1 && t->init[i].y == -1) { f.u_init = -1; f.v_init = -1; }
        else {
            f.u_init = ui[i].x; f.v_init = ui[i].y;
            t->init[i].x = -1; t->init[i].y = -1;
            if (prev_is_last) { f.u_init_vel = (uc[i].x - ui[i].x) / dt_2; f.v_init_vel = (uc[i].y - ui[i].y) / dt_2; }
            else { f.u_init_vel = (up[i].x - ui[i].x) / dt_2; f.v_init_vel = (up[i].y - ui[i].y) / dt_2; }
        }
        if (k < cap) out[k] = f;
        ++k;
    }
    free(uc);
    return k < cap ? k : cap;
}

int lvo_frontend_process(lvo_frontend* fe, const uint8_t* img, int stride, double ts,
                         const lvo_imu* imu, int n_imu, lvo_feature_obs* out, int cap, int* n_out)
{
    const lvo_fe_config* c = &fe->cfg;
    *n_out = 0;
    if (!fe->b_first_img) {                                  /* :134-142 */
        if (n_imu > 0 && imu[0].t - ts <= 0.0) fe->b_first_img = 1;
        else return 0;
    }
    /* createImagePyramids :318-334 */
    lvo_pyramid_free(&fe->curr_pyr);
    if (c->flag_equalize) {
        lvo_clahe_u8(img, c->width, c->height, stride, fe->eq_img, c->width, 3.0, 8, 8);
        lvo_pyramid_build(fe->eq_img, c->width, c->height, c->width, c->patch_size, c->pyramid_levels, &fe->curr_pyr);
    } else {
        lvo_pyramid_build(img, c->width, c->height, stride, c->patch_size, c->pyramid_levels, &fe->curr_pyr);
    }
    lvo_orb_prepare(&fe->curr_pyr, fe->curr_ext, fe->curr_blur);   /* :150 */
    fe->curr_img_time = ts;
    fe->curr_valid = 0;
    int have = 0;
    const double pub_gate = 0.9 * (1.0 / c->pub_frequency);

    if (fe->image_state == 1) {
        /* initializeFirstFrame :337-352 */
        int cap_n = c->max_features_num > 0 ? c->max_features_num : c->width * c->height;
        if (cap_n > fe->cap_new) { fe->new_pts = (lvo_pt2f*)realloc(fe->new_pts, sizeof(lvo_pt2f) * (size_t)cap_n); fe->cap_new = cap_n; }
        fe->n_new = lvo_good_features(&fe->curr_pyr, NULL, c->max_features_num, 0.01, (double)c->min_distance, fe->new_pts, cap_n);
        fe->last_pub_time = ts;
        if (fe->n_new > 20) fe->image_state = 2;
    } else if (fe->image_state == 2) {
        /* initializeFirstFeatures :355-537 */
        lvo_predict_homography(imu, n_imu, fe->prev_img_time, ts, c->R_cam_imu, c->intrinsics, fe->H);
        track_set s; memset(&s, 0, sizeof s);
        if (!track_fresh(fe, fe->new_pts, fe->n_new, 1, &s)) {
            fe->image_state = 1;
        } else {
            track_set* t = &fe->tr;
            ts_reserve(t, s.n);
            t->n = 0;
            for (int i = 0; i < s.n; ++i) {
                int k = t->n++;
                t->prev[k] = s.prev[i]; t->curr[k] = s.curr[i];
                t->init[k].x = -1; t->init[k].y = -1;
                t->id[k] = fe->next_feature_id++;
                t->life[k] = 2;
                memcpy(t->desc + (size_t)32 * k, s.desc + (size_t)32 * i, 32);
            }
            fe->curr_valid = 1;
            fe->n_new = 0;
            if (ts - fe->last_pub_time >= pub_gate) {
                find_new_features(fe);
                *n_out = get_feature_msg(fe, out, cap);
                fe->last_pub_time = ts; fe->pub_counter++;      /* publish :1171-1172 */
                have = 1;
            }
            fe->image_state = 3;
        }
        ts_free(&s);
    } else {
        lvo_predict_homography(imu, n_imu, fe->prev_img_time, ts, c->R_cam_imu, c->intrinsics, fe->H);
        track_features(fe);
        track_new_features(fe);
        if (ts - fe->last_pub_time >= pub_gate) {
            find_new_features(fe);
            *n_out = get_feature_msg(fe, out, cap);
            fe->last_pub_time = ts; fe->pub_counter++;
            have = 1;
        }
    }
    /* :207-216 rotate */
    { lvo_pyramid tmp = fe->prev_pyr; fe->prev_pyr = fe->curr_pyr; fe->curr_pyr = tmp; }
    { uint8_t* p = fe->prev_ext; fe->prev_ext = fe->curr_ext; fe->curr_ext = p; p = fe->prev_blur; fe->prev_blur = fe->curr_blur; fe->curr_blur = p; }
    if (fe->curr_valid) { lvo_pt2f* p = fe->tr.prev; fe->tr.prev = fe->tr.curr; fe->tr.curr = p; }
    /* when curr_pts_ is empty the swap leaves prev_pts_ empty too; the other vectors keep
     * whatever they held (the reference only clears them together, see clear_tracks) */
    else fe->tr.n = 0;
    fe->prev_img_time = ts;
    return have;
}

int lvo_frontend_tracks(const lvo_frontend* fe, uint64_t* ids, lvo_pt2f* pts, int* lifetime,
                        lvo_pt2f* init_pts, uint8_t* desc, int cap)
{
    int n = fe->tr.n < cap ? fe->tr.n : cap;
    for (int i = 0; i < n; ++i) {
        if (ids) ids[i] = fe->tr.id[i];
        if (pts) pts[i] = fe->tr.prev[i];
        if (lifetime) lifetime[i] = fe->tr.life[i];
        if (init_pts) init_pts[i] = fe->tr.init[i];
        if (desc) memcpy(desc + (size_t)32 * i, fe->tr.desc + (size_t)32 * i, 32);
    }
    return n;
}
int lvo_frontend_new_pts(const lvo_frontend* fe, lvo_pt2f* pts, int cap)
{
    int n = fe->n_new < cap ? fe->n_new : cap;
    if (pts) memcpy(pts, fe->new_pts, sizeof(lvo_pt2f) * (size_t)n);
    return n;
}
int lvo_frontend_state(const lvo_frontend* fe) { return fe->image_state; }
void lvo_frontend_lk_stats(const lvo_frontend* fe, uint64_t* pl, uint64_t* it) { *pl = fe->lk_point_levels; *it = fe->lk_iterations; }
