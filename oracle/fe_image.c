/*
 * fe_image.c — ORACLE (test infrastructure, see lvo.h): full-image passes of the front-end.
 * PARITY UNPINNED against the reference (no golden vectors exist; OpenCV is not vendored).
 * Each function restates the published OpenCV algorithm its call site in
 * /root/reference/src/image_processor.cpp or src/ORBDescriptor.cpp invokes.
 */
#include "lvo.h"
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* All-core leg of the CPU baseline (bench.py cpu_baseline): loops whose iterations are independent (image rows,
 * CLAHE tiles, tracks, key points) may be spread over host threads.  Same arithmetic per iteration, so results are bit-identical
 * for any thread count (tests/test_oracle_frontend.py).  Default 1 = LARVIO's own single-threaded design. */
int lvo_threads_ = 1;
void lvo_set_threads(int n) { lvo_threads_ = n < 1 ? 1 : n; }
int lvo_get_threads(void) { return lvo_threads_; }
#define LVO_PAR _Pragma("omp parallel for schedule(static) num_threads(lvo_threads_) if(lvo_threads_ > 1)")

static inline int cv_round_f(float v) { return (int)rintf(v); }   /* cvRound: half-to-even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) [upstream] */
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p; else p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------------ CLAHE
 * [upstream clahe.cpp: CLAHE_CalcLut_Body, CLAHE_Interpolation_Body, CLAHE_Impl::apply]
 * call site image_processor.cpp:322-325. */
void lvo_clahe_u8(const uint8_t* src, int w, int h, int sstride,
                  uint8_t* dst, int dstride, double clip_limit, int tiles_x, int tiles_y)
{
    const int hist_size = 256;
    int ew = w, eh = h;
    uint8_t* ext = NULL;
    const uint8_t* lut_src = src; int lut_stride = sstride;
    if (!(w % tiles_x == 0 && h % tiles_y == 0)) {
        /* copyMakeBorder(src, ext, 0, ty-(h%ty), 0, tx-(w%tx), REFLECT_101) */
        ew = w + (tiles_x - (w % tiles_x));
        eh = h + (tiles_y - (h % tiles_y));
        ext = (uint8_t*)malloc((size_t)ew * eh);
        for (int y = 0; y < eh; ++y) {
            int sy = reflect101(y, h);
            for (int x = 0; x < ew; ++x) ext[(size_t)y * ew + x] = src[(size_t)sy * sstride + reflect101(x, w)];
        }
        lut_src = ext; lut_stride = ew;
    }
    const int tw = ew / tiles_x, th = eh / tiles_y;
    const int tile_total = tw * th;
    const float lut_scale = (float)(hist_size - 1) / tile_total;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_total / hist_size);
        if (clip < 1) clip = 1;
    }
    uint8_t* lut = (uint8_t*)malloc((size_t)tiles_x * tiles_y * hist_size);
    LVO_PAR
    for (int k = 0; k < tiles_x * tiles_y; ++k) {
        const int ty = k / tiles_x, tx = k % tiles_x;
        int hist[256];
        memset(hist, 0, sizeof hist);
        for (int y = 0; y < th; ++y) {
            const uint8_t* p = lut_src + (size_t)(ty * th + y) * lut_stride + tx * tw;
            for (int x = 0; x < tw; ++x) hist[p[x]]++;
        }
        if (clip > 0) {
            int clipped = 0;
            for (int i = 0; i < hist_size; ++i)
                if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
            int batch = clipped / hist_size;
            int residual = clipped - batch * hist_size;
            for (int i = 0; i < hist_size; ++i) hist[i] += batch;
            if (residual != 0) {
                int step = hist_size / residual; if (step < 1) step = 1;
                for (int i = 0; i < hist_size && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        int sum = 0;
        uint8_t* tl = lut + (size_t)k * hist_size;
        for (int i = 0; i < hist_size; ++i) {
            sum += hist[i];
            tl[i] = sat_u8(cv_round_f((float)sum * lut_scale));
        }
    }
    /* interpolation */
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    LVO_PAR
    for (int y = 0; y < h; ++y) {
        float tyf = y * inv_th - 0.5f;
        int ty1 = cv_floor_f(tyf), ty2 = ty1 + 1;
        float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        const uint8_t* p1 = lut + (size_t)ty1 * tiles_x * hist_size;
        const uint8_t* p2 = lut + (size_t)ty2 * tiles_x * hist_size;
        for (int x = 0; x < w; ++x) {
            float txf = x * inv_tw - 0.5f;
            int tx1 = cv_floor_f(txf), tx2 = tx1 + 1;
            float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            int v = src[(size_t)y * sstride + x];
            int i1 = tx1 * hist_size + v, i2 = tx2 * hist_size + v;
            float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
            dst[(size_t)y * dstride + x] = sat_u8(cv_round_f(res));
        }
    }
    free(lut);
    free(ext);
}

/* ------------------------------------------------------------------------ pyrDown
 * [upstream pyramids.cpp pyrDown_<FixPtCast<uchar,8>>] */
void lvo_pyr_down_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int* rows_all = (int*)malloc(sizeof(int) * (size_t)dw * 5 * (size_t)lvo_threads_);
    LVO_PAR
    for (int y = 0; y < dh; ++y) {
        int* rows = rows_all + (size_t)dw * 5 * (size_t)omp_get_thread_num();
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y - 2 + k, h);
            const uint8_t* s = src + (size_t)sy * sstride;
            int* r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; ++x) {
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = reflect101(2 * x, w),
                    x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                r[x] = s[x0] + s[x4] + 4 * (s[x1] + s[x3]) + 6 * s[x2];
            }
        }
        for (int x = 0; x < dw; ++x) {
            int v = rows[x] + rows[4 * dw + x] + 4 * (rows[dw + x] + rows[3 * dw + x]) + 6 * rows[2 * dw + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows_all);
}

/* ------------------------------------------------------------------------ Scharr
 * [upstream lkpyramid.cpp calcSharrDeriv] Ix = [3 10 3]^T (x) [-1 0 1], Iy transposed, int16. */
void lvo_scharr_deriv(const uint8_t* src, int w, int h, int sstride, int16_t* dst, int dstride)
{
    int* t_all = (int*)malloc(sizeof(int) * (size_t)(w + 2) * 2 * (size_t)lvo_threads_);
    LVO_PAR
    for (int y = 0; y < h; ++y) {
        int* t0 = t_all + (size_t)(w + 2) * 2 * (size_t)omp_get_thread_num();
        int* t1 = t0 + (w + 2);
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * sstride;
        const uint8_t* r1 = src + (size_t)y * sstride;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * sstride;
        int* a = t0 + 1; int* b = t1 + 1;
        for (int x = 0; x < w; ++x) {
            a[x] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            b[x] = r2[x] - r0[x];
        }
        int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        a[-1] = a[x0]; a[w] = a[x1];
        b[-1] = b[x0]; b[w] = b[x1];
        int16_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            d[2 * x]     = (int16_t)(a[x + 1] - a[x - 1]);
            d[2 * x + 1] = (int16_t)((b[x + 1] + b[x - 1]) * 3 + b[x] * 10);
        }
    }
    free(t_all);
}

/* ------------------------------------------------------------------------ LK pyramid
 * [upstream lkpyramid.cpp buildOpticalFlowPyramid]; call site image_processor.cpp:329-333 */
static void pad_reflect101_inplace(uint8_t* buf, int w, int h, int pad, int stride)
{
    /* interior already written at (pad,pad); fill the frame */
    for (int y = -pad; y < h + pad; ++y) {
        int sy = reflect101(y, h);
        uint8_t* d = buf + (size_t)(y + pad) * stride + pad;
        const uint8_t* s = buf + (size_t)(sy + pad) * stride + pad;
        for (int x = -pad; x < w + pad; ++x) {
            if (y >= 0 && y < h && x >= 0 && x < w) { x = w - 1; continue; }
            d[x] = s[reflect101(x, w)];
        }
    }
}

void lvo_pyramid_build(const uint8_t* img, int w, int h, int stride, int win, int max_level, lvo_pyramid* out)
{
    memset(out, 0, sizeof *out);
    out->pad = win;
    int lw = w, lh = h;
    for (int level = 0; level <= max_level && level < LVO_MAX_LEVELS; ++level) {
        const int is = lw + 2 * win, ds = 2 * (lw + 2 * win);
        out->w[level] = lw; out->h[level] = lh;
        out->istride[level] = is; out->dstride[level] = ds;
        out->img[level] = (uint8_t*)calloc((size_t)is * (lh + 2 * win), 1);
        out->der[level] = (int16_t*)calloc((size_t)ds * (lh + 2 * win), sizeof(int16_t));
        uint8_t* roi = out->img[level] + (size_t)win * is + win;
        if (level == 0) {
            for (int y = 0; y < lh; ++y) memcpy(roi + (size_t)y * is, img + (size_t)y * stride, (size_t)lw);
        } else {
            const uint8_t* prev = out->img[level - 1] + (size_t)win * out->istride[level - 1] + win;
            lvo_pyr_down_u8(prev, out->w[level - 1], out->h[level - 1], out->istride[level - 1], roi, is);
        }
        pad_reflect101_inplace(out->img[level], lw, lh, win, is);
        lvo_scharr_deriv(roi, lw, lh, is, out->der[level] + (size_t)win * ds + 2 * win, ds);
        out->n_levels = level + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;
    }
}

void lvo_pyramid_free(lvo_pyramid* p)
{
    for (int i = 0; i < LVO_MAX_LEVELS; ++i) { free(p->img[i]); free(p->der[i]); p->img[i] = NULL; p->der[i] = NULL; }
    p->n_levels = 0;
}

/* ------------------------------------------------------------------------ ORB level-0 mosaic + blur
 * ORBDescriptor.cpp:418-484.  Level 0 only: levels>0 are never sampled (levels vector is
 * all zeros at image_processor.cpp:442,677,910), so their resize+blur is dead work.
 * copyMakeBorder(image, ext, 32.., REFLECT_101) is NOT isolated and `image` is a ROI inside
 * the LK buffer: the first `pad` border pixels are that buffer's own (reflect-101) pixels,
 * the rest reflect about the grown image [upstream copy.cpp copyMakeBorder].
 * GaussianBlur(7x7, sigma 2) on a u8 submatrix, not isolated => generic separable filter
 * with 8-bit fixed-point kernel cvRound(k*256) = {18,34,49,55,49,34,18} and
 * (sum + 2^15) >> 16 [upstream filter.cpp / smooth.cpp]. */
void lvo_orb_prepare(const lvo_pyramid* pyr, uint8_t* ext, uint8_t* blur)
{
    const int w = pyr->w[0], h = pyr->h[0], pad = pyr->pad, B = LVO_ORB_BORDER;
    const int es = w + 2 * B, eh = h + 2 * B;
    const int grow = pad < B ? pad : B;                 /* pixels taken from the parent buffer */
    const int gw = w + 2 * grow, gh = h + 2 * grow;
    LVO_PAR
    for (int y = 0; y < eh; ++y) {
        int gy = reflect101(y - B + grow, gh) - grow;   /* coordinate in level-0 frame, in [-grow, h+grow) */
        const uint8_t* s = pyr->img[0] + (size_t)(gy + pad) * pyr->istride[0] + pad;
        for (int x = 0; x < es; ++x) {
            int gx = reflect101(x - B + grow, gw) - grow;
            ext[(size_t)y * es + x] = s[gx];
        }
    }
    memcpy(blur, ext, (size_t)es * eh);
    static const int K[7] = {18, 34, 49, 55, 49, 34, 18};
    int* rows = (int*)malloc(sizeof(int) * (size_t)w * (h + 6));
    LVO_PAR
    for (int y = -3; y < h + 3; ++y) {
        const uint8_t* s = ext + (size_t)(y + B) * es + B;
        int* r = rows + (size_t)(y + 3) * w;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int k = 0; k < 7; ++k) acc += K[k] * s[x + k - 3];
            r[x] = acc;
        }
    }
    LVO_PAR
    for (int y = 0; y < h; ++y) {
        uint8_t* d = blur + (size_t)(y + B) * es + B;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int k = 0; k < 7; ++k) acc += K[k] * rows[(size_t)(y + k) * w + x];
            d[x] = sat_u8((acc + (1 << 15)) >> 16);
        }
    }
    free(rows);
}

/* ------------------------------------------------------------------------ goodFeaturesToTrack
 * [upstream featureselect.cpp goodFeaturesToTrack, corner.cpp cornerMinEigenVal];
 * call sites image_processor.cpp:343, 1035-1036 (quality 0.01, blockSize 3, Sobel 3).
 * Order fixed by the oracle where OpenCV's depends on its filter engine:
 *   Dx = kc*r(y) + ke*(r(y-1)+r(y+1)),  r = s(x+1)-s(x-1);   Dy = t(y+1)-t(y-1),
 *   t = kc*s(x) + ke*(s(x-1)+s(x+1));  ke=(float)(1/3060.), kc=2*ke;
 *   box3x3 = ((c(x-1)+c(x))+c(x+1)) horizontally, then the same vertically (OpenCV uses
 *   running sums; the oracle uses direct 3-tap sums), reflect-101 on the cov map. */
static inline float px(const lvo_pyramid* p, int x, int y)
{   /* level-0 pixel incl. padding (reflect-101 content) */
    return (float)p->img[0][(size_t)(y + p->pad) * p->istride[0] + x + p->pad];
}

void lvo_min_eigen_map(const lvo_pyramid* pyr, float* eig)
{
    const int w = pyr->w[0], h = pyr->h[0];
    const float ke = (float)(1.0 / 3060.0), kc = 2.0f * ke;
    float* cov = (float*)malloc(sizeof(float) * 3 * (size_t)w * h);
    LVO_PAR
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float r0 = px(pyr, x + 1, y - 1) - px(pyr, x - 1, y - 1);
            float r1 = px(pyr, x + 1, y) - px(pyr, x - 1, y);
            float r2 = px(pyr, x + 1, y + 1) - px(pyr, x - 1, y + 1);
            float dx = kc * r1 + ke * (r0 + r2);
            float t0 = kc * px(pyr, x, y - 1) + ke * (px(pyr, x - 1, y - 1) + px(pyr, x + 1, y - 1));
            float t2 = kc * px(pyr, x, y + 1) + ke * (px(pyr, x - 1, y + 1) + px(pyr, x + 1, y + 1));
            float dy = t2 - t0;
            float* c = cov + 3 * ((size_t)y * w + x);
            c[0] = dx * dx; c[1] = dx * dy; c[2] = dy * dy;
        }
    float* hs = (float*)malloc(sizeof(float) * 3 * (size_t)w * h);
    LVO_PAR
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            for (int c = 0; c < 3; ++c)
                hs[3 * ((size_t)y * w + x) + c] =
                    (cov[3 * ((size_t)y * w + xm) + c] + cov[3 * ((size_t)y * w + x) + c]) + cov[3 * ((size_t)y * w + xp) + c];
        }
    LVO_PAR
    for (int y = 0; y < h; ++y) {
        int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        for (int x = 0; x < w; ++x) {
            float s[3];
            for (int c = 0; c < 3; ++c)
                s[c] = (hs[3 * ((size_t)ym * w + x) + c] + hs[3 * ((size_t)y * w + x) + c]) + hs[3 * ((size_t)yp * w + x) + c];
            float a = s[0] * 0.5f, b = s[1], c2 = s[2] * 0.5f;
            eig[(size_t)y * w + x] = (a + c2) - sqrtf((a - c2) * (a - c2) + b * b);
        }
    }
    free(cov); free(hs);
}

typedef struct { float v; int idx; } cand_t;
static int cand_cmp(const void* pa, const void* pb)
{   /* greaterThanPtr [upstream]: value desc, then address desc */
    const cand_t* a = (const cand_t*)pa; const cand_t* b = (const cand_t*)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return a->idx > b->idx ? -1 : a->idx < b->idx ? 1 : 0;
}

int lvo_good_features(const lvo_pyramid* pyr, const uint8_t* mask, int max_corners,
                      double quality, double min_distance, lvo_pt2f* out, int cap)
{
    const int w = pyr->w[0], h = pyr->h[0];
    float* eig = (float*)malloc(sizeof(float) * (size_t)w * h);
    lvo_min_eigen_map(pyr, eig);
    /* minMaxLoc(eig, 0, &maxVal, 0, 0, mask) */
    int have = 0; float maxv = 0.f;
    for (size_t i = 0; i < (size_t)w * h; ++i)
        if (!mask || mask[i]) { if (!have || eig[i] > maxv) { maxv = eig[i]; have = 1; } }
    double max_val = have ? (double)maxv : 0.0;
    const float thresh = (float)(max_val * quality);
    /* threshold TOZERO, 3x3 dilate, local maxima in the interior */
    cand_t* cands = (cand_t*)malloc(sizeof(cand_t) * (size_t)w * h);
    int nc = 0;
    #define TZ(v) ((v) > thresh ? (v) : 0.f)
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float v = TZ(eig[(size_t)y * w + x]);
            if (v == 0.f) continue;
            if (mask && !mask[(size_t)y * w + x]) continue;
            float m = v;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    float u = TZ(eig[(size_t)(y + dy) * w + x + dx]);
                    if (u > m) m = u;
                }
            if (v == m) { cands[nc].v = v; cands[nc].idx = y * w + x; ++nc; }
        }
    #undef TZ
    qsort(cands, (size_t)nc, sizeof(cand_t), cand_cmp);
    int n_out = 0;
    if (min_distance >= 1) {
        const int cell = (int)rint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        /* grid cells hold accepted corner indices (into out) via linked lists */
        int* head = (int*)malloc(sizeof(int) * (size_t)gw * gh);
        int* next = (int*)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
        for (int i = 0; i < gw * gh; ++i) head[i] = -1;
        lvo_pt2f* acc = (lvo_pt2f*)malloc(sizeof(lvo_pt2f) * (size_t)(nc > 0 ? nc : 1));
        int na = 0;
        const float md2 = (float)(min_distance * min_distance);
        for (int i = 0; i < nc; ++i) {
            int y = cands[i].idx / w, x = cands[i].idx - y * w;
            int xc = x / cell, yc = y / cell;
            int x1 = xc - 1, y1 = yc - 1, x2 = xc + 1, y2 = yc + 1;
            if (x1 < 0) x1 = 0; if (y1 < 0) y1 = 0;
            if (x2 > gw - 1) x2 = gw - 1; if (y2 > gh - 1) y2 = gh - 1;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (int j = head[yy * gw + xx]; j >= 0; j = next[j]) {
                        float dx = x - acc[j].x, dy = y - acc[j].y;
                        if (dx * dx + dy * dy < md2) { good = 0; break; }
                    }
            if (good) {
                acc[na].x = (float)x; acc[na].y = (float)y;
                next[na] = head[yc * gw + xc]; head[yc * gw + xc] = na;
                if (n_out < cap) out[n_out] = acc[na];
                ++na; ++n_out;
                if (max_corners > 0 && na == max_corners) break;
            }
        }
        free(head); free(next); free(acc);
    } else {
        for (int i = 0; i < nc; ++i) {
            int y = cands[i].idx / w, x = cands[i].idx - y * w;
            if (n_out < cap) { out[n_out].x = (float)x; out[n_out].y = (float)y; }
            ++n_out;
            if (max_corners > 0 && n_out == max_corners) break;
        }
    }
    free(cands); free(eig);
    return n_out < cap ? n_out : cap;
}
