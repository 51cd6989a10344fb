// Replay harness for the moving-start initialiser (larvio_amd/csrc/be_init.h): reads a recorded start (extrinsics, IMU samples, feature
// messages; written by tests/test_oracle_dynamic_init.py), drives DynInit exactly as lvk_ekf_process does - with the RANSAC stage
// replaced by "every correspondence is an inlier" (host-only; the oracle does the same) - and prints the successful attempt's
// intermediate and final results as one JSON object for the comparison with oracle/dyn_init.py.
#include "../../larvio_amd/csrc/be_init.h"
#include <stdio.h>
using namespace lvk_init;
// stand-in for the RANSAC stage (host-only test): every correspondence is an inlier, the matrix is the normalised 8-point fit of all of them
static bool keep_all(void*, const std::vector<Pt2>& ll, const std::vector<Pt2>& rr, double, double, std::vector<unsigned char>& mask, double* F) { mask.assign(ll.size(), 1); return eight_point(ll, rr, F); }
static void arr(const char* k, const double* v, int n, bool last = false) { printf("\"%s\": [", k); for (int i = 0; i < n; ++i) printf("%s%.17g", i ? ", " : "", v[i]); printf("]%s", last ? "" : ", "); }
// the real RANSAC stage for this host-only harness: the ORACLE's restatement of cv::findFundamentalMat (oracle/liblvo.so; its mask and matrix are
// bit for bit what the product's kernel returns, tests/test_gpu_frontend_stages.py) - loaded at run time when a second argument names the library
#include <dlfcn.h>
struct lvo_pt2f { float x, y; };
typedef int (*lvo_ff_fn)(const lvo_pt2f*, const lvo_pt2f*, int, double, double, unsigned char*, double*);
static lvo_ff_fn g_ff = nullptr;
static bool oracle_ransac(void*, const std::vector<Pt2>& ll, const std::vector<Pt2>& rr, double thresh, double conf, std::vector<unsigned char>& mask, double* F)
{
    const int n = (int)ll.size(); std::vector<lvo_pt2f> a((size_t)n), b((size_t)n);
    for (int i = 0; i < n; ++i) { a[(size_t)i] = {(float)ll[(size_t)i].x, (float)ll[(size_t)i].y}; b[(size_t)i] = {(float)rr[(size_t)i].x, (float)rr[(size_t)i].y}; }
    mask.assign((size_t)n, 0);
    return g_ff(a.data(), b.data(), n, thresh, conf, mask.data(), F) == 1;
}
int main(int argc, char** argv)
{
    FILE* f = argc > 1 ? fopen(argv[1], "r") : nullptr; if (!f) return 2;
    if (argc > 2) { void* h = dlopen(argv[2], RTLD_NOW); g_ff = h ? (lvo_ff_fn)dlsym(h, "lvo_find_fundamental") : nullptr; if (!g_ff) { fprintf(stderr, "cannot load lvo_find_fundamental from %s\n", argv[2]); return 4; } }
    double Rb2c[9], tcb[3], th; int ok = 1;
    for (double& x : Rb2c) ok &= fscanf(f, "%lf", &x) == 1; for (double& x : tcb) ok &= fscanf(f, "%lf", &x) == 1; ok &= fscanf(f, "%lf", &th) == 1;
    int n_imu = 0; ok &= fscanf(f, "%d", &n_imu) == 1; std::vector<lvk_imu> imu((size_t)n_imu);
    for (auto& s : imu) ok &= fscanf(f, "%lf %lf %lf %lf %lf %lf %lf", &s.t, &s.gyro[0], &s.gyro[1], &s.gyro[2], &s.acc[0], &s.acc[1], &s.acc[2]) == 7;
    DynInit d; d.reset(); d.td = 0; d.imu_img_time_th = th; m3_t(Rb2c, d.RIC); memcpy(d.TIC, tcb, 24);
    for (int i = 0; i < 9; ++i) { d.Ma[i] = d.Tg[i] = (i % 4 == 0); d.As[i] = 0; }
    d.ransac = g_ff ? oracle_ransac : keep_all;
    int n_msgs = 0; ok &= fscanf(f, "%d", &n_msgs) == 1;
    if (!ok) return 3;
    for (int m = 0; m < n_msgs; ++m) {
        double ts; int n; if (fscanf(f, "%lf %d", &ts, &n) != 2) return 3;
        std::vector<lvk_feature_obs> o((size_t)n);
        for (auto& x : o) { memset(&x, 0, sizeof x); long long id; if (fscanf(f, "%lld %lf %lf %lf %lf", &id, &x.u, &x.v, &x.u_vel, &x.v_vel) != 5) return 3; x.id = (uint64_t)id; }
        int k = 0; while (k < n_imu && imu[(size_t)k].t < ts + 0.05) ++k;
        int erase = 0;
        if (d.try_init(ts, o.data(), n, imu.data(), k, &erase)) {
            printf("{\"message\": %d, \"erase\": %d, \"l\": %d, \"n_points\": %d, \"state_time\": %.17g, \"scale\": %.17g, ", m, erase, d.diag.l, d.diag.n_points, d.out.state_time, d.diag.scale);
            arr("relR", d.diag.relR, 9); arr("relT", d.diag.relT, 3); arr("g", d.diag.g, 3); arr("q", d.out.q, 4); arr("v", d.out.v, 3); arr("bg", d.out.bg, 3);
            std::vector<double> R, T; for (auto& a : d.diag.sfm_R) R.insert(R.end(), a.begin(), a.end()); for (auto& a : d.diag.sfm_T) T.insert(T.end(), a.begin(), a.end());
            arr("sfm_R", R.data(), (int)R.size()); arr("sfm_T", T.data(), (int)T.size(), true);
            printf("}\n");
            return 0;
        }
    }
    printf("{\"message\": -1}\n");
    return 0;
}
