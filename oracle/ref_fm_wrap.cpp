// ref_fm_wrap.cpp — ORACLE / TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE's own window bookkeeping of the moving-start
// initialiser (/root/reference/src/feature_manager.cpp: addFeatureCheckParallax :45-97, getCorresponding :100-120, removeBack :203-220,
// removeFront :222-243), compiled where it lies against the Eigen stand-in of ref_shim/ (oracle/Makefile, target `ref`).
#include <Initializer/feature_manager.h>
#include <vector>

using namespace larvio;

extern "C" {

void* lvref_fm_create() { return new FeatureManager(); }
void lvref_fm_destroy(void* h) { delete (FeatureManager*)h; }
// one image: n features (id, u, v, u_vel, v_vel).  Returns addFeatureCheckParallax's answer (true = marginalise the OLDEST frame)
int lvref_fm_add(void* h, int frame_count, int n, const long long* ids, const double* uvv, double td)
{
    MonoCameraMeasurement msg; msg.timeStampToSec = 0;
    for (int i = 0; i < n; ++i) {
        MonoFeatureMeasurement f; f.id = ids[i]; f.u = uvv[4 * i]; f.v = uvv[4 * i + 1]; f.u_vel = uvv[4 * i + 2]; f.v_vel = uvv[4 * i + 3];
        f.u_init = f.v_init = -1; f.u_init_vel = f.v_init_vel = 0;
        msg.features.push_back(f);
    }
    return ((FeatureManager*)h)->addFeatureCheckParallax(frame_count, &msg, td) ? 1 : 0;
}
// getCorresponding(l, r): up to cap pairs (x_l, y_l, x_r, y_r); returns the count
int lvref_fm_corresponding(void* h, int l, int r, double* out4, int cap)
{
    const auto c = ((FeatureManager*)h)->getCorresponding(l, r);
    int n = 0;
    for (const auto& p : c) { if (n < cap) { out4[4 * n] = p.first(0); out4[4 * n + 1] = p.first(1); out4[4 * n + 2] = p.second(0); out4[4 * n + 3] = p.second(1); } ++n; }
    return n;
}
void lvref_fm_remove_back(void* h) { ((FeatureManager*)h)->removeBack(); }
int lvref_fm_feature_count(void* h) { return (int)((FeatureManager*)h)->feature.size(); }

}  // extern "C"
