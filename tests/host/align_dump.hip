// Host-only: runs the PRODUCT's visual-inertial alignment (lvk_init::visual_imu_alignment with its pre-integrations, larvio_amd/csrc/be_init.h)
// on a window read from a text file and prints ok, the gyro bias, g and x for the comparison with the reference's own
// src/initial_alignment.cpp compiled in place (tests/test_oracle_ref_align.py).
// file: "n_frames", "tic[3] bg0[3]", then per frame: "R[9] T[3]", and for frames >= 1: "acc0[3] gyr0[3] n" + n x "dt acc[3] gyr[3]"
#include "../../larvio_amd/csrc/be_init.h"
#include <stdio.h>
using namespace lvk_init;
int main(int argc, char** argv)
{
    FILE* f = argc > 1 ? fopen(argv[1], "r") : nullptr; if (!f) return 2;
    int n_cases = 0; if (fscanf(f, "%d", &n_cases) != 1) return 3;
    for (int c = 0; c < n_cases; ++c) {
        int nf = 0; if (fscanf(f, "%d", &nf) != 1 || nf < 2 || nf > WIN + 1) return 3;
        double tic[3], bg0[3]; for (double& x : tic) if (fscanf(f, "%lf", &x) != 1) return 3; for (double& x : bg0) if (fscanf(f, "%lf", &x) != 1) return 3;
        std::vector<Frame> frames((size_t)nf); double Bgs[WIN + 1][3];
        for (int i = 0; i <= WIN; ++i) memcpy(Bgs[i], bg0, 24);
        const double zero[3] = {0, 0, 0};
        for (int j = 0; j < nf; ++j) {
            Frame& fr = frames[(size_t)j];
            for (double& x : fr.R) if (fscanf(f, "%lf", &x) != 1) return 3; for (double& x : fr.T) if (fscanf(f, "%lf", &x) != 1) return 3;
            if (j == 0) continue;
            double h[6]; int n = 0; for (double& x : h) if (fscanf(f, "%lf", &x) != 1) return 3; if (fscanf(f, "%d", &n) != 1) return 3;
            fr.pre.start(h, h + 3, zero, bg0); fr.has_pre = true;
            for (int k = 0; k < n; ++k) { double s[7]; for (double& x : s) if (fscanf(f, "%lf", &x) != 1) return 3; fr.pre.push_back(s[0], s + 1, s + 4); }
        }
        std::vector<Frame*> fp; for (auto& fr : frames) fp.push_back(&fr);
        double g[3] = {0, 0, 0}; std::vector<double> x;
        const bool ok = visual_imu_alignment(fp, Bgs, tic, g, x);
        printf("%d %.17g %.17g %.17g %.17g %.17g %.17g %zu", ok ? 1 : 0, Bgs[0][0], Bgs[0][1], Bgs[0][2], g[0], g[1], g[2], x.size());
        for (double v : x) printf(" %.17g", v);
        printf("\n");
    }
    return 0;
}
