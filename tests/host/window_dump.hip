// Host-only: drives the PRODUCT's window bookkeeping of the moving-start initialiser (lvk_init::DynInit::add_features / corresponding /
// slide_window's removeBack, larvio_amd/csrc/be_init.h) with a message stream read from a text file - window filling, then a slide per
// message, as when every initialisation attempt fails - and prints, for every full window, the correspondences of every frame with the
// newest one, for the comparison with the reference's own FeatureManager compiled in place (tests/test_oracle_ref_window.py).
// file: "n_msgs td", then per message "n" + n x "id u v u_vel v_vel"
#include "../../larvio_amd/csrc/be_init.h"
#include <stdio.h>
using namespace lvk_init;
int main(int argc, char** argv)
{
    FILE* f = argc > 1 ? fopen(argv[1], "r") : nullptr; if (!f) return 2;
    int n_msgs = 0; double td = 0; if (fscanf(f, "%d %lf", &n_msgs, &td) != 2) return 3;
    DynInit d; d.reset();
    for (int m = 0; m < n_msgs; ++m) {
        int n = 0; if (fscanf(f, "%d", &n) != 1) return 3;
        std::vector<lvk_feature_obs> o((size_t)n);
        for (auto& x : o) { memset(&x, 0, sizeof x); long long id; if (fscanf(f, "%lld %lf %lf %lf %lf", &id, &x.u, &x.v, &x.u_vel, &x.v_vel) != 5) return 3; x.id = (uint64_t)id; }
        d.add_features(o.data(), n, td);
        if (d.frame_count == WIN) {
            for (int i = 0; i < WIN; ++i) {
                std::vector<Pt2> a, b; d.corresponding(i, WIN, a, b);
                printf("%d %d %zu", m, i, a.size());
                for (size_t k = 0; k < a.size(); ++k) printf(" %.17g %.17g %.17g %.17g", a[k].x, a[k].y, b[k].x, b[k].y);
                printf("\n");
            }
            d.slide_window();
            printf("%d -1 %zu\n", m, d.tracks.size());
        } else d.frame_count++;
    }
    return 0;
}
