// larvio_main.cpp — the loop of the reference's app/larvioMain.cpp:84-117 (minus Pangolin and the EuRoC readers) over the C++ host
// classes of include/lvk_larvio.hpp.  Input: one binary sequence file written by examples/make_sequence.py (configuration structs,
// IMU samples, frames); output: the final filter state, one "%.17g" number per field, so that a test can compare it bit for bit
// with the Python-driven run of the same library.
//   build: make -C examples        run: examples/larvio_main <sequence.bin>
#include "lvk_larvio.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace {
template <typename T> bool rd(FILE* f, T* out, size_t n = 1) { return std::fread(out, sizeof(T), n, f) == n; }
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s sequence.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    char magic[8]; int n_frames = 0, w = 0, h = 0, n_imu = 0, init_frame = -1;
    lvk_fe_config fcfg; lvk_ekf_config bcfg; double init[23];      // t, q[4], p[3], v[3], bg[3], ba[3], gyro_old[3], acc_old[3]
    if (!rd(f, magic, 8) || std::string(magic, 7) != "LVKSEQ1" || !rd(f, &n_frames) || !rd(f, &w) || !rd(f, &h) || !rd(f, &n_imu) || !rd(f, &init_frame) ||
        !rd(f, &fcfg) || !rd(f, &bcfg) || !rd(f, init, 23)) { std::fprintf(stderr, "bad header\n"); return 2; }
    std::vector<lvk::ImuData> imu_all((size_t)n_imu);
    if (!rd(f, imu_all.data(), imu_all.size())) { std::fprintf(stderr, "bad imu block\n"); return 2; }

    lvk::Context ctx(0);
    if (!ctx.ok()) { std::fprintf(stderr, "larvio_main: %s\n", ctx.error()); return 3; }       // larvioMain.cpp:44-55: initialize() false => exit
    lvk::ImageProcessor ImgProcesser(fcfg, ctx.get());
    if (!ImgProcesser.initialize()) { std::fprintf(stderr, "Image Processer initialization failed!\n"); return 3; }
    lvk::LarVio Estimator(bcfg, ctx.get());
    if (!Estimator.initialize()) { std::fprintf(stderr, "Estimator initialization failed!\n"); return 3; }

    std::vector<lvk::ImuData> imu_msg_buffer;
    std::vector<uint8_t> img((size_t)w * h);
    size_t next_imu = 0; int n_msgs = 0, n_updates = 0;
    for (int i = 0; i < n_frames; ++i) {
        double ts = 0;
        if (!rd(f, &ts) || !rd(f, img.data(), img.size())) { std::fprintf(stderr, "bad frame %d\n", i); return 2; }
        // larvioMain.cpp:98-102: IMU samples up to 0.05 s past the image
        while (next_imu < imu_all.size() && imu_all[next_imu].timeStampToSec - ts < 0.05) imu_msg_buffer.push_back(imu_all[next_imu++]);
        if (i == init_frame)                               // the initialisers are out of scope: start from the state in the file
            lvk_ekf_set_state(Estimator.handle(), init[0], init + 1, init + 5, init + 8, init + 11, init + 14, init + 17, init + 20);
        lvk::ImageData msg = {ts, img.data(), w, h, w};
        lvk::MonoCameraMeasurement features;
        const bool bProcess = ImgProcesser.processImage(msg, imu_msg_buffer, &features);       // :107
        if (bProcess) {
            ++n_msgs;
            if (Estimator.processFeatures(&features, imu_msg_buffer)) ++n_updates;             // :114
        }
    }
    std::fclose(f);
    double s[30]; lvk_ekf_get_state(Estimator.handle(), s);
    std::printf("frames %d messages %d updates %d dim %d\n", n_frames, n_msgs, n_updates, lvk_ekf_dim(Estimator.handle()));
    std::printf("state");
    for (int k = 0; k < 30; ++k) std::printf(" %.17g", s[k]);
    std::printf("\n");
    double T[16]; Estimator.getTbw(T);
    std::printf("Tbw"); for (int k = 0; k < 16; ++k) std::printf(" %.17g", T[k]); std::printf("\n");
    double Pp[36]; Estimator.getPpose(Pp);
    std::printf("Ppose_diag"); for (int k = 0; k < 6; ++k) std::printf(" %.17g", Pp[7 * k]); std::printf("\n");
    return 0;
}
