// larvio_euroc.cpp — the headless form of the reference's dataset driver (/root/reference/app/larvioMain.cpp:27-117, minus
// Pangolin): same command line, same loop, same log files (msckf_2_state.txt / msckf_2_takeoff.txt in the configuration's
// output_dir, written by lvk::LarVio as larvio.cpp:388,446-453 does), running on liblvk_hip.so.
//
//   larvio_euroc path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt] [--max-frames N]
//
// --tum writes "t x y z qx qy qz qw" (body in world, absolute stamps, 17 significant digits) for tools/traj_rmse.py.
// The filter starts with the static initializer (StaticInitializer.cpp); the dynamic (SfM) initializer is outside the hot path.
#include "lvk_dataset.hpp"
#include "lvk_png.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "Usage: %s path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt] [--max-frames N]\n", argv[0]);
        return 1;
    }
    std::string tum_path; long max_frames = -1;
    for (int a = 5; a < argc; ++a) {
        if (!std::strcmp(argv[a], "--tum") && a + 1 < argc) tum_path = argv[++a];
        else if (!std::strcmp(argv[a], "--max-frames") && a + 1 < argc) max_frames = std::atol(argv[++a]);
        else { std::fprintf(stderr, "unknown option %s\n", argv[a]); return 1; }
    }

    // Read sensors (larvioMain.cpp:33-37)
    std::vector<lvk::ImuData> allImuData; std::vector<lvk::ImgInfo> allImgInfo;
    if (!lvk::loadImuFile(argv[1], allImuData) || allImuData.empty()) { std::fprintf(stderr, "cannot read IMU samples from %s\n", argv[1]); return 1; }
    if (!lvk::loadImageList(argv[2], allImgInfo) || allImgInfo.empty()) { std::fprintf(stderr, "cannot read the image list %s\n", argv[2]); return 1; }
    const std::string config_file(argv[4]);

    lvk::Context ctx(0);
    if (!ctx.ok()) { std::fprintf(stderr, "larvio_euroc: %s\n", ctx.error()); return 3; }
    lvk::ImageProcessor ImgProcesser(config_file, ctx.get());                                      // :42-48
    if (!ImgProcesser.initialize()) { std::fprintf(stderr, "Image Processer initialization failed!\n"); return 1; }
    lvk::LarVio Estimator(config_file, ctx.get());                                                  // :50-56
    if (!Estimator.initialize()) { std::fprintf(stderr, "Estimator initialization failed!\n"); return 1; }

    FILE* tum = nullptr;
    if (!tum_path.empty() && !(tum = std::fopen(tum_path.c_str(), "w"))) { std::perror(tum_path.c_str()); return 1; }

    typedef std::chrono::steady_clock Clock;
    double t_fe = 0, t_be = 0, t_io = 0; long n_fe = 0, n_be = 0, n_odo = 0;
    size_t k = 0;
    std::vector<lvk::ImuData> imu_msg_buffer;
    const size_t n_frames = max_frames >= 0 && (size_t)max_frames < allImgInfo.size() ? (size_t)max_frames : allImgInfo.size();
    for (size_t j = 0; j < n_frames; ++j) {
        // get img (:88-95)
        const Clock::time_point t0 = Clock::now();
        const std::string fullPath = std::string(argv[3]) + "/" + allImgInfo[j].imgName;
        lvk::GreyImage image; std::string err;
        if (!lvk::read_png_grey(fullPath, &image, &err)) { std::fprintf(stderr, "%s: %s\n", fullPath.c_str(), err.c_str()); return 1; }
        const double ts = allImgInfo[j].timeStampToSec;
        // get imus (:98-103): everything up to 0.05 s past the image
        while (k < allImuData.size() && allImuData[k].timeStampToSec - ts < 0.05) imu_msg_buffer.push_back(allImuData[k++]);
        const Clock::time_point t1 = Clock::now();

        // process (:106-116)
        lvk::ImageData msg = {ts, image.data.data(), image.width, image.height, image.width};
        lvk::MonoCameraMeasurement features;
        const bool bProcess = ImgProcesser.processImage(msg, imu_msg_buffer, &features);
        const Clock::time_point t2 = Clock::now();
        bool bPubOdo = false;
        if (bProcess) bPubOdo = Estimator.processFeatures(&features, imu_msg_buffer);
        const Clock::time_point t3 = Clock::now();
        t_io += std::chrono::duration<double>(t1 - t0).count();
        t_fe += std::chrono::duration<double>(t2 - t1).count(); ++n_fe;
        if (bProcess) { t_be += std::chrono::duration<double>(t3 - t2).count(); ++n_be; }
        if (bPubOdo) {
            ++n_odo;
            if (tum) {
                double s[30]; lvk_ekf_get_state(Estimator.handle(), s);                            // q stored [x y z w]; p = s[8..10]
                std::fprintf(tum, "%.9f %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", s[0], s[8], s[9], s[10], s[1], s[2], s[3], s[4]);
            }
        }
    }
    if (tum) std::fclose(tum);
    std::printf("frames %ld  feature messages %ld  odometry updates %ld  state dim %d\n", n_fe, n_be, n_odo, lvk_ekf_dim(Estimator.handle()));
    std::printf("front-end %.3f ms/frame   back-end %.3f ms/message   image read+decode %.3f ms/frame\n", n_fe ? 1e3 * t_fe / n_fe : 0.0,
                n_be ? 1e3 * t_be / n_be : 0.0, n_fe ? 1e3 * t_io / n_fe : 0.0);
    if (t_fe + t_be > 0) std::printf("processing rate %.1f frames/s (front-end + back-end, sequential, host buffers)\n", n_fe / (t_fe + t_be));
    return 0;
}
