// larvio_euroc.cpp — the headless form of the reference's dataset driver (/root/reference/app/larvioMain.cpp:27-117, minus
// Pangolin): same command line, same loop, same log files (msckf_2_state.txt / msckf_2_takeoff.txt in the configuration's
// output_dir, written by lvk::LarVio as larvio.cpp:388,446-453 does), running on liblvk_hip.so.
//
//   larvio_euroc path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt] [--max-frames N] [--pipelined]
//
// --tum writes "t x y z qx qy qz qw" (body in world, absolute stamps, 17 significant digits) for tools/traj_rmse.py.
// --pipelined runs the same loop through lvk::VioPipeline: the filter update of a message overlaps the front-end of the next
// frames on a second HIP stream; the trajectory is the same, written from the filter thread's odometry callback.
// The filter starts with the static initializer (StaticInitializer.cpp); the dynamic (SfM) initializer is outside the hot path.
#include "lvk_dataset.hpp"
#include "lvk_png.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void write_tum(FILE* tum, const lvk::LarVio& Estimator)
{
    double s[30]; lvk_ekf_get_state(Estimator.handle(), s);                                        // q stored [x y z w]; p = s[8..10]
    std::fprintf(tum, "%.9f %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", s[0], s[8], s[9], s[10], s[1], s[2], s[3], s[4]);
}

struct OdometrySink { FILE* tum; long n_odo; };
static void on_odometry(void* user, double, const lvk::LarVio& Estimator)
{
    OdometrySink* o = static_cast<OdometrySink*>(user);
    ++o->n_odo;
    if (o->tum) write_tum(o->tum, Estimator);
}

static int run_pipelined(const char* image_dir, const std::vector<lvk::ImuData>& allImuData, const std::vector<lvk::ImgInfo>& allImgInfo, long max_frames,
                         lvk::ImageProcessor& ImgProcesser, lvk::LarVio& Estimator, FILE* tum)
{
    typedef std::chrono::steady_clock Clock;
    lvk::VioPipeline pipe(ImgProcesser, Estimator);
    if (!pipe.ok()) { std::fprintf(stderr, "cannot create the pipeline\n"); return 1; }
    OdometrySink sink = {tum, 0};
    pipe.onOdometry(on_odometry, &sink);
    const size_t n_frames = max_frames >= 0 && (size_t)max_frames < allImgInfo.size() ? (size_t)max_frames : allImgInfo.size();
    size_t k = 0; double t_proc = 0, t_io = 0; long n_msgs = 0, n_upd = 0;
    for (size_t j = 0; j < n_frames; ++j) {
        const Clock::time_point t0 = Clock::now();
        const std::string fullPath = std::string(image_dir) + "/" + allImgInfo[j].imgName;
        lvk::GreyImage image; std::string err;
        if (!lvk::read_png_grey(fullPath, &image, &err)) { std::fprintf(stderr, "%s: %s\n", fullPath.c_str(), err.c_str()); return 1; }
        const double ts = allImgInfo[j].timeStampToSec;
        const size_t k0 = k;
        while (k < allImuData.size() && allImuData[k].timeStampToSec - ts < 0.05) ++k;
        const Clock::time_point t1 = Clock::now();
        if (k > k0) pipe.pushImu(&allImuData[k0], k - k0);
        lvk::ImageData msg = {ts, image.data.data(), image.width, image.height, image.width};
        pipe.processImage(msg);
        t_io += std::chrono::duration<double>(t1 - t0).count();
        t_proc += std::chrono::duration<double>(Clock::now() - t1).count();
    }
    const Clock::time_point t2 = Clock::now();
    if (!pipe.drain(&n_upd, &n_msgs)) { std::fprintf(stderr, "pipeline: %s\n", "an update failed"); return 1; }
    t_proc += std::chrono::duration<double>(Clock::now() - t2).count();
    if (tum) std::fclose(tum);
    std::printf("frames %zu  feature messages %ld  odometry updates %ld  state dim %d\n", n_frames, n_msgs, sink.n_odo, lvk_ekf_dim(Estimator.handle()));
    std::printf("pipelined: %.3f ms/frame in the driver thread   image read+decode %.3f ms/frame\n", n_frames ? 1e3 * t_proc / n_frames : 0.0, n_frames ? 1e3 * t_io / n_frames : 0.0);
    if (t_proc > 0) std::printf("processing rate %.1f frames/s (pipelined, host buffers)\n", n_frames / t_proc);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "Usage: %s path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt] [--max-frames N] [--pipelined]\n", argv[0]);
        return 1;
    }
    std::string tum_path; long max_frames = -1; bool pipelined = false;
    for (int a = 5; a < argc; ++a) {
        if (!std::strcmp(argv[a], "--tum") && a + 1 < argc) tum_path = argv[++a];
        else if (!std::strcmp(argv[a], "--max-frames") && a + 1 < argc) max_frames = std::atol(argv[++a]);
        else if (!std::strcmp(argv[a], "--pipelined")) pipelined = true;
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
    lvk::Context ctx2(0);                                                                           // the filter's own stream when pipelined
    if (pipelined && !ctx2.ok()) { std::fprintf(stderr, "larvio_euroc: %s\n", ctx2.error()); return 3; }
    lvk::LarVio Estimator(config_file, pipelined ? ctx2.get() : ctx.get());                         // :50-56
    if (!Estimator.initialize()) { std::fprintf(stderr, "Estimator initialization failed!\n"); return 1; }

    FILE* tum = nullptr;
    if (!tum_path.empty() && !(tum = std::fopen(tum_path.c_str(), "w"))) { std::perror(tum_path.c_str()); return 1; }
    if (pipelined) return run_pipelined(argv[3], allImuData, allImgInfo, max_frames, ImgProcesser, Estimator, tum);

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
            if (tum) write_tum(tum, Estimator);
        }
    }
    if (tum) std::fclose(tum);
    std::printf("frames %ld  feature messages %ld  odometry updates %ld  state dim %d\n", n_fe, n_be, n_odo, lvk_ekf_dim(Estimator.handle()));
    std::printf("front-end %.3f ms/frame   back-end %.3f ms/message   image read+decode %.3f ms/frame\n", n_fe ? 1e3 * t_fe / n_fe : 0.0,
                n_be ? 1e3 * t_be / n_be : 0.0, n_fe ? 1e3 * t_io / n_fe : 0.0);
    if (t_fe + t_be > 0) std::printf("processing rate %.1f frames/s (front-end + back-end, sequential, host buffers)\n", n_fe / (t_fe + t_be));
    return 0;
}
