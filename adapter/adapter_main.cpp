// adapter/adapter_main.cpp — the loop of the reference's app/larvioMain.cpp:84-117 written against the ADAPTER classes
// (larvio::ImageProcessor / larvio::LarVio with the reference's own signatures), i.e. what larvioMain.cpp does once it is linked
// against lvk_adapter instead of the image_processor / estimator libraries, minus the Pangolin viewer (:57-82,118-202).
// Usage: adapter_main path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt] [--bench N [--no-vis]]
// --bench N: the image files are decoded before the loop starts (the reference's player reads them inside it; a PNG decode is ten times
// a frame of this library) and the LAST N frames of the loop - processImage, processFeatures, every getter of larvioMain.cpp:117-170 -
// are timed with the steady clock; one line "bench frames N seconds S frames_per_s F" is printed (bench.py: adapter_cpp).  --no-vis
// leaves out getVisualImg (the viewer's picture: a device-to-host copy and a drawing pass per odometry message).
// The library's runtime settings (lvk_runtime_env: hardware queues, kernel arguments) need no call here: lvk_context_create applies
// them, and the adapter's initialize() is this process's first HIP call - larvioMain.cpp itself links unchanged.
#include <larvio/image_processor.h>
#include <larvio/larvio.h>

#include "../examples/lvk_dataset.hpp"
#include "../examples/lvk_png.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

using namespace larvio;

int main(int argc, char** argv)
{
    if (argc < 5) { std::fprintf(stderr, "Usage: %s path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt]\n", argv[0]); return 1; }
    std::string tum_path; long bench_n = 0; bool vis_on = true;
    for (int a = 5; a < argc; ++a) {
        if (!std::strcmp(argv[a], "--tum") && a + 1 < argc) tum_path = argv[++a];
        else if (!std::strcmp(argv[a], "--bench") && a + 1 < argc) bench_n = std::atol(argv[++a]);
        else if (!std::strcmp(argv[a], "--no-vis")) vis_on = false;
    }
    std::vector<lvk::ImuData> imu_rows; std::vector<lvk::ImgInfo> allImgInfo;
    if (!lvk::loadImuFile(argv[1], imu_rows) || !lvk::loadImageList(argv[2], allImgInfo)) { std::fprintf(stderr, "cannot read the sensor files\n"); return 1; }
    std::vector<ImuData> allImuData;
    for (const auto& m : imu_rows) allImuData.push_back(ImuData(m.timeStampToSec, m.angular_velocity[0], m.angular_velocity[1], m.angular_velocity[2],
                                                                m.linear_acceleration[0], m.linear_acceleration[1], m.linear_acceleration[2]));
    std::string config_file(argv[4]);

    ImageProcessorPtr ImgProcesser;                                   // larvioMain.cpp:42-48
    ImgProcesser.reset(new ImageProcessor(config_file));
    if (!ImgProcesser->initialize()) { std::fprintf(stderr, "Image Processer initialization failed!\n"); return 1; }
    LarVioPtr Estimator;                                              // :50-55
    Estimator.reset(new LarVio(config_file));
    if (!Estimator->initialize()) { std::fprintf(stderr, "Estimator initialization failed!\n"); return 1; }

    FILE* tum = tum_path.empty() ? nullptr : std::fopen(tum_path.c_str(), "w");
    size_t k = 0; long n_msgs = 0, n_odo = 0, n_stable = 0, n_active = 0;
    std::vector<ImuData> imu_msg_buffer;
    std::vector<lvk::GreyImage> preloaded;
    if (bench_n > 0) {
        preloaded.resize(allImgInfo.size());
        for (size_t j = 0; j < allImgInfo.size(); ++j) { std::string err; if (!lvk::read_png_grey(std::string(argv[3]) + "/" + allImgInfo[j].imgName, &preloaded[j], &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; } }
        if ((size_t)bench_n > allImgInfo.size()) bench_n = (long)allImgInfo.size();
    }
    std::chrono::steady_clock::time_point t_bench;
    for (size_t j = 0; j < allImgInfo.size(); ++j) {
        if (bench_n > 0 && j + (size_t)bench_n == allImgInfo.size()) { (void)Estimator->getTbw(); t_bench = std::chrono::steady_clock::now(); }   // the getter waits for a deferred update: the timed part starts with an idle filter
        lvk::GreyImage image_file; std::string err;
        if (bench_n <= 0 && !lvk::read_png_grey(std::string(argv[3]) + "/" + allImgInfo[j].imgName, &image_file, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        lvk::GreyImage& image = bench_n > 0 ? preloaded[j] : image_file;
        ImageDataPtr imgPtr(new ImgData);                             // :88-95
        imgPtr->timeStampToSec = allImgInfo[j].timeStampToSec;
        imgPtr->image = cv::Mat(image.height, image.width, CV_8UC1, image.data.data()).clone();
        while (k < allImuData.size() && allImuData[k].timeStampToSec - imgPtr->timeStampToSec < 0.05) imu_msg_buffer.push_back(allImuData[k++]);   // :98-103
        MonoCameraMeasurementPtr features = new MonoCameraMeasurement;                                                  // :105
        const bool bProcess = ImgProcesser->processImage(imgPtr, imu_msg_buffer, features);                             // :107
        bool bPubOdo = false;
        if (bProcess) { ++n_msgs; bPubOdo = Estimator->processFeatures(features, imu_msg_buffer); }                     // :114
        delete features;
        if (bPubOdo) {                                                                                                  // :117-170, without the viewer
            ++n_odo;
            const Eigen::Isometry3d T_b_w = Estimator->getTbw();
            const Eigen::Vector3d vel = Estimator->getVel();
            const Eigen::Matrix<double, 6, 6> P_pose = Estimator->getPpose();
            const Eigen::Matrix3d P_vel = Estimator->getPvel();
            std::vector<Eigen::Isometry3d> swPoses; Estimator->getSwPoses(swPoses);
            std::map<FeatureIDType, Eigen::Vector3d> stable, active;
            Estimator->getStableMapPointPositions(stable); Estimator->getActiveeMapPointPositions(active);
            n_stable += (long)stable.size(); n_active = (long)active.size();
            const cv::Mat vis = vis_on ? ImgProcesser->getVisualImg() : cv::Mat();
            if (tum) std::fprintf(tum, "%.9f %.17g %.17g %.17g  %.17g %.17g %.17g  %.6e %.6e  %zu %d\n", imgPtr->timeStampToSec, T_b_w.translation()(0), T_b_w.translation()(1),
                                  T_b_w.translation()(2), vel(0), vel(1), vel(2), P_pose(0, 0), P_vel(0, 0), swPoses.size(), vis.cols * vis.rows * vis.channels());
        }
    }
    if (bench_n > 0) {
        (void)Estimator->getTbw();                                    // the last deferred update is part of the timed work
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_bench).count();
        std::printf("bench frames %ld seconds %.6f frames_per_s %.1f\n", bench_n, sec, (double)bench_n / sec);
    }
    if (tum) std::fclose(tum);
    std::printf("frames %zu  feature messages %ld  odometry updates %ld  stable map points handed out %ld  active at the end %ld\n", allImgInfo.size(), n_msgs, n_odo, n_stable, n_active);
    return 0;
}
