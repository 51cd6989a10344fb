// adapter/adapter_main.cpp — the loop of the reference's app/larvioMain.cpp:84-117 written against the ADAPTER classes
// (larvio::ImageProcessor / larvio::LarVio with the reference's own signatures), i.e. what larvioMain.cpp does once it is linked
// against lvk_adapter instead of the image_processor / estimator libraries, minus the Pangolin viewer (:57-82,118-202).
// Usage: adapter_main path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt]
#include <larvio/image_processor.h>
#include <larvio/larvio.h>

#include "../examples/lvk_dataset.hpp"
#include "../examples/lvk_png.hpp"

#include <cstdio>
#include <cstring>
#include <map>

using namespace larvio;

int main(int argc, char** argv)
{
    if (argc < 5) { std::fprintf(stderr, "Usage: %s path_to_imu/data.csv path_to_cam0/data.csv path_to_cam0/data config_file_path [--tum traj.txt]\n", argv[0]); return 1; }
    std::string tum_path;
    for (int a = 5; a < argc; ++a) if (!std::strcmp(argv[a], "--tum") && a + 1 < argc) tum_path = argv[++a];
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
    for (size_t j = 0; j < allImgInfo.size(); ++j) {
        lvk::GreyImage image; std::string err;
        if (!lvk::read_png_grey(std::string(argv[3]) + "/" + allImgInfo[j].imgName, &image, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
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
            const cv::Mat vis = ImgProcesser->getVisualImg();
            if (tum) std::fprintf(tum, "%.9f %.17g %.17g %.17g  %.17g %.17g %.17g  %.6e %.6e  %zu %d\n", imgPtr->timeStampToSec, T_b_w.translation()(0), T_b_w.translation()(1),
                                  T_b_w.translation()(2), vel(0), vel(1), vel(2), P_pose(0, 0), P_vel(0, 0), swPoses.size(), vis.cols * vis.rows * vis.channels());
        }
    }
    if (tum) std::fclose(tum);
    std::printf("frames %zu  feature messages %ld  odometry updates %ld  stable map points handed out %ld  active at the end %ld\n", allImgInfo.size(), n_msgs, n_odo, n_stable, n_active);
    return 0;
}
