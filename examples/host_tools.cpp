// host_tools.cpp — exposes the host-only helpers of the dataset driver (configuration reader, PNG reader, CSV readers) on the
// command line so that tests/test_host_tools.py can check them on a machine without a GPU.  Not linked against liblvk_hip.so.
//   host_tools config <file.yaml>          the two C-ABI configuration structs, one "name value" line per field
//   host_tools png <in.png> <out.raw>      prints "width height", writes the 8-bit grey pixels
//   host_tools imu <data.csv>              one "%.17g x7" line per sample
//   host_tools images <data.csv>           one "stamp name" line per image
#include "lvk_config.hpp"
#include "lvk_dataset.hpp"
#include "lvk_png.hpp"
#include <cstdio>
#include <cstring>

static void show(const char* name, double v) { std::printf("%s %.17g\n", name, v); }
static void show(const char* name, const double* v, int n) { std::printf("%s", name); for (int i = 0; i < n; ++i) std::printf(" %.17g", v[i]); std::printf("\n"); }

int main(int argc, char** argv)
{
    if (argc >= 3 && !std::strcmp(argv[1], "config")) {
        lvk::ConfigFile f;
        if (!f.open(argv[2])) { std::fprintf(stderr, "config_file error: %s\n", f.error().c_str()); return 1; }
        lvk_fe_config a; lvk_ekf_config b; std::string err;
        if (!lvk::load_fe_config(f, &a, &err) || !lvk::load_ekf_config(f, &b, &err)) { std::fprintf(stderr, "config_file error: %s\n", err.c_str()); return 1; }
        std::printf("output_dir %s\n", f.str("output_dir").c_str());
#define FE(x) show("fe." #x, (double)a.x)
        FE(width); FE(height); FE(pyramid_levels); FE(patch_size); FE(max_iteration); FE(track_precision); FE(max_features_num);
        FE(min_distance); FE(flag_equalize); FE(pub_frequency); FE(distortion_model);
        show("fe.intrinsics", a.intrinsics, 4); show("fe.distortion", a.distortion, 4); show("fe.R_cam_imu", a.R_cam_imu, 9);
#define BE(x) show("ekf." #x, (double)b.x)
        BE(if_fej); BE(estimate_extrin); BE(estimate_td); BE(if_zupt_valid); BE(sw_size); BE(max_track_len); BE(least_observation_number);
        BE(max_features_in_one_grid); BE(aug_grid_rows); BE(aug_grid_cols); BE(pub_frequency); BE(imu_rate); BE(width); BE(height);
        show("ekf.intrinsics", b.intrinsics, 4); show("ekf.T_cam_imu", b.T_cam_imu, 16);
        BE(td); BE(noise_gyro); BE(noise_acc); BE(noise_gyro_bias); BE(noise_acc_bias); BE(noise_feature);
        BE(initial_covariance_orientation); BE(initial_covariance_velocity); BE(initial_covariance_position); BE(initial_covariance_gyro_bias);
        BE(initial_covariance_acc_bias); BE(initial_covariance_extrin_rot); BE(initial_covariance_extrin_trans);
        BE(rotation_threshold); BE(translation_threshold); BE(tracking_rate_threshold); BE(feature_translation_threshold);
        BE(zupt_max_feature_dis); BE(zupt_noise_v); BE(zupt_noise_p); BE(zupt_noise_q); BE(static_duration);
        BE(feature_idp_dim); BE(use_schmidt); BE(calib_imu_instrinsic); BE(max_features); BE(legacy_grid); BE(reserved0);
        return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "png")) {
        lvk::GreyImage img; std::string err;
        if (!lvk::read_png_grey(argv[2], &img, &err)) { std::fprintf(stderr, "%s: %s\n", argv[2], err.c_str()); return 1; }
        FILE* o = std::fopen(argv[3], "wb"); if (!o) { std::perror(argv[3]); return 1; }
        std::fwrite(img.data.data(), 1, img.data.size(), o); std::fclose(o);
        std::printf("%d %d\n", img.width, img.height);
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "imu")) {
        std::vector<lvk::ImuData> v;
        if (!lvk::loadImuFile(argv[2], v)) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        for (size_t i = 0; i < v.size(); ++i)
            std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", v[i].timeStampToSec, v[i].angular_velocity[0], v[i].angular_velocity[1],
                        v[i].angular_velocity[2], v[i].linear_acceleration[0], v[i].linear_acceleration[1], v[i].linear_acceleration[2]);
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "images")) {
        std::vector<lvk::ImgInfo> v;
        if (!lvk::loadImageList(argv[2], v)) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        for (size_t i = 0; i < v.size(); ++i) std::printf("%.17g %s\n", v[i].timeStampToSec, v[i].imgName.c_str());
        return 0;
    }
    std::fprintf(stderr, "usage: host_tools config|png|imu|images ...\n");
    return 2;
}
