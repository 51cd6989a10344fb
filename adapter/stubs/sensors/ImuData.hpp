// STUB of the reference's include/sensors/ImuData.hpp:17-43 for the compile check (same public members).
#pragma once
#include "Eigen/Core"
#include "Eigen/Dense"
namespace larvio {
struct ImuData {
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    ImuData(double t, double wx, double wy, double wz, double ax, double ay, double az)
        : timeStampToSec(t), angular_velocity(wx, wy, wz), linear_acceleration(ax, ay, az) {}
    ImuData(double t, const Eigen::Vector3d& omg, const Eigen::Vector3d& acc) : timeStampToSec(t), angular_velocity(omg), linear_acceleration(acc) {}
    double timeStampToSec;
    Eigen::Vector3d angular_velocity;
    Eigen::Vector3d linear_acceleration;
};
}
