// STUB of the reference's include/larvio/feature_msg.h:15-56 for the compile check (same public members; 72-byte record).
#pragma once
#include <vector>
namespace larvio {
class MonoFeatureMeasurement {
public:
    MonoFeatureMeasurement() : id(0), u(0.0), v(0.0), u_init(0.0), v_init(0.0), u_vel(0.0), v_vel(0.0), u_init_vel(0.0), v_init_vel(0.0) {}
    unsigned long long int id;
    double u, v, u_init, v_init, u_vel, v_vel, u_init_vel, v_init_vel;
};
class MonoCameraMeasurement {
public:
    double timeStampToSec;
    std::vector<MonoFeatureMeasurement> features;
};
typedef MonoCameraMeasurement* MonoCameraMeasurementPtr;
}
