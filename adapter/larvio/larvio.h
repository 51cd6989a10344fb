// adapter/larvio/larvio.h — drop-in replacement of the reference's include/larvio/larvio.h: the same class name, namespace, public
// members and typedefs (/root/reference/include/larvio/larvio.h:37-90,421), so that app/larvioMain.cpp (:50-55,114,139-155) and
// ros_wrapper System.cpp compile and link unchanged; everything private is a handle into liblvk_hip.so.
#ifndef LARVIO_H
#define LARVIO_H

#include <map>
#include <string>
#include <vector>
#include <cstdio>
#include <boost/shared_ptr.hpp>
#include <Eigen/Dense>
#include <Eigen/Geometry>

#include <larvio/feature_msg.h>
#include "sensors/ImuData.hpp"
#include "lvk_c.h"

namespace larvio {

typedef long long int FeatureIDType;      // include/larvio/imu_state.h:24

class LarVio {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

    LarVio(std::string& config_file_);                    // larvio.cpp:40-44
    LarVio(const LarVio&) = delete;
    LarVio operator=(const LarVio&) = delete;
    ~LarVio();                                            // larvio.cpp:47-55

    bool initialize();                                    // larvio.cpp:314-360
    void reset();                                         // declared by the reference (larvio.h:57), never defined there; here: start over
    // larvio.cpp:363-461: erases the IMU samples it consumed from imu_msg_buffer (:511-512); true = publish
    bool processFeatures(MonoCameraMeasurementPtr msg, std::vector<ImuData>& imu_msg_buffer);

    Eigen::Isometry3d getTbw();                           // larvio.cpp:2644-2658
    Eigen::Vector3d getVel();                             // :2661-2668
    Eigen::Matrix<double, 6, 6> getPpose();               // :2671-2688
    Eigen::Matrix3d getPvel();                            // :2691-2699
    void getSwPoses(std::vector<Eigen::Isometry3d>& swPoses);                                            // :2702-2716
    void getStableMapPointPositions(std::map<larvio::FeatureIDType, Eigen::Vector3d>& mMapPoints);       // :2719-2723 (clears on read)
    void getActiveeMapPointPositions(std::map<larvio::FeatureIDType, Eigen::Vector3d>& mMapPoints);      // :2726-2730 (clears on read)

    typedef boost::shared_ptr<LarVio> Ptr;
    typedef boost::shared_ptr<const LarVio> ConstPtr;

  private:
    void writeLogs();
    void finish();                                        // waits for a deferred update, then the logs / map points that follow it
    std::string config_file;
    lvk_ekf_config cfg;
    lvk_context* ctx;
    lvk_ekf* ekf;
    std::FILE *f_state, *f_takeoff;
    bool takeoff_written;
    bool failed; double good_state[30];                   // a (deferred) update failed: the filter is unusable, getters answer from the last good state
    bool blocking, pending;                               // LVK_ADAPTER_BLOCKING=1: processFeatures runs the update before it returns
    std::map<FeatureIDType, Eigen::Vector3d> active_slam_features;   // refreshed after every update (larvio.cpp:455-458), cleared on read
};

typedef LarVio::Ptr LarVioPtr;
typedef LarVio::ConstPtr LarVioConstPtr;

} // namespace larvio

#endif
