// adapter/larvio/image_processor.h — drop-in replacement of the reference's include/larvio/image_processor.h: the same class name,
// namespace, public members and typedefs (/root/reference/include/larvio/image_processor.h:36-68,328), so that app/larvioMain.cpp
// (:42-48,104-107,121) and ros_wrapper System.cpp compile and link unchanged; everything private is a handle into liblvk_hip.so.
// Put this directory before the reference's include/ on the include path and link lvk_adapter + lvk_hip instead of image_processor.
#ifndef IMAGE_PROCESSOR_H
#define IMAGE_PROCESSOR_H

#include <larvio/feature_msg.h>

#include <vector>
#include <string>
#include <boost/shared_ptr.hpp>
#include <opencv2/opencv.hpp>

#include "sensors/ImuData.hpp"
#include "sensors/ImageData.hpp"
#include "lvk_c.h"

namespace larvio {

class ImageProcessor {
public:
  // Constructor (image_processor.cpp:28-33)
  ImageProcessor(std::string& config_file_);
  ImageProcessor(const ImageProcessor&) = delete;
  ImageProcessor operator=(const ImageProcessor&) = delete;
  ~ImageProcessor();

  // Initialize the object (image_processor.cpp:116-126): reads the configuration file, creates the GPU front-end.
  bool initialize();

  // image_processor.cpp:130-219; true if `features` holds this frame's message
  bool processImage(const ImageDataPtr& msg, const std::vector<ImuData>& imu_msg_buffer, MonoCameraMeasurementPtr features);

  // Get publish image (image_processor.h:63-65): the last published frame as RGB with the tracked features marked
  cv::Mat getVisualImg() { vis_wanted = true; if (vis_pending) publishVisual(); return visual_img; }

  typedef boost::shared_ptr<ImageProcessor> Ptr;
  typedef boost::shared_ptr<const ImageProcessor> ConstPtr;

private:
  void publishVisual();          // builds visual_img from the last published frame - when somebody asks for it (getVisualImg)
  std::string config_file;
  lvk_fe_config cfg;
  lvk_context* ctx;
  lvk_frontend* fe;
  std::vector<lvk_feature_obs> out;
  cv::Mat visual_img;
  cv::Mat vis_src;               // the last published frame's image: a snapshot (own pixels) once somebody has asked for pictures, the caller's own cv::Mat before
  bool vis_wanted = false, vis_shared = false;   // getVisualImg has been called at least once; vis_src still shares the caller's pixels
  bool vis_pending = false;      // a frame was published since visual_img was built
  long frames_seen = 0, vis_frame = 0;
};

typedef ImageProcessor::Ptr ImageProcessorPtr;
typedef ImageProcessor::ConstPtr ImageProcessorConstPtr;

} // end namespace larvio

#endif
