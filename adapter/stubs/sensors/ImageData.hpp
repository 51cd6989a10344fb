// STUB of the reference's include/sensors/ImageData.hpp:17-24 for the compile check (same public members).
#pragma once
#include "opencv2/core.hpp"
#include "boost/shared_ptr.hpp"
namespace larvio {
struct ImgData { double timeStampToSec; cv::Mat image; };
typedef boost::shared_ptr<ImgData> ImageDataPtr;
}
