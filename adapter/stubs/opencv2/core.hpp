// STUB for the compile check of adapter/ (the image has no OpenCV): the few members of cv::Mat the adapter touches - size, step,
// data pointer, shared ownership of the pixels, clone().  In a LARVIO tree the real <opencv2/core.hpp> is used instead.
#pragma once
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8UC1 0
#define CV_8UC3 16
namespace cv {
class Mat {
public:
    Mat() : rows(0), cols(0), step(0), data(nullptr), type_(CV_8UC1) {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t ext_step = 0) : rows(r), cols(c), step(ext_step ? ext_step : (size_t)c * (type == CV_8UC3 ? 3 : 1)), data((unsigned char*)ext), type_(type) {}
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type; step = (size_t)c * channels();
        own_ = std::make_shared<std::vector<unsigned char>>((size_t)r * step); data = own_->data();
    }
    Mat clone() const { Mat m(rows, cols, type_); for (int y = 0; y < rows; ++y) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * channels()); return m; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    int type() const { return type_; }
    unsigned char* ptr(int y) { return data + (size_t)y * step; }
    const unsigned char* ptr(int y) const { return data + (size_t)y * step; }
    int rows, cols; size_t step; unsigned char* data;
private:
    int type_; std::shared_ptr<std::vector<unsigned char>> own_;
};
}
