// TEST INFRASTRUCTURE ONLY: the few members of cv::Mat that beamforming-lk_amd/host/aw_processing_unit.{h,cpp} touch,
// so that the exact-signature AWProcessingUnit can be compiled and run on hosts without OpenCV (this image has
// none).  Not OpenCV, not a stand-in for building the reference, and never on the include path of the product
// (the Makefile adds it for tests/host/test_exact_signatures only).  Semantics follow cv::Mat's documented ones.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#define CV_8UC1 0
#define CV_8UC3 16

namespace cv {

struct Size {
    int width = 0, height = 0;
};

class Mat {
public:
    Mat() = default;
    Mat(int rows, int cols, int type) { create(rows, cols, type); }
    void create(int r, int c, int t) {
        rows = r;
        cols = c;
        type_ = t;
        store.assign((size_t) r * c * (t == CV_8UC3 ? 3 : 1), 0);
        data = store.data();
    }
    int type() const { return type_; }
    bool isContinuous() const { return true; }
    Size size() const { return Size{cols, rows}; }
    template <class T>
    T &at(int r, int c) { return reinterpret_cast<T *>(data)[(size_t) r * cols + c]; }

    int rows = 0, cols = 0;
    uint8_t *data = nullptr;

private:
    int type_ = CV_8UC1;
    std::vector<uint8_t> store;
};

}  // namespace cv
