// Host image: tight row-major float32, the boundary type of OpticalFlow2D::ComputeFlow.
// Public interface of the reference's src/data_types/data2d.h:27-65; the raw readers/writers follow
// data2d.cpp:98-231 (u8 values are widened to float, f32 is little-endian as stored).
// Unlike the reference, a failed read leaves the object empty instead of double-freeing.
#pragma once

#include <cstddef>
#include <vector>

class Data2D {
public:
    Data2D() = default;
    Data2D(size_t width, size_t height);

    inline size_t Width() { return width_; }
    inline size_t Height() { return height_; }
    inline float* DataPtr() { return data_.empty() ? nullptr : data_.data(); }
    inline float& Data(size_t x, size_t y) { return data_[y * width_ + x]; }

    void Swap(Data2D& other);
    void ZeroData();

    bool ReadRAWFromFileU8(const char* filename, size_t width, size_t height);
    bool ReadRAWFromFileF32(const char* filename, size_t width, size_t height);
    bool WriteRAWToFileU8(const char* filename);
    bool WriteRAWToFileF32(const char* filename);

private:
    template <typename Sample>
    bool ReadRaw(const char* filename, size_t width, size_t height);

    std::vector<float> data_;
    size_t width_ = 0;
    size_t height_ = 0;
};
