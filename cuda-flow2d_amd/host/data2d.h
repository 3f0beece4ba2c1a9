// Host image: tight row-major float32, the boundary type of OpticalFlow2D::ComputeFlow.
// Public interface of the reference's src/data_types/data2d.h:27-65; the raw readers/writers follow
// data2d.cpp:98-231 (u8 values are widened to float, f32 is little-endian as stored).
// Unlike the reference, a failed read leaves the object empty instead of double-freeing.
//
// Page-locked images: the reference has a compile-time switch ALLOCATE_PINNED_MEMORY (data2d.cpp:34,60-61,80-82:
// cuMemAllocHost instead of new[]); here it is a run-time choice per object, HostMemory::Pinned, or process-wide
// through Data2D::UsePinnedMemory(true) for objects created without saying (the CLI does that once it has a device).
// A pinned image is uploaded and downloaded by DMA at the PCIe rate without blocking the host, which is what
// OpticalFlowBatch2D::ComputeFlowBatch overlaps with the pyramids of other pairs.  When the pinned allocation fails
// (no device, limit reached) the image silently lives in pageable memory: same results, slower copies.
#pragma once

#include <cstddef>

enum class HostMemory { Default, Pageable, Pinned };

class Data2D {
public:
    Data2D() = default;
    Data2D(size_t width, size_t height, HostMemory memory = HostMemory::Default);
    ~Data2D();
    Data2D(Data2D&& other) noexcept;
    Data2D& operator=(Data2D&& other) noexcept;
    Data2D(const Data2D&) = delete;
    Data2D& operator=(const Data2D&) = delete;

    // what HostMemory::Default means from now on (false at start-up: pageable, like the reference as shipped)
    static void UsePinnedMemory(bool on);

    inline size_t Width() { return width_; }
    inline size_t Height() { return height_; }
    inline float* DataPtr() { return data_; }
    inline float& Data(size_t x, size_t y) { return data_[y * width_ + x]; }
    bool IsPinned() const { return pinned_; }

    void Swap(Data2D& other);
    void ZeroData();

    bool ReadRAWFromFileU8(const char* filename, size_t width, size_t height);
    bool ReadRAWFromFileF32(const char* filename, size_t width, size_t height);
    bool WriteRAWToFileU8(const char* filename);
    bool WriteRAWToFileF32(const char* filename);

private:
    template <typename Sample>
    bool ReadRaw(const char* filename, size_t width, size_t height);

    bool Allocate(size_t width, size_t height, HostMemory memory);  // zero-filled; frees what the object held
    void Free();

    float* data_ = nullptr;
    size_t width_ = 0;
    size_t height_ = 0;
    bool pinned_ = false;
    HostMemory requested_ = HostMemory::Default;  // kept across Read*: a pinned image stays pinned when reloaded
};
