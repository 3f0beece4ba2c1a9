#include "data2d.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "flow2d_c_abi.h"

// FLOW2D_DATA2D_NO_DEVICE: a build of this file for host-only tools (flow2d_batch_selftest) that must not link the HIP
// library: a request for page-locked memory is then served from pageable memory, like any failed pinned allocation.
#ifdef FLOW2D_DATA2D_NO_DEVICE
static int HostAlloc(size_t, void**) { return FLOW2D_ERR_DEVICE; }
static void HostFree(void*) {}
#else
static int HostAlloc(size_t bytes, void** p) { return flow2d_host_alloc(nullptr, bytes, p); }
static void HostFree(void* p) { flow2d_host_free(nullptr, p); }
#endif

namespace {
struct FileCloser {
    void operator()(std::FILE* f) const
    {
        if (f) std::fclose(f);
    }
};
using File = std::unique_ptr<std::FILE, FileCloser>;
}  // namespace

namespace {
bool g_default_pinned = false;
}

void Data2D::UsePinnedMemory(bool on) { g_default_pinned = on; }

Data2D::Data2D(size_t width, size_t height, HostMemory memory) { Allocate(width, height, memory); }

Data2D::~Data2D() { Free(); }

Data2D::Data2D(Data2D&& other) noexcept { *this = std::move(other); }

Data2D& Data2D::operator=(Data2D&& other) noexcept
{
    if (this != &other) {
        Free();
        data_ = other.data_;
        width_ = other.width_;
        height_ = other.height_;
        pinned_ = other.pinned_;
        requested_ = other.requested_;
        other.data_ = nullptr;
        other.width_ = other.height_ = 0;
        other.pinned_ = false;
    }
    return *this;
}

// AllocateMemory / FreeMemory of the reference (data2d.cpp:56-91), with the pinned branch chosen at run time.
bool Data2D::Allocate(size_t width, size_t height, HostMemory memory)
{
    Free();
    requested_ = memory;
    const size_t count = width * height;
    if (count == 0) return true;
    const bool want_pinned = memory == HostMemory::Pinned || (memory == HostMemory::Default && g_default_pinned);
    if (want_pinned) {
        void* p = nullptr;
        if (HostAlloc(count * sizeof(float), &p) == FLOW2D_OK && p) {
            data_ = static_cast<float*>(p);
            pinned_ = true;
        }
    }
    if (!data_) {
        data_ = new (std::nothrow) float[count];
        if (!data_) {
            std::printf("Error. Cannot allocate memory on the host.\n");
            return false;
        }
    }
    std::memset(data_, 0, count * sizeof(float));
    width_ = width;
    height_ = height;
    return true;
}

void Data2D::Free()
{
    if (data_) {
        if (pinned_) HostFree(data_);
        else delete[] data_;
    }
    data_ = nullptr;
    width_ = height_ = 0;
    pinned_ = false;
}

void Data2D::Swap(Data2D& other)
{
    std::swap(data_, other.data_);
    std::swap(width_, other.width_);
    std::swap(height_, other.height_);
    std::swap(pinned_, other.pinned_);
    std::swap(requested_, other.requested_);
}

void Data2D::ZeroData()
{
    if (data_) std::memset(data_, 0, width_ * height_ * sizeof(float));
}

// The file must hold exactly width*height samples: short files and files with trailing bytes are
// rejected with the reference's "wrong dimensions" message (data2d.cpp:121-133,158-169).
template <typename Sample>
bool Data2D::ReadRaw(const char* filename, size_t width, size_t height)
{
    File file(std::fopen(filename, "rb"));
    if (!file) {
        std::printf("Cannot open file '%s'.\n", filename);
        return false;
    }
    std::vector<Sample> row(width);
    std::vector<float> pixels(width * height);
    bool ok = true;
    for (size_t y = 0; ok && y < height; ++y) {
        ok = std::fread(row.data(), sizeof(Sample), width, file.get()) == width;
        for (size_t x = 0; ok && x < width; ++x) pixels[y * width + x] = static_cast<float>(row[x]);
    }
    unsigned char extra;
    if (ok && std::fread(&extra, 1, 1, file.get()) != 0) ok = false;
    if (!ok) {
        std::printf("Error reading RAW data from file '%s': wrong dimensions.", filename);
        Free();
        return false;
    }
    if (!Allocate(width, height, requested_)) return false;
    if (!pixels.empty()) std::memcpy(data_, pixels.data(), pixels.size() * sizeof(float));
    return true;
}

bool Data2D::ReadRAWFromFileU8(const char* filename, size_t width, size_t height)
{
    return ReadRaw<unsigned char>(filename, width, height);
}

bool Data2D::ReadRAWFromFileF32(const char* filename, size_t width, size_t height)
{
    return ReadRaw<float>(filename, width, height);
}

bool Data2D::WriteRAWToFileU8(const char* filename)
{
    File file(std::fopen(filename, "wb"));
    if (!file) {
        std::printf("Cannot open file '%s'.\n", filename);
        return false;
    }
    std::vector<unsigned char> row(width_);
    for (size_t y = 0; y < height_; ++y) {
        for (size_t x = 0; x < width_; ++x)
            row[x] = static_cast<unsigned char>(std::min(255.f, std::max(0.f, data_[y * width_ + x])));
        if (std::fwrite(row.data(), 1, width_, file.get()) != width_) {
            std::printf("Error writing RAW data to file '%s'.", filename);
            return false;
        }
    }
    return true;
}

bool Data2D::WriteRAWToFileF32(const char* filename)
{
    File file(std::fopen(filename, "wb"));
    if (!file) {
        std::printf("Cannot open file '%s'.\n", filename);
        return false;
    }
    const size_t count = width_ * height_;
    if (count && std::fwrite(data_, sizeof(float), count, file.get()) != count) {
        std::printf("Error writing RAW data to file '%s'.", filename);
        return false;
    }
    return true;
}
