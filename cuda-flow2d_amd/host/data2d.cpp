#include "data2d.h"

#include <algorithm>
#include <cstdio>
#include <memory>

namespace {
struct FileCloser {
    void operator()(std::FILE* f) const
    {
        if (f) std::fclose(f);
    }
};
using File = std::unique_ptr<std::FILE, FileCloser>;
}  // namespace

Data2D::Data2D(size_t width, size_t height) : data_(width * height, 0.f), width_(width), height_(height) {}

void Data2D::Swap(Data2D& other)
{
    data_.swap(other.data_);
    std::swap(width_, other.width_);
    std::swap(height_, other.height_);
}

void Data2D::ZeroData() { std::fill(data_.begin(), data_.end(), 0.f); }

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
        data_.clear();
        width_ = height_ = 0;
        return false;
    }
    data_.swap(pixels);
    width_ = width;
    height_ = height;
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
    if (count && std::fwrite(data_.data(), sizeof(float), count, file.get()) != count) {
        std::printf("Error writing RAW data to file '%s'.", filename);
        return false;
    }
    return true;
}
