// host_san_check -- the CPU-only parts of the host layer (settings reader, Data2D raw I/O with its error paths, the
// output writers, the parameter bag) driven once under -fsanitize=address,undefined (`make san`; tests/test_sanitizers.py).
// GPU sanitizers do not exist on the pool, so this is what a sanitizer can see of the host layer without a device.
// usage: host_san_check <settings.xml> <scratch dir>     prints "host_san_check ok" and exits 0
#include <cmath>
#include <cstdio>
#include <string>

#include "data2d.h"
#include "io_utils.h"
#include "operation_parameters.h"
#include "settings.h"

#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) {                                                            \
            std::fprintf(stderr, "host_san_check: %s failed (line %d)\n", #cond, __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 3) return 3;
    const std::string dir = argv[2];
    OpticFlow::Settings s;
    EXPECT(s.LoadSettings(argv[1]) == 0);
    EXPECT(s.levels > 0 && s.iterOuter > 0 && s.iterInner > 0 && s.warpScale > 0.f);
    EXPECT(OpticFlow::Settings().LoadSettings(dir + "/no_such_settings.xml") == -1);
    {  // a truncated document must be refused, not read past its end
        std::FILE* f = std::fopen((dir + "/broken.xml").c_str(), "w");
        EXPECT(f != nullptr);
        std::fputs("<?xml version=\"1.0\"?>\n<settings>\n  <input path=\"./\" file1=\"a", f);
        std::fclose(f);
        EXPECT(OpticFlow::Settings().LoadSettings(dir + "/broken.xml") == -1);
    }

    const size_t w = 37, h = 20;
    Data2D a(w, h), b(w, h, HostMemory::Pinned);  // (a pinned request lives in pageable memory in this build)
    for (size_t y = 0; y < h; ++y)
        for (size_t x = 0; x < w; ++x) {
            a.Data(x, y) = static_cast<float>(x) - 0.5f * static_cast<float>(y);
            b.Data(x, y) = std::sin(0.1f * static_cast<float>(x * y));
        }
    EXPECT(a.WriteRAWToFileF32((dir + "/a.raw").c_str()));
    EXPECT(a.WriteRAWToFileU8((dir + "/a8.raw").c_str()));
    Data2D r;
    EXPECT(r.ReadRAWFromFileF32((dir + "/a.raw").c_str(), w, h));
    for (size_t i = 0; i < w * h; ++i) EXPECT(r.DataPtr()[i] == a.DataPtr()[i]);
    EXPECT(!r.ReadRAWFromFileF32((dir + "/a.raw").c_str(), w + 1, h));  // wrong dimensions: refused, object left empty
    EXPECT(r.Width() == 0 && r.DataPtr() == nullptr);
    EXPECT(r.ReadRAWFromFileU8((dir + "/a8.raw").c_str(), w, h));
    EXPECT(!r.ReadRAWFromFileU8((dir + "/nope.raw").c_str(), w, h));
    Data2D moved(std::move(a));
    EXPECT(moved.Width() == w && a.DataPtr() == nullptr);
    moved.Swap(b);
    moved.ZeroData();

    IOUtils::WriteFlowToImageRGB(moved, b, 3.f, dir + "/flow.ppm");
    IOUtils::WriteMagnitudeToFileF32(moved, b, dir + "/magnitude.raw");
    (void)IOUtils::ConvertToRGB(0.f, 0.f);
    (void)IOUtils::ConvertToRGB(1e30f, -1e30f);
    (void)IOUtils::ConvertToRGB(NAN, 1.f);

    OperationParameters bag;
    size_t levels = 5;
    float alpha = 3.5f;
    EXPECT(bag.PushValuePtr("warp_levels_count", &levels));
    EXPECT(!bag.PushValuePtr("warp_levels_count", &alpha));  // an existing key is not overwritten
    size_t out = 0;
    EXPECT(bag.Read("warp_levels_count", out) && out == 5);
    EXPECT(!bag.Read("missing", out));
    bag.Clear();
    EXPECT(bag.GetValuePtr("warp_levels_count") == nullptr);
    std::printf("host_san_check ok\n");
    return 0;
}
