// Output formatting after the hot path: colour-coded flow image (binary PPM) and magnitude raw.
// Interface of the reference's src/utils/io_utils.h:31-79; byte-compatible output
// (io_utils.cpp:35-114,140-225: "P6 \n<nx> <ny> \n255\n" header, Bruhn colour wheel).
#pragma once

#include <string>

#include "data2d.h"

namespace IOUtils {

typedef unsigned char GRAY;

struct RGBColor {
    int r = 0, g = 0, b = 0;
    RGBColor() = default;
    RGBColor(int red, int green, int blue) : r(red), g(green), b(blue) {}
};

// exit(255) when the file cannot be opened, like the reference (io_utils.cpp:47-51,88-92).
void WriteFlowToImageRGB(Data2D& u, Data2D& v, float flowMaxScale, std::string fileName);
void WriteMagnitudeToFileF32(Data2D& u, Data2D& v, std::string fileName);

// Direction -> hue, magnitude (clipped at 1) -> brightness.
RGBColor ConvertToRGB(float x, float y);

inline int ConvertToByte(int num) { return num >= 255 ? 255 : (num > 0 ? num : 0); }

inline GRAY ConvertToGray(float number)
{
    if (number < 0.0f) return 0;
    if (number > 255.0f) return 255;
    return static_cast<GRAY>(number);
}

}  // namespace IOUtils
