#include "io_utils.h"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

namespace IOUtils {

void WriteFlowToImageRGB(Data2D& u, Data2D& v, float flowMaxScale, std::string fileName)
{
    const float factor = 1.0 / flowMaxScale;
    std::ofstream out(fileName.c_str(), std::ios::out | std::ios::binary);
    if (!out.is_open()) {
        std::cerr << "Error: cannot save file " << std::endl;
        std::exit(255);
    }
    const int nx = static_cast<int>(u.Width()), ny = static_cast<int>(u.Height());
    out << "P6 \n" << nx << " " << ny << " \n255\n";
    std::vector<GRAY> row(static_cast<size_t>(nx) * 3);
    for (int y = 0; y < ny; ++y) {
        for (int x = 0; x < nx; ++x) {
            const RGBColor c = ConvertToRGB(u.Data(x, y) * factor, v.Data(x, y) * factor);
            row[3 * x + 0] = ConvertToGray(static_cast<float>(c.r));
            row[3 * x + 1] = ConvertToGray(static_cast<float>(c.g));
            row[3 * x + 2] = ConvertToGray(static_cast<float>(c.b));
        }
        out.write(reinterpret_cast<const char*>(row.data()), row.size());
    }
}

void WriteMagnitudeToFileF32(Data2D& u, Data2D& v, std::string fileName)
{
    std::ofstream out(fileName.c_str(), std::ios::out | std::ios::binary);
    if (!out.is_open()) {
        std::cerr << "Error: cannot save file " << std::endl;
        std::exit(255);
    }
    const int nx = static_cast<int>(u.Width()), ny = static_cast<int>(u.Height());
    std::vector<float> row(nx);
    for (int y = 0; y < ny; ++y) {
        for (int x = 0; x < nx; ++x) row[x] = std::sqrt(u.Data(x, y) * u.Data(x, y) + v.Data(x, y) * v.Data(x, y));
        out.write(reinterpret_cast<const char*>(row.data()), nx * sizeof(float));
    }
}

// Colour wheel of io_utils.cpp:140-225 as a table: the half angle phi/2 in [0, pi] runs through the key
// colours below; inside a segment the colour is the linear blend of its two ends, scaled by the
// (clipped) magnitude and floored.  float/double promotions follow the reference expression by
// expression so the bytes match.
RGBColor ConvertToRGB(float x, float y)
{
    struct Key {
        double at;  // in units of pi
        double r, g, b;
    };
    static const Key keys[] = {
        {0.0, 255.0, 0.0, 0.0},     {0.125, 255.0, 0.0, 255.0}, {0.25, 64.0, 64.0, 255.0}, {0.375, 0.0, 255.0, 255.0},
        {0.5, 0.0, 255.0, 0.0},     {0.75, 255.0, 255.0, 0.0},  {1.0, 255.0, 0.0, 0.0},
    };
    const float Pi = 2.0 * std::acos(0.0);
    float amp = std::sqrt(x * x + y * y);
    if (amp > 1) amp = 1;
    float phi;
    if (x == 0.0f)
        phi = (y >= 0.0f) ? 0.5 * Pi : 1.5 * Pi;
    else if (x > 0.0f)
        phi = (y >= 0.0f) ? std::atan(y / x) : 2.0 * Pi + std::atan(y / x);
    else
        phi = Pi + std::atan(y / x);
    phi = phi / 2.0;

    RGBColor rgb;
    for (int s = 0; s < 6; ++s) {
        const double lo = keys[s].at * Pi, hi = keys[s + 1].at * Pi;
        const bool last = (s == 5);
        if (!(phi >= lo && (last ? phi <= hi : phi < hi))) continue;
        const float beta = (phi - lo) / ((keys[s + 1].at - keys[s].at) * Pi);
        const float alpha = 1.0 - beta;
        rgb.r = static_cast<int>(std::floor(amp * (alpha * keys[s].r + beta * keys[s + 1].r)));
        rgb.g = static_cast<int>(std::floor(amp * (alpha * keys[s].g + beta * keys[s + 1].g)));
        rgb.b = static_cast<int>(std::floor(amp * (alpha * keys[s].b + beta * keys[s + 1].b)));
    }
    rgb.r = ConvertToByte(rgb.r);
    rgb.g = ConvertToByte(rgb.g);
    rgb.b = ConvertToByte(rgb.b);
    return rgb;
}

}  // namespace IOUtils
