// settings.xml reader.  Field names and the schema are those of the reference's OpticFlow::Settings
// (src/utils/settings.h:38-77, settings.cpp:53-144):
//   /OpticalFlow/Input/Path@inputPath, /Input/Mode@Nx,@Ny,@imageType, /Input/Mode/Files@file1,@file2,
//   /Parameters/Method@key, /Parameters/Solver/Iterations@inner,@outer,
//   /Parameters/Solver/Warping@levels,@scaling,@medianRadius, /Parameters/Solver/Model@sigma,@alpha,@e_smooth,@e_data,
//   /Output/Path@outputPath.
// The reference parses with a vendored TinyXML and crashes on a missing element; this reader is a
// ~100-line attribute scanner for the fixed schema and returns -1 instead.  Supersets:
// imageType ("8-bit" | "32-bit", default 32-bit as main.cpp:175 reads F32) and an optional
// /Parameters/Method@dataConstancy ("grey" | "gradient" | "log-derivatives" | "gradient-untiled").
#pragma once

#include <string>

namespace OpticFlow {

class Settings {
public:
    // Input settings
    std::string inputPath;
    std::string outputPath;
    std::string fileName1;
    std::string fileName2;
    std::string imageType = "32-bit";
    std::string dataConstancy = "grey";

    // General
    int width = 0;
    int height = 0;
    float sigma = 0.f;
    float precision = 0.f;
    int medianRadius = 0;

    // Solver settings
    int iterInner = 0;
    int iterOuter = 0;
    float alpha = 0.f;
    float e_smooth = 0.f;
    float e_data = 0.f;
    int levels = 0;
    float warpScale = 0.f;
    float flowScale = 0.f;
    bool press_key = false;

    // 0 on success, -1 if the file cannot be read or a mandatory element/attribute is missing.
    int LoadSettings(std::string fileName);
};

}  // namespace OpticFlow
