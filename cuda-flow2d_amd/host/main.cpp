// flow2d: command-line front end of the MI355X optical-flow path.
// Keeps the argv forms, defaults, exit codes and output files of the reference's src/main.cpp:46-229:
//   flow2d                                   -> ./settings.xml
//   flow2d <settings.xml>
//   flow2d <file1> <file2> <width> <height> <prefix> <outdir/> [<alpha> <sigma>]     (argc 7 / 9)
//   flow2d <file1> <file2> <width> <height> <outdir/>                                (argc 6)
// exit codes: 1 no device, 2 frame load failed, 3 settings error, 255 cannot write PPM/amp, 0 otherwise.
// Documented supersets: imageType="8-bit" selects the u8 reader; inputPath is tried as
// a prefix before the bare file name; argc == 6 no longer dereferences argv[6]; no blocking getchar();
// options --u8, --gradient, --log-derivatives, --device N, --verbose, --sor OMEGA (opt-in red-black SOR, no reference parity)
// may precede the positional arguments.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "device_utils.h"
#include "io_utils.h"
#include "optical_flow_2d.h"
#include "settings.h"

using namespace OpticFlow;

static bool LoadFrame(Data2D& frame, const std::string& input_path, const std::string& name, bool u8, size_t w,
                      size_t h)
{
    std::vector<std::string> candidates;
    if (!input_path.empty()) candidates.push_back(input_path + name);
    candidates.push_back(name);
    for (const std::string& path : candidates) {
        std::FILE* probe = std::fopen(path.c_str(), "rb");
        if (!probe) continue;
        std::fclose(probe);
        return u8 ? frame.ReadRAWFromFileU8(path.c_str(), w, h) : frame.ReadRAWFromFileF32(path.c_str(), w, h);
    }
    std::printf("Cannot open file '%s'.\n", name.c_str());
    return false;
}

int main(int argc, char** argv)
{
    // optional flags first (supersets), then the reference's positional forms
    bool force_u8 = false, verbose = false;
    int device = 0;
    float sor_omega = 0.f;
    DataConstancy data_constancy = DataConstancy::Grey;
    std::vector<char*> args = {argv[0]};
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--u8")) force_u8 = true;
        else if (!std::strcmp(argv[i], "--gradient")) data_constancy = DataConstancy::Gradient;
        else if (!std::strcmp(argv[i], "--gradient-untiled")) data_constancy = DataConstancy::GradientUntiled;
        else if (!std::strcmp(argv[i], "--log-derivatives")) data_constancy = DataConstancy::LogDerivatives;
        else if (!std::strcmp(argv[i], "--verbose")) verbose = true;
        else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--sor") && i + 1 < argc) sor_omega = static_cast<float>(std::atof(argv[++i]));
        else args.push_back(argv[i]);
    }
    const int nargs = static_cast<int>(args.size());

    if (!InitDeviceContext(device)) return 1;
    // the reference's ALLOCATE_PINNED_MEMORY option (data2d.cpp:34), on: frames and flows live in page-locked memory
    // and move by DMA (a failed pinned allocation falls back to pageable memory, same results)
    Data2D::UsePinnedMemory(true);

    std::printf("//----------------------------------------------------------------------//\n");
    std::printf("//       2D optical flow, MI355X (gfx950) HIP path. flow2d 0.1          //\n");
    std::printf("//----------------------------------------------------------------------//\n");

    // defaults of main.cpp:65-86
    size_t width = 584, height = 388;
    size_t warp_levels_count = 50;
    float warp_scale_factor = 0.9f;
    size_t outer_iterations_count = 40;
    size_t inner_iterations_count = 5;
    float equation_alpha = 35.0f;
    float equation_smoothness = 0.001f;
    float equation_data = 0.001f;
    size_t median_radius = 5;
    float gaussian_sigma = 1.5f;
    std::string file_name1 = "rub1.raw", file_name2 = "rub2.raw";
    std::string input_path = "./data/", output_path = "./data/output/", counter;
    bool u8 = force_u8;

    if (nargs == 6 || nargs == 7 || nargs == 9) {
        file_name1 = args[1];
        file_name2 = args[2];
        width = std::atoi(args[3]);
        height = std::atoi(args[4]);
        output_path = (nargs == 6) ? args[5] : args[6];
        if (nargs == 7) counter = args[5];
        if (nargs == 9) {
            equation_alpha = std::atof(args[7]);
            gaussian_sigma = std::atof(args[8]);
            counter = "alpha" + std::string(args[7]) + "_sigma" + std::string(args[8]) + "_";
        }
        input_path.clear();
    } else if (nargs < 3) {
        const std::string settings_file = (nargs == 1) ? "settings.xml" : std::string(args[1]);
        std::cout << "Reading settings: " << settings_file << std::endl;
        Settings settings;
        if (settings.LoadSettings(settings_file)) {
            std::cout << "TERMINATING. Error reading settings: " << settings_file << std::endl;
            return 3;
        }
        std::cout << "OK" << std::endl << std::endl;
        width = settings.width;
        height = settings.height;
        input_path = settings.inputPath;
        output_path = settings.outputPath;
        file_name1 = settings.fileName1;
        file_name2 = settings.fileName2;
        warp_levels_count = settings.levels;
        warp_scale_factor = settings.warpScale;
        outer_iterations_count = settings.iterOuter;
        inner_iterations_count = settings.iterInner;
        equation_alpha = settings.alpha;
        equation_data = settings.e_data;
        equation_smoothness = settings.e_smooth;
        median_radius = settings.medianRadius;
        gaussian_sigma = settings.sigma;
        if (settings.imageType == "8-bit") u8 = true;
        if (settings.dataConstancy == "gradient") data_constancy = DataConstancy::Gradient;
        if (settings.dataConstancy == "gradient-untiled") data_constancy = DataConstancy::GradientUntiled;
        if (settings.dataConstancy == "log-derivatives") data_constancy = DataConstancy::LogDerivatives;
    } else {
        std::cout << "Usage: " << args[0] << " <settings file>. Otherwise settings.xml in the current directory is used"
                  << std::endl;
        return 0;
    }

    DataSize3 data_size = {width, height, 1};
    Data2D frame_0, frame_1;
    if (!LoadFrame(frame_0, input_path, file_name1, u8, width, height) ||
        !LoadFrame(frame_1, input_path, file_name2, u8, width, height)) {
        return 2;
    }

    OpticalFlow2D optical_flow;
    optical_flow.silent = !verbose;
    if (optical_flow.Initialize(data_size, data_constancy)) {
        Data2D flow_u(width, height), flow_v(width, height);
        OperationParameters params;
        params.PushValuePtr("warp_levels_count", &warp_levels_count);
        params.PushValuePtr("warp_scale_factor", &warp_scale_factor);
        params.PushValuePtr("outer_iterations_count", &outer_iterations_count);
        params.PushValuePtr("inner_iterations_count", &inner_iterations_count);
        params.PushValuePtr("equation_alpha", &equation_alpha);
        params.PushValuePtr("equation_smoothness", &equation_smoothness);
        params.PushValuePtr("equation_data", &equation_data);
        params.PushValuePtr("median_radius", &median_radius);
        params.PushValuePtr("gaussian_sigma", &gaussian_sigma);
        if (sor_omega != 0.f) params.PushValuePtr("solver_sor_omega", &sor_omega);
        optical_flow.ComputeFlow(frame_0, frame_1, flow_u, flow_v, params);
        if (!optical_flow.LastRunSucceeded()) {
            // the reference writes whatever its buffers hold after a failed run; no output files here instead
            std::cout << "Error: the flow computation failed, no output written." << std::endl;
            optical_flow.Destroy();
            DestroyDeviceContext();
            return 4;  // superset of the reference's exit codes (0, 1, 2, 3, 255)
        }

        const std::string suffix = "-" + std::to_string(width) + "-" + std::to_string(height) + ".raw";
        flow_u.WriteRAWToFileF32((output_path + counter + "flow-u" + suffix).c_str());
        flow_v.WriteRAWToFileF32((output_path + counter + "flow-v" + suffix).c_str());
        IOUtils::WriteFlowToImageRGB(flow_u, flow_v, 10, output_path + counter + "res.pgm");
        IOUtils::WriteMagnitudeToFileF32(flow_u, flow_v, output_path + counter + "amp" + suffix);
        optical_flow.Destroy();
    }
    DestroyDeviceContext();
    return 0;
}
