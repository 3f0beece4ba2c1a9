// TEST INFRASTRUCTURE ONLY -- a command-line probe over the REFERENCE'S OWN host sources.
//
// This file is ours; everything it calls is the reference's code, compiled by oracle/Makefile
// (target ref-host) from the sources where they lie under /root/reference into oracle/_ref/:
//   src/optical_flow/optical_flow_base_2d.cpp   GetMaxWarpLevel            (:36-59)
//   src/cuda_operations/2d/cuda_operation_convolution_2d.cpp  ComputeGaussianKernel (:83-112)
//   src/data_types/data2d.cpp                   raw readers / writers      (:98-231)
//   src/data_types/operation_parameters.cpp     the parameter bag          (:28-52)
//   src/utils/io_utils.cpp                      colour-wheel PPM, magnitude (:35-225)
//   src/utils/settings.cpp + vendored TinyXML   settings.xml loader        (:53-144)
// The probe prints what those functions return so that tests can pin the oracle and the host layer
// against the reference itself (tests/golden/ref_host_golden.json is its recorded output).
//
// Built with -fno-access-control (the two functions of interest are protected/private members) and
// --unresolved-symbols=ignore-all (the convolution operator's other members call the CUDA driver,
// which this image does not have; those members are never called here).  No header, library or
// tool of the CUDA toolkit is re-created: <cuda.h> resolves to the copy the image ships.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "src/cuda_operations/2d/cuda_operation_convolution_2d.h"
#include "src/data_types/data2d.h"
#include "src/data_types/operation_parameters.h"
#include "src/optical_flow/optical_flow_base_2d.h"
#include "src/utils/io_utils.h"
#include "src/utils/settings.h"

namespace {

struct LevelProbe : OpticalFlowBase2D {
  LevelProbe() : OpticalFlowBase2D("probe") {}
  bool Initialize(const DataSize3&, DataConstancy) override { return true; }
};

unsigned bits(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  return u;
}

int usage() {
  std::fprintf(stderr,
               "ref_host_probe levels W H scale | taps sigma | readu8 in W H out | readf32 in W H out |\n"
               "               writeu8 in_f32 W H out | ppm u v W H maxscale out | amp u v W H out |\n"
               "               settings file.xml | bag\n");
  return 64;
}

bool load(Data2D& d, const char* path, size_t w, size_t h) { return d.ReadRAWFromFileF32(path, w, h); }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return usage();
  std::string cmd = argv[1];
  if (cmd == "levels" && argc == 5) {
    LevelProbe p;
    size_t n = p.GetMaxWarpLevel(std::strtoull(argv[2], nullptr, 10), std::strtoull(argv[3], nullptr, 10),
                                 std::strtof(argv[4], nullptr));
    std::printf("%zu\n", n);
    return 0;
  }
  if (cmd == "taps" && argc == 3) {
    CudaOperationConvolution2D conv;
    conv.ComputeGaussianKernel(std::strtof(argv[2], nullptr), 3, 1.0f);  // the call at :159
    std::printf("%zu", conv.kernel_radius_);
    for (size_t i = 0; i < conv.kernel_length_; ++i) std::printf(" %08x", bits(conv.kernel_[i]));
    std::printf("\n");
    std::fflush(stdout);
    std::_Exit(0);  // the operator's destructor path touches the (absent) driver: leave without it
  }
  if ((cmd == "readu8" || cmd == "readf32") && argc == 6) {
    Data2D d;
    size_t w = std::strtoull(argv[3], nullptr, 10), h = std::strtoull(argv[4], nullptr, 10);
    bool ok = cmd == "readu8" ? d.ReadRAWFromFileU8(argv[2], w, h) : d.ReadRAWFromFileF32(argv[2], w, h);
    std::printf("%d\n", ok ? 1 : 0);
    std::fflush(stdout);
    if (!ok) std::_Exit(2);  // the reference double-frees after a failed read (data2d.cpp:78-89)
    return d.WriteRAWToFileF32(argv[5]) ? 0 : 3;
  }
  if (cmd == "writeu8" && argc == 6) {
    Data2D d;
    if (!load(d, argv[2], std::strtoull(argv[3], nullptr, 10), std::strtoull(argv[4], nullptr, 10))) std::_Exit(2);
    return d.WriteRAWToFileU8(argv[5]) ? 0 : 3;
  }
  if (cmd == "ppm" && argc == 8) {
    Data2D u, v;
    size_t w = std::strtoull(argv[4], nullptr, 10), h = std::strtoull(argv[5], nullptr, 10);
    if (!load(u, argv[2], w, h) || !load(v, argv[3], w, h)) std::_Exit(2);
    IOUtils::WriteFlowToImageRGB(u, v, std::strtof(argv[6], nullptr), argv[7]);
    return 0;
  }
  if (cmd == "amp" && argc == 7) {
    Data2D u, v;
    size_t w = std::strtoull(argv[4], nullptr, 10), h = std::strtoull(argv[5], nullptr, 10);
    if (!load(u, argv[2], w, h) || !load(v, argv[3], w, h)) std::_Exit(2);
    IOUtils::WriteMagnitudeToFileF32(u, v, argv[6]);
    return 0;
  }
  if (cmd == "settings" && argc == 3) {
    OpticFlow::Settings s;
    int rc = s.LoadSettings(argv[2]);
    std::printf("rc %d\n", rc);
    if (rc != 0) return 0;
    std::printf("inputPath %s\noutputPath %s\nfile1 %s\nfile2 %s\n", s.inputPath.c_str(), s.outputPath.c_str(),
                s.fileName1.c_str(), s.fileName2.c_str());
    std::printf("width %d\nheight %d\ninner %d\nouter %d\nlevels %d\nmedianRadius %d\n", s.width, s.height,
                s.iterInner, s.iterOuter, s.levels, s.medianRadius);
    std::printf("sigma %08x\nalpha %08x\ne_smooth %08x\ne_data %08x\nscaling %08x\n", bits(s.sigma), bits(s.alpha),
                bits(s.e_smooth), bits(s.e_data), bits(s.warpScale));
    return 0;
  }
  if (cmd == "bag") {
    OperationParameters p;
    int a = 1, b = 2;
    bool first = p.PushValuePtr("k", &a);
    bool second = p.PushValuePtr("k", &b);  // no overwrite (operation_parameters.cpp:28-35)
    int* got = static_cast<int*>(p.GetValuePtr("k"));
    void* missing = p.GetValuePtr("absent");
    p.Clear();
    void* cleared = p.GetValuePtr("k");
    std::printf("%d %d %d %d %d\n", first ? 1 : 0, second ? 1 : 0, got ? *got : -1, missing ? 1 : 0, cleared ? 1 : 0);
    return 0;
  }
  return usage();
}
