// TEST INFRASTRUCTURE ONLY -- runs the REFERENCE'S OWN device kernels on the MI355X.
//
// oracle/Makefile (target ref-kernels) compiles the six files src/kernels/*_2d.cu of the reference,
// from where they lie under /root/reference, with hipcc for gfx950 into oracle/_ref/<op>_2d.co (the
// kernels are plain `extern "C" __global__` C with threadIdx/__shared__/__constant__, which HIP
// compiles natively).  This file is ours: it binds to those code objects exactly as the reference's
// operator layer binds to its PTX modules -- module file per operator, functions and the
// `container_size` / `c_Kernel` globals looked up BY NAME (cuda_operation_*_2d.cpp Initialize()),
// launch geometry, dynamic LDS size and argument order of each Execute() -- through the HIP module
// API (hipModuleLoad / hipModuleGetFunction / hipModuleGetGlobal / hipModuleLaunchKernel), and
// restates the host-side loops around them (CudaOperationSolve2D::Execute, ComputeFlow) so that the
// product and the CPU oracle can be compared with what the reference's kernels really compute.
// Nothing under cuda-flow2d_amd/ links, loads or calls this.
//
// Every plane is allocated with kGuardRows spare rows above and below the container: several
// reference kernels read (and convolutionRows writes) a few rows / pixels outside the image for
// threads beyond the edge ("race detection"); inside the guard that is harmless.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

struct DataSize3 {  // data_structs.h:31-35
  size_t width, height, pitch;
};

constexpr size_t kGuardRows = 32;

struct Driver {
  bool open = false;
  DataSize3 container{};
  hipModule_t m_add{}, m_conv{}, m_median{}, m_reg{}, m_res{}, m_solve{};
  hipFunction_t f_add{}, f_rows{}, f_cols{}, f_median{}, f_reg{}, f_res_x{}, f_res_y{}, f_phi_ksi{}, f_solve{},
      f_solve_grad{}, f_solve_log{};
  hipDeviceptr_t c_kernel{};
  std::vector<void*> allocations;  // base pointers of guarded planes
  std::vector<float*> pool;        // the twelve containers of ComputeFlow
  char error[512] = {0};
};

Driver g;

int fail(hipError_t e, const char* what) {
  std::snprintf(g.error, sizeof g.error, "%s: %s", what, hipGetErrorString(e));
  return 1;
}

#define HIP_OK(call)                          \
  do {                                        \
    hipError_t e_ = (call);                   \
    if (e_ != hipSuccess) return fail(e_, #call); \
  } while (0)

int load_module(const std::string& dir, const char* op, hipModule_t* mod) {
  std::string path = dir + "/" + op + ".co";
  hipError_t e = hipModuleLoad(mod, path.c_str());
  if (e != hipSuccess) {
    std::snprintf(g.error, sizeof g.error, "hipModuleLoad(%s): %s", path.c_str(), hipGetErrorString(e));
    return 1;
  }
  // `container_size` upload with the size check of e.g. cuda_operation_add_2d.cpp:57-64
  hipDeviceptr_t dev = nullptr;
  size_t bytes = 0;
  HIP_OK(hipModuleGetGlobal(&dev, &bytes, *mod, "container_size"));
  if (bytes != sizeof(DataSize3)) {
    std::snprintf(g.error, sizeof g.error, "%s: container_size is %zu bytes", path.c_str(), bytes);
    return 1;
  }
  HIP_OK(hipMemcpyHtoD(dev, &g.container, sizeof(DataSize3)));
  return 0;
}

unsigned cdiv(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

int launch(hipFunction_t f, unsigned gx, unsigned gy, unsigned bx, unsigned by, unsigned lds, void** args) {
  HIP_OK(hipModuleLaunchKernel(f, gx, gy, 1, bx, by, 1, lds, nullptr, args, nullptr));
  return 0;
}

size_t plane_bytes() { return g.container.pitch * g.container.height; }

// Every operator refuses a level that does not fit the container (the kernels have no such check).
int level_ok(size_t w, size_t h) {
  if (g.open && w >= 1 && h >= 1 && w <= g.container.width && h <= g.container.height) return 0;
  std::snprintf(g.error, sizeof g.error, "level %zux%zu outside the container or driver not open", w, h);
  return 6;
}

// cuda_operation_convolution_2d.cpp:83-112 ComputeGaussianKernel(sigma, 3, 1.0)
int gaussian_taps(float sigma, float* taps, int* radius_out) {
  const size_t precision = 3;
  const float pixel_size = 1.0f;
  size_t radius = (size_t)(precision * sigma / pixel_size);
  if (2 * radius + 1 > 51) return 1;  // c_Kernel[MAX_KERNEL_LENGTH], convolution_2d.cu:49-57
  int r = static_cast<int>(radius);
  for (int i = -r; i <= r; i++) {
    float val = 1.0 / (sigma * std::sqrt(2.0 * 3.1415926)) * std::exp(-(i * i * pixel_size * pixel_size) / (2.0 * sigma * sigma));
    taps[i + r] = val;
  }
  float sum = 0.0;
  for (int i = 0; i < 2 * r + 1; i++) sum = sum + taps[i];
  for (int i = 0; i < 2 * r + 1; i++) taps[i] = taps[i] / sum;
  *radius_out = r;
  return 0;
}

}  // namespace

REF_API const char* refk_last_error() { return g.error; }

REF_API int refk_close() {
  if (!g.open) return 0;
  (void)hipDeviceSynchronize();
  for (void* p : g.allocations) (void)hipFree(p);
  g.allocations.clear();
  g.pool.clear();
  hipModule_t mods[6] = {g.m_add, g.m_conv, g.m_median, g.m_reg, g.m_res, g.m_solve};
  for (hipModule_t m : mods)
    if (m) (void)hipModuleUnload(m);
  g = Driver{};
  return 0;
}

// Initialize() of the six operators: modules <dir>/<op>.co, functions by name, container_size.
REF_API int refk_open(const char* dir, size_t container_w, size_t container_h, size_t pitch_bytes) {
  refk_close();
  if (pitch_bytes % 4 || pitch_bytes < container_w * 4 || !container_w || !container_h) {
    std::snprintf(g.error, sizeof g.error, "bad container %zux%zu pitch %zu", container_w, container_h, pitch_bytes);
    return 1;
  }
  HIP_OK(hipInit(0));
  HIP_OK(hipSetDevice(0));
  g.container = {container_w, container_h, pitch_bytes};
  std::string d(dir);
  if (load_module(d, "add_2d", &g.m_add) || load_module(d, "convolution_2d", &g.m_conv) ||
      load_module(d, "median_2d", &g.m_median) || load_module(d, "registration_2d", &g.m_reg) ||
      load_module(d, "resample_2d", &g.m_res) || load_module(d, "solve_2d", &g.m_solve))
    return 1;
  HIP_OK(hipModuleGetFunction(&g.f_add, g.m_add, "add_2d"));
  HIP_OK(hipModuleGetFunction(&g.f_rows, g.m_conv, "convolutionRowsKernel"));
  HIP_OK(hipModuleGetFunction(&g.f_cols, g.m_conv, "convolutionColumnsKernel"));
  HIP_OK(hipModuleGetFunction(&g.f_median, g.m_median, "median_2d"));
  HIP_OK(hipModuleGetFunction(&g.f_reg, g.m_reg, "registration_2d"));
  HIP_OK(hipModuleGetFunction(&g.f_res_x, g.m_res, "resample_x"));
  HIP_OK(hipModuleGetFunction(&g.f_res_y, g.m_res, "resample_y"));
  HIP_OK(hipModuleGetFunction(&g.f_phi_ksi, g.m_solve, "compute_phi_ksi"));
  HIP_OK(hipModuleGetFunction(&g.f_solve, g.m_solve, "solve_2d"));
  HIP_OK(hipModuleGetFunction(&g.f_solve_grad, g.m_solve, "solve_2d_grad"));
  HIP_OK(hipModuleGetFunction(&g.f_solve_log, g.m_solve, "solve_2d_log"));
  size_t bytes = 0;
  HIP_OK(hipModuleGetGlobal(&g.c_kernel, &bytes, g.m_conv, "c_Kernel"));
  if (bytes != 51 * sizeof(float)) {
    std::snprintf(g.error, sizeof g.error, "c_Kernel is %zu bytes", bytes);
    return 1;
  }
  g.open = true;
  return 0;
}

// One zero-filled container with guard rows; returns the interior pointer.
REF_API float* refk_plane_alloc() {
  if (!g.open) return nullptr;
  size_t bytes = g.container.pitch * (g.container.height + 2 * kGuardRows);
  void* base = nullptr;
  if (hipMalloc(&base, bytes) != hipSuccess) return nullptr;
  if (hipMemset(base, 0, bytes) != hipSuccess) {
    (void)hipFree(base);
    return nullptr;
  }
  g.allocations.push_back(base);
  return reinterpret_cast<float*>(static_cast<char*>(base) + kGuardRows * g.container.pitch);
}

// Whole container (container_h rows of pitch bytes) to / from a host array of the same shape.
REF_API int refk_upload(float* plane, const float* host) {
  HIP_OK(hipMemcpy(plane, host, plane_bytes(), hipMemcpyHostToDevice));
  return 0;
}

REF_API int refk_download(const float* plane, float* host) {
  HIP_OK(hipMemcpy(host, plane, plane_bytes(), hipMemcpyDeviceToHost));
  return 0;
}

// Tight w x h host image <-> top-left corner (CopyData2D{to,From}Device, cuda_utils.cpp:66-105)
REF_API int refk_upload_2d(float* plane, const float* host, size_t w, size_t h) {
  HIP_OK(hipMemcpy2D(plane, g.container.pitch, host, w * 4, w * 4, h, hipMemcpyHostToDevice));
  return 0;
}

REF_API int refk_download_2d(const float* plane, float* host, size_t w, size_t h) {
  HIP_OK(hipMemcpy2D(host, w * 4, plane, g.container.pitch, w * 4, h, hipMemcpyDeviceToHost));
  return 0;
}

REF_API int refk_sync() {
  HIP_OK(hipDeviceSynchronize());
  return 0;
}

// ---- operators: geometry and argument order of each Execute() --------------------------------

// cuda_operation_add_2d.cpp:89-105
REF_API int refk_add(float* operand_0, const float* operand_1, size_t w, size_t h) {
  if (int rc = level_ok(w, h)) return rc;
  void* args[4] = {&operand_0, &operand_1, &w, &h};
  return launch(g.f_add, cdiv(w, 16), cdiv(h, 8), 16, 8, 0, args);
}

// cuda_operation_convolution_2d.cpp:163-278 with given taps (c_Kernel upload, rows 16x4 / columns 4x16,
// four results per thread, 1536 bytes of dynamic LDS requested although the kernels use static LDS)
REF_API int refk_convolution_taps(const float* input, float* output, float* temp, size_t w, size_t h,
                                  const float* taps, int radius) {
  if (int rc = level_ok(w, h)) return rc;
  if (input == output) return 2;  // :151-154
  if (radius < 0 || 2 * radius + 1 > 51 || radius > 16) return 3;
  HIP_OK(hipMemcpyHtoD(g.c_kernel, const_cast<float*>(taps), (2 * radius + 1) * sizeof(float)));
  int iw = static_cast<int>(w), ih = static_cast<int>(h);
  int pitch = static_cast<int>(g.container.pitch / sizeof(float));
  {
    void* args[6] = {&temp, &input, &iw, &ih, &pitch, &radius};
    if (launch(g.f_rows, cdiv(w, 16 * 4), cdiv(h, 4), 16, 4, (4 + 2) * 16 * 4 * sizeof(float), args)) return 1;
  }
  {
    void* args[6] = {&output, &temp, &iw, &ih, &pitch, &radius};
    if (launch(g.f_cols, cdiv(w, 4), cdiv(h, 16 * 4), 4, 16, 4 * (4 + 2) * 16 * sizeof(float), args)) return 1;
  }
  return 0;
}

REF_API int refk_gaussian_taps(float sigma, float* taps, int* radius) { return gaussian_taps(sigma, taps, radius); }

REF_API int refk_convolution(const float* input, float* output, float* temp, size_t w, size_t h, float sigma) {
  float taps[51];
  int radius = 0;
  if (gaussian_taps(sigma, taps, &radius)) return 3;
  return refk_convolution_taps(input, output, temp, w, h, taps, radius);
}

// cuda_operation_median_2d.cpp:77-155.  Returns 0 launched / copied, 4 refused (nothing written).
REF_API int refk_median(const float* input, float* output, size_t w, size_t h, size_t radius) {
  if (int rc = level_ok(w, h)) return rc;
  if (input == output) return 2;
  if (radius == 1) {
    HIP_OK(hipMemcpyDtoD(output, const_cast<float*>(input), plane_bytes()));
    return 0;
  }
  if (radius % 2 == 0) radius -= 1;
  if (!(radius >= 3 && radius <= 7)) return 4;
  int radius_2 = static_cast<int>(radius / 2);
  unsigned lds = (8 + 2 * radius_2) * (8 + 2 * radius_2) * sizeof(float);
  void* args[5] = {&input, &w, &h, &radius, &output};
  return launch(g.f_median, cdiv(w, 8), cdiv(h, 8), 8, 8, lds, args);
}

// cuda_operation_registration_2d.cpp:105-127
REF_API int refk_registration(const float* frame_0, const float* frame_1, const float* flow_u, const float* flow_v,
                              float* output, size_t w, size_t h, float hx, float hy) {
  if (int rc = level_ok(w, h)) return rc;
  if (frame_1 == output) return 2;
  void* args[9] = {&frame_0, &frame_1, &flow_u, &flow_v, &w, &h, &hx, &hy, &output};
  return launch(g.f_reg, cdiv(w, 16), cdiv(h, 8), 16, 8, 0, args);
}

// cuda_operation_resample_2d.cpp:76-152: x pass into temp (rw x h), y pass into output (rw x rh)
REF_API int refk_resample(const float* input, float* output, float* temp, size_t w, size_t h, size_t rw, size_t rh) {
  if (int rc = level_ok(w, h)) return rc;
  if (int rc = level_ok(rw, rh)) return rc;
  if (input == output) return 2;
  {
    void* args[5] = {&input, &temp, &rw, &h, &w};
    if (launch(g.f_res_x, cdiv(rw, 16), cdiv(h, 8), 16, 8, 0, args)) return 1;
  }
  {
    void* args[5] = {&temp, &output, &rw, &rh, &h};
    if (launch(g.f_res_y, cdiv(rw, 16), cdiv(rh, 8), 16, 8, 0, args)) return 1;
  }
  return 0;
}

// cuda_operation_solve_2d.cpp:239-261 (six LDS planes of 18 x 10)
REF_API int refk_phi_ksi(const float* f0, const float* f1, const float* u, const float* v, const float* du,
                         const float* dv, size_t w, size_t h, float hx, float hy, float e_smooth, float e_data,
                         float* phi, float* ksi) {
  if (int rc = level_ok(w, h)) return rc;
  void* args[14] = {&f0, &f1, &u, &v, &du, &dv, &w, &h, &hx, &hy, &e_smooth, &e_data, &phi, &ksi};
  return launch(g.f_phi_ksi, cdiv(w, 16), cdiv(h, 8), 16, 8, 18 * 10 * sizeof(float) * 6, args);
}

// one sweep, cuda_operation_solve_2d.cpp:263-287; kernel and LDS plane count by constancy (:65-82,:181-198)
// constancy: 0 Grey, 1 Gradient, 2 LogDerivatives (data_structs.h:27)
REF_API int refk_sweep(int constancy, const float* f0, const float* f1, const float* u, const float* v,
                       const float* du, const float* dv, const float* phi, const float* ksi, size_t w, size_t h,
                       float hx, float hy, float alpha, float* temp_du, float* temp_dv) {
  if (int rc = level_ok(w, h)) return rc;
  if (constancy < 0 || constancy > 2) return 6;
  hipFunction_t f = constancy == 1 ? g.f_solve_grad : constancy == 2 ? g.f_solve_log : g.f_solve;
  unsigned planes = constancy == 0 ? 8 : 11;
  void* args[15] = {&f0, &f1, &u, &v, &du, &dv, &phi, &ksi, &w, &h, &hx, &hy, &alpha, &temp_du, &temp_dv};
  return launch(f, cdiv(w, 16), cdiv(h, 8), 16, 8, 18 * 10 * sizeof(float) * planes, args);
}

// CudaOperationSolve2D::Execute, cuda_operation_solve_2d.cpp:106-314: zero du/dv, outer x [phi/ksi,
// inner x (sweep, swap, cuStreamSynchronize)], events around it.  The four increment planes are
// taken by pointer and swapped in place; the result is in *du / *dv afterwards.
REF_API int refk_solve(const float* f0, const float* f1, const float* u, const float* v, float** du, float** dv,
                       float* phi, float* ksi, float** temp_du, float** temp_dv, size_t w, size_t h, float hx, float hy,
                       int constancy, size_t outer, size_t inner, float alpha, float e_smooth, float e_data,
                       float* elapsed_ms) {
  hipEvent_t start, stop;
  HIP_OK(hipEventCreate(&start));
  HIP_OK(hipEventCreate(&stop));
  HIP_OK(hipEventRecord(start, nullptr));
  HIP_OK(hipMemset2D(*du, g.container.pitch, 0, w * sizeof(float), g.container.height));
  HIP_OK(hipMemset2D(*dv, g.container.pitch, 0, w * sizeof(float), g.container.height));
  for (size_t i = 0; i < outer; ++i) {
    if (refk_phi_ksi(f0, f1, u, v, *du, *dv, w, h, hx, hy, e_smooth, e_data, phi, ksi)) return 1;
    for (size_t j = 0; j < inner; ++j) {
      if (refk_sweep(constancy, f0, f1, u, v, *du, *dv, phi, ksi, w, h, hx, hy, alpha, *temp_du, *temp_dv)) return 1;
      std::swap(*du, *temp_du);
      std::swap(*dv, *temp_dv);
      HIP_OK(hipStreamSynchronize(nullptr));
    }
  }
  HIP_OK(hipEventRecord(stop, nullptr));
  HIP_OK(hipEventSynchronize(stop));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, start, stop));
  if (elapsed_ms) *elapsed_ms = ms;
  (void)hipEventDestroy(start);
  (void)hipEventDestroy(stop);
  return 0;
}

struct refk_flow_params {  // the nine bag values of optical_flow_2d.cpp:160-168 + the constancy of Initialize
  size_t warp_levels_count;
  float warp_scale_factor;
  size_t outer_iterations_count;
  size_t inner_iterations_count;
  float equation_alpha;
  float equation_smoothness;
  float equation_data;
  size_t median_radius;
  float gaussian_sigma;
  int data_constancy;
};

// optical_flow_base_2d.cpp:36-59
static size_t max_warp_level(size_t width, size_t height, float scale_factor) {
  size_t r_width = 1, r_height = 1, level_counter = 1;
  while (scale_factor < 1.f) {
    float scale = std::pow(scale_factor, static_cast<float>(level_counter));
    r_width = static_cast<size_t>(std::ceil(width * scale));
    r_height = static_cast<size_t>(std::ceil(height * scale));
    if (r_width < 4 || r_height < 4) break;
    ++level_counter;
  }
  if (r_width == 1 || r_height == 1) --level_counter;
  return level_counter;
}

// OpticalFlow2D::ComputeFlow, optical_flow_2d.cpp:142-569, over the reference's kernels.
// Times: whole call incl. H<->D (the reference's own bracket, :173-179 / :548-554) and the finest
// level's solve (cuda_operation_solve_2d.cpp:220,302).
REF_API int refk_compute_flow(const float* frame_0, const float* frame_1, float* flow_u, float* flow_v,
                              const refk_flow_params* p, float* total_ms, float* finest_solve_ms) {
  if (!g.open) return 1;
  const size_t W = g.container.width, H = g.container.height;
  while (g.pool.size() < 12) {
    float* plane = refk_plane_alloc();
    if (!plane) return fail(hipErrorOutOfMemory, "pool");
    g.pool.push_back(plane);
  }
  size_t max_level = max_warp_level(W, H, p->warp_scale_factor);
  int level = static_cast<int>(std::min(p->warp_levels_count, max_level)) - 1;
  if (level < 0 || !(p->warp_scale_factor < 1.f)) return 5;

  hipEvent_t start, stop;
  HIP_OK(hipEventCreate(&start));
  HIP_OK(hipEventCreate(&stop));
  HIP_OK(hipEventRecord(start, nullptr));

  std::vector<float*> stack(g.pool.rbegin(), g.pool.rend());  // LIFO of free containers (:194-211,:560-567)
  auto pop = [&]() {
    float* t = stack.back();
    stack.pop_back();
    return t;
  };
  float* f0 = pop();
  float* f1 = pop();
  float* f0r = pop();
  float* f1r = pop();
  float* u = pop();
  float* v = pop();
  float* du = pop();
  float* dv = pop();

  if (refk_upload_2d(f0, frame_0, W, H) || refk_upload_2d(f1, frame_1, W, H)) return 1;

  if (p->gaussian_sigma > 0.0) {  // :218-246
    float* temp = pop();
    if (refk_convolution(f0, u, temp, W, H, p->gaussian_sigma)) return 1;
    if (refk_convolution(f1, v, temp, W, H, p->gaussian_sigma)) return 1;
    stack.push_back(temp);
    std::swap(f0, u);
    std::swap(f1, v);
  }

  size_t pw = 0, ph = 0;
  while (level >= 0) {
    float scale = std::pow(p->warp_scale_factor, static_cast<float>(level));  // :268-272
    size_t cw = static_cast<size_t>(std::ceil(W * scale));
    size_t ch = static_cast<size_t>(std::ceil(H * scale));
    float hx = W / static_cast<float>(cw);
    float hy = H / static_cast<float>(ch);

    if (level == 0) {  // :280-305
      std::swap(f0, f0r);
      std::swap(f1, f1r);
    } else {
      float* temp = pop();
      if (refk_resample(f0, f0r, temp, W, H, cw, ch) || refk_resample(f1, f1r, temp, W, H, cw, ch)) return 1;
      stack.push_back(temp);
    }

    if (pw == 0) {  // :309-313
      HIP_OK(hipMemset2D(u, g.container.pitch, 0, W * sizeof(float), H));
      HIP_OK(hipMemset2D(v, g.container.pitch, 0, W * sizeof(float), H));
    } else {  // :316-339
      float* temp = pop();
      if (refk_resample(u, du, temp, pw, ph, cw, ch) || refk_resample(v, dv, temp, pw, ph, cw, ch)) return 1;
      std::swap(u, du);
      std::swap(v, dv);
      stack.push_back(temp);
    }

    {  // :344-363
      float* temp = pop();
      if (refk_registration(f0r, f1r, u, v, temp, cw, ch, hx, hy)) return 1;
      std::swap(f1r, temp);
      stack.push_back(temp);
    }

    {  // :366-406
      float* phi = pop();
      float* ksi = pop();
      float* tdu = pop();
      float* tdv = pop();
      float ms = 0.f;
      if (refk_solve(f0r, f1r, u, v, &du, &dv, phi, ksi, &tdu, &tdv, cw, ch, hx, hy, p->data_constancy,
                     p->outer_iterations_count, p->inner_iterations_count, p->equation_alpha,
                     p->equation_smoothness, p->equation_data, &ms))
        return 1;
      if (level == 0 && finest_solve_ms) *finest_solve_ms = ms;
      stack.push_back(tdv);
      stack.push_back(tdu);
      stack.push_back(ksi);
      stack.push_back(phi);
    }

    if (refk_add(u, du, cw, ch) || refk_add(v, dv, cw, ch)) return 1;  // :409-422
    pw = cw;
    ph = ch;
    --level;

    {  // :428-449 -- the output plane is swapped in even when the operator refused the width
      float* temp = pop();
      int rc = refk_median(u, temp, cw, ch, p->median_radius);
      if (rc == 1) return 1;
      std::swap(u, temp);
      rc = refk_median(v, temp, cw, ch, p->median_radius);
      if (rc == 1) return 1;
      std::swap(v, temp);
      stack.push_back(temp);
    }
  }

  if (refk_download_2d(u, flow_u, W, H) || refk_download_2d(v, flow_v, W, H)) return 1;  // :544-545
  HIP_OK(hipEventRecord(stop, nullptr));
  HIP_OK(hipEventSynchronize(stop));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, start, stop));
  if (total_ms) *total_ms = ms;
  (void)hipEventDestroy(start);
  (void)hipEventDestroy(stop);
  return 0;
}
