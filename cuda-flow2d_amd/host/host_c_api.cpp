// C facade over the C++ host layer, for the Python tests and bench.py (plumbing only).
// It drives the same OpticalFlow2D / OperationParameters objects a C++ caller would.
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "device_utils.h"
#include "io_utils.h"
#include "optical_flow_2d.h"
#include "optical_flow_batch_2d.h"
#include "settings.h"

#define HOST_API extern "C" __attribute__((visibility("default")))

struct flow2d_host_params {
    size_t warp_levels_count;
    float warp_scale_factor;
    size_t outer_iterations_count;
    size_t inner_iterations_count;
    float equation_alpha;
    float equation_smoothness;
    float equation_data;
    size_t median_radius;
    float gaussian_sigma;
    int solver_algorithm;
    float sor_omega;
};

struct flow2d_host_flow {
    OpticalFlow2D flow;
    size_t width = 0, height = 0;
};

namespace {
void FillBag(OperationParameters& bag, flow2d_host_params& p)
{
    bag.PushValuePtr("warp_levels_count", &p.warp_levels_count);
    bag.PushValuePtr("warp_scale_factor", &p.warp_scale_factor);
    bag.PushValuePtr("outer_iterations_count", &p.outer_iterations_count);
    bag.PushValuePtr("inner_iterations_count", &p.inner_iterations_count);
    bag.PushValuePtr("equation_alpha", &p.equation_alpha);
    bag.PushValuePtr("equation_smoothness", &p.equation_smoothness);
    bag.PushValuePtr("equation_data", &p.equation_data);
    bag.PushValuePtr("median_radius", &p.median_radius);
    bag.PushValuePtr("gaussian_sigma", &p.gaussian_sigma);
    bag.PushValuePtr("solver_algorithm", &p.solver_algorithm);
    bag.PushValuePtr("solver_sor_omega", &p.sor_omega);
}
}  // namespace

// Process-wide context on `device` (InitDeviceContext).  0 on success.
HOST_API int flow2d_host_init_device(int device) { return InitDeviceContext(device) ? 0 : 1; }
HOST_API void flow2d_host_adopt_context(flow2d_context* ctx) { AdoptDeviceContext(ctx); }
HOST_API flow2d_context* flow2d_host_context(void) { return CurrentDeviceContext(); }
HOST_API void flow2d_host_shutdown(void) { DestroyDeviceContext(); }

// constancy: enum class DataConstancy (0 Grey, 1 Gradient, 2 LogDerivatives, 3 GradientUntiled).  nullptr on failure.
// lone: OpticalFlow2D::lone (1: the object's pairs run alone on the device -- packed strip kernel for under-filled launches; 0: one
// worker among several)
HOST_API flow2d_host_flow* flow2d_host_flow_create(size_t width, size_t height, int constancy, int silent, int lone)
{
    flow2d_host_flow* h = new (std::nothrow) flow2d_host_flow();
    if (!h) return nullptr;
    h->flow.silent = silent != 0;
    h->flow.lone = lone != 0;
    DataSize3 size = {width, height, 1};
    if (!h->flow.Initialize(size, static_cast<DataConstancy>(constancy))) {
        delete h;
        return nullptr;
    }
    h->width = width;
    h->height = height;
    return h;
}

HOST_API void flow2d_host_flow_destroy(flow2d_host_flow* h)
{
    if (!h) return;
    h->flow.Destroy();
    delete h;
}

HOST_API size_t flow2d_host_flow_pitch(flow2d_host_flow* h) { return h ? h->flow.ContainerSize().pitch : 0; }

HOST_API size_t flow2d_host_max_warp_level(flow2d_host_flow* h, size_t width, size_t height, float scale)
{
    return h ? h->flow.GetMaxWarpLevel(width, height, scale) : 0;
}

// GetMaxWarpLevel needs no device (optical_flow_base_2d.cpp:36-59 is pure host arithmetic)
HOST_API size_t flow2d_host_max_warp_level_static(size_t width, size_t height, float scale)
{
    OpticalFlow2D flow;
    return flow.GetMaxWarpLevel(width, height, scale);
}

// OpticalFlow2D::ComputeFlow on tight host images (width*height floats each).  0 on success, 2 when the run
// delivered no flow.
HOST_API int flow2d_host_compute_flow(flow2d_host_flow* h, const float* frame_0, const float* frame_1, float* flow_u,
                                      float* flow_v, const flow2d_host_params* params, float* total_ms)
{
    if (!h || !frame_0 || !frame_1 || !flow_u || !flow_v || !params) return 1;
    const size_t n = h->width * h->height;
    Data2D f0(h->width, h->height), f1(h->width, h->height), u(h->width, h->height), v(h->width, h->height);
    std::memcpy(f0.DataPtr(), frame_0, n * sizeof(float));
    std::memcpy(f1.DataPtr(), frame_1, n * sizeof(float));
    // poison the outputs so that an aborted run (missing key, bad parameter) is visible to the caller
    for (size_t i = 0; i < n; ++i) u.DataPtr()[i] = v.DataPtr()[i] = -12345.f;
    flow2d_host_params p = *params;
    OperationParameters bag;
    FillBag(bag, p);
    h->flow.ComputeFlow(f0, f1, u, v, bag);
    std::memcpy(flow_u, u.DataPtr(), n * sizeof(float));
    std::memcpy(flow_v, v.DataPtr(), n * sizeof(float));
    if (total_ms) *total_ms = h->flow.LastTotalMs();
    return h->flow.LastRunSucceeded() ? 0 : 2;  // 2: the run was refused or an operator failed (outputs keep the poison)
}

// OpticalFlow2D::ComputeFlowDevice: frames and flow already in pitched device containers.  Queued on
// the context's stream, no synchronisation.  0 on success.
HOST_API int flow2d_host_compute_flow_device(flow2d_host_flow* h, void* dev_frame_0, void* dev_frame_1,
                                             void* dev_flow_u, void* dev_flow_v, const flow2d_host_params* params,
                                             int timing_mode)
{
    if (!h || !params) return 1;
    flow2d_host_params p = *params;
    OperationParameters bag;
    FillBag(bag, p);
    h->flow.timing_mode = timing_mode;
    auto dp = [](void* q) { return static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(q)); };
    return h->flow.ComputeFlowDevice(dp(dev_frame_0), dp(dev_frame_1), dp(dev_flow_u), dp(dev_flow_v), bag) ? 0 : 2;
}

// OpticalFlow2D::ComputeFlowSequenceDevice: frame_count device frames, frame_count - 1 flow plane pairs.
HOST_API int flow2d_host_compute_flow_sequence_device(flow2d_host_flow* h, void* const* dev_frames, size_t frame_count,
                                                      void* const* dev_flows_u, void* const* dev_flows_v,
                                                      const flow2d_host_params* params)
{
    if (!h || !params || !dev_frames || !dev_flows_u || !dev_flows_v || frame_count < 2) return 1;
    flow2d_host_params p = *params;
    OperationParameters bag;
    FillBag(bag, p);
    h->flow.timing_mode = 0;
    auto dp = [](void* q) { return static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(q)); };
    std::vector<DevicePtr> frames(frame_count), us(frame_count - 1), vs(frame_count - 1);
    for (size_t k = 0; k < frame_count; ++k) frames[k] = dp(dev_frames[k]);
    for (size_t k = 0; k + 1 < frame_count; ++k) {
        us[k] = dp(dev_flows_u[k]);
        vs[k] = dp(dev_flows_v[k]);
    }
    return h->flow.ComputeFlowSequenceDevice(frames.data(), frame_count, us.data(), vs.data(), bag) ? 0 : 2;
}

// ---- OpticalFlowBatch2D: pairs spread over lanes (stream + OpticalFlow2D + plane pool each) on one GPU -----------
struct flow2d_host_batch {
    OpticalFlowBatch2D batch;
};

HOST_API flow2d_host_batch* flow2d_host_batch_create(size_t width, size_t height, int constancy, size_t lanes, int device,
                                                     size_t group_size)
{
    flow2d_host_batch* h = new (std::nothrow) flow2d_host_batch();
    if (!h) return nullptr;
    DataSize3 size = {width, height, 1};
    if (!h->batch.Initialize(size, static_cast<DataConstancy>(constancy), lanes, device, group_size)) {
        delete h;
        return nullptr;
    }
    return h;
}

HOST_API void flow2d_host_batch_destroy(flow2d_host_batch* h)
{
    if (!h) return;
    h->batch.Destroy();
    delete h;
}

HOST_API size_t flow2d_host_batch_pitch(flow2d_host_batch* h) { return h ? h->batch.ContainerSize().pitch : 0; }
HOST_API size_t flow2d_host_batch_lanes(flow2d_host_batch* h) { return h ? h->batch.Lanes() : 0; }
HOST_API size_t flow2d_host_batch_group_stride(flow2d_host_batch* h) { return h ? h->batch.GroupStrideBytes() : 0; }
HOST_API flow2d_context* flow2d_host_batch_lane_context(flow2d_host_batch* h, size_t lane)
{
    return h ? h->batch.LaneContext(lane) : nullptr;
}
HOST_API void flow2d_host_batch_use_graph(flow2d_host_batch* h, int on)
{
    if (h) h->batch.use_graph = on != 0;
}

// OpticalFlowBatch2D::ComputeFlowBatchDevice: `count` pairs, pair k on lane (first_lane + k) mod lanes.  0 on success.
HOST_API int flow2d_host_batch_compute(flow2d_host_batch* h, size_t count, void* const* dev_frames_0,
                                       void* const* dev_frames_1, void* const* dev_flows_u, void* const* dev_flows_v,
                                       const flow2d_host_params* params, size_t first_lane)
{
    if (!h || !params || (count && (!dev_frames_0 || !dev_frames_1 || !dev_flows_u || !dev_flows_v))) return 1;
    flow2d_host_params p = *params;
    OperationParameters bag;
    FillBag(bag, p);
    auto dp = [](void* q) { return static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(q)); };
    std::vector<DevicePtr> f0(count), f1(count), u(count), v(count);
    for (size_t k = 0; k < count; ++k) {
        f0[k] = dp(dev_frames_0[k]);
        f1[k] = dp(dev_frames_1[k]);
        u[k] = dp(dev_flows_u[k]);
        v[k] = dp(dev_flows_v[k]);
    }
    return h->batch.ComputeFlowBatchDevice(count, f0.data(), f1.data(), u.data(), v.data(), bag, first_lane) ? 0 : 2;
}

// OpticalFlowBatch2D::ComputeFlowBatchDeviceGrouped: `count` independent pairs, grouped by the object.  0 on success.
HOST_API int flow2d_host_batch_compute_grouped(flow2d_host_batch* h, size_t count, void* const* dev_frames_0,
                                               void* const* dev_frames_1, void* const* dev_flows_u,
                                               void* const* dev_flows_v, const flow2d_host_params* params,
                                               size_t first_lane)
{
    if (!h || !params || (count && (!dev_frames_0 || !dev_frames_1 || !dev_flows_u || !dev_flows_v))) return 1;
    flow2d_host_params p = *params;
    OperationParameters bag;
    FillBag(bag, p);
    auto dp = [](void* q) { return static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(q)); };
    std::vector<DevicePtr> f0(count), f1(count), u(count), v(count);
    for (size_t k = 0; k < count; ++k) {
        f0[k] = dp(dev_frames_0[k]);
        f1[k] = dp(dev_frames_1[k]);
        u[k] = dp(dev_flows_u[k]);
        v[k] = dp(dev_flows_v[k]);
    }
    return h->batch.ComputeFlowBatchDeviceGrouped(count, f0.data(), f1.data(), u.data(), v.data(), bag, first_lane) ? 0 : 2;
}

// Host images for the H<->D-inclusive entry: Data2D objects in pageable or page-locked memory (HostMemory::Pinned, the
// reference's ALLOCATE_PINNED_MEMORY option as a run-time choice).  The caller fills / reads them through the pointer.
HOST_API Data2D* flow2d_host_data2d_create(size_t width, size_t height, int pinned)
{
    Data2D* d = new (std::nothrow) Data2D(width, height, pinned ? HostMemory::Pinned : HostMemory::Pageable);
    if (d && !d->DataPtr()) {
        delete d;
        d = nullptr;
    }
    return d;
}
HOST_API void flow2d_host_data2d_destroy(Data2D* d) { delete d; }
HOST_API void flow2d_host_use_pinned_memory(int on) { Data2D::UsePinnedMemory(on != 0); }
HOST_API float* flow2d_host_data2d_ptr(Data2D* d) { return d ? d->DataPtr() : nullptr; }
HOST_API int flow2d_host_data2d_is_pinned(Data2D* d) { return d && d->IsPinned() ? 1 : 0; }

// OpticalFlowBatch2D::ComputeFlowBatch: `count` pairs of host images in, host flows out, uploads and downloads
// pipelined against the lanes' pyramids.  Queued: the flows are complete after flow2d_host_batch_synchronize.
HOST_API int flow2d_host_batch_compute_host(flow2d_host_batch* h, size_t count, Data2D* const* frames_0,
                                            Data2D* const* frames_1, Data2D* const* flows_u, Data2D* const* flows_v,
                                            const flow2d_host_params* params, size_t first_lane)
{
    if (!h || !params) return 1;
    flow2d_host_params p = *params;
    OperationParameters bag;
    FillBag(bag, p);
    return h->batch.ComputeFlowBatch(count, frames_0, frames_1, flows_u, flows_v, bag, first_lane) ? 0 : 2;
}

HOST_API int flow2d_host_batch_synchronize(flow2d_host_batch* h) { return (h && h->batch.Synchronize()) ? 0 : 1; }

// Per-level solve records of the last run (needs timing_mode >= 1 and a synchronised context).
// Writes up to `capacity` records of 7 floats (width, height, solve_ms, kernel_ms, kernel_launches,
// algorithmic bytes per launch, algorithm used) and returns the number of levels.
HOST_API size_t flow2d_host_level_timings(flow2d_host_flow* h, float* records, size_t capacity)
{
    if (!h) return 0;
    std::vector<FlowLevelTiming> t = h->flow.LastLevelTimings();
    for (size_t i = 0; i < t.size() && i < capacity; ++i) {
        records[7 * i + 0] = static_cast<float>(t[i].width);
        records[7 * i + 1] = static_cast<float>(t[i].height);
        records[7 * i + 2] = t[i].solve_ms;
        records[7 * i + 3] = t[i].kernel_ms;
        records[7 * i + 4] = static_cast<float>(t[i].kernel_launches);
        records[7 * i + 5] = static_cast<float>(t[i].bytes_per_launch);
        records[7 * i + 6] = static_cast<float>(t[i].algorithm);
    }
    return t.size();
}

HOST_API void flow2d_host_use_graph(flow2d_host_flow* h, int on)
{
    if (h) h->flow.use_graph = on != 0;
}

HOST_API void flow2d_host_reset_timings(flow2d_host_flow* h)
{
    if (h) h->flow.ResetLevelTimings();
}

// Omitting a bag key must make ComputeFlow print and return with the outputs untouched
// (optical_flow_2d.cpp:160-168).  Returns 1 if the outputs were left untouched.
HOST_API int flow2d_host_missing_key_leaves_outputs(flow2d_host_flow* h, const char* omitted_key)
{
    if (!h) return -1;
    Data2D f0(h->width, h->height), f1(h->width, h->height), u(h->width, h->height), v(h->width, h->height);
    const size_t n = h->width * h->height;
    for (size_t i = 0; i < n; ++i) u.DataPtr()[i] = v.DataPtr()[i] = 77.f;
    flow2d_host_params p = {3, 0.5f, 1, 1, 3.5f, 0.001f, 0.001f, 5, 0.45f, 0, 0.f};
    OperationParameters full, bag;
    FillBag(full, p);
    const char* keys[] = {"warp_levels_count", "warp_scale_factor", "outer_iterations_count",
                          "inner_iterations_count", "equation_alpha", "equation_smoothness",
                          "equation_data", "median_radius", "gaussian_sigma"};
    for (const char* k : keys)
        if (std::strcmp(k, omitted_key) != 0) bag.PushValuePtr(k, full.GetValuePtr(k));
    h->flow.ComputeFlow(f0, f1, u, v, bag);
    for (size_t i = 0; i < n; ++i)
        if (u.DataPtr()[i] != 77.f || v.DataPtr()[i] != 77.f) return 0;
    return 1;
}

// ---- small helpers so the tests can reach Data2D / Settings / IOUtils ---------------------------------
HOST_API int flow2d_host_read_raw(const char* path, size_t width, size_t height, int u8, float* out)
{
    Data2D d;
    const bool ok = u8 ? d.ReadRAWFromFileU8(path, width, height) : d.ReadRAWFromFileF32(path, width, height);
    if (!ok) return 1;
    std::memcpy(out, d.DataPtr(), width * height * sizeof(float));
    return 0;
}

// Data2D::WriteRAWToFileU8 / WriteRAWToFileF32 of a tight width x height image.  0 on success.
HOST_API int flow2d_host_write_raw(const float* data, size_t width, size_t height, int u8, const char* path)
{
    Data2D d(width, height);
    std::memcpy(d.DataPtr(), data, width * height * sizeof(float));
    return (u8 ? d.WriteRAWToFileU8(path) : d.WriteRAWToFileF32(path)) ? 0 : 1;
}

HOST_API int flow2d_host_write_outputs(const float* u, const float* v, size_t width, size_t height,
                                       const char* ppm_path, const char* amp_path, float flow_max_scale)
{
    Data2D du(width, height), dv(width, height);
    std::memcpy(du.DataPtr(), u, width * height * sizeof(float));
    std::memcpy(dv.DataPtr(), v, width * height * sizeof(float));
    IOUtils::WriteFlowToImageRGB(du, dv, flow_max_scale, ppm_path);
    IOUtils::WriteMagnitudeToFileF32(du, dv, amp_path);
    return 0;
}

HOST_API void flow2d_host_convert_to_rgb(float x, float y, int* rgb)
{
    const IOUtils::RGBColor c = IOUtils::ConvertToRGB(x, y);
    rgb[0] = c.r;
    rgb[1] = c.g;
    rgb[2] = c.b;
}

struct flow2d_host_settings {
    int width, height, medianRadius, iterInner, iterOuter, levels, press_key;
    float sigma, alpha, e_smooth, e_data, warpScale;
    char inputPath[512], outputPath[512], fileName1[256], fileName2[256], imageType[32], dataConstancy[32];
};

HOST_API int flow2d_host_load_settings(const char* path, flow2d_host_settings* out)
{
    OpticFlow::Settings s;
    const int rc = s.LoadSettings(path);
    if (rc != 0) return rc;
    out->width = s.width;
    out->height = s.height;
    out->medianRadius = s.medianRadius;
    out->iterInner = s.iterInner;
    out->iterOuter = s.iterOuter;
    out->levels = s.levels;
    out->press_key = s.press_key;
    out->sigma = s.sigma;
    out->alpha = s.alpha;
    out->e_smooth = s.e_smooth;
    out->e_data = s.e_data;
    out->warpScale = s.warpScale;
    std::snprintf(out->inputPath, sizeof(out->inputPath), "%s", s.inputPath.c_str());
    std::snprintf(out->outputPath, sizeof(out->outputPath), "%s", s.outputPath.c_str());
    std::snprintf(out->fileName1, sizeof(out->fileName1), "%s", s.fileName1.c_str());
    std::snprintf(out->fileName2, sizeof(out->fileName2), "%s", s.fileName2.c_str());
    std::snprintf(out->imageType, sizeof(out->imageType), "%s", s.imageType.c_str());
    std::snprintf(out->dataConstancy, sizeof(out->dataConstancy), "%s", s.dataConstancy.c_str());
    return 0;
}

// ---- operator-level access: the reference's plugin API (CudaOperationBase::Initialize / Execute with a
// string-keyed bag of void*) driven from the tests.  `kind`: add, convolution, median, registration, resample,
// solve.  The bag is given as parallel arrays of keys and pointers to the values (exactly what PushValuePtr takes).
#include <memory>
#include <string>

struct flow2d_host_operator {
    std::unique_ptr<CudaOperationBase> op;
    DataSize3 container;
    DataConstancy constancy;
    flow2d_context* ctx;
};

HOST_API flow2d_host_operator* flow2d_host_operator_create(const char* kind, size_t container_width,
                                                           size_t container_height, size_t pitch_bytes, int constancy,
                                                           int omit_container_size)
{
    auto h = std::make_unique<flow2d_host_operator>();
    const std::string k = kind ? kind : "";
    if (k == "add") h->op.reset(new CudaOperationAdd2D());
    else if (k == "convolution") h->op.reset(new CudaOperationConvolution2D());
    else if (k == "median") h->op.reset(new CudaOperationMedian2D());
    else if (k == "registration") h->op.reset(new CudaOperationRegistration2D());
    else if (k == "resample") h->op.reset(new CudaOperationResample2D());
    else if (k == "solve") h->op.reset(new CudaOperationSolve2D());
    else return nullptr;
    h->container = {container_width, container_height, pitch_bytes};
    h->constancy = static_cast<DataConstancy>(constancy);
    h->ctx = CurrentDeviceContext();
    OperationParameters init;
    if (!omit_container_size) init.PushValuePtr("container_size", &h->container);
    init.PushValuePtr("data_constancy", &h->constancy);
    if (!h->op->Initialize(&init)) return nullptr;
    return h.release();
}

HOST_API const char* flow2d_host_operator_name(flow2d_host_operator* h) { return h ? h->op->GetName() : ""; }

HOST_API void flow2d_host_operator_execute(flow2d_host_operator* h, const char* const* keys, void* const* values,
                                           size_t count)
{
    if (!h) return;
    OperationParameters bag;
    for (size_t i = 0; i < count; ++i) bag.PushValuePtr(keys[i], values[i]);
    h->op->Execute(bag);
}

HOST_API void flow2d_host_operator_destroy(flow2d_host_operator* h)
{
    if (!h) return;
    h->op->Destroy();
    delete h;
}

// OperationParameters semantics (operation_parameters.cpp:28-47): no overwrite, nullptr for a missing key.
HOST_API int flow2d_host_bag_selftest(void)
{
    OperationParameters bag;
    int a = 1, b = 2;
    if (!bag.PushValuePtr("k", &a)) return 1;
    if (bag.PushValuePtr("k", &b)) return 2;            // an existing key is kept
    if (bag.GetValuePtr("k") != &a) return 3;
    if (bag.GetValuePtr("missing") != nullptr) return 4;
    bag.Clear();
    if (bag.GetValuePtr("k") != nullptr) return 5;
    return 0;
}
