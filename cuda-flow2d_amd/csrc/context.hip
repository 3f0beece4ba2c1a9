// Device / context / pitched memory / events of the flow2d C-ABI.
// Replaces the CUDA driver calls of src/utils/cuda_utils.cpp:26-105 and
// src/optical_flow/optical_flow_2d.cpp:84-140,309-312,574-577 of the reference.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unistd.h>

#include "common.hpp"

namespace {
thread_local std::string g_last_error;

// Has this process started the HIP runtime?  The runtime opens the kernel driver's device node when it initialises and keeps it
// open: a descriptor of /dev/kfd among the process's own is the sign (no HIP call can answer the question without starting it).
bool hip_runtime_started()
{
    char link[64], target[64];
    for (int fd = 0; fd < 1024; ++fd) {
        std::snprintf(link, sizeof(link), "/proc/self/fd/%d", fd);
        const ssize_t n = readlink(link, target, sizeof(target) - 1);
        if (n > 0 && std::string(target, static_cast<size_t>(n)) == "/dev/kfd") return true;
    }
    return false;
}
}

namespace flow2d {

void set_last_error(const char* what, hipError_t err)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(err);
}

void set_last_error_text(const char* what) { g_last_error = what; }

DeviceGuard::DeviceGuard(const flow2d_context* ctx)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) {
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) {
            set_last_error("hipSetDevice", e);
            ok_ = false;
        }
    }
}

}  // namespace flow2d

extern "C" {

int flow2d_abi_version(void) { return FLOW2D_ABI_VERSION; }

const char* flow2d_status_string(int status)
{
    switch (status) {
        case FLOW2D_OK: return "ok";
        case FLOW2D_ERR_INVALID_ARGUMENT: return "invalid argument";
        case FLOW2D_ERR_NO_DEVICE: return "no usable HIP device";
        case FLOW2D_ERR_DEVICE: return "HIP runtime error";
        case FLOW2D_ERR_OUT_OF_MEMORY: return "out of device memory";
        case FLOW2D_ERR_UNSUPPORTED: return "unsupported parameter";
        default: return "unknown status";
    }
}

const char* flow2d_last_error(void) { return g_last_error.c_str(); }

int flow2d_hw_queues(void)
{
    const char* v = std::getenv("GPU_MAX_HW_QUEUES");
    const int n = v ? std::atoi(v) : 0;
    return n > 0 ? n : 4;
}

// The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable once, when
// it initialises (the first HIP call of the process).  The batched host path keeps four lanes (streams) busy next to whatever
// other stream the process owns (RCCL's, the caller's): with the default, two lanes share a queue and wait behind each other
// (4096^2, four lanes: 205 instead of 224 pairs/s).  The host layer therefore asks for eight queues before its first HIP call.
// (Rounds 2-5 did it from a library constructor -- a side effect of dlopen on the host process that silently did nothing when
// HIP was already running; now it is a call, and it says when it came too late.)
int flow2d_request_hw_queues(int queues)
{
    if (queues < 1) return FLOW2D_ERR_INVALID_ARGUMENT;
    const bool set_by_caller = std::getenv("GPU_MAX_HW_QUEUES") != nullptr;
    if (flow2d_hw_queues() >= queues && (set_by_caller || queues <= 4)) return FLOW2D_OK;  // (unset: the runtime's default of 4)
    if (set_by_caller) {
        flow2d::set_last_error_text("GPU_MAX_HW_QUEUES is set to fewer queues than requested; the caller's value stands");
        return FLOW2D_ERR_UNSUPPORTED;
    }
    if (hip_runtime_started()) {
        flow2d::set_last_error_text("the HIP runtime of this process was started before flow2d_request_hw_queues: it keeps its default of 4 "
                                    "hardware queues (export GPU_MAX_HW_QUEUES, or call this before the first HIP call)");
        return FLOW2D_ERR_UNSUPPORTED;
    }
    setenv("GPU_MAX_HW_QUEUES", std::to_string(queues).c_str(), 1);
    return FLOW2D_OK;
}

int flow2d_device_count(int* count)
{
    if (!count) return FLOW2D_ERR_INVALID_ARGUMENT;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        flow2d::set_last_error("hipGetDeviceCount", e);
        *count = 0;
        return FLOW2D_ERR_NO_DEVICE;
    }
    *count = n;
    return FLOW2D_OK;
}

static int create_context(int device_ordinal, void* stream, bool adopt, flow2d_context** out_ctx)
{
    if (!out_ctx) return FLOW2D_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int n = 0;
    int st = flow2d_device_count(&n);
    if (st != FLOW2D_OK) return st;
    if (device_ordinal < 0 || device_ordinal >= n) return FLOW2D_ERR_NO_DEVICE;
    FLOW2D_HIP_TRY(hipSetDevice(device_ordinal));
    flow2d_context* ctx = new (std::nothrow) flow2d_context();
    if (!ctx) return FLOW2D_ERR_OUT_OF_MEMORY;
    ctx->device = device_ordinal;
    if (adopt) {
        ctx->stream = static_cast<hipStream_t>(stream);
        ctx->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            flow2d::set_last_error("hipStreamCreateWithFlags", e);
            delete ctx;
            return FLOW2D_ERR_DEVICE;
        }
        ctx->owns_stream = true;
    }
    // (stream-ordered on the context's own stream: a call on the NULL stream here would bring the legacy default stream
    //  to life, and with it the lanes of a batch lose their overlap -- measured: config 2 at half its rate)
    if (hipMalloc(reinterpret_cast<void**>(&ctx->fused_fallbacks), 2 * sizeof(unsigned int)) != hipSuccess ||
        hipMemsetAsync(ctx->fused_fallbacks, 0, 2 * sizeof(unsigned int), ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        ctx->fused_fallbacks = nullptr;  // diagnostics only: the kernels run without it
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0)
        ctx->num_cus = cus;
    *out_ctx = ctx;
    return FLOW2D_OK;
}

int flow2d_context_create(int device_ordinal, flow2d_context** out_ctx)
{
    return create_context(device_ordinal, nullptr, false, out_ctx);
}

int flow2d_context_create_on_stream(int device_ordinal, void* hip_stream, flow2d_context** out_ctx)
{
    return create_context(device_ordinal, hip_stream, true, out_ctx);
}

int flow2d_context_destroy(flow2d_context* ctx)
{
    FLOW2D_ENTER(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    flow2d_timing_reset(ctx);
    for (hipEvent_t ev : ctx->event_pool) (void)hipEventDestroy(ev);
    ctx->event_pool.clear();
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->fused_fallbacks) (void)hipFree(ctx->fused_fallbacks);
    if (ctx->clock_probe) (void)hipFree(ctx->clock_probe);
    delete ctx;
    return FLOW2D_OK;
}

int flow2d_context_device(const flow2d_context* ctx, int* device_ordinal)
{
    if (!ctx || !device_ordinal) return FLOW2D_ERR_INVALID_ARGUMENT;
    *device_ordinal = ctx->device;
    return FLOW2D_OK;
}

int flow2d_context_stream(const flow2d_context* ctx, void** hip_stream)
{
    if (!ctx || !hip_stream) return FLOW2D_ERR_INVALID_ARGUMENT;
    *hip_stream = ctx->stream;
    return FLOW2D_OK;
}

int flow2d_synchronize(flow2d_context* ctx)
{
    FLOW2D_ENTER(ctx);
    FLOW2D_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FLOW2D_OK;
}

// word 0: guard trips, word 1: waves of plain-only launches (solve_fused_kernel.hpp)
static int read_fused_counter(flow2d_context* ctx, int word, unsigned long long* waves)
{
    FLOW2D_ENTER(ctx);
    if (!waves) return FLOW2D_ERR_INVALID_ARGUMENT;
    *waves = 0;
    if (!ctx->fused_fallbacks) return FLOW2D_OK;
    unsigned int n = 0;
    FLOW2D_HIP_TRY(hipMemcpyAsync(&n, ctx->fused_fallbacks + word, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
    FLOW2D_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *waves = n;
    return FLOW2D_OK;
}

int flow2d_fused_fallbacks(flow2d_context* ctx, unsigned long long* waves) { return read_fused_counter(ctx, 0, waves); }

int flow2d_fused_plain_waves(flow2d_context* ctx, unsigned long long* waves) { return read_fused_counter(ctx, 1, waves); }

// The shader clock the device holds, sampled by one idle wave per XCD beside whatever else runs: each wave reads the constant
// 100 MHz clock and the shader clock, sleeps (s_sleep: no issue slots taken from the kernels it runs beside) until `duration_us`
// have passed, reads both again and files cycles / time under its XCD.  Eight workgroups of one wave are dealt to the eight XCDs.
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long duration_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0;
    while (t1 - t0 < duration_ticks) {
        __builtin_amdgcn_s_sleep(64);
        t1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * (xcc & 7u)] = t1 - t0;
    out[2 * (xcc & 7u) + 1] = c1 - c0;
}
}  // namespace

int flow2d_clock_probe_start(flow2d_context* ctx, double duration_us)
{
    FLOW2D_ENTER(ctx);
    if (!(duration_us > 0.0) || duration_us > 1e6) return FLOW2D_ERR_INVALID_ARGUMENT;
    if (!ctx->clock_probe) FLOW2D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->clock_probe), 16 * sizeof(unsigned long long)));
    FLOW2D_HIP_TRY(hipMemsetAsync(ctx->clock_probe, 0, 16 * sizeof(unsigned long long), ctx->stream));
    clock_probe_kernel<<<8, 64, 0, ctx->stream>>>(ctx->clock_probe, static_cast<unsigned long long>(duration_us * 100.0));
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_clock_probe_read(flow2d_context* ctx, double* ghz_per_xcd)
{
    FLOW2D_ENTER(ctx);
    if (!ghz_per_xcd) return FLOW2D_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 8; ++i) ghz_per_xcd[i] = 0.0;
    if (!ctx->clock_probe) return FLOW2D_OK;
    unsigned long long t[16];
    FLOW2D_HIP_TRY(hipMemcpyAsync(t, ctx->clock_probe, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
    FLOW2D_HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i)
        if (t[2 * i]) ghz_per_xcd[i] = static_cast<double>(t[2 * i + 1]) / (static_cast<double>(t[2 * i]) * 10.0);  // 100 MHz ticks
    return FLOW2D_OK;
}

int flow2d_mem_info(flow2d_context* ctx, size_t* free_bytes, size_t* total_bytes)
{
    FLOW2D_ENTER(ctx);
    if (!free_bytes || !total_bytes) return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return FLOW2D_OK;
}

int flow2d_device_name(flow2d_context* ctx, char* buf, size_t buf_len)
{
    FLOW2D_ENTER(ctx);
    if (!buf || buf_len == 0) return FLOW2D_ERR_INVALID_ARGUMENT;
    hipDeviceProp_t prop;
    FLOW2D_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    std::snprintf(buf, buf_len, "%s (%s)", prop.name, prop.gcnArchName);
    return FLOW2D_OK;
}

size_t flow2d_plane_pitch_bytes(size_t width)
{
    const size_t align = 256;
    return (width * sizeof(float) + align - 1) / align * align;
}

int flow2d_plane_alloc(flow2d_context* ctx, size_t width, size_t height, void** out_dev_ptr, size_t* out_pitch_bytes)
{
    FLOW2D_ENTER(ctx);
    if (!out_dev_ptr || !out_pitch_bytes || width == 0 || height == 0) return FLOW2D_ERR_INVALID_ARGUMENT;
    size_t pitch = flow2d_plane_pitch_bytes(width);
    void* p = nullptr;
    FLOW2D_HIP_TRY(hipMalloc(&p, pitch * height));
    *out_dev_ptr = p;
    *out_pitch_bytes = pitch;
    return FLOW2D_OK;
}

int flow2d_plane_free(flow2d_context* ctx, void* dev_ptr)
{
    FLOW2D_ENTER(ctx);
    if (!dev_ptr) return FLOW2D_OK;
    FLOW2D_HIP_TRY(hipFree(dev_ptr));
    return FLOW2D_OK;
}

int flow2d_memset_2d(flow2d_context* ctx, void* dev_ptr, size_t pitch_bytes, int byte_value, size_t width_bytes,
                     size_t height)
{
    FLOW2D_ENTER(ctx);
    if (!dev_ptr || width_bytes > pitch_bytes) return FLOW2D_ERR_INVALID_ARGUMENT;
    if (width_bytes == 0 || height == 0) return FLOW2D_OK;
    const size_t stride_bytes = ctx->batch_stride_floats * sizeof(float);
    if (ctx->batch_count > 1 && stride_bytes == pitch_bytes * height) {  // whole containers, one below the other: one call
        FLOW2D_HIP_TRY(hipMemset2DAsync(dev_ptr, pitch_bytes, byte_value, width_bytes, height * ctx->batch_count, ctx->stream));
        return FLOW2D_OK;
    }
    for (unsigned b = 0; b < ctx->batch_count; ++b)
        FLOW2D_HIP_TRY(hipMemset2DAsync(static_cast<char*>(dev_ptr) + b * stride_bytes, pitch_bytes, byte_value, width_bytes,
                                        height, ctx->stream));
    return FLOW2D_OK;
}

int flow2d_copy_h2d_2d(flow2d_context* ctx, void* dst_dev, size_t dst_pitch_bytes, const void* src_host,
                       size_t src_pitch_bytes, size_t width_bytes, size_t height)
{
    FLOW2D_ENTER(ctx);
    if (!dst_dev || !src_host || width_bytes > dst_pitch_bytes || width_bytes > src_pitch_bytes)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipMemcpy2DAsync(dst_dev, dst_pitch_bytes, src_host, src_pitch_bytes, width_bytes, height,
                                    hipMemcpyHostToDevice, ctx->stream));
    return FLOW2D_OK;
}

int flow2d_copy_d2h_2d(flow2d_context* ctx, void* dst_host, size_t dst_pitch_bytes, const void* src_dev,
                       size_t src_pitch_bytes, size_t width_bytes, size_t height)
{
    FLOW2D_ENTER(ctx);
    if (!dst_host || !src_dev || width_bytes > dst_pitch_bytes || width_bytes > src_pitch_bytes)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipMemcpy2DAsync(dst_host, dst_pitch_bytes, src_dev, src_pitch_bytes, width_bytes, height,
                                    hipMemcpyDeviceToHost, ctx->stream));
    return FLOW2D_OK;
}

int flow2d_copy_d2d(flow2d_context* ctx, void* dst_dev, const void* src_dev, size_t bytes)
{
    FLOW2D_ENTER(ctx);
    if (!dst_dev || !src_dev) return FLOW2D_ERR_INVALID_ARGUMENT;
    if (bytes == 0) return FLOW2D_OK;
    for (unsigned b = 0; b < ctx->batch_count; ++b) {
        const size_t off = b * ctx->batch_stride_floats * sizeof(float);
        FLOW2D_HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst_dev) + off, static_cast<const char*>(src_dev) + off, bytes,
                                      hipMemcpyDeviceToDevice, ctx->stream));
    }
    return FLOW2D_OK;
}

int flow2d_host_alloc(flow2d_context* ctx, size_t bytes, void** out_host_ptr)
{
    if (!out_host_ptr || bytes == 0) return FLOW2D_ERR_INVALID_ARGUMENT;
    *out_host_ptr = nullptr;
    if (ctx) {
        flow2d::DeviceGuard guard(ctx);
        if (!guard.ok()) return FLOW2D_ERR_DEVICE;
    }
    (void)hipGetLastError();
    void* p = nullptr;
    FLOW2D_HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    *out_host_ptr = p;
    return FLOW2D_OK;
}

int flow2d_host_free(flow2d_context* ctx, void* host_ptr)
{
    (void)ctx;
    if (!host_ptr) return FLOW2D_OK;
    (void)hipGetLastError();
    FLOW2D_HIP_TRY(hipHostFree(host_ptr));
    return FLOW2D_OK;
}

int flow2d_context_set_batch(flow2d_context* ctx, size_t count, size_t stride_bytes)
{
    // (two-plane launches carry planes x count in grid.z, which ends at 65535)
    if (!ctx || count == 0 || count > FLOW2D_BATCH_MAX || (count > 1 && (stride_bytes == 0 || stride_bytes % 16 != 0)))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    ctx->batch_count = static_cast<unsigned>(count);
    ctx->batch_stride_floats = count > 1 ? stride_bytes / sizeof(float) : 0;
    return FLOW2D_OK;
}

int flow2d_context_set_lone(flow2d_context* ctx, int lone)
{
    if (!ctx) return FLOW2D_ERR_INVALID_ARGUMENT;
    ctx->lone = lone != 0;
    return FLOW2D_OK;
}

int flow2d_capture_begin(flow2d_context* ctx)
{
    FLOW2D_ENTER(ctx);
    if (ctx->timing != 0) {
        flow2d::set_last_error_text("flow2d_capture_begin: switch launch timing off before capturing");
        return FLOW2D_ERR_INVALID_ARGUMENT;
    }
    FLOW2D_HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    return FLOW2D_OK;
}

int flow2d_capture_end(flow2d_context* ctx, void** out_graph_exec)
{
    FLOW2D_ENTER(ctx);
    if (!out_graph_exec) return FLOW2D_ERR_INVALID_ARGUMENT;
    *out_graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    FLOW2D_HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        flow2d::set_last_error("hipGraphInstantiate", e);
        return FLOW2D_ERR_DEVICE;
    }
    *out_graph_exec = exec;
    return FLOW2D_OK;
}

int flow2d_graph_launch(flow2d_context* ctx, void* graph_exec)
{
    FLOW2D_ENTER(ctx);
    if (!graph_exec) return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), ctx->stream));
    return FLOW2D_OK;
}

int flow2d_graph_destroy(flow2d_context* ctx, void* graph_exec)
{
    FLOW2D_ENTER(ctx);
    if (!graph_exec) return FLOW2D_OK;
    FLOW2D_HIP_TRY(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
    return FLOW2D_OK;
}

int flow2d_event_create(flow2d_context* ctx, void** out_event)
{
    FLOW2D_ENTER(ctx);
    if (!out_event) return FLOW2D_ERR_INVALID_ARGUMENT;
    hipEvent_t ev;
    FLOW2D_HIP_TRY(hipEventCreate(&ev));
    *out_event = ev;
    return FLOW2D_OK;
}

int flow2d_event_record(flow2d_context* ctx, void* event)
{
    FLOW2D_ENTER(ctx);
    if (!event) return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), ctx->stream));
    return FLOW2D_OK;
}

int flow2d_event_synchronize(flow2d_context* ctx, void* event)
{
    FLOW2D_ENTER(ctx);
    if (!event) return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return FLOW2D_OK;
}

int flow2d_event_elapsed_ms(flow2d_context* ctx, void* start_event, void* stop_event, float* out_ms)
{
    FLOW2D_ENTER(ctx);
    if (!start_event || !stop_event || !out_ms) return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipEventElapsedTime(out_ms, static_cast<hipEvent_t>(start_event),
                                       static_cast<hipEvent_t>(stop_event)));
    return FLOW2D_OK;
}

int flow2d_stream_wait_event(flow2d_context* ctx, void* event)
{
    FLOW2D_ENTER(ctx);
    if (!event) return FLOW2D_ERR_INVALID_ARGUMENT;
    FLOW2D_HIP_TRY(hipStreamWaitEvent(ctx->stream, static_cast<hipEvent_t>(event), 0));
    return FLOW2D_OK;
}

int flow2d_event_destroy(flow2d_context* ctx, void* event)
{
    FLOW2D_ENTER(ctx);
    if (!event) return FLOW2D_OK;
    FLOW2D_HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return FLOW2D_OK;
}

}  // extern "C"
