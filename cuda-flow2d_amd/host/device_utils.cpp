#include "device_utils.h"

#include <cstdint>
#include <cstdio>

namespace {
// per host thread: a process that drives several GPUs from one thread each (flow2d_batch --gpus N) gives every thread
// its own current context, like the CUDA driver's per-thread current context the reference relies on (cuCtxCreate,
// src/utils/cuda_utils.cpp:43)
thread_local flow2d_context* g_context = nullptr;
thread_local bool g_owned = false;
}  // namespace

bool CheckFlow2DError(int status, const char* where)
{
    if (status == FLOW2D_OK) return false;
    std::printf("flow2d error = %04d \"%s\" in %s. %s\n", status, flow2d_status_string(status), where,
                flow2d_last_error());
    return true;
}

bool InitDeviceContext(int device_ordinal)
{
    if (g_context) return true;
    // eight hardware queues for the lanes of the batched path: must precede the process's first HIP call (a refusal -- the runtime
    // already runs, or the caller exported a smaller value -- is not an error: OpticalFlowBatch2D::Initialize warns when it matters)
    (void)flow2d_request_hw_queues(8);
    int count = 0;
    if (flow2d_device_count(&count) != FLOW2D_OK || count == 0) {
        std::printf("Error: no HIP devices supporting flow2d (gfx950) were found.\n");
        return false;
    }
    if (CheckFlow2DError(flow2d_context_create(device_ordinal, &g_context), "flow2d_context_create")) {
        g_context = nullptr;
        return false;
    }
    g_owned = true;
    char name[256] = {0};
    flow2d_device_name(g_context, name, sizeof(name));
    std::printf("Using HIP device [%d]: %s\n", device_ordinal, name);
    return true;
}

void AdoptDeviceContext(flow2d_context* ctx)
{
    DestroyDeviceContext();
    g_context = ctx;
    g_owned = false;
}

flow2d_context* CurrentDeviceContext() { return g_context; }

void DestroyDeviceContext()
{
    if (g_context && g_owned) flow2d_context_destroy(g_context);
    g_context = nullptr;
    g_owned = false;
}

ScopedDeviceContext::ScopedDeviceContext(flow2d_context* ctx) : previous_(g_context), previous_owned_(g_owned)
{
    g_context = ctx;
    g_owned = false;
}

ScopedDeviceContext::~ScopedDeviceContext()
{
    g_context = previous_;
    g_owned = previous_owned_;
}

bool CopyData2DtoDevice(Data2D& data, DevicePtr device_ptr, size_t device_height, size_t device_pitch)
{
    if (!g_context || data.Height() > device_height || data.Width() * sizeof(float) > device_pitch) return false;
    const size_t row = data.Width() * sizeof(float);
    return !CheckFlow2DError(
        flow2d_copy_h2d_2d(g_context, AsPlane(device_ptr), device_pitch, data.DataPtr(), row, row, data.Height()),
        "CopyData2DtoDevice");
}

bool CopyData2DFromDevice(DevicePtr device_ptr, Data2D& data, size_t device_height, size_t device_pitch)
{
    if (!g_context || data.Height() > device_height || data.Width() * sizeof(float) > device_pitch) return false;
    const size_t row = data.Width() * sizeof(float);
    return !CheckFlow2DError(
        flow2d_copy_d2h_2d(g_context, data.DataPtr(), row, AsPlane(device_ptr), device_pitch, row, data.Height()),
        "CopyData2DFromDevice");
}
