// Device glue of the host layer: the process-wide flow2d context and host<->device image copies.
// Role of the reference's src/utils/cuda_utils.{h,cpp}: InitCudaContextWithFirstAvailableDevice
// (cuda_utils.cpp:26-62), CopyData2DtoDevice / CopyData2DFromDevice (:66-105), CheckCudaError
// (cuda_utils.h:31-51).  Everything goes through the C-ABI of include/flow2d_c_abi.h.
#pragma once

#include <cstdint>

#include "data2d.h"
#include "data_structs.h"
#include "flow2d_c_abi.h"

// Creates the process-wide context on `device_ordinal` (the reference always takes device 0;
// a one-process-per-GPU launcher passes its LOCAL_RANK).  Returns false if no device is usable.
bool InitDeviceContext(int device_ordinal = 0);
// Uses an existing context (not owned) as the process-wide one.
void AdoptDeviceContext(flow2d_context* ctx);
flow2d_context* CurrentDeviceContext();
void DestroyDeviceContext();
// Makes `ctx` the process-wide context for the guard's lifetime and then puts the previous one back, ownership
// included (OpticalFlowBatch2D: each lane's objects bind to the lane's own context when they are initialised).
class ScopedDeviceContext {
public:
    explicit ScopedDeviceContext(flow2d_context* ctx);
    ~ScopedDeviceContext();
    ScopedDeviceContext(const ScopedDeviceContext&) = delete;
    ScopedDeviceContext& operator=(const ScopedDeviceContext&) = delete;

private:
    flow2d_context* previous_;
    bool previous_owned_;
};

// Prints "flow2d error = ..." and returns true when `status` is an error (same polarity as the
// reference's CheckCudaError).
bool CheckFlow2DError(int status, const char* where);

bool CopyData2DtoDevice(Data2D& data, DevicePtr device_ptr, size_t device_height, size_t device_pitch);
bool CopyData2DFromDevice(DevicePtr device_ptr, Data2D& data, size_t device_height, size_t device_pitch);

inline float* AsPlane(DevicePtr p) { return reinterpret_cast<float*>(static_cast<uintptr_t>(p)); }
