#include "cuda_operation_base.h"

#include <cstdio>

#include "device_utils.h"

bool CudaOperationBase::Initialize(const OperationParameters* params)
{
    initialized_ = false;
    if (!params) {
        std::printf("Operation: '%s'. Initialization parameters are missing.\n", GetName());
        return false;
    }
    DataSize3 container{0, 0, 0};
    if (!params->Read<DataSize3>("container_size", container)) {
        std::printf("Operation: '%s'. Missing parameter '%s'.\n", GetName(), "container_size");
        return false;
    }
    flow2d_context* ctx = nullptr;
    if (!params->Read<flow2d_context*>("flow2d_context", ctx)) ctx = CurrentDeviceContext();
    if (!ctx) {
        std::printf("Operation: '%s'. No device context: call InitDeviceContext() first.\n", GetName());
        return false;
    }
    if (container.pitch == 0 || container.pitch % 16 != 0 || container.pitch < container.width * sizeof(float)) {
        std::printf("Operation: '%s'. Bad container pitch %zu.\n", GetName(), container.pitch);
        return false;
    }
    context_ = ctx;
    dev_container_size_ = container;
    initialized_ = true;
    return true;
}

void CudaOperationBase::Execute(OperationParameters&)
{
    std::printf("Warning: '%s' Execute() was not defined.\n", name_);
}

void CudaOperationBase::Destroy()
{
    context_ = nullptr;
    initialized_ = false;
}

CudaOperationBase::~CudaOperationBase() = default;

bool CudaOperationBase::IsInitialized() const
{
    if (!initialized_) {
        std::printf("Error: Operation '%s' was not initialized.\n", name_);
        failed_ = true;
    }
    return initialized_;
}

bool CudaOperationBase::Failed(int status, const char* what) const
{
    if (status == FLOW2D_OK) return false;
    failed_ = true;
    std::printf("Operation '%s': %s failed: %s. %s\n", name_, what, flow2d_status_string(status), flow2d_last_error());
    return true;
}
