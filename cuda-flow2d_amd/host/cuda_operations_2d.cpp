#include "cuda_operations_2d.h"

#include <cstdio>
#include <utility>

#include "device_utils.h"

// ---- Add -----------------------------------------------------------------------------------------
void CudaOperationAdd2D::Execute(OperationParameters& params)
{
    if (!IsInitialized()) return;
    DevicePtr operand_0 = 0, operand_1 = 0;
    DataSize3 data_size{};
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, operand_0, "operand_0");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, operand_1, "operand_1");
    FLOW2D_PARAM_OR_RETURN(params, DataSize3, data_size, "data_size");
    // optional second plane set (not in the reference's bag): both additions in one launch
    DevicePtr operand_0_b = 0, operand_1_b = 0;
    if (params.Read<DevicePtr>("operand_0_b", operand_0_b) && params.Read<DevicePtr>("operand_1_b", operand_1_b)) {
        Failed(flow2d_add_2d_pair(context_, AsPlane(operand_0), AsPlane(operand_1), AsPlane(operand_0_b),
                                  AsPlane(operand_1_b), data_size.width, data_size.height, dev_container_size_.pitch),
               "flow2d_add_2d_pair");
        return;
    }
    Failed(flow2d_add_2d(context_, AsPlane(operand_0), AsPlane(operand_1), data_size.width, data_size.height,
                         dev_container_size_.pitch),
           "flow2d_add_2d");
}

// ---- Convolution -----------------------------------------------------------------------------------
void CudaOperationConvolution2D::ComputeGaussianKernel(float sigma, size_t precision, float pixel_size)
{
    // The C-ABI computes the taps with the reference's arithmetic for (precision 3, pixel size 1),
    // the only combination the reference uses (cuda_operation_convolution_2d.cpp:160).
    int radius = 0;
    kernel_length_ = 0;
    kernel_radius_ = 0;
    if (precision != 3 || pixel_size != 1.0f) {
        std::printf("<%s>: only precision 3 / pixel size 1 Gaussian kernels are supported.\n", GetName());
        failed_ = true;
        return;
    }
    if (Failed(flow2d_gaussian_kernel(sigma, kernel_, &radius), "flow2d_gaussian_kernel")) return;
    kernel_radius_ = static_cast<size_t>(radius);
    kernel_length_ = 2 * kernel_radius_ + 1;
}

void CudaOperationConvolution2D::PrintConvolutionKernel()
{
    if (kernel_length_ == 0) {
        std::printf("Error: Convolution kernel is not initialized.\n");
        return;
    }
    std::printf("Convolution kernel (radius = %zu)\n", kernel_radius_);
    for (size_t i = 0; i < kernel_length_; ++i) std::printf("%.4f ", kernel_[i]);
    std::printf("\n\n");
}

void CudaOperationConvolution2D::Execute(OperationParameters& params)
{
    if (!IsInitialized()) return;
    DevicePtr dev_input = 0, dev_output = 0, dev_temp = 0;
    DataSize3 data_size{};
    float gaussian_sigma = 0.f;
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_input, "dev_input");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_temp, "dev_temp");
    FLOW2D_PARAM_OR_RETURN(params, DataSize3, data_size, "data_size");
    FLOW2D_PARAM_OR_RETURN(params, float, gaussian_sigma, "gaussian_sigma");
    if (dev_input == dev_output) {
        std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
        failed_ = true;
        return;
    }
    ComputeGaussianKernel(gaussian_sigma, 3, 1.0f);
    if (kernel_length_ == 0) return;
    // rows then columns (cuda_operation_convolution_2d.cpp:169-175), fused into one launch: the temp plane
    // stays untouched, the output is bit-identical to the two-pass form
    (void)dev_temp;
    Failed(flow2d_gaussian_blur(context_, AsPlane(dev_output), AsPlane(dev_input), data_size.width, data_size.height,
                                dev_container_size_.pitch, kernel_, static_cast<int>(kernel_radius_)),
           "flow2d_gaussian_blur");
}

// ---- Median ----------------------------------------------------------------------------------------
void CudaOperationMedian2D::Execute(OperationParameters& params)
{
    if (!IsInitialized()) return;
    DevicePtr dev_input = 0, dev_output = 0;
    DataSize3 data_size{};
    size_t radius = 0;
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_input, "dev_input");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
    FLOW2D_PARAM_OR_RETURN(params, DataSize3, data_size, "data_size");
    FLOW2D_PARAM_OR_RETURN(params, size_t, radius, "radius");
    if (dev_input == dev_output) {
        std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
        failed_ = true;
        return;
    }
    DevicePtr dev_input_b = 0, dev_output_b = 0;  // optional second plane (not in the reference's bag)
    const bool pair = params.Read<DevicePtr>("dev_input_b", dev_input_b) &&
                      params.Read<DevicePtr>("dev_output_b", dev_output_b);
    if (pair && dev_input_b == dev_output_b) {
        std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
        failed_ = true;
        return;
    }
    // optional addends (not in the reference's bag): the filter runs over dev_input + dev_addend, formed on the fly -- the
    // pyramid's `u += du` (a single rounded addition per pixel) and the median of u in one pass; dev_input is not modified
    DevicePtr dev_addend = 0, dev_addend_b = 0;
    const bool sum = params.Read<DevicePtr>("dev_addend", dev_addend) && (!pair || params.Read<DevicePtr>("dev_addend_b", dev_addend_b));
    const size_t window = (radius != 1 && radius % 2 == 0) ? radius - 1 : radius;
    if (sum && !(window >= 3 && window <= 7)) {  // nothing to fuse into: add in place, then the plain path
        if (pair)
            Failed(flow2d_add_2d_pair(context_, AsPlane(dev_input), AsPlane(dev_addend), AsPlane(dev_input_b), AsPlane(dev_addend_b),
                                      data_size.width, data_size.height, dev_container_size_.pitch), "flow2d_add_2d_pair");
        else
            Failed(flow2d_add_2d(context_, AsPlane(dev_input), AsPlane(dev_addend), data_size.width, data_size.height,
                                 dev_container_size_.pitch), "flow2d_add_2d");
    }
    if (radius == 1) {  // no filtering: copy the whole container (cuda_operation_median_2d.cpp:100-104)
        Failed(flow2d_copy_d2d(context_, AsPlane(dev_output), AsPlane(dev_input),
                               dev_container_size_.pitch * dev_container_size_.height),
               "flow2d_copy_d2d");
        if (pair)
            Failed(flow2d_copy_d2d(context_, AsPlane(dev_output_b), AsPlane(dev_input_b),
                                   dev_container_size_.pitch * dev_container_size_.height),
                   "flow2d_copy_d2d");
        return;
    }
    if (radius % 2 == 0) {
        std::printf("Warning. Median raduis is even (%zu), decresaing by 1...\n", radius);
        radius -= 1;
    }
    if (radius >= 3 && radius <= 7 && sum) {
        Failed(flow2d_add_median_2d_pair(context_, AsPlane(dev_input), AsPlane(dev_addend), pair ? AsPlane(dev_input_b) : nullptr,
                                         pair ? AsPlane(dev_addend_b) : nullptr, data_size.width, data_size.height,
                                         dev_container_size_.pitch, radius, AsPlane(dev_output),
                                         pair ? AsPlane(dev_output_b) : nullptr),
               "flow2d_add_median_2d_pair");
    } else if (radius >= 3 && radius <= 7 && pair) {
        Failed(flow2d_median_2d_pair(context_, AsPlane(dev_input), AsPlane(dev_input_b), data_size.width,
                                     data_size.height, dev_container_size_.pitch, radius, AsPlane(dev_output),
                                     AsPlane(dev_output_b)),
               "flow2d_median_2d_pair");
    } else if (radius >= 3 && radius <= 7) {
        Failed(flow2d_median_2d(context_, AsPlane(dev_input), data_size.width, data_size.height,
                                dev_container_size_.pitch, radius, AsPlane(dev_output)),
               "flow2d_median_2d");
    } else {
        std::printf("Error. Wrong median raduis (%zu). Supported values: 3, 5, 7\n", radius);
        failed_ = true;
    }
}

// ---- Registration ----------------------------------------------------------------------------------
void CudaOperationRegistration2D::Execute(OperationParameters& params)
{
    if (!IsInitialized()) return;
    DevicePtr dev_frame_0 = 0, dev_frame_1 = 0, dev_flow_u = 0, dev_flow_v = 0, dev_output = 0;
    DataSize3 data_size{};
    float hx = 0.f, hy = 0.f;
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_frame_0, "dev_frame_0");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_frame_1, "dev_frame_1");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_flow_u, "dev_flow_u");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_flow_v, "dev_flow_v");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
    FLOW2D_PARAM_OR_RETURN(params, DataSize3, data_size, "data_size");
    FLOW2D_PARAM_OR_RETURN(params, float, hx, "hx");
    FLOW2D_PARAM_OR_RETURN(params, float, hy, "hy");
    if (dev_frame_1 == dev_output) {
        std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
        failed_ = true;
        return;
    }
    Failed(flow2d_registration_2d(context_, AsPlane(dev_frame_0), AsPlane(dev_frame_1), AsPlane(dev_flow_u),
                                  AsPlane(dev_flow_v), data_size.width, data_size.height, dev_container_size_.pitch,
                                  hx, hy, AsPlane(dev_output)),
           "flow2d_registration_2d");
}

// ---- Resample --------------------------------------------------------------------------------------
void CudaOperationResample2D::Execute(OperationParameters& params)
{
    if (!IsInitialized()) return;
    DevicePtr dev_input = 0, dev_output = 0, dev_temp = 0;
    DataSize3 data_size{}, resample_size{};
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_input, "dev_input");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_temp, "dev_temp");
    FLOW2D_PARAM_OR_RETURN(params, DataSize3, data_size, "data_size");
    FLOW2D_PARAM_OR_RETURN(params, DataSize3, resample_size, "resample_size");
    if (dev_input == dev_output) {
        std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
        failed_ = true;
        return;
    }
    // optional second plane set (not in the reference's bag): two planes of the same geometry per launch
    DevicePtr dev_input_b = 0, dev_output_b = 0, dev_temp_b = 0;
    const bool pair = params.Read<DevicePtr>("dev_input_b", dev_input_b) && params.Read<DevicePtr>("dev_output_b", dev_output_b) &&
                      params.Read<DevicePtr>("dev_temp_b", dev_temp_b);
    // up-sampling in both directions (the flow of the previous level): both passes in one launch, the temp plane unused
    if (resample_size.width >= data_size.width && resample_size.height >= data_size.height) {
        Failed(flow2d_resample_xy_pair(context_, AsPlane(dev_input), AsPlane(dev_output), pair ? AsPlane(dev_input_b) : nullptr,
                                       pair ? AsPlane(dev_output_b) : nullptr, data_size.width, data_size.height,
                                       resample_size.width, resample_size.height, dev_container_size_.pitch),
               "flow2d_resample_xy_pair");
        return;
    }
    if (pair) {
        if (Failed(flow2d_resample_x_pair(context_, AsPlane(dev_input), AsPlane(dev_temp), AsPlane(dev_input_b),
                                          AsPlane(dev_temp_b), resample_size.width, data_size.height, data_size.width,
                                          dev_container_size_.pitch),
                   "flow2d_resample_x_pair"))
            return;
        Failed(flow2d_resample_y_pair(context_, AsPlane(dev_temp), AsPlane(dev_output), AsPlane(dev_temp_b),
                                      AsPlane(dev_output_b), resample_size.width, resample_size.height,
                                      data_size.height, dev_container_size_.pitch),
               "flow2d_resample_y_pair");
        return;
    }
    // x pass into temp at (new width x old height), then y pass (cuda_operation_resample_2d.cpp:99-105)
    if (Failed(flow2d_resample_x(context_, AsPlane(dev_input), AsPlane(dev_temp), resample_size.width,
                                 data_size.height, data_size.width, dev_container_size_.pitch),
               "flow2d_resample_x"))
        return;
    Failed(flow2d_resample_y(context_, AsPlane(dev_temp), AsPlane(dev_output), resample_size.width,
                             resample_size.height, data_size.height, dev_container_size_.pitch),
           "flow2d_resample_y");
}

// ---- Solve -----------------------------------------------------------------------------------------
bool CudaOperationSolve2D::Initialize(const OperationParameters* params)
{
    if (!CudaOperationBase::Initialize(params)) return false;
    DataConstancy constancy = DataConstancy::Grey;
    if (!params->Read<DataConstancy>("data_constancy", constancy)) {
        std::printf("Operation: '%s'. Missing parameter '%s'.\n", GetName(), "data_constancy");
        initialized_ = false;
        return false;
    }
    init_constancy_ = constancy;
    return true;
}

void CudaOperationSolve2D::Execute(OperationParameters& params)
{
    if (!IsInitialized()) return;
    DevicePtr dev_frame_0 = 0, dev_frame_1 = 0, dev_flow_u = 0, dev_flow_v = 0, dev_phi = 0, dev_ksi = 0;
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_frame_0, "dev_frame_0");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_frame_1, "dev_frame_1");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_flow_u, "dev_flow_u");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_flow_v, "dev_flow_v");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_phi, "dev_phi");
    FLOW2D_PARAM_OR_RETURN(params, DevicePtr, dev_ksi, "dev_ksi");

    // taken by pointer: the caller's variables are swapped so the result ends in dev_flow_du / dv
    DevicePtr *du_ptr = nullptr, *dv_ptr = nullptr, *tdu_ptr = nullptr, *tdv_ptr = nullptr;
    FLOW2D_PARAM_PTR_OR_RETURN(params, DevicePtr, du_ptr, "dev_flow_du");
    FLOW2D_PARAM_PTR_OR_RETURN(params, DevicePtr, dv_ptr, "dev_flow_dv");
    FLOW2D_PARAM_PTR_OR_RETURN(params, DevicePtr, tdu_ptr, "dev_temp_du");
    FLOW2D_PARAM_PTR_OR_RETURN(params, DevicePtr, tdv_ptr, "dev_temp_dv");

    flow2d_solve_params p{};
    DataSize3 data_size{};
    DataConstancy data_constancy = DataConstancy::Grey;
    FLOW2D_PARAM_OR_RETURN(params, size_t, p.outer_iterations_count, "outer_iterations_count");
    FLOW2D_PARAM_OR_RETURN(params, size_t, p.inner_iterations_count, "inner_iterations_count");
    FLOW2D_PARAM_OR_RETURN(params, float, p.equation_alpha, "equation_alpha");
    FLOW2D_PARAM_OR_RETURN(params, float, p.equation_smoothness, "equation_smoothness");
    FLOW2D_PARAM_OR_RETURN(params, float, p.equation_data, "equation_data");
    FLOW2D_PARAM_OR_RETURN(params, float, p.hx, "hx");
    FLOW2D_PARAM_OR_RETURN(params, float, p.hy, "hy");
    FLOW2D_PARAM_OR_RETURN(params, DataSize3, data_size, "data_size");
    FLOW2D_PARAM_OR_RETURN(params, DataConstancy, data_constancy, "data_constancy");
    int algorithm = FLOW2D_SOLVER_AUTO;
    params.Read<int>("solver_algorithm", algorithm);
    float sor_omega = 0.f;  // superset key: opt-in red-black SOR instead of the reference's Jacobi sweeps
    params.Read<float>("solver_sor_omega", sor_omega);

    // The reference picks the sweep kernel at Initialize and only the LDS size at Execute
    // (cuda_operation_solve_2d.cpp:65-82,181-198); the kernel chosen at Initialize wins.
    (void)data_constancy;
    p.width = data_size.width;
    p.height = data_size.height;
    p.pitch_bytes = dev_container_size_.pitch;
    p.container_height = dev_container_size_.height;
    // the kernel is chosen by the constancy given to Initialize (cuda_operation_solve_2d.cpp:65-82)
    p.data_constancy = init_constancy_ == DataConstancy::Gradient          ? FLOW2D_CONSTANCY_GRADIENT
                       : init_constancy_ == DataConstancy::LogDerivatives  ? FLOW2D_CONSTANCY_LOG_DERIVATIVES
                       : init_constancy_ == DataConstancy::GradientUntiled ? FLOW2D_CONSTANCY_GRADIENT_UNTILED
                                                                           : FLOW2D_CONSTANCY_GREY;
    p.algorithm = algorithm;
    p.sor_omega = sor_omega;

    void *ev_start = nullptr, *ev_stop = nullptr;
    if (!silent) {
        flow2d_event_create(context_, &ev_start);
        flow2d_event_create(context_, &ev_stop);
        flow2d_event_record(context_, ev_start);
    }
    int result_in_temp = 0;
    const bool failed = Failed(
        flow2d_solve_level(context_, AsPlane(dev_frame_0), AsPlane(dev_frame_1), AsPlane(dev_flow_u),
                           AsPlane(dev_flow_v), AsPlane(*du_ptr), AsPlane(*dv_ptr), AsPlane(dev_phi), AsPlane(dev_ksi),
                           AsPlane(*tdu_ptr), AsPlane(*tdv_ptr), &p, &result_in_temp),
        "flow2d_solve_level");
    if (!failed && result_in_temp) {
        std::swap(*du_ptr, *tdu_ptr);
        std::swap(*dv_ptr, *tdv_ptr);
    }
    if (!silent) {
        // per-level solve time, the reference's second timer (cuda_operation_solve_2d.cpp:302-311)
        float ms = 0.f;
        flow2d_event_record(context_, ev_stop);
        flow2d_event_synchronize(context_, ev_stop);
        flow2d_event_elapsed_ms(context_, ev_start, ev_stop, &ms);
        std::printf(" solve %4zu x%4zu: %8.4fs\n", p.width, p.height, ms / 1000.);
        flow2d_event_destroy(context_, ev_start);
        flow2d_event_destroy(context_, ev_stop);
    }
}
