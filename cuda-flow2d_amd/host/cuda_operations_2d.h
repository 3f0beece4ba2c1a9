// The six operators of the 2D flow path.  Class names, bag keys and pointee types are those of the
// reference's src/cuda_operations/2d/cuda_operation_{add,convolution,median,registration,resample,
// solve}_2d.{h,cpp}; see each Execute for the keys (device pointers travel as DevicePtr, the same
// 64-bit representation as CUdeviceptr).
#pragma once

#include "cuda_operation_base.h"

// keys: operand_0, operand_1 (DevicePtr), data_size (DataSize3).  cuda_operation_add_2d.cpp:77-105
class CudaOperationAdd2D : public CudaOperationBase {
public:
    CudaOperationAdd2D() : CudaOperationBase("CUDA Add 2D") {}
    void Execute(OperationParameters& params) override;
};

// keys: dev_input, dev_output, dev_temp (DevicePtr), data_size (DataSize3), gaussian_sigma (float).
// cuda_operation_convolution_2d.cpp:133-176
class CudaOperationConvolution2D : public CudaOperationBase {
public:
    CudaOperationConvolution2D() : CudaOperationBase("CUDA Convolution 2D") {}
    void Execute(OperationParameters& params) override;
    // Host-side Gaussian taps (precision 3, pixel size 1), cuda_operation_convolution_2d.cpp:83-112
    void ComputeGaussianKernel(float sigma, size_t precision, float pixel_size);
    void PrintConvolutionKernel();

private:
    float kernel_[51] = {0};
    size_t kernel_radius_ = 0;
    size_t kernel_length_ = 0;
};

// keys: dev_input, dev_output (DevicePtr), data_size (DataSize3), radius (size_t; the window width).
// cuda_operation_median_2d.cpp:77-155
// Optional, not in the reference's bag: dev_input_b, dev_output_b (a second plane in the same launch); dev_addend,
// dev_addend_b (the filter runs over input + addend formed on the fly -- the pyramid's add_2d followed by median_2d in
// one launch; the input planes are not modified, except for width 1, where the operator adds in place and copies).
class CudaOperationMedian2D : public CudaOperationBase {
public:
    CudaOperationMedian2D() : CudaOperationBase("CUDA Median 2D") {}
    void Execute(OperationParameters& params) override;
};

// keys: dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_output (DevicePtr), data_size, hx, hy (float).
// cuda_operation_registration_2d.cpp:77-127
class CudaOperationRegistration2D : public CudaOperationBase {
public:
    CudaOperationRegistration2D() : CudaOperationBase("CUDA Registration 2D") {}
    void Execute(OperationParameters& params) override;
};

// keys: dev_input, dev_output, dev_temp (DevicePtr), data_size, resample_size (DataSize3).
// cuda_operation_resample_2d.cpp:76-107
// Optional second plane set: dev_input_b, dev_output_b, dev_temp_b.  When both directions up-sample, the two passes run
// in one launch and dev_temp is left untouched (same results).
class CudaOperationResample2D : public CudaOperationBase {
public:
    CudaOperationResample2D() : CudaOperationBase("CUDA Resample 2D") {}
    void Execute(OperationParameters& params) override;
};

// keys: dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_phi, dev_ksi (DevicePtr by value);
// dev_flow_du, dev_flow_dv, dev_temp_du, dev_temp_dv (DevicePtr, swapped IN PLACE through the bag so
// that after return the result is in the caller's dev_flow_du / dev_flow_dv variables);
// data_constancy (DataConstancy), outer_iterations_count, inner_iterations_count (size_t),
// equation_alpha, equation_smoothness, equation_data, hx, hy (float), data_size (DataSize3).
// Optional superset keys: solver_algorithm (int, flow2d_solver_algorithm), solver_sor_omega (float; opt-in
// red-black SOR with that factor instead of Jacobi sweeps -- no parity with the reference).
// cuda_operation_solve_2d.cpp:106-314
class CudaOperationSolve2D : public CudaOperationBase {
public:
    CudaOperationSolve2D() : CudaOperationBase("CUDA Solve 2D") {}
    bool Initialize(const OperationParameters* params = nullptr) override;
    void Execute(OperationParameters& params) override;

    bool silent = false;

private:
    DataConstancy init_constancy_ = DataConstancy::Grey;
};
