// Boundary PODs of the flow2d host layer.  Same names, members and meaning as the reference's
// src/data_types/data_structs.h:25-35 so caller code compiles unchanged.
#pragma once

#include <cstddef>

enum class Methods { OpticalFlow, Correlation };

// Grey, Gradient, LogDerivatives: the reference's values (src/data_types/data_structs.h:27).  GradientUntiled is an
// opt-in extra of this implementation (flow2d_c_abi.h, FLOW2D_CONSTANCY_GRADIENT_UNTILED): gradient constancy whose
// second derivatives use the true neighbours instead of the reference's 16x8 launch tiles.
enum class DataConstancy { Grey, Gradient, LogDerivatives, GradientUntiled };

struct DataSize3 {
    size_t width;
    size_t height;
    size_t pitch;  // bytes
};

// Raw device address as it travels through OperationParameters.  Same width and representation as
// the CUdeviceptr the reference stores in its bags (optical_flow_2d.cpp:194-211).
typedef unsigned long long DevicePtr;
