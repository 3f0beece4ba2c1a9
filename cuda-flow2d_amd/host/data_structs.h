// Boundary PODs of the flow2d host layer.  Same names, members and meaning as the reference's
// src/data_types/data_structs.h:25-35 so caller code compiles unchanged.
#pragma once

#include <cstddef>

enum class Methods { OpticalFlow, Correlation };

// LogDerivatives is accepted by the type but not implemented by the MI355X path (out of scope).
enum class DataConstancy { Grey, Gradient, LogDerivatives };

struct DataSize3 {
    size_t width;
    size_t height;
    size_t pitch;  // bytes
};

// Raw device address as it travels through OperationParameters.  Same width and representation as
// the CUdeviceptr the reference stores in its bags (optical_flow_2d.cpp:194-211).
typedef unsigned long long DevicePtr;
