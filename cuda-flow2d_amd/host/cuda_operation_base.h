// Operator ("plugin") base class of the host layer.  Same public interface as the reference's
// src/cuda_operations/cuda_operation_base.h:29-53 -- Initialize(const OperationParameters*),
// Execute(OperationParameters&), Destroy(), GetName() -- so operator-level callers are drop-in.
// Where the reference loads `<exe>/kernels/<op>.ptx` and resolves kernels by symbol name, an
// operator here binds to the C-ABI launchers of include/flow2d_c_abi.h on a flow2d_context.
#pragma once

#include "data_structs.h"
#include "flow2d_c_abi.h"
#include "operation_parameters.h"

class CudaOperationBase {
public:
    const char* GetName() const { return name_; }

    // Init keys: "container_size" (DataSize3, pitch in bytes) is mandatory, as in the reference;
    // "flow2d_context" (flow2d_context*) is optional -- default is the process-wide context of
    // device_utils.h (the reference relies on the implicit current CUcontext).
    virtual bool Initialize(const OperationParameters* params = nullptr);
    virtual void Execute(OperationParameters& params);
    virtual void Destroy();

    virtual ~CudaOperationBase();

    // Execute() is void like the reference's (which prints and goes on after a failed driver call, leaving stale
    // buffers to be swapped in: SURVEY section 5).  A superset for callers that want to know: true when an Execute()
    // since the last call refused its arguments, missed a key or had a C-ABI call fail; reading clears it.
    bool TakeFailure()
    {
        const bool f = failed_;
        failed_ = false;
        return f;
    }

protected:
    explicit CudaOperationBase(const char* name) : name_(name) {}

    bool IsInitialized() const;
    // Reports a failed C-ABI call the way the reference reports a failed driver call: print, go on.
    bool Failed(int status, const char* what) const;

    flow2d_context* context_ = nullptr;
    DataSize3 dev_container_size_{0, 0, 0};
    bool initialized_ = false;
    mutable bool failed_ = false;  // sticky until TakeFailure()

private:
    const char* name_ = nullptr;
};

// Fetches bag value KEY of type TYPE into VAR; on a missing key prints the reference's message
// (common_utils.h:34-43) and returns from the calling operator.
#define FLOW2D_PARAM_OR_RETURN(PARAMS, TYPE, VAR, KEY)                                         \
    do {                                                                                       \
        if (!(PARAMS).Read<TYPE>((KEY), (VAR))) {                                              \
            std::printf("Operation: '%s'. Missing parameter '%s'.\n", GetName(), (KEY));       \
            failed_ = true;                                                                    \
            return;                                                                            \
        }                                                                                      \
    } while (0)

#define FLOW2D_PARAM_PTR_OR_RETURN(PARAMS, TYPE, PTR, KEY)                                     \
    do {                                                                                       \
        (PTR) = static_cast<TYPE*>((PARAMS).GetValuePtr((KEY)));                               \
        if (!(PTR)) {                                                                          \
            std::printf("Operation: '%s'. Missing parameter '%s'.\n", GetName(), (KEY));       \
            failed_ = true;                                                                    \
            return;                                                                            \
        }                                                                                      \
    } while (0)
