// Independent image pairs on one GPU, several in flight: the host side of the batched path (the
// reference has no counterpart -- it holds one context and blocks the host after every sweep,
// src/utils/cuda_utils.cpp:43, cuda_operation_solve_2d.cpp:291).
//
// A pair's pyramid is a chain of several hundred dependent launches whose coarse levels cannot fill the chip, so
// pairs are spread over `lanes` independent (HIP stream, OpticalFlow2D, plane pool) triples: pair k of a call goes to
// lane (first_lane + k) mod lanes, the calls of consecutive pairs are issued round-robin over the lanes so that one
// pair's launch-bound coarse levels overlap another pair's fine levels, and with use_graph every (lane, buffers,
// parameters) combination is recorded once as a HIP graph and replayed (one host call per pair).  Nothing is
// synchronised until Synchronize().  Every pair's flow is bit-identical to OpticalFlow2D::ComputeFlowDevice on that
// pair alone: the lanes share nothing but the device.
//
// Every lane needs a hardware queue of its own: the HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES
// (default 4) queues, and a lane that shares a queue waits behind its neighbour (measured at 4096^2, four lanes:
// 205 instead of 224 pairs/s when one more stream -- an idle one, or RCCL's -- exists in the process).  libflow2d_hip.so
// therefore sets GPU_MAX_HW_QUEUES=8 when it is loaded and the variable is unset (flow2d_hw_queues()); only a process
// that initialised HIP before loading the library has to export the variable itself.
#pragma once

#include <cstddef>
#include <memory>
#include <vector>

#include "optical_flow_2d.h"

class OpticalFlowBatch2D {
public:
    OpticalFlowBatch2D();
    ~OpticalFlowBatch2D();

    // `lanes` contexts (one stream each) on `device`, each with an OpticalFlow2D of `data_size` and its plane pool
    // (12 + 2 planes per lane).  false when the device, a stream or the memory is not to be had.
    // group_size > 1: every lane computes lock-step groups of that many pairs (OpticalFlow2D::group_size): an entry of
    // ComputeFlowBatchDevice is then a group, its four planes tall containers with pair g GroupStrideBytes() * g behind
    // the pointer.  Mid-size frames (1080p, 1024^2) gain a third in throughput: one launch per level holds the group.
    bool Initialize(const DataSize3& data_size, DataConstancy data_constancy, size_t lanes, int device = 0,
                    size_t group_size = 1);

    // `count` pairs already in pitched device containers of ContainerSize() (any allocation of this device):
    // dev_frames_*[k] are read, dev_flows_*[k] written.  Queued, not synchronised.  false on a bad argument or when
    // a pair's run was refused (the other pairs are still queued).
    bool ComputeFlowBatchDevice(size_t count, const DevicePtr* dev_frames_0, const DevicePtr* dev_frames_1,
                                const DevicePtr* dev_flows_u, const DevicePtr* dev_flows_v, OperationParameters& params,
                                size_t first_lane = 0);

    bool Synchronize();  // waits for every lane's stream
    void Destroy();

    bool use_graph = true;
    bool silent = true;

    size_t Lanes() const { return lanes_.size(); }
    DataSize3 ContainerSize() const;
    size_t GroupSize() const { return group_size_; }
    size_t GroupStrideBytes() const;
    flow2d_context* LaneContext(size_t lane) const;

private:
    struct Lane {
        flow2d_context* context = nullptr;
        OpticalFlow2D flow;
    };
    std::vector<std::unique_ptr<Lane>> lanes_;
    size_t group_size_ = 1;
};
