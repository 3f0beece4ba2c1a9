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
// 205 instead of 224 pairs/s when one more stream -- an idle one, or RCCL's -- exists in the process).  The host
// layer therefore asks for eight queues before its first HIP call (InitDeviceContext -> flow2d_request_hw_queues(8), which sets
// GPU_MAX_HW_QUEUES when the caller has not); only a process that started HIP before that has to export the variable itself.
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

    // Lock-step groups formed by the object itself (group_size > 1): `count` INDEPENDENT pairs, every plane a container
    // of its own (any allocation of this device, ContainerSize()); consecutive runs of up to GroupSize() pairs become
    // one group each (OpticalFlow2D::ComputeFlowGroupDevice: gather, one launch per kernel for the group, hand back -- or, when
    // the run's planes already sit GroupStrideBytes() apart in all four roles, the pyramid on them in place, no copies),
    // group k on lane (first_lane + k) mod lanes.  The last group may be smaller.  Mid-size frames (1024^2, 1080p) gain
    // a third in throughput over one pair per lane; each pair's flow is bit-identical to its own ComputeFlowDevice.
    bool ComputeFlowBatchDeviceGrouped(size_t count, const DevicePtr* dev_frames_0, const DevicePtr* dev_frames_1,
                                       const DevicePtr* dev_flows_u, const DevicePtr* dev_flows_v,
                                       OperationParameters& params, size_t first_lane = 0);

    // The same for HOST images -- the bracket of the reference's own timer, uploads and downloads included
    // (optical_flow_2d.cpp:173-179,214-215,544-554): entry k's frames are copied into its lane's staging planes, its
    // pyramid runs, and its flows are copied out, all queued on the lane's stream; the lanes overlap each other, so one
    // lane's DMA transfers run beside the others' kernels and the host waits for nothing.  frames_*[k] are read,
    // flows_*[k] (pre-allocated, ContainerSize() wide and high) written; all four must stay alive and untouched until
    // Synchronize().  Images in page-locked memory (Data2D(w, h, HostMemory::Pinned)) move by DMA at the PCIe rate
    // without blocking the host; pageable images are staged by the runtime and stall it (same results).  With
    // group_size > 1 consecutive pairs form the lock-step groups, so count must be a multiple of it.  Results are those
    // of OpticalFlow2D::ComputeFlow per pair, bit for bit.
    bool ComputeFlowBatch(size_t count, Data2D* const* frames_0, Data2D* const* frames_1, Data2D* const* flows_u,
                          Data2D* const* flows_v, OperationParameters& params, size_t first_lane = 0);

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
        DevicePtr staging[4] = {0, 0, 0, 0};  // host entry: frame 0, frame 1, flow u, flow v (group_size containers tall)
    };
    bool InitHostEntry();
    std::vector<std::unique_ptr<Lane>> lanes_;
    size_t group_size_ = 1;
    int device_ = 0;
};
