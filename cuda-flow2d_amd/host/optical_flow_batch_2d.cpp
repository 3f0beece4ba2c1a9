#include "optical_flow_batch_2d.h"

#include <cstdio>

#include "device_utils.h"

OpticalFlowBatch2D::OpticalFlowBatch2D() = default;

OpticalFlowBatch2D::~OpticalFlowBatch2D() { Destroy(); }

bool OpticalFlowBatch2D::Initialize(const DataSize3& data_size, DataConstancy data_constancy, size_t lanes, int device,
                                    size_t group_size)
{
    Destroy();
    group_size_ = group_size;
    if (lanes == 0 || lanes > 64) {
        std::printf("Error: OpticalFlowBatch2D: %zu lanes (1..64).\n", lanes);
        return false;
    }
    if (static_cast<size_t>(flow2d_hw_queues()) < lanes && !silent)
        std::printf("Warning: OpticalFlowBatch2D: %zu lanes on %d hardware queues (GPU_MAX_HW_QUEUES): lanes will share queues.\n",
                    lanes, flow2d_hw_queues());
    // OpticalFlow2D binds to the process's current context at Initialize (like the reference's operators bind to the
    // current CUDA context): each lane's own context is made current for its Initialize, the caller's is put back.
    bool ok = true;
    for (size_t i = 0; i < lanes && ok; ++i) {
        auto lane = std::make_unique<Lane>();
        if (CheckFlow2DError(flow2d_context_create(device, &lane->context), "flow2d_context_create")) {
            ok = false;
            break;
        }
        {
            ScopedDeviceContext current(lane->context);
            lane->flow.silent = silent;
            lane->flow.group_size = group_size;
            ok = lane->flow.Initialize(data_size, data_constancy);
            if (!ok) lane->flow.Destroy();
        }
        lanes_.push_back(std::move(lane));
    }
    if (!ok) Destroy();
    return ok;
}

bool OpticalFlowBatch2D::ComputeFlowBatchDevice(size_t count, const DevicePtr* dev_frames_0, const DevicePtr* dev_frames_1,
                                                const DevicePtr* dev_flows_u, const DevicePtr* dev_flows_v,
                                                OperationParameters& params, size_t first_lane)
{
    if (lanes_.empty() || (count != 0 && (!dev_frames_0 || !dev_frames_1 || !dev_flows_u || !dev_flows_v))) return false;
    bool ok = true;
    // pair k -> lane (first_lane + k) mod lanes; issuing in pair order IS round-robin over the lanes
    for (size_t k = 0; k < count; ++k) {
        Lane& lane = *lanes_[(first_lane + k) % lanes_.size()];
        lane.flow.use_graph = use_graph;
        lane.flow.timing_mode = 0;
        ok &= lane.flow.ComputeFlowDevice(dev_frames_0[k], dev_frames_1[k], dev_flows_u[k], dev_flows_v[k], params);
    }
    return ok;
}

bool OpticalFlowBatch2D::Synchronize()
{
    bool ok = true;
    for (auto& lane : lanes_)
        if (lane->context) ok &= !CheckFlow2DError(flow2d_synchronize(lane->context), "flow2d_synchronize");
    return ok;
}

void OpticalFlowBatch2D::Destroy()
{
    for (auto& lane : lanes_) {
        if (lane->context) (void)flow2d_synchronize(lane->context);
        lane->flow.Destroy();
        if (lane->context) flow2d_context_destroy(lane->context);
        lane->context = nullptr;
    }
    lanes_.clear();
}

DataSize3 OpticalFlowBatch2D::ContainerSize() const
{
    return lanes_.empty() ? DataSize3{0, 0, 0} : lanes_.front()->flow.ContainerSize();
}

size_t OpticalFlowBatch2D::GroupStrideBytes() const
{
    return lanes_.empty() ? 0 : lanes_.front()->flow.GroupStrideBytes();
}

flow2d_context* OpticalFlowBatch2D::LaneContext(size_t lane) const
{
    return lane < lanes_.size() ? lanes_[lane]->context : nullptr;
}
