#include "optical_flow_batch_2d.h"

#include <algorithm>
#include <cstdio>

#include "device_utils.h"

OpticalFlowBatch2D::OpticalFlowBatch2D() = default;

OpticalFlowBatch2D::~OpticalFlowBatch2D() { Destroy(); }

bool OpticalFlowBatch2D::Initialize(const DataSize3& data_size, DataConstancy data_constancy, size_t lanes, int device,
                                    size_t group_size)
{
    Destroy();
    group_size_ = group_size;
    device_ = device;
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
            lane->flow.lone = lanes == 1;  // several lanes fill the device with each other's work: no second stream per lane
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

bool OpticalFlowBatch2D::ComputeFlowBatchDeviceGrouped(size_t count, const DevicePtr* dev_frames_0,
                                                       const DevicePtr* dev_frames_1, const DevicePtr* dev_flows_u,
                                                       const DevicePtr* dev_flows_v, OperationParameters& params,
                                                       size_t first_lane)
{
    if (lanes_.empty() || (count != 0 && (!dev_frames_0 || !dev_frames_1 || !dev_flows_u || !dev_flows_v))) return false;
    bool ok = true;
    size_t group = 0;
    for (size_t k = 0; k < count; k += group_size_, ++group) {
        const size_t n = std::min(group_size_, count - k);
        Lane& lane = *lanes_[(first_lane + group) % lanes_.size()];
        lane.flow.use_graph = use_graph;
        lane.flow.timing_mode = 0;
        ok &= lane.flow.ComputeFlowGroupDevice(n, dev_frames_0 + k, dev_frames_1 + k, dev_flows_u + k, dev_flows_v + k, params);
    }
    return ok;
}

// Per lane one set of staging planes: frame 0, frame 1, flow u, flow v, group-tall (first host-entry call).
bool OpticalFlowBatch2D::InitHostEntry()
{
    const DataSize3 size = ContainerSize();
    for (auto& lane : lanes_)
        for (DevicePtr& plane : lane->staging) {
            if (plane) continue;
            void* p = nullptr;
            size_t pitch = 0;
            if (CheckFlow2DError(flow2d_plane_alloc(lane->context, size.width, size.height * group_size_, &p, &pitch),
                                 "flow2d_plane_alloc") ||
                pitch != size.pitch)
                return false;
            plane = static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(p));
        }
    return true;
}

bool OpticalFlowBatch2D::ComputeFlowBatch(size_t count, Data2D* const* frames_0, Data2D* const* frames_1,
                                          Data2D* const* flows_u, Data2D* const* flows_v, OperationParameters& params,
                                          size_t first_lane)
{
    if (lanes_.empty() || (count != 0 && (!frames_0 || !frames_1 || !flows_u || !flows_v))) return false;
    if (count % group_size_ != 0) {
        std::printf("Error: OpticalFlowBatch2D: %zu pairs do not fill lock-step groups of %zu.\n", count, group_size_);
        return false;
    }
    const DataSize3 size = ContainerSize();
    for (size_t k = 0; k < count; ++k) {
        Data2D* images[4] = {frames_0[k], frames_1[k], flows_u[k], flows_v[k]};
        for (Data2D* image : images)
            if (!image || image->Width() != size.width || image->Height() != size.height || !image->DataPtr()) {
                std::printf("Error: OpticalFlowBatch2D: pair %zu: every image must be %zu x %zu.\n", k, size.width, size.height);
                return false;
            }
    }
    if (count == 0) return true;
    if (!InitHostEntry()) return false;
    const size_t row = size.width * sizeof(float);
    const size_t stride = GroupStrideBytes();
    bool ok = true;
    // Upload, pyramid and download of an entry are queued on ITS LANE's stream, in that order; what overlaps are the
    // lanes: while one lane's DMA engines move its frames or flows, the others compute.  (A first version with a
    // dedicated upload and a dedicated download stream chained to the lanes by events ran 9 % slower: 199 against 218
    // pairs/s at 4096^2, 228 with the copies taken out.)  The stream order also makes one staging set per lane enough.
    for (size_t entry = 0; entry * group_size_ < count; ++entry) {
        Lane& lane = *lanes_[(first_lane + entry) % lanes_.size()];
        char* planes[4];
        for (int i = 0; i < 4; ++i) planes[i] = reinterpret_cast<char*>(AsPlane(lane.staging[i]));
        for (size_t g = 0; g < group_size_; ++g) {
            const size_t k = entry * group_size_ + g;
            ok &= !CheckFlow2DError(flow2d_copy_h2d_2d(lane.context, planes[0] + g * stride, size.pitch,
                                                       frames_0[k]->DataPtr(), row, row, size.height), "flow2d_copy_h2d_2d");
            ok &= !CheckFlow2DError(flow2d_copy_h2d_2d(lane.context, planes[1] + g * stride, size.pitch,
                                                       frames_1[k]->DataPtr(), row, row, size.height), "flow2d_copy_h2d_2d");
        }
        lane.flow.use_graph = use_graph;
        lane.flow.timing_mode = 0;
        ok &= lane.flow.ComputeFlowDevice(lane.staging[0], lane.staging[1], lane.staging[2], lane.staging[3], params);
        for (size_t g = 0; g < group_size_; ++g) {
            const size_t k = entry * group_size_ + g;
            ok &= !CheckFlow2DError(flow2d_copy_d2h_2d(lane.context, flows_u[k]->DataPtr(), row, planes[2] + g * stride,
                                                       size.pitch, row, size.height), "flow2d_copy_d2h_2d");
            ok &= !CheckFlow2DError(flow2d_copy_d2h_2d(lane.context, flows_v[k]->DataPtr(), row, planes[3] + g * stride,
                                                       size.pitch, row, size.height), "flow2d_copy_d2h_2d");
        }
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
        for (DevicePtr& plane : lane->staging) {
            if (plane && lane->context) flow2d_plane_free(lane->context, AsPlane(plane));
            plane = 0;
        }
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
