#include "optical_flow_2d.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "device_utils.h"

// ---- base ------------------------------------------------------------------------------------------
size_t OpticalFlowBase2D::GetMaxWarpLevel(size_t width, size_t height, float scale_factor) const
{
    // Level n is usable while ceil(W * s^n) >= 4 and ceil(H * s^n) >= 4, all in float
    // (optical_flow_base_2d.cpp:36-59).
    size_t level_w = 1, level_h = 1, levels = 1;
    for (; scale_factor < 1.f; ++levels) {
        const float s = std::pow(scale_factor, static_cast<float>(levels));
        level_w = static_cast<size_t>(std::ceil(width * s));
        level_h = static_cast<size_t>(std::ceil(height * s));
        if (level_w < 4 || level_h < 4) break;
    }
    if (level_w == 1 || level_h == 1) --levels;
    return levels;
}

bool OpticalFlowBase2D::IsInitialized() const
{
    if (!initialized_) std::printf("Error: '%s' was not initialized.\n", name_);
    return initialized_;
}

void OpticalFlowBase2D::ComputeFlow(Data2D&, Data2D&, Data2D&, Data2D&, OperationParameters&)
{
    std::printf("Warning: '%s' ComputeFlow() was not defined.\n", name_);
}

void OpticalFlowBase2D::Destroy() { initialized_ = false; }

OpticalFlowBase2D::~OpticalFlowBase2D() = default;

// ---- OpticalFlow2D ---------------------------------------------------------------------------------
OpticalFlow2D::OpticalFlow2D() : OpticalFlowBase2D("Optical Flow 2D MI355X") {}

OpticalFlow2D::~OpticalFlow2D() { Destroy(); }

bool OpticalFlow2D::Initialize(const DataSize3& data_size, DataConstancy data_constancy)
{
    if (initialized_) Destroy();
    context_ = CurrentDeviceContext();
    if (!context_) {
        std::printf("Error: '%s': no device context, call InitDeviceContext() first.\n", GetName());
        return false;
    }
    if (data_size.width < 4 || data_size.height < 4) {
        std::printf("Error: '%s': frames must be at least 4 x 4.\n", GetName());
        return false;
    }
    if (group_size == 0 || group_size > 64) {
        std::printf("Error: '%s': group size %zu (1..64).\n", GetName(), group_size);
        return false;
    }
    group_ = group_size;
    (void)flow2d_context_set_lone(context_, lone ? 1 : 0);
    dev_container_size_ = data_size;
    dev_container_size_.pitch = 0;
    data_constancy_ = data_constancy;
    initialized_ = InitMemory() && InitOperations();
    if (!initialized_) Destroy();
    return initialized_;
}

bool OpticalFlow2D::InitMemory()
{
    if (!silent) std::printf("Allocating memory on the device...\n");
    size_t free_bytes = 0, total_bytes = 0;
    if (CheckFlow2DError(flow2d_mem_info(context_, &free_bytes, &total_bytes), "flow2d_mem_info")) return false;
    const size_t pitch = flow2d_plane_pitch_bytes(dev_container_size_.width);
    // (+ the two packed x-pass planes; a lock-step group holds every plane group_ containers tall)
    const size_t needed = pitch * dev_container_size_.height * group_ * (kContainersCount + 3);
    if (!silent)
        std::printf("Available\t:\t%.0fMB / %.0fMB\nNeeded\t\t:\t%.0fMB\n", free_bytes / 1048576.f,
                    total_bytes / 1048576.f, needed / 1048576.f);
    if (needed >= free_bytes) return false;  // same silent refusal as optical_flow_2d.cpp:109-113
    for (size_t i = 0; i < kContainersCount; ++i) {
        void* plane = nullptr;
        size_t got_pitch = 0;
        if (CheckFlow2DError(flow2d_plane_alloc(context_, dev_container_size_.width, dev_container_size_.height * group_,
                                                &plane, &got_pitch),
                             "flow2d_plane_alloc") ||
            got_pitch != pitch) {
            std::printf("Error during device memory allocation.");
            return false;
        }
        all_planes_.push_back(static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(plane)));
    }
    free_planes_ = all_planes_;
    for (DevicePtr& packed : packed_frames_) {  // outside the pool: they hold one pair's x-resampled rows of all levels
        void* plane = nullptr;
        size_t got_pitch = 0;
        if (CheckFlow2DError(flow2d_plane_alloc(context_, dev_container_size_.width, dev_container_size_.height * group_,
                                                &plane, &got_pitch),
                             "flow2d_plane_alloc") ||
            got_pitch != pitch) {
            std::printf("Error during device memory allocation.");
            return false;
        }
        packed = static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(plane));
    }
    {  // ... and the plane the warp writes when the levels are stacked (RunPyramid); without it the levels are resampled one by one
        void* plane = nullptr;
        size_t got_pitch = 0;
        if (flow2d_plane_alloc(context_, dev_container_size_.width, dev_container_size_.height * group_, &plane, &got_pitch) == FLOW2D_OK) {
            if (got_pitch == pitch) level_warp_plane_ = static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(plane));
            else flow2d_plane_free(context_, plane);
        }
    }
    dev_container_size_.pitch = pitch;
    return true;
}

bool OpticalFlow2D::InitOperations()
{
    if (dev_container_size_.pitch == 0) {
        std::printf("Initialization failed. Device pitch is 0.\n");
        return false;
    }
    OperationParameters op;
    op.PushValuePtr("container_size", &dev_container_size_);
    op.PushValuePtr("data_constancy", &data_constancy_);
    op.PushValuePtr("flow2d_context", &context_);
    CudaOperationBase* ops[] = {&cuop_add_, &cuop_convolution_, &cuop_median_,
                                &cuop_register_, &cuop_resample_, &cuop_solve_};
    for (CudaOperationBase* cuop : ops) {
        const bool ok = cuop->Initialize(&op);
        if (!silent) std::printf("%-18s: %s\n", cuop->GetName(), ok ? "OK" : "FAILED");
        if (!ok) return false;
    }
    return true;
}

void OpticalFlow2D::Destroy()
{
    CudaOperationBase* ops[] = {&cuop_add_, &cuop_convolution_, &cuop_median_,
                                &cuop_register_, &cuop_resample_, &cuop_solve_};
    for (CudaOperationBase* cuop : ops) cuop->Destroy();
    if (context_) {
        if (!all_planes_.empty()) flow2d_synchronize(context_);
        DropGraphs();
        FreeSequenceCache();
        if (free_planes_.size() != all_planes_.size())
            std::printf("Warning. Not all device memory allocations were freed.\n");
        for (DevicePtr p : all_planes_) flow2d_plane_free(context_, AsPlane(p));
        for (DevicePtr& p : packed_frames_) {
            if (p) flow2d_plane_free(context_, AsPlane(p));
            p = 0;
        }
        for (DevicePtr& p : group_staging_) {
            if (p) flow2d_plane_free(context_, AsPlane(p));
            p = 0;
        }
        if (level_warp_plane_) flow2d_plane_free(context_, AsPlane(level_warp_plane_));
        level_warp_plane_ = 0;
    }
    all_planes_.clear();
    free_planes_.clear();
    initialized_ = false;
}

DevicePtr OpticalFlow2D::Acquire()
{
    DevicePtr p = free_planes_.back();
    free_planes_.pop_back();
    return p;
}

void OpticalFlow2D::Release(DevicePtr p) { free_planes_.push_back(p); }

void OpticalFlow2D::FreeSequenceCache()
{
    for (FramePyramid& pyramid : sequence_cache_) {
        if (context_) {
            if (pyramid.blurred) flow2d_plane_free(context_, AsPlane(pyramid.blurred));
            for (DevicePtr p : pyramid.levels)
                if (p) flow2d_plane_free(context_, AsPlane(p));
        }
        pyramid = FramePyramid();
    }
}

// The plane of `pyramid` that holds pyramid level `level` (>= 1): container width, `rows` rows, same pitch as the
// 12 containers.  Allocated on first use and whenever a later call needs more rows.  0 on failure.
DevicePtr OpticalFlow2D::SequenceLevelPlane(FramePyramid& pyramid, size_t level, size_t rows)
{
    if (pyramid.levels.size() <= level) {
        pyramid.levels.resize(level + 1, 0);
        pyramid.level_rows.resize(level + 1, 0);
    }
    if (pyramid.levels[level] && pyramid.level_rows[level] >= rows) return pyramid.levels[level];
    if (pyramid.levels[level]) {
        flow2d_synchronize(context_);  // the old plane may still be read by queued work
        flow2d_plane_free(context_, AsPlane(pyramid.levels[level]));
        pyramid.levels[level] = 0;
    }
    void* plane = nullptr;
    size_t pitch = 0;
    if (CheckFlow2DError(flow2d_plane_alloc(context_, dev_container_size_.width, rows, &plane, &pitch), "flow2d_plane_alloc"))
        return 0;
    if (pitch != dev_container_size_.pitch) {
        flow2d_plane_free(context_, static_cast<float*>(plane));
        std::printf("Error: '%s': sequence cache pitch %zu differs from the container pitch %zu.\n", GetName(), pitch,
                    dev_container_size_.pitch);
        return 0;
    }
    pyramid.levels[level] = static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(plane));
    pyramid.level_rows[level] = rows;
    return pyramid.levels[level];
}

bool OpticalFlow2D::ComputeFlowSequenceDevice(const DevicePtr* dev_frames, size_t frame_count, const DevicePtr* dev_flows_u,
                                              const DevicePtr* dev_flows_v, OperationParameters& params)
{
    if (!IsInitialized() || !dev_frames || !dev_flows_u || !dev_flows_v || frame_count < 2) return false;
    if (group_ > 1) {
        std::printf("Error: '%s': sequences and lock-step groups do not combine.\n", GetName());
        return false;
    }
    for (size_t k = 0; k < frame_count; ++k)
        if (!dev_frames[k] || (k + 1 < frame_count && (!dev_flows_u[k] || !dev_flows_v[k]))) return false;
    float gaussian_sigma = 0.f;
    params.Read<float>("gaussian_sigma", gaussian_sigma);
    for (FramePyramid& pyramid : sequence_cache_) {
        pyramid.valid = false;
        if (gaussian_sigma > 0.f && !pyramid.blurred) {  // a plane for the blurred frame
            void* plane = nullptr;
            size_t pitch = 0;
            if (CheckFlow2DError(flow2d_plane_alloc(context_, dev_container_size_.width, dev_container_size_.height,
                                                    &plane, &pitch),
                                 "flow2d_plane_alloc") ||
                pitch != dev_container_size_.pitch)
                return false;
            pyramid.blurred = static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(plane));
        }
    }
    bool ok = true;
    for (size_t k = 0; ok && k + 1 < frame_count; ++k) {
        FramePyramid& first = sequence_cache_[k % 2];          // frame k: built as the second frame of pair k-1
        FramePyramid& second = sequence_cache_[(k + 1) % 2];   // frame k+1: built by this pair
        second.valid = false;
        sequence_frames_[0] = &first;
        sequence_frames_[1] = &second;
        dev_frame_0_ = Acquire();  // unused by a sequence pair, kept for the pool's bookkeeping
        dev_frame_1_ = Acquire();
        dev_flow_u_ = Acquire();
        dev_flow_v_ = Acquire();
        caller_frame_0_ = dev_frames[k];
        caller_frame_1_ = dev_frames[k + 1];
        caller_flow_u_ = dev_flows_u[k];
        caller_flow_v_ = dev_flows_v[k];
        ok = RunPyramid(params);
        caller_frame_0_ = caller_frame_1_ = caller_flow_u_ = caller_flow_v_ = 0;
        sequence_frames_[0] = sequence_frames_[1] = nullptr;
        Release(dev_frame_0_);
        Release(dev_frame_1_);
        Release(dev_flow_u_);
        Release(dev_flow_v_);
        if (ok) first.valid = second.valid = true;
    }
    return ok;
}

void OpticalFlow2D::ResetLevelTimings()
{
    if (context_) flow2d_timing_reset(context_);
}

std::vector<FlowLevelTiming> OpticalFlow2D::LastLevelTimings()
{
    std::vector<FlowLevelTiming> out;
    size_t n = 0;
    if (!context_ || flow2d_timing_count(context_, &n) != FLOW2D_OK) return out;
    for (size_t i = 0; i < n; ++i) {
        flow2d_timing_record r;
        if (flow2d_timing_get(context_, i, &r) != FLOW2D_OK) break;
        out.push_back({r.width, r.height, r.elapsed_ms, r.kernel_ms, r.kernel_launches, r.algorithm,
                       r.algorithmic_bytes_per_launch});
    }
    return out;
}

void OpticalFlow2D::ComputeFlow(Data2D& frame_0, Data2D& frame_1, Data2D& flow_u, Data2D& flow_v,
                                OperationParameters& params)
{
    last_run_ok_ = false;
    if (!IsInitialized()) return;
    if (group_ > 1) {
        std::printf("Error: '%s': ComputeFlow takes one pair; lock-step groups go through ComputeFlowDevice.\n", GetName());
        return;
    }
    const size_t W = dev_container_size_.width, H = dev_container_size_.height;
    if (frame_0.Width() != W || frame_0.Height() != H || frame_1.Width() != W || frame_1.Height() != H ||
        flow_u.Width() != W || flow_u.Height() != H || flow_v.Width() != W || flow_v.Height() != H) {
        std::printf("Error: '%s': frame / flow sizes do not match the initialised size %zu x %zu.\n", GetName(), W, H);
        return;
    }
    std::printf("\nStarting optical flow computation...\n");
    void *ev_start = nullptr, *ev_stop = nullptr;
    flow2d_event_create(context_, &ev_start);
    flow2d_event_create(context_, &ev_stop);
    flow2d_event_record(context_, ev_start);

    dev_frame_0_ = Acquire();
    dev_frame_1_ = Acquire();
    dev_flow_u_ = Acquire();
    dev_flow_v_ = Acquire();
    bool ok = CopyData2DtoDevice(frame_0, dev_frame_0_, H, dev_container_size_.pitch) &&
              CopyData2DtoDevice(frame_1, dev_frame_1_, H, dev_container_size_.pitch);
    ok = ok && RunPyramid(params);
    if (ok) {
        ok = CopyData2DFromDevice(dev_flow_u_, flow_u, H, dev_container_size_.pitch) &&
             CopyData2DFromDevice(dev_flow_v_, flow_v, H, dev_container_size_.pitch);
    }
    last_run_ok_ = ok;
    flow2d_event_record(context_, ev_stop);
    flow2d_event_synchronize(context_, ev_stop);  // the only host wait of a pair
    flow2d_event_elapsed_ms(context_, ev_start, ev_stop, &last_total_ms_);
    std::printf("Total GPU computation time: % 4.4fs\n", last_total_ms_ / 1000.);
    flow2d_event_destroy(context_, ev_start);
    flow2d_event_destroy(context_, ev_stop);

    Release(dev_frame_0_);
    Release(dev_frame_1_);
    Release(dev_flow_u_);
    Release(dev_flow_v_);
}

namespace {
// Everything a recorded pyramid depends on: which entry recorded it ('P': a pair or a tall group through
// ComputeFlowDevice, 'G': a group gathered from scattered planes -- a group of ONE scattered pair names the same four
// planes as the pair entry but records gather -> pyramid of one instance -> hand back), the number of instances, the
// caller buffers and the nine (+1) parameters.
std::vector<unsigned char> GraphKey(char entry, size_t instances, const std::vector<DevicePtr>& planes, OperationParameters& params)
{
    std::vector<unsigned char> key;
    auto put = [&key](const void* p, size_t n) {
        const unsigned char* q = static_cast<const unsigned char*>(p);
        key.insert(key.end(), q, q + n);
    };
    put(&entry, sizeof(entry));
    put(&instances, sizeof(instances));
    put(planes.data(), planes.size() * sizeof(DevicePtr));
    const char* size_keys[] = {"warp_levels_count", "outer_iterations_count", "inner_iterations_count", "median_radius"};
    const char* float_keys[] = {"warp_scale_factor", "equation_alpha", "equation_smoothness", "equation_data",
                                "gaussian_sigma"};
    for (const char* k : size_keys) {
        size_t v = ~size_t(0);
        params.Read<size_t>(k, v);
        put(&v, sizeof(v));
    }
    for (const char* k : float_keys) {
        float v = -1.f;
        params.Read<float>(k, v);
        put(&v, sizeof(v));
    }
    int algorithm = FLOW2D_SOLVER_AUTO;
    params.Read<int>("solver_algorithm", algorithm);
    put(&algorithm, sizeof(algorithm));
    float sor_omega = 0.f;
    params.Read<float>("solver_sor_omega", sor_omega);
    put(&sor_omega, sizeof(sor_omega));
    return key;
}
}  // namespace

void OpticalFlow2D::DropGraphs()
{
    for (auto& kv : graphs_)
        if (kv.second.exec && context_) flow2d_graph_destroy(context_, kv.second.exec);
    graphs_.clear();
}

// Replays the graph recorded under `key`, recording it first (by running `queue` under stream capture) when the
// combination is new.
bool OpticalFlow2D::ReplayOrRecord(std::vector<unsigned char> key, const std::function<bool()>& queue)
{
    auto it = graphs_.find(key);
    if (it == graphs_.end()) {
        if (graphs_.size() >= kMaxGraphs) {
            // full: the least recently replayed graph goes.  It may still be running on the stream (replays are not
            // synchronised by contract), so the stream is drained first.
            flow2d_synchronize(context_);
            auto oldest = graphs_.begin();
            for (auto g = graphs_.begin(); g != graphs_.end(); ++g)
                if (g->second.last_use < oldest->second.last_use) oldest = g;
            if (oldest->second.exec) flow2d_graph_destroy(context_, oldest->second.exec);
            graphs_.erase(oldest);
        }
        if (CheckFlow2DError(flow2d_capture_begin(context_), "flow2d_capture_begin")) return false;
        const bool queued = queue();
        void* exec = nullptr;
        const bool ended = !CheckFlow2DError(flow2d_capture_end(context_, &exec), "flow2d_capture_end");
        if (!queued || !ended) {
            if (exec) flow2d_graph_destroy(context_, exec);
            return false;
        }
        it = graphs_.emplace(std::move(key), RecordedGraph{exec, 0}).first;
    }
    it->second.last_use = ++graph_clock_;
    return !CheckFlow2DError(flow2d_graph_launch(context_, it->second.exec), "flow2d_graph_launch");
}

bool OpticalFlow2D::ComputeFlowDevice(DevicePtr dev_frame_0, DevicePtr dev_frame_1, DevicePtr dev_flow_u,
                                      DevicePtr dev_flow_v, OperationParameters& params)
{
    if (!IsInitialized() || !dev_frame_0 || !dev_frame_1 || !dev_flow_u || !dev_flow_v) return false;
    active_group_ = group_;
    if (!use_graph || timing_mode != 0) return QueuePair(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, params);
    return ReplayOrRecord(GraphKey('P', group_, {dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v}, params), [&] {
        return QueuePair(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, params);
    });
}

bool OpticalFlow2D::ComputeFlowGroupDevice(size_t count, const DevicePtr* dev_frames_0, const DevicePtr* dev_frames_1,
                                           const DevicePtr* dev_flows_u, const DevicePtr* dev_flows_v,
                                           OperationParameters& params)
{
    if (!IsInitialized() || !dev_frames_0 || !dev_frames_1 || !dev_flows_u || !dev_flows_v) return false;
    if (count == 0 || count > group_) {
        std::printf("Error: '%s': a group of %zu pairs (1..%zu).\n", GetName(), count, group_);
        return false;
    }
    std::vector<DevicePtr> planes;
    for (size_t g = 0; g < count; ++g) {
        if (!dev_frames_0[g] || !dev_frames_1[g] || !dev_flows_u[g] || !dev_flows_v[g]) return false;
        planes.insert(planes.end(), {dev_frames_0[g], dev_frames_1[g], dev_flows_u[g], dev_flows_v[g]});
    }
    // Pairs whose planes already sit one container apart -- frame 0 of pair g exactly GroupStrideBytes() * g behind frame 0 of pair 0,
    // and so for frame 1, u and v: a caller that keeps a group's pairs in four tall allocations -- ARE a group as laid out: the
    // pyramid runs on the caller's planes, nothing is gathered or handed back (round 6: the two copies were 4.4 % of a 4096^2 group
    // of eight, 5.8 % of a 1024^2 group of 32).
    const size_t stride = GroupStrideBytes();
    bool in_place = true;
    for (size_t g = 1; g < count && in_place; ++g)
        in_place = dev_frames_0[g] == dev_frames_0[0] + g * stride && dev_frames_1[g] == dev_frames_1[0] + g * stride &&
                   dev_flows_u[g] == dev_flows_u[0] + g * stride && dev_flows_v[g] == dev_flows_v[0] + g * stride;
    if (in_place) {
        auto queue = [&] {
            active_group_ = count;
            const bool ok = QueuePair(dev_frames_0[0], dev_frames_1[0], dev_flows_u[0], dev_flows_v[0], params);
            active_group_ = group_;
            return ok;
        };
        if (!use_graph || timing_mode != 0) return queue();
        return ReplayOrRecord(GraphKey('I', count, {dev_frames_0[0], dev_frames_1[0], dev_flows_u[0], dev_flows_v[0]}, params), queue);
    }
    for (DevicePtr& p : group_staging_) {  // the tall staging containers, at the first scattered group
        if (p) continue;
        void* plane = nullptr;
        size_t pitch = 0;
        if (CheckFlow2DError(flow2d_plane_alloc(context_, dev_container_size_.width, dev_container_size_.height * group_,
                                                &plane, &pitch),
                             "flow2d_plane_alloc") ||
            pitch != dev_container_size_.pitch)
            return false;
        p = static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(plane));
    }
    if (!use_graph || timing_mode != 0)
        return QueueScatteredGroup(count, dev_frames_0, dev_frames_1, dev_flows_u, dev_flows_v, params);
    return ReplayOrRecord(GraphKey('G', count, planes, params), [&] {
        return QueueScatteredGroup(count, dev_frames_0, dev_frames_1, dev_flows_u, dev_flows_v, params);
    });
}

// gather the frames -> the group's pyramid on the staging containers -> hand the flows back
bool OpticalFlow2D::QueueScatteredGroup(size_t count, const DevicePtr* dev_frames_0, const DevicePtr* dev_frames_1,
                                        const DevicePtr* dev_flows_u, const DevicePtr* dev_flows_v,
                                        OperationParameters& params)
{
    const size_t stride = GroupStrideBytes();
    auto slot = [&](int which, size_t g) -> void* { return reinterpret_cast<char*>(AsPlane(group_staging_[which])) + g * stride; };
    std::vector<const void*> src;
    std::vector<void*> dst;
    for (size_t g = 0; g < count; ++g) {
        src.push_back(AsPlane(dev_frames_0[g])), dst.push_back(slot(0, g));
        src.push_back(AsPlane(dev_frames_1[g])), dst.push_back(slot(1, g));
    }
    // one launch per FLOW2D_COPY_PLANES_MAX planes (the pointer tables travel in the kernel arguments): groups of up to 32
    // pairs gather with one launch, the largest (64) with two
    auto copy_planes = [&]() {
        for (size_t first = 0; first < src.size(); first += FLOW2D_COPY_PLANES_MAX) {
            const size_t n = std::min<size_t>(FLOW2D_COPY_PLANES_MAX, src.size() - first);
            if (CheckFlow2DError(flow2d_copy_planes(context_, n, src.data() + first, dst.data() + first, dev_container_size_.pitch,
                                                    dev_container_size_.width, dev_container_size_.height),
                                 "flow2d_copy_planes"))
                return false;
        }
        return true;
    };
    if (!copy_planes()) return false;
    active_group_ = count;
    const bool ok = QueuePair(group_staging_[0], group_staging_[1], group_staging_[2], group_staging_[3], params);
    active_group_ = group_;
    if (!ok) return false;
    src.clear(), dst.clear();
    for (size_t g = 0; g < count; ++g) {
        src.push_back(slot(2, g)), dst.push_back(AsPlane(dev_flows_u[g]));
        src.push_back(slot(3, g)), dst.push_back(AsPlane(dev_flows_v[g]));
    }
    return copy_planes();
}

bool OpticalFlow2D::QueuePair(DevicePtr dev_frame_0, DevicePtr dev_frame_1, DevicePtr dev_flow_u,
                              DevicePtr dev_flow_v, OperationParameters& params)
{
    const size_t bytes = dev_container_size_.pitch * dev_container_size_.height;
    // a lock-step group: from here to the end of the run every launch, memset and device copy of this context acts on
    // all group_ pairs (instance g of every plane, pool and caller alike, GroupStrideBytes() * g behind its pointer)
    if (group_ > 1 && CheckFlow2DError(flow2d_context_set_batch(context_, active_group_, GroupStrideBytes()), "flow2d_context_set_batch"))
        return false;
    dev_frame_0_ = Acquire();
    dev_frame_1_ = Acquire();
    dev_flow_u_ = Acquire();
    dev_flow_v_ = Acquire();
    // The pyramid consumes its frame planes (level-0 swaps), so without a pre-blur it works on copies; with one
    // the blur is the only reader of the frames and reads the caller's planes in place.  The flow of the last
    // level leaves through its median, which writes the caller's planes directly.
    float gaussian_sigma = 0.f;
    params.Read<float>("gaussian_sigma", gaussian_sigma);
    const bool frames_in_place = gaussian_sigma > 0.f;
    bool ok = true;
    if (frames_in_place) {
        caller_frame_0_ = dev_frame_0;
        caller_frame_1_ = dev_frame_1;
    } else {
        ok = !CheckFlow2DError(flow2d_copy_d2d(context_, AsPlane(dev_frame_0_), AsPlane(dev_frame_0), bytes), "copy") &&
             !CheckFlow2DError(flow2d_copy_d2d(context_, AsPlane(dev_frame_1_), AsPlane(dev_frame_1), bytes), "copy");
    }
    caller_flow_u_ = dev_flow_u;
    caller_flow_v_ = dev_flow_v;
    ok = ok && RunPyramid(params);
    caller_frame_0_ = caller_frame_1_ = caller_flow_u_ = caller_flow_v_ = 0;
    Release(dev_frame_0_);
    Release(dev_frame_1_);
    Release(dev_flow_u_);
    Release(dev_flow_v_);
    if (group_ > 1) flow2d_context_set_batch(context_, 1, 0);
    return ok;
}

// The coarse-to-fine loop on device planes: pre-blur, then per level { resample frames from full
// resolution, resample the flow from the previous level, warp frame 1, solve, u += du, median }.
// Order, buffer roles and float level-size arithmetic follow optical_flow_2d.cpp:160-449.
// On entry dev_frame_0_/1_ hold the frames; on success dev_flow_u_/v_ hold the flow.
bool OpticalFlow2D::RunPyramid(OperationParameters& params)
{
    size_t warp_levels_count = 0, outer_iterations_count = 0, inner_iterations_count = 0, median_radius = 0;
    float warp_scale_factor = 0.f, equation_alpha = 0.f, equation_smoothness = 0.f, equation_data = 0.f;
    float gaussian_sigma = 0.f;
    struct {
        const char* key;
        bool ok;
    } reads[] = {
        {"warp_levels_count", params.Read<size_t>("warp_levels_count", warp_levels_count)},
        {"warp_scale_factor", params.Read<float>("warp_scale_factor", warp_scale_factor)},
        {"outer_iterations_count", params.Read<size_t>("outer_iterations_count", outer_iterations_count)},
        {"inner_iterations_count", params.Read<size_t>("inner_iterations_count", inner_iterations_count)},
        {"equation_alpha", params.Read<float>("equation_alpha", equation_alpha)},
        {"equation_smoothness", params.Read<float>("equation_smoothness", equation_smoothness)},
        {"equation_data", params.Read<float>("equation_data", equation_data)},
        {"median_radius", params.Read<size_t>("median_radius", median_radius)},
        {"gaussian_sigma", params.Read<float>("gaussian_sigma", gaussian_sigma)},
    };
    for (const auto& r : reads)
        if (!r.ok) {
            std::printf("Operation: '%s'. Missing parameter '%s'.\n", GetName(), r.key);
            return false;
        }
    int solver_algorithm = FLOW2D_SOLVER_AUTO;
    params.Read<int>("solver_algorithm", solver_algorithm);
    float solver_sor_omega = 0.f;
    params.Read<float>("solver_sor_omega", solver_sor_omega);

    DataSize3 original_size = {dev_container_size_.width, dev_container_size_.height, 0};
    const size_t max_level = GetMaxWarpLevel(original_size.width, original_size.height, warp_scale_factor);
    int level = static_cast<int>(std::min(warp_levels_count, max_level)) - 1;
    if (level < 0 || !(warp_scale_factor < 1.f)) {
        // the reference would skip the loop and hand back stale buffers: refuse instead
        std::printf("Error: '%s': no pyramid level to run (levels %zu, scale %g).\n", GetName(), warp_levels_count,
                    warp_scale_factor);
        return false;
    }

    {  // widths the median operator accepts: 1 (copy), 3..8 (even widths use width - 1); anything else would
       // make the reference swap in a stale buffer (cuda_operation_median_2d.cpp:150-152)
        const size_t eff = (median_radius != 1 && median_radius % 2 == 0) ? median_radius - 1 : median_radius;
        if (!(median_radius == 1 || (median_radius != 2 && eff >= 3 && eff <= 7))) {
            std::printf("Error. Wrong median raduis (%zu). Supported values: 3, 5, 7\n", median_radius);
            return false;
        }
    }

    flow2d_timing_enable(context_, timing_mode);
    // per-launch brackets (mode 2) only on the finest level: that is the kernel the roofline is quoted on
    flow2d_timing_launch_filter(context_, dev_container_size_.width, dev_container_size_.height);

    DevicePtr frame_0 = dev_frame_0_, frame_1 = dev_frame_1_, flow_u = dev_flow_u_, flow_v = dev_flow_v_;
    DevicePtr frame_0_res = Acquire(), frame_1_res = Acquire(), flow_du = Acquire(), flow_dv = Acquire();
    OperationParameters op;
    // The operators' Execute() is void, like the reference's, whose ComputeFlow never learns of a failed launch and
    // swaps the stale output plane in.  Here every operator's failure flag is collected and the
    // run is abandoned at the end of the level it happened in: ComputeFlow then leaves the caller's flow untouched,
    // ComputeFlowDevice returns false and no graph of the broken run is kept.
    bool failed = false;

    const bool sequence = sequence_frames_[0] != nullptr;

    // The reference resamples both frames from FULL resolution at every level (optical_flow_2d.cpp:284-303): one read
    // of each frame per level.  Here the x passes of all levels > 0 are one trip over the frames (every row read once,
    // the x-resampled rows of all levels written side by side into a packed plane per frame; same cell sums, same
    // bits); a level's y pass then reads its segment.  Used when the segments fit one row (scale factors up to ~0.5).
    std::vector<size_t> packed_width, packed_column;
    bool packed = false;
    if (!sequence && level >= 2 && original_size.width <= 15360 && level <= FLOW2D_RESAMPLE_MAX_LEVELS) {
        size_t column = 0;
        for (int l = level; l >= 1; --l) {
            const float s = std::pow(warp_scale_factor, static_cast<float>(l));
            const size_t lw = static_cast<size_t>(std::ceil(original_size.width * s));
            packed_width.push_back(lw);
            packed_column.push_back(column);
            column += (lw + 3) / 4 * 4;  // 16-byte aligned segments (what a plane pointer must be)
        }
        packed = column <= dev_container_size_.pitch / sizeof(float);
    }
    const int first_level = level;
    // Round 6: with the x passes done in one trip, the y passes of ALL levels are one launch too (flow2d_resample_y_levels; they were
    // seven launches of 7-39 us for a config-3 pair, most of them far too small to fill the device).  Every level then needs a plane
    // region of its own: the levels sit one below the other in the two planes that otherwise hold "the current level's frames"
    // (their heights sum to less than the container's for scale factors up to 0.5), and the warp of a level writes the plane kept
    // for it instead of replacing the level's frame 1.  Same kernels' arithmetic on the same values: same bits.
    std::vector<size_t> level_row(static_cast<size_t>(first_level) + 1, 0), level_width(level_row.size(), 0), level_height(level_row.size(), 0);
    bool stacked = false;
    if (packed && level_warp_plane_) {
        size_t rows = 0;
        for (int l = first_level; l >= 1; --l) {
            const float s = std::pow(warp_scale_factor, static_cast<float>(l));
            level_row[static_cast<size_t>(l)] = rows;
            level_width[static_cast<size_t>(l)] = static_cast<size_t>(std::ceil(original_size.width * s));
            level_height[static_cast<size_t>(l)] = static_cast<size_t>(std::ceil(original_size.height * s));
            rows += level_height[static_cast<size_t>(l)];
        }
        stacked = rows <= dev_container_size_.height;
    }

    if (sequence) {  // a frame's level 0 is blurred once (or is the caller's own plane) and then only read
        DevicePtr callers[2] = {caller_frame_0_, caller_frame_1_};
        DevicePtr* level0[2] = {&frame_0, &frame_1};
        DevicePtr temp = Acquire();
        for (int i = 0; i < 2; ++i) {
            FramePyramid& pyramid = *sequence_frames_[i];
            if (gaussian_sigma > 0.0) {
                pyramid.level0 = pyramid.blurred;
                if (!pyramid.valid) {
                    op.Clear();
                    op.PushValuePtr("dev_input", &callers[i]);
                    op.PushValuePtr("dev_output", &pyramid.level0);
                    op.PushValuePtr("dev_temp", &temp);
                    op.PushValuePtr("data_size", &original_size);
                    op.PushValuePtr("gaussian_sigma", &gaussian_sigma);
                    cuop_convolution_.Execute(op);
                failed |= cuop_convolution_.TakeFailure();
                }
            } else {
                pyramid.level0 = callers[i];
            }
            *level0[i] = pyramid.level0;
        }
        Release(temp);
    } else if (gaussian_sigma > 0.0 && caller_frame_0_) {  // frames still in the caller's planes: blur them into ours
        DevicePtr temp = Acquire();
        DevicePtr sources[2] = {caller_frame_0_, caller_frame_1_};
        DevicePtr* targets[2] = {&frame_0, &frame_1};
        for (int i = 0; i < 2; ++i) {
            op.Clear();
            op.PushValuePtr("dev_input", &sources[i]);
            op.PushValuePtr("dev_output", targets[i]);
            op.PushValuePtr("dev_temp", &temp);  // (not touched: the blur is one launch)
            op.PushValuePtr("data_size", &original_size);
            op.PushValuePtr("gaussian_sigma", &gaussian_sigma);
            cuop_convolution_.Execute(op);
            failed |= cuop_convolution_.TakeFailure();
        }
        Release(temp);
    } else if (gaussian_sigma > 0.0) {  // optical_flow_2d.cpp:218-246: blur into the flow planes, then swap roles
        DevicePtr temp = Acquire();
        DevicePtr* io[2][2] = {{&frame_0, &flow_u}, {&frame_1, &flow_v}};
        for (auto& pair : io) {
            op.Clear();
            op.PushValuePtr("dev_input", pair[0]);
            op.PushValuePtr("dev_output", pair[1]);
            op.PushValuePtr("dev_temp", &temp);
            op.PushValuePtr("data_size", &original_size);
            op.PushValuePtr("gaussian_sigma", &gaussian_sigma);
            cuop_convolution_.Execute(op);
            failed |= cuop_convolution_.TakeFailure();
            std::swap(*pair[0], *pair[1]);
        }
        Release(temp);
    }

    if (packed) {
        if (CheckFlow2DError(flow2d_resample_x_levels(context_, AsPlane(frame_0), AsPlane(packed_frames_[0]),
                                                      AsPlane(frame_1), AsPlane(packed_frames_[1]),
                                                      original_size.width, original_size.height,
                                                      dev_container_size_.pitch, packed_width.size(),
                                                      packed_width.data(), packed_column.data()),
                             "flow2d_resample_x_levels"))
            failed = true;
        if (stacked) {  // every level's y pass, both frames, one launch
            std::vector<size_t> widths, heights, rows;
            for (int l = first_level; l >= 1; --l) {
                widths.push_back(level_width[static_cast<size_t>(l)]);
                heights.push_back(level_height[static_cast<size_t>(l)]);
                rows.push_back(level_row[static_cast<size_t>(l)]);
            }
            if (CheckFlow2DError(flow2d_resample_y_levels(context_, AsPlane(packed_frames_[0]), AsPlane(frame_0_res), AsPlane(packed_frames_[1]),
                                                          AsPlane(frame_1_res), original_size.height, dev_container_size_.pitch, widths.size(),
                                                          widths.data(), heights.data(), packed_column.data(), rows.data()),
                                 "flow2d_resample_y_levels"))
                failed = true;
        }
    }

    DataSize3 current_size = {0, 0, 0}, prev_size = {0, 0, 0};
    for (; level >= 0; --level) {
        const float scale = std::pow(warp_scale_factor, static_cast<float>(level));
        current_size.width = static_cast<size_t>(std::ceil(original_size.width * scale));
        current_size.height = static_cast<size_t>(std::ceil(original_size.height * scale));
        float hx = original_size.width / static_cast<float>(current_size.width);
        float hy = original_size.height / static_cast<float>(current_size.height);
        if (!silent) std::printf("Solve level %2d (%4zu x%4zu) \n", level, current_size.width, current_size.height);

        // frames: level 0 uses the (blurred) full-resolution planes, others are resampled from them
        DevicePtr sequence_level[2] = {frame_0, frame_1};  // sequence only: this level's read-only frame planes
        if (sequence) {
            DevicePtr temp = level > 0 ? Acquire() : 0;
            for (int i = 0; i < 2 && level > 0; ++i) {
                FramePyramid& pyramid = *sequence_frames_[i];
                DevicePtr plane = SequenceLevelPlane(pyramid, static_cast<size_t>(level), current_size.height);
                if (!plane) {
                    Release(temp);
                    Release(frame_0_res);
                    Release(frame_1_res);
                    Release(flow_du);
                    Release(flow_dv);
                    return false;
                }
                if (!pyramid.valid) {
                    op.Clear();
                    op.PushValuePtr("dev_input", i == 0 ? &frame_0 : &frame_1);
                    op.PushValuePtr("dev_output", &plane);
                    op.PushValuePtr("dev_temp", &temp);
                    op.PushValuePtr("data_size", &original_size);
                    op.PushValuePtr("resample_size", &current_size);
                    cuop_resample_.Execute(op);
                failed |= cuop_resample_.TakeFailure();
                }
                sequence_level[i] = plane;
            }
            if (temp) Release(temp);
        } else if (stacked) {  // every level's frames are in place already
        } else if (level == 0) {
            std::swap(frame_0, frame_0_res);
            std::swap(frame_1, frame_1_res);
        } else if (packed) {  // x pass done for all levels: this level's y pass, both frames in one launch
            const size_t column = packed_column[static_cast<size_t>(first_level - level)];
            if (CheckFlow2DError(flow2d_resample_y_pair(context_, AsPlane(packed_frames_[0]) + column, AsPlane(frame_0_res),
                                                        AsPlane(packed_frames_[1]) + column, AsPlane(frame_1_res),
                                                        current_size.width, current_size.height, original_size.height,
                                                        dev_container_size_.pitch),
                                 "flow2d_resample_y_pair"))
                failed = true;
        } else {  // both frames of the level in one resample call (two planes per launch)
            DevicePtr temp = Acquire(), temp_b = Acquire();
            op.Clear();
            op.PushValuePtr("dev_input", &frame_0);
            op.PushValuePtr("dev_output", &frame_0_res);
            op.PushValuePtr("dev_temp", &temp);
            op.PushValuePtr("dev_input_b", &frame_1);
            op.PushValuePtr("dev_output_b", &frame_1_res);
            op.PushValuePtr("dev_temp_b", &temp_b);
            op.PushValuePtr("data_size", &original_size);
            op.PushValuePtr("resample_size", &current_size);
            cuop_resample_.Execute(op);
            failed |= cuop_resample_.TakeFailure();
            Release(temp_b);
            Release(temp);
        }

        // flow: zero at the coarsest level, otherwise previous level -> this level (no magnitude scaling).  Round 6: both ride with
        // the warp of the level (flow2d_upsample_registration_2d: the resample and the registration operators' arithmetic on the
        // same values in one launch -- (u, v) handed on in registers instead of through their planes; at the coarsest level the
        // zeros, over the level's region instead of two memsets of the whole container).
        const bool upsample = prev_size.width != 0;
        // backward registration of `level_1` by the level's flow into `output` (after bringing the flow to the level's size)
        auto warp = [&](DevicePtr level_0, DevicePtr level_1, DevicePtr output) {
            int err;
            if (upsample) {
                err = flow2d_upsample_registration_2d(context_, AsPlane(flow_u), AsPlane(flow_v), prev_size.width, prev_size.height,
                                                      AsPlane(flow_du), AsPlane(flow_dv), AsPlane(level_0), AsPlane(level_1),
                                                      current_size.width, current_size.height, dev_container_size_.pitch, hx, hy,
                                                      AsPlane(output));
                std::swap(flow_u, flow_du);
                std::swap(flow_v, flow_dv);
            } else {
                err = flow2d_upsample_registration_2d(context_, nullptr, nullptr, 0, 0, AsPlane(flow_u), AsPlane(flow_v), AsPlane(level_0),
                                                      AsPlane(level_1), current_size.width, current_size.height,
                                                      dev_container_size_.pitch, hx, hy, AsPlane(output));
            }
            if (CheckFlow2DError(err, "flow2d_upsample_registration_2d")) failed = true;
        };

        DevicePtr solve_frame_0 = frame_0_res;  // what the solver reads as frame 0 of this level
        DevicePtr solve_frame_1 = 0;            // ... and as (warped) frame 1: frame_1_res unless set
        if (stacked) {  // the level planes stay where the y pass of all levels put them; the warp writes the plane kept for it
            const size_t at = level > 0 ? level_row[static_cast<size_t>(level)] * dev_container_size_.pitch : 0;
            DevicePtr level_0 = (level > 0 ? frame_0_res : frame_0) + at, level_1 = (level > 0 ? frame_1_res : frame_1) + at;
            warp(level_0, level_1, level_warp_plane_);
            solve_frame_0 = level_0;
            solve_frame_1 = level_warp_plane_;
        } else if (sequence) {  // the level planes are kept for the next pair: warp into the pool plane, read the rest
            warp(sequence_level[0], sequence_level[1], frame_1_res);
            solve_frame_0 = sequence_level[0];
        } else {  // backward registration of frame 1 by the current flow; the warped frame replaces it
            DevicePtr temp = Acquire();
            warp(frame_0_res, frame_1_res, temp);
            std::swap(frame_1_res, temp);
            Release(temp);
        }

        {  // lagged-diffusivity fixed point: the hot loop
            DevicePtr phi = Acquire(), ksi = Acquire(), temp_du = Acquire(), temp_dv = Acquire();
            op.Clear();
            op.PushValuePtr("dev_frame_0", &solve_frame_0);
            op.PushValuePtr("dev_frame_1", solve_frame_1 ? &solve_frame_1 : &frame_1_res);
            op.PushValuePtr("dev_flow_u", &flow_u);
            op.PushValuePtr("dev_flow_v", &flow_v);
            op.PushValuePtr("dev_flow_du", &flow_du);
            op.PushValuePtr("dev_flow_dv", &flow_dv);
            op.PushValuePtr("dev_phi", &phi);
            op.PushValuePtr("dev_ksi", &ksi);
            op.PushValuePtr("dev_temp_du", &temp_du);
            op.PushValuePtr("dev_temp_dv", &temp_dv);
            op.PushValuePtr("data_constancy", &data_constancy_);
            op.PushValuePtr("outer_iterations_count", &outer_iterations_count);
            op.PushValuePtr("inner_iterations_count", &inner_iterations_count);
            op.PushValuePtr("equation_alpha", &equation_alpha);
            op.PushValuePtr("equation_smoothness", &equation_smoothness);
            op.PushValuePtr("equation_data", &equation_data);
            op.PushValuePtr("data_size", &current_size);
            op.PushValuePtr("hx", &hx);
            op.PushValuePtr("hy", &hy);
            op.PushValuePtr("solver_algorithm", &solver_algorithm);
            op.PushValuePtr("solver_sor_omega", &solver_sor_omega);
            cuop_solve_.silent = true;  // per-level printing would need a host wait; timings are collected instead
            cuop_solve_.Execute(op);
            failed |= cuop_solve_.TakeFailure();
            Release(phi);
            Release(ksi);
            Release(temp_du);
            Release(temp_dv);
        }

        prev_size = current_size;
        if (failed) break;

        {  // u += du, v += dv and the median of u and v after every level, the finest included -- one launch: the filter
           // reads u + du (a single rounded addition, add_2d.cu:33-46) as it goes, the plane of sums is never stored
            DevicePtr temp = Acquire(), temp_b = Acquire();
            // the last median of a ComputeFlowDevice run delivers the result into the caller's planes
            const bool deliver = level == 0 && caller_flow_u_ != 0 && caller_flow_v_ != 0;
            DevicePtr out_u = deliver ? caller_flow_u_ : temp, out_v = deliver ? caller_flow_v_ : temp_b;
            op.Clear();
            op.PushValuePtr("dev_input", &flow_u);
            op.PushValuePtr("dev_output", &out_u);
            op.PushValuePtr("dev_input_b", &flow_v);
            op.PushValuePtr("dev_output_b", &out_v);
            op.PushValuePtr("dev_addend", &flow_du);
            op.PushValuePtr("dev_addend_b", &flow_dv);
            op.PushValuePtr("data_size", &current_size);
            op.PushValuePtr("radius", &median_radius);
            cuop_median_.Execute(op);
            failed |= cuop_median_.TakeFailure();
            if (!deliver) {
                std::swap(flow_u, temp);
                std::swap(flow_v, temp_b);
            }
            Release(temp_b);
            Release(temp);
        }
    }

    // hand the roles back: the caller reads the flow from dev_flow_u_/v_ and releases all four
    if (!sequence) {  // (a sequence pair's frame planes belong to the sequence cache or to the caller)
        dev_frame_0_ = frame_0;
        dev_frame_1_ = frame_1;
    }
    dev_flow_u_ = flow_u;
    dev_flow_v_ = flow_v;
    Release(frame_0_res);
    Release(frame_1_res);
    Release(flow_du);
    Release(flow_dv);
    flow2d_timing_enable(context_, 0);
    if (failed) std::printf("Error: '%s': an operator failed; the flow of this run is not valid.\n", GetName());
    return !failed;
}
