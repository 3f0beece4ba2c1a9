// Coarse-to-fine orchestrator of the flow2d hot path.
// Public interface of the reference's OpticalFlowBase2D / OpticalFlow2D
// (src/optical_flow/optical_flow_base_2d.h:29-52, src/optical_flow/optical_flow_2d.h:43-71):
//   bool Initialize(const DataSize3&, DataConstancy = Grey);
//   void ComputeFlow(Data2D& frame_0, Data2D& frame_1, Data2D& flow_u, Data2D& flow_v, OperationParameters&);
//   void Destroy();   bool silent;
// ComputeFlow bag keys and pointee types (optical_flow_2d.cpp:160-168): warp_levels_count size_t,
// warp_scale_factor float, outer_iterations_count size_t, inner_iterations_count size_t,
// equation_alpha float, equation_smoothness float, equation_data float, median_radius size_t,
// gaussian_sigma float.  Optional superset keys: solver_algorithm int (flow2d_solver_algorithm),
// solver_sor_omega float (opt-in red-black SOR; the default 0 keeps the reference's Jacobi sweeps).
//
// MI355X-first differences (results unchanged): every launch of a pair is queued on one HIP stream
// with no host synchronisation until the flow is copied back (the reference blocks after every
// sweep, cuda_operation_solve_2d.cpp:291); frames may already live in HBM (ComputeFlowDevice).
#pragma once

#include <cstddef>
#include <functional>
#include <map>
#include <vector>

#include "cuda_operations_2d.h"
#include "data2d.h"
#include "data_structs.h"
#include "operation_parameters.h"

class OpticalFlowBase2D {
public:
    const char* GetName() const { return name_; }

    virtual bool Initialize(const DataSize3& data_size, DataConstancy data_constancy = DataConstancy::Grey) = 0;
    virtual void ComputeFlow(Data2D& frame_0, Data2D& frame_1, Data2D& flow_u, Data2D& flow_v,
                             OperationParameters& params);
    virtual void Destroy();
    virtual ~OpticalFlowBase2D();

    // Number of usable pyramid levels (src/optical_flow/optical_flow_base_2d.cpp:36-59); public here so
    // callers and tests can size a run.
    size_t GetMaxWarpLevel(size_t width, size_t height, float scale_factor) const;

protected:
    explicit OpticalFlowBase2D(const char* name) : name_(name) {}
    bool IsInitialized() const;

    bool initialized_ = false;
    DataConstancy data_constancy_ = DataConstancy::Grey;

private:
    const char* name_ = nullptr;
};

struct FlowLevelTiming {
    size_t width, height;
    float solve_ms;   // device time of the level's whole solve call (reference timer, cuda_operation_solve_2d.cpp:220,302)
    float kernel_ms;  // timing_mode 2: summed launch durations of the level's dominant solver kernel, else -1
    int kernel_launches;
    int algorithm;
    double bytes_per_launch;  // algorithmic bytes one launch of that kernel accounts for
};

class OpticalFlow2D : public OpticalFlowBase2D {
public:
    OpticalFlow2D();
    ~OpticalFlow2D() override;

    bool Initialize(const DataSize3& data_size, DataConstancy data_constancy = DataConstancy::Grey) override;
    void ComputeFlow(Data2D& frame_0, Data2D& frame_1, Data2D& flow_u, Data2D& flow_v,
                     OperationParameters& params) override;
    void Destroy() override;

    // Same computation for frames that already sit in pitched device containers of the initialised
    // size (pitch = ContainerSize().pitch).  dev_frame_* are read, dev_flow_* are written.  Nothing is
    // synchronised: the work is queued on the context's stream.  Returns false on a bad argument.
    bool ComputeFlowDevice(DevicePtr dev_frame_0, DevicePtr dev_frame_1, DevicePtr dev_flow_u, DevicePtr dev_flow_v,
                           OperationParameters& params);

    // Flows of an image sequence: flow k is the flow from frames[k] to frames[k + 1] (frame_count - 1 flows), each
    // bit-identical to ComputeFlowDevice on that pair.  Every frame's pre-blurred plane and pyramid levels are
    // computed once and serve first as the second, then as the first frame of consecutive pairs.
    // Frames are only read; queued on the context's stream, launched eagerly (no graph).
    bool ComputeFlowSequenceDevice(const DevicePtr* dev_frames, size_t frame_count, const DevicePtr* dev_flows_u,
                                   const DevicePtr* dev_flows_v, OperationParameters& params);

    // When set, ComputeFlowDevice records the whole pyramid of a pair into a HIP graph the first time it
    // sees a (buffers, parameters) combination and replays it afterwards: one host call instead of
    // several hundred launches.  Ignored while timing_mode != 0 (events are not captured).
    bool use_graph = false;

    // Lock-step groups of independent pairs (set BEFORE Initialize; 1 = off).  Every plane of the pool becomes
    // `group_size` containers tall and ComputeFlowDevice computes group_size pairs at once: pair g of each of the four
    // caller planes lives GroupStrideBytes() * g behind the pointer passed (tall containers, pairs one below the other).
    // Every launch then holds the work of the whole group (flow2d_context_set_batch), so a level of a mid-size frame
    // fills the chip and the launch-bound coarse levels cost one launch per group instead of one per pair; each pair's
    // flow is bit-identical to its own ComputeFlowDevice.  Not available for ComputeFlow (host images) and sequences.
    size_t group_size = 1;
    size_t GroupStrideBytes() const { return dev_container_size_.pitch * dev_container_size_.height; }

    // A lock-step group formed from `count` (1 .. group_size) independent pairs, every plane a container of its own
    // anywhere on the device: the frames are gathered into the object's tall staging containers (one launch,
    // flow2d_copy_planes), the group is computed like a tall-container group of `count` pairs, and the flows are handed
    // back to the callers' planes (one launch).  Queued, not synchronised; with use_graph the whole sequence is recorded
    // per (planes, parameters) combination and replayed.  Each pair's flow is bit-identical to its own ComputeFlowDevice.
    bool ComputeFlowGroupDevice(size_t count, const DevicePtr* dev_frames_0, const DevicePtr* dev_frames_1,
                                const DevicePtr* dev_flows_u, const DevicePtr* dev_flows_v, OperationParameters& params);

    const DataSize3& ContainerSize() const { return dev_container_size_; }
    // Device time of the last ComputeFlow (events around upload..download), milliseconds.
    float LastTotalMs() const { return last_total_ms_; }
    // ComputeFlow is void, like the reference's: whether the last call delivered a flow (false after a missing key,
    // a parameter no level can run with, or a failed operator -- the caller's flow images are then left untouched)
    bool LastRunSucceeded() const { return last_run_ok_; }
    int timing_mode = 0;  // flow2d_timing_enable mode used during a run (0 off, 1 per level, 2 + per kernel launch)
    // One record per level solved since the last ResetLevelTimings() (oldest first); call after the
    // context has been synchronised.  Records accumulate across runs while timing_mode is non-zero.
    std::vector<FlowLevelTiming> LastLevelTimings();
    void ResetLevelTimings();

    bool silent = false;

    // Set BEFORE Initialize.  lone: this object runs its pairs one after the other with nothing else of the same job beside them on
    // the device (the CLI, a lone ComputeFlowDevice user, a batch of one lane) -- latency is what counts, and strip launches that leave
    // half the wave slots empty use the build of the strip kernel with packed arithmetic (flow2d_context_set_lone).
    // OpticalFlowBatch2D clears it for its lanes when there are several: in a pipeline the other lanes' work fills the device.
    bool lone = true;

private:
    bool InitMemory();
    bool InitOperations();
    bool RunPyramid(OperationParameters& params);
    bool QueuePair(DevicePtr dev_frame_0, DevicePtr dev_frame_1, DevicePtr dev_flow_u, DevicePtr dev_flow_v,
                   OperationParameters& params);
    bool QueueScatteredGroup(size_t count, const DevicePtr* dev_frames_0, const DevicePtr* dev_frames_1,
                             const DevicePtr* dev_flows_u, const DevicePtr* dev_flows_v, OperationParameters& params);
    bool ReplayOrRecord(std::vector<unsigned char> key, const std::function<bool()>& queue);
    void DropGraphs();
    DevicePtr Acquire();
    void Release(DevicePtr p);

    static constexpr size_t kContainersCount = 12;  // optical_flow_2d.h:45 of the reference
    DataSize3 dev_container_size_{0, 0, 0};
    size_t group_ = 1;  // group_size as it was at Initialize
    size_t active_group_ = 1;  // pairs of the group being queued (a scattered group may be smaller than group_)
    DevicePtr group_staging_[4] = {0, 0, 0, 0};  // ComputeFlowGroupDevice: tall frame 0, frame 1, flow u, flow v (first use)
    std::vector<DevicePtr> all_planes_;
    std::vector<DevicePtr> free_planes_;
    DevicePtr dev_frame_0_ = 0, dev_frame_1_ = 0, dev_flow_u_ = 0, dev_flow_v_ = 0;  // valid inside a run
    // Two planes beside the pool: the x-resampled rows of all pyramid levels of frame 0 / frame 1, side by side
    // (flow2d_resample_x_levels: one read of each frame for the x passes of the whole pyramid)
    DevicePtr packed_frames_[2] = {0, 0};
    // ComputeFlowDevice only: the caller's planes.  With a pre-blur the frames are read once (by the blur), so
    // they are read in place instead of copied; the last level's median writes the caller's flow planes.
    DevicePtr caller_frame_0_ = 0, caller_frame_1_ = 0, caller_flow_u_ = 0, caller_flow_v_ = 0;
    // ComputeFlowSequenceDevice only: a frame's blurred full-resolution plane (level 0; the caller's own plane when
    // there is no pre-blur) and its resampled levels, kept from one pair to the next.
    struct FramePyramid {
        DevicePtr blurred = 0;             // owned: the pre-blurred frame (allocated on first use)
        DevicePtr level0 = 0;              // what level 0 reads: `blurred`, or the caller's plane without a pre-blur
        std::vector<DevicePtr> levels;     // [l] for l >= 1, container width x level height
        std::vector<size_t> level_rows;
        bool valid = false;
    };
    FramePyramid sequence_cache_[2];
    FramePyramid* sequence_frames_[2] = {nullptr, nullptr};  // non-null inside a sequence pair: frame 0 / frame 1
    DevicePtr SequenceLevelPlane(FramePyramid& pyramid, size_t level, size_t rows);
    void FreeSequenceCache();
    flow2d_context* context_ = nullptr;
    // One plane beside the pool: the warped frame of a level, when the levels of both frames are computed up front into plane
    // regions of their own (RunPyramid: "stacked" levels) and therefore cannot be overwritten by the warp
    DevicePtr level_warp_plane_ = 0;
    float last_total_ms_ = 0.f;
    bool last_run_ok_ = false;
    // recorded pyramids, keyed by the caller buffers and parameters they were recorded for
    struct RecordedGraph {
        void* exec = nullptr;
        unsigned long long last_use = 0;
    };
    std::map<std::vector<unsigned char>, RecordedGraph> graphs_;
    unsigned long long graph_clock_ = 0;
    static constexpr size_t kMaxGraphs = 32;

    CudaOperationAdd2D cuop_add_;
    CudaOperationConvolution2D cuop_convolution_;
    CudaOperationMedian2D cuop_median_;
    CudaOperationRegistration2D cuop_register_;
    CudaOperationResample2D cuop_resample_;
    CudaOperationSolve2D cuop_solve_;
};
