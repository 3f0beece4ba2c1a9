// Shared declarations of the HIP side of the flow2d C-ABI (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "flow2d_c_abi.h"

struct flow2d_timing_slot {
    flow2d_timing_record rec;
    hipEvent_t start;
    hipEvent_t stop;
    std::vector<hipEvent_t> kernel_events;  // mode 2: start/stop pairs of the dominant kernel's launches
};

struct flow2d_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int timing = 0;  // flow2d_timing_enable mode
    size_t timing_min_w = 0, timing_min_h = 0;  // flow2d_timing_launch_filter
    std::vector<hipEvent_t> event_pool;          // recycled timing events
    std::vector<flow2d_timing_slot> timings;
    int num_cus = 256;
    // flow2d_context_set_batch: every launch runs `batch_count` instances, instance b on plane pointers + b * stride
    unsigned batch_count = 1;
    // flow2d_context_set_lone: nothing else of the caller's job runs beside this context's launches
    bool lone = false;
    size_t batch_stride_floats = 0;
    // device counter: waves of the fused kernel that repeated their strip with the plain division (flow2d_fused_fallbacks)
    unsigned int* fused_fallbacks = nullptr;
    unsigned long long* clock_probe = nullptr;  // flow2d_clock_probe_start: per XCD, ticks of the 100 MHz clock and shader cycles
};

// How a kernel finds its instance of a batched launch: grid.z = planes x batch_count, plane = z % planes (the
// u / v or frame 0 / frame 1 choice of the two-plane launches), instance = z / planes.
struct BatchArg {
    unsigned planes;
    unsigned reserved;
    unsigned long long stride;  // floats between consecutive instances of every plane
};

namespace flow2d {

void set_last_error(const char* what, hipError_t err);
void set_last_error_text(const char* what);

// Makes ctx->device current for the calling thread for the duration of a call.
struct DeviceGuard {
    explicit DeviceGuard(const flow2d_context* ctx);
    bool ok() const { return ok_; }
    bool ok_ = true;
};

inline bool plane_args_ok(const void* p, size_t w, size_t h, size_t pitch_bytes)
{
    return p != nullptr && w > 0 && h > 0 && (pitch_bytes % 16) == 0 && pitch_bytes >= w * sizeof(float) &&
           (reinterpret_cast<uintptr_t>(p) % 16) == 0 && w < (1u << 30) && h < (1u << 30);
}

inline unsigned div_up(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

inline BatchArg batch_arg(const flow2d_context* ctx, unsigned planes)
{
    return BatchArg{planes, 0u, static_cast<unsigned long long>(ctx->batch_stride_floats)};
}
inline unsigned batch_z(const flow2d_context* ctx, unsigned planes) { return planes * ctx->batch_count; }

}  // namespace flow2d

#define FLOW2D_HIP_TRY(expr)                              \
    do {                                                  \
        hipError_t flow2d_e_ = (expr);                    \
        if (flow2d_e_ != hipSuccess) {                    \
            ::flow2d::set_last_error(#expr, flow2d_e_);   \
            return (flow2d_e_ == hipErrorOutOfMemory) ? FLOW2D_ERR_OUT_OF_MEMORY : FLOW2D_ERR_DEVICE; \
        }                                                 \
    } while (0)

// (hipGetLastError() is sticky per thread: an error some earlier, already reported or deliberately ignored runtime call
// left behind must not be taken for a failed launch of this entry, so the slate is wiped on the way in.)
#define FLOW2D_ENTER(ctx)                                               \
    if ((ctx) == nullptr) return FLOW2D_ERR_INVALID_ARGUMENT;           \
    ::flow2d::DeviceGuard flow2d_guard_(ctx);                           \
    if (!flow2d_guard_.ok()) return FLOW2D_ERR_DEVICE;                  \
    (void)hipGetLastError()

#define FLOW2D_CHECK_LAUNCH() FLOW2D_HIP_TRY(hipGetLastError())

__device__ __forceinline__ unsigned batch_plane(const BatchArg& b) { return blockIdx.z % b.planes; }
__device__ __forceinline__ size_t batch_offset(const BatchArg& b)
{
    return static_cast<size_t>(blockIdx.z / b.planes) * static_cast<size_t>(b.stride);
}

// XCD-aware tile order of a one-dimensional grid (stencil kernels whose workgroups re-read their neighbours' rows and
// columns).  The hardware deals workgroups to the eight XCDs in turn -- workgroup id % 8 -- and every XCD has an L2 of
// its own, so with a plain 2-D grid the x neighbours of a tile run on OTHER XCDs and every halo column is fetched from
// HBM again (64-byte sectors of the neighbour's 256-byte row: 1.7x the algorithmic bytes measured on the sweep kernels).
// Here XCD k owns the horizontal band of tile rows [k * rows_per_xcd, (k + 1) * rows_per_xcd) and walks it row-major:
// both the x and the y neighbours of a tile are workgroups of the same XCD, a few ids apart.  Which XCD gets id % 8 == k
// is the hardware's business and only affects speed.
struct XcdTiles {
    unsigned tiles_x, tiles_y, rows_per_xcd;
};
inline XcdTiles xcd_tiles(unsigned tiles_x, unsigned tiles_y) { return XcdTiles{tiles_x, tiles_y, (tiles_y + 7u) / 8u}; }
inline unsigned xcd_grid(const XcdTiles& t) { return 8u * t.rows_per_xcd * t.tiles_x; }
__device__ __forceinline__ bool xcd_tile(const XcdTiles& t, unsigned id, unsigned& tile_x, unsigned& tile_y)
{
    const unsigned xcd = id & 7u, j = id >> 3;
    tile_y = xcd * t.rows_per_xcd + j / t.tiles_x;
    tile_x = j % t.tiles_x;
    return tile_y < t.tiles_y;  // the last band may be short: its surplus workgroups leave at once
}

// Reflect-without-repeat index of the solver / median halos: -k -> k, n-1+k -> n-1-k.
__device__ __forceinline__ int mirror_index(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - i - 2 : i;
}
