// Pyramid-side kernels of the flow2d hot path for gfx950: flow update (add), Gaussian pre-blur,
// area-weighted resampling and backward bilinear registration (warp).
//
// These are HBM-bound streaming / gather kernels that run a handful of times per pyramid level
// (< 3 % of a level's bytes).  Layout: one wave = 64 consecutive pixels of one row so
// every global access of a wave is one contiguous 256-byte row segment; blocks are 64x4.
// All arithmetic follows the reference's operation order (no FMA contraction) -- see each kernel.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>

#include "common.hpp"

namespace {

constexpr int kBlockX = 64;
constexpr int kBlockY = 4;

inline dim3 grid_for(size_t w, size_t h) { return dim3(flow2d::div_up(w, kBlockX), flow2d::div_up(h, kBlockY)); }

// ---- add_2d: src/kernels/add_2d.cu:33-46 --------------------------------------------------------
// float4 body (four pixels per lane, 1 KiB per wave-instruction) + scalar tail.
// Two independent planes per launch (grid.z = 2 picks the second set): u += du and v += dv, the two frames of a
// level, ... come in pairs, and on the small levels a launch costs more than its work.
__global__ __launch_bounds__(256) void add_2d_kernel(float* __restrict__ op0_a, const float* __restrict__ op1_a,
                                                     float* __restrict__ op0_b, const float* __restrict__ op1_b, int w,
                                                     int h, int pitch, BatchArg batch)
{
    float* __restrict__ op0 = (batch_plane(batch) ? op0_b : op0_a) + batch_offset(batch);
    const float* __restrict__ op1 = (batch_plane(batch) ? op1_b : op1_a) + batch_offset(batch);
    const int x4 = (blockIdx.x * kBlockX + threadIdx.x) * 4;
    const int y = blockIdx.y * kBlockY + threadIdx.y;
    if (y >= h || x4 >= w) return;
    const size_t off = static_cast<size_t>(y) * pitch + x4;
    if (x4 + 3 < w) {
        float4 a = *reinterpret_cast<const float4*>(op0 + off);
        const float4 b = *reinterpret_cast<const float4*>(op1 + off);
        a.x += b.x;
        a.y += b.y;
        a.z += b.z;
        a.w += b.w;
        *reinterpret_cast<float4*>(op0 + off) = a;
    } else {
        for (int i = 0; x4 + i < w; ++i) op0[off + i] += op1[off + i];
    }
}

// ---- Gaussian rows / columns: src/kernels/convolution_2d.cu:74-168, 181-261 ---------------------
// Zero padding outside the image; sum accumulated j = -r..r with tap[r - j] (fp32, no FMA).
struct GaussTaps {
    float t[51];
};

template <bool kRows>
__global__ __launch_bounds__(256) void gauss_kernel(float* __restrict__ dst, const float* __restrict__ src, int w,
                                                    int h, int pitch, int radius, GaussTaps taps)
{
    const int x = blockIdx.x * kBlockX + threadIdx.x;
    const int y = blockIdx.y * kBlockY + threadIdx.y;
    if (x >= w || y >= h) return;
    float sum = 0.f;
    for (int j = -radius; j <= radius; ++j) {
        float s;
        if (kRows) {
            const int xx = x + j;
            s = (xx >= 0 && xx < w) ? src[static_cast<size_t>(y) * pitch + xx] : 0.f;
        } else {
            const int yy = y + j;
            s = (yy >= 0 && yy < h) ? src[static_cast<size_t>(yy) * pitch + x] : 0.f;
        }
        sum += taps.t[radius - j] * s;
    }
    dst[static_cast<size_t>(y) * pitch + x] = sum;
}

// Rows pass and columns pass of the Gaussian in one launch.  A 64x4 workgroup owns a 64 x kBlurTile tile:
// it evaluates the rows pass for the tile's rows plus `radius` rows above and below into LDS (rows outside
// the image are zero, exactly what the columns pass of the reference sees through its zero padding), then
// the columns pass reads LDS.  Same taps, same accumulation order, fp32 intermediate: bit-identical to the
// two-launch form, with the intermediate plane never touching DRAM.
constexpr int kBlurTile = 32;
constexpr int kBlurMaxRadius = 25;

__global__ __launch_bounds__(256) void gauss_fused_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                          int w, int h, int pitch, int radius, GaussTaps taps)
{
    __shared__ float rows[kBlurTile + 2 * kBlurMaxRadius][kBlockX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * kBlockX + tx;
    const int y0 = blockIdx.y * kBlurTile;
    const int staged = kBlurTile + 2 * radius;
    for (int i = ty; i < staged; i += kBlockY) {
        const int y = y0 - radius + i;
        float sum = 0.f;
        if (y >= 0 && y < h && x < w) {
            const float* line = src + static_cast<size_t>(y) * pitch;
            for (int j = -radius; j <= radius; ++j) {
                const int xx = x + j;
                const float s = (xx >= 0 && xx < w) ? line[xx] : 0.f;
                sum += taps.t[radius - j] * s;
            }
        }
        rows[i][tx] = sum;
    }
    __syncthreads();
    if (x >= w) return;
    for (int i = ty; i < kBlurTile; i += kBlockY) {
        const int y = y0 + i;
        if (y >= h) break;
        float sum = 0.f;
        for (int j = -radius; j <= radius; ++j) sum += taps.t[radius - j] * rows[i + radius + j][tx];
        dst[static_cast<size_t>(y) * pitch + x] = sum;
    }
}

// Streaming form of the same two passes for small radii (R <= kBlurStreamMaxRadius, i.e. sigma < 2.4): a wave
// owns 64 columns and walks down the image.  Per image row every lane loads ONE value, fetches the R values
// on either side from the adjacent lanes (chained DPP shifts; the shift-in value 0 is exactly the reference's
// zero padding at the image border, and columns right of the image are loaded as 0), forms the rows-pass sum
// and keeps the sums of the last 2R+1 rows in registers; the columns pass of output row y runs as soon as row
// y+R has gone through the rows pass.  Same taps, same accumulation order (j = -R..R, sum += tap * value,
// fp32, no FMA): bit-identical to the two launches, one load and one store per pixel.
constexpr int kBlurStreamMaxRadius = 6;

__device__ __forceinline__ float blur_lane_left(float v)  // lane i <- lane i-1, 0 into lane 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float blur_lane_right(float v)  // lane i <- lane i+1, 0 into lane 63
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// rows pass of one image row at this lane's column, from the lane's own value c
template <int R>
__device__ __forceinline__ float blur_row_sum(float c, const GaussTaps& taps)
{
    float left[R + 1], right[R + 1];
    left[0] = right[0] = c;
    // (the lane shifts as one run at raised issue priority: a DPP instruction takes an issue turn alone, and a wave that gets its 2R of
    //  them through back to back leaves the turns in between to pairs of plain instructions -- round 5's rule for the strip kernel,
    //  there applied by csrc/issue_priority.py; 39 -> 34 us per 4096^2 frame)
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int k = 1; k <= R; ++k) {
        left[k] = blur_lane_left(left[k - 1]);
        right[k] = blur_lane_right(right[k - 1]);
    }
    __builtin_amdgcn_s_setprio(0);
    float sum = 0.f;
#pragma unroll
    for (int j = -R; j <= R; ++j) sum += taps.t[R - j] * (j < 0 ? left[-j] : right[j]);
    return sum;
}

// Step J of 2R+1: image row `ry` enters ring slot J, output row ry - R leaves.
// keep ? v : +0.f as a bit mask.  (Written as a select, the compiler turns it into a branch around the LOAD of v with a
// wait for it inside: every row's load was waited for where it was issued, and a wave walked its strip one memory
// latency per row.)
__device__ __forceinline__ float value_or_zero(float v, bool keep)
{
    return __int_as_float(__float_as_int(v) & -static_cast<int>(keep));
}

constexpr int kBlurAhead = 4;  // (six or eight rows in flight are SLOWER: 36.6 -> 37-40 -> 40.4 us, profiles/r06_experiments)  image rows in flight per lane: a wave has one load per row, and a row is a microsecond away

template <int R, int J>
__device__ __forceinline__ void blur_step(float (&ring)[2 * R + 1], float (&next)[kBlurAhead], float* __restrict__ dst,
                                          const float* __restrict__ src, int ry, int y0, int y1, int h, int pitch,
                                          int xc, bool in_image, bool lane_stores, const GaussTaps& taps)
{
    constexpr int N = 2 * R + 1;
    const float c = next[0];  // row ry, requested kBlurAhead steps ago
    {
#pragma unroll
        for (int i = 0; i + 1 < kBlurAhead; ++i) next[i] = next[i + 1];
        const int rn = ry + kBlurAhead;
        const float v = src[static_cast<size_t>(min(max(rn, 0), h - 1)) * pitch + xc];
        next[kBlurAhead - 1] = value_or_zero(v, in_image && rn >= 0 && rn < h);
    }
    ring[J] = blur_row_sum<R>(c, taps);
    const int yo = ry - R;
    if (yo >= y0 && yo < y1) {  // wave-uniform
        float sum = 0.f;
#pragma unroll
        for (int j = -R; j <= R; ++j) sum += taps.t[R - j] * ring[(J + N - R + j) % N];  // row yo + j
        if (lane_stores) dst[static_cast<size_t>(yo) * pitch + xc] = sum;
    }
}

template <int R, size_t... Js>
__device__ __forceinline__ void blur_steps(float (&ring)[2 * R + 1], float (&next)[kBlurAhead], float* __restrict__ dst,
                                           const float* __restrict__ src, int ry, int y0, int y1, int h, int pitch,
                                           int xc, bool in_image, bool lane_stores, const GaussTaps& taps,
                                           std::index_sequence<Js...>)
{
    (blur_step<R, static_cast<int>(Js)>(ring, next, dst, src, ry + static_cast<int>(Js), y0, y1, h, pitch, xc, in_image,
                                        lane_stores, taps),
     ...);
}

template <int R>
__global__ __launch_bounds__(256) void gauss_stream_kernel(float* __restrict__ dst, const float* __restrict__ src, int w,
                                                           int h, int pitch, int rows_per_strip, GaussTaps taps,
                                                           BatchArg batch)
{
    dst += batch_offset(batch);
    src += batch_offset(batch);
    constexpr int N = 2 * R + 1, kValid = 64 - 2 * R;
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip * kValid >= w) return;
    const int x = strip * kValid - R + lane;
    const int xc = min(max(x, 0), w - 1);
    const bool in_image = x >= 0 && x < w;
    const bool lane_stores = lane >= R && lane < 64 - R && x < w;
    const int y0 = blockIdx.y * rows_per_strip;
    const int y1 = min(y0 + rows_per_strip, h);
    float ring[N];
#pragma unroll
    for (int i = 0; i < N; ++i) ring[i] = 0.f;
    // rows y0-R .. y1-1+R go through the rows pass (rows outside the image are zero rows)
    const int r_first = y0 - R, r_last = y1 - 1 + R;
    float next[kBlurAhead];
#pragma unroll
    for (int i = 0; i < kBlurAhead; ++i) {
        const int rn = r_first + i;
        const float v = src[static_cast<size_t>(min(max(rn, 0), h - 1)) * pitch + xc];
        next[i] = value_or_zero(v, in_image && rn >= 0 && rn < h);
    }
    for (int ry = r_first; ry <= r_last; ry += N)
        blur_steps<R>(ring, next, dst, src, ry, y0, y1, h, pitch, xc, in_image, lane_stores, taps,
                      std::make_index_sequence<N>{});
}

template <int R>
void launch_gauss_stream(flow2d_context* ctx, float* dst, const float* src, size_t width, size_t height,
                         size_t pitch_bytes, const GaussTaps& t)
{
    const long strips = flow2d::div_up(width, 64 - 2 * R);
    const long want_waves = (ctx->num_cus > 0 ? ctx->num_cus : 256) * 4 * 4;  // four waves per SIMD
    long rows = 128;
    while (rows > 16 && strips * (long)flow2d::div_up(height, rows) * (long)ctx->batch_count < want_waves) rows /= 2;
    const dim3 grid(flow2d::div_up(strips, 4), flow2d::div_up(height, rows), flow2d::batch_z(ctx, 1));
    gauss_stream_kernel<R><<<grid, 256, 0, ctx->stream>>>(dst, src, (int)width, (int)height, (int)(pitch_bytes / 4),
                                                          (int)rows, t, flow2d::batch_arg(ctx, 1));
}

// ---- area-weighted resampling: src/kernels/resample_2d.cu:34-75 (x), :77-118 (y) ----------------
template <bool kAlongX>
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ in_a, float* __restrict__ out_a,
                                                       const float* __restrict__ in_b, float* __restrict__ out_b,
                                                       int out_w, int out_h, int in_n, int pitch, float delta,
                                                       float normalization, int rows_per_thread, BatchArg batch)
{
    // delta = in_n / (float) out_n, normalization = out_n / (float) in_n (resample_2d.cu:46-47): the same for every
    // output, so the host evaluates the two float divisions (the same IEEE operations) instead of every thread
    const float* __restrict__ in = (batch_plane(batch) ? in_b : in_a) + batch_offset(batch);
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    const int x = blockIdx.x * kBlockX + threadIdx.x;
    if (x >= out_w) return;
    // a thread takes rows_per_thread outputs of its column: four on the finer levels (one output per thread was bound by
    // wave launches there), one on the coarse ones (few outputs, each a long chain of cells)
    for (int row = 0; row < rows_per_thread; ++row) {
    const int y = (blockIdx.y * rows_per_thread + row) * kBlockY + threadIdx.y;
    if (y >= out_h) return;
    const unsigned g = kAlongX ? x : y;
    const float left_f = static_cast<float>(g) * delta;
    const float right_f = static_cast<float>(g + 1u) * delta;
    const int left_i = static_cast<int>(floorf(left_f));
    const int right_i = min(in_n, static_cast<int>(ceilf(right_f)));
    const int cells = right_i - left_i;
    const float* base = kAlongX ? in + static_cast<size_t>(y) * pitch : in + x;
    const size_t stride = kAlongX ? 1 : pitch;
    float value = 0.f;
    // The cells are summed in order (parity), but their loads do not depend on the sum.  Only the first and the last cell
    // carry a fraction other than 1 (resample_2d.cu:58-66; v * 1.f is v); the ones between are plain additions, their
    // loads issued 32 (long chains: the y pass of the coarsest levels of a large frame sums 2048 rows per output with a
    // handful of outputs -- pure latency, 147 -> see profiles/r04 at 8192^2) or 8 at a time.
    if (cells == 1) {
        value += base[static_cast<size_t>(left_i) * stride] * delta;
    } else if (cells > 1) {
        const int last = left_i + cells - 1;
        value += base[static_cast<size_t>(left_i) * stride] * (static_cast<float>(left_i + 1) - left_f);
        int k = left_i + 1;
        for (; k + 32 <= last; k += 32) {
            float v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = base[static_cast<size_t>(k + i) * stride];
#pragma unroll
            for (int i = 0; i < 32; ++i) value += v[i];
        }
        for (; k + 8 <= last; k += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = base[static_cast<size_t>(k + i) * stride];
#pragma unroll
            for (int i = 0; i < 8; ++i) value += v[i];
        }
        for (; k < last; ++k) value += base[static_cast<size_t>(k) * stride];
        value += base[static_cast<size_t>(last) * stride] * (right_f - static_cast<float>(last));
    }
    out[static_cast<size_t>(y) * pitch + x] = value * normalization;
    }
}

// Both passes in one launch, for up-sampling (the flow of the previous pyramid level, optical_flow_2d.cpp:314-338): every
// output evaluates the x pass for the one or two input rows of its y cells -- the same left-to-right cell sum, the same
// normalisation, rounded to float like the temp plane the reference stores it in -- and then the y pass over those
// values.  Same operations in the same order as resample_kernel<true> into a temp followed by resample_kernel<false>,
// hence the same bits; the temp plane (as large as the output in x) is neither written nor read: 160 instead of 288 MB
// for the two flow planes at 2048^2 -> 4096^2.
// (delta = in / out and normalization = out / in of either direction are the same for every output: the host evaluates
//  the two float divisions -- the same IEEE operations -- once instead of every thread four times; an output's x cells and
//  their fractions are worked out once and used for each input row of its y cells)
constexpr int kResampleXYRows = 8;
struct ResampleXY {
    float delta_x, norm_x, delta_y, norm_y;
};

// the x cells of one output column (resample_2d.cu:46-55): worked out once per thread
struct ResampleXCells {
    int left_i, cells_x;
    float first_x, last_x;
};
__device__ __forceinline__ ResampleXCells resample_x_cells(int x, int in_w, const ResampleXY& k)
{
    const float left_f = static_cast<float>(static_cast<unsigned>(x)) * k.delta_x;
    const float right_f = static_cast<float>(static_cast<unsigned>(x) + 1u) * k.delta_x;
    ResampleXCells c;
    c.left_i = static_cast<int>(floorf(left_f));
    c.cells_x = min(in_w, static_cast<int>(ceilf(right_f))) - c.left_i;
    c.first_x = c.cells_x == 1 ? k.delta_x : static_cast<float>(c.left_i + 1) - left_f;
    c.last_x = c.cells_x == 1 ? k.delta_x : right_f - static_cast<float>(c.left_i + c.cells_x - 1);
    return c;
}

// output (column of `c`, row y): the y pass of resample_2d.cu:77-118 over x-pass values computed on the spot
__device__ __forceinline__ float resample_xy_value(const float* __restrict__ in, const ResampleXCells& c, int y, int in_w, int in_h,
                                                   int pitch, const ResampleXY& k)
{
    const int left_i = c.left_i, cells_x = c.cells_x;
    const float first_x = c.first_x, last_x = c.last_x;
    const float top_f = static_cast<float>(static_cast<unsigned>(y)) * k.delta_y;
    const float bottom_f = static_cast<float>(static_cast<unsigned>(y) + 1u) * k.delta_y;
    const int top_i = static_cast<int>(floorf(top_f));
    const int cells = min(in_h, static_cast<int>(ceilf(bottom_f))) - top_i;
    float value = 0.f;
    if (cells_x <= 2 && cells <= 2) {
        // up-sampling: an output has one or two cells in either direction (a third one only where the rounding of
        // g * delta and (g + 1) * delta pushes them more than 1 apart) -- straight-line code, the second cell's
        // product selected in or out; the sums start from 0.f and add in cell order like the loops below
        const float* __restrict__ row0 = in + static_cast<size_t>(top_i) * pitch + left_i;
        const float* __restrict__ row1 = in + static_cast<size_t>(min(top_i + 1, in_h - 1)) * pitch + left_i;
        const int second = min(left_i + 1, in_w - 1) - left_i;
        const float a0 = row0[0], a1 = row0[second], b0 = row1[0], b1 = row1[second];
        float xa = 0.f + a0 * first_x, xb = 0.f + b0 * first_x;
        const float xa2 = xa + a1 * last_x, xb2 = xb + b1 * last_x;
        xa = cells_x == 2 ? xa2 : xa;
        xb = cells_x == 2 ? xb2 : xb;
        const float first_y = cells == 1 ? k.delta_y : static_cast<float>(top_i + 1) - top_f;
        const float last_y = bottom_f - static_cast<float>(top_i + 1);
        value = 0.f + (xa * k.norm_x) * first_y;
        const float value2 = value + (xb * k.norm_x) * last_y;
        value = cells == 2 ? value2 : value;
    } else {
        for (int j = 0; j < cells; ++j) {
            float frac = 1.f;
            if (j == 0) frac = static_cast<float>(top_i + 1) - top_f;
            if (j == cells - 1) frac = bottom_f - static_cast<float>(top_i + j);
            if (cells == 1) frac = k.delta_y;
            const float* __restrict__ row = in + static_cast<size_t>(top_i + j) * pitch + left_i;
            float x_pass = 0.f;  // the x pass of this input row: first cell, whole cells, last cell, in that order (:56-72)
            for (int i = 0; i < cells_x; ++i) {
                const float fx = i == 0 ? first_x : (i == cells_x - 1 ? last_x : 1.f);
                x_pass += row[i] * fx;
            }
            value += (x_pass * k.norm_x) * frac;
        }
    }
    return value * k.norm_y;
}

__global__ __launch_bounds__(256) void resample_xy_kernel(const float* __restrict__ in_a, float* __restrict__ out_a,
                                                          const float* __restrict__ in_b, float* __restrict__ out_b,
                                                          int out_w, int out_h, int in_w, int in_h, int pitch,
                                                          ResampleXY k, BatchArg batch)
{
    const float* __restrict__ in = (batch_plane(batch) ? in_b : in_a) + batch_offset(batch);
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    const int x = blockIdx.x * kBlockX + threadIdx.x;
    if (x >= out_w) return;
    const ResampleXCells c = resample_x_cells(x, in_w, k);
    // a thread walks kResampleXYRows output rows of its column (a wave that lives for one output each spends its time
    // being launched: 80 -> 51 us for the two 4096^2 flow planes; all thirty-two loads of the eight rows issued before the first is
    // used: 51 -> 61 us, round 6 -- the inputs are cache hits, the kernel is bound by its stores)
    for (int i = 0; i < kResampleXYRows; ++i) {
        const int y = (blockIdx.y * kResampleXYRows + i) * kBlockY + threadIdx.y;
        if (y >= out_h) return;
        out[static_cast<size_t>(y) * pitch + x] = resample_xy_value(in, c, y, in_w, in_h, pitch, k);
    }
}

// Down-sampling along x with a large ratio (the frames are resampled from FULL resolution at every level,
// optical_flow_2d.cpp:284-303): with one output per lane the lanes of a wave read 64 different cache
// lines per load.  Here each wave (one image row, 64 outputs) stages the contiguous input span of its
// outputs through LDS in coalesced chunks and the lanes then walk their cells in LDS.  Every output still
// accumulates its cells left to right in fp32, so the result is bit-identical to resample_kernel<true>.
constexpr int kResampleChunk = 2048;                                   // floats of input per wave and pass
constexpr int kResampleLdsPerWave = kResampleChunk + kResampleChunk / 32;  // one pad word per 32: conflict-free strides

__global__ __launch_bounds__(256) void resample_x_lds_kernel(const float* __restrict__ in_a, float* __restrict__ out_a,
                                                             const float* __restrict__ in_b, float* __restrict__ out_b,
                                                             int out_w, int out_h, int in_w, int pitch, BatchArg batch)
{
    const float* __restrict__ in = (batch_plane(batch) ? in_b : in_a) + batch_offset(batch);
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    __shared__ float lds[4][kResampleLdsPerWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + wave;
    const bool row_ok = y < out_h;
    const bool active = row_ok && x < out_w;
    const float delta = static_cast<float>(in_w) / static_cast<float>(out_w);
    const float normalization = static_cast<float>(out_w) / static_cast<float>(in_w);
    // this lane's cells, exactly as resample_2d.cu:46-53
    const int gx = min(x, out_w - 1);
    const float left_f = static_cast<float>(static_cast<unsigned>(gx)) * delta;
    const float right_f = static_cast<float>(static_cast<unsigned>(gx) + 1u) * delta;
    const int left_i = static_cast<int>(floorf(left_f));
    const int right_i = min(in_w, static_cast<int>(ceilf(right_f)));
    const int cells = right_i - left_i;
    // span of the whole wave: first lane's left cell .. last valid lane's right cell (block-uniform)
    const int x_first = blockIdx.x * 64, x_last = min(x_first + 63, out_w - 1);
    const int lo = static_cast<int>(floorf(static_cast<float>(static_cast<unsigned>(x_first)) * delta));
    const int hi = min(in_w, static_cast<int>(ceilf(static_cast<float>(static_cast<unsigned>(x_last) + 1u) * delta)));
    const float* row = in + static_cast<size_t>(min(y, out_h - 1)) * pitch;
    float value = 0.f;
    for (int c0 = lo; c0 < hi; c0 += kResampleChunk) {
        const int c1 = min(c0 + kResampleChunk, hi);
        for (int i = c0 + lane; i < c1; i += 64) {
            const int k = i - c0;
            lds[wave][k + (k >> 5)] = row[i];
        }
        __syncthreads();
        if (active) {
            const int b = max(left_i, c0), e = min(right_i, c1);
            for (int i = b; i < e; ++i) {
                const int j = i - left_i;
                float frac = 1.f;
                if (j == 0) frac = static_cast<float>(left_i + 1) - left_f;
                if (j == cells - 1) frac = right_f - static_cast<float>(left_i + j);
                if (cells == 1) frac = delta;
                const int k = i - c0;
                value += lds[wave][k + (k >> 5)] * frac;
            }
        }
        __syncthreads();
    }
    if (active) out[static_cast<size_t>(y) * pitch + x] = value * normalization;
}

// One trip over the input for the x pass of several pyramid levels (flow2d_resample_x_levels): a workgroup owns one
// image row, stages it in LDS (one pad word per 32, so the lanes' strided cell walks are conflict-free for the
// power-of-two ratios of a 0.5 pyramid) and produces every level's outputs from there.  Per output the cells are
// accumulated left to right exactly as in resample_kernel<true> (resample_2d.cu:46-72): same bits.
//
// An output of a level `in_w / out_w` cells wide is one dependent chain of that many additions, so the deep levels of a
// large frame (8192 -> 4: 2048 cells per output, 4 outputs per row) are all latency and no parallelism.  Levels at
// least a workgroup wide are walked one after the other with every thread busy; ALL narrower levels are then walked
// together, one output per thread, so the row's critical path is its longest chain instead of the sum of them
// (8192^2, 12 levels, both frames: 3.7 ms -> see profiles/).  Inside a chain only the first and the last cell carry
// a fraction other than 1 (resample_2d.cu:58-66), the others are plain additions with eight LDS reads in flight.
// Which wave of a workgroup takes the longest chains: it rotates with the workgroup, so that the long-chain waves of the
// workgroups sharing a CU do not all sit on the last SIMD.  (Only speed depends on it; (block >> 8) + (block >> 3), which
// follows the dealing of workgroups to XCDs and CUs more closely, measured the same.)
__device__ __forceinline__ unsigned long_chain_wave(unsigned block) { return block; }

struct ResampleLevels {
    int count;
    int out_w[FLOW2D_RESAMPLE_MAX_LEVELS];
    int col[FLOW2D_RESAMPLE_MAX_LEVELS];
};

__device__ __forceinline__ float resample_x_output_lds(const float* __restrict__ row, int in_w, int g, float delta,
                                                       float normalization)
{
    const float left_f = static_cast<float>(static_cast<unsigned>(g)) * delta;
    const float right_f = static_cast<float>(static_cast<unsigned>(g) + 1u) * delta;
    const int left_i = static_cast<int>(floorf(left_f));
    const int right_i = min(in_w, static_cast<int>(ceilf(right_f)));
    const int cells = right_i - left_i;
    auto cell = [&](int k) { return row[k + (k >> 5)]; };
    float value = 0.f;
    if (cells == 1) {
        value += cell(left_i) * delta;
    } else if (cells > 1) {
        const int last = left_i + cells - 1;
        value += cell(left_i) * (static_cast<float>(left_i + 1) - left_f);
        // the cells between: fraction 1, value += row[k] * 1.f is value += row[k].  Up to the next multiple of 32 one by
        // one; then whole blocks of 32 cells, which lie contiguously in LDS behind ONE address (the pad word follows each
        // block): 32 reads with immediate offsets in flight and one addition per cell -- the dependent additions are what
        // a 2048-cell chain costs, not the address arithmetic and the read latency of every single cell
        int k = left_i + 1;
        const int head_end = min(last, (k + 31) & ~31);
        for (; k < head_end; ++k) value += cell(k);
        for (; k + 32 <= last; k += 32) {
            const float* __restrict__ block = row + k + (k >> 5);
            float c[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) c[i] = block[i];
#pragma unroll
            for (int i = 0; i < 32; ++i) value += c[i];
        }
        for (; k < last; ++k) value += cell(k);
        value += cell(last) * (right_f - static_cast<float>(last));
    }
    return value * normalization;
}

__global__ __launch_bounds__(256) void resample_x_levels_kernel(const float* __restrict__ in_a, float* __restrict__ out_a,
                                                                const float* __restrict__ in_b, float* __restrict__ out_b,
                                                                int in_w, int pitch, ResampleLevels lv, BatchArg batch)
{
    extern __shared__ float row[];
    const float* __restrict__ in = (batch_plane(batch) ? in_b : in_a) + batch_offset(batch);
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    const size_t line = static_cast<size_t>(blockIdx.x) * pitch;
    for (int i = threadIdx.x; i < in_w; i += 256) row[i + (i >> 5)] = in[line + i];
    __syncthreads();
    // levels at least a workgroup wide: one after the other, all threads busy
    int narrow_total = 0;
    for (int l = 0; l < lv.count; ++l) {
        const int out_w = lv.out_w[l];
        if (out_w < 256) {
            narrow_total += out_w;
            continue;
        }
        const float delta = static_cast<float>(in_w) / static_cast<float>(out_w);
        const float normalization = static_cast<float>(out_w) / static_cast<float>(in_w);
        for (int g = threadIdx.x; g < out_w; g += 256)
            out[line + lv.col[l] + g] = resample_x_output_lds(row, in_w, g, delta, normalization);
    }
    // all narrower levels together: item i of their concatenation -> (level, g).  The concatenation runs from the widest
    // to the narrowest level, i.e. from the shortest to the longest chains, so the long chains meet in ONE wave (dealt over
    // all four, each wave would walk the longest chain with a few lanes active: four times the LDS and VALU instructions);
    // which wave that is rotates with the workgroup -- the last wave of EVERY workgroup sits on the same SIMD of its CU,
    // which then issued the additions of all the long chains while the other three SIMDs idled.
    for (int first = 0; first < narrow_total; first += 256) {
        const int i = first + static_cast<int>((threadIdx.x + 64u * long_chain_wave(blockIdx.x)) & 255u);
        if (i >= narrow_total) continue;
        int g = i, level = -1;
        for (int l = 0; l < lv.count; ++l) {
            const int out_w = lv.out_w[l];
            if (out_w >= 256 || level >= 0) continue;
            if (g < out_w)
                level = l;
            else
                g -= out_w;
        }
        const int out_w = lv.out_w[level];
        const float delta = static_cast<float>(in_w) / static_cast<float>(out_w);
        const float normalization = static_cast<float>(out_w) / static_cast<float>(in_w);
        out[line + lv.col[level] + g] = resample_x_output_lds(row, in_w, g, delta, normalization);
    }
}

// The same for levels whose ratio in_w / out_w is a power of two -- every level of a 0.5 pyramid on a frame whose width
// divides evenly (4096, 8192, 1024, 1920 = 15 * 128 ...).  Then delta is an integer R, every output sums exactly R whole
// cells (all fractions of resample_2d.cu:58-66 are 1, and c * 1.f is c) and normalization is 1 / R: the output is
// ((0 + c[0]) + c[1] + ... + c[R-1]) * (1 / R), additions in cell order.
//  * Levels with R <= 32: a thread owns 32 consecutive cells of the row -- eight 16-byte loads, a whole 128-byte line per
//    lane -- and sums them for every such level out of REGISTERS: no LDS traffic, no address arithmetic per cell, no
//    bank conflicts (the general kernel walks LDS with lane strides of 2 ... 16 words there: 2- to 16-way conflicts).
//  * Levels with R >= 64 (long chains): the cells go to LDS once (32 contiguous words per thread, 16-byte aligned: four pad
//    words per 32 cells and four more per 256, which spreads the cell positions the chains of a wave touch in one step over
//    the banks), and one output per thread walks its R / 32 blocks there: eight 16-byte reads per block, issued a block
//    ahead, and one dependent addition per cell -- a lone wave issues an instruction every four to five cycles whatever its
//    kind, so one read per cell (ds_read_b32) doubled the time of a 2048-cell chain (level 8192 -> 4 alone: 257 us).  Chains of one length share a wave (the concatenation runs from the shortest to
//    the longest), and the wave with the longest ones rotates with the workgroup over the four SIMDs of the CU.
// 8192^2, 11 levels, both frames: 868 us (general kernel, round 3) -> see profiles/r04_experiments.
constexpr int kPow2Cells = 32;  // cells per thread
__device__ __forceinline__ int pow2_lds_index(int k) { return k + 4 * ((k >> 5) + (k >> 8)); }

template <int R>
__device__ __forceinline__ void pow2_emit(const float (&c)[kPow2Cells], float* __restrict__ dst)
{
    constexpr int kOutputs = kPow2Cells / R;
    constexpr float kNorm = 1.f / static_cast<float>(R);
    float r[kOutputs];
#pragma unroll
    for (int o = 0; o < kOutputs; ++o) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j) v += c[o * R + j];
        r[o] = v * kNorm;
    }
    if constexpr (kOutputs >= 4) {
#pragma unroll
        for (int o = 0; o < kOutputs; o += 4) *reinterpret_cast<float4*>(dst + o) = make_float4(r[o], r[o + 1], r[o + 2], r[o + 3]);
    } else if constexpr (kOutputs == 2) {
        *reinterpret_cast<float2*>(dst) = make_float2(r[0], r[1]);
    } else {
        dst[0] = r[0];
    }
}

__global__ __launch_bounds__(256) void resample_x_levels_pow2_kernel(const float* __restrict__ in_a, float* __restrict__ out_a,
                                                                     const float* __restrict__ in_b, float* __restrict__ out_b,
                                                                     int in_w, int pitch, ResampleLevels lv, int deep_total,
                                                                     BatchArg batch)
{
    extern __shared__ float row[];
    const float* __restrict__ in = (batch_plane(batch) ? in_b : in_a) + batch_offset(batch);
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    const size_t line = static_cast<size_t>(blockIdx.x) * pitch;
    const int t = threadIdx.x;
    if (t * kPow2Cells < in_w) {  // (in_w is a multiple of 32: whole threads)
        float c[kPow2Cells];
        const float4* __restrict__ src = reinterpret_cast<const float4*>(in + line + static_cast<size_t>(t) * kPow2Cells);
#pragma unroll
        for (int q = 0; q < kPow2Cells / 4; ++q) {
            const float4 v = src[q];
            c[4 * q] = v.x, c[4 * q + 1] = v.y, c[4 * q + 2] = v.z, c[4 * q + 3] = v.w;
        }
        if (deep_total > 0) {
            float4* __restrict__ mine = reinterpret_cast<float4*>(row + pow2_lds_index(t * kPow2Cells));
#pragma unroll
            for (int q = 0; q < kPow2Cells / 4; ++q) mine[q] = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
        }
        for (int l = 0; l < lv.count; ++l) {
            const int ratio = in_w / lv.out_w[l];  // a power of two (the launcher checked)
            if (ratio > kPow2Cells) continue;
            float* __restrict__ dst = out + line + lv.col[l] + t * (kPow2Cells / ratio);
            switch (ratio) {
                case 2: pow2_emit<2>(c, dst); break;
                case 4: pow2_emit<4>(c, dst); break;
                case 8: pow2_emit<8>(c, dst); break;
                case 16: pow2_emit<16>(c, dst); break;
                default: pow2_emit<32>(c, dst); break;
            }
        }
    }
    if (deep_total == 0) return;
    __syncthreads();
    // the long chains: item i of the concatenation of the levels with R >= 64 (in the order given: the pyramid lists its
    // levels from the finest to the coarsest, i.e. from the shortest chains to the longest)
    for (int first = 0; first < deep_total; first += 256) {
        const int i = first + static_cast<int>((threadIdx.x + 64u * long_chain_wave(blockIdx.x)) & 255u);
        if (i >= deep_total) continue;
        int g = i, level = -1;
        for (int l = 0; l < lv.count; ++l) {
            const int out_w = lv.out_w[l];
            if (in_w / out_w <= kPow2Cells || level >= 0) continue;
            if (g < out_w)
                level = l;
            else
                g -= out_w;
        }
        const int out_w = lv.out_w[level];
        const int blocks = (in_w / out_w) / kPow2Cells;
        float value = 0.f;
        // the reads of the next block are issued before the additions of the current one (two register sets, the blocks of
        // a chain taken in pairs: R >= 64 means an even number of them)
        auto load = [&](float4 (&dst)[kPow2Cells / 4], int b) {
            const float4* __restrict__ block = reinterpret_cast<const float4*>(row + pow2_lds_index((g * blocks + b) * kPow2Cells));
#pragma unroll
            for (int q = 0; q < kPow2Cells / 4; ++q) dst[q] = block[q];
        };
        auto add = [&](const float4 (&src)[kPow2Cells / 4]) {
#pragma unroll
            for (int q = 0; q < kPow2Cells / 4; ++q) {
                value += src[q].x;
                value += src[q].y;
                value += src[q].z;
                value += src[q].w;
            }
        };
        float4 a[kPow2Cells / 4], n[kPow2Cells / 4];
        // (the scheduling barriers keep the reads where they are written: left alone, the scheduler sinks them below the
        //  additions of the block before, right in front of their first use, and every block waits for its read latency)
        load(a, 0);
        for (int b = 0; b < blocks; b += 2) {
            load(n, b + 1);
            __builtin_amdgcn_sched_barrier(0);
            add(a);
            __builtin_amdgcn_sched_barrier(0);
            load(a, min(b + 2, blocks - 1));  // (after the last pair: a read nothing uses)
            __builtin_amdgcn_sched_barrier(0);
            add(n);
            __builtin_amdgcn_sched_barrier(0);
        }
        out[line + lv.col[level] + g] = value * (static_cast<float>(out_w) / static_cast<float>(in_w));
    }
}

// ---- backward registration: src/kernels/registration_2d.cu:34-73 --------------------------------
// (1.f / hx and 1.f / hy are the same for every pixel: evaluated by the host, the same IEEE division.  A thread takes
//  kRegistrationRows pixels of its column: their flow values are requested together, then the four frame values each.)
constexpr int kRegistrationRows = 4;

// one pixel of registration_2d.cu:34-73 (c = its offset in the planes)
__device__ __forceinline__ float registered_value(const float* __restrict__ f0, const float* __restrict__ f1, int gx, int gy, size_t c,
                                                  float uu, float vv, int w, int h, int pitch, float inv_hx, float inv_hy)
{
    const float x_f = static_cast<float>(gx) + (uu * inv_hx);
    const float y_f = static_cast<float>(gy) + (vv * inv_hy);
    if ((x_f < 0.f) || (x_f > static_cast<float>(w - 1)) || (y_f < 0.f) || (y_f > static_cast<float>(h - 1)) || isnan(x_f) ||
        isnan(y_f))
        return f0[c];
    const int x = static_cast<int>(floorf(x_f));
    const int y = static_cast<int>(floorf(y_f));
    const float dx = x_f - static_cast<float>(x);
    const float dy = y_f - static_cast<float>(y);
    const int x1 = min(w - 1, x + 1);
    const int y1 = min(h - 1, y + 1);
    // x and x1 lie in the column pair (xb, xb + 1) with xb = min(x, w - 2): each row's two values come as ONE eight-byte gather (the
    // target takes dword-aligned dwordx2 loads) instead of two -- the kernel is bound by its gathers' address processing, not by bytes
    // (round 6: registration alone 69 -> 63 us at 4096^2, the one-launch warp 23 -> 18 us at 2048^2).  (w = 1: xb = 0 and the second column is row padding, never selected.)
    const int xb = max(min(x, w - 2), 0);
    const float* r0 = f1 + static_cast<size_t>(y) * pitch + xb;
    const float* r1 = f1 + static_cast<size_t>(y1) * pitch + xb;
    const float a0 = r0[0], a1 = r0[1], b0 = r1[0], b1 = r1[1];
    const bool x_second = x != xb, x1_second = x1 != xb;
    const float r0x = x_second ? a1 : a0, r0x1 = x1_second ? a1 : a0, r1x = x_second ? b1 : b0, r1x1 = x1_second ? b1 : b0;
    return (1.f - dx) * (1.f - dy) * r0x + (dx) * (1.f - dy) * r0x1 + (1.f - dx) * (dy)*r1x + (dx) * (dy)*r1x1;
}

__global__ __launch_bounds__(256) void registration_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                           const float* __restrict__ u, const float* __restrict__ v,
                                                           int w, int h, int pitch, float inv_hx, float inv_hy,
                                                           float* __restrict__ out, BatchArg batch)
{
    f0 += batch_offset(batch);
    f1 += batch_offset(batch);
    u += batch_offset(batch);
    v += batch_offset(batch);
    out += batch_offset(batch);
    const int gx = blockIdx.x * kBlockX + threadIdx.x;
    if (gx >= w) return;
    float uu[kRegistrationRows], vv[kRegistrationRows];
#pragma unroll
    for (int i = 0; i < kRegistrationRows; ++i) {
        const int gy = min((blockIdx.y * kRegistrationRows + i) * kBlockY + threadIdx.y, h - 1);
        const size_t c = static_cast<size_t>(gy) * pitch + gx;
        uu[i] = u[c];
        vv[i] = v[c];
    }
    // (all sixteen gathers of the thread's four rows issued before the first is used: no faster, 67.7 against 68.2-69.5 us at 4096^2 --
    //  the kernel moves 4.9 TB/s as it is; profiles/r06_experiments)
#pragma unroll
    for (int i = 0; i < kRegistrationRows; ++i) {
        const int gy = (blockIdx.y * kRegistrationRows + i) * kBlockY + threadIdx.y;
        if (gy >= h) return;
        const size_t c = static_cast<size_t>(gy) * pitch + gx;
        out[c] = registered_value(f0, f1, gx, gy, c, uu[i], vv[i], w, h, pitch, inv_hx, inv_hy);
    }
}

// The flow of the previous level brought to this level's size AND frame 1 warped by it, in one launch (round 6;
// optical_flow_2d.cpp:314-338 followed by :341-362): a thread evaluates (u, v) of its pixels exactly as resample_xy_kernel does,
// stores them and hands them to the warp of the same pixels instead of a second kernel reading them back -- one launch and 8 bytes
// per pixel less at every level but the coarsest; the same operations on the same values, the same bits.
// DOUBLE: the level is exactly twice the previous one in both directions (every level of a 0.5 pyramid over power-of-two frames, the fine
// levels of most others).  Then delta = 0.5 and every output has ONE cell in either direction -- g * 0.5 and (g + 1) * 0.5 are exact,
// ceil((g + 1) / 2) - floor(g / 2) = 1 for every g -- so of the general form's four loads per plane and pixel only the first one counts,
// and the four rows of a thread, taken ADJACENT here (4j .. 4j + 3; h is even: whole pairs are inside the level or below it), share two
// input rows: 4 loads instead of 32 in front of the thread's sixteen gathers (4096^2: 120 -> see profiles/r06_experiments).  The value
// goes through the general form's operations for one cell -- ((0 + a * delta_x) * norm_x, 0 + that * delta_y, * norm_y) -- bit for bit.
// ZERO: the coarsest level -- no previous flow: (u, v) = 0 is stored and warped by (the reference fills both planes with two memsets
// of the whole container first, optical_flow_2d.cpp:308-313; what lies outside the level's region of a plane is never read).
enum { kUpsampleGeneral = 0, kUpsampleDouble = 1, kUpsampleZero = 2 };
template <int MODE>
__global__ __launch_bounds__(256) void upsample_registration_kernel(const float* __restrict__ in_u, const float* __restrict__ in_v,
                                                                    float* __restrict__ out_u, float* __restrict__ out_v,
                                                                    const float* __restrict__ f0, const float* __restrict__ f1,
                                                                    float* __restrict__ warped, int w, int h, int in_w, int in_h,
                                                                    int pitch, ResampleXY k, float inv_hx, float inv_hy, BatchArg batch)
{
    if (MODE != kUpsampleZero) {  // (no previous flow at the coarsest level: null planes)
        in_u += batch_offset(batch);
        in_v += batch_offset(batch);
    }
    out_u += batch_offset(batch);
    out_v += batch_offset(batch);
    f0 += batch_offset(batch);
    f1 += batch_offset(batch);
    warped += batch_offset(batch);
    const int gx = blockIdx.x * kBlockX + threadIdx.x;
    if (gx >= w) return;
    constexpr bool DOUBLE = MODE == kUpsampleDouble;
    const ResampleXCells cells = resample_x_cells(gx, in_w, k);
    float uu[kRegistrationRows], vv[kRegistrationRows];
    int rows[kRegistrationRows];
    if (MODE == kUpsampleZero) {
#pragma unroll
        for (int i = 0; i < kRegistrationRows; ++i) {
            rows[i] = (blockIdx.y * kRegistrationRows + i) * kBlockY + threadIdx.y;
            uu[i] = vv[i] = 0.f;
        }
    } else if (DOUBLE) {
        static_assert(kRegistrationRows == 4, "two input rows per thread");
        const int base = (blockIdx.y * kBlockY + threadIdx.y) * kRegistrationRows;
        float a[2], b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t at = static_cast<size_t>(min(base + 2 * j, h - 1) >> 1) * pitch + cells.left_i;
            a[j] = in_u[at];
            b[j] = in_v[at];
        }
#pragma unroll
        for (int i = 0; i < kRegistrationRows; ++i) {
            rows[i] = base + i;
            const float xu = 0.f + a[i >> 1] * k.delta_x, xv = 0.f + b[i >> 1] * k.delta_x;
            const float yu = 0.f + (xu * k.norm_x) * k.delta_y, yv = 0.f + (xv * k.norm_x) * k.delta_y;
            uu[i] = yu * k.norm_y;
            vv[i] = yv * k.norm_y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kRegistrationRows; ++i) {
            rows[i] = (blockIdx.y * kRegistrationRows + i) * kBlockY + threadIdx.y;
            const int gy = min(rows[i], h - 1);  // (below the level: a valid row again)
            uu[i] = resample_xy_value(in_u, cells, gy, in_w, in_h, pitch, k);
            vv[i] = resample_xy_value(in_v, cells, gy, in_w, in_h, pitch, k);
        }
    }
#pragma unroll
    for (int i = 0; i < kRegistrationRows; ++i) {
        const int gy = rows[i];
        if (gy >= h) return;
        const size_t c = static_cast<size_t>(gy) * pitch + gx;
        out_u[c] = uu[i];
        out_v[c] = vv[i];
        warped[c] = registered_value(f0, f1, gx, gy, c, uu[i], vv[i], w, h, pitch, inv_hx, inv_hy);
    }
}

}  // namespace

namespace {
// ---- the y passes of several levels in one launch (round 6) -----------------------------------------------------------
// After flow2d_resample_x_levels the packed planes hold, for every level, the x-resampled rows at full height; each level's y
// pass (resample_2d.cu:77-118) reads its column segment.  One launch per level cost a config-3 pair seven launches of 7-39 us,
// most of them far too small to fill the device; here block rows [first_y[l], first_y[l + 1]) of the grid belong to level l and
// run the body of resample_kernel<false> on that level's geometry: the same cell sums in the same order, the same bits.
struct ResampleYLevels {
    int count;
    int out_w[FLOW2D_RESAMPLE_MAX_LEVELS], out_h[FLOW2D_RESAMPLE_MAX_LEVELS], col[FLOW2D_RESAMPLE_MAX_LEVELS];
    int out_row[FLOW2D_RESAMPLE_MAX_LEVELS];   // first row of the level's plane region in the output planes
    int first_block[FLOW2D_RESAMPLE_MAX_LEVELS + 1];  // the level's blocks in the one-dimensional grid (blocks_x[l] per block row)
    int blocks_x[FLOW2D_RESAMPLE_MAX_LEVELS];
    int rows_per_thread[FLOW2D_RESAMPLE_MAX_LEVELS];
    int max_cells[FLOW2D_RESAMPLE_MAX_LEVELS];  // an upper bound of the cells of one output of the level
    float delta[FLOW2D_RESAMPLE_MAX_LEVELS], normalization[FLOW2D_RESAMPLE_MAX_LEVELS];
};

// The outputs of one thread when every output has at most MAXC cells (the fine levels: a ratio of 2 has two cells per output, 4
// four, 8 eight): ALL loads of the thread's rows first, then the sums -- in resample_kernel<false>'s order (first cell with its
// fraction, the middle ones plain, the last with its fraction), so the same bits.  One output at a time left two loads in
// flight per thread and the level-1 pass latency-bound at 2.6 TB/s.
template <int MAXC, int ROWS>
__device__ __forceinline__ void resample_y_rows(const float* __restrict__ base, float* __restrict__ out, int x, int block_y, int out_h,
                                                int in_h, int pitch, float delta, float normalization)
{
    float v[ROWS][MAXC];
    int cells[ROWS], ys[ROWS];
    float first_f[ROWS], last_f[ROWS];
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
        const int y = (block_y * ROWS + row) * kBlockY + static_cast<int>(threadIdx.y);
        ys[row] = y;
        const int yc = min(y, out_h - 1);  // (a thread below the level computes a valid row again and does not store it)
        const float left_f = static_cast<float>(static_cast<unsigned>(yc)) * delta;
        const float right_f = static_cast<float>(static_cast<unsigned>(yc) + 1u) * delta;
        const int left_i = static_cast<int>(floorf(left_f));
        const int right_i = min(in_h, static_cast<int>(ceilf(right_f)));
        cells[row] = right_i - left_i;
        first_f[row] = cells[row] == 1 ? delta : static_cast<float>(left_i + 1) - left_f;
        last_f[row] = right_f - static_cast<float>(left_i + cells[row] - 1);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) v[row][c] = base[static_cast<size_t>(min(left_i + c, in_h - 1)) * pitch];
    }
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
        float value = 0.f;
        value += cells[row] >= 1 ? v[row][0] * first_f[row] : 0.f;
#pragma unroll
        for (int c = 1; c < MAXC; ++c) {
            const float plain = value + v[row][c], last = value + v[row][c] * last_f[row];
            value = c < cells[row] - 1 ? plain : (c == cells[row] - 1 ? last : value);
        }
        if (ys[row] < out_h) out[static_cast<size_t>(ys[row]) * pitch + x] = value * normalization;
    }
}

__global__ __launch_bounds__(256) void resample_y_levels_kernel(const float* __restrict__ in_a, float* __restrict__ out_a,
                                                                const float* __restrict__ in_b, float* __restrict__ out_b,
                                                                int in_h, int pitch, ResampleYLevels lv, BatchArg batch)
{
    // which level this block belongs to: one batch of scalar loads and sixteen compares (a search loop is a chain of dependent
    // scalar loads in front of every block), unused entries hold INT_MAX
    int l = 0;
#pragma unroll
    for (int i = 1; i < FLOW2D_RESAMPLE_MAX_LEVELS; ++i) l += static_cast<int>(blockIdx.x) >= lv.first_block[i] ? 1 : 0;
    const int out_w = lv.out_w[l], out_h = lv.out_h[l], rows_per_thread = lv.rows_per_thread[l];
    const float delta = lv.delta[l], normalization = lv.normalization[l];
    const float* __restrict__ in = (batch_plane(batch) ? in_b : in_a) + batch_offset(batch) + lv.col[l];
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch) + static_cast<size_t>(lv.out_row[l]) * pitch;
    const int block = static_cast<int>(blockIdx.x) - lv.first_block[l], blocks_x = lv.blocks_x[l];
    const int block_y = block / blocks_x;
    const int x = (block - block_y * blocks_x) * kBlockX + threadIdx.x;
    if (x >= out_w) return;
    const float* base = in + x;
    // the fine levels: every load of the thread in flight at once (the level's largest cell count is known to the host)
    const int max_cells = lv.max_cells[l];
    if (rows_per_thread == 4 && max_cells <= 2) return resample_y_rows<2, 4>(base, out, x, block_y, out_h, in_h, pitch, delta, normalization);
    if (rows_per_thread == 4 && max_cells <= 4) return resample_y_rows<4, 4>(base, out, x, block_y, out_h, in_h, pitch, delta, normalization);
    if (rows_per_thread == 4 && max_cells <= 9) return resample_y_rows<9, 4>(base, out, x, block_y, out_h, in_h, pitch, delta, normalization);
    if (rows_per_thread == 1 && max_cells <= 9) return resample_y_rows<9, 1>(base, out, x, block_y, out_h, in_h, pitch, delta, normalization);
    for (int row = 0; row < rows_per_thread; ++row) {
        const int y = (block_y * rows_per_thread + row) * kBlockY + threadIdx.y;
        if (y >= out_h) return;
        const float left_f = static_cast<float>(static_cast<unsigned>(y)) * delta;
        const float right_f = static_cast<float>(static_cast<unsigned>(y) + 1u) * delta;
        const int left_i = static_cast<int>(floorf(left_f));
        const int right_i = min(in_h, static_cast<int>(ceilf(right_f)));
        const int cells = right_i - left_i;
        const size_t stride = pitch;
        float value = 0.f;
        // (the body of resample_kernel<false>: cells summed in order, only the first and the last carry a fraction)
        if (cells == 1) {
            value += base[static_cast<size_t>(left_i) * stride] * delta;
        } else if (cells > 1) {
            const int last = left_i + cells - 1;
            value += base[static_cast<size_t>(left_i) * stride] * (static_cast<float>(left_i + 1) - left_f);
            int k = left_i + 1;
            for (; k + 32 <= last; k += 32) {
                float v[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = base[static_cast<size_t>(k + i) * stride];
#pragma unroll
                for (int i = 0; i < 32; ++i) value += v[i];
            }
            for (; k < last; k += 8) {  // eight at a time, the last batch partly (its loads clamped, its additions skipped): no cell waits alone
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = base[static_cast<size_t>(min(k + i, last - 1)) * stride];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float next = value + v[i];
                    value = k + i < last ? next : value;
                }
            }
            value += base[static_cast<size_t>(last) * stride] * (right_f - static_cast<float>(last));
        }
        out[static_cast<size_t>(y) * pitch + x] = value * normalization;
    }
}

}  // namespace

extern "C" {

static int launch_add(flow2d_context* ctx, float* operand_0, const float* operand_1, float* operand_0_b,
                      const float* operand_1_b, size_t width, size_t height, size_t pitch_bytes)
{
    FLOW2D_ENTER(ctx);
    if (!flow2d::plane_args_ok(operand_0, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(operand_1, width, height, pitch_bytes))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const bool pair = operand_0_b || operand_1_b;
    if (pair && (!flow2d::plane_args_ok(operand_0_b, width, height, pitch_bytes) ||
                 !flow2d::plane_args_ok(operand_1_b, width, height, pitch_bytes) || operand_0_b == operand_0 ||
                 operand_0_b == operand_1 || operand_0 == operand_1_b))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const unsigned planes = pair ? 2 : 1;
    dim3 grid(flow2d::div_up(width, kBlockX * 4), flow2d::div_up(height, kBlockY), flow2d::batch_z(ctx, planes));
    add_2d_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(operand_0, operand_1, operand_0_b, operand_1_b,
                                                                   (int)width, (int)height, (int)(pitch_bytes / 4),
                                                                   flow2d::batch_arg(ctx, planes));
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_add_2d(flow2d_context* ctx, float* operand_0, const float* operand_1, size_t width, size_t height,
                  size_t pitch_bytes)
{
    return launch_add(ctx, operand_0, operand_1, nullptr, nullptr, width, height, pitch_bytes);
}

int flow2d_add_2d_pair(flow2d_context* ctx, float* operand_0_a, const float* operand_1_a, float* operand_0_b,
                       const float* operand_1_b, size_t width, size_t height, size_t pitch_bytes)
{
    if (!operand_0_b || !operand_1_b) return FLOW2D_ERR_INVALID_ARGUMENT;
    return launch_add(ctx, operand_0_a, operand_1_a, operand_0_b, operand_1_b, width, height, pitch_bytes);
}

// ---- flow2d_copy_planes: up to 64 independent planes copied by ONE launch (pointer tables in the kernel arguments) ----
namespace {
struct PlaneTable {
    const float* src[FLOW2D_COPY_PLANES_MAX];
    float* dst[FLOW2D_COPY_PLANES_MAX];
};
// one float4 per lane, rows of a plane in blockIdx.y, the plane in blockIdx.z; the row tail that is no whole float4
// goes float by float
__global__ __launch_bounds__(256) void copy_planes_kernel(PlaneTable t, int width, int height, int pitch_floats)
{
    const float* src = t.src[blockIdx.z];
    float* dst = t.dst[blockIdx.z];
    const int quads = width >> 2;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const size_t row = static_cast<size_t>(y) * pitch_floats;
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x)
            reinterpret_cast<float4*>(dst + row)[q] = reinterpret_cast<const float4*>(src + row)[q];
        const int x = (quads << 2) + blockIdx.x * blockDim.x + threadIdx.x;
        if (x < width) dst[row + x] = src[row + x];
    }
}
}  // namespace

int flow2d_copy_planes(flow2d_context* ctx, size_t count, const void* const* src_planes, void* const* dst_planes,
                       size_t pitch_bytes, size_t width, size_t height)
{
    FLOW2D_ENTER(ctx);
    if (count == 0) return FLOW2D_OK;
    if (!src_planes || !dst_planes || count > FLOW2D_COPY_PLANES_MAX) return FLOW2D_ERR_INVALID_ARGUMENT;
    PlaneTable t{};
    for (size_t i = 0; i < count; ++i) {
        if (!flow2d::plane_args_ok(src_planes[i], width, height, pitch_bytes) ||
            !flow2d::plane_args_ok(dst_planes[i], width, height, pitch_bytes) || src_planes[i] == dst_planes[i])
            return FLOW2D_ERR_INVALID_ARGUMENT;
        t.src[i] = static_cast<const float*>(src_planes[i]);
        t.dst[i] = static_cast<float*>(dst_planes[i]);
    }
    const unsigned bx = std::max(1u, std::min(8u, flow2d::div_up(width / 4 + 1, 256)));
    const unsigned by = static_cast<unsigned>(std::min<size_t>(height, 1024));
    copy_planes_kernel<<<dim3(bx, by, static_cast<unsigned>(count)), 256, 0, ctx->stream>>>(t, static_cast<int>(width), static_cast<int>(height),
                                                                                   static_cast<int>(pitch_bytes / 4));
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_gaussian_kernel(float sigma, float* taps, int* out_radius)
{
    if (!taps || !out_radius || !(sigma > 0.f)) return FLOW2D_ERR_INVALID_ARGUMENT;
    const size_t precision = 3;
    const float pixel_size = 1.0f;
    const size_t radius = static_cast<size_t>(precision * sigma / pixel_size);
    if (2 * radius + 1 > 51) return FLOW2D_ERR_UNSUPPORTED;
    const int r = static_cast<int>(radius);
    const double amplitude = 1.0 / (static_cast<double>(sigma) * std::sqrt(2.0 * 3.1415926));
    const double two_sigma_sq = 2.0 * static_cast<double>(sigma) * static_cast<double>(sigma);
    for (int i = -r; i <= r; ++i) {
        const float neg_dist_sq = -(static_cast<float>(i * i) * pixel_size * pixel_size);  // float, as in the reference
        taps[i + r] = static_cast<float>(amplitude * std::exp(static_cast<double>(neg_dist_sq) / two_sigma_sq));
    }
    float sum = 0.0;
    for (int i = 0; i < 2 * r + 1; ++i) sum = sum + taps[i];
    for (int i = 0; i < 2 * r + 1; ++i) taps[i] = taps[i] / sum;
    *out_radius = r;
    return FLOW2D_OK;
}

static int launch_gauss(flow2d_context* ctx, bool rows, float* dst, const float* src, size_t width, size_t height,
                        size_t pitch_bytes, const float* taps, int radius)
{
    FLOW2D_ENTER(ctx);
    if (!flow2d::plane_args_ok(dst, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(src, width, height, pitch_bytes) || !taps || dst == src)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (radius < 0 || 2 * radius + 1 > 51) return FLOW2D_ERR_UNSUPPORTED;
    GaussTaps t;
    for (int i = 0; i < 51; ++i) t.t[i] = i < 2 * radius + 1 ? taps[i] : 0.f;
    for (unsigned b = 0; b < ctx->batch_count; ++b) {  // (not on the pyramid's path: one launch per batch instance)
        const size_t off = b * ctx->batch_stride_floats;
        if (rows)
            gauss_kernel<true><<<grid_for(width, height), dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
                dst + off, src + off, (int)width, (int)height, (int)(pitch_bytes / 4), radius, t);
        else
            gauss_kernel<false><<<grid_for(width, height), dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
                dst + off, src + off, (int)width, (int)height, (int)(pitch_bytes / 4), radius, t);
    }
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_convolution_rows(flow2d_context* ctx, float* dst, const float* src, size_t width, size_t height,
                            size_t pitch_bytes, const float* taps, int radius)
{
    return launch_gauss(ctx, true, dst, src, width, height, pitch_bytes, taps, radius);
}

int flow2d_convolution_columns(flow2d_context* ctx, float* dst, const float* src, size_t width, size_t height,
                               size_t pitch_bytes, const float* taps, int radius)
{
    return launch_gauss(ctx, false, dst, src, width, height, pitch_bytes, taps, radius);
}

int flow2d_gaussian_blur(flow2d_context* ctx, float* dst, const float* src, size_t width, size_t height,
                         size_t pitch_bytes, const float* taps, int radius)
{
    FLOW2D_ENTER(ctx);
    if (!flow2d::plane_args_ok(dst, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(src, width, height, pitch_bytes) || !taps || dst == src)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (radius < 0 || radius > kBlurMaxRadius) return FLOW2D_ERR_UNSUPPORTED;
    GaussTaps t;
    for (int i = 0; i < 51; ++i) t.t[i] = i < 2 * radius + 1 ? taps[i] : 0.f;
    static_assert(kBlurStreamMaxRadius == 6, "one case per streamed radius");
    switch (radius) {
        case 1: launch_gauss_stream<1>(ctx, dst, src, width, height, pitch_bytes, t); break;
        case 2: launch_gauss_stream<2>(ctx, dst, src, width, height, pitch_bytes, t); break;
        case 3: launch_gauss_stream<3>(ctx, dst, src, width, height, pitch_bytes, t); break;
        case 4: launch_gauss_stream<4>(ctx, dst, src, width, height, pitch_bytes, t); break;
        case 5: launch_gauss_stream<5>(ctx, dst, src, width, height, pitch_bytes, t); break;
        case 6: launch_gauss_stream<6>(ctx, dst, src, width, height, pitch_bytes, t); break;
        default: {
            const dim3 grid(flow2d::div_up(width, kBlockX), flow2d::div_up(height, kBlurTile));
            for (unsigned b = 0; b < ctx->batch_count; ++b)  // (large radii only: one launch per batch instance)
                gauss_fused_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
                    dst + b * ctx->batch_stride_floats, src + b * ctx->batch_stride_floats, (int)width, (int)height,
                    (int)(pitch_bytes / 4), radius, t);
        }
    }
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

static int launch_resample(flow2d_context* ctx, bool along_x, const float* input, float* output, const float* input_b,
                           float* output_b, size_t out_width, size_t out_height, size_t in_extent, size_t pitch_bytes)
{
    FLOW2D_ENTER(ctx);
    const size_t in_w = along_x ? in_extent : out_width;
    const size_t in_h = along_x ? out_height : in_extent;
    if (!flow2d::plane_args_ok(input, in_w, in_h, pitch_bytes) ||
        !flow2d::plane_args_ok(output, out_width, out_height, pitch_bytes) || input == output)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const bool pair = input_b || output_b;
    if (pair && (!flow2d::plane_args_ok(input_b, in_w, in_h, pitch_bytes) ||
                 !flow2d::plane_args_ok(output_b, out_width, out_height, pitch_bytes) || input_b == output_b ||
                 output_b == output || output_b == input || output == input_b))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const unsigned planes = pair ? 2 : 1;
    const unsigned z = flow2d::batch_z(ctx, planes);
    const BatchArg batch = flow2d::batch_arg(ctx, planes);
    if (along_x && in_extent >= 2 * out_width)  // strong down-sampling: coalesced staging through LDS
        resample_x_lds_kernel<<<dim3(flow2d::div_up(out_width, 64), flow2d::div_up(out_height, 4), z), 256, 0,
                                ctx->stream>>>(input, output, input_b, output_b, (int)out_width, (int)out_height,
                                               (int)in_extent, (int)(pitch_bytes / 4), batch);
    else {
        const int rows_per_thread = out_width * out_height >= (size_t)512 * 512 ? 4 : 1;
        dim3 grid = grid_for(out_width, flow2d::div_up(out_height, rows_per_thread));
        grid.z = z;
        const float out_n = static_cast<float>(along_x ? out_width : out_height), in_n = static_cast<float>(in_extent);
        const float delta = in_n / out_n, normalization = out_n / in_n;
        if (along_x)
            resample_kernel<true><<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
                input, output, input_b, output_b, (int)out_width, (int)out_height, (int)in_extent, (int)(pitch_bytes / 4),
                delta, normalization, rows_per_thread, batch);
        else
            resample_kernel<false><<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
                input, output, input_b, output_b, (int)out_width, (int)out_height, (int)in_extent, (int)(pitch_bytes / 4),
                delta, normalization, rows_per_thread, batch);
    }
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_resample_x_levels(flow2d_context* ctx, const float* input_a, float* packed_a, const float* input_b,
                             float* packed_b, size_t in_width, size_t height, size_t pitch_bytes, size_t level_count,
                             const size_t* out_widths, const size_t* column_offsets)
{
    FLOW2D_ENTER(ctx);
    if (!out_widths || !column_offsets || level_count == 0) return FLOW2D_ERR_INVALID_ARGUMENT;
    if (level_count > FLOW2D_RESAMPLE_MAX_LEVELS || in_width > 15360) return FLOW2D_ERR_UNSUPPORTED;
    const bool pair = input_b || packed_b;
    if (!flow2d::plane_args_ok(input_a, in_width, height, pitch_bytes) || !packed_a || input_a == packed_a ||
        (reinterpret_cast<uintptr_t>(packed_a) % 16) != 0)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (pair && (!flow2d::plane_args_ok(input_b, in_width, height, pitch_bytes) || !packed_b || input_b == packed_b ||
                 packed_b == packed_a || packed_b == input_a || packed_a == input_b ||
                 (reinterpret_cast<uintptr_t>(packed_b) % 16) != 0))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    ResampleLevels lv{};
    lv.count = static_cast<int>(level_count);
    const size_t pitch = pitch_bytes / 4;
    for (size_t l = 0; l < level_count; ++l) {
        if (out_widths[l] == 0 || out_widths[l] >= (1u << 30) || column_offsets[l] % 4 != 0 ||
            column_offsets[l] + out_widths[l] > pitch)
            return FLOW2D_ERR_INVALID_ARGUMENT;
        for (size_t m = 0; m < l; ++m)  // segments must not overlap
            if (column_offsets[l] < column_offsets[m] + out_widths[m] && column_offsets[m] < column_offsets[l] + out_widths[l])
                return FLOW2D_ERR_INVALID_ARGUMENT;
        lv.out_w[l] = static_cast<int>(out_widths[l]);
        lv.col[l] = static_cast<int>(column_offsets[l]);
    }
    const unsigned planes = pair ? 2 : 1;
    {   // every ratio a power of two, the frame a whole number of 32-cell threads of one workgroup: the register kernel
        bool pow2 = in_width % kPow2Cells == 0 && in_width <= 256 * kPow2Cells && (pitch % 4) == 0;
        int deep_total = 0;
        for (size_t l = 0; l < level_count && pow2; ++l) {
            const size_t ratio = in_width / out_widths[l];
            pow2 = ratio >= 2 && ratio * out_widths[l] == in_width && (ratio & (ratio - 1)) == 0;
            if (ratio > kPow2Cells) deep_total += static_cast<int>(out_widths[l]);
        }
        if (pow2) {
            const size_t words = in_width + 4 * (in_width / 32 + in_width / 256) + 8;
            resample_x_levels_pow2_kernel<<<dim3(static_cast<unsigned>(height), 1, flow2d::batch_z(ctx, planes)), 256,
                                            deep_total ? words * sizeof(float) : 0, ctx->stream>>>(
                input_a, packed_a, input_b, packed_b, static_cast<int>(in_width), static_cast<int>(pitch), lv, deep_total,
                flow2d::batch_arg(ctx, planes));
            FLOW2D_CHECK_LAUNCH();
            return FLOW2D_OK;
        }
    }
    const size_t lds_bytes = (in_width + in_width / 32 + 1) * sizeof(float);
    resample_x_levels_kernel<<<dim3(static_cast<unsigned>(height), 1, flow2d::batch_z(ctx, planes)), 256, lds_bytes,
                               ctx->stream>>>(input_a, packed_a, input_b, packed_b, static_cast<int>(in_width),
                                              static_cast<int>(pitch), lv, flow2d::batch_arg(ctx, planes));
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_resample_y_levels(flow2d_context* ctx, const float* packed_a, float* output_a, const float* packed_b, float* output_b,
                             size_t in_height, size_t pitch_bytes, size_t level_count, const size_t* out_widths,
                             const size_t* out_heights, const size_t* column_offsets, const size_t* output_rows)
{
    FLOW2D_ENTER(ctx);
    if (!out_widths || !out_heights || !column_offsets || !output_rows || level_count == 0 || !packed_a || !output_a ||
        packed_a == output_a || pitch_bytes == 0 || pitch_bytes % 16 != 0 || in_height == 0)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (level_count > FLOW2D_RESAMPLE_MAX_LEVELS) return FLOW2D_ERR_UNSUPPORTED;
    const bool pair = packed_b || output_b;
    if (pair && (!packed_b || !output_b || packed_b == output_b || output_b == output_a || output_b == packed_a || output_a == packed_b))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const size_t pitch = pitch_bytes / 4;
    ResampleYLevels lv{};
    lv.count = static_cast<int>(level_count);
    long blocks = 0;
    for (size_t l = 0; l <= FLOW2D_RESAMPLE_MAX_LEVELS; ++l) lv.first_block[l] = 0x7fffffff;
    for (size_t l = 0; l < level_count; ++l) {
        if (out_widths[l] == 0 || out_heights[l] == 0 || out_heights[l] >= (1u << 30) || column_offsets[l] + out_widths[l] > pitch ||
            output_rows[l] >= (1u << 30))
            return FLOW2D_ERR_INVALID_ARGUMENT;
        for (size_t m = 0; m < l; ++m)  // the levels' output regions must not overlap
            if (output_rows[l] < output_rows[m] + out_heights[m] && output_rows[m] < output_rows[l] + out_heights[l])
                return FLOW2D_ERR_INVALID_ARGUMENT;
        lv.out_w[l] = static_cast<int>(out_widths[l]), lv.out_h[l] = static_cast<int>(out_heights[l]);
        lv.col[l] = static_cast<int>(column_offsets[l]), lv.out_row[l] = static_cast<int>(output_rows[l]);
        // delta = in_n / (float) out_n, normalization = out_n / (float) in_n (resample_2d.cu:88-89), as launch_resample evaluates them
        const float out_n = static_cast<float>(out_heights[l]), in_n = static_cast<float>(in_height);
        lv.delta[l] = in_n / out_n, lv.normalization[l] = out_n / in_n;
        lv.rows_per_thread[l] = out_widths[l] * out_heights[l] >= (size_t)512 * 512 ? 4 : 1;
        // cells of output g = ceil((g + 1) delta) - floor(g delta) in float arithmetic: at most ceil(delta) + 1 (+1 for the roundings)
        lv.max_cells[l] = lv.delta[l] == std::floor(lv.delta[l]) && lv.delta[l] < 1e6f ? static_cast<int>(lv.delta[l])
                                                                                      : static_cast<int>(std::min(std::ceil(lv.delta[l]) + 2.f, 1e9f));
        lv.first_block[l] = static_cast<int>(blocks);
        lv.blocks_x[l] = static_cast<int>(flow2d::div_up(out_widths[l], kBlockX));
        blocks += static_cast<long>(lv.blocks_x[l]) * static_cast<long>(flow2d::div_up(flow2d::div_up(out_heights[l], lv.rows_per_thread[l]), kBlockY));
        if (blocks >= 0x7fffffffl) return FLOW2D_ERR_UNSUPPORTED;
    }
    const unsigned planes = pair ? 2 : 1;
    resample_y_levels_kernel<<<dim3(static_cast<unsigned>(blocks), 1, flow2d::batch_z(ctx, planes)), dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
        packed_a, output_a, packed_b, output_b, static_cast<int>(in_height), static_cast<int>(pitch), lv, flow2d::batch_arg(ctx, planes));
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_resample_x(flow2d_context* ctx, const float* input, float* output, size_t out_width, size_t out_height,
                      size_t in_width, size_t pitch_bytes)
{
    return launch_resample(ctx, true, input, output, nullptr, nullptr, out_width, out_height, in_width, pitch_bytes);
}

int flow2d_resample_x_pair(flow2d_context* ctx, const float* input_a, float* output_a, const float* input_b,
                           float* output_b, size_t out_width, size_t out_height, size_t in_width, size_t pitch_bytes)
{
    if (!input_b || !output_b) return FLOW2D_ERR_INVALID_ARGUMENT;
    return launch_resample(ctx, true, input_a, output_a, input_b, output_b, out_width, out_height, in_width, pitch_bytes);
}

int flow2d_resample_y(flow2d_context* ctx, const float* input, float* output, size_t out_width, size_t out_height,
                      size_t in_height, size_t pitch_bytes)
{
    return launch_resample(ctx, false, input, output, nullptr, nullptr, out_width, out_height, in_height, pitch_bytes);
}

int flow2d_resample_y_pair(flow2d_context* ctx, const float* input_a, float* output_a, const float* input_b,
                           float* output_b, size_t out_width, size_t out_height, size_t in_height, size_t pitch_bytes)
{
    if (!input_b || !output_b) return FLOW2D_ERR_INVALID_ARGUMENT;
    return launch_resample(ctx, false, input_a, output_a, input_b, output_b, out_width, out_height, in_height,
                           pitch_bytes);
}

int flow2d_resample_xy_pair(flow2d_context* ctx, const float* input_a, float* output_a, const float* input_b,
                            float* output_b, size_t in_width, size_t in_height, size_t out_width, size_t out_height,
                            size_t pitch_bytes)
{
    FLOW2D_ENTER(ctx);
    if (!flow2d::plane_args_ok(input_a, in_width, in_height, pitch_bytes) ||
        !flow2d::plane_args_ok(output_a, out_width, out_height, pitch_bytes) || input_a == output_a)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const bool pair = input_b || output_b;
    if (pair && (!flow2d::plane_args_ok(input_b, in_width, in_height, pitch_bytes) ||
                 !flow2d::plane_args_ok(output_b, out_width, out_height, pitch_bytes) || input_b == output_b ||
                 output_b == output_a || output_b == input_a || output_a == input_b))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const unsigned planes = pair ? 2 : 1;
    dim3 grid = grid_for(out_width, flow2d::div_up(out_height, kResampleXYRows));
    grid.z = flow2d::batch_z(ctx, planes);
    const ResampleXY k{static_cast<float>(in_width) / static_cast<float>(out_width),
                       static_cast<float>(out_width) / static_cast<float>(in_width),
                       static_cast<float>(in_height) / static_cast<float>(out_height),
                       static_cast<float>(out_height) / static_cast<float>(in_height)};
    resample_xy_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
        input_a, output_a, input_b, output_b, (int)out_width, (int)out_height, (int)in_width, (int)in_height,
        (int)(pitch_bytes / 4), k, flow2d::batch_arg(ctx, planes));
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

int flow2d_registration_2d(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                           const float* flow_v, size_t width, size_t height, size_t pitch_bytes, float hx, float hy,
                           float* output)
{
    FLOW2D_ENTER(ctx);
    if (!flow2d::plane_args_ok(frame_0, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(frame_1, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(flow_u, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(flow_v, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(output, width, height, pitch_bytes) || output == frame_1 || output == frame_0 ||
        !(hx > 0.f) || !(hy > 0.f))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    dim3 grid = grid_for(width, flow2d::div_up(height, kRegistrationRows));
    grid.z = flow2d::batch_z(ctx, 1);
    registration_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
        frame_0, frame_1, flow_u, flow_v, (int)width, (int)height, (int)(pitch_bytes / 4), 1.f / hx, 1.f / hy, output,
        flow2d::batch_arg(ctx, 1));
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

// flow_u / flow_v of the previous level (in_width x in_height) resampled to width x height into out_u / out_v -- the bits of
// flow2d_resample_xy_pair -- and frame_1 warped by them into `output` -- the bits of flow2d_registration_2d: one launch.
// flow_u = flow_v = NULL with in_width = in_height = 0 (the coarsest level): out_u = out_v = 0 over width x height, warped by that.
int flow2d_upsample_registration_2d(flow2d_context* ctx, const float* flow_u, const float* flow_v, size_t in_width, size_t in_height,
                                    float* out_u, float* out_v, const float* frame_0, const float* frame_1, size_t width, size_t height,
                                    size_t pitch_bytes, float hx, float hy, float* output)
{
    FLOW2D_ENTER(ctx);
    const bool zero = !flow_u && !flow_v && in_width == 0 && in_height == 0;  // the coarsest level: no previous flow
    if ((!zero && (!flow2d::plane_args_ok(flow_u, in_width, in_height, pitch_bytes) || !flow2d::plane_args_ok(flow_v, in_width, in_height, pitch_bytes))) ||
        !flow2d::plane_args_ok(out_u, width, height, pitch_bytes) || !flow2d::plane_args_ok(out_v, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(frame_0, width, height, pitch_bytes) || !flow2d::plane_args_ok(frame_1, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(output, width, height, pitch_bytes) || !(hx > 0.f) || !(hy > 0.f))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const float* reads[] = {flow_u, flow_v, frame_0, frame_1};
    const float* writes[] = {out_u, out_v, output};
    for (int i = 0; i < 3; ++i) {
        for (const float* r : reads)
            if (writes[i] == r) return FLOW2D_ERR_INVALID_ARGUMENT;
        for (int j = i + 1; j < 3; ++j)
            if (writes[i] == writes[j]) return FLOW2D_ERR_INVALID_ARGUMENT;
    }
    dim3 grid = grid_for(width, flow2d::div_up(height, kRegistrationRows));
    grid.z = flow2d::batch_z(ctx, 1);
    const ResampleXY k = zero ? ResampleXY{1.f, 1.f, 1.f, 1.f}
                              : ResampleXY{static_cast<float>(in_width) / static_cast<float>(width),
                                           static_cast<float>(width) / static_cast<float>(in_width),
                                           static_cast<float>(in_height) / static_cast<float>(height),
                                           static_cast<float>(height) / static_cast<float>(in_height)};
    auto launch = [&](auto mode) {
        upsample_registration_kernel<decltype(mode)::value><<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
            flow_u, flow_v, out_u, out_v, frame_0, frame_1, output, (int)width, (int)height, (int)in_width, (int)in_height,
            (int)(pitch_bytes / 4), k, 1.f / hx, 1.f / hy, flow2d::batch_arg(ctx, 1));
    };
    if (zero)
        launch(std::integral_constant<int, kUpsampleZero>{});
    else if (width == 2 * in_width && height == 2 * in_height)
        launch(std::integral_constant<int, kUpsampleDouble>{});
    else
        launch(std::integral_constant<int, kUpsampleGeneral>{});
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}


}  // extern "C"
