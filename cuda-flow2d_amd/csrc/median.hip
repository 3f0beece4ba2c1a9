// r x r median of a flow plane (r in {3,5,7}) for gfx950.
// Restates src/kernels/median_2d.cu:87-299 of the reference: window [x-r/2, x+r/2]^2 with
// reflect-without-repeat borders, output = element r*r/2 of the ascending window.  The reference
// sorts the window with a per-thread insertion sort; selection is order-free, so any exact
// selection network gives the same value -- except where the order of the reference's sort shows:
// a window holding a NaN (`temp < window[j]` is false for it, so NaNs never move and cut the window into
// separately sorted runs) or zeros of both signs (-0 < +0 is false either way: they keep their gather order).
// Windows holding a NaN or a -0 are therefore recomputed by exact_median(), which evaluates what that
// insertion sort leaves at position r*r/2 (median_2d.cu:52-63,281-296); everything else takes the networks.
//
// Generic kernel (any r): the window lives in registers and goes through a Batcher odd-even merge network
// whose comparator list is generated at compile time; comparators that cannot influence the median are
// removed by the compiler's dead-code elimination.  One wave = 64 consecutive pixels of a row.
//
// r = 5 (the reference's default) has a streaming kernel: a wave owns 64 columns and walks down the image.
// Per image row it loads ONE value per lane, takes the four horizontal neighbours from the adjacent lanes
// (DPP), sorts the 5-tuple (9 comparators) and keeps the sorted tuples of six consecutive rows in
// registers.  Two vertically adjacent medians share four of their five rows, so each step merges those four
// sorted tuples once and finishes both medians from there (median5_pair_network.inc: an 81-comparator merge network,
// generated and exhaustively verified by tools/gen_median_network.py, lowered by tools/median_select3.py to one-result
// instructions -- v_min / v_max and the three-input v_min3 / v_max3 / v_med3 -- and shortened under the same exhaustive
// check: about 80 instructions for two pixels instead of the 123 two-input halves the comparators leave alive, the
// five-value sorter of a row 13 instead of 18).  Per pixel: 1 load instead of 25.
#include <array>
#include <utility>

#include "common.hpp"

namespace {

constexpr int kBlockX = 64;
constexpr int kBlockY = 4;

constexpr int next_pow2(int n)
{
    int p = 1;
    while (p < n) p *= 2;
    return p;
}

struct Comparator {
    int a, b;
};

// Batcher's odd-even merge sort on P = next_pow2(N) wires; comparators touching a wire >= N are
// dropped (those wires would hold +inf and never move).
template <int N>
constexpr int network_size()
{
    constexpr int P = next_pow2(N);
    int count = 0;
    for (int p = 1; p < P; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < P - j - k - 1 ? k - 1 : P - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < N) ++count;
    return count;
}

template <int N>
constexpr std::array<Comparator, network_size<N>()> make_network()
{
    constexpr int P = next_pow2(N);
    std::array<Comparator, network_size<N>()> net{};
    int count = 0;
    for (int p = 1; p < P; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < P - j - k - 1 ? k - 1 : P - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < N) {
                        net[count].a = i + j;
                        net[count].b = i + j + k;
                        ++count;
                    }
    return net;
}

template <int N, size_t... I>
__device__ __forceinline__ void run_network(float (&v)[N], std::index_sequence<I...>)
{
    constexpr auto net = make_network<N>();
    (
        [&] {
            const float lo = fminf(v[net[I].a], v[net[I].b]);
            const float hi = fmaxf(v[net[I].a], v[net[I].b]);
            v[net[I].a] = lo;
            v[net[I].b] = hi;
        }(),
        ...);
}

// What the filter reads: a plane, or (ADD) the sum of two planes formed on the fly -- the pyramid's `u += du` followed
// by the median of u (optical_flow_2d.cpp:414-446) in one pass: add_2d's sum is a single rounded addition
// (add_2d.cu:33-46), so the median of the sums is the same value, and the plane of sums is neither written nor re-read.
template <bool ADD>
struct Source {
    const float* __restrict__ in;
    const float* __restrict__ add;
    __device__ __forceinline__ float operator[](size_t i) const
    {
        if (ADD) return in[i] + add[i];
        return in[i];
    }
};

// The streaming kernels address a plane as base pointer (wave-uniform: a scalar register pair) + one 32-bit per-lane BYTE offset
// (global_load / global_store ... saddr), like the strip kernel of the solver: a 64-bit address per access was two to three
// vector instructions each, a quarter of the kernel.  Planes of 4 GiB and more take the generic kernel (launch_median).
__device__ __forceinline__ float plane_load(const float* plane, unsigned byte_offset)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(plane) + static_cast<size_t>(byte_offset));
}
__device__ __forceinline__ void plane_store(float* plane, unsigned byte_offset, float value)
{
    *reinterpret_cast<float*>(reinterpret_cast<char*>(plane) + static_cast<size_t>(byte_offset)) = value;
}

// v_cmp_class mask: signalling NaN, quiet NaN, negative zero
constexpr int kSpecialClass = 0x1 | 0x2 | 0x20;
__device__ __forceinline__ bool is_special(float v) { return __builtin_amdgcn_classf(v, kSpecialClass); }

// What the reference's insertion sort (median_2d.cu:52-63) leaves at index r*r/2 of the window gathered in its
// order (descending offsets, :281-287), for ANY contents.  With `temp < window[j]` as the only comparison
//  * a NaN is never moved and never passed: the NaNs stay at their gather positions and the values between two
//    of them are sorted among themselves;
//  * equal values (and -0 / +0, which compare equal) keep their gather order (the sort is stable).
// So the result is the NaN sitting at r*r/2, or the element of stable rank r*r/2 - a in the run [a, b) of gather
// positions between the nearest NaNs on either side.  O(n^2) loads (L1 hits); only rare windows come here.
template <int R, bool ADD>
__device__ __noinline__ float exact_median(const Source<ADD> in, int x, int y, int w, int h, int pitch)
{
    constexpr int N = R * R, R2 = R / 2, M = N / 2;
    auto at = [&](int k) {
        const int iy = k / R, ix = k - iy * R;
        return in[static_cast<size_t>(mirror_index(y - iy + R2, h)) * pitch + mirror_index(x - ix + R2, w)];
    };
    const float vm = at(M);
    if (vm != vm) return vm;
    int a = 0, b = N;
    for (int k = 0; k < N; ++k) {
        const float v = at(k);
        if (v != v) {
            if (k < M) a = k + 1;
            else if (b == N) b = k;
        }
    }
    const int target = M - a;
    for (int i = a; i < b; ++i) {
        const float vi = at(i);
        int rank = 0;
        for (int j = a; j < b; ++j) {
            const float vj = at(j);
            rank += (vj < vi || (vj == vi && j < i)) ? 1 : 0;
        }
        if (rank == target) return vi;
    }
    return vm;  // not reached: the ranks of a run are a permutation of 0 .. b-a-1
}

// (grid.z = 2 filters a second, independent plane in the same launch: the flow's u and v)
template <int R, bool ADD>
__global__ __launch_bounds__(256) void median_kernel(const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                     const float* __restrict__ add_a, const float* __restrict__ add_b, int w,
                                                     int h, int pitch, float* __restrict__ out_a,
                                                     float* __restrict__ out_b, BatchArg batch)
{
    const Source<ADD> in{(batch_plane(batch) ? in_b : in_a) + batch_offset(batch),
                         ADD ? (batch_plane(batch) ? add_b : add_a) + batch_offset(batch) : nullptr};
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    constexpr int R2 = R / 2;
    constexpr int N = R * R;
    const int x = blockIdx.x * kBlockX + threadIdx.x;
    const int y = blockIdx.y * kBlockY + threadIdx.y;
    if (x >= w || y >= h) return;
    int xs[R];
#pragma unroll
    for (int i = 0; i < R; ++i) xs[i] = mirror_index(x + i - R2, w);
    float v[N];
    bool special = false;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const size_t row = static_cast<size_t>(mirror_index(y + j - R2, h)) * pitch;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[j * R + i] = in[row + xs[i]];
            special |= is_special(v[j * R + i]);
        }
    }
    run_network<N>(v, std::make_index_sequence<network_size<N>()>{});
    float result = v[N / 2];
    if (special) result = exact_median<R, ADD>(in, x, y, w, h, pitch);
    out[static_cast<size_t>(y) * pitch + x] = result;
}

// ---- r = 5, streaming -----------------------------------------------------------------------------------------
// One-result selection programs (tools/median_select3.py): node (inputs + i) = kind(node a, node b, node c)
struct SelectOp {
    int kind, a, b, c;  // kind 0 min, 1 max (two inputs), 2 min3, 3 max3, 4 med3
};
template <int KIND>
__device__ __forceinline__ float select_op(float a, float b, float c)
{
    // (the windows that reach these programs hold no NaN and no -0 -- those are redone by exact_median -- so minimum, maximum and
    //  median of three are selections of one of their operands; fminf(fminf(a, b), c) with a single-use inner result is v_min3_f32)
    if constexpr (KIND == 0) return fminf(a, b);
    else if constexpr (KIND == 1) return fmaxf(a, b);
    else if constexpr (KIND == 2) return fminf(fminf(a, b), c);
    else if constexpr (KIND == 3) return fmaxf(fmaxf(a, b), c);
    else return __builtin_amdgcn_fmed3f(a, b, c);
}
#include "median5_pair_network.inc"

// runs PROGRAM on n[0 .. INPUTS): afterwards n[INPUTS + i] holds node i's value (everything stays in registers: the indices are
// compile-time constants, values nobody reads are never computed)
template <int INPUTS, int OPS, const SelectOp (&PROGRAM)[OPS], size_t... I>
__device__ __forceinline__ void run_select_program(float (&n)[INPUTS + OPS], std::index_sequence<I...>)
{
    ((n[INPUTS + I] = select_op<PROGRAM[I].kind>(n[PROGRAM[I].a], n[PROGRAM[I].b], n[PROGRAM[I].c])), ...);
}

__device__ __forceinline__ void sort5(float (&t)[5])
{
    float n[5 + kSort5Ops];
#pragma unroll
    for (int i = 0; i < 5; ++i) n[i] = t[i];
    run_select_program<5, kSort5Ops, kSort5Program>(n, std::make_index_sequence<kSort5Ops>{});
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i] = n[kSort5Out[i]];
}

__device__ __forceinline__ float lane_left(float v)  // lane i <- lane i-1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v)  // lane i <- lane i+1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

constexpr int kStreamValid = 60;  // lanes 2..61 of a wave have both neighbours on either side

// the values a lane loads for one image row: its own column, or (EDGE) the five mirrored columns -- RAW, as they come
// from memory: the sum with the addend and the NaN / -0 check happen where the row is consumed (sorted_tuple), two
// steps later, so that nothing waits for a load at the place it is issued
template <bool EDGE, bool ADD>
struct RowLoad {
    float v[EDGE ? 5 : 1];
    float a[ADD ? (EDGE ? 5 : 1) : 1];
};

template <bool EDGE, bool ADD>
__device__ __forceinline__ RowLoad<EDGE, ADD> load_row(const Source<ADD> in, int row, int h, int pitch, int xc,
                                                       const int (&xm)[5])
{
    const unsigned line = static_cast<unsigned>(min(max(mirror_index(row, h), 0), h - 1)) * static_cast<unsigned>(pitch);
    RowLoad<EDGE, ADD> r;
    r.a[0] = 0.f;
#pragma unroll
    for (int i = 0; i < (EDGE ? 5 : 1); ++i) {
        const unsigned at = (line + static_cast<unsigned>(EDGE ? xm[i] : xc)) * 4u;
        r.v[i] = plane_load(in.in, at);
        if (ADD) r.a[i] = plane_load(in.add, at);
    }
    return r;
}

// sorted 5-tuple (x-2 .. x+2) of the loaded row.  `special` collects whether any value this lane loaded is a NaN or a
// -0 (one v_cmp_class per value; the strip is re-checked per pixel only when some lane of the wave saw one)
template <bool EDGE, bool ADD>
__device__ __forceinline__ void sorted_tuple(const RowLoad<EDGE, ADD>& r, float (&t)[5], bool& special)
{
    if (EDGE) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            t[i] = ADD ? r.v[i] + r.a[i] : r.v[i];
            special |= is_special(t[i]);
        }
    } else {
        const float c = ADD ? r.v[0] + r.a[0] : r.v[0];
        special |= is_special(c);
        const float l1 = lane_left(c), r1 = lane_right(c);
        t[0] = lane_left(l1);
        t[1] = l1;
        t[2] = c;
        t[3] = r1;
        t[4] = lane_right(r1);
    }
    sort5(t);
}

// One wave, one strip of 64 columns, rows [y0, y1).  Step I of three (the ring of six row slots advances by
// two rows per step, so three steps bring every slot back to its place and all indices are constants).
template <bool EDGE, int I, bool ADD>
__device__ __forceinline__ void median5_step(float (&ring)[6][5], RowLoad<EDGE, ADD> (&next)[4], const Source<ADD> in,
                                             float* __restrict__ out, int ya, int y1, int h, int pitch, int x, int xc,
                                             const int (&xm)[5], bool lane_stores, bool& special)
{
    // rows ya+2 and ya+3 were requested two steps ago (a step is shorter than the way to memory and back, and with one
    // step of distance the wait for the rows also waited for the previous step's stores); request rows ya+6, ya+7
    sorted_tuple<EDGE, ADD>(next[0], ring[(2 * I + 4) % 6], special);
    sorted_tuple<EDGE, ADD>(next[1], ring[(2 * I + 5) % 6], special);
    next[0] = next[2];
    next[1] = next[3];
    next[2] = load_row<EDGE, ADD>(in, ya + 6, h, pitch, xc, xm);
    next[3] = load_row<EDGE, ADD>(in, ya + 7, h, pitch, xc, xm);
    float v[kMedianPairInputs + kMedianPairOps];
#pragma unroll
    for (int g = 0; g < 6; ++g)
#pragma unroll
        for (int e = 0; e < 5; ++e) v[5 * g + e] = ring[(2 * I + g) % 6][e];
    run_select_program<kMedianPairInputs, kMedianPairOps, kMedianPairProgram>(v, std::make_index_sequence<kMedianPairOps>{});
    if (lane_stores) {
        const unsigned at = (static_cast<unsigned>(ya) * static_cast<unsigned>(pitch) + static_cast<unsigned>(x)) * 4u;
        plane_store(out, at, v[kMedianPairOutA]);
        if (ya + 1 < y1) plane_store(out, at + static_cast<unsigned>(pitch) * 4u, v[kMedianPairOutB]);
    }
}

template <bool EDGE, bool ADD>
__device__ __forceinline__ void median5_strip(const Source<ADD> in, float* __restrict__ out, int w, int h, int pitch,
                                              int x, int y0, int y1, bool lane_stores)
{
    const int xc = min(max(x, 0), w - 1);
    int xm[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) xm[i] = min(max(mirror_index(x + i - 2, w), 0), w - 1);
    float ring[6][5];
    bool special = false;
    // rows y0-2 .. y0+1 fill slots 0..3; rows y0+2, y0+3 are the first pair in flight
    {
        RowLoad<EDGE, ADD> first[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) first[g] = load_row<EDGE, ADD>(in, y0 - 2 + g, h, pitch, xc, xm);
#pragma unroll
        for (int g = 0; g < 4; ++g) sorted_tuple<EDGE, ADD>(first[g], ring[g], special);
#pragma unroll
        for (int e = 0; e < 5; ++e) ring[4][e] = ring[5][e] = 0.f;
    }
    RowLoad<EDGE, ADD> next[4] = {load_row<EDGE, ADD>(in, y0 + 2, h, pitch, xc, xm), load_row<EDGE, ADD>(in, y0 + 3, h, pitch, xc, xm),
                                  load_row<EDGE, ADD>(in, y0 + 4, h, pitch, xc, xm), load_row<EDGE, ADD>(in, y0 + 5, h, pitch, xc, xm)};
    for (int ya = y0; ya < y1; ya += 6) {
        median5_step<EDGE, 0, ADD>(ring, next, in, out, ya, y1, h, pitch, x, xc, xm, lane_stores, special);
        if (ya + 2 >= y1) break;
        median5_step<EDGE, 1, ADD>(ring, next, in, out, ya + 2, y1, h, pitch, x, xc, xm, lane_stores, special);
        if (ya + 4 >= y1) break;
        median5_step<EDGE, 2, ADD>(ring, next, in, out, ya + 4, y1, h, pitch, x, xc, xm, lane_stores, special);
    }
    // Some lane of this wave loaded a NaN or a -0: every window of the strip is a subset of what the wave loaded,
    // so look at each stored pixel's window again and redo those that hold one the way the reference's sort would.
    if (__builtin_amdgcn_ballot_w64(special) != 0 && lane_stores) {
        for (int y = y0; y < y1; ++y) {
            bool hit = false;
            for (int j = -2; j <= 2; ++j) {
                const size_t line = static_cast<size_t>(mirror_index(y + j, h)) * pitch;
                for (int i = -2; i <= 2; ++i) hit |= is_special(in[line + mirror_index(x + i, w)]);
            }
            if (hit) out[static_cast<size_t>(y) * pitch + x] = exact_median<5, ADD>(in, x, y, w, h, pitch);
        }
    }
}

template <bool ADD>
__global__ __launch_bounds__(256) void median5_stream_kernel(const float* __restrict__ in_a,
                                                             const float* __restrict__ in_b,
                                                             const float* __restrict__ add_a,
                                                             const float* __restrict__ add_b, int w, int h, int pitch,
                                                             int rows_per_strip, float* __restrict__ out_a,
                                                             float* __restrict__ out_b, BatchArg batch)
{
    const Source<ADD> in{(batch_plane(batch) ? in_b : in_a) + batch_offset(batch),
                         ADD ? (batch_plane(batch) ? add_b : add_a) + batch_offset(batch) : nullptr};
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x_first = strip * kStreamValid - 2;
    if (strip * kStreamValid >= w) return;
    const int x = x_first + lane;
    const int y0 = blockIdx.y * rows_per_strip;
    const int y1 = min(y0 + rows_per_strip, h);
    const bool lane_stores = lane >= 2 && lane < 62 && x < w;
    // a wave whose 64 columns lie inside the image takes its horizontal neighbours from the adjacent lanes;
    // at the left and right image border the mirrored columns are loaded instead
    const bool edge = x_first < 0 || x_first + 63 > w - 1;
    if (__builtin_amdgcn_readfirstlane(edge))
        median5_strip<true, ADD>(in, out, w, h, pitch, x, y0, y1, lane_stores);
    else
        median5_strip<false, ADD>(in, out, w, h, pitch, x, y0, y1, lane_stores);
}

// ---- r = 3, streaming (round 6) -----------------------------------------------------------------------------------
// The same scheme one size down: a lane loads ONE value per image row, takes its two horizontal neighbours from the adjacent
// lanes (DPP) and sorts the triple with three instructions (v_min3 / v_med3 / v_max3 of the same three values); with the
// triples of three rows sorted, the median of the nine is med3(max of the minima, median of the medians, min of the maxima) --
// four instructions per pixel.  A step finishes two vertically adjacent pixels from four rows.  The generic kernel gathered nine
// values per pixel: 66 us per 4096^2 plane, slower than the 5 x 5 filter.
constexpr int kStream3Valid = 62;  // lanes 1..62 of a wave have a neighbour on either side

template <bool EDGE, bool ADD>
struct Row3Load {
    float v[EDGE ? 3 : 1];
    float a[ADD ? (EDGE ? 3 : 1) : 1];
};

template <bool EDGE, bool ADD>
__device__ __forceinline__ Row3Load<EDGE, ADD> load_row3(const Source<ADD> in, int row, int h, int pitch, int xc, const int (&xm)[3])
{
    const unsigned line = static_cast<unsigned>(min(max(mirror_index(row, h), 0), h - 1)) * static_cast<unsigned>(pitch);
    Row3Load<EDGE, ADD> r;
    r.a[0] = 0.f;
#pragma unroll
    for (int i = 0; i < (EDGE ? 3 : 1); ++i) {
        const unsigned at = (line + static_cast<unsigned>(EDGE ? xm[i] : xc)) * 4u;
        r.v[i] = plane_load(in.in, at);
        if (ADD) r.a[i] = plane_load(in.add, at);
    }
    return r;
}

// (minimum, median, maximum) of the row's three values x-1, x, x+1
template <bool EDGE, bool ADD>
__device__ __forceinline__ void sorted_triple(const Row3Load<EDGE, ADD>& r, float (&t)[3], bool& special)
{
    float a, b, c;
    if (EDGE) {
        a = ADD ? r.v[0] + r.a[0] : r.v[0], b = ADD ? r.v[1] + r.a[1] : r.v[1], c = ADD ? r.v[2] + r.a[2] : r.v[2];
        special = special || is_special(a) || is_special(b) || is_special(c);
    } else {
        b = ADD ? r.v[0] + r.a[0] : r.v[0];
        special |= is_special(b);
        a = lane_left(b), c = lane_right(b);
    }
    t[0] = fminf(fminf(a, b), c);
    t[1] = __builtin_amdgcn_fmed3f(a, b, c);
    t[2] = fmaxf(fmaxf(a, b), c);
}

__device__ __forceinline__ float median_of_sorted_rows(const float (&p)[3], const float (&q)[3], const float (&r)[3])
{
    return __builtin_amdgcn_fmed3f(fmaxf(fmaxf(p[0], q[0]), r[0]), __builtin_amdgcn_fmed3f(p[1], q[1], r[1]), fminf(fminf(p[2], q[2]), r[2]));
}

// Step I of two (the ring of four row slots advances by two rows per step): output rows ya, ya + 1 from rows ya-1 .. ya+2.
template <bool EDGE, int I, bool ADD>
__device__ __forceinline__ void median3_step(float (&ring)[4][3], Row3Load<EDGE, ADD> (&next)[4], const Source<ADD> in,
                                             float* __restrict__ out, int ya, int y1, int h, int pitch, int x, int xc,
                                             const int (&xm)[3], bool lane_stores, bool& special)
{
    // rows ya+1 and ya+2 were requested two steps ago; request rows ya+5, ya+6
    sorted_triple<EDGE, ADD>(next[0], ring[(2 * I + 2) % 4], special);
    sorted_triple<EDGE, ADD>(next[1], ring[(2 * I + 3) % 4], special);
    next[0] = next[2];
    next[1] = next[3];
    next[2] = load_row3<EDGE, ADD>(in, ya + 5, h, pitch, xc, xm);
    next[3] = load_row3<EDGE, ADD>(in, ya + 6, h, pitch, xc, xm);
    const float a = median_of_sorted_rows(ring[(2 * I) % 4], ring[(2 * I + 1) % 4], ring[(2 * I + 2) % 4]);
    const float b = median_of_sorted_rows(ring[(2 * I + 1) % 4], ring[(2 * I + 2) % 4], ring[(2 * I + 3) % 4]);
    if (lane_stores) {
        const unsigned at = (static_cast<unsigned>(ya) * static_cast<unsigned>(pitch) + static_cast<unsigned>(x)) * 4u;
        plane_store(out, at, a);
        if (ya + 1 < y1) plane_store(out, at + static_cast<unsigned>(pitch) * 4u, b);
    }
}

template <bool EDGE, bool ADD>
__device__ __forceinline__ void median3_strip(const Source<ADD> in, float* __restrict__ out, int w, int h, int pitch,
                                              int x, int y0, int y1, bool lane_stores)
{
    const int xc = min(max(x, 0), w - 1);
    int xm[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) xm[i] = min(max(mirror_index(x + i - 1, w), 0), w - 1);
    float ring[4][3];
    bool special = false;
    {  // rows y0-1, y0 fill slots 0, 1; rows y0+1 .. y0+4 are the first two pairs in flight
        Row3Load<EDGE, ADD> first[2] = {load_row3<EDGE, ADD>(in, y0 - 1, h, pitch, xc, xm), load_row3<EDGE, ADD>(in, y0, h, pitch, xc, xm)};
        sorted_triple<EDGE, ADD>(first[0], ring[0], special);
        sorted_triple<EDGE, ADD>(first[1], ring[1], special);
#pragma unroll
        for (int e = 0; e < 3; ++e) ring[2][e] = ring[3][e] = 0.f;
    }
    Row3Load<EDGE, ADD> next[4] = {load_row3<EDGE, ADD>(in, y0 + 1, h, pitch, xc, xm), load_row3<EDGE, ADD>(in, y0 + 2, h, pitch, xc, xm),
                                   load_row3<EDGE, ADD>(in, y0 + 3, h, pitch, xc, xm), load_row3<EDGE, ADD>(in, y0 + 4, h, pitch, xc, xm)};
    for (int ya = y0; ya < y1; ya += 4) {
        median3_step<EDGE, 0, ADD>(ring, next, in, out, ya, y1, h, pitch, x, xc, xm, lane_stores, special);
        if (ya + 2 >= y1) break;
        median3_step<EDGE, 1, ADD>(ring, next, in, out, ya + 2, y1, h, pitch, x, xc, xm, lane_stores, special);
    }
    // Some lane of this wave loaded a NaN or a -0: look at each stored pixel's window again and redo those that hold one the way
    // the reference's sort would (like the 5 x 5 strip).
    if (__builtin_amdgcn_ballot_w64(special) != 0 && lane_stores) {
        for (int y = y0; y < y1; ++y) {
            bool hit = false;
            for (int j = -1; j <= 1; ++j) {
                const size_t line = static_cast<size_t>(mirror_index(y + j, h)) * pitch;
                for (int i = -1; i <= 1; ++i) hit |= is_special(in[line + mirror_index(x + i, w)]);
            }
            if (hit) out[static_cast<size_t>(y) * pitch + x] = exact_median<3, ADD>(in, x, y, w, h, pitch);
        }
    }
}

template <bool ADD>
__global__ __launch_bounds__(256) void median3_stream_kernel(const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                             const float* __restrict__ add_a, const float* __restrict__ add_b, int w, int h,
                                                             int pitch, int rows_per_strip, float* __restrict__ out_a,
                                                             float* __restrict__ out_b, BatchArg batch)
{
    const Source<ADD> in{(batch_plane(batch) ? in_b : in_a) + batch_offset(batch),
                         ADD ? (batch_plane(batch) ? add_b : add_a) + batch_offset(batch) : nullptr};
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x_first = strip * kStream3Valid - 1;
    if (strip * kStream3Valid >= w) return;
    const int x = x_first + lane;
    const int y0 = blockIdx.y * rows_per_strip;
    const int y1 = min(y0 + rows_per_strip, h);
    const bool lane_stores = lane >= 1 && lane < 63 && x < w;
    const bool edge = x_first < 0 || x_first + 63 > w - 1;
    if (__builtin_amdgcn_readfirstlane(edge))
        median3_strip<true, ADD>(in, out, w, h, pitch, x, y0, y1, lane_stores);
    else
        median3_strip<false, ADD>(in, out, w, h, pitch, x, y0, y1, lane_stores);
}

// ---- r = 7, streaming (round 4) -----------------------------------------------------------------------------------
// The same scheme one size up: a lane loads ONE value per image row, takes three neighbours per side from the adjacent
// lanes (DPP), sorts the 7-tuple (16 comparators) and keeps the sorted tuples of eight consecutive rows in registers.  Two
// vertically adjacent medians share six of their seven rows: median7_pair_network.inc (a 212-comparator merge network generated and
// verified exhaustively over the 8^8 sorted 0-1 inputs by tools/gen_median7_network.py, lowered to 254 one-result instructions with
// three-input minima, maxima and medians by tools/median_select3.py) merges the six shared tuples once and finishes both.
// The generic kernel gathers 49 values per pixel and sorts them with a pruned Batcher network: 412 us per 4096^2 plane.
#include "median7_pair_network.inc"

__device__ __forceinline__ void sort7(float (&t)[7])
{
    float n[7 + kSort7Ops];
#pragma unroll
    for (int i = 0; i < 7; ++i) n[i] = t[i];
    run_select_program<7, kSort7Ops, kSort7Program>(n, std::make_index_sequence<kSort7Ops>{});
#pragma unroll
    for (int i = 0; i < 7; ++i) t[i] = n[kSort7Out[i]];
}

constexpr int kStream7Valid = 58;  // lanes 3..60 of a wave have three neighbours on either side

template <bool EDGE, bool ADD>
struct Row7Load {
    float v[EDGE ? 7 : 1];
    float a[ADD ? (EDGE ? 7 : 1) : 1];
};

template <bool EDGE, bool ADD>
__device__ __forceinline__ Row7Load<EDGE, ADD> load_row7(const Source<ADD> in, int row, int h, int pitch, int xc,
                                                         const int (&xm)[7])
{
    const unsigned line = static_cast<unsigned>(min(max(mirror_index(row, h), 0), h - 1)) * static_cast<unsigned>(pitch);
    Row7Load<EDGE, ADD> r;
    r.a[0] = 0.f;
#pragma unroll
    for (int i = 0; i < (EDGE ? 7 : 1); ++i) {
        const unsigned at = (line + static_cast<unsigned>(EDGE ? xm[i] : xc)) * 4u;
        r.v[i] = plane_load(in.in, at);
        if (ADD) r.a[i] = plane_load(in.add, at);
    }
    return r;
}

template <bool EDGE, bool ADD>
__device__ __forceinline__ void sorted_tuple7(const Row7Load<EDGE, ADD>& r, float (&t)[7], bool& special)
{
    if (EDGE) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            t[i] = ADD ? r.v[i] + r.a[i] : r.v[i];
            special |= is_special(t[i]);
        }
    } else {
        const float c = ADD ? r.v[0] + r.a[0] : r.v[0];
        special |= is_special(c);
        const float l1 = lane_left(c), r1 = lane_right(c);
        const float l2 = lane_left(l1), r2 = lane_right(r1);
        t[0] = lane_left(l2);
        t[1] = l2;
        t[2] = l1;
        t[3] = c;
        t[4] = r1;
        t[5] = r2;
        t[6] = lane_right(r2);
    }
    sort7(t);
}

// Step I of four: the ring of eight row slots advances by two rows per step.
template <bool EDGE, int I, bool ADD>
__device__ __forceinline__ void median7_step(float (&ring)[8][7], Row7Load<EDGE, ADD> (&next)[4], const Source<ADD> in,
                                             float* __restrict__ out, int ya, int y1, int h, int pitch, int x, int xc,
                                             const int (&xm)[7], bool lane_stores, bool& special)
{
    // rows ya+3 and ya+4 were requested two steps ago; request rows ya+7, ya+8
    sorted_tuple7<EDGE, ADD>(next[0], ring[(2 * I + 6) % 8], special);
    sorted_tuple7<EDGE, ADD>(next[1], ring[(2 * I + 7) % 8], special);
    next[0] = next[2];
    next[1] = next[3];
    next[2] = load_row7<EDGE, ADD>(in, ya + 7, h, pitch, xc, xm);
    next[3] = load_row7<EDGE, ADD>(in, ya + 8, h, pitch, xc, xm);
    float v[kMedian7PairInputs + kMedian7PairOps];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int e = 0; e < 7; ++e) v[7 * g + e] = ring[(2 * I + g) % 8][e];
    run_select_program<kMedian7PairInputs, kMedian7PairOps, kMedian7PairProgram>(v, std::make_index_sequence<kMedian7PairOps>{});
    if (lane_stores) {
        const unsigned at = (static_cast<unsigned>(ya) * static_cast<unsigned>(pitch) + static_cast<unsigned>(x)) * 4u;
        plane_store(out, at, v[kMedian7PairOutA]);
        if (ya + 1 < y1) plane_store(out, at + static_cast<unsigned>(pitch) * 4u, v[kMedian7PairOutB]);
    }
}

template <bool EDGE, bool ADD>
__device__ __forceinline__ void median7_strip(const Source<ADD> in, float* __restrict__ out, int w, int h, int pitch,
                                              int x, int y0, int y1, bool lane_stores)
{
    const int xc = min(max(x, 0), w - 1);
    int xm[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) xm[i] = min(max(mirror_index(x + i - 3, w), 0), w - 1);
    float ring[8][7];
    bool special = false;
    // rows y0-3 .. y0+2 fill slots 0..5; rows y0+3, y0+4 are the first pair in flight
    {
        Row7Load<EDGE, ADD> first[6];
#pragma unroll
        for (int g = 0; g < 6; ++g) first[g] = load_row7<EDGE, ADD>(in, y0 - 3 + g, h, pitch, xc, xm);
#pragma unroll
        for (int g = 0; g < 6; ++g) sorted_tuple7<EDGE, ADD>(first[g], ring[g], special);
#pragma unroll
        for (int e = 0; e < 7; ++e) ring[6][e] = ring[7][e] = 0.f;
    }
    Row7Load<EDGE, ADD> next[4] = {load_row7<EDGE, ADD>(in, y0 + 3, h, pitch, xc, xm), load_row7<EDGE, ADD>(in, y0 + 4, h, pitch, xc, xm),
                                   load_row7<EDGE, ADD>(in, y0 + 5, h, pitch, xc, xm), load_row7<EDGE, ADD>(in, y0 + 6, h, pitch, xc, xm)};
    for (int ya = y0; ya < y1; ya += 8) {
        median7_step<EDGE, 0, ADD>(ring, next, in, out, ya, y1, h, pitch, x, xc, xm, lane_stores, special);
        if (ya + 2 >= y1) break;
        median7_step<EDGE, 1, ADD>(ring, next, in, out, ya + 2, y1, h, pitch, x, xc, xm, lane_stores, special);
        if (ya + 4 >= y1) break;
        median7_step<EDGE, 2, ADD>(ring, next, in, out, ya + 4, y1, h, pitch, x, xc, xm, lane_stores, special);
        if (ya + 6 >= y1) break;
        median7_step<EDGE, 3, ADD>(ring, next, in, out, ya + 6, y1, h, pitch, x, xc, xm, lane_stores, special);
    }
    // a NaN or a -0 somewhere in what the wave loaded: the windows that hold one are redone the way the reference's sort would
    if (__builtin_amdgcn_ballot_w64(special) != 0 && lane_stores) {
        for (int y = y0; y < y1; ++y) {
            bool hit = false;
            for (int j = -3; j <= 3; ++j) {
                const size_t line = static_cast<size_t>(mirror_index(y + j, h)) * pitch;
                for (int i = -3; i <= 3; ++i) hit |= is_special(in[line + mirror_index(x + i, w)]);
            }
            if (hit) out[static_cast<size_t>(y) * pitch + x] = exact_median<7, ADD>(in, x, y, w, h, pitch);
        }
    }
}

template <bool ADD>
__global__ __launch_bounds__(256) void median7_stream_kernel(const float* __restrict__ in_a,
                                                             const float* __restrict__ in_b,
                                                             const float* __restrict__ add_a,
                                                             const float* __restrict__ add_b, int w, int h, int pitch,
                                                             int rows_per_strip, float* __restrict__ out_a,
                                                             float* __restrict__ out_b, BatchArg batch)
{
    const Source<ADD> in{(batch_plane(batch) ? in_b : in_a) + batch_offset(batch),
                         ADD ? (batch_plane(batch) ? add_b : add_a) + batch_offset(batch) : nullptr};
    float* __restrict__ out = (batch_plane(batch) ? out_b : out_a) + batch_offset(batch);
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x_first = strip * kStream7Valid - 3;
    if (strip * kStream7Valid >= w) return;
    const int x = x_first + lane;
    const int y0 = blockIdx.y * rows_per_strip;
    const int y1 = min(y0 + rows_per_strip, h);
    const bool lane_stores = lane >= 3 && lane < 61 && x < w;
    const bool edge = x_first < 0 || x_first + 63 > w - 1;
    if (__builtin_amdgcn_readfirstlane(edge))
        median7_strip<true, ADD>(in, out, w, h, pitch, x, y0, y1, lane_stores);
    else
        median7_strip<false, ADD>(in, out, w, h, pitch, x, y0, y1, lane_stores);
}

// rows per strip: even, tall enough that the four start-up rows are noise, short enough to fill the chip
int median5_rows_per_strip(const flow2d_context* ctx, size_t w, size_t h)
{
    const long strips_x = flow2d::div_up(w, kStreamValid);
    const long want_waves = (ctx->num_cus > 0 ? ctx->num_cus : 256) * 4 * 4;  // four waves per SIMD
    long rows = 64;
    while (rows > 8 && strips_x * (long)flow2d::div_up(h, rows) * (long)ctx->batch_count < want_waves) rows /= 2;
    return (int)rows;
}

}  // namespace

template <bool ADD>
static int launch_median(flow2d_context* ctx, const float* input, const float* input_b, const float* addend,
                         const float* addend_b, size_t width, size_t height, size_t pitch_bytes, size_t window,
                         float* output, float* output_b)
{
    FLOW2D_ENTER(ctx);
    if (!flow2d::plane_args_ok(input, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(output, width, height, pitch_bytes) || input == output)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const bool pair = input_b || output_b;
    if (pair && (!flow2d::plane_args_ok(input_b, width, height, pitch_bytes) ||
                 !flow2d::plane_args_ok(output_b, width, height, pitch_bytes) || input_b == output_b ||
                 output_b == output || output_b == input || output == input_b))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (ADD && (!flow2d::plane_args_ok(addend, width, height, pitch_bytes) || addend == output || addend == output_b ||
                (pair && (!flow2d::plane_args_ok(addend_b, width, height, pitch_bytes) || addend_b == output ||
                          addend_b == output_b))))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (window != 3 && window != 5 && window != 7) return FLOW2D_ERR_UNSUPPORTED;
    // the mirror rule needs every reflected index inside the image
    if (width <= window / 2 || height <= window / 2) return FLOW2D_ERR_UNSUPPORTED;
    const unsigned planes = pair ? 2 : 1;
    const unsigned z = flow2d::batch_z(ctx, planes);
    const BatchArg batch = flow2d::batch_arg(ctx, planes);
    const dim3 grid(flow2d::div_up(width, kBlockX), flow2d::div_up(height, kBlockY), z);
    const dim3 block(kBlockX, kBlockY);
    const int w = (int)width, h = (int)height, pitch = (int)(pitch_bytes / 4);
    const bool offsets_fit = static_cast<unsigned long long>(pitch_bytes) * height < (1ull << 32);  // the streaming kernels' 32-bit byte offsets
    if (window == 3 && width >= 4 && height >= 8 && offsets_fit) {
        const int rows = median5_rows_per_strip(ctx, width, height);
        const dim3 sgrid(flow2d::div_up(flow2d::div_up(width, kStream3Valid), 4), flow2d::div_up(height, rows), z);
        median3_stream_kernel<ADD><<<sgrid, 256, 0, ctx->stream>>>(input, input_b, addend, addend_b, w, h, pitch, rows, output,
                                                                   output_b, batch);
        FLOW2D_CHECK_LAUNCH();
        return FLOW2D_OK;
    }
    if (window == 5 && width >= 8 && height >= 8 && offsets_fit) {  // mirrored rows/columns up to 3 beyond the border stay inside
        const int rows = median5_rows_per_strip(ctx, width, height);
        const dim3 sgrid(flow2d::div_up(flow2d::div_up(width, kStreamValid), 4), flow2d::div_up(height, rows), z);
        median5_stream_kernel<ADD><<<sgrid, 256, 0, ctx->stream>>>(input, input_b, addend, addend_b, w, h, pitch, rows, output,
                                                                   output_b, batch);
        FLOW2D_CHECK_LAUNCH();
        return FLOW2D_OK;
    }
    if (window == 7 && width >= 12 && height >= 12 && offsets_fit) {  // mirrored rows/columns up to 5 beyond the border stay inside
        const int rows = median5_rows_per_strip(ctx, width, height);
        const dim3 sgrid(flow2d::div_up(flow2d::div_up(width, kStream7Valid), 4), flow2d::div_up(height, rows), z);
        median7_stream_kernel<ADD><<<sgrid, 256, 0, ctx->stream>>>(input, input_b, addend, addend_b, w, h, pitch, rows, output,
                                                                   output_b, batch);
        FLOW2D_CHECK_LAUNCH();
        return FLOW2D_OK;
    }
    switch (window) {
        case 3: median_kernel<3, ADD><<<grid, block, 0, ctx->stream>>>(input, input_b, addend, addend_b, w, h, pitch, output, output_b, batch); break;
        case 5: median_kernel<5, ADD><<<grid, block, 0, ctx->stream>>>(input, input_b, addend, addend_b, w, h, pitch, output, output_b, batch); break;
        default: median_kernel<7, ADD><<<grid, block, 0, ctx->stream>>>(input, input_b, addend, addend_b, w, h, pitch, output, output_b, batch); break;
    }
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

extern "C" int flow2d_median_2d(flow2d_context* ctx, const float* input, size_t width, size_t height,
                                size_t pitch_bytes, size_t window, float* output)
{
    return launch_median<false>(ctx, input, nullptr, nullptr, nullptr, width, height, pitch_bytes, window, output, nullptr);
}

extern "C" int flow2d_median_2d_pair(flow2d_context* ctx, const float* input_a, const float* input_b, size_t width,
                                     size_t height, size_t pitch_bytes, size_t window, float* output_a, float* output_b)
{
    if (!input_b || !output_b) return FLOW2D_ERR_INVALID_ARGUMENT;
    return launch_median<false>(ctx, input_a, input_b, nullptr, nullptr, width, height, pitch_bytes, window, output_a, output_b);
}

extern "C" int flow2d_add_median_2d_pair(flow2d_context* ctx, const float* input_a, const float* addend_a, const float* input_b,
                                         const float* addend_b, size_t width, size_t height, size_t pitch_bytes, size_t window,
                                         float* output_a, float* output_b)
{
    if (!addend_a || ((input_b || output_b || addend_b) && !(input_b && output_b && addend_b))) return FLOW2D_ERR_INVALID_ARGUMENT;
    return launch_median<true>(ctx, input_a, input_b, addend_a, addend_b, width, height, pitch_bytes, window, output_a, output_b);
}
