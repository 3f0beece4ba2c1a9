// Fused outer iteration of the variational solver for gfx950: robust weights (phi, ksi) + ALL inner
// Jacobi sweeps of one outer iteration in ONE pass over HBM.
//
// The reference launches compute_phi_ksi once and solve_2d* `inner` times per outer iteration
// (cuda_operation_solve_2d.cpp:238-299), moving 32 + 40*inner bytes per pixel through DRAM.  Here a
// wavefront owns a strip of 64 columns and streams down the image row by row with every plane it needs
// held in registers: each lane keeps, for its column, 3-row sliding windows of the inputs and of the
// intermediate flow of every sweep, and the per-pixel coefficients in a short delay line.  Sweep k of
// row r-2-k runs as soon as sweep k-1 has produced row r-1-k (time skewing), so one trip down the strip
// performs phi/ksi and all sweeps while reading f0, f1, u, v, du, dv once and writing du, dv once
// (about 32 B per pixel instead of 32 + 40*inner).
//
//  - x neighbours come from the adjacent lanes with DPP wave shifts (no LDS, no barriers; the four waves
//    of a workgroup are independent strips).  A strip computes INNER+1 halo columns per side redundantly,
//    so 64 - 2*(INNER+1) columns of each wave are stored.
//  - y neighbours are the lane's own registers (sliding windows, rotated by unrolling the row loop).
//  - Image borders follow the reference's reflect rule (-1 -> 1, n -> n-2) by substituting the opposite
//    neighbour at the border pixel; rows/columns outside the image are computed on clamped addresses and
//    never reach a stored value.
//  - Every pixel goes through exactly the expressions of solve_2d.cu (solver_math.hpp), in the same
//    order, without FMA contraction: results are bit-identical to the per-sweep kernels and the oracle.
//
//  - The ten divisions of a row step (two per sweep, solve_2d.cu:363,367) are three instructions each: with
//    y = RN(1 / den) taken once per pixel and outer iteration by a true division, q0 = n * y, r = fma(-q0, den, n),
//    q = fma(r, y, q0) is RN(n / den) bit for bit -- for every pair of fp32 significands, checked exhaustively on the
//    device (tools/ubench/markstein_exhaustive.hip, 7.0e13 pairs), and therefore for all operands that keep q0 and r
//    inside the normal range.  The kernel checks exactly that as it goes (den within [2^-30, 2^40], no numerator in
//    (0, 2^-80), no stored value infinite or NaN); a wave that sees anything else repeats its strip with the plain
//    division, so the result never depends on the shortcut.
//  - Border strips are shorter than interior ones (FusedPlan): a wave on an image border executes about a fifth more
//    instructions per row, and a launch -- one round of waves -- lasts as long as its slowest wave.
//
// Bound: vector-instruction issue, not HBM.  A SIMD hands out one issue turn per ~4.3 cycles; two waves share a turn only where
// both instructions are plain ones (no DPP, no packed arithmetic, no transcendental) and a scalar instruction costs a turn too --
// hence the build (csrc/Makefile: no packed fp32 in this kernel's device code, issue_priority.py around the DPP / transcendental
// runs) and the care for every scalar instruction in the row step.  DESIGN.md section 3.1.1.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <utility>

#include "common.hpp"
#include "solve_fused_args.hpp"
#include "solver_math.hpp"

// Developer probes (stamps and stall histogram, compute-only / memory-only / no-halo / packed-plane timing builds) are compile-time
// constants of solve_fused_probes.hpp, all false in the product library.  One build option is not a probe (csrc/Makefile sets it
// for the instances it compiles WITH packed fp32 arithmetic): FLOW2D_FUSED_PACKED, the face products as v_pk_mul_f32 ... op_sel.
#include "solve_fused_probes.hpp"

namespace {

using namespace flow2d_math;
using flow2d::FusedArgs;

namespace probe = flow2d_probe;

// A plane row is addressed as base pointer (a scalar register pair) + one 32-bit per-lane byte offset that all planes
// share (global_load / global_store ... saddr): two scalar registers per plane.  (Buffer descriptors, four scalar
// registers per plane, pushed the kernel past the scalar register file: the descriptors were spilled to vector-register
// lanes and read back, 24 v_readlane per row step.)  Planes stay below 4 GiB (fused_addressable).
__device__ __forceinline__ float plane_load(const float* plane, unsigned byte_offset)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(plane) + static_cast<size_t>(byte_offset));
}
__device__ __forceinline__ void plane_store(float* plane, unsigned byte_offset, float value)
{
    *reinterpret_cast<float*>(reinterpret_cast<char*>(plane) + static_cast<size_t>(byte_offset)) = value;
}

// One input row of a lane's column on its way from memory: the frames, the flow, the increment.
struct RowFetch {
    float f0, f1;
    v2f uv, duv;
};
typedef float v4f __attribute__((ext_vector_type(4)));
// byte_offset: (row * pitch + column) * 4, the offset all planes share
__device__ __forceinline__ RowFetch fetch_row(const FusedArgs& a, unsigned byte_offset)
{
    if constexpr (probe::kPackedPlanes) {  // probe: (f0, f1, u, v) of a pixel side by side in one plane, (du, dv) in another
        const v4f q = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(a.pack_in) + static_cast<size_t>(byte_offset * 4u));
        const v2f d = *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(a.pack_duv) + static_cast<size_t>(byte_offset * 2u));
        return RowFetch{q.x, q.y, v2f{q.z, q.w}, d};
    } else {
        // the increment planes are read whether or not the launch treats them as zero (first outer iteration: they are valid planes
        // with stale contents) and the zero is selected when the row is committed: no branch around two loads in every step
        return RowFetch{plane_load(a.f0, byte_offset), plane_load(a.f1, byte_offset), v2f{plane_load(a.u, byte_offset), plane_load(a.v, byte_offset)},
                        v2f{plane_load(a.du, byte_offset), plane_load(a.dv, byte_offset)}};
    }
}
__device__ __forceinline__ void store_row(const FusedArgs& a, unsigned byte_offset, float du, float dv)
{
    if constexpr (probe::kPackedPlanes) {
        *reinterpret_cast<v2f*>(reinterpret_cast<char*>(a.pack_out) + static_cast<size_t>(byte_offset * 2u)) = v2f{du, dv};
    } else {
        plane_store(a.out_du, byte_offset, du);
        plane_store(a.out_dv, byte_offset, dv);
    }
}
__device__ __forceinline__ unsigned row_offset(const FusedArgs& a, int row, int xc)
{
    return (static_cast<unsigned>(row) * static_cast<unsigned>(a.pitch) + static_cast<unsigned>(xc)) * 4u;
}

// lane i receives lane i-1 (wave_shr:1) / lane i+1 (wave_shl:1); the end lanes of the wave receive 0
// (bound_ctrl), which only ever reaches halo columns.  No "old" operand, so the move can fold into the
// consuming VALU instruction.
__device__ __forceinline__ float from_left(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}


// static per-pixel coefficients of one outer iteration (what the sweeps need besides the moving flow)
struct Coef {
    v2f wx, wy;    // (w_x+, w_x-), (w_y+, w_y-): face diffusivity * neighbour weight (solve_2d.cu:337-346)
    v2f uvc;       // (u, v) of the pixel
    v2f den;       // (ksi * J11 + sumH, ksi * J22 + sumH) (solve_2d.cu:363,367)
    v2f rden;      // (RN(1 / den.x), RN(1 / den.y)): three-step division only
    v2f J13_23;    // (J13, J23)
    float ksi, J12;
};

// n / d through the prepared reciprocal y = RN(1 / d): RN(n / d) exactly whenever q0 and r stay in the normal range
// (see the header).
__device__ __forceinline__ float div3(float n, float d, float y)
{
    const float q0 = n * y;
    const float r = __builtin_fmaf(-q0, d, n);
    return __builtin_fmaf(r, y, q0);
}

// 1.0f / d in three instructions: the hardware approximation and one fma pair.  RN(1 / d) for EVERY normal d whose
// reciprocal is normal -- all 2 113 929 217 bit patterns of [2^-126, 2^126] tried on the device
// (tools/ubench/rcp_sqrt_exhaustive.hip; the compiler's correctly rounded division is eleven, five of them quarter-rate).
// The kernel only ever needs two of them side by side -- the two denominators of a pixel, the two robustifiers of a pixel --
// so the fma steps are packed instructions (v_pk_fma_f32), each half the scalar sequence bit for bit.
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f rcp3_pair(v2f d)
{
    const v2f y0 = v2f{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const v2f e = fma2(-d, y0, v2f{1.f, 1.f});
    return fma2(e, y0, y0);
}

// 1.f / (2.f * sqrtf(s)) in nine instructions instead of 28: the hardware root corrected by its residual times half the
// hardware reciprocal root is RN(sqrt(s)), and rcp3 of twice that (exact) is the quotient -- for every s whose
// 2 sqrt(s) lies within the guarded denominator range (every bit pattern tried, same ubench).  twice_root goes to the
// guard: a zero, an infinity, a NaN or a negative argument leave a NaN or an out-of-range value there.
__device__ __forceinline__ v2f half_inverse_root_pair(v2f s, v2f& twice_root)
{
    const v2f g0 = v2f{__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y)};
    const v2f h = 0.5f * v2f{__builtin_amdgcn_rsqf(s.x), __builtin_amdgcn_rsqf(s.y)};
    const v2f r = fma2(-g0, g0, s);
    const v2f g = fma2(r, h, g0);
    twice_root = 2.f * g;
    return rcp3_pair(twice_root);
}

// What the proof does not cover is recorded per lane in four integer accumulators -- integer min / max on the operands'
// bit patterns, vector ALU only (a comparison per division would go through the scalar unit: measured, it costs more than
// the divisions it guards) -- and judged once, after the strip:
//   tiny: min of (bits(n) << 1) - 1 over the numerators: the shift drops the sign, the decrement sends a zero (harmless:
//         q0 = r = q = 0) to the top; a non-zero numerator below 2^-80 lands below kTinyLimit.  The one zero that is not
//         harmless is a numerator of exactly -0, whose quotient is -0 while the three steps give +0; it takes a -0 in the
//         flow planes to produce one (the sum of the four face terms is -0 only if all four are):
//   zero: signed min of the raw bits of every flow value read: INT_MIN exactly when one of them is a -0
//   den : float minimum and maximum of the denominators (v_min3_f32 / v_max3_f32: two instructions for a pair instead of
//         two integer subtractions and a v_max3_u32): outside [2^-30, 2^40] for a denominator that is too small, too
//         large, zero, negative or infinite.  A NaN passes both (minNum / maxNum return the other operand) -- and turns
//         du, dv of its pixel into NaN in the first sweep, in either form of the division, which `out` sees
//   out : max of bits(du, dv) << 1 over the stored results: above kOutLimit for an infinity or a NaN
#define FLOW2D_GUARD_PIN(x) asm volatile("" : "+v"(x))
struct DivGuard {
    unsigned tiny, out;
    float den_lo, den_hi;
    int zero;
};
constexpr unsigned kTinyLimit = 2u * 0x17800000u - 1u;         // 2^-80 = 0x17800000
constexpr float kDenLow = 0x1p-30f, kDenHigh = 0x1p40f;
constexpr unsigned kOutLimit = 0xfefffffeu;                    // FLT_MAX << 1
// (the empty asm pins each update where it is written: left alone, the compiler sinks all updates of a ring turn to the
//  loop latch and keeps the sixty numerators of the turn alive until then)
__device__ __forceinline__ void guard_numerators(DivGuard& g, float nu, float nv)
{
    g.tiny = min(g.tiny, min((__float_as_uint(nu) << 1) - 1u, (__float_as_uint(nv) << 1) - 1u));
    FLOW2D_GUARD_PIN(g.tiny);
}
__device__ __forceinline__ void guard_flow_row(DivGuard& g, v2f uv, v2f duv)
{
    // read as a signed integer, -0 (0x80000000) is the smallest value there is: a signed minimum over the raw bits
    // reaches INT_MIN exactly when some value is a -0 (two v_min3_i32 per row instead of four v_xor and two minima)
    g.zero = min(min(g.zero, __float_as_int(uv.x)), __float_as_int(uv.y));
    g.zero = min(min(g.zero, __float_as_int(duv.x)), __float_as_int(duv.y));
    FLOW2D_GUARD_PIN(g.zero);
}
__device__ __forceinline__ void guard_denominators(DivGuard& g, float du, float dv)
{
    // (the operands are results of ordinary arithmetic, never the direct result of a transcendental instruction: see mul_by_x)
    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(g.den_lo) : "v"(du), "v"(dv));
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(g.den_hi) : "v"(du), "v"(dv));
}
__device__ __forceinline__ void guard_results(DivGuard& g, float du, float dv)
{
    g.out = max(g.out, max(__float_as_uint(du) << 1, __float_as_uint(dv) << 1));
    FLOW2D_GUARD_PIN(g.out);
}
// x / d for the grid-spacing divisors 2h and 4h (wave-uniform; y = RN(1 / d) comes from the host): an exact multiply
// when they are powers of two, the three-step division under the numerator guard otherwise, the plain division in the
// fallback pass.  (A spacing outside [2^-30, 2^40] sends the whole launch to the fallback: FusedArgs::plain_only.)
template <bool POW2, bool FAST>
__device__ __forceinline__ float spacing_quotient(DivGuard& g, float n, float d, float y)
{
    if (POW2) return n * y;
    if (!FAST) return n / d;
    g.tiny = min(g.tiny, (__float_as_uint(n) << 1) - 1u);
    FLOW2D_GUARD_PIN(g.tiny);
    return div3(n, d, y);
}
template <bool POW2, bool FAST>
__device__ __forceinline__ v2f spacing_quotient2(DivGuard& g, v2f n, float d, float y)
{
    if (POW2) return n * y;
    if (!FAST) return v2f{n.x / d, n.y / d};
    guard_numerators(g, n.x, n.y);
    const v2f q0 = n * y;  // div3 on both halves
    return fma2(fma2(-q0, v2f{d, d}, n), v2f{y, y}, q0);
}

// (n.x / d.x, n.y / d.y): the x and the y spacing's quotients of one pixel
template <bool POW2, bool FAST>
__device__ __forceinline__ v2f spacing_quotient_pair(DivGuard& g, v2f n, v2f d, v2f y)
{
    if (POW2) return n * y;
    if (!FAST) return v2f{n.x / d.x, n.y / d.y};
    guard_numerators(g, n.x, n.y);
    const v2f q0 = n * y;  // div3 on both halves
    return fma2(fma2(-q0, d, n), y, q0);
}

__device__ __forceinline__ bool guard_tripped(const DivGuard& g)
{
    return g.tiny < kTinyLimit || !(g.den_lo >= kDenLow) || !(g.den_hi <= kDenHigh) || g.out > kOutLimit || g.zero == static_cast<int>(0x80000000u);
}

// (w.x * d.x, w.x * d.y) and (w.y * d.x, w.y * d.y): v_pk_mul_f32 reading ONE half of w for both products (op_sel).  The
// compiler builds dup_x(w) * d from a register pair it first assembles with moves -- five v_mov_b32 per sweep for the four
// face weights, in every sweep anew; the products are the same.  (The compiler does not look into an asm statement when it
// pads data hazards: the operands here are results of ordinary or packed arithmetic, never the direct result of a
// transcendental instruction, which on gfx950 needs a wait state before an ordinary VALU instruction may read it.)
__device__ __forceinline__ v2f mul_by_x(v2f w, v2f d)
{
#ifndef FLOW2D_FUSED_PACKED
    return v2f{w.x * d.x, w.x * d.y};
#else
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(d));
    return r;
#endif
}
__device__ __forceinline__ v2f mul_by_y(v2f w, v2f d)
{
#ifndef FLOW2D_FUSED_PACKED
    return v2f{w.y * d.x, w.y * d.y};
#else
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(w), "v"(d));
    return r;
#endif
}
// keeps the vectoriser from pairing two scalar operations into a packed one (an empty statement: no instruction)
__device__ __forceinline__ float scalar_only(float v)
{
    asm("" : "+v"(v));
    return v;
}
// ksi_argument (solver_math.hpp) with the rows of the brightness tensor as pairs -- 15 instructions instead of 26; every
// product and every sum is one of the scalar form's, in its order (J12 = fx fy is formed as fy fx in one of its two copies)
__device__ __forceinline__ float ksi_argument_packed(float fx, float fy, float ft, v2f duv, float e_data)
{
    const v2f fxy = v2f{fx, fy};
    const v2f J11_12 = mul_by_x(fxy, fxy), J12_22 = mul_by_y(fxy, fxy);  // (fx fx, fx fy), (fy fx, fy fy)
    const v2f J13_23 = fxy * v2f{ft, ft};
    const float J33 = ft * ft;
    // ((J11 du + J12 dv + J13) du, (J12 du + J22 dv + J23) dv) and (J13 du, J23 dv)
    const v2f rows = (mul_by_x(duv, J11_12) + mul_by_y(duv, J12_22) + J13_23) * duv;
    const v2f last = J13_23 * duv;
    float s = rows.x + rows.y + (last.x + last.y + J33);
    s = static_cast<float>(s > 0) * s;
    return s + e_data * e_data;
}
// sum_flux2 (solver_math.hpp) over the four neighbour differences, same order of additions
__device__ __forceinline__ v2f flux_of_differences(v2f wx, v2f wy, v2f dR, v2f dL, v2f dD, v2f dU)
{
    return mul_by_x(wx, dR) + mul_by_y(wx, dL) + mul_by_x(wy, dD) + mul_by_y(wy, dU);
}

__device__ __forceinline__ v2f from_left2(v2f v) { return v2f{from_left(v.x), from_left(v.y)}; }
__device__ __forceinline__ v2f from_right2(v2f v) { return v2f{from_right(v.x), from_right(v.y)}; }
__device__ __forceinline__ v2f pick2(bool c, v2f a, v2f b) { return v2f{c ? a.x : b.x, c ? a.y : b.y}; }

// GRAD: 0 brightness constancy, 1 gradient constancy with the reference's 16x8 tile rule, 2 gradient constancy with
// true neighbours (FLOW2D_CONSTANCY_GRADIENT_UNTILED), 3 solve_2d_log (solve_2d.cu:391-669): as 1 on log(I + 1),
// and every x/y neighbour of the tensor's first derivatives, of phi in the face weights and of the flow in the sweeps
// is the pixel's own value at a 16x8 block edge (that kernel's halo offsets are 0, :448,462,476,490).  phi and ksi
// themselves come from compute_phi_ksi in every mode: brightness tensor, true neighbours.
template <int INNER, int GRAD>
struct Strip {
    static constexpr int kHalo = INNER + 1;                          // halo rows above and below a strip
    static constexpr int kHaloLanes = probe::kNoHalo ? 0 : kHalo;    // halo columns left and right of it (probe: none, wrong edges)
    static constexpr int kValid = 64 - 2 * kHaloLanes;
    static constexpr int kRing = ((INNER + 1 + 2) / 3) * 3;  // coefficient ring, a multiple of the 3-row windows

    // 3-row sliding windows, slot = row mod 3; (u, v) and (du, dv) travel as pairs
    float f0w[3], f1w[3];
    v2f uvw[3], duvw[3];
    float phiw[3];
    float fxw[3], fyw[3], ftw[3];  // GRAD only
    float lf0w[3], lf1w[3];        // GRAD == 3 only: log(frame + 1) rows
    v2f UV[INNER][3];                // UV[k] = (u + du^k, v + dv^k) rows around the row sweep k+1 is working on
    v2f duvc[INNER];                 // (du^k, dv^k) of the row sweep k+1 processes in the current step (Jacobi reads dv only)
    Coef C[kRing];
    // brightness derivatives and ksi of the row stage W consumes next (produced by stage P one step earlier)
    float p_fx, p_fy, p_ft, p_ksi;
    // two prefetched input rows in flight: row r+1 (n_*, fetched a step ago) and row r+2 (m_*, fetched in this step)
    RowFetch n, m;
    // continue_sweeps only: the sweeps' starting increment of row r-2 (start_cur) and the row fetched for the
    // next step (n_start)
    v2f start_cur, n_start;
    DivGuard guard;  // three-step division: operands outside the proven range leave their mark here
    // stamps probe: the wave's stalls at the row commit -- lane b of stall[0] counts those of floor(log2(cycles)) == b, lane b of
    // stall[1] adds their cycles up, lane p of stall[2] the cycles of the row steps 4 p .. 4 p + 3 of the strip
    unsigned stall[3];
    unsigned long long stall_t0, stall_t1;  // the clock before and after the last step's wait (scalar registers)
};

// stamps probe: wait for the row this step commits (its L loads; the L of the next row may stay in flight) between two readings
// of the shader clock and file the difference.  The readings come back through the scalar data cache: the wave meets them at
// the end of the step's first stage, not here.
template <int INNER, int GRAD>
__device__ __forceinline__ void timed_row_wait(Strip<INNER, GRAD>& s, int step)
{
    constexpr int kLoadsPerRow = probe::kPackedPlanes ? 2 : 6;
    // (file the PREVIOUS step's pair of readings: they have long arrived, no wait for the scalar cache inside the step)
    const unsigned dt = static_cast<unsigned>(s.stall_t1 - s.stall_t0);
    s.stall_t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerRow) : "memory");
    s.stall_t1 = __builtin_amdgcn_s_memtime();
    const int bin = 31 - __builtin_clz(dt | 1u), prog = (step >> 2) & 63;
    const int lane = static_cast<int>(threadIdx.x & 63u);
    s.stall[0] += lane == bin ? 1u : 0u;
    s.stall[1] += lane == bin ? dt : 0u;
    s.stall[2] += lane == prog ? dt : 0u;
}

// exchange probe (with no halo lanes): what handing edge columns from wave to wave would cost a row step -- one barrier among the
// four waves of the workgroup, n 8-byte LDS writes of live values and n two-address LDS reads whose results reach the guard
template <int INNER, int GRAD>
__device__ __forceinline__ void exchange_cost(Strip<INNER, GRAD>& s, int J)
{
    constexpr int n = probe::kExchange;
    __shared__ v2f slab[2][n > 0 ? n : 1][256 + 2];
    const int col = threadIdx.x + 1, par = J & 1;
#pragma unroll
    for (int i = 0; i < n; ++i) slab[par][i][col] = i < INNER ? s.UV[i][0] : s.uvw[i % 3];
    __builtin_amdgcn_s_barrier();
    unsigned acc = 0u;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const v2f l = slab[par ^ 1][i][col - 1], r = slab[par ^ 1][i][col + 1];
        acc |= __float_as_uint(l.x) & __float_as_uint(r.y);
    }
    s.guard.out = max(s.guard.out, acc & 1u);
}

// EDGE = false: the strip touches no image border, so the reflect substitutions (a v_cndmask per
// neighbour fetch) are compiled out; EDGE = true keeps them.  Chosen per wave (wave-uniform branch).
// POW2: 2h and 4h are powers of two, so dividing by them is an exact multiply by the reciprocal.
// CONT: the launch continues the sweeps of an outer iteration (FusedArgs::continue_sweeps); a template value so
// that the ordinary launch carries none of it.
// T: index of the step in the strip's start-up (0 = the strip's first input row), or -1 in the steady state.  A
// stored row y needs sweep k on rows y-(INNER-k) .. y+(INNER-k) only, so during the first steps of a strip the later
// stages would work on rows nothing depends on: stage P is first needed at step 2, stage W at step 3, sweep k at
// step 3 + 2k.  The start-up steps are peeled off the row loop and compiled without those stages.
// FAST: the sweeps divide through the prepared reciprocal (div3); false = plain division (the fallback pass).
// SOR: the INNER stages are red-black half-sweeps of the opt-in successive over-relaxation (flow2d_solve_params.sor_omega;
// not a reference mode, the reference is Jacobi): stage k relaxes the pixels with (x + y) % 2 == (k - 1) % 2 in place --
// du' = (1 - omega) du + omega gs_du with the Gauss-Seidel value of the same point update, dv' sees the relaxed du' --
// and passes the other colour through.  A half-sweep reaches one pixel like a Jacobi sweep, so the skewed pipeline, its
// halo and its start-up are those of INNER Jacobi sweeps; INNER / 2 iterations per launch.
template <int INNER, int GRAD, bool EDGE, bool POW2, bool CONT, bool FAST, bool SOR, int J, int T = -1>
__device__ __forceinline__ void strip_step(Strip<INNER, GRAD>& s, const FusedArgs& a, int r, int x, int xc, bool at_l,
                                           bool at_r, bool lane_stores, int y0, int y1, float hx_2,
                                           float hy_2)
{
    using S = Strip<INNER, GRAD>;
    constexpr int kRing = S::kRing;
    // slots of input rows r, r-1, r-2 (J == r - first_row mod kRing, kRing % 3 == 0)
    constexpr int s0 = J % 3, s1 = (J + 2) % 3, s2 = (J + 1) % 3;
    const int w = a.w, h = a.h;
    if (!EDGE) at_l = at_r = false;

    // ---- commit the prefetched row r (its slot still holds row r-3: sweep 1 needs that row's dv) ----------
    constexpr bool cont = CONT;
    const v2f duv_row3 = cont ? s.start_cur : s.duvw[s0];  // start_cur still is row r-3 here
    if (cont) {
        if (FAST) guard_flow_row(s.guard, s.n_start, s.n_start);
        s.start_cur = s.n_start;  // row r-2
        const int rs = min(max(r - 1, 0), h - 1);
        const unsigned off = (static_cast<unsigned>(rs) * static_cast<unsigned>(a.pitch) + static_cast<unsigned>(xc)) * 4u;
        s.n_start = v2f{plane_load(a.start_du, off), plane_load(a.start_dv, off)};
    }
    if constexpr (probe::kStalls && T < 0) timed_row_wait(s, r - 1 - (y0 - S::kHalo));
    if constexpr (probe::kExchange >= 0 && T < 0) exchange_cost(s, J);
    s.f0w[s0] = s.n.f0;
    s.f1w[s0] = s.n.f1;
    s.uvw[s0] = s.n.uv;
    s.duvw[s0] = v2f{a.zero_increment ? 0.f : s.n.duv.x, a.zero_increment ? 0.f : s.n.duv.y};
    if (FAST) guard_flow_row(s.guard, s.n.uv, s.duvw[s0]);
    if (GRAD == 3) {
        s.lf0w[s0] = log1p_frame(s.n.f0);
        s.lf1w[s0] = log1p_frame(s.n.f1);
    }
    {  // row r+1 arrived a step ago; fetch row r+2 (clamped: rows outside the image are never used by a stored pixel)
        s.n = s.m;
        constexpr int kAhead = 2;
        // (a strip without EDGE keeps kHalo + 1 + kAhead rows away from the top and bottom border -- the kernel's edge test -- so
        //  its prefetch row needs no clamp: two scalar instructions less per step, each of which costs the wave an issue turn)
        // compute-only probe: every row folded onto eight cache-resident rows
        const int rn = probe::kComputeOnly ? (r + kAhead) & 7 : EDGE ? min(max(r + kAhead, 0), h - 1) : r + kAhead;
        s.m = fetch_row(a, row_offset(a, rn, xc));
    }
    if constexpr (probe::kMemoryOnly) {  // the strip's loads and stores without its arithmetic
        const int rk = r - 2 - INNER;
        if (lane_stores && rk >= y0 && rk < y1)
            store_row(a, row_offset(a, rk, xc), s.f0w[s0] + s.uvw[s0].x + s.duvw[s0].x, s.f1w[s0] + s.uvw[s0].y + s.duvw[s0].y);
        return;
    }
    constexpr bool run_P = T < 0 || T >= 2, run_W = T < 0 || T >= 3;
    if (T >= 0) __builtin_amdgcn_sched_barrier(0);  // keep the straight-line start-up from being interleaved across steps

    // ---- stage P, row rp = r-1: phi, brightness derivatives, ksi (solve_2d.cu:138-197) -------------------
    const int rp = r - 1;
    float fx = 0.f, fy = 0.f, ft = 0.f, ksi = 0.f;
    auto stage_P = [&]() {
    if (run_P) {
        const bool top = EDGE && (rp == 0), bot = EDGE && (rp == h - 1);
        v2f xnum, ynum;  // numerators of (dux, dvx) and (duy, dvy): aP - aM + bP - bM, solve_2d.cu:141-157
        if (!EDGE) {
            // no border in this strip: the lane shifts ride as DPP operands of scalar subtractions / additions (three of
            // the four x neighbours of a component; a packed operation needs all of them moved into registers first)
            const v2f uvc = s.uvw[s1], duvc = s.duvw[s1];
            xnum = v2f{scalar_only(scalar_only(scalar_only(from_right(uvc.x) - from_left(uvc.x)) + from_right(duvc.x)) - from_left(duvc.x)),
                       scalar_only(scalar_only(scalar_only(from_right(uvc.y) - from_left(uvc.y)) + from_right(duvc.y)) - from_left(duvc.y))};
            ynum = diff4_num2(s.uvw[s0], s.uvw[s2], s.duvw[s0], s.duvw[s2]);
        } else {
            // cross-lane reads happen with every lane active; the border substitution is a select afterwards
            const v2f uv_l0 = from_left2(s.uvw[s1]), uv_r0 = from_right2(s.uvw[s1]);
            const v2f duv_l0 = from_left2(s.duvw[s1]), duv_r0 = from_right2(s.duvw[s1]);
            const v2f uvL = pick2(at_l, uv_r0, uv_l0), uvR = pick2(at_r, uv_l0, uv_r0);
            const v2f duvL = pick2(at_l, duv_r0, duv_l0), duvR = pick2(at_r, duv_l0, duv_r0);
            const v2f uvU = pick2(top, s.uvw[s0], s.uvw[s2]), uvD = pick2(bot, s.uvw[s2], s.uvw[s0]);
            const v2f duvU = pick2(top, s.duvw[s0], s.duvw[s2]), duvD = pick2(bot, s.duvw[s2], s.duvw[s0]);
            xnum = diff4_num2(uvR, uvL, duvR, duvL);
            ynum = diff4_num2(uvD, uvU, duvD, duvU);
        }
        const v2f dx = spacing_quotient2<POW2, FAST>(s.guard, xnum, a.two_hx, a.inv_two_hx);
        const v2f dy = spacing_quotient2<POW2, FAST>(s.guard, ynum, a.two_hy, a.inv_two_hy);
        float t_phi = 1.f, t_ksi = 1.f;  // 2 sqrt(.) of the two robustifiers, for the guard
        float phi_arg = 0.f;
        if (FAST) {  // phi_argument (solver_math.hpp), the four squares as two packed products; the root follows with ksi's
            const v2f dx2 = dx * dx, dy2 = dy * dy;
            phi_arg = dx2.x + dy2.x + dx2.y + dy2.y + a.e_smooth * a.e_smooth;
        } else {
            s.phiw[s1] = phi_value(dx.x, dy.x, dx.y, dy.y, a.e_smooth);
        }

        const float f0c = s.f0w[s1], f1c = s.f1w[s1];
        float fx_num;
        if (!EDGE) {
            fx_num = scalar_only(scalar_only(scalar_only(from_right(f0c) - from_left(f0c)) + from_right(f1c)) - from_left(f1c));
        } else {
            const float f0l0 = from_left(f0c), f0r0 = from_right(f0c), f1l0 = from_left(f1c), f1r0 = from_right(f1c);
            const float f0L = at_l ? f0r0 : f0l0, f0R = at_r ? f0l0 : f0r0;
            const float f1L = at_l ? f1r0 : f1l0, f1R = at_r ? f1l0 : f1r0;
            fx_num = f0R - f0L + f1R - f1L;
        }
        const float f0U = top ? s.f0w[s0] : s.f0w[s2], f0D = bot ? s.f0w[s2] : s.f0w[s0];
        const float f1U = top ? s.f1w[s0] : s.f1w[s2], f1D = bot ? s.f1w[s2] : s.f1w[s0];
        {  // the two quotients side by side (the numerators' last operations write whichever registers the pair needs)
            const v2f fxy = spacing_quotient_pair<POW2, FAST>(s.guard, v2f{fx_num, f0D - f0U + f1D - f1U}, v2f{a.four_hx, a.four_hy},
                                                              v2f{a.inv_four_hx, a.inv_four_hy});
            fx = fxy.x, fy = fxy.y;
        }
        ft = f1c - f0c;
        if (FAST) {
            v2f twice;
            const v2f roots = half_inverse_root_pair(v2f{phi_arg, ksi_argument_packed(fx, fy, ft, s.duvw[s1], a.e_data)}, twice);
            s.phiw[s1] = roots.x;
            ksi = roots.y;
            t_phi = twice.x, t_ksi = twice.y;
            guard_denominators(s.guard, t_phi, t_ksi);
        } else {
            ksi = ksi_value(fx, fy, ft, s.duvw[s1].x, s.duvw[s1].y, a.e_data);
        }
        if (GRAD == 3) {
            // first derivatives of log(I + 1) with the block rule of solve_2d_log (:519-535 over the halo of :446-503)
            const bool x_lo = (x & 15) == 0, x_hi = (x & 15) == 15, y_lo = (rp & 7) == 0, y_hi = (rp & 7) == 7;
            const float l0c = s.lf0w[s1], l1c = s.lf1w[s1];
            const float l0l0 = from_left(l0c), l0r0 = from_right(l0c), l1l0 = from_left(l1c), l1r0 = from_right(l1c);
            const float l0L = x_lo ? l0c : l0l0, l0R = x_hi ? l0c : (at_r ? l0l0 : l0r0);
            const float l1L = x_lo ? l1c : l1l0, l1R = x_hi ? l1c : (at_r ? l1l0 : l1r0);
            const float l0U = y_lo ? l0c : s.lf0w[s2], l0D = y_hi ? l0c : (bot ? s.lf0w[s2] : s.lf0w[s0]);
            const float l1U = y_lo ? l1c : s.lf1w[s2], l1D = y_hi ? l1c : (bot ? s.lf1w[s2] : s.lf1w[s0]);
            s.fxw[s1] = spacing_quotient<POW2, FAST>(s.guard, l0R - l0L + l1R - l1L, a.four_hx, a.inv_four_hx);
            s.fyw[s1] = spacing_quotient<POW2, FAST>(s.guard, l0D - l0U + l1D - l1U, a.four_hy, a.inv_four_hy);
            s.ftw[s1] = l1c - l0c;
        } else if (GRAD) {
            s.fxw[s1] = fx;
            s.fyw[s1] = fy;
            s.ftw[s1] = ft;
        }
    }
    };
    stage_P();

    // ---- stage W, row rw = r-2: face weights and the motion tensor -> coefficient ring --------------------
    // phi ring: slot s1 holds row r-1 (just written), s2 row r-2, s0 row r-3
    const int rw = r - 2;
    auto stage_W = [&]() {
    if (run_W) {
        constexpr int cw = (J + 2 * kRing - 2) % kRing;
        Coef& c = s.C[cw];
        const bool top = EDGE && (rw == 0), bot = EDGE && (rw == h - 1);
        const float pc = s.phiw[s2];
        const float pl0 = from_left(pc), pr0 = from_right(pc);
        v2f p_rl;                                                                        // (phi[x+1], phi[x-1])
        float pU, pD;
        if (GRAD == 3) {  // own value at the 16x8 block edge, reflected pixel where the block leaves the image
            p_rl = v2f{(x & 15) == 15 ? pc : (at_r ? pl0 : pr0), (x & 15) == 0 ? pc : pl0};
            pU = (rw & 7) == 0 ? pc : s.phiw[s0];
            pD = (rw & 7) == 7 ? pc : (bot ? s.phiw[s0] : s.phiw[s1]);
        } else {
            p_rl = v2f{at_r ? pl0 : pr0, at_l ? pr0 : pl0};
            pU = top ? s.phiw[s1] : s.phiw[s0];
            pD = bot ? s.phiw[s0] : s.phiw[s1];
        }
        // hx_2, hy_2 hold HALF the neighbour weight alpha / h^2 here (FusedArgs::half_hx_2): face_phi (a + b) / 2.f times the weight
        // is the one rounding of (a + b) w / 2 either way -- dividing a sum of two robustifiers by two is exact (they are 0 or at
        // least 2.7e-20: 1 / (2 sqrt(FLT_MAX))), and so is halving the weight (the launcher refuses weights below 2^-100) -- so
        // (a + b) * (w / 2) has the same bits in one instruction less per face pair (round 5)
        const float yp = EDGE ? static_cast<float>(rw < h - 1) * hy_2 : hy_2;
        const float ym = EDGE ? static_cast<float>(rw > 0) * hy_2 : hy_2;
        // face_phi * (xp, xm), solve_2d.cu:337-346: xp = [x < w-1] * alpha / hx^2, xm = [x > 0] * alpha / hx^2; an interior
        // strip has no image border, so both are the uniform alpha / hx^2 there
        if (!EDGE && GRAD != 3) {  // the neighbours as DPP operands of the two additions
            c.wx = v2f{scalar_only(from_right(pc) + pc), scalar_only(from_left(pc) + pc)} * v2f{hx_2, hx_2};
        } else {
            c.wx = (p_rl + pc) * (EDGE ? v2f{at_r ? 0.f : hx_2, at_l ? 0.f : hx_2} : v2f{hx_2, hx_2});
        }
        c.wy = (v2f{pD, pU} + pc) * v2f{yp, ym};
        const float sumH = sum_weights(c.wx.x, c.wx.y, c.wy.x, c.wy.y);
        const float c_ksi = s.p_ksi;
        c.uvc = s.uvw[s2];
        v2f J11_22, c_J13_23;
        float c_J12;
        if (!GRAD) {
            const v2f fxy = v2f{s.p_fx, s.p_fy};
            J11_22 = fxy * fxy;
            c_J12 = s.p_fx * s.p_fy;
            c_J13_23 = fxy * s.p_ft;
        } else {
            // second derivatives inside the reference's 16x8 blocks, own value replicated at block and
            // image edges (solve_2d.cu:816-841,872-876); fx/fy/ft rings: s1 = row r-1, s2 = r-2, s0 = r-3
            const float fxc = s.fxw[s2], fyc = s.fyw[s2], ftc = s.ftw[s2];
            // cross-lane reads first, with every lane active; select afterwards (a DPP read under a
            // divergent branch would see disabled source lanes)
            const float fx_l0 = from_left(fxc), fx_r0 = from_right(fxc);
            const float ft_l0 = from_left(ftc), ft_r0 = from_right(ftc);
            float fx_l, fx_r, ft_l, ft_r, fx_u, fx_d, fy_u, fy_d, ft_u, ft_d;
            if (GRAD == 1 || GRAD == 3) {  // the reference's tile rule: own value at the 16x8 block edge and at the image edge
                const int tx = x & 15, ty = rw & 7;
                // (a strip without EDGE holds no pixel of the last column or row: its tile rule is the block rule alone)
                const bool x_lo = (tx == 0), x_hi = (tx == 15) || (EDGE && x == w - 1);
                const bool y_lo = (ty == 0), y_hi = (ty == 7) || (EDGE && rw == h - 1);
                fx_l = x_lo ? fxc : fx_l0, fx_r = x_hi ? fxc : fx_r0;
                ft_l = x_lo ? ftc : ft_l0, ft_r = x_hi ? ftc : ft_r0;
                fx_u = y_lo ? fxc : s.fxw[s0], fx_d = y_hi ? fxc : s.fxw[s1];
                fy_u = y_lo ? fyc : s.fyw[s0], fy_d = y_hi ? fyc : s.fyw[s1];
                ft_u = y_lo ? ftc : s.ftw[s0], ft_d = y_hi ? ftc : s.ftw[s1];
            } else {  // true neighbours, reflected at the image border like the first derivatives
                fx_l = at_l ? fx_r0 : fx_l0, fx_r = at_r ? fx_l0 : fx_r0;
                ft_l = at_l ? ft_r0 : ft_l0, ft_r = at_r ? ft_l0 : ft_r0;
                fx_u = top ? s.fxw[s1] : s.fxw[s0], fx_d = bot ? s.fxw[s0] : s.fxw[s1];
                fy_u = top ? s.fyw[s1] : s.fyw[s0], fy_d = bot ? s.fyw[s0] : s.fyw[s1];
                ft_u = top ? s.ftw[s1] : s.ftw[s0], ft_d = bot ? s.ftw[s0] : s.ftw[s1];
            }
            // float(1.0 / (2.0 * h)): double, rounded to float (solve_2d.cu:868-869)
            const v2f hxy_1 = v2f{a.hx_1, a.hy_1};
            // The second derivatives as pairs -- the selected neighbours land in whichever registers the pairs need -- and
            // gradient_tensor (solver_math.hpp) on them: A = (fxx, fxy), B = (fxy, fyy), Ft = (fxt, fyt);
            // (J11, J22) = A A + B B, J12 = A.x B.x + A.y B.y, (J13, J23) = Ft.x A + Ft.y B: the scalar form's products and sums
            // in its order, 15 instructions instead of 24.
            const v2f A = (v2f{fx_r, fx_d} - v2f{fx_l, fx_u}) * hxy_1;
            const v2f Ft = (v2f{ft_r, ft_d} - v2f{ft_l, ft_u}) * hxy_1;
            const float fyy = (fy_d - fy_u) * a.hy_1;
            const v2f B = v2f{A.y, fyy};
            J11_22 = A * A + B * B;
            const v2f AB = A * B;
            c_J12 = AB.x + AB.y;
            c_J13_23 = mul_by_x(Ft, A) + mul_by_y(Ft, B);
        }
        const v2f c_den = c_ksi * J11_22 + sumH;  // update_denominator for u and v
        v2f c_rden = v2f{0.f, 0.f};
        if (FAST) {
            c_rden = rcp3_pair(c_den);
            guard_denominators(s.guard, c_den.x, c_den.y);
        }
        c.den = c_den;
        c.rden = c_rden;
        c.J13_23 = c_J13_23;
        c.ksi = c_ksi;
        c.J12 = c_J12;
    }
    };
    // (u + du, v + dv) of row r-2 enters sweep 1's window
    if (run_W) s.UV[0][s2] = s.uvw[s2] + (cont ? s.start_cur : s.duvw[s2]);
    stage_W();
    // stage P's outputs of this step are what stage W consumes in the next one
    s.p_fx = fx;
    s.p_fy = fy;
    s.p_ft = ft;
    s.p_ksi = ksi;

    // ---- sweeps k = 1..INNER, row rk = r-2-k (solve_2d.cu:349-367) ------------------------------------------
    v2f old = duv_row3;  // (du^0, dv^0) of row r-3
#pragma unroll
    for (int k = 1; k <= INNER; ++k) {
        if (T >= 0 && T < 3 + 2 * k) continue;  // start-up: this sweep's row feeds nothing yet
        const int rk = r - 2 - k;
        // window slots of rows rk-1, rk, rk+1 (rk = r-2-k  ->  slot (J - 2 - k) mod 3)
        const int sc = (J + 3 * 8 - 2 - k) % 3, su = (sc + 2) % 3, sd = (sc + 1) % 3;
        const int ck = (J + 4 * kRing - 2 - k) % kRing;  // a constant once the sweep loop is unrolled
        const Coef& c = s.C[ck];
        const v2f den = c.den, rden = c.rden, J13_23 = c.J13_23;
        const float ksi = c.ksi, J12 = c.J12;
        const bool top = EDGE && (rk == 0), bot = EDGE && (rk == h - 1);
        const v2f n_c = s.UV[k - 1][sc], centre = c.uvc;
        // neighbour minus centre, component by component for the x neighbours: a scalar subtraction takes the lane shift
        // as a DPP operand (v_sub_f32_dpp), a packed one needs the shifted pair assembled by two v_mov_b32_dpp first.
        // The border substitutions select among the differences -- the same values as differences of the selected.
        const v2f d_l0 = v2f{scalar_only(from_left(n_c.x) - centre.x), scalar_only(from_left(n_c.y) - centre.y)};
        const v2f d_r0 = v2f{scalar_only(from_right(n_c.x) - centre.x), scalar_only(from_right(n_c.y) - centre.y)};
        const v2f d_u0 = s.UV[k - 1][su] - centre, d_d0 = s.UV[k - 1][sd] - centre;
        v2f dL, dR, dU, dD;
        if (GRAD == 3) {  // solve_2d_log: the flow neighbours follow the block rule too (:612-633)
            const v2f d_c = n_c - centre;
            dL = pick2((x & 15) == 0, d_c, d_l0);
            dR = pick2((x & 15) == 15, d_c, pick2(at_r, d_l0, d_r0));
            dU = pick2((rk & 7) == 0, d_c, d_u0);
            dD = pick2((rk & 7) == 7, d_c, pick2(bot, d_u0, d_d0));
        } else {
            dL = pick2(at_l, d_r0, d_l0), dR = pick2(at_r, d_l0, d_r0);
            dU = pick2(top, d_d0, d_u0), dD = pick2(bot, d_u0, d_d0);
        }
        const v2f sums = flux_of_differences(c.wx, c.wy, dR, dL, dD, dU);  // (sumU, sumV)
        const float dv_in = old.y;
        float du_new, dv_new;
        if (SOR) {  // point_update_sor (solver_math.hpp): the Gauss-Seidel values blended with the old ones, then the colour select
            const float nu = ksi * (-J13_23.x - J12 * dv_in) + sums.x;
            const float gs_du = FAST ? div3(nu, den.x, rden.x) : nu / den.x;
            du_new = a.sor_keep * old.x + a.sor_omega * gs_du;
            const float nv = ksi * (-J13_23.y - J12 * du_new) + sums.y;
            const float gs_dv = FAST ? div3(nv, den.y, rden.y) : nv / den.y;
            dv_new = a.sor_keep * old.y + a.sor_omega * gs_dv;
            const bool mine = ((x + rk + k - 1) & 1) == 0;  // half-sweep k relaxes colour (k - 1) % 2
            du_new = mine ? du_new : old.x;
            dv_new = mine ? dv_new : old.y;
            if (FAST) {
                guard_numerators(s.guard, nu, nv);
                if (k == INNER) guard_results(s.guard, du_new, dv_new);
            }
        } else if (FAST) {  // the coupled 2x2 update of solve_2d.cu:361-367 (point_update) with the three-step division
            const float nu = ksi * (-J13_23.x - J12 * dv_in) + sums.x;
            du_new = div3(nu, den.x, rden.x);
            const float nv = ksi * (-J13_23.y - J12 * du_new) + sums.y;
            dv_new = div3(nv, den.y, rden.y);
            guard_numerators(s.guard, nu, nv);
            if (k == INNER) guard_results(s.guard, du_new, dv_new);
        } else {
            point_update(ksi, den.x, den.y, J12, J13_23.x, J13_23.y, sums.x, sums.y, dv_in, du_new, dv_new);
        }
        if (k < INNER) {
            s.UV[k][sc] = c.uvc + v2f{du_new, dv_new};
            old = s.duvc[k];                    // (du^k, dv^k) of row r-3-k, produced by this sweep one step ago
            s.duvc[k] = v2f{du_new, dv_new};    // of row r-2-k, for the next step
        } else if (lane_stores) {  // (rk is in [y0, y1) in every step that gets here: run_strip's start-up and r_last)
            store_row(a, row_offset(a, probe::kComputeOnly ? rk & 7 : rk, xc), du_new, dv_new);
        }
    }
}

// one turn of the ring starting at ring position J0 (the start-up ends there)
template <int INNER, int GRAD, bool EDGE, bool POW2, bool CONT, bool FAST, bool SOR, int J0, size_t... Is>
__device__ __forceinline__ void strip_steps(Strip<INNER, GRAD>& s, const FusedArgs& a, int r_base, int x, int xc,
                                            bool at_l, bool at_r, bool lane_stores, int y0, int y1, 
                                            float hx_2, float hy_2, std::index_sequence<Is...>)
{
    constexpr int kRing = Strip<INNER, GRAD>::kRing;
    (strip_step<INNER, GRAD, EDGE, POW2, CONT, FAST, SOR, (J0 + static_cast<int>(Is)) % kRing>(s, a, r_base + static_cast<int>(Is), x, xc, at_l,
                                                                                    at_r, lane_stores, y0, y1, hx_2, hy_2),
     ...);
}

// the last, partial turn of the ring: the steps up to r_last only (wave-uniform guards)
template <int INNER, int GRAD, bool EDGE, bool POW2, bool CONT, bool FAST, bool SOR, int J0, size_t... Is>
__device__ __forceinline__ void strip_tail(Strip<INNER, GRAD>& s, const FusedArgs& a, int r_base, int r_last, int x, int xc,
                                           bool at_l, bool at_r, bool lane_stores, int y0, int y1, float hx_2,
                                           float hy_2, std::index_sequence<Is...>)
{
    constexpr int kRing = Strip<INNER, GRAD>::kRing;
    ((r_base + static_cast<int>(Is) <= r_last
          ? strip_step<INNER, GRAD, EDGE, POW2, CONT, FAST, SOR, (J0 + static_cast<int>(Is)) % kRing>(s, a, r_base + static_cast<int>(Is), x, xc,
                                                                                               at_l, at_r, lane_stores, y0, y1, hx_2,
                                                                                               hy_2)
          : (void)0),
     ...);
}

template <int INNER, int GRAD, bool EDGE, bool POW2, bool CONT, bool FAST, bool SOR, size_t... Ts>
__device__ __forceinline__ void strip_startup(Strip<INNER, GRAD>& s, const FusedArgs& a, int r_first, int x, int xc,
                                              bool at_l, bool at_r, bool lane_stores, int y0, int y1, 
                                              float hx_2, float hy_2, std::index_sequence<Ts...>)
{
    constexpr int kRing = Strip<INNER, GRAD>::kRing;
    (strip_step<INNER, GRAD, EDGE, POW2, CONT, FAST, SOR, static_cast<int>(Ts) % kRing, static_cast<int>(Ts)>(
         s, a, r_first + static_cast<int>(Ts), x, xc, at_l, at_r, lane_stores, y0, y1, hx_2, hy_2),
     ...);
}

// One trip of a wave down its strip: state set-up, the peeled start-up steps, the row loop, the partial last ring turn.
// Returns whether any lane met operands the three-step division is not proven for (always false with FAST = false).
template <int INNER, int GRAD, bool EDGE, bool POW2, bool CONT, bool FAST, bool SOR>
__device__ __forceinline__ bool run_strip(const FusedArgs& a, int x, int xc, bool at_l, bool at_r, bool lane_stores, int y0,
                                          int y1, float hx_2, float hy_2, unsigned* stall_out = nullptr)
{
    using S = Strip<INNER, GRAD>;
    S s;
    s.stall[0] = s.stall[1] = s.stall[2] = 0u;
    s.stall_t0 = s.stall_t1 = 0ull;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        s.f0w[i] = s.f1w[i] = s.phiw[i] = 0.f;
        s.uvw[i] = s.duvw[i] = v2f{0.f, 0.f};
        s.fxw[i] = s.fyw[i] = s.ftw[i] = 0.f;
        s.lf0w[i] = s.lf1w[i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < INNER; ++k) {
        s.duvc[k] = v2f{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) s.UV[k][i] = v2f{0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < S::kRing; ++i) {
        s.C[i] = Coef{};
        s.C[i].den = s.C[i].rden = v2f{1.f, 1.f};
    }
    s.p_fx = s.p_fy = s.p_ft = s.p_ksi = 0.f;
    s.guard = DivGuard{0xffffffffu, 0u, 1.f, 1.f, 0x7fffffff};

    // first input row: the strip's first stored row needs INNER+1 rows of halo above it
    const int r_first = y0 - S::kHalo;
    {
        s.n = fetch_row(a, row_offset(a, min(max(r_first, 0), a.h - 1), xc));
        s.m = fetch_row(a, row_offset(a, min(max(r_first + 1, 0), a.h - 1), xc));
        s.start_cur = s.n_start = v2f{0.f, 0.f};
        if (CONT) {  // the first step commits row r_first - 2 of the starting increment
            const size_t os = static_cast<size_t>(min(max(r_first - 2, 0), a.h - 1)) * a.pitch + xc;
            s.n_start = v2f{a.start_du[os], a.start_dv[os]};
        }
    }
    // the last stored row y1-1 leaves the last sweep at input row (y1-1) + 2 + INNER
    const int r_last = y1 - 1 + 2 + INNER;
    // start-up steps, then the row loop.  The start-up is exactly the 2 INNER + 3 steps before the last sweep first produces a row
    // anybody stores (row y0 leaves it at input row y0 + 2 + INNER = r_first + 2 INNER + 3), so EVERY later step stores a row of
    // [y0, y1): the steady-state steps carry no row test (five scalar instructions per step, each an issue turn of the wave).  The
    // row loop therefore starts at ring position kPeel % kRing, not 0.
    constexpr int kPeel = 2 * INNER + 3, kJ0 = kPeel % S::kRing;
    strip_startup<INNER, GRAD, EDGE, POW2, CONT, FAST, SOR>(s, a, r_first, x, xc, at_l, at_r, lane_stores, y0, y1, hx_2, hy_2,
                                                       std::make_index_sequence<kPeel>{});
    int r = r_first + kPeel;
    for (; r + S::kRing - 1 <= r_last; r += S::kRing) {
        strip_steps<INNER, GRAD, EDGE, POW2, CONT, FAST, SOR, kJ0>(s, a, r, x, xc, at_l, at_r, lane_stores, y0, y1, hx_2, hy_2,
                                                              std::make_index_sequence<S::kRing>{});
    }
    strip_tail<INNER, GRAD, EDGE, POW2, CONT, FAST, SOR, kJ0>(s, a, r, r_last, x, xc, at_l, at_r, lane_stores, y0, y1, hx_2, hy_2,
                                                         std::make_index_sequence<S::kRing - 1>{});
    if constexpr (probe::kStamps)
        if (stall_out) stall_out[0] = s.stall[0], stall_out[1] = s.stall[1], stall_out[2] = s.stall[2];
    return guard_tripped(s.guard);
}

template <int INNER, int GRAD, bool POW2, bool CONT, bool SOR = false>
__global__ __launch_bounds__(256, 2) void fused_outer_kernel(FusedArgs a)
{
    using S = Strip<INNER, GRAD>;
    const int lane = threadIdx.x & 63;
    const unsigned long long stamp_r0 = probe::kStamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long stamp_c0 = probe::kStamps ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned stall[3] = {0u, 0u, 0u};
    // launch block id -> block column bx and strip by of the plan (solve_fused_args.hpp: every XCD a contiguous, equally heavy run)
    int bx, by;
    const bool uniform = a.rows_interior == a.rows_edge;
    if (!fused_block_of(a, static_cast<int>(blockIdx.x), bx, by)) return;
    const int strip_x = bx * 4 + (threadIdx.x >> 6);
    if (strip_x * S::kValid >= a.w) return;  // whole wave leaves; waves never synchronise with each other
    {  // instance of a batched launch
        const size_t off = static_cast<size_t>(blockIdx.z) * static_cast<size_t>(a.batch_stride);
        a.f0 += off, a.f1 += off, a.u += off, a.v += off, a.du += off, a.dv += off, a.out_du += off, a.out_dv += off;
        if (CONT) a.start_du += off, a.start_dv += off;
    }
    const int x = strip_x * S::kValid - S::kHaloLanes + lane;
    const int xc = min(max(x, 0), a.w - 1);
    int y0, y1;
    if (uniform || bx == 0 || bx == a.blocks_x - 1) {
        y0 = by * a.rows_edge;
        y1 = min(y0 + a.rows_edge, a.h);
    } else if (by == 0) {
        y0 = 0, y1 = a.rows_edge;
    } else if (by == a.strips_interior - 1) {
        y0 = a.h - a.rows_edge, y1 = a.h;
    } else {
        y0 = a.rows_edge + (by - 1) * a.rows_interior;
        y1 = min(y0 + a.rows_interior, a.h - a.rows_edge);
    }
    if (y0 >= y1) return;  // (a middle strip the rounding of rows_interior left empty)
    const bool at_l = (x == 0), at_r = (x == a.w - 1);
    const bool lane_stores = lane >= S::kHaloLanes && lane < 64 - S::kHaloLanes && x < a.w;
    const float hx_2 = a.half_hx_2, hy_2 = a.half_hy_2;  // HALF the neighbour weights: see stage W

    // does any row or column this wave touches sit on an image border?  (a superset test is fine)
    const int x_first = strip_x * S::kValid - S::kHaloLanes;
    // (rows: the strip reads y0 - kHalo .. y1 + kHalo + 1 + rows in flight; the interior body fetches them unclamped)
    const bool edge = x_first <= 0 || x_first + 63 >= a.w - 1 || y0 <= S::kHalo + 1 || y1 + S::kHalo + 4 >= a.h;
    bool bad = a.plain_only != 0;
    if (bad)
        ;
    else if (__builtin_amdgcn_readfirstlane(edge))
        bad = run_strip<INNER, GRAD, true, POW2, CONT, true, SOR>(a, x, xc, at_l, at_r, lane_stores, y0, y1, hx_2, hy_2, stall);
    else
        bad = run_strip<INNER, GRAD, false, POW2, CONT, true, SOR>(a, x, xc, at_l, at_r, lane_stores, y0, y1, hx_2, hy_2, stall);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
        // some lane's operands left the range the three-step division is proven for (or the launch's grid spacing did:
        // plain_only): the whole strip with the plain division (same stores, now from the reference's own arithmetic)
        (void)run_strip<INNER, GRAD, true, POW2, CONT, false, SOR>(a, x, xc, at_l, at_r, lane_stores, y0, y1, hx_2, hy_2);
        // word 0 counts guard trips, word 1 the waves of launches that never tried the short forms
        if (a.fallback_count && lane == 0) atomicAdd(a.fallback_count + (a.plain_only ? 1 : 0), 1u);
    }
    if constexpr (probe::kStamps) {  // the wave's stamp (probe::kStampWords words) and its stall histogram (probe::kStallRows x 64)
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
        unsigned slot = 0u;
        if (lane == 0) slot = atomicAdd(a.stamp_count, 1u) % probe::kStampWaves;
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (lane == 0) {
            unsigned long long* o = a.stamps + static_cast<size_t>(slot) * probe::kStampWords;
            o[0] = stamp_r0, o[1] = r1, o[2] = c1 - stamp_c0;
            unsigned hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            o[3] = hw_id, o[4] = xcc_id;
            o[5] = blockIdx.x | (static_cast<unsigned long long>(blockIdx.z) << 32);
            o[6] = (threadIdx.x >> 6) | (static_cast<unsigned>(edge) << 8) | (static_cast<unsigned long long>(strip_x) << 32);
            o[7] = static_cast<unsigned>(y0) | (static_cast<unsigned long long>(y1) << 32);
        }
        if (slot < probe::kStallWaves)
            for (int i = 0; i < probe::kStallRows; ++i) a.stalls[(static_cast<size_t>(slot) * probe::kStallRows + i) * 64 + lane] = stall[i];
    }
}

#define FUSED_LAUNCH(N)                                                                 \
    do {                                                                                \
        fused_outer_kernel<N, GRAD, POW2, CONT><<<grid, 256, 0, stream>>>(a); \
        return 0;                                                                       \
    } while (0)

#define FUSED_LAUNCH_SOR(N)                                                                   \
    do {                                                                                      \
        fused_outer_kernel<N, GRAD, POW2, CONT, true><<<grid, 256, 0, stream>>>(a); \
        return 0;                                                                             \
    } while (0)

template <int GRAD, bool POW2, bool CONT>
int launch_for_inner_cont(int inner, dim3 grid, hipStream_t stream, const FusedArgs& a)
{
    if constexpr (probe::kDevInstances) {  // developer builds: only the instantiations of the 4096^2 benchmark, half a minute
        if constexpr (GRAD <= 1 && POW2 && !CONT) {
            if (inner == 5) FUSED_LAUNCH(5);
            if (inner == 2) FUSED_LAUNCH(2);
        }
        return 1;
    } else {
    if (a.sor_omega != 0.f) {  // red-black half-sweeps: 2 or 4 stages = one or two iterations per launch (not for solve_2d_log)
        if constexpr (GRAD != 3) {
            if (inner == 2) FUSED_LAUNCH_SOR(2);
            if (inner == 4) FUSED_LAUNCH_SOR(4);
        }
        return 1;
    }
    switch (inner) {
        case 1: FUSED_LAUNCH(1);
        case 2: FUSED_LAUNCH(2);
        case 3: FUSED_LAUNCH(3);
        case 4: FUSED_LAUNCH(4);
        case 5: FUSED_LAUNCH(5);
        default: return 1;
    }
    }
}

template <int GRAD, bool POW2>
int launch_for_inner(int inner, dim3 grid, hipStream_t stream, const FusedArgs& a)
{
    return a.continue_sweeps ? launch_for_inner_cont<GRAD, POW2, true>(inner, grid, stream, a)
                             : launch_for_inner_cont<GRAD, POW2, false>(inner, grid, stream, a);
}

}  // namespace
