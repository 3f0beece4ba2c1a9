// Developer probes of the fused strip kernel (solve_fused_kernel.hpp), in one place.  They exist in developer builds only
// (-DFLOW2D_DEV_BUILD, built into ab/*.so and loaded through FLOW2D_HIP_LIB; tools/ab_time.sh, tools/ab_bench.sh); the product
// library has none of them and reads no environment variable.  The kernel tests them as compile-time constants (`if constexpr`),
// so a probe that is off leaves no instruction and no register behind.
//   FLOW2D_FUSED_DEV            only the instantiations of the 4096^2 benchmark (inner 5 and 2, power-of-two spacing): half a minute
//   FLOW2D_FUSED_STAMPS         per-wave time stamps: start, end, shader cycles, where it ran (tools/fused_wave_stamps.py)
//   FLOW2D_FUSED_STALLS         (with STAMPS) ... and the histogram of each wave's waits at the row commit, by duration and by the
//                               wave's progress; about twenty more instructions per row step (+3-5 % on the launch)
//   FLOW2D_FUSED_COMPUTE_ONLY   timing probe, WRONG results: every row folded onto eight cache-resident rows
//   FLOW2D_FUSED_MEMORY_ONLY    timing probe, WRONG results: the strip's loads and stores without its arithmetic
//   FLOW2D_FUSED_NO_HALO        timing probe, WRONG results at strip edges: no halo lanes -- 64 stored columns per wave, the bound
//                               of any design that exchanges edge columns between waves instead of recomputing them
//   FLOW2D_FUSED_EXCHANGE=n     (with NO_HALO) ... plus what such an exchange would cost: one s_barrier and n 8-byte LDS writes + n
//                               two-address LDS reads per row step (0: the barrier alone)
//   FLOW2D_FUSED_SIDE_LAST      A/B: the side blocks of a border-aware plan all in the last XCD's run, as in rounds 3-5
//   FLOW2D_FUSED_PACKED_PLANES  f0, f1, u, v read as ONE float4 plane and du, dv as one float2 plane in, one out (the launcher
//                               packs and unpacks around the launch): 8 -> 3 vector-memory instructions per row step
// Settled questions of rounds 2-5 (short ring, no lane shifts, three rows in flight, stagger, non-temporal accesses, register
// budgets, stage order, instruction injection, priority turns, full weights, plain division, block order, waves per SIMD) are
// no longer switches: their results are in profiles/r0N_experiments/.
#pragma once

#if (defined(FLOW2D_FUSED_DEV) || defined(FLOW2D_FUSED_STAMPS) || defined(FLOW2D_FUSED_STALLS) || defined(FLOW2D_FUSED_COMPUTE_ONLY) || defined(FLOW2D_FUSED_MEMORY_ONLY) || \
     defined(FLOW2D_FUSED_NO_HALO) || defined(FLOW2D_FUSED_EXCHANGE) || defined(FLOW2D_FUSED_PACKED_PLANES) || defined(FLOW2D_FUSED_SIDE_LAST)) && !defined(FLOW2D_DEV_BUILD)
#error "the fused kernel's probes need -DFLOW2D_DEV_BUILD: they are not part of the product library"
#endif

namespace flow2d_probe {

#ifdef FLOW2D_FUSED_DEV
constexpr bool kDevInstances = true;
#else
constexpr bool kDevInstances = false;
#endif
#ifdef FLOW2D_FUSED_STAMPS
constexpr bool kStamps = true;
#else
constexpr bool kStamps = false;
#endif
#ifdef FLOW2D_FUSED_STALLS
constexpr bool kStalls = true;
#else
constexpr bool kStalls = false;
#endif
#ifdef FLOW2D_FUSED_COMPUTE_ONLY
constexpr bool kComputeOnly = true;
#else
constexpr bool kComputeOnly = false;
#endif
#ifdef FLOW2D_FUSED_MEMORY_ONLY
constexpr bool kMemoryOnly = true;
#else
constexpr bool kMemoryOnly = false;
#endif
#ifdef FLOW2D_FUSED_NO_HALO
constexpr bool kNoHalo = true;
#else
constexpr bool kNoHalo = false;
#endif
#ifdef FLOW2D_FUSED_EXCHANGE
constexpr int kExchange = FLOW2D_FUSED_EXCHANGE;
#else
constexpr int kExchange = -1;  // no exchange probe (0: the barrier alone)
#endif
#ifdef FLOW2D_FUSED_SIDE_LAST
constexpr bool kSideLast = true;
#else
constexpr bool kSideLast = false;
#endif
#ifdef FLOW2D_FUSED_PACKED_PLANES
constexpr bool kPackedPlanes = true;
#else
constexpr bool kPackedPlanes = false;
#endif

// per wave: start / end on the 100 MHz clock, shader cycles, HW_ID, XCC_ID, block id, wave in block | edge << 8, y0 | y1 << 32
constexpr int kStampWords = 8, kStampWaves = 1 << 16;
// per wave, 64 lanes each: stalls at the row commit by floor(log2(cycles)) -- how many, and their cycles --, and the stall
// cycles by the wave's progress (four row steps per lane)
constexpr int kStallRows = 3, kStallWaves = 1 << 12;

}  // namespace flow2d_probe
