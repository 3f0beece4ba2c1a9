// Proof by exhaustion for the three-instruction division of the fused solver kernel (solve_fused_kernel.hpp):
//
//     y  = RN(1 / d)                 once per pixel and outer iteration (a true, correctly rounded division)
//     q0 = RN(n * y)
//     r  = RN(n - q0 * d)            one fma
//     q  = RN(q0 + r * y)            one fma
//
// Claim checked here: q == RN(n / d), bit for bit, for EVERY pair of fp32 significands (2^23 x 2^23 pairs, n and d in
// [1, 2)).  All four steps commute exactly with scaling n or d by a power of two as long as nothing leaves the normal
// range, so the significand check covers every normal (n, d) whose q0, r and q stay normal or are exactly zero -- which
// is what the kernel's guard establishes at run time: d within [2^-30, 2^40], n zero or at least 2^-80 in magnitude
// (then q0 >= 2^-121 is normal and r, a multiple of 2^-128 at least, is exact), the result finite.  The second kernel
// applies that very guard (the kernel's own integer test on the bit patterns) to random operands of every magnitude,
// tiny, denormal and huge numerators included: whatever passes the guard must equal n / d.
// Two cheaper guards were tried and are NOT sufficient: the class of r alone (r == 0 is ambiguous: a true zero, or a
// residual below the denormal grid rounded away -- 8.3e6 wrong quotients of 1.7e10 passed) and the wave's sticky IEEE
// exception flags (TRAPSTS.EXCP stays 0 on gfx950 unless traps are enabled).
//
// (Markstein's theorem gives q == RN(n / d) when q0 is a faithful rounding of n / d; RN(n * RN(1 / d)) can be 1.5 ulp
// off when n < d, so the theorem alone does not cover this sequence.)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/markstein_exhaustive.hip -o gpurun_out/markstein
// Run:   gpurun_out/markstein [first_chunk last_chunk]     (64 chunks of 2^17 divisors; all of them: about a minute)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Report {
    unsigned long long mismatches;
    unsigned long long unfaithful_q0;  // |q0 - RN(n/d)| > 1 ulp: the premise of Markstein's theorem fails
    unsigned long long guard_trips;    // kernel 2: pairs the guard sends to the fallback
    unsigned long long guard_misses;   // kernel 2: guard passed, result wrong
    unsigned int listed;
    unsigned int list[256][2];  // first mismatching (n bits, d bits)
};

__device__ __forceinline__ float three_step(float n, float d, float y, float* r_out)
{
    const float q0 = n * y;
    const float r = __builtin_fmaf(-q0, d, n);
    *r_out = r;
    return __builtin_fmaf(r, y, q0);
}

__global__ __launch_bounds__(256) void all_significands(unsigned d_first, Report* rep)
{
    const unsigned dsig = d_first + blockIdx.x * blockDim.x + threadIdx.x;
    const float d = __uint_as_float(0x3f800000u | dsig);
    const float y = 1.0f / d;
    unsigned long long bad = 0, unfaithful = 0;
    for (unsigned nsig = 0; nsig < (1u << 23); ++nsig) {
        const float n = __uint_as_float(0x3f800000u | nsig);
        const float want = n / d;
        float r;
        const float got = three_step(n, d, y, &r);
        const int off = (int)__float_as_uint(n * y) - (int)__float_as_uint(want);
        unfaithful += (off > 1 || off < -1);
        if (__float_as_uint(got) != __float_as_uint(want)) {
            ++bad;
            const unsigned slot = atomicAdd(&rep->listed, 1u);
            if (slot < 256) {
                rep->list[slot][0] = __float_as_uint(n);
                rep->list[slot][1] = __float_as_uint(d);
            }
        }
    }
    if (bad) atomicAdd(&rep->mismatches, bad);
    if (unfaithful) atomicAdd(&rep->unfaithful_q0, unfaithful);
}

__device__ __forceinline__ unsigned long long lcg(unsigned long long& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return s;
}

// the run-time guard of the kernel (solve_fused_kernel.hpp, DivGuard): tiny numerators by an integer test on the bit pattern
// -- the shift drops the sign, the decrement sends a zero to the top -- and non-finite results
__device__ __forceinline__ bool guard_ok(float n, float q)
{
    const unsigned tiny = (__float_as_uint(n) << 1) - 1u;   // kTinyLimit = 2 * bits(2^-80) - 1
    const unsigned out = __float_as_uint(q) << 1;           // kOutLimit  = bits(FLT_MAX) << 1
    return !(tiny < 2u * 0x17800000u - 1u) && !(out > 0xfefffffeu);
}

__global__ __launch_bounds__(256) void random_magnitudes(unsigned long long seed, int per_thread, Report* rep)
{
    unsigned long long s = seed + 0x9e3779b97f4a7c15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long trips = 0, misses = 0;
    for (int i = 0; i < per_thread; ++i) {
        const unsigned long long a = lcg(s), b = lcg(s);
        // n: any finite float, either sign, zeros and denormals included; d: positive, exponent in [-30, 40]
        const float n = __uint_as_float((unsigned)(a >> 32));
        const unsigned dexp = 127u - 30u + (unsigned)((b >> 40) % 71u);
        const float d = __uint_as_float((dexp << 23) | ((unsigned)(b >> 8) & 0x7fffffu));
        if (__builtin_isnan(n) || __builtin_isinf(n)) continue;
        const float y = 1.0f / d;
        float r;
        const float got = three_step(n, d, y, &r);
        const float want = n / d;
        // (a numerator of exactly -0 gives +0 where n / d gives -0: the kernel keeps -0 out of its numerators by sending
        //  any flow plane that holds a -0 to the fallback, see guard_flow_row; it is excluded here)
        if (!guard_ok(n, got) || __float_as_uint(n) == 0x80000000u) {
            ++trips;
        } else if (__float_as_uint(got) != __float_as_uint(want)) {  // bit for bit, the sign of a zero included
            ++misses;
            const unsigned slot = atomicAdd(&rep->listed, 1u);
            if (slot < 256) {
                rep->list[slot][0] = __float_as_uint(n);
                rep->list[slot][1] = __float_as_uint(d);
            }
        }
    }
    if (trips) atomicAdd(&rep->guard_trips, trips);
    if (misses) atomicAdd(&rep->guard_misses, misses);
}

int main(int argc, char** argv)
{
    const int first = argc > 2 ? std::atoi(argv[1]) : 0, last = argc > 2 ? std::atoi(argv[2]) : 63;
    Report* dev;
    Report host;
    std::memset(&host, 0, sizeof(host));
    if (hipMalloc(&dev, sizeof(Report)) != hipSuccess) return 2;
    hipMemcpy(dev, &host, sizeof(host), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int chunk = first; chunk <= last; ++chunk) {  // 2^17 divisors per launch, 2^23 numerators each
        all_significands<<<(1u << 17) / 256, 256>>>((unsigned)chunk << 17, dev);
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
        std::printf("chunk %2d of 64: divisor significands [%#x, %#x): mismatches so far %llu, q0 more than 1 ulp off %llu\n",
                    chunk, (unsigned)chunk << 17, (unsigned)(chunk + 1) << 17, host.mismatches, host.unfaithful_q0);
        std::fflush(stdout);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double pairs = (double)(last - first + 1) * (1u << 17) * (double)(1u << 23);
    std::printf("significand pairs checked: %.4g in %.1f s; three-step quotient != n / d: %llu; RN(n * RN(1/d)) more than 1 ulp "
                "off: %llu\n", pairs, ms * 1e-3, host.mismatches, host.unfaithful_q0);
    for (unsigned i = 0; i < host.listed && i < 16; ++i)
        std::printf("  mismatch: n = %#010x  d = %#010x\n", host.list[i][0], host.list[i][1]);
    const unsigned long long sig_mismatches = host.mismatches;

    host.listed = 0;
    hipMemcpy(dev, &host, sizeof(host), hipMemcpyHostToDevice);
    const int blocks = 256 * 16, per_thread = 4096;
    for (int pass = 0; pass < 4; ++pass) random_magnitudes<<<blocks, 256>>>(12345ull + pass, per_thread, dev);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
    std::printf("random magnitudes: %.4g pairs (n any finite float bit pattern, d in [2^-30, 2^41)): sent to the fallback by the "
                "guard %llu, passed the guard and differ from n / d: %llu\n", 4.0 * blocks * 256 * per_thread, host.guard_trips,
                host.guard_misses);
    for (unsigned i = 0; i < host.listed && i < 16; ++i)
        std::printf("  guard miss: n = %#010x  d = %#010x\n", host.list[i][0], host.list[i][1]);
    return (sig_mismatches || host.guard_misses) ? 1 : 0;
}
