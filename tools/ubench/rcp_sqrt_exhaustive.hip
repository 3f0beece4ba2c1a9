// Proof by exhaustion for the short forms of the coefficient stage of the fused solver kernel (solve_fused_kernel.hpp):
//
//   reciprocal   y0 = v_rcp_f32(d);  e = fma(-d, y0, 1);  y = fma(e, y0, y0)             == 1.0f / d ?
//   square root  g0 = v_sqrt_f32(s); h = 0.5f * v_rsq_f32(s); r = fma(-g0, g0, s); g = fma(r, h, g0)   == sqrtf(s) ?
//   phi / ksi    t = 2 * g;  e = fma(-t, h, 1);  p = fma(e, h, h)                         == 1.f / (2.f * sqrtf(s)) ?
//
// These are functions of ONE float, so every bit pattern of the guarded range is tried (not only the significands): the
// hardware approximations are free to depend on the exponent.  Each candidate is compared bit for bit with the
// compiler's correctly rounded expression built with the library's flags (-O3 -ffp-contract=off -fno-fast-math).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/ubench/rcp_sqrt_exhaustive.hip -o gpurun_out/rcp_sqrt
// Run:   gpurun_out/rcp_sqrt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

constexpr int kCandidates = 12;
struct Report {
    unsigned long long mismatches[kCandidates];
    unsigned lowest[kCandidates], highest[kCandidates];  // bit patterns of the smallest / largest argument that differs
};

__device__ __forceinline__ void note(Report* rep, int c, unsigned bits, bool bad)
{
    if (bad) {
        atomicAdd(&rep->mismatches[c], 1ull);
        atomicMin(&rep->lowest[c], bits);
        atomicMax(&rep->highest[c], bits);
    }
}

__device__ __forceinline__ bool differ(float a, float b) { return __float_as_uint(a) != __float_as_uint(b); }

// [first, last] bit patterns, one per thread
__global__ __launch_bounds__(256) void reciprocals(unsigned first, unsigned last, Report* rep)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + first;
    if (i > last) return;
    const float d = __uint_as_float((unsigned)i);
    const float want = 1.0f / d;
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    note(rep, 0, (unsigned)i, differ(y1, want));
    const float e2 = __builtin_fmaf(-d, y1, 1.0f);
    const float y2 = __builtin_fmaf(e2, y1, y1);
    note(rep, 1, (unsigned)i, differ(y2, want));
    note(rep, 2, (unsigned)i, differ(y0, want));  // how often the bare approximation is already right (for information)
}

__global__ __launch_bounds__(256) void roots(unsigned first, unsigned last, Report* rep)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + first;
    if (i > last) return;
    const float s = __uint_as_float((unsigned)i);
    const float want = sqrtf(s);
    const float want_p = 1.f / (2.f * want);
    // S1: hardware root, one residual step through half the hardware reciprocal root
    const float g0 = __builtin_amdgcn_sqrtf(s);
    const float h = 0.5f * __builtin_amdgcn_rsqf(s);
    const float r = __builtin_fmaf(-g0, g0, s);
    const float g1 = __builtin_fmaf(r, h, g0);
    note(rep, 3, (unsigned)i, differ(g1, want));
    // S2: reciprocal root only (one transcendental), coupled iteration
    {
        const float r0 = __builtin_amdgcn_rsqf(s);
        float g = s * r0, hh = 0.5f * r0;
        const float e = __builtin_fmaf(-hh, g, 0.5f);
        g = __builtin_fmaf(g, e, g);
        hh = __builtin_fmaf(hh, e, hh);
        const float d = __builtin_fmaf(-g, g, s);
        g = __builtin_fmaf(d, hh, g);
        note(rep, 4, (unsigned)i, differ(g, want));
        // P3: 1 / (2 g) from the refined hh of the same iteration
        const float t = 2.f * g;
        const float e1 = __builtin_fmaf(-t, hh, 1.0f);
        const float p = __builtin_fmaf(e1, hh, hh);
        note(rep, 7, (unsigned)i, differ(p, want_p));
    }
    // P1: 1 / (2 g1) seeded with h = rsq / 2 (no further transcendental)
    {
        const float t = 2.f * g1;
        const float e1 = __builtin_fmaf(-t, h, 1.0f);
        const float p = __builtin_fmaf(e1, h, h);
        note(rep, 5, (unsigned)i, differ(p, want_p));
        const float e2 = __builtin_fmaf(-t, p, 1.0f);
        const float p2 = __builtin_fmaf(e2, p, p);
        note(rep, 6, (unsigned)i, differ(p2, want_p));
    }
    // P2: 1 / (2 g1) seeded with the hardware reciprocal of 2 g1
    {
        const float t = 2.f * g1;
        const float y0 = __builtin_amdgcn_rcpf(t);
        const float e1 = __builtin_fmaf(-t, y0, 1.0f);
        const float p = __builtin_fmaf(e1, y0, y0);
        note(rep, 8, (unsigned)i, differ(p, want_p));
    }
    // S3: hardware root, residual step through the hardware reciprocal of the root
    {
        const float hh = 0.5f * __builtin_amdgcn_rcpf(g0);
        const float g = __builtin_fmaf(r, hh, g0);
        note(rep, 9, (unsigned)i, differ(g, want));
    }
    note(rep, 10, (unsigned)i, differ(g0, want));  // bare v_sqrt_f32 (for information)
}

static const char* kNames[kCandidates] = {
    "R1  rcp + one fma pair                          vs 1.0f / d",
    "R2  rcp + two fma pairs                         vs 1.0f / d",
    "    bare v_rcp_f32                              vs 1.0f / d",
    "S1  v_sqrt + residual * (v_rsq / 2)             vs sqrtf(s)",
    "S2  v_rsq coupled iteration (one transcendental) vs sqrtf(s)",
    "P1  S1, then one fma pair seeded with v_rsq / 2 vs 1.f / (2.f * sqrtf(s))",
    "P1' S1, then two fma pairs seeded with v_rsq / 2 vs 1.f / (2.f * sqrtf(s))",
    "P3  S2, then one fma pair seeded with refined h vs 1.f / (2.f * sqrtf(s))",
    "P2  S1, then v_rcp(2 g) + one fma pair          vs 1.f / (2.f * sqrtf(s))",
    "S3  v_sqrt + residual * (v_rcp(g0) / 2)         vs sqrtf(s)",
    "    bare v_sqrt_f32                             vs sqrtf(s)",
    "",
};

static int run(const char* what, bool is_roots, unsigned first, unsigned last, Report* dev)
{
    Report host;
    std::memset(&host, 0, sizeof(host));
    std::memset(host.lowest, 0xff, sizeof(host.lowest));
    (void)hipMemcpy(dev, &host, sizeof(host), hipMemcpyHostToDevice);
    const unsigned long long n = (unsigned long long)last - first + 1;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (is_roots) roots<<<blocks, 256>>>(first, last, dev);
    else reciprocals<<<blocks, 256>>>(first, last, dev);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    (void)hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
    std::printf("%s: bit patterns [%#010x, %#010x], %llu values\n", what, first, last, n);
    for (int c = 0; c < kCandidates; ++c) {
        const bool mine = is_roots ? (c >= 3 && c <= 10) : c < 3;
        if (!mine) continue;
        std::printf("  %-72s mismatches %llu", kNames[c], host.mismatches[c]);
        if (host.mismatches[c]) std::printf("  (arguments from %#010x to %#010x)", host.lowest[c], host.highest[c]);
        std::printf("\n");
    }
    return 0;
}

int main()
{
    Report* dev;
    if (hipMalloc(&dev, sizeof(Report)) != hipSuccess) return 2;
    // the ranges the kernel's guard admits: denominators in [2^-30, 2^40]; 2 sqrt(s) in the same range, i.e. s in [2^-62, 2^78]
    run("reciprocal, d in [2^-30, 2^40]", false, 0x30800000u, 0x53800000u, dev);
    run("reciprocal, every positive normal d with a normal reciprocal [2^-126, 2^126]", false, 0x00800000u, 0x7e800000u, dev);
    run("root, s in [2^-62, 2^78]", true, 0x20800000u, 0x66800000u, dev);
    // (the guard sees 2 RN(sqrt(s)), which rounds into its range from a little outside: two more binades on either side)
    run("root, s in [2^-64, 2^80]", true, 0x1f800000u, 0x67800000u, dev);
    run("root, every positive normal s", true, 0x00800000u, 0x7f7fffffu, dev);
    run("root, upper half: s in [1, FLT_MAX]", true, 0x3f800000u, 0x7f7fffffu, dev);
    return 0;
}
