// Micro-benchmark: issue cost of a correctly rounded fp32 division / sqrt on gfx950 relative to plain VALU ops.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/div_rate.hip -o gpurun_out/div_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, float seed, int iters)
{
    // 8 independent chains per lane so the pipe stays full
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
    float d = seed * 1.0001f + 1.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = a[i] * d + 0.25f;            // mul + add (2 VALU)
            if (MODE == 1) a[i] = a[i] / d + 0.25f;            // division + add
            if (MODE == 2) a[i] = sqrtf(a[i]) + d;             // sqrt + add
            if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]) + d;  // bare v_rcp + add
            if (MODE == 4) a[i] = 1.f / (2.f * sqrtf(a[i] + d));   // the phi/ksi pattern
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.25f, iters);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.25f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int blocks = 256 * 8, iters = 2000;  // 8 waves per SIMD
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const char* names[] = {"mul+add", "div+add", "sqrt+add", "rcp+add", "1/(2*sqrt)"};
    float ms[5] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters),
                   run<3>(out, blocks, iters), run<4>(out, blocks, iters)};
    // wave-ops per SIMD: blocks*4 waves / 1024 SIMDs * iters * 8
    const double ops_per_simd = (double)blocks * 4 / 1024 * iters * 8;
    for (int m = 0; m < 5; ++m)
        printf("%-12s %8.3f ms  -> %.1f ns per op-group per SIMD (%.1f cycles at 2.4 GHz)\n", names[m], ms[m],
               ms[m] * 1e6 / ops_per_simd, ms[m] * 1e6 / ops_per_simd * 2.4);
    return 0;
}
