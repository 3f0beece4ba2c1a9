// Micro-benchmark (round 4): does the issue rate of a wave's VALU stream depend on the SIZE of the loop body (instruction
// fetch) and on the instruction ENCODING (4-byte VOP2 against 8-byte VOP3)?  tools/ubench/issue_cycles.hip measures
// 128-instruction loops, which live in the instruction buffers; the fused solver kernel streams an 18-20 KB loop body per
// ring turn through every wave.  Same method: cycles by s_memtime inside the kernel, one workgroup of 256 x W threads
// per CU (W waves per SIMD), independent accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/ifetch_rate.hip -o build_ubench/ifetch_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct Stamp {
    unsigned long long cycles, real;
};

#define STR2(x) #x
#define STR(x) STR2(x)

// REPT groups of 8 independent instructions per loop trip
#define BENCH(NAME, REPT, INSTR)                                                                                  \
    __global__ void NAME(float* out, Stamp* stamps, int iters)                                                    \
    {                                                                                                             \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,    \
              a7 = a0 + 7, b = 1.0001f;                                                                           \
        __syncthreads();                                                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                           \
        for (int i = 0; i < iters; ++i) {                                                                         \
            asm volatile(".rept " STR(REPT) "\n" INSTR " %0, %0, %8\n" INSTR " %1, %1, %8\n" INSTR " %2, %2, %8\n" \
                         INSTR " %3, %3, %8\n" INSTR " %4, %4, %8\n" INSTR " %5, %5, %8\n" INSTR " %6, %6, %8\n"   \
                         INSTR " %7, %7, %8\n.endr\n"                                                             \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                         : "v"(b));                                                                               \
        }                                                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                       \
        if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0}; \
    }

BENCH(add32_128, 16, "v_add_f32_e32")      // 128 instr, 0.5 KB
BENCH(add32_2k, 256, "v_add_f32_e32")      // 2048 instr, 8 KB
BENCH(add32_4k, 512, "v_add_f32_e32")      // 4096 instr, 16 KB
BENCH(add32_8k, 1024, "v_add_f32_e32")     // 8192 instr, 32 KB
BENCH(add32_24k, 3072, "v_add_f32_e32")    // 24576 instr, 96 KB (beyond a 64 KB instruction cache)
BENCH(add64_128, 16, "v_add_f32_e64")      // 128 instr, 1 KB
BENCH(add64_2k, 256, "v_add_f32_e64")      // 16 KB
BENCH(add64_4k, 512, "v_add_f32_e64")      // 32 KB
BENCH(add64_12k, 1536, "v_add_f32_e64")    // 96 KB

typedef void (*kern_t)(float*, Stamp*, int);

static void run(const char* name, kern_t k, int per_trip, long body_bytes, float* out, Stamp* stamps, int cus)
{
    printf("%-12s body %6.1f KB:", name, body_bytes / 1024.0);
    for (int w = 1; w <= 4; ++w) {
        const int threads = 256 * w;
        const int iters = std::max(4, 4000000 / per_trip);
        for (int warm = 0; warm < 2; ++warm) k<<<cus, threads>>>(out, stamps, iters);
        k<<<cus, threads>>>(out, stamps, iters);
        hipDeviceSynchronize();
        const int waves = cus * threads / 64;
        std::vector<Stamp> h(waves);
        hipMemcpy(h.data(), stamps, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
        std::vector<double> cyc, clk;
        for (const Stamp& s : h) {
            cyc.push_back((double)s.cycles / iters / per_trip);
            clk.push_back((double)s.cycles / (double)s.real * 0.1);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        printf("  %dw: %5.2f cyc/instr/wave (p90 %5.2f) [%4.2f per SIMD] %.2f GHz", w, cyc[cyc.size() / 2],
               cyc[cyc.size() * 9 / 10], cyc[cyc.size() / 2] / w, clk[clk.size() / 2]);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    Stamp* stamps;
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(float));
    hipMalloc(&stamps, (size_t)cus * 16 * sizeof(Stamp));
    for (int i = 0; i < 50; ++i) add32_128<<<cus, 512>>>(out, stamps, 20000);
    hipDeviceSynchronize();
    printf("cycles per VALU instruction per wave (median over waves), W = waves per SIMD (one workgroup of 256 W threads per CU)\n");
    run("add e32", add32_128, 128, 128 * 4, out, stamps, cus);
    run("add e32", add32_2k, 2048, 2048 * 4, out, stamps, cus);
    run("add e32", add32_4k, 4096, 4096 * 4, out, stamps, cus);
    run("add e32", add32_8k, 8192, 8192 * 4, out, stamps, cus);
    run("add e32", add32_24k, 24576, 24576 * 4, out, stamps, cus);
    run("add e64", add64_128, 128, 128 * 8, out, stamps, cus);
    run("add e64", add64_2k, 2048, 2048 * 8, out, stamps, cus);
    run("add e64", add64_4k, 4096, 4096 * 8, out, stamps, cus);
    run("add e64", add64_12k, 12288, 12288 * 8, out, stamps, cus);
    return 0;
}
