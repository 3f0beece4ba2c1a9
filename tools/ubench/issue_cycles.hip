// Micro-benchmark: CYCLES per wave64 VALU instruction on gfx950, counted in the kernel with s_memtime (so the
// result does not depend on the clock the chip happens to hold), at 1..4 co-resident waves per SIMD, for the
// instruction kinds the fused solver kernel is made of.  Also reports the shader clock the run held
// (s_memtime ticks per s_memrealtime tick x 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/issue_cycles.hip -o gpurun_out/issue_cycles
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

struct Stamp {
    unsigned long long cycles, real;
};

// BODY runs `iters` times; `per_trip` = instructions of interest per trip
#define BENCH_KERNEL(NAME, DECL, BODY, SINK)                                                     \
    __global__ void NAME(float* out, Stamp* stamps, int iters)                                   \
    {                                                                                            \
        DECL;                                                                                    \
        __syncthreads();                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                          \
        for (int i = 0; i < iters; ++i) {                                                        \
            BODY;                                                                                \
        }                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SINK;                                       \
        if ((threadIdx.x & 63) == 0) {                                                           \
            Stamp s{t1 - t0, r1 - r0};                                                           \
            stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;                            \
        }                                                                                        \
    }

#define ACC8 float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f, c = 0.5f
#define IO8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
#define SUM8 (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define I8(OP, TAIL)                                                                                          \
    OP " %0, %0, " TAIL "\n" OP " %1, %1, " TAIL "\n" OP " %2, %2, " TAIL "\n" OP " %3, %3, " TAIL "\n"       \
    OP " %4, %4, " TAIL "\n" OP " %5, %5, " TAIL "\n" OP " %6, %6, " TAIL "\n" OP " %7, %7, " TAIL "\n"

// 64 instructions per trip
BENCH_KERNEL(k_add_indep, ACC8, asm volatile(REP4(REP4(I8("v_add_f32", "%8")) ) IO8); asm volatile("" ::: "memory"), SUM8)
BENCH_KERNEL(k_fma_indep, ACC8, asm volatile(REP4(REP4(I8("v_fma_f32", "%8, %9"))) IO8), SUM8)
BENCH_KERNEL(k_fma_dep, ACC8, asm volatile(REP16(REP4("v_fma_f32 %0, %0, %8, %9\n")) IO8), SUM8)
typedef float v2f __attribute__((ext_vector_type(2)));
#define PKACC8 v2f a0 = v2f{(float)threadIdx.x, 1.f}; v2f a1 = a0 + 1.f; v2f a2 = a0 + 2.f; v2f a3 = a0 + 3.f; v2f a4 = a0 + 4.f; \
               v2f a5 = a0 + 5.f; v2f a6 = a0 + 6.f; v2f a7 = a0 + 7.f; v2f b = v2f{1.0001f, 0.999f}; v2f c = b
BENCH_KERNEL(k_pk_indep, PKACC8, asm volatile(REP4(REP4(I8("v_pk_mul_f32", "%8"))) IO8), (SUM8).x)
BENCH_KERNEL(k_rcp_indep, ACC8,
             asm volatile(REP4(REP4("v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\n"
                                    "v_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n")) IO8),
             SUM8)
BENCH_KERNEL(k_dpp_indep, ACC8,
             asm volatile(REP4(REP4("v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %2, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %3, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %4, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %5, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %6, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %7, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) IO8),
             SUM8)
// an add whose source is the neighbour lane (DPP folded into the consumer), fed by the previous add: the pattern of
// the solver's x-neighbour fetches
BENCH_KERNEL(k_add_dpp_dep, ACC8,
             asm volatile(REP4(REP4("s_nop 1\nv_add_f32_dpp %0, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "s_nop 1\nv_add_f32_dpp %1, %2, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "s_nop 1\nv_add_f32_dpp %2, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "s_nop 1\nv_add_f32_dpp %3, %0, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) IO8),
             SUM8)

// compiler-generated streams (what the solver really runs): correctly rounded division and sqrt, 8 independent
// chains per trip, and one dependent chain
BENCH_KERNEL(k_div_indep, ACC8, a0 = b / a0; a1 = b / a1; a2 = b / a2; a3 = b / a3; a4 = c / a4; a5 = c / a5; a6 = c / a6;
             a7 = c / a7; asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)), SUM8)
BENCH_KERNEL(k_div_dep, ACC8, a0 = b / a0; a0 = c / a0; a0 = b / a0; a0 = c / a0; a0 = b / a0; a0 = c / a0; a0 = b / a0;
             a0 = c / a0; asm volatile("" : "+v"(a0)), SUM8)
BENCH_KERNEL(k_sqrt_indep, ACC8, a0 = sqrtf(a0); a1 = sqrtf(a1); a2 = sqrtf(a2); a3 = sqrtf(a3); a4 = sqrtf(a4);
             a5 = sqrtf(a5); a6 = sqrtf(a6); a7 = sqrtf(a7);
             asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)), SUM8)

typedef void (*kern_t)(float*, Stamp*, int);

struct Result {
    double cycles_per_trip, ghz, wall_ns_per_trip;
};

static Result run(kern_t k, int waves_per_simd, float* out, Stamp* stamps, int iters, int num_cus)
{
    // one workgroup of 256 * waves_per_simd threads per CU: its waves spread over the CU's four SIMDs
    const int threads = 256 * waves_per_simd;
    const int waves = num_cus * threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int warm = 0; warm < 3; ++warm) k<<<num_cus, threads>>>(out, stamps, iters);
    hipEventRecord(e0);
    k<<<num_cus, threads>>>(out, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(waves);
    hipMemcpy(h.data(), stamps, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (const Stamp& s : h) {
        cyc.push_back((double)s.cycles / iters);
        clk.push_back((double)s.cycles / (double)s.real * 0.1);  // s_memrealtime: 100 MHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return {cyc[cyc.size() / 2], clk[clk.size() / 2], ms * 1e6 / iters};
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
    struct Entry {
        const char* name;
        kern_t k;
        int per_trip;
        const char* unit;
    };
    const Entry table[] = {
        {"v_add_f32 independent", k_add_indep, 128, "instruction"},
        {"v_fma_f32 independent", k_fma_indep, 128, "instruction"},
        {"v_fma_f32 dependent chain", k_fma_dep, 64, "instruction"},
        {"v_pk_mul_f32 independent", k_pk_indep, 128, "instruction"},
        {"v_rcp_f32 independent", k_rcp_indep, 128, "instruction"},
        {"v_mov_b32_dpp independent", k_dpp_indep, 128, "instruction"},
        {"s_nop 1 + v_add_f32_dpp on fresh value", k_add_dpp_dep, 64, "pair"},
        {"a / b correctly rounded, 8 independent", k_div_indep, 8, "division"},
        {"a / b correctly rounded, dependent chain", k_div_dep, 8, "division"},
        {"sqrtf correctly rounded, 8 independent", k_sqrt_indep, 8, "sqrt"},
    };
    const int iters = 20000;
    // keep the chip busy for a while first so the clock governor has settled
    for (int i = 0; i < 200; ++i) k_fma_indep<<<cus, 512>>>(out, stamps, iters);
    hipDeviceSynchronize();
    printf("cycles per <unit> per WAVE (median over waves), [cycles per unit per SIMD], shader clock GHz; columns = waves per SIMD\n");
    for (const Entry& e : table) {
        printf("%-42s", e.name);
        for (int w = 1; w <= 4; ++w) {
            const Result r = run(e.k, w, out, stamps, iters, cus);
            printf("  %dw: %7.2f [%6.2f] %.2fGHz", w, r.cycles_per_trip / e.per_trip, r.cycles_per_trip / e.per_trip / w, r.ghz);
        }
        printf("  per %s\n", e.unit);
    }
    return 0;
}
