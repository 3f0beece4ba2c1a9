// Micro-benchmark: issue cost of individual gfx950 VALU instructions (inline asm, so the opcode is what is
// measured), as independent streams (throughput) and as one dependent chain (latency), at 1, 2 and 8 waves
// per SIMD.  Reported relative to v_add_f32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

// INDEP: 8 independent accumulators; DEP: one accumulator used 8 times in a row
#define KERNEL(NAME, INDEP_ASM, DEP_ASM)                                                        \
    __global__ void NAME##_indep(float* out, int iters)                                         \
    {                                                                                           \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
              a7 = a0 + 7, b = 1.0001f, c = 0.5f;                                               \
        for (int i = 0; i < iters; ++i) {                                                       \
            asm volatile(INDEP_ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c));                                                     \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;      \
    }                                                                                           \
    __global__ void NAME##_dep(float* out, int iters)                                           \
    {                                                                                           \
        float a0 = threadIdx.x, b = 1.0001f, c = 0.5f;                                          \
        for (int i = 0; i < iters; ++i) {                                                       \
            asm volatile(DEP_ASM : "+v"(a0) : "v"(b), "v"(c));                                  \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0;                                        \
    }

#define I8(OP, TAIL)                                                                                          \
    OP " %0, %0, " TAIL "\n" OP " %1, %1, " TAIL "\n" OP " %2, %2, " TAIL "\n" OP " %3, %3, " TAIL "\n"       \
    OP " %4, %4, " TAIL "\n" OP " %5, %5, " TAIL "\n" OP " %6, %6, " TAIL "\n" OP " %7, %7, " TAIL "\n"
#define D8(OP, TAIL, B) REP8(OP " %0, %0, " TAIL "\n")

KERNEL(add, I8("v_add_f32", "%8"), REP8("v_add_f32 %0, %0, %1\n"))
KERNEL(mul, I8("v_mul_f32", "%8"), REP8("v_mul_f32 %0, %0, %1\n"))
KERNEL(fma, I8("v_fma_f32", "%8, %9"), REP8("v_fma_f32 %0, %0, %1, %2\n"))
KERNEL(fma_neg, I8("v_fma_f32", "-%8, %9"), REP8("v_fma_f32 %0, %0, -%1, %2\n"))
KERNEL(fmac, "v_fmac_f32 %0, %8, %9\nv_fmac_f32 %1, %8, %9\nv_fmac_f32 %2, %8, %9\nv_fmac_f32 %3, %8, %9\n"
             "v_fmac_f32 %4, %8, %9\nv_fmac_f32 %5, %8, %9\nv_fmac_f32 %6, %8, %9\nv_fmac_f32 %7, %8, %9\n",
       REP8("v_fmac_f32 %0, %1, %2\n"))
KERNEL(max3, I8("v_max3_f32", "%8, %9"), REP8("v_max3_f32 %0, %0, %1, %2\n"))
KERNEL(min3u, I8("v_min3_u32", "%8, %9"), REP8("v_min3_u32 %0, %0, %1, %2\n"))
KERNEL(lshladd, I8("v_lshl_add_u32", "1, %9"), REP8("v_lshl_add_u32 %0, %0, 1, %2\n"))
KERNEL(rcp, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\n"
            "v_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n",
       REP8("v_rcp_f32 %0, %0\n"))
KERNEL(sqrt, "v_sqrt_f32 %0, %0\nv_sqrt_f32 %1, %1\nv_sqrt_f32 %2, %2\nv_sqrt_f32 %3, %3\nv_sqrt_f32 %4, %4\n"
             "v_sqrt_f32 %5, %5\nv_sqrt_f32 %6, %6\nv_sqrt_f32 %7, %7\n",
       REP8("v_sqrt_f32 %0, %0\n"))
KERNEL(divscale, "v_div_scale_f32 %0, vcc, %0, %8, %9\nv_div_scale_f32 %1, vcc, %1, %8, %9\n"
                 "v_div_scale_f32 %2, vcc, %2, %8, %9\nv_div_scale_f32 %3, vcc, %3, %8, %9\n"
                 "v_div_scale_f32 %4, vcc, %4, %8, %9\nv_div_scale_f32 %5, vcc, %5, %8, %9\n"
                 "v_div_scale_f32 %6, vcc, %6, %8, %9\nv_div_scale_f32 %7, vcc, %7, %8, %9\n",
       REP8("v_div_scale_f32 %0, vcc, %0, %1, %2\n"))
KERNEL(divfmas, I8("v_div_fmas_f32", "%8, %9"), REP8("v_div_fmas_f32 %0, %0, %1, %2\n"))
KERNEL(divfixup, I8("v_div_fixup_f32", "%8, %9"), REP8("v_div_fixup_f32 %0, %0, %1, %2\n"))
KERNEL(dpp, "v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_mov_b32_dpp %4, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_mov_b32_dpp %5, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_mov_b32_dpp %6, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_mov_b32_dpp %7, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n",
       REP8("s_nop 1\nv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"))

// packed fp32: 64-bit operands
#define KERNEL2(NAME, OP)                                                                                     \
    __global__ void NAME##_indep(float* out, int iters)                                                       \
    {                                                                                                         \
        typedef float v2f __attribute__((ext_vector_type(2)));                                                \
        v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, \
            a6 = a0 + 6.f, a7 = a0 + 7.f, b = {1.0001f, 0.999f};                                              \
        for (int i = 0; i < iters; ++i) {                                                                     \
            asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" \
                         OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"                             \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b));                                                                           \
        }                                                                                                     \
        v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                               \
    }                                                                                                         \
    __global__ void NAME##_dep(float* out, int iters)                                                         \
    {                                                                                                         \
        typedef float v2f __attribute__((ext_vector_type(2)));                                                \
        v2f a0 = {(float)threadIdx.x, 1.f}, b = {1.0001f, 0.999f};                                            \
        for (int i = 0; i < iters; ++i) {                                                                     \
            asm volatile(REP8(OP " %0, %0, %1\n") : "+v"(a0) : "v"(b));                                       \
        }                                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y;                                             \
    }
KERNEL2(pkadd, "v_pk_add_f32")
KERNEL2(pkmul, "v_pk_mul_f32")

typedef void (*kern_t)(float*, int);

static float time_kernel(kern_t k, float* out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, iters);
    hipEventRecord(e0);
    k<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms;
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    struct Entry { const char* name; kern_t indep, dep; };
    const Entry table[] = {
        {"v_add_f32", add_indep, add_dep},       {"v_mul_f32", mul_indep, mul_dep},
        {"v_fma_f32", fma_indep, fma_dep},       {"v_fma_f32 (neg)", fma_neg_indep, fma_neg_dep},
        {"v_fmac_f32", fmac_indep, fmac_dep},    {"v_max3_f32", max3_indep, max3_dep},
        {"v_min3_u32", min3u_indep, min3u_dep},  {"v_lshl_add_u32", lshladd_indep, lshladd_dep},
        {"v_pk_add_f32", pkadd_indep, pkadd_dep}, {"v_pk_mul_f32", pkmul_indep, pkmul_dep},
        {"v_rcp_f32", rcp_indep, rcp_dep},       {"v_sqrt_f32", sqrt_indep, sqrt_dep},
        {"v_div_scale_f32", divscale_indep, divscale_dep}, {"v_div_fmas_f32", divfmas_indep, divfmas_dep},
        {"v_div_fixup_f32", divfixup_indep, divfixup_dep}, {"v_mov_b32_dpp", dpp_indep, dpp_dep},
    };
    const int iters = 4000;
    // waves per SIMD: blocks of 256 threads = 4 waves = one wave on each SIMD of a CU
    const int wps[] = {1, 2, 3, 4, 6, 8};
    printf("ns per instruction per SIMD (8 instructions per loop trip; 256 CUs x 4 SIMDs); columns = waves per SIMD\n");
    printf("%-18s", "independent");
    for (int w : wps) printf(" %8dw", w);
    printf(" | dependent %6dw %8dw\n", 1, 2);
    for (const Entry& e : table) {
        printf("%-18s", e.name);
        for (int w : wps) printf(" %9.3f", time_kernel(e.indep, out, 256 * w, iters) * 1e6 / ((double)w * iters * 8));
        printf(" | %16.3f %9.3f\n", time_kernel(e.dep, out, 256, iters) * 1e6 / ((double)iters * 8),
               time_kernel(e.dep, out, 512, iters) * 1e6 / (2.0 * iters * 8));
    }
    return 0;
}
