// Micro-benchmark: how long a wave64 VALU instruction that NEEDS the previous one's result waits on gfx950 -- by the way the
// value travels (same register as destination and source 0, source 1, the fma accumulator, a second register, a packed pair, a
// DPP operand) and by how many independent chains the stream interleaves -- at 1..3 waves per SIMD.  Cycles counted in the kernel
// with s_memtime.  The fused solver's sweeps are one chain of about 85 dependent instructions per row step.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dep_latency.hip -o gpurun_out/dep_latency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

typedef float v2f __attribute__((ext_vector_type(2)));
struct Stamp {
    unsigned long long cycles, real;
};

#define BENCH_KERNEL(NAME, BODY)                                                                      \
    __global__ void NAME(float* out, Stamp* stamps, int iters)                                        \
    {                                                                                                 \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = 1.0001f, c = 0.5f;                                                                  \
        v2f p0 = v2f{a0, a1}, p1 = v2f{a2, a3}, p2 = v2f{a4, a5}, p3 = v2f{a6, a7};                    \
        __syncthreads();                                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                               \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b), "v"(c)); \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y; \
        if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0}; \
    }

// every body: 64 instructions
BENCH_KERNEL(k_indep8, REP4(REP4("v_add_f32 %0, %0, %12\nv_add_f32 %1, %1, %12\nv_add_f32 %2, %2, %12\nv_add_f32 %3, %3, %12\n")))  // (4 chains of distance 4)
BENCH_KERNEL(k_same_src0, REP16(REP4("v_add_f32 %0, %0, %12\n")))
BENCH_KERNEL(k_same_src1, REP16(REP4("v_add_f32 %0, %12, %0\n")))
BENCH_KERNEL(k_fma_acc, REP16(REP4("v_fma_f32 %0, %12, %13, %0\n")))
BENCH_KERNEL(k_fma_src0, REP16(REP4("v_fma_f32 %0, %0, %12, %13\n")))
BENCH_KERNEL(k_pingpong, REP16("v_add_f32 %1, %0, %12\nv_add_f32 %0, %1, %12\nv_add_f32 %1, %0, %12\nv_add_f32 %0, %1, %12\n"))
BENCH_KERNEL(k_mul_add, REP16("v_mul_f32 %1, %0, %12\nv_add_f32 %0, %1, %13\nv_mul_f32 %1, %0, %12\nv_add_f32 %0, %1, %13\n"))
BENCH_KERNEL(k_mul_fma, REP16("v_mul_f32 %1, %0, %12\nv_fma_f32 %2, %1, %13, %0\nv_mul_f32 %1, %2, %12\nv_fma_f32 %0, %1, %13, %2\n"))
BENCH_KERNEL(k_ilp2, REP16("v_mul_f32 %1, %0, %12\nv_mul_f32 %3, %2, %12\nv_add_f32 %0, %1, %13\nv_add_f32 %2, %3, %13\n"))
BENCH_KERNEL(k_ilp3, REP4(REP4("v_mul_f32 %1, %0, %12\nv_mul_f32 %3, %2, %12\nv_mul_f32 %5, %4, %12\nv_add_f32 %0, %1, %13\n")  // (not an exact multiple: 4 x 16)
                           ))
BENCH_KERNEL(k_ilp4, REP4(REP4("v_mul_f32 %1, %0, %12\nv_mul_f32 %3, %2, %12\nv_mul_f32 %5, %4, %12\nv_mul_f32 %7, %6, %12\n"))
                           )
BENCH_KERNEL(k_dpp_chain, REP16("v_add_f32_dpp %1, %0, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                "v_add_f32_dpp %0, %1, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                "v_add_f32_dpp %1, %0, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                "v_add_f32_dpp %0, %1, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"))
BENCH_KERNEL(k_rcp_chain, REP16("v_rcp_f32 %1, %0\nv_add_f32 %0, %1, %12\nv_rcp_f32 %1, %0\nv_add_f32 %0, %1, %12\n"))


// ---- mixed streams: the classes the solver kernel is made of, in its proportions (per 32: 15 plain, 11 packed, 5 DPP, 1 transcendental)
#define PL(i) "v_add_f32 %" #i ", %" #i ", %12\n"
#define PK(i) "v_pk_mul_f32 %" #i ", %" #i ", %" #i "\n"
#define DP(i, j) "v_add_f32_dpp %" #i ", %" #j ", %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define TR(i) "v_rcp_f32 %" #i ", %" #i "\n"
BENCH_KERNEL(k_mix_plain_pk, REP16(PL(0) PK(8) PL(1) PK(9)))
BENCH_KERNEL(k_mix_plain_dpp, REP16(PL(0) DP(4, 5) PL(1) DP(5, 4)))
BENCH_KERNEL(k_mix_interleaved, REP4(PL(0) PK(8) PL(1) DP(4, 5) PK(9) PL(2) PL(3) PK(10) DP(5, 4) PL(0) PK(11) PL(1) PL(2) PK(8) DP(6, 7) PL(3)) \
                                 REP4(PK(9) PL(0) PL(1) PK(10) DP(7, 6) PL(2) PK(11) PL(3) PL(0) PK(8) DP(4, 5) PL(1) PK(9) PL(2) TR(6) PK(10)))
BENCH_KERNEL(k_mix_grouped, REP4(PL(0) PL(1) PL(2) PL(3) PL(0) PL(1) PL(2) PL(3) PK(8) PK(9) PK(10) PK(11) PK(8) DP(4, 5) DP(5, 4) DP(6, 7)) \
                             REP4(PL(0) PL(1) PL(2) PL(3) PL(0) PL(1) PL(2) PK(9) PK(10) PK(11) PK(8) PK(9) PK(10) DP(7, 6) DP(4, 5) TR(6)))
// one packed / DPP / transcendental instruction among 7 or 15 plain ones (64 instructions per trip)
#define P7 PL(0) PL(1) PL(2) PL(3) PL(0) PL(1) PL(2)
#define P15 P7 PL(3) P7
#define REP8(X) X X X X X X X X
BENCH_KERNEL(k_pk_1in8, REP8(P7 PK(8)))
BENCH_KERNEL(k_pk_1in16, REP4(P15 PK(8)))
BENCH_KERNEL(k_dpp_1in8, REP8(P7 DP(4, 5)))
BENCH_KERNEL(k_dpp_1in16, REP4(P15 DP(4, 5)))
BENCH_KERNEL(k_tr_1in16, REP4(P15 TR(6)))
BENCH_KERNEL(k_pk_only, REP16(PK(8) PK(9) PK(10) PK(11)))
BENCH_KERNEL(k_pk_dpp, REP16(PK(8) DP(4, 5) PK(9) DP(5, 4)))
// lower densities, clumps, and the other instruction kinds of the kernel
#define P31 P15 PL(3) P15
#define P63 P31 PL(3) P31
#define IN3(i) "v_min3_u32 %" #i ", %" #i ", %5, %6\n"
#define CND(i) "v_cndmask_b32 %" #i ", %5, %6, vcc\n"
#define LDS(i) "ds_swizzle_b32 %" #i ", %5 offset:0x1f\n"
BENCH_KERNEL(k_pk_1in32, P31 PK(8) P31 PK(8))
BENCH_KERNEL(k_pk_1in64, P63 PK(8))
BENCH_KERNEL(k_clump_48_16, P31 P15 PL(0) PL(1) REP4(PK(8) PK(9) PK(10) PK(11)))
BENCH_KERNEL(k_min3_1in8, REP8(P7 IN3(4)))
BENCH_KERNEL(k_cnd_1in8, REP8(P7 CND(4)))
BENCH_KERNEL(k_lds_1in16, REP4(P15 LDS(4)) "s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_fma_only, REP16("v_fma_f32 %0, %0, %12, %13\nv_fma_f32 %1, %1, %12, %13\nv_fma_f32 %2, %2, %12, %13\nv_fma_f32 %3, %3, %12, %13\n"))
BENCH_KERNEL(k_fma_pk_1in8, REP8("v_fma_f32 %0, %0, %12, %13\nv_fma_f32 %1, %1, %12, %13\nv_fma_f32 %2, %2, %12, %13\nv_fma_f32 %3, %3, %12, %13\n" \
                                 "v_fma_f32 %0, %0, %12, %13\nv_fma_f32 %1, %1, %12, %13\nv_fma_f32 %2, %2, %12, %13\n" PK(8)))

typedef void (*kern_t)(float*, Stamp*, int);

static double run(kern_t k, int waves_per_simd, float* out, Stamp* stamps, int iters, int num_cus)
{
    const int threads = 256 * waves_per_simd;
    const int waves = num_cus * threads / 64;
    for (int warm = 0; warm < 2; ++warm) k<<<num_cus, threads>>>(out, stamps, iters);
    k<<<num_cus, threads>>>(out, stamps, iters);
    hipDeviceSynchronize();
    std::vector<Stamp> h(waves);
    hipMemcpy(h.data(), stamps, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (const Stamp& s : h) cyc.push_back((double)s.cycles / iters);
    std::sort(cyc.begin(), cyc.end());
    return cyc[cyc.size() / 2];
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
    };
    const Entry table[] = {
        {"independent (four chains of distance four)", k_indep8},
        {"v_add dst = src0 = previous result", k_same_src0},
        {"v_add dst = src1 = previous result", k_same_src1},
        {"v_fma, previous result as the accumulator (src2)", k_fma_acc},
        {"v_fma, previous result as src0", k_fma_src0},
        {"v_add ping-pong through two registers", k_pingpong},
        {"v_mul -> v_add -> v_mul ... (two registers)", k_mul_add},
        {"v_mul -> v_fma(prev, ., older) ... (three registers)", k_mul_fma},
        {"two such chains interleaved", k_ilp2},
        {"three multiplies + one dependent add", k_ilp3},
        {"four independent multiplies (distance four)", k_ilp4},
        {"v_add_f32_dpp on the previous result (no s_nop: timing only)", k_dpp_chain},
        {"v_rcp -> v_add -> v_rcp ...", k_rcp_chain},
        {"mixed: plain, packed, plain, packed ...", k_mix_plain_pk},
        {"mixed: plain, DPP, plain, DPP ...", k_mix_plain_dpp},
        {"mixed: the solver's proportions, classes interleaved", k_mix_interleaved},
        {"mixed: the solver's proportions, classes in groups", k_mix_grouped},
        {"7 plain + 1 packed", k_pk_1in8},
        {"15 plain + 1 packed", k_pk_1in16},
        {"7 plain + 1 DPP", k_dpp_1in8},
        {"15 plain + 1 DPP", k_dpp_1in16},
        {"15 plain + 1 transcendental", k_tr_1in16},
        {"packed only", k_pk_only},
        {"packed, DPP, packed, DPP ...", k_pk_dpp},
        {"31 plain + 1 packed", k_pk_1in32},
        {"63 plain + 1 packed", k_pk_1in64},
        {"48 plain, then 16 packed", k_clump_48_16},
        {"7 plain + 1 v_min3_u32", k_min3_1in8},
        {"7 plain + 1 v_cndmask_b32", k_cnd_1in8},
        {"15 plain + 1 ds_swizzle_b32 (waited for once per 64)", k_lds_1in16},
        {"v_fma_f32 only (four chains)", k_fma_only},
        {"7 v_fma_f32 + 1 packed", k_fma_pk_1in8},
    };
    const int iters = 20000;
    for (int i = 0; i < 100; ++i) k_indep8<<<cus, 512>>>(out, stamps, iters);
    hipDeviceSynchronize();
    printf("cycles per instruction per WAVE (median over waves); columns = waves per SIMD\n");
    for (const Entry& e : table) {
        printf("%-62s", e.name);
        for (int w = 1; w <= 4; ++w) printf("  %dw: %6.2f", w, run(e.k, w, out, stamps, iters, cus) / 64.0);
        printf("\n");
    }
    return 0;
}
