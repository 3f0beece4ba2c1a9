// Micro-benchmark (round 5): does the PLACEMENT of a vector instruction's operands in the register file change what it costs
// on gfx950?  tools/ubench/pair_latency.hip measured 5.99 cycles for plain v_add_f32 on v64.. where dep_latency.hip (compiler-
// allocated low registers) had 4.44.  Same harness: cycles per instruction per wave at 1..4 waves per SIMD, fixed registers.
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-inline-asm tools/ubench/regbank.hip -o build_ubench/regbank
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))
#define REP64(X) REP16(REP4(X))

struct Stamp {
    unsigned long long cycles, real;
};

#define CLOBBERS                                                                                                              \
    "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v32", "v33", "v34", \
        "v35", "v36", "v37", "v38", "v39", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",   \
        "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v96", \
        "v97", "v120", "v121"

#define BENCH_KERNEL(NAME, BODY)                                                                                      \
    __global__ __launch_bounds__(1024) void NAME(float* out, Stamp* stamps, int iters)                                \
    {                                                                                                                 \
        asm volatile("s_nop 0" ::: CLOBBERS);                                                                         \
        __syncthreads();                                                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                               \
        for (int i = 0; i < iters; ++i) {                                                                             \
            asm volatile(BODY : : : CLOBBERS);                                                                        \
        }                                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                   \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                               \
        float r;                                                                                                      \
        asm volatile("v_add_f32 %0, v4, v64\nv_add_f32 %0, %0, v96\n" : "=v"(r) : : CLOBBERS);                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                               \
        if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0};  \
    }

// (values are whatever the registers hold: timing does not depend on the data)
#define ADD(d, a, b) "v_add_f32 v" #d ", v" #a ", v" #b "\n"
#define FMA(d, a, b, c) "v_fma_f32 v" #d ", v" #a ", v" #b ", v" #c "\n"
#define PKM(d, a, b) "v_pk_mul_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1]\n"
BENCH_KERNEL(k_low_chain, REP64(ADD(4, 4, 5)))
BENCH_KERNEL(k_low_4chains, REP16(ADD(4, 4, 12) ADD(5, 5, 12) ADD(6, 6, 12) ADD(7, 7, 12)))
BENCH_KERNEL(k_low_4chains_b, REP16(ADD(4, 4, 13) ADD(5, 5, 13) ADD(6, 6, 13) ADD(7, 7, 13)))
BENCH_KERNEL(k_v64_chain, REP64(ADD(64, 64, 89)))
BENCH_KERNEL(k_v64_chain_b, REP64(ADD(64, 64, 65)))
BENCH_KERNEL(k_v64_same_bank, REP64(ADD(64, 64, 68)))
BENCH_KERNEL(k_v64_4chains, REP16(ADD(64, 64, 89) ADD(65, 65, 89) ADD(66, 66, 89) ADD(67, 67, 89)))
BENCH_KERNEL(k_v64_4chains_nb, REP16(ADD(64, 64, 90) ADD(65, 65, 91) ADD(66, 66, 88) ADD(67, 67, 89)))
BENCH_KERNEL(k_v128_chain, REP64(ADD(96, 96, 97)))
BENCH_KERNEL(k_v192_chain, REP64(ADD(120, 120, 121)))
BENCH_KERNEL(k_far_apart, REP64(ADD(4, 4, 121)))
BENCH_KERNEL(k_dst_other, REP16(ADD(8, 4, 5) ADD(9, 4, 5) ADD(10, 4, 5) ADD(11, 4, 5)))
BENCH_KERNEL(k_const_src, REP64("v_add_f32 v64, 1.0, v64\n"))
BENCH_KERNEL(k_sgpr_src, REP64("v_add_f32 v64, s2, v64\n"))
BENCH_KERNEL(k_fma_3banks, REP16(FMA(8, 4, 5, 6) FMA(9, 5, 6, 7) FMA(10, 6, 7, 4) FMA(11, 7, 4, 5)))
BENCH_KERNEL(k_fma_same_bank, REP16(FMA(8, 4, 12, 16) FMA(9, 5, 13, 17) FMA(10, 6, 14, 18) FMA(11, 7, 15, 19)))
BENCH_KERNEL(k_fma_2same, REP16(FMA(8, 4, 12, 5) FMA(9, 5, 13, 6) FMA(10, 6, 14, 7) FMA(11, 7, 15, 4)))
BENCH_KERNEL(k_pk_low, REP16(PKM(4, 4, 12) PKM(6, 6, 12) PKM(8, 8, 12) PKM(10, 10, 12)))
BENCH_KERNEL(k_pk_v64, REP16(PKM(64, 64, 88) PKM(66, 66, 88) PKM(68, 68, 88) PKM(70, 70, 88)))
BENCH_KERNEL(k_pk_samebank, REP16(PKM(64, 64, 72) PKM(66, 66, 74) PKM(68, 68, 76) PKM(70, 70, 78)))
BENCH_KERNEL(k_pk_offbank, REP16(PKM(64, 64, 74) PKM(66, 66, 76) PKM(68, 68, 78) PKM(70, 70, 72)))

typedef void (*kern_t)(float*, Stamp*, int);

static double run(kern_t k, int waves_per_simd, float* out, Stamp* stamps, int iters, int num_cus)
{
    const int threads = 256 * waves_per_simd;
    const int waves = num_cus * threads / 64;
    for (int warm = 0; warm < 2; ++warm) k<<<num_cus, threads>>>(out, stamps, iters);
    k<<<num_cus, threads>>>(out, stamps, iters);
    (void)hipDeviceSynchronize();
    std::vector<Stamp> h(waves);
    (void)hipMemcpy(h.data(), stamps, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (const Stamp& s : h) cyc.push_back((double)s.cycles / iters);
    std::sort(cyc.begin(), cyc.end());
    return cyc[cyc.size() / 2];
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    Stamp* stamps;
    (void)hipMalloc(&out, (size_t)cus * 1024 * sizeof(float));
    (void)hipMalloc(&stamps, (size_t)cus * 16 * sizeof(Stamp));
    struct Entry {
        const char* name;
        kern_t k;
    };
    const Entry table[] = {
        {"v_add v4, v4, v5 (chain, low registers)", k_low_chain},
        {"four chains v4..v7 += v12", k_low_4chains},
        {"four chains v4..v7 += v13", k_low_4chains_b},
        {"v_add v64, v64, v89 (chain)", k_v64_chain},
        {"v_add v64, v64, v65 (chain)", k_v64_chain_b},
        {"v_add v64, v64, v68 (both sources bank 0)", k_v64_same_bank},
        {"four chains v64..v67 += v89", k_v64_4chains},
        {"four chains v64..v67 += v90, v91, v88, v89 (no shared bank)", k_v64_4chains_nb},
        {"v_add v96, v96, v97 (chain)", k_v128_chain},
        {"v_add v120, v120, v121 (chain)", k_v192_chain},
        {"v_add v4, v4, v121 (chain)", k_far_apart},
        {"v_add v8..v11 = v4 + v5 (independent, other destination)", k_dst_other},
        {"v_add v64, 1.0, v64 (inline constant)", k_const_src},
        {"v_add v64, s2, v64 (scalar source)", k_sgpr_src},
        {"v_fma, three sources in three banks", k_fma_3banks},
        {"v_fma, three sources in ONE bank", k_fma_same_bank},
        {"v_fma, two sources in one bank", k_fma_2same},
        {"v_pk_mul v[4:5].. low registers", k_pk_low},
        {"v_pk_mul v[64:65]..", k_pk_v64},
        {"v_pk_mul, sources in the same banks", k_pk_samebank},
        {"v_pk_mul, sources in other banks", k_pk_offbank},
    };
    const int iters = 20000;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int i = 0; i < 50; ++i) k_low_chain<<<cus, 512>>>(out, stamps, iters);
    (void)hipDeviceSynchronize();
    printf("cycles per instruction per WAVE (median over waves); columns = waves per SIMD\n");
    for (const Entry& e : table) {
        printf("%-62s", e.name);
        for (int w = 1; w <= 4; ++w) printf("  %dw: %6.2f", w, run(e.k, w, out, stamps, iters, cus) / 64.0);
        printf("\n");
    }
    return 0;
}
