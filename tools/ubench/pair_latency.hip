// Micro-benchmark (round 5): what a DEPENDENT pair of vector instructions costs a wave on gfx950 when the
// producer is a packed, DPP or transcendental instruction -- tools/ubench/dep_latency.hip measured plain producers only -- and
// which instruction kinds put two co-resident waves into the slow sharing mode (two issue turns per instruction and wave).
// Cycles per instruction per WAVE (s_memtime) at 1..4 waves per SIMD, like dep_latency.hip; bodies use fixed registers v64..v95
// (clobbered), so that the halves of a register pair can be named.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pair_latency.hip -o build_ubench/pair_latency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP2(X) X X
#define REP4(X) X X X X
#define REP8(X) REP4(X) REP4(X)
#define REP16(X) REP4(REP4(X))
#define REP32(X) REP16(X) REP16(X)
#define REP64(X) REP16(REP4(X))

struct Stamp {
    unsigned long long cycles, real;
};

#define CLOBBERS                                                                                                          \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", \
        "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "vcc", "s20", "s21", \
        "a0", "scc"

// INSTRS: instructions per trip of BODY (the bodies are not all 64 long)
#define BENCH_KERNEL(NAME, BODY)                                                                                       \
    __global__ void NAME(float* out, Stamp* stamps, int iters)                                                         \
    {                                                                                                                  \
        asm volatile("v_cvt_f32_u32 v64, %0\n" ::"v"(threadIdx.x + 3) : CLOBBERS);                                      \
        asm volatile(                                                                                                  \
            "v_add_f32 v65, 1.0, v64\nv_add_f32 v66, 1.0, v65\nv_add_f32 v67, 1.0, v66\nv_add_f32 v68, 1.0, v67\n"       \
            "v_add_f32 v69, 1.0, v68\nv_add_f32 v70, 1.0, v69\nv_add_f32 v71, 1.0, v70\nv_add_f32 v72, 1.0, v71\n"       \
            "v_add_f32 v73, 1.0, v72\nv_add_f32 v74, 1.0, v73\nv_add_f32 v75, 1.0, v74\nv_add_f32 v76, 1.0, v75\n"       \
            "v_add_f32 v77, 1.0, v76\nv_add_f32 v78, 1.0, v77\nv_add_f32 v79, 1.0, v78\nv_add_f32 v80, 1.0, v79\n"       \
            "v_add_f32 v81, 1.0, v80\nv_add_f32 v82, 1.0, v81\nv_add_f32 v83, 1.0, v82\nv_add_f32 v84, 1.0, v83\n"       \
            "v_add_f32 v85, 1.0, v84\nv_add_f32 v86, 1.0, v85\nv_add_f32 v87, 1.0, v86\nv_add_f32 v88, 1.0, v87\n"       \
            "v_mov_b32 v89, 1.0\nv_mov_b32 v90, 0.5\nv_mov_b32 v91, 1.0\nv_mov_b32 v92, 0.5\n"                         \
            "v_mov_b32 v93, 1.0\nv_mov_b32 v94, 1.0\nv_mov_b32 v95, 0\n" ::                                             \
                : CLOBBERS);                                                                                           \
        __syncthreads();                                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                \
        for (int i = 0; i < iters; ++i) {                                                                              \
            asm volatile(BODY : : : CLOBBERS);                                                                         \
        }                                                                                                              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                \
        float r;                                                                                                       \
        asm volatile("v_add_f32 %0, v64, v72\nv_add_f32 %0, %0, v80\nv_add_f32 %0, %0, v84\n" : "=v"(r) : : CLOBBERS);  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                \
        if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0};   \
    }

// v89 = v91 = v93 = v94 = 1.0 and v90 = v92 = 0.5: multiplying by (v[89:90]) etc. keeps values finite for a while; nothing
// here depends on the data (the solver kernel times the same on zeros, noise and images).
#define DPPS " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

// ---- dependent pairs: producer -> consumer -> producer ... (one chain; every instruction needs the previous one's result)
BENCH_KERNEL(k_plain_chain, REP64("v_add_f32 v64, v64, v89\n"))
BENCH_KERNEL(k_pk_chain, REP64("v_pk_mul_f32 v[64:65], v[64:65], v[88:89]\n"))
BENCH_KERNEL(k_pkfma_chain, REP64("v_pk_fma_f32 v[64:65], v[64:65], v[88:89], v[90:91]\n"))
BENCH_KERNEL(k_pk_then_plain, REP32("v_pk_mul_f32 v[64:65], v[64:65], v[88:89]\nv_add_f32 v64, v65, v89\n"))
BENCH_KERNEL(k_pk_then_plain_hi, REP32("v_pk_mul_f32 v[64:65], v[64:65], v[88:89]\nv_add_f32 v65, v64, v89\n"))
BENCH_KERNEL(k_plain_then_dppsrc, REP32("v_add_f32 v64, v65, v89\ns_nop 1\nv_sub_f32_dpp v65, v64, v89 wave_shr:1" DPPS))
BENCH_KERNEL(k_plain_then_dppsrc_nonop, REP32("v_add_f32 v64, v65, v89\nv_sub_f32_dpp v65, v64, v89 wave_shr:1" DPPS))
BENCH_KERNEL(k_dpp_then_plain, REP32("v_sub_f32_dpp v65, v64, v89 wave_shr:1" DPPS "v_add_f32 v64, v65, v89\n"))
BENCH_KERNEL(k_rcp_then_plain, REP32("v_rcp_f32 v64, v65\ns_nop 0\nv_add_f32 v65, v64, v89\n"))
BENCH_KERNEL(k_sqrt_then_plain, REP32("v_sqrt_f32 v64, v65\ns_nop 0\nv_add_f32 v65, v64, v89\n"))
BENCH_KERNEL(k_rcp_then_pkfma, REP32("v_rcp_f32 v64, v65\ns_nop 0\nv_pk_fma_f32 v[64:65], v[64:65], v[88:89], v[90:91]\n"))
BENCH_KERNEL(k_pk_opsel_chain, REP64("v_pk_mul_f32 v[64:65], v[64:65], v[88:89] op_sel_hi:[0,1]\n"))
// the sweep's own u -> v chain (14 plain instructions, each needing the one before), then the packed UV update and 4 packed / 4 DPP
// instructions of the next sweep's neighbour terms that need it: one sweep of the strip kernel in miniature (25 instructions x 2 + 14)
#define SWEEP_CHAIN                                                                                                          \
    "v_mul_f32 v66, v70, v64\nv_sub_f32 v66, v71, v66\nv_mul_f32 v66, v72, v66\nv_add_f32 v66, v66, v80\nv_mul_f32 v67, v73, v66\n" \
    "v_fma_f32 v68, -v67, v74, v66\nv_fmac_f32 v67, v68, v73\nv_mul_f32 v68, v70, v67\nv_sub_f32 v68, v75, v68\nv_mul_f32 v68, v72, v68\n" \
    "v_add_f32 v68, v81, v68\nv_mul_f32 v69, v76, v68\nv_fma_f32 v64, -v69, v77, v68\nv_fmac_f32 v69, v64, v76\n"
#define SWEEP_HEAD                                                                                                           \
    "v_pk_add_f32 v[82:83], v[78:79], v[66:67]\n"                                                                            \
    "v_sub_f32_dpp v84, v82, v78 wave_shr:1" DPPS "v_sub_f32_dpp v85, v83, v79 wave_shr:1" DPPS                              \
    "v_sub_f32_dpp v86, v82, v78 wave_shl:1" DPPS "v_sub_f32_dpp v87, v83, v79 wave_shl:1" DPPS                              \
    "v_pk_mul_f32 v[84:85], v[88:89], v[84:85] op_sel_hi:[0,1]\nv_pk_mul_f32 v[86:87], v[88:89], v[86:87] op_sel:[1,0]\n"    \
    "v_pk_add_f32 v[80:81], v[84:85], v[86:87]\nv_pk_add_f32 v[80:81], v[80:81], v[90:91]\nv_pk_add_f32 v[80:81], v[80:81], v[92:93]\n"
BENCH_KERNEL(k_sweep_mini, REP2(SWEEP_HEAD SWEEP_CHAIN) SWEEP_CHAIN)

// ---- which kinds switch two co-resident waves into the slow mode: 7 plain + 1 X (64 instructions per trip)
#define PL(i) "v_add_f32 v" #i ", v" #i ", v89\n"
#define P7 PL(64) PL(65) PL(66) PL(67) PL(64) PL(65) PL(66)
#define P15 P7 PL(67) P7
#define ONE_IN_8(X) REP8(P7 X)
#define ONE_IN_16(X) REP4(P15 X)
BENCH_KERNEL(k_all_plain, REP8(P7 PL(67)))
BENCH_KERNEL(k_x_pk_add, ONE_IN_8("v_pk_add_f32 v[72:73], v[72:73], v[88:89]\n"))
BENCH_KERNEL(k_x_pk_mov, ONE_IN_8("v_pk_mov_b32 v[72:73], v[74:75], v[76:77]\n"))
BENCH_KERNEL(k_x_dpp_wave_shr, ONE_IN_8("v_mov_b32_dpp v72, v73 wave_shr:1" DPPS))
BENCH_KERNEL(k_x_dpp_row_shr, ONE_IN_8("v_mov_b32_dpp v72, v73 row_shr:1" DPPS))
BENCH_KERNEL(k_x_dpp_quad, ONE_IN_8("v_mov_b32_dpp v72, v73 quad_perm:[0,0,1,2]" DPPS))
BENCH_KERNEL(k_x_dpp_row_ror, ONE_IN_8("v_mov_b32_dpp v72, v73 row_ror:1" DPPS))
BENCH_KERNEL(k_x_dpp_mirror, ONE_IN_8("v_mov_b32_dpp v72, v73 row_mirror" DPPS))
BENCH_KERNEL(k_x_sdwa, ONE_IN_8("v_mov_b32_sdwa v72, v73 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"))
BENCH_KERNEL(k_x_rcp, ONE_IN_8("v_rcp_f32 v72, v73\n"))
BENCH_KERNEL(k_x_fma_vop3, ONE_IN_8("v_fma_f32 v72, -v73, v74, v75\n"))
BENCH_KERNEL(k_x_lshl_add, ONE_IN_8("v_lshl_add_u32 v72, v73, 1, -1\n"))
BENCH_KERNEL(k_x_cmp, ONE_IN_8("v_cmp_lt_f32 vcc, v72, v73\n"))
BENCH_KERNEL(k_x_readlane, ONE_IN_8("v_readfirstlane_b32 s20, v72\n"))
BENCH_KERNEL(k_x_salu, ONE_IN_8("s_mul_i32 s20, s21, 7\n"))  // (no SCC write: the loop's compare sits before the asm body)
BENCH_KERNEL(k_x_snop, ONE_IN_8("s_nop 0\n"))
BENCH_KERNEL(k_x_mul_i32, ONE_IN_8("v_mul_lo_u32 v72, v73, v74\n"))
BENCH_KERNEL(k_x_f64, ONE_IN_8("v_add_f64 v[72:73], v[72:73], v[74:75]\n"))
BENCH_KERNEL(k_x_cvt_pk, ONE_IN_8("v_cvt_pk_bf16_f32 v72, v73, v74\n"))
BENCH_KERNEL(k_x_accread, ONE_IN_8("v_accvgpr_write_b32 a0, v72\n"))
BENCH_KERNEL(k_x_bperm, ONE_IN_16("ds_bpermute_b32 v72, v95, v73\n") "s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_x_swizzle, ONE_IN_16("ds_swizzle_b32 v72, v73 offset:0x1f\n") "s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_x_permlane, ONE_IN_8("v_permlane32_swap_b32 v72, v73\n"))

typedef void (*kern_t)(float*, Stamp*, int);

static double g_clock_ghz = 0.0;  // shader clock of the last run(): s_memtime cycles per s_memrealtime tick (100 MHz), median over waves
static double run(kern_t k, int waves_per_simd, float* out, Stamp* stamps, int iters, int num_cus)
{
    const int threads = 256 * waves_per_simd;
    const int waves = num_cus * threads / 64;
    for (int warm = 0; warm < 2; ++warm) k<<<num_cus, threads>>>(out, stamps, iters);
    k<<<num_cus, threads>>>(out, stamps, iters);
    hipDeviceSynchronize();
    std::vector<Stamp> h(waves);
    hipMemcpy(h.data(), stamps, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (const Stamp& s : h) cyc.push_back((double)s.cycles / iters), clk.push_back(s.real ? (double)s.cycles / (double)s.real * 0.1 : 0.0);
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    g_clock_ghz = clk[clk.size() / 2];
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
        int instrs;  // vector / LDS instructions per trip (s_nop and s_waitcnt not counted)
    };
    const Entry table[] = {
        {"chain: v_add_f32 -> v_add_f32", k_plain_chain, 64},
        {"chain: v_pk_mul_f32 -> v_pk_mul_f32", k_pk_chain, 64},
        {"chain: v_pk_fma_f32 -> v_pk_fma_f32", k_pkfma_chain, 64},
        {"chain: v_pk_mul_f32 (op_sel) -> v_pk_mul_f32", k_pk_opsel_chain, 64},
        {"chain: v_pk_mul_f32 -> v_add_f32 (reads .y, writes .x) -> ...", k_pk_then_plain, 64},
        {"chain: v_pk_mul_f32 -> v_add_f32 (reads .x, writes .y) -> ...", k_pk_then_plain_hi, 64},
        {"chain: v_add_f32 -> s_nop 1 -> v_sub_f32_dpp (DPP operand) -> ...", k_plain_then_dppsrc, 64},
        {"chain: v_add_f32 -> v_sub_f32_dpp (no s_nop: timing only) -> ...", k_plain_then_dppsrc_nonop, 64},
        {"chain: v_sub_f32_dpp -> v_add_f32 -> ...", k_dpp_then_plain, 64},
        {"chain: v_rcp_f32 -> s_nop 0 -> v_add_f32 -> ...", k_rcp_then_plain, 64},
        {"chain: v_sqrt_f32 -> s_nop 0 -> v_add_f32 -> ...", k_sqrt_then_plain, 64},
        {"chain: v_rcp_f32 -> s_nop 0 -> v_pk_fma_f32 -> ...", k_rcp_then_pkfma, 64},
        {"a sweep in miniature (10 packed/DPP head + 14 plain chain) x 2 + chain", k_sweep_mini, 62},
        {"64 plain (four chains)", k_all_plain, 64},
        {"7 plain + 1 v_pk_add_f32", k_x_pk_add, 64},
        {"7 plain + 1 v_pk_mov_b32", k_x_pk_mov, 64},
        {"7 plain + 1 v_mov_b32_dpp wave_shr:1", k_x_dpp_wave_shr, 64},
        {"7 plain + 1 v_mov_b32_dpp row_shr:1", k_x_dpp_row_shr, 64},
        {"7 plain + 1 v_mov_b32_dpp quad_perm", k_x_dpp_quad, 64},
        {"7 plain + 1 v_mov_b32_dpp row_ror:1", k_x_dpp_row_ror, 64},
        {"7 plain + 1 v_mov_b32_dpp row_mirror", k_x_dpp_mirror, 64},
        {"7 plain + 1 v_mov_b32_sdwa", k_x_sdwa, 64},
        {"7 plain + 1 v_rcp_f32", k_x_rcp, 64},
        {"7 plain + 1 v_fma_f32 (VOP3, neg)", k_x_fma_vop3, 64},
        {"7 plain + 1 v_lshl_add_u32", k_x_lshl_add, 64},
        {"7 plain + 1 v_cmp_lt_f32 vcc", k_x_cmp, 64},
        {"7 plain + 1 v_readfirstlane_b32", k_x_readlane, 64},
        {"7 plain + 1 s_mul_i32", k_x_salu, 64},
        {"7 plain + 1 s_nop 0", k_x_snop, 64},
        {"7 plain + 1 v_mul_lo_u32", k_x_mul_i32, 64},
        {"7 plain + 1 v_add_f64", k_x_f64, 64},
        {"7 plain + 1 v_cvt_pk_bf16_f32", k_x_cvt_pk, 64},
        {"7 plain + 1 v_accvgpr_write_b32", k_x_accread, 64},
        {"15 plain + 1 ds_bpermute_b32 (waited for once per 64)", k_x_bperm, 64},
        {"15 plain + 1 ds_swizzle_b32 (waited for once per 64)", k_x_swizzle, 64},
        {"7 plain + 1 v_permlane32_swap_b32", k_x_permlane, 64},
    };
    const int iters = 20000;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int i = 0; i < 100; ++i) k_all_plain<<<cus, 512>>>(out, stamps, iters);
    hipDeviceSynchronize();
    printf("cycles per instruction per WAVE (median over waves); columns = waves per SIMD\n");
    for (const Entry& e : table) {
        printf("%-72s", e.name);
        for (int w = 1; w <= 4; ++w) printf("  %dw: %6.2f", w, run(e.k, w, out, stamps, iters, cus) / e.instrs);
        printf("\n");
    }
    // Does the chip hold its clock when the vector ALUs are kept busy?  Long launches (0.1-0.2 s each) of three streams at 1..4
    // waves per SIMD: cycles per instruction per wave, the shader clock held, and what that makes in wave-instructions per
    // SIMD and microsecond.
    printf("\nlong launches: cycles per instruction per wave @ clock GHz -> instructions per SIMD per us\n");
    const Entry longs[] = {{"64 plain (four chains)", k_all_plain, 64}, {"v_pk_fma_f32 chain", k_pkfma_chain, 64},
                           {"a sweep in miniature", k_sweep_mini, 62}};
    for (const Entry& e : longs) {
        printf("%-30s", e.name);
        for (int w = 1; w <= 4; ++w) {
            const double c = run(e.k, w, out, stamps, iters * 25, cus) / e.instrs;
            printf("  %dw: %5.2f @ %.2f -> %5.0f", w, c, g_clock_ghz, w / c * g_clock_ghz * 1e3);
        }
        printf("\n");
    }
    return 0;
}
