// Launcher of the fused outer-iteration kernel: strip plan, arguments, dispatch to the instance objects
// (solve_fused_instance.hip; the kernel itself is solve_fused_kernel.hpp).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "common.hpp"
#include "solve_fused_args.hpp"
#include "solve_fused_probes.hpp"

namespace {

// true when x is a normal power of two whose reciprocal (and 1/(2x), 1/(4x)) is exactly representable
bool is_power_of_two(float x)
{
    int e = 0;
    return x > 0.f && std::frexp(x, &e) == 0.5f && e > -100 && e < 100;
}

}  // namespace

namespace flow2d {

namespace probe = flow2d_probe;

#ifdef FLOW2D_DEV_BUILD
// developer builds: one device buffer for the wave stamps of all instance objects, their counter behind them, and one for the
// waves' stall histograms (allocated at the first launch)
static unsigned long long* stamp_buffer()
{
    static unsigned long long* p = [] {
        void* q = nullptr;
        const size_t bytes = size_t(probe::kStampWaves) * probe::kStampWords * sizeof(unsigned long long) + sizeof(unsigned long long);
        (void)hipMalloc(&q, bytes);
        (void)hipMemset(q, 0, bytes);
        return static_cast<unsigned long long*>(q);
    }();
    return p;
}
static unsigned int* stamp_counter() { return reinterpret_cast<unsigned int*>(stamp_buffer() + size_t(probe::kStampWaves) * probe::kStampWords); }
static unsigned int* stall_buffer()
{
    static unsigned int* p = [] {
        void* q = nullptr;
        const size_t bytes = size_t(probe::kStallWaves) * probe::kStallRows * 64 * sizeof(unsigned int);
        (void)hipMalloc(&q, bytes);
        (void)hipMemset(q, 0, bytes);
        return static_cast<unsigned int*>(q);
    }();
    return p;
}

// packed-planes probe: (f0, f1, u, v) of a pixel side by side in one float4 plane, (du, dv) in a float2 plane; the launcher packs
// before and unpacks after every launch (the kernel alone is what the probe times: stamps or a kernel trace)
struct PackedPlanes {
    float *in = nullptr, *duv = nullptr, *out = nullptr;
    size_t floats = 0;
};
static PackedPlanes& packed_planes(size_t plane_floats)
{
    static PackedPlanes p;
    if (p.floats < plane_floats) {
        (void)hipFree(p.in), (void)hipFree(p.duv), (void)hipFree(p.out);
        (void)hipMalloc(reinterpret_cast<void**>(&p.in), plane_floats * 16);
        (void)hipMalloc(reinterpret_cast<void**>(&p.duv), plane_floats * 8);
        (void)hipMalloc(reinterpret_cast<void**>(&p.out), plane_floats * 8);
        p.floats = plane_floats;
    }
    return p;
}
__global__ void pack_planes_kernel(const float* f0, const float* f1, const float* u, const float* v, const float* du, const float* dv,
                                   float4* in, float2* duv, size_t n)
{
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i < n) in[i] = make_float4(f0[i], f1[i], u[i], v[i]), duv[i] = make_float2(du[i], dv[i]);
}
__global__ void unpack_planes_kernel(const float2* out, float* du, float* dv, size_t n)
{
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i < n) du[i] = out[i].x, dv[i] = out[i].y;
}
#endif

bool fused_supports(size_t inner) { return inner >= 1 && inner <= 5; }

// Stage W multiplies the sum of two robustifiers by HALF the neighbour weight alpha / h^2 instead of halving the sum first: the
// same bits as long as halving the weight is exact, i.e. the weight is zero or a normal number with room below it
// (solve_fused_kernel.hpp, stage W).  A weight below 2^-100 (alpha of 1e-30 and less) sends the level to the other kernels.
bool fused_weights_ok(float hx, float hy, float alpha)
{
    const float wx = alpha / (hx * hx), wy = alpha / (hy * hy);
    auto ok = [](float w) { return w == 0.f || (std::fabs(w) >= 0x1p-100f && std::isfinite(w)) || std::isnan(w); };
    return ok(wx) && ok(wy);
}

// The fused kernel addresses a plane through a buffer descriptor with 32-bit byte offsets (row offset as the scalar
// offset, column as a 32-bit vector offset): the plane, height x pitch bytes, must stay below 4 GiB.
bool fused_addressable(size_t h, size_t pitch_bytes) { return h != 0 && pitch_bytes <= 0xffffffffull / h; }

// Strip heights of a launch.  A wave spends (rows + 2*inner + 3) row steps on `rows` stored rows, so tall strips
// waste less; but the launch should fill the chip in whole co-resident rounds (two 256-thread workgroups per CU), and a
// round lasts as long as its slowest wave.  Cost model in row steps of an interior wave: a wave on an image border runs
// the EDGE body (kEdgeCost times the instructions per row), a full round of 2 workgroups per CU costs 2 x the slowest
// wave's steps (VALU-issue bound), a last round with at most one workgroup per CU 1.3 x (a lone wave per SIMD cannot
// saturate the VALU).  The peeled start-up steps run without the stages whose rows nothing depends on yet
// (strip_step), which is worth about six whole steps at inner = 5; the estimate below weighs the stages by their
// instruction counts (stage P 102, stage W 20, a sweep 46, the rest 16).
// Candidates: uniform strips (every launch has border waves, so its rounds run at the EDGE body's pace), and for every
// number of strips per column the border-aware partition whose border strips are shorter by the cost ratio
// (FusedArgs::rows_edge), with the first and last block column cut into strips of that height throughout.
struct FusedPlan {
    int rows_interior, rows_edge, strips_interior, blocks_x, blocks;  // blocks: the launch's grid (per batch instance)
};
// EDGE body / interior body: 1.2-1.26 in VALU instructions per row step (swept 1.0 ... 1.38 in rounds 3 and 5: 1.22)
static const double kEdgeCost = 1.22;

static FusedPlan fused_plan_search(const flow2d_context* ctx, size_t w, size_t h, size_t inner, long instances);

// The plan is a pure function of (w, h, inner, instances, CUs) and its search walks O(h) candidates: memoised per host thread
// (a thread drives the lanes of one device; eager paths -- no graph, timing modes -- call this before every launch).
FusedPlan fused_plan(const flow2d_context* ctx, size_t w, size_t h, size_t inner, long instances)
{
    thread_local std::map<std::array<long, 5>, FusedPlan> cache;
    const std::array<long, 5> key{(long)w, (long)h, (long)inner, instances, (long)ctx->num_cus};
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() > 4096) cache.clear();
        it = cache.emplace(key, fused_plan_search(ctx, w, h, inner, instances)).first;
    }
    return it->second;
}

static FusedPlan fused_plan_search(const flow2d_context* ctx, size_t w, size_t h, size_t inner, long instances)
{
    const int valid = probe::kNoHalo ? 64 : 64 - 2 * ((int)inner + 1);
    const long blocks_x = (div_up(w, valid) + 3) / 4;
    const long cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const long cap = cus * 2;
    const int peel = 3 + 2 * (int)inner;  // run_strip's start-up steps
    double saved = 2 * 102.0 + 3 * 20.0;
    for (int k = 1; k <= (int)inner; ++k) saved += 46.0 * std::min(3 + 2 * k, peel);
    const double halo = (double)(2 * (long)inner + 3) - saved / (138.0 + 46.0 * (double)inner);  // the last ring turn is partial
    const long batch = instances;  // the instances of a batched launch share the chip
    // What a plan costs: the launch's duration on an empty chip (rounds of waves, below) PLUS kWorkWeight x its total work in
    // rounds of the full chip.  The first term alone gives a small level many short strips -- every wave slot filled, and every
    // strip paying its 2 inner + 3 start-up rows again -- which is the fastest lone launch, but in a pipeline of lanes the chip
    // is saturated by the other lanes' launches and the redundant rows are vector instructions somebody waits for.  Weights
    // 0 / 1 / 2 / 3 on one box (profiles/r04_experiments/plan_work_weight_ab.txt): config 2 3 993-4 103 / 4 107-4 120 / 4 199-4 211 /
    // 4 220-4 266 pairs/s, rub1-rub2 1 535-1 540 / 1 586-1 591 / 1 622-1 630 / 1 622-1 625, configs 3, 4, 5 unchanged; a lone 1080p
    // pair 1.10 / 1.09 / 1.11 / 1.14 ms, a lone 1024^2 pair 0.67 / 0.65 / 0.65 / 0.67 ms: two.  Round 6, far beyond (where a 4096^2
    // launch turns into half a round of twice as long strips): 2 / 6 / 24 / 100 -> config 3 350-351 / 349-350 / 347-348 / 333-335,
    // config 5 100.1-100.3 / 100.6-100.7 / 100.0-100.5 / 97.2-97.6 pairs/s, a lone config-3 pair 3.46-3.50 / 3.65-3.69 / 3.92-3.94 / 4.88 ms.
#ifdef FLOW2D_DEV_BUILD
    static const double bias = std::getenv("FLOW2D_PLAN_BIAS") ? std::atof(std::getenv("FLOW2D_PLAN_BIAS")) : 2.0;
#else
    const double bias = 2.0;
#endif
    auto rounds = [&](long blocks, double slowest) {
        const long full = blocks / cap, rem = blocks % cap;
        return full * 2.0 * slowest + (rem == 0 ? 0.0 : (rem <= cus ? 1.3 * slowest : 2.0 * slowest)) +
               bias * 2.0 * slowest * (double)blocks / (double)cap;
    };
    double best = 1e300;
    FusedPlan plan{1, 1, (int)h, (int)blocks_x, (int)(blocks_x * (long)h)};
    for (long ny = 1; ny <= (long)h; ++ny) {  // uniform strips
        const long rows = (long)((h + ny - 1) / ny);
        if ((long)((h + rows - 1) / rows) != ny) continue;  // same ny reachable with fewer rows: skip duplicates
        const double cost = rounds(blocks_x * ny * batch, kEdgeCost * ((double)rows + halo));
        if (cost < best - 1e-9) best = cost, plan = FusedPlan{(int)rows, (int)rows, (int)ny, (int)blocks_x, (int)(blocks_x * ny)};
    }
    for (long ny = 3; blocks_x >= 3 && ny <= (long)h / 2; ++ny) {  // border-aware: ny strips per interior column
        // rows_edge = (rows_interior + halo) / kEdgeCost - halo and 2 rows_edge + (ny - 2) rows_interior = h
        const double ri_real = ((double)h + 2.0 * halo - 2.0 * halo / kEdgeCost) / ((double)(ny - 2) + 2.0 / kEdgeCost);
        long re = (long)std::floor((ri_real + halo) / kEdgeCost - halo);
        if (re < 1 || 2 * re >= (long)h) continue;
        // (the kernel sends a strip to the border body when it ends within halo + 4 rows of the image's last row -- its interior
        //  body prefetches unclamped --, so with border strips that short the second strip from the bottom would run the border
        //  body at interior length and outlast the plan's estimate: ADVICE r05)
        if (re <= (long)inner + 1 + 3) continue;
        const long ri = ((long)h - 2 * re + (ny - 3)) / (ny - 2);
        if (ri < re) continue;
        const long ny_edge = ((long)h + re - 1) / re;
        const long blocks = ((blocks_x - 2) * ny + 2 * ny_edge) * batch;
        const double slowest = std::max((double)ri + halo, kEdgeCost * ((double)re + halo));
        const double cost = rounds(blocks, slowest);
        if (cost < best - 1e-9)
            best = cost, plan = FusedPlan{(int)ri, (int)re, (int)ny, (int)blocks_x, (int)((blocks_x - 2) * ny + 2 * ny_edge)};
    }
    return plan;
}

// The launch order of a plan's blocks (fused_block_of, solve_fused_args.hpp): every XCD a contiguous run of the plan's order, and
// the side blocks of a border-aware plan dealt evenly over those runs -- as long as every run has room for its share.
void fused_order_for(const FusedPlan& plan, FusedArgs& a)
{
    a.blocks_per_xcd = (plan.blocks + 7) / 8;
    a.side_blocks = 0;
    if (plan.rows_interior != plan.rows_edge && !probe::kSideLast) {
        const int side = plan.blocks - (plan.blocks_x - 2) * plan.strips_interior;
        const int last_run = plan.blocks - 7 * a.blocks_per_xcd;
        if (side > 0 && last_run >= (side + 7) / 8) a.side_blocks = side;
    }
}

// Rows [y0, y1) of the strip `by` in block column `bx` (what the kernel computes for its waves)
void fused_rows_of(const FusedArgs& a, int bx, int by, int& y0, int& y1)
{
    const bool uniform = a.rows_interior == a.rows_edge;
    if (uniform || bx == 0 || bx == a.blocks_x - 1) {
        y0 = by * a.rows_edge;
        y1 = std::min(y0 + a.rows_edge, a.h);
    } else if (by == 0) {
        y0 = 0, y1 = a.rows_edge;
    } else if (by == a.strips_interior - 1) {
        y0 = a.h - a.rows_edge, y1 = a.h;
    } else {
        y0 = a.rows_edge + (by - 1) * a.rows_interior;
        y1 = std::min(y0 + a.rows_interior, a.h - a.rows_edge);
    }
}

// One outer iteration: reads du/dv (previous outer iteration), writes out_du/out_dv (after `inner` sweeps).
int launch_fused_outer(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                       const float* v, const float* du, const float* dv, size_t w, size_t h, size_t pitch_bytes,
                       float hx, float hy, float alpha, float e_smooth, float e_data, size_t inner, float* out_du,
                       float* out_dv, int rows_per_strip, bool zero_increment, const float* start_du,
                       const float* start_dv, float sor_omega)
{
    // sor_omega != 0: the `inner` stages are red-black half-sweeps (2 or 4: one or two iterations per launch)
    if (!fused_supports(inner) || !fused_addressable(h, pitch_bytes)) return FLOW2D_ERR_UNSUPPORTED;
    if (sor_omega != 0.f && (inner != 2 && inner != 4)) return FLOW2D_ERR_UNSUPPORTED;
    if (!fused_weights_ok(hx, hy, alpha)) return FLOW2D_ERR_UNSUPPORTED;
    // rows_per_strip > 0: uniform strips of that height (developer override); 0: the planner's choice
    // A lock-step group whose every instance fills the chip on its own with long strips (128 rows and more: 4096^2 and
    // up) is launched instance by instance: nothing is gained by one launch of several rounds, and the strips are then
    // planned -- and show in a kernel trace -- exactly as for a single pair.  Smaller levels share a launch (grid.z),
    // which lets the planner give them longer strips.
    const bool split = ctx->batch_count > 1 && fused_plan(ctx, w, h, inner, 1).rows_interior >= 128;
    const unsigned instances_per_launch = split ? 1u : ctx->batch_count;
    FusedPlan plan = fused_plan(ctx, w, h, inner, (long)instances_per_launch);
    if (rows_per_strip > 0)
        plan = FusedPlan{rows_per_strip, rows_per_strip, (int)div_up(h, rows_per_strip), plan.blocks_x,
                         plan.blocks_x * (int)div_up(h, rows_per_strip)};
    FusedArgs a{f0, f1, u, v, du, dv, out_du, out_dv, (int)w, (int)h, (int)(pitch_bytes / 4), plan.rows_interior,
                plan.rows_edge, plan.strips_interior, plan.blocks_x,
                zero_increment ? 1 : 0, start_du, start_dv, (start_du && start_dv) ? 1 : 0, hx, hy, alpha, e_smooth,
                e_data,
                2.f * hx, 2.f * hy, 4.f * hx, 4.f * hy, 1.f / (2.f * hx), 1.f / (2.f * hy), 1.f / (4.f * hx), 1.f / (4.f * hy),
                static_cast<float>(1.0 / (2.0 * hx)), static_cast<float>(1.0 / (2.0 * hy)), alpha / (hx * hx), alpha / (hy * hy),
                0.5f * (alpha / (hx * hx)), 0.5f * (alpha / (hy * hy)),
                sor_omega, 1.f - sor_omega,
                0, plan.blocks, 0, 0, static_cast<unsigned long long>(ctx->batch_stride_floats),
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                ctx->fused_fallbacks};
    // The first outer iteration of a level starts from du = dv = 0: the kernel loads the increment planes all the same (no branch
    // around two loads in every row step) and selects the zero -- so let those loads go to (u, v), whose lines the same step has just
    // fetched, instead of dragging two planes of stale data through HBM (a 4096^2 launch 583 -> 450 MB; round 6).
    if (zero_increment) a.du = u, a.dv = v;
#ifdef FLOW2D_DEV_BUILD
    if (probe::kStamps) a.stamps = stamp_buffer(), a.stamp_count = stamp_counter(), a.stalls = stall_buffer();
    const size_t plane_floats = h * (pitch_bytes / 4);
    if (probe::kPackedPlanes) {
        if (ctx->batch_count > 1 || a.continue_sweeps || plane_floats * 16 > 0xffffffffull) return FLOW2D_ERR_UNSUPPORTED;
        PackedPlanes& pp = packed_planes(plane_floats);
        a.pack_in = pp.in, a.pack_duv = pp.duv, a.pack_out = pp.out;
        pack_planes_kernel<<<(unsigned)div_up(plane_floats, 256), 256, 0, ctx->stream>>>(f0, f1, u, v, du, dv, reinterpret_cast<float4*>(pp.in),
                                                                                    reinterpret_cast<float2*>(pp.duv), plane_floats);
    }
#endif
    {  // 2h and 4h as three-step divisors (non-power-of-two spacings): within [2^-30, 2^40] like every guarded denominator
        const float lo = std::min(a.two_hx, a.two_hy), hi = std::max(a.four_hx, a.four_hy);
        if (!(lo >= 0x1p-30f && hi <= 0x1p40f)) a.plain_only = 1;
    }
    // XCD-aware block order: x-adjacent blocks share their halo columns, y-adjacent strips their halo rows; in one XCD
    // they meet in its L2 (reads of a 4096^2 launch 541 -> 455 MB; worth 1-3 % of the launch since the round-3 kernel
    // is within reach of the memory system).
    fused_order_for(plan, a);
    const dim3 grid(a.blocks_per_xcd ? a.blocks_per_xcd * 8 : plan.blocks, 1, instances_per_launch);
    const bool pow2 = is_power_of_two(hx) && is_power_of_two(hy);
    int rc = 0;
    for (unsigned first = 0; first < ctx->batch_count && rc == 0; first += instances_per_launch) {
    if (first) {  // the next instance of a split group: every plane one batch stride further
        const size_t off = static_cast<size_t>(ctx->batch_stride_floats) * instances_per_launch;
        a.f0 += off, a.f1 += off, a.u += off, a.v += off, a.du += off, a.dv += off, a.out_du += off, a.out_dv += off;
        if (a.continue_sweeps) a.start_du += off, a.start_dv += off;
    }
    const int in = static_cast<int>(inner);
    // A context that runs alone (flow2d_context_set_lone) and a launch of at most one workgroup per CU -- one wave per SIMD: nothing
    // to share issue turns with -- take the build with packed arithmetic: fewer, wider instructions (the same IEEE operations).
    const bool packed_build = ctx->lone && (long)plan.blocks * (long)instances_per_launch <= (long)(ctx->num_cus > 0 ? ctx->num_cus : 256);
    rc = 1;
    if (packed_build && constancy == FLOW2D_CONSTANCY_GRADIENT)
        rc = pow2 ? fused_launch_g1_p1_k(in, grid, ctx->stream, a) : fused_launch_g1_p0_k(in, grid, ctx->stream, a);
    else if (packed_build && constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED)
        rc = pow2 ? fused_launch_g2_p1_k(in, grid, ctx->stream, a) : fused_launch_g2_p0_k(in, grid, ctx->stream, a);
    else if (packed_build && constancy != FLOW2D_CONSTANCY_LOG_DERIVATIVES)
        rc = pow2 ? fused_launch_g0_p1_k(in, grid, ctx->stream, a) : fused_launch_g0_p0_k(in, grid, ctx->stream, a);
    if (rc == 0)  // (the packed build ran; it holds no kernels for continued sweeps: those fall through to the pipeline's build)
        ;
    else if (constancy == FLOW2D_CONSTANCY_GRADIENT)
        rc = pow2 ? fused_launch_g1_p1(in, grid, ctx->stream, a) : fused_launch_g1_p0(in, grid, ctx->stream, a);
    else if (constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED)
        rc = pow2 ? fused_launch_g2_p1(in, grid, ctx->stream, a) : fused_launch_g2_p0(in, grid, ctx->stream, a);
    else if (constancy == FLOW2D_CONSTANCY_LOG_DERIVATIVES)
        rc = pow2 ? fused_launch_g3_p1(in, grid, ctx->stream, a) : fused_launch_g3_p0(in, grid, ctx->stream, a);
    else
        rc = pow2 ? fused_launch_g0_p1(in, grid, ctx->stream, a) : fused_launch_g0_p0(in, grid, ctx->stream, a);
    }
    if (rc) return FLOW2D_ERR_UNSUPPORTED;
#ifdef FLOW2D_DEV_BUILD
    if (probe::kPackedPlanes)
        unpack_planes_kernel<<<(unsigned)div_up(plane_floats, 256), 256, 0, ctx->stream>>>(reinterpret_cast<const float2*>(a.pack_out), out_du, out_dv,
                                                                                      plane_floats);
#endif
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

}  // namespace flow2d

// Diagnostics: the blocks a strip launch of this geometry would run, in launch order -- (block column, strip, first row, end row)
// per launch block id, -1 for ids beyond the plan (the grid is a multiple of eight).  tests/test_gpu_fused.py checks that the order
// is a permutation of the plan and the rows a partition of the image.
extern "C" FLOW2D_API int flow2d_fused_block_order(flow2d_context* ctx, size_t width, size_t height, size_t inner, size_t instances,
                                                   int* out, size_t capacity_blocks, size_t* grid_blocks)
{
    if (!ctx || !out || !grid_blocks || !flow2d::fused_supports(inner) || width == 0 || height == 0 || instances == 0)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    const flow2d::FusedPlan plan = flow2d::fused_plan(ctx, width, height, inner, (long)instances);
    flow2d::FusedArgs a{};
    a.w = (int)width, a.h = (int)height;
    a.rows_interior = plan.rows_interior, a.rows_edge = plan.rows_edge, a.strips_interior = plan.strips_interior, a.blocks_x = plan.blocks_x;
    a.blocks = plan.blocks;
    flow2d::fused_order_for(plan, a);
    const size_t grid = (size_t)a.blocks_per_xcd * 8;
    *grid_blocks = grid;
    if (grid > capacity_blocks) return FLOW2D_ERR_INVALID_ARGUMENT;
    for (size_t id = 0; id < grid; ++id) {
        int bx = -1, by = -1, y0 = -1, y1 = -1;
        if (flow2d::fused_block_of(a, (int)id, bx, by)) flow2d::fused_rows_of(a, bx, by, y0, y1);
        out[4 * id + 0] = bx, out[4 * id + 1] = by, out[4 * id + 2] = y0, out[4 * id + 3] = y1;
    }
    return FLOW2D_OK;
}

#ifdef FLOW2D_DEV_BUILD
// developer builds only: the wave stamps recorded since the last call (kStampWords words per wave), newest launches last, and -- when
// stalls is not null -- the stall histograms of the first min(waves, kStallWaves) of them (kStallRows x 64 words per wave)
extern "C" FLOW2D_API int flow2d_dev_fused_stamps(unsigned long long* out, size_t max_waves, size_t* waves, unsigned int* stalls)
{
    namespace probe = flow2d_probe;
    if (!probe::kStamps) return FLOW2D_ERR_UNSUPPORTED;
    unsigned int n = 0;
    if (hipMemcpy(&n, flow2d::stamp_counter(), sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return FLOW2D_ERR_DEVICE;
    const size_t take = std::min<size_t>(std::min<size_t>(n, probe::kStampWaves), max_waves);
    if (take && hipMemcpy(out, flow2d::stamp_buffer(), take * probe::kStampWords * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return FLOW2D_ERR_DEVICE;
    const size_t hist = std::min<size_t>(take, probe::kStallWaves);
    if (stalls && hist &&
        hipMemcpy(stalls, flow2d::stall_buffer(), hist * probe::kStallRows * 64 * sizeof(unsigned int), hipMemcpyDeviceToHost) != hipSuccess)
        return FLOW2D_ERR_DEVICE;
    (void)hipMemset(flow2d::stamp_counter(), 0, sizeof(unsigned int));
    *waves = take;
    return FLOW2D_OK;
}
#endif
