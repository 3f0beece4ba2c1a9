// The solver's fixed-point loop of one pyramid level and its launch timing.
// Replaces the launch loop of CudaOperationSolve2D::Execute
// (src/cuda_operations/2d/cuda_operation_solve_2d.cpp:229-300) without its per-sweep host
// synchronisation (:291): everything is queued on the context's stream.
#include <algorithm>
#include <cstdlib>
#include <utility>

#include "common.hpp"

namespace flow2d {
int launch_phi_ksi(flow2d_context* ctx, const float* f0, const float* f1, const float* u, const float* v,
                   const float* du, const float* dv, size_t w, size_t h, size_t pitch_bytes, float hx, float hy,
                   float e_smooth, float e_data, float* phi, float* ksi);
int launch_sweep(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u, const float* v,
                 const float* du, const float* dv, const float* phi, const float* ksi, size_t w, size_t h,
                 size_t pitch_bytes, float hx, float hy, float alpha, float* tdu, float* tdv);
int launch_sor_iteration(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                         const float* v, float* du, float* dv, const float* phi, const float* ksi, size_t w, size_t h,
                         size_t pitch_bytes, float hx, float hy, float alpha, float omega);
bool small_level_supports(size_t w, size_t h);
int launch_small_level(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                       const float* v, size_t w, size_t h, size_t pitch_bytes, float hx, float hy, float alpha,
                       float e_smooth, float e_data, size_t outer, size_t inner, float* out_du, float* out_dv);
bool fused_supports(size_t inner);
bool fused_addressable(size_t h, size_t pitch_bytes);
bool fused_weights_ok(float hx, float hy, float alpha);
bool tiled_supports(int constancy, size_t inner);
int launch_tiled_outer(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                       const float* v, const float* du, const float* dv, size_t w, size_t h, size_t pitch_bytes, float hx,
                       float hy, float alpha, float e_smooth, float e_data, size_t inner, float* out_du, float* out_dv,
                       bool zero_increment, float sor_omega);

int launch_fused_outer(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                       const float* v, const float* du, const float* dv, size_t w, size_t h, size_t pitch_bytes,
                       float hx, float hy, float alpha, float e_smooth, float e_data, size_t inner, float* out_du,
                       float* out_dv, int rows_per_strip, bool zero_increment, const float* start_du,
                       const float* start_dv, float sor_omega);
}  // namespace flow2d

namespace {
// mode-2 timing: one more event on the stream, appended to the slot's start/stop list
hipError_t take_event(flow2d_context* ctx, hipEvent_t* ev)
{
    if (!ctx->event_pool.empty()) {
        *ev = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return hipSuccess;
    }
    return hipEventCreate(ev);
}

hipError_t mark(flow2d_context* ctx, flow2d_timing_slot* slot)
{
    hipEvent_t ev;
    hipError_t e = take_event(ctx, &ev);
    if (e != hipSuccess) return e;
    slot->kernel_events.push_back(ev);
    return hipEventRecord(ev, ctx->stream);
}
}  // namespace

// largest level (pixels) AUTO gives to the tiled kernel (developer builds, -DFLOW2D_DEV_BUILD: the environment variable
// FLOW2D_TILED_MAX_PIXELS overrides; the product library reads no environment variable here).  Measured
// against the strips as they are at the end of round 3 (tools/time_levels.py; level solve 10 x 5, tiles / strips, ms):
// Grey 384^2 0.111 / 0.115, 512^2 0.119 / 0.165, 640^2 0.193 / 0.170, 768^2 0.253 / 0.181, 1024^2 0.340 / 0.260;
// Gradient 512^2 0.127 / 0.183, 640^2 0.207 / 0.182, 1024^2 0.366 / 0.280, 1920 x 1080 0.679 / 0.393.
static size_t tiled_max_pixels(int data_constancy)
{
#ifdef FLOW2D_DEV_BUILD
    static const long long forced = std::getenv("FLOW2D_TILED_MAX_PIXELS") ? std::atoll(std::getenv("FLOW2D_TILED_MAX_PIXELS")) : -1;
    if (forced >= 0) return static_cast<size_t>(forced);
#endif
    (void)data_constancy;  // the same crossover for the brightness and the gradient terms
    return static_cast<size_t>(600) * 600;
}

// `instances`: pairs of a lock-step group that share every launch (flow2d_context_set_batch; 1 for a single pair)
static int solver_algorithm_for(int requested, size_t width, size_t height, size_t pitch_bytes, size_t outer, size_t inner,
                                int data_constancy, size_t instances)
{
    if (requested < FLOW2D_SOLVER_AUTO || requested > FLOW2D_SOLVER_TILED) return -1;
    if (requested == FLOW2D_SOLVER_AUTO) {
        // Up to tiled_max_pixels (600^2) the outer iteration runs on small LDS tiles (solve_tile.hip): a strip wave
        // needs (rows + halo) x ~1 us whatever the level size, tiles spread a small level over the whole chip
        // (level solve 10 x 5 at 256^2: 0.07 against 0.11 ms, at 512^2 0.12 against 0.17, at 640^2 0.19 against 0.17).
        // That includes the coarsest levels: ten launches of a few
        // 8 x 8 tiles take 0.05 ms at 16 x 16 ... 64 x 32, the single-workgroup kernel 0.06 ... 0.11 ms.
        // A lock-step group moves the crossover down with the square of its size: its instances share the strips' launch
        // (longer strips, the planner sees all of them) and, in a pipeline of several lanes, a few strip waves find room
        // beside another lane's kernels where 1024-thread tile workgroups with their LDS wait (config 2, groups of four:
        // 3 416-3 439 pairs/s with the tiles up to 600^2 per instance, 3 524-3 567 up to 64^2 ... 128^2; config 4, groups
        // of eight: 2 168-2 197 -> 2 224-2 239; groups of two at 4096^2: no difference).
        if (inner >= 2 && width * height * instances * instances <= tiled_max_pixels(data_constancy) &&
            flow2d::tiled_supports(data_constancy, inner))
            return FLOW2D_SOLVER_TILED;
        // Where the tiled kernel does not apply (solve_2d_log, a single sweep per outer iteration), levels up to 64 x 32
        // run whole in one launch on one CU (solve_small.hip; 0.06-0.10 ms against 0.16-0.19 ms for 60 per-sweep launches;
        // at 64 x 64 four pixels per thread spill and lose).
        if (flow2d::small_level_supports(width, height) && height <= 32) return FLOW2D_SOLVER_SINGLE_WORKGROUP;
        // Everything else goes through the fused strip kernel (above 512^2 it wins over per-sweep launches by 1.7-2.9x).
        // A single sweep per outer iteration leaves nothing to fuse, and a plane of 4 GiB or more is beyond the fused
        // kernel's 32-bit buffer offsets: both take the per-sweep kernels.
        return (inner >= 2 && flow2d::fused_addressable(height, pitch_bytes)) ? FLOW2D_SOLVER_FUSED : FLOW2D_SOLVER_PER_SWEEP;
    }
    if (requested == FLOW2D_SOLVER_TILED && !flow2d::tiled_supports(data_constancy, inner)) return -1;
    if (requested == FLOW2D_SOLVER_SINGLE_WORKGROUP && !flow2d::small_level_supports(width, height)) return -1;
    if (requested == FLOW2D_SOLVER_FUSED) {
        if (inner == 0 && outer != 0) return -1;  // nothing to fuse: an outer iteration without sweeps leaves du, dv as they are
        if (!flow2d::fused_addressable(height, pitch_bytes)) return -1;
    }
    return requested;
}

extern "C" {

// The algorithm flow2d_solve_level runs for a request (never AUTO) on a single pair, or -1 when the requested one cannot
// run the level.  Pure host logic, no device needed.  (In a lock-step group AUTO gives fewer levels to the tiles: above.)
int flow2d_solver_algorithm_for(int requested, size_t width, size_t height, size_t pitch_bytes, size_t outer, size_t inner,
                                int data_constancy)
{
    return solver_algorithm_for(requested, width, height, pitch_bytes, outer, inner, data_constancy, 1);
}

int flow2d_solve_level(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                       const float* flow_v, float* flow_du, float* flow_dv, float* phi, float* ksi, float* temp_du,
                       float* temp_dv, const flow2d_solve_params* p, int* result_in_temp)
{
    FLOW2D_ENTER(ctx);
    if (!p || !result_in_temp) return FLOW2D_ERR_INVALID_ARGUMENT;
    const float* planes[] = {frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi, ksi, temp_du, temp_dv};
    for (int i = 0; i < 10; ++i) {
        if (!flow2d::plane_args_ok(planes[i], p->width, p->height, p->pitch_bytes)) return FLOW2D_ERR_INVALID_ARGUMENT;
        for (int j = 4; j < 10; ++j)  // the six written planes must be distinct from everything else
            if (i != j && planes[i] == planes[j]) return FLOW2D_ERR_INVALID_ARGUMENT;
    }
    if (p->width < 2 || p->height < 2 || p->container_height < p->height || !(p->hx > 0.f) || !(p->hy > 0.f))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (p->data_constancy != FLOW2D_CONSTANCY_GREY && p->data_constancy != FLOW2D_CONSTANCY_GRADIENT &&
        p->data_constancy != FLOW2D_CONSTANCY_GRADIENT_UNTILED && p->data_constancy != FLOW2D_CONSTANCY_LOG_DERIVATIVES)
        return FLOW2D_ERR_UNSUPPORTED;
    if (p->algorithm < FLOW2D_SOLVER_AUTO || p->algorithm > FLOW2D_SOLVER_TILED)
        return FLOW2D_ERR_INVALID_ARGUMENT;

    const bool sor = p->sor_omega != 0.f;
    if (sor && (!(p->sor_omega > 0.f) || !(p->sor_omega < 2.f))) return FLOW2D_ERR_INVALID_ARGUMENT;
    if (sor && p->data_constancy == FLOW2D_CONSTANCY_LOG_DERIVATIVES) return FLOW2D_ERR_UNSUPPORTED;
    int algorithm = solver_algorithm_for(p->algorithm, p->width, p->height, p->pitch_bytes, p->outer_iterations_count,
                                         p->inner_iterations_count, p->data_constancy, ctx->batch_count);
    // Red-black SOR (opt-in): temporally blocked in the strip kernel and in the LDS tiles with half-sweeps for stages (round 5: one
    // launch per two iterations instead of a phi / ksi launch and two half-sweep launches per iteration).  AUTO picks tiles or
    // strips by the level's size like for Jacobi (the tiles up to two iterations per outer iteration; the single-workgroup kernel
    // has no such stages); only what the strips cannot address, or an explicit request, takes the half-sweep launches.
    if (sor) {
        // stages of a launch = half-sweeps: the tiles hold up to four (two iterations), the strips chain launches of four
        const size_t stages = 2 * p->inner_iterations_count;
        const bool can_fuse = p->inner_iterations_count >= 1 && flow2d::fused_addressable(p->height, p->pitch_bytes);
        const bool can_tile = p->inner_iterations_count >= 1 && stages <= 4 && flow2d::tiled_supports(p->data_constancy, stages);
        const bool auto_tiles = can_tile && solver_algorithm_for(FLOW2D_SOLVER_AUTO, p->width, p->height, p->pitch_bytes,
                                                                 p->outer_iterations_count, stages, p->data_constancy,
                                                                 ctx->batch_count) == FLOW2D_SOLVER_TILED;
        if (p->algorithm == FLOW2D_SOLVER_FUSED && !can_fuse) return FLOW2D_ERR_UNSUPPORTED;
        if (p->algorithm == FLOW2D_SOLVER_TILED && !can_tile) return FLOW2D_ERR_UNSUPPORTED;
        if (p->algorithm == FLOW2D_SOLVER_SINGLE_WORKGROUP) return FLOW2D_ERR_UNSUPPORTED;
        if (p->algorithm == FLOW2D_SOLVER_AUTO)
            algorithm = auto_tiles ? FLOW2D_SOLVER_TILED : (can_fuse ? FLOW2D_SOLVER_FUSED : FLOW2D_SOLVER_PER_SWEEP);
        else
            algorithm = p->algorithm;
    }
    if (algorithm < 0) return FLOW2D_ERR_UNSUPPORTED;
    // the strips multiply with half the neighbour weight alpha / h^2, which must be exactly representable (solve_fused.hip)
    if (algorithm == FLOW2D_SOLVER_FUSED && !flow2d::fused_weights_ok(p->hx, p->hy, p->equation_alpha)) {
        if (p->algorithm == FLOW2D_SOLVER_FUSED) return FLOW2D_ERR_UNSUPPORTED;
        algorithm = FLOW2D_SOLVER_PER_SWEEP;
    }

    flow2d_timing_slot* slot = nullptr;
    if (ctx->timing) {
        flow2d_timing_slot s{};
        FLOW2D_HIP_TRY(take_event(ctx, &s.start));
        FLOW2D_HIP_TRY(take_event(ctx, &s.stop));
        ctx->timings.push_back(s);
        slot = &ctx->timings.back();
        FLOW2D_HIP_TRY(hipEventRecord(slot->start, ctx->stream));
    }

    // du = dv = 0 over level width x container height (cuda_operation_solve_2d.cpp:229-232).  The fused path
    // starts its first outer iteration from zero increments without reading the planes, so it needs no memset.
    const bool one_per_outer = algorithm == FLOW2D_SOLVER_FUSED || algorithm == FLOW2D_SOLVER_TILED;
    if (algorithm == FLOW2D_SOLVER_PER_SWEEP || (one_per_outer && p->outer_iterations_count == 0)) {
        for (unsigned b = 0; b < ctx->batch_count; ++b) {
            FLOW2D_HIP_TRY(hipMemset2DAsync(flow_du + b * ctx->batch_stride_floats, p->pitch_bytes, 0,
                                            p->width * sizeof(float), p->container_height, ctx->stream));
            FLOW2D_HIP_TRY(hipMemset2DAsync(flow_dv + b * ctx->batch_stride_floats, p->pitch_bytes, 0,
                                            p->width * sizeof(float), p->container_height, ctx->stream));
        }
    }

    const bool per_launch = slot && ctx->timing >= 2 && p->width >= ctx->timing_min_w && p->height >= ctx->timing_min_h;
    float* du = flow_du;
    float* dv = flow_dv;
    float* tdu = temp_du;
    float* tdv = temp_dv;
    int launches = 0;
    if (algorithm == FLOW2D_SOLVER_SINGLE_WORKGROUP) {
        if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
        int st = flow2d::launch_small_level(ctx, p->data_constancy, frame_0, frame_1, flow_u, flow_v, p->width, p->height,
                                            p->pitch_bytes, p->hx, p->hy, p->equation_alpha, p->equation_smoothness,
                                            p->equation_data, p->outer_iterations_count, p->inner_iterations_count,
                                            du, dv);
        if (st != FLOW2D_OK) return st;
        if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
        ++launches;
    }
#ifdef FLOW2D_DEV_BUILD  // uniform strips of that many rows instead of the planner's choice
    static const int rows_env = std::getenv("FLOW2D_FUSED_ROWS") ? std::atoi(std::getenv("FLOW2D_FUSED_ROWS")) : 0;
#else
    const int rows_env = 0;
#endif
    // Fused path: one launch per outer iteration does phi/ksi and up to 5 sweeps (solve_fused_kernel.hpp); phi and ksi
    // are not materialised, so their planes serve as a third (du, dv) pair.  More than 5 sweeps per outer
    // iteration are split into equal chunks: every chunk rebuilds the coefficients from the outer iteration's
    // starting pair (same arithmetic, same values) and continues the sweeps from the previous chunk's result.
    float* pair_u[3] = {flow_du, temp_du, phi};
    float* pair_v[3] = {flow_dv, temp_dv, ksi};
    int source = 0;  // pair holding du, dv at the start of the outer iteration
    const size_t inner = p->inner_iterations_count;
    // (red-black iterations: two half-sweep stages each, at most two iterations per launch)
    const size_t per_launch_max = sor ? 2 : 5;
    const size_t chunks = algorithm == FLOW2D_SOLVER_FUSED ? std::max<size_t>(1, (inner + per_launch_max - 1) / per_launch_max) : 0;
    for (size_t i = 0; algorithm == FLOW2D_SOLVER_FUSED && i < p->outer_iterations_count; ++i) {
        int in = source;
        for (size_t c = 0; c < chunks; ++c) {
            const size_t sweeps = inner / chunks + (c < inner % chunks ? 1 : 0);
            int out = 0;
            while (out == source || out == in) ++out;
            const int rows = rows_env > 0 ? rows_env : 0;  // 0: the launcher plans the strip heights (solve_fused.hip, FusedPlan)
            if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
            int st = flow2d::launch_fused_outer(ctx, p->data_constancy, frame_0, frame_1, flow_u, flow_v, pair_u[source],
                                                pair_v[source], p->width, p->height, p->pitch_bytes, p->hx, p->hy,
                                                p->equation_alpha, p->equation_smoothness, p->equation_data,
                                                sor ? 2 * sweeps : sweeps, pair_u[out], pair_v[out], rows, i == 0,
                                                c == 0 ? nullptr : pair_u[in], c == 0 ? nullptr : pair_v[in],
                                                sor ? p->sor_omega : 0.f);
            if (st != FLOW2D_OK) return st;
            if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
            in = out;
            ++launches;
        }
        source = in;
    }
    // Tiled path: one launch per outer iteration, ping-pong between the caller's two pairs.  (Round 5 tried TWO outer
    // iterations per launch of the 8 x 8 and 16 x 16 tiles over a halo of 2 (inner + 1) pixels -- half the launches of the
    // launch-bound levels: the regions of 32 x 32 / 40 x 40 pixels recompute 16x / 6x the tile and a lone 1024^2 pair took 0.76
    // instead of 0.71 ms, a lone 4096^2 pair 4.24 instead of 4.08-4.17, the pipelined rates 1 % less:
    // profiles/r05_experiments/tile_two_outer_ab.txt.  Not kept.)
    for (size_t i = 0; algorithm == FLOW2D_SOLVER_TILED && i < p->outer_iterations_count; ++i) {
        const int out = source == 0 ? 1 : 0;
        if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
        int st = flow2d::launch_tiled_outer(ctx, p->data_constancy, frame_0, frame_1, flow_u, flow_v, pair_u[source],
                                            pair_v[source], p->width, p->height, p->pitch_bytes, p->hx, p->hy,
                                            p->equation_alpha, p->equation_smoothness, p->equation_data,
                                            sor ? 2 * inner : inner, pair_u[out], pair_v[out], i == 0, sor ? p->sor_omega : 0.f);
        if (st != FLOW2D_OK) return st;
        if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
        source = out;
        ++launches;
    }
    if (algorithm == FLOW2D_SOLVER_FUSED && source == 2) {  // the caller only knows two pairs: hand the result over
        for (unsigned b = 0; b < ctx->batch_count; ++b) {
            const size_t off = b * ctx->batch_stride_floats;
            FLOW2D_HIP_TRY(hipMemcpy2DAsync(flow_du + off, p->pitch_bytes, phi + off, p->pitch_bytes,
                                            p->width * sizeof(float), p->height, hipMemcpyDeviceToDevice, ctx->stream));
            FLOW2D_HIP_TRY(hipMemcpy2DAsync(flow_dv + off, p->pitch_bytes, ksi + off, p->pitch_bytes,
                                            p->width * sizeof(float), p->height, hipMemcpyDeviceToDevice, ctx->stream));
        }
        source = 0;
    }
    if (one_per_outer && source == 1) {
        std::swap(du, tdu);
        std::swap(dv, tdv);
    }
    for (size_t i = 0; algorithm == FLOW2D_SOLVER_PER_SWEEP && i < p->outer_iterations_count; ++i) {
        int st = flow2d::launch_phi_ksi(ctx, frame_0, frame_1, flow_u, flow_v, du, dv, p->width, p->height,
                                        p->pitch_bytes, p->hx, p->hy, p->equation_smoothness, p->equation_data, phi,
                                        ksi);
        if (st != FLOW2D_OK) return st;
        for (size_t j = 0; sor && j < p->inner_iterations_count; ++j) {  // opt-in: in place, no ping-pong
            st = flow2d::launch_sor_iteration(ctx, p->data_constancy, frame_0, frame_1, flow_u, flow_v, du, dv, phi, ksi,
                                              p->width, p->height, p->pitch_bytes, p->hx, p->hy, p->equation_alpha,
                                              p->sor_omega);
            if (st != FLOW2D_OK) return st;
            launches += 2;
        }
        for (size_t j = 0; !sor && j < p->inner_iterations_count; ++j) {
            if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
            st = flow2d::launch_sweep(ctx, p->data_constancy, frame_0, frame_1, flow_u, flow_v, du, dv, phi, ksi,
                                      p->width, p->height, p->pitch_bytes, p->hx, p->hy, p->equation_alpha, tdu, tdv);
            if (st != FLOW2D_OK) return st;
            if (per_launch) FLOW2D_HIP_TRY(mark(ctx, slot));
            std::swap(du, tdu);
            std::swap(dv, tdv);
            ++launches;
        }
    }
    *result_in_temp = (du != flow_du) ? 1 : 0;

    if (slot) {
        FLOW2D_HIP_TRY(hipEventRecord(slot->stop, ctx->stream));
        slot->rec.width = p->width;
        slot->rec.height = p->height;
        slot->rec.outer = p->outer_iterations_count;
        slot->rec.inner = p->inner_iterations_count;
        slot->rec.data_constancy = p->data_constancy;
        slot->rec.algorithm = algorithm;
        slot->rec.kernel_launches = launches;
        slot->rec.elapsed_ms = -1.f;
        slot->rec.kernel_ms = -1.f;
        double per_px = 40.0;  // one Jacobi sweep
        if (algorithm == FLOW2D_SOLVER_FUSED)  // an outer iteration's bytes, spread over its launches
            per_px = (32.0 + 40.0 * p->inner_iterations_count) / static_cast<double>(std::max<size_t>(1, chunks));
        if (algorithm == FLOW2D_SOLVER_TILED) per_px = 32.0 + 40.0 * p->inner_iterations_count;
        if (algorithm == FLOW2D_SOLVER_SINGLE_WORKGROUP)
            per_px = p->outer_iterations_count * (32.0 + 40.0 * p->inner_iterations_count);
        slot->rec.algorithmic_bytes_per_launch = per_px * static_cast<double>(p->width) * static_cast<double>(p->height);
    }
    return FLOW2D_OK;
}

int flow2d_timing_enable(flow2d_context* ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return FLOW2D_ERR_INVALID_ARGUMENT;
    ctx->timing = mode;
    return FLOW2D_OK;
}

int flow2d_timing_launch_filter(flow2d_context* ctx, size_t min_width, size_t min_height)
{
    if (!ctx) return FLOW2D_ERR_INVALID_ARGUMENT;
    ctx->timing_min_w = min_width;
    ctx->timing_min_h = min_height;
    return FLOW2D_OK;
}

int flow2d_timing_count(flow2d_context* ctx, size_t* count)
{
    if (!ctx || !count) return FLOW2D_ERR_INVALID_ARGUMENT;
    *count = ctx->timings.size();
    return FLOW2D_OK;
}

int flow2d_timing_get(flow2d_context* ctx, size_t index, flow2d_timing_record* out)
{
    FLOW2D_ENTER(ctx);
    if (!out || index >= ctx->timings.size()) return FLOW2D_ERR_INVALID_ARGUMENT;
    flow2d_timing_slot& s = ctx->timings[index];
    if (s.rec.elapsed_ms < 0.f) {
        FLOW2D_HIP_TRY(hipEventSynchronize(s.stop));
        FLOW2D_HIP_TRY(hipEventElapsedTime(&s.rec.elapsed_ms, s.start, s.stop));
        if (!s.kernel_events.empty()) {
            float total = 0.f;
            for (size_t k = 0; k + 1 < s.kernel_events.size(); k += 2) {
                float ms = 0.f;
                FLOW2D_HIP_TRY(hipEventElapsedTime(&ms, s.kernel_events[k], s.kernel_events[k + 1]));
                total += ms;
            }
            s.rec.kernel_ms = total;
        }
    }
    *out = s.rec;
    return FLOW2D_OK;
}

int flow2d_timing_reset(flow2d_context* ctx)
{
    FLOW2D_ENTER(ctx);
    for (auto& s : ctx->timings) {  // events go back to the pool (destroyed with the context)
        ctx->event_pool.push_back(s.start);
        ctx->event_pool.push_back(s.stop);
        for (hipEvent_t ev : s.kernel_events) ctx->event_pool.push_back(ev);
    }
    ctx->timings.clear();
    return FLOW2D_OK;
}

}  // extern "C"
