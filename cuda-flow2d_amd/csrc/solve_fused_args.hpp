// Arguments of the fused outer-iteration kernel (solve_fused_kernel.hpp), shared by its instance translation units
// (solve_fused_instance.hip, one object per data term and spacing kind) and the launcher (solve_fused.hip).
#pragma once

#include <hip/hip_runtime.h>

namespace flow2d {

struct FusedArgs {
    const float* f0;
    const float* f1;
    const float* u;
    const float* v;
    const float* du;
    const float* dv;
    float* out_du;
    float* out_dv;
    int w, h, pitch;
    // Strip heights (FusedPlan): a wave whose strip touches an image border runs the EDGE body, about a fifth more
    // instructions per row than an interior wave's, and a launch is one round of waves -- as long as its slowest wave.
    // So border strips are shorter: the first and the last strip of every column of strips hold rows_edge rows, the ones
    // between them rows_interior (strips_interior strips per column in all); the first and the last BLOCK in x (they
    // hold the strips on the left / right border) are cut into strips of rows_edge throughout.  rows_interior ==
    // rows_edge: uniform strips.
    int rows_interior, rows_edge, strips_interior;
    // The grid is one-dimensional over the blocks that have a strip (a two-dimensional grid would hold empty blocks,
    // and with blocks dealt to the eight XCDs in turn the working ones would pile up on some of them): block id ->
    // (block column, strip) by blocks_x, the block columns of the image.
    int blocks_x;
    int zero_increment;  // first outer iteration: du = dv = 0, the planes are not read (and need no memset)
    // More sweeps per outer iteration than one launch holds: a later launch of the same outer iteration rebuilds
    // the coefficients from the same du/dv (identical arithmetic, identical values) and continues the sweeps
    // from the previous launch's result in start_du/start_dv.
    const float* start_du;
    const float* start_dv;
    int continue_sweeps;
    float hx, hy, alpha, e_smooth, e_data;
    // Wave-uniform constants of the level, evaluated on the HOST in the reference's float / double arithmetic (the same
    // IEEE operations the kernel would perform) so that they arrive as kernel arguments in scalar registers: computed in
    // the kernel they are vector-ALU results, which the compiler broadcasts into vector register pairs, hoists out of the
    // row loop and -- the register file being full -- spills, one scratch reload per use and row step.
    float two_hx, two_hy, four_hx, four_hy;          // 2h, 4h (solve_2d.cu:141-171)
    float inv_two_hx, inv_two_hy, inv_four_hx, inv_four_hy;  // their reciprocals (exact when h is a power of two)
    float hx_1, hy_1;                                // float(1.0 / (2.0 * h)), solve_2d.cu:868-869
    float hx_2, hy_2;                                // alpha / (h * h), solve_2d.cu:337-340
    float half_hx_2, half_hy_2;                      // ... halved (exactly: the launcher checks), what stage W multiplies with
    // opt-in red-black SOR (not a reference mode): omega in (0, 2) makes the launch's `inner` stages half-sweeps; 0 = Jacobi.
    // sor_keep = 1.f - omega, evaluated on the host in float like the per-sweep kernel and the oracle do
    float sor_omega, sor_keep;
    int plain_only;                   // a grid spacing outside the range the three-step division is proven for: every wave
                                      // takes the fallback pass (plain divisions) at once
    int blocks, blocks_per_xcd;       // working blocks of the plan; ceil(blocks / 8), or 0 for the plain block order
    int side_blocks;                  // border-aware plans: the blocks of the first and last block column (the last side_blocks of
                                      // the plan's order), dealt evenly over the XCDs' runs; 0: every XCD a plain run of the order
    unsigned long long batch_stride;  // floats between the instances of a batched launch (blockIdx.z)
    // developer probes (solve_fused_probes.hpp; null in the product library, whose kernels never read them): per-wave time stamps
    // with their counter and stall histograms; the frames and the flow of a pixel as one float4 plane, the increment as float2 planes
    unsigned long long* stamps;
    unsigned int* stamp_count;
    unsigned int* stalls;
    const float* pack_in;
    const float* pack_duv;
    float* pack_out;
    unsigned int* fallback_count;     // [0] waves that repeated their strip with the plain division, [1] waves of plain_only
                                      // launches (diagnostics; may be null)
};

// Launch block id -> block of the plan (block column bx, strip by); false: the id is beyond the plan's blocks.
//
// Plan order: uniform strips row by row; a border-aware plan first the interior block columns (strips_interior strips each, row by
// row), then the first and the last block column alternately (strips of rows_edge: "side blocks").  Workgroups are dealt to the
// eight XCDs in turn, and every XCD takes a contiguous run of the order, so that x-adjacent blocks (shared halo columns) and
// y-adjacent strips (shared halo rows) meet in one L2.  Round 6: a launch lasts as long as its slowest XCD -- per-wave stamps
// (profiles/r06_experiments) show the SIMDs of one XCD finishing within 1 % of each other and the XCDs 11 % apart: their clocks
// differ by 3-4 %, and the plain run gave the last XCD ALL the side blocks, three of whose four strips are short interior
// ones.  Now every XCD's run is its share of the interior order followed by its share of the side blocks (the same rows of the
// image, so the locality stays): equal work per XCD, the clocks' spread is what remains.
__host__ __device__ inline bool fused_block_of(const FusedArgs& a, int launch_id, int& bx, int& by)
{
    int id = launch_id;
    int side = -1;  // >= 0: a side block, by its index in the order's tail
    if (a.blocks_per_xcd) {
        const int k = launch_id & 7, j = launch_id >> 3, start = k * a.blocks_per_xcd;
        if (start + j >= a.blocks) return false;
        id = start + j;
        if (a.side_blocks) {
            const int side_lo = (k * a.side_blocks) >> 3, side_hi = ((k + 1) * a.side_blocks) >> 3;
            const int run = (a.blocks - start < a.blocks_per_xcd) ? a.blocks - start : a.blocks_per_xcd;
            const int interior_run = run - (side_hi - side_lo);
            if (j < interior_run) id = start - side_lo + j;
            else side = side_lo + (j - interior_run);
        }
    } else if (id >= a.blocks) {
        return false;
    }
    const bool uniform = a.rows_interior == a.rows_edge;
    const int inner_cols = a.blocks_x - 2, inner_blocks = inner_cols * a.strips_interior;
    if (uniform) {
        bx = id % a.blocks_x, by = id / a.blocks_x;
    } else if (side < 0 && id < inner_blocks) {
        bx = 1 + id % inner_cols, by = id / inner_cols;
    } else {
        const int j = side >= 0 ? side : id - inner_blocks;
        bx = (j & 1) ? a.blocks_x - 1 : 0, by = j >> 1;
    }
    return true;
}

// One entry per instance object: launches fused_outer_kernel<inner, GRAD, POW2, CONT> (CONT from a.continue_sweeps); returns
// non-zero when the object holds no such instantiation (inner outside 1..5, or a developer build's reduced set).
#define FLOW2D_FUSED_LAUNCHER(g, p) int fused_launch_g##g##_p##p(int inner, dim3 grid, hipStream_t stream, const FusedArgs& a)
FLOW2D_FUSED_LAUNCHER(0, 0);
FLOW2D_FUSED_LAUNCHER(0, 1);
FLOW2D_FUSED_LAUNCHER(1, 0);
FLOW2D_FUSED_LAUNCHER(1, 1);
FLOW2D_FUSED_LAUNCHER(2, 0);
FLOW2D_FUSED_LAUNCHER(2, 1);
FLOW2D_FUSED_LAUNCHER(3, 0);
FLOW2D_FUSED_LAUNCHER(3, 1);
// the packed build of the same kernels (not of the log-derivative term, whose one build is packed already)
#define FLOW2D_FUSED_LONE_LAUNCHER(g, p) int fused_launch_g##g##_p##p##_k(int inner, dim3 grid, hipStream_t stream, const FusedArgs& a)
FLOW2D_FUSED_LONE_LAUNCHER(0, 0);
FLOW2D_FUSED_LONE_LAUNCHER(0, 1);
FLOW2D_FUSED_LONE_LAUNCHER(1, 0);
FLOW2D_FUSED_LONE_LAUNCHER(1, 1);
FLOW2D_FUSED_LONE_LAUNCHER(2, 0);
FLOW2D_FUSED_LONE_LAUNCHER(2, 1);

}  // namespace flow2d
