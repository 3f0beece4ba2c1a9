// One instance object of the fused outer-iteration kernel: the instantiations of ONE data term (GRAD 0..3) and ONE kind of
// grid spacing (POW2 0 / 1) -- inner 1..5, with and without continued sweeps: 20 kernels.  csrc/Makefile compiles this
// file eight times (-DFLOW2D_FUSED_INSTANCE_GRAD=g -DFLOW2D_FUSED_INSTANCE_POW2=p), so that the 80 kernels of the library
// build side by side instead of in one translation unit of four minutes.
#include "solve_fused_kernel.hpp"

#if !defined(FLOW2D_FUSED_INSTANCE_GRAD) || !defined(FLOW2D_FUSED_INSTANCE_POW2)
#error "compile with -DFLOW2D_FUSED_INSTANCE_GRAD=0..3 -DFLOW2D_FUSED_INSTANCE_POW2=0|1 (csrc/Makefile)"
#endif

namespace flow2d {

#ifdef FLOW2D_FUSED_INSTANCE_LONE  // the packed build for under-filled launches of a context that runs alone (csrc/Makefile)
#define FLOW2D_FUSED_INSTANCE_NAME2(g, p) fused_launch_g##g##_p##p##_k
#else
#define FLOW2D_FUSED_INSTANCE_NAME2(g, p) fused_launch_g##g##_p##p
#endif
#define FLOW2D_FUSED_INSTANCE_NAME(g, p) FLOW2D_FUSED_INSTANCE_NAME2(g, p)

int FLOW2D_FUSED_INSTANCE_NAME(FLOW2D_FUSED_INSTANCE_GRAD, FLOW2D_FUSED_INSTANCE_POW2)(int inner, dim3 grid, hipStream_t stream,
                                                                                   const FusedArgs& a)
{
#ifdef FLOW2D_FUSED_INSTANCE_LONE  // (ten kernels instead of twenty: a launch that continues an outer iteration's sweeps takes the pipeline's build)
    if (a.continue_sweeps) return 1;
    return launch_for_inner_cont<FLOW2D_FUSED_INSTANCE_GRAD, FLOW2D_FUSED_INSTANCE_POW2 != 0, false>(inner, grid, stream, a);
#else
    return launch_for_inner<FLOW2D_FUSED_INSTANCE_GRAD, FLOW2D_FUSED_INSTANCE_POW2 != 0>(inner, grid, stream, a);
#endif
}

}  // namespace flow2d
