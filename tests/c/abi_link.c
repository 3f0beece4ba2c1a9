/* Plain-C client of the C-ABI (include/flow2d_c_abi.h): what a reference-side binding in any language relies on --
 * C linkage, plain pointers and sizes, int status codes.  Compiled and run by tests/test_host_cpu.py without a GPU:
 * everything called here must work (or fail with a status code, never crash) when no device is present. */
#include <stdio.h>
#include <string.h>

#include "flow2d_c_abi.h"

int main(void)
{
    float taps[51];
    int radius = -1, count = -1, st;
    if (flow2d_abi_version() != 1) return 10;
    if (flow2d_plane_pitch_bytes(584) != 2560) return 11;
    if (flow2d_gaussian_kernel(1.5f, taps, &radius) != FLOW2D_OK || radius != 4) return 12;
    if (!(taps[4] > taps[3] && taps[3] > taps[0])) return 13;
    if (strcmp(flow2d_status_string(FLOW2D_OK), flow2d_status_string(FLOW2D_ERR_UNSUPPORTED)) == 0) return 14;
    st = flow2d_device_count(&count);
    if (st != FLOW2D_OK && st != FLOW2D_ERR_NO_DEVICE) return 15;
    if (flow2d_solver_algorithm_for(FLOW2D_SOLVER_AUTO, 4096, 4096, flow2d_plane_pitch_bytes(4096), 10, 5,
                                    FLOW2D_CONSTANCY_GRADIENT) != FLOW2D_SOLVER_FUSED)
        return 16;
    if (flow2d_context_set_batch(NULL, 2, 4096) != FLOW2D_ERR_INVALID_ARGUMENT) return 17;
    printf("flow2d C-ABI v%d, %d device(s), sigma 1.5 -> radius %d\n", flow2d_abi_version(), count < 0 ? 0 : count, radius);
    return 0;
}
