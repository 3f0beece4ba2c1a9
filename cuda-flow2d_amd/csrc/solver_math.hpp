// Pointwise arithmetic of the variational solver, shared by the per-sweep kernels (solve.hip) and the
// fused outer-iteration kernel (solve_fused_kernel.hpp) so both evaluate literally the same expressions.
// Operation order and float/double promotions are those of the reference's src/kernels/solve_2d.cu
// (lines cited per function); built with -ffp-contract=off, so nothing is fused.
#pragma once

#include <hip/hip_runtime.h>

namespace flow2d_math {

// central difference of (a + da) as the reference sums it: (aP - aM + daP - daM) / den
// (solve_2d.cu:141-157 with den = 2*h; :164-171 with den = 4*h for the two frames)
__device__ __forceinline__ float diff4(float aP, float aM, float bP, float bM, float den)
{
    return (aP - aM + bP - bM) / den;
}

// smoothness diffusivity, solve_2d.cu:161-162: phi = 1 / (2 sqrt(argument))
__device__ __forceinline__ float phi_argument(float dux, float duy, float dvx, float dvy, float e_smooth)
{
    return dux * dux + duy * duy + dvx * dvx + dvy * dvy + e_smooth * e_smooth;
}
__device__ __forceinline__ float phi_value(float dux, float duy, float dvx, float dvy, float e_smooth)
{
    return 1.f / (2.f * sqrtf(phi_argument(dux, duy, dvx, dvy, e_smooth)));
}

// data-term robustifier from the brightness tensor, solve_2d.cu:176-196: ksi = 1 / (2 sqrt(argument))
__device__ __forceinline__ float ksi_argument(float fx, float fy, float ft, float du, float dv, float e_data)
{
    const float J11 = fx * fx, J22 = fy * fy, J33 = ft * ft, J12 = fx * fy, J13 = fx * ft, J23 = fy * ft;
    float s = (J11 * du + J12 * dv + J13) * du + (J12 * du + J22 * dv + J23) * dv + (J13 * du + J23 * dv + J33);
    s = static_cast<float>(s > 0) * s;
    return s + e_data * e_data;
}
__device__ __forceinline__ float ksi_value(float fx, float fy, float ft, float du, float dv, float e_data)
{
    return 1.f / (2.f * sqrtf(ksi_argument(fx, fy, ft, du, dv, e_data)));
}

// face diffusivity, solve_2d.cu:343-346
__device__ __forceinline__ float face_phi(float neighbour, float centre) { return (neighbour + centre) / 2.f; }

// sumH of solve_2d.cu:349 from the four face weights w = face_phi * {xp,xm,yp,ym}
__device__ __forceinline__ float sum_weights(float wxp, float wxm, float wyp, float wym) { return (wxp + wxm + wyp + wym); }

// sumU / sumV of solve_2d.cu:350-359: neighbours are the full flow (u + du) of the previous sweep
__device__ __forceinline__ float sum_flux(float wxp, float wxm, float wyp, float wym, float nR, float nL, float nD,
                                          float nU, float centre)
{
    return wxp * (nR - centre) + wxm * (nL - centre) + wyp * (nD - centre) + wym * (nU - centre);
}

// denominators of the point update, solve_2d.cu:363,367: constant over the sweeps of an outer iteration
__device__ __forceinline__ float update_denominator(float ksi, float Jnn, float sumH) { return ksi * Jnn + sumH; }

// the coupled 2x2 update of solve_2d.cu:361-367 (dv' uses the fresh du')
__device__ __forceinline__ void point_update(float ksi, float den_u, float den_v, float J12, float J13, float J23,
                                             float sumU, float sumV, float dv_old, float& du_new, float& dv_new)
{
    du_new = (ksi * (-J13 - J12 * dv_old) + sumU) / den_u;
    dv_new = (ksi * (-J23 - J12 * du_new) + sumV) / den_v;
}

// Opt-in successive over-relaxation of the same 2x2 point system (NOT in the reference, which is Jacobi with
// omega = 1): the Gauss-Seidel value is blended with the old one, and dv sees the relaxed du.
__device__ __forceinline__ void point_update_sor(float ksi, float den_u, float den_v, float J12, float J13, float J23,
                                                 float sumU, float sumV, float du_old, float dv_old, float omega,
                                                 float& du_new, float& dv_new)
{
    const float gs_du = (ksi * (-J13 - J12 * dv_old) + sumU) / den_u;
    du_new = (1.f - omega) * du_old + omega * gs_du;
    const float gs_dv = (ksi * (-J23 - J12 * du_new) + sumV) / den_v;
    dv_new = (1.f - omega) * dv_old + omega * gs_dv;
}

// x / d for the grid-spacing divisors (2h, 4h).  When d is a power of two, x * (1/d) is the same
// correctly rounded value as x / d (1/d is exact and both round the same real number), so the caller
// may pass inv_d and pow2 = true to replace the ~10-instruction division by one multiply.
template <bool POW2>
__device__ __forceinline__ float div_spacing(float x, float d, float inv_d)
{
    return POW2 ? x * inv_d : x / d;
}

template <bool POW2>
__device__ __forceinline__ float diff4s(float aP, float aM, float bP, float bM, float den, float inv_den)
{
    return div_spacing<POW2>(aP - aM + bP - bM, den, inv_den);
}

// gradient-constancy tensor from the second derivatives, solve_2d.cu:879-884
__device__ __forceinline__ void gradient_tensor(float fxx, float fxy, float fyy, float fxt, float fyt, float& J11,
                                                float& J22, float& J12, float& J13, float& J23)
{
    J11 = fxx * fxx + fxy * fxy;
    J22 = fxy * fxy + fyy * fyy;
    J12 = fxx * fxy + fxy * fyy;
    J13 = fxx * fxt + fxy * fyt;
    J23 = fxy * fxt + fyy * fyt;
}

// log(I + 1.0f) of a frame value, solve_2d.cu:519-535 (solve_2d_log).  logf is the device library's; the
// reference's kernel compiled for this GPU calls the same function, so the bits agree with it (a CPU libm's logf
// may differ in the last place).
__device__ __forceinline__ float log1p_frame(float value) { return logf(value + 1.0f); }

// ---- two-wide forms: the u and v equations share their weights, so the pair (u, v) goes through
// v_pk_add_f32 / v_pk_mul_f32 (one VALU issue for both fields).  Component by component these are the
// scalar expressions above, in the same order; nothing is contracted into an FMA.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f dup_x(v2f a) { return a.xx; }
__device__ __forceinline__ v2f dup_y(v2f a) { return a.yy; }

// (aP - aM + bP - bM) per component, solve_2d.cu:141-157
__device__ __forceinline__ v2f diff4_num2(v2f aP, v2f aM, v2f bP, v2f bM) { return aP - aM + bP - bM; }

// sumU and sumV of solve_2d.cu:350-359 at once: wx = (w_x+, w_x-), wy = (w_y+, w_y-)
__device__ __forceinline__ v2f sum_flux2(v2f wx, v2f wy, v2f nR, v2f nL, v2f nD, v2f nU, v2f centre)
{
    return dup_x(wx) * (nR - centre) + dup_y(wx) * (nL - centre) + dup_x(wy) * (nD - centre) + dup_y(wy) * (nU - centre);
}

}  // namespace flow2d_math
