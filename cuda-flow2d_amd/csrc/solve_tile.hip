// One outer iteration of the variational solver on small LDS tiles: the kernel shape for the MID-SIZE and small
// pyramid levels (64 x 33 ... 896 x 896 pixels).
//
// The fused strip kernel (solve_fused_kernel.hpp) gives a wave a 64-column strip and lets it walk down the image; a wave
// issues one instruction every ~4 cycles whatever it is, so a launch lasts (rows + halo) x ~1.1 us however few strips
// there are -- at 256 x 256 twelve microseconds for work that would occupy the chip's vector ALUs for half a
// microsecond.  Here the same outer iteration (compute_phi_ksi + `inner` Jacobi sweeps, solve_2d.cu:43-377) is cut into
// tiles of 8 x 8, 16 x 16 or 32 x 32 pixels, one workgroup each with one thread per pixel (32 x 32: per two pixels) of
// the tile and its (inner + 1)-pixel halo (20 x 20, 28 x 28 or 44 x 44 pixels), so that even a 64 x 64 level spreads over
// 64 compute units: the region lives in LDS, a pixel's coefficients in its thread's registers, and a sweep is one pass
// between two workgroup barriers.  The halo is recomputed by every tile (6x / 3x / 1.9x the pixels of the tile itself
// in the first stage, shrinking by one ring per stage), which is why the large levels stay with the strips (1.3x).
//
// Arithmetic: the solver_math.hpp expressions in the reference's order, no FMA contraction -- the same bits as the
// per-sweep, fused and single-workgroup kernels and as the oracle.
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "solver_math.hpp"

namespace {

using namespace flow2d_math;

constexpr int kMaxInner = 5;

struct TileArgs {
    const float* f0;
    const float* f1;
    const float* u;
    const float* v;
    const float* du;
    const float* dv;
    float* out_du;
    float* out_dv;
    int w, h, pitch;
    int inner;
    int zero_increment;  // first outer iteration of a level: du = dv = 0, the planes are not read
    float hx, hy, alpha, e_smooth, e_data;
    float sor_omega;  // SOR instantiations only: the `inner` stages are red-black half-sweeps (round 5)
    unsigned long long batch_stride;  // floats between the instances of a batched launch (blockIdx.z)
};

// GRAD: 0 brightness constancy (solve_2d), 1 gradient constancy with the reference's 16x8 block rule (solve_2d_grad),
// 2 gradient constancy over true neighbours (FLOW2D_CONSTANCY_GRADIENT_UNTILED)
// (two workgroups of up to 1024 threads per CU: at most 64 VGPRs, so that one tile's barriers and LDS round trips
// hide behind the other's arithmetic)
// POW2: 2h and 4h are powers of two (every level of a 0.5 pyramid), so the six divisions by them in compute_phi_ksi
// are exact multiplies by the reciprocal (solver_math.hpp, div_spacing).
// SOR (round 5): the stages are half-sweeps of the opt-in red-black successive over-relaxation -- stage k relaxes the pixels
// with (x + y) % 2 == (k - 1) % 2 (point_update_sor) and passes the other colour through; a half-sweep reaches one pixel like a
// Jacobi sweep, so rings, halo and barriers are those of `inner` Jacobi sweeps: two iterations per launch at most.
template <int TX, int TY, int GRAD, int kThreads, bool POW2, bool SOR>
__global__ __launch_bounds__(kThreads, 8) void tile_outer_kernel(TileArgs a)
{
    constexpr int kHalo = kMaxInner + 1;
    constexpr int RW = TX + 2 * kHalo, RH = TY + 2 * kHalo, RN = RW * RH;
    constexpr int PPT = (RN + kThreads - 1) / kThreads;  // pixels per thread
    constexpr int kPlanes = GRAD ? 10 : 7;
    __shared__ float lds[kPlanes * RN];
    {
        const size_t off = static_cast<size_t>(blockIdx.z) * static_cast<size_t>(a.batch_stride);
        a.f0 += off, a.f1 += off, a.u += off, a.v += off, a.du += off, a.dv += off, a.out_du += off, a.out_dv += off;
    }
    float* const P_u = lds + 0 * RN;   // later: ping (u + du^k)
    float* const P_v = lds + 1 * RN;   //        ping (v + dv^k)
    float* const P_du = lds + 2 * RN;  // later: pong
    float* const P_dv = lds + 3 * RN;
    float* const P_f0 = lds + 4 * RN;
    float* const P_f1 = lds + 5 * RN;
    float* const P_phi = lds + 6 * RN;
    float* const P_fx = lds + (GRAD ? 7 : 0) * RN;
    float* const P_fy = lds + (GRAD ? 8 : 0) * RN;
    float* const P_ft = lds + (GRAD ? 9 : 0) * RN;

    const int w = a.w, h = a.h;
    const int x_org = blockIdx.x * TX - kHalo, y_org = blockIdx.y * TY - kHalo;
    // the launch may hold fewer sweeps than the halo was sized for: the valid rings are counted from the tile outwards
    const int slack = kMaxInner - a.inner;

    // ---- this thread's pixels: pixel j is region slot p = threadIdx.x + j * 256 -------------------------------------
    // One word of facts per pixel instead of coordinates and neighbour slots in registers:
    //   bit 0..3  left / right / up / down neighbour lies on the other side (reflect rule of solve_2d.cu:75-134 at the
    //             image border; at the region's own edge too, which only keeps the access inside the region: such a
    //             pixel is in ring 0 and never valid)
    //   bit 4..7  the pixel is in image column 0 / w-1, row 0 / h-1      bit 8  inside the image
    //   bit 16..  its ring (distance from the region's edge), less the rings this launch does not need
    int facts[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = threadIdx.x + j * kThreads;
        const int rx = p % RW, ry = p / RW;
        const int gx = x_org + rx, gy = y_org + ry;
        const bool inside = p < RN && gx >= 0 && gx < w && gy >= 0 && gy < h;
        const int ring = max(min(min(rx, RW - 1 - rx), min(ry, RH - 1 - ry)) - slack, 0);
        facts[j] = (gx == 0 || rx == 0 ? 1 : 0) | (gx == w - 1 || rx == RW - 1 ? 2 : 0) | (gy == 0 || ry == 0 ? 4 : 0) |
                   (gy == h - 1 || ry == RH - 1 ? 8 : 0) | (gx == 0 ? 16 : 0) | (gx == w - 1 ? 32 : 0) | (gy == 0 ? 64 : 0) |
                   (gy == h - 1 ? 128 : 0) | (inside ? 256 : 0) | (ring << 16);
    }
    auto slot = [&](int j) { return static_cast<int>(threadIdx.x) + j * kThreads; };
    auto inside = [&](int j) { return (facts[j] & 256) != 0; };
    auto ring_of = [&](int j) { return facts[j] >> 16; };
    auto left_of = [&](int j) { return slot(j) + ((facts[j] & 1) ? 1 : -1); };
    auto right_of = [&](int j) { return slot(j) + ((facts[j] & 2) ? -1 : 1); };
    auto up_of = [&](int j) { return slot(j) + ((facts[j] & 4) ? RW : -RW); };
    auto down_of = [&](int j) { return slot(j) + ((facts[j] & 8) ? -RW : RW); };
    auto global_x = [&](int j) { return x_org + slot(j) % RW; };
    auto global_y = [&](int j) { return y_org + slot(j) / RW; };

    // ---- load ----------------------------------------------------------------------------------------------------
    float uc[PPT], vc[PPT], dv_cur[PPT], du0[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        uc[j] = vc[j] = dv_cur[j] = du0[j] = 0.f;
        if (inside(j)) {
            const size_t o = static_cast<size_t>(global_y(j)) * a.pitch + global_x(j);
            uc[j] = a.u[o];
            vc[j] = a.v[o];
            if (!a.zero_increment) {
                du0[j] = a.du[o];
                dv_cur[j] = a.dv[o];
            }
            P_u[slot(j)] = uc[j];
            P_v[slot(j)] = vc[j];
            P_du[slot(j)] = du0[j];
            P_dv[slot(j)] = dv_cur[j];
            P_f0[slot(j)] = a.f0[o];
            P_f1[slot(j)] = a.f1[o];
        }
    }
    __syncthreads();

    // ---- stage A (rings >= 1): phi, brightness derivatives, ksi -- compute_phi_ksi, solve_2d.cu:138-197 ----------
    float fx[PPT], fy[PPT], ft[PPT], ksi[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        fx[j] = fy[j] = ft[j] = ksi[j] = 0.f;
        if (inside(j) && ring_of(j) >= 1) {
            const int p = slot(j), nl = left_of(j), nr = right_of(j), nu = up_of(j), nd = down_of(j);
            const float dux = diff4s<POW2>(P_u[nr], P_u[nl], P_du[nr], P_du[nl], 2.f * a.hx, 1.f / (2.f * a.hx));
            const float duy = diff4s<POW2>(P_u[nd], P_u[nu], P_du[nd], P_du[nu], 2.f * a.hy, 1.f / (2.f * a.hy));
            const float dvx = diff4s<POW2>(P_v[nr], P_v[nl], P_dv[nr], P_dv[nl], 2.f * a.hx, 1.f / (2.f * a.hx));
            const float dvy = diff4s<POW2>(P_v[nd], P_v[nu], P_dv[nd], P_dv[nu], 2.f * a.hy, 1.f / (2.f * a.hy));
            P_phi[p] = phi_value(dux, duy, dvx, dvy, a.e_smooth);
            fx[j] = diff4s<POW2>(P_f0[nr], P_f0[nl], P_f1[nr], P_f1[nl], 4.f * a.hx, 1.f / (4.f * a.hx));
            fy[j] = diff4s<POW2>(P_f0[nd], P_f0[nu], P_f1[nd], P_f1[nu], 4.f * a.hy, 1.f / (4.f * a.hy));
            ft[j] = P_f1[p] - P_f0[p];
            ksi[j] = ksi_value(fx[j], fy[j], ft[j], du0[j], dv_cur[j], a.e_data);
            if (GRAD) {
                P_fx[p] = fx[j];
                P_fy[p] = fy[j];
                P_ft[p] = ft[j];
            }
        }
    }
    __syncthreads();

    // ---- stage B (rings >= 2): face weights, motion tensor, denominators; u + du, v + dv become visible -----------
    const float hx_2 = a.alpha / (a.hx * a.hx);
    const float hy_2 = a.alpha / (a.hy * a.hy);
    float wxp[PPT], wxm[PPT], wyp[PPT], wym[PPT], den_u[PPT], den_v[PPT], J12[PPT], J13[PPT], J23[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        wxp[j] = wxm[j] = wyp[j] = wym[j] = J12[j] = J13[j] = J23[j] = 0.f;
        den_u[j] = den_v[j] = 1.f;
        if (inside(j) && ring_of(j) >= 2) {
            const int p = slot(j), nl = left_of(j), nr = right_of(j), nu = up_of(j), nd = down_of(j);
            const float xp = static_cast<float>((facts[j] & 32) == 0) * hx_2;   // x < w - 1
            const float xm = static_cast<float>((facts[j] & 16) == 0) * hx_2;   // x > 0
            const float yp = static_cast<float>((facts[j] & 128) == 0) * hy_2;  // y < h - 1
            const float ym = static_cast<float>((facts[j] & 64) == 0) * hy_2;   // y > 0
            const float pc = P_phi[p];
            wxp[j] = face_phi(P_phi[nr], pc) * xp;
            wxm[j] = face_phi(P_phi[nl], pc) * xm;
            wyp[j] = face_phi(P_phi[nd], pc) * yp;
            wym[j] = face_phi(P_phi[nu], pc) * ym;
            const float sumH = sum_weights(wxp[j], wxm[j], wyp[j], wym[j]);
            float J11, J22;
            if (!GRAD) {
                J11 = fx[j] * fx[j];
                J22 = fy[j] * fy[j];
                J12[j] = fx[j] * fy[j];
                J13[j] = fx[j] * ft[j];
                J23[j] = fy[j] * ft[j];
            } else {
                int xa, xb, ya, yb;  // slots of the second derivatives' neighbours
                if (GRAD == 1) {     // own value at the reference's 16x8 block edge and at the image edge (:816-841,872-876)
                    const int gx = global_x(j), gy = global_y(j);
                    xa = ((gx & 15) == 0) ? p : p - 1;
                    xb = ((gx & 15) == 15 || (facts[j] & 32)) ? p : p + 1;
                    ya = ((gy & 7) == 0) ? p : p - RW;
                    yb = ((gy & 7) == 7 || (facts[j] & 128)) ? p : p + RW;
                } else {             // true neighbours, reflected at the image border
                    xa = nl, xb = nr, ya = nu, yb = nd;
                }
                const float hx_1 = 1.0 / (2.0 * a.hx);  // double, rounded to float (solve_2d.cu:868-869)
                const float hy_1 = 1.0 / (2.0 * a.hy);
                const float fxx = (P_fx[xb] - P_fx[xa]) * hx_1;
                const float fxy = (P_fx[yb] - P_fx[ya]) * hy_1;
                const float fyy = (P_fy[yb] - P_fy[ya]) * hy_1;
                const float fxt = (P_ft[xb] - P_ft[xa]) * hx_1;
                const float fyt = (P_ft[yb] - P_ft[ya]) * hy_1;
                gradient_tensor(fxx, fxy, fyy, fxt, fyt, J11, J22, J12[j], J13[j], J23[j]);
            }
            den_u[j] = update_denominator(ksi[j], J11, sumH);
            den_v[j] = update_denominator(ksi[j], J22, sumH);
        }
        if (inside(j)) {  // every ring: the sweeps read these as neighbours (the planes of u, v are no longer needed as such)
            P_u[slot(j)] = uc[j] + du0[j];
            P_v[slot(j)] = vc[j] + dv_cur[j];
        }
    }
    __syncthreads();

    // ---- sweeps k = 1 .. inner (rings >= k + 1): solve_2d.cu:349-367; ping-pong between the two plane pairs --------
    float* ping_u = P_u;
    float* ping_v = P_v;
    float* pong_u = P_du;
    float* pong_v = P_dv;
    float du_new[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) du_new[j] = SOR ? du0[j] : 0.f;  // (SOR: the increment a half-sweep blends with)
    for (int k = 1; k <= a.inner; ++k) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            if (inside(j) && ring_of(j) >= k + 1) {
                const int nl = left_of(j), nr = right_of(j), nu = up_of(j), nd = down_of(j);
                const float sumU = sum_flux(wxp[j], wxm[j], wyp[j], wym[j], ping_u[nr], ping_u[nl], ping_u[nd],
                                            ping_u[nu], uc[j]);
                const float sumV = sum_flux(wxp[j], wxm[j], wyp[j], wym[j], ping_v[nr], ping_v[nl], ping_v[nd],
                                            ping_v[nu], vc[j]);
                float ndv;
                if (SOR) {
                    float ndu;
                    point_update_sor(ksi[j], den_u[j], den_v[j], J12[j], J13[j], J23[j], sumU, sumV, du_new[j], dv_cur[j],
                                     a.sor_omega, ndu, ndv);
                    const bool mine = ((global_x(j) + global_y(j) + k - 1) & 1) == 0;  // half-sweep k relaxes colour (k - 1) % 2
                    du_new[j] = mine ? ndu : du_new[j];
                    ndv = mine ? ndv : dv_cur[j];
                } else {
                    point_update(ksi[j], den_u[j], den_v[j], J12[j], J13[j], J23[j], sumU, sumV, dv_cur[j], du_new[j], ndv);
                }
                dv_cur[j] = ndv;
                pong_u[slot(j)] = uc[j] + du_new[j];
                pong_v[slot(j)] = vc[j] + ndv;
            }
        }
        __syncthreads();
        float* t = ping_u;
        ping_u = pong_u;
        pong_u = t;
        t = ping_v;
        ping_v = pong_v;
        pong_v = t;
    }

#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        if (inside(j) && ring_of(j) >= a.inner + 1) {  // the tile itself
            const size_t o = static_cast<size_t>(global_y(j)) * a.pitch + global_x(j);
            a.out_du[o] = du_new[j];
            a.out_dv[o] = dv_cur[j];
        }
    }
}

template <int TX, int TY, int kThreads, bool POW2, bool SOR>
void launch_tiles_sor(int grad, dim3 grid, hipStream_t stream, const TileArgs& a)
{
    if (grad == 1)
        tile_outer_kernel<TX, TY, 1, kThreads, POW2, SOR><<<grid, kThreads, 0, stream>>>(a);
    else if (grad == 2)
        tile_outer_kernel<TX, TY, 2, kThreads, POW2, SOR><<<grid, kThreads, 0, stream>>>(a);
    else
        tile_outer_kernel<TX, TY, 0, kThreads, POW2, SOR><<<grid, kThreads, 0, stream>>>(a);
}
template <int TX, int TY, int kThreads, bool POW2>
void launch_tiles_pow2(int grad, dim3 grid, hipStream_t stream, const TileArgs& a)
{
    if (a.sor_omega != 0.f)
        launch_tiles_sor<TX, TY, kThreads, POW2, true>(grad, grid, stream, a);
    else
        launch_tiles_sor<TX, TY, kThreads, POW2, false>(grad, grid, stream, a);
}

// true when x is a normal power of two whose reciprocal (and 1/(2x), 1/(4x)) is exactly representable
bool is_power_of_two(float x)
{
    int e = 0;
    return x > 0.f && std::frexp(x, &e) == 0.5f && e > -100 && e < 100;
}

template <int TX, int TY, int kThreads>
void launch_tiles(int grad, dim3 grid, hipStream_t stream, const TileArgs& a)
{
    if (is_power_of_two(a.hx) && is_power_of_two(a.hy))
        launch_tiles_pow2<TX, TY, kThreads, true>(grad, grid, stream, a);
    else
        launch_tiles_pow2<TX, TY, kThreads, false>(grad, grid, stream, a);
}

}  // namespace

namespace flow2d {

bool tiled_supports(int constancy, size_t inner)
{
    return inner >= 1 && inner <= kMaxInner &&
           (constancy == FLOW2D_CONSTANCY_GREY || constancy == FLOW2D_CONSTANCY_GRADIENT ||
            constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED);
}

// One outer iteration: reads du/dv (previous outer iteration, unless zero_increment), writes out_du/out_dv.
// sor_omega != 0: the `inner` stages are red-black half-sweeps (an even number: inner / 2 iterations).
int launch_tiled_outer(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                       const float* v, const float* du, const float* dv, size_t w, size_t h, size_t pitch_bytes, float hx,
                       float hy, float alpha, float e_smooth, float e_data, size_t inner, float* out_du, float* out_dv,
                       bool zero_increment, float sor_omega)
{
    if (!tiled_supports(constancy, inner) || (sor_omega != 0.f && (inner & 1))) return FLOW2D_ERR_UNSUPPORTED;
    TileArgs a{f0, f1, u, v, du, dv, out_du, out_dv, (int)w, (int)h, (int)(pitch_bytes / 4), (int)inner,
               zero_increment ? 1 : 0, hx, hy, alpha, e_smooth, e_data, sor_omega,
               static_cast<unsigned long long>(ctx->batch_stride_floats)};
    const int grad = constancy == FLOW2D_CONSTANCY_GRADIENT ? 1 : (constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED ? 2 : 0);
    // Tile size by level size (measured on MI355X, level solve of 10 x 5, Grey / Gradient, ms; fused strips for comparison):
    //            8x8            16x16          32x32          strips
    //   64^2     0.052 / 0.055  0.061 / 0.064                 0.110 / 0.120
    //   256^2    0.096 / 0.104  0.068 / 0.071  0.110 / 0.118  0.128 / 0.149
    //   384^2    0.176 / 0.193  0.128 / 0.137  0.119 / 0.128  0.169 / 0.162
    //   512^2    0.268 / 0.297  0.167 / 0.180  0.120 / 0.132  0.199 / 0.217
    //   640^2    0.392 / 0.451  0.247 / 0.273  0.198 / 0.223  0.246 / 0.270
    //   800^2                   0.350 / 0.387  0.272 / 0.298  0.287 / 0.311
    //   1024^2                  0.531 / 0.594  0.356 / 0.422  0.375 / 0.407   (-> AUTO keeps the strips from 896^2 on)
    // (h = 1, i.e. the power-of-two spacing of a 0.5 pyramid; an interior-tile variant without the border facts was
    // measured too: 32 x 32 +3 %, not kept)
    // 8 x 8 tiles (one pixel per thread over the 20 x 20 region, 6x the tile's pixels) while that still spreads the
    // level thinly; 16 x 16 (28 x 28 region, 3x) up to 352^2; 32 x 32 (44 x 44 region, two pixels per thread, 1.9x)
    // above: fewer, larger workgroups, two of them per CU (64 VGPRs), so one tile's barriers hide behind the other's
    // arithmetic.
#ifdef FLOW2D_DEV_BUILD  // force a tile size: 1 = 8 x 8, 2 = 16 x 16, 3 = 32 x 32
    static const int variant = std::getenv("FLOW2D_TILE_VARIANT") ? std::atoi(std::getenv("FLOW2D_TILE_VARIANT")) : 0;
#else
    const int variant = 0;
#endif
    const bool tiny = w * h <= 160 * 160, mid = w * h <= 352 * 352;
    if (variant == 1 || (variant == 0 && tiny))
        launch_tiles<8, 8, 512>(grad, dim3(div_up(w, 8), div_up(h, 8), ctx->batch_count), ctx->stream, a);
    else if (variant == 2 || (variant == 0 && mid))
        launch_tiles<16, 16, 1024>(grad, dim3(div_up(w, 16), div_up(h, 16), ctx->batch_count), ctx->stream, a);
    else
        launch_tiles<32, 32, 1024>(grad, dim3(div_up(w, 32), div_up(h, 32), ctx->batch_count), ctx->stream, a);
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

}  // namespace flow2d
