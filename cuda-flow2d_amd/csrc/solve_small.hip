// Whole-level solver for the coarse pyramid levels (up to 64 x 64 pixels) in ONE launch.
//
// At these sizes the reference's loop -- outer x (1 + inner) dependent launches, each followed by a host
// synchronisation (cuda_operation_solve_2d.cpp:238-299) -- is pure launch latency: about 3.4 us per
// launch even when queued back to back, 0.2 ms per level for 10 x 5 sweeps.  Here a single 1024-thread
// workgroup keeps the whole level on one CU: per-pixel state (u, v, du, dv, image derivatives, motion
// tensor, face weights, denominators) lives in registers, four pixels per thread; only what neighbours
// must see goes through LDS planes with a one-pixel reflected halo (u, v, du, dv for phi; phi for the face
// weights; u+du, v+dv for the Jacobi sweeps).  A sweep costs two workgroup barriers instead of a kernel
// boundary.  All outer and inner iterations run inside the launch.
//
// Arithmetic: the same solver_math.hpp expressions as the per-sweep and fused kernels, hence the same bits.
#include "common.hpp"
#include "solver_math.hpp"

namespace {

using namespace flow2d_math;

constexpr int kMaxSide = 64;               // level width and height limit
constexpr int kStride = kMaxSide + 2;      // LDS row length incl. halo
constexpr int kPlane = kStride * kStride;  // floats per LDS plane

enum Plane { kU = 0, kV, kDU, kDV, kUU, kVV, kPhi, kPlaneCount };

struct SmallArgs {
    const float* f0;
    const float* f1;
    const float* u;
    const float* v;
    float* out_du;
    float* out_dv;
    int w, h, pitch;
    int outer, inner;
    float hx, hy, alpha, e_smooth, e_data;
    int untiled;  // gradient constancy only: true neighbours instead of the reference's 16x8 tile rule
    int log;      // solve_2d_log (solve_2d.cu:391-669): tensor from log(I + 1); x/y neighbours of the frames, of phi in
                  // the face weights and of the flow in the sweeps are the pixel's own value at a 16x8 block edge
};

__device__ __forceinline__ int at(int x, int y) { return (y + 1) * kStride + (x + 1); }

// Stores a pixel and the halo slots that mirror it (reflect without repeat: -1 <- 1, n <- n-2).
__device__ __forceinline__ void put(float* plane, int x, int y, int w, int h, float value)
{
    plane[at(x, y)] = value;
    if (x == 1) plane[at(-1, y)] = value;
    if (x == w - 2) plane[at(w, y)] = value;
    if (y == 1) plane[at(x, -1)] = value;
    if (y == h - 2) plane[at(x, h)] = value;
}

// kPx = pixels per thread (rows ty*kPx .. ty*kPx+kPx-1 of column tx): 1, 2 or 4 for levels up to 16, 32, 64 rows
template <bool GRAD, int kPx>
__global__ __launch_bounds__(1024) void small_level_kernel(SmallArgs a)
{
    __shared__ float lds[kPlaneCount * kPlane];
    float* const P_u = lds + kU * kPlane;
    float* const P_v = lds + kV * kPlane;
    float* const P_du = lds + kDU * kPlane;
    float* const P_dv = lds + kDV * kPlane;
    float* const P_uu = lds + kUU * kPlane;
    float* const P_vv = lds + kVV * kPlane;
    float* const P_phi = lds + kPhi * kPlane;

    const int w = a.w, h = a.h;
    const int x = threadIdx.x;       // 0..63
    const int y_base = threadIdx.y * kPx;
    const bool col_ok = x < w;

    // neighbour columns / rows of the face weights and of the sweeps: x-1, x+1, y-1, y+1 (the LDS halo holds the
    // reflected pixel), except in solve_2d_log mode, where a 16x8 block edge sees the pixel itself
    const bool logm = GRAD && a.log;
    const int xl_n = (logm && (x & 15) == 0) ? x : x - 1;
    const int xr_n = (logm && (x & 15) == 15) ? x : x + 1;
    auto up_of = [&](int y) { return (logm && (y & 7) == 0) ? y : y - 1; };
    auto down_of = [&](int y) { return (logm && (y & 7) == 7) ? y : y + 1; };

    float uc[kPx], vc[kPx], du[kPx], dv[kPx];
    float fx[kPx], fy[kPx], ft[kPx];          // brightness derivatives (ksi; Grey tensor)
    // Gradient constancy keeps its tensor in registers; for Grey the entries are products of fx, fy, ft and
    // are re-formed where needed (fewer live registers than holding five more values per pixel)
    constexpr int kJ = GRAD ? kPx : 1;
    float gJ11[kJ], gJ22[kJ], gJ12[kJ], gJ13[kJ], gJ23[kJ];
    bool ok[kPx];
    auto tensor = [&](int j, float& J11, float& J22, float& J12, float& J13, float& J23) {
        if (GRAD) {
            J11 = gJ11[j % kJ]; J22 = gJ22[j % kJ]; J12 = gJ12[j % kJ]; J13 = gJ13[j % kJ]; J23 = gJ23[j % kJ];
        } else {
            J11 = fx[j] * fx[j]; J22 = fy[j] * fy[j]; J12 = fx[j] * fy[j]; J13 = fx[j] * ft[j]; J23 = fy[j] * ft[j];
        }
    };

    // ---- prologue: frames into LDS (planes kUU/kVV are free until the first sweep), derivatives ----------
#pragma unroll
    for (int j = 0; j < kPx; ++j) {
        const int y = y_base + j;
        ok[j] = col_ok && y < h;
        uc[j] = vc[j] = du[j] = dv[j] = 0.f;
        if (ok[j]) {
            const size_t o = static_cast<size_t>(y) * a.pitch + x;
            put(P_uu, x, y, w, h, a.f0[o]);
            put(P_vv, x, y, w, h, a.f1[o]);
            uc[j] = a.u[o];
            vc[j] = a.v[o];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPx; ++j) {
        const int y = y_base + j;
        fx[j] = fy[j] = ft[j] = 0.f;
        if (ok[j]) {
            fx[j] = diff4(P_uu[at(x + 1, y)], P_uu[at(x - 1, y)], P_vv[at(x + 1, y)], P_vv[at(x - 1, y)], 4.f * a.hx);
            fy[j] = diff4(P_uu[at(x, y + 1)], P_uu[at(x, y - 1)], P_vv[at(x, y + 1)], P_vv[at(x, y - 1)], 4.f * a.hy);
            ft[j] = P_vv[at(x, y)] - P_uu[at(x, y)];
        }
    }
    if (GRAD) {
        // what the tensor differentiates: fx, fy, ft of the frames, or (log mode) of log(frame + 1) with the block rule
        float gx[kPx], gy[kPx], gt[kPx];
#pragma unroll
        for (int j = 0; j < kPx; ++j) {
            gx[j] = fx[j];
            gy[j] = fy[j];
            gt[j] = ft[j];
        }
        if (logm) {
            float r0[kPx], r1[kPx];
#pragma unroll
            for (int j = 0; j < kPx; ++j) {
                r0[j] = ok[j] ? P_uu[at(x, y_base + j)] : 0.f;
                r1[j] = ok[j] ? P_vv[at(x, y_base + j)] : 0.f;
            }
            __syncthreads();  // every brightness read of the raw frames is done
#pragma unroll
            for (int j = 0; j < kPx; ++j)
                if (ok[j]) {
                    put(P_uu, x, y_base + j, w, h, log1p_frame(r0[j]));
                    put(P_vv, x, y_base + j, w, h, log1p_frame(r1[j]));
                }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kPx; ++j) {
                const int y = y_base + j;
                if (ok[j]) {  // solve_2d.cu:519-535
                    gx[j] = diff4(P_uu[at(xr_n, y)], P_uu[at(xl_n, y)], P_vv[at(xr_n, y)], P_vv[at(xl_n, y)], 4.f * a.hx);
                    gy[j] = diff4(P_uu[at(x, down_of(y))], P_uu[at(x, up_of(y))], P_vv[at(x, down_of(y))],
                                  P_vv[at(x, up_of(y))], 4.f * a.hy);
                    gt[j] = P_vv[at(x, y)] - P_uu[at(x, y)];
                }
            }
        }
        // second derivatives inside the reference's 16x8 blocks with the block's edge value replicated
        // (solve_2d.cu:816-841,872-884); planes kDU, kDV, kPhi hold fx, fy, ft for this step only
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPx; ++j)
            if (ok[j]) {
                P_du[at(x, y_base + j)] = gx[j];
                P_dv[at(x, y_base + j)] = gy[j];
                P_phi[at(x, y_base + j)] = gt[j];
            }
        __syncthreads();
        const float hx_1 = 1.0 / (2.0 * a.hx);  // double, rounded to float (solve_2d.cu:868-869)
        const float hy_1 = 1.0 / (2.0 * a.hy);
#pragma unroll
        for (int j = 0; j < kPx; ++j) {
            const int y = y_base + j;
            gJ11[j % kJ] = gJ22[j % kJ] = gJ12[j % kJ] = gJ13[j % kJ] = gJ23[j % kJ] = 0.f;
            if (ok[j]) {
                // tile rule of the reference, or (untiled mode) the true neighbours reflected at the border
                const int xa = a.untiled ? mirror_index(x - 1, w) : (((x & 15) == 0) ? x : x - 1);
                const int xb = a.untiled ? mirror_index(x + 1, w) : (((x & 15) == 15 || x == w - 1) ? x : x + 1);
                const int ya = a.untiled ? mirror_index(y - 1, h) : (((y & 7) == 0) ? y : y - 1);
                const int yb = a.untiled ? mirror_index(y + 1, h) : (((y & 7) == 7 || y == h - 1) ? y : y + 1);
                const float fxx = (P_du[at(xb, y)] - P_du[at(xa, y)]) * hx_1;
                const float fxy = (P_du[at(x, yb)] - P_du[at(x, ya)]) * hy_1;
                const float fyy = (P_dv[at(x, yb)] - P_dv[at(x, ya)]) * hy_1;
                const float fxt = (P_phi[at(xb, y)] - P_phi[at(xa, y)]) * hx_1;
                const float fyt = (P_phi[at(x, yb)] - P_phi[at(x, ya)]) * hy_1;
                gradient_tensor(fxx, fxy, fyy, fxt, fyt, gJ11[j % kJ], gJ22[j % kJ], gJ12[j % kJ], gJ13[j % kJ], gJ23[j % kJ]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPx; ++j)
        if (ok[j]) {
            put(P_u, x, y_base + j, w, h, uc[j]);
            put(P_v, x, y_base + j, w, h, vc[j]);
        }

    const float hx_2 = a.alpha / (a.hx * a.hx);
    const float hy_2 = a.alpha / (a.hy * a.hy);
    const float xp = static_cast<float>(x < w - 1) * hx_2;
    const float xm = static_cast<float>(x > 0) * hx_2;

    for (int outer = 0; outer < a.outer; ++outer) {
        // ---- du, dv of the previous outer iteration become visible to the neighbours ---------------------
#pragma unroll
        for (int j = 0; j < kPx; ++j)
            if (ok[j]) {
                put(P_du, x, y_base + j, w, h, du[j]);
                put(P_dv, x, y_base + j, w, h, dv[j]);
            }
        __syncthreads();

        // ---- phi (to LDS) and ksi (registers): compute_phi_ksi, solve_2d.cu:138-197 ----------------------
        float ksi[kPx];
#pragma unroll
        for (int j = 0; j < kPx; ++j) {
            __builtin_amdgcn_sched_barrier(0);  // one pixel at a time: keeps the LDS reads of four pixels from being hoisted together
            const int y = y_base + j;
            ksi[j] = 0.f;
            if (ok[j]) {
                const float dux = diff4(P_u[at(x + 1, y)], P_u[at(x - 1, y)], P_du[at(x + 1, y)], P_du[at(x - 1, y)], 2.f * a.hx);
                const float duy = diff4(P_u[at(x, y + 1)], P_u[at(x, y - 1)], P_du[at(x, y + 1)], P_du[at(x, y - 1)], 2.f * a.hy);
                const float dvx = diff4(P_v[at(x + 1, y)], P_v[at(x - 1, y)], P_dv[at(x + 1, y)], P_dv[at(x - 1, y)], 2.f * a.hx);
                const float dvy = diff4(P_v[at(x, y + 1)], P_v[at(x, y - 1)], P_dv[at(x, y + 1)], P_dv[at(x, y - 1)], 2.f * a.hy);
                put(P_phi, x, y, w, h, phi_value(dux, duy, dvx, dvy, a.e_smooth));
                ksi[j] = ksi_value(fx[j], fy[j], ft[j], du[j], dv[j], a.e_data);
            }
        }
        __syncthreads();

        // ---- face weights, denominators; u+du, v+dv of this outer iteration's start ------------------------
        float wxp[kPx], wxm[kPx], wyp[kPx], wym[kPx], den_u[kPx], den_v[kPx];
#pragma unroll
        for (int j = 0; j < kPx; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            const int y = y_base + j;
            wxp[j] = wxm[j] = wyp[j] = wym[j] = 0.f;
            den_u[j] = den_v[j] = 1.f;
            if (ok[j]) {
                const float yp = static_cast<float>(y < h - 1) * hy_2;
                const float ym = static_cast<float>(y > 0) * hy_2;
                const float pc = P_phi[at(x, y)];
                wxp[j] = face_phi(P_phi[at(xr_n, y)], pc) * xp;
                wxm[j] = face_phi(P_phi[at(xl_n, y)], pc) * xm;
                wyp[j] = face_phi(P_phi[at(x, down_of(y))], pc) * yp;
                wym[j] = face_phi(P_phi[at(x, up_of(y))], pc) * ym;
                const float sumH = sum_weights(wxp[j], wxm[j], wyp[j], wym[j]);
                float J11, J22, J12, J13, J23;
                tensor(j, J11, J22, J12, J13, J23);
                den_u[j] = update_denominator(ksi[j], J11, sumH);
                den_v[j] = update_denominator(ksi[j], J22, sumH);
                put(P_uu, x, y, w, h, uc[j] + du[j]);
                put(P_vv, x, y, w, h, vc[j] + dv[j]);
            }
        }
        __syncthreads();

        // ---- inner Jacobi sweeps: solve_2d.cu:349-367 --------------------------------------------------------
        for (int inner = 0; inner < a.inner; ++inner) {
            float ndu[kPx], ndv[kPx];
#pragma unroll
            for (int j = 0; j < kPx; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                const int y = y_base + j;
                ndu[j] = ndv[j] = 0.f;
                if (ok[j]) {
                    const float sumU = sum_flux(wxp[j], wxm[j], wyp[j], wym[j], P_uu[at(xr_n, y)], P_uu[at(xl_n, y)],
                                                P_uu[at(x, down_of(y))], P_uu[at(x, up_of(y))], uc[j]);
                    const float sumV = sum_flux(wxp[j], wxm[j], wyp[j], wym[j], P_vv[at(xr_n, y)], P_vv[at(xl_n, y)],
                                                P_vv[at(x, down_of(y))], P_vv[at(x, up_of(y))], vc[j]);
                    float J11, J22, J12, J13, J23;
                    tensor(j, J11, J22, J12, J13, J23);
                    point_update(ksi[j], den_u[j], den_v[j], J12, J13, J23, sumU, sumV, dv[j], ndu[j], ndv[j]);
                }
            }
            __syncthreads();  // every neighbour read of this sweep is done: Jacobi
#pragma unroll
            for (int j = 0; j < kPx; ++j)
                if (ok[j]) {
                    du[j] = ndu[j];
                    dv[j] = ndv[j];
                    put(P_uu, x, y_base + j, w, h, uc[j] + du[j]);
                    put(P_vv, x, y_base + j, w, h, vc[j] + dv[j]);
                }
            __syncthreads();
        }
    }

#pragma unroll
    for (int j = 0; j < kPx; ++j)
        if (ok[j]) {
            const size_t o = static_cast<size_t>(y_base + j) * a.pitch + x;
            a.out_du[o] = du[j];
            a.out_dv[o] = dv[j];
        }
}

}  // namespace

namespace flow2d {

bool small_level_supports(size_t w, size_t h) { return w >= 2 && h >= 2 && w <= kMaxSide && h <= kMaxSide; }

// All outer x inner iterations of one level in a single launch; the result is written to out_du / out_dv.
int launch_small_level(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                       const float* v, size_t w, size_t h, size_t pitch_bytes, float hx, float hy, float alpha,
                       float e_smooth, float e_data, size_t outer, size_t inner, float* out_du, float* out_dv)
{
    if (!small_level_supports(w, h)) return FLOW2D_ERR_UNSUPPORTED;
    if (ctx->batch_count > 1) {  // one workgroup per instance, one launch each
        const unsigned n = ctx->batch_count;
        const size_t s = ctx->batch_stride_floats;
        ctx->batch_count = 1;
        int st = FLOW2D_OK;
        for (unsigned b = 0; b < n && st == FLOW2D_OK; ++b)
            st = launch_small_level(ctx, constancy, f0 + b * s, f1 + b * s, u + b * s, v + b * s, w, h, pitch_bytes, hx, hy,
                                    alpha, e_smooth, e_data, outer, inner, out_du + b * s, out_dv + b * s);
        ctx->batch_count = n;
        return st;
    }
    SmallArgs a{f0, f1, u, v, out_du, out_dv, (int)w, (int)h, (int)(pitch_bytes / 4), (int)outer, (int)inner,
                hx, hy, alpha, e_smooth, e_data, 0, 0};
    const dim3 block(kMaxSide, 1024 / kMaxSide);
    const bool grad = constancy != FLOW2D_CONSTANCY_GREY;
    a.untiled = constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED ? 1 : 0;
    a.log = constancy == FLOW2D_CONSTANCY_LOG_DERIVATIVES ? 1 : 0;
    const int px = h <= 16 ? 1 : (h <= 32 ? 2 : 4);
    if (px == 1)
        grad ? small_level_kernel<true, 1><<<1, block, 0, ctx->stream>>>(a)
             : small_level_kernel<false, 1><<<1, block, 0, ctx->stream>>>(a);
    else if (px == 2)
        grad ? small_level_kernel<true, 2><<<1, block, 0, ctx->stream>>>(a)
             : small_level_kernel<false, 2><<<1, block, 0, ctx->stream>>>(a);
    else
        grad ? small_level_kernel<true, 4><<<1, block, 0, ctx->stream>>>(a)
             : small_level_kernel<false, 4><<<1, block, 0, ctx->stream>>>(a);
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

}  // namespace flow2d
