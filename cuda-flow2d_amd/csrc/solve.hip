// Per-sweep solver kernels of the flow2d hot path for gfx950: the robust weights (phi, ksi) and
// one Jacobi sweep of the 2-field Euler-Lagrange system: brightness, gradient and log-derivative constancy.
// One launch here corresponds to one launch of the reference
// (compute_phi_ksi / solve_2d / solve_2d_log / solve_2d_grad, src/kernels/solve_2d.cu:43-198, 200-377, 391-669,
// 683-952).
//
// Mapping: one wave = 64 consecutive pixels of one image row, so each of the 8 input planes is
// read as contiguous 256-byte row segments; the x+-1 / y+-1 neighbours of a wave come from the same
// or the adjacent rows' segments.  For those to be L2 hits the neighbouring workgroups must run on the
// same XCD (each XCD has an L2 of its own): the grid is one-dimensional and XCD-aware (common.hpp,
// XcdTiles: every XCD walks a horizontal band of the level row-major), so HBM sees each plane about
// once: 40 B per pixel-sweep, 32 B per pixel for phi/ksi.  With the plain 2-D grid of rounds 1-2 the x
// neighbours of a tile sat on other XCDs and the kernels moved 1.69x their algorithmic bytes.  Arithmetic keeps the reference's order of operations; the file is
// built with -ffp-contract=off so no multiply-add is fused.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "solver_math.hpp"

namespace {

constexpr int kBlockX = 64;
constexpr int kBlockY = 4;

struct Neighbourhood {
    size_t c, l, r, u, d;  // element offsets of centre / left / right / up / down (mirrored at borders)
};

__device__ __forceinline__ Neighbourhood neighbourhood(int x, int y, int w, int h, int pitch)
{
    const size_t rc = static_cast<size_t>(y) * pitch;
    const size_t ru = static_cast<size_t>(mirror_index(y - 1, h)) * pitch;
    const size_t rd = static_cast<size_t>(mirror_index(y + 1, h)) * pitch;
    Neighbourhood n;
    n.c = rc + x;
    n.l = rc + mirror_index(x - 1, w);
    n.r = rc + mirror_index(x + 1, w);
    n.u = ru + x;
    n.d = rd + x;
    return n;
}

// fx, fy, ft of solve_2d.cu:311-321 (identical in compute_phi_ksi :164-174 and solve_2d_grad :798-808)
__device__ __forceinline__ void image_derivatives(const float* __restrict__ f0, const float* __restrict__ f1,
                                                  const Neighbourhood& n, float hx, float hy, float& fx, float& fy,
                                                  float& ft)
{
    fx = flow2d_math::diff4(f0[n.r], f0[n.l], f1[n.r], f1[n.l], 4.f * hx);
    fy = flow2d_math::diff4(f0[n.d], f0[n.u], f1[n.d], f1[n.u], 4.f * hy);
    ft = f1[n.c] - f0[n.c];
}

// ---- compute_phi_ksi: src/kernels/solve_2d.cu:43-198 ---------------------------------------------
__global__ __launch_bounds__(256) void phi_ksi_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                      const float* __restrict__ u, const float* __restrict__ v,
                                                      const float* __restrict__ du, const float* __restrict__ dv,
                                                      XcdTiles tiles, int w, int h, int pitch, float hx, float hy, float e_smooth,
                                                      float e_data, float* __restrict__ phi, float* __restrict__ ksi)
{
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + threadIdx.x;
    const int y = tile_y * kBlockY + threadIdx.y;
    if (x >= w || y >= h) return;
    const Neighbourhood n = neighbourhood(x, y, w, h, pitch);

    using namespace flow2d_math;
    const float dux = diff4(u[n.r], u[n.l], du[n.r], du[n.l], 2.f * hx);
    const float duy = diff4(u[n.d], u[n.u], du[n.d], du[n.u], 2.f * hy);
    const float dvx = diff4(v[n.r], v[n.l], dv[n.r], dv[n.l], 2.f * hx);
    const float dvy = diff4(v[n.d], v[n.u], dv[n.d], dv[n.u], 2.f * hy);
    phi[n.c] = phi_value(dux, duy, dvx, dvy, e_smooth);

    float fx, fy, ft;
    image_derivatives(f0, f1, n, hx, hy, fx, fy, ft);
    ksi[n.c] = ksi_value(fx, fy, ft, du[n.c], dv[n.c], e_data);
}

// The pointwise Jacobi update shared by solve_2d (solve_2d.cu:332-374) and solve_2d_grad (:889-931).
// SOR = false: Jacobi, the reference's sweep (reads du/dv, writes the separate tdu/tdv planes).
// SOR = true: opt-in relaxation in place (tdu == du, tdv == dv) of the pixels of one colour; their four
// neighbours have the other colour, so a half-sweep only reads values it does not write.
template <bool SOR = false>
__device__ __forceinline__ void jacobi_update(const float* __restrict__ u, const float* __restrict__ v,
                                              const float* du, const float* dv, const float* __restrict__ phi,
                                              const float* __restrict__ ksi, const Neighbourhood& n, int x, int y,
                                              int w, int h, float hx, float hy, float alpha, float J11, float J22,
                                              float J12, float J13, float J23, float* tdu, float* tdv,
                                              float omega = 1.f)
{
    const float hx_2 = alpha / (hx * hx);
    const float hy_2 = alpha / (hy * hy);
    const float xp = static_cast<float>(x < w - 1) * hx_2;
    const float xm = static_cast<float>(x > 0) * hx_2;
    const float yp = static_cast<float>(y < h - 1) * hy_2;
    const float ym = static_cast<float>(y > 0) * hy_2;

    using namespace flow2d_math;
    const float pc = phi[n.c];
    const float wxp = face_phi(phi[n.r], pc) * xp;
    const float wxm = face_phi(phi[n.l], pc) * xm;
    const float wyp = face_phi(phi[n.d], pc) * yp;
    const float wym = face_phi(phi[n.u], pc) * ym;
    const float sumH = sum_weights(wxp, wxm, wyp, wym);
    const float sumU = sum_flux(wxp, wxm, wyp, wym, u[n.r] + du[n.r], u[n.l] + du[n.l], u[n.d] + du[n.d],
                                u[n.u] + du[n.u], u[n.c]);
    const float sumV = sum_flux(wxp, wxm, wyp, wym, v[n.r] + dv[n.r], v[n.l] + dv[n.l], v[n.d] + dv[n.d],
                                v[n.u] + dv[n.u], v[n.c]);
    float r_du, r_dv;
    const float k = ksi[n.c];
    if (SOR)
        point_update_sor(k, update_denominator(k, J11, sumH), update_denominator(k, J22, sumH), J12, J13, J23, sumU,
                         sumV, du[n.c], dv[n.c], omega, r_du, r_dv);
    else
        point_update(k, update_denominator(k, J11, sumH), update_denominator(k, J22, sumH), J12, J13, J23, sumU, sumV,
                     dv[n.c], r_du, r_dv);
    tdu[n.c] = r_du;
    tdv[n.c] = r_dv;
}

// ---- solve_2d: src/kernels/solve_2d.cu:200-377 ---------------------------------------------------
__global__ __launch_bounds__(256) void sweep_grey_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                         const float* __restrict__ u, const float* __restrict__ v,
                                                         const float* __restrict__ du, const float* __restrict__ dv,
                                                         const float* __restrict__ phi, const float* __restrict__ ksi,
                                                         XcdTiles tiles, int w, int h, int pitch, float hx, float hy, float alpha,
                                                         float* __restrict__ tdu, float* __restrict__ tdv)
{
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + threadIdx.x;
    const int y = tile_y * kBlockY + threadIdx.y;
    if (x >= w || y >= h) return;
    const Neighbourhood n = neighbourhood(x, y, w, h, pitch);
    float fx, fy, ft;
    image_derivatives(f0, f1, n, hx, hy, fx, fy, ft);
    jacobi_update(u, v, du, dv, phi, ksi, n, x, y, w, h, hx, hy, alpha, fx * fx, fy * fy, fx * fy, fx * ft, fy * ft,
                  tdu, tdv);
}

// ---- solve_2d, streaming form (round 4) -----------------------------------------------------------------------------
// The tile form above lives for one row segment: 35 loads, a wait, the update, two stores, and the wave is gone -- its
// 672 MB per 4096^2 sweep take 165 us (4.1 TB/s) however little of them is re-read, the same time in which the rounds-1/2
// grid moved 1.7x the bytes.  Here a wave owns a strip of 62 columns (one halo lane on either side) and walks down
// `rows` image rows: every plane row is ONE load per lane (a 256-byte segment per wave), requested kSweepAhead rows
// before it is needed; the y neighbours are the 3-row windows in the lane's registers, the x neighbours come from the
// adjacent lanes by DPP.  The reflect rule needs no selects: the row above the image IS row 1 and the column left of it
// IS column 1, so the halo rows / lanes simply load the mirrored address (mirror_index) and everything else follows.
// Same expressions as jacobi_update (solve_2d.cu:311-374) in the same order: bit-identical.
// Where it pays (same box, tile form / streaming form, us per sweep): 4096^2 190 / 166 (another box: 165 / 164), 8192^2 651 /
// 661, but 2048^2 30 / 35 and 1920 x 1080 15.8 / 19.5 -- ten planes of a level up to about 6 Mpixel live in the 256 MB
// Infinity Cache, where the short-lived waves of the tile form reach 5.4-5.6 TB/s; beyond it both forms are held at
// 4.0-4.4 TB/s by HBM itself (ten separate plane streams; skewing the planes' base addresses against each other buys 3 %:
// profiles/r04_experiments).  The streaming form takes the levels of 8 Mpixel and more.
constexpr int kSweepValid = 62;
constexpr int kSweepAhead = 3;
constexpr size_t kSweepStreamMinPixels = size_t(8) << 20;

__device__ __forceinline__ float sweep_from_left(float v)  // lane i <- lane i-1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float sweep_from_right(float v)  // lane i <- lane i+1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

struct SweepRow {  // one image row of the planes a sweep reads, as loaded
    float f0, f1, u, v, du, dv, phi, ksi;
};

struct SweepPlanes {
    const float* f0;
    const float* f1;
    const float* u;
    const float* v;
    const float* du;
    const float* dv;
    const float* phi;
    const float* ksi;
};

// FRAMES_AHEAD: the frames of row `row + 1` with the other planes of row `row` (the gradient terms)
// LOG: the frames as log(I + 1) (solve_2d_log, solve_2d.cu:519-535)
template <bool FRAMES_AHEAD, bool LOG>
__device__ __forceinline__ SweepRow sweep_load(const SweepPlanes& p, int row, int h, int pitch, int xm)
{
    const size_t at = static_cast<size_t>(min(max(mirror_index(row, h), 0), h - 1)) * pitch + xm;
    const size_t af = FRAMES_AHEAD ? static_cast<size_t>(min(max(mirror_index(row + 1, h), 0), h - 1)) * pitch + xm : at;
    const float a = p.f0[af], b = p.f1[af];
    return SweepRow{LOG ? flow2d_math::log1p_frame(a) : a, LOG ? flow2d_math::log1p_frame(b) : b, p.u[at], p.v[at], p.du[at],
                    p.dv[at], p.phi[at], p.ksi[at]};
}

// GRAD: solve_2d_grad (solve_2d.cu:683-952).  Its tensor needs fx, fy, ft of the rows y-1, y, y+1 (second derivatives inside
// the reference's 16x8 blocks, the block's own edge value replicated: :816-841), i.e. the frames one row further ahead than
// the other planes: a window slot of row q then carries the frames' row q + 1, the derivatives of row y + 1 are formed when
// slot y + 1 arrives and kept in a 3-row window of their own.  Strips start on multiples of 16 rows, so the derivative row
// above a strip's first row is never read (y % 8 == 0 takes the pixel's own value).
// SOR (round 5): one red-black half-sweep of the opt-in successive over-relaxation IN PLACE (tdu == du, tdv == dv): only the
// pixels with (x + y) % 2 == colour are relaxed and stored.  Their four neighbours have the other colour, which no wave writes
// in this launch, so rows loaded ahead, halo rows and halo lanes only ever deliver values that are still the old ones where
// they count.
// MODE (round 5): 0 solve_2d; 1 solve_2d_grad; 2 the gradient term over TRUE neighbours (FLOW2D_CONSTANCY_GRADIENT_UNTILED, not
// a reference mode): the tensor's differences of fx, fy, ft reach across block edges, reflected at the image border by selects
// -- a derivative row computed from mirrored rows is NOT the mirrored derivative row (fy changes sign) -- and the derivative
// row above the strip is formed in the prologue; 3 solve_2d_log (solve_2d.cu:391-669): as 1 on log(I + 1), and the first
// derivatives, phi in the face weights and the flow neighbours of the sweep take the pixel's own value at a 16x8 block edge
// (that kernel's halo offsets are 0: :448,462,476,490).
template <int MODE, bool SOR>
__global__ __launch_bounds__(256) void sweep_stream_kernel(SweepPlanes p, XcdTiles tiles, int w, int h, int pitch, int rows,
                                                           float hx, float hy, float alpha, float* tdu, float* tdv, float omega,
                                                           int colour)
{
    constexpr bool GRAD = MODE != 0, LOG = MODE == 3, UNTILED = MODE == 2;
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    // halo lanes per side: one for the x neighbours of the planes; the gradient term also differentiates fx and ft in x, whose
    // own x neighbours must be true values, so it keeps two
    constexpr int kHalo = GRAD ? 2 : 1, kValid = 64 - 2 * kHalo;
    const int lane = threadIdx.x & 63;
    const int strip = tile_x * 4 + (threadIdx.x >> 6);
    if (strip * kValid >= w) return;  // whole wave
    const int x = strip * kValid - kHalo + lane;
    const int xm = min(max(mirror_index(x, w), 0), w - 1);
    const bool stores = lane >= kHalo && lane < 64 - kHalo && x < w;
    const int y0 = tile_y * rows, y1 = min(y0 + rows, h);
    using namespace flow2d_math;
    const float hx_2 = alpha / (hx * hx);
    const float hy_2 = alpha / (hy * hy);
    const float xp = static_cast<float>(x < w - 1) * hx_2;
    const float xm_w = static_cast<float>(x > 0) * hx_2;

    // window slot = (row + 1) mod 3; rows y0 - 1 and y0 first, then the rows in flight
    SweepRow win[3];
    SweepRow ahead[kSweepAhead];
    win[0] = sweep_load<GRAD, LOG>(p, y0 - 1, h, pitch, xm);
    win[1] = sweep_load<GRAD, LOG>(p, y0, h, pitch, xm);
#pragma unroll
    for (int i = 0; i < kSweepAhead; ++i) ahead[i] = sweep_load<GRAD, LOG>(p, y0 + 1 + i, h, pitch, xm);

    const bool x_lo = (x & 15) == 0, x_hi = (x & 15) == 15 || x == w - 1;
    const bool x_hi15 = (x & 15) == 15;             // solve_2d_log's own neighbours: the block edge only (the image edge reflects)
    const bool at_l = x == 0, at_r = x == w - 1;    // UNTILED: the reflect rule of the image border
    // frame derivatives of row q from the frames' rows above / at / below it, solve_2d.cu:311-321 (:798-808; log: :519-535 with the
    // block rule in x and y)
    auto derivatives = [&](int q, float f0u, float f1u, float f0c, float f1c, float f0d, float f1d, float& fx, float& fy, float& ft) {
        const float f0l0 = sweep_from_left(f0c), f0r0 = sweep_from_right(f0c), f1l0 = sweep_from_left(f1c), f1r0 = sweep_from_right(f1c);
        if (LOG) {
            const bool q_lo = (q & 7) == 0, q_hi = (q & 7) == 7;
            fx = diff4(x_hi15 ? f0c : f0r0, x_lo ? f0c : f0l0, x_hi15 ? f1c : f1r0, x_lo ? f1c : f1l0, 4.f * hx);
            fy = diff4(q_hi ? f0c : f0d, q_lo ? f0c : f0u, q_hi ? f1c : f1d, q_lo ? f1c : f1u, 4.f * hy);
        } else {
            fx = diff4(f0r0, f0l0, f1r0, f1l0, 4.f * hx);
            fy = diff4(f0d, f0u, f1d, f1u, 4.f * hy);
        }
        ft = f1c - f0c;
    };
    // GRAD: (fx, fy, ft) of the rows y-1, y, y+1; slot of row q = (q + 1) mod 3 like the plane windows
    float dfx[3] = {0.f, 0.f, 0.f}, dfy[3] = {0.f, 0.f, 0.f}, dft[3] = {0.f, 0.f, 0.f};
    const float hx_1 = 1.0 / (2.0 * hx);  // evaluated in double, rounded to float (solve_2d.cu:868-869)
    const float hy_1 = 1.0 / (2.0 * hy);
    if (GRAD) {  // row y0: the frames' rows y0 - 1 (loaded here), y0 (in slot y0 - 1) and y0 + 1 (in slot y0)
        auto frame = [&](const float* plane, int row) {
            const float value = plane[static_cast<size_t>(min(max(mirror_index(row, h), 0), h - 1)) * pitch + xm];
            return LOG ? log1p_frame(value) : value;
        };
        const float a1 = frame(p.f0, y0 - 1), b1 = frame(p.f1, y0 - 1);
        derivatives(y0, a1, b1, win[0].f0, win[0].f1, win[1].f0, win[1].f1, dfx[1], dfy[1], dft[1]);
        dfx[0] = dfx[1], dfy[0] = dfy[1], dft[0] = dft[1];  // (row y0 - 1: never read by the block rule, y0 % 8 == 0)
        if (UNTILED)  // true neighbours: row y0 - 1 from the frames' rows y0 - 2, y0 - 1, y0 (row 0 takes the row below instead)
            derivatives(y0 - 1, frame(p.f0, y0 - 2), frame(p.f1, y0 - 2), a1, b1, win[0].f0, win[0].f1, dfx[0], dfy[0], dft[0]);
    }

    auto step = [&](int y, int sy, const SweepRow& up, const SweepRow& c, const SweepRow& down) {
        float J11, J22, J12, J13, J23;
        if (!GRAD) {
            float fx, fy, ft;
            derivatives(y, up.f0, up.f1, c.f0, c.f1, down.f0, down.f1, fx, fy, ft);
            J11 = fx * fx, J22 = fy * fy, J12 = fx * fy, J13 = fx * ft, J23 = fy * ft;
        } else {
            // sy = slot of row y; derivatives of row y + 1 from the frames' rows y (slot y - 1), y + 1 (slot y), y + 2 (slot y + 1)
            const int su = (sy + 2) % 3, sd = (sy + 1) % 3;
            derivatives(y + 1, up.f0, up.f1, c.f0, c.f1, down.f0, down.f1, dfx[sd], dfy[sd], dft[sd]);
            const float fxc = dfx[sy], fyc = dfy[sy], ftc = dft[sy];
            // cross-lane reads with every lane active, the block rule as selects afterwards (solve_2d.cu:816-841)
            const float fx_l0 = sweep_from_left(fxc), fx_r0 = sweep_from_right(fxc);
            const float ft_l0 = sweep_from_left(ftc), ft_r0 = sweep_from_right(ftc);
            float fx_l, fx_r, ft_l, ft_r, fx_u, fx_d, fy_u, fy_d, ft_u, ft_d;
            if (UNTILED) {  // true neighbours, reflected at the image border
                const bool top = y == 0, bot = y == h - 1;
                fx_l = at_l ? fx_r0 : fx_l0, fx_r = at_r ? fx_l0 : fx_r0;
                ft_l = at_l ? ft_r0 : ft_l0, ft_r = at_r ? ft_l0 : ft_r0;
                fx_u = top ? dfx[sd] : dfx[su], fx_d = bot ? dfx[su] : dfx[sd];
                fy_u = top ? dfy[sd] : dfy[su], fy_d = bot ? dfy[su] : dfy[sd];
                ft_u = top ? dft[sd] : dft[su], ft_d = bot ? dft[su] : dft[sd];
            } else {
                const bool y_lo = (y & 7) == 0, y_hi = (y & 7) == 7 || y == h - 1;
                fx_l = x_lo ? fxc : fx_l0, fx_r = x_hi ? fxc : fx_r0;
                ft_l = x_lo ? ftc : ft_l0, ft_r = x_hi ? ftc : ft_r0;
                fx_u = y_lo ? fxc : dfx[su], fx_d = y_hi ? fxc : dfx[sd];
                fy_u = y_lo ? fyc : dfy[su], fy_d = y_hi ? fyc : dfy[sd];
                ft_u = y_lo ? ftc : dft[su], ft_d = y_hi ? ftc : dft[sd];
            }
            const float fxx = (fx_r - fx_l) * hx_1;
            const float fxy = (fx_d - fx_u) * hy_1;
            const float fyy = (fy_d - fy_u) * hy_1;
            const float fxt = (ft_r - ft_l) * hx_1;
            const float fyt = (ft_d - ft_u) * hy_1;
            gradient_tensor(fxx, fxy, fyy, fxt, fyt, J11, J22, J12, J13, J23);
        }
        // jacobi_update, value form
        const float yp = static_cast<float>(y < h - 1) * hy_2;
        const float ym = static_cast<float>(y > 0) * hy_2;
        const float pc = c.phi;
        const float su = c.u + c.du, sv = c.v + c.dv;  // the neighbours' full flow, formed once per pixel
        const float p_r0 = sweep_from_right(pc), p_l0 = sweep_from_left(pc);
        const float su_r0 = sweep_from_right(su), su_l0 = sweep_from_left(su), sv_r0 = sweep_from_right(sv), sv_l0 = sweep_from_left(sv);
        // solve_2d_log: own value at the 16x8 block edge for phi and the flow as well (neighbourhood_log, :612-633)
        const bool b_lo = LOG && (y & 7) == 0, b_hi = LOG && (y & 7) == 7, e_lo = LOG && x_lo, e_hi = LOG && x_hi15;
        const float wxp = face_phi(e_hi ? pc : p_r0, pc) * xp;
        const float wxm = face_phi(e_lo ? pc : p_l0, pc) * xm_w;
        const float wyp = face_phi(b_hi ? pc : down.phi, pc) * yp;
        const float wym = face_phi(b_lo ? pc : up.phi, pc) * ym;
        const float sumH = sum_weights(wxp, wxm, wyp, wym);
        const float sumU = sum_flux(wxp, wxm, wyp, wym, e_hi ? su : su_r0, e_lo ? su : su_l0, b_hi ? su : down.u + down.du,
                                    b_lo ? su : up.u + up.du, c.u);
        const float sumV = sum_flux(wxp, wxm, wyp, wym, e_hi ? sv : sv_r0, e_lo ? sv : sv_l0, b_hi ? sv : down.v + down.dv,
                                    b_lo ? sv : up.v + up.dv, c.v);
        float r_du, r_dv;
        const float k = c.ksi;
        if (SOR)
            point_update_sor(k, update_denominator(k, J11, sumH), update_denominator(k, J22, sumH), J12, J13, J23, sumU, sumV,
                             c.du, c.dv, omega, r_du, r_dv);
        else
            point_update(k, update_denominator(k, J11, sumH), update_denominator(k, J22, sumH), J12, J13, J23, sumU, sumV, c.dv,
                         r_du, r_dv);
        if (stores && (!SOR || ((x + y) & 1) == colour)) {
            const size_t at = static_cast<size_t>(y) * pitch + x;
            tdu[at] = r_du;
            tdv[at] = r_dv;
        }
    };

    // three rows per trip, so that every window slot is a compile-time index
    for (int y = y0; y < y1; y += 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (y + j >= y1) break;  // wave-uniform
            // row y + j + 1 arrives: the oldest of the rows in flight; request row y + j + 1 + kSweepAhead
            win[(j + 2) % 3] = ahead[0];
#pragma unroll
            for (int i = 0; i + 1 < kSweepAhead; ++i) ahead[i] = ahead[i + 1];
            ahead[kSweepAhead - 1] = sweep_load<GRAD, LOG>(p, y + j + 1 + kSweepAhead, h, pitch, xm);
            step(y + j, (j + 1) % 3, win[j % 3], win[(j + 1) % 3], win[(j + 2) % 3]);
        }
    }
}

// ---- solve_2d_grad: src/kernels/solve_2d.cu:683-952 ----------------------------------------------
// The reference differentiates fx, fy, ft inside each 16x8 thread block with the block's own edge
// value replicated into the halo (:816-841), so the motion tensor depends on that tiling.  A 64x8
// workgroup here covers four such blocks; fx/fy/ft go through LDS and neighbours are taken with the
// same replicate-at-16x8-edge rule.  Where the image edge falls inside a block the reference reads
// an unwritten LDS slot (undefined); this kernel replicates the edge pixel there.
constexpr int kGradTileX = 16;
constexpr int kGradTileY = 8;

__global__ __launch_bounds__(512) void sweep_grad_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                         const float* __restrict__ u, const float* __restrict__ v,
                                                         const float* __restrict__ du, const float* __restrict__ dv,
                                                         const float* __restrict__ phi, const float* __restrict__ ksi,
                                                         XcdTiles tiles, int w, int h, int pitch, float hx, float hy, float alpha,
                                                         float* __restrict__ tdu, float* __restrict__ tdv)
{
    __shared__ float s_fx[kGradTileY][kBlockX];
    __shared__ float s_fy[kGradTileY][kBlockX];
    __shared__ float s_ft[kGradTileY][kBlockX];

    const int tx = threadIdx.x, ty = threadIdx.y;
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + tx;
    const int y = tile_y * kGradTileY + ty;
    const bool inside = x < w && y < h;
    Neighbourhood n{};
    if (inside) {
        n = neighbourhood(x, y, w, h, pitch);
        float fx, fy, ft;
        image_derivatives(f0, f1, n, hx, hy, fx, fy, ft);
        s_fx[ty][tx] = fx;
        s_fy[ty][tx] = fy;
        s_ft[ty][tx] = ft;
    }
    __syncthreads();
    if (!inside) return;

    const int xa = (tx % kGradTileX == 0) ? tx : tx - 1;
    const int xb = (tx % kGradTileX == kGradTileX - 1 || x == w - 1) ? tx : tx + 1;
    const int ya = (ty == 0) ? ty : ty - 1;
    const int yb = (ty == kGradTileY - 1 || y == h - 1) ? ty : ty + 1;

    const float hx_1 = 1.0 / (2.0 * hx);  // evaluated in double, rounded to float (solve_2d.cu:868-869)
    const float hy_1 = 1.0 / (2.0 * hy);

    const float fxx = (s_fx[ty][xb] - s_fx[ty][xa]) * hx_1;
    const float fxy = (s_fx[yb][tx] - s_fx[ya][tx]) * hy_1;
    const float fyy = (s_fy[yb][tx] - s_fy[ya][tx]) * hy_1;
    const float fxt = (s_ft[ty][xb] - s_ft[ty][xa]) * hx_1;
    const float fyt = (s_ft[yb][tx] - s_ft[ya][tx]) * hy_1;

    float J11, J22, J12, J13, J23;
    flow2d_math::gradient_tensor(fxx, fxy, fyy, fxt, fyt, J11, J22, J12, J13, J23);
    jacobi_update(u, v, du, dv, phi, ksi, n, x, y, w, h, hx, hy, alpha, J11, J22, J12, J13, J23, tdu, tdv);
}

// ---- solve_2d_log: src/kernels/solve_2d.cu:391-669 (DataConstancy::LogDerivatives) ---------------------
// solve_2d_grad on log(I + 1.0f) -- with one more tile effect: the halo offsets of this kernel are
// `global_x - 1 + 1` etc. (= 0, :448,462,476,490), so the halo of EVERY plane (frames, u, v, du, dv, phi, ksi)
// holds the 16x8 block's own edge pixel.  Inside a block a neighbour is the true one; a block that sticks out of
// the image has its out-of-range threads load the reflected pixel (:434-435), which the last in-range pixel reads.
__device__ __forceinline__ Neighbourhood neighbourhood_log(int x, int y, int w, int h, int pitch)
{
    const int xl = (x % kGradTileX == 0) ? x : x - 1;
    const int xr = (x % kGradTileX == kGradTileX - 1) ? x : mirror_index(x + 1, w);
    const int yu = (y % kGradTileY == 0) ? y : y - 1;
    const int yd = (y % kGradTileY == kGradTileY - 1) ? y : mirror_index(y + 1, h);
    const size_t rc = static_cast<size_t>(y) * pitch;
    Neighbourhood n;
    n.c = rc + x;
    n.l = rc + xl;
    n.r = rc + xr;
    n.u = static_cast<size_t>(yu) * pitch + x;
    n.d = static_cast<size_t>(yd) * pitch + x;
    return n;
}

__global__ __launch_bounds__(512) void sweep_log_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                        const float* __restrict__ u, const float* __restrict__ v,
                                                        const float* __restrict__ du, const float* __restrict__ dv,
                                                        const float* __restrict__ phi, const float* __restrict__ ksi,
                                                        XcdTiles tiles, int w, int h, int pitch, float hx, float hy, float alpha,
                                                        float* __restrict__ tdu, float* __restrict__ tdv)
{
    __shared__ float s_fx[kGradTileY][kBlockX];
    __shared__ float s_fy[kGradTileY][kBlockX];
    __shared__ float s_ft[kGradTileY][kBlockX];

    const int tx = threadIdx.x, ty = threadIdx.y;
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + tx;
    const int y = tile_y * kGradTileY + ty;
    const bool inside = x < w && y < h;
    Neighbourhood n{};
    if (inside) {
        n = neighbourhood_log(x, y, w, h, pitch);
        using flow2d_math::log1p_frame;
        s_fx[ty][tx] = flow2d_math::diff4(log1p_frame(f0[n.r]), log1p_frame(f0[n.l]), log1p_frame(f1[n.r]),
                                          log1p_frame(f1[n.l]), 4.f * hx);   // :519-522
        s_fy[ty][tx] = flow2d_math::diff4(log1p_frame(f0[n.d]), log1p_frame(f0[n.u]), log1p_frame(f1[n.d]),
                                          log1p_frame(f1[n.u]), 4.f * hy);   // :523-526
        s_ft[ty][tx] = log1p_frame(f1[n.c]) - log1p_frame(f0[n.c]);          // :534-535
    }
    __syncthreads();
    if (!inside) return;

    // fx, fy, ft: own value replicated at the block edge (:542-565); unwritten slot at the image edge -> own value
    const int xa = (tx % kGradTileX == 0) ? tx : tx - 1;
    const int xb = (tx % kGradTileX == kGradTileX - 1 || x == w - 1) ? tx : tx + 1;
    const int ya = (ty == 0) ? ty : ty - 1;
    const int yb = (ty == kGradTileY - 1 || y == h - 1) ? ty : ty + 1;
    const float hx_1 = 1.0 / (2.0 * hx);  // :584-585
    const float hy_1 = 1.0 / (2.0 * hy);
    const float fxx = (s_fx[ty][xb] - s_fx[ty][xa]) * hx_1;
    const float fxy = (s_fx[yb][tx] - s_fx[ya][tx]) * hy_1;
    const float fyy = (s_fy[yb][tx] - s_fy[ya][tx]) * hy_1;
    const float fxt = (s_ft[ty][xb] - s_ft[ty][xa]) * hx_1;
    const float fyt = (s_ft[yb][tx] - s_ft[ya][tx]) * hy_1;
    float J11, J22, J12, J13, J23;
    flow2d_math::gradient_tensor(fxx, fxy, fyy, fxt, fyt, J11, J22, J12, J13, J23);
    // the smoothness term sees the same block-edge replication through n (:612-633)
    jacobi_update(u, v, du, dv, phi, ksi, n, x, y, w, h, hx, hy, alpha, J11, J22, J12, J13, J23, tdu, tdv);
}

// ---- gradient constancy with true neighbours (FLOW2D_CONSTANCY_GRADIENT_UNTILED; not a reference kernel) ------
// Same motion tensor as solve_2d_grad, but the second derivatives are central differences of fx, fy, ft over the
// real neighbours x+-1, y+-1 (reflected at the image border like every other neighbour of the solver) instead of
// being cut at the reference's 16x8 launch tiles.  This per-sweep form simply evaluates fx, fy, ft at the four
// neighbours (all L1/L2 hits); the fused and single-workgroup kernels keep them in registers / LDS.
__device__ __forceinline__ void untiled_gradient_tensor(const float* __restrict__ f0, const float* __restrict__ f1, int x,
                                                        int y, int w, int h, int pitch, float hx, float hy, float& J11,
                                                        float& J22, float& J12, float& J13, float& J23)
{
    const int xa = mirror_index(x - 1, w), xb = mirror_index(x + 1, w);
    const int ya = mirror_index(y - 1, h), yb = mirror_index(y + 1, h);
    float fx_a, fy_a, ft_a, fx_b, fy_b, ft_b, fx_u, fy_u, ft_u, fx_d, fy_d, ft_d;
    image_derivatives(f0, f1, neighbourhood(xa, y, w, h, pitch), hx, hy, fx_a, fy_a, ft_a);
    image_derivatives(f0, f1, neighbourhood(xb, y, w, h, pitch), hx, hy, fx_b, fy_b, ft_b);
    image_derivatives(f0, f1, neighbourhood(x, ya, w, h, pitch), hx, hy, fx_u, fy_u, ft_u);
    image_derivatives(f0, f1, neighbourhood(x, yb, w, h, pitch), hx, hy, fx_d, fy_d, ft_d);
    const float hx_1 = 1.0 / (2.0 * hx);  // double, rounded to float, as in solve_2d.cu:868-869
    const float hy_1 = 1.0 / (2.0 * hy);
    const float fxx = (fx_b - fx_a) * hx_1;
    const float fxy = (fx_d - fx_u) * hy_1;
    const float fyy = (fy_d - fy_u) * hy_1;
    const float fxt = (ft_b - ft_a) * hx_1;
    const float fyt = (ft_d - ft_u) * hy_1;
    flow2d_math::gradient_tensor(fxx, fxy, fyy, fxt, fyt, J11, J22, J12, J13, J23);
}

__global__ __launch_bounds__(256) void sweep_grad_untiled_kernel(
    const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ u, const float* __restrict__ v,
    const float* __restrict__ du, const float* __restrict__ dv, const float* __restrict__ phi,
    const float* __restrict__ ksi, XcdTiles tiles, int w, int h, int pitch, float hx, float hy, float alpha, float* __restrict__ tdu,
    float* __restrict__ tdv)
{
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + threadIdx.x;
    const int y = tile_y * kBlockY + threadIdx.y;
    if (x >= w || y >= h) return;
    float J11, J22, J12, J13, J23;
    untiled_gradient_tensor(f0, f1, x, y, w, h, pitch, hx, hy, J11, J22, J12, J13, J23);
    jacobi_update(u, v, du, dv, phi, ksi, neighbourhood(x, y, w, h, pitch), x, y, w, h, hx, hy, alpha, J11, J22, J12,
                  J13, J23, tdu, tdv);
}

__global__ __launch_bounds__(256) void sor_grad_untiled_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                               const float* __restrict__ u, const float* __restrict__ v,
                                                               float* du, float* dv, const float* __restrict__ phi,
                                                               const float* __restrict__ ksi, XcdTiles tiles, int w, int h, int pitch,
                                                               float hx, float hy, float alpha, float omega, int colour)
{
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + threadIdx.x;
    const int y = tile_y * kBlockY + threadIdx.y;
    if (x >= w || y >= h || ((x + y) & 1) != colour) return;
    float J11, J22, J12, J13, J23;
    untiled_gradient_tensor(f0, f1, x, y, w, h, pitch, hx, hy, J11, J22, J12, J13, J23);
    jacobi_update<true>(u, v, du, dv, phi, ksi, neighbourhood(x, y, w, h, pitch), x, y, w, h, hx, hy, alpha, J11, J22,
                        J12, J13, J23, du, dv, omega);
}

// ---- opt-in red-black SOR half-sweeps (no counterpart in the reference; BASELINE.json names the scheme) ---
// One launch relaxes the pixels with (x + y) % 2 == colour in place.  A full iteration = colour 0 then colour 1.
__global__ __launch_bounds__(256) void sor_grey_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                       const float* __restrict__ u, const float* __restrict__ v,
                                                       float* du, float* dv, const float* __restrict__ phi,
                                                       const float* __restrict__ ksi, XcdTiles tiles, int w, int h, int pitch,
                                                       float hx, float hy, float alpha, float omega, int colour)
{
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + threadIdx.x;
    const int y = tile_y * kBlockY + threadIdx.y;
    if (x >= w || y >= h || ((x + y) & 1) != colour) return;
    const Neighbourhood n = neighbourhood(x, y, w, h, pitch);
    float fx, fy, ft;
    image_derivatives(f0, f1, n, hx, hy, fx, fy, ft);
    jacobi_update<true>(u, v, du, dv, phi, ksi, n, x, y, w, h, hx, hy, alpha, fx * fx, fy * fy, fx * fy, fx * ft,
                        fy * ft, du, dv, omega);
}

__global__ __launch_bounds__(512) void sor_grad_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                       const float* __restrict__ u, const float* __restrict__ v,
                                                       float* du, float* dv, const float* __restrict__ phi,
                                                       const float* __restrict__ ksi, XcdTiles tiles, int w, int h, int pitch,
                                                       float hx, float hy, float alpha, float omega, int colour)
{
    __shared__ float s_fx[kGradTileY][kBlockX];
    __shared__ float s_fy[kGradTileY][kBlockX];
    __shared__ float s_ft[kGradTileY][kBlockX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    unsigned tile_x, tile_y;
    if (!xcd_tile(tiles, blockIdx.x, tile_x, tile_y)) return;
    const int x = tile_x * kBlockX + tx;
    const int y = tile_y * kGradTileY + ty;
    const bool inside = x < w && y < h;
    Neighbourhood n{};
    if (inside) {
        n = neighbourhood(x, y, w, h, pitch);
        image_derivatives(f0, f1, n, hx, hy, s_fx[ty][tx], s_fy[ty][tx], s_ft[ty][tx]);
    }
    __syncthreads();
    if (!inside || ((x + y) & 1) != colour) return;
    const int xa = (tx % kGradTileX == 0) ? tx : tx - 1;
    const int xb = (tx % kGradTileX == kGradTileX - 1 || x == w - 1) ? tx : tx + 1;
    const int ya = (ty == 0) ? ty : ty - 1;
    const int yb = (ty == kGradTileY - 1 || y == h - 1) ? ty : ty + 1;
    const float hx_1 = 1.0 / (2.0 * hx);
    const float hy_1 = 1.0 / (2.0 * hy);
    const float fxx = (s_fx[ty][xb] - s_fx[ty][xa]) * hx_1;
    const float fxy = (s_fx[yb][tx] - s_fx[ya][tx]) * hy_1;
    const float fyy = (s_fy[yb][tx] - s_fy[ya][tx]) * hy_1;
    const float fxt = (s_ft[ty][xb] - s_ft[ty][xa]) * hx_1;
    const float fyt = (s_ft[yb][tx] - s_ft[ya][tx]) * hy_1;
    float J11, J22, J12, J13, J23;
    flow2d_math::gradient_tensor(fxx, fxy, fyy, fxt, fyt, J11, J22, J12, J13, J23);
    jacobi_update<true>(u, v, du, dv, phi, ksi, n, x, y, w, h, hx, hy, alpha, J11, J22, J12, J13, J23, du, dv, omega);
}

bool solver_planes_ok(const float* const* planes, int count, size_t w, size_t h, size_t pitch_bytes)
{
    for (int i = 0; i < count; ++i)
        if (!flow2d::plane_args_ok(planes[i], w, h, pitch_bytes)) return false;
    return w >= 2 && h >= 2;
}

}  // namespace

namespace flow2d {

int launch_phi_ksi(flow2d_context* ctx, const float* f0, const float* f1, const float* u, const float* v,
                   const float* du, const float* dv, size_t w, size_t h, size_t pitch_bytes, float hx, float hy,
                   float e_smooth, float e_data, float* phi, float* ksi)
{
    if (ctx->batch_count > 1) {  // the per-sweep kernels are not batched: one launch per instance
        const unsigned n = ctx->batch_count;
        const size_t s = ctx->batch_stride_floats;
        ctx->batch_count = 1;
        int st = FLOW2D_OK;
        for (unsigned b = 0; b < n && st == FLOW2D_OK; ++b)
            st = launch_phi_ksi(ctx, f0 + b * s, f1 + b * s, u + b * s, v + b * s, du + b * s, dv + b * s, w, h,
                                pitch_bytes, hx, hy, e_smooth, e_data, phi + b * s, ksi + b * s);
        ctx->batch_count = n;
        return st;
    }
    const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kBlockY));
        const dim3 grid(xcd_grid(tiles));
    phi_ksi_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(f0, f1, u, v, du, dv, tiles, (int)w, (int)h,
                                                                     (int)(pitch_bytes / 4), hx, hy, e_smooth, e_data,
                                                                     phi, ksi);
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

// levels of 8 Mpixel and more (below, the ten planes sit in the Infinity Cache and the short-lived waves of the tile forms are faster)
static bool sweep_streams(size_t w, size_t h) { return w >= 2 * kSweepValid && h >= 8 && w * h >= kSweepStreamMinPixels; }
// (rows per strip, round 6: 16 / 32 / 64 rows fetch 718 / 655 / 626 MB per 4096^2 sweep -- the halo rows of a strip are re-read by
//  its neighbour -- for 537 MB of algorithmic reads, and take 155.6 / 152.7-155.8 / 165-167 us on one box, 158-161 / 162-165 us on
//  another (8192^2: 555-573 / 569-602 us): the bytes 32 rows save come out of the caches, the time they cost does not -- 16 stays;
//  profiles/r06_experiments/per_sweep_strip_rows_ab*.txt)
static unsigned sweep_strip_rows()
{
#ifdef FLOW2D_DEV_BUILD  // (developer builds: strip height override, a multiple of 16)
    if (const char* e = std::getenv("FLOW2D_SWEEP_ROWS")) return std::max(16, std::atoi(e) / 16 * 16);
#endif
    return 16;
}
static XcdTiles sweep_stream_tiles(size_t w, size_t h, int constancy)
{
    const unsigned valid = constancy == FLOW2D_CONSTANCY_GREY ? kSweepValid : kSweepValid - 2;  // (the gradient terms: two halo lanes per side)
    return xcd_tiles(div_up(div_up(w, valid), 4), div_up(h, sweep_strip_rows()));
}

int launch_sweep(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u, const float* v,
                 const float* du, const float* dv, const float* phi, const float* ksi, size_t w, size_t h,
                 size_t pitch_bytes, float hx, float hy, float alpha, float* tdu, float* tdv)
{
    if (ctx->batch_count > 1) {
        const unsigned n = ctx->batch_count;
        const size_t s = ctx->batch_stride_floats;
        ctx->batch_count = 1;
        int st = FLOW2D_OK;
        for (unsigned b = 0; b < n && st == FLOW2D_OK; ++b)
            st = launch_sweep(ctx, constancy, f0 + b * s, f1 + b * s, u + b * s, v + b * s, du + b * s, dv + b * s,
                              phi + b * s, ksi + b * s, w, h, pitch_bytes, hx, hy, alpha, tdu + b * s, tdv + b * s);
        ctx->batch_count = n;
        return st;
    }
    if (sweep_streams(w, h)) {
        // streaming form: strips of 62 columns (60: the gradient terms), four to a workgroup, 16 rows to a strip: the two halo
        // rows of a strip are its y neighbours' rows, which the XCD-aware tile order keeps in the same L2.  (us per 4096^2 /
        // 8192^2 Grey sweep by strip height, one box: 8 rows 169 / 570, 16 160 / 564, 24 160 / 615, 33 167 / 610, 64 168 / 617,
        // 128 171 / 599, 256 190 / 630.)
        const XcdTiles tiles = sweep_stream_tiles(w, h, constancy);
        const SweepPlanes planes{f0, f1, u, v, du, dv, phi, ksi};
        const dim3 grid(xcd_grid(tiles));
        const int iw = (int)w, ih = (int)h, ip = (int)(pitch_bytes / 4), rows = (int)sweep_strip_rows();
        if (constancy == FLOW2D_CONSTANCY_GRADIENT)
            sweep_stream_kernel<1, false><<<grid, 256, 0, ctx->stream>>>(planes, tiles, iw, ih, ip, rows, hx, hy, alpha, tdu, tdv, 1.f, 0);
        else if (constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED)
            sweep_stream_kernel<2, false><<<grid, 256, 0, ctx->stream>>>(planes, tiles, iw, ih, ip, rows, hx, hy, alpha, tdu, tdv, 1.f, 0);
        else if (constancy == FLOW2D_CONSTANCY_LOG_DERIVATIVES)
            sweep_stream_kernel<3, false><<<grid, 256, 0, ctx->stream>>>(planes, tiles, iw, ih, ip, rows, hx, hy, alpha, tdu, tdv, 1.f, 0);
        else
            sweep_stream_kernel<0, false><<<grid, 256, 0, ctx->stream>>>(planes, tiles, iw, ih, ip, rows, hx, hy, alpha, tdu, tdv, 1.f, 0);
    } else if (constancy == FLOW2D_CONSTANCY_GRADIENT) {
        const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kGradTileY));
        const dim3 grid(xcd_grid(tiles));
        sweep_grad_kernel<<<grid, dim3(kBlockX, kGradTileY), 0, ctx->stream>>>(
            f0, f1, u, v, du, dv, phi, ksi, tiles, (int)w, (int)h, (int)(pitch_bytes / 4), hx, hy, alpha, tdu, tdv);
    } else if (constancy == FLOW2D_CONSTANCY_LOG_DERIVATIVES) {
        const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kGradTileY));
        const dim3 grid(xcd_grid(tiles));
        sweep_log_kernel<<<grid, dim3(kBlockX, kGradTileY), 0, ctx->stream>>>(
            f0, f1, u, v, du, dv, phi, ksi, tiles, (int)w, (int)h, (int)(pitch_bytes / 4), hx, hy, alpha, tdu, tdv);
    } else if (constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED) {
        const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kBlockY));
        const dim3 grid(xcd_grid(tiles));
        sweep_grad_untiled_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
            f0, f1, u, v, du, dv, phi, ksi, tiles, (int)w, (int)h, (int)(pitch_bytes / 4), hx, hy, alpha, tdu, tdv);
    } else {
        const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kBlockY));
        const dim3 grid(xcd_grid(tiles));
        sweep_grey_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
            f0, f1, u, v, du, dv, phi, ksi, tiles, (int)w, (int)h, (int)(pitch_bytes / 4), hx, hy, alpha, tdu, tdv);
    }
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}

// One red-black SOR iteration (two half-sweeps) in place on du / dv.
int launch_sor_iteration(flow2d_context* ctx, int constancy, const float* f0, const float* f1, const float* u,
                         const float* v, float* du, float* dv, const float* phi, const float* ksi, size_t w, size_t h,
                         size_t pitch_bytes, float hx, float hy, float alpha, float omega)
{
    if (ctx->batch_count > 1) {
        const unsigned n = ctx->batch_count;
        const size_t s = ctx->batch_stride_floats;
        ctx->batch_count = 1;
        int st = FLOW2D_OK;
        for (unsigned b = 0; b < n && st == FLOW2D_OK; ++b)
            st = launch_sor_iteration(ctx, constancy, f0 + b * s, f1 + b * s, u + b * s, v + b * s, du + b * s, dv + b * s,
                                      phi + b * s, ksi + b * s, w, h, pitch_bytes, hx, hy, alpha, omega);
        ctx->batch_count = n;
        return st;
    }
    for (int colour = 0; colour < 2; ++colour) {
        if (sweep_streams(w, h)) {
            // (round 5) the half-sweep as a streaming strip kernel, in place: the same strips as the Jacobi form.
            // Invariant of the in-place form (tdu == du, tdv == dv): the halo lanes and halo rows of a strip load pixels
            // of the colour being relaxed that a NEIGHBOURING wave stores in this launch -- formally a race, in effect none: a
            // relaxed pixel's update reads its four neighbours, which have the OTHER colour and are only read in this launch, and
            // its own old value, which the owning wave loads itself; own-colour values from halo lanes / rows are dead (the kernel's
            // du / dv pointers carry no __restrict__, so the compiler may not merge or hoist those loads across the stores).
            const XcdTiles tiles = sweep_stream_tiles(w, h, constancy);
            const SweepPlanes planes{f0, f1, u, v, du, dv, phi, ksi};
            const dim3 grid(xcd_grid(tiles));
            const int iw = (int)w, ih = (int)h, ip = (int)(pitch_bytes / 4), rows = (int)sweep_strip_rows();
            if (constancy == FLOW2D_CONSTANCY_GRADIENT)
                sweep_stream_kernel<1, true><<<grid, 256, 0, ctx->stream>>>(planes, tiles, iw, ih, ip, rows, hx, hy, alpha, du, dv, omega, colour);
            else if (constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED)
                sweep_stream_kernel<2, true><<<grid, 256, 0, ctx->stream>>>(planes, tiles, iw, ih, ip, rows, hx, hy, alpha, du, dv, omega, colour);
            else
                sweep_stream_kernel<0, true><<<grid, 256, 0, ctx->stream>>>(planes, tiles, iw, ih, ip, rows, hx, hy, alpha, du, dv, omega, colour);
        } else if (constancy == FLOW2D_CONSTANCY_GRADIENT) {
            const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kGradTileY));
        const dim3 grid(xcd_grid(tiles));
            sor_grad_kernel<<<grid, dim3(kBlockX, kGradTileY), 0, ctx->stream>>>(
                f0, f1, u, v, du, dv, phi, ksi, tiles, (int)w, (int)h, (int)(pitch_bytes / 4), hx, hy, alpha, omega, colour);
        } else if (constancy == FLOW2D_CONSTANCY_GRADIENT_UNTILED) {
            const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kBlockY));
        const dim3 grid(xcd_grid(tiles));
            sor_grad_untiled_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
                f0, f1, u, v, du, dv, phi, ksi, tiles, (int)w, (int)h, (int)(pitch_bytes / 4), hx, hy, alpha, omega, colour);
        } else {
            const XcdTiles tiles = xcd_tiles(div_up(w, kBlockX), div_up(h, kBlockY));
        const dim3 grid(xcd_grid(tiles));
            sor_grey_kernel<<<grid, dim3(kBlockX, kBlockY), 0, ctx->stream>>>(
                f0, f1, u, v, du, dv, phi, ksi, tiles, (int)w, (int)h, (int)(pitch_bytes / 4), hx, hy, alpha, omega, colour);
        }
        FLOW2D_CHECK_LAUNCH();
    }
    return FLOW2D_OK;
}

}  // namespace flow2d

extern "C" {

int flow2d_compute_phi_ksi(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                           const float* flow_v, const float* flow_du, const float* flow_dv, size_t width,
                           size_t height, size_t pitch_bytes, float hx, float hy, float equation_smoothness,
                           float equation_data, float* phi, float* ksi)
{
    FLOW2D_ENTER(ctx);
    const float* planes[] = {frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi, ksi};
    if (!solver_planes_ok(planes, 8, width, height, pitch_bytes) || !(hx > 0.f) || !(hy > 0.f) || phi == ksi)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 6; ++i)
        if (planes[i] == phi || planes[i] == ksi) return FLOW2D_ERR_INVALID_ARGUMENT;
    return flow2d::launch_phi_ksi(ctx, frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, width, height, pitch_bytes,
                                  hx, hy, equation_smoothness, equation_data, phi, ksi);
}

static int sweep_entry(flow2d_context* ctx, int constancy, const float* frame_0, const float* frame_1,
                       const float* flow_u, const float* flow_v, const float* flow_du, const float* flow_dv,
                       const float* phi, const float* ksi, size_t width, size_t height, size_t pitch_bytes, float hx,
                       float hy, float alpha, float* temp_du, float* temp_dv)
{
    FLOW2D_ENTER(ctx);
    const float* planes[] = {frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi, ksi, temp_du, temp_dv};
    if (!solver_planes_ok(planes, 10, width, height, pitch_bytes) || !(hx > 0.f) || !(hy > 0.f) ||
        temp_du == temp_dv)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 8; ++i)  // Jacobi: the sweep must not write a plane it reads
        if (planes[i] == temp_du || planes[i] == temp_dv) return FLOW2D_ERR_INVALID_ARGUMENT;
    return flow2d::launch_sweep(ctx, constancy, frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi, ksi, width,
                                height, pitch_bytes, hx, hy, alpha, temp_du, temp_dv);
}

int flow2d_solve_2d_sor(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                        const float* flow_v, float* flow_du, float* flow_dv, const float* phi, const float* ksi,
                        size_t width, size_t height, size_t pitch_bytes, float hx, float hy, float equation_alpha,
                        float omega, int data_constancy)
{
    FLOW2D_ENTER(ctx);
    const float* planes[] = {frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi, ksi};
    if (!solver_planes_ok(planes, 8, width, height, pitch_bytes) || !(hx > 0.f) || !(hy > 0.f) || flow_du == flow_dv ||
        !(omega > 0.f) || !(omega < 2.f))
        return FLOW2D_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 8; ++i)
        if (i != 4 && i != 5 && (planes[i] == flow_du || planes[i] == flow_dv)) return FLOW2D_ERR_INVALID_ARGUMENT;
    if (data_constancy != FLOW2D_CONSTANCY_GREY && data_constancy != FLOW2D_CONSTANCY_GRADIENT &&
        data_constancy != FLOW2D_CONSTANCY_GRADIENT_UNTILED)
        return FLOW2D_ERR_UNSUPPORTED;
    return flow2d::launch_sor_iteration(ctx, data_constancy, frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi,
                                        ksi, width, height, pitch_bytes, hx, hy, equation_alpha, omega);
}

int flow2d_solve_2d(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                    const float* flow_v, const float* flow_du, const float* flow_dv, const float* phi,
                    const float* ksi, size_t width, size_t height, size_t pitch_bytes, float hx, float hy,
                    float equation_alpha, float* temp_du, float* temp_dv)
{
    return sweep_entry(ctx, FLOW2D_CONSTANCY_GREY, frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi, ksi,
                       width, height, pitch_bytes, hx, hy, equation_alpha, temp_du, temp_dv);
}

int flow2d_solve_2d_grad(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                         const float* flow_v, const float* flow_du, const float* flow_dv, const float* phi,
                         const float* ksi, size_t width, size_t height, size_t pitch_bytes, float hx, float hy,
                         float equation_alpha, float* temp_du, float* temp_dv)
{
    return sweep_entry(ctx, FLOW2D_CONSTANCY_GRADIENT, frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi, ksi,
                       width, height, pitch_bytes, hx, hy, equation_alpha, temp_du, temp_dv);
}

int flow2d_solve_2d_log(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                        const float* flow_v, const float* flow_du, const float* flow_dv, const float* phi,
                        const float* ksi, size_t width, size_t height, size_t pitch_bytes, float hx, float hy,
                        float equation_alpha, float* temp_du, float* temp_dv)
{
    return sweep_entry(ctx, FLOW2D_CONSTANCY_LOG_DERIVATIVES, frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi,
                       ksi, width, height, pitch_bytes, hx, hy, equation_alpha, temp_du, temp_dv);
}

int flow2d_solve_2d_grad_untiled(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                                 const float* flow_v, const float* flow_du, const float* flow_dv, const float* phi,
                                 const float* ksi, size_t width, size_t height, size_t pitch_bytes, float hx, float hy,
                                 float equation_alpha, float* temp_du, float* temp_dv)
{
    return sweep_entry(ctx, FLOW2D_CONSTANCY_GRADIENT_UNTILED, frame_0, frame_1, flow_u, flow_v, flow_du, flow_dv, phi,
                       ksi, width, height, pitch_bytes, hx, hy, equation_alpha, temp_du, temp_dv);
}

}  // extern "C"
