/*
 * flow2d_oracle.c -- CPU restatement of the coarse-to-fine variational optical-flow path
 * of axruff/cuda-flow2d, in plain C.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call it.  The shipped path (cuda-flow2d_amd/) never links,
 * loads or falls back to anything in oracle/.
 *
 * PARITY STATUS: pinned against the reference itself, two ways (the reference ships no tests or vectors of its own):
 *   (1) its device kernels: src/kernels/{add,convolution,median,registration,resample,solve}_2d.cu are compiled
 *       for gfx950 from where they lie (oracle/Makefile -> oracle/_ref/<op>_2d.co, no FMA contraction) and run on an
 *       MI355X through oracle/ref_driver.cpp (symbol names, launch geometry and argument order of the reference's
 *       operator layer).  Their outputs on seeded inputs -- every kernel, CudaOperationSolve2D::Execute, and whole
 *       ComputeFlow runs on rub1/rub2 (settings.xml values and main.cpp defaults) and synthetic pairs -- are the
 *       fixture tests/golden/ref_kernels_golden.npz (tests/golden/make_ref_golden.py); this file reproduces them
 *       bit for bit (tests/test_oracle.py::test_ref_golden_*), NaN / signed-zero median windows included.
 *       Two documented exceptions: solve_2d_grad / solve_2d_log where the image edge falls inside a 16x8 block
 *       (the reference reads unwritten shared memory there; see oracle_solve_2d_grad), and solve_2d_log's logf
 *       (device library there, libm here: last-place differences).
 *   (2) its host code: GetMaxWarpLevel, ComputeGaussianKernel (and, for the product's host layer, Data2D IO,
 *       the colour-wheel writers, Settings, OperationParameters) are compiled from where they lie into
 *       oracle/_ref/ref_host_probe and run in the build container; outputs in tests/golden/ref_host_golden.npz.
 *   Besides: an independent numpy restatement (oracle/np_restatement.py) that agrees bit for bit, and the
 *   anchors SURVEY.md 8(c) recorded.  The reference's host orchestration (optical_flow_2d.cpp, the operator
 *   classes) needs libcuda and is not built; ref_driver.cpp restates it around the real kernels.
 *
 * Every function follows one reference function; the file:line it restates is cited above it
 * (paths relative to the reference repository root).  Arithmetic is fp32 with the reference's
 * operation order and promotions; build with -ffp-contract=off (see oracle/Makefile) so no
 * multiply-add is fused -- the same rule the HIP kernels are built under.
 *
 * Planes are row-major float arrays addressed as p[y * pitch + x], pitch in ELEMENTS
 * (the reference's IND(X,Y) with container_size.pitch / sizeof(float)).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* Reflect-without-repeat index used by the solver and median halos:
 * -k -> k, n-1+k -> n-1-k   (solve_2d.cu:75-76,88-89,101-102; median_2d.cu:107-108,115-116,123-124) */
static inline long mirror_index(long i, long n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - i - 2;
    return i;
}

/* ------------------------------------------------------------------------------------------
 * Thread control for the timing baseline (bench.py cpu_baseline).
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Level-count rule.  src/optical_flow/optical_flow_base_2d.cpp:36-59 (GetMaxWarpLevel)
 * ---------------------------------------------------------------------------------------- */
ORACLE_API size_t oracle_max_warp_level(size_t width, size_t height, float scale_factor)
{
    size_t rw = 1, rh = 1, level = 1;
    while (scale_factor < 1.f) {
        float s = powf(scale_factor, (float)level);
        rw = (size_t)ceilf((float)width * s);
        rh = (size_t)ceilf((float)height * s);
        if (rw < 4 || rh < 4) break;
        ++level;
    }
    if (rw == 1 || rh == 1) --level;
    return level;
}

/* Size and grid spacing of pyramid level `level`.  src/optical_flow/optical_flow_2d.cpp:268-272 */
ORACLE_API void oracle_level_geometry(size_t width, size_t height, float scale_factor, int level,
                                      size_t* lw, size_t* lh, float* hx, float* hy)
{
    float s = powf(scale_factor, (float)level);
    *lw = (size_t)ceilf((float)width * s);
    *lh = (size_t)ceilf((float)height * s);
    *hx = (float)width / (float)(*lw);
    *hy = (float)height / (float)(*lh);
}

/* ------------------------------------------------------------------------------------------
 * Gaussian taps.  src/cuda_operations/2d/cuda_operation_convolution_2d.cpp:83-112
 * (ComputeGaussianKernel with precision = 3, pixel_size = 1.0, as called at :160)
 * taps must hold >= 2*radius+1 floats (the reference caps the length at 51).
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int oracle_gaussian_taps(float sigma, float* taps, int* radius_out)
{
    const size_t precision = 3;
    const float pixel_size = 1.0f;
    size_t radius = (size_t)((float)precision * sigma / pixel_size);
    int r = (int)radius;
    if (2 * r + 1 > 51) return 1;
    for (int i = -r; i <= r; ++i) {
        float num = -((float)(i * i) * pixel_size * pixel_size);           /* float */
        double arg = (double)num / (2.0 * (double)sigma * (double)sigma);   /* double */
        double amp = 1.0 / ((double)sigma * sqrt(2.0 * 3.1415926));
        taps[i + r] = (float)(amp * exp(arg));
    }
    float sum = 0.0f;
    for (int i = 0; i < 2 * r + 1; ++i) sum = sum + taps[i];
    for (int i = 0; i < 2 * r + 1; ++i) taps[i] = taps[i] / sum;
    *radius_out = r;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Separable Gaussian, zero padding outside the image, accumulation j = -r..r with tap[r-j].
 * src/kernels/convolution_2d.cu:74-168 (rows), :181-261 (columns)
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_convolution_rows(float* dst, const float* src, int w, int h, int pitch,
                                        const float* taps, int radius)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            float sum = 0;
            for (int j = -radius; j <= radius; ++j) {
                int xx = x + j;
                float s = (xx >= 0 && xx < w) ? src[(size_t)y * pitch + xx] : 0.f;
                sum += taps[radius - j] * s;
            }
            dst[(size_t)y * pitch + x] = sum;
        }
    }
}

ORACLE_API void oracle_convolution_cols(float* dst, const float* src, int w, int h, int pitch,
                                        const float* taps, int radius)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            float sum = 0;
            for (int j = -radius; j <= radius; ++j) {
                int yy = y + j;
                float s = (yy >= 0 && yy < h) ? src[(size_t)yy * pitch + x] : 0.f;
                sum += taps[radius - j] * s;
            }
            dst[(size_t)y * pitch + x] = sum;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Area-weighted 1-D resampling.  src/kernels/resample_2d.cu:34-75 (x), :77-118 (y)
 * ---------------------------------------------------------------------------------------- */
static inline float resample_cell_sum(const float* in, size_t stride, size_t g, size_t out_n, size_t in_n)
{
    float delta = (float)in_n / (float)out_n;
    float norm = (float)out_n / (float)in_n;
    float left_f = (float)(unsigned)g * delta;
    float right_f = (float)((unsigned)g + 1u) * delta;
    int left_i = (int)floorf(left_f);
    int right_c = (int)ceilf(right_f);
    int right_i = (int)in_n < right_c ? (int)in_n : right_c;
    int cells = right_i - left_i;
    float value = 0.f;
    for (int j = 0; j < cells; ++j) {
        float frac = 1.f;
        if (j == 0) frac = (float)(left_i + 1) - left_f;
        if (j == cells - 1) frac = right_f - (float)(left_i + j);
        if (cells == 1) frac = delta;
        value += in[(size_t)(left_i + j) * stride] * frac;
    }
    return value * norm;
}

ORACLE_API void oracle_resample_x(const float* in, float* out, size_t out_w, size_t out_h,
                                  size_t in_w, size_t pitch)
{
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)out_h; ++y)
        for (size_t x = 0; x < out_w; ++x)
            out[(size_t)y * pitch + x] = resample_cell_sum(in + (size_t)y * pitch, 1, x, out_w, in_w);
}

ORACLE_API void oracle_resample_y(const float* in, float* out, size_t out_w, size_t out_h,
                                  size_t in_h, size_t pitch)
{
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)out_h; ++y)
        for (size_t x = 0; x < out_w; ++x)
            out[(size_t)y * pitch + x] = resample_cell_sum(in + x, pitch, (size_t)y, out_h, in_h);
}

/* x pass into temp (out_w x in_h), then y pass.
 * src/cuda_operations/2d/cuda_operation_resample_2d.cpp:99-105 */
ORACLE_API void oracle_resample(const float* in, float* out, float* temp, size_t in_w, size_t in_h,
                                size_t out_w, size_t out_h, size_t pitch)
{
    oracle_resample_x(in, temp, out_w, in_h, in_w, pitch);
    oracle_resample_y(temp, out, out_w, out_h, in_h, pitch);
}

/* ------------------------------------------------------------------------------------------
 * Backward bilinear registration (warp).  src/kernels/registration_2d.cu:34-73
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_registration_2d(const float* f0, const float* f1, const float* u, const float* v,
                                       size_t w, size_t h, size_t pitch, float hx, float hy, float* out)
{
    const float inv_hx = 1.f / hx, inv_hy = 1.f / hy;
    const float xmax = (float)(w - 1), ymax = (float)(h - 1);
#pragma omp parallel for schedule(static)
    for (long yy = 0; yy < (long)h; ++yy) {
        for (size_t xx = 0; xx < w; ++xx) {
            size_t c = (size_t)yy * pitch + xx;
            float x_f = (float)(unsigned)xx + (u[c] * inv_hx);
            float y_f = (float)(unsigned)yy + (v[c] * inv_hy);
            if ((x_f < 0.) || (x_f > xmax) || (y_f < 0.) || (y_f > ymax) || isnan(x_f) || isnan(y_f)) {
                out[c] = f0[c];
            } else {
                int x = (int)floorf(x_f), y = (int)floorf(y_f);
                float dx = x_f - (float)x, dy = y_f - (float)y;
                int x1 = (int)(w - 1) < x + 1 ? (int)(w - 1) : x + 1;
                int y1 = (int)(h - 1) < y + 1 ? (int)(h - 1) : y + 1;
                float value = (1.f - dx) * (1.f - dy) * f1[(size_t)y * pitch + x] +
                              (dx) * (1.f - dy) * f1[(size_t)y * pitch + x1] +
                              (1.f - dx) * (dy)*f1[(size_t)y1 * pitch + x] +
                              (dx) * (dy)*f1[(size_t)y1 * pitch + x1];
                out[c] = value;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Robust weights: smoothness diffusivity phi and data penaliser ksi.
 * src/kernels/solve_2d.cu:43-198 (compute_phi_ksi)
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_compute_phi_ksi(const float* f0, const float* f1, const float* u, const float* v,
                                       const float* du, const float* dv, size_t w, size_t h, size_t pitch,
                                       float hx, float hy, float e_smooth, float e_data, float* phi, float* ksi)
{
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) {
        size_t rc = (size_t)y * pitch;
        size_t ru = (size_t)mirror_index(y - 1, (long)h) * pitch;
        size_t rd = (size_t)mirror_index(y + 1, (long)h) * pitch;
        for (long x = 0; x < (long)w; ++x) {
            size_t xl = (size_t)mirror_index(x - 1, (long)w), xr = (size_t)mirror_index(x + 1, (long)w);
            float dux = (u[rc + xr] - u[rc + xl] + du[rc + xr] - du[rc + xl]) / (2.f * hx);
            float duy = (u[rd + x] - u[ru + x] + du[rd + x] - du[ru + x]) / (2.f * hy);
            float dvx = (v[rc + xr] - v[rc + xl] + dv[rc + xr] - dv[rc + xl]) / (2.f * hx);
            float dvy = (v[rd + x] - v[ru + x] + dv[rd + x] - dv[ru + x]) / (2.f * hy);
            phi[rc + x] = 1.f / (2.f * sqrtf(dux * dux + duy * duy + dvx * dvx + dvy * dvy + e_smooth * e_smooth));

            float fx = (f0[rc + xr] - f0[rc + xl] + f1[rc + xr] - f1[rc + xl]) / (4.f * hx);
            float fy = (f0[rd + x] - f0[ru + x] + f1[rd + x] - f1[ru + x]) / (4.f * hy);
            float ft = f1[rc + x] - f0[rc + x];
            float J11 = fx * fx, J22 = fy * fy, J33 = ft * ft, J12 = fx * fy, J13 = fx * ft, J23 = fy * ft;
            float a = du[rc + x], b = dv[rc + x];
            float s = (J11 * a + J12 * b + J13) * a + (J12 * a + J22 * b + J23) * b + (J13 * a + J23 * b + J33);
            s = (float)(s > 0) * s;
            ksi[rc + x] = 1.f / (2.f * sqrtf(s + e_data * e_data));
        }
    }
}

/* One Jacobi sweep given the motion tensor entries of a pixel; the part shared by
 * solve_2d (solve_2d.cu:332-374) and solve_2d_grad (solve_2d.cu:889-931). */
static inline void jacobi_update(const float* u, const float* v, const float* du, const float* dv, const float* phi,
                                 const float* ksi, size_t rc, size_t ru, size_t rd, long x, size_t xl, size_t xr,
                                 long y, long w, long h, float hx_2, float hy_2, float J11, float J22, float J12,
                                 float J13, float J23, float* out_du, float* out_dv)
{
    float xp = (float)(x < w - 1) * hx_2;
    float xm = (float)(x > 0) * hx_2;
    float yp = (float)(y < h - 1) * hy_2;
    float ym = (float)(y > 0) * hy_2;
    float pc = phi[rc + x];
    float phi_xp = (phi[rc + xr] + pc) / 2.f;
    float phi_xm = (phi[rc + xl] + pc) / 2.f;
    float phi_yp = (phi[rd + x] + pc) / 2.f;
    float phi_ym = (phi[ru + x] + pc) / 2.f;
    float sumH = (xp * phi_xp + xm * phi_xm + yp * phi_yp + ym * phi_ym);
    float uc = u[rc + x], vc = v[rc + x];
    float sumU = phi_xp * xp * (u[rc + xr] + du[rc + xr] - uc) + phi_xm * xm * (u[rc + xl] + du[rc + xl] - uc) +
                 phi_yp * yp * (u[rd + x] + du[rd + x] - uc) + phi_ym * ym * (u[ru + x] + du[ru + x] - uc);
    float sumV = phi_xp * xp * (v[rc + xr] + dv[rc + xr] - vc) + phi_xm * xm * (v[rc + xl] + dv[rc + xl] - vc) +
                 phi_yp * yp * (v[rd + x] + dv[rd + x] - vc) + phi_ym * ym * (v[ru + x] + dv[ru + x] - vc);
    float k = ksi[rc + x];
    float r_du = (k * (-J13 - J12 * dv[rc + x]) + sumU) / (k * J11 + sumH);
    float r_dv = (k * (-J23 - J12 * r_du) + sumV) / (k * J22 + sumH);
    out_du[rc + x] = r_du;
    out_dv[rc + x] = r_dv;
}

/* ------------------------------------------------------------------------------------------
 * One Jacobi sweep, brightness constancy.  src/kernels/solve_2d.cu:200-377 (solve_2d)
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_solve_2d(const float* f0, const float* f1, const float* u, const float* v, const float* du,
                                const float* dv, const float* phi, const float* ksi, size_t w, size_t h,
                                size_t pitch, float hx, float hy, float alpha, float* tdu, float* tdv)
{
    const float hx_2 = alpha / (hx * hx), hy_2 = alpha / (hy * hy);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) {
        size_t rc = (size_t)y * pitch;
        size_t ru = (size_t)mirror_index(y - 1, (long)h) * pitch;
        size_t rd = (size_t)mirror_index(y + 1, (long)h) * pitch;
        for (long x = 0; x < (long)w; ++x) {
            size_t xl = (size_t)mirror_index(x - 1, (long)w), xr = (size_t)mirror_index(x + 1, (long)w);
            float fx = (f0[rc + xr] - f0[rc + xl] + f1[rc + xr] - f1[rc + xl]) / (4.f * hx);
            float fy = (f0[rd + x] - f0[ru + x] + f1[rd + x] - f1[ru + x]) / (4.f * hy);
            float ft = f1[rc + x] - f0[rc + x];
            jacobi_update(u, v, du, dv, phi, ksi, rc, ru, rd, x, xl, xr, y, (long)w, (long)h, hx_2, hy_2, fx * fx,
                          fy * fy, fx * fy, fx * ft, fy * ft, tdu, tdv);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * One Jacobi sweep, gradient constancy.  src/kernels/solve_2d.cu:683-952 (solve_2d_grad)
 *
 * The reference evaluates fx, fy, ft per pixel and then differentiates them INSIDE each 16x8
 * thread block, with the block's own edge value replicated into the halo (:816-841).  The
 * second derivatives therefore depend on the launch tiling (block 16x8 anchored at the image
 * origin, cuda_operation_solve_2d.cpp:164,234-235); this restatement reproduces that rule.
 * Where the image edge falls inside a block the reference reads a shared-memory slot no
 * thread wrote (undefined); here that slot is defined as the edge pixel's own value
 * (replication), the same rule as at a block edge.
 * ---------------------------------------------------------------------------------------- */
#define GRAD_TILE_X 16
#define GRAD_TILE_Y 8
/* Neighbour indices for the second derivatives.  tiled != 0: the reference's rule (own value replicated at the
 * 16x8 block edge and at the image edge).  tiled == 0 ("untiled", SURVEY 8 f2, NOT a reference mode): the true
 * neighbours everywhere, with the reflect rule of the first derivatives (-1 -> 1, n -> n-2) at the image border. */
static void grad_neighbours(long i, long n, long tile, int tiled, long* lo, long* hi)
{
    if (tiled) {
        *lo = (i % tile == 0) ? i : i - 1;
        *hi = (i % tile == tile - 1 || i == n - 1) ? i : i + 1;
    } else {
        *lo = mirror_index(i - 1, n);
        *hi = mirror_index(i + 1, n);
    }
}

/* Neighbour indices of the EIGHT BASE PLANES in solve_2d_log (solve_2d.cu:448,462,476,490): the halo offsets
 * there are `global_x - 1 + 1`, `global_x + 1 - 1`, ... = 0, so every 16x8 block's halo holds the block's own edge
 * pixel; inside a block the neighbour is the true one, and a block that sticks out of the image has its
 * out-of-range threads load the reflected pixel (:434-435), which is what the last in-range pixel then reads. */
static void log_neighbours(long i, long n, long tile, long* lo, long* hi)
{
    *lo = (i % tile == 0) ? i : i - 1;
    *hi = (i % tile == tile - 1) ? i : mirror_index(i + 1, n);
}

/* mode 0: Gradient over true neighbours (not a reference mode); 1: solve_2d_grad (:683-952);
 * 2: solve_2d_log (:391-669) -- as 1 on log(I + 1.0f), with the block-edge replication applied to the frames,
 * u, v, du, dv, phi and ksi as well.  lg0 / lg1, when non-NULL, are precomputed log(I + 1) planes of frame_0 /
 * frame_1 (same pitch); NULL -> logf from libm, which is within an ulp of, but not bit-identical to, a GPU's. */
static void solve_2d_grad_any(const float* f0, const float* f1, const float* u, const float* v, const float* du,
                              const float* dv, const float* phi, const float* ksi, size_t w, size_t h, size_t pitch,
                              float hx, float hy, float alpha, float* tdu, float* tdv, int mode, const float* lg0,
                              const float* lg1)
{
    const float hx_2 = alpha / (hx * hx), hy_2 = alpha / (hy * hy);
    const float hx_1 = (float)(1.0 / (2.0 * (double)hx));
    const float hy_1 = (float)(1.0 / (2.0 * (double)hy));
    const int tiled = mode != 0, logm = mode == 2;
    float* fxp = (float*)malloc(sizeof(float) * w * h);
    float* fyp = (float*)malloc(sizeof(float) * w * h);
    float* ftp = (float*)malloc(sizeof(float) * w * h);
    float* l0 = NULL;
    float* l1 = NULL;
    if (logm && !(lg0 && lg1)) {
        l0 = (float*)malloc(sizeof(float) * pitch * h);
        l1 = (float*)malloc(sizeof(float) * pitch * h);
#pragma omp parallel for schedule(static)
        for (long y = 0; y < (long)h; ++y)
            for (long x = 0; x < (long)w; ++x) {
                l0[y * pitch + x] = logf(f0[y * pitch + x] + 1.0f);
                l1[y * pitch + x] = logf(f1[y * pitch + x] + 1.0f);
            }
        lg0 = l0;
        lg1 = l1;
    }
    const float* g0 = logm ? lg0 : f0;
    const float* g1 = logm ? lg1 : f1;
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) {
        long yu = mirror_index(y - 1, (long)h), yd = mirror_index(y + 1, (long)h);
        if (logm) log_neighbours(y, (long)h, GRAD_TILE_Y, &yu, &yd);
        size_t rc = (size_t)y * pitch, ru = (size_t)yu * pitch, rd = (size_t)yd * pitch;
        for (long x = 0; x < (long)w; ++x) {
            long xl = mirror_index(x - 1, (long)w), xr = mirror_index(x + 1, (long)w);
            if (logm) log_neighbours(x, (long)w, GRAD_TILE_X, &xl, &xr);
            fxp[y * w + x] = (g0[rc + xr] - g0[rc + xl] + g1[rc + xr] - g1[rc + xl]) / (4.f * hx);
            fyp[y * w + x] = (g0[rd + x] - g0[ru + x] + g1[rd + x] - g1[ru + x]) / (4.f * hy);
            ftp[y * w + x] = g1[rc + x] - g0[rc + x];
        }
    }
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) {
        long yn = mirror_index(y - 1, (long)h), ys = mirror_index(y + 1, (long)h);
        if (logm) log_neighbours(y, (long)h, GRAD_TILE_Y, &yn, &ys);
        size_t rc = (size_t)y * pitch, ru = (size_t)yn * pitch, rd = (size_t)ys * pitch;
        long yu, yd;
        grad_neighbours(y, (long)h, GRAD_TILE_Y, tiled, &yu, &yd);
        for (long x = 0; x < (long)w; ++x) {
            long xl = mirror_index(x - 1, (long)w), xr = mirror_index(x + 1, (long)w);
            if (logm) log_neighbours(x, (long)w, GRAD_TILE_X, &xl, &xr);
            long xa, xb;
            grad_neighbours(x, (long)w, GRAD_TILE_X, tiled, &xa, &xb);
            float fxx = (fxp[y * w + xb] - fxp[y * w + xa]) * hx_1;
            float fxy = (fxp[yd * w + x] - fxp[yu * w + x]) * hy_1;
            float fyy = (fyp[yd * w + x] - fyp[yu * w + x]) * hy_1;
            float fxt = (ftp[y * w + xb] - ftp[y * w + xa]) * hx_1;
            float fyt = (ftp[yd * w + x] - ftp[yu * w + x]) * hy_1;
            float J11 = fxx * fxx + fxy * fxy;
            float J22 = fxy * fxy + fyy * fyy;
            float J12 = fxx * fxy + fxy * fyy;
            float J13 = fxx * fxt + fxy * fyt;
            float J23 = fxy * fxt + fyy * fyt;
            jacobi_update(u, v, du, dv, phi, ksi, rc, ru, rd, x, (size_t)xl, (size_t)xr, y, (long)w, (long)h, hx_2,
                          hy_2, J11, J22, J12, J13, J23, tdu, tdv);
        }
    }
    free(fxp);
    free(fyp);
    free(ftp);
    free(l0);
    free(l1);
}

ORACLE_API void oracle_solve_2d_grad(const float* f0, const float* f1, const float* u, const float* v,
                                     const float* du, const float* dv, const float* phi, const float* ksi, size_t w,
                                     size_t h, size_t pitch, float hx, float hy, float alpha, float* tdu, float* tdv)
{
    solve_2d_grad_any(f0, f1, u, v, du, dv, phi, ksi, w, h, pitch, hx, hy, alpha, tdu, tdv, 1, NULL, NULL);
}

/* Gradient constancy with true neighbours (constancy 2 of the product; not a reference kernel). */
ORACLE_API void oracle_solve_2d_grad_untiled(const float* f0, const float* f1, const float* u, const float* v,
                                             const float* du, const float* dv, const float* phi, const float* ksi,
                                             size_t w, size_t h, size_t pitch, float hx, float hy, float alpha,
                                             float* tdu, float* tdv)
{
    solve_2d_grad_any(f0, f1, u, v, du, dv, phi, ksi, w, h, pitch, hx, hy, alpha, tdu, tdv, 0, NULL, NULL);
}

/* One Jacobi sweep on the logarithmic derivatives.  src/kernels/solve_2d.cu:391-669 (solve_2d_log), selected by
 * DataConstancy::LogDerivatives (cuda_operation_solve_2d.cpp:75-77).  Like solve_2d_grad its result depends on the
 * 16x8 launch tiling, and where the image edge falls inside a block the last pixel's fx/fy/ft neighbour is an
 * unwritten shared-memory slot in the reference (defined here as the pixel's own value). */
ORACLE_API void oracle_solve_2d_log(const float* f0, const float* f1, const float* u, const float* v,
                                    const float* du, const float* dv, const float* phi, const float* ksi, size_t w,
                                    size_t h, size_t pitch, float hx, float hy, float alpha, float* tdu, float* tdv)
{
    solve_2d_grad_any(f0, f1, u, v, du, dv, phi, ksi, w, h, pitch, hx, hy, alpha, tdu, tdv, 2, NULL, NULL);
}

/* The same with log(frame + 1.0f) supplied by the caller (to take the libm out of a comparison). */
ORACLE_API void oracle_solve_2d_log_planes(const float* lg0, const float* lg1, const float* u, const float* v,
                                           const float* du, const float* dv, const float* phi, const float* ksi,
                                           size_t w, size_t h, size_t pitch, float hx, float hy, float alpha,
                                           float* tdu, float* tdv)
{
    solve_2d_grad_any(lg0, lg1, u, v, du, dv, phi, ksi, w, h, pitch, hx, hy, alpha, tdu, tdv, 2, lg0, lg1);
}

/* ------------------------------------------------------------------------------------------
 * Opt-in red-black SOR iteration (NOT a reference kernel; the reference is Jacobi).  Restates
 * flow2d_solve_2d_sor of the product for the tests: pixels with (x + y) even, then odd, are relaxed in place;
 * du_new = (1 - w) du + w * gs_du, dv sees the relaxed du.  constancy: 0 Grey, 1 Gradient (reference tile rule),
 * 2 Gradient with true neighbours.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_solve_2d_sor(const float* f0, const float* f1, const float* u, const float* v, float* du,
                                    float* dv, const float* phi, const float* ksi, size_t w, size_t h, size_t pitch,
                                    float hx, float hy, float alpha, float omega, int constancy)
{
    const float hx_2 = alpha / (hx * hx), hy_2 = alpha / (hy * hy);
    const float hx_1 = (float)(1.0 / (2.0 * (double)hx));
    const float hy_1 = (float)(1.0 / (2.0 * (double)hy));
    float* fxp = (float*)malloc(sizeof(float) * w * h);
    float* fyp = (float*)malloc(sizeof(float) * w * h);
    float* ftp = (float*)malloc(sizeof(float) * w * h);
    for (long y = 0; y < (long)h; ++y) {
        size_t rc = (size_t)y * pitch;
        size_t ru = (size_t)mirror_index(y - 1, (long)h) * pitch;
        size_t rd = (size_t)mirror_index(y + 1, (long)h) * pitch;
        for (long x = 0; x < (long)w; ++x) {
            size_t xl = (size_t)mirror_index(x - 1, (long)w), xr = (size_t)mirror_index(x + 1, (long)w);
            fxp[y * w + x] = (f0[rc + xr] - f0[rc + xl] + f1[rc + xr] - f1[rc + xl]) / (4.f * hx);
            fyp[y * w + x] = (f0[rd + x] - f0[ru + x] + f1[rd + x] - f1[ru + x]) / (4.f * hy);
            ftp[y * w + x] = f1[rc + x] - f0[rc + x];
        }
    }
    for (int colour = 0; colour < 2; ++colour) {
#pragma omp parallel for schedule(static)
        for (long y = 0; y < (long)h; ++y) {
            size_t rc = (size_t)y * pitch;
            size_t ru = (size_t)mirror_index(y - 1, (long)h) * pitch;
            size_t rd = (size_t)mirror_index(y + 1, (long)h) * pitch;
            long yu, yd;
            grad_neighbours(y, (long)h, GRAD_TILE_Y, constancy == 1, &yu, &yd);
            for (long x = 0; x < (long)w; ++x) {
                if (((x + y) & 1) != colour) continue;
                size_t xl = (size_t)mirror_index(x - 1, (long)w), xr = (size_t)mirror_index(x + 1, (long)w);
                float J11, J22, J12, J13, J23;
                if (constancy == 1 || constancy == 2) {
                    long xa, xb;
                    grad_neighbours(x, (long)w, GRAD_TILE_X, constancy == 1, &xa, &xb);
                    float fxx = (fxp[y * w + xb] - fxp[y * w + xa]) * hx_1;
                    float fxy = (fxp[yd * w + x] - fxp[yu * w + x]) * hy_1;
                    float fyy = (fyp[yd * w + x] - fyp[yu * w + x]) * hy_1;
                    float fxt = (ftp[y * w + xb] - ftp[y * w + xa]) * hx_1;
                    float fyt = (ftp[yd * w + x] - ftp[yu * w + x]) * hy_1;
                    J11 = fxx * fxx + fxy * fxy;
                    J22 = fxy * fxy + fyy * fyy;
                    J12 = fxx * fxy + fxy * fyy;
                    J13 = fxx * fxt + fxy * fyt;
                    J23 = fxy * fxt + fyy * fyt;
                } else {
                    float fx = fxp[y * w + x], fy = fyp[y * w + x], ft = ftp[y * w + x];
                    J11 = fx * fx; J22 = fy * fy; J12 = fx * fy; J13 = fx * ft; J23 = fy * ft;
                }
                float xp = (float)(x < (long)w - 1) * hx_2, xm = (float)(x > 0) * hx_2;
                float yp = (float)(y < (long)h - 1) * hy_2, ym = (float)(y > 0) * hy_2;
                float pc = phi[rc + x];
                float wxp = (phi[rc + xr] + pc) / 2.f * xp, wxm = (phi[rc + xl] + pc) / 2.f * xm;
                float wyp = (phi[rd + x] + pc) / 2.f * yp, wym = (phi[ru + x] + pc) / 2.f * ym;
                float sumH = (wxp + wxm + wyp + wym);
                float uc = u[rc + x], vc = v[rc + x];
                float sumU = wxp * (u[rc + xr] + du[rc + xr] - uc) + wxm * (u[rc + xl] + du[rc + xl] - uc) +
                             wyp * (u[rd + x] + du[rd + x] - uc) + wym * (u[ru + x] + du[ru + x] - uc);
                float sumV = wxp * (v[rc + xr] + dv[rc + xr] - vc) + wxm * (v[rc + xl] + dv[rc + xl] - vc) +
                             wyp * (v[rd + x] + dv[rd + x] - vc) + wym * (v[ru + x] + dv[ru + x] - vc);
                float k = ksi[rc + x];
                float gs_du = (k * (-J13 - J12 * dv[rc + x]) + sumU) / (k * J11 + sumH);
                float du_new = (1.f - omega) * du[rc + x] + omega * gs_du;
                float gs_dv = (k * (-J23 - J12 * du_new) + sumV) / (k * J22 + sumH);
                float dv_new = (1.f - omega) * dv[rc + x] + omega * gs_dv;
                du[rc + x] = du_new;
                dv[rc + x] = dv_new;
            }
        }
    }
    free(fxp);
    free(fyp);
    free(ftp);
}

/* Level loop with SOR iterations in place of the Jacobi sweeps (same outer structure as oracle_solve_level). */
ORACLE_API void oracle_solve_level_sor(const float* f0, const float* f1, const float* u, const float* v, float* du,
                                       float* dv, float* phi, float* ksi, size_t w, size_t h, size_t pitch,
                                       size_t container_h, float hx, float hy, float alpha, float e_smooth,
                                       float e_data, size_t outer, size_t inner, int constancy, float omega)
{
    for (size_t y = 0; y < container_h; ++y) {
        memset(du + y * pitch, 0, w * sizeof(float));
        memset(dv + y * pitch, 0, w * sizeof(float));
    }
    for (size_t i = 0; i < outer; ++i) {
        oracle_compute_phi_ksi(f0, f1, u, v, du, dv, w, h, pitch, hx, hy, e_smooth, e_data, phi, ksi);
        for (size_t j = 0; j < inner; ++j)
            oracle_solve_2d_sor(f0, f1, u, v, du, dv, phi, ksi, w, h, pitch, hx, hy, alpha, omega, constancy);
    }
}

/* ------------------------------------------------------------------------------------------
 * op0 += op1.  src/kernels/add_2d.cu:33-46
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_add_2d(float* op0, const float* op1, size_t w, size_t h, size_t pitch)
{
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y)
        for (size_t x = 0; x < w; ++x) op0[(size_t)y * pitch + x] += op1[(size_t)y * pitch + x];
}

/* ------------------------------------------------------------------------------------------
 * r x r median ("radius" is the window WIDTH: 3, 5 or 7), mirror borders, full insertion sort,
 * element r*r/2.  src/kernels/median_2d.cu:87-299, sort at :52-63
 * Returns 0 on success, 1 for an unsupported width (the reference prints an error and writes
 * nothing, cuda_operation_median_2d.cpp:150-152).
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int oracle_median_2d(const float* in, size_t w, size_t h, size_t pitch, size_t radius, float* out)
{
    if (radius != 3 && radius != 5 && radius != 7) return 1;
    const int r2 = (int)radius / 2;
    const int n = (int)(radius * radius);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) {
        float win[49];
        for (long x = 0; x < (long)w; ++x) {
            int k = 0;
            /* same gather order as the reference: descending offsets (median_2d.cu:281-287) */
            for (int iy = 0; iy < (int)radius; ++iy) {
                long yy = mirror_index(y - iy + r2, (long)h);
                for (int ix = 0; ix < (int)radius; ++ix) {
                    long xx = mirror_index(x - ix + r2, (long)w);
                    win[k++] = in[(size_t)yy * pitch + (size_t)xx];
                }
            }
            for (int i = 0; i < n; ++i) {
                float t = win[i];
                int j = i - 1;
                for (; j >= 0 && t < win[j]; --j) win[j + 1] = win[j];
                win[j + 1] = t;
            }
            out[(size_t)y * pitch + x] = win[n / 2];
        }
    }
    return 0;
}

/* Median operator semantics incl. width 1 (copy) and even widths (width-1).
 * src/cuda_operations/2d/cuda_operation_median_2d.cpp:99-153
 * Returns 0 when `out` was written, 1 when the width is unsupported (out untouched). */
ORACLE_API int oracle_median_op(const float* in, size_t w, size_t h, size_t pitch, size_t container_h, size_t radius,
                                float* out)
{
    if (radius == 1) {
        memcpy(out, in, pitch * container_h * sizeof(float));
        return 0;
    }
    if (radius % 2 == 0) radius -= 1;
    if (radius >= 3 && radius <= 7) return oracle_median_2d(in, w, h, pitch, radius, out);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Fixed-point loops of one level.  src/cuda_operations/2d/cuda_operation_solve_2d.cpp:229-300
 * du, dv are zeroed (level width x container height), then outer x [phi/ksi, inner x sweep+swap].
 * The four pointers are swapped in place exactly like the reference (:288-289); on return
 * *du / *dv hold the result.  constancy: 0 = Grey (solve_2d), 1 = Gradient (solve_2d_grad), 2 = Gradient with
 * true neighbours (not a reference mode), 3 = LogDerivatives (solve_2d_log).
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_solve_level(const float* f0, const float* f1, const float* u, const float* v, float** du,
                                   float** dv, float* phi, float* ksi, float** tdu, float** tdv, size_t w, size_t h,
                                   size_t pitch, size_t container_h, float hx, float hy, float alpha, float e_smooth,
                                   float e_data, size_t outer, size_t inner, int constancy)
{
    for (size_t y = 0; y < container_h; ++y) {
        memset(*du + y * pitch, 0, w * sizeof(float));
        memset(*dv + y * pitch, 0, w * sizeof(float));
    }
    for (size_t i = 0; i < outer; ++i) {
        oracle_compute_phi_ksi(f0, f1, u, v, *du, *dv, w, h, pitch, hx, hy, e_smooth, e_data, phi, ksi);
        for (size_t j = 0; j < inner; ++j) {
            if (constancy == 1)
                oracle_solve_2d_grad(f0, f1, u, v, *du, *dv, phi, ksi, w, h, pitch, hx, hy, alpha, *tdu, *tdv);
            else if (constancy == 2)
                oracle_solve_2d_grad_untiled(f0, f1, u, v, *du, *dv, phi, ksi, w, h, pitch, hx, hy, alpha, *tdu, *tdv);
            else if (constancy == 3)
                oracle_solve_2d_log(f0, f1, u, v, *du, *dv, phi, ksi, w, h, pitch, hx, hy, alpha, *tdu, *tdv);
            else
                oracle_solve_2d(f0, f1, u, v, *du, *dv, phi, ksi, w, h, pitch, hx, hy, alpha, *tdu, *tdv);
            float* t = *du; *du = *tdu; *tdu = t;
            t = *dv; *dv = *tdv; *tdv = t;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Whole coarse-to-fine loop.  src/optical_flow/optical_flow_2d.cpp:142-569 (ComputeFlow)
 * frame_0/frame_1/flow_u/flow_v are tight W x H host images (Data2D layout, data2d.h:33-35).
 * Twelve W x H planes are used like the reference's container pool (optical_flow_2d.h:45);
 * pitch == W here.  Optional stage dump: when `dump` is non-NULL it is called after each
 * stage with a tag, the level and the plane (used to build per-stage golden vectors).
 * Returns 0, or 1 when no level would be run (levels == 0 or scale >= 1; the reference then
 * returns stale buffers, SURVEY H1 -- treated as an input error here).
 * ---------------------------------------------------------------------------------------- */
typedef void (*oracle_dump_fn)(const char* tag, int level, const float* plane, size_t w, size_t h, size_t pitch,
                               void* user);

typedef struct {
    size_t warp_levels_count;
    float warp_scale_factor;
    size_t outer_iterations_count;
    size_t inner_iterations_count;
    float equation_alpha;
    float equation_smoothness;
    float equation_data;
    size_t median_radius;
    float gaussian_sigma;
    int data_constancy; /* 0 Grey, 1 Gradient (reference tile rule), 2 Gradient with true neighbours, 3 Log */
    float sor_omega;    /* 0: Jacobi sweeps (reference).  (0,2): opt-in red-black SOR iterations instead */
} oracle_flow_params;

#define SWAPP(a, b)      \
    do {                 \
        float* t_ = (a); \
        (a) = (b);       \
        (b) = t_;        \
    } while (0)

ORACLE_API int oracle_compute_flow(const float* frame_0, const float* frame_1, float* flow_u, float* flow_v,
                                   size_t W, size_t H, const oracle_flow_params* p, oracle_dump_fn dump, void* user,
                                   double* finest_solve_seconds)
{
    const size_t pitch = W;
    const size_t plane = W * H;
    size_t max_level = oracle_max_warp_level(W, H, p->warp_scale_factor);
    size_t lv = p->warp_levels_count < max_level ? p->warp_levels_count : max_level;
    int level = (int)lv - 1;
    if (level < 0 || !(p->warp_scale_factor < 1.f)) return 1;

    float* pool = (float*)calloc(plane * 12, sizeof(float));
    if (!pool) return 2;
    float* f0 = pool + 0 * plane;
    float* f1 = pool + 1 * plane;
    float* f0r = pool + 2 * plane;
    float* f1r = pool + 3 * plane;
    float* u = pool + 4 * plane;
    float* v = pool + 5 * plane;
    float* du = pool + 6 * plane;
    float* dv = pool + 7 * plane;
    float* t0 = pool + 8 * plane;
    float* t1 = pool + 9 * plane;
    float* t2 = pool + 10 * plane;
    float* t3 = pool + 11 * plane;

    memcpy(f0, frame_0, plane * sizeof(float));
    memcpy(f1, frame_1, plane * sizeof(float));

    /* Gaussian pre-blur (optical_flow_2d.cpp:218-246) */
    if (p->gaussian_sigma > 0.0) {
        float taps[51];
        int radius = 0;
        if (oracle_gaussian_taps(p->gaussian_sigma, taps, &radius)) {
            free(pool);
            return 3;
        }
        oracle_convolution_rows(t0, f0, (int)W, (int)H, (int)pitch, taps, radius);
        oracle_convolution_cols(u, t0, (int)W, (int)H, (int)pitch, taps, radius);
        oracle_convolution_rows(t0, f1, (int)W, (int)H, (int)pitch, taps, radius);
        oracle_convolution_cols(v, t0, (int)W, (int)H, (int)pitch, taps, radius);
        SWAPP(f0, u);
        SWAPP(f1, v);
        if (dump) {
            dump("blur0", -1, f0, W, H, pitch, user);
            dump("blur1", -1, f1, W, H, pitch, user);
        }
    }

    size_t pw = 0, ph = 0;
    while (level >= 0) {
        size_t cw, ch;
        float hx, hy;
        oracle_level_geometry(W, H, p->warp_scale_factor, level, &cw, &ch, &hx, &hy);

        /* frames: full resolution -> level (optical_flow_2d.cpp:278-305) */
        if (level == 0) {
            SWAPP(f0, f0r);
            SWAPP(f1, f1r);
        } else {
            oracle_resample(f0, f0r, t0, W, H, cw, ch, pitch);
            oracle_resample(f1, f1r, t0, W, H, cw, ch, pitch);
        }
        if (dump) {
            dump("frame0_res", level, f0r, cw, ch, pitch, user);
            dump("frame1_res", level, f1r, cw, ch, pitch, user);
        }

        /* flow: previous level -> this level (optical_flow_2d.cpp:308-341) */
        if (pw == 0) {
            memset(u, 0, plane * sizeof(float));
            memset(v, 0, plane * sizeof(float));
        } else {
            oracle_resample(u, du, t0, pw, ph, cw, ch, pitch);
            oracle_resample(v, dv, t0, pw, ph, cw, ch, pitch);
            SWAPP(u, du);
            SWAPP(v, dv);
        }
        if (dump) {
            dump("flow_u_res", level, u, cw, ch, pitch, user);
            dump("flow_v_res", level, v, cw, ch, pitch, user);
        }

        /* warp frame 1 back by the current flow (optical_flow_2d.cpp:344-363) */
        oracle_registration_2d(f0r, f1r, u, v, cw, ch, pitch, hx, hy, t0);
        SWAPP(f1r, t0);
        if (dump) dump("warped", level, f1r, cw, ch, pitch, user);

        /* solve (optical_flow_2d.cpp:366-406) */
        {
            double t_start = 0.0;
#ifdef _OPENMP
            t_start = omp_get_wtime();
#endif
            if (p->sor_omega != 0.f)
                oracle_solve_level_sor(f0r, f1r, u, v, du, dv, t0, t1, cw, ch, pitch, H, hx, hy, p->equation_alpha,
                                       p->equation_smoothness, p->equation_data, p->outer_iterations_count,
                                       p->inner_iterations_count, p->data_constancy, p->sor_omega);
            else
                oracle_solve_level(f0r, f1r, u, v, &du, &dv, t0, t1, &t2, &t3, cw, ch, pitch, H, hx, hy,
                                   p->equation_alpha, p->equation_smoothness, p->equation_data,
                                   p->outer_iterations_count, p->inner_iterations_count, p->data_constancy);
#ifdef _OPENMP
            if (level == 0 && finest_solve_seconds) *finest_solve_seconds = omp_get_wtime() - t_start;
#endif
        }
        if (dump) {
            dump("phi", level, t0, cw, ch, pitch, user);
            dump("ksi", level, t1, cw, ch, pitch, user);
            dump("du", level, du, cw, ch, pitch, user);
            dump("dv", level, dv, cw, ch, pitch, user);
        }

        /* u += du, v += dv (optical_flow_2d.cpp:409-422) */
        oracle_add_2d(u, du, cw, ch, pitch);
        oracle_add_2d(v, dv, cw, ch, pitch);
        if (dump) {
            dump("flow_u_add", level, u, cw, ch, pitch, user);
            dump("flow_v_add", level, v, cw, ch, pitch, user);
        }

        pw = cw;
        ph = ch;
        --level;

        /* median of u and v after EVERY level (optical_flow_2d.cpp:428-449); the caller swaps
         * the output in even when the operator refused the width and wrote nothing. */
        oracle_median_op(u, cw, ch, pitch, H, p->median_radius, t0);
        SWAPP(u, t0);
        oracle_median_op(v, cw, ch, pitch, H, p->median_radius, t0);
        SWAPP(v, t0);
        if (dump) {
            dump("flow_u_med", level + 1, u, cw, ch, pitch, user);
            dump("flow_v_med", level + 1, v, cw, ch, pitch, user);
        }
    }

    memcpy(flow_u, u, plane * sizeof(float));
    memcpy(flow_v, v, plane * sizeof(float));
    free(pool);
    return 0;
}
