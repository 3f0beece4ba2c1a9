// Micro-benchmark (round 5): the HBM ceiling of the per-sweep solver kernels.  A Jacobi sweep of the
// reference (src/kernels/solve_2d.cu:200-377) reads eight fp32 planes and writes two; this kernel does exactly that and next
// to nothing else (the eight values are summed, the sum goes to both outputs), in the strip geometry of sweep_stream_kernel
// (csrc/solve.hip): a wave owns a strip of columns and walks down `rows` rows, four strips to a workgroup, workgroups dealt
// to the XCDs in horizontal bands.  Variants: with and without the sweep's halo (one lane per side, one row above and below),
// 4 or 16 bytes per lane, strip heights, all ten planes or a plain two-stream copy.  Reports us per pass and TB/s of
// ALGORITHMIC bytes (40 B per pixel for the ten-plane forms, 8 B for the copy) -- the figure SURVEY 8(d) prices a sweep with.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream10.hip -o build_ubench/stream10
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct Planes {
    const float* in[8];
    float* out[2];
};

__device__ __forceinline__ int mirror(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - i - 2 : i;
}

// VEC floats per lane; HALO: the sweep's halo lanes and rows (re-read from the neighbouring strips, as the real kernel does);
// NIN planes read, NOUT written
// (HALO as an int below: 0 none; 1 strips of 62 columns, one halo lane per side (rows start at 248-byte multiples); 2 strips
//  of 64 ALIGNED columns, the halo rows as before, the two halo columns of the planes that need x neighbours -- all but one --
//  by a second load instruction in which only lanes 0 and 63 are active)
template <int VEC, int HALO, int NIN, int NOUT>
__global__ __launch_bounds__(256) void stream_kernel(Planes p, int w, int h, int pitch, int rows, unsigned tiles_x, unsigned tiles_y,
                                                     unsigned rows_per_xcd)
{
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const unsigned tile_y = xcd * rows_per_xcd + j / tiles_x, tile_x = j % tiles_x;
    if (tile_y >= tiles_y) return;
    constexpr int kHalo = HALO == 1 ? 1 : 0, kValid = 64 - 2 * kHalo;
    constexpr int kRowHalo = HALO ? 1 : 0;
    const int lane = threadIdx.x & 63;
    const int strip = tile_x * 4 + (threadIdx.x >> 6);
    if (strip * kValid * VEC >= w) return;
    const int x = (strip * kValid - kHalo + lane) * VEC;
    const int xm = min(max(mirror(x, w), 0), w - VEC);
    const bool stores = lane >= kHalo && lane < 64 - kHalo && x < w;
    const int y0 = tile_y * rows, y1 = min(y0 + rows, h);
    typedef float vec __attribute__((ext_vector_type(VEC)));
    vec carry = 0.f;  // what the halo rows contribute, so that their loads cannot be dropped
    // HALO == 2: lane 0 fetches column x - 1, lane 63 column x + 1 (mirrored at the image border)
    const bool halo_lane = HALO == 2 && (lane == 0 || lane == 63);
    const int xh = min(max(mirror(lane == 0 ? x - 1 : x + 1, w), 0), w - 1);
    for (int y = y0 - kRowHalo; y < y1 + kRowHalo; ++y) {
        const size_t row = static_cast<size_t>(min(max(mirror(y, h), 0), h - 1)) * pitch;
        const size_t at = row + xm;
        vec s = carry;
#pragma unroll
        for (int i = 0; i < NIN; ++i) s += *reinterpret_cast<const vec*>(p.in[i] + at);
        if (HALO == 2) {
            float t = 0.f;
            if (halo_lane) {
#pragma unroll
                for (int i = 0; i + 1 < NIN; ++i) t += p.in[i][row + xh];
            }
            s[0] += t;
        }
        if (y < y0 || y >= y1) {
            carry = s * 1e-30f;
            continue;
        }
        if (stores) {
            const size_t to = static_cast<size_t>(y) * pitch + x;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) *reinterpret_cast<vec*>(p.out[i] + to) = s;
            if (NOUT == 0 && s[0] == 123.456f) *reinterpret_cast<vec*>(p.out[0] + to) = s;  // (never: keeps the loads)
        }
    }
}

struct Case {
    const char* name;
    int vec, halo, nin, nout, rows;
};

template <int VEC, int HALO, int NIN, int NOUT>
static float time_case(const Planes& p, int w, int h, int pitch, int rows, int reps)
{
    const int valid = (HALO == 1 ? 62 : 64) * VEC;
    const unsigned strips_x = (w + valid - 1) / valid, tiles_x = (strips_x + 3) / 4, tiles_y = (h + rows - 1) / rows;
    const unsigned rows_per_xcd = (tiles_y + 7) / 8;
    const dim3 grid(8 * rows_per_xcd * tiles_x);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) stream_kernel<VEC, HALO, NIN, NOUT><<<grid, 256>>>(p, w, h, pitch, rows, tiles_x, tiles_y, rows_per_xcd);
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) stream_kernel<VEC, HALO, NIN, NOUT><<<grid, 256>>>(p, w, h, pitch, rows, tiles_x, tiles_y, rows_per_xcd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main(int argc, char** argv)
{
    const int sizes[] = {4096, 8192};
    for (int size : sizes) {
        const int w = size, h = size, pitch = size;
        Planes p;
        const size_t bytes = static_cast<size_t>(pitch) * h * sizeof(float);
        for (int i = 0; i < 8; ++i) {
            (void)hipMalloc(reinterpret_cast<void**>(const_cast<float**>(&p.in[i])), bytes);
            (void)hipMemset(const_cast<float*>(p.in[i]), 0, bytes);
        }
        for (int i = 0; i < 2; ++i) (void)hipMalloc(reinterpret_cast<void**>(&p.out[i]), bytes);
        (void)hipDeviceSynchronize();
        const int reps = 20;
        printf("== %d x %d, ten planes of %.0f MiB (hipMalloc: every plane on a 2 MiB boundary)\n", w, h, bytes / 1048576.0);
        auto report = [&](const char* name, float us, double bytes_per_px) {
            printf("%-78s %8.1f us  %6.2f TB/s algorithmic\n", name, us, bytes_per_px * w * h / us / 1e6);
        };
        for (int rows : {16, 32, 64, 128}) {
            char name[128];
            snprintf(name, sizeof name, "8 read + 2 written, 4 B per lane, halo lanes and rows, strips of %d rows", rows);
            report(name, time_case<1, 1, 8, 2>(p, w, h, pitch, rows, reps), 40.0);
        }
        for (int rows : {16, 32, 64, 128}) {
            char name[128];
            snprintf(name, sizeof name, "8 read + 2 written, 4 B per lane, aligned strips + halo rows + 2-lane halo loads, %d rows", rows);
            report(name, time_case<1, 2, 8, 2>(p, w, h, pitch, rows, reps), 40.0);
        }
        for (int rows : {16, 64, 256}) {
            char name[128];
            snprintf(name, sizeof name, "8 read + 2 written, 4 B per lane, no halo, strips of %d rows", rows);
            report(name, time_case<1, 0, 8, 2>(p, w, h, pitch, rows, reps), 40.0);
        }
        for (int rows : {16, 64, 256}) {
            char name[128];
            snprintf(name, sizeof name, "8 read + 2 written, 16 B per lane, no halo, strips of %d rows", rows);
            report(name, time_case<4, 0, 8, 2>(p, w, h, pitch, rows, reps), 40.0);
        }
        report("6 read + 2 written (the fused strip kernel's planes), 4 B per lane, 128 rows", time_case<1, 0, 6, 2>(p, w, h, pitch, 128, reps), 32.0);
        report("4 read + 1 written, 16 B per lane, 64 rows", time_case<4, 0, 4, 1>(p, w, h, pitch, 64, reps), 20.0);
        report("1 read + 1 written (copy), 4 B per lane, 64 rows", time_case<1, 0, 1, 1>(p, w, h, pitch, 64, reps), 8.0);
        report("1 read + 1 written (copy), 16 B per lane, 64 rows", time_case<4, 0, 1, 1>(p, w, h, pitch, 64, reps), 8.0);
        report("8 read + 0 written, 16 B per lane, 64 rows", time_case<4, 0, 8, 0>(p, w, h, pitch, 64, reps), 32.0);
        for (int i = 0; i < 8; ++i) (void)hipFree(const_cast<float*>(p.in[i]));
        for (int i = 0; i < 2; ++i) (void)hipFree(p.out[i]);
    }
    return 0;
}
