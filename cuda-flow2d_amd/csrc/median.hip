// r x r median of a flow plane (r in {3,5,7}) for gfx950.
// Restates src/kernels/median_2d.cu:87-299 of the reference: window [x-r/2, x+r/2]^2 with
// reflect-without-repeat borders, output = element r*r/2 of the ascending window.  The reference
// sorts the window with a per-thread insertion sort; selection is order-free, so any exact
// selection network gives the same value (ties only differ in the sign of zero).
//
// Here the window lives in registers and goes through a Batcher odd-even merge network whose
// comparator list is generated at compile time; comparators that cannot influence the median
// are removed by the compiler's dead-code elimination.  One wave = 64 consecutive pixels of a row.
#include <array>
#include <utility>

#include "common.hpp"

namespace {

constexpr int kBlockX = 64;
constexpr int kBlockY = 4;

constexpr int next_pow2(int n)
{
    int p = 1;
    while (p < n) p *= 2;
    return p;
}

struct Comparator {
    int a, b;
};

// Batcher's odd-even merge sort on P = next_pow2(N) wires; comparators touching a wire >= N are
// dropped (those wires would hold +inf and never move).
template <int N>
constexpr int network_size()
{
    constexpr int P = next_pow2(N);
    int count = 0;
    for (int p = 1; p < P; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < P - j - k - 1 ? k - 1 : P - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < N) ++count;
    return count;
}

template <int N>
constexpr std::array<Comparator, network_size<N>()> make_network()
{
    constexpr int P = next_pow2(N);
    std::array<Comparator, network_size<N>()> net{};
    int count = 0;
    for (int p = 1; p < P; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < P - j - k - 1 ? k - 1 : P - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < N) {
                        net[count].a = i + j;
                        net[count].b = i + j + k;
                        ++count;
                    }
    return net;
}

template <int N, size_t... I>
__device__ __forceinline__ void run_network(float (&v)[N], std::index_sequence<I...>)
{
    constexpr auto net = make_network<N>();
    (
        [&] {
            const float lo = fminf(v[net[I].a], v[net[I].b]);
            const float hi = fmaxf(v[net[I].a], v[net[I].b]);
            v[net[I].a] = lo;
            v[net[I].b] = hi;
        }(),
        ...);
}

template <int R>
__global__ __launch_bounds__(256) void median_kernel(const float* __restrict__ in, int w, int h, int pitch,
                                                     float* __restrict__ out)
{
    constexpr int R2 = R / 2;
    constexpr int N = R * R;
    const int x = blockIdx.x * kBlockX + threadIdx.x;
    const int y = blockIdx.y * kBlockY + threadIdx.y;
    if (x >= w || y >= h) return;
    int xs[R];
#pragma unroll
    for (int i = 0; i < R; ++i) xs[i] = mirror_index(x + i - R2, w);
    float v[N];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float* row = in + static_cast<size_t>(mirror_index(y + j - R2, h)) * pitch;
#pragma unroll
        for (int i = 0; i < R; ++i) v[j * R + i] = row[xs[i]];
    }
    run_network<N>(v, std::make_index_sequence<network_size<N>()>{});
    out[static_cast<size_t>(y) * pitch + x] = v[N / 2];
}

}  // namespace

extern "C" int flow2d_median_2d(flow2d_context* ctx, const float* input, size_t width, size_t height,
                                size_t pitch_bytes, size_t window, float* output)
{
    FLOW2D_ENTER(ctx);
    if (!flow2d::plane_args_ok(input, width, height, pitch_bytes) ||
        !flow2d::plane_args_ok(output, width, height, pitch_bytes) || input == output)
        return FLOW2D_ERR_INVALID_ARGUMENT;
    if (window != 3 && window != 5 && window != 7) return FLOW2D_ERR_UNSUPPORTED;
    // the mirror rule needs every reflected index inside the image
    if (width <= window / 2 || height <= window / 2) return FLOW2D_ERR_UNSUPPORTED;
    const dim3 grid(flow2d::div_up(width, kBlockX), flow2d::div_up(height, kBlockY));
    const dim3 block(kBlockX, kBlockY);
    const int w = (int)width, h = (int)height, pitch = (int)(pitch_bytes / 4);
    switch (window) {
        case 3: median_kernel<3><<<grid, block, 0, ctx->stream>>>(input, w, h, pitch, output); break;
        case 5: median_kernel<5><<<grid, block, 0, ctx->stream>>>(input, w, h, pitch, output); break;
        default: median_kernel<7><<<grid, block, 0, ctx->stream>>>(input, w, h, pitch, output); break;
    }
    FLOW2D_CHECK_LAUNCH();
    return FLOW2D_OK;
}
