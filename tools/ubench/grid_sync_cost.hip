// Micro-benchmark (round 4): what does a grid-wide barrier cost on MI355X?  VERDICT r03 item 6(i) proposes to run all outer
// iterations of a small pyramid level in ONE launch, with a cooperative grid barrier between outer iterations, instead of
// one 5-12 us tile launch per iteration.  Between two outer iterations every tile must see the flow increments its
// neighbours wrote -- workgroups on other XCDs included, i.e. through memory, not through one L2.  Measured here, per
// barrier, for a cooperative launch (hipLaunchCooperativeKernel: all workgroups resident by construction):
//   * cooperative_groups grid.sync()                      (the runtime's barrier: release / acquire fences + counter)
//   * a hand-written counter barrier with device-scope release / acquire (what a persistent tile kernel would use)
// Every workgroup also writes and reads a few words of its neighbour's data across the barrier, so that the fences are
// not optimised into nothing.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_sync_cost.hip -o build_ubench/grid_sync_cost
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>

namespace cg = cooperative_groups;

__global__ void k_grid_sync(float* data, int iters)
{
    cg::grid_group grid = cg::this_grid();
    const unsigned n = gridDim.x, b = blockIdx.x;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x < 64) data[(size_t)b * 64 + threadIdx.x] = acc + i;
        grid.sync();
        if (threadIdx.x < 64) acc += data[(size_t)((b + 1) % n) * 64 + threadIdx.x];
        grid.sync();
    }
    if (threadIdx.x < 64) data[(size_t)b * 64 + threadIdx.x] = acc;
}

__global__ void k_counter_barrier(float* data, unsigned* counter, int iters)
{
    const unsigned n = gridDim.x, b = blockIdx.x;
    float acc = 0.f;
    unsigned target = 0;
    auto barrier = [&]() {
        __syncthreads();
        target += n;
        if (threadIdx.x == 0) {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);  // device scope by default on a global address
            while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    };
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x < 64) __builtin_nontemporal_store(acc + i, &data[(size_t)b * 64 + threadIdx.x]);
        __threadfence();
        barrier();
        if (threadIdx.x < 64) acc += __builtin_nontemporal_load(&data[(size_t)((b + 1) % n) * 64 + threadIdx.x]);
        barrier();
    }
    if (threadIdx.x < 64) data[(size_t)b * 64 + threadIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* data;
    unsigned* counter;
    hipMalloc(&data, (size_t)4096 * 64 * sizeof(float));
    hipMalloc(&counter, sizeof(unsigned));
    const int iters = 200;
    printf("grid-wide barrier cost, cooperative launch, %d CUs; us per barrier (two per iteration, %d iterations)\n", cus, iters);
    for (int threads : {256, 1024}) {
        for (int per_cu : {1, 2}) {
            const int blocks = cus * per_cu;
            if (threads * per_cu > 2048) continue;
            for (int variant = 0; variant < 2; ++variant) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipMemset(counter, 0, sizeof(unsigned));
                    hipMemset(data, 0, (size_t)4096 * 64 * sizeof(float));
                    int it = iters;
                    void* args0[] = {&data, &it};
                    void* args1[] = {&data, &counter, &it};
                    hipEventRecord(e0);
                    hipError_t err = variant == 0
                                         ? hipLaunchCooperativeKernel((void*)k_grid_sync, dim3(blocks), dim3(threads), args0, 0, 0)
                                         : hipLaunchCooperativeKernel((void*)k_counter_barrier, dim3(blocks), dim3(threads), args1, 0, 0);
                    hipEventRecord(e1);
                    if (err != hipSuccess || hipEventSynchronize(e1) != hipSuccess) {
                        printf("launch failed: %s\n", hipGetErrorString(err));
                        return 1;
                    }
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("  %4d workgroups x %4d threads  %-34s %6.2f us per barrier\n", blocks, threads,
                       variant == 0 ? "cooperative_groups grid.sync()" : "counter barrier, release / acquire", best * 1e3 / (2.0 * iters));
            }
        }
    }
    return 0;
}
