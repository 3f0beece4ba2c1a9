// flow2d_batch -- a batch of independent image pairs over the GPUs of one node, all C++ (SURVEY Appendix C,
// BASELINE.json configs[3]): pair k goes to rank k mod world, every rank runs its pairs through OpticalFlowBatch2D (lanes,
// lock-step groups, graph replay) and the flow fields are gathered on rank 0 over RCCL.  The reference has no counterpart:
// it holds one context on device 0 (src/utils/cuda_utils.cpp:43) and computes one pair per process run (src/main.cpp).
//
// Collectives (librccl, directly): ncclBroadcast of rank 0's parameter block, so that every rank solves with rank 0's
// parameters; grouped ncclSend / ncclRecv of every rank's block of flow fields to rank 0; a one-word ncclAllReduce as
// the barrier around the timed region.  No data-path collective: the pairs are independent.
//
// Ranks: `--gpus N` starts N ranks as threads of this process, rank r on device r (ncclCommInitRank with an id shared in
// memory); `--rank R --world N --id-file PATH [--device D]` is one rank of N separate processes (rank 0 writes the
// ncclUniqueId to PATH, the others wait for it).
//
// Pairs: `--pairs-dir DIR` reads DIR/pair_%04d_0.raw and _1.raw (tight little-endian float32, the reference's raw format,
// src/data_types/data2d.cpp:140-178); without it the SURVEY 8(d) synthetic pair k (shift (2 cos k, 2 sin k)) is generated.
// `--out-dir DIR`: rank 0 writes DIR/flow_%04d_u.raw and _v.raw after the gather.  Rank 0 prints one JSON line.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "data2d.h"
#include "device_utils.h"
#include "optical_flow_batch_2d.h"

namespace {

struct ParameterBlock {  // what rank 0 broadcasts: plain numbers only
    double width, height, pairs_total, lanes, group, repeat;
    double levels, scale, outer, inner, alpha, e_smooth, e_data, median, sigma, constancy;
};

struct Options {
    ParameterBlock p{1920, 1080, 8, 4, 8, 1, 8, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5, 0};
    int gpus = 1, rank = -1, world = 1, device = -1;
    std::string id_file, pairs_dir, out_dir;
};

#define CHECK_HIP(expr)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "flow2d_batch: %s: %s\n", #expr, hipGetErrorString(e_));    \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define CHECK_NCCL(expr)                                                                     \
    do {                                                                                     \
        ncclResult_t e_ = (expr);                                                            \
        if (e_ != ncclSuccess) {                                                             \
            std::fprintf(stderr, "flow2d_batch: %s: %s\n", #expr, ncclGetErrorString(e_));   \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

// I0 = 128 + 60 sin(2 pi x / 64) cos(2 pi y / 48) + 30 sin(2 pi (x + 2 y) / 23.7), I1(x, y) = I0(x - dx, y - dy)
void SyntheticPair(Data2D& f0, Data2D& f1, double dx, double dy)
{
    const double two_pi = 2.0 * 3.14159265358979323846;
    auto image = [&](double x, double y) {
        return 128.0 + 60.0 * std::sin(two_pi * x / 64.0) * std::cos(two_pi * y / 48.0) + 30.0 * std::sin(two_pi * (x + 2.0 * y) / 23.7);
    };
    for (size_t y = 0; y < f0.Height(); ++y)
        for (size_t x = 0; x < f0.Width(); ++x) {
            f0.Data(x, y) = static_cast<float>(image(static_cast<double>(x), static_cast<double>(y)));
            f1.Data(x, y) = static_cast<float>(image(static_cast<double>(x) - dx, static_cast<double>(y) - dy));
        }
}

uint64_t Fnv1a(const void* data, size_t bytes, uint64_t h = 1469598103934665603ull)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

struct Shared {  // what the threads of --gpus N share
    ncclUniqueId id;
    std::atomic<int> failed{0};
};

int RankMain(const Options& opt, int rank, int world, int device, const ncclUniqueId& id)
{
    CHECK_HIP(hipSetDevice(device));
    ncclComm_t comm;
    CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    // ---- rank 0's parameter block on every rank ------------------------------------------------------------------
    ParameterBlock* dev_block = nullptr;
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dev_block), sizeof(ParameterBlock)));
    ParameterBlock block = opt.p;
    if (rank != 0) std::memset(&block, 0, sizeof(block));  // whatever this rank was started with does not count
    CHECK_HIP(hipMemcpyAsync(dev_block, &block, sizeof(block), hipMemcpyHostToDevice, stream));
    CHECK_NCCL(ncclBroadcast(dev_block, dev_block, sizeof(ParameterBlock), ncclUint8, 0, comm, stream));
    CHECK_HIP(hipMemcpyAsync(&block, dev_block, sizeof(block), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    const size_t width = static_cast<size_t>(block.width), height = static_cast<size_t>(block.height);
    const size_t total = static_cast<size_t>(block.pairs_total), repeat = std::max<size_t>(1, static_cast<size_t>(block.repeat));
    size_t group = std::max<size_t>(1, static_cast<size_t>(block.group));

    // ---- this rank's pairs: k mod world == rank ---------------------------------------------------------------------
    std::vector<size_t> mine;
    for (size_t k = static_cast<size_t>(rank); k < total; k += static_cast<size_t>(world)) mine.push_back(k);
    const size_t per_rank = (total + world - 1) / world;  // block size of the gather (ranks with fewer pairs pad)
    group = std::min(group, std::max<size_t>(1, mine.size()));

    OpticalFlowBatch2D batch;
    batch.silent = true;
    const DataSize3 size = {width, height, 1};
    if (!batch.Initialize(size, static_cast<DataConstancy>(static_cast<int>(block.constancy)),
                          static_cast<size_t>(block.lanes), device, group)) {
        std::fprintf(stderr, "flow2d_batch: rank %d: OpticalFlowBatch2D::Initialize failed\n", rank);
        return 1;
    }
    const size_t pitch = batch.ContainerSize().pitch;
    const size_t plane_bytes = pitch * height;
    flow2d_context* ctx = batch.LaneContext(0);

    // frames: one container per plane; flows: ONE allocation [per_rank][2][height][pitch], the block the gather moves
    std::vector<void*> frames(2 * mine.size(), nullptr);
    char* flows = nullptr;
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&flows), std::max<size_t>(1, per_rank) * 2 * plane_bytes));
    CHECK_HIP(hipMemsetAsync(flows, 0, std::max<size_t>(1, per_rank) * 2 * plane_bytes, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    {
        Data2D f0(width, height, HostMemory::Pinned), f1(width, height, HostMemory::Pinned);
        for (size_t i = 0; i < mine.size(); ++i) {
            const size_t k = mine[i];
            if (!opt.pairs_dir.empty()) {
                char name[64];
                std::snprintf(name, sizeof(name), "/pair_%04zu_0.raw", k);
                const bool a = f0.ReadRAWFromFileF32((opt.pairs_dir + name).c_str(), width, height);
                std::snprintf(name, sizeof(name), "/pair_%04zu_1.raw", k);
                const bool b = f1.ReadRAWFromFileF32((opt.pairs_dir + name).c_str(), width, height);
                if (!a || !b) return 2;  // the CLI's exit code for a frame that cannot be loaded
            } else {
                SyntheticPair(f0, f1, 2.0 * std::cos(static_cast<double>(k)), 2.0 * std::sin(static_cast<double>(k)));
            }
            Data2D* images[2] = {&f0, &f1};
            for (int j = 0; j < 2; ++j) {
                size_t got = 0;
                if (flow2d_plane_alloc(ctx, width, height, &frames[2 * i + j], &got) != FLOW2D_OK || got != pitch) return 1;
                if (flow2d_copy_h2d_2d(ctx, frames[2 * i + j], pitch, images[j]->DataPtr(), width * 4, width * 4, height) != FLOW2D_OK)
                    return 1;
            }
            if (flow2d_synchronize(ctx) != FLOW2D_OK) return 1;  // the host images are reused for the next pair
        }
    }

    size_t warp_levels = static_cast<size_t>(block.levels), outer = static_cast<size_t>(block.outer),
           inner = static_cast<size_t>(block.inner), median = static_cast<size_t>(block.median);
    float scale = static_cast<float>(block.scale), alpha = static_cast<float>(block.alpha),
          e_smooth = static_cast<float>(block.e_smooth), e_data = static_cast<float>(block.e_data),
          sigma = static_cast<float>(block.sigma);
    OperationParameters params;
    params.PushValuePtr("warp_levels_count", &warp_levels);
    params.PushValuePtr("warp_scale_factor", &scale);
    params.PushValuePtr("outer_iterations_count", &outer);
    params.PushValuePtr("inner_iterations_count", &inner);
    params.PushValuePtr("equation_alpha", &alpha);
    params.PushValuePtr("equation_smoothness", &e_smooth);
    params.PushValuePtr("equation_data", &e_data);
    params.PushValuePtr("median_radius", &median);
    params.PushValuePtr("gaussian_sigma", &sigma);

    std::vector<DevicePtr> f0s, f1s, us, vs;
    auto as_ptr = [](void* p) { return static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(p)); };
    for (size_t i = 0; i < mine.size(); ++i) {
        f0s.push_back(as_ptr(frames[2 * i]));
        f1s.push_back(as_ptr(frames[2 * i + 1]));
        us.push_back(as_ptr(flows + (2 * i) * plane_bytes));
        vs.push_back(as_ptr(flows + (2 * i + 1) * plane_bytes));
    }
    auto pass = [&]() {
        return batch.ComputeFlowBatchDeviceGrouped(mine.size(), f0s.data(), f1s.data(), us.data(), vs.data(), params) &&
               batch.Synchronize();
    };
    float* dev_word = nullptr;
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dev_word), sizeof(float)));
    CHECK_HIP(hipMemsetAsync(dev_word, 0, sizeof(float), stream));
    auto barrier = [&]() -> int {
        CHECK_NCCL(ncclAllReduce(dev_word, dev_word, 1, ncclFloat, ncclSum, comm, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        return 0;
    };

    // ---- warm-up (records the graphs), then the timed passes -----------------------------------------------------------
    if (!pass() || barrier()) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t r = 0; r < repeat; ++r)
        if (!pass()) return 1;
    if (barrier()) return 1;
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // ---- gather: every rank's block of flow fields to rank 0 (grouped send / recv) --------------------------------
    char* gathered = nullptr;
    const size_t block_bytes = per_rank * 2 * plane_bytes;
    if (rank == 0) CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&gathered), std::max<size_t>(1, block_bytes) * world));
    const auto g0 = std::chrono::steady_clock::now();
    CHECK_NCCL(ncclGroupStart());
    if (rank == 0) {
        for (int r = 1; r < world; ++r)
            CHECK_NCCL(ncclRecv(gathered + static_cast<size_t>(r) * block_bytes, block_bytes, ncclUint8, r, comm, stream));
    } else if (block_bytes) {
        CHECK_NCCL(ncclSend(flows, block_bytes, ncclUint8, 0, comm, stream));
    }
    CHECK_NCCL(ncclGroupEnd());
    if (rank == 0 && block_bytes) CHECK_HIP(hipMemcpyAsync(gathered, flows, block_bytes, hipMemcpyDeviceToDevice, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    const double gather_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();

    int rc = 0;
    if (rank == 0) {
        // pair k sits in rank (k mod world)'s block at position k / world
        uint64_t digest = 1469598103934665603ull;
        Data2D u(width, height), v(width, height);
        for (size_t k = 0; k < total; ++k) {
            const char* base = gathered + (k % world) * block_bytes + (k / world) * 2 * plane_bytes;
            CHECK_HIP(hipMemcpy2D(u.DataPtr(), width * 4, base, pitch, width * 4, height, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy2D(v.DataPtr(), width * 4, base + plane_bytes, pitch, width * 4, height, hipMemcpyDeviceToHost));
            digest = Fnv1a(u.DataPtr(), width * height * 4, digest);
            digest = Fnv1a(v.DataPtr(), width * height * 4, digest);
            if (!opt.out_dir.empty()) {
                char name[64];
                std::snprintf(name, sizeof(name), "/flow_%04zu_u.raw", k);
                const bool a = u.WriteRAWToFileF32((opt.out_dir + name).c_str());
                std::snprintf(name, sizeof(name), "/flow_%04zu_v.raw", k);
                const bool b = v.WriteRAWToFileF32((opt.out_dir + name).c_str());
                if (!a || !b) rc = 255;  // the reference's exit code for an output file that cannot be written
            }
        }
        const double pairs = static_cast<double>(total) * static_cast<double>(repeat);
        std::printf("{\"tool\": \"flow2d_batch\", \"world\": %d, \"pairs\": %zu, \"repeat\": %zu, \"width\": %zu, \"height\": %zu, "
                    "\"lanes\": %zu, \"group\": %zu, \"seconds\": %.6f, \"pairs_per_s\": %.3f, \"mpixel_iters_per_s\": %.1f, "
                    "\"gather\": \"ncclSend/ncclRecv to rank 0\", \"gather_bytes_per_rank\": %zu, \"gather_ms\": %.3f, "
                    "\"flows_fnv1a\": \"%016llx\"}\n",
                    world, total, repeat, width, height, batch.Lanes(), group, seconds, pairs / seconds,
                    pairs * static_cast<double>(width * height) * static_cast<double>(outer * inner) / seconds / 1e6, block_bytes,
                    gather_seconds * 1e3, static_cast<unsigned long long>(digest));
        std::fflush(stdout);
    }

    batch.Destroy();
    for (void* p : frames)
        if (p) (void)hipFree(p);
    (void)hipFree(flows);
    if (gathered) (void)hipFree(gathered);
    (void)hipFree(dev_word);
    (void)hipFree(dev_block);
    (void)hipStreamDestroy(stream);
    ncclCommDestroy(comm);
    return rc;
}

bool ParseArgs(int argc, char** argv, Options& o)
{
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](double& out) {
            if (i + 1 >= argc) return false;
            out = std::atof(argv[++i]);
            return true;
        };
        auto text = [&](std::string& out) {
            if (i + 1 >= argc) return false;
            out = argv[++i];
            return true;
        };
        double d = 0;
        bool ok = true;
        if (a == "--width") ok = value(o.p.width);
        else if (a == "--height") ok = value(o.p.height);
        else if (a == "--pairs") ok = value(o.p.pairs_total);
        else if (a == "--lanes") ok = value(o.p.lanes);
        else if (a == "--group") ok = value(o.p.group);
        else if (a == "--repeat") ok = value(o.p.repeat);
        else if (a == "--levels") ok = value(o.p.levels);
        else if (a == "--scale") ok = value(o.p.scale);
        else if (a == "--outer") ok = value(o.p.outer);
        else if (a == "--inner") ok = value(o.p.inner);
        else if (a == "--alpha") ok = value(o.p.alpha);
        else if (a == "--e-smooth") ok = value(o.p.e_smooth);
        else if (a == "--e-data") ok = value(o.p.e_data);
        else if (a == "--median") ok = value(o.p.median);
        else if (a == "--sigma") ok = value(o.p.sigma);
        else if (a == "--constancy") ok = value(o.p.constancy);  // enum class DataConstancy: 0 Grey, 1 Gradient, 2 LogDerivatives
        else if (a == "--gpus") { ok = value(d); o.gpus = static_cast<int>(d); }
        else if (a == "--rank") { ok = value(d); o.rank = static_cast<int>(d); }
        else if (a == "--world") { ok = value(d); o.world = static_cast<int>(d); }
        else if (a == "--device") { ok = value(d); o.device = static_cast<int>(d); }
        else if (a == "--id-file") ok = text(o.id_file);
        else if (a == "--pairs-dir") ok = text(o.pairs_dir);
        else if (a == "--out-dir") ok = text(o.out_dir);
        else ok = false;
        if (!ok) {
            std::fprintf(stderr, "flow2d_batch: bad argument '%s'\n", a.c_str());
            return false;
        }
    }
    return true;
}

}  // namespace

int main(int argc, char** argv)
{
    Options opt;
    if (!ParseArgs(argc, argv, opt)) {
        std::fprintf(stderr,
                     "usage: flow2d_batch [--gpus N | --rank R --world N --id-file PATH [--device D]] [--pairs K] [--width W] "
                     "[--height H]\n       [--lanes L] [--group G] [--repeat R] [--levels n] [--scale s] [--outer n] [--inner n] "
                     "[--alpha a] [--median m] [--sigma s]\n       [--constancy c] [--pairs-dir DIR] [--out-dir DIR]\n");
        return 3;
    }
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) {
        std::fprintf(stderr, "flow2d_batch: no HIP device (the flow2d path has no CPU fallback)\n");
        return 1;  // the CLI's exit code for "no device"
    }
    if (opt.rank >= 0) {  // one rank of `world` processes: the id travels through a file
        ncclUniqueId id;
        if (opt.id_file.empty() || opt.rank >= opt.world) return 3;
        if (opt.rank == 0) {
            if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
            const std::string tmp = opt.id_file + ".tmp";
            std::FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return 255;
            std::fclose(f);
            if (std::rename(tmp.c_str(), opt.id_file.c_str()) != 0) return 255;
        } else {
            std::FILE* f = nullptr;
            for (int tries = 0; tries < 600 && !(f = std::fopen(opt.id_file.c_str(), "rb")); ++tries) usleep(100000);
            if (!f || std::fread(&id, sizeof(id), 1, f) != 1) return 1;
            std::fclose(f);
        }
        return RankMain(opt, opt.rank, opt.world, opt.device >= 0 ? opt.device : opt.rank % devices, id);
    }
    if (opt.gpus < 1 || opt.gpus > devices) {
        std::fprintf(stderr, "flow2d_batch: --gpus %d with %d device(s)\n", opt.gpus, devices);
        return 1;
    }
    Shared shared;
    if (ncclGetUniqueId(&shared.id) != ncclSuccess) return 1;
    std::vector<std::thread> threads;
    std::vector<int> codes(opt.gpus, 0);
    for (int r = 1; r < opt.gpus; ++r)
        threads.emplace_back([&, r] { codes[r] = RankMain(opt, r, opt.gpus, r, shared.id); });
    codes[0] = RankMain(opt, 0, opt.gpus, 0, shared.id);
    for (std::thread& t : threads) t.join();
    for (int c : codes)
        if (c) return c;
    return 0;
}
