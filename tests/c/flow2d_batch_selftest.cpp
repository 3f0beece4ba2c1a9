// flow2d_batch_selftest -- the rank logic of flow2d_batch (batch_driver.cpp) at any world size on a machine without a GPU.
//
// Same RunBatchRank as the product, with the two interfaces of batch_driver.h backed by
//   * LoopbackComm: the ranks are threads of this process; broadcast, all-reduce and gather go through a hub in shared
//     memory (mutex + condition variable, one rendezvous per collective) -- what librccl does over xGMI, reduced to its
//     meaning;
//   * StubDevice: planes live in host memory (pitch = width * 4 rounded up to 256 B, like flow2d_plane_pitch_bytes) and
//     the "flow" of a pair is a stamp of its two frames: u = 2 f0 + 1, v = f1 - f0 + levels (so a result that reached the
//     wrong file, came from the wrong rank's block or was computed with another rank's parameters is a wrong value).
// There is no optical flow here, by design: this binary exists for tests/test_batch_driver.py (pair -> rank, padded gather
// blocks, flow_%04d files, and that a rank which fails locally takes every rank out with the same exit code instead of
// leaving them blocked in a collective).  It links neither HIP nor RCCL and nothing in the product links it.
//
// usage: flow2d_batch_selftest --world N [flow2d_batch's job flags] [--fail-rank R --fail-phase init|load|warmup|pass|gather-alloc]
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "batch_driver.h"

namespace {

// ---- the hub the loopback ranks meet in ------------------------------------------------------------------------------
class LoopbackHub {
public:
    explicit LoopbackHub(int world) : world_(world), pointers_(world, nullptr), values_(world, 0) {}
    int World() const { return world_; }
    // every rank calls Meet with its pointer and value; the last one to arrive runs `op` over all of them, then all leave
    template <typename Op>
    void Meet(int rank, void* pointer, int value, Op op, int* value_out)
    {
        std::unique_lock<std::mutex> lock(mutex_);
        pointers_[rank] = pointer;
        values_[rank] = value;
        const unsigned long long round = round_;
        if (++arrived_ == world_) {
            op(pointers_, values_);
            result_ = values_[0];
            arrived_ = 0;
            ++round_;
            cv_.notify_all();
        } else {
            cv_.wait(lock, [&] { return round_ != round; });
        }
        if (value_out) *value_out = result_;
    }

private:
    int world_;
    std::mutex mutex_;
    std::condition_variable cv_;
    int arrived_ = 0;
    unsigned long long round_ = 0;
    std::vector<void*> pointers_;
    std::vector<int> values_;
    int result_ = 0;
};

class LoopbackComm : public BatchComm {
public:
    LoopbackComm(LoopbackHub& hub, int rank) : hub_(hub), rank_(rank) {}
    int Rank() const override { return rank_; }
    int World() const override { return hub_.World(); }
    bool Broadcast(void* buffer, size_t bytes, int root) override
    {
        hub_.Meet(rank_, buffer, 0, [&](std::vector<void*>& p, std::vector<int>&) {
            for (size_t r = 0; r < p.size(); ++r)
                if (static_cast<int>(r) != root) std::memcpy(p[r], p[root], bytes);
        }, nullptr);
        return true;
    }
    bool AllReduceMax(int* value) override
    {
        hub_.Meet(rank_, nullptr, *value, [&](std::vector<void*>&, std::vector<int>& v) {
            int m = v[0];
            for (int x : v) m = x > m ? x : m;
            v[0] = m;
        }, value);
        return true;
    }
    bool GatherToRoot(const void* send, void* recv, size_t block_bytes) override
    {
        // rank 0 publishes its receive buffer, the others their blocks; the copies are made by whoever arrives last
        void* mine = rank_ == 0 ? recv : const_cast<void*>(send);
        const void* root_send = send;
        hub_.Meet(rank_, mine, 0, [&](std::vector<void*>& p, std::vector<int>&) {
            for (size_t r = 1; r < p.size(); ++r)
                if (block_bytes) std::memcpy(static_cast<char*>(p[0]) + r * block_bytes, p[r], block_bytes);
        }, nullptr);
        if (rank_ == 0 && block_bytes) std::memcpy(recv, root_send, block_bytes);
        return true;
    }

private:
    LoopbackHub& hub_;
    int rank_;
};

// ---- a rank's "device": host memory and stamps --------------------------------------------------------------------------
struct Failure {
    int rank = -1;
    std::string phase;
};

class StubDevice : public BatchDevice {
public:
    StubDevice(int rank, const Failure& failure) : rank_(rank), failure_(failure) {}
    bool Fails(const char* phase) const { return rank_ == failure_.rank && failure_.phase == phase; }
    bool Initialize(size_t width, size_t height, int, size_t lanes, size_t group) override
    {
        if (Fails("init")) return false;
        width_ = width, height_ = height, lanes_ = lanes, group_ = group;
        pitch_ = (width * 4 + 255) / 256 * 256;
        return true;
    }
    size_t PitchBytes() const override { return pitch_; }
    size_t Lanes() const override { return lanes_; }
    HostMemory StagingMemory() const override { return HostMemory::Pageable; }
    void* Alloc(size_t bytes) override
    {
        if (Fails("gather-alloc") && ++allocations_ && bytes > 4 * pitch_ * height_ && passes_ > 0) return nullptr;
        return std::calloc(1, bytes ? bytes : 1);
    }
    void Free(void* p) override { std::free(p); }
    bool Upload(void* dst, const void* host, size_t bytes) override
    {
        std::memcpy(dst, host, bytes);
        return true;
    }
    bool Download(void* host, const void* src, size_t bytes) override
    {
        std::memcpy(host, src, bytes);
        return true;
    }
    bool UploadPlane(void* dst_plane, Data2D& image) override
    {
        if (Fails("load")) return false;
        for (size_t y = 0; y < image.Height(); ++y)
            std::memcpy(static_cast<char*>(dst_plane) + y * pitch_, image.DataPtr() + y * image.Width(), image.Width() * 4);
        return true;
    }
    bool DownloadPlane(Data2D& image, const void* src_plane) override
    {
        for (size_t y = 0; y < image.Height(); ++y)
            std::memcpy(image.DataPtr() + y * image.Width(), static_cast<const char*>(src_plane) + y * pitch_, image.Width() * 4);
        return true;
    }
    void BeginPhase(const char* name) override { phase_ = name; }
    bool Synchronize() override { return true; }
    bool QueuePass(size_t count, void* const* frames_0, void* const* frames_1, void* const* flows_u, void* const* flows_v,
                   OperationParameters& params, size_t first_lane) override
    {
        if (Fails(phase_ == "warmup" ? "warmup" : "pass") || first_lane >= lanes_) return false;
        ++passes_;
        size_t levels = 0;
        if (!params.Read("warp_levels_count", levels)) return false;
        for (size_t i = 0; i < count; ++i)
            for (size_t y = 0; y < height_; ++y) {
                const float* a = reinterpret_cast<const float*>(static_cast<const char*>(frames_0[i]) + y * pitch_);
                const float* b = reinterpret_cast<const float*>(static_cast<const char*>(frames_1[i]) + y * pitch_);
                float* u = reinterpret_cast<float*>(static_cast<char*>(flows_u[i]) + y * pitch_);
                float* v = reinterpret_cast<float*>(static_cast<char*>(flows_v[i]) + y * pitch_);
                for (size_t x = 0; x < width_; ++x) {
                    u[x] = 2.f * a[x] + 1.f;
                    v[x] = b[x] - a[x] + static_cast<float>(levels);
                }
            }
        return true;
    }
    void Destroy() override {}

private:
    int rank_;
    Failure failure_;
    size_t width_ = 0, height_ = 0, lanes_ = 0, group_ = 0, pitch_ = 0;
    int passes_ = 0, allocations_ = 0;
    std::string phase_;
};

struct SelfTestFlags {
    int world = 2;
    Failure failure;
};

int ExtraFlag(int argc, char** argv, int i, void* user)
{
    SelfTestFlags& f = *static_cast<SelfTestFlags*>(user);
    const std::string a = argv[i];
    if (i + 1 >= argc) return 0;
    if (a == "--fail-rank") {
        f.failure.rank = std::atoi(argv[i + 1]);
        return 2;
    }
    if (a == "--fail-phase") {
        f.failure.phase = argv[i + 1];
        return 2;
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    BatchOptions opt;
    SelfTestFlags flags;
    opt.p.width = 64, opt.p.height = 48, opt.p.lanes = 2, opt.p.group = 2;
    if (!ParseBatchArgs(argc, argv, opt, ExtraFlag, &flags)) {
        std::fprintf(stderr, "usage: flow2d_batch_selftest --world N %s       [--fail-rank R --fail-phase init|load|warmup|pass|gather-alloc]\n",
                     BatchUsage());
        return 3;
    }
    const int world = opt.world;
    if (world < 1 || world > 64) return 3;
    LoopbackHub hub(world);
    std::vector<int> codes(world, 0);
    std::vector<std::thread> threads;
    auto rank_main = [&](int rank) {
        BatchOptions mine = opt;
        // only rank 0's parameters count: give the others nonsense so that a rank that skipped the broadcast shows
        if (rank != 0) mine.p = BatchParameterBlock{1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        LoopbackComm comm(hub, rank);
        StubDevice device(rank, flags.failure);
        codes[rank] = RunBatchRank(mine, comm, device);
    };
    for (int r = 1; r < world; ++r) threads.emplace_back(rank_main, r);
    rank_main(0);
    for (std::thread& t : threads) t.join();
    int code = 0;
    bool same = true;
    for (int c : codes) {
        code = c > code ? c : code;
        same = same && c == codes[0];
    }
    if (!same) {
        std::fprintf(stderr, "flow2d_batch_selftest: the ranks disagree on the exit code:");
        for (int c : codes) std::fprintf(stderr, " %d", c);
        std::fprintf(stderr, "\n");
        return 99;
    }
    return code;
}
