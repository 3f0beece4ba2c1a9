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
// The communicator comes up through StartBatchRank like the product's (side-channel agreements before and after the
// connector's rendezvous); the side channel is a ThreadRendezvous, or -- with --file-rendezvous PREFIX -- the FileRendezvous the
// product's separate rank processes use (here still threads, every one with an object of its own over the same files).
//
// usage: flow2d_batch_selftest --world N [flow2d_batch's job flags] [--file-rendezvous PREFIX [--run-id NONCE]]
//            [--fail-rank R --fail-phase comm-prepare|comm-connect|broadcast|broadcast-absent|allreduce-absent|gather|gather-absent|
//                                        init|load|warmup|pass|gather-alloc]
//        "x" phases: the rank's collective fails AFTER it took part (a local failure); "x-absent": the rank never enters the
//        collective, raises the side channel's flag and leaves -- its peers are blocked in theirs until they see the flag.
#include <chrono>
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
    // every rank calls Meet with its pointer and value; the last one to arrive runs `op` over all of them, then all leave.
    // false: the side channel's flag went up while this rank was waiting (a peer left without entering the collective) --
    // what ncclCommAbort does for a rank blocked in librccl.
    template <typename Op>
    bool Meet(int rank, void* pointer, int value, Op op, int* value_out, const RankRendezvous& side)
    {
        std::unique_lock<std::mutex> lock(mutex_);
        if (side.Raised()) return false;  // (sticky: once a rank has left, no collective completes any more)
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
            while (round_ == round) {
                if (side.Raised()) {
                    --arrived_;
                    return false;
                }
                cv_.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(2));  // (pthread_cond_timedwait: known to TSan)
            }
        }
        if (value_out) *value_out = result_;
        return true;
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

struct Failure {
    int rank = -1;
    std::string phase;
};

class LoopbackComm : public BatchComm, public CommConnector {
public:
    LoopbackComm(LoopbackHub& hub, int rank, RankRendezvous& side, const Failure& failure) : hub_(hub), rank_(rank), side_(side), failure_(failure) {}
    bool Fails(const char* phase) const { return rank_ == failure_.rank && failure_.phase == phase; }
    // the connector: nothing to prepare or connect in shared memory, but both can be made to fail
    bool Prepare() override { return !Fails("comm-prepare"); }
    bool Connect() override
    {
        connected_ = !Fails("comm-connect");
        return connected_;
    }
    void Abort() override { connected_ = false; }
    BatchComm& Comm() override { return *this; }
    int Rank() const override { return rank_; }
    int World() const override { return hub_.World(); }
    bool Broadcast(void* buffer, size_t bytes, int root) override
    {
        if (Absent("broadcast-absent")) return false;
        const bool met = hub_.Meet(rank_, buffer, 0, [&](std::vector<void*>& p, std::vector<int>&) {
            for (size_t r = 0; r < p.size(); ++r)
                if (static_cast<int>(r) != root) std::memcpy(p[r], p[root], bytes);
        }, nullptr, side_);
        return met && !Fails("broadcast");
    }
    bool AllReduceMax(int* value) override
    {
        if (++all_reduces_ == 3 && Absent("allreduce-absent")) return false;
        return hub_.Meet(rank_, nullptr, *value, [&](std::vector<void*>&, std::vector<int>& v) {
            int m = v[0];
            for (int x : v) m = x > m ? x : m;
            v[0] = m;
        }, value, side_);
    }
    bool GatherToRoot(const void* send, void* recv, size_t block_bytes) override
    {
        if (Absent("gather-absent")) return false;
        // rank 0 publishes its receive buffer, the others their blocks; the copies are made by whoever arrives last
        void* mine = rank_ == 0 ? recv : const_cast<void*>(send);
        const void* root_send = send;
        const bool met = hub_.Meet(rank_, mine, 0, [&](std::vector<void*>& p, std::vector<int>&) {
            for (size_t r = 1; r < p.size(); ++r)
                if (block_bytes) std::memcpy(static_cast<char*>(p[0]) + r * block_bytes, p[r], block_bytes);
        }, nullptr, side_);
        if (met && rank_ == 0 && block_bytes) std::memcpy(recv, root_send, block_bytes);
        return met && !Fails("gather");
    }

private:
    // the rank cannot enter the collective at all: like the product's back end it raises the flag before it returns
    bool Absent(const char* phase)
    {
        if (!Fails(phase)) return false;
        side_.Raise();
        return true;
    }
    LoopbackHub& hub_;
    int rank_;
    RankRendezvous& side_;
    Failure failure_;
    bool connected_ = false;
    int all_reduces_ = 0;
};

// ---- a rank's "device": host memory and stamps --------------------------------------------------------------------------
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
    std::string file_prefix;  // --file-rendezvous: the product's FileRendezvous instead of the ThreadRendezvous
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
    if (a == "--file-rendezvous") {
        f.file_prefix = argv[i + 1];
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
        std::fprintf(stderr, "usage: flow2d_batch_selftest --world N %s       [--file-rendezvous PREFIX] [--fail-rank R --fail-phase PHASE]\n",
                     BatchUsage());
        return 3;
    }
    const int world = opt.world;
    if (world < 1 || world > 64) return 3;
    // the ranks are threads of this process: without --run-id they share an id nobody else has (a default id would make the
    // files a failed job leaves behind count as this job's own)
    if (!flags.file_prefix.empty() && opt.run_id.empty()) opt.run_id = FreshRunId();
    LoopbackHub hub(world);
    ThreadRendezvous shared_side(world, 20.0);
    std::vector<int> codes(world, 0);
    std::vector<std::thread> threads;
    auto rank_main = [&](int rank) {
        BatchOptions mine = opt;
        // only rank 0's parameters count: give the others nonsense so that a rank that skipped the broadcast shows
        if (rank != 0) mine.p = BatchParameterBlock{1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        FileRendezvous file_side(flags.file_prefix, opt.run_id, rank, world, 20.0);
        RankRendezvous& side = flags.file_prefix.empty() ? static_cast<RankRendezvous&>(shared_side) : file_side;
        LoopbackComm comm(hub, rank, side, flags.failure);
        StubDevice device(rank, flags.failure);
        codes[rank] = StartBatchRank(mine, side, comm, device);
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
