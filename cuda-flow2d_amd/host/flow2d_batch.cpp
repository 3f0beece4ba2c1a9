// flow2d_batch -- a batch of independent image pairs over the GPUs of one node, all C++ (SURVEY Appendix C,
// BASELINE.json configs[3]): pair k goes to rank k mod world, every rank runs its pairs through OpticalFlowBatch2D (lanes,
// lock-step groups, graph replay) and the flow fields are gathered on rank 0 over RCCL.  The reference has no counterpart:
// it holds one context on device 0 (src/utils/cuda_utils.cpp:43) and computes one pair per process run (src/main.cpp).
//
// This file is the product front end: the librccl back end of BatchComm, the HIP / OpticalFlowBatch2D back end of
// BatchDevice, and the two ways of starting ranks.  The rank logic itself -- parameter broadcast, pair -> rank mapping,
// padded gather blocks, status agreements, output files -- is batch_driver.cpp, which the CPU self-test
// (tests/c/flow2d_batch_selftest.cpp) runs at world sizes 2, 3 and 8 over an in-process loopback.
//
// Collectives (librccl, directly): ncclBroadcast of rank 0's parameter block; a one-word ncclAllReduce (maximum) as the
// barrier around the timed region and as the status agreement before every phase; grouped ncclSend / ncclRecv of every
// rank's block of flow fields to rank 0.  No data-path collective: the pairs are independent.
//
// Ranks: `--gpus N` starts N ranks as threads of this process, rank r on device r (ncclCommInitRank with an id shared in
// memory); `--rank R --world N --id-file PATH [--run-id NONCE] [--device D]` is one rank of N separate processes (rank 0
// writes run id + ncclUniqueId to PATH, the others wait for a file with their run id; tools/run_batch8.sh starts them).
// Either way the communicator comes up under a side channel's agreements and every wait watches its flag
// (batch_driver.h, StartBatchRank): a rank that fails before, inside or after ncclCommInitRank takes the others out with it.
//
// Pairs: `--pairs-dir DIR` reads DIR/pair_%04d_0.raw and _1.raw (tight little-endian float32, the reference's raw format,
// src/data_types/data2d.cpp:140-178); without it the SURVEY 8(d) synthetic pair k (shift (2 cos k, 2 sin k)) is generated.
// `--out-dir DIR`: rank 0 writes DIR/flow_%04d_u.raw and _v.raw after the gather.  Rank 0 prints one JSON line.
// Exit code: 0, or -- the same on every rank -- 1 (device / run failed), 2 (a frame could not be loaded), 255 (an output
// file could not be written); 3 for a bad command line.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "batch_driver.h"
#include "device_utils.h"
#include "optical_flow_batch_2d.h"

namespace {

bool HipOk(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    std::fprintf(stderr, "flow2d_batch: %s: %s\n", what, hipGetErrorString(e));
    return false;
}
bool NcclOk(ncclResult_t e, const char* what)
{
    if (e == ncclSuccess) return true;
    std::fprintf(stderr, "flow2d_batch: %s: %s\n", what, ncclGetErrorString(e));
    return false;
}
#define HIP_OK(expr) HipOk((expr), #expr)
#define NCCL_OK(expr) NcclOk((expr), #expr)

// ---- BatchComm over librccl ------------------------------------------------------------------------------------------
// Also the CommConnector that brings it up under the side channel's agreements (batch_driver.h): Prepare() makes everything
// that can fail locally -- stream, the all-reduce word -- BEFORE anybody enters ncclCommInitRank, so a rank that fails there
// keeps the others out of it.  Every wait polls the side channel's flag: when a peer has left outside an agreement, or the
// communicator reports an asynchronous error, the rank aborts the communicator (ncclCommAbort needs no peer) and returns
// false instead of sitting in hipStreamSynchronize for ever.
class RcclComm : public BatchComm, public CommConnector {
public:
    RcclComm(int rank, int world, int device, const ncclUniqueId& id, RankRendezvous& side)
        : rank_(rank), world_(world), device_(device), id_(id), side_(side)
    {
    }
    ~RcclComm() override
    {
        if (word_) (void)hipFree(word_);
        if (stream_) (void)hipStreamDestroy(stream_);
        if (comm_) ncclCommDestroy(comm_);
    }
    bool Prepare() override
    {
        return HIP_OK(hipSetDevice(device_)) && HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking)) &&
               HIP_OK(hipMalloc(reinterpret_cast<void**>(&word_), sizeof(int)));
    }
    bool Connect() override { return NCCL_OK(ncclCommInitRank(&comm_, world_, id_, rank_)); }
    void Abort() override
    {
        if (comm_) (void)ncclCommAbort(comm_);
        comm_ = nullptr;
    }
    BatchComm& Comm() override { return *this; }
    int Rank() const override { return rank_; }
    int World() const override { return world_; }
    bool Broadcast(void* buffer, size_t bytes, int root) override
    {
        return Usable() && Check(NCCL_OK(ncclBroadcast(buffer, buffer, bytes, ncclUint8, root, comm_, stream_))) && Wait();
    }
    bool AllReduceMax(int* value) override
    {
        return Usable() && Check(HIP_OK(hipMemcpyAsync(word_, value, sizeof(int), hipMemcpyHostToDevice, stream_)) &&
                                 NCCL_OK(ncclAllReduce(word_, word_, 1, ncclInt32, ncclMax, comm_, stream_)) &&
                                 HIP_OK(hipMemcpyAsync(value, word_, sizeof(int), hipMemcpyDeviceToHost, stream_))) &&
               Wait();
    }
    bool GatherToRoot(const void* send, void* recv, size_t block_bytes) override
    {
        if (!Usable() || !Check(NCCL_OK(ncclGroupStart()))) return false;
        bool ok = true;
        if (rank_ == 0) {
            for (int r = 1; r < world_ && ok; ++r)
                if (block_bytes)
                    ok = NCCL_OK(ncclRecv(static_cast<char*>(recv) + static_cast<size_t>(r) * block_bytes, block_bytes, ncclUint8, r,
                                          comm_, stream_));
        } else if (block_bytes) {
            ok = NCCL_OK(ncclSend(send, block_bytes, ncclUint8, 0, comm_, stream_));
        }
        ok = NCCL_OK(ncclGroupEnd()) && ok;
        if (ok && rank_ == 0 && block_bytes) ok = HIP_OK(hipMemcpyAsync(recv, send, block_bytes, hipMemcpyDeviceToDevice, stream_));
        return Check(ok) && Wait();
    }

private:
    bool Usable()
    {
        if (comm_ && !side_.Raised()) return true;
        Abort();
        return false;
    }
    // a collective this rank could not even enqueue: the peers are (or will be) blocked in theirs -- tell them
    bool Check(bool enqueued)
    {
        if (!enqueued) {
            side_.Raise();
            Abort();
        }
        return enqueued;
    }
    // the stream's work, with an eye on the side channel and on the communicator's asynchronous errors
    bool Wait()
    {
        for (unsigned spins = 0;; ++spins) {
            const hipError_t e = hipStreamQuery(stream_);
            if (e == hipSuccess) return true;
            ncclResult_t async = ncclSuccess;
            const bool broken = e != hipErrorNotReady || ncclCommGetAsyncError(comm_, &async) != ncclSuccess || async != ncclSuccess;
            if (broken || ((spins & 63u) == 63u && side_.Raised())) {
                if (broken) {
                    std::fprintf(stderr, "flow2d_batch: rank %d: the communicator failed while a collective was in flight\n", rank_);
                    side_.Raise();
                } else {
                    std::fprintf(stderr, "flow2d_batch: rank %d: another rank left; aborting the collective in flight\n", rank_);
                }
                Abort();
                return false;
            }
            if (spins > 2000) usleep(50);  // (the first microseconds busy: the all-reduce is the barrier of the timed passes)
        }
    }

    int rank_, world_, device_;
    ncclUniqueId id_;
    RankRendezvous& side_;
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    int* word_ = nullptr;
};

// ---- BatchDevice over HIP + OpticalFlowBatch2D -----------------------------------------------------------------------------
class HipBatchDevice : public BatchDevice {
public:
    // the calling thread's current device is `device`.  Copies and memsets run on a stream of the object's own: nothing
    // here may touch the NULL stream, whose implicit synchronisation would serialise the lanes of the batch.
    explicit HipBatchDevice(int device) : device_(device)
    {
        batch_.silent = true;
        if (!HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking))) stream_ = nullptr;
    }
    ~HipBatchDevice() override
    {
        if (stream_) (void)hipStreamDestroy(stream_);
    }
    bool Initialize(size_t width, size_t height, int data_constancy, size_t lanes, size_t group) override
    {
        height_ = height;
        const DataSize3 size = {width, height, 1};
        if (!batch_.Initialize(size, static_cast<DataConstancy>(data_constancy), lanes, device_, group)) return false;
        context_ = batch_.LaneContext(0);
        return context_ != nullptr;
    }
    size_t PitchBytes() const override { return batch_.ContainerSize().pitch; }
    size_t Lanes() const override { return batch_.Lanes(); }
    HostMemory StagingMemory() const override { return HostMemory::Pinned; }
    void* Alloc(size_t bytes) override
    {
        void* p = nullptr;
        if (!stream_ || !HIP_OK(hipMalloc(&p, bytes))) return nullptr;
        if (!HIP_OK(hipMemsetAsync(p, 0, bytes, stream_)) || !HIP_OK(hipStreamSynchronize(stream_))) {
            (void)hipFree(p);
            return nullptr;
        }
        return p;
    }
    void Free(void* p) override
    {
        if (p) (void)hipFree(p);
    }
    bool Upload(void* dst, const void* host, size_t bytes) override
    {
        return HIP_OK(hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, stream_)) && HIP_OK(hipStreamSynchronize(stream_));
    }
    bool Download(void* host, const void* src, size_t bytes) override
    {
        return HIP_OK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, stream_)) && HIP_OK(hipStreamSynchronize(stream_));
    }
    bool UploadPlane(void* dst_plane, Data2D& image) override
    {
        // on the lane's stream, and waited for: the host image is reused for the next pair
        return flow2d_copy_h2d_2d(context_, dst_plane, PitchBytes(), image.DataPtr(), image.Width() * 4, image.Width() * 4,
                                  image.Height()) == FLOW2D_OK &&
               flow2d_synchronize(context_) == FLOW2D_OK;
    }
    bool DownloadPlane(Data2D& image, const void* src_plane) override
    {
        return HIP_OK(hipMemcpy2DAsync(image.DataPtr(), image.Width() * 4, src_plane, PitchBytes(), image.Width() * 4,
                                       image.Height(), hipMemcpyDeviceToHost, stream_)) &&
               HIP_OK(hipStreamSynchronize(stream_));
    }
    bool QueuePass(size_t count, void* const* frames_0, void* const* frames_1, void* const* flows_u, void* const* flows_v,
                   OperationParameters& params, size_t first_lane) override
    {
        auto as_ptr = [](void* p) { return static_cast<DevicePtr>(reinterpret_cast<uintptr_t>(p)); };
        std::vector<DevicePtr> f0s, f1s, us, vs;
        for (size_t i = 0; i < count; ++i) {
            f0s.push_back(as_ptr(frames_0[i]));
            f1s.push_back(as_ptr(frames_1[i]));
            us.push_back(as_ptr(flows_u[i]));
            vs.push_back(as_ptr(flows_v[i]));
        }
        return batch_.ComputeFlowBatchDeviceGrouped(count, f0s.data(), f1s.data(), us.data(), vs.data(), params, first_lane);
    }
    bool Synchronize() override { return batch_.Synchronize(); }
    void Destroy() override { batch_.Destroy(); }

private:
    int device_;
    size_t height_ = 0;
    hipStream_t stream_ = nullptr;
    OpticalFlowBatch2D batch_;
    flow2d_context* context_ = nullptr;
};

int RankMain(const BatchOptions& opt, int rank, int world, int device, const ncclUniqueId& id, RankRendezvous& side)
{
    (void)hipSetDevice(device);  // (checked in RcclComm::Prepare, under the side channel's agreement)
    RcclComm comm(rank, world, device, id, side);
    HipBatchDevice dev(device);
    return StartBatchRank(opt, side, comm, dev);
}

}  // namespace

int main(int argc, char** argv)
{
    BatchOptions opt;
    if (!ParseBatchArgs(argc, argv, opt)) {
        std::fprintf(stderr, "usage: flow2d_batch %s", BatchUsage());
        return 3;
    }
    (void)flow2d_request_hw_queues(8);  // four lanes + RCCL's stream per device: before the process's first HIP call
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) {
        std::fprintf(stderr, "flow2d_batch: no HIP device (the flow2d path has no CPU fallback)\n");
        return 1;  // the CLI's exit code for "no device"
    }
    if (opt.rank >= 0) {  // one rank of `world` processes: the id travels through a file
        ncclUniqueId id;
        if (opt.id_file.empty() || opt.rank >= opt.world) return 3;
        // The file holds "<run id>\n" + the id.  Rank 0 removes whatever an earlier run left at the path before it writes,
        // the readers skip a file that carries another run id, and rank 0 removes the file again once every rank is past the
        // communicator's rendezvous.  The run id comes from the launcher (tools/run_batch8.sh makes a fresh one, and a fresh
        // directory, per job) and is MANDATORY for a job of several processes: a failed job leaves its flag and posts behind on
        // purpose, and under a default id the next job at the same path would take them for its own (every rank leaving at once
        // with "another rank failed": ADVICE r05).  A lone rank has no peer to agree an id with and makes its own.
        if (opt.run_id.empty() && opt.world > 1) {
            std::fprintf(stderr, "flow2d_batch: --rank with --world %d needs --run-id NONCE, the same fresh value on every rank of the job\n", opt.world);
            return 3;
        }
        const std::string run_id = opt.run_id.empty() ? FreshRunId() : opt.run_id;
        const std::string head = run_id + "\n";
        FileRendezvous side(opt.id_file, run_id, opt.rank, opt.world);
        if (opt.rank == 0) {
            std::remove(opt.id_file.c_str());
            // (a rank 0 that cannot publish the id raises the side channel's flag: its peers stop polling for the file at once and
            //  every rank returns 1 -- not 120 s later, and not with a code of rank 0's own)
            bool published = ncclGetUniqueId(&id) == ncclSuccess;
            if (published) {
                const std::string tmp = opt.id_file + ".tmp";
                std::FILE* f = std::fopen(tmp.c_str(), "wb");
                published = f && std::fwrite(head.data(), 1, head.size(), f) == head.size() && std::fwrite(&id, sizeof(id), 1, f) == 1;
                if (f) published = (std::fclose(f) == 0) && published;
                published = published && std::rename(tmp.c_str(), opt.id_file.c_str()) == 0;
                if (!published) std::remove(tmp.c_str());
            }
            if (!published) {
                std::fprintf(stderr, "flow2d_batch: rank 0: cannot publish the communicator's id at %s\n", opt.id_file.c_str());
                side.Raise();
                return 1;
            }
        } else {
            bool have = false;
            for (int tries = 0; tries < 1200 && !have && !side.Raised(); ++tries) {
                std::vector<char> text(head.size() + sizeof(id));
                std::FILE* f = std::fopen(opt.id_file.c_str(), "rb");
                if (f) {
                    have = std::fread(text.data(), 1, text.size(), f) == text.size() && std::memcmp(text.data(), head.data(), head.size()) == 0;
                    std::fclose(f);
                }
                if (have) std::memcpy(&id, text.data() + head.size(), sizeof(id));
                else usleep(100000);
            }
            if (!have) {
                std::fprintf(stderr, "flow2d_batch: rank %d: no id file of run '%s' at %s\n", opt.rank, run_id.c_str(), opt.id_file.c_str());
                side.Raise();
                return 1;
            }
        }
        const int code = RankMain(opt, opt.rank, opt.world, opt.device >= 0 ? opt.device : opt.rank % devices, id, side);
        if (opt.rank == 0) std::remove(opt.id_file.c_str());
        return code;
    }
    if (opt.gpus < 1 || opt.gpus > devices) {
        std::fprintf(stderr, "flow2d_batch: --gpus %d with %d device(s)\n", opt.gpus, devices);
        return 1;
    }
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    ThreadRendezvous side(opt.gpus);  // the ranks' side channel: memory of this process
    std::vector<std::thread> threads;
    std::vector<int> codes(opt.gpus, 0);
    for (int r = 1; r < opt.gpus; ++r) threads.emplace_back([&, r] { codes[r] = RankMain(opt, r, opt.gpus, r, id, side); });
    codes[0] = RankMain(opt, 0, opt.gpus, 0, id, side);
    for (std::thread& t : threads) t.join();
    int code = 0;
    for (int c : codes) code = c > code ? c : code;
    return code;
}
