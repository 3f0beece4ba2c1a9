// The rank logic of flow2d_batch -- independent image pairs over the GPUs of one node (BASELINE.json
// configs[3]) -- written against two small interfaces, so that the very same code runs with
//   * librccl + HIP + OpticalFlowBatch2D          (the product: flow2d_batch.cpp), and
//   * an in-process loopback + host memory + stamps (tests/c/flow2d_batch_selftest.cpp: world sizes 2, 3, 8 on a machine without
//     a GPU; the pair -> rank mapping, the padded gather blocks and the failure protocol are what is under test).
// The reference has no counterpart: it holds one context on device 0 (src/utils/cuda_utils.cpp:43) and computes one
// pair per process run (src/main.cpp).
//
// Protocol of a rank (RunBatchRank):
//   1. broadcast of rank 0's parameter block: every rank solves with rank 0's parameters
//   2. pair k -> rank k mod world; the rank loads (or generates) its pairs and uploads them          -> agreement
//   3. warm-up passes (record the graphs: one pass per flow-buffer set)                             -> agreement
//   4. `repeat` timed passes, queued back to back -- pass r starts on lane (r x groups) mod lanes and writes flow-buffer
//      set r mod lanes, so consecutive passes overlap on the GPU like the steps of bench.py -- one wait at the end  -> agreement
//   5. gather: every rank's block [per_rank][2][height][pitch] of flow fields to rank 0 (ranks with fewer pairs pad)
//   6. rank 0 writes flow_%04d_{u,v}.raw, prints one JSON line                                      -> agreement
// An "agreement" is a one-word all-reduce (maximum) of the ranks' status codes: it is the barrier of the timed region,
// and it is how a rank that failed locally takes the others down with it instead of leaving them blocked in the next
// collective -- every rank keeps joining the collectives until an agreement tells all of them to stop, and all of them
// return the same non-zero code (the largest: 1 device / run, 2 frame not loadable, 255 output not writable).  A
// collective that fails on a rank AFTER it took part (the broadcast, the gather) is a local failure like any other: the
// rank goes on to the next agreement with code 1.
//
// Bringing the communicator up (StartBatchRank) cannot use the communicator, so it goes through a SIDE CHANNEL the ranks
// share without it (RankRendezvous: memory for ranks that are threads, small files for ranks that are processes):
//   a. every rank's local prerequisites (device selected, stream, buffers)      -> side-channel agreement
//   b. the communicator's own rendezvous (ncclCommInitRank), entered only when (a) succeeded everywhere
//                                                                              -> side-channel agreement; a rank that
//      connected while another did not aborts its communicator instead of using it
// The side channel also carries one flag any rank can RAISE when it has to leave outside an agreement (a collective that
// failed before the rank took part, a torn communicator): the back ends' waits poll it, so the peers of such a rank
// leave their blocked collective within milliseconds (ncclCommAbort) instead of waiting for ever.  Every rank then
// returns 1.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "data2d.h"
#include "data_structs.h"
#include "operation_parameters.h"

// ---- the collectives ------------------------------------------------------------------------------------------------
// Five entry points (creation is the back end's constructor): buffers are memory of the BatchDevice the rank runs on.
class BatchComm {
public:
    virtual ~BatchComm() {}
    virtual int Rank() const = 0;
    virtual int World() const = 0;
    // rank `root`'s `bytes` at `buffer` into every rank's `buffer`
    virtual bool Broadcast(void* buffer, size_t bytes, int root) = 0;
    // on return *value is the maximum over all ranks (also the barrier: nobody returns before everybody entered)
    virtual bool AllReduceMax(int* value) = 0;
    // every rank's `block_bytes` at `send` to rank 0's `recv + rank * block_bytes` (rank 0's own block too);
    // `recv` is only looked at on rank 0
    virtual bool GatherToRoot(const void* send, void* recv, size_t block_bytes) = 0;
};

// ---- the side channel: agreements and the abort flag that do not need the communicator --------------------------------
class RankRendezvous {
public:
    virtual ~RankRendezvous() {}
    // Every rank posts `ok` for `stage` (0, 1, ...: in the same order on every rank); true only when ALL ranks posted true.
    // A rank that never arrives, or a raised flag, makes the others give up (false) instead of waiting for ever.
    virtual bool AllOk(int stage, bool ok) = 0;
    virtual void Raise() = 0;         // "I am leaving outside an agreement": sticky, seen by every rank
    virtual bool Raised() const = 0;  // cheap enough to poll every millisecond
};

// ranks = threads of one process: ONE object shared by all of them
class ThreadRendezvous : public RankRendezvous {
public:
    explicit ThreadRendezvous(int world, double timeout_seconds = 120.0);
    ~ThreadRendezvous() override;
    bool AllOk(int stage, bool ok) override;
    void Raise() override;
    bool Raised() const override;

private:
    struct State;
    State* state_;
};

// ranks = processes that share a directory: rank r posts stage s as the file <prefix>.s<s>.r<r> holding "<run id> <0|1>",
// the flag is the file <prefix>.abort.<run id>.  A stale file of an earlier run carries another run id and is ignored;
// every rank removes what it wrote when it is destroyed.
// A run id nobody else uses (process id + clock): for a job whose ranks all live in one process, or have no peers.  A job of
// several PROCESSES must be given one id by its launcher (--run-id; tools/run_batch8.sh does): with a default id a failed job's
// flag and posts, which stay behind on purpose, would be taken for the next job's own.
std::string FreshRunId();

class FileRendezvous : public RankRendezvous {
public:
    FileRendezvous(const std::string& prefix, const std::string& run_id, int rank, int world, double timeout_seconds = 120.0);
    ~FileRendezvous() override;
    bool AllOk(int stage, bool ok) override;
    void Raise() override;
    bool Raised() const override;

private:
    std::string prefix_, run_id_;
    int rank_, world_;
    double timeout_;
    bool failed_ = false;
    std::vector<std::string> written_;
};

// ---- bringing a rank's communicator up ---------------------------------------------------------------------------------
class CommConnector {
public:
    virtual ~CommConnector() {}
    virtual bool Prepare() = 0;  // local prerequisites of the communicator; no rendezvous inside
    virtual bool Connect() = 0;  // the communicator's own rendezvous; entered only when every rank is prepared
    virtual void Abort() = 0;    // a peer did not connect: tear down what Connect built, without a collective
    virtual BatchComm& Comm() = 0;
};

// ---- memory and computation of one rank --------------------------------------------------------------------------------
class BatchDevice {
public:
    virtual ~BatchDevice() {}
    // lanes x (stream, OpticalFlow2D, plane pool), lock-step groups of `group` pairs
    virtual bool Initialize(size_t width, size_t height, int data_constancy, size_t lanes, size_t group) = 0;
    virtual size_t PitchBytes() const = 0;  // of every plane handed to Pass (valid after Initialize)
    virtual size_t Lanes() const = 0;
    virtual HostMemory StagingMemory() const = 0;  // what the host images of the loader should live in
    virtual void* Alloc(size_t bytes) = 0;         // zero-filled; nullptr on failure
    virtual void Free(void* p) = 0;
    virtual bool Upload(void* dst, const void* host, size_t bytes) = 0;
    virtual bool Download(void* host, const void* src, size_t bytes) = 0;
    virtual bool UploadPlane(void* dst_plane, Data2D& image) = 0;          // tight host rows -> pitched rows
    virtual bool DownloadPlane(Data2D& image, const void* src_plane) = 0;  // and back
    // Queues the flows of `count` independent pairs (planes of PitchBytes() x height), the first group of them on lane
    // `first_lane`, the following groups on the following lanes; nothing is waited for until Synchronize().
    virtual bool QueuePass(size_t count, void* const* frames_0, void* const* frames_1, void* const* flows_u,
                           void* const* flows_v, OperationParameters& params, size_t first_lane) = 0;
    virtual bool Synchronize() = 0;
    virtual void BeginPhase(const char* name) { (void)name; }  // "warmup" / "timed": a hook for the self-test's failure injection
    virtual void Destroy() = 0;
};

struct BatchParameterBlock {  // what rank 0 broadcasts: plain numbers only
    double width, height, pairs_total, lanes, group, repeat;
    double levels, scale, outer, inner, alpha, e_smooth, e_data, median, sigma, constancy;
};

struct BatchOptions {
    BatchParameterBlock p{1920, 1080, 8, 4, 8, 1, 8, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5, 0};
    int gpus = 1, rank = -1, world = 1, device = -1;
    std::string id_file, run_id, pairs_dir, out_dir;
    bool print_layout = false;  // rank 0 adds the pair -> (rank, slot, byte offset) table to its JSON line
};

// false (after a message on stderr) for an argument it does not know; `extra` lets a front end take flags of its own
// (returns how many argv entries it consumed at position i, 0 = not mine).
bool ParseBatchArgs(int argc, char** argv, BatchOptions& options, int (*extra)(int argc, char** argv, int i, void* user) = nullptr,
                    void* user = nullptr);
const char* BatchUsage();

// ---- the layout of the job (pure functions: what the CPU tests pin) ---------------------------------------------------
// pairs of `rank`: k = rank, rank + world, ... < total
std::vector<size_t> PairsOfRank(size_t total, int world, int rank);
// pairs in a rank's gather block: ceil(total / world) -- ranks with fewer pairs pad
size_t PairsPerBlock(size_t total, int world);
// where pair k's (u, v) planes sit in rank 0's gathered buffer: block of rank k mod world, slot k / world
size_t GatheredOffset(size_t k, int world, size_t per_rank, size_t plane_bytes);

// SURVEY 8(d) synthetic pair: I0 = 128 + 60 sin(2 pi x / 64) cos(2 pi y / 48) + 30 sin(2 pi (x + 2 y) / 23.7), I1 = I0 shifted
void SyntheticBatchPair(Data2D& f0, Data2D& f1, double dx, double dy);
uint64_t Fnv1a(const void* data, size_t bytes, uint64_t h = 1469598103934665603ull);

// One rank of the job.  Returns the job's exit code -- the same on every rank: 0, or the largest local failure code.
int RunBatchRank(const BatchOptions& options, BatchComm& comm, BatchDevice& device);
// The same with the communicator still to be brought up: steps (a) and (b) above, then RunBatchRank.
int StartBatchRank(const BatchOptions& options, RankRendezvous& rendezvous, CommConnector& connector, BatchDevice& device);
