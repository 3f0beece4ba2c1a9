// Rank logic of flow2d_batch (see batch_driver.h): no HIP, no RCCL in this file -- only the two interfaces.
#include "batch_driver.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include <unistd.h>

std::vector<size_t> PairsOfRank(size_t total, int world, int rank)
{
    std::vector<size_t> mine;
    for (size_t k = static_cast<size_t>(rank); k < total; k += static_cast<size_t>(world)) mine.push_back(k);
    return mine;
}

size_t PairsPerBlock(size_t total, int world)
{
    return (total + static_cast<size_t>(world) - 1) / static_cast<size_t>(world);
}

size_t GatheredOffset(size_t k, int world, size_t per_rank, size_t plane_bytes)
{
    const size_t block_bytes = per_rank * 2 * plane_bytes;
    return (k % static_cast<size_t>(world)) * block_bytes + (k / static_cast<size_t>(world)) * 2 * plane_bytes;
}

void SyntheticBatchPair(Data2D& f0, Data2D& f1, double dx, double dy)
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

uint64_t Fnv1a(const void* data, size_t bytes, uint64_t h)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

const char* BatchUsage()
{
    return "[--gpus N | --rank R --world N --id-file PATH [--run-id NONCE] [--device D]] [--pairs K] [--width W] [--height H]\n"
           "       [--lanes L] [--group G] [--repeat R] [--levels n] [--scale s] [--outer n] [--inner n] [--alpha a] [--median m]\n"
           "       [--sigma s] [--e-smooth e] [--e-data e] [--constancy c] [--pairs-dir DIR] [--out-dir DIR] [--print-layout]\n";
}

bool ParseBatchArgs(int argc, char** argv, BatchOptions& o, int (*extra)(int, char**, int, void*), void* user)
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
        else if (a == "--constancy") ok = value(o.p.constancy);  // enum class DataConstancy: 0 Grey, 1 Gradient, 2 LogDerivatives, 3 GradientUntiled
        else if (a == "--gpus") { ok = value(d); o.gpus = static_cast<int>(d); }
        else if (a == "--rank") { ok = value(d); o.rank = static_cast<int>(d); }
        else if (a == "--world") { ok = value(d); o.world = static_cast<int>(d); }
        else if (a == "--device") { ok = value(d); o.device = static_cast<int>(d); }
        else if (a == "--id-file") ok = text(o.id_file);
        else if (a == "--run-id") ok = text(o.run_id);
        else if (a == "--pairs-dir") ok = text(o.pairs_dir);
        else if (a == "--out-dir") ok = text(o.out_dir);
        else if (a == "--print-layout") o.print_layout = true;
        else {
            const int taken = extra ? extra(argc, argv, i, user) : 0;
            ok = taken > 0;
            if (ok) i += taken - 1;
        }
        if (!ok) {
            std::fprintf(stderr, "flow2d_batch: bad argument '%s'\n", a.c_str());
            return false;
        }
    }
    return true;
}

// ---- the side channel ---------------------------------------------------------------------------------------------------
struct ThreadRendezvous::State {
    int world;
    double timeout;
    std::mutex mutex;
    std::condition_variable cv;
    int arrived = 0;
    bool all = true, result = false;
    unsigned long long round = 0;
    std::atomic<bool> raised{false};
};

ThreadRendezvous::ThreadRendezvous(int world, double timeout_seconds) : state_(new State)
{
    state_->world = world;
    state_->timeout = timeout_seconds;
}
ThreadRendezvous::~ThreadRendezvous() { delete state_; }

bool ThreadRendezvous::AllOk(int /*stage*/, bool ok)
{
    State& s = *state_;
    std::unique_lock<std::mutex> lock(s.mutex);
    s.all = s.all && ok;
    const unsigned long long round = s.round;
    if (++s.arrived == s.world) {
        s.result = s.all;
        s.arrived = 0;
        s.all = true;
        ++s.round;
        s.cv.notify_all();
        return s.result && !s.raised.load();
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(s.timeout);
    while (s.round == round) {
        if (s.raised.load() || std::chrono::steady_clock::now() >= deadline) {
            // a rank that never arrives, or one that left outside an agreement: nobody may count this or a later round as agreed
            s.raised.store(true);
            s.cv.notify_all();
            return false;
        }
        // (a system_clock deadline: pthread_cond_timedwait, which ThreadSanitizer knows; wait_for goes through pthread_cond_clockwait)
        s.cv.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(5));
    }
    return s.result && !s.raised.load();
}
void ThreadRendezvous::Raise()
{
    state_->raised.store(true);
    state_->cv.notify_all();
}
bool ThreadRendezvous::Raised() const { return state_->raised.load(); }

std::string FreshRunId()
{
    const auto now = std::chrono::steady_clock::now().time_since_epoch();
    return "p" + std::to_string(static_cast<long>(getpid())) + "t" +
           std::to_string(static_cast<long long>(std::chrono::duration_cast<std::chrono::microseconds>(now).count()));
}

FileRendezvous::FileRendezvous(const std::string& prefix, const std::string& run_id, int rank, int world, double timeout_seconds)
    : prefix_(prefix), run_id_(run_id.empty() ? "0" : run_id), rank_(rank), world_(world), timeout_(timeout_seconds)
{
}
FileRendezvous::~FileRendezvous()
{
    // after a failed agreement or a raised flag the files stay: a slower rank must still be able to read them (they carry the
    // run id, so a later run ignores them; tools/run_batch8.sh works in a directory of its own and removes it)
    if (!failed_)
        for (const std::string& f : written_) std::remove(f.c_str());
}

namespace {
bool WriteWhole(const std::string& path, const std::string& text)
{
    const std::string tmp = path + ".tmp";
    std::FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
    return std::fclose(f) == 0 && ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}
// "<run id> <0|1>": 1 / 0, or -1 for a missing, foreign or unfinished file
int ReadPost(const std::string& path, const std::string& run_id)
{
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return -1;
    char buffer[256] = {0};
    const size_t n = std::fread(buffer, 1, sizeof(buffer) - 1, f);
    std::fclose(f);
    const std::string text(buffer, n);
    if (text == run_id + " 1") return 1;
    if (text == run_id + " 0") return 0;
    return -1;
}
}  // namespace

bool FileRendezvous::AllOk(int stage, bool ok)
{
    auto name = [&](int r) { return prefix_ + ".s" + std::to_string(stage) + ".r" + std::to_string(r); };
    const std::string mine = name(rank_);
    if (!WriteWhole(mine, run_id_ + (ok ? " 1" : " 0"))) {
        Raise();
        return false;
    }
    written_.push_back(mine);
    if (!ok) failed_ = true;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_);
    bool all = ok;
    for (int r = 0; r < world_; ++r) {
        if (r == rank_) continue;
        int post = -1;
        while ((post = ReadPost(name(r), run_id_)) < 0) {
            if (Raised() || std::chrono::steady_clock::now() >= deadline) {
                Raise();
                return false;
            }
            usleep(2000);
        }
        all = all && post == 1;
    }
    all = all && !Raised();
    if (!all) failed_ = true;
    return all;
}
void FileRendezvous::Raise()
{
    failed_ = true;
    const std::string flag = prefix_ + ".abort." + run_id_;
    if (WriteWhole(flag, run_id_)) written_.push_back(flag);
}
bool FileRendezvous::Raised() const { return access((prefix_ + ".abort." + run_id_).c_str(), F_OK) == 0; }

int StartBatchRank(const BatchOptions& options, RankRendezvous& rendezvous, CommConnector& connector, BatchDevice& device)
{
    // (a) local prerequisites, agreed on before anybody enters the communicator's own rendezvous
    const bool prepared = connector.Prepare();
    if (!rendezvous.AllOk(0, prepared)) {
        std::fprintf(stderr, "flow2d_batch: %s before the communicator was set up; no rank enters it\n",
                     prepared ? "another rank failed" : "this rank failed");
        return 1;
    }
    // (b) the communicator; a rank that is connected while a peer is not must not use what it has
    const bool connected = connector.Connect();
    if (!rendezvous.AllOk(1, connected)) {
        std::fprintf(stderr, "flow2d_batch: the communicator did not come up on %s\n", connected ? "another rank" : "this rank");
        if (connected) connector.Abort();
        return 1;
    }
    return RunBatchRank(options, connector.Comm(), device);
}

namespace {

// everything a rank allocated, released on every way out of RunBatchRank
struct RankBuffers {
    BatchDevice& device;
    void* dev_block = nullptr;
    std::vector<void*> frames;
    void* flows = nullptr;
    void* gathered = nullptr;
    bool initialized = false;
    explicit RankBuffers(BatchDevice& d) : device(d) {}
    ~RankBuffers()
    {
        if (initialized) device.Destroy();
        for (void* p : frames) device.Free(p);
        device.Free(flows);
        device.Free(gathered);
        device.Free(dev_block);
    }
};

}  // namespace

int RunBatchRank(const BatchOptions& opt, BatchComm& comm, BatchDevice& device)
{
    const int rank = comm.Rank(), world = comm.World();
    RankBuffers mem(device);
    // the status agreement: the largest code of all ranks; a collective that itself fails leaves nothing to agree with
    auto agree = [&](int local) {
        int all = local;
        if (!comm.AllReduceMax(&all)) {
            std::fprintf(stderr, "flow2d_batch: rank %d: the status all-reduce failed\n", rank);
            return std::max(local, 1);
        }
        if (all != 0 && local == 0)
            std::fprintf(stderr, "flow2d_batch: rank %d: another rank failed (code %d), leaving\n", rank, all);
        return all;
    };

    // ---- 1. rank 0's parameter block on every rank --------------------------------------------------------------------
    BatchParameterBlock block = opt.p;
    if (rank != 0) std::memset(&block, 0, sizeof(block));  // whatever this rank was started with does not count
    int status = 0;
    mem.dev_block = device.Alloc(sizeof(block));
    if (!mem.dev_block || !device.Upload(mem.dev_block, &block, sizeof(block))) status = 1;
    if (int all = agree(status)) return all;
    // a broadcast (or the download behind it) that fails HERE after the rank took part is a local failure: the peers are
    // on their way into the next agreement, and so is this rank (a collective this rank never entered is the back end's
    // business: it raises the side channel's flag, and the agreement below fails on every rank)
    if (!comm.Broadcast(mem.dev_block, sizeof(block), 0) || !device.Download(&block, mem.dev_block, sizeof(block))) {
        std::fprintf(stderr, "flow2d_batch: rank %d: the parameter broadcast failed\n", rank);
        status = 1;
    }
    if (int all = agree(status)) return all;
    // every rank judges the same block, so they all leave together (3: the code of a bad command line)
    auto whole = [](double v, double low, double high) { return v >= low && v <= high && v == std::floor(v); };
    if (!whole(block.width, 1, 1 << 20) || !whole(block.height, 1, 1 << 20) || !whole(block.pairs_total, 0, 1e9) ||
        !whole(block.lanes, 1, 64) || !whole(block.group, 1, 1e9) || !whole(block.repeat, 1, 1e9) || !whole(block.levels, 1, 64) ||
        !whole(block.outer, 1, 1e6) || !whole(block.inner, 1, 1e6) || !whole(block.median, 0, 64) || !whole(block.constancy, 0, 3) ||
        !(block.scale > 0.0 && block.scale < 1.0)) {
        if (rank == 0) std::fprintf(stderr, "flow2d_batch: parameters out of range\nusage: flow2d_batch %s", BatchUsage());
        return 3;
    }
    const size_t width = static_cast<size_t>(block.width), height = static_cast<size_t>(block.height);
    const size_t total = static_cast<size_t>(block.pairs_total), repeat = std::max<size_t>(1, static_cast<size_t>(block.repeat));
    size_t group = std::max<size_t>(1, static_cast<size_t>(block.group));

    // ---- 2. this rank's pairs: k mod world == rank -----------------------------------------------------------------------
    const std::vector<size_t> mine = PairsOfRank(total, world, rank);
    const size_t per_rank = PairsPerBlock(total, world);  // block size of the gather (ranks with fewer pairs pad)
    group = std::min(group, std::max<size_t>(1, mine.size()));
    size_t pitch = 0, plane_bytes = 0, sets = 1;
    if (!device.Initialize(width, height, static_cast<int>(block.constancy), static_cast<size_t>(block.lanes), group)) {
        std::fprintf(stderr, "flow2d_batch: rank %d: the device side could not be initialised (%zu x %zu, %zu lanes, groups of %zu)\n",
                     rank, width, height, static_cast<size_t>(block.lanes), group);
        status = 1;
    } else {
        mem.initialized = true;
        pitch = device.PitchBytes();
        plane_bytes = pitch * height;
        // frames: one container per plane; flows: `sets` blocks [per_rank][2][height][pitch] in ONE allocation -- a block is
        // what the gather moves; with repeated passes every lane's worth of passes in flight writes a block of its own
        sets = repeat > 1 ? std::max<size_t>(1, device.Lanes()) : 1;
        mem.frames.assign(2 * mine.size(), nullptr);
        mem.flows = device.Alloc(sets * std::max<size_t>(1, per_rank) * 2 * plane_bytes);
        if (!mem.flows) status = 1;
        Data2D f0(width, height, device.StagingMemory()), f1(width, height, device.StagingMemory());
        for (size_t i = 0; i < mine.size() && status == 0; ++i) {
            const size_t k = mine[i];
            if (!opt.pairs_dir.empty()) {
                char name[64];
                std::snprintf(name, sizeof(name), "/pair_%04zu_0.raw", k);
                const bool a = f0.ReadRAWFromFileF32((opt.pairs_dir + name).c_str(), width, height);
                std::snprintf(name, sizeof(name), "/pair_%04zu_1.raw", k);
                const bool b = f1.ReadRAWFromFileF32((opt.pairs_dir + name).c_str(), width, height);
                if (!a || !b) {
                    status = 2;  // the CLI's exit code for a frame that cannot be loaded
                    break;
                }
            } else {
                SyntheticBatchPair(f0, f1, 2.0 * std::cos(static_cast<double>(k)), 2.0 * std::sin(static_cast<double>(k)));
            }
            Data2D* images[2] = {&f0, &f1};
            for (int j = 0; j < 2 && status == 0; ++j) {
                mem.frames[2 * i + j] = device.Alloc(plane_bytes);
                if (!mem.frames[2 * i + j] || !device.UploadPlane(mem.frames[2 * i + j], *images[j])) status = 1;
            }
        }
    }
    if (int all = agree(status)) return all;

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

    const size_t block_bytes = per_rank * 2 * plane_bytes;
    const size_t set_stride = std::max<size_t>(1, per_rank) * 2 * plane_bytes;
    const size_t lanes = std::max<size_t>(1, device.Lanes());
    const size_t groups = (mine.size() + group - 1) / group;  // lock-step groups of a pass, one lane each
    std::vector<void*> f0s, f1s;
    for (size_t i = 0; i < mine.size(); ++i) {
        f0s.push_back(mem.frames[2 * i]);
        f1s.push_back(mem.frames[2 * i + 1]);
    }
    // pass r: flow-buffer set r mod sets, first group on lane (r x groups) mod lanes -- a function of r mod lanes when
    // sets == lanes, so the graphs the warm-up records are the ones the timed passes replay
    auto queue_pass = [&](size_t r) {
        char* base = static_cast<char*>(mem.flows) + (r % sets) * set_stride;
        std::vector<void*> us, vs;
        for (size_t i = 0; i < mine.size(); ++i) {
            us.push_back(base + (2 * i) * plane_bytes);
            vs.push_back(base + (2 * i + 1) * plane_bytes);
        }
        return device.QueuePass(mine.size(), f0s.data(), f1s.data(), us.data(), vs.data(), params, (r * groups) % lanes);
    };

    // ---- 3. warm-up (records the graphs), 4. the timed passes; the agreements are the barriers around them ------------------
    device.BeginPhase("warmup");
    for (size_t r = 0; r < sets && status == 0; ++r)
        if (!queue_pass(r)) status = 1;
    if (!device.Synchronize()) status = std::max(status, 1);
    if (int all = agree(status)) return all;
    device.BeginPhase("timed");
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t r = 0; r < repeat && status == 0; ++r)
        if (!queue_pass(r)) status = 1;
    if (!device.Synchronize()) status = std::max(status, 1);
    // the rank's own time, taken before the closing agreement (whose all-reduce is not part of the passes); the job's time
    // is the slowest rank's: a second one-word all-reduce (maximum) of the microseconds
    const double own_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (int all = agree(status)) return all;
    int micros = static_cast<int>(std::min(own_seconds * 1e6, 2.0e9));
    if (!comm.AllReduceMax(&micros)) status = 1;
    const double seconds = std::max(1e-6, micros * 1e-6);
    const char* result_block = static_cast<char*>(mem.flows) + ((repeat - 1) % sets) * set_stride;  // the last pass's flows

    // ---- 5. gather: every rank's block of flow fields to rank 0 ----------------------------------------------------------------
    if (rank == 0) {
        mem.gathered = device.Alloc(std::max<size_t>(1, block_bytes) * static_cast<size_t>(world));
        if (!mem.gathered) status = 1;
    }
    if (int all = agree(status)) return all;
    const auto g0 = std::chrono::steady_clock::now();
    if (!comm.GatherToRoot(result_block, mem.gathered, block_bytes)) {  // (a local failure: on to the closing agreement)
        std::fprintf(stderr, "flow2d_batch: rank %d: the gather failed\n", rank);
        status = 1;
    }
    if (int all = agree(status)) return all;  // before rank 0 writes files or prints: a failed job leaves no result line
    const double gather_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();

    // ---- 6. rank 0: files and the JSON line -----------------------------------------------------------------------------------
    if (rank == 0 && status == 0) {
        uint64_t digest = 1469598103934665603ull;
        Data2D u(width, height), v(width, height);
        for (size_t k = 0; k < total && status != 1; ++k) {
            // pair k sits in rank (k mod world)'s block at position k / world
            const char* base = static_cast<const char*>(mem.gathered) + GatheredOffset(k, world, per_rank, plane_bytes);
            if (!device.DownloadPlane(u, base) || !device.DownloadPlane(v, base + plane_bytes)) {
                status = 1;
                break;
            }
            digest = Fnv1a(u.DataPtr(), width * height * 4, digest);
            digest = Fnv1a(v.DataPtr(), width * height * 4, digest);
            if (!opt.out_dir.empty()) {
                char name[64];
                std::snprintf(name, sizeof(name), "/flow_%04zu_u.raw", k);
                const bool a = u.WriteRAWToFileF32((opt.out_dir + name).c_str());
                std::snprintf(name, sizeof(name), "/flow_%04zu_v.raw", k);
                const bool b = v.WriteRAWToFileF32((opt.out_dir + name).c_str());
                if (!a || !b) status = 255;  // the reference's exit code for an output file that cannot be written
            }
        }
        if (status != 1) {
            const double pairs = static_cast<double>(total) * static_cast<double>(repeat);
            std::string layout;
            if (opt.print_layout) {
                layout = ", \"layout\": [";
                for (size_t k = 0; k < total; ++k) {
                    char item[96];
                    std::snprintf(item, sizeof(item), "%s[%zu, %zu, %zu, %zu]", k ? ", " : "", k, k % static_cast<size_t>(world),
                                  k / static_cast<size_t>(world), GatheredOffset(k, world, per_rank, plane_bytes));
                    layout += item;
                }
                layout += "]";
            }
            std::printf("{\"tool\": \"flow2d_batch\", \"world\": %d, \"pairs\": %zu, \"repeat\": %zu, \"width\": %zu, \"height\": %zu, "
                        "\"lanes\": %zu, \"group\": %zu, \"passes_in_flight\": %zu, \"seconds\": %.6f, \"pairs_per_s\": %.3f, \"mpixel_iters_per_s\": %.1f, "
                        "\"gather\": \"every rank's block to rank 0\", \"pairs_per_block\": %zu, \"pitch_bytes\": %zu, "
                        "\"gather_bytes_per_rank\": %zu, \"gather_ms\": %.3f, \"flows_fnv1a\": \"%016llx\"%s}\n",
                        world, total, repeat, width, height, device.Lanes(), group, sets, seconds, pairs / seconds,
                        pairs * static_cast<double>(width * height) * static_cast<double>(outer * inner) / seconds / 1e6, per_rank,
                        pitch, block_bytes, gather_seconds * 1e3, static_cast<unsigned long long>(digest), layout.c_str());
            std::fflush(stdout);
        }
    }
    return agree(status);
}
