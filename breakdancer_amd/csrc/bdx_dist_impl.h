// Chromosome-sharded runs: one whole-genome result from chromosomes spread over several GPUs (include/bdx.h, bdx_dist_*).
// Included at the end of bdx_api.hip (it drives the stage functions of that translation unit).
//
// The path shards by chromosome -- regions never span tids (BreakDancer.cpp:216) -- and what a single breakdancer-max run
// couples across chromosomes is small: the pass-1 statistics (window, lambda, densities: BamSummary.cpp:129-150,
// BreakDancerMax.cpp:83-116), the running counters sampled at region boundaries, the read that closes a chromosome's
// last candidate region (BreakDancer.cpp:202-231), the region numbering / flush cadence (BreakDancer.cpp:254-259), and the
// inter-chromosomal read pairs (ARP_CTX).  Every rank (one per GPU) holds ALL of its chromosomes in ONE context and runs the
// single-context launch sequence over them -- K1 ... K6 and the table kernel, one launch each whatever the number of
// chromosomes --, with per-chromosome tables (k9_shard.hip) carrying what crosses a chromosome boundary.  Region ids are
// genome-wide on every rank, so flush windows, order keys and the walk itself are the single run's.  FIVE collectives per run
// (round 5: eleven -- six all-reduces, two all-to-alls, three gathers, each behind a host round trip):
//   A   all-reduce (host words): pass-1 counters, per-file reference lengths, per-chromosome totals and owners, every chromosome's
//       first anomalous read, and what the one all-to-all will carry -- inter-chromosomal reads by mate chromosome, census records by
//       owner (the compaction runs in front of it: it needs nothing of the other ranks)
//   B   all-reduce (host words): every chromosome's region count -> genome-wide region ids (every entry has one owner: a sum is a gather)
//   X   ONE all-to-all: per destination the inter-chromosomal join records -- a CTX read whose mate lies on a later chromosome of
//       another rank goes there and joins that rank's reads (k7_exchange.hip) --, the census records of the names it owns, and the read
//       lengths of the flush windows this rank closes.  From here to its finished table a rank waits for nobody: which components span
//       ranks is known without an exchange (a region that sent a join record away, or was joined with a foreign one, is an end of one)
//   S   all-reduce (host words): what every rank will send to rank 0 (pair groups of the components that span ranks or that its device
//       walk leaves, rows and list entries of its table), and whether a read name misbehaved
//   G   ONE gather: every rank's package -- its region records, those pair groups, its finished table (rows sorted by order key) -- to
//       rank 0, which walks the gathered groups in its result context (K6 once more, on the device) and merges the tables by key
// Payloads stay in HBM: the all-to-all and the gather run on device buffers through RCCL (ncclAllToAllv, grouped ncclSend / ncclRecv),
// xGMI between the GPUs of a node; the three all-reduces carry a few KB of host words (staged through device words for RCCL).  A second
// backend runs the ranks as threads of one process (tests on a single GPU; a host program that drives several GPUs itself).
#include <dlfcn.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>

namespace {

constexpr int kDistPhases = 18;   // bdx_dist_get_phase_ms / bdx_dist_phase_name
constexpr size_t kGatherHostMax = 2048;   // gathered pair groups up to which rank 0's HOST walks them (more: K6 on the device)

// ------------------------------------------------------------------------------------------------------------------
// communicators
// ------------------------------------------------------------------------------------------------------------------
struct Comm {
    int rank = 0, world = 1;
    std::string err;
    virtual ~Comm() {}
    uint32_t n_allreduce = 0, n_alltoall = 0, n_gather = 0;   // collectives entered since begin_run (bdx_dist_get_collectives)
    virtual void begin_run() {}   // start of a bdx_dist_run (collective)
    virtual void abort() {}       // this rank leaves the run between two collectives: wake whoever waits for it
    // in place sum over the ranks of n 64-bit words in HOST memory, without the device: false = this backend has no such path (the caller
    // stages the words through device memory and calls allreduce_u64)
    virtual bool allreduce_host_u64(uint64_t*, size_t, bool* done) { *done = false; return true; }
    // in place sum over the ranks of n 64-bit words in device memory
    virtual bool allreduce_u64(uint64_t* dev, size_t n, hipStream_t s) = 0;
    // rank r sends scount[d] words from send + sdispl[d] to rank d and receives rcount[d] words from rank d at recv + rdispl[d]
    virtual bool alltoallv_u64(const uint64_t* send, const size_t* scount, const size_t* sdispl, uint64_t* recv, const size_t* rcount,
                               const size_t* rdispl, hipStream_t s) = 0;
    // every rank sends n bytes (n a multiple of 8) to the root, which places rank r's at recv + displ[r] (count[r] bytes)
    virtual bool gatherv_bytes(const void* send, size_t n, void* recv, const size_t* count, const size_t* displ, int root, hipStream_t s) = 0;
};

// ---- RCCL (one process per GPU) -- resolved at run time so that single-GPU users do not depend on librccl ----
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, bdx_unique_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllToAllv)(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    std::string err, path;
    int version = 0;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("librccl not found: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        AllToAllv = (decltype(AllToAllv))sym("ncclAllToAllv");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        GetVersion = (decltype(GetVersion))dlsym(lib, "ncclGetVersion");
        if (GetVersion) (void)GetVersion(&version);
        {   // which file the loader resolved the name to (a process that imported torch first has torch's bundled copy mapped already)
            Dl_info info{};
            if (GetUniqueId && dladdr((void*)GetUniqueId, &info) && info.dli_fname) path = info.dli_fname;
        }
        return err.empty();
    }
};
RcclApi& rccl() { static RcclApi a; return a; }
constexpr int kNcclUint8 = 1, kNcclUint64 = 5, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values of rccl.h

struct RcclComm : Comm {
    void* comm = nullptr;
    bool ok(int rc, const char* what) {
        if (rc == 0) return true;
        err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
        return false;
    }
    ~RcclComm() override { if (comm) (void)rccl().CommDestroy(comm); }
    void begin_run() override { n_allreduce = n_alltoall = n_gather = 0; }
    bool allreduce_u64(uint64_t* dev, size_t n, hipStream_t s) override {
        ++n_allreduce;
        return ok(rccl().AllReduce(dev, dev, n, kNcclUint64, kNcclSum, comm, s), "ncclAllReduce");
    }
    bool alltoallv_u64(const uint64_t* send, const size_t* scount, const size_t* sdispl, uint64_t* recv, const size_t* rcount,
                       const size_t* rdispl, hipStream_t s) override {
        ++n_alltoall;
        return ok(rccl().AllToAllv(send, scount, sdispl, recv, rcount, rdispl, kNcclUint64, comm, s), "ncclAllToAllv");
    }
    bool gatherv_bytes(const void* send, size_t n, void* recv, const size_t* count, const size_t* displ, int root, hipStream_t s) override {
        ++n_gather;
        if (!ok(rccl().GroupStart(), "ncclGroupStart")) return false;
        bool good = true;
        if (n) good = ok(rccl().Send(send, n, kNcclUint8, root, comm, s), "ncclSend");
        if (good && rank == root)
            for (int r = 0; r < world && good; ++r)
                if (count[r]) good = ok(rccl().Recv((char*)recv + displ[r], count[r], kNcclUint8, r, comm, s), "ncclRecv");
        const bool ended = ok(rccl().GroupEnd(), "ncclGroupEnd");
        return good && ended;
    }
};

// ---- ranks as threads of one process: collectives as device-to-device copies around a barrier ----
struct ThreadGroup {
    int world = 1;
    std::atomic<int> arrived{0};
    std::atomic<uint64_t> generation{0};
    std::vector<const void*> ptr;      // what every rank published for the collective in progress
    std::vector<const size_t*> cnt, dsp;
    std::vector<std::vector<uint64_t>> host;  // allreduce staging
    std::vector<uint64_t*> hptr;       // host all-reduce: every rank's words
    std::atomic<bool> failed{false};   // a collective of the run in progress went wrong on some rank
    std::atomic<bool> aborted{false};  // a rank left bdx_dist_run early: nobody may wait for it at a barrier
    explicit ThreadGroup(int w) : world(w), ptr(w), cnt(w), dsp(w), host(w), hptr(w) {}
    // false: the group was aborted (by a rank that gave up between two collectives) -- the caller fails its collective.
    // The ranks spin (a run's barriers are microseconds apart and a condition variable's wake-up cost 30-50 us each, twice per
    // collective: the larger part of round 5's 0.08-0.17 ms per all-reduce); a rank that has waited for a while yields its core.
    bool barrier() {
        if (aborted.load(std::memory_order_acquire)) return false;
        const uint64_t g = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == world) {
            arrived.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
            return true;
        }
        for (uint32_t spin = 0;; ++spin) {
            if (generation.load(std::memory_order_acquire) != g) return true;
            if (aborted.load(std::memory_order_acquire)) return generation.load(std::memory_order_acquire) != g;
            if (spin < 20000u) __builtin_ia32_pause(); else std::this_thread::yield();
        }
    }
    void abort() { aborted.store(true, std::memory_order_release); }
    // every rank calls this at the start of a run, before its first collective: the flags of the previous run are history.
    // (Two barriers: nobody clears while somebody may still be reading, nobody proceeds before the flags are clear.)
    void begin_run(int rank) {
        if (aborted.load()) return;  // an aborted group stays aborted: its ranks are out of step for good
        barrier();
        if (rank == 0) failed.store(false);
        barrier();
    }
};

struct ThreadComm : Comm {
    std::shared_ptr<ThreadGroup> g;
    bool hip(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        err = std::string(what) + ": " + hipGetErrorString(e);
        g->failed.store(true);
        return false;
    }
    bool gone() { err = "another rank left the run"; return false; }
    void begin_run() override { n_allreduce = n_alltoall = n_gather = 0; g->begin_run(rank); }
    void abort() override { g->abort(); }
    bool allreduce_host_u64(uint64_t* v, size_t n, bool* done) override {
        *done = true;
        ++n_allreduce;
        g->hptr[rank] = v;
        if (!g->barrier()) return gone();
        std::vector<uint64_t>& sum = g->host[rank];
        sum.assign(n, 0);
        for (int r = 0; r < world; ++r)
            for (size_t i = 0; i < n; ++i) sum[i] += g->hptr[r][i];
        if (!g->barrier()) return gone();  // (everybody has read everybody's words)
        memcpy(v, sum.data(), n * 8);
        return !g->failed;
    }
    bool allreduce_u64(uint64_t* dev, size_t n, hipStream_t s) override {
        ++n_allreduce;
        std::vector<uint64_t>& mine = g->host[rank];
        mine.resize(n);
        bool good = hip(hipMemcpyAsync(mine.data(), dev, n * 8, hipMemcpyDeviceToHost, s), "hipMemcpyAsync") && hip(hipStreamSynchronize(s), "sync");
        if (!g->barrier()) return gone();
        std::vector<uint64_t> sum(n, 0);
        for (int r = 0; r < world; ++r)
            for (size_t i = 0; i < n && i < g->host[r].size(); ++i) sum[i] += g->host[r][i];
        if (!g->barrier()) return gone();  // (everybody has read everybody's words)
        good = good && hip(hipMemcpyAsync(dev, sum.data(), n * 8, hipMemcpyHostToDevice, s), "hipMemcpyAsync") && hip(hipStreamSynchronize(s), "sync");
        return good && !g->failed;
    }
    bool alltoallv_u64(const uint64_t* send, const size_t* scount, const size_t* sdispl, uint64_t* recv, const size_t* rcount,
                       const size_t* rdispl, hipStream_t s) override {
        ++n_alltoall;
        bool good = hip(hipStreamSynchronize(s), "sync");  // the send buffer is complete
        g->ptr[rank] = send; g->cnt[rank] = scount; g->dsp[rank] = sdispl;
        if (!g->barrier()) return gone();
        for (int r = 0; r < world && good; ++r) {
            const size_t n = g->cnt[r][rank];
            if (n != rcount[r]) { err = "all-to-all counts disagree"; g->failed.store(true); good = false; break; }
            if (n) good = hip(hipMemcpyAsync(recv + rdispl[r], (const uint64_t*)g->ptr[r] + g->dsp[r][rank], n * 8, hipMemcpyDefault, s), "hipMemcpyAsync");
        }
        good = good && hip(hipStreamSynchronize(s), "sync");
        if (!g->barrier()) return gone();  // (the send buffers may be reused)
        return good && !g->failed;
    }
    bool gatherv_bytes(const void* send, size_t n, void* recv, const size_t* count, const size_t* displ, int root, hipStream_t s) override {
        ++n_gather;
        bool good = hip(hipStreamSynchronize(s), "sync");
        g->ptr[rank] = send;
        if (!g->barrier()) return gone();
        if (rank == root) {
            for (int r = 0; r < world && good; ++r)
                if (count[r]) good = hip(hipMemcpyAsync((char*)recv + displ[r], g->ptr[r], count[r], hipMemcpyDefault, s), "hipMemcpyAsync");
            good = good && hip(hipStreamSynchronize(s), "sync");
        }
        (void)n;
        if (!g->barrier()) return gone();
        return good && !g->failed;
    }
};

}  // namespace

struct bdx_dist {
    int device = 0, ntids = 0, nlibs = 0, nbams = 0, nkeys = 0, w0 = 0;
    bdx_opts opts{};
    std::vector<bdx_lib> libs;
    std::unique_ptr<Comm> comm;
    bdx_ctx* reads = nullptr;        // ALL chromosomes this rank owns, one after the other in ascending order: one launch sequence
    bdx_ctx* util = nullptr;         // rank 0: holds the result
    int last_tid = -1;               // bdx_dist_chromosome: the chromosomes must be fed in ascending order
    size_t sorted_n = 0;             // the reads whose reference-id column has been checked (check_order): this many ...
    const void* sorted_ptr = nullptr;  // ... at this address
    size_t n_at_last = 0;
    bool use_check = false;
    DevBuf b_words, b_tab, b_send, b_recv, b_pack, b_all, b_nsend, b_nrecv, b_ntab, b_nflag, b_foreign, b_rg_rec, b_rg_pk, b_x, b_merge, b_chk, b_bucket;
    PinBuf h_words;                  // staging of the all-reduces' words (pinned: the copies either side of a collective are asynchronous)
    PinBuf h_tab;                    // what the small kernels report: per-chromosome tables, counts, ready words
    hipEvent_t ev_side = nullptr;    // the name census runs on the context's second stream, beside the joins
    uint32_t seq = 0;
    std::string err;
    uint64_t ctx_sent = 0, ctx_received = 0, gathered_bytes = 0;
    float ms_total = 0, ms_exchange = 0;
    bool ran = false;
    float phase_ms[kDistPhases] = {0};   // bdx_dist_get_phase_ms: where the last run's time went on this rank
    bool table_lent = false;         // (one rank) the result context holds this rank's pinned table buffers until the next run
    bool collect_support = false;    // bdx_dist_set_collect_support: the supporting reads of every SV (-g / -d) come with the result
};

namespace {

int dfail(bdx_dist* d, int code, const std::string& msg) {
    if (d) d->err = msg;
    return code;
}
#define DHIP(d, expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return dfail(d, BDX_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define DCTX(d, c, expr)                                                                     \
    do {                                                                                     \
        const int _rc = (expr);                                                              \
        if (_rc != BDX_OK) return dfail(d, _rc, std::string(#expr) + ": " + (c)->err);       \
    } while (0)

// host vector -> device words -> all-reduce -> host vector
int allreduce_host(bdx_dist* d, std::vector<uint64_t>& v, hipStream_t s) {
    if (v.empty()) return BDX_OK;
    // (one rank: the sum is the vector itself -- no copies, no collective, and above all no wait for the stream: the device walk that is
    // enqueued when the host's share is agreed on keeps running beside the host's walk, as in bdx_run)
    if (d->comm->world == 1) return BDX_OK;
    {   // (ranks of one process: the words never leave the host)
        bool done = false;
        if (!d->comm->allreduce_host_u64(v.data(), v.size(), &done)) return dfail(d, BDX_EHIP, d->comm->err);
        if (done) return BDX_OK;
    }
    DHIP(d, d->b_words.ensure(v.size() * 8));
    DHIP(d, d->h_words.ensure(v.size() * 8));
    memcpy(d->h_words.p, v.data(), v.size() * 8);
    DHIP(d, hipMemcpyAsync(d->b_words.p, d->h_words.p, v.size() * 8, hipMemcpyHostToDevice, s));
    if (!d->comm->allreduce_u64(d->b_words.as<uint64_t>(), v.size(), s)) return dfail(d, BDX_EHIP, d->comm->err);
    DHIP(d, hipMemcpyAsync(d->h_words.p, d->b_words.p, v.size() * 8, hipMemcpyDeviceToHost, s));
    DHIP(d, hipStreamSynchronize(s));
    memcpy(v.data(), d->h_words.p, v.size() * 8);
    return BDX_OK;
}

int dist_create_common(bdx_dist** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids, int w0, int device,
                       std::unique_ptr<Comm> comm) {
    if (!out || !opts || !libs || nlibs < 1 || nbams < 1 || ntids < 1) return BDX_EINVAL;
    if (comm->world >= kMaxRanks || ntids >= (1 << 24) - 1) return BDX_ELIMIT;   // (one table package per rank and one for rank 0's result context: k9_merge_tables)
    bdx_dist* d = new (std::nothrow) bdx_dist;
    if (!d) return BDX_ENOMEM;
    d->device = device; d->ntids = ntids; d->nlibs = nlibs; d->nbams = nbams; d->w0 = w0;
    d->opts = *opts;
    d->libs.assign(libs, libs + nlibs);
    d->nkeys = opts->cn_lib ? nlibs : nbams;
    d->comm = std::move(comm);
    int rc = bdx_create(&d->reads, opts, libs, nlibs, nbams, ntids, w0, device);
    if (rc == BDX_OK && d->comm->rank == 0) rc = bdx_create(&d->util, opts, libs, nlibs, nbams, ntids, w0, device);
    if (rc != BDX_OK) { if (d->reads) bdx_destroy(d->reads); delete d; return rc; }
    d->reads->force_direct_join = true;
    if (hipEventCreateWithFlags(&d->ev_side, hipEventDisableTiming) != hipSuccess) { bdx_dist_destroy(d); return BDX_EHIP; }
    *out = d;
    return BDX_OK;
}

inline size_t nkeys_words(int nkeys) { return (size_t)nkeys + 1; }

// pinned report area of a rank: ready words, then the tables the small kernels write
struct TabLayout {
    size_t flags = 0, tidtab = 0, first = 0, rtab = 0, cnts = 0, misc = 0, up_off = 0, up_tail = 0, up_misc = 0, up_cur = 0, up_stats = 0, up_counts = 0, words = 0;
    TabLayout(int ntids, int ncols, int nkeys, int ncnt, int world) {
        size_t o = 16;
        tidtab = o; o += (size_t)(ntids + 1) * (1 + ncols) + 2;
        first = o; o += (size_t)ntids * 4;
        rtab = o; o += (size_t)ntids + 3;
        cnts = o; o += (size_t)ntids + 2 * (size_t)world;   // (inter-chromosomal reads by mate chromosome | census records by owner)
        misc = o; o += 16;
        // staging of the small tables that go UP to the device: pinned, so that the copies are asynchronous and the vectors they were
        // built in need not outlive them (every table has its own place: nothing is overwritten within a run)
        up_off = o; o += (size_t)ntids * (1 + nkeys);
        up_tail = o; o += (size_t)ntids * 4;
        up_misc = o; o += (size_t)3 * ntids + 1;
        up_cur = o; o += (size_t)2 * world;
        up_stats = o; o += (size_t)2 + ncnt + 64;
        up_counts = o; o += sizeof(StageCounts) / 4;
        words = o;
    }
};

// spin on a pinned word a one-thread kernel sets behind the kernels whose results it announces
bool wait_word(volatile uint32_t* w, uint32_t value) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        if (*w == value) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
        __builtin_ia32_pause();
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) return false;
    }
}

}  // namespace

extern "C" {

int bdx_dist_unique_id(bdx_unique_id* out) {
    if (!out) return BDX_EINVAL;
    if (!rccl().load()) return BDX_EHIP;
    return rccl().GetUniqueId(out) == 0 ? BDX_OK : BDX_EHIP;
}

int bdx_dist_create(bdx_dist** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids,
                    int max_read_window_size0, int device, int rank, int world, const bdx_unique_id* id) {
    if (!out || world < 1 || rank < 0 || rank >= world || !id) return BDX_EINVAL;
    if (!rccl().load()) return BDX_EHIP;
    if (hipSetDevice(device) != hipSuccess) return BDX_EHIP;
    std::unique_ptr<RcclComm> c(new RcclComm);
    c->rank = rank; c->world = world;
    if (const int nrc = rccl().CommInitRank(&c->comm, world, *id, rank)) {   // (no handle yet to keep the message in)
        fprintf(stderr, "[bdx] ncclCommInitRank(rank %d of %d, device %d) failed: %s (%d)\n", rank, world, device,
                rccl().GetErrorString ? rccl().GetErrorString(nrc) : "RCCL error", nrc);
        return BDX_EHIP;
    }
    return dist_create_common(out, opts, libs, nlibs, nbams, ntids, max_read_window_size0, device, std::move(c));
}

int bdx_dist_create_threads(bdx_dist** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids,
                            int max_read_window_size0, const int* devices, int world) {
    if (!out || !devices || world < 1) return BDX_EINVAL;
    std::shared_ptr<ThreadGroup> g(new ThreadGroup(world));
    for (int r = 0; r < world; ++r) out[r] = nullptr;
    // every rank's context (device set-up, streams, pinned words) on its own thread: side by side on different devices
    std::vector<int> rcs(world, BDX_OK);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            std::unique_ptr<ThreadComm> c(new ThreadComm);
            c->rank = r; c->world = world; c->g = g;
            rcs[r] = dist_create_common(&out[r], opts, libs, nlibs, nbams, ntids, max_read_window_size0, devices[r], std::move(c));
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < world; ++r)
        if (rcs[r] != BDX_OK) {
            const int rc = rcs[r];
            for (int q = 0; q < world; ++q) { if (out[q]) bdx_dist_destroy(out[q]); out[q] = nullptr; }
            return rc;
        }
    return BDX_OK;
}

void bdx_dist_destroy(bdx_dist* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->reads) bdx_destroy(d->reads);
    if (d->util) bdx_destroy(d->util);
    for (DevBuf* b : {&d->b_words, &d->b_tab, &d->b_send, &d->b_recv, &d->b_pack, &d->b_all, &d->b_nsend, &d->b_nrecv, &d->b_ntab, &d->b_nflag, &d->b_foreign,
                      &d->b_rg_rec, &d->b_rg_pk, &d->b_x, &d->b_merge, &d->b_chk, &d->b_bucket})
        b->release();
    d->h_tab.release();
    d->h_words.release();
    if (d->ev_side) (void)hipEventDestroy(d->ev_side);
    delete d;
}

const char* bdx_dist_last_error(const bdx_dist* d) { return d ? d->err.c_str() : ""; }
int bdx_dist_rank(const bdx_dist* d) { return d ? d->comm->rank : -1; }
int bdx_dist_world(const bdx_dist* d) { return d ? d->comm->world : 0; }

// One context per rank takes all of the rank's chromosomes; the handle is the same for every tid.  What the caller must keep to
// is the order: a rank's chromosomes ascending, each chromosome's records together (the position-sorted stream gives both).
bdx_ctx* bdx_dist_chromosome(bdx_dist* d, int tid) {
    if (!d || tid < 0 || tid >= d->ntids) return nullptr;
    bdx_ctx* c = d->reads;
    if (tid < d->last_tid && c->n != d->n_at_last) {
        d->err = "bdx_dist_chromosome: a rank's chromosomes are fed in ascending order";
        return nullptr;
    }
    if (tid != d->last_tid) { d->last_tid = tid; }
    d->n_at_last = c->n;
    return c;
}

int bdx_dist_reset_reads(bdx_dist* d) {
    if (!d) return BDX_EINVAL;
    const int rc = bdx_reset_reads(d->reads);
    if (rc != BDX_OK) return dfail(d, rc, d->reads->err);
    d->last_tid = -1;
    d->n_at_last = 0;
    d->ran = false;
    d->sorted_n = 0; d->sorted_ptr = nullptr;   // (the next set of reads may have this one's count and address: its order is checked again)
    return BDX_OK;
}

namespace {
// The chromosome table is found by binary searches in the reference-id column (k9_tid_table_kernel): that column must be ascending with
// every id inside the header's sequences -- a rank's chromosomes fed in order, each chromosome's records together.  bdx_dist_chromosome
// sees to the order of the CALLS; what the batches hold is checked here, once per set of reads.
int check_order(bdx_dist* d) {
    bdx_ctx* c = d->reads;
    if (!c->n || (d->sorted_n == c->n && d->sorted_ptr == (const void*)c->d.tid)) return BDX_OK;
    DHIP(d, d->b_chk.ensure(64));
    uint32_t* err = d->b_chk.as<uint32_t>();
    uint32_t bad = 0;
    DHIP(d, hipMemsetAsync(err, 0, 4, c->stream));
    if (c->copy_pending) { DHIP(d, hipStreamWaitEvent(c->stream, c->ev_copy, 0)); }
    launch_k9_check_sorted((const int32_t*)c->d.tid, c->n, d->ntids, err, c->stream);
    DHIP(d, hipMemcpyAsync(&bad, err, 4, hipMemcpyDeviceToHost, c->stream));
    DHIP(d, hipStreamSynchronize(c->stream));
    if (bad) return dfail(d, BDX_EINVAL, "the reads of a rank are not in ascending order of their reference ids (or an id lies outside the header's sequences)");
    d->sorted_n = c->n; d->sorted_ptr = (const void*)c->d.tid;
    return BDX_OK;
}
}  // namespace

int bdx_dist_prepare(bdx_dist* d) {
    if (!d) return BDX_EINVAL;
    bdx_ctx* c = d->reads;
    DHIP(d, hipSetDevice(d->device));
    if (bdx_warm_up(d->device) != BDX_OK) return dfail(d, BDX_EHIP, "bdx_warm_up");
    {
        const int orc = check_order(d);
        if (orc != BDX_OK) return orc;
    }
    // the later stages' buffers for the prior a first run goes by (bdx_reserve does the same for a single context), with K6's
    // per-region arrays sized for the genome's regions rather than this rank's
    {
        const int ncols = 2 + d->nkeys, ncnt = d->nlibs * kNumFlags + d->nlibs + d->nbams, world = d->comm->world;
        const TabLayout L(d->ntids, ncols, d->nkeys, ncnt, world);
        DHIP(d, d->h_tab.ensure(L.words * 4));
        const size_t v1_words = (size_t)d->ntids * (ncols + 7 + (size_t)world) + ncnt + d->nbams + 11 * (size_t)world + (size_t)world * world + 64;   // (the first all-reduce's vector, the largest)
        DHIP(d, d->h_words.ensure(v1_words * 8));
        DHIP(d, d->b_words.ensure(v1_words * 8));
        DHIP(d, d->b_tab.ensure(((size_t)d->ntids * (nkeys_words(d->nkeys) + 12) + 8 * (size_t)world + 64) * 4));
        if (c->n && c->n < ((size_t)1 << 32)) {   // pass 1's tables for the reads that are loaded (a store that grew while it was filled lost them)
            const int rc = pass1_prepare(c, (uint32_t)((c->n + kTile - 1) / kTile));
            if (rc != BDX_OK) return dfail(d, rc, c->err);
            c->k1_live = false;
        }
    }
    if (c->n >= (1u << 20) && !c->ran) {
        const uint64_t prior = (uint64_t)c->n / 32 + 4096;
        if (prior <= kMaxAnomalous) {
            const uint32_t keep = c->k6_cap;
            c->k6_cap = (uint32_t)std::min<uint64_t>(prior, kMaxRegions);
            c->table_in_hbm = d->comm->world > 1;
            c->groups_in_hbm = d->comm->world > 1;
            const int rc = presize_stages_here(c, (uint32_t)prior);
            c->k6_cap = keep;
            if (rc != BDX_OK) return dfail(d, rc, c->err);
            const size_t nr = (size_t)prior;
            DHIP(d, d->b_rg_rec.ensure(nr * sizeof(RegionRec)));
            DHIP(d, d->b_rg_pk.ensure(nr * 2 * d->nkeys * 4));
            DHIP(d, d->b_x.ensure(nr + nr / 8 + 1024));   // (taint bytes: one per region / anomalous read of the upper bound)
            const size_t nn = (size_t)prior;
            DHIP(d, d->b_ntab.ensure((size_t)k7_names_slots(nn) * 16));
            DHIP(d, d->b_pack.ensure(nn * 40)); DHIP(d, d->b_all.ensure(nn * 40));
            DHIP(d, d->b_foreign.ensure(nn * 20 / 8 + 64));
            DHIP(d, d->b_send.ensure(nn * 20 + 4096)); DHIP(d, d->b_recv.ensure(nn * 20 + 4096));   // (a census record of 16 bytes per anomalous read; an eighth of them inter-chromosomal and travelling, 32 bytes each)
            if (d->comm->rank == 0) {   // the genome's region table in pinned memory (regions: about a tenth of the anomalous reads)
                const size_t nreg = (size_t)c->n / 256 + 4096;
                DHIP(d, c->h_regs.ensure(nreg * sizeof(RegionRec)));
                DHIP(d, c->h_pk.ensure(nreg * 2 * d->nkeys * 4));
            }
        }
    }
    return BDX_OK;
}

int bdx_dist_get_phase_ms(const bdx_dist* d, float* out, int n) {
    if (!d || !out) return BDX_EINVAL;
    for (int i = 0; i < n; ++i) out[i] = i < kDistPhases ? d->phase_ms[i] : 0.0f;
    return BDX_OK;
}

const char* bdx_dist_phase_name(int i) {
    static const char* names[kDistPhases] = {
        "pass1_compaction_first_reads", "allreduce_statistics_first_reads_exchange_counts", "", "", "rebase_and_region_cut", "allreduce_regions",
        "globalize_and_pack", "alltoall_join_records_census_windows", "", "",
        "joins_components_walk_table", "allreduce_package_sizes", "", "gather_packages_to_rank0", "rank0_only_merge", "replay_route",
        "rank0_only_host_walk", "rank0_only_device_walk_of_gathered_groups"};
    return i >= 0 && i < kDistPhases ? names[i] : "";
}

int bdx_dist_set_collect_support(bdx_dist* d, int on) {
    if (!d) return BDX_EINVAL;
    d->collect_support = on != 0;
    return BDX_OK;
}

bdx_ctx* bdx_dist_result(bdx_dist* d) { return d && d->ran && d->comm->rank == 0 ? d->util : nullptr; }

// a test / measurement switch (bdx_set_debug's names) for the contexts of this rank: its own and, on rank 0, the result context
int bdx_dist_set_debug(bdx_dist* d, const char* name, int value) {
    if (!d || !name) return BDX_EINVAL;
    int rc = d->reads ? bdx_set_debug(d->reads, name, value) : BDX_OK;
    if (rc == BDX_OK && d->util) rc = bdx_set_debug(d->util, name, value);
    return rc;
}

int bdx_dist_get_exchange(const bdx_dist* d, uint64_t* ctx_records_sent, uint64_t* ctx_records_received, uint64_t* gathered_bytes,
                          float* ms_total, float* ms_exchange) {
    if (!d) return BDX_EINVAL;
    if (!d->ran) return BDX_ESTATE;
    if (ctx_records_sent) *ctx_records_sent = d->ctx_sent;
    if (ctx_records_received) *ctx_records_received = d->ctx_received;
    if (gathered_bytes) *gathered_bytes = d->gathered_bytes;
    if (ms_total) *ms_total = d->ms_total;
    if (ms_exchange) *ms_exchange = d->ms_exchange;
    return BDX_OK;
}

int bdx_dist_get_collectives(const bdx_dist* d, uint32_t out[3], const char** backend, int* rccl_version) {
    if (!d) return BDX_EINVAL;
    if (!d->ran) return BDX_ESTATE;
    if (out) { out[0] = d->comm->n_allreduce; out[1] = d->comm->n_alltoall; out[2] = d->comm->n_gather; }
    const bool is_rccl = dynamic_cast<const RcclComm*>(d->comm.get()) != nullptr;
    if (backend) *backend = is_rccl ? rccl().path.c_str() : "threads";
    if (rccl_version) *rccl_version = is_rccl ? rccl().version : 0;
    return BDX_OK;
}

int bdx_dist_owner(uint64_t name_key, int world) { return world > 0 ? (int)exchange_owner(name_key, (uint32_t)world) : -1; }

// longest-processing-time packing: chromosomes in descending weight, each onto the least loaded rank (ties: lower rank)
int bdx_dist_plan(const uint64_t* weight, int ntids, int world, int* rank_of_tid) {
    if (!weight || !rank_of_tid || ntids < 0 || world < 1) return BDX_EINVAL;
    std::vector<int> order(ntids);
    for (int i = 0; i < ntids; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight[a] > weight[b]; });
    std::vector<uint64_t> load(world, 0);
    for (int t : order) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[r] < load[best]) best = r;
        rank_of_tid[t] = best;
        load[best] += weight[t];
    }
    return BDX_OK;
}

// A failure that only one rank sees (its own data, its own device) must not make that rank leave while the others enter
// the next collective: they would wait for it for ever.  Every stage between two collectives therefore runs as a local
// phase whose status travels with the next all-reduce (one word per rank behind the payload); all ranks look at the
// summed words and give up together.  What cannot be folded (a device allocation failing between the size exchange and
// the all-to-all) aborts the communicator of the thread backend, which wakes its waiters with an error.
struct RunStatus {
    int rc = BDX_OK;      // this rank's first failure
    std::string msg;
};

static int agreed_failure(bdx_dist* d, const RunStatus& st, const uint64_t* words, int world) {
    int first = -1, code = BDX_OK;
    for (int q = 0; q < world; ++q)
        if (words[q]) { first = q; code = (int)words[q]; break; }
    if (first < 0) return BDX_OK;
    if (st.rc != BDX_OK) return dfail(d, st.rc, st.msg);
    return dfail(d, code, "rank " + std::to_string(first) + " failed (" + bdx_strerror(code) + "); all ranks stop");
}

// the result of a rank's context becomes the result context's: the pinned buffers the device assembled the table in change
// hands (no copy), the counters are taken over
static void swap_table_buffers(bdx_ctx* U, bdx_ctx* C) {
    std::swap(U->h_sv_out, C->h_sv_out); std::swap(U->h_lib_index, C->h_lib_index); std::swap(U->h_lib_pairs, C->h_lib_pairs);
    std::swap(U->h_cn_key, C->h_cn_key); std::swap(U->h_cn_value, C->h_cn_value); std::swap(U->h_ltail_dev, C->h_ltail_dev);
}
static void adopt_table(bdx_ctx* U, bdx_ctx* C) {
    swap_table_buffers(U, C);
    std::swap(U->walk, C->walk); std::swap(U->log_tail, C->log_tail);
    U->materialized = C->materialized; U->rows_packed = C->rows_packed;
    U->n_sv_total = C->n_sv_total; U->n_terms_total = C->n_terms_total; U->n_cn_total = C->n_cn_total; U->n_printed = C->n_printed;
    U->n_sv_host = C->n_sv_host; U->n_groups_total = C->n_groups_total;
    U->counts = C->counts;
    C->k6 = K6Arrays{};   // (its pointers into the swapped buffers are history)
}

int bdx_dist_run(bdx_dist* d) {
    if (!d) return BDX_EINVAL;
    const auto t_begin = std::chrono::steady_clock::now();
    Comm& comm = *d->comm;
    const int world = comm.world, rank = comm.rank;
    const int nlibs = d->nlibs, nbams = d->nbams, nkeys = d->nkeys, ntids = d->ntids;
    const int ncnt = nlibs * kNumFlags + nlibs + nbams, ncols = 2 + nkeys, nkeys2 = 2 * nkeys;
    DHIP(d, hipSetDevice(d->device));
    bdx_ctx* C = d->reads;
    bdx_ctx* U = d->util;
    if (C->sizing.load(std::memory_order_acquire)) return dfail(d, BDX_ESTATE, "the buffers of the later stages are being sized on another thread (bdx_bamdec_finish has not returned)");
    hipStream_t s = C->stream;
    d->ran = false;
    d->ctx_sent = d->ctx_received = d->gathered_bytes = 0;
    if (d->table_lent && U) {   // (the previous result is history: its buffers go back to the context that fills them)
        swap_table_buffers(U, C);
        U->ran = false;
        d->table_lent = false;
    }
    if (U) { U->reg = nullptr; U->nreg = 0; U->rpk = nullptr; }
    ++d->seq;
    comm.begin_run();
    RunStatus st;
    // a phase: local work between two collectives; its failure is recorded, not returned.  Phases and collectives alternate:
    // phase_ms[2k] = the k-th phase, phase_ms[2k + 1] = the collective behind it (which includes waiting for the slowest rank)
    int n_phase = 0;
    for (float& x : d->phase_ms) x = 0;
    auto phase = [&](const std::function<int()>& body) {
        const auto tp = std::chrono::steady_clock::now();
        if (st.rc == BDX_OK) {
            d->err.clear();
            const int rc = body();
            if (rc != BDX_OK) { st.rc = rc; st.msg = d->err; }
        }
        if (2 * n_phase < kDistPhases) d->phase_ms[2 * n_phase] += ms_between(tp, std::chrono::steady_clock::now());
    };
    struct Stamp {
        bdx_dist* d; int& n; std::chrono::steady_clock::time_point t;
        ~Stamp() { if (2 * n + 1 < kDistPhases) d->phase_ms[2 * n + 1] += ms_between(t, std::chrono::steady_clock::now()); ++n; }
    };
    // all-reduce of v with the ranks' status words appended; afterwards every rank knows whether anybody failed
    auto exchange = [&](std::vector<uint64_t>& v) -> int {
        Stamp stamp{d, n_phase, std::chrono::steady_clock::now()};
        const size_t at = v.size();
        v.resize(at + (size_t)world, 0);
        v[at + (size_t)rank] = (uint64_t)st.rc;
        const int rc = allreduce_host(d, v, s);
        if (rc != BDX_OK) { comm.abort(); return rc; }
        const int f = agreed_failure(d, st, &v[at], world);
        v.resize(at);
        return f;
    };
    auto leave = [&](int rc) { comm.abort(); return rc; };  // failures past the last foldable point
    // BDX_DIST_TRACE=1 (a measurement / debugging aid, like BDX_ALLOC_TRACE): waits for the stream at every step and names it on stderr
    static const bool tracing = getenv("BDX_DIST_TRACE") != nullptr;
    auto trace = [&](const char* what) {
        if (!tracing) return;
        const hipError_t e = hipStreamSynchronize(s);
        fprintf(stderr, "[bdx dist %d/%d] %s: %s (%.3f ms)\n", rank, world, what, hipGetErrorString(e), ms_between(t_begin, std::chrono::steady_clock::now()));
    };

    // the report area (pinned) and the small device tables
    const TabLayout L(ntids, ncols, nkeys, ncnt, world);
    DHIP(d, d->h_tab.ensure(L.words * 4));
    uint32_t* H = d->h_tab.as<uint32_t>();
    volatile uint32_t* flags = (volatile uint32_t*)H;
    // device tables: tid_off [ntids][1 + nkeys] | tid_tail [ntids][4] | roff [ntids] | owner [ntids] | tid_start [ntids + 1] | cnt [4 world] | n_total [8] |
    // rbase u64 [ntids + 1] | inter-chromosomal reads by mate chromosome [ntids]
    const size_t o_off = 0, o_tail = o_off + (size_t)ntids * (1 + nkeys), o_roff = o_tail + (size_t)ntids * 4, o_owner = o_roff + ntids,
                 o_start = o_owner + ntids, o_cnt = o_start + ntids + 1, o_ntot = o_cnt + 4 * (size_t)world, o_rbase = (o_ntot + 8 + 1) / 2 * 2,
                 o_mt = o_rbase + 2 * ((size_t)ntids + 1), tab_words = o_mt + (size_t)ntids;
    DHIP(d, d->b_tab.ensure(tab_words * 4));
    uint32_t* T = d->b_tab.as<uint32_t>();
    if ((uint64_t)world * (uint64_t)ntids > (1u << 22)) return dfail(d, BDX_ELIMIT, "ranks x sequences beyond the first all-reduce's table (2^22 words)");   // (every rank alike)

    // ---- A: pass 1 over all of this rank's chromosomes, where each chromosome starts and what the counters read there; the compaction
    // (it needs nothing of the other ranks); what the compact records say before anything is known of the other ranks: every chromosome's
    // first anomalous read, the inter-chromosomal reads by mate chromosome, the census records by owner.  ONE all-reduce carries all of it
    // (round 5: three -- statistics, first reads + exchange counts, regions) ----
    const size_t tw = (size_t)ncols;  // per chromosome: anomalous reads, normal pairs, proper reads per key
    const size_t W2 = (size_t)world * world;
    const size_t at_tot = (size_t)ncnt + nbams, at_reads = at_tot + (size_t)ntids * tw, at_claim = at_reads + ntids, at_owner = at_claim + ntids,
                 at_first = at_owner + ntids, at_mt = at_first + (size_t)ntids * 3, at_cen = at_mt + (size_t)world * ntids, at_flags = at_cen + W2;
    std::vector<uint64_t> v1(at_flags + 3, 0);
    std::vector<uint32_t> tidtab((size_t)(ntids + 1) * (1 + ncols), 0);   // [t][0] first read, [t][1 + c] counters in front of it
    const bool solo = world == 1;   // nothing travels: no census, no counts, no all-to-all, no gather
    uint32_t na = 0;                // this rank's anomalous reads
    uint32_t* UP = H;               // (the staging places of L.up_*)
    phase([&]() -> int {
        C->table_in_hbm = world > 1;
        C->groups_in_hbm = world > 1;
        C->defer_walk = false;
        C->k6_cap = 0; C->k6_r_rec = nullptr; C->k6_r_pk = nullptr; C->k6_taint = nullptr; C->k3_tid_tail = nullptr;
        {   // (a set of reads bdx_dist_prepare has not seen: its order is checked before the chromosome table is searched in it)
            const int orc = check_order(d);
            if (orc != BDX_OK) return orc;
        }
        DCTX(d, C, do_pass1(C, 0, false, false));   // (its record is waited for together with the chromosome table)
        TidTableParams tp{};
        tp.tid = C->d.tid; tp.lib = C->d.lib; tp.cls = C->b_cls.as<uint8_t>(); tp.n = C->n; tp.ntiles = C->ntiles; tp.tstride = C->tstride;
        tp.ntids = ntids; tp.nkeys = nkeys; tp.nlibs = nlibs; tp.ncols = ncols; tp.libs = C->b_libs.as<DevLib>();
        tp.tile_tot = C->b_tile_tot.as<uint32_t>(); tp.tile_pre = C->b_tile_pre.as<uint32_t>(); tp.chunk_base = C->fp_deferred.chunk_base;
        tp.chunk_super = C->fp_deferred.chunk_super; tp.p1 = C->b_p1.as<Pass1>(); tp.out = H + L.tidtab;
        launch_k9_tid_table(tp, s);
        launch_k9_signal(H + 0, d->seq, s);
        DCTX(d, C, wait_pass1(C));
        if (!wait_word(flags + 0, d->seq)) {
            DHIP(d, hipStreamSynchronize(s));
            if (flags[0] != d->seq) return dfail(d, BDX_EINTERNAL, "the chromosome table did not arrive: its kernels were not launched");
        }
        trace("chromosome table");
        memcpy(tidtab.data(), H + L.tidtab, tidtab.size() * 4);
        const uint32_t* terr = H + L.tidtab + tidtab.size();
        if (terr[0] || terr[1]) return dfail(d, BDX_EINVAL, "record with a reference id outside [0, ntids)");
        na = C->p1.n_anom;
        // the compaction, and what its records say
        {
            UploadList ul{};
            ul.fill(T + o_cnt, 0u, (size_t)world * 4 + 8);   // (the exchange's counters, the error words of the later kernels)
            ul.fill(T + o_mt, 0u, (size_t)ntids);
            launch_k9_upload(ul, s);
        }
        DCTX(d, C, do_compact(C, 0, nullptr, true));
        trace("compaction");
        memset(H + L.first, 0, (size_t)ntids * 16);
        if (na) {
            FirstCountsParams fc{};
            fc.cp = C->cp; fc.mtid_col = C->d.mtid; fc.n_ptr = &C->b_p1.as<Pass1>()->n_anom; fc.ntids = ntids; fc.world = (uint32_t)world;
            fc.first_tab = H + L.first; fc.cnt_mtid = T + o_mt; fc.cnt_owner = T + o_cnt + world;
            launch_k9_first_counts(fc, na, s);
            if (solo) {
                launch_k9_signal(H + 1, d->seq, s);
            } else {
                UploadList rl{};   // (device words -> the pinned report area, then the ready word: one launch each)
                rl.copy(H + L.cnts, T + o_mt, (size_t)ntids);
                rl.copy(H + L.cnts + ntids, T + o_cnt + world, (size_t)world);
                launch_k9_upload(rl, s);
                launch_k9_signal(H + 1, d->seq, s);
            }
            if (!wait_word(flags + 1, d->seq)) {
                DHIP(d, hipStreamSynchronize(s));
                if (flags[1] != d->seq) return dfail(d, BDX_EINTERNAL, "the chromosomes' first reads did not arrive: its kernels were not launched");
            }
            trace("first reads and counts");
        } else {
            C->k4 = K4Arrays{};
        }
        for (int i = 0; i < ncnt; ++i) v1[i] = C->cnt_local[i];
        for (int b = 0; b < nbams; ++b) v1[ncnt + b] = C->p1.ref_len[b];
        for (int t = 0; t < ntids; ++t) {
            const uint32_t* a = &tidtab[(size_t)t * (1 + ncols)];
            const uint32_t* b = a + (1 + ncols);
            const uint64_t nreads = (uint64_t)b[0] - a[0];
            if (!nreads) continue;
            for (int c = 0; c < ncols; ++c) v1[at_tot + (size_t)t * tw + c] = (uint32_t)(b[1 + c] - a[1 + c]);
            v1[at_reads + t] = nreads;
            v1[at_claim + t] = 1;
            v1[at_owner + t] = (uint64_t)rank + 1;
            const uint32_t* f = H + L.first + (size_t)t * 4;
            if (na && f[0]) {   // {has one, its read length, normal pairs of THIS chromosome in front of it (the chromosomes before are added by everybody alike)}
                v1[at_first + (size_t)t * 3] = 1; v1[at_first + (size_t)t * 3 + 1] = f[1];
                v1[at_first + (size_t)t * 3 + 2] = (uint32_t)(f[2] - a[1 + kColNormal]);
            }
        }
        if (na && !solo) {
            for (int t = 0; t < ntids; ++t) v1[at_mt + (size_t)rank * ntids + t] = H[L.cnts + t];
            for (int q = 0; q < world; ++q) v1[at_cen + (size_t)rank * world + q] = H[L.cnts + ntids + q];
        }
        v1[at_flags] = d->collect_support ? 1 : 0;
        if (C->n) v1[at_flags + (C->use_check ? 1 : 2)] = 1;
        return BDX_OK;
    });
    int rc = exchange(v1);
    if (rc != BDX_OK) return rc;
    const uint64_t want_support = v1[at_flags];
    if (want_support != 0 && want_support != (uint64_t)world) return dfail(d, BDX_EINVAL, "bdx_dist_set_collect_support is set on some ranks only");
    if (v1[at_flags + 1] && v1[at_flags + 2]) return dfail(d, BDX_EINVAL, "bdx_use_name_check is set on some ranks' contexts only");
    const bool with_check = v1[at_flags + 1] != 0;
    std::vector<int32_t> owner(ntids, -1);
    for (int t = 0; t < ntids; ++t) {
        if (v1[at_claim + t] > 1) return dfail(d, BDX_EINVAL, "chromosome " + std::to_string(t) + " has reads on more than one rank");
        if (v1[at_claim + t]) owner[t] = (int32_t)v1[at_owner + t] - 1;
    }
    std::vector<uint64_t> read_base((size_t)ntids + 1, 0);   // a chromosome's first read in the merged stream (position sorted: chromosomes ascending)
    for (int t = 0; t < ntids; ++t) read_base[(size_t)t + 1] = read_base[t] + v1[at_reads + (size_t)t];
    std::vector<uint32_t> cnt_g(ncnt);
    for (int i = 0; i < ncnt; ++i) cnt_g[i] = (uint32_t)v1[i];
    uint32_t covered = 0;  // BamSummary.cpp:123-126: a uint32 maximum compared against each file's size_t sum
    for (int b = 0; b < nbams; ++b)
        if ((uint64_t)covered < v1[ncnt + b]) covered = (uint32_t)v1[ncnt + b];
    const int32_t window = window_from(C, cnt_g.data(), covered);
    auto tot = [&](int tid, int k) { return v1[at_tot + (size_t)tid * tw + k]; };
    std::vector<uint64_t> base((size_t)(ntids + 1) * tw, 0);  // exclusive prefix over the chromosomes in stream order
    for (int t = 0; t < ntids; ++t)
        for (size_t k = 0; k < tw; ++k) base[(size_t)(t + 1) * tw + k] = base[(size_t)t * tw + k] + tot(t, (int)k);
    const uint64_t na_all = base[(size_t)ntids * tw];
    // (the same sum on every rank: all of them return here, together)
    if (na_all > kMaxAnomalous) return dfail(d, BDX_ELIMIT, "more than 2^31 anomalous reads in one run");
    int last_anom_tid = -1;
    for (int t = 0; t < ntids; ++t)
        if (tot(t, 0) > 0) last_anom_tid = t;
    // every chromosome's first anomalous read with the genome's normal-pair count: {has one, read length, count}
    std::vector<uint64_t> v2((size_t)ntids * 3, 0);
    for (int t = 0; t < ntids; ++t)
        if (v1[at_first + (size_t)t * 3]) {
            v2[(size_t)t * 3] = 1; v2[(size_t)t * 3 + 1] = v1[at_first + (size_t)t * 3 + 1];
            v2[(size_t)t * 3 + 2] = (uint32_t)(base[(size_t)t * tw + kColNormal] + v1[at_first + (size_t)t * 3 + 2]);
        }
    // what the one all-to-all will carry, per destination: inter-chromosomal join records (to the owner of the mate's later chromosome),
    // census records (to the owner of the name key), window read lengths (below, once the regions are counted)
    std::vector<uint32_t> h_cnt(world, 0), h_ncnt(world, 0), r_cnt(world, 0), r_ncnt(world, 0);
    if (!solo)
        for (int q = 0; q < world; ++q) {
            for (int t = 0; t < ntids; ++t) {
                if (owner[t] < 0) continue;
                if (owner[t] == q && q != rank) h_cnt[q] += (uint32_t)v1[at_mt + (size_t)rank * ntids + t];
                if (owner[t] == rank && q != rank) r_cnt[q] += (uint32_t)v1[at_mt + (size_t)q * ntids + t];
            }
            h_ncnt[q] = (uint32_t)v1[at_cen + (size_t)rank * world + q];
            r_ncnt[q] = (uint32_t)v1[at_cen + (size_t)q * world + rank];
        }
    size_t nsend = 0, nrecv = 0, nnrecv = 0;
    for (int q = 0; q < world; ++q) { nsend += h_cnt[q]; nrecv += r_cnt[q]; nnrecv += r_ncnt[q]; }
    {   // (the same matrices on every rank: the limits trip everywhere at once)
        for (int r = 0; r < world && !solo; ++r) {
            uint64_t col = 0, ncol = 0;
            for (int q = 0; q < world; ++q) {
                ncol += v1[at_cen + (size_t)q * world + r];
                for (int t = 0; t < ntids; ++t)
                    if (owner[t] == r && q != r) col += v1[at_mt + (size_t)q * ntids + t];
            }
            if (col > (1u << 28) || ncol > 0x7FFFFFFFull) return dfail(d, BDX_ELIMIT, "too many inter-chromosomal join records / names on one rank");
        }
    }

    // ---- B: the statistics every kernel from here on runs with; the compact records' counters start where the chromosomes in front
    // (anybody's) left them; regions -- the first anomalous read of the next chromosome that has one closes a chromosome's last candidate.
    // The second all-reduce: every chromosome's region count ----
    std::vector<uint64_t> v3((size_t)ntids + 1, 0);
    std::vector<uint32_t> rtab((size_t)ntids + 3, 0);   // this rank's regions: first region of every chromosome, count, last_maxq
    n_phase = 2;   // (slots 4 / 5)
    phase([&]() -> int {
        DCTX(d, C, set_pass1(C, cnt_g.data(), covered, window, false));
        {
            uint32_t* st_up = UP + L.up_stats;
            st_up[0] = covered; st_up[1] = (uint32_t)window;
            memcpy(st_up + 2, cnt_g.data(), (size_t)ncnt * 4);
            memcpy(st_up + 2 + ncnt, C->key_density.data(), C->key_density.size() * 4);
            // (window and covered length -- Pass1's first words --, flag histogram, densities; with anomalous reads the chromosomes' offsets, owners
            // and closing reads: ONE launch that reads them from the pinned report area)
            UploadList ul{};
            ul.copy(C->b_p1.p, st_up, 2);
            ul.copy(C->b_cnt.p, st_up + 2, (size_t)ncnt);
            ul.copy(C->b_kdens.p, st_up + 2 + ncnt, C->key_density.size());
            if (na) {
                uint32_t* up = UP + L.up_off;
                for (int t = 0; t < ntids; ++t) {
                    const uint32_t* a = &tidtab[(size_t)t * (1 + ncols)];
                    for (int k = 0; k < 1 + nkeys; ++k) up[(size_t)t * (1 + nkeys) + k] = (uint32_t)base[(size_t)t * tw + 1 + k] - a[1 + 1 + k];
                }
                ul.copy(T + o_off, up, (size_t)ntids * (1 + nkeys));
                uint32_t* um = UP + L.up_misc;   // owner [ntids] | first read of every chromosome in this context [ntids + 1]   (roff follows later)
                for (int t = 0; t < ntids; ++t) { um[t] = (uint32_t)owner[t]; um[(size_t)ntids + t] = tidtab[(size_t)t * (1 + ncols)]; }
                um[(size_t)2 * ntids] = tidtab[(size_t)ntids * (1 + ncols)];
                ul.copy(T + o_owner, um, (size_t)2 * ntids + 1);
                uint32_t* tails = UP + L.up_tail;
                int nx = -1;
                for (int t = ntids - 1; t >= 0; --t) {
                    uint32_t* q = &tails[(size_t)t * 4];
                    q[0] = nx >= 0 ? 1u : 0u;
                    q[1] = nx >= 0 ? (uint32_t)v2[(size_t)nx * 3 + 1] : 0u;
                    q[2] = nx >= 0 ? (uint32_t)v2[(size_t)nx * 3 + 2] : (uint32_t)base[(size_t)ntids * tw + 1];
                    q[3] = 0;
                    if (v2[(size_t)t * 3]) nx = t;
                }
                ul.copy(T + o_tail, tails, (size_t)ntids * 4);
            }
            launch_k9_upload(ul, s);
        }
        if (!na) return BDX_OK;
        launch_k9_rebase(C->cp, &C->b_p1.as<Pass1>()->n_anom, na, nkeys, T + o_off, nullptr, s);
        C->k3_tid_tail = T + o_tail;
        DCTX(d, C, do_cut(C, 0, 0, 0, false, true));
        trace("region cut");
        launch_k9_tid_regions(C->b_r_rec.as<RegionRec>(), C->b_counts.as<StageCounts>(), ntids, H + L.rtab, s);
        launch_k9_signal(H + 2, d->seq, s);
        if (!wait_word(flags + 2, d->seq)) {
            DHIP(d, hipStreamSynchronize(s));
            if (flags[2] != d->seq) return dfail(d, BDX_EINTERNAL, "the chromosomes' region counts did not arrive: its kernels were not launched");
        }
        memcpy(rtab.data(), H + L.rtab, rtab.size() * 4);
        for (int t = 0; t < ntids; ++t) v3[t] = rtab[t + 1] - rtab[t];
        if (last_anom_tid >= 0 && owner[last_anom_tid] == rank) v3[ntids] = rtab[ntids + 2];
        return BDX_OK;
    });
    rc = exchange(v3);
    if (rc != BDX_OK) return rc;
    std::vector<uint64_t> rbase(ntids + 1, 0);
    for (int t = 0; t < ntids; ++t) rbase[t + 1] = rbase[t] + v3[t];
    const uint64_t NR = rbase[ntids];
    if (NR > kMaxRegions) return dfail(d, BDX_ELIMIT, "too many regions for the packed group key");  // (all ranks alike)
    const int32_t lm = (int32_t)(uint32_t)v3[ntids];
    const uint32_t nr_local = rtab[ntids + 1];
    std::vector<uint64_t> nr_of_rank(world, 0);
    for (int t = 0; t < ntids; ++t)
        if (owner[t] >= 0) nr_of_rank[owner[t]] += v3[t];
    const uint32_t period = (uint32_t)std::max(1, d->opts.buffer_size + 1);
    const uint32_t NW = (uint32_t)(NR / period);
    const uint32_t capG = (uint32_t)std::max<uint64_t>(std::max<uint64_t>(C->na_alloc, NR), 1);
    // flush windows whose last region (id (w + 1) period - 1) is rank q's: its read length is what every rank's walk uses at that flush
    std::vector<uint32_t> nwin(world, 0);
    for (int t = 0; t < ntids; ++t)
        if (owner[t] >= 0 && rbase[t + 1] > rbase[t]) nwin[owner[t]] += (uint32_t)(std::min<uint64_t>(rbase[t + 1], (uint64_t)NW * period) / period - std::min<uint64_t>(rbase[t], (uint64_t)NW * period) / period);

    // the result context takes the run's statistics; with no region anywhere the run is over
    auto finish_result = [&]() -> int {
        d->ran = true;
        d->ms_total = ms_between(t_begin, std::chrono::steady_clock::now());
        return BDX_OK;
    };
    if (rank == 0) {
        DCTX(d, U, set_pass1(U, cnt_g.data(), covered, window, false));
        U->n = 0;
        U->p1 = Pass1{};
        U->p1.n_anom = (uint32_t)na_all; U->p1.covered_ref_len = covered; U->p1.window = window;
        U->replayed = false;
        U->sup_off.clear(); U->sup_idx.clear(); U->sup_flag.clear();
        U->collect_support = false;
        if (!U->walk_scratch) U->walk_scratch = walk_scratch_new();
    }
    if (!NR) {   // no region anywhere: an empty table
        if (rank == 0) {
            U->regions.clear(); U->r_pk.clear(); U->reg = nullptr; U->nreg = 0; U->rpk = nullptr;
            U->walk.clear(); U->log_tail.clear();
            U->n_sv_total = U->n_terms_total = U->n_cn_total = U->n_printed = U->n_sv_host = U->n_groups_total = 0;
            memset(&U->counts, 0, sizeof(U->counts));
            U->materialized = true; U->ran = true; U->stage = 4;
            if (want_support) { U->collect_support = true; U->sup_off.assign(1, 0); }   // (no SV, no supporting read: an empty list, not a missing one)
        }
        return finish_result();
    }

    const auto t_x0 = std::chrono::steady_clock::now();
    // ---- C: genome-wide region ids; ONE all-to-all carries, per destination, the inter-chromosomal join records (32 bytes), the census
    // records (16) and the read lengths of the flush windows this rank closes (8) -- round 5: two all-to-alls and, for the windows and the
    // taint bytes, an all-reduce of device words.  No host round trip from here to the rank's finished table ----
    constexpr size_t kxw = sizeof(ExchangeEntry) / 8;
    std::vector<size_t> scount(world, 0), sdispl(world, 0), rcount(world, 0), rdispl(world, 0);   // in 64-bit words
    size_t swords = 0, rwords = 0;
    for (int q = 0; q < world; ++q) {
        const size_t sw = (size_t)h_cnt[q] * kxw + (size_t)h_ncnt[q] * 2 + (q != rank ? nwin[rank] : 0);
        const size_t rw = (size_t)r_cnt[q] * kxw + (size_t)r_ncnt[q] * 2 + (q != rank ? nwin[q] : 0);
        scount[q] = sw; sdispl[q] = swords; swords += round_up(sw, 4);   // (blocks start on 32-byte boundaries: a join record is stored as one)
        rcount[q] = rw; rdispl[q] = rwords; rwords += round_up(rw, 4);
    }
    SegList seg_ctx{}, seg_cen{}, seg_win{};
    {
        uint32_t a = 0, b = 0, c = 0;
        seg_ctx.n = seg_cen.n = seg_win.n = world;
        for (int q = 0; q < world; ++q) {
            seg_ctx.off[q] = rdispl[q]; seg_ctx.start[q] = a; a += r_cnt[q];
            seg_cen.off[q] = rdispl[q] + (size_t)r_cnt[q] * kxw; seg_cen.start[q] = b; b += r_ncnt[q];
            seg_win.off[q] = rdispl[q] + (size_t)r_cnt[q] * kxw + (size_t)r_ncnt[q] * 2; seg_win.start[q] = c; c += q != rank ? nwin[q] : 0;
        }
        seg_ctx.start[world] = a; seg_cen.start[world] = b; seg_win.start[world] = c;
    }
    const uint32_t nwin_recv = seg_win.start[world];
    uint8_t* taint = nullptr;       // [capG] bytes: this rank's regions that are an end of a group formed on another rank, or of one formed here with another rank's region
    RegionRec* regs = nullptr;      // rank 0: the genome's region table (pinned: the copy runs beside the result context's walk)
    uint32_t* pk = nullptr;
    bool regs_pending = false;
    ExchangeSrc xs{};
    {
        const auto tp = std::chrono::steady_clock::now();
        auto body = [&]() -> int {
            DHIP(d, d->b_send.ensure(std::max<size_t>(swords, 4) * 8));
            DHIP(d, d->b_recv.ensure(std::max<size_t>(rwords, 4) * 8));
            DHIP(d, d->b_rg_rec.ensure((size_t)NR * sizeof(RegionRec)));
            DHIP(d, d->b_rg_pk.ensure(std::max<size_t>((size_t)NR * nkeys2 * 4, 16)));
            DHIP(d, C->b_out_deg.ensure((size_t)capG * 6 * 4));
            DHIP(d, d->b_x.ensure((size_t)capG + 64));
            taint = d->b_x.as<uint8_t>();
            DHIP(d, hipMemsetAsync(d->b_rg_rec.p, 0, (size_t)NR * sizeof(RegionRec), s));
            DHIP(d, hipMemsetAsync(taint, 0, (size_t)capG, s));
            {   // (the chromosomes' region offsets and the exchange's cursors: one launch, as above)
                UploadList ul{};
                uint32_t* ur = UP + L.up_misc + (size_t)2 * ntids + 1;
                for (int t = 0; t < ntids; ++t) ur[t] = (uint32_t)rbase[t] - rtab[t];
                ul.copy(T + o_roff, ur, (size_t)ntids);
                if (na && !solo) {   // where a destination's join records / census records start in the ONE send buffer, in records of their own size
                    uint32_t* cur = UP + L.up_cur;
                    for (int q = 0; q < world; ++q) {
                        cur[q] = (uint32_t)(sdispl[q] / kxw);
                        cur[(size_t)world + q] = (uint32_t)((sdispl[q] + (size_t)h_cnt[q] * kxw) / 2);
                    }
                    ul.copy(T + o_cnt + 2 * (size_t)world, cur, (size_t)world * 2);
                }
                launch_k9_upload(ul, s);
            }
            GlobalizeParams gp{};
            gp.tid = C->cp.tid; gp.region_of = C->k3.region_of; gp.n_ptr = &C->b_p1.as<Pass1>()->n_anom;
            gp.r_rec = C->b_r_rec.as<RegionRec>(); gp.r_pk = C->b_r_pk.as<uint32_t>(); gp.nr_local = nr_local; gp.roff = T + o_roff;
            gp.rg_rec = d->b_rg_rec.as<RegionRec>(); gp.rg_pk = d->b_rg_pk.as<uint32_t>(); gp.nkeys2 = nkeys2;
            gp.scratch = C->b_out_deg.as<uint32_t>(); gp.cap = capG; gp.counts = C->b_counts.as<StageCounts>(); gp.nr_global = (uint32_t)NR; gp.last_maxq = lm;
            launch_k9_globalize(gp, na, s);
            trace("globalize");
            C->k6_cap = (uint32_t)NR; C->k6_r_rec = d->b_rg_rec.as<RegionRec>(); C->k6_r_pk = d->b_rg_pk.as<uint32_t>();
            C->k6_taint = solo ? nullptr : taint;
            if (solo && rank == 0) {   // one rank: the genome's table is this rank's -- to pinned memory beside the joins, for the host's share of the walk and the result
                if (C->h_regs.ensure((size_t)NR * sizeof(RegionRec)) != hipSuccess || C->h_pk.ensure(std::max<size_t>((size_t)NR * nkeys2 * 4, 16)) != hipSuccess)
                    return dfail(d, BDX_ENOMEM, "region table");
                regs = C->h_regs.as<RegionRec>();
                pk = C->h_pk.as<uint32_t>();
                DHIP(d, hipEventRecord(C->ev_copy, s));
                DHIP(d, hipStreamWaitEvent(C->copy_stream, C->ev_copy, 0));
                static_assert(sizeof(RegionRec) % 4 == 0, "copied by words");
                UploadList ul{};   // (a kernel, not copy commands: the first device-to-host copy command of a process sets up a copy-engine queue, round 5)
                ul.copy(regs, d->b_rg_rec.p, (size_t)NR * sizeof(RegionRec) / 4);
                if (nkeys2) ul.copy(pk, d->b_rg_pk.p, (size_t)NR * nkeys2);
                launch_k9_upload(ul, C->copy_stream);
            }
            if (!solo) {
                if (nwin[rank]) {
                    WindowDst wd{};
                    wd.world = world;
                    for (int q = 0; q < world; ++q)
                        wd.dst[q] = q == rank ? nullptr : d->b_send.as<unsigned long long>() + sdispl[q] + (size_t)h_cnt[q] * kxw + (size_t)h_ncnt[q] * 2;
                    launch_k9_window_pack(d->b_rg_rec.as<RegionRec>(), (uint32_t)NR, period, wd, nwin[rank], T + o_ntot + 3, s);
                }
                if (na) {
                    xs.key = C->cp.key; xs.check = C->cp.check; xs.meta = C->cp.meta; xs.tid = C->cp.tid; xs.idx = C->cp.idx; xs.mtid_col = C->d.mtid;
                    xs.region_of = C->k3.region_of; xs.n_ptr = &C->b_p1.as<Pass1>()->n_anom; xs.owner_of_tid = (const int32_t*)(T + o_owner);
                    xs.ntids = ntids; xs.me = rank; xs.world = (uint32_t)world; xs.taint = taint;
                    launch_k7_scatter(xs, na, T + o_cnt + 2 * (size_t)world, d->b_send.as<ExchangeEntry>(), d->b_send.as<unsigned long long>(), s);
                }
            }
            trace("scatter");
            return BDX_OK;
        };
        d->err.clear();
        const int local_rc = st.rc == BDX_OK ? body() : BDX_OK;
        if (local_rc != BDX_OK) { st.rc = local_rc; st.msg = d->err; }
        d->phase_ms[6] += ms_between(tp, std::chrono::steady_clock::now());
        n_phase = 5;   // (slots 6 / 7 were this stretch and the all-to-all; 10 / 11 follow)
    }
    if (st.rc != BDX_OK && world > 1) {
        // (a rank that cannot take part in the all-to-all: the others would wait for it -- the communicator is given up)
        return leave(dfail(d, st.rc, st.msg));
    }
    if (st.rc != BDX_OK) return dfail(d, st.rc, st.msg);
    {
        const auto tp = std::chrono::steady_clock::now();
        if (!solo && !comm.alltoallv_u64(d->b_send.as<uint64_t>(), scount.data(), sdispl.data(), d->b_recv.as<uint64_t>(), rcount.data(), rdispl.data(), s))
            return leave(dfail(d, BDX_EHIP, comm.err));
        d->ctx_sent = nsend; d->ctx_received = nrecv;
        d->phase_ms[7] += ms_between(tp, std::chrono::steady_clock::now());
    }

    // ---- D: the joins (own reads + foreign entries), the name census beside them, the pair groups per region, the components, the device's
    // walk of the ones that lie inside this rank, the rank's table -- enqueued in one go: nothing of it waits for the host or for another rank.
    // A component that spans ranks is known as such WITHOUT an exchange: a region that sent a join record away is an end of a group another
    // rank forms (k7_scatter marks it), a region joined with a foreign entry by a gate-passing group is one too (k6_pairs marks it) -- its
    // groups go to rank 0 (round 5 all-reduced the marks).  The third all-reduce: what every rank will send to rank 0, and whether a name misbehaved ----
    // (a negative -s: shifted region ids are the read-level walk's business below; the pair model's device walk is not enqueued for them, as in bdx_run)
    const bool ph_opt = 0 > d->opts.min_len && 0.0f < (float)d->opts.seq_coverage_lim;
    const bool force_host = (rank == 0 ? U->host_walk_only : false) || C->host_walk_only || d->opts.min_read_pair < 1 || ph_opt;
    const uint32_t ph = (na_all && ph_opt) ? 1u : 0u;
    const bool replay_known = want_support != 0 || ph != 0;   // (the read-level walk on rank 0 serves these whatever the names look like)
    auto t_x1 = std::chrono::steady_clock::now();
    uint64_t irregular = 0;
    uint64_t mine_counts[10] = {0};
    phase([&]() -> int {
        const unsigned long long* R = d->b_recv.as<unsigned long long>();
        if (nwin_recv) launch_k9_window_unpack(d->b_rg_rec.as<RegionRec>(), (uint32_t)NR, period, R, seg_win, nwin_recv, T + o_ntot + 4, s);
        if (nnrecv) {   // the census of the names this rank owns: on the second stream, beside the joins (its verdict is read below)
            const uint32_t slots = k7_names_slots(nnrecv);
            DHIP(d, d->b_ntab.ensure((size_t)slots * 16));
            DHIP(d, d->b_nflag.ensure(16));
            DHIP(d, hipEventRecord(d->ev_side, s));
            DHIP(d, hipStreamWaitEvent(C->copy_stream, d->ev_side, 0));
            launch_k7_names_clear(d->b_ntab.as<unsigned long long>(), slots, d->b_nflag.as<uint32_t>(), C->copy_stream);
            launch_k7_names_census_seg(R, seg_cen, (uint32_t)nnrecv, d->b_ntab.as<unsigned long long>(), slots, d->b_nflag.as<uint32_t>(), C->copy_stream);
        }
        if (na) {
            if ((uint64_t)C->na_alloc + nrecv > (1u << 28)) return dfail(d, BDX_ELIMIT, "too many join entries on one rank");
            const uint32_t nf = (uint32_t)nrecv;
            DHIP(d, d->b_foreign.ensure(std::max<size_t>(nf, 1) * 20 + 64));
            uint64_t* fkey = d->b_foreign.as<uint64_t>();
            uint64_t* fcheck = fkey + std::max<size_t>(nf, 1);
            int32_t* fregion = (int32_t*)(fcheck + std::max<size_t>(nf, 1));
            launch_k7_unpack_seg(R, seg_ctx, nf, fkey, with_check ? fcheck : nullptr, fregion, &C->b_p1.as<Pass1>()->n_anom, T + o_ntot, s);
            Entries en{};
            en.key = C->cp.key; en.check = C->cp.check; en.region = C->k3.region_of; en.meta = C->cp.meta; en.isize = C->cp.isize;
            en.n_local = &C->b_p1.as<Pass1>()->n_anom; en.fkey = fkey; en.fcheck = fcheck; en.fregion = fregion; en.want_pair_lo = 1;
            DCTX(d, C, do_join_local(C, C->na_alloc + nf, en, T + o_ntot, true));
        }
        trace("join");
        t_x1 = std::chrono::steady_clock::now();
        DCTX(d, C, do_k6(C, force_host, 1));
        DCTX(d, C, do_k6(C, force_host, 2));   // (components and, right behind them, the device's walk)
        trace("components and walk");
        if (solo && rank == 0) {   // one rank: its host walks what the device walk leaves, as in bdx_run -- on the table that went to pinned memory beside the joins
            DHIP(d, hipStreamSynchronize(C->copy_stream));
            decode_regions(C, regs, pk, (uint32_t)NR, 0, true);
        }
        if (!wait_flag(C, 1, C->seq)) {
            if (C->poll) DHIP(d, hipStreamSynchronize(s)); else DHIP(d, hipEventSynchronize(C->ev_groups));
            if (!flag_arrived(C, 1)) return dfail(d, BDX_EINTERNAL, "the pair groups did not arrive: their kernels were not launched");
        }
        C->counts = *C->h_counts.as<StageCounts>();
        if (C->counts.irregular) irregular = 1;   // a read name seen more than twice: the run is replayed read by read on rank 0 (below)
        if (C->counts.overflow) return dfail(d, BDX_EINTERNAL, "group list overflow");
        if (nnrecv) {
            uint32_t flag = 0;
            DHIP(d, hipMemcpyAsync(&flag, d->b_nflag.p, 4, hipMemcpyDeviceToHost, C->copy_stream));
            DHIP(d, hipStreamSynchronize(C->copy_stream));
            if (flag) irregular = 1;
        }
        if (nwin_recv) {
            uint32_t werr = 0;
            DHIP(d, hipMemcpy(&werr, T + o_ntot + 4, 4, hipMemcpyDeviceToHost));
            if (werr) return dfail(d, BDX_EINTERNAL, "window read lengths of the exchange do not fit the region table");
        }
        if (irregular || replay_known) return BDX_OK;   // (the table of this model is not wanted)
        C->walk.clear();
        if (solo && rank == 0) {
            decode_groups(C, C->h_groups.as<GroupRec>(), C->counts.n_groups, 0);
            C->last_big_groups = (int64_t)C->counts.n_groups + C->counts.n_groups_big;
            const auto tw0 = std::chrono::steady_clock::now();
            DCTX(d, C, host_walk(C, lm, na_all != 0));
            d->phase_ms[16] = ms_between(tw0, std::chrono::steady_clock::now());
        }
        C->counts.last_maxq = lm;
        DCTX(d, C, do_k6_table(C));
        DCTX(d, C, finish_table(C));
        trace("table");
        mine_counts[0] = C->counts.n_groups;   // (several ranks: the groups of the components that go to rank 0)
        mine_counts[1] = C->n_sv_total; mine_counts[2] = C->n_terms_total; mine_counts[3] = C->n_cn_total; mine_counts[4] = C->n_printed;
        mine_counts[5] = C->n_sv_host; mine_counts[6] = C->counts.n_pairs; mine_counts[7] = C->n_groups_total; mine_counts[8] = C->counts.n_old;
        return BDX_OK;
    });
    d->ms_exchange = ms_between(t_x0, t_x1);
    std::vector<uint64_t> v5((size_t)world * 10 + 1, 0);
    if (st.rc == BDX_OK)
        for (int k = 0; k < 10; ++k) v5[(size_t)rank * 10 + k] = mine_counts[k];
    v5[(size_t)world * 10] = irregular;
    rc = exchange(v5);
    if (rc != BDX_OK) return rc;
    // some rank met a read name more than twice -- or the caller wants the reads behind every SV, which only the read-level walk knows
    // ... or -s is negative: the very first anomalous read of the genome then registers a read-less region 0 (BreakDancer.cpp:216-231,
    // 244-252; bdx_run's `ph`), every real region's id shifts by one and with it the flush cadence -- the read-level walk knows how
    // (ReadWalkInput::phantom), the pair model's kernels do not
    const bool replay = v5[(size_t)world * 10] != 0 || replay_known;

    // the genome's region table on rank 0: the ranks' dense tables placed by genome-wide id (several ranks: from a gather's packages)
    size_t region_bytes = 0;
    std::vector<size_t> gcount(world), gdispl(world);
    const size_t rrec = sizeof(RegionRec), rpk = (size_t)nkeys2 * 4;
    for (int q = 0; q < world; ++q) { gcount[q] = round_up((size_t)nr_of_rank[q] * (rrec + rpk), 8); gdispl[q] = region_bytes; region_bytes += gcount[q]; }
    if (!solo && (uint64_t)nr_local != nr_of_rank[rank]) return leave(dfail(d, BDX_EINTERNAL, "region counts of the chromosomes do not add up"));
    // rank 0, several ranks: the packages' region records (at all + base_off + gdispl[q]) -> the result context's table in HBM and, on the side stream, in pinned memory
    auto place_regions = [&](size_t base_off, bool with_pk = true) -> int {
        GatherDesc D{};
        D.world = world;
        uint32_t max_nr = 0;
        for (int q = 0; q < world; ++q) {
            const size_t nr = (size_t)nr_of_rank[q];
            D.p[q] = GatherPackage{base_off + gdispl[q], base_off + gdispl[q] + nr * rrec, 0, (uint32_t)nr, 0};
            max_nr = std::max(max_nr, (uint32_t)nr);
        }
        if (C->h_regs.ensure((size_t)NR * rrec) != hipSuccess || C->h_pk.ensure(std::max<size_t>((size_t)NR * rpk, 16)) != hipSuccess) return dfail(d, BDX_ENOMEM, "region table");
        regs = C->h_regs.as<RegionRec>();
        pk = C->h_pk.as<uint32_t>();
        if (U->b_r_rec.ensure((size_t)NR * rrec) != hipSuccess || U->b_r_pk.ensure(std::max<size_t>((size_t)NR * rpk, 16)) != hipSuccess) return dfail(d, BDX_ENOMEM, "region table");
        DHIP(d, hipMemcpyAsync(T + o_rbase, rbase.data(), ((size_t)ntids + 1) * 8, hipMemcpyHostToDevice, s));
        launch_k8_place_regions((const char*)d->b_all.p, D, max_nr, (const uint64_t*)(T + o_rbase), ntids, nkeys2, U->b_r_rec.as<RegionRec>(), U->b_r_pk.as<uint32_t>(),
                                T + o_ntot + 1, s);
        DHIP(d, hipEventRecord(C->ev_copy, s));
        DHIP(d, hipStreamWaitEvent(C->copy_stream, C->ev_copy, 0));
        UploadList ul{};
        ul.copy(regs, U->b_r_rec.p, (size_t)NR * rrec / 4);
        if (nkeys2 && with_pk) ul.copy(pk, U->b_r_pk.p, (size_t)NR * rpk / 4);   // (without: the rows the host's walk touches follow, launch_k9_pk_rows)
        launch_k9_upload(ul, C->copy_stream);
        regs_pending = true;
        return BDX_OK;
    };
    auto wait_regions = [&]() -> int {
        if (regs_pending) {
            DHIP(d, hipStreamSynchronize(C->copy_stream));
            regs_pending = false;
            uint32_t perr = 0;
            DHIP(d, hipMemcpy(&perr, T + o_ntot + 1, 4, hipMemcpyDeviceToHost));
            if (perr) return dfail(d, BDX_EINTERNAL, "region table of the gather does not add up");
        }
        return BDX_OK;
    };
    // this rank's region records and their prefix samples, one after the other, at dst (device memory)
    auto pack_regions = [&](char* dst) -> bool {
        if (!nr_local) return true;
        return hipMemcpyAsync(dst, C->b_r_rec.p, (size_t)nr_local * rrec, hipMemcpyDeviceToDevice, s) == hipSuccess &&
               hipMemcpyAsync(dst + (size_t)nr_local * rrec, C->b_r_pk.p, (size_t)nr_local * rpk, hipMemcpyDeviceToDevice, s) == hipSuccess;
    };

    // ---- a read name seen more than twice (clashing names across merged files): the pair model does not hold, and the reference's
    // behaviour (ReadRegionData.cpp:108-113,152-175, SvBuilder.cpp:101-118) depends on the order of ALL sightings.  Every rank sends
    // its region records and the compact records of its chromosomes (name key, genome-wide region id, meta, |isize|, tid, second name
    // hash, index in the chromosome's stream: 40 bytes per anomalous read) to rank 0, which replays the run read by read (H2,
    // bdx_walk_reads.cpp) on the gathered region table.  The same route serves bdx_dist_set_collect_support: the supporting reads of an
    // SV (-g / -d dumps, BreakDancer.cpp:514-534) are known to the read-level walk only ----
    if (replay) {
        const auto t_r0 = std::chrono::steady_clock::now();
        constexpr size_t kw = 5;
        DHIP(d, hipStreamSynchronize(s));   // (K6's kernels were enqueued on the pair model: let them finish, their results are dropped)
        if (!solo) {
            if (d->b_pack.ensure(std::max<size_t>(gcount[rank], 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "region package"));
            if (!pack_regions((char*)d->b_pack.p)) return leave(dfail(d, BDX_EHIP, "region package"));
            if (rank == 0 && d->b_all.ensure(std::max<size_t>(region_bytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
            if (!comm.gatherv_bytes(d->b_pack.p, gcount[rank], d->b_all.p, gcount.data(), gdispl.data(), 0, s)) return leave(dfail(d, BDX_EHIP, comm.err));
            d->gathered_bytes += region_bytes;
            if (rank == 0) {
                const int pr = place_regions(0);
                if (pr != BDX_OK) return leave(pr);
                const int wr = wait_regions();   // (the packages sit in b_all, which the records' gather takes next)
                if (wr != BDX_OK) return leave(wr);
            }
        }
        std::vector<size_t> rp_count(world), rp_displ(world);
        size_t rp_bytes = 0;
        std::vector<uint64_t> na_of_rank(world, 0);
        for (int t = 0; t < ntids; ++t)
            if (owner[t] >= 0) na_of_rank[owner[t]] += tot(t, 0);
        for (int q = 0; q < world; ++q) { rp_count[q] = (size_t)na_of_rank[q] * kw * 8; rp_displ[q] = rp_bytes; rp_bytes += rp_count[q]; }
        if (d->b_pack.ensure(std::max<size_t>(rp_count[rank], 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "record package"));
        if (na) launch_k9_pack_replay(C->cp, C->k3.region_of, &C->b_p1.as<Pass1>()->n_anom, na, T + o_start, d->b_pack.as<unsigned long long>(), s);
        if (rank == 0 && d->b_all.ensure(std::max<size_t>(rp_bytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
        if (!comm.gatherv_bytes(d->b_pack.p, rp_count[rank], d->b_all.p, rp_count.data(), rp_displ.data(), 0, s)) return leave(dfail(d, BDX_EHIP, comm.err));
        DHIP(d, hipStreamSynchronize(s));
        d->gathered_bytes += rp_bytes;
        if (rank == 0) {
            std::vector<uint64_t> rp_host(rp_bytes / 8);
            if (rp_bytes) DHIP(d, hipMemcpy(rp_host.data(), d->b_all.p, rp_bytes, hipMemcpyDeviceToHost));
            // the records of all chromosomes in stream order: chromosomes ascending, each rank's package holds its own in order
            if (rp_host.size() != (size_t)na_all * kw) return dfail(d, BDX_EINTERNAL, "compact records of the gather do not add up");
            std::vector<uint64_t> key(na_all), chk(with_check ? na_all : 0);
            std::vector<int32_t> reg(na_all), isz(na_all);
            std::vector<uint32_t> meta(na_all);
            std::vector<uint64_t> sidx(want_support ? na_all : 0);   // index in the merged stream
            std::vector<uint64_t> at(ntids);
            for (int t = 0; t < ntids; ++t) at[t] = base[(size_t)t * tw];
            for (size_t i = 0; i < (size_t)na_all; ++i) {
                const uint64_t w0 = rp_host[i * kw], w1 = rp_host[i * kw + 1], w2 = rp_host[i * kw + 2];
                const uint32_t t = (uint32_t)(w2 >> 32);
                if (t >= (uint32_t)ntids || at[t] >= base[(size_t)(t + 1) * tw]) return dfail(d, BDX_EINTERNAL, "compact records of the gather do not add up");
                const size_t o = (size_t)at[t]++;
                key[o] = w0; reg[o] = (int32_t)(uint32_t)w1; meta[o] = (uint32_t)(w1 >> 32); isz[o] = (int32_t)(uint32_t)w2;
                if (ph && reg[o] >= 0) reg[o] += (int32_t)ph;   // (the read-less region 0 in front: replay_reads does the same)
                if (with_check) chk[o] = rp_host[i * kw + 3];
                if (want_support) sidx[o] = read_base[t] + rp_host[i * kw + 4];
            }
            if (solo) {   // (one rank: its table went to pinned memory beside the walk)
                if (!regs) {
                    if (C->h_regs.ensure((size_t)NR * rrec) != hipSuccess || C->h_pk.ensure(std::max<size_t>((size_t)NR * rpk, 16)) != hipSuccess) return dfail(d, BDX_ENOMEM, "region table");
                    regs = C->h_regs.as<RegionRec>(); pk = C->h_pk.as<uint32_t>();
                    DHIP(d, hipMemcpy(regs, d->b_rg_rec.p, (size_t)NR * rrec, hipMemcpyDeviceToHost));
                    if (nkeys2) DHIP(d, hipMemcpy(pk, d->b_rg_pk.p, (size_t)NR * rpk, hipMemcpyDeviceToHost));
                }
            }
            // (a region's first read: its index in its rank's list -> in the genome-wide one)
            {
                std::vector<uint64_t> seen(world, 0), adj(ntids, 0);
                for (int t = 0; t < ntids; ++t)
                    if (owner[t] >= 0) { adj[t] = base[(size_t)t * tw] - seen[owner[t]]; seen[owner[t]] += tot(t, 0); }
                decode_regions(U, regs, pk, (uint32_t)NR, ph, false);
                for (size_t r = ph; r < NR + ph; ++r) U->regions[r].first += (uint32_t)adj[U->regions[r].tid];
            }
            memset(&U->counts, 0, sizeof(U->counts));
            U->counts.n_regions = (uint32_t)NR;
            U->counts.last_maxq = lm;
            if (with_check) unify_names(key.data(), chk.data(), (size_t)na_all);
            std::vector<uint32_t> sup;
            U->collect_support = want_support != 0;
            DCTX(d, U, replay_arrays(U, (uint32_t)na_all, key.data(), reg.data(), meta.data(), isz.data(), ph, want_support ? &sup : nullptr));
            if (want_support) {   // compact indices -> indices in the merged stream, and the reads' flags
                U->sup_idx.resize(sup.size());
                U->sup_flag.resize(sup.size());
                for (size_t i = 0; i < sup.size(); ++i) { U->sup_idx[i] = sidx[sup[i]]; U->sup_flag[i] = (uint8_t)meta_flag(meta[sup[i]]); }
            }
            U->p1.n_anom = (uint32_t)na_all;
        }
        d->phase_ms[15] = ms_between(t_r0, std::chrono::steady_clock::now());
        return finish_result();
    }

    if (solo) {
        adopt_table(U, C);
        d->table_lent = true;
        U->reg = C->reg; U->nreg = C->nreg; U->rpk = C->rpk;   // (the genome's region table stays in this rank's pinned buffers until the next run)
        C->reg = nullptr; C->nreg = 0; C->rpk = nullptr;
        U->counts.n_regions = (uint32_t)NR;
        U->ran = true; U->stage = 4;
        d->phase_ms[14] = 0;
        return finish_result();
    }

    // ---- E: ONE gather takes every rank's package to rank 0: its region records, the pair groups of the components it could not walk
    // alone, its finished table (rows sorted by order key) -- round 5: three gathers with two all-reduces of sizes between them.  Rank 0 then
    // walks the gathered groups in its result context (K6 once more: the groups bucketed by later region stand where a rank's K6 has its
    // reads, components of up to 64 regions on the device, the rest by the host -- round 5's host walked all of them, BreakDancer.cpp:266-346
    // being one global walk) and merges the ranks' tables and that one by order key into the result context's pinned buffers ----
    const auto t_m0 = std::chrono::steady_clock::now();
    TableDesc TD{};
    TD.world = world;
    std::vector<size_t> pcount(world), pdispl(world), grp_off(world);
    size_t pbytes = 0, ng_all = 0;
    uint64_t n_sv_all = 0, n_terms_all = 0, n_cn_all = 0, n_printed_all = 0, n_pairs_all = 0, n_groups_all = 0, n_old_all = 0, n_sv_host_all = 0;
    uint32_t max_sv = 0;
    for (int q = 0; q < world; ++q) {
        const uint64_t* c = &v5[(size_t)q * 10];
        const size_t ng = (size_t)c[0], nsv = (size_t)c[1], nt = (size_t)c[2], nc = (size_t)c[3];
        size_t o = pbytes + gcount[q];          // (the package starts with the rank's region records: gdispl[q] is NOT their place here, pdispl[q] is)
        grp_off[q] = o; o += ng * sizeof(GroupRec);
        TablePackage& P = TD.p[q];
        P.n_sv = (uint32_t)nsv; P.n_terms = (uint32_t)nt; P.n_cn = (uint32_t)nc;
        P.rows_off = o; o += round_up(nsv * sizeof(SvOut), 8);
        P.keys_off = o; o += nsv * 8;
        P.lib_index_off = o; o += round_up(nt * 4, 8);
        P.lib_pairs_off = o; o += round_up(nt * 4, 8);
        P.ltail_off = o; o += nt * 8;
        P.cn_key_off = o; o += round_up(nc * 4, 8);
        P.cn_value_off = o; o += round_up(nc * 4, 8);
        pdispl[q] = pbytes; pcount[q] = o - pbytes; pbytes = o;
        ng_all += ng;
        n_sv_all += nsv; n_terms_all += nt; n_cn_all += nc; n_printed_all += c[4]; n_sv_host_all += c[5];
        n_pairs_all += c[6]; n_groups_all += c[7]; n_old_all += c[8];
        max_sv = std::max(max_sv, (uint32_t)nsv);
    }
    if (tracing && rank == 0) fprintf(stderr, "[bdx dist] pair groups gathered on rank 0: %zu\n", ng_all);
    if (ng_all > kMaxAnomalous / 4) return dfail(d, BDX_ELIMIT, "too many pair groups of components that span ranks");   // (the same sums on every rank)
    if (max_sv >= (1u << 26) || n_sv_all > 0xFFFFFFF0ull) return dfail(d, BDX_ELIMIT, "too many SV candidates for the merge");
    {
        const TablePackage& P = TD.p[rank];
        if (d->b_pack.ensure(std::max<size_t>(pcount[rank], 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "package for rank 0"));
        char* pp = (char*)d->b_pack.p - pdispl[rank];
        bool good = pack_regions((char*)d->b_pack.p);
        auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes && good) good = hipMemcpyAsync(pp + off, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess; };
        put(grp_off[rank], C->k4.g_rec, (size_t)v5[(size_t)rank * 10] * sizeof(GroupRec));
        put(P.rows_off, C->b_sv_out.p, (size_t)P.n_sv * sizeof(SvOut)); put(P.keys_off, C->b_sv_key.p, (size_t)P.n_sv * 8);
        put(P.lib_index_off, C->b_lib_index_out.p, (size_t)P.n_terms * 4); put(P.lib_pairs_off, C->b_lib_pairs_out.p, (size_t)P.n_terms * 4);
        put(P.ltail_off, C->b_ltail_out.p, (size_t)P.n_terms * 8);
        put(P.cn_key_off, C->b_cn_key_out.p, (size_t)P.n_cn * 4); put(P.cn_value_off, C->b_cn_value_out.p, (size_t)P.n_cn * 4);
        if (!good) return leave(dfail(d, BDX_EHIP, "package for rank 0"));
        trace("package");
        if (rank == 0 && d->b_all.ensure(std::max<size_t>(pbytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
        const auto tg = std::chrono::steady_clock::now();
        if (!comm.gatherv_bytes(d->b_pack.p, pcount[rank], d->b_all.p, pcount.data(), pdispl.data(), 0, s)) return leave(dfail(d, BDX_EHIP, comm.err));
        d->phase_ms[13] += ms_between(tg, std::chrono::steady_clock::now());
        d->gathered_bytes += pbytes;
        trace("gathered");
    }
    if (rank != 0) {
        d->phase_ms[14] = ms_between(t_m0, std::chrono::steady_clock::now());
        return finish_result();
    }
    // ---- rank 0 alone from here ----
    uint64_t u_counts[9] = {0};
    {
        // (place_regions takes the packages' region records at base + gdispl[q]; here they start the packages: their own displacements)
        for (int q = 0; q < world; ++q) gdispl[q] = pdispl[q];
        const int pr = place_regions(0, false);
        if (pr != BDX_OK) return pr;
    }
    if (ng_all) {
        const auto tw0 = std::chrono::steady_clock::now();
        const uint32_t ng = (uint32_t)ng_all, nr = (uint32_t)NR;
        // FEW gathered groups (a real genome's inter-chromosomal clusters: dozens of components) are walked by rank 0's HOST: K6 once more on
        // the device is twenty launches -- 0.3 ms of rank 0's time alone for 52 components -- where the host walks them in microseconds.  MANY
        // (thousands of translocations) stay on the device: round 5's host took 1.1-1.3 ms for 13 k groups.  Test switch "gather_walk" on the
        // result context: 1 always the device, 2 always the host.
        const bool host_gather = U->dbg_gather_walk == 2 || (U->dbg_gather_walk == 0 && ng_all <= kGatherHostMax);
        uint32_t* bucket_err = nullptr;   // (device walk: the bucket kernel's verdict)
        if (host_gather) {
            hipStream_t su = U->stream;
            const uint32_t capU = 2 * std::max(nr, ng) + 2;   // (sized as for the device walk below: a candidate consumes at least one pair group)
            DHIP(d, U->b_counts.ensure(sizeof(StageCounts))); DHIP(d, U->h_counts.ensure(sizeof(StageCounts)));
            DHIP(d, U->b_cnt.ensure((size_t)ncnt * 4)); DHIP(d, U->b_p1.ensure(sizeof(Pass1))); DHIP(d, U->b_kdens.ensure(64 * 4));
            DHIP(d, U->h_flags.ensure(64)); DHIP(d, U->h_groups.ensure(((size_t)ng + 1) * sizeof(GroupRec)));
            // the gathered groups: out of every rank's package into pinned memory, behind the gather
            {
                GroupRec* hg = U->h_groups.as<GroupRec>();
                size_t at = 0;
                for (int q = 0; q < world; ++q) {
                    const size_t nq = (size_t)v5[(size_t)q * 10];
                    if (nq) DHIP(d, hipMemcpyAsync(hg + at, (const char*)d->b_all.p + grp_off[q], nq * sizeof(GroupRec), hipMemcpyDeviceToHost, s));
                    at += nq;
                }
            }
            DHIP(d, hipEventRecord(d->ev_side, s));   // (the gather, the region table placed behind it, the groups' copies)
            DHIP(d, hipStreamWaitEvent(su, d->ev_side, 0));
            {
                StageCounts sc{};
                sc.n_regions = nr; sc.last_maxq = lm;
                memcpy(UP + L.up_counts, &sc, sizeof(sc));
                const uint32_t* st_up = UP + L.up_stats;
                UploadList ul{};
                ul.copy(U->b_counts.p, UP + L.up_counts, sizeof(StageCounts) / 4);
                ul.copy(U->b_p1.p, st_up, 2);
                ul.copy(U->b_cnt.p, st_up + 2, (size_t)ncnt);
                ul.copy(U->b_kdens.p, st_up + 2 + ncnt, U->key_density.size());
                launch_k9_upload(ul, su);
            }
            ++U->seq;
            U->na_alloc = 0; U->k6_cap = capU;
            U->k6_r_rec = U->b_r_rec.as<RegionRec>(); U->k6_r_pk = U->b_r_pk.as<uint32_t>(); U->k6_taint = nullptr;
            U->k6_in_groups = nullptr; U->k6_in_goff = nullptr;
            U->cp = Compact{}; U->k3 = K3Arrays{}; U->k4 = K4Arrays{};
            U->k4.g_rec = U->h_groups.as<GroupRec>(); U->k4.g_cap = ng + 1;
            U->table_in_hbm = true; U->groups_in_hbm = false; U->defer_walk = false;
            memset(&U->counts, 0, sizeof(U->counts));
            DCTX(d, U, do_k6(U, false, 4));   // (K6's arrays, no launch: the table stage below finds no candidate of the device's; NOT force_host -- that zeroes
                                              // the host candidates' order keys, which the merge of the ranks' tables goes by)
            DHIP(d, hipMemsetAsync(U->k6.own_nsv, 0, (size_t)capU * 3 * 4, su));   // own_nsv | own_nacc | own_ncn: no vertex has candidates of its own
            DHIP(d, hipStreamSynchronize(s));   // (the groups are in pinned memory)
            U->counts.n_regions = nr; U->counts.n_groups = ng; U->counts.last_maxq = lm;
        } else {
            // (K6 sizes its lists for a context's anomalous reads -- candidates and list entries <= reads / 2, each consuming a read pair; here
            // a candidate consumes at least one pair GROUP and a list entry is a part of one: twice the groups stands for the reads)
            const uint32_t capU = 2 * std::max(nr, ng) + 2;
            // buckets: cnt | goff | cur ([nr + 1] each) | scan workspace | n | err, then the groups in bucket order
            const size_t w_scan = 2 * ((size_t)scan_grid(nr + 1) + 2), o_goff = (size_t)nr + 1, o_cur = 2 * o_goff, o_ws = 3 * o_goff, o_n = o_ws + w_scan, o_err = o_n + 1,
                         o_grp = (o_err + 1 + 3) / 4 * 4;
            DHIP(d, d->b_bucket.ensure(o_grp * 4 + (size_t)ng * sizeof(GroupRec)));
            uint32_t* B = d->b_bucket.as<uint32_t>();
            GroupRec* sorted = (GroupRec*)(B + o_grp);
            SegList seg_grp{};
            seg_grp.n = world;
            {
                uint32_t a = 0;
                for (int q = 0; q < world; ++q) { seg_grp.off[q] = grp_off[q] / 8; seg_grp.start[q] = a; a += (uint32_t)v5[(size_t)q * 10]; }
                seg_grp.start[world] = a;
            }
            hipStream_t su = U->stream;
            DHIP(d, hipEventRecord(d->ev_side, s));   // (the gather, and the region table placed behind it)
            DHIP(d, hipStreamWaitEvent(su, d->ev_side, 0));
            launch_k9_bucket_groups((const unsigned long long*)d->b_all.p, seg_grp, ng, nr, B, B + o_goff, B + o_cur, sorted, B + o_ws, B + o_n, B + o_err, su);
            bucket_err = B + o_err;
            // the result context as a K6 context: the genome's statistics and region table, no reads
            DHIP(d, U->b_counts.ensure(sizeof(StageCounts))); DHIP(d, U->h_counts.ensure(sizeof(StageCounts)));
            DHIP(d, U->b_cnt.ensure((size_t)ncnt * 4)); DHIP(d, U->b_p1.ensure(sizeof(Pass1))); DHIP(d, U->b_kdens.ensure(64 * 4));
            DHIP(d, U->h_flags.ensure(64)); DHIP(d, U->h_groups.ensure(((size_t)ng + 1) * sizeof(GroupRec)));
            DHIP(d, U->b_out_deg.ensure((size_t)capU * 6 * 4));
            {
                StageCounts sc{};
                sc.n_regions = nr; sc.last_maxq = lm;
                memcpy(UP + L.up_counts, &sc, sizeof(sc));
                const uint32_t* st_up = UP + L.up_stats;   // (covered, window | flag histogram | densities: as this rank's own context got them)
                UploadList ul{};
                ul.copy(U->b_counts.p, UP + L.up_counts, sizeof(StageCounts) / 4);
                ul.copy(U->b_p1.p, st_up, 2);
                ul.copy(U->b_cnt.p, st_up + 2, (size_t)ncnt);
                ul.copy(U->b_kdens.p, st_up + 2 + ncnt, U->key_density.size());
                launch_k9_upload(ul, su);
            }
            launch_k6_scratch_init(U->b_out_deg.as<uint32_t>(), capU, su);
            ++U->seq;
            U->na_alloc = 0; U->k6_cap = capU;
            U->k6_r_rec = U->b_r_rec.as<RegionRec>(); U->k6_r_pk = U->b_r_pk.as<uint32_t>(); U->k6_taint = nullptr;
            U->k6_in_groups = sorted; U->k6_in_goff = B + o_goff;
            U->cp = Compact{}; U->k3 = K3Arrays{}; U->k4 = K4Arrays{};
            U->k4.g_rec = U->h_groups.as<GroupRec>(); U->k4.g_cap = ng + 1;
            U->table_in_hbm = true; U->groups_in_hbm = false; U->defer_walk = false;
            if (U->big_walk_mode < 0) U->last_big_groups = 1 << 20;   // (components of 5..64 regions on the device as well: what is left is the host's, sequentially)
            // (components that span ranks are mostly a translocation's two regions and their neighbours: three rounds of label propagation settle
            // them -- the eight of a context that walks large components are seven launches on rank 0's own part of the run; what has not
            // converged fails the closure check and is the host's)
            if (!U->dbg_label_rounds) U->dbg_label_rounds = 3;
            memset(&U->counts, 0, sizeof(U->counts));
            DCTX(d, U, do_k6(U, force_host, 0));
            if (!wait_flag(U, 1, U->seq)) {
                DHIP(d, hipStreamSynchronize(su));
                if (!flag_arrived(U, 1)) return dfail(d, BDX_EINTERNAL, "the pair groups of the gathered components did not arrive: their kernels were not launched");
            }
            U->counts = *U->h_counts.as<StageCounts>();
            if (U->counts.overflow) return dfail(d, BDX_EINTERNAL, "group list overflow (gathered components)");
        }
        decode_groups(U, U->h_groups.as<GroupRec>(), U->counts.n_groups, 0);
        U->last_big_groups = (int64_t)U->counts.n_groups + U->counts.n_groups_big;
        if (U->counts.n_groups) {   // (the host's share of this walk reads the table in pinned memory: its copy ran beside the device's walk)
            // ... and the proper-read samples of the regions its groups name: those rows only
            // (on the stream of the table's copy, behind it: the result context's own stream is still walking)
            launch_k9_pk_rows(U->h_groups.as<GroupRec>(), U->counts.n_groups, U->b_r_pk.as<uint32_t>(), pk, (uint32_t)nkeys2, nr, C->copy_stream);
            regs_pending = true;
            const int wr = wait_regions();
            if (wr != BDX_OK) return wr;
        }
        decode_regions(U, regs, pk, nr, 0, true);
        const auto tw1 = std::chrono::steady_clock::now();
        DCTX(d, U, host_walk(U, lm, na_all != 0));
        d->phase_ms[16] = ms_between(tw1, std::chrono::steady_clock::now());
        U->counts.last_maxq = lm;
        DCTX(d, U, do_k6_table(U));
        DCTX(d, U, finish_table(U));
        if (bucket_err) {   // (the bucket kernel's verdict, read once the table is there: a wait for it in front of the host's share of the walk kept the host from
            // enqueueing the table stage while the device walked -- 40 us of rank 0's time alone; a group outside the table is skipped by the kernels)
            uint32_t berr = 0;
            DHIP(d, hipMemcpyAsync(&berr, bucket_err, 4, hipMemcpyDeviceToHost, U->stream));
            DHIP(d, hipStreamSynchronize(U->stream));
            if (berr) return dfail(d, BDX_EINTERNAL, "a gathered pair group names a region outside the genome's table");
        }
        u_counts[0] = U->n_sv_total; u_counts[1] = U->n_terms_total; u_counts[2] = U->n_cn_total; u_counts[3] = U->n_printed;
        u_counts[4] = U->n_sv_host; u_counts[6] = U->n_groups_total; u_counts[7] = U->counts.n_old; u_counts[8] = U->counts.n_groups;
        d->phase_ms[17] = ms_between(tw0, std::chrono::steady_clock::now()) - d->phase_ms[16];
        TablePackage& P = TD.p[world];   // the table of the gathered components, where the result context's K6 left it (byte offsets relative to the gather buffer)
        P.n_sv = (uint32_t)u_counts[0]; P.n_terms = (uint32_t)u_counts[1]; P.n_cn = (uint32_t)u_counts[2];
        auto off = [&](const DevBuf& b) { return (uint64_t)((uintptr_t)b.p - (uintptr_t)d->b_all.p); };
        P.rows_off = off(U->b_sv_out); P.keys_off = off(U->b_sv_key); P.lib_index_off = off(U->b_lib_index_out); P.lib_pairs_off = off(U->b_lib_pairs_out);
        P.ltail_off = off(U->b_ltail_out); P.cn_key_off = off(U->b_cn_key_out); P.cn_value_off = off(U->b_cn_value_out);
        n_sv_all += u_counts[0]; n_terms_all += u_counts[1]; n_cn_all += u_counts[2]; n_printed_all += u_counts[3]; n_sv_host_all += u_counts[4];
        n_groups_all += u_counts[6]; n_old_all += u_counts[7];
        max_sv = std::max(max_sv, (uint32_t)u_counts[0]);
        TD.world = world + 1;
        if (max_sv >= (1u << 26) || n_sv_all > 0xFFFFFFF0ull) return dfail(d, BDX_ELIMIT, "too many SV candidates for the merge");
        DHIP(d, hipEventRecord(d->ev_side, U->stream));   // (its table kernel: finish_table has seen its ready word, this orders the streams)
        DHIP(d, hipStreamWaitEvent(s, d->ev_side, 0));
    }
    {
        const uint32_t n_total = (uint32_t)n_sv_all;
        DHIP(d, U->h_sv_out.ensure(std::max<size_t>(n_total, 1) * sizeof(SvOut)));
        DHIP(d, U->h_lib_index.ensure(std::max<size_t>(n_terms_all, 1) * 4)); DHIP(d, U->h_lib_pairs.ensure(std::max<size_t>(n_terms_all, 1) * 4));
        DHIP(d, U->h_ltail_dev.ensure(std::max<size_t>(n_terms_all, 1) * 8));
        DHIP(d, U->h_cn_key.ensure(std::max<size_t>(n_cn_all, 1) * 4 + 16)); DHIP(d, U->h_cn_value.ensure(std::max<size_t>(n_cn_all, 1) * 4 + 16));
        if (n_total) {
            const size_t ws_words = ((size_t)scan_grid(n_total) + 4) * 4;
            DHIP(d, d->b_merge.ensure((size_t)n_total * 12 + ws_words * 4 + sizeof(TableDesc) + 128));
            uint2* begins = d->b_merge.as<uint2>();
            uint32_t* src = (uint32_t*)(begins + n_total);
            uint32_t* ws = (uint32_t*)(((uintptr_t)(src + n_total) + 15) & ~(uintptr_t)15);
            TableDesc* d_td = (TableDesc*)(((uintptr_t)(ws + ws_words) + 15) & ~(uintptr_t)15);
            DHIP(d, hipMemcpyAsync(d_td, &TD, sizeof(TableDesc), hipMemcpyHostToDevice, s));
            DHIP(d, hipMemcpyAsync(T + o_ntot + 2, &n_total, 4, hipMemcpyHostToDevice, s));
            MergeOut mo{U->h_sv_out.as<SvOut>(), U->h_lib_index.as<int32_t>(), U->h_lib_pairs.as<int32_t>(), d->opts.fisher ? U->h_ltail_dev.as<double>() : nullptr,
                        U->h_cn_key.as<int32_t>(), U->h_cn_value.as<float>()};
            if (tracing) {   // are the ranks' tables sorted by key, and are the keys distinct?
                DHIP(d, hipStreamSynchronize(s));
                for (int q = 0; q < TD.world; ++q) {
                    std::vector<unsigned long long> kk(TD.p[q].n_sv);
                    if (!kk.empty()) DHIP(d, hipMemcpy(kk.data(), (const char*)d->b_all.p + TD.p[q].keys_off, kk.size() * 8, hipMemcpyDeviceToHost));
                    size_t bad = 0;
                    for (size_t i = 1; i < kk.size(); ++i) bad += kk[i] < kk[i - 1];
                    fprintf(stderr, "[bdx dist] table of rank %d: %zu rows, %zu out of key order; first keys", q, kk.size(), bad);
                    for (size_t i = 0; i < kk.size() && i < 6; ++i) fprintf(stderr, " (T %llu own %llu start %llu)", kk[i] >> 34, (kk[i] >> 33) & 1, (kk[i] >> 7) & 0x3FFFFFF);
                    fprintf(stderr, "\n");
                }
            }
            DHIP(d, hipMemsetAsync(src, 0xFF, (size_t)n_total * 4, s));
            launch_k9_merge_tables((const char*)d->b_all.p, d_td, TD.world, n_total, max_sv, src, begins, ws, T + o_ntot + 2, mo, s);
        }
        DHIP(d, hipStreamSynchronize(s));
        DHIP(d, hipGetLastError());
        trace("merge");
        U->walk.clear(); U->log_tail.clear();
        U->n_sv_total = n_total; U->n_terms_total = (uint32_t)n_terms_all; U->n_cn_total = (uint32_t)n_cn_all; U->n_printed = (uint32_t)n_printed_all;
        U->n_sv_host = (uint32_t)n_sv_host_all; U->n_groups_total = (uint32_t)n_groups_all;
        memset(&U->counts, 0, sizeof(U->counts));
        U->counts.n_regions = (uint32_t)NR; U->counts.last_maxq = lm; U->counts.n_pairs = (uint32_t)n_pairs_all; U->counts.n_old = (uint32_t)n_old_all;
        U->counts.n_sv_dev = n_total - U->n_sv_host; U->counts.n_groups = (uint32_t)u_counts[8];   // (what the host's walk took: components the device walk leaves out)
        U->materialized = false; U->rows_packed = true;   // (k9_merge_tables writes SvWire rows, like a single context's table kernel)
        {   // the genome's region table: in rank 0's pinned buffers until the next run
            const int wr = wait_regions();
            if (wr != BDX_OK) return wr;
            decode_regions(U, regs, pk, (uint32_t)NR, 0, true);
            C->reg = nullptr; C->nreg = 0; C->rpk = nullptr;
        }
        if (d->opts.fisher) {  // Fisher's combination (BreakDancer.cpp:71-81) uses the host's exp / log
            materialize(U);
            finish_scores(U->opts, U->log_tail.data(), U->walk.svs.data(), U->walk.svs.size(), &U->n_printed);
        }
        U->ran = true; U->stage = 4;
    }
    d->phase_ms[14] = std::max(0.0f, ms_between(t_m0, std::chrono::steady_clock::now()) - d->phase_ms[16] - d->phase_ms[17] - d->phase_ms[13]);
    return finish_result();
}

}  // extern "C"
