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
// genome-wide on every rank, so flush windows, order keys and the walk itself are the single run's.  Between the stages:
//   C1  all-reduce: pass-1 counters, per-file reference lengths, per-chromosome totals and owners
//   C2  all-reduce: each chromosome's first anomalous read;  C3: its region count  (every entry has one owner: a sum is a gather)
//   C4  all-reduce of the exchange's count matrix, then ONE all-to-all of the CTX join records -- a CTX read whose mate lies on a
//       later chromosome of another rank goes there and joins that rank's reads (k7_exchange.hip) -- and one of the name census
//   C5  gather of the ranks' region tables on rank 0 (the result holds the genome's table; runs beside the joins)
//   C6  all-reduce (payload stays in HBM): taint bytes of the regions that gate-passing groups connect ACROSS ranks, and the read
//       length of each flush window's last region.  Components inside one rank are walked and scored where they live (K6);
//       tainted ones go to the host's share
//   C7  all-reduce: size of every rank's host share;  C8: gather of the host shares on rank 0, which walks them (H1)
//   C9  gather of the ranks' finished tables (rows sorted by order key) on rank 0, merged by key into the result context
// Payloads stay in HBM: the collectives run on device buffers through RCCL (ncclAllReduce, ncclAllToAllv, grouped
// ncclSend / ncclRecv), xGMI between the GPUs of a node.  A second backend runs the ranks as threads of one process
// (tests on a single GPU; a host program that drives several GPUs itself).
#include <dlfcn.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>

namespace {

constexpr int kDistPhases = 18;   // bdx_dist_get_phase_ms / bdx_dist_phase_name

// ------------------------------------------------------------------------------------------------------------------
// communicators
// ------------------------------------------------------------------------------------------------------------------
struct Comm {
    int rank = 0, world = 1;
    std::string err;
    virtual ~Comm() {}
    virtual void begin_run() {}   // start of a bdx_dist_run (collective)
    virtual void abort() {}       // this rank leaves the run between two collectives: wake whoever waits for it
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
    std::string err;
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
    bool allreduce_u64(uint64_t* dev, size_t n, hipStream_t s) override {
        return ok(rccl().AllReduce(dev, dev, n, kNcclUint64, kNcclSum, comm, s), "ncclAllReduce");
    }
    bool alltoallv_u64(const uint64_t* send, const size_t* scount, const size_t* sdispl, uint64_t* recv, const size_t* rcount,
                       const size_t* rdispl, hipStream_t s) override {
        return ok(rccl().AllToAllv(send, scount, sdispl, recv, rcount, rdispl, kNcclUint64, comm, s), "ncclAllToAllv");
    }
    bool gatherv_bytes(const void* send, size_t n, void* recv, const size_t* count, const size_t* displ, int root, hipStream_t s) override {
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
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> ptr;      // what every rank published for the collective in progress
    std::vector<const size_t*> cnt, dsp;
    std::vector<std::vector<uint64_t>> host;  // allreduce staging
    std::atomic<bool> failed{false};   // a collective of the run in progress went wrong on some rank
    std::atomic<bool> aborted{false};  // a rank left bdx_dist_run early: nobody may wait for it at a barrier
    explicit ThreadGroup(int w) : world(w), ptr(w), cnt(w), dsp(w), host(w) {}
    // false: the group was aborted (by a rank that gave up between two collectives) -- the caller fails its collective
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted.load()) return false;
        const uint64_t g = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != g || aborted.load(); });
        return generation != g;
    }
    void abort() {
        { std::lock_guard<std::mutex> lk(mu); aborted.store(true); }
        cv.notify_all();
    }
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
    void begin_run() override { g->begin_run(rank); }
    void abort() override { g->abort(); }
    bool allreduce_u64(uint64_t* dev, size_t n, hipStream_t s) override {
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
    DevBuf b_words, b_tab, b_send, b_recv, b_pack, b_all, b_nsend, b_nrecv, b_ntab, b_nflag, b_foreign, b_rg_rec, b_rg_pk, b_x, b_merge, b_chk;
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
    if (comm->world > kMaxRanks || ntids >= (1 << 24) - 1) return BDX_ELIMIT;
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
    size_t flags = 0, tidtab = 0, first = 0, rtab = 0, cnts = 0, misc = 0, up_off = 0, up_tail = 0, up_misc = 0, up_cur = 0, up_stats = 0, words = 0;
    TabLayout(int ntids, int ncols, int nkeys, int ncnt, int world) {
        size_t o = 16;
        tidtab = o; o += (size_t)(ntids + 1) * (1 + ncols) + 2;
        first = o; o += (size_t)ntids * 4;
        rtab = o; o += (size_t)ntids + 3;
        cnts = o; o += (size_t)2 * world;
        misc = o; o += 16;
        // staging of the small tables that go UP to the device: pinned, so that the copies are asynchronous and the vectors they were
        // built in need not outlive them (every table has its own place: nothing is overwritten within a run)
        up_off = o; o += (size_t)ntids * (1 + nkeys);
        up_tail = o; o += (size_t)ntids * 4;
        up_misc = o; o += (size_t)3 * ntids + 1;
        up_cur = o; o += (size_t)2 * world;
        up_stats = o; o += (size_t)2 + ncnt + 64;
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
                      &d->b_rg_rec, &d->b_rg_pk, &d->b_x, &d->b_merge, &d->b_chk})
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
        DHIP(d, d->h_words.ensure(((size_t)d->ntids * (ncols + 4) + ncnt + d->nbams + 2 * (size_t)world * world + 64) * 8));
        DHIP(d, d->b_words.ensure(((size_t)d->ntids * (ncols + 4) + ncnt + d->nbams + 2 * (size_t)world * world + 64) * 8));
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
            DHIP(d, d->b_x.ensure((nr / 8 + nr / 64 + (size_t)d->comm->world + 64) * 8));
            const size_t nn = (size_t)prior;
            DHIP(d, d->b_nsend.ensure(nn * 16)); DHIP(d, d->b_nrecv.ensure(nn * 16));
            DHIP(d, d->b_ntab.ensure((size_t)k7_names_slots(nn) * 16));
            DHIP(d, d->b_pack.ensure(nn * 40)); DHIP(d, d->b_all.ensure(nn * 40));
            DHIP(d, d->b_foreign.ensure(nn * 20 / 8 + 64));
            DHIP(d, d->b_send.ensure(nn * 4)); DHIP(d, d->b_recv.ensure(nn * 4));   // (an eighth of the anomalous reads inter-chromosomal and travelling)
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
        "pass1_and_chromosome_table", "allreduce_statistics", "compaction_and_rebase", "allreduce_first_reads", "region_cut", "allreduce_regions",
        "globalize_and_exchange_counts", "allreduce_exchange_sizes", "exchange_join_pair_groups", "allreduce_taint_and_windows",
        "components_walk", "allreduce_host_share", "host_share_and_table", "allreduce_table_sizes", "rank0_only_merge", "replay_route",
        "rank0_only_host_walk", ""};
    return i >= 0 && i < kDistPhases ? names[i] : "";
}

int bdx_dist_set_collect_support(bdx_dist* d, int on) {
    if (!d) return BDX_EINVAL;
    d->collect_support = on != 0;
    return BDX_OK;
}

bdx_ctx* bdx_dist_result(bdx_dist* d) { return d && d->ran && d->comm->rank == 0 ? d->util : nullptr; }

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
    U->materialized = C->materialized;
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
    // the same for words that live in HBM (`words` of them, `world` more behind them for the status): the payload stays on the device
    auto exchange_dev = [&](uint64_t* dev, size_t words) -> int {
        Stamp stamp{d, n_phase, std::chrono::steady_clock::now()};
        std::vector<uint64_t> tail((size_t)world, 0);
        tail[(size_t)rank] = (uint64_t)st.rc;
        bool good = hipMemcpyAsync(dev + words, tail.data(), (size_t)world * 8, hipMemcpyHostToDevice, s) == hipSuccess;
        good = good && comm.allreduce_u64(dev, words + (size_t)world, s);
        good = good && hipMemcpyAsync(tail.data(), dev + words, (size_t)world * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        if (!good) { comm.abort(); return dfail(d, BDX_EHIP, comm.err.empty() ? "all-reduce of device words" : comm.err); }
        return agreed_failure(d, st, tail.data(), world);
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
    // device tables: tid_off [ntids][1 + nkeys] | tid_tail [ntids][4] | roff [ntids] | owner [ntids] | tid_start [ntids + 1] | cnt [4 world] | n_total [4] | rbase u64 [ntids + 1]
    const size_t o_off = 0, o_tail = o_off + (size_t)ntids * (1 + nkeys), o_roff = o_tail + (size_t)ntids * 4, o_owner = o_roff + ntids,
                 o_start = o_owner + ntids, o_cnt = o_start + ntids + 1, o_ntot = o_cnt + 4 * (size_t)world, o_rbase = (o_ntot + 4 + 1) / 2 * 2,
                 tab_words = o_rbase + 2 * ((size_t)ntids + 1);
    DHIP(d, d->b_tab.ensure(tab_words * 4));
    uint32_t* T = d->b_tab.as<uint32_t>();

    // ---- pass 1 over all of this rank's chromosomes; where each chromosome starts and what the counters read there.  C1 ----
    const size_t tw = (size_t)ncols;  // per chromosome: anomalous reads, normal pairs, proper reads per key
    const size_t at_tot = (size_t)ncnt + nbams, at_reads = at_tot + (size_t)ntids * tw, at_claim = at_reads + ntids, at_owner = at_claim + ntids,
                 at_flags = at_owner + ntids;
    std::vector<uint64_t> v1(at_flags + 3, 0);
    std::vector<uint32_t> tidtab((size_t)(ntids + 1) * (1 + ncols), 0);   // [t][0] first read, [t][1 + c] counters in front of it
    phase([&]() -> int {
        C->table_in_hbm = world > 1;
        C->groups_in_hbm = world > 1;
        C->defer_walk = world > 1;
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
    const uint32_t na = C->p1.n_anom;   // this rank's

    // ---- compaction; the counters of every chromosome start where the chromosomes in front of it (anybody's) left them; what the
    // exchange will carry, per destination.  C2: every chromosome's first anomalous read, and the exchange's count matrices ----
    const size_t W2 = (size_t)world * world;
    const size_t at_mat = (size_t)ntids * 3;
    std::vector<uint64_t> v2(at_mat + 2 * W2, 0);   // (send-count matrices of the CTX records and the census records: row = sender)
    std::vector<uint32_t> h_cnt(world, 0), h_ncnt(world, 0);
    ExchangeSrc xs{};
    uint32_t* UP = H;   // (the staging places of L.up_*)
    phase([&]() -> int {
        DCTX(d, C, set_pass1(C, cnt_g.data(), covered, window, false));
        {   // the statistics every kernel from here on runs with: window and covered length (Pass1's first words), flag histogram, densities
            uint32_t* st_up = UP + L.up_stats;
            st_up[0] = covered; st_up[1] = (uint32_t)window;
            memcpy(st_up + 2, cnt_g.data(), (size_t)ncnt * 4);
            memcpy(st_up + 2 + ncnt, C->key_density.data(), C->key_density.size() * 4);
            // (these tables, the exchange counters' start values and -- with anomalous reads -- the chromosomes' offsets and owners go up in
            // ONE launch that reads them from the pinned report area: as six copy / fill commands they were six 5 us blits with their gaps)
            UploadList ul{};
            ul.copy(C->b_p1.p, st_up, 2);
            ul.copy(C->b_cnt.p, st_up + 2, (size_t)ncnt);
            ul.copy(C->b_kdens.p, st_up + 2 + ncnt, C->key_density.size());
            // (the exchange counters and the error words of the later kernels: also on a rank without anomalous reads -- rank 0 checks the word
            // k8_place_regions leaves whether or not it holds reads itself)
            ul.fill(T + o_cnt, 0u, (size_t)world * 4 + 4);
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
            }
            launch_k9_upload(ul, s);
        }
        DCTX(d, C, do_compact(C, 0, nullptr, true));
        trace("compaction");
        if (!na) { C->k4 = K4Arrays{}; return BDX_OK; }
        memset(H + L.first, 0, (size_t)ntids * 16);
        launch_k9_rebase(C->cp, &C->b_p1.as<Pass1>()->n_anom, na, nkeys, T + o_off, H + L.first, s);
        xs.key = C->cp.key; xs.check = C->cp.check; xs.meta = C->cp.meta; xs.tid = C->cp.tid; xs.idx = C->cp.idx; xs.mtid_col = C->d.mtid;
        xs.region_of = nullptr; xs.n_ptr = &C->b_p1.as<Pass1>()->n_anom; xs.owner_of_tid = (const int32_t*)(T + o_owner);
        xs.ntids = ntids; xs.me = rank; xs.world = (uint32_t)world;
        if (world == 1) {   // (a lone rank: nothing travels, nothing is counted or packed for the exchange; the ready word behind the rebase)
            H[L.cnts] = 0; H[L.cnts + 1] = 0;
            launch_k9_signal(H + 1, d->seq, s);
        } else {
            launch_k7_count(xs, na, T + o_cnt, s);
            launch_k9_report(T + o_cnt, H + L.cnts, 2 * (uint32_t)world, H + 1, d->seq, s);
        }
        if (!wait_word(flags + 1, d->seq)) {
            DHIP(d, hipStreamSynchronize(s));
            if (flags[1] != d->seq) return dfail(d, BDX_EINTERNAL, "the chromosomes' first reads did not arrive: its kernels were not launched");
        }
        trace("rebase and counts");
        for (int t = 0; t < ntids; ++t) {
            const uint32_t* f = H + L.first + (size_t)t * 4;
            if (f[0]) { v2[(size_t)t * 3] = 1; v2[(size_t)t * 3 + 1] = f[1]; v2[(size_t)t * 3 + 2] = f[2]; }
        }
        for (int q = 0; q < world; ++q) { h_cnt[q] = H[L.cnts + q]; h_ncnt[q] = H[L.cnts + world + q]; }
        for (int q = 0; q < world; ++q) { v2[at_mat + (size_t)rank * world + q] = h_cnt[q]; v2[at_mat + W2 + (size_t)rank * world + q] = h_ncnt[q]; }
        return BDX_OK;
    });
    rc = exchange(v2);
    if (rc != BDX_OK) return rc;
    std::vector<size_t> scount(world), sdispl(world), rcount(world), rdispl(world);
    std::vector<size_t> nscount(world), nsdispl(world), nrcount(world), nrdispl(world);   // the name census: two words per read
    size_t nsend = 0, nrecv = 0, nnsend = 0, nnrecv = 0;
    constexpr size_t kxw = sizeof(ExchangeEntry) / 8;
    for (int q = 0; q < world; ++q) { scount[q] = (size_t)h_cnt[q] * kxw; sdispl[q] = nsend * kxw; nsend += h_cnt[q]; }
    for (int q = 0; q < world; ++q) { nscount[q] = (size_t)h_ncnt[q] * 2; nsdispl[q] = nnsend * 2; nnsend += h_ncnt[q]; }
    for (int q = 0; q < world; ++q) { rcount[q] = (size_t)v2[at_mat + (size_t)q * world + rank] * kxw; rdispl[q] = nrecv * kxw; nrecv += v2[at_mat + (size_t)q * world + rank]; }
    for (int q = 0; q < world; ++q) { nrcount[q] = (size_t)v2[at_mat + W2 + (size_t)q * world + rank] * 2; nrdispl[q] = nnrecv * 2; nnrecv += v2[at_mat + W2 + (size_t)q * world + rank]; }
    // One rank: nothing travels -- its join sees every sighting of every name itself (a third one is K4's to report, as in bdx_run), so the
    // census of names is not taken, and neither all-to-all is called (the CTX records of a lone rank are all its own: none is packed).
    const bool solo = world == 1;
    if (solo) nnrecv = 0;
    {
        // (the column sums are the same table on every rank: the limits trip everywhere at once)
        for (int r = 0; r < world; ++r) {
            uint64_t col = 0, ncol = 0;
            for (int q = 0; q < world; ++q) { col += v2[at_mat + (size_t)q * world + r]; ncol += v2[at_mat + W2 + (size_t)q * world + r]; }
            if (col > (1u << 28) || ncol > 0x7FFFFFFFull) return dfail(d, BDX_ELIMIT, "too many inter-chromosomal join records / names on one rank");
        }
    }

    // ---- regions; the first anomalous read of the next chromosome that has one closes a chromosome's last candidate.  C3 ----
    std::vector<uint64_t> v3((size_t)ntids + 1, 0);
    std::vector<uint32_t> rtab((size_t)ntids + 3, 0);   // this rank's regions: first region of every chromosome, count, last_maxq
    phase([&]() -> int {
        // (the exchange's buffers, sized by counts that are known since C2: a rank without reads still receives census records)
        DHIP(d, d->b_send.ensure(std::max<size_t>(nsend, 1) * sizeof(ExchangeEntry)));
        DHIP(d, d->b_nsend.ensure(std::max<size_t>(nnsend, 1) * 16));
        DHIP(d, d->b_recv.ensure(std::max<size_t>(nrecv, 1) * sizeof(ExchangeEntry)));
        DHIP(d, d->b_nrecv.ensure(std::max<size_t>(nnrecv, 1) * 16));
        if (!na) return BDX_OK;
        {
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
            DHIP(d, hipMemcpyAsync(T + o_tail, tails, (size_t)ntids * 16, hipMemcpyHostToDevice, s));
        }
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
    // ---- genome-wide region ids, the CTX records and the census records packed per destination; C4: ONE all-to-all each.  No
    // host round trip from here to the pair groups: failures of this stretch travel with the next all-reduce ----
    uint64_t* X = nullptr;          // [NW] window read lengths | taint bytes (capG) | status words
    const size_t x_taint = NW, x_words = (size_t)NW + ((size_t)capG + 7) / 8;
    RegionRec* regs = nullptr;      // rank 0: the genome's region table (pinned: the copy runs beside the joins)
    uint32_t* pk = nullptr;
    bool regs_pending = false;
    int local_rc = BDX_OK;          // (what `phase` would record: this stretch ends in point-to-point collectives, which cannot carry it)
    {
        const auto tp = std::chrono::steady_clock::now();
        auto body = [&]() -> int {
            DHIP(d, d->b_rg_rec.ensure((size_t)NR * sizeof(RegionRec)));
            DHIP(d, d->b_rg_pk.ensure(std::max<size_t>((size_t)NR * nkeys2 * 4, 16)));
            DHIP(d, C->b_out_deg.ensure((size_t)capG * 6 * 4));
            DHIP(d, d->b_x.ensure((x_words + (size_t)world + 8) * 8));
            X = d->b_x.as<uint64_t>();
            DHIP(d, hipMemsetAsync(d->b_rg_rec.p, 0, (size_t)NR * sizeof(RegionRec), s));
            DHIP(d, hipMemsetAsync(X, 0, (x_words + (size_t)world) * 8, s));
            {   // (the chromosomes' region offsets and the exchange's cursors: one launch, as above)
                UploadList ul{};
                uint32_t* ur = UP + L.up_misc + (size_t)2 * ntids + 1;
                for (int t = 0; t < ntids; ++t) ur[t] = (uint32_t)rbase[t] - rtab[t];
                ul.copy(T + o_roff, ur, (size_t)ntids);
                if (na) {
                    uint32_t* cur = UP + L.up_cur;
                    for (int q = 0; q < world; ++q) { cur[q] = (uint32_t)(sdispl[q] / kxw); cur[(size_t)world + q] = (uint32_t)(nsdispl[q] / 2); }
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
            launch_k9_window_collect(d->b_rg_rec.as<RegionRec>(), (uint32_t)NR, period, (unsigned long long*)X, s);
            C->k6_cap = (uint32_t)NR; C->k6_r_rec = d->b_rg_rec.as<RegionRec>(); C->k6_r_pk = d->b_rg_pk.as<uint32_t>();
            C->k6_taint = world > 1 ? (uint8_t*)(X + x_taint) : nullptr;
            if (na && world > 1) {
                xs.region_of = C->k3.region_of;
                launch_k7_scatter(xs, na, T + o_cnt + 2 * (size_t)world, d->b_send.as<ExchangeEntry>(), d->b_nsend.as<unsigned long long>(), s);
            }
            trace("scatter");
            return BDX_OK;
        };
        d->err.clear();
        local_rc = st.rc == BDX_OK ? body() : BDX_OK;
        if (local_rc != BDX_OK) { st.rc = local_rc; st.msg = d->err; }
        d->phase_ms[6] += ms_between(tp, std::chrono::steady_clock::now());
        n_phase = 4;   // (slots 6 / 7: this stretch and the all-to-alls)
    }
    if (st.rc != BDX_OK && world > 1) {
        // (a rank that cannot take part in the all-to-all: the others would wait for it -- the communicator is given up)
        return leave(dfail(d, st.rc, st.msg));
    }
    if (st.rc != BDX_OK) return dfail(d, st.rc, st.msg);
    {
        const auto tp = std::chrono::steady_clock::now();
        // C4: the all-to-all of the CTX records (four 64-bit words each), then the census records
        if (solo && nsend) return leave(dfail(d, BDX_EINTERNAL, "a lone rank packed inter-chromosomal records for another"));
        if (!solo && !comm.alltoallv_u64(d->b_send.as<uint64_t>(), scount.data(), sdispl.data(), d->b_recv.as<uint64_t>(), rcount.data(), rdispl.data(), s))
            return leave(dfail(d, BDX_EHIP, comm.err));
        if (!solo && !comm.alltoallv_u64(d->b_nsend.as<uint64_t>(), nscount.data(), nsdispl.data(), d->b_nrecv.as<uint64_t>(), nrcount.data(), nrdispl.data(), s))
            return leave(dfail(d, BDX_EHIP, comm.err));
        d->ctx_sent = nsend; d->ctx_received = nrecv;
        d->phase_ms[7] += ms_between(tp, std::chrono::steady_clock::now());
    }

    // ---- the genome's region table on rank 0 (C5: a gather of the ranks' dense tables; with one rank it is there already) ----
    size_t region_bytes = 0;
    {
        std::vector<size_t> gcount(world), gdispl(world);
        const size_t rrec = sizeof(RegionRec), rpk = (size_t)nkeys2 * 4;
        for (int q = 0; q < world; ++q) { gcount[q] = round_up((size_t)nr_of_rank[q] * (rrec + rpk), 8); gdispl[q] = region_bytes; region_bytes += gcount[q]; }
        if (world > 1) {
            const size_t mine = gcount[rank];
            if (d->b_pack.ensure(std::max<size_t>(mine, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "region package"));
            if (nr_local) {
                if (hipMemcpyAsync(d->b_pack.p, C->b_r_rec.p, (size_t)nr_local * rrec, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                    hipMemcpyAsync((char*)d->b_pack.p + (size_t)nr_local * rrec, C->b_r_pk.p, (size_t)nr_local * rpk, hipMemcpyDeviceToDevice, s) != hipSuccess)
                    return leave(dfail(d, BDX_EHIP, "region package"));
            }
            if ((uint64_t)nr_local != nr_of_rank[rank]) return leave(dfail(d, BDX_EINTERNAL, "region counts of the chromosomes do not add up"));
            if (rank == 0 && d->b_all.ensure(std::max<size_t>(region_bytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
            if (!comm.gatherv_bytes(d->b_pack.p, mine, d->b_all.p, gcount.data(), gdispl.data(), 0, s)) return leave(dfail(d, BDX_EHIP, comm.err));
            d->gathered_bytes += region_bytes;
        }
        if (rank == 0) {
            if (C->h_regs.ensure((size_t)NR * rrec) != hipSuccess || C->h_pk.ensure(std::max<size_t>((size_t)NR * rpk, 16)) != hipSuccess)
                return leave(dfail(d, BDX_ENOMEM, "region table"));
            regs = C->h_regs.as<RegionRec>();
            pk = C->h_pk.as<uint32_t>();
            const RegionRec* src_rec = d->b_rg_rec.as<RegionRec>();
            const uint32_t* src_pk = d->b_rg_pk.as<uint32_t>();
            if (world > 1) {
                GatherDesc D{};
                D.world = world;
                uint32_t max_nr = 0;
                for (int q = 0; q < world; ++q) {
                    const size_t nr = (size_t)nr_of_rank[q];
                    D.p[q] = GatherPackage{gdispl[q], gdispl[q] + nr * rrec, 0, (uint32_t)nr, 0};
                    max_nr = std::max(max_nr, (uint32_t)nr);
                }
                if (U->b_r_rec.ensure((size_t)NR * rrec) != hipSuccess || U->b_r_pk.ensure(std::max<size_t>((size_t)NR * rpk, 16)) != hipSuccess)
                    return leave(dfail(d, BDX_ENOMEM, "region table"));
                if (hipMemcpyAsync(T + o_rbase, rbase.data(), ((size_t)ntids + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess) return leave(dfail(d, BDX_EHIP, "region table"));
                launch_k8_place_regions((const char*)d->b_all.p, D, max_nr, (const uint64_t*)(T + o_rbase), ntids, nkeys2, U->b_r_rec.as<RegionRec>(), U->b_r_pk.as<uint32_t>(),
                                        T + o_ntot + 1, s);
                src_rec = U->b_r_rec.as<RegionRec>(); src_pk = U->b_r_pk.as<uint32_t>();
            }
            // (on the context's second stream, behind what has been enqueued so far: the joins and the pair groups do not wait for it)
            // (written into the pinned table by a kernel, not by copy commands: the first device-to-host copy command of a process sets up a
            // copy-engine queue -- 6 ms in front of the joins of a process's first run, BDX_DIST_TRACE -- and the kernel's stores cross PCIe
            // as the result tables of bdx_run do)
            if (hipEventRecord(C->ev_copy, s) != hipSuccess || hipStreamWaitEvent(C->copy_stream, C->ev_copy, 0) != hipSuccess)
                return leave(dfail(d, BDX_EHIP, "region table"));
            {
                static_assert(sizeof(RegionRec) % 4 == 0, "copied by words");
                UploadList ul{};
                ul.copy(regs, src_rec, (size_t)NR * rrec / 4);
                if (nkeys2) ul.copy(pk, src_pk, (size_t)NR * rpk / 4);
                launch_k9_upload(ul, C->copy_stream);
            }
            regs_pending = true;
        }
    }

    auto wait_regions = [&]() -> int {
        if (regs_pending) {
            DHIP(d, hipStreamSynchronize(C->copy_stream));
            regs_pending = false;
            if (world > 1) {
                uint32_t perr = 0;
                DHIP(d, hipMemcpy(&perr, T + o_ntot + 1, 4, hipMemcpyDeviceToHost));
                if (perr) return dfail(d, BDX_EINTERNAL, "region table of the gather does not add up");
            }
        }
        return BDX_OK;
    };

    // ---- the joins (own reads + foreign entries), the name census, the pair groups per region.  C6: taint bytes, window lengths ----
    // (a negative -s: shifted region ids are the read-level walk's business below; the pair model's device walk is not enqueued for them, as in bdx_run)
    const bool ph_opt = 0 > d->opts.min_len && 0.0f < (float)d->opts.seq_coverage_lim;
    const bool force_host = (rank == 0 ? U->host_walk_only : false) || C->host_walk_only || d->opts.min_read_pair < 1 || ph_opt;
    auto t_x1 = std::chrono::steady_clock::now();
    phase([&]() -> int {
        if (nnrecv) {   // the census of the names this rank owns: on the second stream, beside the joins (its verdict is read at the next all-reduce)
            const uint32_t slots = k7_names_slots(nnrecv);
            DHIP(d, d->b_ntab.ensure((size_t)slots * 16));
            DHIP(d, d->b_nflag.ensure(16));
            DHIP(d, hipEventRecord(d->ev_side, s));
            DHIP(d, hipStreamWaitEvent(C->copy_stream, d->ev_side, 0));
            launch_k7_names_clear(d->b_ntab.as<unsigned long long>(), slots, d->b_nflag.as<uint32_t>(), C->copy_stream);
            launch_k7_names_census(d->b_nrecv.as<unsigned long long>(), (uint32_t)nnrecv, d->b_ntab.as<unsigned long long>(), slots, d->b_nflag.as<uint32_t>(),
                                   C->copy_stream);
        }
        if (na) {
            if ((uint64_t)C->na_alloc + nrecv > (1u << 28)) return dfail(d, BDX_ELIMIT, "too many join entries on one rank");
            const uint32_t nf = (uint32_t)nrecv;
            DHIP(d, d->b_foreign.ensure(std::max<size_t>(nf, 1) * 20 + 64));
            uint64_t* fkey = d->b_foreign.as<uint64_t>();
            uint64_t* fcheck = fkey + std::max<size_t>(nf, 1);
            int32_t* fregion = (int32_t*)(fcheck + std::max<size_t>(nf, 1));
            launch_k7_unpack(d->b_recv.as<ExchangeEntry>(), nf, fkey, with_check ? fcheck : nullptr, fregion, &C->b_p1.as<Pass1>()->n_anom, T + o_ntot, s);
            Entries en{};
            en.key = C->cp.key; en.check = C->cp.check; en.region = C->k3.region_of; en.meta = C->cp.meta; en.isize = C->cp.isize;
            en.n_local = &C->b_p1.as<Pass1>()->n_anom; en.fkey = fkey; en.fcheck = fcheck; en.fregion = fregion; en.want_pair_lo = 1;
            DCTX(d, C, do_join_local(C, C->na_alloc + nf, en, T + o_ntot, true));
        }
        trace("join");
        t_x1 = std::chrono::steady_clock::now();
        DCTX(d, C, do_k6(C, force_host, 1));
        trace("pair groups");
        return BDX_OK;
    });
    if (world > 1) {
        rc = exchange_dev(X, x_words);
        if (rc != BDX_OK) return rc;
    } else {
        ++n_phase;
    }
    d->ms_exchange = ms_between(t_x0, t_x1);

    // ---- components, the device's share of the walk; the host's share (pair groups of the components that are large, or span
    // ranks) goes to rank 0.  C7: its size per rank, and whether a name misbehaved ----
    uint64_t irregular = 0;
    phase([&]() -> int {
        if (world > 1) launch_k9_window_apply(d->b_rg_rec.as<RegionRec>(), (uint32_t)NR, period, (const unsigned long long*)X, s);
        DCTX(d, C, do_k6(C, force_host, 2));
        trace("components and walk");
        if (rank == 0) {   // while the device walks: the genome's region table for the host's share of the walk and for the result
            const int wr = wait_regions();
            if (wr != BDX_OK) return wr;
            decode_regions(C, regs, pk, (uint32_t)NR, 0, true);   // (read where it arrived, in pinned memory: the host's share of the walk touches little of it)
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
        return BDX_OK;
    });
    std::vector<uint64_t> v5((size_t)world + 1, 0);
    if (st.rc == BDX_OK) v5[(size_t)rank] = C->counts.n_groups;
    v5[(size_t)world] = irregular;
    rc = exchange(v5);
    if (rc != BDX_OK) return rc;
    // some rank met a read name more than twice -- or the caller wants the reads behind every SV, which only the read-level walk knows
    // ... or -s is negative: the very first anomalous read of the genome then registers a read-less region 0 (BreakDancer.cpp:216-231,
    // 244-252; bdx_run's `ph`), every real region's id shifts by one and with it the flush cadence -- the read-level walk knows how
    // (ReadWalkInput::phantom), the pair model's kernels do not
    const uint32_t ph = (na_all && ph_opt) ? 1u : 0u;
    const bool replay = v5[(size_t)world] != 0 || want_support != 0 || ph != 0;


    // ---- a read name seen more than twice (clashing names across merged files): the pair model does not hold, and the reference's
    // behaviour (ReadRegionData.cpp:108-113,152-175, SvBuilder.cpp:101-118) depends on the order of ALL sightings.  Every rank sends
    // the compact records of its chromosomes (name key, genome-wide region id, meta, |isize|, tid, second name hash, index in the
    // chromosome's stream: 40 bytes per anomalous read) to rank 0, which replays the run read by read (H2, bdx_walk_reads.cpp) on the
    // gathered region table.  The same route serves bdx_dist_set_collect_support: the supporting reads of an SV (-g / -d dumps,
    // BreakDancer.cpp:514-534) are known to the read-level walk only ----
    if (replay) {
        const auto t_r0 = std::chrono::steady_clock::now();
        constexpr size_t kw = 5;
        std::vector<size_t> rp_count(world), rp_displ(world);
        size_t rp_bytes = 0;
        std::vector<uint64_t> na_of_rank(world, 0);
        for (int t = 0; t < ntids; ++t)
            if (owner[t] >= 0) na_of_rank[owner[t]] += tot(t, 0);
        for (int q = 0; q < world; ++q) { rp_count[q] = (size_t)na_of_rank[q] * kw * 8; rp_displ[q] = rp_bytes; rp_bytes += rp_count[q]; }
        DHIP(d, hipStreamSynchronize(s));   // (K6's kernels were enqueued on the pair model: let them finish, their results are dropped)
        if (d->b_pack.ensure(std::max<size_t>(rp_count[rank], 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "record package"));
        if (na) launch_k9_pack_replay(C->cp, C->k3.region_of, &C->b_p1.as<Pass1>()->n_anom, na, T + o_start, d->b_pack.as<unsigned long long>(), s);
        if (rank == 0 && d->b_all.ensure(std::max<size_t>(rp_bytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
        if (rank == 0) { const int wr = wait_regions(); if (wr != BDX_OK) return leave(wr); }   // (the region package sat in b_all)
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

    // ---- C8: the host's share -> rank 0; every rank finishes its own table ----
    std::vector<size_t> hcount(world), hdispl(world);
    size_t hbytes = 0, ng_all = 0;
    for (int q = 0; q < world; ++q) { hcount[q] = (size_t)v5[q] * sizeof(GroupRec); hdispl[q] = hbytes; hbytes += hcount[q]; ng_all += (size_t)v5[q]; }
    std::vector<GroupRec> host_groups;
    if (world > 1) {
        const void* mine = C->k4.g_rec ? (const void*)C->k4.g_rec : d->b_pack.p;
        if (rank == 0) { const int wr = wait_regions(); if (wr != BDX_OK) return leave(wr); }   // (the region package sat in b_all)
        if (rank == 0 && d->b_all.ensure(std::max<size_t>(hbytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
        if (!comm.gatherv_bytes(mine, hcount[rank], d->b_all.p, hcount.data(), hdispl.data(), 0, s)) return leave(dfail(d, BDX_EHIP, comm.err));
        d->gathered_bytes += hbytes;
        if (rank == 0 && ng_all) {
            host_groups.resize(ng_all);
            if (hipMemcpyAsync(host_groups.data(), d->b_all.p, hbytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
                return leave(dfail(d, BDX_EHIP, "host share of the pair groups"));
        }
    }
    uint64_t mine_counts[8] = {0};
    phase([&]() -> int {
        C->walk.clear();
        // (several ranks: the device's share of the walk starts here, behind the gather of the host's share -- C7's all-reduce and that
        // gather did not have to wait for it, and rank 0 walks the gathered groups while every device walks its own)
        if (C->defer_walk) DCTX(d, C, do_k6(C, force_host, 3));
        if (rank == 0) {
            const GroupRec* g = world > 1 ? host_groups.data() : C->h_groups.as<GroupRec>();
            decode_groups(C, g, (uint32_t)ng_all, 0);
            C->last_big_groups = (int64_t)ng_all + C->counts.n_groups_big;
            const auto tw0 = std::chrono::steady_clock::now();
            DCTX(d, C, host_walk(C, lm, na_all != 0));
            d->phase_ms[16] = ms_between(tw0, std::chrono::steady_clock::now());
        }
        C->counts.last_maxq = lm;
        trace("host walk");
        DCTX(d, C, do_k6_table(C));
        DCTX(d, C, finish_table(C));
        trace("table");
        mine_counts[0] = C->n_sv_total; mine_counts[1] = C->n_terms_total; mine_counts[2] = C->n_cn_total; mine_counts[3] = C->n_printed;
        mine_counts[4] = C->n_sv_host; mine_counts[5] = C->counts.n_pairs; mine_counts[6] = C->n_groups_total; mine_counts[7] = C->counts.n_old;
        return BDX_OK;
    });
    if (world == 1) {
        adopt_table(U, C);
        d->table_lent = true;
        U->reg = C->reg; U->nreg = C->nreg; U->rpk = C->rpk;   // (the genome's region table stays in this rank's pinned buffers until the next run)
        C->reg = nullptr; C->nreg = 0; C->rpk = nullptr;
        U->counts.n_regions = (uint32_t)NR;
        U->ran = true; U->stage = 4;
        d->phase_ms[14] = 0;
        return finish_result();
    }
    std::vector<uint64_t> v6((size_t)world * 8, 0);
    if (st.rc == BDX_OK)
        for (int k = 0; k < 8; ++k) v6[(size_t)rank * 8 + k] = mine_counts[k];
    rc = exchange(v6);
    if (rc != BDX_OK) return rc;

    // ---- C9: the ranks' tables -> rank 0, which merges them by order key into the result context's pinned buffers ----
    const auto t_m0 = std::chrono::steady_clock::now();
    TableDesc TD{};
    TD.world = world;
    std::vector<size_t> tcount(world), tdispl(world);
    size_t tbytes = 0;
    uint64_t n_sv_all = 0, n_terms_all = 0, n_cn_all = 0, n_printed_all = 0, n_pairs_all = 0, n_groups_all = 0, n_old_all = 0;
    uint32_t max_sv = 0;
    for (int q = 0; q < world; ++q) {
        const size_t nsv = (size_t)v6[(size_t)q * 8], nt = (size_t)v6[(size_t)q * 8 + 1], nc = (size_t)v6[(size_t)q * 8 + 2];
        size_t o = tbytes;
        TablePackage& P = TD.p[q];
        P.n_sv = (uint32_t)nsv; P.n_terms = (uint32_t)nt; P.n_cn = (uint32_t)nc;
        P.rows_off = o; o += round_up(nsv * sizeof(SvOut), 8);
        P.keys_off = o; o += nsv * 8;
        P.lib_index_off = o; o += round_up(nt * 4, 8);
        P.lib_pairs_off = o; o += round_up(nt * 4, 8);
        P.ltail_off = o; o += nt * 8;
        P.cn_key_off = o; o += round_up(nc * 4, 8);
        P.cn_value_off = o; o += round_up(nc * 4, 8);
        tdispl[q] = tbytes; tcount[q] = o - tbytes; tbytes = o;
        n_sv_all += nsv; n_terms_all += nt; n_cn_all += nc; n_printed_all += v6[(size_t)q * 8 + 3];
        n_pairs_all += v6[(size_t)q * 8 + 5]; n_groups_all += v6[(size_t)q * 8 + 6]; n_old_all += v6[(size_t)q * 8 + 7];
        max_sv = std::max(max_sv, (uint32_t)nsv);
    }
    if (max_sv >= (1u << 26) || n_sv_all > 0xFFFFFFF0ull) return dfail(d, BDX_ELIMIT, "too many SV candidates for the merge");
    {
        const TablePackage& P = TD.p[rank];
        if (d->b_pack.ensure(std::max<size_t>(tcount[rank], 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "table package"));
        char* pp = (char*)d->b_pack.p - tdispl[rank];
        bool good = true;
        auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes && good) good = hipMemcpyAsync(pp + off, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess; };
        put(P.rows_off, C->b_sv_out.p, (size_t)P.n_sv * sizeof(SvOut)); put(P.keys_off, C->b_sv_key.p, (size_t)P.n_sv * 8);
        put(P.lib_index_off, C->b_lib_index_out.p, (size_t)P.n_terms * 4); put(P.lib_pairs_off, C->b_lib_pairs_out.p, (size_t)P.n_terms * 4);
        put(P.ltail_off, C->b_ltail_out.p, (size_t)P.n_terms * 8);
        put(P.cn_key_off, C->b_cn_key_out.p, (size_t)P.n_cn * 4); put(P.cn_value_off, C->b_cn_value_out.p, (size_t)P.n_cn * 4);
        if (!good) return leave(dfail(d, BDX_EHIP, "table package"));
        trace("table packed");
        if (rank == 0 && d->b_all.ensure(std::max<size_t>(tbytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
        if (!comm.gatherv_bytes(d->b_pack.p, tcount[rank], d->b_all.p, tcount.data(), tdispl.data(), 0, s)) return leave(dfail(d, BDX_EHIP, comm.err));
        d->gathered_bytes += tbytes;
        trace("tables gathered");
    }
    if (rank == 0) {
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
            MergeOut mo{U->h_sv_out.as<SvOut>(), U->h_lib_index.as<int32_t>(), U->h_lib_pairs.as<int32_t>(), U->h_ltail_dev.as<double>(),
                        U->h_cn_key.as<int32_t>(), U->h_cn_value.as<float>()};
            if (tracing) {   // are the ranks' tables sorted by key, and are the keys distinct?
                DHIP(d, hipStreamSynchronize(s));
                for (int q = 0; q < world; ++q) {
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
            launch_k9_merge_tables((const char*)d->b_all.p, d_td, world, n_total, max_sv, src, begins, ws, T + o_ntot + 2, mo, s);
        }
        DHIP(d, hipStreamSynchronize(s));
        DHIP(d, hipGetLastError());
        trace("merge");
        U->walk.clear(); U->log_tail.clear();
        U->n_sv_total = n_total; U->n_terms_total = (uint32_t)n_terms_all; U->n_cn_total = (uint32_t)n_cn_all; U->n_printed = (uint32_t)n_printed_all;
        U->n_sv_host = (uint32_t)v6[4]; U->n_groups_total = (uint32_t)n_groups_all;
        memset(&U->counts, 0, sizeof(U->counts));
        U->counts.n_regions = (uint32_t)NR; U->counts.last_maxq = lm; U->counts.n_pairs = (uint32_t)n_pairs_all; U->counts.n_old = (uint32_t)n_old_all;
        U->counts.n_sv_dev = n_total - U->n_sv_host; U->counts.n_groups = (uint32_t)ng_all;
        U->materialized = false;
        U->reg = C->reg; U->nreg = C->nreg; U->rpk = C->rpk;   // (in rank 0's pinned buffers until the next run)
        C->reg = nullptr; C->nreg = 0; C->rpk = nullptr;
        if (d->opts.fisher) {  // Fisher's combination (BreakDancer.cpp:71-81) uses the host's exp / log
            materialize(U);
            finish_scores(U->opts, U->log_tail.data(), U->walk.svs.data(), U->walk.svs.size(), &U->n_printed);
        }
        U->ran = true; U->stage = 4;
    }
    d->phase_ms[14] = ms_between(t_m0, std::chrono::steady_clock::now());
    return finish_result();
}

}  // extern "C"
