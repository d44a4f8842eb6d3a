// Chromosome-sharded runs: one whole-genome result from chromosomes spread over several GPUs (include/bdx.h, bdx_dist_*).
// Included at the end of bdx_api.hip (it drives the stage functions of that translation unit).
//
// The path shards by chromosome -- regions never span tids (BreakDancer.cpp:216) -- and what a single breakdancer-max run
// couples across chromosomes is small: the pass-1 statistics (window, lambda, densities: BamSummary.cpp:129-150,
// BreakDancerMax.cpp:83-116), the running counters sampled at region boundaries, the read that closes a chromosome's
// last candidate region (BreakDancer.cpp:202-231), the region numbering / flush cadence (BreakDancer.cpp:254-259), and the
// inter-chromosomal read pairs (-t, ARP_CTX).  Every rank (one per GPU) runs K1-K4 on its own chromosomes; between the
// stages the ranks exchange
//   C1  an all-reduce of the pass-1 counters, per-file reference lengths and per-chromosome totals,
//   C2  an all-reduce of each chromosome's first anomalous read and C3 of its region count / last read length
//       (every table entry is owned by exactly one rank, so a sum is a gather),
//   C4  ONE all-to-all of the CTX join records to owner(name key) (k7_exchange.hip): the only exchange on the data path.
//       Pairs with both mates on one chromosome never leave their GPU,
//   C5  a gather of the region tables and pair groups to rank 0, which walks the region graph (build_connection is
//       inherently ordered: BreakDancer.cpp:266-346) and scores the candidates (K5).
// Payloads stay in HBM: the collectives run on device buffers through RCCL (ncclAllReduce, ncclAllToAllv, grouped
// ncclSend / ncclRecv), xGMI between the GPUs of a node.  A second backend runs the ranks as threads of one process
// (tests on a single GPU; a host program that drives several GPUs itself).
#include <dlfcn.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>

namespace {

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
    std::map<int, bdx_ctx*> chrom;   // the chromosomes this rank owns
    bdx_ctx* util = nullptr;         // joins the CTX records this rank owns, and on rank 0 walks and holds the result
    DevBuf b_words, b_cnt, b_send, b_recv, b_pack, b_all, b_gin;
    DevBuf b_nsend, b_nrecv, b_ntab, b_ninfo, b_nft, b_nflag;   // the name census (k7_exchange.hip)
    std::string err;
    uint64_t ctx_sent = 0, ctx_received = 0, gathered_bytes = 0;
    float ms_total = 0, ms_exchange = 0;
    bool ran = false;
    float phase_ms[12] = {0};        // bdx_dist_get_phase_ms: where the last run's time went on this rank
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
int allreduce_host(bdx_dist* d, std::vector<uint64_t>& v) {
    if (v.empty()) return BDX_OK;
    hipStream_t s = d->util->stream;
    DHIP(d, d->b_words.ensure(v.size() * 8));
    DHIP(d, hipMemcpyAsync(d->b_words.p, v.data(), v.size() * 8, hipMemcpyHostToDevice, s));
    if (!d->comm->allreduce_u64(d->b_words.as<uint64_t>(), v.size(), s)) return dfail(d, BDX_EHIP, d->comm->err);
    DHIP(d, hipMemcpyAsync(v.data(), d->b_words.p, v.size() * 8, hipMemcpyDeviceToHost, s));
    DHIP(d, hipStreamSynchronize(s));
    return BDX_OK;
}

int dist_create_common(bdx_dist** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids, int w0, int device,
                       std::unique_ptr<Comm> comm) {
    if (!out || !opts || !libs || nlibs < 1 || nbams < 1 || ntids < 1) return BDX_EINVAL;
    if (opts->min_len < 0) return BDX_ELIMIT;  // (a negative -s registers a read-less region 0: single-context runs only)
    if (comm->world > kMaxRanks) return BDX_ELIMIT;
    bdx_dist* d = new (std::nothrow) bdx_dist;
    if (!d) return BDX_ENOMEM;
    d->device = device; d->ntids = ntids; d->nlibs = nlibs; d->nbams = nbams; d->w0 = w0;
    d->opts = *opts;
    d->libs.assign(libs, libs + nlibs);
    d->nkeys = opts->cn_lib ? nlibs : nbams;
    d->comm = std::move(comm);
    const int rc = bdx_create(&d->util, opts, libs, nlibs, nbams, ntids, w0, device);
    if (rc != BDX_OK) { delete d; return rc; }
    *out = d;
    return BDX_OK;
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
    for (int r = 0; r < world; ++r) {
        std::unique_ptr<ThreadComm> c(new ThreadComm);
        c->rank = r; c->world = world; c->g = g;
        const int rc = dist_create_common(&out[r], opts, libs, nlibs, nbams, ntids, max_read_window_size0, devices[r], std::move(c));
        if (rc != BDX_OK) {
            for (int q = 0; q < r; ++q) { bdx_dist_destroy(out[q]); out[q] = nullptr; }
            return rc;
        }
    }
    return BDX_OK;
}

void bdx_dist_destroy(bdx_dist* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    for (auto& kv : d->chrom) bdx_destroy(kv.second);
    if (d->util) bdx_destroy(d->util);
    for (DevBuf* b : {&d->b_words, &d->b_cnt, &d->b_send, &d->b_recv, &d->b_pack, &d->b_all, &d->b_gin, &d->b_nsend, &d->b_nrecv, &d->b_ntab, &d->b_ninfo,
                      &d->b_nft, &d->b_nflag})
        b->release();
    delete d;
}

const char* bdx_dist_last_error(const bdx_dist* d) { return d ? d->err.c_str() : ""; }
int bdx_dist_rank(const bdx_dist* d) { return d ? d->comm->rank : -1; }
int bdx_dist_world(const bdx_dist* d) { return d ? d->comm->world : 0; }

bdx_ctx* bdx_dist_chromosome(bdx_dist* d, int tid) {
    if (!d || tid < 0 || tid >= d->ntids) return nullptr;
    auto f = d->chrom.find(tid);
    if (f != d->chrom.end()) return f->second;
    bdx_ctx* c = nullptr;
    if (bdx_create(&c, &d->opts, d->libs.data(), d->nlibs, d->nbams, d->ntids, d->w0, d->device) != BDX_OK) return nullptr;
    c->groups_in_hbm = true;   // (its pair groups go to rank 0 from HBM)
    d->chrom[tid] = c;
    return c;
}

int bdx_dist_get_phase_ms(const bdx_dist* d, float* out, int n) {
    if (!d || !out) return BDX_EINVAL;
    for (int i = 0; i < n; ++i) out[i] = i < 12 ? d->phase_ms[i] : 0.0f;
    return BDX_OK;
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

static int agreed_failure(bdx_dist* d, const RunStatus& st, const std::vector<uint64_t>& v, size_t at, int world) {
    int first = -1, code = BDX_OK;
    for (int q = 0; q < world; ++q)
        if (v[at + q]) { first = q; code = (int)v[at + q]; break; }
    if (first < 0) return BDX_OK;
    if (st.rc != BDX_OK) return dfail(d, st.rc, st.msg);
    return dfail(d, code, "rank " + std::to_string(first) + " failed (" + bdx_strerror(code) + "); all ranks stop");
}

int bdx_dist_run(bdx_dist* d) {
    if (!d) return BDX_EINVAL;
    const auto t_begin = std::chrono::steady_clock::now();
    Comm& comm = *d->comm;
    const int world = comm.world, rank = comm.rank;
    const int nlibs = d->nlibs, nbams = d->nbams, nkeys = d->nkeys, ntids = d->ntids;
    const int ncnt = nlibs * kNumFlags + nlibs + nbams;
    DHIP(d, hipSetDevice(d->device));
    bdx_ctx* U = d->util;
    hipStream_t us = U->stream;
    d->ran = false;
    d->ctx_sent = d->ctx_received = d->gathered_bytes = 0;
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
        if (2 * n_phase < 12) d->phase_ms[2 * n_phase] += ms_between(tp, std::chrono::steady_clock::now());
    };
    // all-reduce of v with the ranks' status words appended; afterwards every rank knows whether anybody failed
    auto exchange = [&](std::vector<uint64_t>& v) -> int {
        const auto tp = std::chrono::steady_clock::now();
        struct Stamp { bdx_dist* d; int& n; std::chrono::steady_clock::time_point t; ~Stamp() { if (2 * n + 1 < 12) d->phase_ms[2 * n + 1] += ms_between(t, std::chrono::steady_clock::now()); ++n; } } stamp{d, n_phase, tp};
        const size_t at = v.size();
        v.resize(at + (size_t)world, 0);
        v[at + (size_t)rank] = (uint64_t)st.rc;
        const int rc = allreduce_host(d, v);
        if (rc != BDX_OK) { comm.abort(); return rc; }
        const int f = agreed_failure(d, st, v, at, world);
        v.resize(at);
        return f;
    };
    auto leave = [&](int rc) { comm.abort(); return rc; };  // failures past the last foldable point

    // ---- pass 1 on the own chromosomes; C1: counters, per-file reference lengths, per-chromosome totals ----
    const size_t tw = 2 + (size_t)nkeys;  // per chromosome: anomalous reads, normal pairs, proper reads per key
    const size_t at_reads = (size_t)ncnt + nbams + (size_t)ntids * tw;  // then: reads per chromosome; ranks that want the supporting reads
    std::vector<uint64_t> v1(at_reads + (size_t)ntids + 1, 0);
    v1[at_reads + (size_t)ntids] = d->collect_support ? 1 : 0;
    phase([&]() -> int {
        for (auto& kv : d->chrom) DCTX(d, kv.second, do_pass1(kv.second, 0, false, false));  // all enqueued, then waited for
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            DCTX(d, c, wait_pass1(c));
            for (int i = 0; i < ncnt; ++i) v1[i] += c->cnt_local[i];
            for (int b = 0; b < nbams; ++b) v1[ncnt + b] += c->p1.ref_len[b];
            uint64_t* t = &v1[(size_t)ncnt + nbams + (size_t)kv.first * tw];
            t[0] = c->p1.n_anom; t[1] = c->p1.n_normal;
            for (int k = 0; k < nkeys; ++k) t[2 + k] = c->p1.key_tot[k];
            v1[at_reads + (size_t)kv.first] = c->n;
        }
        DCTX(d, U, do_pass1(U));  // (no reads: brings the utility context's buffers up)
        return BDX_OK;
    });
    int rc = exchange(v1);
    if (rc != BDX_OK) return rc;
    const uint64_t want_support = v1[at_reads + (size_t)ntids];
    if (want_support != 0 && want_support != (uint64_t)world) return dfail(d, BDX_EINVAL, "bdx_dist_set_collect_support is set on some ranks only");
    std::vector<uint64_t> read_base((size_t)ntids + 1, 0);   // a chromosome's first read in the merged stream (position sorted: chromosomes ascending)
    for (int t = 0; t < ntids; ++t) read_base[(size_t)t + 1] = read_base[t] + v1[at_reads + (size_t)t];
    std::vector<uint32_t> cnt_g(ncnt);
    for (int i = 0; i < ncnt; ++i) cnt_g[i] = (uint32_t)v1[i];
    uint32_t covered = 0;  // BamSummary.cpp:123-126: a uint32 maximum compared against each file's size_t sum
    for (int b = 0; b < nbams; ++b)
        if ((uint64_t)covered < v1[ncnt + b]) covered = (uint32_t)v1[ncnt + b];
    const int32_t window = window_from(U, cnt_g.data(), covered);
    auto tot = [&](int tid, int k) { return v1[(size_t)ncnt + nbams + (size_t)tid * tw + k]; };
    std::vector<uint64_t> base((size_t)(ntids + 1) * tw, 0);  // exclusive prefix over the chromosomes in stream order
    for (int t = 0; t < ntids; ++t)
        for (size_t k = 0; k < tw; ++k) base[(size_t)(t + 1) * tw + k] = base[(size_t)t * tw + k] + tot(t, (int)k);
    // (the same sum on every rank: all of them return here, together)
    if (base[(size_t)ntids * tw] > kMaxAnomalous) return dfail(d, BDX_ELIMIT, "more than 2^31 anomalous reads in one run");

    // ---- compaction with the counters of the chromosomes in front; C2: every chromosome's first anomalous read ----
    std::vector<uint64_t> v2((size_t)ntids * 3, 0);
    phase([&]() -> int {
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            DCTX(d, c, set_pass1(c, cnt_g.data(), covered, window, true));
            std::vector<uint32_t> pkb(nkeys);
            for (int k = 0; k < nkeys; ++k) pkb[k] = (uint32_t)base[(size_t)kv.first * tw + 2 + k];
            DCTX(d, c, do_compact(c, (uint32_t)base[(size_t)kv.first * tw + 1], pkb.data(), false));
        }
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            if (!c->p1.n_anom) continue;
            uint32_t meta = 0, nn = 0;
            DHIP(d, hipMemcpyAsync(&meta, c->cp.meta, 4, hipMemcpyDeviceToHost, c->stream));
            DHIP(d, hipMemcpyAsync(&nn, c->cp.nn, 4, hipMemcpyDeviceToHost, c->stream));
            DHIP(d, hipStreamSynchronize(c->stream));
            uint64_t* t = &v2[(size_t)kv.first * 3];
            t[0] = 1 | (c->use_check ? 2 : 0); t[1] = (uint64_t)meta_qlen(meta); t[2] = nn;
        }
        return BDX_OK;
    });
    rc = exchange(v2);
    if (rc != BDX_OK) return rc;
    // (every chromosome has one owner, so the sums are the owners' words) the second name hash: on all chromosomes or on none
    bool with_check = false, without_check = false;
    for (int t = 0; t < ntids; ++t) {
        if (v2[(size_t)t * 3] & 2) with_check = true;
        else if (v2[(size_t)t * 3] & 1) without_check = true;
    }
    if (with_check && without_check) return dfail(d, BDX_EINVAL, "bdx_use_name_check is set on some chromosomes' contexts only");

    // ---- regions; the first anomalous read of the next chromosome closes a chromosome's last candidate.  C3 ----
    std::vector<int> next_anom(ntids, -1);
    for (int t = ntids - 1, nx = -1; t >= 0; --t) {
        next_anom[t] = nx;
        if (tot(t, 0) > 0) nx = t;
    }
    std::vector<uint64_t> v3((size_t)ntids * 2, 0);
    uint32_t* d_cnt = nullptr;      // [world] records per destination
    uint32_t* d_cur = nullptr;      // [world] scatter cursors
    uint32_t *d_ncnt = nullptr, *d_ncur = nullptr;
    std::vector<uint32_t> h_cnt(world, 0), h_ncnt(world, 0);
    phase([&]() -> int {
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            const int nx = next_anom[kv.first];
            if (nx >= 0) DCTX(d, c, do_cut(c, 1, (int32_t)v2[(size_t)nx * 3 + 1], (uint32_t)v2[(size_t)nx * 3 + 2], false, true));
            else DCTX(d, c, do_cut(c, 0, 0, 0, false, true));
        }
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            DCTX(d, c, readback(c, false));
            v3[(size_t)kv.first * 2] = c->counts.n_regions;
            v3[(size_t)kv.first * 2 + 1] = (uint32_t)c->counts.last_maxq;
        }
        // ---- joins: pairs within a chromosome where they are; CTX records to owner(name key) ----
        DHIP(d, d->b_cnt.ensure((size_t)world * 16 + 64));
        d_cnt = d->b_cnt.as<uint32_t>();
        d_cur = d_cnt + world;
        d_ncnt = d_cnt + 2 * world;    // the name census: every anomalous read's key
        d_ncur = d_cnt + 3 * world;
        DHIP(d, hipMemsetAsync(d_cnt, 0, (size_t)world * 16, us));
        DHIP(d, hipStreamSynchronize(us));
        // (the chromosomes' streams are independent: the counts are complete once each has been waited for, below)
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            const uint32_t na = c->p1.n_anom;
            if (na) launch_k7_count(c->cp.key, c->cp.meta, &c->b_p1.as<Pass1>()->n_anom, na, (uint32_t)world, d_cnt, c->stream);
            if (na) launch_k7_names_count(c->cp.key, &c->b_p1.as<Pass1>()->n_anom, na, (uint32_t)world, d_ncnt, c->stream);
        }
        for (auto& kv : d->chrom) DHIP(d, hipStreamSynchronize(kv.second->stream));
        DHIP(d, hipMemcpy(h_cnt.data(), d_cnt, (size_t)world * 4, hipMemcpyDeviceToHost));
        DHIP(d, hipMemcpy(h_ncnt.data(), d_ncnt, (size_t)world * 4, hipMemcpyDeviceToHost));
        return BDX_OK;
    });
    rc = exchange(v3);
    if (rc != BDX_OK) return rc;
    std::vector<uint64_t> rbase(ntids + 1, 0);
    for (int t = 0; t < ntids; ++t) rbase[t + 1] = rbase[t] + v3[(size_t)t * 2];
    const uint64_t NR = rbase[ntids];
    if (NR > kMaxRegions) return dfail(d, BDX_ELIMIT, "too many regions for the packed group key");  // (all ranks alike)
    int last_anom_tid = -1;
    for (int t = 0; t < ntids; ++t)
        if (tot(t, 0) > 0) last_anom_tid = t;

    const auto t_x0 = std::chrono::steady_clock::now();
    // Everything a rank can do before it knows what the others send happens in front of the count exchange, so that its
    // failure still travels with it: the send buffer (sized by the rank's own counts), the chromosomes' own joins, the
    // packing of the CTX records per destination.
    std::vector<size_t> scount(world), sdispl(world), rcount(world), rdispl(world);
    std::vector<size_t> nscount(world), nsdispl(world), nrcount(world), nrdispl(world);   // the name census: two words per read
    size_t nsend = 0, nrecv = 0, nnsend = 0, nnrecv = 0;
    constexpr size_t kxw = sizeof(ExchangeEntry) / 8;
    for (int q = 0; q < world; ++q) { scount[q] = (size_t)h_cnt[q] * kxw; sdispl[q] = nsend * kxw; nsend += h_cnt[q]; }
    for (int q = 0; q < world; ++q) { nscount[q] = (size_t)h_ncnt[q] * 2; nsdispl[q] = nnsend * 2; nnsend += h_ncnt[q]; }
    phase([&]() -> int {
        DHIP(d, d->b_send.ensure(std::max<size_t>(nsend, 1) * sizeof(ExchangeEntry)));
        DHIP(d, d->b_nsend.ensure(std::max<size_t>(nnsend, 1) * 16));
        {
            std::vector<uint32_t> cur(world);
            for (int q = 0; q < world; ++q) cur[q] = (uint32_t)(sdispl[q] / kxw);
            DHIP(d, hipMemcpy(d_cur, cur.data(), (size_t)world * 4, hipMemcpyHostToDevice));
            for (int q = 0; q < world; ++q) cur[q] = (uint32_t)(nsdispl[q] / 2);
            DHIP(d, hipMemcpy(d_ncur, cur.data(), (size_t)world * 4, hipMemcpyHostToDevice));
        }
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            const uint32_t na = c->p1.n_anom;
            if (!na) continue;
            // the chromosome's own pairs: K4 on its compact reads, pair groups with genome-wide region ids
            Entries en{};
            en.key = c->cp.key; en.check = c->cp.check; en.region = c->k3.region_of; en.meta = c->cp.meta; en.isize = c->cp.isize;
            en.region_base = (int32_t)rbase[kv.first];
            DCTX(d, c, do_join_local(c, na, en, &c->b_p1.as<Pass1>()->n_anom, false));
            launch_k7_scatter(c->cp.key, c->cp.check, c->k3.region_of, c->cp.meta, c->cp.isize, &c->b_p1.as<Pass1>()->n_anom, na, (uint32_t)world,
                              (uint32_t)base[(size_t)kv.first * tw], (int32_t)rbase[kv.first], d_cur, d->b_send.as<ExchangeEntry>(), c->stream);
            launch_k7_names_scatter(c->cp.key, c->cp.check, c->cp.meta, &c->b_p1.as<Pass1>()->n_anom, na, (uint32_t)world, (uint32_t)kv.first, d_ncur,
                                    d->b_nsend.as<unsigned long long>(), c->stream);
        }
        for (auto& kv : d->chrom) DHIP(d, hipStreamSynchronize(kv.second->stream));
        return BDX_OK;
    });
    std::vector<uint64_t> v4((size_t)world * world * 2, 0);  // send-count matrices (CTX records, census records): row = sender
    const size_t W2 = (size_t)world * world;
    for (int q = 0; q < world; ++q) { v4[(size_t)rank * world + q] = h_cnt[q]; v4[W2 + (size_t)rank * world + q] = h_ncnt[q]; }
    rc = exchange(v4);
    if (rc != BDX_OK) return rc;
    for (int q = 0; q < world; ++q) { rcount[q] = (size_t)v4[(size_t)q * world + rank] * kxw; rdispl[q] = nrecv * kxw; nrecv += v4[(size_t)q * world + rank]; }
    for (int q = 0; q < world; ++q) { nrcount[q] = (size_t)v4[W2 + (size_t)q * world + rank] * 2; nrdispl[q] = nnrecv * 2; nnrecv += v4[W2 + (size_t)q * world + rank]; }
    if (nnrecv > 0x7FFFFFFFull) return dfail(d, BDX_ELIMIT, "too many anomalous reads' names on one rank");
    {
        // (the column sums are the same table on every rank: the limit trips everywhere at once)
        for (int r = 0; r < world; ++r) {
            uint64_t col = 0;
            for (int q = 0; q < world; ++q) col += v4[(size_t)q * world + r];
            if (col > kMaxAnomalous) return dfail(d, BDX_ELIMIT, "too many inter-chromosomal join records on one rank");
        }
    }
    if (d->b_recv.ensure(std::max<size_t>(nrecv, 1) * sizeof(ExchangeEntry)) != hipSuccess)
        return leave(dfail(d, BDX_ENOMEM, "receive buffer of the all-to-all"));
    // C4: the all-to-all of the CTX records (three 64-bit words each)
    if (!comm.alltoallv_u64(d->b_send.as<uint64_t>(), scount.data(), sdispl.data(), d->b_recv.as<uint64_t>(), rcount.data(), rdispl.data(), us))
        return leave(dfail(d, BDX_EHIP, comm.err));
    if (d->b_nrecv.ensure(std::max<size_t>(nnrecv, 1) * 16) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "receive buffer of the name census"));
    if (!comm.alltoallv_u64(d->b_nsend.as<uint64_t>(), nscount.data(), nsdispl.data(), d->b_nrecv.as<uint64_t>(), nrcount.data(), nrdispl.data(), us))
        return leave(dfail(d, BDX_EHIP, comm.err));
    d->ctx_sent = nsend; d->ctx_received = nrecv;
    // join what arrived
    uint32_t ng_ctx = 0;
    auto t_x1 = std::chrono::steady_clock::now();
    size_t nreg_mine = 0, ng_mine = 0, pack_bytes = 0;
    uint64_t irregular = 0;
    const size_t rrec = sizeof(RegionRec), rpk = (size_t)2 * nkeys * 4, grec = sizeof(GroupRec);
    phase([&]() -> int {
        if (nnrecv) {   // the census of the names this rank owns
            uint32_t slots = 1024;
            while (slots < 2 * nnrecv) slots <<= 1;
            DHIP(d, d->b_ntab.ensure((size_t)slots * 8)); DHIP(d, d->b_ninfo.ensure((size_t)slots * 8)); DHIP(d, d->b_nft.ensure((size_t)slots * 4));
            DHIP(d, d->b_nflag.ensure(16));
            DHIP(d, hipMemsetAsync(d->b_ntab.p, 0xFF, (size_t)slots * 8, us));
            DHIP(d, hipMemsetAsync(d->b_ninfo.p, 0, (size_t)slots * 8, us));
            DHIP(d, hipMemsetAsync(d->b_nft.p, 0xFF, (size_t)slots * 4, us));
            DHIP(d, hipMemsetAsync(d->b_nflag.p, 0, 4, us));
            launch_k7_names_census(d->b_nrecv.as<unsigned long long>(), (uint32_t)nnrecv, d->b_ntab.as<unsigned long long>(), d->b_ninfo.as<unsigned long long>(),
                                   d->b_nft.as<uint32_t>(), slots - 1, d->b_nflag.as<uint32_t>(), us);
            uint32_t flag = 0;
            DHIP(d, hipMemcpyAsync(&flag, d->b_nflag.p, 4, hipMemcpyDeviceToHost, us));
            DHIP(d, hipStreamSynchronize(us));
            if (flag) irregular = 1;
        }
        if (nrecv) {
            const uint32_t n32 = (uint32_t)nrecv;
            DHIP(d, U->b_x_key.ensure(nrecv * 8)); DHIP(d, U->b_x_order.ensure(nrecv * 4)); DHIP(d, U->b_x_region.ensure(nrecv * 4));
            DHIP(d, U->b_x_meta.ensure(nrecv * 4)); DHIP(d, U->b_x_isize.ensure(nrecv * 4)); DHIP(d, U->b_x_n.ensure(16));
            if (with_check) DHIP(d, U->b_x_check.ensure(nrecv * 8));
            launch_k7_unpack(d->b_recv.as<ExchangeEntry>(), n32, U->b_x_key.as<uint64_t>(), with_check ? U->b_x_check.as<uint64_t>() : nullptr,
                             U->b_x_order.as<uint32_t>(), U->b_x_region.as<int32_t>(), U->b_x_meta.as<uint32_t>(), U->b_x_isize.as<int32_t>(), us);
            DHIP(d, hipMemcpyAsync(U->b_x_n.p, &n32, 4, hipMemcpyHostToDevice, us));
            DHIP(d, hipMemsetAsync(U->b_counts.p, 0, sizeof(StageCounts), us));
            Entries en{};
            en.key = U->b_x_key.as<uint64_t>(); en.region = U->b_x_region.as<int32_t>(); en.order = U->b_x_order.as<uint32_t>();
            en.check = with_check ? U->b_x_check.as<uint64_t>() : nullptr;
            en.meta = U->b_x_meta.as<uint32_t>(); en.isize = U->b_x_isize.as<int32_t>();
            U->groups_in_hbm = true;    // (packaged for rank 0 from HBM like the chromosomes' own)
            const int jrc = do_join_local(U, n32, en, U->b_x_n.as<uint32_t>(), false);
            U->groups_in_hbm = false;   // (rank 0's walk reads K6's groups on the host)
            if (jrc != BDX_OK) return dfail(d, jrc, "do_join_local: " + U->err);
            DHIP(d, hipMemcpyAsync(U->h_counts.p, U->b_counts.p, sizeof(StageCounts), hipMemcpyDeviceToHost, us));
            DHIP(d, hipStreamSynchronize(us));
            const StageCounts sc = *U->h_counts.as<StageCounts>();
            if (sc.irregular) irregular = 1;   // a read name seen more than twice: the run is replayed read by read on rank 0 (below)
            if (sc.overflow) return dfail(d, BDX_EINTERNAL, "group list overflow");
            ng_ctx = sc.n_groups;
        }
        t_x1 = std::chrono::steady_clock::now();

        // ---- C5: region tables and pair groups to rank 0 ----
        // this rank's package: per owned chromosome (ascending) its region records and prefix samples, then all pair groups
        ng_mine = ng_ctx;
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            if (!c->p1.n_anom) continue;
            DCTX(d, c, readback(c, true));
            if (c->counts.irregular) irregular = 1;
            nreg_mine += c->counts.n_regions;
            ng_mine += c->counts.n_groups;
        }
        pack_bytes = round_up(nreg_mine * (rrec + rpk) + ng_mine * grec, 8);
        DHIP(d, d->b_pack.ensure(std::max<size_t>(pack_bytes, 8)));
        char* p = (char*)d->b_pack.p;
        for (auto& kv : d->chrom) {  // records of all own chromosomes, then their prefix samples, then the groups
            bdx_ctx* c = kv.second;
            const size_t nr = c->p1.n_anom ? c->counts.n_regions : 0;
            if (nr) DHIP(d, hipMemcpyAsync(p, c->b_r_rec.p, nr * rrec, hipMemcpyDeviceToDevice, us));
            p += nr * rrec;
        }
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            const size_t nr = c->p1.n_anom ? c->counts.n_regions : 0;
            if (nr) DHIP(d, hipMemcpyAsync(p, c->b_r_pk.p, nr * rpk, hipMemcpyDeviceToDevice, us));
            p += nr * rpk;
        }
        for (auto& kv : d->chrom) {
            bdx_ctx* c = kv.second;
            const size_t ng = c->p1.n_anom ? c->counts.n_groups : 0;
            if (ng) DHIP(d, hipMemcpyAsync(p, c->k4.g_rec, ng * grec, hipMemcpyDefault, us));
            p += ng * grec;
        }
        if (ng_ctx) DHIP(d, hipMemcpyAsync(p, U->k4.g_rec, (size_t)ng_ctx * grec, hipMemcpyDefault, us));
        return BDX_OK;
    });
    std::vector<uint64_t> v5((size_t)world * 3 + 1, 0);
    if (st.rc == BDX_OK) { v5[(size_t)rank * 3] = nreg_mine; v5[(size_t)rank * 3 + 1] = ng_mine; v5[(size_t)rank * 3 + 2] = pack_bytes; }
    v5[(size_t)world * 3] = irregular;
    rc = exchange(v5);
    if (rc != BDX_OK) return rc;
    // some rank met a read name more than twice -- or the caller wants the reads behind every SV, which only the read-level walk knows
    const bool replay = v5[(size_t)world * 3] != 0 || want_support != 0;
    std::vector<size_t> gcount(world), gdispl(world);
    size_t all_bytes = 0, ng_all = 0;
    for (int q = 0; q < world; ++q) { gcount[q] = (size_t)v5[(size_t)q * 3 + 2]; gdispl[q] = all_bytes; all_bytes += gcount[q]; ng_all += v5[(size_t)q * 3 + 1]; }
    if (rank == 0 && d->b_all.ensure(std::max<size_t>(all_bytes, 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
    if (!comm.gatherv_bytes(d->b_pack.p, pack_bytes, d->b_all.p, gcount.data(), gdispl.data(), 0, us)) return leave(dfail(d, BDX_EHIP, comm.err));
    DHIP(d, hipStreamSynchronize(us));
    d->gathered_bytes = all_bytes;
    d->ms_exchange = ms_between(t_x0, t_x1);

    // ---- a read name seen more than twice (clashing names across merged files): the pair model does not hold, and the reference's
    // behaviour (ReadRegionData.cpp:108-113,152-175, SvBuilder.cpp:101-118) depends on the order of ALL sightings.  Every rank sends
    // the compact records of its chromosomes (name key, genome-wide region id, meta, |isize|, tid, second name hash, index in the
    // chromosome's stream: 40 bytes per anomalous read) to rank 0, which replays the run read by read (H2, bdx_walk_reads.cpp) on the
    // gathered region table.  The same route serves bdx_dist_set_collect_support: the supporting reads of an SV (-g / -d dumps,
    // BreakDancer.cpp:514-534) are known to the read-level walk only ----
    std::vector<uint64_t> rp_host;   // rank 0: all ranks' records
    std::vector<size_t> rp_count(world), rp_displ(world);
    if (replay) {
        std::vector<uint64_t> mine_rec;
        phase([&]() -> int {
            for (auto& kv : d->chrom) {
                bdx_ctx* c = kv.second;
                const uint32_t na = c->p1.n_anom;
                if (!na) continue;
                std::vector<uint64_t> key(na), chk(na, 0);
                std::vector<int32_t> reg(na), isz(na);
                std::vector<uint32_t> meta(na), ridx(na);
                DHIP(d, hipStreamSynchronize(c->stream));
                DHIP(d, hipMemcpy(ridx.data(), c->cp.idx, (size_t)na * 4, hipMemcpyDeviceToHost));
                DHIP(d, hipMemcpy(key.data(), c->cp.key, (size_t)na * 8, hipMemcpyDeviceToHost));
                if (c->cp.check) DHIP(d, hipMemcpy(chk.data(), c->cp.check, (size_t)na * 8, hipMemcpyDeviceToHost));
                DHIP(d, hipMemcpy(reg.data(), c->k3.region_of, (size_t)na * 4, hipMemcpyDeviceToHost));
                DHIP(d, hipMemcpy(meta.data(), c->cp.meta, (size_t)na * 4, hipMemcpyDeviceToHost));
                DHIP(d, hipMemcpy(isz.data(), c->cp.isize, (size_t)na * 4, hipMemcpyDeviceToHost));
                const uint64_t rb = rbase[kv.first];
                for (uint32_t j = 0; j < na; ++j) {
                    const uint32_t g = reg[j] < 0 ? 0xFFFFFFFFu : (uint32_t)(reg[j] + (int64_t)rb);
                    mine_rec.push_back(key[j]);
                    mine_rec.push_back((uint64_t)g | ((uint64_t)meta[j] << 32));
                    mine_rec.push_back((uint64_t)(uint32_t)isz[j] | ((uint64_t)(uint32_t)kv.first << 32));
                    mine_rec.push_back(chk[j]);
                    mine_rec.push_back(ridx[j]);
                }
            }
            DHIP(d, d->b_pack.ensure(std::max<size_t>(mine_rec.size() * 8, 8)));
            if (!mine_rec.empty()) DHIP(d, hipMemcpyAsync(d->b_pack.p, mine_rec.data(), mine_rec.size() * 8, hipMemcpyHostToDevice, us));
            DHIP(d, hipStreamSynchronize(us));
            return BDX_OK;
        });
        std::vector<uint64_t> v6(world, 0);
        if (st.rc == BDX_OK) v6[rank] = mine_rec.size() * 8;
        rc = exchange(v6);
        if (rc != BDX_OK) return rc;
        size_t rp_bytes = 0;
        for (int q = 0; q < world; ++q) { rp_count[q] = (size_t)v6[q]; rp_displ[q] = rp_bytes; rp_bytes += rp_count[q]; }
        // (rank 0's region package is still in b_all: copy it out before the buffer takes the records)
        std::vector<char> keep_regions;
        if (rank == 0 && all_bytes) {
            keep_regions.resize(all_bytes);
            if (hipMemcpy(keep_regions.data(), d->b_all.p, all_bytes, hipMemcpyDeviceToHost) != hipSuccess) return leave(dfail(d, BDX_EHIP, "gather buffer"));
        }
        if (rank == 0 && d->b_all.ensure(std::max<size_t>(std::max(rp_bytes, all_bytes), 8)) != hipSuccess) return leave(dfail(d, BDX_ENOMEM, "gather buffer"));
        if (!comm.gatherv_bytes(d->b_pack.p, mine_rec.size() * 8, d->b_all.p, rp_count.data(), rp_displ.data(), 0, us)) return leave(dfail(d, BDX_EHIP, comm.err));
        DHIP(d, hipStreamSynchronize(us));
        d->gathered_bytes += rp_bytes;
        if (rank == 0) {
            rp_host.resize(rp_bytes / 8);
            if (rp_bytes) DHIP(d, hipMemcpy(rp_host.data(), d->b_all.p, rp_bytes, hipMemcpyDeviceToHost));
            if (all_bytes) DHIP(d, hipMemcpy(d->b_all.p, keep_regions.data(), all_bytes, hipMemcpyHostToDevice));   // (read back below as usual)
        }
    }

    // (from here on nothing is collective any more: rank 0 finishes on its own)
    if (rank == 0) {
        // The packages sit in HBM (b_all): the region table is assembled there -- every record to rbase[tid] + its rank among its
        // chromosome's records (k8_place_regions) -- and, unless the run is replayed, gets K6's slot space by a scan; the host takes
        // ONE copy of the finished table (the getters and the host's share of the walk read it).
        const uint64_t na_all = base[(size_t)ntids * tw];
        const int nkeys2 = 2 * nkeys;
        GatherDesc D{};
        D.world = world;
        uint32_t max_nr = 0, max_ng = 0;
        size_t nr_sum = 0;
        for (int q = 0; q < world; ++q) {
            const size_t nr = (size_t)v5[(size_t)q * 3], ng = (size_t)v5[(size_t)q * 3 + 1];
            D.p[q] = GatherPackage{gdispl[q], gdispl[q] + nr * rrec, gdispl[q] + nr * (rrec + rpk), (uint32_t)nr, (uint32_t)ng};
            max_nr = std::max(max_nr, (uint32_t)nr); max_ng = std::max(max_ng, (uint32_t)ng);
            nr_sum += nr;
        }
        if (nr_sum != NR) return dfail(d, BDX_EINTERNAL, "region table of the gather does not add up");
        std::vector<RegionRec> regs(NR);
        std::vector<uint32_t> pk((size_t)NR * nkeys2);
        // scratch of the assembly, behind the bucketed groups: rbase[ntids + 1] | goff[NR + 2] | cnt[NR + 1] | scan ws | {n, err, slots}
        const size_t cap_alloc = (size_t)std::max<uint64_t>(std::max<uint64_t>(na_all, NR), 1);
        const size_t g_bytes = round_up(std::max<size_t>(ng_all, 1) * sizeof(GroupRec), 8);
        const size_t ws_words = scan_grid((uint32_t)std::max<uint64_t>(NR, 1)) + 2;
        DHIP(d, d->b_gin.ensure(g_bytes + ((size_t)ntids + 1) * 8 + ((size_t)NR + 2 + (size_t)NR + 1 + ws_words + 4) * 4));
        GroupRec* d_groups = d->b_gin.as<GroupRec>();
        uint64_t* d_rbase = (uint64_t*)((char*)d->b_gin.p + g_bytes);
        uint32_t* d_goff = (uint32_t*)(d_rbase + ntids + 1);
        uint32_t* d_gcnt = d_goff + NR + 2;
        uint32_t* d_ws = d_gcnt + NR + 1;
        uint32_t* d_misc = d_ws + ws_words;   // [0] number of regions, [1] error flag, [2] slots
        uint32_t misc[4] = {(uint32_t)NR, 0, 0, 0};
        if (NR) {
            DHIP(d, U->b_r_rec.ensure(cap_alloc * sizeof(RegionRec))); DHIP(d, U->b_r_pk.ensure(cap_alloc * nkeys2 * 4));
            DHIP(d, hipMemcpyAsync(d_rbase, rbase.data(), ((size_t)ntids + 1) * 8, hipMemcpyHostToDevice, us));
            DHIP(d, hipMemcpyAsync(d_misc, misc, 16, hipMemcpyHostToDevice, us));
            DHIP(d, hipMemsetAsync(d_gcnt, 0, ((size_t)NR + 1) * 4, us));
            launch_k8_place_regions((const char*)d->b_all.p, D, max_nr, d_rbase, ntids, nkeys2, U->b_r_rec.as<RegionRec>(), U->b_r_pk.as<uint32_t>(),
                                    d_misc + 1, us);
            if (!replay) launch_k8_slot_space(U->b_r_rec.as<RegionRec>(), (uint32_t)NR, d_misc, d_misc + 2, d_ws, us);
            DHIP(d, hipMemcpyAsync(regs.data(), U->b_r_rec.p, NR * sizeof(RegionRec), hipMemcpyDeviceToHost, us));
            DHIP(d, hipMemcpyAsync(pk.data(), U->b_r_pk.p, pk.size() * 4, hipMemcpyDeviceToHost, us));
            DHIP(d, hipMemcpyAsync(misc, d_misc, 16, hipMemcpyDeviceToHost, us));
            DHIP(d, hipStreamSynchronize(us));
            if (misc[1]) return dfail(d, BDX_EINTERNAL, "region table of the gather does not add up");
        }
        auto groups_to_host = [&](std::vector<GroupRec>& groups) -> int {   // the packages' pair groups as they came (the host-only walk)
            groups.resize(ng_all);
            size_t gi = 0;
            for (int q = 0; q < world; ++q) {
                if (D.p[q].ng) DHIP(d, hipMemcpy(&groups[gi], (const char*)d->b_all.p + D.p[q].groups_off, (size_t)D.p[q].ng * grec, hipMemcpyDeviceToHost));
                gi += D.p[q].ng;
            }
            return BDX_OK;
        };
        DCTX(d, U, set_pass1(U, cnt_g.data(), covered, window, true));
        U->n = 0;
        const int32_t lm = last_anom_tid >= 0 ? (int32_t)(uint32_t)v3[(size_t)last_anom_tid * 2 + 1] : 0;
        if (replay) {
            // the records of all chromosomes in stream order: chromosomes ascending, each rank's package holds its own in order
            constexpr size_t kw = 5;
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
                if (with_check) chk[o] = rp_host[i * kw + 3];
                if (want_support) sidx[o] = read_base[t] + rp_host[i * kw + 4];
            }
            // (a region's first read: its index in its chromosome's list -> in the genome-wide one)
            for (size_t r = 0; r < NR; ++r) regs[r].first += (uint32_t)base[(size_t)regs[r].tid * tw];
            decode_regions(U, regs.data(), pk.data(), (uint32_t)NR, 0, false);
            U->counts.n_regions = (uint32_t)NR;
            U->counts.last_maxq = lm;
            if (with_check) unify_names(key.data(), chk.data(), (size_t)na_all);
            std::vector<uint32_t> sup;
            U->collect_support = want_support != 0;
            DCTX(d, U, replay_arrays(U, (uint32_t)na_all, key.data(), reg.data(), meta.data(), isz.data(), 0, want_support ? &sup : nullptr));
            if (want_support) {   // compact indices -> indices in the merged stream, and the reads' flags
                U->sup_idx.resize(sup.size());
                U->sup_flag.resize(sup.size());
                for (size_t i = 0; i < sup.size(); ++i) { U->sup_idx[i] = sidx[sup[i]]; U->sup_flag[i] = (uint8_t)meta_flag(meta[sup[i]]); }
            }
            U->p1.n_anom = (uint32_t)na_all;
            d->ran = true;
            d->ms_total = ms_between(t_begin, std::chrono::steady_clock::now());
            return BDX_OK;
        }
        const bool host_only = U->host_walk_only;  // (bdx_set_host_walk on the result context's owner: the whole walk on the host)
        // slot space of K6: region r owns the slots [first, first + n) -- its reads' places in a single-context run; here
        // simply the regions laid end to end (every group owns at least one read of its later region, so they suffice)
        const uint64_t slots = misc[2];   // (k8_slot_space: regs[r].first = the slots of the regions before r)
        if (host_only || NR == 0 || ng_all == 0 || slots > kMaxAnomalous) {
            std::vector<GroupRec> groups;
            if (const int gr = groups_to_host(groups)) return gr;
            decode_regions(U, regs.data(), pk.data(), (uint32_t)NR, 0, false);
            decode_groups(U, groups.data(), (uint32_t)ng_all, 0);
            U->counts.n_regions = (uint32_t)NR;
            DCTX(d, U, host_walk(U, lm, last_anom_tid >= 0));
            DCTX(d, U, score_host_terms(U));
            DCTX(d, U, finish_host_walk(U));
        } else {
            // the device walk of a single-context run (K6), fed with the gathered groups bucketed by their later region: a histogram, a
            // scan and a scatter over the packages where they lie (k8_bucket_groups)
            const uint32_t cap = (uint32_t)slots;
            launch_k8_bucket_groups((const char*)d->b_all.p, D, max_ng, (uint32_t)NR, d_misc, d_gcnt, d_goff, d_groups, d_ws, d_misc + 1, us);
            DHIP(d, U->b_out_deg.ensure((size_t)cap * 6 * 4));
            DHIP(d, hipMemcpyAsync(U->b_cnt.p, cnt_g.data(), (size_t)ncnt * 4, hipMemcpyHostToDevice, us));
            DHIP(d, hipMemcpyAsync(U->b_kdens.p, U->key_density.data(), U->key_density.size() * 4, hipMemcpyHostToDevice, us));
            StageCounts sc{};
            sc.n_regions = (uint32_t)NR; sc.last_maxq = lm;
            DHIP(d, hipMemcpyAsync(U->b_counts.p, &sc, sizeof(sc), hipMemcpyHostToDevice, us));
            DHIP(d, hipMemcpyAsync(misc, d_misc, 16, hipMemcpyDeviceToHost, us));
            DHIP(d, hipStreamSynchronize(us));  // (host vectors above go out of use)
            if (misc[1]) return dfail(d, BDX_EINTERNAL, "pair groups of the gather name regions that do not exist");
            launch_k6_scratch_init(U->b_out_deg.as<uint32_t>(), cap, us);
            U->na_alloc = cap;
            U->k3 = K3Arrays{}; U->k4 = K4Arrays{}; U->cp = Compact{};
            U->k4.g_cap = (uint32_t)ng_all + 1;
            DHIP(d, U->h_groups.ensure((size_t)U->k4.g_cap * sizeof(GroupRec)));
            U->k4.g_rec = U->h_groups.as<GroupRec>();
            U->k6_in_groups = d_groups; U->k6_in_goff = d_goff;
            U->counts.last_maxq = lm;
            const bool force_host = U->host_walk_only || U->opts.min_read_pair < 1;
            int rk = do_k6(U, force_host);
            U->k6_in_groups = nullptr; U->k6_in_goff = nullptr;
            if (rk != BDX_OK) return dfail(d, rk, "do_k6: " + U->err);
            decode_regions(U, regs.data(), pk.data(), (uint32_t)NR, 0, false);
            if (!wait_flag(U, 1, U->seq)) {
                if (U->poll) DHIP(d, hipStreamSynchronize(us)); else DHIP(d, hipEventSynchronize(U->ev_groups));
            }
            U->counts = *U->h_counts.as<StageCounts>();
            if (U->counts.overflow) return dfail(d, BDX_EINTERNAL, "group list overflow");
            decode_groups(U, U->h_groups.as<GroupRec>(), U->counts.n_groups, 0);
            DCTX(d, U, host_walk(U, lm, true));
            DCTX(d, U, do_k6_table(U));
            DCTX(d, U, finish_table(U));
        }
        U->p1.n_anom = (uint32_t)base[(size_t)ntids * tw];
    } else {
        DCTX(d, U, set_pass1(U, cnt_g.data(), covered, window, true));
    }
    d->ran = true;
    d->ms_total = ms_between(t_begin, std::chrono::steady_clock::now());
    return BDX_OK;
}

}  // extern "C"
