// libbdx: context, HBM-resident read store, stage orchestration and the extern "C" boundary (include/bdx.h).
//
// One context = one GPU = one HIP stream.  The whole record stream stays resident in HBM (a 30x human
// genome is ~33 GB of SoA, well inside 288 GB), so pass 1 and pass 2 of the reference collapse into one
// read of the data: K1 -> finalize -> K2 -> K3 -> K4 on the device, one small readback, the host walk,
// K5 for the scores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <sys/mman.h>
#include <atomic>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/bdx.h"
#include "bdx_dev.h"
#include "bdx_k3.h"
#include "bdx_shard.h"
#include "bdx_scan.h"
#include "bdx_walk.h"
#include "bdx_bam_dev.h"

using namespace bdx;

namespace {

// BDX_ALLOC_TRACE=1: every allocation of a context's buffers with its size and duration on stderr
inline bool alloc_trace() { static const bool on = getenv("BDX_ALLOC_TRACE") != nullptr; return on; }

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t b) {
        if (b <= bytes) return hipSuccess;
        const auto t0 = std::chrono::steady_clock::now();
        const bool had = p != nullptr;
        if (p) (void)hipFree(p);   // (waits for the device to go idle: steady-state code must not get here)
        p = nullptr;
        bytes = 0;
        size_t want = b + b / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (alloc_trace()) fprintf(stderr, "[bdx alloc] device %12zu B %8.1f us%s\n", want, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), had ? " (regrown: hipFree first)" : "");
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    unsigned flags = hipHostMallocDefault;
    // A large buffer is anonymous memory on transparent huge pages, registered with the runtime (hipHostRegister): 0.02 s per 0.5 GB
    // against hipHostMalloc's 0.09-0.11 -- pinning is paid per page -- and a quarter less to hand back when the process ends
    // (tools/pin_probe.hip, profiles/r05_pin_probe.txt); the device sees it at the same address.  Small buffers -- the words the host
    // polls, the records kernels and host exchange mid-run -- stay with hipHostMalloc (fine-grained by default).  bdx_set_process_option("pin_malloc", 1): all of them.
    void* map_base = nullptr;
    size_t map_len = 0;
    static std::atomic<bool>& registered_switch() { static std::atomic<bool> on{true}; return on; }   // bdx_set_process_option("pin_malloc", 1) turns it off
    static bool use_registered() { return registered_switch().load(std::memory_order_relaxed); }
    hipError_t ensure(size_t b) {
        if (b <= bytes) return hipSuccess;
        release();
        size_t want = b + b / 8 + 256;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipErrorOutOfMemory;
        constexpr size_t kHuge = (size_t)2 << 20;
        if (want >= 2 * kHuge && flags == hipHostMallocDefault && use_registered()) {
            const size_t len = (want + kHuge - 1) & ~(kHuge - 1);
            void* base = mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (base != MAP_FAILED) {
                void* al = (void*)(((uintptr_t)base + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
                (void)madvise(al, len, MADV_HUGEPAGE);
                for (size_t o = 0; o < len; o += 4096) ((volatile char*)al)[o] = 0;   // (faulted in before it is pinned: one fault per huge page)
                void* dev = nullptr;
                if (hipHostRegister(al, len, hipHostRegisterMapped) == hipSuccess && hipHostGetDevicePointer(&dev, al, 0) == hipSuccess && dev == al) {
                    p = al; map_base = base; map_len = len + kHuge; want = len; e = hipSuccess;
                } else {
                    (void)hipHostUnregister(al);
                    (void)hipGetLastError();
                    munmap(base, len + kHuge);
                }
            }
        }
        if (e != hipSuccess) {
            e = hipHostMalloc(&p, want, flags);
            // (small buffers hold the words the host polls and the counters kernels report: a block the allocator hands out again may still
            // hold another context's ready word -- the same sequence number -- and a poll would return before the kernel has run)
            if (e == hipSuccess && want <= ((size_t)1 << 20)) memset(p, 0, want);
        }
        if (alloc_trace()) fprintf(stderr, "[bdx alloc] pinned %12zu B %8.1f us%s\n", want, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), map_base ? " (registered huge pages)" : "");
        if (e == hipSuccess) bytes = want; else p = nullptr;
        return e;
    }
    void release() {
        if (p && map_base) { (void)hipHostUnregister(p); munmap(map_base, map_len); }
        else if (p) (void)hipHostFree(p);
        p = nullptr; bytes = 0; map_base = nullptr; map_len = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

constexpr int kNumStages = 12;
constexpr int kK1MaxGrid = 8192;  // measured best on MI355X (tools/k1_probe.hip; again at the end of round 2, BDX_K1_GRID: 2048 74.6 us, 4096 70.9,
                                  // 8192 69.9, 12288 72.5, 16384 73.4): 256 CUs x 32 workgroups queued, 4 independent waves each

}  // namespace

struct bdx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bdx_opts opts{};
    std::vector<bdx_lib> libs;
    int nlibs = 0, nbams = 0, ntids = 0, nkeys = 0, w0 = 0;
    std::string err;

    // resident reads
    ReadsSoA d{};
    size_t n = 0, cap = 0;
    bool adopted = false;
    DevBuf b_tid, b_pos, b_mtid, b_mpos, b_isize, b_flag, b_qlen, b_mapq, b_lib, b_bam, b_key, b_check;
    bool groups_in_hbm = false;       // (sharded runs) the join's pair groups stay in HBM instead of pinned host memory
    bool defer_walk = false;          // (sharded runs over several ranks) do_k6 part 2 stops behind the components; the device walk (part 3) is
                                      // enqueued once the host's share has left for rank 0, and runs beside the collectives and rank 0's walk
    DevBuf b_groups;
    bool use_check = false;           // bdx_use_name_check: every batch carries a second hash of the read name, mates must agree in it too

    // stage buffers
    DevBuf b_libs, b_cls, b_tile_tot, b_tile_pre, b_tile_mono, b_blk_cnt, b_cnt, b_p1, b_fold, b_stash, b_chunk_tot;
    DevBuf b_c_tid, b_c_pos, b_c_isize, b_c_meta, b_c_key, b_c_check, b_c_idx, b_c_nn, b_c_pk;
    DevBuf b_cand, b_pre_q, b_pre_rev, b_pre_nonctx, b_c_first, b_c_maxq, b_c_rid, b_region_of, b_ws_u4, b_ws_u32, b_totals, b_counts;
    DevBuf b_bcnt, b_boff, b_bcur, b_e_key, b_e_idx, b_partner, b_t_key, b_t_idx;
    DevBuf b_x_key, b_x_check, b_x_order, b_x_region, b_x_meta, b_x_isize, b_x_n;
    DevBuf b_lib_mean;
    DevBuf b_r_rec, b_r_pk, b_out_deg, b_parts, b_kdens, b_rs, b_slot, b_members, b_own, b_lib_stage, b_cn_stage,
        b_t_lambda, b_t_k, b_ws6;
    PinBuf h_p1, h_cnt, h_counts, h_regs, h_pk, h_groups, h_terms;
    DevBuf b_sv_src, b_dlists, b_ltail, b_pair_lo;
    PinBuf h_hs_rec, h_hs_aux, h_hs_lists, h_printed;
    DevBuf b_ins, b_member_ids;
    PinBuf h_flags;                   // [0] pass 1 ready, [1] host's groups ready, [2] final table ready (= run sequence number)
    uint32_t seq = 0;
    // test / measurement switches (bdx_set_debug): all off by default
    int dbg_no_stash = 0, dbg_max_chunks = 0, dbg_finalize2_fold = 0, dbg_no_forward = 0, dbg_scan3 = 0, dbg_label_rounds = 0, dbg_k1_grid = 0,
        dbg_end_write_value = 0, dbg_walk_lanes = 0, dbg_ins_plain = 0, dbg_gather_walk = 0, dbg_region_dma = 0, dbg_join_fwd = 0, dbg_regions_copy = 0, dbg_asm_plain = 0;
    bool region_dma_now = false;      // this run's region table goes to the host by a copy command once the host knows its size (see bdx_run)
    uint32_t lb_seq = 0;              // launches of look-back scans so far: every launch stamps its words with its own number (bdx_scan.h)
    uint32_t k1_event_period = 4;     // K1 is bracketed by HIP events on every n-th run (an event pair idles the GPU ~10 us)
    float k1_ms_last = 0;
    bool k1_timed = false;
    bool stage_timing = false;        // HIP events around K2 / K3 / K4+K6 (each costs a few microseconds of idle GPU)
    bool poll = true;                 // BDX_NO_POLL=1: wait with stream / event synchronisation only
    bool materialized = true;         // c->walk holds the final table (false: it still sits in the pinned buffers only)
    bool rows_packed = false;         // ... as SvWire rows (48 bytes: what a single-context run's table kernel writes over PCIe), not SvOut
    uint32_t n_sv_total = 0, n_groups_total = 0, n_terms_total = 0, n_cn_total = 0;
    PinBuf h_counts0, h_counts2, h_sv_out, h_lib_index, h_lib_pairs, h_cn_key, h_cn_value, h_ltail_dev;
    hipEvent_t ev_groups = nullptr, ev_regions = nullptr;
    int big_walk_mode = -1;           // BDX_BIG_WALK=1 / 0: components of 5..64 regions always / never walked on the device; default: when
                                      // the host's share is large enough to matter (see do_k6)
    int64_t last_big_groups = -1;     // groups of such components in the previous run of this context (device + host share)
    FinalizeParams fp_deferred{};     // second level of the pass-1 finalisation, to be run by K2's launch (enqueue-ahead runs)
    bool finalize2_deferred = false;
    bool bucketed_join = false;       // BDX_BUCKETED_JOIN=1: use the partitioned LDS join at every size (it is the path for > 4 M entries)
    bool host_walk_only = false;      // BDX_HOST_WALK=1: every component goes through the host walk (A/B testing of K6)
    K6Arrays k6{};
    const GroupRec* k6_in_groups = nullptr;  // (a caller that holds pair groups as aggregates: K6 takes them instead of counting pairs)
    const uint32_t* k6_in_goff = nullptr;
    // sharded runs (bdx_dist_*): this context holds several chromosomes of a genome, in ascending order (bdx_shard.h)
    bool force_direct_join = false;   // the join takes foreign entries: the direct table at every size
    uint32_t k6_cap = 0;              // capacity of K6's per-region arrays: the GENOME's regions (0: this context's anomalous reads)
    const RegionRec* k6_r_rec = nullptr;   // the region table laid out by genome-wide id (this rank's regions, n == 0 elsewhere)
    const uint32_t* k6_r_pk = nullptr;
    uint8_t* k6_taint = nullptr;
    const uint32_t* k3_tid_tail = nullptr;
    bool table_in_hbm = false;        // the final table stays in HBM (with its order keys): rank 0 merges the ranks' tables
    DevBuf b_sv_out, b_lib_index_out, b_lib_pairs_out, b_cn_key_out, b_cn_value_out, b_ltail_out, b_sv_key;

    // results
    bool ran = false;
    Pass1 p1{};
    StageCounts counts{};
    std::vector<uint32_t> cnt;        // adopted counters: hist[nlibs*11], lib_cnt[nlibs], bam_cnt[nbams]
    std::vector<uint32_t> cnt_local;  // this context's own counters
    uint32_t g_covered = 0;
    int32_t g_window = 0;
    uint32_t nn_base = 0;
    int stage = 0;                    // 0 nothing, 1 pass 1 done, 2 statistics adopted, 3 regions cut, 4 walked
    uint32_t ntiles = 0, tstride = 0;
    Compact cp{};
    K3Arrays k3{};
    K4Arrays k4{};
    std::vector<double> log_tail;
    bool collect_support = false;
    std::vector<uint32_t> ov_cnt;     // bdx_set_pass1_statistics: restored pass-1 counters that replace the run's own
    uint32_t ov_covered = 0;
    bool replayed = false;            // the last run went through the read-level host replay (a read name seen more than twice)
    bool use_stash = false;           // K1 leaves ready-made records of the anomalous reads for K2 (at most kStashKeys counter keys)
    // Sizing passes (bdx_reserve, the BAM decoder's sizing thread): the stage functions called with a `Sizing` only grow the buffers of the
    // stages behind pass 1 -- they read the context's configuration and touch those buffers, nothing of its run state (no flag on the
    // context says "sizing": round 5's did, and a run beside the sizing thread saw it and launched nothing).  While one is in flight the
    // entry points that launch those stages refuse with BDX_ESTATE; its error text goes to sizing_err (the feeding thread owns `err`).
    std::atomic<int> sizing{0};
    std::string sizing_err;
    std::vector<uint32_t> sup_off;    // [n_svs + 1]
    std::vector<uint64_t> sup_idx;
    std::vector<uint8_t> sup_flag;
    std::vector<float> seqcov, lib_density, key_density;
    std::vector<HostRegion> regions;  // owned copies (staged runs, phantom shift); otherwise reg/rpk point into pinned memory
    std::vector<uint32_t> r_pk;
    const HostRegion* reg = nullptr;
    size_t nreg = 0;
    const uint32_t* rpk = nullptr;
    std::vector<GroupPart> parts;
    WalkResult walk;
    WalkScratch* walk_scratch = nullptr;
    uint32_t n_printed = 0;
    uint32_t n_sv_host = 0;
    // enqueue-ahead: a context that has just run an input of the same size sizes the later stages from that run's count of
    // anomalous reads (+12 %) and enqueues them before the pass-1 record is back, so the device does not idle at the
    // host's decision; finalize2_kernel neutralises them if the guess was too small and the host runs them again
    int speculate = 2;                // enqueue-ahead: 0 off (BDX_NO_SPECULATE=1), 1 sized from a prior on the read count only, 2 (default)
                                      // from the previous run of the same input where there is one, else from the prior
    int spec_test = 0;                // BDX_SPEC_TEST=1: guess half of the last count (forces the retry path)
    uint32_t last_na = 0;
    size_t last_n = 0;
    uint32_t na_alloc = 0;            // the count the later stages are sized and launched with
    bool region_of_fused = false;
    uint32_t join_table_clean = 0;    // slots of the direct join table already set to -1 (by K2), 0 = none
    float stage_ms[kNumStages] = {0};
    hipEvent_t ev[8] = {nullptr};

    // ---- streamed input (bdx_push / bdx_acquire_batch + bdx_submit_batch) ----
    hipStream_t copy_stream = nullptr;  // H2D copies of the batches; the classifier follows each batch on `stream` behind ev_copy
    hipEvent_t ev_copy = nullptr;
    bool copy_pending = false;          // copies enqueued since the last run: `stream` has to wait for ev_copy
    // pass 1 as the reads arrive: the tile tables are laid out for k1_cap_tiles tiles and tiles [0, k1_done) are classified
    bool k1_live = false;
    uint32_t k1_done = 0, k1_cap_tiles = 0;
    struct Stage {                      // one pinned staging buffer of the ring
        PinBuf buf;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    };
    Stage ring[4];
    int ring_next = 0, ring_cur = -1;
    // name keys that stay in the caller's pinned memory (bdx_push): one segment per batch; host == nullptr: that range of
    // keys was copied into the resident column
    struct KeySeg { uint64_t begin; const uint64_t* host; const uint16_t* host_qlen; const uint64_t* host_check; };
    std::vector<KeySeg> key_segs;
    DevBuf b_seg, b_done, b_lb;
};

namespace {

thread_local std::string* t_err_sink = nullptr;   // a sizing pass on a thread of its own: its messages do not go to the context's `err`
int fail(bdx_ctx* c, int code, const std::string& msg) {
    if (t_err_sink) *t_err_sink = msg;
    else if (c) c->err = msg;
    return code;
}
// what a sizing pass hands the stage functions instead of the context's run state
struct Sizing { uint32_t na; };
// entry points that launch the stages behind pass 1: not while their buffers are being sized on another thread
#define NOT_WHILE_SIZING(c)                                                                                                              \
    do {                                                                                                                                 \
        if ((c)->sizing.load(std::memory_order_acquire))                                                                                 \
            return fail(c, BDX_ESTATE, "the buffers of the later stages are being sized on another thread (bdx_bamdec_finish has not returned)"); \
    } while (0)
int hipfail(bdx_ctx* c, hipError_t e, const char* what) {
    return fail(c, BDX_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(ctx, expr)                                   \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return hipfail(ctx, _e, #expr); \
    } while (0)

size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// The stamp of the next look-back launch on this context.  A state array may see a given stamp only once (a word of an
// earlier launch with the same stamp would read as already published), so the stamp counts LAUNCHES, not runs -- the cut and
// the table stage can be launched again without a new pass 1 -- and when the 30-bit counter comes round the arrays start
// from zero again.
hipError_t next_lb_stamp(bdx_ctx* c, uint32_t* out) {
    if (((++c->lb_seq) & 0x3FFFFFFFu) == 0) {
        if (c->b_lb.p) { const hipError_t e = hipMemsetAsync(c->b_lb.p, 0, c->b_lb.bytes, c->stream); if (e != hipSuccess) return e; }
        if (c->b_ws6.p) { const hipError_t e = hipMemsetAsync(c->b_ws6.p, 0, c->b_ws6.bytes, c->stream); if (e != hipSuccess) return e; }
        ++c->lb_seq;
    }
    *out = c->lb_seq & 0x3FFFFFFFu;
    return hipSuccess;
}

int alloc_reads(bdx_ctx* c, size_t cap) {
    cap = round_up(std::max<size_t>(cap, 1), 1024);
    if (cap <= c->cap) return BDX_OK;
    if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->k1_live = false;  // the tile tables were laid out for the old capacity: bdx_run classifies from the first tile again
    struct Col { DevBuf* b; size_t esz; const void** slot; };
    Col cols[] = {{&c->b_tid, 4, (const void**)&c->d.tid},     {&c->b_pos, 4, (const void**)&c->d.pos},
                  {&c->b_mtid, 4, (const void**)&c->d.mtid},   {&c->b_mpos, 4, (const void**)&c->d.mpos},
                  {&c->b_isize, 4, (const void**)&c->d.isize}, {&c->b_flag, 2, (const void**)&c->d.flag},
                  {&c->b_qlen, 2, (const void**)&c->d.qlen},   {&c->b_mapq, 1, (const void**)&c->d.mapq},
                  {&c->b_lib, 1, (const void**)&c->d.lib},     {&c->b_bam, 1, (const void**)&c->d.bam},
                  {&c->b_key, 8, (const void**)&c->d.key},     {&c->b_check, 8, (const void**)&c->d.check}};
    for (Col& col : cols) {
        if (col.b == &c->b_check && !c->use_check) continue;
        DevBuf nb;
        HIPCHK(c, nb.ensure(cap * col.esz));
        if (c->n) HIPCHK(c, hipMemcpyAsync(nb.p, col.b->p, c->n * col.esz, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        col.b->release();
        *col.b = nb;
        *col.slot = nb.p;
    }
    c->cap = cap;
    return BDX_OK;
}

// columns of a staging buffer: every column starts 8-byte aligned (the stride is a multiple of 64 records)
void stage_view(const bdx_ctx::Stage& st, bdx_batch_buf* out) {
    char* p = (char*)st.buf.p;
    const size_t K = st.cap;
    out->name_key = (uint64_t*)p; p += K * 8;
    out->name_check = (uint64_t*)p; p += K * 8;
    out->tid = (int32_t*)p; p += K * 4;
    out->pos = (int32_t*)p; p += K * 4;
    out->mtid = (int32_t*)p; p += K * 4;
    out->mpos = (int32_t*)p; p += K * 4;
    out->isize = (int32_t*)p; p += K * 4;
    out->flag = (uint16_t*)p; p += K * 2;
    out->qlen = (uint16_t*)p; p += K * 2;
    out->mapq = (uint8_t*)p; p += K;
    out->lib = (uint8_t*)p; p += K;
    out->bam = (uint8_t*)p;
    out->capacity = K;
}

int pass1_prepare(bdx_ctx* c, uint32_t tiles_cap);
int presize_stages(bdx_ctx* c, uint32_t na);
int presize_stages_here(bdx_ctx* c, uint32_t na);
int pass1_classify(bdx_ctx* c, uint32_t upto, bool timed);

constexpr uint32_t kStreamTilesMin = 4096;  // classify behind a batch only once this many new tiles (1 M reads) are complete

// One batch of host records into the resident store: H2D copies on the copy stream, then -- for a store that is being
// filled from empty -- the classifier over the tiles the batch completed, on the compute stream behind the copies.
// lazy_keys: the batch is the caller's own memory and stays valid until bdx_run returns, so pinned name keys need not
// travel: K2 fetches the keys of the anomalous reads (about 1 %) straight from there.
int enqueue_batch(bdx_ctx* c, const bdx_batch& b, bool lazy_keys) {
    if (c->n + b.n > c->cap) {
        const int rc = alloc_reads(c, std::max(c->n + b.n, c->cap * 2));
        if (rc != BDX_OK) return rc;
    }
    const size_t o = c->n, n = b.n;
    hipStream_t s = c->copy_stream;
    if (o == 0) {  // a fresh store: pass 1 runs as the reads arrive
        c->key_segs.clear();
        const uint64_t tiles = (c->cap + kTile - 1) / kTile;
        if (tiles <= 0xFFFFFFFFull) {
            const int rc = pass1_prepare(c, (uint32_t)tiles);
            if (rc != BDX_OK) return rc;
            c->k1_live = true;
        }
    }
    // (one stream: splitting the columns over two copy streams measured 10.8 ms against 9.1 ms for 15 M records)
    hipStream_t s2 = s;
    HIPCHK(c, hipMemcpyAsync((void*)(c->d.tid + o), b.tid, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync((void*)(c->d.pos + o), b.pos, n * 4, hipMemcpyHostToDevice, s2));
    HIPCHK(c, hipMemcpyAsync((void*)(c->d.mtid + o), b.mtid, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync((void*)(c->d.mpos + o), b.mpos, n * 4, hipMemcpyHostToDevice, s2));
    HIPCHK(c, hipMemcpyAsync((void*)(c->d.isize + o), b.isize, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync((void*)(c->d.flag + o), b.flag, n * 2, hipMemcpyHostToDevice, s2));
    HIPCHK(c, hipMemcpyAsync((void*)(c->d.mapq + o), b.mapq, n, hipMemcpyHostToDevice, s2));
    // (with one library / one source file the kernels take every index as 0 and do not read the column)
    if (c->nlibs > 1) HIPCHK(c, hipMemcpyAsync((void*)(c->d.lib + o), b.lib, n, hipMemcpyHostToDevice, s2));
    if (c->nbams > 1) HIPCHK(c, hipMemcpyAsync((void*)(c->d.bam + o), b.bam, n, hipMemcpyHostToDevice, s2));
    // name keys and read lengths are only needed for the anomalous reads (about 1 %): pinned ones stay where they are
    const uint64_t* dev_view = nullptr;
    const uint16_t* dev_qlen = nullptr;
    const uint64_t* dev_check = nullptr;
    if (lazy_keys && c->key_segs.size() < 64) {
        auto device_view = [](const void* hp) -> void* {
            hipPointerAttribute_t attr{};
            void* dp = nullptr;
            if (hipPointerGetAttributes(&attr, hp) == hipSuccess && attr.type == hipMemoryTypeHost &&
                hipHostGetDevicePointer(&dp, (void*)hp, 0) == hipSuccess)
                return dp;
            (void)hipGetLastError();  // ordinary pageable memory: not an error
            return nullptr;
        };
        dev_view = (const uint64_t*)device_view(b.name_key);
        dev_qlen = dev_view ? (const uint16_t*)device_view(b.qlen) : nullptr;
        if (dev_qlen && c->use_check) dev_check = (const uint64_t*)device_view(b.name_check);
        if (!dev_qlen || (c->use_check && !dev_check)) dev_view = nullptr;
    }
    if (!dev_view) {
        HIPCHK(c, hipMemcpyAsync((void*)(c->d.key + o), b.name_key, n * 8, hipMemcpyHostToDevice, s2));
        HIPCHK(c, hipMemcpyAsync((void*)(c->d.qlen + o), b.qlen, n * 2, hipMemcpyHostToDevice, s));
        if (c->use_check) HIPCHK(c, hipMemcpyAsync((void*)(c->d.check + o), b.name_check, n * 8, hipMemcpyHostToDevice, s2));
    }
    if (c->key_segs.empty() || dev_view || c->key_segs.back().host)
        c->key_segs.push_back(bdx_ctx::KeySeg{(uint64_t)o, dev_view, dev_qlen, dev_check});
    HIPCHK(c, hipEventRecord(c->ev_copy, s));
    c->copy_pending = true;
    c->n += n;
    c->ran = false;
    if (c->k1_live) {
        const uint32_t full = (uint32_t)(c->n / kTile);
        if (full >= c->k1_done + kStreamTilesMin) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
            c->copy_pending = false;
            const int rc = pass1_classify(c, full, false);
            if (rc != BDX_OK) return rc;
        }
    }
    return BDX_OK;
}

}  // namespace

extern "C" {

void bdx_opts_default(bdx_opts* o) {  // common/Options.cpp:27-41
    if (!o) return;
    o->min_len = 7; o->cut_sd = 3; o->max_sd = 1000000000; o->min_map_qual = 35; o->min_read_pair = 2;
    o->seq_coverage_lim = 1000; o->buffer_size = 100; o->transchr_rearrange = 0; o->fisher = 0;
    o->illumina_long_insert = 0; o->cn_lib = 0; o->print_af = 0; o->score_threshold = 30; o->chr_restricted = 0;
}

const char* bdx_strerror(int code) {
    switch (code) {
        case BDX_OK: return "ok";
        case BDX_EINVAL: return "invalid argument";
        case BDX_ENOMEM: return "out of memory";
        case BDX_EHIP: return "HIP runtime error";
        case BDX_ESTATE: return "call out of order";
        case BDX_ELIMIT: return "limit exceeded";
        default: return "internal error";
    }
}
const char* bdx_last_error(const bdx_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }
int bdx_device(const bdx_ctx* ctx) { return ctx ? ctx->device : -1; }
void* bdx_stream(const bdx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int bdx_create(bdx_ctx** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids,
               int max_read_window_size0, int device) {
    if (!out || !opts || !libs || nlibs < 1 || nbams < 1) return BDX_EINVAL;
    if (nlibs > 255 || nbams > 254) return BDX_ELIMIT;
    const int nkeys = opts->cn_lib ? nlibs : nbams;
    if (nkeys > 60) return BDX_ELIMIT;
    for (int i = 0; i < nlibs; ++i)
        if (libs[i].bam_index < 0 || libs[i].bam_index >= nbams) return BDX_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BDX_EHIP;
    if (hipSetDevice(device) != hipSuccess) return BDX_EHIP;
    bdx_ctx* c = new (std::nothrow) bdx_ctx;
    if (!c) return BDX_ENOMEM;
    c->device = device;
    c->opts = *opts;
    c->libs.assign(libs, libs + nlibs);
    c->nlibs = nlibs; c->nbams = nbams; c->ntids = ntids; c->nkeys = nkeys; c->w0 = max_read_window_size0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return BDX_EHIP; }
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete c; return BDX_EHIP; }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return BDX_EHIP; }
    if (hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess) { delete c; return BDX_EHIP; }
    for (auto& st : c->ring)
        if (hipEventCreateWithFlags(&st.done, hipEventDisableTiming) != hipSuccess) { delete c; return BDX_EHIP; }
    if (hipEventCreateWithFlags(&c->ev_groups, hipEventDisableTiming) != hipSuccess) { delete c; return BDX_EHIP; }
    if (hipEventCreateWithFlags(&c->ev_regions, hipEventDisableTiming) != hipSuccess) { delete c; return BDX_EHIP; }
    std::vector<DevLib> dl(nlibs);
    for (int i = 0; i < nlibs; ++i) {
        dl[i].upper = libs[i].uppercutoff;
        dl[i].lower = libs[i].lowercutoff;
        dl[i].min_mapq = libs[i].min_mapping_quality < 0 ? opts->min_map_qual : libs[i].min_mapping_quality;
        dl[i].key = opts->cn_lib ? i : libs[i].bam_index;
    }
    std::vector<float> means(nlibs);
    for (int i = 0; i < nlibs; ++i) means[i] = libs[i].mean_insertsize;
    if (c->b_libs.ensure(nlibs * sizeof(DevLib)) != hipSuccess ||
        hipMemcpy(c->b_libs.p, dl.data(), nlibs * sizeof(DevLib), hipMemcpyHostToDevice) != hipSuccess ||
        c->b_lib_mean.ensure(nlibs * 4) != hipSuccess ||
        hipMemcpy(c->b_lib_mean.p, means.data(), nlibs * 4, hipMemcpyHostToDevice) != hipSuccess) {
        bdx_destroy(c);
        return BDX_EHIP;
    }
    *out = c;
    return BDX_OK;
}

void bdx_destroy(bdx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& st : c->ring) {
        st.buf.release();
        if (st.done) (void)hipEventDestroy(st.done);
    }
    c->b_seg.release();
    c->b_done.release();
    c->b_lb.release();
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    DevBuf* bufs[] = {&c->b_tid, &c->b_pos, &c->b_mtid, &c->b_mpos, &c->b_isize, &c->b_flag, &c->b_qlen, &c->b_mapq, &c->b_lib,
                      &c->b_bam, &c->b_key, &c->b_check, &c->b_c_check, &c->b_x_check, &c->b_groups, &c->b_libs, &c->b_cls, &c->b_stash, &c->b_chunk_tot, &c->b_tile_tot, &c->b_tile_pre, &c->b_tile_mono,
                      &c->b_blk_cnt, &c->b_cnt, &c->b_p1, &c->b_c_tid, &c->b_c_pos, &c->b_c_isize,
                      &c->b_c_meta, &c->b_c_key, &c->b_c_idx, &c->b_c_nn, &c->b_c_pk, &c->b_cand, &c->b_pre_q, &c->b_pre_rev,
                      &c->b_pre_nonctx, &c->b_c_first, &c->b_c_maxq, &c->b_c_rid, &c->b_region_of, &c->b_ws_u4, &c->b_ws_u32, &c->b_totals,
                      &c->b_counts, &c->b_bcnt, &c->b_boff, &c->b_bcur, &c->b_e_key, &c->b_e_idx, &c->b_partner, &c->b_t_key,
                      &c->b_t_idx, &c->b_x_key, &c->b_x_order, &c->b_x_region,
                      &c->b_x_meta, &c->b_x_isize, &c->b_x_n, &c->b_fold, &c->b_lib_mean, &c->b_pair_lo, &c->b_sv_src, &c->b_dlists, &c->b_ltail, &c->b_r_rec, &c->b_r_pk, &c->b_out_deg,
                      &c->b_parts, &c->b_kdens, &c->b_rs, &c->b_slot, &c->b_members, &c->b_own, &c->b_lib_stage,
                      &c->b_cn_stage, &c->b_t_lambda, &c->b_t_k, &c->b_ws6, &c->b_ins, &c->b_member_ids,
                      &c->b_sv_out, &c->b_lib_index_out, &c->b_lib_pairs_out, &c->b_cn_key_out, &c->b_cn_value_out, &c->b_ltail_out, &c->b_sv_key};
    for (DevBuf* b : bufs) b->release();
    PinBuf* pins[] = {&c->h_p1, &c->h_cnt, &c->h_counts, &c->h_regs, &c->h_pk, &c->h_groups, &c->h_terms, &c->h_flags, &c->h_hs_rec, &c->h_hs_aux, &c->h_hs_lists, &c->h_printed, &c->h_counts0, &c->h_counts2,
                      &c->h_sv_out, &c->h_lib_index, &c->h_lib_pairs, &c->h_cn_key, &c->h_cn_value, &c->h_ltail_dev};
    for (PinBuf* b : pins) b->release();
    if (c->walk_scratch) walk_scratch_free(c->walk_scratch);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_groups) (void)hipEventDestroy(c->ev_groups);
    if (c->ev_regions) (void)hipEventDestroy(c->ev_regions);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bdx_reserve(bdx_ctx* c, size_t n_reads) {
    if (!c) return BDX_EINVAL;
    if (c->adopted) return fail(c, BDX_ESTATE, "reads were adopted from the caller");
    NOT_WHILE_SIZING(c);
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = alloc_reads(c, n_reads);
    if (rc != BDX_OK) return rc;
    // The buffers of the stages behind pass 1 as well, for the prior a first run sizes them by (1/32 of the reads anomalous, see
    // bdx_run): ~60 device and pinned allocations, 3-4 ms at 15 M reads, most of it page pinning -- here they happen while the
    // caller still decodes or copies, not inside its first bdx_run.  Small inputs size theirs exactly, when they run.
    if (n_reads >= (1u << 20) && !c->ran) {
        const uint64_t prior = (uint64_t)n_reads / 32 + 4096;
        if (prior <= kMaxAnomalous) return presize_stages_here(c, (uint32_t)prior);
    }
    return BDX_OK;
}

int bdx_push(bdx_ctx* c, const bdx_batch* b) {
    if (!c || !b) return BDX_EINVAL;
    if (c->adopted) return fail(c, BDX_ESTATE, "reads were adopted from the caller");
    if (b->n == 0) return BDX_OK;
    if (!b->tid || !b->pos || !b->mtid || !b->mpos || !b->isize || !b->flag || !b->qlen || !b->mapq || !b->lib || !b->bam ||
        !b->name_key || (c->use_check && !b->name_check))
        return fail(c, BDX_EINVAL, "null array in batch");
    HIPCHK(c, hipSetDevice(c->device));
    return enqueue_batch(c, *b, true);
}

int bdx_acquire_batch(bdx_ctx* c, size_t capacity, bdx_batch_buf* out) {
    if (!c || !out || capacity == 0) return BDX_EINVAL;
    if (c->adopted) return fail(c, BDX_ESTATE, "reads were adopted from the caller");
    if (c->ring_cur >= 0) return fail(c, BDX_ESTATE, "the previous batch was not submitted");
    HIPCHK(c, hipSetDevice(c->device));
    bdx_ctx::Stage& st = c->ring[c->ring_next];
    if (st.busy) {  // all staging buffers are in flight: wait for the oldest copy
        HIPCHK(c, hipEventSynchronize(st.done));
        st.busy = false;
    }
    st.cap = round_up(capacity, 64);
    HIPCHK(c, st.buf.ensure(st.cap * 43 + 64));
    stage_view(st, out);
    c->ring_cur = c->ring_next;
    c->ring_next = (c->ring_next + 1) % 4;
    return BDX_OK;
}

int bdx_submit_batch(bdx_ctx* c, size_t n) {
    if (!c) return BDX_EINVAL;
    if (c->ring_cur < 0) return fail(c, BDX_ESTATE, "no batch was acquired");
    bdx_ctx::Stage& st = c->ring[c->ring_cur];
    c->ring_cur = -1;
    if (n == 0) return BDX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    bdx_batch_buf v{};
    stage_view(st, &v);  // the layout bdx_acquire_batch handed out
    if (n > v.capacity) return fail(c, BDX_EINVAL, "more records than the acquired batch holds");
    bdx_batch b{v.tid, v.pos, v.mtid, v.mpos, v.isize, v.flag, v.qlen, v.mapq, v.lib, v.bam, v.name_key, n, v.name_check};
    const int rc = enqueue_batch(c, b, false);  // (the buffer is recycled: its name keys travel with the other columns)
    if (rc != BDX_OK) return rc;
    HIPCHK(c, hipEventRecord(st.done, c->copy_stream));
    st.busy = true;
    return BDX_OK;
}

int bdx_reset_reads(bdx_ctx* c) {
    if (!c) return BDX_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->adopted) {
        c->d = ReadsSoA{};
        c->cap = 0;
        c->adopted = false;
    }
    c->n = 0;
    c->ran = false;
    c->k1_live = false;
    c->k1_done = 0;
    c->copy_pending = false;
    c->key_segs.clear();
    c->ring_cur = -1;
    for (auto& st : c->ring) st.busy = false;
    return BDX_OK;
}

int bdx_set_device_reads(bdx_ctx* c, const bdx_batch* b) {
    if (!c || !b) return BDX_EINVAL;
    if (c->n && !c->adopted) return fail(c, BDX_ESTATE, "context already holds pushed reads");
    const void* ptrs[] = {b->tid, b->pos, b->mtid, b->mpos, b->isize, b->flag, b->qlen, b->mapq, b->lib, b->bam, b->name_key,
                          c->use_check ? (const void*)b->name_check : (const void*)b->name_key};
    for (const void* p : ptrs)
        if (b->n && (!p || ((uintptr_t)p & 15))) return fail(c, BDX_EINVAL, "device arrays must be non-null and 16-byte aligned");
    c->d.tid = b->tid; c->d.pos = b->pos; c->d.mtid = b->mtid; c->d.mpos = b->mpos; c->d.isize = b->isize;
    c->d.flag = b->flag; c->d.qlen = b->qlen; c->d.mapq = b->mapq; c->d.lib = b->lib; c->d.bam = b->bam; c->d.key = b->name_key;
    c->d.check = c->use_check ? b->name_check : nullptr;
    c->n = b->n;
    c->cap = b->n;
    c->adopted = true;
    c->ran = false;
    c->k1_live = false;
    c->key_segs.clear();
    return BDX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Stages.  bdx_run chains them on one context; the bdx_stage_* entry points expose the same stages so that
// several contexts (one per chromosome, spread over GPUs) can exchange the few global quantities between them.
// ---------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// Spin on a word of pinned host memory that a kernel sets once its results are written there.  Much shorter than the
// wake-up of a blocking stream / event wait; falls back to the caller's blocking wait if the word does not show up.
bool wait_flag(const bdx_ctx* c, int idx, uint32_t value) {
    if (!c->poll || !c->h_flags.p) return false;
    volatile uint32_t* f = (volatile uint32_t*)c->h_flags.p + idx;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        if (*f == value) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
        __builtin_ia32_pause();
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return false;
    }
}

// behind a poll that timed out and a stream that has drained: was the word written at all?  (A kernel that was never launched -- a
// launch that failed, a stage that returned early -- leaves zeros where its results should be; they must not pass for an empty input.)
bool flag_arrived(const bdx_ctx* c, int idx) {
    return !c->poll || !c->h_flags.p || ((volatile uint32_t*)c->h_flags.p)[idx] == c->seq;
}

// Tell the host that everything enqueued so far has completed: a stream write-value into a polled pinned word, or (polling
// off / the stream operation unavailable) an event.  The word must be written AFTER a kernel boundary behind the kernels
// whose results it announces (their stores to host memory come from several compute dies; only the end of the kernel
// orders them).  On this runtime the write-value command is itself a one-thread kernel (__amd_rocclr_streamOpsWrite in
// the kernel trace, ~4 us on the stream), so where another kernel follows anyway its first thread sets the word instead
// (K6Arrays::flag_regions / flag_groups); the command remains for the end of the run.
int signal_ready(bdx_ctx* c, int idx, hipEvent_t ev) {
    if (c->poll) {
        if (hipStreamWriteValue32(c->stream, c->h_flags.as<uint32_t>() + idx, c->seq, 0) == hipSuccess) return BDX_OK;
        (void)hipGetLastError();
        c->poll = false;  // not supported here: blocking waits from now on
    }
    if (ev) {
        hipError_t e = hipEventRecord(ev, c->stream);
        if (e != hipSuccess) return hipfail(c, e, "hipEventRecord");
    }
    return BDX_OK;
}

float ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<float, std::milli>(b - a).count();
}

// K1 + finalize: class bytes, per-tile tables, *local* pass-1 counters
int do_pass1(bdx_ctx* c, uint32_t na_cap = 0, bool wait = true, bool defer_second = false);
int wait_pass1(bdx_ctx* c);

// Start of a pass 1: per-tile tables laid out for tiles_cap tiles, counters and tables at their start values.
int pass1_prepare(bdx_ctx* c, uint32_t tiles_cap) {
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int nlibs = c->nlibs, nbams = c->nbams, nkeys = c->nkeys;
    const int ncols = 2 + nkeys, ncnt = nlibs * kNumFlags + nlibs + nbams;
    const uint32_t tstride = (uint32_t)round_up(std::max<uint32_t>(tiles_cap, 16), 16);
    c->tstride = tstride;
    c->k1_cap_tiles = tiles_cap;
    c->k1_done = 0;
    HIPCHK(c, c->b_cls.ensure(std::max<size_t>((size_t)tiles_cap * kTile, 16)));
    // K1's ready-made records for K2: 2 B per read of address space, written (and later read) only where reads are anomalous
    c->use_stash = !c->dbg_no_stash && nkeys <= kStashKeys;   // (no_stash: K2 gathers everything from the columns)
    if (c->use_stash) HIPCHK(c, c->b_stash.ensure(std::max<size_t>((size_t)tiles_cap * kStashCap * sizeof(StashRec), 64)));
    HIPCHK(c, c->b_tile_tot.ensure((size_t)ncols * tstride * 4));
    HIPCHK(c, c->b_tile_pre.ensure((size_t)ncols * tstride * 4));
    HIPCHK(c, c->b_tile_mono.ensure((size_t)nbams * tstride * sizeof(MonoRec)));
    HIPCHK(c, c->b_blk_cnt.ensure((size_t)kCntCopies * ncnt * 4));
    HIPCHK(c, c->b_cnt.ensure((size_t)ncnt * 4));
    HIPCHK(c, c->b_p1.ensure(sizeof(Pass1)));
    HIPCHK(c, c->h_p1.ensure(sizeof(Pass1)));
    HIPCHK(c, c->h_cnt.ensure((size_t)ncnt * 4));
    HIPCHK(c, c->h_counts.ensure(sizeof(StageCounts)));
    HIPCHK(c, c->b_counts.ensure(sizeof(StageCounts)));
    const size_t w_tot = (size_t)ncols * tstride, w_mono = (size_t)nbams * tstride * sizeof(MonoRec) / 4;
    if (w_tot > 0xFFFFFFFFull || w_mono > 0xFFFFFFFFull) {  // beyond the init kernel's 32-bit word counts
        HIPCHK(c, hipMemsetAsync(c->b_tile_tot.p, 0, w_tot * 4, s));
        HIPCHK(c, hipMemsetAsync(c->b_tile_mono.p, 0xFF, w_mono * 4, s));
    }
    InitList il{};
    int k = 0;
    auto add = [&](void* p, size_t words, uint32_t value) { il.ptr[k] = (uint32_t*)p; il.words[k] = (uint32_t)words; il.value[k] = value; ++k; };
    if (w_tot <= 0xFFFFFFFFull && w_mono <= 0xFFFFFFFFull) { add(c->b_tile_tot.p, w_tot, 0u); add(c->b_tile_mono.p, w_mono, 0xFFFFFFFFu); }
    add(c->b_blk_cnt.p, (size_t)kCntCopies * ncnt, 0u);
    add(c->b_p1.p, sizeof(Pass1) / 4, 0u);
    add(c->b_counts.p, sizeof(StageCounts) / 4, 0u);
    HIPCHK(c, c->b_done.ensure(64));
    add(c->b_done.p, 1, 0u);
    il.n = k;
    launch_init(il, s);
    return BDX_OK;
}

// K1 over the tiles [k1_done, upto) of the c->n reads that are (or, behind ev_copy, will be) in HBM
int pass1_classify(bdx_ctx* c, uint32_t upto, bool timed) {
    hipStream_t s = c->stream;
    if (upto <= c->k1_done) return BDX_OK;
    K1Params k1{};
    k1.r = c->d; k1.n = c->n; k1.ntiles = upto; k1.tstride = c->tstride; k1.tile0 = c->k1_done;
    k1.nlibs = c->nlibs; k1.nbams = c->nbams; k1.nkeys = c->nkeys;
    k1.max_sd = c->opts.max_sd; k1.opt_t = c->opts.transchr_rearrange; k1.opt_l = c->opts.illumina_long_insert;
    k1.libs = c->b_libs.as<DevLib>(); k1.cls = c->b_cls.as<uint8_t>(); k1.tile_tot = c->b_tile_tot.as<uint32_t>();
    k1.tile_mono = c->b_tile_mono.as<MonoRec>();
    k1.blk_cnt = c->b_blk_cnt.as<uint32_t>();
    k1.stash = c->use_stash ? c->b_stash.as<StashRec>() : nullptr;
    const uint32_t span = upto - c->k1_done;
    const uint32_t grid_cap = c->dbg_k1_grid > 0 ? (uint32_t)c->dbg_k1_grid : (uint32_t)kK1MaxGrid;  // (tuning probe)
    const int grid1 = (int)std::min<uint32_t>((span + kWaves - 1) / kWaves, grid_cap);
    launch_k1(k1, grid1, k1_lds_bytes(c->nlibs, c->nbams, c->nkeys), s, timed ? c->ev[0] : nullptr, timed ? c->ev[1] : nullptr);
    c->k1_done = upto;
    return BDX_OK;
}

int do_pass1(bdx_ctx* c, uint32_t na_cap, bool wait, bool defer_second) {
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int nlibs = c->nlibs, nbams = c->nbams, nkeys = c->nkeys;
    const int ncols = 2 + nkeys, ncnt = nlibs * kNumFlags + nlibs + nbams;
    if (c->n >= ((size_t)1 << 32)) return fail(c, BDX_ELIMIT, "more than 2^32 - 1 reads in one context (read indices and the prefix counters are 32-bit)");
    const uint32_t ntiles = (uint32_t)((c->n + kTile - 1) / kTile);
    c->ntiles = ntiles;
    c->ran = false; c->stage = 0; c->replayed = false;
    c->na_alloc = 0;
    c->regions.clear(); c->r_pk.clear(); c->parts.clear();
    c->reg = nullptr; c->nreg = 0; c->rpk = nullptr;
    c->walk.clear();
    if (!c->walk_scratch) c->walk_scratch = walk_scratch_new();
    c->n_printed = 0;
    memset(&c->counts, 0, sizeof(c->counts));
    for (float& m : c->stage_ms) m = 0;

    if (c->copy_pending) {  // batches pushed since the last classifier launch
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_copy, 0));
        c->copy_pending = false;
    }
    // a store that was filled from empty has its first tiles classified already; anything else starts from tile 0
    if (!(c->k1_live && c->k1_cap_tiles >= ntiles)) {
        const int rc = pass1_prepare(c, ntiles);
        if (rc != BDX_OK) return rc;
    }
    c->k1_live = false;  // (consumed: a repeated run classifies everything again)
    const uint32_t tstride = c->tstride;
    const bool time_k1 = (c->stage_timing || c->seq % c->k1_event_period == 0) && c->k1_done == 0 && ntiles > 0;
    {
        const int rc = pass1_classify(c, ntiles, time_k1);
        if (rc != BDX_OK) return rc;
    }
    FinalizeParams fp{};
    fp.ntiles = ntiles; fp.tstride = tstride; fp.nblk = 0;
    fp.nfold = std::max<uint32_t>(1, std::min<uint32_t>(64, (ntiles + 1023) / 1024));
    HIPCHK(c, c->b_fold.ensure((size_t)nbams * fp.nfold * sizeof(MonoRec)));
    fp.fold_part = c->b_fold.as<MonoRec>();
    fp.nlibs = nlibs; fp.nbams = nbams; fp.nkeys = nkeys; fp.ncols = ncols; fp.ncnt = ncnt; fp.w0 = c->w0;
    fp.tile_tot = c->b_tile_tot.as<uint32_t>(); fp.tile_pre = c->b_tile_pre.as<uint32_t>(); fp.tile_mono = c->b_tile_mono.as<MonoRec>();
    {   // the tile-total columns are scanned in chunks of whole rounds of a workgroup (4096 super tiles), at most kMaxChunks of them
        const uint32_t nsuper = (ntiles + kK2TilesPerWave - 1) / kK2TilesPerWave;
        const uint32_t round = 4096;
        const uint32_t rounds = std::max<uint32_t>(1, (nsuper + round - 1) / round);
        // (tests: chunks of several rounds at small sizes)
        const uint32_t max_chunks = c->dbg_max_chunks > 0 ? (uint32_t)std::min(c->dbg_max_chunks, (int)kMaxChunks) : (uint32_t)kMaxChunks;
        fp.chunk_super = round * ((rounds + max_chunks - 1) / max_chunks);
        fp.nchunk = std::max<uint32_t>(1, (nsuper + fp.chunk_super - 1) / fp.chunk_super);
        HIPCHK(c, c->b_chunk_tot.ensure((size_t)ncols * kMaxChunks * 8));
        fp.chunk_tot = c->b_chunk_tot.as<uint32_t>();
        fp.chunk_base = fp.chunk_tot + (size_t)ncols * kMaxChunks;
    }
    fp.blk_cnt = c->b_blk_cnt.as<uint32_t>(); fp.cnt = c->b_cnt.as<uint32_t>(); fp.p1 = c->b_p1.as<Pass1>();
    HIPCHK(c, c->b_kdens.ensure(64 * 4));
    fp.libs = c->b_libs.as<DevLib>(); fp.cn_lib = c->opts.cn_lib; fp.key_density = c->b_kdens.as<float>();
    fp.cnt_host = c->h_cnt.as<uint32_t>(); fp.p1_host = c->h_p1.as<Pass1>();  // written by the kernel: no copy commands
    memset(c->h_p1.p, 0, sizeof(Pass1));
    HIPCHK(c, c->h_flags.ensure(64));
    ++c->seq;
    fp.flag_host = c->h_flags.as<uint32_t>(); fp.flag_value = c->seq;
    fp.na_cap = na_cap;
    // The second level as the job of finalize_kernel's last workgroup (BDX_FINALIZE2_FOLD=1) was measured and lost: the
    // device-scope fences it needs right behind K1's 15 MB of class bytes cost more (step 0.324 ms) than the launch (0.308 ms)
    const bool fold = c->dbg_finalize2_fold != 0;
    fp.done = fold ? c->b_done.as<uint32_t>() : nullptr;
    // enqueue-ahead: K2 follows without a host decision in between, so its launch takes the one-workgroup second level along
    c->fp_deferred = fp;
    c->finalize2_deferred = defer_second;
    launch_finalize(fp, s, !defer_second);
    c->k1_timed = time_k1;
    return wait ? wait_pass1(c) : BDX_OK;
}

// the pass-1 record and counters, written into pinned memory by finalize2_kernel
int wait_pass1(bdx_ctx* c) {
    const int ncnt = c->nlibs * kNumFlags + c->nlibs + c->nbams;
    if (!wait_flag(c, 0, c->seq)) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // (the word is set by the kernel that writes the record, whether or not the host polls for it: a stream that has drained without it
        // means that kernel was never launched -- an empty pass-1 record must not pass for a BAM without reads)
        if (c->h_flags.p && *((volatile uint32_t*)c->h_flags.p) != c->seq) return fail(c, BDX_EINTERNAL, "the pass-1 record did not arrive: its kernel was not launched");
    }
    c->p1 = *c->h_p1.as<Pass1>();
    c->cnt_local.assign(c->h_cnt.as<uint32_t>(), c->h_cnt.as<uint32_t>() + ncnt);
    if (c->k1_timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->k1_ms_last = ms;
    }
    c->stage_ms[0] = c->k1_ms_last;  // the latest measured launch
    if (c->p1.n_anom > kMaxAnomalous) return fail(c, BDX_ELIMIT, "more than 2^31 anomalous reads in one context");
    // capacity of the later stages: the count plus the headroom an enqueue-ahead run of the same input will ask for, so that
    // its buffers are these buffers (an enqueue-ahead run has set its guess already)
    if (!c->na_alloc && c->p1.n_anom) c->na_alloc = (uint32_t)std::min<uint64_t>((uint64_t)c->p1.n_anom + c->p1.n_anom / 8 + 1024, kMaxAnomalous);
    c->stage = 1;
    return BDX_OK;
}

// final window from the (global) counters: BreakDancerMax.cpp:109-116
int32_t window_from(const bdx_ctx* c, const uint32_t* cnt, uint32_t covered) {
    int W = c->w0;
    for (int i = 0; i < c->nlibs; ++i) {
        const int nd = (int)(cnt[i * kNumFlags + F_LARGE] + cnt[i * kNumFlags + F_SMALL]);
        const int tmp = nd > 0 ? (int)((float)covered / (float)nd) : 50;
        W = std::min(W, tmp);
    }
    return W;
}

// adopt the pass-1 statistics the rest of the path runs with (the context's own, or all-reduced ones)
int set_pass1(bdx_ctx* c, const uint32_t* cnt, uint32_t covered, int32_t window, bool upload) {
    const int nlibs = c->nlibs, nkeys = c->nkeys;
    const int ncnt = nlibs * kNumFlags + nlibs + c->nbams;
    c->cnt.assign(cnt, cnt + ncnt);
    c->g_covered = covered;
    c->g_window = window;
    // host-side scalars of main() (BreakDancerMax.cpp:88-107, BamSummary.cpp:140-149), float32 like the reference
    const uint32_t* lib_cnt = c->cnt.data() + nlibs * kNumFlags;
    const uint32_t* bam_cnt = lib_cnt + nlibs;
    c->seqcov.assign(nlibs, 0.f);
    c->lib_density.assign(nlibs, 0.f);
    c->key_density.assign(nkeys, 0.000001f);
    for (int i = 0; i < nlibs; ++i) {
        float covg = 0;
        if (lib_cnt[i] != 0 && covered != 0) covg = float(lib_cnt[i]) * c->libs[i].readlens / covered;
        c->seqcov[i] = covg;
        float dens = 0.000001f;
        if (c->opts.cn_lib) {
            if (lib_cnt[i] != 0) dens = float(lib_cnt[i]) / covered;
        } else {
            dens = float(bam_cnt[c->libs[i].bam_index]) / covered;
        }
        c->lib_density[i] = dens;
        c->key_density[c->opts.cn_lib ? i : c->libs[i].bam_index] = dens;
    }
    // the region cut reads the window from the device copy of Pass1
    struct { uint32_t covered; int32_t window; } hdr{covered, window};
    static_assert(offsetof(Pass1, covered_ref_len) == 0 && offsetof(Pass1, window) == 4, "Pass1 header layout");
    if (upload) {  // a single-context run adopts its own statistics: the device copy already holds them
        HIPCHK(c, hipMemcpyAsync(c->b_p1.p, &hdr, sizeof(hdr), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->stage = 2;
    return BDX_OK;
}

// K2: compact anomalous reads (prefix counters offset by the bases of earlier shards)
int do_compact(bdx_ctx* c, uint32_t nn_base, const uint32_t* pk_base, bool prepare_join, const Sizing* sz = nullptr) {
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int nkeys = c->nkeys;
    const uint32_t na = sz ? sz->na : c->na_alloc;
    if (c->stage_timing && !sz) HIPCHK(c, hipEventRecord(c->ev[2], s));
    Compact cp_sz{};
    K3Arrays k3_sz{};
    Compact& cp = sz ? cp_sz : c->cp;
    K3Arrays& k3 = sz ? k3_sz : c->k3;
    cp = Compact{};
    k3 = K3Arrays{};
    if (na) {
        if (na > kMaxAnomalous) return fail(c, BDX_ELIMIT, "more than 2^31 anomalous reads in one context");
        const size_t cap = na;
        HIPCHK(c, c->b_c_tid.ensure(cap * 4)); HIPCHK(c, c->b_c_pos.ensure(cap * 4)); HIPCHK(c, c->b_c_isize.ensure(cap * 4));
        HIPCHK(c, c->b_c_meta.ensure(cap * 4)); HIPCHK(c, c->b_c_key.ensure(cap * 8)); HIPCHK(c, c->b_c_nn.ensure(cap * 4));
        HIPCHK(c, c->b_c_idx.ensure(cap * 4));
        if (c->use_check) HIPCHK(c, c->b_c_check.ensure(cap * 8));
        HIPCHK(c, c->b_c_pk.ensure(cap * 4 * nkeys));
        cp.tid = c->b_c_tid.as<int32_t>(); cp.pos = c->b_c_pos.as<int32_t>(); cp.isize = c->b_c_isize.as<int32_t>();
        cp.meta = c->b_c_meta.as<uint32_t>(); cp.key = c->b_c_key.as<uint64_t>(); cp.nn = c->b_c_nn.as<uint32_t>();
        cp.idx = c->b_c_idx.as<uint32_t>();
        cp.check = c->use_check ? c->b_c_check.as<uint64_t>() : nullptr;
        cp.pk = c->b_c_pk.as<uint32_t>(); cp.cap = na;
        K2Params k2{};
        k2.r = c->d; k2.n = c->n; k2.ntiles = c->ntiles; k2.tstride = c->tstride; k2.nkeys = nkeys; k2.nlibs = c->nlibs; k2.libs = c->b_libs.as<DevLib>();
        k2.cls = c->b_cls.as<uint8_t>(); k2.tile_pre = c->b_tile_pre.as<uint32_t>(); k2.c = cp;
        k2.tile_tot = c->b_tile_tot.as<uint32_t>(); k2.stash = c->use_stash ? c->b_stash.as<StashRec>() : nullptr;
        k2.chunk_tot = c->fp_deferred.chunk_tot; k2.chunk_base = c->fp_deferred.chunk_base; k2.chunk_super = c->fp_deferred.chunk_super;
        k2.nn_base = nn_base;
        for (int k = 0; k < nkeys; ++k) k2.pk_base[k] = pk_base ? pk_base[k] : 0u;
        // scratch of the later stages cleared by this launch: the per-candidate max read length of K3, and (when this
        // context joins its own reads) the slot indices of K4's direct table
        HIPCHK(c, c->b_c_maxq.ensure(cap * 4));
        k2.fill_ptr[0] = c->b_c_maxq.as<uint32_t>(); k2.fill_words[0] = na; k2.fill_value[0] = 0u;
        if (!sz) c->join_table_clean = 0;
        if (prepare_join && (c->force_direct_join || (!c->bucketed_join && na <= kDirectJoinMax))) {
            const uint32_t slots = direct_join_slots(na);
            HIPCHK(c, c->b_t_key.ensure((size_t)slots * 8)); HIPCHK(c, c->b_partner.ensure((size_t)na * 4));
            k2.fill_ptr[1] = c->b_t_key.as<uint32_t>(); k2.fill_words[1] = 2 * slots; k2.fill_value[1] = 0xFFFFFFFFu;
            k2.fill_ptr[2] = c->b_partner.as<uint32_t>(); k2.fill_words[2] = na; k2.fill_value[2] = 0xFFFFFFFFu;
            HIPCHK(c, c->b_pair_lo.ensure((size_t)na * 4));
            k2.fill_ptr[3] = c->b_pair_lo.as<uint32_t>(); k2.fill_words[3] = na; k2.fill_value[3] = 0xFFFFFFFFu;
            if (!sz) c->join_table_clean = slots;
        }
        if (sz) return BDX_OK;
        {   // name keys the caller's pinned batches still hold (bdx_push): one segment per batch
            bool any_host = false;
            for (auto const& sg : c->key_segs) any_host |= sg.host != nullptr;
            if (any_host && !c->adopted) {
                const size_t ns = c->key_segs.size();
                std::vector<uint64_t> tab(4 * ns + 1);
                for (size_t i = 0; i < ns; ++i) {
                    const bdx_ctx::KeySeg& sg = c->key_segs[i];
                    tab[i] = sg.begin;
                    tab[ns + 1 + i] = (uint64_t)(uintptr_t)(sg.host ? sg.host - sg.begin : c->d.key);
                    tab[2 * ns + 1 + i] = (uint64_t)(uintptr_t)(sg.host ? sg.host_qlen - sg.begin : c->d.qlen);
                    tab[3 * ns + 1 + i] = (uint64_t)(uintptr_t)(sg.host ? sg.host_check - sg.begin : c->d.check);
                }
                tab[ns] = c->n;
                HIPCHK(c, c->b_seg.ensure(tab.size() * 8));
                HIPCHK(c, hipMemcpyAsync(c->b_seg.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, s));
                HIPCHK(c, hipStreamSynchronize(s));  // (tab is a local; this path is PCIe-bound anyway)
                k2.nseg = (int)ns;
                k2.seg_begin = c->b_seg.as<uint64_t>();
                k2.seg_ptr = (const uint64_t* const*)(c->b_seg.as<uint64_t>() + ns + 1);
                k2.seg_qlen = (const uint16_t* const*)(c->b_seg.as<uint64_t>() + 2 * ns + 1);
                k2.seg_check = (const uint64_t* const*)(c->b_seg.as<uint64_t>() + 3 * ns + 1);
            }
        }
        launch_k2(k2, k2_lds_bytes(nkeys), s, c->finalize2_deferred ? &c->fp_deferred : nullptr);
        c->finalize2_deferred = false;
    }
    if (sz) return BDX_OK;
    if (c->finalize2_deferred) {  // (no K2 launch to ride on)
        launch_finalize2_only(c->fp_deferred, s);
        c->finalize2_deferred = false;
    }
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[3], s));
    c->nn_base = nn_base;
    return BDX_OK;
}

// K3: cut regions.  In a whole-genome run the last candidate of a chromosome is closed by the first anomalous read
// of the next chromosome, which still counts for its nucleotide sum / max read length / normal-pair count
// (BreakDancer.cpp:202-231): has_next / next_qlen / next_nn carry that read across contexts.
// a result whose region table is read where the device left it (c->reg == h_regs.p: bdx_run on a large table, sharded runs) gets its own copy:
// called before the pinned buffers may be reallocated under it (a sizing pass for a larger input) or handed back (bdx_trim_results)
void own_borrowed_regions(bdx_ctx* c) {
    if (!c->reg || (const void*)c->reg != c->h_regs.p) return;
    const HostRegion* r = c->reg;
    const uint32_t* pk = c->rpk;
    const size_t n = c->nreg;
    std::vector<HostRegion> keep(r, r + n);
    std::vector<uint32_t> keep_pk;
    if (pk) keep_pk.assign(pk, pk + n * 2 * (size_t)c->nkeys);
    c->regions.swap(keep); c->r_pk.swap(keep_pk);
    c->reg = c->regions.data(); c->rpk = c->r_pk.data();
}

int do_cut(bdx_ctx* c, int has_next, int32_t next_qlen, uint32_t next_nn, bool for_k6, bool keep_dev = false, const Sizing* sz = nullptr) {
    HIPCHK(c, hipSetDevice(c->device));
    if (sz) own_borrowed_regions(c);   // (a sizing pass may grow the pinned table below.  bdx_reserve sizes nothing once the context has run, and the decoder's
                                       // pass follows a reset -- no live result should be here; if one is, it keeps a copy)
    hipStream_t s = c->stream;
    const int nkeys = c->nkeys;
    const uint32_t na = sz ? sz->na : c->na_alloc;
    const uint32_t nn_base = c->nn_base;
    Compact& cp = c->cp;
    K3Arrays k3_sz{};
    K3Arrays& k3 = sz ? k3_sz : c->k3;
    if (na) {
        const size_t cap = na;
        DevBuf* u32bufs[] = {&c->b_cand, &c->b_pre_q, &c->b_pre_rev, &c->b_pre_nonctx, &c->b_c_first, &c->b_c_maxq, &c->b_c_rid,
                             &c->b_region_of};
        for (DevBuf* b : u32bufs) HIPCHK(c, b->ensure(cap * 4));
        // the region table and (below) the group list are written by the kernels straight into pinned host memory:
        // they are write-once, read-never on the device, so the PCIe writes overlap the kernels and no D2H copy is needed
        const bool hbm_only = keep_dev && !for_k6;   // sharded runs: a chromosome's table is sent on from HBM, nobody reads it on this host
        if (!hbm_only) {
            HIPCHK(c, c->h_regs.ensure(cap * sizeof(RegionRec)));
            HIPCHK(c, c->h_pk.ensure(cap * 2 * nkeys * 4));
        }
        const size_t nblk = scan_grid(na, 1) + 1;  // (sized for one element per thread, the finest split the scans use)
        HIPCHK(c, c->b_ws_u4.ensure(nblk * sizeof(U4)));
        HIPCHK(c, c->b_ws_u32.ensure(nblk * 4));
        HIPCHK(c, c->b_totals.ensure(64));
        k3.cap = na;
        k3.cand = c->b_cand.as<int32_t>(); k3.pre_q = c->b_pre_q.as<uint32_t>(); k3.pre_rev = c->b_pre_rev.as<uint32_t>();
        k3.pre_nonctx = c->b_pre_nonctx.as<uint32_t>(); k3.c_first = c->b_c_first.as<uint32_t>();
        k3.c_maxq = c->b_c_maxq.as<int32_t>(); k3.c_rid = c->b_c_rid.as<int32_t>(); k3.region_of = c->b_region_of.as<int32_t>();
        k3.r_rec = c->h_regs.as<RegionRec>(); k3.r_pk = c->h_pk.as<uint32_t>();
        if (for_k6) {  // the device-side SV assembly reads the region table back: keep a copy in HBM
            HIPCHK(c, c->b_r_rec.ensure(cap * sizeof(RegionRec)));
            HIPCHK(c, c->b_r_pk.ensure(cap * 2 * nkeys * 4));
            HIPCHK(c, c->b_out_deg.ensure(cap * 6 * 4));
            k3.r_rec_dev = c->b_r_rec.as<RegionRec>(); k3.r_pk_dev = c->b_r_pk.as<uint32_t>(); k3.out_deg = c->b_out_deg.as<uint32_t>();
            // with the direct join right behind K3, that kernel forwards the table to the host
            const bool no_forward = c->dbg_no_forward != 0;
            k3.host_copy_later = (!no_forward && !c->bucketed_join && na <= kDirectJoinMax) ? 1 : 0;
        }
        const bool three_launch = c->dbg_scan3 != 0;  // (A/B: the block-sums / rescan pair of launches)
        if (!three_launch) {
            const size_t words = 5 * nblk;
            if (c->b_lb.bytes < words * 8) {  // the look-back words must start out zero; afterwards every run brings its own stamp
                HIPCHK(c, c->b_lb.ensure(words * 8));
                HIPCHK(c, hipMemsetAsync(c->b_lb.p, 0, c->b_lb.bytes, s));
            }
            k3.lb_state = c->b_lb.as<unsigned long long>();
            if (!sz) HIPCHK(c, next_lb_stamp(c, &k3.lb_stamp));
        }
        k3.ws_u4 = c->b_ws_u4.as<U4>(); k3.head_total = (U4*)c->b_totals.p; k3.ws_u32 = c->b_ws_u32.as<uint32_t>();
        k3.acc_total = (uint32_t*)((char*)c->b_totals.p + 32); k3.counts = c->b_counts.as<StageCounts>();
        if (for_k6) {
            HIPCHK(c, c->h_counts0.ensure(sizeof(StageCounts)));
            if (!sz) memset(c->h_counts0.p, 0, sizeof(StageCounts));
            k3.counts_host = c->h_counts0.as<StageCounts>();
        }
        if (hbm_only) {  // (no pinned mirror: pinning 24 chromosomes' tables cost a sharded run tens of milliseconds)
            HIPCHK(c, c->b_r_rec.ensure(cap * sizeof(RegionRec)));
            HIPCHK(c, c->b_r_pk.ensure(cap * 2 * nkeys * 4));
            k3.r_rec = c->b_r_rec.as<RegionRec>(); k3.r_pk = c->b_r_pk.as<uint32_t>();
            k3.r_rec_dev = nullptr; k3.r_pk_dev = nullptr;
            k3.host_copy_later = 0;
        }
        if (sz) return BDX_OK;
        K3Tail tail{has_next, next_qlen, next_nn, c->k3_tid_tail};
        // single-context runs that take the direct join let that kernel do k3_region_of_kernel's work
        c->region_of_fused = for_k6 && !c->bucketed_join && na <= kDirectJoinMax;
        launch_k3(k3, cp, c->b_p1.as<Pass1>(), na, c->opts.min_len, c->opts.seq_coverage_lim, nkeys, nn_base, tail, !c->region_of_fused, s);
        if (for_k6 && !k3.host_copy_later) {  // the region table is in pinned memory
            const int rc = signal_ready(c, 3, c->ev_regions);
            if (rc != BDX_OK) return rc;
        }
    }
    if (sz) return BDX_OK;
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[4], s));
    c->stage = 3;
    return BDX_OK;
}

// K4 on the context's own reads (single-context run)
int do_join_local(bdx_ctx* c, uint32_t n, const Entries& en, const uint32_t* n_ptr, bool join_only, const Sizing* sz = nullptr) {
    hipStream_t s = c->stream;
    K4Arrays k4_sz{};
    K4Arrays& k4 = sz ? k4_sz : c->k4;
    k4 = K4Arrays{};
    if (!n) return BDX_OK;
    k4.g_cap = n / 2 + 1;
    if (c->groups_in_hbm) {   // sharded runs: the groups are packaged for rank 0 from HBM
        HIPCHK(c, c->b_groups.ensure((size_t)k4.g_cap * sizeof(GroupRec)));
        k4.g_rec = c->b_groups.as<GroupRec>();
    } else {
        HIPCHK(c, c->h_groups.ensure((size_t)k4.g_cap * sizeof(GroupRec)));
        k4.g_rec = c->h_groups.as<GroupRec>();
    }
    // (foreign entries of a sharded run have no partner[] / pair_lo[] entry: sized for the context's own reads, as K2 cleared them)
    const size_t n_own = en.n_local ? std::min<size_t>(n, std::max<uint32_t>(c->na_alloc, 1u)) : n;
    HIPCHK(c, c->b_partner.ensure(n_own * 4));
    k4.partner = c->b_partner.as<int32_t>();
    if (sz) return BDX_OK;  // (the direct table is sized by do_compact; the bucketed join sizes its own when it runs)
    if (c->force_direct_join || (!c->bucketed_join && n <= kDirectJoinMax)) {
        uint32_t slots = direct_join_slots(n);
        // (the foreign entries of a sharded run come on top of the reads K2 sized the table for: it still has room at half its load)
        if (c->join_table_clean && (uint64_t)n * 2 <= c->join_table_clean) slots = c->join_table_clean;
        const bool want_lo = en.c_rid || en.want_pair_lo;
        if (c->join_table_clean != slots) {
            HIPCHK(c, c->b_t_key.ensure((size_t)slots * 8));
            HIPCHK(c, hipMemsetAsync(c->b_t_key.p, 0xFF, (size_t)slots * 8, s));
            HIPCHK(c, hipMemsetAsync(c->b_partner.p, 0xFF, n_own * 4, s));
            if (en.want_pair_lo) {
                HIPCHK(c, c->b_pair_lo.ensure(n_own * 4));
                HIPCHK(c, hipMemsetAsync(c->b_pair_lo.p, 0xFF, n_own * 4, s));
            }
        }
        k4.pair_lo = ((c->join_table_clean == slots && want_lo) || en.want_pair_lo) ? c->b_pair_lo.as<int32_t>() : nullptr;  // preset to -1 by K2
        c->join_table_clean = 0;
        k4.direct = 1; k4.t_mask = slots - 1;
        k4.t_key = c->b_t_key.as<uint64_t>(); k4.t_idx = c->b_t_idx.as<int32_t>();
        if (join_only) launch_k4_join_only(k4, en, n_ptr, n, c->b_counts.as<StageCounts>(), s);
        else launch_k4(k4, en, n_ptr, n, c->b_counts.as<StageCounts>(), s);
        return BDX_OK;
    }
    uint32_t nb = 1, lg = 0;
    while (nb < (uint32_t)kMaxBuckets && (size_t)nb * 384 < n) { nb <<= 1; ++lg; }  // ~256-512 entries per bucket: >= 1 workgroup per CU early
    k4.nbuckets = nb; k4.log2b = lg;
    if ((size_t)nb * 4 > c->b_bcnt.bytes) {  // (re)allocated: the partition histogram has to start from zero once
        HIPCHK(c, c->b_bcnt.ensure((size_t)kMaxBuckets * 4));
        HIPCHK(c, hipMemsetAsync(c->b_bcnt.p, 0, c->b_bcnt.bytes, s));
    }
    HIPCHK(c, c->b_boff.ensure((nb + 1) * 4)); HIPCHK(c, c->b_bcur.ensure(nb * 4));
    HIPCHK(c, c->b_e_key.ensure((size_t)n * 8)); HIPCHK(c, c->b_e_idx.ensure((size_t)n * 4));
    HIPCHK(c, c->b_t_key.ensure((size_t)n * 16)); HIPCHK(c, c->b_t_idx.ensure((size_t)n * 8));
    k4.bcnt = c->b_bcnt.as<uint32_t>(); k4.boff = c->b_boff.as<uint32_t>(); k4.bcur = c->b_bcur.as<uint32_t>();
    k4.e_key = c->b_e_key.as<uint64_t>(); k4.e_idx = c->b_e_idx.as<uint32_t>();
    k4.t_key = c->b_t_key.as<uint64_t>(); k4.t_idx = c->b_t_idx.as<int32_t>();
    if (join_only) launch_k4_join_only(k4, en, n_ptr, n, c->b_counts.as<StageCounts>(), s);
    else launch_k4(k4, en, n_ptr, n, c->b_counts.as<StageCounts>(), s);
    return BDX_OK;
}

// counts + region table (+ groups) to the host
int readback(bdx_ctx* c, bool with_groups) {
    hipStream_t s = c->stream;
    const uint32_t na = c->p1.n_anom;
    if (!na) return BDX_OK;
    HIPCHK(c, hipMemcpyAsync(c->h_counts.p, c->b_counts.p, sizeof(StageCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    c->counts = *c->h_counts.as<StageCounts>();
    if (c->counts.overflow) return fail(c, BDX_EINTERNAL, "group list overflow");
    (void)with_groups;  // regions / prefix samples / groups already sit in pinned host memory (see do_cut / do_join_local)
    return BDX_OK;
}

// region table for the host side.  borrow: rr / pk stay valid (the context's pinned buffers) and no id shift is
// needed, so the walk and the getters read them in place.
constexpr uint32_t kBorrowRegionsMin = 32768;   // regions from which on the host's share of the walk may read the table in pinned memory (bdx_run)
void decode_regions(bdx_ctx* c, const RegionRec* rr, const uint32_t* pk, uint32_t nr, uint32_t ph, bool borrow) {
    static_assert(sizeof(HostRegion) == sizeof(RegionRec) && offsetof(HostRegion, first) == offsetof(RegionRec, first), "region layout");
    const int nkeys = c->nkeys;
    if (borrow && !ph) {
        c->reg = (const HostRegion*)rr; c->nreg = nr; c->rpk = pk;
        return;
    }
    c->regions.resize(nr + ph);
    if (ph) c->regions[0] = HostRegion{-1, -1, -1, 0, 0, 0, 0, 0, 0};
    if (nr) memcpy(c->regions.data() + ph, rr, (size_t)nr * sizeof(RegionRec));
    c->r_pk.assign((size_t)ph * 2 * nkeys, 0u);
    c->r_pk.insert(c->r_pk.end(), pk, pk + (size_t)nr * 2 * nkeys);
    c->reg = c->regions.data(); c->nreg = c->regions.size(); c->rpk = c->r_pk.data();
}

void decode_groups(bdx_ctx* c, const GroupRec* gr, uint32_t ng, uint32_t ph) {
    c->parts.resize(ng);
    for (uint32_t i = 0; i < ng; ++i) {
        const uint64_t k = gr[i].key;
        c->parts[i] = GroupPart{(uint32_t)(k >> 38) + ph, (uint32_t)((k >> 12) & ((1u << 26) - 1)) + ph, (uint8_t)(k & 15),
                                (uint8_t)((k >> 4) & 255), gr[i].pairs, gr[i].sum_isize};
    }
}

// K6 on the context's own regions (single-context runs): pair groups per region, SV assembly of the components that need
// no traversal, everything else listed for the host walk; then the dense results and K5 for the device-assembled SVs.
// part: 0 the whole first half; 1 up to and including k6_pairs_kernel, 2 the rest; 3 the deferred device walk; 4 the arrays and no launch
int do_k6(bdx_ctx* c, bool force_host, int part = 0, const Sizing* sz = nullptr) {
    hipStream_t s = c->stream;
    K6Arrays a_sz{};
    K6Arrays& a = sz ? a_sz : c->k6;
    if (part == 2) {
        if (!a.cap) return BDX_OK;
        launch_k6_components(a, a.cap, s);
        if (!c->poll) {
            const int rc = signal_ready(c, 1, c->ev_groups);
            if (rc != BDX_OK) return rc;
        }
        if (!c->defer_walk) launch_k6_walk(a, a.cap, s);
        return BDX_OK;
    }
    if (part == 3) {   // (the deferred device walk)
        if (a.cap) launch_k6_walk(a, a.cap, s);
        return BDX_OK;
    }
    const uint32_t na = std::max(sz ? sz->na : c->na_alloc, c->k6_cap);   // (sharded runs: K6's arrays are indexed by genome-wide region id)
    const int nkeys = c->nkeys, nlibs = c->nlibs;
    a = K6Arrays{};
    if (!na) return BDX_OK;
    const size_t cap = na;
    HIPCHK(c, c->b_out_deg.ensure(cap * 6 * 4));
    HIPCHK(c, c->b_parts.ensure(cap * sizeof(PartRec)));
    HIPCHK(c, c->b_rs.ensure(cap * sizeof(RegSum)));
    HIPCHK(c, c->b_members.ensure(cap * kK6MaxMembers * sizeof(MemberInfo)));
    HIPCHK(c, c->b_own.ensure(cap * 7 * 4 + 64 * 4 * 4));
    // Components of 5..64 regions cost one more launch (k6_walk_big_kernel) and a member table of 256 B per label.  Few of
    // them are walked by the host behind the device's own walk for free; many (dense data) make the host walk the longest
    // stage.  Without a previous run to go by, the number of anomalous reads decides.
    const int big_walk = c->big_walk_mode >= 0 ? c->big_walk_mode : (c->last_big_groups >= 0 ? c->last_big_groups > 2000 : na > 500000u);
    if (big_walk) HIPCHK(c, c->b_member_ids.ensure(cap * kK6BigMembers * 4));
    a.sv_cap = na / 2 + 1; a.term_cap = na / 2 + 1; a.cn_cap = (na / 2 + 1) * (uint32_t)nkeys;
    a.lib_stride = (uint32_t)std::min(nlibs, kK6LibStride);
    HIPCHK(c, c->b_slot.ensure(cap * sizeof(SvOut)));
    HIPCHK(c, c->b_lib_stage.ensure(cap * a.lib_stride * sizeof(LibStage)));
    HIPCHK(c, c->b_cn_stage.ensure(cap * (size_t)nkeys * sizeof(CnStage) + 16));
    HIPCHK(c, c->b_t_lambda.ensure((size_t)a.term_cap * 8)); HIPCHK(c, c->b_t_k.ensure((size_t)a.term_cap * 4));
    const size_t nblk = scan_grid(na, 1) + 1;
    if (c->table_in_hbm) {   // (rank 0 of a sharded run merges the ranks' tables: this one goes there from HBM)
        HIPCHK(c, c->b_sv_out.ensure((size_t)a.sv_cap * sizeof(SvOut))); HIPCHK(c, c->b_sv_key.ensure((size_t)a.sv_cap * 8));
        HIPCHK(c, c->b_lib_index_out.ensure((size_t)a.term_cap * 4)); HIPCHK(c, c->b_lib_pairs_out.ensure((size_t)a.term_cap * 4));
        HIPCHK(c, c->b_cn_key_out.ensure((size_t)a.cn_cap * 4 + 16)); HIPCHK(c, c->b_cn_value_out.ensure((size_t)a.cn_cap * 4 + 16));
        HIPCHK(c, c->b_ltail_out.ensure((size_t)a.term_cap * 8));
    } else {
        HIPCHK(c, c->h_sv_out.ensure((size_t)a.sv_cap * sizeof(SvOut)));
        HIPCHK(c, c->h_lib_index.ensure((size_t)a.term_cap * 4)); HIPCHK(c, c->h_lib_pairs.ensure((size_t)a.term_cap * 4));
        HIPCHK(c, c->h_cn_key.ensure((size_t)a.cn_cap * 4 + 16)); HIPCHK(c, c->h_cn_value.ensure((size_t)a.cn_cap * 4 + 16));
        HIPCHK(c, c->h_ltail_dev.ensure((size_t)a.term_cap * 8));
    }
    HIPCHK(c, c->h_counts2.ensure(sizeof(StageCounts)));
    HIPCHK(c, c->b_sv_src.ensure((size_t)a.sv_cap * 16)); HIPCHK(c, c->b_ltail.ensure((size_t)a.term_cap * 8));
    HIPCHK(c, c->b_dlists.ensure((size_t)a.term_cap * 4 + (size_t)a.cn_cap * 8 + 64));
    {   // the inserted list (k6_insert_kernel): device order keys padded to a power of two for the sort
        size_t p2 = 1;
        while (p2 < a.sv_cap) p2 <<= 1;
        const size_t svc = a.sv_cap;
        const size_t nsort = std::min<size_t>(svc, kK6RankSortMax);   // (k6_ranksort_kernel's output: lists of that many entries at most)
        HIPCHK(c, c->b_ins.ensure(p2 * 12 + svc * 8 + (svc + 1) * 16 + nsort * 12 + nsort * 4 * kK6RankSlices + 64));
        a.old_key = c->b_ins.as<uint64_t>(); a.hs_key_dev = a.old_key + p2;
        a.old_slot = (uint32_t*)(a.hs_key_dev + svc); a.ins_T = a.old_slot + p2; a.ins_src = a.ins_T + svc + 1;
        a.ins_pre_l = a.ins_src + svc + 1; a.ins_pre_c = a.ins_pre_l + svc + 1;
        a.sorted_key = (uint64_t*)(((uintptr_t)(a.ins_pre_c + svc + 1) + 7) & ~(uintptr_t)7); a.sorted_slot = (uint32_t*)(a.sorted_key + nsort);
        // (k6_ranksort_kernel is launched, and its ranks are read, where the list of candidates placed by key CAN outgrow what k6_insert_kernel's
        // one workgroup sorts in LDS (2,048 entries) -- a -t run's candidates are all of that kind: 5-10 k of them at a genome share took that
        // workgroup's bitonic sort in HBM half a millisecond.  Smaller inputs do without the launch)
        a.rank_part = a.sv_cap > 2048u ? a.sorted_slot + nsort : nullptr;
    }
    a.sv_begin = c->b_sv_src.as<uint2>(); a.sv_src = (uint32_t*)(a.sv_begin + a.sv_cap); a.ltail = c->b_ltail.as<double>();
    a.d_lib_index = c->b_dlists.as<int32_t>(); a.d_cn_key = a.d_lib_index + a.term_cap; a.d_cn_value = (float*)(a.d_cn_key + a.cn_cap);
    a.cap = na;
    a.r_rec = c->k6_r_rec ? c->k6_r_rec : c->b_r_rec.as<RegionRec>(); a.r_pk = c->k6_r_pk ? c->k6_r_pk : c->b_r_pk.as<uint32_t>();
    a.taint = c->k6_taint;
    a.region_of = c->k3.region_of; a.partner = c->k4.partner; a.pair_lo = c->k4.pair_lo; a.meta = c->cp.meta; a.isize = c->cp.isize;
    a.in_groups = c->k6_in_groups; a.in_goff = c->k6_in_goff; a.first_of = c->k6_in_groups ? c->k6_in_goff : nullptr;
    a.parts = c->b_parts.as<PartRec>();
    a.rs = c->b_rs.as<RegSum>();
    a.out_deg = c->b_out_deg.as<uint32_t>(); a.label = a.out_deg + cap; a.bad_v = a.out_deg + 2 * cap; a.bad = a.out_deg + 3 * cap;
    a.mcount = a.out_deg + 4 * cap; a.pcount = a.out_deg + 5 * cap;
    a.members = c->b_members.as<MemberInfo>();
    a.own_nsv = c->b_own.as<uint32_t>(); a.own_nacc = a.own_nsv + cap; a.own_ncn = a.own_nsv + 2 * cap; a.own_first = a.own_nsv + 3 * cap; a.slot_next = a.own_nsv + 4 * cap; a.owners = a.own_nsv + 5 * cap; a.owners_big = a.own_nsv + 6 * cap; a.emit_part = a.own_nsv + 7 * cap;
    a.member_ids = big_walk ? c->b_member_ids.as<uint32_t>() : nullptr;
    a.sv_stage = c->b_slot.as<SvOut>(); a.lib_stage = c->b_lib_stage.as<LibStage>(); a.cn_stage = c->b_cn_stage.as<CnStage>();
    if (c->table_in_hbm) {
        a.sv_out = c->b_sv_out.as<SvOut>(); a.lib_index = c->b_lib_index_out.as<int32_t>(); a.lib_pairs = c->b_lib_pairs_out.as<int32_t>();
        a.cn_key = c->b_cn_key_out.as<int32_t>(); a.cn_value = c->b_cn_value_out.as<float>();
        a.sv_key = c->b_sv_key.as<unsigned long long>();
    } else {
        a.sv_out = c->h_sv_out.as<SvOut>(); a.lib_index = c->h_lib_index.as<int32_t>(); a.lib_pairs = c->h_lib_pairs.as<int32_t>();
        a.cn_key = c->h_cn_key.as<int32_t>(); a.cn_value = c->h_cn_value.as<float>();
    }
    a.sv_vx = a.sv_key ? a.sv_src + a.sv_cap : nullptr;
    a.t_lambda = c->b_t_lambda.as<double>(); a.t_k = c->b_t_k.as<int32_t>();
    a.g_rec = c->k4.g_rec; a.g_cap = c->k4.g_cap;
    {   // look-back words of the table scan: zero once, afterwards every run brings its own stamp
        const size_t words = 4 * nblk;
        if (c->b_ws6.bytes < words * 8) {
            HIPCHK(c, c->b_ws6.ensure(words * 8));
            HIPCHK(c, hipMemsetAsync(c->b_ws6.p, 0, c->b_ws6.bytes, s));
        }
        a.lb_state = c->b_ws6.as<unsigned long long>();
        if (!sz) HIPCHK(c, next_lb_stamp(c, &a.lb_stamp));
    }
    if (sz) return BDX_OK;
    a.counts = c->b_counts.as<StageCounts>();
    // run constants: the flag histogram is the device's own reduced counter table (a single-context run adopts its own
    // statistics), the read densities per counter key travel in the kernel arguments
    a.hist = c->b_cnt.as<uint32_t>();
    a.key_density = c->b_kdens.as<float>();
    a.lib_mean = c->b_lib_mean.as<float>();
    a.counts_host = c->h_counts.as<StageCounts>();
    a.counts_host2 = c->h_counts2.as<StageCounts>();
    memset(c->h_counts.p, 0, sizeof(StageCounts));
    memset(c->h_counts2.p, 0, sizeof(StageCounts));
    a.p1 = c->b_p1.as<Pass1>();
    a.nlibs = nlibs; a.nkeys = nkeys; a.min_read_pair = c->opts.min_read_pair; a.chr_restricted = c->opts.chr_restricted;
    a.period = std::max(1, c->opts.buffer_size + 1);
    a.force_host = force_host ? 1 : 0;
    a.big_walk = big_walk;
    a.ins_plain = c->dbg_ins_plain;
    a.asm_plain = c->dbg_asm_plain;
    a.walk_lanes = c->dbg_walk_lanes > 0 ? std::min(c->dbg_walk_lanes, 32) : 32;  // (measured at configs[1]: 64 regions per wave 23.7 us, 32: 22.4 us, 16: 24.5 us)
    {
        const int rounds = c->dbg_label_rounds;
        // (long chains need more rounds to agree on one label; with the general walk on, the step is long enough not to care)
        a.label_rounds = rounds ? rounds : (a.big_walk ? kK6LabelRoundsBig : kK6LabelRounds);
    }
    if (c->poll) {  // ready words set by the kernels themselves (first thread of k6_pairs_kernel / first wave of k6_walk_kernel)
        a.flag_value = c->seq;
        a.flag_groups = c->h_flags.as<uint32_t>() + 1;
        if (c->k3.host_copy_later && !c->region_dma_now) a.flag_regions = c->h_flags.as<uint32_t>() + 3;
        a.mirror_in_walk = (force_host || c->defer_walk) ? 0 : 1;  // (k6_walk_kernel follows k6_emit_kernel unless everything goes to the host, or the walk waits for the ranks' collectives)
    }
    if (part == 4) return BDX_OK;   // (the arrays only: rank 0 of a sharded run whose host walks the few gathered groups -- the table stage follows)
    if (part == 1) {
        launch_k6_pairs(a, na, s);
        return BDX_OK;
    }
    launch_k6_groups(a, na, s);
    if (!c->poll) {  // the host's share of the groups is complete
        const int rc = signal_ready(c, 1, c->ev_groups);
        if (rc != BDX_OK) return rc;
    }
    launch_k6_walk(a, na, s);
    return BDX_OK;
}

// bdx_reserve: the stage functions run for their allocations only
int presize_stages(bdx_ctx* c, uint32_t na) {
    // (may run on a thread of its own beside the thread that feeds the context: it touches the later stages' buffers and nothing else --
    // see bdx_ctx::sizing.  Its messages go to sizing_err.)
    struct Guard {
        bdx_ctx* c;
        std::string* keep;
        explicit Guard(bdx_ctx* c_) : c(c_), keep(t_err_sink) { c->sizing.fetch_add(1, std::memory_order_acq_rel); c->sizing_err.clear(); t_err_sink = &c->sizing_err; }
        ~Guard() { t_err_sink = keep; c->sizing.fetch_sub(1, std::memory_order_acq_rel); }
    } guard(c);
    const Sizing sz{na};
    int r = do_compact(c, 0, nullptr, true, &sz);
    if (r == BDX_OK) r = do_cut(c, 0, 0, 0, true, false, &sz);
    if (r == BDX_OK) r = do_join_local(c, na, Entries{}, nullptr, true, &sz);
    if (r == BDX_OK) r = do_k6(c, false, 0, &sz);
    return r;
}

// a sizing pass on the caller's own thread: its message is the context's
int presize_stages_here(bdx_ctx* c, uint32_t na) {
    const int r = presize_stages(c, na);
    if (r != BDX_OK) c->err = c->sizing_err;
    return r;
}

// second half of K6, enqueued once the host walk has produced its share: the host's SV candidates (pinned memory) are
// interleaved with the device's by the compaction, K5 scores every term, the score kernel finishes every candidate --
// the final table is assembled by the device in pinned host memory, in the reference's output order.
int do_k6_table(bdx_ctx* c) {
    hipStream_t s = c->stream;
    K6Arrays& a = c->k6;
    const uint32_t na = a.cap;
    if (!na) return BDX_OK;
    const WalkResult& H = c->walk;
    const uint32_t nh = (uint32_t)H.svs.size(), nt = (uint32_t)H.terms.size(), nc = (uint32_t)H.cn_key.size();
    a.nh = nh;
    if (nh) {
        const uint64_t period = (uint64_t)std::max(1, c->opts.buffer_size + 1);
        HIPCHK(c, c->h_hs_rec.ensure((size_t)nh * sizeof(SvOut)));
        HIPCHK(c, c->h_hs_aux.ensure((size_t)nh * 12 + 16));
        HIPCHK(c, c->h_hs_lists.ensure((size_t)nt * 16 + (size_t)nc * 8 + 16));
        memcpy(c->h_hs_rec.p, H.svs.data(), (size_t)nh * sizeof(HostSv));
        uint64_t* hkey = c->h_hs_aux.as<uint64_t>();
        uint32_t* hcnt = (uint32_t*)(hkey + nh);
        for (uint32_t j = 0; j < nh; ++j) {
            // order key (see K6Arrays): a traversal started at one of its window's own vertices is placed right before the
            // device's candidates of that start vertex, one started from an earlier window's vertex before all of its window
            const uint64_t key = H.sv_key[j];
            const bool from_old = !((key >> 32) & 1ull);
            const uint64_t start = key & 0xffffffffull;
            const uint64_t T = from_old ? (key >> 33) * period : start;
            hkey[j] = (T << 34) | (from_old ? 0ull : 1ull << 33) | (start << 7);
            if (a.force_host) hkey[j] = 0;  // no device candidates to interleave with (and the ids may carry the phantom shift)
            hcnt[j] = (uint32_t)H.svs[j].sv.lib_count | ((uint32_t)H.svs[j].sv.cn_count << 16);
        }
        double* lam = c->h_hs_lists.as<double>();
        int32_t* li = (int32_t*)(lam + nt);
        int32_t* lp = li + nt;
        int32_t* ck = lp + nt;
        float* cv = (float*)(ck + nc);
        for (uint32_t i = 0; i < nt; ++i) { lam[i] = H.terms[i].lambda; li[i] = H.lib_index[i]; lp[i] = H.lib_pairs[i]; }
        for (uint32_t i = 0; i < nc; ++i) { ck[i] = H.cn_key[i]; cv[i] = H.cn_value[i]; }
        a.hs_rec = c->h_hs_rec.as<SvOut>(); a.hs_key = hkey; a.hs_cnt = hcnt;
        a.hs_lambda = lam; a.hs_lib_index = li; a.hs_lib_pairs = lp; a.hs_cn_key = ck; a.hs_cn_value = cv;
    }
    // (K5 runs inside the table kernel.  The terms' log tails go to the host for Fisher's combination only -- BreakDancer.cpp:71-81 uses the
    // host's exp / log --; otherwise they are 8 bytes per term over PCIe that nobody reads)
    a.ltail_host = c->table_in_hbm ? c->b_ltail_out.as<double>() : (c->opts.fisher ? c->h_ltail_dev.as<double>() : nullptr);
    // (the table kernel's launch: for the regions the pair groups' report named -- the host has read it by now -- not for their upper bound)
    a.fin_regions = (c->counts.n_regions && c->counts.n_regions <= na) ? c->counts.n_regions : na;
    HIPCHK(c, c->h_printed.ensure((size_t)k6_score_grid(a) * 4));
    a.printed_host = c->h_printed.as<uint32_t>();
    // The word the host polls for the end of the run is set by a one-thread kernel behind the table kernel (a kernel boundary
    // orders it behind that kernel's stores to host memory).  A stream write-value command does the same as a one-thread kernel of
    // the runtime's own, but starts 5 us after the kernel before it has ended; back-to-back launches follow each other at once.
    const bool write_value = c->dbg_end_write_value != 0;  // (A/B)
    if (c->poll && !write_value) { a.flag_done = c->h_flags.as<uint32_t>() + 2; a.flag_value = c->seq; }
    a.wire_rows = c->table_in_hbm ? 0 : 1;   // (a table that stays in HBM for rank 0's merge keeps its full rows)
    c->rows_packed = a.wire_rows != 0;
    launch_k6_table(a, na, std::log(10), c->opts.score_threshold, c->opts.fisher ? 0 : 1, s);
    if (!a.flag_done) {  // (without polling: finish_table waits for the stream)
        const int rc = signal_ready(c, 2, nullptr);
        if (rc != BDX_OK) return rc;
    }
    return BDX_OK;
}

// copy the final table out of the pinned buffers the device assembled it in (getters, support lists, Fisher scores)
int materialize(bdx_ctx* c) {
    if (c->materialized) return BDX_OK;
    WalkResult& M = c->walk;
    M.clear();
    const uint32_t n = c->n_sv_total, nt = c->n_terms_total, nc = c->n_cn_total;
    if (c->rows_packed) {
        // the rows crossed the link as SvWire: chromosomes and strand counts are the regions' (SvBuilder.cpp:18-99 takes them from there too),
        // the list offsets the running sums of the counts in table order
        const SvWire* w = c->h_sv_out.as<SvWire>();
        M.svs.resize(n);
        int32_t lb = 0, cb = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const SvWire& r = w[i];
            HostSv& o = M.svs[i];
            bdx_sv& v = o.sv;
            const HostRegion& ra = c->reg[r.region[0]];
            const HostRegion& rb = r.region[1] >= 0 ? c->reg[r.region[1]] : ra;
            v.chr[0] = ra.tid; v.chr[1] = rb.tid;
            v.pos[0] = r.pos[0]; v.pos[1] = r.pos[1];
            v.fwd[0] = (int32_t)(ra.n - ra.rev); v.rev[0] = (int32_t)ra.rev;
            v.fwd[1] = (int32_t)(rb.n - rb.rev); v.rev[1] = (int32_t)rb.rev;
            v.flag = (int32_t)((r.bits >> 16) & 15u); v.size = r.size; v.score = r.score; v.num_reads = r.num_reads; v.printed = (int32_t)((r.bits >> 23) & 1u);
            v.region[0] = r.region[0]; v.region[1] = r.region[1];
            v.lib_begin = lb; v.lib_count = (int32_t)(r.bits & 255u); lb += v.lib_count;
            v.cn_begin = cb; v.cn_count = (int32_t)((r.bits >> 8) & 255u); cb += v.cn_count;
            v.allele_frequency = r.allele_frequency; v.logp = r.logp;
            o.grp_mask = (r.bits >> 20) & 7u; o.start = r.start;
        }
    } else {
        const HostSv* d = c->h_sv_out.as<HostSv>();
        M.svs.assign(d, d + n);
    }
    M.lib_index.assign(c->h_lib_index.as<int32_t>(), c->h_lib_index.as<int32_t>() + nt);
    M.lib_pairs.assign(c->h_lib_pairs.as<int32_t>(), c->h_lib_pairs.as<int32_t>() + nt);
    M.cn_key.assign(c->h_cn_key.as<int32_t>(), c->h_cn_key.as<int32_t>() + nc);
    M.cn_value.assign(c->h_cn_value.as<float>(), c->h_cn_value.as<float>() + nc);
    if (c->opts.fisher) c->log_tail.assign(c->h_ltail_dev.as<double>(), c->h_ltail_dev.as<double>() + nt); else c->log_tail.clear();
    M.n_groups = c->n_groups_total;
    c->materialized = true;
    return BDX_OK;
}

// end of a single-context run: wait for the device's table
int finish_table(bdx_ctx* c) {
    hipStream_t s = c->stream;
    const auto tf0 = std::chrono::steady_clock::now();
    if (!wait_flag(c, 2, c->seq)) {
        HIPCHK(c, hipStreamSynchronize(s));
        if (!flag_arrived(c, 2)) return fail(c, BDX_EINTERNAL, "the SV table did not arrive: its kernels were not launched");
    }
    const auto tf1 = std::chrono::steady_clock::now();
    static_assert(sizeof(HostSv) == sizeof(SvOut) && offsetof(HostSv, grp_mask) == offsetof(SvOut, grp_mask) &&
                      offsetof(HostSv, start) == offsetof(SvOut, start), "SV record layout");
    c->n_sv_host = (uint32_t)c->walk.svs.size();
    c->n_groups_total = c->walk.n_groups + c->counts.n_groups_dev;
    const StageCounts c2 = *c->h_counts2.as<StageCounts>();
    if (c2.overflow) return fail(c, BDX_EINTERNAL, "SV list overflow");
    c->n_sv_total = c2.n_sv_dev; c->n_terms_total = c2.n_terms_dev; c->n_cn_total = c2.n_cn_dev;
    c->counts.n_sv_dev = c2.n_sv_dev - c->n_sv_host;
    {
        // printed candidates: every workgroup of the score kernel left its count (no run: none)
        const uint32_t g = c->k6.printed_host ? k6_score_grid(c->k6) : 0u;
        const uint32_t* pb = c->k6.printed_host;
        uint32_t np = 0;
        for (uint32_t i = 0; i < g; ++i) np += pb[i];
        c->n_printed = np;
    }
    c->counts.n_old = c2.n_old;
    c->materialized = false;
    if (c->opts.fisher && !c->table_in_hbm) {  // Fisher's combination (BreakDancer.cpp:71-81) uses the host's exp / log (sharded: after the merge)
        materialize(c);
        finish_scores(c->opts, c->log_tail.data(), c->walk.svs.data(), c->walk.svs.size(), &c->n_printed);
    }
    const auto tf2 = std::chrono::steady_clock::now();
    c->stage_ms[8] = ms_between(tf0, tf1);
    c->stage_ms[9] = ms_between(tf1, tf2);
    c->stage_ms[10] = 0;
    c->ran = true;
    c->stage = 4;
    return BDX_OK;
}

// H1 walk over c->regions / c->r_pk / c->parts with the adopted pass-1 statistics
int host_walk(bdx_ctx* c, int32_t last_maxq, bool any_anomalous) {
    HIPCHK(c, hipSetDevice(c->device));
    WalkInput wi{};
    wi.opts = c->opts; wi.libs = c->libs.data(); wi.nlibs = c->nlibs; wi.nbams = c->nbams; wi.nkeys = c->nkeys;
    wi.hist = c->cnt.data(); wi.covered_ref_len = c->g_covered; wi.key_density = c->key_density.data();
    wi.regions = c->reg; wi.nregions = c->nreg; wi.r_pk = c->rpk; wi.parts = &c->parts; wi.last_maxq = last_maxq;
    wi.any_anomalous = any_anomalous;
    c->walk.clear();
    if (!c->parts.empty()) greedy_walk(wi, c->walk_scratch, c->walk);
    return BDX_OK;
}

// K5 for the host walk's terms (staged runs: the whole walk is the host's)
int score_host_terms(bdx_ctx* c) {
    hipStream_t s = c->stream;
    const uint32_t nt = (uint32_t)c->walk.terms.size();
    if (nt) {
        // zero-copy: the kernel reads lambda / k from pinned host memory and writes the log tails back into it (a few
        // tens of KB; the PCIe traffic overlaps the kernel and no copy engine round trips are paid)
        HIPCHK(c, c->h_terms.ensure((size_t)nt * 20 + 16));
        double* hl = c->h_terms.as<double>();
        int32_t* hk = (int32_t*)(hl + nt);
        double* ho = (double*)((char*)c->h_terms.p + (((size_t)nt * 12 + 7) & ~(size_t)7));
        for (uint32_t i = 0; i < nt; ++i) { hl[i] = c->walk.terms[i].lambda; hk[i] = c->walk.terms[i].k; }
        launch_k5(hl, hk, ho, nt, s);
    }
    return BDX_OK;
}

// staged runs: the walk, the list and the score combination are the host's
int finish_host_walk(bdx_ctx* c) {
    hipStream_t s = c->stream;
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    const uint32_t nt = (uint32_t)c->walk.terms.size();
    const double* host_tail = nt ? (const double*)((char*)c->h_terms.p + (((size_t)nt * 12 + 7) & ~(size_t)7)) : nullptr;
    c->log_tail.assign(host_tail, host_tail + nt);
    finish_scores(c->opts, c->log_tail.data(), c->walk.svs.data(), c->walk.svs.size(), &c->n_printed);
    c->n_sv_host = c->n_sv_total = (uint32_t)c->walk.svs.size();
    c->n_terms_total = nt; c->n_cn_total = (uint32_t)c->walk.cn_key.size();
    c->n_groups_total = c->walk.n_groups;
    c->counts.n_sv_dev = 0;
    c->materialized = true;
    c->ran = true;
    c->stage = 4;
    return BDX_OK;
}

// Supporting reads per SV, in SvBuilder's observation order (SvBuilder.cpp:101-118): reads of region A then region B
// are visited in stream order and a pair is recorded when its second mate shows up, second mate first.
int collect_support(bdx_ctx* c, uint32_t ph) {
    c->sup_off.assign(1, 0);
    c->sup_idx.clear();
    c->sup_flag.clear();
    const uint32_t na = c->p1.n_anom;
    std::vector<int32_t> partner(na), region_of(na);
    std::vector<uint32_t> idx(na), meta(na);
    if (na) {
        HIPCHK(c, hipMemcpy(partner.data(), c->k4.partner, (size_t)na * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(region_of.data(), c->k3.region_of, (size_t)na * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(idx.data(), c->cp.idx, (size_t)na * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(meta.data(), c->cp.meta, (size_t)na * 4, hipMemcpyDeviceToHost));
    }
    std::vector<std::pair<uint32_t, uint32_t>> pairs;  // (second-observed j, its mate p)
    for (const HostSv& hs : c->walk.svs) {
        pairs.clear();
        const uint32_t rA = (uint32_t)hs.sv.region[0], rB = (uint32_t)hs.sv.region[1];
        const uint32_t glo[3] = {rA, rA, rB}, ghi_[3] = {rA, rB, rB};
        for (uint32_t g = 0; g < 3; ++g) {
            if (!(hs.grp_mask & (1u << g))) continue;
            const uint32_t lo = glo[g], hi = ghi_[g];
            const uint32_t f = c->reg[hi].first, n = c->reg[hi].n;
            for (uint32_t j = f; j < f + n; ++j) {
                const int32_t p = partner[j];
                if (p < 0 || (uint32_t)p >= j) continue;
                if ((uint32_t)(region_of[p] + (int32_t)ph) != lo) continue;
                pairs.emplace_back(j, (uint32_t)p);
            }
        }
        std::sort(pairs.begin(), pairs.end());
        for (auto const& pr : pairs) {
            c->sup_idx.push_back(idx[pr.first]); c->sup_flag.push_back((uint8_t)meta_flag(meta[pr.first]));
            c->sup_idx.push_back(idx[pr.second]); c->sup_flag.push_back((uint8_t)meta_flag(meta[pr.second]));
        }
        c->sup_off.push_back((uint32_t)c->sup_idx.size());
    }
    return BDX_OK;
}

// The read-level replay (H2) over compact records in stream order -- name key, accepted region id (or -1), meta, |isize| -- on the
// context's region table (c->reg / c->rpk) and pass-1 statistics; scores from K5.  Shared by the single-context replay below and
// by rank 0 of a sharded run, which gathers the records of all chromosomes first.
int replay_arrays(bdx_ctx* c, uint32_t na, const uint64_t* key, const int32_t* region_of, const uint32_t* meta, const int32_t* isize, uint32_t ph,
                  std::vector<uint32_t>* sup) {
    ReadWalkInput ri{};
    WalkInput& wi = ri.base;
    wi.opts = c->opts; wi.libs = c->libs.data(); wi.nlibs = c->nlibs; wi.nbams = c->nbams; wi.nkeys = c->nkeys;
    wi.hist = c->cnt.data(); wi.covered_ref_len = c->g_covered; wi.key_density = c->key_density.data();
    wi.regions = c->reg; wi.nregions = c->nreg; wi.r_pk = c->rpk; wi.parts = nullptr; wi.last_maxq = c->counts.last_maxq;
    wi.any_anomalous = na != 0;
    ri.n_reads = na; ri.key = key; ri.region_of = region_of; ri.meta = meta; ri.isize = isize;
    ri.phantom = ph;
    if (sup) { ri.support_off = &c->sup_off; ri.support = sup; }
    c->walk.clear();
    read_level_walk(ri, c->walk);
    c->counts.n_groups = 0; c->counts.n_pairs = 0;
    int rc = score_host_terms(c);
    if (rc == BDX_OK) rc = finish_host_walk(c);
    if (rc != BDX_OK) return rc;
    c->replayed = true;
    return BDX_OK;
}

// The read-level replay tells names apart by one 64-bit word.  With a second name hash in the stream, two reads are the same
// name only if both words agree: every (key, check) pair gets a number of its own, which then serves as the key.
void unify_names(uint64_t* key, const uint64_t* check, size_t n) {
    struct PairHash {
        size_t operator()(const std::pair<uint64_t, uint64_t>& p) const { return (size_t)(p.first ^ (p.second * 0x9E3779B97F4A7C15ull)); }
    };
    std::unordered_map<std::pair<uint64_t, uint64_t>, uint64_t, PairHash> ids;
    ids.reserve(n);
    for (size_t i = 0; i < n; ++i) key[i] = ids.emplace(std::make_pair(key[i], check[i]), (uint64_t)ids.size()).first->second;
}

// A read name occurs more than twice (StageCounts::irregular): everything behind the region cut is replayed one read at a
// time on the host (H2), from the compact records; the scores still come from K5.  Names clashing across merged BAMs are the
// usual cause -- the reference keeps running on them (ReadRegionData.cpp:108-113), so does this.
int replay_reads(bdx_ctx* c, uint32_t ph) {
    hipStream_t s = c->stream;
    HIPCHK(c, hipStreamSynchronize(s));  // K6's kernels were enqueued on the pair model: let them finish, their results are dropped
    const uint32_t na = c->p1.n_anom;
    std::vector<uint64_t> key(na);
    std::vector<int32_t> region_of(na), isize(na);
    std::vector<uint32_t> meta(na);
    HIPCHK(c, hipMemcpy(key.data(), c->cp.key, (size_t)na * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(region_of.data(), c->k3.region_of, (size_t)na * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(meta.data(), c->cp.meta, (size_t)na * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(isize.data(), c->cp.isize, (size_t)na * 4, hipMemcpyDeviceToHost));
    if (c->cp.check && na) {
        std::vector<uint64_t> check(na);
        HIPCHK(c, hipMemcpy(check.data(), c->cp.check, (size_t)na * 8, hipMemcpyDeviceToHost));
        unify_names(key.data(), check.data(), na);
    }
    if (ph)
        for (int32_t& r : region_of)
            if (r >= 0) r += (int32_t)ph;
    std::vector<uint32_t> sup;
    int rc = replay_arrays(c, na, key.data(), region_of.data(), meta.data(), isize.data(), ph, c->collect_support ? &sup : nullptr);
    if (rc != BDX_OK) return rc;
    if (c->collect_support) {  // compact indices -> stream indices and flags
        std::vector<uint32_t> idx(na);
        HIPCHK(c, hipMemcpy(idx.data(), c->cp.idx, (size_t)na * 4, hipMemcpyDeviceToHost));
        c->sup_idx.resize(sup.size());
        c->sup_flag.resize(sup.size());
        for (size_t i = 0; i < sup.size(); ++i) { c->sup_idx[i] = idx[sup[i]]; c->sup_flag[i] = (uint8_t)meta_flag(meta[sup[i]]); }
    }
    return BDX_OK;
}

}  // namespace

extern "C" {

int bdx_run(bdx_ctx* c) {
    if (!c) return BDX_EINVAL;
    NOT_WHILE_SIZING(c);
    const auto t_begin = std::chrono::steady_clock::now();
    hipStream_t s = c->stream;
    // The very first anomalous read "breaks" an empty accumulator (start = end = -1, no reads).  With a negative
    // -s that empty candidate passes process_breakpoint's test (0 > min_len, coverage 0) and the reference
    // registers a read-less region 0 (BreakDancer.cpp:216-231, 244-252); every real region id shifts by one.
    const bool ph_opt = 0 > c->opts.min_len && 0.0f < (float)c->opts.seq_coverage_lim;
    // shifted region ids and a non-positive -r are left to the host walk entirely
    const bool force_host = c->host_walk_only || ph_opt || c->opts.min_read_pair < 1;
    // K2 .. K6 (first half) for c->na_alloc anomalous reads
    auto enqueue_middle = [&]() -> int {
        c->region_dma_now = false;
        int r = do_compact(c, 0, nullptr, true);
        if (r != BDX_OK) return r;
        r = do_cut(c, 0, 0, 0, true);
        if (r != BDX_OK) return r;
        if (!c->na_alloc) return BDX_OK;
        // the region table is final after K3: the host takes its copy while the device joins the mates
        Entries en{};
        en.key = c->cp.key; en.check = c->cp.check; en.region = c->k3.region_of; en.meta = c->cp.meta; en.isize = c->cp.isize;
        if (c->region_of_fused) {
            en.cand = c->k3.cand; en.c_rid = c->k3.c_rid; en.region_out = c->k3.region_of;
            en.k6_scratch = c->k3.out_deg; en.scratch_cap = c->k3.cap;
            // (the join kernel forwards the region table to pinned host memory.  A kernel of its own on the copy stream, beside the join, was
            // measured in round 6: 1.525 against 1.530 ms at a genome share -- the join does not wait for those 5.7 MB; profiles/r06_genome_ab.txt)
            c->region_dma_now = c->k3.host_copy_later && c->poll && c->dbg_region_dma != 0;
            if (c->region_dma_now) {
                // (the join kernel only says that K3 is through: the host reads the region count and has the copy engine fetch the table)
                en.flag_host = c->h_flags.as<uint32_t>() + 4; en.flag_value = c->seq;
            } else if (c->k3.host_copy_later) {
                en.r_rec_dev = c->k3.r_rec_dev; en.r_pk_dev = c->k3.r_pk_dev; en.r_rec_host = c->k3.r_rec; en.r_pk_host = c->k3.r_pk;
                en.counts = c->b_counts.as<StageCounts>(); en.nkeys2 = 2 * c->nkeys;
                en.fwd_blocks = c->dbg_join_fwd < 0 ? 0xFFFFFFFFu : (uint32_t)c->dbg_join_fwd;
            }
        }
        r = do_join_local(c, c->na_alloc, en, &c->b_p1.as<Pass1>()->n_anom, true);
        if (r != BDX_OK) return r;
        if (c->k3.host_copy_later && !c->poll) {  // the join kernel has forwarded the region table to pinned memory
            r = signal_ready(c, 3, c->ev_regions);  // (polling: the next kernel, k6_pairs_kernel, sets the ready word itself)
            if (r != BDX_OK) return r;
        }
        return do_k6(c, force_host);
    };
    // enqueue-ahead when this context has just run an input of the same size
    uint32_t guess = 0;
    const bool restored = !c->ov_cnt.empty();  // (restored statistics are adopted between pass 1 and the rest: nothing ahead)
    if (!restored && c->speculate == 2 && c->ran && c->last_n == c->n && c->last_na) {
        guess = c->spec_test ? std::max(1u, c->last_na / 2) : c->last_na + c->last_na / 8 + 1024;
        if (guess > kMaxAnomalous) guess = 0;
    } else if (!restored && c->speculate && c->n >= (1u << 20) && (c->speculate == 1 || c->n < (1u << 25))) {
        // no history: a prior.  Anomalous reads are a few percent of a sorted BAM at most (1 % at configs[1]); 1/32 of the
        // reads covers that with room, costs a few microseconds of oversized grids when it is generous, and one more pass
        // over the (short) later stages when it is not.  Small inputs are not worth it: their whole run is launch latency.
        // (measured at configs[1], 1.1 % anomalous: prior n/32 0.309 ms per step, n/64 0.300 ms, exact sizing after the read-back
        // 0.307 ms, sizing from the previous run 0.295 ms -- an oversized launch grid costs about what the host round trip does)
        // (a GPU's share of a genome -- 2^25 reads and more -- does not take the prior by itself: K1's half millisecond covers the launches either way,
        // and a prior twice the truth, eighteen times with -t, costs its oversized grids and tables: 1.534 against 1.514 ms, 0.92 against 0.83 with -t.
        // The compaction ALONE ahead under the prior, the rest sized exactly, was built too: 1.656 against 1.612 ms, 0.90 against 0.83 -- the tables K2
        // clears for K3 and K4 are sized by the guess.  profiles/r06_genome_ab.txt)
        const uint64_t prior = (uint64_t)c->n / (c->spec_test ? 4096 : 32) + 4096;
        guess = prior > kMaxAnomalous ? 0 : (uint32_t)prior;
    }
    int rc;
    if (restored) {
        // restored pass-1 statistics (a cache written by an earlier run: ConfigLoader.cpp:19-23): the classifier still runs --
        // pass 2 needs its class bytes -- but window, densities, lambda and the printed counters come from the cache
        rc = do_pass1(c);
        if (rc != BDX_OK) return rc;
        rc = set_pass1(c, c->ov_cnt.data(), c->ov_covered, window_from(c, c->ov_cnt.data(), c->ov_covered), true);
        if (rc != BDX_OK) return rc;
        HIPCHK(c, hipMemcpyAsync(c->b_cnt.p, c->ov_cnt.data(), c->ov_cnt.size() * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_kdens.p, c->key_density.data(), c->key_density.size() * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));
        rc = enqueue_middle();
        if (rc != BDX_OK) return rc;
    } else if (guess) {
        rc = do_pass1(c, guess, false, true);
        if (rc != BDX_OK) return rc;
        c->na_alloc = guess;
        rc = enqueue_middle();
        if (rc != BDX_OK) return rc;
        rc = wait_pass1(c);
        if (rc != BDX_OK) return rc;
        rc = set_pass1(c, c->cnt_local.data(), c->p1.covered_ref_len, c->p1.window, false);
        if (rc != BDX_OK) return rc;
        if (c->p1.n_anom > guess) {  // more anomalous reads than guessed: the enqueued stages saw none; run them properly
            HIPCHK(c, hipStreamSynchronize(s));
            HIPCHK(c, hipMemcpy(&c->b_p1.as<Pass1>()->n_anom, &c->p1.n_anom, 4, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemset(c->b_counts.p, 0, sizeof(StageCounts)));
            ++c->seq;  // fresh ready-words: the ones of the neutralised launches are already set
            c->na_alloc = c->p1.n_anom;
            rc = enqueue_middle();
            if (rc != BDX_OK) return rc;
        }
    } else {
        rc = do_pass1(c);
        if (rc != BDX_OK) return rc;
        rc = set_pass1(c, c->cnt_local.data(), c->p1.covered_ref_len, c->p1.window, false);
        if (rc != BDX_OK) return rc;
        rc = enqueue_middle();
        if (rc != BDX_OK) return rc;
    }
    c->last_n = c->n;
    c->last_na = c->p1.n_anom;
    const uint32_t na = c->p1.n_anom;
    const uint32_t ph = (na && ph_opt) ? 1u : 0u;
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[5], s));
    const auto t_h0 = std::chrono::steady_clock::now();
    auto t_h1 = t_h0;
    if (na && c->region_dma_now) {
        if (!wait_flag(c, 4, c->seq)) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (!flag_arrived(c, 4)) return fail(c, BDX_EINTERNAL, "the region count did not arrive: its kernels were not launched");
        }
        const size_t nr = c->h_counts0.as<StageCounts>()->n_regions;
        if (nr) {
            HIPCHK(c, hipMemcpyAsync(c->h_regs.p, c->b_r_rec.p, nr * sizeof(RegionRec), hipMemcpyDeviceToHost, c->copy_stream));
            if (c->nkeys) HIPCHK(c, hipMemcpyAsync(c->h_pk.p, c->b_r_pk.p, nr * 2 * (size_t)c->nkeys * 4, hipMemcpyDeviceToHost, c->copy_stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    } else if (na) {
        if (!wait_flag(c, 3, c->seq)) {
            if (c->poll) HIPCHK(c, hipStreamSynchronize(s)); else HIPCHK(c, hipEventSynchronize(c->ev_regions));
            if (!flag_arrived(c, 3)) return fail(c, BDX_EINTERNAL, "the region table did not arrive: its kernels were not launched");
        }
    }
    if (na) {
        // (the table sits in pinned memory the device has just written: one streaming copy into ordinary memory is much
        // cheaper than the walk's scattered reads of it)
        if ((uint64_t)c->h_counts0.as<StageCounts>()->n_regions + ph > kMaxRegions) {   // (region ids are 26-bit fields of the packed group key)
            HIPCHK(c, hipStreamSynchronize(s));
            return fail(c, BDX_ELIMIT, "more than 2^26 - 2 accepted regions in one context");
        }
        // The host's share of the walk reads the table where the device left it, in pinned memory.  (Until round 6 the table was copied into
        // ordinary memory first -- "one streaming copy is cheaper than the walk's scattered reads", true when the host walked every component:
        // 5.7 MB at a genome share, 0.28 ms of this thread between the join kernel's end and the pair groups' arrival 0.14 ms later; the
        // host's walk of its ~1,200 groups then started late and the device stood idle in front of the table stage.)  A large share
        // (debug "regions_copy" = 1: always) still gets its copy, once the groups have said how large it is.
        const uint32_t nr_host = c->h_counts0.as<StageCounts>()->n_regions;
        // (a small table is copied at once, as ever: its copy fits between the join kernel's end and the groups' arrival -- 32 k regions are 70 us)
        const bool copy_first = c->dbg_regions_copy == 1 || (c->dbg_regions_copy == 0 && nr_host < kBorrowRegionsMin);
        if (copy_first) decode_regions(c, c->h_regs.as<RegionRec>(), c->h_pk.as<uint32_t>(), nr_host, ph, false);
        if (!wait_flag(c, 1, c->seq)) {
            if (c->poll) HIPCHK(c, hipStreamSynchronize(s)); else HIPCHK(c, hipEventSynchronize(c->ev_groups));
            if (!flag_arrived(c, 1)) return fail(c, BDX_EINTERNAL, "the pair groups did not arrive: their kernels were not launched");
        }
        c->counts = *c->h_counts.as<StageCounts>();
        if (!copy_first) decode_regions(c, c->h_regs.as<RegionRec>(), c->h_pk.as<uint32_t>(), nr_host, ph,
                                        c->dbg_regions_copy == 2 || (uint64_t)c->counts.n_groups * 8 < nr_host);   // (borrowed while the walk touches a fraction of the table:
                                        // a walk of 20 k groups over 15 k regions was 46 us faster on its copy)
        if (c->counts.irregular) {  // a read name seen more than twice: the pair model does not hold (see bdx_walk_reads.cpp)
            t_h1 = std::chrono::steady_clock::now();
            rc = replay_reads(c, ph);
            if (rc != BDX_OK) return rc;
            const auto t_r = std::chrono::steady_clock::now();
            c->stage_ms[4] = ms_between(t_h0, t_h1); c->stage_ms[5] = ms_between(t_h1, t_r); c->stage_ms[7] = ms_between(t_begin, t_r);
            return BDX_OK;
        }
        if (c->counts.overflow) return fail(c, BDX_EINTERNAL, "group list overflow");
        decode_groups(c, c->h_groups.as<GroupRec>(), c->counts.n_groups, ph);
        c->last_big_groups = (int64_t)c->counts.n_groups + c->counts.n_groups_big;  // (the host's share: mostly such components)
    }
    t_h1 = std::chrono::steady_clock::now();
    rc = host_walk(c, c->counts.last_maxq, na != 0);
    if (rc != BDX_OK) return rc;
    const auto t_h2 = std::chrono::steady_clock::now();
    if (na) {
        rc = do_k6_table(c);
        if (rc != BDX_OK) return rc;
        rc = finish_table(c);
    } else {
        rc = finish_host_walk(c);
    }
    if (rc != BDX_OK) return rc;
    if (c->collect_support) {
        materialize(c);
        rc = collect_support(c, ph);
        if (rc != BDX_OK) return rc;
    }
    const auto t_end = std::chrono::steady_clock::now();
    if (c->stage_timing) {
        (void)hipEventSynchronize(c->ev[5]);  // (complete by now; makes the elapsed-time queries valid)
        auto evms = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[a], c->ev[b]); return ms; };
        c->stage_ms[1] = evms(2, 3);
        c->stage_ms[2] = evms(3, 4);
        c->stage_ms[3] = evms(4, 5);
    }
    c->stage_ms[4] = ms_between(t_h0, t_h1);
    c->stage_ms[5] = ms_between(t_h1, t_h2);
    c->stage_ms[6] = ms_between(t_h2, t_end);
    c->stage_ms[7] = ms_between(t_begin, t_end);
    return BDX_OK;
}

// ---- staged entry points (multi-context / multi-GPU runs) ------------------------------------------------------
int bdx_stage_pass1(bdx_ctx* c) {
    if (!c) return BDX_EINVAL;
    NOT_WHILE_SIZING(c);
    return do_pass1(c);
}

int bdx_get_pass1_local(const bdx_ctx* c, uint32_t* counters, uint64_t* ref_len_per_bam, uint32_t* totals) {
    if (!c) return BDX_EINVAL;
    if (c->stage < 1) return BDX_ESTATE;
    if (counters) memcpy(counters, c->cnt_local.data(), c->cnt_local.size() * 4);
    if (ref_len_per_bam)
        for (int b = 0; b < c->nbams; ++b) ref_len_per_bam[b] = c->p1.ref_len[b];
    if (totals) {
        totals[0] = c->p1.n_anom; totals[1] = c->p1.n_normal;
        for (int k = 0; k < c->nkeys; ++k) totals[2 + k] = c->p1.key_tot[k];
    }
    return BDX_OK;
}

int bdx_set_pass1_global(bdx_ctx* c, const uint32_t* counters, uint32_t covered_ref_len, int32_t window) {
    if (!c || !counters) return BDX_EINVAL;
    if (c->stage < 1) return BDX_ESTATE;
    if (window < 0) window = window_from(c, counters, covered_ref_len);
    return set_pass1(c, counters, covered_ref_len, window, true);
}

int bdx_stage_compact(bdx_ctx* c, uint32_t nn_base, const uint32_t* pk_base, int32_t* first_qlen, uint32_t* first_nn) {
    if (!c) return BDX_EINVAL;
    if (c->stage < 2) return BDX_ESTATE;
    NOT_WHILE_SIZING(c);
    if (c->opts.min_len < 0) return fail(c, BDX_ELIMIT, "staged runs do not support a negative -s");
    int rc = do_compact(c, nn_base, pk_base, false);
    if (rc != BDX_OK) return rc;
    if (first_qlen) *first_qlen = 0;
    if (first_nn) *first_nn = 0;
    if (c->p1.n_anom) {  // the first anomalous read of this chromosome closes the previous chromosome's last candidate
        uint32_t meta = 0, nn = 0;
        HIPCHK(c, hipMemcpyAsync(&meta, c->cp.meta, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&nn, c->cp.nn, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (first_qlen) *first_qlen = meta_qlen(meta);
        if (first_nn) *first_nn = nn;
    }
    return BDX_OK;
}

int bdx_stage_regions(bdx_ctx* c, int has_next, int32_t next_qlen, uint32_t next_nn) {
    if (!c) return BDX_EINVAL;
    if (c->stage < 2) return BDX_ESTATE;
    NOT_WHILE_SIZING(c);
    int rc = do_cut(c, has_next, next_qlen, next_nn, false);
    if (rc != BDX_OK) return rc;
    return readback(c, false);
}

int bdx_get_stage_regions(const bdx_ctx* c, uint32_t* n_regions, uint32_t* n_anomalous, int32_t* last_maxq) {
    if (!c) return BDX_EINVAL;
    if (c->stage < 3) return BDX_ESTATE;
    if (n_regions) *n_regions = c->counts.n_regions;
    if (n_anomalous) *n_anomalous = c->p1.n_anom;
    if (last_maxq) *last_maxq = c->counts.last_maxq;
    return BDX_OK;
}

int bdx_get_region_records(const bdx_ctx* c, bdx_region_rec* out, uint32_t* pk, size_t cap) {
    if (!c) return BDX_EINVAL;
    if (c->stage < 3 || !c->h_regs.p) return BDX_ESTATE;   // (no table yet, or handed back by bdx_trim_results)
    const size_t n = std::min<size_t>(cap, c->counts.n_regions);
    static_assert(sizeof(bdx_region_rec) == sizeof(RegionRec), "region record layout");
    if (out && n) memcpy(out, c->h_regs.p, n * sizeof(RegionRec));
    if (pk && n) memcpy(pk, c->h_pk.p, n * 2 * c->nkeys * 4);
    return BDX_OK;
}

int bdx_get_compact(const bdx_ctx* c, uint64_t* key, int32_t* region, uint32_t* meta, int32_t* isize, size_t cap) {
    if (!c) return BDX_EINVAL;
    if (c->stage < 3) return BDX_ESTATE;
    const size_t n = std::min<size_t>(cap, c->p1.n_anom);
    if (!n) return BDX_OK;
    if (hipSetDevice(c->device) != hipSuccess) return BDX_EHIP;
    if (key && hipMemcpy(key, c->cp.key, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return BDX_EHIP;
    if (region && hipMemcpy(region, c->k3.region_of, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return BDX_EHIP;
    if (meta && hipMemcpy(meta, c->cp.meta, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return BDX_EHIP;
    if (isize && hipMemcpy(isize, c->cp.isize, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return BDX_EHIP;
    return BDX_OK;
}

int bdx_join_entries(bdx_ctx* c, size_t n, const uint64_t* key, const uint32_t* order, const int32_t* region, const uint32_t* meta,
                     const int32_t* isize, bdx_group* out, size_t cap, uint32_t* n_groups, uint32_t* n_pairs) {
    if (!c || (n && (!key || !order || !region || !meta || !isize))) return BDX_EINVAL;
    if (n > kMaxAnomalous) return fail(c, BDX_ELIMIT, "too many join entries");
    NOT_WHILE_SIZING(c);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (n_groups) *n_groups = 0;
    if (n_pairs) *n_pairs = 0;
    if (!n) return BDX_OK;
    HIPCHK(c, c->b_x_key.ensure(n * 8)); HIPCHK(c, c->b_x_order.ensure(n * 4)); HIPCHK(c, c->b_x_region.ensure(n * 4));
    HIPCHK(c, c->b_x_meta.ensure(n * 4)); HIPCHK(c, c->b_x_isize.ensure(n * 4)); HIPCHK(c, c->b_x_n.ensure(16));
    HIPCHK(c, c->b_counts.ensure(sizeof(StageCounts))); HIPCHK(c, c->h_counts.ensure(sizeof(StageCounts)));
    const uint32_t n32 = (uint32_t)n;
    HIPCHK(c, hipMemcpyAsync(c->b_x_key.p, key, n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->b_x_order.p, order, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->b_x_region.p, region, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->b_x_meta.p, meta, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->b_x_isize.p, isize, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->b_x_n.p, &n32, 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->b_counts.p, 0, sizeof(StageCounts), s));
    Entries en{};
    en.key = c->b_x_key.as<uint64_t>(); en.region = c->b_x_region.as<int32_t>(); en.order = c->b_x_order.as<uint32_t>();
    en.meta = c->b_x_meta.as<uint32_t>(); en.isize = c->b_x_isize.as<int32_t>();
    int rc = do_join_local(c, n32, en, c->b_x_n.as<uint32_t>(), false);
    if (rc != BDX_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_counts.p, c->b_counts.p, sizeof(StageCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    const StageCounts sc = *c->h_counts.as<StageCounts>();
    if (sc.irregular) return fail(c, BDX_ELIMIT, "a read name occurs more than twice: staged runs cannot replay it (run the chromosomes in one context)");
    if (sc.overflow) return fail(c, BDX_EINTERNAL, "group list overflow");
    if (n_groups) *n_groups = sc.n_groups;
    if (n_pairs) *n_pairs = sc.n_pairs;
    const size_t ng = std::min<size_t>(cap, sc.n_groups);
    static_assert(sizeof(bdx_group) == sizeof(GroupRec), "group record layout");
    if (out && ng) memcpy(out, c->k4.g_rec, ng * sizeof(GroupRec));
    return BDX_OK;
}

int bdx_stage_walk(bdx_ctx* c, size_t nregions, const bdx_region_rec* regions, const uint32_t* pk, size_t ngroups,
                   const bdx_group* groups, int32_t last_maxq, int any_anomalous) {
    if (!c || (nregions && (!regions || !pk)) || (ngroups && !groups)) return BDX_EINVAL;
    if (c->stage < 2) return BDX_ESTATE;
    NOT_WHILE_SIZING(c);
    decode_regions(c, (const RegionRec*)regions, pk, (uint32_t)nregions, 0, false);
    decode_groups(c, (const GroupRec*)groups, (uint32_t)ngroups, 0);
    c->counts.n_regions = (uint32_t)nregions;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = host_walk(c, last_maxq, any_anomalous != 0);
    if (rc != BDX_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = score_host_terms(c);
    if (rc == BDX_OK) rc = finish_host_walk(c);
    c->stage_ms[5] = ms_between(t0, t1);
    c->stage_ms[6] = ms_between(t1, std::chrono::steady_clock::now());
    return rc;
}

int bdx_get_summary(const bdx_ctx* c, bdx_summary* o) {
    if (!c || !o) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    o->n_reads = c->n; o->n_anomalous = c->p1.n_anom; o->covered_ref_len = c->g_covered; o->window = c->g_window;
    o->n_candidates = c->counts.n_cand; o->n_regions = (uint32_t)c->nreg; o->n_pairs = c->counts.n_pairs;
    o->n_groups = c->n_groups_total; o->n_svs = c->n_sv_total; o->n_svs_printed = c->n_printed;
    return BDX_OK;
}

int bdx_get_counters(const bdx_ctx* c, uint32_t* lib_read_count, uint32_t* bam_read_count, uint32_t* flag_hist, float* seqcov,
                     float* density) {
    if (!c) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    const uint32_t* hist = c->cnt.data();
    const uint32_t* lib_cnt = hist + c->nlibs * kNumFlags;
    const uint32_t* bam_cnt = lib_cnt + c->nlibs;
    if (flag_hist) memcpy(flag_hist, hist, (size_t)c->nlibs * kNumFlags * 4);
    if (lib_read_count) memcpy(lib_read_count, lib_cnt, (size_t)c->nlibs * 4);
    if (bam_read_count) memcpy(bam_read_count, bam_cnt, (size_t)c->nbams * 4);
    if (seqcov) memcpy(seqcov, c->seqcov.data(), (size_t)c->nlibs * 4);
    if (density) memcpy(density, c->lib_density.data(), (size_t)c->nlibs * 4);
    return BDX_OK;
}

int bdx_get_regions(const bdx_ctx* c, bdx_region* out, size_t cap) {
    if (!c || (!out && cap)) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    const size_t n = std::min(cap, c->nreg);
    for (size_t i = 0; i < n; ++i) {
        const HostRegion& r = c->reg[i];
        const int valid = c->opts.chr_restricted ? (int)r.nonctx : (int)r.n;
        out[i] = bdx_region{r.tid, r.start, r.end, (int32_t)r.nnormal, (int32_t)(r.n - r.rev), (int32_t)r.rev, (int32_t)r.n,
                            valid >= c->opts.min_read_pair ? 1 : 0, r.maxq};
    }
    return BDX_OK;
}

int bdx_get_svs(const bdx_ctx* c, bdx_sv* out, size_t cap) {
    if (!c || (!out && cap)) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    materialize(const_cast<bdx_ctx*>(c));
    const size_t n = std::min(cap, c->walk.svs.size());
    for (size_t i = 0; i < n; ++i) out[i] = c->walk.svs[i].sv;
    return BDX_OK;
}

int bdx_trim_results(bdx_ctx* c) {
    if (!c) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    NOT_WHILE_SIZING(c);
    HIPCHK(c, hipSetDevice(c->device));
    materialize(c);
    own_borrowed_regions(c);   // (a region table read where the device left it: the getters go on from a copy)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (PinBuf* b : {&c->h_regs, &c->h_pk, &c->h_groups, &c->h_sv_out, &c->h_lib_index, &c->h_lib_pairs, &c->h_cn_key, &c->h_cn_value, &c->h_ltail_dev}) b->release();
    return BDX_OK;
}

int bdx_get_sv_lists(const bdx_ctx* c, int32_t* lib_index, int32_t* lib_pairs, size_t lib_cap, int32_t* cn_key, float* cn_value,
                     size_t cn_cap) {
    if (!c) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    materialize(const_cast<bdx_ctx*>(c));
    const size_t nl = std::min(lib_cap, c->walk.lib_index.size());
    if (lib_index) memcpy(lib_index, c->walk.lib_index.data(), nl * 4);
    if (lib_pairs) memcpy(lib_pairs, c->walk.lib_pairs.data(), nl * 4);
    const size_t nc = std::min(cn_cap, c->walk.cn_key.size());
    if (cn_key) memcpy(cn_key, c->walk.cn_key.data(), nc * 4);
    if (cn_value) memcpy(cn_value, c->walk.cn_value.data(), nc * 4);
    return BDX_OK;
}

int bdx_set_collect_support(bdx_ctx* c, int on) {
    if (!c) return BDX_EINVAL;
    c->collect_support = on != 0;
    return BDX_OK;
}

int bdx_get_sv_support(const bdx_ctx* c, uint32_t* sv_offsets, uint64_t* read_index, uint8_t* read_flag, size_t cap, size_t* n_total) {
    if (!c) return BDX_EINVAL;
    if (!c->ran || !c->collect_support || c->sup_off.size() != c->walk.svs.size() + 1) return BDX_ESTATE;
    if (n_total) *n_total = c->sup_idx.size();
    if (sv_offsets) memcpy(sv_offsets, c->sup_off.data(), c->sup_off.size() * 4);
    const size_t n = std::min(cap, c->sup_idx.size());
    if (read_index && n) memcpy(read_index, c->sup_idx.data(), n * 8);
    if (read_flag && n) memcpy(read_flag, c->sup_flag.data(), n);
    return BDX_OK;
}

int bdx_get_read_class(const bdx_ctx* c, uint8_t* out, size_t cap) {
    if (!c || !out) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    const size_t n = std::min(cap, c->n);
    if (n && hipMemcpy(out, c->b_cls.p, n, hipMemcpyDeviceToHost) != hipSuccess) return BDX_EHIP;
    return BDX_OK;
}

int bdx_set_stage_timing(bdx_ctx* c, int on) {
    if (!c) return BDX_EINVAL;
    c->stage_timing = on != 0;
    return BDX_OK;
}

int bdx_set_enqueue_ahead(bdx_ctx* c, int on) {
    if (!c) return BDX_EINVAL;
    c->speculate = on < 0 ? 0 : (on > 2 ? 2 : on);
    return BDX_OK;
}

int bdx_run_many(bdx_ctx* const* ctxs, size_t n, int in_flight) {
    if (!ctxs && n) return BDX_EINVAL;
    for (size_t i = 0; i < n; ++i) {
        if (!ctxs[i]) return BDX_EINVAL;
        for (size_t k = 0; k < i; ++k)
            if (ctxs[k] == ctxs[i]) return BDX_EINVAL;   // (one context cannot run twice at a time)
    }
    const size_t workers = std::min<size_t>(n, (size_t)std::max(1, in_flight));
    if (workers <= 1) {
        for (size_t i = 0; i < n; ++i) {
            const int rc = bdx_run(ctxs[i]);
            if (rc != BDX_OK) return rc;
        }
        return BDX_OK;
    }
    // one host thread per context in flight: a run's waits (pass-1 statistics, the exact sizes of the later stages, the table's
    // arrival) are the thread's own, and the kernels of the contexts overlap on the GPU -- each context has its own streams
    std::atomic<size_t> next{0};
    std::atomic<int> first_error{BDX_OK};
    std::vector<std::thread> pool;
    pool.reserve(workers);
    for (size_t w = 0; w < workers; ++w)
        pool.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n || first_error.load() != BDX_OK) return;
                const int rc = bdx_run(ctxs[i]);
                if (rc != BDX_OK) {
                    int ok = BDX_OK;
                    first_error.compare_exchange_strong(ok, rc);
                }
            }
        });
    for (auto& t : pool) t.join();
    return first_error.load();
}

int bdx_use_name_check(bdx_ctx* c, int on) {
    if (!c) return BDX_EINVAL;
    if ((on != 0) == c->use_check) return BDX_OK;
    if (c->n) return fail(c, BDX_ESTATE, "the second name hash is switched while the context holds no reads");
    HIPCHK(c, hipSetDevice(c->device));
    c->use_check = on != 0;
    if (c->use_check && c->cap && !c->adopted) {
        HIPCHK(c, c->b_check.ensure(c->cap * 8));
        c->d.check = c->b_check.as<uint64_t>();
    }
    if (!c->use_check) c->d.check = nullptr;
    return BDX_OK;
}

int bdx_set_debug(bdx_ctx* c, const char* name, int value) {
    if (!c || !name) return BDX_EINVAL;
    struct { const char* n; int* p; } ints[] = {{"no_stash", &c->dbg_no_stash}, {"max_chunks", &c->dbg_max_chunks}, {"finalize2_fold", &c->dbg_finalize2_fold},
                                                 {"no_forward", &c->dbg_no_forward}, {"scan3", &c->dbg_scan3}, {"label_rounds", &c->dbg_label_rounds},
                                                 {"k1_grid", &c->dbg_k1_grid}, {"end_write_value", &c->dbg_end_write_value}, {"spec_test", &c->spec_test},
                                                 {"walk_lanes", &c->dbg_walk_lanes}, {"ins_plain", &c->dbg_ins_plain}, {"gather_walk", &c->dbg_gather_walk}, {"region_dma", &c->dbg_region_dma}, {"join_fwd", &c->dbg_join_fwd}, {"regions_copy", &c->dbg_regions_copy}, {"asm_plain", &c->dbg_asm_plain},
                                                 {"big_walk", &c->big_walk_mode}};
    for (auto& e : ints)
        if (!strcmp(name, e.n)) { *e.p = value; return BDX_OK; }
    if (!strcmp(name, "bucketed_join")) { c->bucketed_join = value != 0; return BDX_OK; }
    if (!strcmp(name, "no_poll")) { c->poll = value == 0; return BDX_OK; }
    if (!strcmp(name, "k1_event_period")) { c->k1_event_period = (uint32_t)std::max(1, value); return BDX_OK; }
    if (!strcmp(name, "pin_noncoherent")) {   // (before the first run: the result tables' pinned buffers are allocated non-coherent)
        for (PinBuf* b : {&c->h_sv_out, &c->h_lib_index, &c->h_lib_pairs, &c->h_cn_key, &c->h_cn_value, &c->h_ltail_dev, &c->h_regs, &c->h_pk})
            b->flags = value ? hipHostMallocNonCoherent : hipHostMallocDefault;
        return BDX_OK;
    }
    return fail(c, BDX_EINVAL, std::string("unknown debug switch ") + name);
}

int bdx_set_host_walk(bdx_ctx* c, int on) {
    if (!c) return BDX_EINVAL;
    c->host_walk_only = on != 0;
    return BDX_OK;
}

int bdx_get_walk_split(const bdx_ctx* c, uint32_t* n_sv_device, uint32_t* n_sv_host, uint32_t* n_groups_host) {
    if (!c) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    if (n_sv_device) *n_sv_device = c->counts.n_sv_dev;
    if (n_sv_host) *n_sv_host = c->n_sv_host;
    if (n_groups_host) *n_groups_host = c->counts.n_groups;
    return BDX_OK;
}

int bdx_get_cross_window_svs(const bdx_ctx* c, uint32_t* n_sv_device) {
    if (!c || !n_sv_device) return BDX_EINVAL;
    if (!c->ran) return BDX_ESTATE;
    *n_sv_device = c->counts.n_old;
    return BDX_OK;
}

int bdx_set_pass1_statistics(bdx_ctx* c, const uint32_t* counters, uint32_t covered_ref_len) {
    if (!c) return BDX_EINVAL;
    if (!counters) { c->ov_cnt.clear(); return BDX_OK; }
    c->ov_cnt.assign(counters, counters + (size_t)c->nlibs * kNumFlags + c->nlibs + c->nbams);
    c->ov_covered = covered_ref_len;
    c->ran = false;
    return BDX_OK;
}

int bdx_was_replayed(const bdx_ctx* c) { return c && c->ran && c->replayed ? 1 : 0; }

int bdx_get_timings(const bdx_ctx* c, float* ms, int cap) {
    if (!c || !ms) return 0;
    const int n = std::min(cap, kNumStages);
    for (int i = 0; i < n; ++i) ms[i] = c->stage_ms[i];
    return n;
}

int bdx_classify(const bdx_opts* opts, const bdx_lib* libs, int nlibs, const bdx_batch* b, uint8_t* cls_out, int device) {
    if (!opts || !libs || !b || !cls_out) return BDX_EINVAL;
    int nbams = 1;
    for (int i = 0; i < nlibs; ++i) nbams = std::max(nbams, libs[i].bam_index + 1);
    for (size_t i = 0; i < b->n; ++i) nbams = std::max(nbams, (int)b->bam[i] + 1);
    bdx_ctx* c = nullptr;
    int rc = bdx_create(&c, opts, libs, nlibs, nbams, 0, 100000000, device);
    if (rc != BDX_OK) return rc;
    rc = bdx_push(c, b);
    if (rc == BDX_OK) rc = bdx_run(c);
    if (rc == BDX_OK) rc = bdx_get_read_class(c, cls_out, b->n);
    bdx_destroy(c);
    return rc;
}

int bdx_set_process_option(const char* name, int value) {
    if (!name) return BDX_EINVAL;
    if (!strcmp(name, "pin_malloc")) { PinBuf::registered_switch().store(value == 0); return BDX_OK; }
    return BDX_EINVAL;
}

int bdx_warm_up(int device) {
    if (hipSetDevice(device) != hipSuccess) return BDX_EHIP;
    warm_k1(nullptr); warm_k2(nullptr); warm_k3(nullptr); warm_k4(nullptr); warm_k5(nullptr); warm_k6(nullptr); warm_k7(nullptr); warm_k9(nullptr);
    return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? BDX_OK : BDX_EHIP;
}

int bdx_poisson_log_upper_tail(const double* lambda, const int32_t* k, double* out, size_t n, int device) {
    if (!lambda || !k || !out) return BDX_EINVAL;
    if (n == 0) return BDX_OK;
    if (hipSetDevice(device) != hipSuccess) return BDX_EHIP;
    DevBuf bl, bk, bo;
    if (bl.ensure(n * 8) != hipSuccess || bk.ensure(n * 4) != hipSuccess || bo.ensure(n * 8) != hipSuccess) return BDX_ENOMEM;
    int rc = BDX_OK;
    if (hipMemcpy(bl.p, lambda, n * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(bk.p, k, n * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = BDX_EHIP;
    if (rc == BDX_OK) {
        launch_k5(bl.as<double>(), bk.as<int32_t>(), bo.as<double>(), (uint32_t)n, nullptr);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = BDX_EHIP;
    }
    if (rc == BDX_OK && hipMemcpy(out, bo.p, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = BDX_EHIP;
    bl.release(); bk.release(); bo.release();
    return rc;
}

}  // extern "C"

#include "bdx_dist_impl.h"
#include "bdx_bamdec_impl.h"
