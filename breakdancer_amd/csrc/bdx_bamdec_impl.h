// bdx_bamdec_*: a BAM file decoded on the GPU (include/bdx.h).  Included at the end of bdx_api.hip: in sink mode the decoded
// records go straight into a context's resident store and the classifier follows them, like the batches of the staging ring.
//
// The caller hands over the file as it is on disk, in pieces of whole BGZF members copied into pinned staging buffers
// (bdx_bamdec_acquire / bdx_bamdec_submit) together with the members' table (payload offset and length, inflated length: the
// 18-byte headers and 8-byte footers are the only bytes the host looks at).  A piece is the unit of TRANSFER (a staging buffer
// is free again as soon as its copy is through); the unit of WORK is a batch of pieces, launched once it holds enough members
// to fill the GPU: one member is one wave and takes ~9 ms alone, so a launch of a few hundred members costs what one of
// 8,000 does (measured: 61 launches of 755 members each, 8.6 ms apiece, back to back).  Per batch, on three streams:
//   copy     compressed bytes (per piece) + the batch's block table -> HBM
//   inflate  KZ, one wave per member, into a ring of inflated bytes
//   records  one batch behind (a record may end in the next batch): KB chain -> stitch -> fields -> ordered compaction into the
//            destination columns; the running state (next record boundary, counts, errors) lives in device memory, so the
//            host never waits for a batch -- it reads a progress record in pinned memory when it wants to know
// The ring holds a few batches; a batch that does not fit behind its predecessor starts at the ring's front again, and the
// front of it is mirrored behind the predecessor so that the record that straddles the two stays contiguous.
//
// Rules the feeding thread's path keeps (each one was a stall of an inflate launch's length when it was broken):
//   * nothing is freed or grown while batches are in flight -- hipFree waits for the device -- so a slot's tables, the record
//     stage's scratch, the raw columns and the destination are sized once, for a full batch (and the batches a store must take
//     before their record counts are known);
//   * with a sink the copy and record streams are the sink's own: four streams in all, one per hardware queue of the runtime
//     (two streams on one queue run in order: a copy's completion marker would sit behind an inflate launch);
//   * a wait on a stream comes before that stream is given its waits for the inflate launches;
//   * the first two batches are smaller (the GPU starts after a few milliseconds of reading), staging buffers are pinned by
//     threads of their own while the first piece is read.
#include <deque>

namespace {

constexpr size_t kBamMargin = (size_t)kMaxDeviceRecord + 65536;   // bytes mirrored behind the piece in front of a wrap
constexpr int kBamSlots = 4;                                       // batches in flight (device-side compressed bytes, tables, status)
#ifndef BDX_BAM_STAGING
#define BDX_BAM_STAGING 12
#endif
constexpr int kBamStaging = BDX_BAM_STAGING;   // pinned staging buffers.  A caller holds several (it reads pieces ahead of the one it submits) and the REST are what the copy engine
                                               // has queued: with six, four of them being read, two copies in flight did not keep it busy -- a piece every 0.26 ms for copies of
                                               // 0.16 ms, 32 GB/s (profiles/r05_genome_probe_mid.txt); pinned one after the other by ONE thread, in the order they are asked for
constexpr size_t kBatchBytesDefault = (size_t)384 << 20;           // compressed bytes per batch
constexpr size_t kBatchBlocksDefault = 7680;                       // members per batch: with the piece that takes it over this, still within the wave slots the GPU has for this kernel
                                                                   // (256 CUs x 32 = 8,192).  The kernel's waves give way as they get ahead (s_setprio), so a launch's members end together:
                                                                   // members beyond the slots would start when all the others end -- 13.6 ms per launch at 8,448 against 9.0 at 7,680 -- 7,936 left no room for the record stages that run beside it: some launches at 13 ms again

struct BamPiece {
    uint64_t seq = 0;            // 1-based
    uint64_t ring_beg = 0, ring_end = 0;
    uint64_t mirror_end = 0;     // = ring_end, or behind it where the front of a wrapped successor is mirrored
    uint64_t prev_end = 0;       // a wrapped piece: where its predecessor ends (the carried record boundary is relative to that)
    uint32_t nblk = 0;
    bool wrapped = false;        // starts at the ring's front although its predecessor does not end at the ring's end
    bool records_done = false;   // its record stage has been enqueued
    int slot = 0;
    hipEvent_t ev_inflated = nullptr, ev_records = nullptr;
};

}  // namespace

struct bdx_bamdec {
    int device = 0;
    bdx_ctx* sink = nullptr;
    std::string err;
    hipStream_t s_copy = nullptr, s_inf = nullptr, s_inf2 = nullptr, s_rec = nullptr;   // (s_inf == s_inf2 == s_rec unless stream_mode != 0: bdx_bamdec_create)
    // A decoder that feeds a context copies on the context's copy stream; with the context's compute and side streams and the decoder's
    // own that makes four streams in all.  The HIP runtime spreads a process's streams over four hardware queues, and two streams on one
    // queue run in order -- with six, the completion of a 0.3 ms copy waited behind a 19 ms inflate launch.
    bool borrowed_copy = false, borrowed_rec = false;   // (a stream of the sink's: not the decoder's to destroy)
    bool own_inf_stream = false;      // (stream_mode 1 / 2) the inflate launches have a stream of their own; else they run in s_rec
    // pinned staging: one piece's compressed bytes and the caller's member table
    struct Staging {
        PinBuf h_comp, h_tab;
        hipEvent_t ev_copied = nullptr;   // H2D of this buffer done
        bool busy = false;
        size_t cap = 0;                   // table entries
        std::atomic<int> pinned{1};       // 0: the pinning thread has not got to this buffer yet (bdx_bamdec_params::piece_bytes), 1: ready
        size_t cap_bytes = 0;             // bytes the last bdx_bamdec_acquire promised
        hipError_t pin_status = hipSuccess;
    } staging[kBamStaging];
    std::thread pin_thread;              // (pins the staging buffers one after the other: page pinning does not run in parallel with itself -- and then
                                         // allocates the batch slots' and the record stage's buffers, which the feeding thread would otherwise size as it
                                         // first meets them, in the middle of the first batches: profiles/r05_cli_timeline.txt, the gaps before 150 ms)
    std::atomic<int> slot_ready[kBamSlots];   // 0: the pin thread has not got to this slot's buffers yet, 1: the feeding thread may look at them
    std::atomic<int> rec_ready{1};            // (the same for the record stage's scratch and raw columns)
    int next_staging = 0, held_staging = 0;   // the held_staging buffers before next_staging are acquired and not submitted yet (oldest first)
    // a batch: the compressed bytes of its pieces back to back in HBM, its member table, the inflate status words
    struct Slot {
        DevBuf d_comp, d_blocks, d_status;
        PinBuf h_blocks;                  // the device-format table, built as the pieces arrive
        hipEvent_t ev_copied = nullptr;   // all of the batch's bytes and its table are in HBM
        hipEvent_t ev_free = nullptr;     // the batch's record stage is through (its device buffers are reusable)
        bool busy = false, open = false;
        size_t bytes = 0, nblk = 0, cap_blk = 0;
        uint64_t ulen = 0;
    } slot[kBamSlots];
    int cur_slot = 0;
    size_t batch_bytes = kBatchBytesDefault, batch_blocks = kBatchBlocksDefault;   // the LARGEST batch
    size_t round_blocks = kBatchBlocksDefault;   // one round of the wave slots: the batches grow from a third of it to batch_blocks (bdx_bamdec_submit)
    DevBuf d_ring;
    size_t ring_bytes = 0;        // usable bytes (the allocation has kBamMargin more)
    uint64_t cursor = 0;
    std::deque<BamPiece> pieces;  // submitted, oldest first; dropped once their record stage is enqueued and a successor exists
    std::vector<hipEvent_t> ev_pool;
    uint64_t n_pieces = 0;
    // where the feeding thread's time goes (bdx_bamdec_host_ms): [0] waiting for a staging buffer's copy, [1] pinning staging memory,
    // [2] waiting for a batch slot, [3] a slot's buffers, [4] the piece's copy calls, [5] a batch's launch, [6] a record stage's
    // launches, [7] feeding the classifier
    // [8] first inflate launch, [9] bdx_bamdec_finish's return: ms after the decoder's creation (or its last re-arming)
    double host_ms[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // ([10] of [7]: sizing the later stages' buffers, [11] of [7]: classifier launches;
                                                                       //  [12] / [13]: the inflate kernel's own time by HIP events / its launches, bdx_bamdec_params::time_kernels)
    bool time_kernels = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kz_events;   // (time_kernels) an event pair around every inflate launch, read by bdx_bamdec_finish
    std::chrono::steady_clock::time_point t_armed = std::chrono::steady_clock::now();
    bool finished = false, any_submitted = false;
    // record stage scratch (one piece at a time on s_rec)
    DevBuf d_cb, d_offs, d_base, d_scan, d_state;
    size_t rec_cap_blk = 0;             // members d_cb / d_offs / d_base are sized for
    DevBuf r_tid, r_pos, r_mtid, r_mpos, r_isize, r_flag, r_qlen, r_mapq, r_lib, r_keep, r_key, r_check;
    uint32_t raw_cap = 0;
    // read groups
    DevBuf d_rg_hash, d_rg_off, d_rg_chars, d_rg_lib;
    RgTable rg{};
    RecordFilterDev filt{};
    uint8_t bam_index = 0;
    // own destination (no sink)
    DevBuf o_tid, o_pos, o_mtid, o_mpos, o_isize, o_flag, o_qlen, o_mapq, o_lib, o_bam, o_key, o_check;
    size_t own_cap = 0;
    // progress record in pinned memory: [0] kept records, [1] error | past_region << 8 | redo << 32, [2] raw records, [3] sequence
    PinBuf h_progress;
    uint64_t confirmed = 0;       // kept records known to be in the destination
    uint64_t confirmed_seq = 0;
    uint64_t bound_in_flight = 0; // upper bound of the records of pieces whose compaction is not confirmed yet
    std::deque<std::pair<uint64_t, uint64_t>> bounds;  // (sequence, bound)
    std::deque<std::pair<uint64_t, hipEvent_t>> rec_events;  // (sequence, records-done event) for the sink's classifier
    float ms_inflate = 0;
    uint64_t inflated_bytes = 0, compressed_bytes = 0;
    size_t expected_bytes = 0;    // compressed bytes the caller announced (0: unknown)
    uint64_t first_batch_bytes = 0;
    bool presized = false;        // sink mode: the later stages' buffers have been sized from the first batch's record density
    std::thread presize_thread;   // (the sizing runs beside the decode; joined by bdx_bamdec_finish)
    int presize_rc = 0;
    std::deque<std::pair<uint64_t, uint64_t>> batch_end_bytes;   // (batch sequence, compressed bytes submitted up to its end)
    uint64_t records_at_arm = 0;  // what the sink held when the decoder was created / armed again
    uint64_t bytes_at_arm = 0;    // compressed bytes submitted before that
};

namespace {

int bfail(bdx_bamdec* d, int code, const std::string& msg) {
    if (d) d->err = msg;
    return code;
}
#define BHIP(d, expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return bfail(d, BDX_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

hipEvent_t bam_event(bdx_bamdec* d) {
    if (!d->ev_pool.empty()) { hipEvent_t e = d->ev_pool.back(); d->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

struct BamTimer {
    double& acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit BamTimer(double& a) : acc(a) {}
    ~BamTimer() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

DstColumns bam_dst(bdx_bamdec* d, uint64_t* cap) {
    DstColumns c{};
    if (d->sink) {
        bdx_ctx* k = d->sink;
        c.tid = (int32_t*)k->d.tid; c.pos = (int32_t*)k->d.pos; c.mtid = (int32_t*)k->d.mtid; c.mpos = (int32_t*)k->d.mpos;
        c.isize = (int32_t*)k->d.isize; c.flag = (uint16_t*)k->d.flag; c.qlen = (uint16_t*)k->d.qlen; c.mapq = (uint8_t*)k->d.mapq;
        c.lib = (uint8_t*)k->d.lib; c.bam = (uint8_t*)k->d.bam; c.key = (uint64_t*)k->d.key;
        c.check = (uint64_t*)k->d.check;   // (null unless bdx_use_name_check)
        *cap = k->cap;
    } else {
        c.tid = d->o_tid.as<int32_t>(); c.pos = d->o_pos.as<int32_t>(); c.mtid = d->o_mtid.as<int32_t>(); c.mpos = d->o_mpos.as<int32_t>();
        c.isize = d->o_isize.as<int32_t>(); c.flag = d->o_flag.as<uint16_t>(); c.qlen = d->o_qlen.as<uint16_t>(); c.mapq = d->o_mapq.as<uint8_t>();
        c.lib = d->o_lib.as<uint8_t>(); c.bam = d->o_bam.as<uint8_t>(); c.key = d->o_key.as<uint64_t>();
        c.check = d->o_check.as<uint64_t>();
        *cap = d->own_cap;
    }
    return c;
}

// the decoder's own destination columns with room for `cap` records (contents kept)
int bam_own_reserve(bdx_bamdec* d, size_t cap, uint64_t keep_records) {
    if (cap <= d->own_cap) return BDX_OK;
    cap = round_up(cap, 1024);
    struct Col { DevBuf* b; size_t esz; };
    Col cols[] = {{&d->o_tid, 4}, {&d->o_pos, 4}, {&d->o_mtid, 4}, {&d->o_mpos, 4}, {&d->o_isize, 4}, {&d->o_flag, 2}, {&d->o_qlen, 2},
                  {&d->o_mapq, 1}, {&d->o_lib, 1}, {&d->o_bam, 1}, {&d->o_key, 8}, {&d->o_check, 8}};
    if (d->own_cap) BHIP(d, hipStreamSynchronize(d->s_rec));
    for (Col& c : cols) {
        DevBuf nb;
        BHIP(d, nb.ensure(cap * c.esz));
        if (keep_records && c.b->p) BHIP(d, hipMemcpy(nb.p, c.b->p, keep_records * c.esz, hipMemcpyDeviceToDevice));
        c.b->release();
        *c.b = nb;
    }
    d->own_cap = cap;
    return BDX_OK;
}

// what the host knows of the device's progress (never blocks)
void bam_poll(bdx_bamdec* d) {
    if (!d->h_progress.p) return;
    volatile uint64_t* pr = (volatile uint64_t*)d->h_progress.p;
    const uint64_t seq = pr[3];
    if (seq <= d->confirmed_seq) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    d->confirmed = pr[0];
    d->confirmed_seq = seq;
    while (!d->bounds.empty() && d->bounds.front().first <= seq) { d->bound_in_flight -= d->bounds.front().second; d->bounds.pop_front(); }
}

// sink mode: the classifier over the tiles that the confirmed records complete
int bam_feed_classifier(bdx_bamdec* d, bool final) {
    bdx_ctx* c = d->sink;
    if (!c) return BDX_OK;
    BamTimer t7(d->host_ms[7]);
    hipEvent_t latest = nullptr;
    while (!d->rec_events.empty() && d->rec_events.front().first <= d->confirmed_seq) {
        if (latest) d->ev_pool.push_back(latest);
        latest = d->rec_events.front().second;
        d->rec_events.pop_front();
    }
    if (d->confirmed > c->n) {
        c->n = (size_t)d->confirmed;
        c->ran = false;
    }
    // The stages behind pass 1 get their buffers (~60 allocations; with a genome's share of records 0.25-0.35 s of page pinning for the
    // result tables) on a thread of their own, once the first four batches' records are counted (not earlier: its allocations would
    // contend with the first batches' own): their records per byte applied to the
    // bytes the caller announced (or, without an announcement, to what has been submitted once the last piece is in).  Until round 5 the
    // feeding thread did this itself behind its last piece -- the GPU was through with the file long before (profiles/r05_genome_probe_before.txt).
    if (!d->presized && (d->confirmed_seq >= 4 || (d->finished && d->confirmed_seq >= 1)) && (d->finished || d->expected_bytes) && !d->batch_end_bytes.empty()) {
        uint64_t bytes_counted = 0;
        for (auto const& be : d->batch_end_bytes)
            if (be.first <= d->confirmed_seq) bytes_counted = be.second;
        const uint64_t total = d->finished ? d->compressed_bytes - d->bytes_at_arm : std::max<uint64_t>(d->expected_bytes, d->compressed_bytes - d->bytes_at_arm);
        const uint64_t counted = d->confirmed > d->records_at_arm ? d->confirmed - d->records_at_arm : 0;
        if (bytes_counted && counted) {
            d->presized = true;
            const double est = (double)counted / (double)bytes_counted * (double)total * 1.10 + (double)d->records_at_arm;
            if (est >= (double)(1u << 20) && !c->ran) {
                const uint64_t prior = (uint64_t)est / 32 + 4096;
                if (prior <= kMaxAnomalous) {
                    const int device = d->device;
                    d->presize_thread = std::thread([d, c, prior, device] {
                        const auto t0 = std::chrono::steady_clock::now();
                        int rc = hipSetDevice(device) == hipSuccess ? BDX_OK : BDX_EHIP;
                        if (rc == BDX_OK) rc = presize_stages(c, (uint32_t)prior);
                        d->presize_rc = rc;
                        d->host_ms[10] = ms_between(t0, std::chrono::steady_clock::now());
                    });
                }
            }
        }
    }
    if (latest) {
        const hipError_t e = hipStreamWaitEvent(c->stream, latest, 0);
        d->ev_pool.push_back(latest);
        if (e != hipSuccess) return bfail(d, BDX_EHIP, "hipStreamWaitEvent");
    }
    if (c->k1_live) {
        const uint32_t full = (uint32_t)(c->n / kTile);
        if (full >= c->k1_done + kStreamTilesMin || (final && full > c->k1_done)) {
            BamTimer t11(d->host_ms[11]);
            const int rc = pass1_classify(c, full, false);
            if (rc != BDX_OK) return bfail(d, rc, c->err);
        }
    }
    return BDX_OK;
}

// is_last: 0 more pieces follow, 1 the file ends here, 2 the caller stops here on purpose (a region read through the index: the
// record that runs past the cut is dropped, not an error)
int bam_record_stage(bdx_bamdec* d, BamPiece& p, const BamPiece* next, int is_last) {
    BamTimer t6(d->host_ms[6]);
    while (!d->rec_ready.load(std::memory_order_acquire)) std::this_thread::yield();   // (the pin thread is sizing the stage's buffers)
    hipStream_t s = d->s_rec;
    bdx_bamdec::Slot& sl = d->slot[p.slot];
    // (sizes first: a wait on the stream, should one be needed, must not find this stage's own waits for the inflate launches in it)
    const uint32_t nblk = p.nblk;
    // upper bound of the piece's records (36 bytes is the smallest record) -> the raw columns
    const uint64_t bound = (p.ring_end - p.ring_beg) / 36 + 2;
    if (bound > d->raw_cap) {
        // (batches differ by a piece's worth of members: half as much again, so that no later batch comes back here -- the wait
        // below is for the record stage before this one, which waits for an inflate launch: the feeding thread would stand still)
        BHIP(d, hipStreamSynchronize(s));
        const size_t cap = round_up(std::max<size_t>((size_t)bound + bound / 2, (d->batch_blocks + d->batch_blocks / 4 + 64) * (size_t)65536 / 36), 1024);
        BHIP(d, d->r_tid.ensure(cap * 4)); BHIP(d, d->r_pos.ensure(cap * 4)); BHIP(d, d->r_mtid.ensure(cap * 4)); BHIP(d, d->r_mpos.ensure(cap * 4));
        BHIP(d, d->r_isize.ensure(cap * 4)); BHIP(d, d->r_flag.ensure(cap * 2)); BHIP(d, d->r_qlen.ensure(cap * 2)); BHIP(d, d->r_mapq.ensure(cap));
        BHIP(d, d->r_lib.ensure(cap)); BHIP(d, d->r_keep.ensure(cap)); BHIP(d, d->r_key.ensure(cap * 8)); BHIP(d, d->r_check.ensure(cap * 8));
        BHIP(d, d->d_scan.ensure((cap / 256 + 8) * 4));
        d->raw_cap = (uint32_t)cap;
    }
    if (nblk > d->rec_cap_blk || !d->d_base.p) {   // (growing them frees them first, which waits for the device: sized for a full batch at once;
                                                    // a batch without members still has its record count written)
        const size_t nb = std::max<size_t>((size_t)nblk + nblk / 2, d->batch_blocks + d->batch_blocks / 4 + 64);
        d->rec_cap_blk = nb;
        BHIP(d, hipStreamSynchronize(s));
        BHIP(d, d->d_cb.ensure(nb * sizeof(ChainBlock)));
        BHIP(d, d->d_offs.ensure(nb * kRecSlots * 2));
        BHIP(d, d->d_base.ensure((nb + 2) * 4));
    }
    // destination capacity: everything that may still arrive from pieces in flight must fit
    bam_poll(d);
    const uint64_t need = d->confirmed + d->bound_in_flight + bound;
    if (d->sink) {
        bdx_ctx* c = d->sink;
        if (need > c->cap) {
            BHIP(d, hipStreamSynchronize(s));   // pieces in flight write through the old columns
            BHIP(d, hipStreamSynchronize(c->stream));   // (and the classifier reads them)
            bam_poll(d);
            c->n = (size_t)d->confirmed;
            const int rc = alloc_reads(c, std::max<size_t>((size_t)need, c->cap + c->cap / 2));
            if (rc != BDX_OK) return bfail(d, rc, c->err);
        }
    } else {
        if (need > d->own_cap) {
            BHIP(d, hipStreamSynchronize(s));
            bam_poll(d);
            const int rc = bam_own_reserve(d, std::max<size_t>((size_t)need, d->own_cap + d->own_cap / 2), d->confirmed);
            if (rc != BDX_OK) return rc;
        }
    }
    uint64_t avail_end = p.ring_end;
    // (the batch's own inflate launch and its successor's run on two streams: neither implies the other)
    BHIP(d, hipStreamWaitEvent(s, p.ev_inflated, 0));
    if (next) {
        BHIP(d, hipStreamWaitEvent(s, next->ev_inflated, 0));
        if (next->wrapped) {   // mirror the front of the ring behind this piece: the straddling record stays contiguous
            const size_t n = std::min<uint64_t>(kBamMargin, next->ring_end - next->ring_beg);
            BHIP(d, hipMemcpyAsync((char*)d->d_ring.p + p.ring_end, (char*)d->d_ring.p + next->ring_beg, n, hipMemcpyDeviceToDevice, s));
            avail_end = p.ring_end + n;
        } else {
            avail_end = next->ring_end;
        }
    }
    const uint8_t* u = d->d_ring.as<uint8_t>();
    const BgzfBlock* blocks = sl.d_blocks.as<BgzfBlock>();
    ChainBlock* cb = d->d_cb.as<ChainBlock>();
    uint16_t* offs = d->d_offs.as<uint16_t>();
    uint32_t* base = d->d_base.as<uint32_t>();
    PieceState* st = d->d_state.as<PieceState>();
    launch_kb_chain(u, blocks, nblk, avail_end, d->filt.n_targets, cb, offs, s);
    launch_kb_stitch(u, blocks, nblk, avail_end, is_last, cb, offs, base, st, sl.d_status.as<uint32_t>(), p.wrapped ? p.prev_end : 0,
                     p.wrapped ? p.ring_beg : 0, s);
    RawColumns raw{d->r_tid.as<int32_t>(), d->r_pos.as<int32_t>(), d->r_mtid.as<int32_t>(), d->r_mpos.as<int32_t>(), d->r_isize.as<int32_t>(),
                   d->r_flag.as<uint16_t>(), d->r_qlen.as<uint16_t>(), d->r_mapq.as<uint8_t>(), d->r_lib.as<uint8_t>(), d->r_keep.as<uint8_t>(),
                   d->r_key.as<uint64_t>(), d->r_check.as<uint64_t>()};
    launch_kb_extract(u, blocks, nblk, cb, offs, base, d->rg, d->filt, raw, st, s);
    uint64_t dst_cap = 0;
    DstColumns dst = bam_dst(d, &dst_cap);
    launch_kb_compact(raw, (uint32_t)std::min<uint64_t>(bound, d->raw_cap), dst, dst_cap, d->bam_index, d->d_scan.as<uint32_t>(), st,
                      (volatile uint64_t*)d->h_progress.p, p.seq, s);
    p.ev_records = bam_event(d);
    if (!p.ev_records) return bfail(d, BDX_EHIP, "hipEventCreate");
    BHIP(d, hipEventRecord(p.ev_records, s));
    // the slot's device buffers (block table, status) are free once the record stage is through
    BHIP(d, hipEventRecord(sl.ev_free, s));
    d->bounds.emplace_back(p.seq, bound);
    d->bound_in_flight += bound;
    if (d->sink) {
        hipEvent_t e = bam_event(d);
        if (!e) return bfail(d, BDX_EHIP, "hipEventCreate");
        BHIP(d, hipEventRecord(e, s));
        d->rec_events.emplace_back(p.seq, e);
    }
    p.records_done = true;
    return BDX_OK;
}

}  // namespace

extern "C" {

int bdx_bamdec_create(bdx_bamdec** out, bdx_ctx* sink, const bdx_bamdec_params* p) {
    if (!out || !p || p->n_targets < 0 || p->bam_index < 0 || p->bam_index > 254) return BDX_EINVAL;
    if (p->n_read_groups && (!p->rg_ids || !p->rg_lib)) return BDX_EINVAL;
    const int device = sink ? sink->device : p->device;
    if (hipSetDevice(device) != hipSuccess) return BDX_EHIP;
    bdx_bamdec* d = new (std::nothrow) bdx_bamdec;
    if (!d) return BDX_ENOMEM;
    d->device = device;
    d->sink = sink;
    for (auto& r : d->slot_ready) r.store(1);
    d->bam_index = (uint8_t)p->bam_index;
    d->filt.only_tid = p->only_tid; d->filt.beg = p->region_beg; d->filt.end = p->region_end; d->filt.n_targets = p->n_targets;
    d->filt.keep_all = (p->record_mode & 1) ? 1 : 0; d->filt.mapq_only = (p->record_mode & 2) ? 1 : 0;
    auto bad = [&](int code) { bdx_bamdec_destroy(d); return code; };
    static const bool create_trace = getenv("BDX_BAMDEC_TRACE") != nullptr;   // (where a decoder's set-up time goes, on stderr)
    const auto t_c0 = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (create_trace) fprintf(stderr, "[bdx bamdec create] %s at %.3f ms\n", what, ms_between(t_c0, std::chrono::steady_clock::now()));
    };
    // Streams.  The copies go through the sink's copy stream (or one of the decoder's own); the inflate launches and the record stages
    // share ONE stream of the decoder's, in order: KZ(b + 1), records(b), KZ(b + 2), records(b + 1) ...
    // Beside an inflate launch a record stage crawls -- a launch takes every wave slot, the record kernels' workgroups wait for slots that
    // free up (kb_stitch 6 ms instead of 0.03) and the launch's own last workgroups start late behind them: 12.4 ms per launch of 7,680
    // members in the pipeline against 8.6 ms alone (profiles/r05_genome_timeline_before.txt) -- and nothing can be kept going beside a
    // kernel that outlives its batch either: on this stack a kernel submitted to an idle queue starts only when every running kernel has
    // ended (tools/latecomer_probe.hip, tools/chain_probe.hip), and a CU-masked queue holds 16 waves per CU (tools/cumask_probe.hip).
    // Taking turns costs the record stages' own time, ~1 ms per 7,680 members.  The stream is NOT the sink's compute stream: the runtime
    // spreads a process's streams over four hardware queues, two streams on one queue run in order, and a copy's completion marker behind
    // a 30 ms inflate launch kept the feeder waiting for its staging buffers; the classifier, on the sink's stream, follows the record
    // stages through their events.  bdx_bamdec_params::stream_mode 1: the inflate launches in a third stream, beside the record stages, as until round 4.
    if (sink && sink->copy_stream) {
        d->s_copy = sink->copy_stream;
        d->borrowed_copy = true;
    } else if (hipStreamCreateWithFlags(&d->s_copy, hipStreamNonBlocking) != hipSuccess) {
        return bad(BDX_EHIP);
    }
    {
        const bool prio = p->stream_mode == 2;   // (experiment: the three streams with queue priorities -- record stages high, inflate low)
        const bool third = p->stream_mode == 1 || prio;
        int pr_least = 0, pr_greatest = 0;
        if (prio && hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) != hipSuccess) return bad(BDX_EHIP);
        if (prio) {
            if (hipStreamCreateWithPriority(&d->s_rec, hipStreamNonBlocking, pr_greatest) != hipSuccess) return bad(BDX_EHIP);
        } else if (third && sink && sink->stream) {
            d->s_rec = sink->stream;
            d->borrowed_rec = true;
        } else if (hipStreamCreateWithFlags(&d->s_rec, hipStreamNonBlocking) != hipSuccess) {
            return bad(BDX_EHIP);
        }
        if (third) {
            if (prio ? hipStreamCreateWithPriority(&d->s_inf, hipStreamNonBlocking, pr_least) != hipSuccess
                     : hipStreamCreateWithFlags(&d->s_inf, hipStreamNonBlocking) != hipSuccess) return bad(BDX_EHIP);
            d->own_inf_stream = true;
        } else {
            d->s_inf = d->s_rec;
        }
    }
    d->s_inf2 = d->s_inf;
    for (auto& sl : d->slot)
        if (hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sl.ev_free, hipEventDisableTiming) != hipSuccess)
            return bad(BDX_EHIP);
    for (auto& st : d->staging)
        if (hipEventCreateWithFlags(&st.ev_copied, hipEventDisableTiming) != hipSuccess) return bad(BDX_EHIP);
    mark("streams and events");
    d->expected_bytes = p->expected_bytes;
    d->time_kernels = p->time_kernels != 0;
    // A launch of ONE round of the wave slots ends with the slots draining (58 GB/s of inflated bytes against 68 in launches of four rounds
    // and 70 with a 16 GB file in one launch, profiles/r05_genome_inflate_alone.txt), so a large input is decoded in batches of up to four
    // rounds -- 30,720 members, 2 GB inflated; the buffers grow with them, which a small file would pay for in its set-up: one round per
    // 2.5 GB announced.  A caller's batch_blocks is taken as it is; batch_rounds overrides the rounds.
    {
        size_t rounds = std::max<size_t>(1, std::min<size_t>(4, p->expected_bytes / ((size_t)2560 << 20)));
        if (p->batch_rounds > 0) rounds = (size_t)std::min(16, p->batch_rounds);
        if (p->batch_blocks) { d->batch_blocks = d->round_blocks = p->batch_blocks; rounds = 1; }
        else d->batch_blocks = d->round_blocks * rounds;
        d->batch_bytes = p->batch_bytes ? p->batch_bytes : kBatchBytesDefault * rounds;
    }
    // read groups
    {
        const uint32_t n = p->n_read_groups;
        std::vector<uint64_t> hash(n);
        std::vector<uint32_t> off(n + 1, 0);
        std::string chars;
        for (uint32_t i = 0; i < n; ++i) {
            const char* id = p->rg_ids[i] ? p->rg_ids[i] : "";
            const size_t l = strlen(id);
            uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)l;
            size_t k = 0;
            for (; k + 8 <= l; k += 8) { uint64_t w; memcpy(&w, id + k, 8); h = name_hash_step(h, w); }
            uint64_t w = 0;
            if (k < l) memcpy(&w, id + k, l - k);
            hash[i] = name_hash_finish(h, w);
            off[i] = (uint32_t)chars.size();
            chars.append(id, l);
        }
        off[n] = (uint32_t)chars.size();
        if (d->d_rg_hash.ensure(std::max<size_t>(n, 1) * 8) != hipSuccess || d->d_rg_off.ensure(((size_t)n + 1) * 4) != hipSuccess ||
            d->d_rg_chars.ensure(std::max<size_t>(chars.size(), 1)) != hipSuccess || d->d_rg_lib.ensure(std::max<size_t>(n, 1)) != hipSuccess)
            return bad(BDX_ENOMEM);
        if ((n && hipMemcpy(d->d_rg_hash.p, hash.data(), (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) ||
            hipMemcpy(d->d_rg_off.p, off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
            (!chars.empty() && hipMemcpy(d->d_rg_chars.p, chars.data(), chars.size(), hipMemcpyHostToDevice) != hipSuccess) ||
            (n && hipMemcpy(d->d_rg_lib.p, p->rg_lib, n, hipMemcpyHostToDevice) != hipSuccess))
            return bad(BDX_EHIP);
        d->rg.hash = d->d_rg_hash.as<uint64_t>(); d->rg.off = d->d_rg_off.as<uint32_t>(); d->rg.chars = d->d_rg_chars.as<char>();
        d->rg.lib = d->d_rg_lib.as<uint8_t>(); d->rg.n = n; d->rg.fallback = p->fallback_lib;
        d->rg.missing = p->missing_lib_plus1 > 0 ? (uint8_t)(p->missing_lib_plus1 - 1) : p->fallback_lib;
    }
    mark("read-group tables");
    // ring of inflated bytes
    d->ring_bytes = p->ring_bytes ? p->ring_bytes : ((size_t)3 << 30);
    if (d->ring_bytes < ((size_t)1 << 20)) d->ring_bytes = (size_t)1 << 20;
    if (d->d_ring.ensure(d->ring_bytes + kBamMargin + 64) != hipSuccess) return bad(BDX_ENOMEM);
    d->ring_bytes = d->d_ring.bytes - kBamMargin - 64;
    if (d->d_state.ensure(sizeof(PieceState)) != hipSuccess) return bad(BDX_ENOMEM);
    PieceState st{};
    st.next_start = p->first_record_offset;
    if (hipMemcpy(d->d_state.p, &st, sizeof(st), hipMemcpyHostToDevice) != hipSuccess) return bad(BDX_EHIP);
    if (d->h_progress.ensure(64) != hipSuccess) return bad(BDX_ENOMEM);
    memset(d->h_progress.p, 0, 64);
    mark("ring and state");
    if (sink && sink->adopted) return bad(BDX_ESTATE);   // (before the pinning threads start: a decoder that fails from here on is destroyed at once)
    // (behind the decoder's own pinned allocation: page pinning does not run in parallel with itself)
    if (p->piece_bytes && p->piece_blocks) {
        for (auto& st : d->staging) st.pinned.store(0);
        for (auto& r : d->slot_ready) r.store(0);
        d->rec_ready.store(0);
        const size_t nb = p->piece_bytes + 64, nt = p->piece_blocks * sizeof(bdx_bgzf_block);
        const size_t slot_bytes = d->batch_bytes + p->piece_bytes + 4096, slot_blocks = d->batch_blocks + p->piece_blocks + 64;
        d->pin_thread = std::thread([d, nb, nt, device, slot_bytes, slot_blocks] {
            hipError_t e = hipSetDevice(device);
            auto pin = [&](bdx_bamdec::Staging& st) {
                if (e == hipSuccess) e = st.h_comp.ensure(nb);
                if (e == hipSuccess) e = st.h_tab.ensure(nt);
                st.pin_status = e;
                st.pinned.store(1, std::memory_order_release);
            };
            auto slot = [&](int k) {   // (what bam_open_batch / bam_launch_batch would size: a failure here is met again, and reported, there)
                bdx_bamdec::Slot& sl = d->slot[k];
                if (e == hipSuccess && sl.d_comp.ensure(slot_bytes) == hipSuccess && sl.h_blocks.ensure(slot_blocks * sizeof(BgzfBlock)) == hipSuccess) {
                    (void)sl.d_blocks.ensure(slot_blocks * sizeof(BgzfBlock));
                    (void)sl.d_status.ensure(slot_blocks * 4);
                }
                d->slot_ready[k].store(1, std::memory_order_release);
            };
            // in the order the feeding thread asks for them: the first pieces' buffers, the first batch's slot, the rest
            pin(d->staging[0]);
            slot(0);
            for (int i = 1; i < kBamStaging; ++i) { pin(d->staging[i]); if (i == 2) slot(1); if (i == 5) slot(2); }
            for (int k = 3; k < kBamSlots; ++k) slot(k);
            if (e == hipSuccess) {   // the record stage's scratch and raw columns, for the largest batch (bam_record_stage's sizes)
                const size_t nbk = d->batch_blocks + d->batch_blocks / 4 + 64;
                const size_t cap = round_up(nbk * (size_t)65536 / 36, 1024);
                bool ok = cap <= 0xFFFFFFFFull;
                for (DevBuf* b : {&d->r_tid, &d->r_pos, &d->r_mtid, &d->r_mpos, &d->r_isize}) ok = ok && b->ensure(cap * 4) == hipSuccess;
                ok = ok && d->r_flag.ensure(cap * 2) == hipSuccess && d->r_qlen.ensure(cap * 2) == hipSuccess && d->r_mapq.ensure(cap) == hipSuccess &&
                     d->r_lib.ensure(cap) == hipSuccess && d->r_keep.ensure(cap) == hipSuccess && d->r_key.ensure(cap * 8) == hipSuccess && d->r_check.ensure(cap * 8) == hipSuccess &&
                     d->d_scan.ensure((cap / 256 + 8) * 4) == hipSuccess;
                if (ok) d->raw_cap = (uint32_t)cap;
                if (ok && d->d_cb.ensure(nbk * sizeof(ChainBlock)) == hipSuccess && d->d_offs.ensure(nbk * kRecSlots * 2) == hipSuccess && d->d_base.ensure((nbk + 2) * 4) == hipSuccess)
                    d->rec_cap_blk = nbk;
            }
            d->rec_ready.store(1, std::memory_order_release);
        });
    }
    if (sink) {
        if (sink->adopted) return bad(BDX_ESTATE);
        if (sink->n == 0 && p->expected_bytes) {   // (a record takes 50-150 bytes of BAM; a store that is too small grows)
            // ... plus room for what the batches in flight could hold at most (36 bytes is the smallest record): the store must be able to
            // take them before their record counts are known, and growing it means waiting for the device and copying the columns
            const size_t in_flight = (size_t)kBamSlots * (d->batch_blocks + d->batch_blocks / 4 + 64) * 65536 / 36;
            const size_t want = std::min<size_t>(p->expected_bytes / 48 + ((size_t)1 << 20) + std::min<size_t>(in_flight, p->expected_bytes * 4), 0xFFFFFFFFull - 1024);
            if (sink->cap < want && alloc_reads(sink, want) != BDX_OK) return bad(BDX_ENOMEM);
            mark("sink store");
        }
        if (sink->n == 0 && sink->cap) {   // pass 1 runs as the records arrive
            sink->key_segs.clear();
            const uint64_t tiles = (sink->cap + kTile - 1) / kTile;
            if (tiles <= 0xFFFFFFFFull) {
                if (pass1_prepare(sink, (uint32_t)tiles) != BDX_OK) return bad(BDX_EHIP);
                sink->k1_live = true;
            }
        }
        mark("pass-1 tables");
        if (sink->key_segs.empty() || sink->key_segs.back().host) sink->key_segs.push_back(bdx_ctx::KeySeg{(uint64_t)sink->n, nullptr, nullptr, nullptr});
        d->confirmed = sink->n;
        d->records_at_arm = sink->n;
        // (records this decoder appends come behind what the store already holds)
        st.n_kept = sink->n;
        if (hipMemcpy(d->d_state.p, &st, sizeof(st), hipMemcpyHostToDevice) != hipSuccess) return bad(BDX_EHIP);
    }
    if (!sink && p->expected_bytes) {   // (the decoder's own columns, sized like a sink's store: growing them waits for the device and copies)
        const size_t in_flight = (size_t)kBamSlots * (d->batch_blocks + d->batch_blocks / 4 + 64) * 65536 / 36;
        const size_t want = std::min<size_t>(p->expected_bytes / 48 + ((size_t)1 << 20) + std::min<size_t>(in_flight, p->expected_bytes * 4), 0xFFFFFFFFull - 1024);
        if (bam_own_reserve(d, want, 0) != BDX_OK) return bad(BDX_ENOMEM);
    }
    mark("done");
    *out = d;
    return BDX_OK;
}

void bdx_bamdec_destroy(bdx_bamdec* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->presize_thread.joinable()) d->presize_thread.join();
    if (d->pin_thread.joinable()) d->pin_thread.join();   // (before anything is released: the thread may still be pinning staging buffers)
    for (hipStream_t s : {d->s_copy, d->s_inf, d->s_inf2, d->s_rec})
        if (s) (void)hipStreamSynchronize(s);
    for (auto& sl : d->slot) {
        sl.h_blocks.release(); sl.d_comp.release(); sl.d_blocks.release(); sl.d_status.release();
        if (sl.ev_copied) (void)hipEventDestroy(sl.ev_copied);
        if (sl.ev_free) (void)hipEventDestroy(sl.ev_free);
    }
    for (auto& st : d->staging) {
        st.h_comp.release(); st.h_tab.release();
        if (st.ev_copied) (void)hipEventDestroy(st.ev_copied);
    }
    for (auto& p : d->pieces) {
        if (p.ev_inflated) (void)hipEventDestroy(p.ev_inflated);
        if (p.ev_records) (void)hipEventDestroy(p.ev_records);
    }
    for (auto& e : d->rec_events) (void)hipEventDestroy(e.second);
    for (hipEvent_t e : d->ev_pool) (void)hipEventDestroy(e);
    for (auto& pr : d->kz_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (DevBuf* b : {&d->d_ring, &d->d_cb, &d->d_offs, &d->d_base, &d->d_scan, &d->d_state, &d->r_tid, &d->r_pos, &d->r_mtid, &d->r_mpos, &d->r_isize,
                      &d->r_flag, &d->r_qlen, &d->r_mapq, &d->r_lib, &d->r_keep, &d->r_key, &d->r_check, &d->o_check, &d->d_rg_hash, &d->d_rg_off, &d->d_rg_chars, &d->d_rg_lib,
                      &d->o_tid, &d->o_pos, &d->o_mtid, &d->o_mpos, &d->o_isize, &d->o_flag, &d->o_qlen, &d->o_mapq, &d->o_lib, &d->o_bam, &d->o_key})
        b->release();
    d->h_progress.release();
    if (!d->own_inf_stream) d->s_inf = nullptr;
    d->s_inf2 = nullptr;
    if (d->borrowed_copy) d->s_copy = nullptr;
    if (d->borrowed_rec) d->s_rec = nullptr;
    for (hipStream_t s : {d->s_copy, d->s_inf, d->s_inf2, d->s_rec})
        if (s) (void)hipStreamDestroy(s);
    delete d;
}

const char* bdx_bamdec_last_error(const bdx_bamdec* d) { return d ? d->err.c_str() : ""; }

int bdx_bamdec_acquire(bdx_bamdec* d, size_t bytes, size_t max_blocks, void** buf, bdx_bgzf_block** blocks) {
    if (!d || !buf || !blocks || bytes == 0 || max_blocks == 0) return BDX_EINVAL;
    if (d->held_staging >= kBamStaging) return bfail(d, BDX_ESTATE, "every staging buffer is acquired: submit a piece first");
    if (d->finished) return bfail(d, BDX_ESTATE, "the decoder has finished");
    BHIP(d, hipSetDevice(d->device));
    bdx_bamdec::Staging& st = d->staging[d->next_staging];
    if (st.busy) {
        BamTimer t(d->host_ms[0]);
        BHIP(d, hipEventSynchronize(st.ev_copied));
        st.busy = false;
    }
    {
        BamTimer t(d->host_ms[1]);
        while (!st.pinned.load(std::memory_order_acquire)) std::this_thread::yield();
        BHIP(d, st.pin_status);
        BHIP(d, st.h_comp.ensure(bytes + 64));
        BHIP(d, st.h_tab.ensure(max_blocks * sizeof(bdx_bgzf_block)));
    }
    *buf = st.h_comp.p;
    *blocks = st.h_tab.as<bdx_bgzf_block>();
    st.cap = max_blocks;
    st.cap_bytes = bytes;
    ++d->held_staging;
    d->next_staging = (d->next_staging + 1) % kBamStaging;
    return BDX_OK;
}

}  // extern "C"

namespace {

// The batch in slot si goes to work: a place in the ring, its table to HBM, the inflate launch; the record stage of the batch
// in front of it (which now has its successor's bytes behind it), and its own if it is the last.
int bam_launch_batch(bdx_bamdec* d, int si, bool last) {
    BamTimer t5(d->host_ms[5]);
    bdx_bamdec::Slot& sl = d->slot[si];
    const uint64_t ulen = sl.ulen;
    const size_t nblocks = sl.nblk;
    if (ulen * 4 > d->ring_bytes) return bfail(d, BDX_ELIMIT, "batch too large for the inflate ring (it holds four batches)");
    BamPiece p;
    p.seq = ++d->n_pieces;
    p.slot = si;
    p.nblk = (uint32_t)nblocks;
    // Batches alternate between two streams: a launch whose members outnumber the GPU's wave slots ends with the slots draining
    // (a member takes ~10 ms however many run beside it), and the next batch's waves fill them as they come free.
    hipStream_t s_inf = (p.seq & 1) ? d->s_inf : d->s_inf2;
    if (d->cursor + ulen > d->ring_bytes) { p.wrapped = !d->pieces.empty(); d->cursor = 0; }
    p.ring_beg = d->cursor;
    p.ring_end = p.mirror_end = d->cursor + ulen;
    d->cursor = p.ring_end;
    if (p.wrapped && !d->pieces.empty()) {   // the front of this batch will be mirrored behind its predecessor
        BamPiece& prev = d->pieces.back();
        prev.mirror_end = prev.ring_end + std::min<uint64_t>(kBamMargin, ulen);
        p.prev_end = prev.ring_end;
    }
    BgzfBlock* tb = sl.h_blocks.as<BgzfBlock>();
    for (size_t i = 0; i < nblocks; ++i) tb[i].out_off += p.ring_beg;   // (offsets within the batch so far)
    // the ring bytes this batch will overwrite must have been consumed: the record stages of the batches that still live there
    // (its own mirror, should its successor wrap, is written by ITS record stage, on the stream on which all older batches'
    // record stages have run by then)
    for (auto& q : d->pieces) {
        const bool overlap = q.ring_beg < p.ring_end && p.ring_beg < q.mirror_end;
        if (!overlap) continue;
        // (a batch whose record stage still waits for its successor -- this batch -- cannot give its bytes up: the ring is too small)
        if (!q.records_done) return bfail(d, BDX_ELIMIT, "inflate ring too small for the batches in flight");
        BHIP(d, hipStreamWaitEvent(s_inf, q.ev_records, 0));
    }
    // (sized for what the slot can hold, once: growing a buffer frees it first, and hipFree waits for the device)
    BHIP(d, sl.d_blocks.ensure(std::max<size_t>(std::max(nblocks, sl.cap_blk), 1) * sizeof(BgzfBlock)));
    BHIP(d, sl.d_status.ensure(std::max<size_t>(std::max(nblocks, sl.cap_blk), 1) * 4));
    if (nblocks) BHIP(d, hipMemcpyAsync(sl.d_blocks.p, tb, nblocks * sizeof(BgzfBlock), hipMemcpyHostToDevice, d->s_copy));
    BHIP(d, hipEventRecord(sl.ev_copied, d->s_copy));
    BHIP(d, hipStreamWaitEvent(s_inf, sl.ev_copied, 0));
    hipEvent_t kz0 = nullptr, kz1 = nullptr;
    if (d->time_kernels && hipEventCreate(&kz0) == hipSuccess && hipEventCreate(&kz1) == hipSuccess) (void)hipEventRecord(kz0, s_inf);
    launch_kz_inflate(sl.d_comp.as<uint8_t>(), sl.d_blocks.as<BgzfBlock>(), (uint32_t)nblocks, d->d_ring.as<uint8_t>(), sl.d_status.as<uint32_t>(), s_inf);
    if (kz0 && kz1) { (void)hipEventRecord(kz1, s_inf); d->kz_events.emplace_back(kz0, kz1); }
    p.ev_inflated = bam_event(d);
    if (!p.ev_inflated) return bfail(d, BDX_EHIP, "hipEventCreate");
    BHIP(d, hipEventRecord(p.ev_inflated, s_inf));
    sl.busy = true;
    sl.open = false;
    if (p.seq == 1) d->first_batch_bytes = sl.bytes;
    d->batch_end_bytes.emplace_back(p.seq, d->compressed_bytes - d->bytes_at_arm);
    while (d->batch_end_bytes.size() > 64) d->batch_end_bytes.pop_front();
    if (d->host_ms[8] == 0) d->host_ms[8] = ms_between(d->t_armed, std::chrono::steady_clock::now());
    d->inflated_bytes += ulen;
    d->pieces.push_back(p);
    d->cur_slot = (si + 1) % kBamSlots;
    if (d->pieces.size() >= 2) {
        BamPiece& prev = d->pieces[d->pieces.size() - 2];
        if (!prev.records_done) {
            const int rc = bam_record_stage(d, prev, &d->pieces.back(), 0);
            if (rc != BDX_OK) return rc;
        }
    }
    if (last) {
        const int rc = bam_record_stage(d, d->pieces.back(), nullptr, 1);
        if (rc != BDX_OK) return rc;
        d->finished = true;
    }
    // forget batches whose ring bytes nobody can need any more: all but the last few
    while (d->pieces.size() > (size_t)kBamSlots + 2 && d->pieces.front().records_done) {
        BamPiece& f = d->pieces.front();
        // (a later batch that lands on its bytes waits for ev_records; once the event has completed that wait is void)
        if (hipEventQuery(f.ev_records) != hipSuccess) break;
        d->ev_pool.push_back(f.ev_inflated);
        d->ev_pool.push_back(f.ev_records);
        d->pieces.pop_front();
    }
    return BDX_OK;
}

// the batch that takes the next piece: the open one, or slot cur_slot once its previous tenant's record stage is through
int bam_open_batch(bdx_bamdec* d, size_t piece_bytes, size_t piece_blocks) {
    bdx_bamdec::Slot& sl = d->slot[d->cur_slot];
    if (sl.open) return BDX_OK;
    while (!d->slot_ready[d->cur_slot].load(std::memory_order_acquire)) std::this_thread::yield();   // (its buffers are being allocated by the pin thread)
    if (sl.busy) {
        BamTimer t(d->host_ms[2]);
        BHIP(d, hipEventSynchronize(sl.ev_free));
        sl.busy = false;
    }
    BamTimer t3(d->host_ms[3]);
    // room for a batch plus the piece that takes it over the threshold (and the kernel's input ring reads ~1.1 KiB behind a payload)
    BHIP(d, sl.d_comp.ensure(d->batch_bytes + piece_bytes + 4096));
    const size_t cap = d->batch_blocks + piece_blocks + 64;
    BHIP(d, sl.h_blocks.ensure(cap * sizeof(BgzfBlock)));
    sl.cap_blk = sl.h_blocks.bytes / sizeof(BgzfBlock);
    sl.bytes = 0; sl.nblk = 0; sl.ulen = 0;
    sl.open = true;
    return BDX_OK;
}

}  // namespace

extern "C" {

int bdx_bamdec_submit(bdx_bamdec* d, size_t bytes, size_t nblocks, int last) {
    if (!d) return BDX_EINVAL;
    if (d->held_staging <= 0) return bfail(d, BDX_ESTATE, "no piece was acquired");
    BHIP(d, hipSetDevice(d->device));
    bdx_bamdec::Staging& st = d->staging[(d->next_staging + kBamStaging - d->held_staging) % kBamStaging];   // the oldest piece held
    --d->held_staging;
    if (nblocks > st.cap) return bfail(d, BDX_EINVAL, "more blocks than the acquired table holds");
    if (bytes > st.cap_bytes) return bfail(d, BDX_EINVAL, "more bytes than the acquired piece holds");
    const bdx_bgzf_block* hb = st.h_tab.as<bdx_bgzf_block>();
    for (size_t i = 0; i < nblocks; ++i)   // (offset near 2^64 must not wrap the sum)
        if (hb[i].inflated_len > 65536 || hb[i].offset > bytes || hb[i].payload_len > bytes - hb[i].offset) return bfail(d, BDX_EINVAL, "BGZF block table does not fit the piece");
    int rc = bam_open_batch(d, bytes, nblocks);
    if (rc != BDX_OK) return rc;
    {   // a piece that does not fit the open batch's buffers any more: that batch goes first
        bdx_bamdec::Slot& cur = d->slot[d->cur_slot];
        if (cur.nblk && (cur.bytes + bytes + 4096 > cur.d_comp.bytes || cur.nblk + nblocks > cur.cap_blk)) {
            rc = bam_launch_batch(d, d->cur_slot, false);
            if (rc == BDX_OK) rc = bam_open_batch(d, bytes, nblocks);
            if (rc != BDX_OK) return rc;
        }
    }
    bdx_bamdec::Slot& sl = d->slot[d->cur_slot];
    if (sl.bytes + bytes + 4096 > sl.d_comp.bytes || sl.nblk + nblocks > sl.cap_blk) return bfail(d, BDX_ELIMIT, "piece larger than a batch");
    {
        BamTimer t(d->host_ms[4]);
        if (bytes) BHIP(d, hipMemcpyAsync((char*)sl.d_comp.p + sl.bytes, st.h_comp.p, bytes, hipMemcpyHostToDevice, d->s_copy));
        BHIP(d, hipEventRecord(st.ev_copied, d->s_copy));
    }
    st.busy = true;
    BgzfBlock* tb = sl.h_blocks.as<BgzfBlock>() + sl.nblk;
    for (size_t i = 0; i < nblocks; ++i) {
        tb[i].in_off = sl.bytes + hb[i].offset; tb[i].in_len = hb[i].payload_len; tb[i].out_off = sl.ulen; tb[i].out_len = hb[i].inflated_len;
        sl.ulen += hb[i].inflated_len;
    }
    sl.nblk += nblocks;
    sl.bytes += (bytes + 7) & ~(size_t)7;
    d->compressed_bytes += bytes;
    d->any_submitted = true;
    // the first batches are smaller: the GPU starts on an eighth of a round of its wave slots (five pieces of the file) while the next ones
    // are read -- a third, two thirds, a round, two, then four (a member takes its 4-8 ms however few run beside it)
    const uint64_t started = d->n_pieces;
    const size_t rb = d->round_blocks;
    const size_t goal_blocks = std::min(d->batch_blocks, started == 0 ? std::max<size_t>(rb / 8, 1) : started == 1 ? std::max<size_t>(rb / 3, 1) : started == 2 ? std::max<size_t>(2 * rb / 3, 1) :
                                                         started == 3 ? rb : started == 4 ? 2 * rb : 4 * rb);
    if (last || sl.nblk >= goal_blocks || sl.bytes >= d->batch_bytes) {
        rc = bam_launch_batch(d, d->cur_slot, last != 0);
        if (rc != BDX_OK) return rc;
    }
    bam_poll(d);
    return bam_feed_classifier(d, false);
}

int bdx_bamdec_progress(bdx_bamdec* d, uint64_t* n_records, uint64_t* n_raw, int* past_region, uint32_t* error) {
    if (!d) return BDX_EINVAL;
    bam_poll(d);
    volatile uint64_t* pr = (volatile uint64_t*)d->h_progress.p;
    if (n_records) *n_records = d->confirmed;
    if (n_raw) *n_raw = pr[2];
    if (past_region) *past_region = (int)((pr[1] >> 8) & 0xFF);
    if (error) *error = (uint32_t)(pr[1] & 0xFF);
    return BDX_OK;
}

int bdx_bamdec_finish(bdx_bamdec* d, uint64_t* n_records) {
    if (!d) return BDX_EINVAL;
    BHIP(d, hipSetDevice(d->device));
    if (!d->finished) {
        // the caller stops early (a region read through the index): what has been submitted is decoded as far as its bytes go
        if (d->slot[d->cur_slot].open && d->slot[d->cur_slot].nblk) {
            const int rc = bam_launch_batch(d, d->cur_slot, false);
            if (rc != BDX_OK) return rc;
        }
        if (!d->pieces.empty() && !d->pieces.back().records_done) {
            const int rc = bam_record_stage(d, d->pieces.back(), nullptr, 2);
            if (rc != BDX_OK) return rc;
        }
        d->finished = true;
        d->held_staging = 0;   // (pieces acquired ahead and never submitted -- a caller that stopped early -- are dropped)
    }
    BHIP(d, hipStreamSynchronize(d->s_copy));
    BHIP(d, hipStreamSynchronize(d->s_inf));
    BHIP(d, hipStreamSynchronize(d->s_inf2));
    BHIP(d, hipStreamSynchronize(d->s_rec));
    PieceState st{};
    BHIP(d, hipMemcpy(&st, d->d_state.p, sizeof(st), hipMemcpyDeviceToHost));
    // inflate status of the slots still around is folded into the state by the stitch kernel
    if (st.error) {
        static const char* what[] = {"", "corrupt BAM record chain", "BAM record larger than the device path handles", "destination full",
                                     "truncated BAM record", "corrupt BGZF block"};
        return bfail(d, st.error == 2 ? BDX_ELIMIT : BDX_EINVAL, what[st.error < 6 ? st.error : 1]);
    }
    d->confirmed = st.n_kept;
    d->confirmed_seq = d->n_pieces;
    d->bounds.clear();
    d->bound_in_flight = 0;
    if (d->presize_thread.joinable()) {
        d->presize_thread.join();
        if (d->presize_rc != BDX_OK) return bfail(d, d->presize_rc, d->sink ? d->sink->sizing_err : "sizing the later stages");
    }
    if (d->sink) {
        const int rc = bam_feed_classifier(d, true);
        // (a file of fewer than four batches: the sizing thread is started by this last feed -- and is through before finish returns: the
        // caller's bdx_run grows the same buffers.  Round 5's sizing pass signalled through a flag on the context that a run beside it
        // saw: an EMPTY table with status 0, tools/determinism_probe.py.  The pass now works on an argument of its own, and a run that
        // finds it in flight refuses with BDX_ESTATE -- bdx_ctx::sizing)
        if (d->presize_thread.joinable()) {
            d->presize_thread.join();
            if (rc == BDX_OK && d->presize_rc != BDX_OK) return bfail(d, d->presize_rc, d->sink->sizing_err);
        }
        if (rc != BDX_OK) return rc;
        d->sink->n = (size_t)st.n_kept;
        d->sink->ran = false;
    }
    if (n_records) *n_records = st.n_kept;
    d->host_ms[9] = ms_between(d->t_armed, std::chrono::steady_clock::now());
    for (auto& pr : d->kz_events) {   // (everything has been waited for above)
        float ms = 0;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { d->host_ms[12] += ms; d->host_ms[13] += 1; }
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    d->kz_events.clear();
    return BDX_OK;
}

// A finished decoder takes another stretch of the same file: all of its buffers, streams and events stay (a decoder's set-up is tens
// of milliseconds -- a 3 GiB ring, four batches' buffers, pinned staging, a stream --, a sequence read through the index is a few), the
// running state starts over: the region filter, where the first record begins, the record chain.  With a sink the records are
// appended behind what its store holds.
int bdx_bamdec_rearm(bdx_bamdec* d, int32_t only_tid, int32_t region_beg, int32_t region_end, uint64_t first_record_offset, size_t expected_bytes) {
    if (!d) return BDX_EINVAL;
    if (!d->finished) return bfail(d, BDX_ESTATE, "bdx_bamdec_finish first");
    BHIP(d, hipSetDevice(d->device));
    for (hipStream_t s : {d->s_copy, d->s_inf, d->s_inf2, d->s_rec})
        if (s) BHIP(d, hipStreamSynchronize(s));
    d->filt.only_tid = only_tid; d->filt.beg = region_beg; d->filt.end = region_end;
    for (auto& st : d->staging) st.busy = false;
    d->next_staging = 0; d->held_staging = 0;
    for (auto& sl : d->slot) { sl.busy = false; sl.open = false; sl.bytes = 0; sl.nblk = 0; sl.ulen = 0; }
    d->cur_slot = 0;
    d->cursor = 0;
    for (auto& p : d->pieces) {
        if (p.ev_inflated) d->ev_pool.push_back(p.ev_inflated);
        if (p.ev_records) d->ev_pool.push_back(p.ev_records);
    }
    d->pieces.clear();
    for (auto& e : d->rec_events) d->ev_pool.push_back(e.second);
    d->rec_events.clear();
    d->bounds.clear();
    d->bound_in_flight = 0;
    d->batch_end_bytes.clear();
    d->bytes_at_arm = d->compressed_bytes;
    d->records_at_arm = d->sink ? d->sink->n : 0;
    d->finished = false; d->any_submitted = false;
    d->host_ms[8] = d->host_ms[9] = 0;
    d->t_armed = std::chrono::steady_clock::now();
    d->expected_bytes = expected_bytes;
    PieceState st{};
    st.next_start = first_record_offset;
    st.n_kept = d->sink ? d->sink->n : 0;
    BHIP(d, hipMemcpy(d->d_state.p, &st, sizeof(st), hipMemcpyHostToDevice));
    if (d->h_progress.p) {
        volatile uint64_t* pr = (volatile uint64_t*)d->h_progress.p;
        pr[0] = st.n_kept; pr[1] = 0; pr[2] = 0;   // (the sequence word stays: piece numbers go on counting)
    }
    d->confirmed = st.n_kept;
    d->confirmed_seq = d->n_pieces;
    if (d->sink && (d->sink->key_segs.empty() || d->sink->key_segs.back().host))
        d->sink->key_segs.push_back(bdx_ctx::KeySeg{(uint64_t)d->sink->n, nullptr, nullptr, nullptr});
    return BDX_OK;
}

int bdx_bamdec_fetch(bdx_bamdec* d, uint64_t first, uint64_t n, const bdx_batch_buf* out) {
    if (!d || !out) return BDX_EINVAL;
    if (d->sink) return bfail(d, BDX_ESTATE, "the records went into the sink context");
    if (!d->finished) return bfail(d, BDX_ESTATE, "bdx_bamdec_finish first");
    if (first + n > d->confirmed || n > out->capacity) return bfail(d, BDX_EINVAL, "range beyond the decoded records");
    if (!n) return BDX_OK;
    BHIP(d, hipSetDevice(d->device));
    // (only the columns the caller gave room for: a merge of several files needs tid, pos and flag)
    if (out->tid) BHIP(d, hipMemcpy(out->tid, d->o_tid.as<int32_t>() + first, n * 4, hipMemcpyDeviceToHost));
    if (out->pos) BHIP(d, hipMemcpy(out->pos, d->o_pos.as<int32_t>() + first, n * 4, hipMemcpyDeviceToHost));
    if (out->mtid) BHIP(d, hipMemcpy(out->mtid, d->o_mtid.as<int32_t>() + first, n * 4, hipMemcpyDeviceToHost));
    if (out->mpos) BHIP(d, hipMemcpy(out->mpos, d->o_mpos.as<int32_t>() + first, n * 4, hipMemcpyDeviceToHost));
    if (out->isize) BHIP(d, hipMemcpy(out->isize, d->o_isize.as<int32_t>() + first, n * 4, hipMemcpyDeviceToHost));
    if (out->flag) BHIP(d, hipMemcpy(out->flag, d->o_flag.as<uint16_t>() + first, n * 2, hipMemcpyDeviceToHost));
    if (out->qlen) BHIP(d, hipMemcpy(out->qlen, d->o_qlen.as<uint16_t>() + first, n * 2, hipMemcpyDeviceToHost));
    if (out->mapq) BHIP(d, hipMemcpy(out->mapq, d->o_mapq.as<uint8_t>() + first, n, hipMemcpyDeviceToHost));
    if (out->lib) BHIP(d, hipMemcpy(out->lib, d->o_lib.as<uint8_t>() + first, n, hipMemcpyDeviceToHost));
    if (out->bam) BHIP(d, hipMemcpy(out->bam, d->o_bam.as<uint8_t>() + first, n, hipMemcpyDeviceToHost));
    if (out->name_key) BHIP(d, hipMemcpy(out->name_key, d->o_key.as<uint64_t>() + first, n * 8, hipMemcpyDeviceToHost));
    if (out->name_check) BHIP(d, hipMemcpy(out->name_check, d->o_check.as<uint64_t>() + first, n * 8, hipMemcpyDeviceToHost));
    return BDX_OK;
}

int bdx_bamdec_stats(const bdx_bamdec* d, uint64_t* compressed_bytes, uint64_t* inflated_bytes, uint64_t* pieces, uint64_t* blocks_walked_twice) {
    if (!d) return BDX_EINVAL;
    if (compressed_bytes) *compressed_bytes = d->compressed_bytes;
    if (inflated_bytes) *inflated_bytes = d->inflated_bytes;
    if (pieces) *pieces = d->n_pieces;
    if (blocks_walked_twice) *blocks_walked_twice = d->h_progress.p ? (((volatile uint64_t*)d->h_progress.p)[1] >> 32) : 0;
    return BDX_OK;
}

}  // extern "C"

namespace {

// the decoders' columns gathered into the context's store behind the `base` records it holds (0: the store is filled from empty)
int gather_decoded(bdx_ctx* c, bdx_bamdec* const* decs, int k, const uint8_t* src_file, const uint32_t* src_index, uint64_t n, uint64_t base) {
    if (!c || !decs || k < 1 || k > kMaxGatherSources || (n && (!src_file || !src_index))) return BDX_EINVAL;
    if (c->adopted) return fail(c, BDX_ESTATE, "the context's reads are not its own");
    if (base + n > 0xFFFFFFFFull - 1024) return fail(c, BDX_ELIMIT, "one context holds at most 2^32 - 1 reads");
    HIPCHK(c, hipSetDevice(c->device));
    GatherSources src{};
    src.k = k;
    uint64_t total = 0;
    for (int b = 0; b < k; ++b) {
        bdx_bamdec* d = decs[b];
        if (!d || d->sink || !d->finished || d->device != c->device) return fail(c, BDX_EINVAL, "a decoder that is finished, keeps its own columns and sits on the context's device");
        GatherSource& g = src.s[b];
        g.tid = d->o_tid.as<int32_t>(); g.pos = d->o_pos.as<int32_t>(); g.mtid = d->o_mtid.as<int32_t>(); g.mpos = d->o_mpos.as<int32_t>();
        g.isize = d->o_isize.as<int32_t>(); g.flag = d->o_flag.as<uint16_t>(); g.qlen = d->o_qlen.as<uint16_t>(); g.mapq = d->o_mapq.as<uint8_t>();
        g.lib = d->o_lib.as<uint8_t>(); g.bam = d->o_bam.as<uint8_t>(); g.key = d->o_key.as<uint64_t>(); g.check = d->o_check.as<uint64_t>();
        g.n = d->confirmed;
        total += d->confirmed;
    }
    if (n != total) return fail(c, BDX_EINVAL, "the merge order does not cover the decoders' records");
    if (!n) return BDX_OK;
    const int rc = alloc_reads(c, (size_t)(base + n));   // (keeps what the store holds)
    if (rc != BDX_OK) return rc;
    DevBuf d_file, d_index, d_err;
    struct Release { DevBuf &a, &b, &c; ~Release() { a.release(); b.release(); c.release(); } } release{d_file, d_index, d_err};
    HIPCHK(c, d_file.ensure((size_t)n)); HIPCHK(c, d_index.ensure((size_t)n * 4)); HIPCHK(c, d_err.ensure(16));
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(d_file.p, src_file, (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_index.p, src_index, (size_t)n * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_err.p, 0, 4, s);
    uint32_t err = 0;
    if (e == hipSuccess) {
        DstColumns dst{};
        dst.tid = (int32_t*)c->d.tid + base; dst.pos = (int32_t*)c->d.pos + base; dst.mtid = (int32_t*)c->d.mtid + base; dst.mpos = (int32_t*)c->d.mpos + base;
        dst.isize = (int32_t*)c->d.isize + base; dst.flag = (uint16_t*)c->d.flag + base; dst.qlen = (uint16_t*)c->d.qlen + base; dst.mapq = (uint8_t*)c->d.mapq + base;
        dst.lib = (uint8_t*)c->d.lib + base; dst.bam = (uint8_t*)c->d.bam + base; dst.key = (uint64_t*)c->d.key + base;
        dst.check = c->d.check ? (uint64_t*)c->d.check + base : nullptr;
        launch_kb_gather(src, d_file.as<uint8_t>(), d_index.as<uint32_t>(), n, dst, d_err.as<uint32_t>(), s);
        e = hipMemcpyAsync(&err, d_err.p, 4, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    HIPCHK(c, e);
    if (err) return fail(c, BDX_EINVAL, "the merge order names a record that does not exist");
    c->n = (size_t)(base + n);
    c->ran = false;
    c->k1_live = false;
    if (!base) c->key_segs.clear();
    return BDX_OK;
}

}  // namespace

extern "C" {

int bdx_merge_decoded(bdx_ctx* c, bdx_bamdec* const* decs, int k, const uint8_t* src_file, const uint32_t* src_index, uint64_t n) {
    if (c && c->n) return fail(c, BDX_ESTATE, "the context already holds reads");
    return gather_decoded(c, decs, k, src_file, src_index, n, 0);
}

int bdx_append_decoded(bdx_ctx* c, bdx_bamdec* const* decs, int k, const uint8_t* src_file, const uint32_t* src_index, uint64_t n) {
    if (!c) return BDX_EINVAL;
    for (auto const& sg : c->key_segs)
        if (sg.host) return fail(c, BDX_ESTATE, "the context holds batches whose name keys are still in the caller's memory (bdx_push)");
    return gather_decoded(c, decs, k, src_file, src_index, n, c->n);
}

int bdx_bamdec_host_ms(const bdx_bamdec* d, float* out, int n) {
    if (!d || !out) return BDX_EINVAL;
    for (int i = 0; i < n; ++i) out[i] = i < 14 ? (float)d->host_ms[i] : 0.0f;
    return BDX_OK;
}

// Kernel-level entry point for the parity tests and bdx-inflate-check: BGZF members inflated by KZ, host buffers in and out.
int bdx_inflate_blocks(int device, const void* compressed, size_t bytes, const bdx_bgzf_block* blocks, size_t nblocks, void* out, size_t out_bytes,
                       uint32_t* status, float* kernel_ms) {
    if (!compressed || !blocks || !out || !status) return BDX_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return BDX_EHIP;
    std::vector<BgzfBlock> tb(nblocks);
    uint64_t o = 0;
    for (size_t i = 0; i < nblocks; ++i) {
        if (blocks[i].inflated_len > 65536 || blocks[i].offset > bytes || blocks[i].payload_len > bytes - blocks[i].offset) return BDX_EINVAL;
        tb[i].in_off = blocks[i].offset; tb[i].in_len = blocks[i].payload_len; tb[i].out_off = o; tb[i].out_len = blocks[i].inflated_len;
        o += blocks[i].inflated_len;
    }
    if (o > out_bytes) return BDX_EINVAL;
    DevBuf d_in, d_out, d_tb, d_st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = BDX_OK;
    auto done = [&](int code) {
        d_in.release(); d_out.release(); d_tb.release(); d_st.release();
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        return code;
    };
    if (d_in.ensure(bytes + 2048) != hipSuccess || d_out.ensure(o + 64) != hipSuccess || d_tb.ensure(std::max<size_t>(nblocks, 1) * sizeof(BgzfBlock)) != hipSuccess ||
        d_st.ensure(std::max<size_t>(nblocks, 1) * 4) != hipSuccess)
        return done(BDX_ENOMEM);
    if (hipMemset(d_in.p, 0, bytes + 2048) != hipSuccess || hipMemcpy(d_in.p, compressed, bytes, hipMemcpyHostToDevice) != hipSuccess ||
        (nblocks && hipMemcpy(d_tb.p, tb.data(), nblocks * sizeof(BgzfBlock), hipMemcpyHostToDevice) != hipSuccess))
        return done(BDX_EHIP);
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(BDX_EHIP);
    (void)hipEventRecord(e0, nullptr);
    // BDX_KZ_PROF=<file>: the kernel's own clocks per member, for tools/bamdec_probe.py
    DevBuf d_prof;
    static const char* const prof_env = getenv("BDX_KZ_PROF");   // (tracing: read once per process)
    const char* prof_path = prof_env;
    if (prof_path && nblocks && (d_prof.ensure(nblocks * 64) != hipSuccess || hipMemset(d_prof.p, 0, nblocks * 64) != hipSuccess)) prof_path = nullptr;
    launch_kz_inflate(d_in.as<uint8_t>(), d_tb.as<BgzfBlock>(), (uint32_t)nblocks, d_out.as<uint8_t>(), d_st.as<uint32_t>(), nullptr,
                          prof_path ? d_prof.as<unsigned long long>() : nullptr);
    (void)hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return done(BDX_EHIP);
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, e0, e1);
    if (prof_path && nblocks) {
        std::vector<unsigned long long> hp(nblocks * 8);
        if (hipMemcpy(hp.data(), d_prof.p, nblocks * 64, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(prof_path, "wb")) { fwrite(hp.data(), 8, hp.size(), f); fclose(f); }
    }
    d_prof.release();
    if ((o && hipMemcpy(out, d_out.p, o, hipMemcpyDeviceToHost) != hipSuccess) ||
        (nblocks && hipMemcpy(status, d_st.p, nblocks * 4, hipMemcpyDeviceToHost) != hipSuccess))
        return done(BDX_EHIP);
    return done(rc);
}

}  // extern "C"
