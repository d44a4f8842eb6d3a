#include "producer.h"

#include <algorithm>
#include <cctype>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <unistd.h>
#include <memory>
#include <thread>
#include <queue>
#include <stdexcept>
#include <future>
#include <mutex>
#include <unordered_map>

#include "bam_reader.h"
#include "column_reader.h"

namespace bdhost {

bdx_batch ReadStream::batch() const {
    bdx_batch b;
    b.tid = tid.data(); b.pos = pos.data(); b.mtid = mtid.data(); b.mpos = mpos.data(); b.isize = isize.data();
    b.flag = flag.data(); b.qlen = qlen.data(); b.mapq = mapq.data(); b.lib = lib.data(); b.bam = bam.data();
    b.name_key = name_key.data();
    b.name_check = name_check.data();
    b.n = size();
    return b;
}

namespace {

// samtools region strings as the reference accepts them for -o (bam_aux.c:107-160 bam_parse_region): "name",
// "name:beg" or "name:beg-end" (1-based, commas allowed); a name that itself contains ':' is tried as a whole
template <class Reader>
bool parse_region(const Reader& rd, const std::string& str, int& tid, int& beg, int& end) {
    std::string s;
    for (char c : str)
        if (!isspace((unsigned char)c)) s += c;
    size_t l = s.size(), name_end = l;
    const size_t colon = s.rfind(':');
    if (colon != std::string::npos) name_end = colon;
    tid = -1;
    if (name_end < l) {
        int n_hyphen = 0;
        size_t i = name_end + 1;
        for (; i < l; ++i) {
            if (s[i] == '-') ++n_hyphen;
            else if (!isdigit((unsigned char)s[i]) && s[i] != ',') break;
        }
        if (i < l || n_hyphen > 1) name_end = l;
        tid = rd.tid_of(s.substr(0, name_end));
        if (tid < 0) {
            tid = rd.tid_of(str);
            if (tid < 0) return false;
            name_end = l;
        }
    } else {
        tid = rd.tid_of(str);
        if (tid < 0) return false;
    }
    if (name_end < l) {
        std::string t;
        for (size_t i = name_end + 1; i < l; ++i)
            if (s[i] != ',') t += s[i];
        beg = atoi(t.c_str());
        const size_t h = t.find('-');
        end = h != std::string::npos ? atoi(t.c_str() + h + 1) : 1 << 29;
        if (beg > 0) --beg;
    } else {
        beg = 0;
        end = 1 << 29;
    }
    return beg <= end;
}

struct Stream {
    std::unique_ptr<BamReader> rd;
    int bam_index = 0;
    int only_tid = -1;
    int beg = 0, end = 1 << 29;
    BamRecord cur{};
    bool valid = false;
    // reader filter of the reference: primary (not secondary / supplementary) and tid >= 0
    // (io/AlignmentFilter.hpp:24-34, io/BamIo.cpp:11-18); -o keeps one tid (RegionLimitedBamReader.hpp:63-71)
    bool advance() {
        while (rd->next(cur)) {
            if (cur.flag & (0x100 | 0x800)) continue;
            if (cur.tid < 0) continue;
            // bam_iter_read keeps the records of the region that overlap it (bam_index.c:571-576 is_overlap)
            if (only_tid >= 0 && (cur.tid != only_tid || !((uint32_t)cur.end_pos > (uint32_t)beg && (uint32_t)cur.pos < (uint32_t)end))) continue;
            return valid = true;
        }
        return valid = false;
    }
};

struct StreamGreater {  // BamMerger::Stream::operator> (io/BamMerger.cpp:40-61), adapted through deref_compare
    bool operator()(const Stream* a, const Stream* b) const {
        const BamRecord &x = a->cur, &y = b->cur;
        if (x.tid > y.tid) return true;
        if (y.tid > x.tid) return false;
        if (x.pos > y.pos) return true;
        if (y.pos > x.pos) return false;
        return ((x.flag >> 4) & 1) > ((y.flag >> 4) & 1);
    }
};


// opens the BAMs and runs the k-way merge, calling f(stream_index, record, bam_index, library) per merged record
template <class F>
void merge_streams(const BamConfig& cfg, const std::string& chr, int threads, std::vector<std::string>* targets, F&& f) {
    std::vector<std::unique_ptr<Stream>> streams;
    for (size_t b = 0; b < cfg.num_bams(); ++b) {
        std::unique_ptr<Stream> s(new Stream);
        s->rd.reset(new BamReader(cfg.bam_files()[b], threads));
        s->bam_index = (int)b;
        if (!chr.empty() && !parse_region(*s->rd, chr, s->only_tid, s->beg, s->end))
            throw std::runtime_error("Failed to parse bam region '" + chr + "' in file " + cfg.bam_files()[b] + ". ");
        streams.push_back(std::move(s));
    }
    if (streams.empty()) throw std::runtime_error("BamMerger created with no input streams!");
    if (targets) *targets = streams[0]->rd->target_names();

    // read-group string -> library index (io/BamConfig.hpp:62-72), cached per distinct RG value
    std::unordered_map<std::string, uint8_t> rg_cache;
    for (auto const& kv : cfg.readgroup_index()) rg_cache[kv.first] = (uint8_t)kv.second;
    const uint8_t fallback = (uint8_t)cfg.fallback_library();

    std::priority_queue<Stream*, std::vector<Stream*>, StreamGreater> pq;
    for (auto& s : streams)
        if (s->advance()) pq.push(s.get());
    // The records were decoded on other cores: their bytes are cold here, so runs of one read group are recognised by the
    // 64-bit key the decoder computed, and the RG string itself is only read the first time a key is seen.
    std::string rgtmp;
    std::unordered_map<uint64_t, uint8_t> by_key;
    uint64_t last_key = ~0ull;
    uint8_t last_lib = fallback;
    uint64_t index = 0;
    auto lib_of = [&](const BamRecord& r) {
        if (r.rg_key == last_key) return last_lib;  // runs of one RG
        auto hit = by_key.find(r.rg_key);  // (a key stands for its string: two RG ids with one 64-bit key are not expected)
        if (hit == by_key.end()) {
            rgtmp.assign(r.rg ? r.rg : "", r.rg ? r.l_rg : 0);
            auto it = rg_cache.find(rgtmp);
            hit = by_key.emplace(r.rg_key, it != rg_cache.end() ? it->second : fallback).first;
        }
        last_key = r.rg_key;
        return last_lib = hit->second;
    };
    if (pq.size() == 1) {  // one BAM: nothing to merge
        Stream* s = pq.top();
        do {
            f(index++, s->cur, s->bam_index, lib_of(s->cur));
        } while (s->advance());
        return;
    }
    while (!pq.empty()) {
        Stream* s = pq.top();
        pq.pop();
        const BamRecord& r = s->cur;
        f(index++, r, s->bam_index, lib_of(r));
        if (s->advance()) pq.push(s);
    }
}

}  // namespace

namespace {

// appends records to the sink's current batch and submits it when it is full
class BatchWriter {
public:
    BatchWriter(BatchSink& sink, size_t batch_records) : sink_(sink), batch_(batch_records) {}
    void append_range(const ColumnChunk& c, size_t lo, size_t hi, uint8_t bam) {
        while (lo < hi) {
            if (!open_) open();
            const size_t m = std::min(hi - lo, buf_.capacity - used_);
            memcpy(buf_.tid + used_, c.tid.data() + lo, m * 4); memcpy(buf_.pos + used_, c.pos.data() + lo, m * 4);
            memcpy(buf_.mtid + used_, c.mtid.data() + lo, m * 4); memcpy(buf_.mpos + used_, c.mpos.data() + lo, m * 4);
            memcpy(buf_.isize + used_, c.isize.data() + lo, m * 4);
            memcpy(buf_.flag + used_, c.flag.data() + lo, m * 2); memcpy(buf_.qlen + used_, c.qlen.data() + lo, m * 2);
            memcpy(buf_.mapq + used_, c.mapq.data() + lo, m); memcpy(buf_.lib + used_, c.lib.data() + lo, m);
            memset(buf_.bam + used_, bam, m);
            memcpy(buf_.name_key + used_, c.name_key.data() + lo, m * 8);
            memcpy(buf_.name_check + used_, c.name_check.data() + lo, m * 8);
            used_ += m; lo += m; total_ += m;
            if (used_ == buf_.capacity) close();
        }
    }
    void append_one(const ColumnChunk& c, size_t i, uint8_t bam) {
        if (!open_) open();
        const size_t u = used_;
        buf_.tid[u] = c.tid[i]; buf_.pos[u] = c.pos[i]; buf_.mtid[u] = c.mtid[i]; buf_.mpos[u] = c.mpos[i]; buf_.isize[u] = c.isize[i];
        buf_.flag[u] = c.flag[i]; buf_.qlen[u] = c.qlen[i]; buf_.mapq[u] = c.mapq[i]; buf_.lib[u] = c.lib[i]; buf_.bam[u] = bam;
        buf_.name_key[u] = c.name_key[i];
        buf_.name_check[u] = c.name_check[i];
        ++used_; ++total_;
        if (used_ == buf_.capacity) close();
    }
    void finish() { if (open_) close(); }
    size_t total() const { return total_; }

private:
    void open() {
        const auto t0 = std::chrono::steady_clock::now();
        buf_ = sink_.acquire(batch_); used_ = 0; open_ = true;
        acquire_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void close() {
        const auto t0 = std::chrono::steady_clock::now();
        sink_.submit(used_); open_ = false;
        submit_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
public:
    double acquire_s_ = 0, submit_s_ = 0, copy_s_ = 0;
private:
    BatchSink& sink_;
    size_t batch_;
    bdx_batch_buf buf_{};
    size_t used_ = 0, total_ = 0;
    bool open_ = false;
};

struct Cursor {  // one file's position in the merge: record i of the chunk the reader handed out last
    std::unique_ptr<ColumnReader> rd;
    const ColumnChunk* chunk = nullptr;
    size_t i = 0;
    uint8_t bam = 0;
    bool advance() {  // to the next record; false at the end of the file
        if (chunk && ++i < chunk->size()) return true;
        do {
            chunk = rd->next();
            i = 0;
        } while (chunk && chunk->size() == 0);
        return chunk != nullptr;
    }
    bool first() {
        do {
            chunk = rd->next();
            i = 0;
        } while (chunk && chunk->size() == 0);
        return chunk != nullptr;
    }
};

struct CursorGreater {  // BamMerger::Stream::operator> (io/BamMerger.cpp:40-61): tid, pos, strand
    bool operator()(const Cursor* a, const Cursor* b) const {
        const ColumnChunk &x = *a->chunk, &y = *b->chunk;
        const size_t i = a->i, j = b->i;
        if (x.tid[i] > y.tid[j]) return true;
        if (y.tid[j] > x.tid[i]) return false;
        if (x.pos[i] > y.pos[j]) return true;
        if (y.pos[j] > x.pos[i]) return false;
        return ((x.flag[i] >> 4) & 1) > ((y.flag[j] >> 4) & 1);
    }
};

struct VectorSink : BatchSink {  // batches appended to a ReadStream
    ReadStream& out;
    size_t base = 0;
    explicit VectorSink(ReadStream& o) : out(o) {}
    bdx_batch_buf acquire(size_t cap) override {
        base = out.size();
        const size_t n = base + cap;
        out.tid.resize(n); out.pos.resize(n); out.mtid.resize(n); out.mpos.resize(n); out.isize.resize(n); out.flag.resize(n);
        out.qlen.resize(n); out.mapq.resize(n); out.lib.resize(n); out.bam.resize(n); out.name_key.resize(n); out.name_check.resize(n);
        bdx_batch_buf b{};
        b.tid = out.tid.data() + base; b.pos = out.pos.data() + base; b.mtid = out.mtid.data() + base; b.mpos = out.mpos.data() + base;
        b.isize = out.isize.data() + base; b.flag = out.flag.data() + base; b.qlen = out.qlen.data() + base;
        b.mapq = out.mapq.data() + base; b.lib = out.lib.data() + base; b.bam = out.bam.data() + base;
        b.name_key = out.name_key.data() + base;
        b.name_check = out.name_check.data() + base;
        b.capacity = cap;
        return b;
    }
    void submit(size_t n) override {
        const size_t m = base + n;
        out.tid.resize(m); out.pos.resize(m); out.mtid.resize(m); out.mpos.resize(m); out.isize.resize(m); out.flag.resize(m);
        out.qlen.resize(m); out.mapq.resize(m); out.lib.resize(m); out.bam.resize(m); out.name_key.resize(m); out.name_check.resize(m);
    }
};

}  // namespace

size_t produce_stream(const BamConfig& cfg, const std::string& chr, int threads, std::vector<std::string>* targets, BatchSink& sink,
                      size_t batch_records) {
    const LibraryResolver libs(cfg);
    const size_t nb = cfg.num_bams();
    if (nb == 0) throw std::runtime_error("BamMerger created with no input streams!");
    const int per = std::max(1, threads / (int)nb);
    std::vector<std::unique_ptr<Cursor>> cur;
    for (size_t b = 0; b < nb; ++b) {
        std::unique_ptr<Cursor> c(new Cursor);
        c->rd.reset(new ColumnReader(cfg.bam_files()[b], per, &libs));
        c->bam = (uint8_t)b;
        RecordFilter f;
        if (!chr.empty() && !parse_region(*c->rd, chr, f.only_tid, f.beg, f.end))
            throw std::runtime_error("Failed to parse bam region '" + chr + "' in file " + cfg.bam_files()[b] + ". ");
        c->rd->start(f);
        cur.push_back(std::move(c));
    }
    if (targets) *targets = cur[0]->rd->target_names();
    BatchWriter w(sink, batch_records);
    if (nb == 1) {  // one file: nothing to merge, whole chunks are copied
        Cursor& c = *cur[0];
        while (const ColumnChunk* ch = c.rd->next()) {
            const auto t0 = std::chrono::steady_clock::now();
            if (ch->size()) w.append_range(*ch, 0, ch->size(), c.bam);
            w.copy_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        w.finish();
        if (getenv("BDX_BAM_PROFILE"))
            fprintf(stderr, "[producer] consumer: copying into batches %.3f s (of which acquire %.3f s, submit %.3f s)\n", w.copy_s_, w.acquire_s_, w.submit_s_);
        return w.total();
    }
    // the reference's k-way merge: a priority queue on (tid, pos, strand), same push / pop sequence as BamMerger
    std::priority_queue<Cursor*, std::vector<Cursor*>, CursorGreater> pq;
    for (auto& c : cur)
        if (c->first()) pq.push(c.get());
    while (!pq.empty()) {
        Cursor* c = pq.top();
        pq.pop();
        w.append_one(*c->chunk, c->i, c->bam);
        if (c->advance()) pq.push(c);
    }
    w.finish();
    return w.total();
}


// ---- device-side decode of a one-BAM configuration ----
namespace {

inline uint32_t dle32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t dle16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// Pieces of a file read AHEAD of the one being cut into members: a pool of threads that lives as long as the decode, fed with slices of 1 MiB.
// The slices are copied out of the file's MAPPING (the reader's own, ColumnReader::mapped) when there is one: pread() of the same bytes from the
// page cache fed the copy engine 39-42 GB/s, memcpy from the mapping 51 -- the engine's own rate -- with half the threads
// (tools/feed_probe.hip, profiles/r05_feed_probe.txt); BDX_READ=pread keeps the system call.
// (Page cache -> pinned memory is a plain copy, one core moves 2-5 GB/s of it through pread.  Until round 4 every piece was read by threads started for
// it -- 244 pieces x 7 threads for a 2 GB file, the piece all they had to work on: 26 GB/s on 16 CPUs; a pool working on up to four pieces at
// a time moves 40 GB/s on the same box, tools/register_probe.hip.)
class ReadPool {
public:
    struct Piece {
        std::mutex mu;
        std::condition_variable cv;
        size_t left = 0;
        bool failed = false;
    };
    explicit ReadPool(int n) {
        for (int i = 0; i < std::max(1, n); ++i) th_.emplace_back([this] { work(); });
    }
    ~ReadPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // bytes [off, off + n) of fd into dst; pc->left counts the slices still on their way
    void read(int fd, const uint8_t* map, size_t off, uint8_t* dst, size_t n, Piece* pc) {
        const size_t slice = (size_t)1 << 20;
        const size_t k = (n + slice - 1) / slice;
        { std::lock_guard<std::mutex> lk(pc->mu); pc->left += k; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t o = 0; o < n; o += slice) q_.push_back(Task{fd, map, off + o, dst + o, std::min(slice, n - o), pc});
        }
        cv_.notify_all();
    }
    static bool wait(Piece* pc) {   // true: every byte arrived
        std::unique_lock<std::mutex> lk(pc->mu);
        pc->cv.wait(lk, [pc] { return pc->left == 0; });
        return !pc->failed;
    }

private:
    struct Task { int fd; const uint8_t* map; size_t off; uint8_t* dst; size_t n; Piece* pc; };
    void work() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                t = q_.front();
                q_.pop_front();
            }
            bool ok = true;
            if (t.map) {
                memcpy(t.dst, t.map + t.off, t.n);
                // (the pages' mappings go again at once, here, on sixteen threads: left to the end, tearing down 16 GB of populated page
                // tables took the thread that unmaps the file 0.2 s)
                const uintptr_t lo = ((uintptr_t)(t.map + t.off) + 4095) & ~(uintptr_t)4095, hi = (uintptr_t)(t.map + t.off + t.n) & ~(uintptr_t)4095;
                if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_DONTNEED);
            } else for (size_t done = 0; done < t.n;) {
                const ssize_t r = pread(t.fd, t.dst + done, t.n - done, (off_t)(t.off + done));
                if (r <= 0) { ok = false; break; }
                done += (size_t)r;
            }
            std::lock_guard<std::mutex> lk(t.pc->mu);
            if (!ok) t.pc->failed = true;
            if (--t.pc->left == 0) t.pc->cv.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::deque<Task> q_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};

}  // namespace

namespace {

inline uint32_t le16_at(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32_at(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Decoders that have done their work.  Releasing one (a 3 GiB ring, four batches' buffers, pinned staging, streams) takes the
// driver tens of milliseconds, and the clustering run does not need the memory back: they are kept until
// release_device_decoders(), which the CLI calls only when it walks its destructors.
std::vector<bdx_bamdec*> g_retired;
std::mutex g_retired_mu;
void retire(bdx_bamdec* d) {
    if (!d) return;
    std::lock_guard<std::mutex> lk(g_retired_mu);
    g_retired.push_back(d);
}

// One file of the configuration through a device-side decoder.  With a sink the records go into its store (and are classified as
// they arrive); without, they stay in the decoder's own columns and the decoder is handed back (*keep) for the merge.
// whole_tid >= 0 (sharded runs): ALL records of that sequence -- through the index, which must be there; span_bytes: what the sequence
// takes of the file (sizes the decoder's buffers and the sink's store instead of "the rest of the file")
// reuse (sharded runs): in / out -- a finished decoder of the same file and sink is armed again (bdx_bamdec_rearm) instead of a new one
// being set up; sized_for: the largest span it will be used for (its buffers are sized once)
size_t decode_on_device(const BamConfig& cfg, size_t bam_index, const std::string& chr, int threads, std::vector<std::string>* targets, bdx_ctx* ctx,
                        int device, bool* unsupported, bdx_bamdec** keep, int whole_tid = -1, size_t span_bytes = 0, bdx_bamdec** reuse = nullptr,
                        size_t sized_for = 0) {
    const std::string& path = cfg.bam_files()[bam_index];
    ColumnReader hdr(path, 1, nullptr);   // (the header: reference names, where the first record lies; the index for -o)
    RecordFilter f;
    if (whole_tid >= 0) {
        f.only_tid = whole_tid;
        f.beg = 0;                // (every placed record of the sequence: the device's overlap test compares unsigned)
        f.end = 0x7FFFFFFF;
    } else if (!chr.empty() && !parse_region(hdr, chr, f.only_tid, f.beg, f.end))
        throw std::runtime_error("Failed to parse bam region '" + chr + "' in file " + path + ". ");
    if (targets) *targets = hdr.target_names();
    size_t member_off = 0;
    uint64_t rec_off = 0;
    bool seeked = false;
    hdr.locate(f, &member_off, &rec_off, &seeked);
    const size_t file_size = hdr.mapped_size();

    std::vector<std::string> ids;
    std::vector<uint8_t> libs;
    for (auto const& kv : cfg.readgroup_index()) { ids.push_back(kv.first); libs.push_back((uint8_t)kv.second); }
    std::vector<const char*> idp;
    for (auto const& s : ids) idp.push_back(s.c_str());
    bdx_bamdec_params p{};
    p.device = device;
    p.n_targets = (int32_t)hdr.target_names().size();
    p.bam_index = (int32_t)bam_index;
    p.only_tid = f.only_tid; p.region_beg = f.beg; p.region_end = f.end;
    p.n_read_groups = (uint32_t)ids.size();
    p.rg_ids = idp.empty() ? nullptr : idp.data();
    p.rg_lib = libs.empty() ? nullptr : libs.data();
    p.fallback_lib = (uint8_t)cfg.fallback_library();
    p.first_record_offset = rec_off;
    // pieces (transfer units) of 16 MiB; the decoder gathers them into batches of >= 8192 members before it launches kernels.
    // Small files get small buffers.  Test knobs: BDX_BAM_PIECE_BYTES, BDX_BAM_BATCH_BLOCKS, BDX_BAM_RING_BYTES
    const size_t kPiece = getenv("BDX_BAM_PIECE_BYTES") ? (size_t)std::max(1ll, atoll(getenv("BDX_BAM_PIECE_BYTES"))) : ((size_t)8 << 20);   // (a staging buffer costs ~0.22 ms per MiB to pin and is reused dozens of times)
    size_t rest = file_size - std::min(file_size, member_off);
    if (span_bytes) rest = std::min(rest, span_bytes);
    const size_t this_span = rest;
    if (sized_for) rest = std::max(rest, std::min(sized_for, file_size));   // (the buffers of a decoder that is armed again: for its largest stretch)
    // (a file of less than a batch: buffers of its size; else the decoder picks the batches -- up to four rounds of the wave slots for a large input)
    p.batch_bytes = rest < ((size_t)384 << 20) ? ((rest + ((size_t)1 << 20)) >> 20) << 20 : 0;
    p.expected_bytes = rest;
    p.batch_blocks = getenv("BDX_BAM_BATCH_BLOCKS") ? (size_t)std::max(1ll, atoll(getenv("BDX_BAM_BATCH_BLOCKS"))) : 0;
    // The ring of inflated bytes holds FOUR batches (the one whose records are being read, its successor, the one being inflated, slack).
    // A batch is at most the decoder's largest -- rounds x 7,680 members and the piece that took it there, 64 KiB each -- or everything that
    // is read (a BGZF member inflates to at most ~16 x its size; 8 x is assumed of a whole stretch: beyond that the decoder says BDX_ELIMIT
    // and the host reader takes the file).  Sharded runs keep one decoder per rank for all of its sequences: sized for the largest.
    {
        // An inflate launch should take several rounds of the GPU's wave slots once the input is large (a one-round launch ends with the slots
        // draining: 58 against 68-70 GB/s of inflated bytes on the level-1 file, 109 against 168 on a level-6 file of reference-drawn reads) --
        // and "large" is about the INFLATED bytes: a real 30x BAM compresses 5 x, the random-base test files 1.57 x.  The ratio is read off
        // the first members (BSIZE / ISIZE of their headers); one round per 4 GB of inflated bytes expected, at most four.
        {
            const uint8_t* m = hdr.mapped();
            size_t off = member_off, comp = 0, infl = 0;
            for (int k = 0; k < 64 && off + 28 <= file_size; ++k) {
                if (m[off] != 31 || m[off + 1] != 139 || le16_at(m + off + 10) != 6 || m[off + 12] != 'B' || m[off + 13] != 'C') break;   // (the usual BGZF header: else no estimate)
                const size_t total = (size_t)le16_at(m + off + 16) + 1;
                if (total < 26 || off + total > file_size) break;
                comp += total; infl += le32_at(m + off + total - 4);
                off += total;
            }
            if (comp && infl && !p.batch_blocks) {
                const double expected_inflated = (double)rest * (double)infl / (double)comp;
                p.batch_rounds = (int32_t)std::max(1.0, std::min(4.0, expected_inflated / (double)((size_t)4 << 30)));
            }
        }
        // (test / measurement knobs of the CLI; the library reads no environment variable for them: they travel in the parameters)
        if (const char* br = getenv("BDX_BAM_BATCH_ROUNDS")) p.batch_rounds = std::max(1, std::min(16, atoi(br)));
        if (const char* ks = getenv("BDX_KZ_STREAM")) p.stream_mode = !strcmp(ks, "own") ? 1 : !strcmp(ks, "prio") ? 2 : 0;
        if (getenv("BDX_TIMING")) p.time_kernels = 1;   // (the timing lines say what the inflate kernel took inside the pipeline)
        const size_t rounds = p.batch_rounds ? (size_t)p.batch_rounds : std::max<size_t>(1, std::min<size_t>(4, rest / ((size_t)2560 << 20)));
        const size_t blocks = p.batch_blocks ? p.batch_blocks : 7680 * rounds;
        const size_t batch_inflated = (blocks + kPiece / 16384 + 64) * 65536;
        p.ring_bytes = std::max<size_t>((size_t)64 << 20, std::min<size_t>(rest * 32, 4 * batch_inflated));
        if (span_bytes) p.ring_bytes = std::max<size_t>(p.ring_bytes, (size_t)256 << 20);
    }
    if (const char* rb = getenv("BDX_BAM_RING_BYTES")) p.ring_bytes = (size_t)std::max(1ll, atoll(rb));
    // (what bdx_bamdec_acquire will ask for: a piece, a member cut at the piece's end carried over from the one before, and slack)
    p.piece_bytes = std::min(kPiece, rest) + 65536 + 65536;   // (in front: the tail of the piece before; behind: slack)
    p.piece_blocks = kPiece / 512 + 4096;
    bdx_bamdec* dec = reuse ? *reuse : nullptr;
    const auto t_create = std::chrono::steady_clock::now();
    int rc;
    if (dec) {
        rc = bdx_bamdec_rearm(dec, f.only_tid, f.beg, f.end, rec_off, this_span);
        if (rc != BDX_OK) throw std::runtime_error(std::string("bdx_bamdec_rearm: ") + bdx_strerror(rc) + " (" + bdx_bamdec_last_error(dec) + ")");
    } else {
        rc = bdx_bamdec_create(&dec, ctx, &p);
        if (rc != BDX_OK) throw std::runtime_error(std::string("bdx_bamdec_create: ") + bdx_strerror(rc));
        if (reuse) *reuse = dec;
    }
    const double create_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create).count();
    struct Guard { bdx_bamdec* d; ~Guard() { retire(d); } } guard{reuse ? nullptr : dec};   // (a reused decoder is its caller's to retire)
    auto check = [&](int r, const char* what) {
        if (r != BDX_OK) throw std::runtime_error(std::string(what) + ": " + bdx_strerror(r) + " (" + bdx_bamdec_last_error(dec) + ") in " + path);
    };
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Failed to open samfile " + path);
    struct Fd { int fd; ~Fd() { close(fd); } } fdg{fd};

    // pieces of about kPiece bytes, cut at member boundaries: the bytes behind the last whole member of a piece open the next
    const bool timing = getenv("BDX_TIMING") != nullptr;
    double t_acquire = 0, t_read = 0, t_scan = 0, t_submit = 0, t_finish = 0;
    auto clk = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    size_t npieces = 0;
    std::vector<uint8_t> carry;
    const size_t max_blocks = p.piece_blocks;   // (a piece of more members than that -- 512 bytes each on average -- is left to the host reader)
    bool stop = false, too_many_members = false, gave_up_in_submit = false;
    // (a sequence read through the index: its records end where the index says -- nothing behind that is read, let alone inflated)
    const size_t read_end = span_bytes ? std::min(file_size, member_off + span_bytes) : file_size;
    // The file is read AHEAD of the piece being cut: up to kAhead pieces are on their way into staging buffers (ReadPool) while this
    // thread finds the members of the oldest one and submits it.  A piece's bytes land kLead bytes into its buffer: the tail of the
    // piece before it -- the member cut at that piece's end, known only once that piece has been cut -- is put in front of them.
    const size_t kLead = 65536;
    // (a region read through the index stops at the first record behind it: nothing is read in vain.  Test knob: BDX_BAM_AHEAD)
    const int kAhead = seeked ? 1 : getenv("BDX_BAM_AHEAD") ? std::max(1, std::min(4, atoi(getenv("BDX_BAM_AHEAD")))) : 4;
    struct Ahead { uint8_t* b = nullptr; bdx_bgzf_block* tab = nullptr; size_t off = 0, want = 0; ReadPool::Piece pc; };
    std::deque<std::unique_ptr<Ahead>> ahead;
    ReadPool pool(threads);
    const char* read_how = getenv("BDX_READ");
    const uint8_t* read_map = read_how && !strcmp(read_how, "pread") ? nullptr : hdr.mapped();
    size_t next_off = member_off;
    bool queued_all = false;
    struct Drain {   // (whatever ends the loop: no thread may still be writing into a staging buffer when the decoder goes on)
        std::deque<std::unique_ptr<Ahead>>& a;
        ~Drain() { for (auto& x : a) (void)ReadPool::wait(&x->pc); }
    } drain{ahead};
    while (!stop) {
        while (!queued_all && (int)ahead.size() < kAhead) {
            std::unique_ptr<Ahead> a(new Ahead);
            a->off = next_off;
            a->want = std::min(kPiece, read_end - std::min(read_end, next_off));
            void* buf = nullptr;
            auto t0 = clk();
            check(bdx_bamdec_acquire(dec, kLead + a->want + 65536, max_blocks, &buf, &a->tab), "bdx_bamdec_acquire");
            t_acquire += since(t0);
            a->b = (uint8_t*)buf;
            if (a->want) pool.read(fd, read_map, a->off, a->b + kLead, a->want, &a->pc);
            next_off += a->want;
            if (next_off >= read_end) queued_all = true;
            ahead.push_back(std::move(a));
        }
        if (ahead.empty()) break;
        std::unique_ptr<Ahead> cur = std::move(ahead.front());
        ahead.pop_front();
        ++npieces;
        auto t0 = clk();
        const bool read_ok = ReadPool::wait(&cur->pc);
        t_read += since(t0);
        t0 = clk();
        if (!read_ok) throw std::runtime_error("cannot read " + path);
        if (carry.size() > kLead) throw std::runtime_error("BGZF member larger than 64 KiB: " + path);
        const size_t base = kLead - carry.size();
        uint8_t* b = cur->b + base;
        if (!carry.empty()) memcpy(b, carry.data(), carry.size());
        const size_t have = carry.size() + cur->want;
        const size_t off = cur->off + cur->want;
        const bool at_eof = off >= file_size;
        bdx_bgzf_block* tab = cur->tab;
        // the members
        size_t q = 0, nb = 0;
        while (q + 18 <= have) {
            const uint8_t* h = b + q;
            if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF file: " + path);
            const size_t xlen = dle16(h + 10);
            if (q + 12 + xlen > have) break;
            int bsize = -1;
            for (size_t x = 12; x + 4 <= 12 + xlen;) {
                const size_t slen = dle16(h + x + 2);
                if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (int)dle16(h + x + 4);
                x += 4 + slen;
            }
            if (bsize < 0) throw std::runtime_error("BGZF block without BC field: " + path);
            const size_t total = (size_t)bsize + 1;
            if (total < 12 + xlen + 8) throw std::runtime_error("truncated BGZF file: " + path);
            if (q + total > have) break;
            const uint32_t ulen = dle32(h + total - 4);
            if (ulen > 65536) throw std::runtime_error("BGZF block larger than 64 KiB: " + path);
            if (ulen) {   // (members that inflate to nothing -- the EOF marker, flush blocks -- are skipped)
                if (nb >= max_blocks) { too_many_members = true; break; }
                tab[nb].offset = base + q + 12 + xlen;
                tab[nb].payload_len = (uint32_t)(total - 12 - xlen - 8);
                tab[nb].inflated_len = ulen;
                ++nb;
            }
            q += total;
        }
        if (too_many_members) break;
        if (at_eof && q != have) throw std::runtime_error("truncated BGZF file: " + path);
        carry.assign(b + q, b + have);
        if (q == 0 && !at_eof && cur->want && off < read_end) throw std::runtime_error("BGZF member larger than a piece: " + path);
        t_scan += since(t0);
        t0 = clk();
        // (the bytes in front of `base` travel with the piece -- at most 64 KiB of 8 MiB -- and no table entry points at them)
        {
            const int src = bdx_bamdec_submit(dec, base + q, nb, at_eof ? 1 : 0);
            if (src == BDX_ELIMIT && unsupported) { gave_up_in_submit = true; break; }   // (a stretch that inflates beyond what the ring was sized for, ...: the host reader's)
            check(src, "bdx_bamdec_submit");
        }
        t_submit += since(t0);
        if (at_eof) break;
        if (off >= read_end) break;   // (the end of the sequence's span: the stream is cut here, bdx_bamdec_finish drops a record that runs past it)
        if (seeked) {   // a region read through the index: nothing of it lies behind the first record past it
            int past = 0;
            check(bdx_bamdec_progress(dec, nullptr, nullptr, &past, nullptr), "bdx_bamdec_progress");
            if (past) stop = true;
        }
    }
    for (auto& x : ahead) (void)ReadPool::wait(&x->pc);   // (pieces read ahead and not needed: their readers are through before the decoder is)
    uint64_t n = 0;
    const auto tf = clk();
    rc = bdx_bamdec_finish(dec, &n);   // (also when a piece was refused: what is in flight is waited for before the caller starts over)
    if (gave_up_in_submit) rc = BDX_ELIMIT;
    t_finish = since(tf);
    if (timing) {
        float hm[14];
        if (bdx_bamdec_host_ms(dec, hm, 14) == BDX_OK) {
            if (hm[13] > 0) {
                uint64_t cb = 0, ib = 0, np_ = 0, tw = 0;
                (void)bdx_bamdec_stats(dec, &cb, &ib, &np_, &tw);
                fprintf(stderr, "[bdx timing] inflate kernel in the pipeline: %.1f ms in %d launches (HIP events), %.3f GB inflated from %.3f GB: %.1f GB/s of inflated bytes\n",
                        hm[12], (int)hm[13], (double)ib * 1e-9, (double)cb * 1e-9, hm[12] > 0 ? (double)ib * 1e-6 / hm[12] : 0.0);
            }
            fprintf(stderr, "[bdx timing] of the classifier feed: sizing the later stages %.1f ms, classifier launches %.1f ms\n", hm[10], hm[11]);
            // (the rate the file moves at once the GPU has its first batch: what a file of any size approaches)
            const double steady = (hm[9] - hm[8]) * 1e-3;
            fprintf(stderr, "[bdx timing] steady state: %.3f GB of BAM (%zu pieces) between the first inflate launch (%.3f s after the decoder's set-up) and the last record "
                            "(%.3f s): %.3f s, %.2f GB/s\n", (double)(read_end - member_off) * 1e-9, npieces, hm[8] * 1e-3, hm[9] * 1e-3, steady,
                    steady > 0 ? (double)(read_end - member_off) * 1e-9 / steady : 0.0);
        }
        if (bdx_bamdec_host_ms(dec, hm, 8) == BDX_OK)
            fprintf(stderr, "[bdx timing] inside the decoder (ms): staging wait %.1f, staging pinning %.1f, slot wait %.1f, slot buffers %.1f, copy calls %.1f, "
                            "batch launches %.1f (of which record stages %.1f), classifier feed %.1f\n", hm[0], hm[1], hm[2], hm[3], hm[4], hm[5], hm[6], hm[7]);
    }
    if (timing)
        fprintf(stderr, "[bdx timing] device decode: decoder set up in %.3f s; %zu pieces; waiting for a staging buffer %.3f s, reading the file %.3f s (%d threads), member tables %.3f s, "
                        "enqueueing %.3f s, waiting for the GPU at the end %.3f s\n", create_s, npieces, t_acquire, t_read, threads, t_scan, t_submit, t_finish);
    if ((rc == BDX_ELIMIT || too_many_members) && unsupported) { *unsupported = true; return 0; }
    if (too_many_members) throw std::runtime_error("too many BGZF members in a piece: " + path);
    check(rc, "bdx_bamdec_finish");
    if (keep) { *keep = dec; guard.d = nullptr; }
    return (size_t)n;
}

// BamMerger's order over the files' (tid, pos, strand) columns: the same priority queue, the same push / pop sequence as the host
// producer's merge (io/BamMerger.cpp:40-61, 78-110).  One short cut, for queues that hold at most ONE other file (two files in all,
// the tumour / normal case): while the file just taken from stays STRICTLY below the other it would come out on top again, so its
// records are taken without touching the queue.  With more files every record goes through the queue: a push that rises to the top
// and the pop behind it rearrange the others, and their arrangement is what decides the ties that follow.
struct KeyCursor {
    const int32_t *tid, *pos;
    const uint16_t* flag;
    size_t i, n;
    uint8_t file;
};
struct KeyGreater {
    bool operator()(const KeyCursor* a, const KeyCursor* b) const {
        const size_t i = a->i, j = b->i;
        if (a->tid[i] > b->tid[j]) return true;
        if (b->tid[j] > a->tid[i]) return false;
        if (a->pos[i] > b->pos[j]) return true;
        if (b->pos[j] > a->pos[i]) return false;
        return ((a->flag[i] >> 4) & 1) > ((b->flag[j] >> 4) & 1);
    }
};

}  // namespace

namespace {

// Two files.  What the priority queue does with two streams is a rule with one bit of memory: the file that emitted last goes on
// only while its next record is STRICTLY below the other file's; otherwise the other file is taken (ties: the file that waited).
// Nothing ahead of a key influences what happens below it, so the merge can be cut wherever the memory cannot matter -- at a key
// that only ONE of the files holds -- and the pieces merged by threads of their own.
struct TwoWay {
    const int32_t *tid[2], *pos[2];
    const uint16_t* flag[2];
    size_t n[2];
    bool less(int a, size_t i, int b, size_t j) const {   // (tid, pos, strand): io/BamMerger.cpp:40-61
        if (tid[a][i] != tid[b][j]) return tid[a][i] < tid[b][j];
        if (pos[a][i] != pos[b][j]) return pos[a][i] < pos[b][j];
        return ((flag[a][i] >> 4) & 1) < ((flag[b][j] >> 4) & 1);
    }
    // records [i, ie) of file 0 and [j, je) of file 1 into the output from i + j on; `last`: the file that emitted before them
    void merge(size_t i, size_t ie, size_t j, size_t je, int last, uint8_t* src_file, uint32_t* src_index) const {
        size_t o = i + j;
        while (i < ie && j < je) {
            const bool take0 = last == 0 ? less(0, i, 1, j) : !less(1, j, 0, i);
            if (take0) { src_file[o] = 0; src_index[o] = (uint32_t)i; ++i; last = 0; }
            else { src_file[o] = 1; src_index[o] = (uint32_t)j; ++j; last = 1; }
            ++o;
        }
        for (; i < ie; ++i, ++o) { src_file[o] = 0; src_index[o] = (uint32_t)i; }
        for (; j < je; ++j, ++o) { src_file[o] = 1; src_index[o] = (uint32_t)j; }
    }
    // (a file is sorted by reference and position; the strands at one position come in any order: seams are sought by position)
    bool before(int a, size_t i, int b, size_t j) const { return tid[a][i] != tid[b][j] ? tid[a][i] < tid[b][j] : pos[a][i] < pos[b][j]; }
    size_t lower_bound(int f, int kf, size_t ki) const {   // first record of file f whose position is not before that of record ki of file kf
        size_t lo = 0, hi = n[f];
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (before(f, mid, kf, ki)) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
};

void merge_two(const TwoWay& w, int threads, uint8_t* src_file, uint32_t* src_index, int first_last = 1) {
    // seams: (i, j) with every record before them at a position before every record behind them, and file 1 holding nothing at record i's position
    std::vector<std::pair<size_t, size_t>> seam{{0, 0}};
    const int want = std::max(1, threads);
    for (int t = 1; t < want; ++t) {
        size_t i = w.n[0] * (size_t)t / (size_t)want;
        bool found = false;
        for (int tries = 0; tries < 64 && i < w.n[0]; ++tries) {
            i = w.lower_bound(0, 0, i);                    // the first record of file 0 at this position
            const size_t j = w.lower_bound(1, 0, i);
            if (j == w.n[1] || w.before(0, i, 1, j)) {     // file 1 has no record there: the seam cannot see a tie
                if (i > seam.back().first || j > seam.back().second) seam.emplace_back(i, j);
                found = true;
                break;
            }
            size_t lo = i, hi = w.n[0];                    // on to file 0's next position
            while (lo < hi) {
                const size_t mid = (lo + hi) / 2;
                if (!w.before(0, i, 0, mid)) lo = mid + 1; else hi = mid;
            }
            i = lo;
        }
        (void)found;   // (no seam here: the piece before it is simply longer)
    }
    seam.emplace_back(w.n[0], w.n[1]);
    std::vector<std::thread> th;
    for (size_t s = 0; s + 1 < seam.size(); ++s) {
        // (the file that emitted before a seam: nothing at a seam depends on it; the very first piece starts as the queue does -- file 0 on a tie)
        // (first_last: the caller merges a stretch of a longer stream and knows which file emitted in front of it)
        const int last = s == 0 ? first_last : 1;
        auto job = [&w, &seam, s, last, src_file, src_index] { w.merge(seam[s].first, seam[s + 1].first, seam[s].second, seam[s + 1].second, last, src_file, src_index); };
        if (s + 2 < seam.size()) th.emplace_back(job); else job();
    }
    for (auto& t : th) t.join();
}

}  // namespace

void merge_order(const std::vector<const int32_t*>& tid, const std::vector<const int32_t*>& pos, const std::vector<const uint16_t*>& flag,
                 const std::vector<size_t>& n, std::vector<uint8_t>& src_file, std::vector<uint32_t>& src_index, int threads, int emitted_last) {
    const size_t k = n.size();
    size_t total = 0;
    for (size_t b = 0; b < k; ++b) total += n[b];
    src_file.resize(total);
    src_index.resize(total);
    if (k == 2 && threads > 0) {
        const TwoWay w{{tid[0], tid[1]}, {pos[0], pos[1]}, {flag[0], flag[1]}, {n[0], n[1]}};
        merge_two(w, total < ((size_t)1 << 16) && threads < 1000 ? 1 : std::min(threads, 64), src_file.data(), src_index.data(), emitted_last == 0 ? 0 : 1);
        return;
    }
    std::vector<KeyCursor> cur(k);
    std::priority_queue<KeyCursor*, std::vector<KeyCursor*>, KeyGreater> pq;
    for (size_t b = 0; b < k; ++b) {
        cur[b] = KeyCursor{tid[b], pos[b], flag[b], 0, n[b], (uint8_t)b};
        if (n[b]) pq.push(&cur[b]);
    }
    const KeyGreater greater;
    size_t o = 0;
    while (!pq.empty()) {
        KeyCursor* c = pq.top();
        pq.pop();
        src_file[o] = c->file; src_index[o] = (uint32_t)c->i; ++o;
        ++c->i;
        while (c->i < c->n && (pq.empty() || (pq.size() == 1 && greater(pq.top(), c)))) {   // strictly below the only other file
            src_file[o] = c->file; src_index[o] = (uint32_t)c->i; ++o;
            ++c->i;
        }
        if (c->i < c->n) pq.push(c);
    }
}

namespace {
// (tid, pos, strand) of the LAST record of sequence tid in a file, by the host reader through the index: the tail of the sequence is
// decoded (a window of 32 kb in front of the sequence's end, widened while it holds no record).  false: the file has none
bool last_key_of(const std::string& path, int tid, int32_t* pos, int* strand) {
    for (int64_t back = 2 * 16384;; back *= 32) {
        ColumnReader rd(path, 1, nullptr);
        const int64_t len = (size_t)tid < rd.target_lengths().size() ? (int64_t)rd.target_lengths()[tid] : ((int64_t)1 << 29);
        RecordFilter f;
        f.only_tid = tid;
        f.beg = (int)std::max<int64_t>(0, len - back);
        f.end = 0x7FFFFFFF;
        rd.start(f);
        bool any = false;
        while (const ColumnChunk* c = rd.next())
            if (c->size()) { any = true; *pos = c->pos.back(); *strand = (c->flag.back() >> 4) & 1; }
        if (any) return true;
        if (f.beg == 0) return false;
    }
}
}  // namespace

// Indexed BAMs, the chromosomes of one whole-genome run spread over ranks (bdx_dist_*): every rank's thread reads the BGZF ranges of ITS
// chromosomes (the index says where they lie) and decodes them on ITS GPU -- the reference's answer to "one chromosome" is the same indexed
// seek (io/RegionLimitedBamReader.hpp:43-71: bam_index_load, bam_iter_query).  ONE file: straight into the rank's context.  SEVERAL files
// (the tumour / normal pair): per chromosome one decoder per file over that file's range (the decoders are the rank's, armed again for every
// chromosome), the merge order worked out from three columns as for one GPU (merge_order: BamMerger's queue, io/BamMerger.cpp:40-126),
// one gather in HBM behind what the rank's store holds (bdx_append_decoded).  Nothing is decoded twice and no record crosses the host.
// unsupported: a file without an index (or one that does not cover the header's sequences), nothing was done.
size_t produce_sharded_on_device(const BamConfig& cfg, int threads, std::vector<std::string>* targets, const std::vector<bdx_dist*>& ranks,
                                 const std::vector<int>& devices, const std::vector<int>& rank_of, bool* unsupported) {
    *unsupported = true;
    const size_t nb = cfg.num_bams();
    if (nb < 1 || nb > 16) return 0;
    std::vector<std::string> names;
    struct Span { size_t begin = 0, end = 0; bool has = false; };
    std::vector<std::vector<Span>> span(nb);
    for (size_t b = 0; b < nb; ++b) {
        ColumnReader hdr(cfg.bam_files()[b], 1, nullptr);
        if (b == 0) names = hdr.target_names();
        else if (hdr.target_names() != names) return 0;   // (files that disagree about the sequences: the host reader's error message)
        span[b].resize(names.size());
        bool any_index = false;
        for (size_t t = 0; t < names.size(); ++t) {
            bool empty = false;
            span[b][t].has = hdr.index_span((int)t, &span[b][t].begin, &span[b][t].end, &empty);
            if (span[b][t].has || empty) any_index = true;
            if (!span[b][t].has && !empty) return 0;   // (no index, or one that does not cover the header's sequences: the host producer takes the files)
        }
        if (!any_index) return 0;
    }
    if (targets) *targets = names;
    const int world = (int)ranks.size();
    std::vector<size_t> n_of(world, 0);
    std::vector<std::string> errs(world);
    std::vector<int> gave_up(world, 0);
    std::vector<std::thread> th;
    const int per = std::max(2, threads / std::max(1, world));
    const bool timing = getenv("BDX_TIMING") != nullptr;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            try {
                size_t bytes_mine = 0;
                std::vector<size_t> largest(nb, 0);
                for (size_t b = 0; b < nb; ++b)
                    for (size_t t = 0; t < names.size(); ++t)
                        if (rank_of[t] == r && span[b][t].has) {
                            bytes_mine += span[b][t].end - span[b][t].begin;
                            largest[b] = std::max(largest[b], span[b][t].end - span[b][t].begin);
                        }
                const size_t largest_any = *std::max_element(largest.begin(), largest.end());
                bool reserved = false;
                std::vector<bdx_bamdec*> decs(nb, nullptr);   // ONE decoder per rank and file, armed again for every chromosome
                struct Retire { std::vector<bdx_bamdec*>& d; ~Retire() { for (bdx_bamdec* x : d) retire(x); } } retire_guard{decs};
                std::vector<std::vector<int32_t>> tid(nb), pos(nb);
                std::vector<std::vector<uint16_t>> flag(nb);
                std::vector<uint8_t> src_file;
                std::vector<uint32_t> src_index;
                for (size_t t = 0; t < names.size(); ++t) {
                    if (rank_of[t] != r) continue;
                    std::vector<size_t> files;   // the files that hold records of this chromosome
                    for (size_t b = 0; b < nb; ++b)
                        if (span[b][t].has) files.push_back(b);
                    if (files.empty()) continue;
                    bdx_ctx* c = bdx_dist_chromosome(ranks[r], (int)t);
                    if (!c) throw std::runtime_error(std::string("bdx_dist_chromosome: ") + bdx_dist_last_error(ranks[r]));
                    if (bdx_use_name_check(c, 1) != BDX_OK) throw std::runtime_error("bdx_use_name_check");
                    if (!reserved) {   // the rank's store for ALL of its chromosomes (a record takes 50-150 bytes of BAM): no growing between them
                        // (plus what the decoder must assume of a batch in flight before its records are counted: 36 bytes is the smallest record)
                        if (bdx_reserve(c, std::min<size_t>(bytes_mine / 48 + largest_any * 8 / 36 + ((size_t)1 << 20), 0xFFFFFFFFull - 1024)) != BDX_OK) throw std::runtime_error("bdx_reserve");
                        reserved = true;
                    }
                    bool un = false;
                    const auto tc = std::chrono::steady_clock::now();
                    if (nb == 1) {
                        // (the decoder reports the store's record count: the rank's total so far)
                        n_of[r] = decode_on_device(cfg, 0, "", per, nullptr, c, devices[r], &un, nullptr, (int)t, span[0][t].end - span[0][t].begin, &decs[0], largest[0]);
                        if (un) { gave_up[r] = 1; return; }
                    } else {
                        std::vector<size_t> n;
                        std::vector<bdx_bamdec*> use;
                        std::vector<const int32_t*> ptid, ppos;
                        std::vector<const uint16_t*> pflag;
                        size_t total = 0;
                        for (size_t b : files) {
                            const size_t nr = decode_on_device(cfg, b, "", per, nullptr, nullptr, devices[r], &un, nullptr, (int)t, span[b][t].end - span[b][t].begin, &decs[b], largest[b]);
                            if (un) { gave_up[r] = 1; return; }
                            tid[b].resize(nr); pos[b].resize(nr); flag[b].resize(nr);
                            bdx_batch_buf out{};
                            out.tid = tid[b].data(); out.pos = pos[b].data(); out.flag = flag[b].data();
                            out.capacity = nr;
                            const int frc = bdx_bamdec_fetch(decs[b], 0, nr, &out);
                            if (frc != BDX_OK) throw std::runtime_error(std::string("bdx_bamdec_fetch: ") + bdx_strerror(frc) + " (" + bdx_bamdec_last_error(decs[b]) + ")");
                            n.push_back(nr); use.push_back(decs[b]);
                            ptid.push_back(tid[b].data()); ppos.push_back(pos[b].data()); pflag.push_back(flag[b].data());
                            total += nr;
                        }
                        // (BamMerger's order among the files that hold the chromosome: a file without records of it is not in the queue at that point either)
                        // The reference keeps ONE queue across chromosome boundaries (io/BamMerger.cpp:40-126): where the files' first records of a
                        // chromosome tie (same position and strand -- the first mappable base behind a telomere's Ns), the file whose record has been
                        // waiting at the top wins, and that is the one that did NOT emit the genome's last record in front of the chromosome.  Two
                        // files (the tumour / normal pair): worked out from the files' last records on the chromosomes before -- whoever's rank they
                        // were on.  More files: the tie goes to the lowest file index, as a queue filled afresh would have it (only the
                        // order of equal-key records at a chromosome's first position can differ from the reference there).
                        int emitted_last = 1;
                        if (files.size() == 2 && n[0] && n[1] && ppos[0][0] == ppos[1][0] && ((pflag[0][0] >> 4) & 1) == ((pflag[1][0] >> 4) & 1)) {
                            int prev[2] = {-1, -1};
                            for (int x = 0; x < 2; ++x)
                                for (int tp = (int)t - 1; tp >= 0 && prev[x] < 0; --tp)
                                    if (span[files[x]][tp].has) prev[x] = tp;
                            if (prev[0] >= 0 && prev[1] < 0) emitted_last = 0;
                            else if (prev[0] < 0) emitted_last = 1;   // (file 1 alone emitted before; or neither did: the queue's first fill, file 0 on top)
                            else if (prev[0] != prev[1]) emitted_last = prev[0] > prev[1] ? 0 : 1;
                            else {
                                int32_t lp[2] = {0, 0};
                                int ls[2] = {0, 0};
                                bool have = true;
                                for (int x = 0; x < 2 && have; ++x) have = last_key_of(cfg.bam_files()[files[x]], prev[x], &lp[x], &ls[x]);
                                if (have && (lp[0] != lp[1] || ls[0] != ls[1])) emitted_last = (lp[0] > lp[1] || (lp[0] == lp[1] && ls[0] > ls[1])) ? 0 : 1;
                                // (equal last keys as well: what happened in front of THAT tie decides -- left as a fresh queue would have it)
                            }
                        }
                        merge_order(ptid, ppos, pflag, n, src_file, src_index, std::max(1, per), emitted_last);
                        const int mrc = bdx_append_decoded(c, use.data(), (int)use.size(), src_file.data(), src_index.data(), total);
                        if (mrc == BDX_ELIMIT) { gave_up[r] = 1; return; }
                        if (mrc != BDX_OK) throw std::runtime_error(std::string("bdx_append_decoded: ") + bdx_strerror(mrc) + " (" + bdx_last_error(c) + ")");
                        n_of[r] += total;
                    }
                    if (timing)
                        fprintf(stderr, "[bdx timing] rank %d: %s (%zu file%s) in %.3f s\n", r, names[t].c_str(), files.size(), files.size() == 1 ? "" : "s",
                                std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count());
                }
            } catch (std::exception const& e) {
                errs[r] = e.what();
            }
        });
    for (auto& t : th) t.join();
    for (auto const& e : errs)
        if (!e.empty()) throw std::runtime_error(e);
    for (int g : gave_up)
        if (g) return 0;   // (a record the device path does not take: the caller starts over with the host producer)
    *unsupported = false;
    size_t n = 0;
    for (size_t x : n_of) n += x;
    return n;
}

size_t produce_on_device(const BamConfig& cfg, const std::string& chr, int threads, std::vector<std::string>* targets, bdx_ctx* ctx,
                         bool* unsupported) {
    if (unsupported) *unsupported = false;
    const size_t nb = cfg.num_bams();
    if (nb == 0) throw std::runtime_error("BamMerger created with no input streams!");
    if (nb == 1) return decode_on_device(cfg, 0, chr, threads, targets, ctx, 0, unsupported, nullptr);
    // several files: each decoded into its decoder's own columns, the merge order worked out from three of them, one gather in HBM
    if (nb > 16) { if (unsupported) *unsupported = true; return 0; }
    const bool timing = getenv("BDX_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<bdx_bamdec*> decs(nb, nullptr);
    struct Guard { std::vector<bdx_bamdec*>& d; ~Guard() { for (bdx_bamdec* x : d) retire(x); } } guard{decs};
    std::vector<size_t> n(nb, 0);
    const int device = bdx_device(ctx);
    std::vector<std::vector<int32_t>> tid(nb), pos(nb);
    std::vector<std::vector<uint16_t>> flag(nb);
    std::vector<const int32_t*> ptid(nb), ppos(nb);
    std::vector<const uint16_t*> pflag(nb);
    std::vector<std::future<int>> fetched(nb);   // a file's key columns come to the host while the next file is decoded
    size_t total = 0;
    for (size_t b = 0; b < nb; ++b) {
        bool un = false;
        n[b] = decode_on_device(cfg, b, chr, threads, b == 0 ? targets : nullptr, nullptr, device, &un, &decs[b]);
        if (un) {
            for (auto& f : fetched) if (f.valid()) f.wait();
            if (unsupported) *unsupported = true;
            return 0;
        }
        total += n[b];
        fetched[b] = std::async(std::launch::async, [&, b]() -> int {
            tid[b].resize(n[b]); pos[b].resize(n[b]); flag[b].resize(n[b]);
            bdx_batch_buf out{};
            out.tid = tid[b].data(); out.pos = pos[b].data(); out.flag = flag[b].data();
            out.capacity = n[b];
            return bdx_bamdec_fetch(decs[b], 0, n[b], &out);
        });
    }
    const double t_dec = since();
    int fetch_rc = BDX_OK;
    size_t fetch_bad = 0;
    for (size_t b = 0; b < nb; ++b) {
        const int rc = fetched[b].get();
        if (rc != BDX_OK && fetch_rc == BDX_OK) { fetch_rc = rc; fetch_bad = b; }
        ptid[b] = tid[b].data(); ppos[b] = pos[b].data(); pflag[b] = flag[b].data();
    }
    if (fetch_rc != BDX_OK)
        throw std::runtime_error(std::string("bdx_bamdec_fetch: ") + bdx_strerror(fetch_rc) + " (" + bdx_bamdec_last_error(decs[fetch_bad]) + ")");
    if (total > 0xFFFFFFFFull - 1024) { if (unsupported) *unsupported = true; return 0; }
    const double t_keys = since();
    std::vector<uint8_t> src_file;
    std::vector<uint32_t> src_index;
    merge_order(ptid, ppos, pflag, n, src_file, src_index, std::max(1, threads));
    const double t_merge = since();
    const int rc = bdx_merge_decoded(ctx, decs.data(), (int)nb, src_file.data(), src_index.data(), total);
    if (rc != BDX_OK) throw std::runtime_error(std::string("bdx_merge_decoded: ") + bdx_strerror(rc) + " (" + bdx_last_error(ctx) + ")");
    if (timing)
        fprintf(stderr, "[bdx timing] %zu files decoded on the GPU in %.3f s, their keys fetched in %.3f s, merge order in %.3f s, gather in %.3f s\n", nb, t_dec,
                t_keys - t_dec, t_merge - t_keys, since() - t_merge);
    return total;
}


void release_device_decoders() {
    std::lock_guard<std::mutex> lk(g_retired_mu);
    for (bdx_bamdec* d : g_retired) bdx_bamdec_destroy(d);
    g_retired.clear();
}

void read_targets(const BamConfig& cfg, std::vector<std::string>& names, std::vector<uint32_t>& lengths) {
    if (cfg.num_bams() == 0) throw std::runtime_error("BamMerger created with no input streams!");
    ColumnReader rd(cfg.bam_files()[0], 1, nullptr);  // (parses the header only: nothing is decoded before start())
    names = rd.target_names();
    lengths = rd.target_lengths();
}

void produce_merged_by_columns(const BamConfig& cfg, const std::string& chr, int threads, ReadStream& out) {
    const LibraryResolver libs(cfg);
    const size_t nb = cfg.num_bams();
    if (nb == 0) throw std::runtime_error("BamMerger created with no input streams!");
    std::vector<ReadStream> per(nb);
    for (size_t b = 0; b < nb; ++b) {
        ColumnReader rd(cfg.bam_files()[b], std::max(1, threads), &libs);
        RecordFilter f;
        if (!chr.empty() && !parse_region(rd, chr, f.only_tid, f.beg, f.end))
            throw std::runtime_error("Failed to parse bam region '" + chr + "' in file " + cfg.bam_files()[b] + ". ");
        rd.start(f);
        if (b == 0) out.targets = rd.target_names();
        VectorSink sink(per[b]);
        BatchWriter w(sink, 1 << 16);
        while (const ColumnChunk* ch = rd.next())
            if (ch->size()) w.append_range(*ch, 0, ch->size(), (uint8_t)b);
        w.finish();
    }
    std::vector<const int32_t*> tid(nb), pos(nb);
    std::vector<const uint16_t*> flag(nb);
    std::vector<size_t> n(nb);
    for (size_t b = 0; b < nb; ++b) { tid[b] = per[b].tid.data(); pos[b] = per[b].pos.data(); flag[b] = per[b].flag.data(); n[b] = per[b].size(); }
    std::vector<uint8_t> src_file;
    std::vector<uint32_t> src_index;
    // (BDX_DUMP_MERGE=queue: the priority queue itself; =pieces: the two-file rule cut into many pieces even on a small input)
    const char* how = getenv("BDX_DUMP_MERGE");
    merge_order(tid, pos, flag, n, src_file, src_index, how && !strcmp(how, "queue") ? 0 : how && !strcmp(how, "pieces") ? 1007 : 4);
    const size_t total = src_file.size();
    out.tid.resize(total); out.pos.resize(total); out.mtid.resize(total); out.mpos.resize(total); out.isize.resize(total); out.flag.resize(total);
    out.qlen.resize(total); out.mapq.resize(total); out.lib.resize(total); out.bam.resize(total); out.name_key.resize(total); out.name_check.resize(total);
    for (size_t i = 0; i < total; ++i) {
        const ReadStream& s = per[src_file[i]];
        const size_t j = src_index[i];
        out.tid[i] = s.tid[j]; out.pos[i] = s.pos[j]; out.mtid[i] = s.mtid[j]; out.mpos[i] = s.mpos[j]; out.isize[i] = s.isize[j]; out.flag[i] = s.flag[j];
        out.qlen[i] = s.qlen[j]; out.mapq[i] = s.mapq[j]; out.lib[i] = s.lib[j]; out.bam[i] = s.bam[j]; out.name_key[i] = s.name_key[j];
        out.name_check[i] = s.name_check[j];
    }
}

void produce(const BamConfig& cfg, const std::string& chr, int threads, ReadStream& out) {
    VectorSink sink(out);
    produce_stream(cfg, chr, threads, &out.targets, sink);
}

void collect_reads(const BamConfig& cfg, const std::string& chr, int threads, const std::vector<uint64_t>& wanted,
                   std::vector<SupportRead>& out) {
    out.assign(wanted.size(), SupportRead());
    size_t w = 0;
    static const char* nt16 = "=ACMGRSVTWYHKDBN";  // bam_nt16_rev_table
    merge_streams(cfg, chr, threads, nullptr, [&](uint64_t index, const BamRecord& r, int, uint8_t lib) {
        if (w >= wanted.size() || wanted[w] != index) return;
        SupportRead& sr = out[w++];
        sr.tid = r.tid; sr.pos = r.pos; sr.l_qseq = r.l_qseq; sr.bdqual = r.bdqual; sr.lib = lib; sr.rev = (r.flag & 0x10) != 0;
        sr.name.assign(r.qname, r.l_qname);
        sr.bases.resize(r.l_qseq > 0 ? r.l_qseq : 0);
        for (int i = 0; i < r.l_qseq; ++i) sr.bases[i] = nt16[(r.seq[i >> 1] >> ((~i & 1) << 2)) & 0xf];
        sr.has_qual = r.l_qseq > 0 && r.qual[0] != 0xff;
        if (sr.has_qual) sr.qual.assign((const char*)r.qual, r.l_qseq);
    });
}

}  // namespace bdhost
