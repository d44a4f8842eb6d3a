#include "column_reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <chrono>

#include "bam_reader.h"
#include "fast_inflate.h"

namespace bdhost {

namespace {

constexpr size_t kPieceBlocksDefault = 128;  // BGZF blocks per piece: <= 8 MiB inflated, ~40 k records
constexpr size_t kIndexAhead = 64;           // blocks indexed beyond a piece before it is queued (a record that straddles)

size_t env_or(const char* name, size_t dflt) {
    const char* v = getenv(name);
    const long long x = v ? atoll(v) : 0;
    return x > 0 ? (size_t)x : dflt;
}
// test knob: BDX_BAM_PIECE_BLOCKS=1 makes every BGZF block its own piece (every piece boundary a guessed record start)
const size_t kPieceBlocks = env_or("BDX_BAM_PIECE_BLOCKS", kPieceBlocksDefault);
const bool kProfile = getenv("BDX_BAM_PROFILE") != nullptr;
std::atomic<long long> g_inflate_ns{0}, g_parse_ns{0}, g_wait_ns{0}, g_redo{0}, g_scan_ns{0}, g_idle_ns{0};
inline long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

void inflate_raw(const uint8_t* src, size_t clen, uint8_t* dst, size_t ulen, const std::string& path) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib inflateInit2 failed");
    zs.next_in = const_cast<Bytef*>(src);
    zs.avail_in = (uInt)clen;
    zs.next_out = dst;
    zs.avail_out = (uInt)ulen;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) throw std::runtime_error("corrupt BGZF block in " + path);
}

// header of the BGZF member at `off`: payload offset / length and inflated size; false at the end of the file
bool bgzf_member(const uint8_t* map, size_t size, size_t off, const std::string& path, size_t* coff, size_t* clen, uint32_t* ulen,
                 size_t* total) {
    const size_t avail = size - off;
    if (avail == 0) return false;
    if (avail < 18) throw std::runtime_error("truncated BGZF file: " + path);
    const uint8_t* h = map + off;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF file: " + path);
    const uint16_t xlen = le16(h + 10);
    if (avail < (size_t)12 + xlen) throw std::runtime_error("truncated BGZF file: " + path);
    int bsize = -1;
    for (size_t x = 12; x + 4 <= (size_t)12 + xlen;) {
        const uint16_t slen = le16(h + x + 2);
        if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = le16(h + x + 4);
        x += 4 + (size_t)slen;
    }
    if (bsize < 0) throw std::runtime_error("BGZF block without BC field: " + path);
    *total = (size_t)bsize + 1;
    if (avail < *total || *total < (size_t)12 + xlen + 8) throw std::runtime_error("truncated BGZF file: " + path);
    *ulen = le32(h + *total - 4);
    if (*ulen > 65536) throw std::runtime_error("BGZF block larger than 64 KiB: " + path);
    *coff = off + 12 + xlen;
    *clen = *total - 12 - xlen - 8;
    return true;
}

}  // namespace

LibraryResolver::LibraryResolver(const BamConfig& cfg) : fallback_((uint8_t)cfg.fallback_library()) {
    for (auto const& kv : cfg.readgroup_index()) by_rg_[kv.first] = (uint8_t)kv.second;
}

uint8_t LibraryResolver::of(const char* rg, uint32_t len) const {
    auto it = by_rg_.find(std::string(rg ? rg : "", rg ? len : 0));
    return it != by_rg_.end() ? it->second : fallback_;
}

ColumnReader::ColumnReader(const std::string& path, int threads, const LibraryResolver* libs)
    : path_(path), threads_(threads < 1 ? 1 : threads), libs_(libs) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Failed to open samfile " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("Failed to open samfile " + path); }
    map_size_ = (size_t)st.st_size;
    if (map_size_) {
        void* m = mmap(nullptr, map_size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); throw std::runtime_error("Failed to map samfile " + path); }
        map_ = (const uint8_t*)m;
        if (!getenv("BDX_BAM_NO_MADV")) madvise(m, map_size_, MADV_SEQUENTIAL);
    }
    close(fd);
    read_header();
}

ColumnReader::~ColumnReader() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_work_.notify_all();
    cv_free_.notify_all();
    cv_done_.notify_all();
    if (scanner_.joinable()) scanner_.join();
    for (auto& t : threads_v_)
        if (t.joinable()) t.join();
    if (map_) munmap((void*)map_, map_size_);
    if (kProfile)
        fprintf(stderr, "[bam] %s: inflate %.3f s, parse+columns %.3f s (summed over %d workers), workers idle %.3f s, block index %.3f s, consumer waited %.3f s, "
                        "pieces decoded again %lld\n", path_.c_str(), g_inflate_ns / 1e9, g_parse_ns / 1e9, threads_, g_idle_ns / 1e9, g_scan_ns / 1e9,
                g_wait_ns / 1e9, (long long)g_redo);
}

int ColumnReader::tid_of(const std::string& name) const {
    for (size_t i = 0; i < targets_.size(); ++i)
        if (targets_[i] == name) return (int)i;
    return -1;
}

// The header (magic, SAM text, reference names) is read block by block on the calling thread; what is remembered is the
// BGZF block that holds the first record and the record's offset inside it.
void ColumnReader::read_header() {
    std::vector<uint8_t> h;
    std::vector<std::pair<size_t, size_t>> starts;  // (compressed offset, inflated offset) of every block read so far
    size_t off = 0;
    auto more = [&]() -> bool {
        size_t coff, clen, total;
        uint32_t ulen;
        if (!bgzf_member(map_, map_size_, off, path_, &coff, &clen, &ulen, &total)) return false;
        starts.emplace_back(off, h.size());
        const size_t at = h.size();
        h.resize(at + ulen);
        if (ulen) inflate_raw(map_ + coff, clen, h.data() + at, ulen, path_);
        off += total;
        return true;
    };
    auto need = [&](size_t n) {
        while (h.size() < n)
            if (!more()) throw std::runtime_error(path_ + " is not a valid bam file");
    };
    need(12);
    if (memcmp(h.data(), "BAM\1", 4) != 0) throw std::runtime_error(path_ + " is not a valid bam file");
    const uint32_t l_text = le32(h.data() + 4);
    size_t p = 8 + (size_t)l_text;
    need(p + 4);
    const uint32_t n_ref = le32(h.data() + p);
    p += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        need(p + 4);
        const uint32_t l = le32(h.data() + p);
        need(p + 4 + (size_t)l + 4);
        targets_.emplace_back((const char*)h.data() + p + 4, l ? l - 1 : 0);
        target_len_.push_back(le32(h.data() + p + 4 + (size_t)l));
        p += 4 + (size_t)l + 4;
    }
    // the block that holds byte p of the inflated stream (the next block if the header ends exactly at a block boundary)
    size_t k = starts.size();
    while (k > 0 && starts[k - 1].second > p) --k;
    if (k > 0 && p < h.size()) {
        first_block_coff_ = starts[k - 1].first;
        first_rec_abs_ = p - starts[k - 1].second;
    } else {
        first_block_coff_ = off;
        first_rec_abs_ = 0;
    }
}

void ColumnReader::locate(const RecordFilter& f, size_t* member_offset, uint64_t* record_offset, bool* seeked) {
    if (started_) throw std::logic_error("ColumnReader::locate after start");
    bool used_index = false;
    if (f.only_tid >= 0 && !getenv("BDX_BAM_NO_INDEX")) used_index = seek_with_index(f.only_tid, f.beg);
    *member_offset = first_block_coff_;
    *record_offset = first_rec_abs_;
    *seeked = used_index;
}

void ColumnReader::start(const RecordFilter& f) {
    if (started_) throw std::logic_error("ColumnReader::start called twice");
    started_ = true;
    filter_ = f;
    // -o with a BAM index next to the file: start at the region instead of at the first record (the reference reads regions
    // through the index as well, bam_iter_query behind io/RegionLimitedBamReader.hpp:33-71); without one the whole file is
    // decoded and filtered
    if (f.only_tid >= 0 && !getenv("BDX_BAM_NO_INDEX")) seeked_ = seek_with_index(f.only_tid, f.beg);
    expected_abs_ = first_rec_abs_;
    next_scan_ = first_block_coff_;
    // pieces in flight: only columns are kept per piece (~1.4 MB), so the decoders can run far ahead of a consumer that is
    // still waiting for its sink (the GPU runtime takes ~0.3 s to come up)
    const size_t npieces = std::max<size_t>(8 * (size_t)threads_, 64);
    for (size_t i = 0; i < npieces; ++i) {
        pool_.emplace_back(new Piece);
        free_.push_back(pool_.back().get());
    }
    scanner_ = std::thread([this] { scan_blocks(); });
    for (int t = 0; t < threads_; ++t) threads_v_.emplace_back([this] { worker(); });
}

// The linear index of <path>.bai (BAM specification 5.2) gives, per 16 kb window, the smallest virtual offset of a record
// that overlaps the window; decoding starts there.  Any problem with the index file simply means no seek.
bool ColumnReader::seek_with_index(int tid, int beg) {
    std::string bai = path_ + ".bai";
    FILE* f = fopen(bai.c_str(), "rb");
    if (!f && path_.size() > 4 && path_.compare(path_.size() - 4, 4, ".bam") == 0) {
        bai = path_.substr(0, path_.size() - 4) + ".bai";
        f = fopen(bai.c_str(), "rb");
    }
    if (!f) return false;
    std::vector<uint8_t> d;
    {
        uint8_t buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
    }
    size_t p = 0;
    auto u32 = [&](uint32_t& v) { if (p + 4 > d.size()) return false; v = le32(d.data() + p); p += 4; return true; };
    auto u64 = [&](uint64_t& v) {
        if (p + 8 > d.size()) return false;
        v = (uint64_t)le32(d.data() + p) | ((uint64_t)le32(d.data() + p + 4) << 32);
        p += 8;
        return true;
    };
    uint32_t magic, n_ref;
    if (!u32(magic) || magic != 0x01494142u || !u32(n_ref) || (uint32_t)tid >= n_ref) return false;
    for (uint32_t r = 0; r < n_ref; ++r) {
        uint32_t n_bin;
        if (!u32(n_bin)) return false;
        uint64_t min_chunk = ~0ull;
        for (uint32_t b = 0; b < n_bin; ++b) {
            uint32_t bin, n_chunk;
            if (!u32(bin) || !u32(n_chunk)) return false;
            for (uint32_t c = 0; c < n_chunk; ++c) {
                uint64_t cb, ce;
                if (!u64(cb) || !u64(ce)) return false;
                if (bin != 37450 && cb < min_chunk) min_chunk = cb;  // (37450: the metadata pseudo-bin of newer indexers)
            }
        }
        uint32_t n_intv;
        if (!u32(n_intv)) return false;
        if ((int)r != tid) {
            if (p + 8ull * n_intv > d.size()) return false;
            p += 8ull * n_intv;
            continue;
        }
        uint64_t off = 0;
        if (n_intv) {
            uint32_t w = std::min<uint32_t>((uint32_t)std::max(beg, 0) >> 14, n_intv - 1);
            for (;; --w) {  // a window without an entry: the nearest one before it
                const size_t q = p + 8ull * w;
                if (q + 8 > d.size()) return false;
                off = (uint64_t)le32(d.data() + q) | ((uint64_t)le32(d.data() + q + 4) << 32);
                if (off || w == 0) break;
            }
        }
        if (!off) off = min_chunk == ~0ull ? 0 : min_chunk;
        if (!off) return false;  // nothing indexed for this sequence: let the full scan find out
        const size_t coff = (size_t)(off >> 16);
        if (coff >= map_size_) return false;
        first_block_coff_ = coff;
        first_rec_abs_ = off & 0xFFFF;
        return true;
    }
    return false;
}

bool ColumnReader::index_span(int tid, size_t* begin, size_t* end, bool* empty) const {
    if (empty) *empty = false;
    if (getenv("BDX_BAM_NO_INDEX")) return false;
    std::string bai = path_ + ".bai";
    FILE* f = fopen(bai.c_str(), "rb");
    if (!f && path_.size() > 4 && path_.compare(path_.size() - 4, 4, ".bam") == 0) {
        bai = path_.substr(0, path_.size() - 4) + ".bai";
        f = fopen(bai.c_str(), "rb");
    }
    if (!f) return false;
    std::vector<uint8_t> d;
    {
        uint8_t buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
    }
    size_t p = 0;
    auto u32 = [&](uint32_t& v) { if (p + 4 > d.size()) return false; v = le32(d.data() + p); p += 4; return true; };
    auto u64 = [&](uint64_t& v) {
        if (p + 8 > d.size()) return false;
        v = (uint64_t)le32(d.data() + p) | ((uint64_t)le32(d.data() + p + 4) << 32);
        p += 8;
        return true;
    };
    uint32_t magic, n_ref;
    if (!u32(magic) || magic != 0x01494142u || !u32(n_ref) || (uint32_t)tid >= n_ref) return false;
    for (uint32_t r = 0; r < n_ref; ++r) {
        uint32_t n_bin;
        if (!u32(n_bin)) return false;
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t b = 0; b < n_bin; ++b) {
            uint32_t bin, n_chunk;
            if (!u32(bin) || !u32(n_chunk)) return false;
            for (uint32_t c = 0; c < n_chunk; ++c) {
                uint64_t cb, ce;
                if (!u64(cb) || !u64(ce)) return false;
                if (bin == 37450) continue;   // (the metadata pseudo-bin of newer indexers)
                lo = std::min(lo, cb);
                hi = std::max(hi, ce);
            }
        }
        uint32_t n_intv;
        if (!u32(n_intv)) return false;
        if (p + 8ull * n_intv > d.size()) return false;
        p += 8ull * n_intv;
        if ((int)r != tid) continue;
        if (lo == ~0ull) { if (empty) *empty = true; return false; }
        *begin = (size_t)(lo >> 16);
        *end = std::min<size_t>(map_size_, (size_t)(hi >> 16) + 65536 + 28);
        return *begin < map_size_;
    }
    return false;
}

// Index blocks until block i exists (anyone may advance the index: a worker whose last record runs past the blocks indexed
// so far does it itself).  Returns false if the file ends before block i.
bool ColumnReader::wait_for_block(size_t i) {
    std::lock_guard<std::mutex> lk(mu_);
    const long long t0 = kProfile ? now_ns() : 0;
    struct Acc { long long t0; ~Acc() { if (kProfile) g_scan_ns += now_ns() - t0; } } acc{t0};
    while (blocks_.size() <= i && !scan_done_) {
        size_t coff, clen, total;
        uint32_t ulen;
        if (!bgzf_member(map_, map_size_, next_scan_, path_, &coff, &clen, &ulen, &total)) {
            scan_done_ = true;
            break;
        }
        next_scan_ += total;
        if (ulen == 0) continue;  // members that inflate to nothing (the EOF marker, flush blocks)
        blocks_.push_back(Block{coff, clen, ulen, total_ulen_});
        total_ulen_ += ulen;
    }
    return blocks_.size() > i;
}

ColumnReader::Piece* ColumnReader::acquire_piece() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_free_.wait(lk, [&] { return stop_ || !free_.empty(); });
    if (stop_) return nullptr;
    Piece* p = free_.back();
    free_.pop_back();
    return p;
}

void ColumnReader::release_piece(Piece* p) {
    {
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(p);
    }
    cv_free_.notify_one();
}

void ColumnReader::scan_blocks() {
    try {
        size_t b0 = 0, index = 0;
        for (;;) {
            const bool have = wait_for_block(b0);
            if (!have) break;
            (void)wait_for_block(b0 + kPieceBlocks + kIndexAhead);
            Piece* p = acquire_piece();
            if (!p) return;
            {
                std::lock_guard<std::mutex> lk(mu_);
                const size_t b1 = std::min(b0 + kPieceBlocks, blocks_.size());
                p->index = index++;
                p->b0 = b0; p->b1 = b1;
                p->abs_begin = blocks_[b0].uabs;
                p->abs_end = blocks_[b1 - 1].uabs + blocks_[b1 - 1].ulen;
                p->found_start = false; p->first_abs = p->next_abs = 0;
                p->cols.clear(); p->error.clear(); p->done = false;
                work_.push_back(p);
                order_.push_back(p);
                b0 = b1;
            }
            cv_work_.notify_one();
            cv_done_.notify_all();
        }
    } catch (std::exception const& e) {
        std::lock_guard<std::mutex> lk(mu_);
        scan_error_ = e.what();
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        all_queued_ = true;
    }
    cv_work_.notify_all();
    cv_done_.notify_all();
}

void ColumnReader::worker() {
    Scratch sc;  // this thread's inflate buffer, reused for every piece it takes
    for (;;) {
        Piece* p = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu_);
            const long long t0 = kProfile ? now_ns() : 0;
            cv_work_.wait(lk, [&] { return stop_ || !work_.empty() || all_queued_; });
            if (kProfile) g_idle_ns += now_ns() - t0;
            if (stop_) return;
            if (work_.empty()) {
                if (all_queued_) return;
                continue;
            }
            p = work_.front();
            work_.pop_front();
        }
        try {
            decode_piece(sc, *p, p->index == 0, first_rec_abs_);
        } catch (std::exception const& e) {
            p->error = e.what();
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            p->done = true;
        }
        cv_done_.notify_all();
    }
}

void ColumnReader::inflate_into(Scratch& sc, const Piece& p, size_t block) {
    Block b;
    if (block >= sc.first_block && block - sc.first_block < sc.blocks.size()) {
        b = sc.blocks[block - sc.first_block];  // (the piece's own blocks were copied out under one lock)
    } else {
        std::lock_guard<std::mutex> lk(mu_);
        b = blocks_[block];
    }
    const size_t at = (size_t)(b.uabs - p.abs_begin);
    if (sc.buf.size() < at + b.ulen + 16) sc.buf.resize(std::max(at + b.ulen + 16, sc.buf.size() * 2));
    // own decoder first (fast_inflate.cpp); zlib decides about anything it does not like, and takes the blocks whose payload
    // ends too close to the end of the mapping for the decoder's 8-byte loads
    static const bool zlib_only = getenv("BDX_BAM_ZLIB") != nullptr;
    if (zlib_only || b.coff + b.clen + 32 > map_size_ ||
        !fast_inflate(map_ + b.coff, b.clen, sc.buf.data() + at, b.ulen, sc.buf.size() - (at + b.ulen)))
        inflate_raw(map_ + b.coff, b.clen, sc.buf.data() + at, b.ulen, path_);
    // the member's CRC-32 (the four bytes behind its payload) against what came out: a flipped bit in a stored or literal-heavy block
    // inflates without complaint (htslib checks it too; BDX_BAM_NO_CRC=1 skips the check)
    static const bool no_crc = getenv("BDX_BAM_NO_CRC") != nullptr;
    if (!no_crc && b.coff + b.clen + 4 <= map_size_ && crc32_fast(sc.buf.data() + at, b.ulen) != le32(map_ + b.coff + b.clen))
        throw std::runtime_error("BGZF block fails its CRC-32 (corrupt file): " + path_);
    sc.filled = at + b.ulen;
}

// Records that START inside the piece's own blocks -> columns.  known_start: start_abs is a true record boundary (the first
// piece; a piece the consumer decodes again); otherwise the first boundary is guessed -- three records in a row whose field
// ranges, size equation and read name fit -- and the consumer checks it against the chain of true boundaries.
void ColumnReader::decode_piece(Scratch& sc, Piece& p, bool known_start, uint64_t start_abs) {
    const size_t own = (size_t)(p.abs_end - p.abs_begin);
    const long long t_in = kProfile ? now_ns() : 0;
    if (sc.piece != p.index + 1) {  // (the scratch buffer may still hold this piece: a piece the consumer decodes again)
        sc.piece = p.index + 1;
        sc.filled = 0;
        sc.extra_blocks = 0;
        if (sc.buf.size() < own + 65536) sc.buf.resize(own + 65536);
        {
            std::lock_guard<std::mutex> lk(mu_);
            sc.first_block = p.b0;
            sc.blocks.assign(blocks_.begin() + (std::ptrdiff_t)p.b0, blocks_.begin() + (std::ptrdiff_t)p.b1);
        }
        for (size_t b = p.b0; b < p.b1; ++b) inflate_into(sc, p, b);
    }
    const long long t_pa = kProfile ? now_ns() : 0;
    if (kProfile) g_inflate_ns += t_pa - t_in;
    struct Acc { long long t0; ~Acc() { if (kProfile) g_parse_ns += now_ns() - t0; } } acc{t_pa};
    p.cols.clear();
    p.past_region = false;
    size_t pos;
    if (known_start) {
        pos = (size_t)(start_abs - p.abs_begin);
    } else {
        pos = bam_guess_record_start(sc.buf.data(), 0, own, sc.filled, (int32_t)targets_.size());
        if (pos >= own) {  // no record starts in here that can be recognised: the consumer sorts it out
            p.found_start = false;
            return;
        }
    }
    p.found_start = true;
    p.first_abs = p.abs_begin + pos;
    auto extend = [&]() -> bool {  // one more block behind the piece's own
        const size_t nb = p.b1 + sc.extra_blocks;
        if (!wait_for_block(nb)) return false;
        inflate_into(sc, p, nb);
        ++sc.extra_blocks;
        return true;
    };
    std::string last_rg;
    uint8_t last_lib = libs_ ? libs_->of(nullptr, 0) : 0;
    bool have_last = false;
    BamRecord r;
    ColumnChunk& c = p.cols;
    {
        const size_t guess = own / 180 + 16;
        c.tid.reserve(guess); c.pos.reserve(guess); c.mtid.reserve(guess); c.mpos.reserve(guess); c.isize.reserve(guess);
        c.flag.reserve(guess); c.qlen.reserve(guess); c.mapq.reserve(guess); c.lib.reserve(guess); c.name_key.reserve(guess); c.name_check.reserve(guess);
    }
    while (pos < own) {
        while (pos + 4 > sc.filled)
            if (!extend()) {
                if (known_start) throw std::runtime_error("truncated BAM record in " + path_);
                p.found_start = false;
                return;
            }
        const uint32_t bs = le32(sc.buf.data() + pos);
        // from a guessed boundary a record of more than a few MiB is taken for a wrong guess (the consumer decodes the piece again
        // from the true boundary): a misread size word must not make this worker inflate and hold half a gigabyte
        bool ok = bs >= 32 && bs <= (known_start ? (1u << 29) : (4u << 20));
        if (ok) {
            while (pos + 4 + (size_t)bs > sc.filled)
                if (!extend()) {
                    if (known_start) throw std::runtime_error("truncated BAM record in " + path_);
                    p.found_start = false;
                    return;
                }
            const uint8_t* q = sc.buf.data() + pos;
            const uint32_t l_name = q[12], n_cigar = le16(q + 16);
            const int32_t l_seq = (int32_t)le32(q + 20);
            ok = l_seq >= 0 && 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq <= bs;
        }
        if (!ok) {
            if (known_start) throw std::runtime_error("corrupt BAM record in " + path_);
            p.found_start = false;  // a wrong guess
            return;
        }
        BamReader::parse_record(sc.buf.data() + pos, r);
        pos += 4 + (size_t)bs;
        // reader filter of the reference: primary (not secondary / supplementary) and tid >= 0
        // (io/AlignmentFilter.hpp:24-34, io/BamIo.cpp:11-18); -o keeps the records overlapping one region
        if (r.flag & (0x100 | 0x800)) continue;
        if (r.tid < 0) {
            if (seeked_) {  // unplaced reads come last in a sorted file
                p.past_region = true;
                p.next_abs = p.abs_begin + pos;
                return;
            }
            continue;
        }
        if (filter_.only_tid >= 0 &&
            (r.tid != filter_.only_tid || !((uint32_t)r.end_pos > (uint32_t)filter_.beg && (uint32_t)r.pos < (uint32_t)filter_.end))) {
            // a coordinate-sorted file holds nothing of the region behind the first record past it
            if (seeked_ && (r.tid > filter_.only_tid || (r.tid == filter_.only_tid && r.pos >= filter_.end))) {
                p.past_region = true;
                p.next_abs = p.abs_begin + pos;
                return;
            }
            continue;
        }
        uint8_t lib = 0;
        if (libs_) {
            const uint32_t lr = r.rg ? r.l_rg : 0;
            if (have_last && lr == last_rg.size() && (lr == 0 || memcmp(r.rg, last_rg.data(), lr) == 0)) {
                lib = last_lib;  // runs of one read group
            } else {
                lib = libs_->of(r.rg, lr);
                last_rg.assign(r.rg ? r.rg : "", lr);
                last_lib = lib;
                have_last = true;
            }
        }
        c.tid.push_back(r.tid); c.pos.push_back(r.pos); c.mtid.push_back(r.mtid); c.mpos.push_back(r.mpos); c.isize.push_back(r.isize);
        c.flag.push_back(r.flag);
        c.qlen.push_back((uint16_t)(r.l_qseq > 65535 ? 65535 : (r.l_qseq < 0 ? 0 : r.l_qseq)));
        c.mapq.push_back(r.bdqual);
        c.lib.push_back(lib);
        c.name_key.push_back(hash_name(r.qname, r.l_qname));
        c.name_check.push_back(check_name(r.qname, r.l_qname));
    }
    p.next_abs = p.abs_begin + pos;
}

const ColumnChunk* ColumnReader::next() {
    if (!started_) throw std::logic_error("ColumnReader::next before start");
    if (current_) {
        release_piece(current_);
        current_ = nullptr;
    }
    if (region_done_) return nullptr;
    for (;;) {
        Piece* p = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu_);
            const long long t0 = kProfile ? now_ns() : 0;
            cv_done_.wait(lk, [&] { return stop_ || (!order_.empty() && order_.front()->done) || (order_.empty() && all_queued_); });
            if (kProfile) g_wait_ns += now_ns() - t0;
            if (!scan_error_.empty() && order_.empty()) throw std::runtime_error(scan_error_);
            if (order_.empty()) {
                if (expected_abs_ != total_ulen_ && !(total_ulen_ == 0 && expected_abs_ == first_rec_abs_))
                    throw std::runtime_error("truncated BAM record in " + path_);
                return nullptr;
            }
            p = order_.front();
            order_.pop_front();
        }
        if (!p->error.empty()) {
            const std::string e = p->error;
            release_piece(p);
            throw std::runtime_error(e);
        }
        const uint64_t X = expected_abs_;
        if (X >= p->abs_end) {  // the record that began earlier covers the whole piece
            release_piece(p);
            continue;
        }
        if (!p->found_start || p->first_abs != X) {  // a wrong or missing guess: decode the piece again from the true boundary
            if (kProfile) ++g_redo;
            try {
                decode_piece(redo_, *p, true, X);
            } catch (...) {
                release_piece(p);
                throw;
            }
        }
        expected_abs_ = p->next_abs;
        current_ = p;
        if (p->past_region) {  // the rest of the file is behind the region: stop the decoders
            region_done_ = true;
            {
                std::lock_guard<std::mutex> lk(mu_);
                stop_ = true;
            }
            cv_work_.notify_all();
            cv_free_.notify_all();
        }
        return &p->cols;
    }
}

}  // namespace bdhost
