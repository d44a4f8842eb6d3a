#include "bam_reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace bdhost {

namespace {

constexpr size_t kMaxBlocksPerFillDefault = 2048;  // <= 128 MiB decompressed per fill
constexpr size_t kSegBytesDefault = 2u << 20;      // decompressed bytes per decode thread, at least

// test knobs: BDX_BAM_FILL_BLOCKS / BDX_BAM_SEG_BYTES shrink the batches / the decode segments so that small files go
// through many batch hand-overs and many guessed record boundaries
size_t env_or(const char* name, size_t dflt) {
    const char* v = getenv(name);
    const long long x = v ? atoll(v) : 0;
    return x > 0 ? (size_t)x : dflt;
}
const size_t kMaxBlocksPerFill = env_or("BDX_BAM_FILL_BLOCKS", kMaxBlocksPerFillDefault);
const size_t kSegBytes = env_or("BDX_BAM_SEG_BYTES", kSegBytesDefault);

inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct Block {
    size_t coff, clen;   // deflate payload within comp_
    size_t uoff, ulen;   // destination within buf_
};

void inflate_block(const uint8_t* src, size_t clen, uint8_t* dst, size_t ulen) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib inflateInit2 failed");
    zs.next_in = const_cast<Bytef*>(src);
    zs.avail_in = (uInt)clen;
    zs.next_out = dst;
    zs.avail_out = (uInt)ulen;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) throw std::runtime_error("corrupt BGZF block");
}

}  // namespace

uint64_t hash_name(const char* s, size_t n) {
    // 64-bit multiply-xorshift over 8-byte words (names are short; collisions ~ n^2 / 2^65)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, s + i, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    uint64_t w = 0;
    if (i < n) memcpy(&w, s + i, n - i);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    h ^= h >> 32;
    return h;
}

uint64_t check_name(const char* s, size_t n) {
    // the second name hash (bdx_batch::name_check): csrc/bdx_bam_dev.h name_check_*, the same function as the device decoder's
    uint64_t h = 0xD6E8FEB86659FD93ull + (uint64_t)n * 0x9FB21C651E98DF25ull;
    auto step = [](uint64_t a, uint64_t w) {
        a ^= w;
        a = (a << 27) | (a >> 37);
        return a * 0x9FB21C651E98DF25ull + 0x52DCE729ull;
    };
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, s + i, 8);
        h = step(h, w);
    }
    uint64_t w = 0;
    if (i < n) memcpy(&w, s + i, n - i);
    h = step(h, w);
    h ^= h >> 33;
    h *= 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    h *= 0x165667B19E3779F9ull;
    return h ^ (h >> 32);
}

namespace {
constexpr size_t kFrontGap = 4u << 20;  // room in front of a batch for the unparsed tail of the previous one
}  // namespace

void BamReader::Chunk::reserve(size_t n) {
    if (n <= cap) return;
    std::unique_ptr<uint8_t[]> nd(new uint8_t[n]);  // (not value-initialised: 128 MiB of zeroes per batch is real time)
    if (end > beg) memcpy(nd.get() + beg, data.get() + beg, end - beg);
    data = std::move(nd);
    cap = n;
}

BamReader::BamReader(const std::string& path, int threads, size_t fill_blocks)
    : path_(path), threads_(threads < 1 ? 1 : threads), fill_blocks_(fill_blocks ? fill_blocks : kMaxBlocksPerFill) {
    {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("Failed to open samfile " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("Failed to open samfile " + path); }
        map_size_ = (size_t)st.st_size;
        if (map_size_) {
            void* m = mmap(nullptr, map_size_, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); throw std::runtime_error("Failed to map samfile " + path); }
            map_ = (const uint8_t*)m;
            madvise(m, map_size_, MADV_SEQUENTIAL);
        }
        close(fd);
    }
    if (!ensure(12) || memcmp(at(), "BAM\1", 4) != 0) throw std::runtime_error(path + " is not a valid bam file");
    const uint32_t l_text = le32(at() + 4);
    cur_ += 8;
    if (!ensure((size_t)l_text + 4)) throw std::runtime_error(path + " is not a valid bam file");
    header_text_.assign((const char*)at(), l_text);
    cur_ += l_text;
    const uint32_t n_ref = le32(at());
    cur_ += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!ensure(4)) throw std::runtime_error(path + " is not a valid bam file");
        const uint32_t l = le32(at());
        if (!ensure((size_t)4 + l + 4)) throw std::runtime_error(path + " is not a valid bam file");
        targets_.emplace_back((const char*)at() + 4, l ? l - 1 : 0);
        cur_ += 4 + l + 4;
    }
    // the records follow: decode the rest of this batch now, and let the helper thread prepare the next one
    records_mode_ = true;
    Chunk& c = chunk_[cur_chunk_];
    c.beg = cur_;
    c.end = end_;
    parse_chunk(c);
    part_ = rec_ = 0;
    const Chunk* cur = &c;
    const int other = cur_chunk_ ^ 1;
    next_ready_ = std::async(std::launch::async, [this, other, cur] { return fill_and_parse(chunk_[other], cur); });
}

BamReader::~BamReader() {
    if (next_ready_.valid()) {
        try { next_ready_.get(); } catch (...) {}
    }
    if (map_) munmap((void*)map_, map_size_);
}

int BamReader::tid_of(const std::string& name) const {
    for (size_t i = 0; i < targets_.size(); ++i)
        if (targets_[i] == name) return (int)i;
    return -1;
}

bool BamReader::fill(Chunk& c) {
    std::vector<Block> blocks;
    size_t uoff = kFrontGap;
    // the compressed file is mapped, not read: the inflate threads take their input straight from the page cache
    while (blocks.size() < fill_blocks_) {
        const size_t avail = map_size_ - comp_off_;
        if (avail == 0) break;
        if (avail < 18) throw std::runtime_error("truncated BGZF file: " + path_);
        const uint8_t* h = map_ + comp_off_;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF file: " + path_);
        const uint16_t xlen = le16(h + 10);
        if (avail < (size_t)12 + xlen) throw std::runtime_error("truncated BGZF file: " + path_);
        int bsize = -1;
        for (size_t x = 12; x + 4 <= (size_t)12 + xlen;) {
            const uint16_t slen = le16(h + x + 2);
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = le16(h + x + 4);
            x += 4 + (size_t)slen;
        }
        if (bsize < 0) throw std::runtime_error("BGZF block without BC field: " + path_);
        const size_t total = (size_t)bsize + 1;
        if (avail < total || total < (size_t)12 + xlen + 8) throw std::runtime_error("truncated BGZF file: " + path_);
        const uint32_t isize = le32(h + total - 4);
        if (isize > 65536) throw std::runtime_error("BGZF block larger than 64 KiB: " + path_);
        Block b;
        b.coff = comp_off_ + 12 + xlen;
        b.clen = total - 12 - xlen - 8;
        b.uoff = uoff;
        b.ulen = isize;
        uoff += isize;
        comp_off_ += total;
        if (isize) blocks.push_back(b);
    }
    if (blocks.empty()) return false;
    const uint8_t* comp = map_;
    c.beg = c.end = 0;
    c.reserve(std::max(uoff, kFrontGap + (size_t)40 * 1024 * 1024));  // sized once: fresh pages under eight writers are slow
    uint8_t* out = c.data.get();
    const int nt = (int)std::min<size_t>((size_t)threads_, blocks.size());
    if (nt <= 1) {
        for (const Block& b : blocks) inflate_block(comp + b.coff, b.clen, out + b.uoff, b.ulen);
    } else {
        std::vector<std::thread> th;
        std::vector<std::string> errs(nt);
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                try {
                    for (size_t i = (size_t)t; i < blocks.size(); i += (size_t)nt)
                        inflate_block(comp + blocks[i].coff, blocks[i].clen, out + blocks[i].uoff, blocks[i].ulen);
                } catch (std::exception const& e) { errs[t] = e.what(); }
            });
        for (auto& x : th) x.join();
        for (auto& e : errs)
            if (!e.empty()) throw std::runtime_error(e + ": " + path_);
    }
    c.beg = kFrontGap;
    c.end = uoff;
    return true;
}

void BamReader::attach_tail(Chunk& n, const uint8_t* src, size_t tail) {
    if (!tail) return;
    if (tail <= n.beg) {
        memcpy(n.data.get() + n.beg - tail, src, tail);
        n.beg -= tail;
    } else {  // a tail longer than the gap (a record of several MiB): make room the slow way
        std::unique_ptr<uint8_t[]> nd(new uint8_t[tail + (n.end - n.beg) + kFrontGap]);
        memcpy(nd.get() + kFrontGap, src, tail);
        memcpy(nd.get() + kFrontGap + tail, n.data.get() + n.beg, n.end - n.beg);
        n.cap = tail + (n.end - n.beg) + kFrontGap;
        n.end = kFrontGap + tail + (n.end - n.beg);
        n.beg = kFrontGap;
        n.data = std::move(nd);
    }
}

// Header mode (constructor): the next batch is inflated synchronously, the unread bytes of the current one go in front.
bool BamReader::advance() {
    const int nxt = cur_chunk_ ^ 1;
    Chunk& n = chunk_[nxt];
    if (!fill(n)) return false;
    if (end_ > cur_) attach_tail(n, chunk_[cur_chunk_].data.get() + cur_, end_ - cur_);
    cur_chunk_ = nxt;
    cur_ = n.beg;
    end_ = n.end;
    return true;
}

bool BamReader::ensure(size_t need) {
    while (end_ - cur_ < need)
        if (!advance()) return end_ - cur_ >= need;
    return true;
}

namespace {

// Does a BAM record start at data[pos]?  Field ranges, the size equation and the read name have to fit; `next` receives
// the position right after it.
bool plausible_record(const uint8_t* data, size_t pos, size_t end, int32_t n_targets, size_t* next) {
    if (pos + 36 > end) return false;
    const uint8_t* p = data + pos;
    const uint32_t bs = le32(p);
    if (bs < 32 || bs > (1u << 26)) return false;
    const int32_t tid = (int32_t)le32(p + 4), rpos = (int32_t)le32(p + 8);
    const uint32_t l_name = p[12], n_cigar = le16(p + 16);
    const int32_t l_seq = (int32_t)le32(p + 20), mtid = (int32_t)le32(p + 24), mpos = (int32_t)le32(p + 28);
    if (tid < -1 || tid >= n_targets || mtid < -1 || mtid >= n_targets || rpos < -1 || mpos < -1 || l_seq < 0 || l_name < 1) return false;
    const uint64_t need = 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > bs) return false;
    if (pos + 36 + l_name > end) return false;
    const uint8_t* nm = p + 36;
    if (nm[l_name - 1] != 0) return false;
    for (uint32_t i = 0; i + 1 < l_name; ++i)
        if (nm[i] < 33 || nm[i] > 126) return false;
    *next = pos + 4 + (size_t)bs;
    return true;
}

}  // namespace

// first position in [from, seg_end) where three records in a row look valid; seg_end if there is none
size_t bam_guess_record_start(const uint8_t* data, size_t from, size_t seg_end, size_t end, int32_t n_targets) {
    for (size_t pos = from; pos < seg_end; ++pos) {
        size_t a, b, c;
        if (plausible_record(data, pos, end, n_targets, &a) && plausible_record(data, a, end, n_targets, &b) &&
            plausible_record(data, b, end, n_targets, &c))
            return pos;
    }
    return seg_end;
}

namespace {

// decode the records that start in [start, seg_end) and are complete before `end`; returns where it stopped.  `trusted`:
// start is a known record boundary, so a record that does not add up means a corrupt file (otherwise: a wrong guess)
size_t parse_range(const uint8_t* data, size_t start, size_t seg_end, size_t end, bool trusted, const std::string& path,
                   std::vector<BamRecord>& out) {
    size_t pos = start;
    while (pos < seg_end) {
        if (pos + 4 > end) break;
        const uint8_t* p = data + pos;
        const uint32_t bs = le32(p);
        if (pos + 4 + (size_t)bs > end) {
            if (trusted && bs > (1u << 30)) throw std::runtime_error("corrupt BAM record in " + path);
            break;  // incomplete: the rest of it comes with the next batch
        }
        bool ok = bs >= 32;
        if (ok) {
            const uint32_t l_name = p[12], n_cigar = le16(p + 16);
            const int32_t l_seq = (int32_t)le32(p + 20);
            ok = l_seq >= 0 && 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq <= bs;
        }
        if (!ok) {
            if (trusted) throw std::runtime_error("corrupt BAM record in " + path);
            break;
        }
        out.emplace_back();
        BamRecord& r = out.back();
        BamReader::parse_record(p, r);
        r.name_key = hash_name(r.qname, r.l_qname);
        r.rg_key = r.rg ? (hash_name(r.rg, r.l_rg) | 1ull) : 0ull;
        pos += 4 + (size_t)bs;
    }
    return pos;
}

}  // namespace

// Every complete record of the batch, decoded by several threads.  A record's position is only known from the sizes of
// all records before it, so every thread but the first GUESSES where the first record of its segment starts (three
// records in a row whose fields, size equation and read name fit) and decodes from there; afterwards the guesses are
// checked against the chain of true boundaries, and a segment whose guess was wrong is decoded again from the right place.
void BamReader::parse_chunk(Chunk& c) {
    const uint8_t* data = c.data.get();
    const size_t n = c.end - c.beg;
    int P = (int)std::min<size_t>({(size_t)std::max(1, threads_), (size_t)16, n / kSegBytes + 1});
    if ((int)c.parts.size() < P) c.parts.resize(P);
    for (auto& v : c.parts) v.clear();
    std::vector<size_t> seg(P + 1), start(P), stop(P);
    for (int i = 0; i <= P; ++i) seg[i] = c.beg + n * (size_t)i / (size_t)P;
    const int32_t nt = (int32_t)targets_.size();
    std::vector<std::string> errs(P);
    auto work = [&](int i) {
        try {
            c.parts[i].reserve((seg[i + 1] - seg[i]) / 160 + 16);
            start[i] = i == 0 ? c.beg : bam_guess_record_start(data, seg[i], seg[i + 1], c.end, nt);
            stop[i] = parse_range(data, start[i], seg[i + 1], c.end, i == 0, path_, c.parts[i]);
        } catch (std::exception const& e) { errs[i] = e.what(); }
    };
    if (P == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 1; i < P; ++i) th.emplace_back(work, i);
        work(0);
        for (auto& t : th) t.join();
    }
    for (auto& e : errs)
        if (!e.empty()) throw std::runtime_error(e);
    size_t pos = stop[0];
    for (int i = 1; i < P; ++i) {
        if (pos >= seg[i + 1]) { c.parts[i].clear(); continue; }  // the record that began earlier covers the whole segment
        if (start[i] == pos) { pos = stop[i]; continue; }
        c.parts[i].clear();                                        // wrong guess: decode the segment from the true boundary
        pos = parse_range(data, pos, seg[i + 1], c.end, true, path_, c.parts[i]);
    }
    c.tail = pos;
}

bool BamReader::fill_and_parse(Chunk& c, const Chunk* prev) {
    static const bool prof = getenv("BDX_BAM_PROFILE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (!fill(c)) {
        if (prev && prev->end > prev->tail) throw std::runtime_error("truncated BAM record in " + path_);
        return false;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (prev && prev->end > prev->tail) attach_tail(c, prev->data.get() + prev->tail, prev->end - prev->tail);
    parse_chunk(c);
    if (prof) {
        const auto t2 = std::chrono::steady_clock::now();
        size_t nrec = 0;
        for (auto const& v : c.parts) nrec += v.size();
        fprintf(stderr, "[bam] batch: inflate %.1f ms (%zu bytes), decode %.1f ms (%zu records, %zu parts)\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), c.end - c.beg,
                std::chrono::duration<double, std::milli>(t2 - t1).count(), nrec, c.parts.size());
    }
    return true;
}

// Record mode: the batch after the current one is inflated AND decoded on a helper thread while the caller consumes the
// current one; here the caller takes it over and starts the one after it (whose front gap receives this one's tail).
bool BamReader::advance_records() {
    if (!next_ready_.valid()) return false;
    static const bool prof = getenv("BDX_BAM_PROFILE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const bool got = next_ready_.get();
    if (prof) fprintf(stderr, "[bam] consumer waited %.1f ms for the next batch\n",
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (!got) return false;
    const int prev = cur_chunk_;
    cur_chunk_ ^= 1;
    part_ = rec_ = 0;
    const Chunk* cur = &chunk_[cur_chunk_];
    next_ready_ = std::async(std::launch::async, [this, prev, cur] { return fill_and_parse(chunk_[prev], cur); });
    return true;
}

bool BamReader::next(BamRecord& r) {
    while (true) {
        const Chunk& c = chunk_[cur_chunk_];
        while (part_ < c.parts.size()) {
            if (rec_ < c.parts[part_].size()) {
                r = c.parts[part_][rec_++];
                return true;
            }
            ++part_;
            rec_ = 0;
        }
        if (!advance_records()) return false;
    }
}

void BamReader::parse_record(const uint8_t* rec, BamRecord& r) {
    const uint32_t bs = le32(rec);
    const uint8_t* p = rec + 4;
    const uint8_t* end = p + bs;
    r.tid = (int32_t)le32(p);
    r.pos = (int32_t)le32(p + 4);
    const uint32_t l_read_name = p[8];
    r.mapq = p[9];
    const uint32_t n_cigar = le16(p + 12);
    r.flag = le16(p + 14);
    r.l_qseq = (int32_t)le32(p + 16);
    r.mtid = (int32_t)le32(p + 20);
    r.mpos = (int32_t)le32(p + 24);
    r.isize = (int32_t)le32(p + 28);
    r.qname = (const char*)p + 32;
    r.l_qname = l_read_name ? l_read_name - 1 : 0;
    const uint8_t* q = p + 32 + l_read_name + 4 * (size_t)n_cigar;
    {   // samtools bam_calend (bam.c:17-45) for the region overlap test; the rarely used 'B' operator is ignored
        const uint8_t* cg = p + 32 + l_read_name;
        int32_t endp = r.pos;
        for (uint32_t k = 0; k < n_cigar; ++k) {
            const uint32_t c = le32(cg + 4 * k), op = c & 15, len = c >> 4;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) endp += (int32_t)len;  // M D N = X consume the reference
        }
        r.end_pos = n_cigar ? endp : r.pos + 1;
    }
    r.seq = q;
    q += ((size_t)r.l_qseq + 1) / 2;
    r.qual = q;
    q += (size_t)r.l_qseq;
    r.rg = nullptr;
    r.l_rg = 0;
    r.bdqual = r.mapq;
    bool have_am = false;
    // one sweep over the aux block: RG:Z and AM:<int> (bam_aux_get / bam_aux2i semantics)
    while (q + 3 <= end) {
        const uint8_t t0 = q[0], t1 = q[1], ty = q[2];
        q += 3;
        size_t sz = 0;
        long ival = 0;
        bool is_int = false;
        const size_t left = (size_t)(end - q);  // (a truncated aux block ends the sweep: nothing is read past the record)
        switch (ty) {
            case 'A': sz = 1; break;
            case 'c': sz = 1; if (left >= 1) { ival = (int8_t)q[0]; is_int = true; } break;
            case 'C': sz = 1; if (left >= 1) { ival = q[0]; is_int = true; } break;
            case 's': sz = 2; if (left >= 2) { ival = (int16_t)le16(q); is_int = true; } break;
            case 'S': sz = 2; if (left >= 2) { ival = le16(q); is_int = true; } break;
            case 'i': sz = 4; if (left >= 4) { ival = (int32_t)le32(q); is_int = true; } break;
            case 'I': sz = 4; if (left >= 4) { ival = (long)le32(q); is_int = true; } break;
            case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z':
            case 'H': {
                const uint8_t* z = (const uint8_t*)memchr(q, 0, (size_t)(end - q));
                if (!z) { q = end; continue; }
                if (t0 == 'R' && t1 == 'G' && ty == 'Z' && !r.rg) { r.rg = (const char*)q; r.l_rg = (uint32_t)(z - q); }
                q = z + 1;
                continue;
            }
            case 'B': {
                if (q + 5 > end) { q = end; continue; }
                const uint8_t sub = q[0];
                const uint32_t cnt = le32(q + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                const size_t bytes = (size_t)cnt * es;
                q = bytes > (size_t)(end - q - 5) ? end : q + 5 + bytes;
                continue;
            }
            default: q = end; continue;
        }
        if (sz > left) { q = end; continue; }
        if (t0 == 'A' && t1 == 'M' && !have_am) { r.bdqual = (uint8_t)(is_int ? ival : 0); have_am = true; }  // bam_aux2i: 0 for non-integer types
        q += sz;
    }
}

}  // namespace bdhost
