#include "bam_reader.h"

#include <zlib.h>

#include <cstring>
#include <stdexcept>
#include <thread>

namespace bdhost {

namespace {

constexpr size_t kReadChunk = 32u << 20;   // compressed bytes fetched per read()
constexpr size_t kMaxBlocksPerFill = 2048;  // <= 128 MiB decompressed per fill

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

BamReader::BamReader(const std::string& path, int threads) : path_(path), threads_(threads < 1 ? 1 : threads) {
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) throw std::runtime_error("Failed to open samfile " + path);
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
}

BamReader::~BamReader() {
    if (next_ready_.valid()) {
        try { next_ready_.get(); } catch (...) {}
    }
    if (fp_) fclose(fp_);
}

int BamReader::tid_of(const std::string& name) const {
    for (size_t i = 0; i < targets_.size(); ++i)
        if (targets_[i] == name) return (int)i;
    return -1;
}

bool BamReader::fill(Chunk& c) {
    std::vector<Block> blocks;
    size_t uoff = kFrontGap;
    while (true) {
        if (!eof_ && comp_.size() - comp_off_ < (size_t)(128u << 10)) {  // top up the compressed window
            comp_.erase(comp_.begin(), comp_.begin() + comp_off_);
            comp_off_ = 0;
            const size_t old = comp_.size();
            comp_.resize(old + kReadChunk);
            const size_t got = fread(comp_.data() + old, 1, kReadChunk, fp_);
            comp_.resize(old + got);
            if (got < kReadChunk) eof_ = true;
        }
        while (blocks.size() < kMaxBlocksPerFill) {
            const size_t avail = comp_.size() - comp_off_;
            if (avail < 18) break;
            const uint8_t* h = comp_.data() + comp_off_;
            if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF file: " + path_);
            const uint16_t xlen = le16(h + 10);
            if (avail < (size_t)12 + xlen) break;
            int bsize = -1;
            for (size_t x = 12; x + 4 <= (size_t)12 + xlen;) {
                const uint16_t slen = le16(h + x + 2);
                if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = le16(h + x + 4);
                x += 4 + (size_t)slen;
            }
            if (bsize < 0) throw std::runtime_error("BGZF block without BC field: " + path_);
            const size_t total = (size_t)bsize + 1;
            if (avail < total) break;
            const uint32_t isize = le32(h + total - 4);
            Block b;
            b.coff = comp_off_ + 12 + xlen;
            b.clen = total - 12 - xlen - 8;
            b.uoff = uoff;
            b.ulen = isize;
            uoff += isize;
            comp_off_ += total;
            if (isize) blocks.push_back(b);
        }
        if (!blocks.empty()) break;
        if (eof_) {
            if (comp_.size() - comp_off_ != 0) throw std::runtime_error("truncated BGZF file: " + path_);
            return false;
        }
    }
    c.beg = c.end = 0;
    c.reserve(std::max(uoff, kFrontGap + (size_t)40 * 1024 * 1024));  // sized once: fresh pages under eight writers are slow
    uint8_t* out = c.data.get();
    const int nt = (int)std::min<size_t>((size_t)threads_, blocks.size());
    if (nt <= 1) {
        for (const Block& b : blocks) inflate_block(comp_.data() + b.coff, b.clen, out + b.uoff, b.ulen);
    } else {
        std::vector<std::thread> th;
        std::vector<std::string> errs(nt);
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                try {
                    for (size_t i = (size_t)t; i < blocks.size(); i += (size_t)nt)
                        inflate_block(comp_.data() + blocks[i].coff, blocks[i].clen, out + blocks[i].uoff, blocks[i].ulen);
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

// The batch after the current one is inflated on a helper thread while the caller parses; here the caller takes it over
// (moving its unparsed tail into the gap in front of the new batch) and starts the one after it.
bool BamReader::advance() {
    if (!started_) {
        started_ = true;
        next_ready_ = std::async(std::launch::async, [this] { return fill(chunk_[0]); });
        cur_chunk_ = 1;  // (so that the first take-over lands on chunk 0)
    }
    if (!next_ready_.valid()) return false;
    const bool got = next_ready_.get();
    if (!got) return false;
    const int nxt = cur_chunk_ ^ 1;
    Chunk& n = chunk_[nxt];
    const size_t tail = end_ - cur_;
    if (tail) {
        const uint8_t* src = chunk_[cur_chunk_].data.get() + cur_;
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
    const int prev = cur_chunk_;
    cur_chunk_ = nxt;
    cur_ = n.beg;
    end_ = n.end;
    next_ready_ = std::async(std::launch::async, [this, prev] { return fill(chunk_[prev]); });
    return true;
}

bool BamReader::ensure(size_t need) {
    while (end_ - cur_ < need)
        if (!advance()) return end_ - cur_ >= need;
    return true;
}

bool BamReader::next(BamRecord& r) {
    if (!ensure(4)) return false;
    const uint32_t bs = le32(at());
    if (!ensure((size_t)4 + bs)) throw std::runtime_error("truncated BAM record in " + path_);
    {   // The batch was just written by the inflate threads on other cores: every record starts on cold lines, and the next
        // record's address is only known from this one's size.  Neighbouring records have similar sizes, so the lines
        // where the next few records should start are requested now (a wrong guess costs nothing).
        const uint8_t* p = at();
        const size_t step = (size_t)4 + bs;
        if (cur_ + 5 * step + 128 < end_) {
            __builtin_prefetch(p + step); __builtin_prefetch(p + step + 64);
            __builtin_prefetch(p + 2 * step); __builtin_prefetch(p + 2 * step + 64);
            __builtin_prefetch(p + 3 * step); __builtin_prefetch(p + 4 * step);
        }
    }
    parse_record(at(), r);
    cur_ += 4 + bs;
    return true;
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
        switch (ty) {
            case 'A': sz = 1; break;
            case 'c': sz = 1; ival = (int8_t)q[0]; is_int = true; break;
            case 'C': sz = 1; ival = q[0]; is_int = true; break;
            case 's': sz = 2; ival = (int16_t)le16(q); is_int = true; break;
            case 'S': sz = 2; ival = le16(q); is_int = true; break;
            case 'i': sz = 4; ival = (int32_t)le32(q); is_int = true; break;
            case 'I': sz = 4; ival = (long)le32(q); is_int = true; break;
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
                q += 5 + (size_t)cnt * es;
                continue;
            }
            default: q = end; continue;
        }
        if (t0 == 'A' && t1 == 'M' && !have_am) { r.bdqual = (uint8_t)(is_int ? ival : 0); have_am = true; }  // bam_aux2i: 0 for non-integer types
        q += sz;
    }
}

}  // namespace bdhost
