// bd_oracle_bam.cpp -- TEST INFRASTRUCTURE (part of the CPU oracle, never linked into or called by the product).
//
// A single-threaded, reference-shaped BAM front end for the oracle: BGZF blocks are inflated one at a time with zlib
// and every record is copied out, filtered and turned into one oracle record, the way the reference's reader stack does
// (samtools 0.1.19 bgzf_read -> bam_read1 behind io/BamReader.hpp:62-70; filter io/AlignmentFilter.hpp:24-34,
// io/BamIo.cpp:11-18; per-record fields, AM / RG aux scan io/Alignment.cpp:12-29,45-64; RG -> library
// io/AlignmentSource.hpp:48-65, io/BamConfig.hpp:62-72).  The reference decodes every BAM twice -- once per file for
// BamSummary (io/BamSummary.cpp:129-150), once merged for BreakDancer::run (breakdancer/BreakDancer.cpp:131-144) --
// so bdo_bam_load takes the number of decode passes to run; the records of the last pass feed bdo_set_stream.
//
// Used by tests (the oracle must reproduce the golden outputs from the BAM fixtures through this decoder as well as
// through the pure-Python one) and by bench.py's cpu_baseline leg (the CPU cost of the reference-shaped path from BAM).
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int bdo_lib_of_readgroup(void* p, const char* rg);
void bdo_set_targets(void* p, int n, const char** names);
void bdo_set_stream(void* p, int bam, int64_t n, const int32_t* tid, const int32_t* pos, const int32_t* mtid,
                    const int32_t* mpos, const int32_t* isize, const uint16_t* flag, const int32_t* qlen,
                    const uint8_t* bdqual, const int32_t* lib, const uint64_t* name);
}

namespace {

struct Bgzf {  // sequential reader over the concatenated BGZF members of one file
    FILE* f = nullptr;
    std::vector<uint8_t> comp, block;
    size_t at = 0;
    bool eof = false;
    std::string err;

    bool next_block() {
        uint8_t h[18];
        size_t got = fread(h, 1, 18, f);
        if (got == 0) { eof = true; return false; }
        if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') {
            err = "not a BGZF block";
            return false;
        }
        const unsigned bsize = (h[16] | (h[17] << 8)) + 1u;
        const unsigned xlen = h[10] | (h[11] << 8);
        if (bsize < 12u + xlen + 8u) { err = "bad BGZF block size"; return false; }
        comp.resize(bsize - 18);
        if (fread(comp.data(), 1, comp.size(), f) != comp.size()) { err = "truncated BGZF block"; return false; }
        const size_t skip = 12 + xlen - 18;  // the rest of the extra field (none for the standard 6-byte BC subfield)
        const size_t clen = comp.size() - skip - 8;
        const uint8_t* tail = comp.data() + comp.size() - 8;
        const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536) { err = "BGZF block larger than 64 KiB"; return false; }
        block.resize(isize);
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2"; return false; }
        zs.next_in = comp.data() + skip; zs.avail_in = (uInt)clen;
        zs.next_out = block.data(); zs.avail_out = (uInt)isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize) { err = "inflate failed"; return false; }
        if (crc32(crc32(0L, Z_NULL, 0), block.data(), isize) != crc) { err = "BGZF CRC mismatch"; return false; }  // (bgzf.c checks it too)
        at = 0;
        return true;
    }
    // copy n bytes out, crossing block boundaries (what bgzf_read does)
    bool read(void* dst, size_t n) {
        uint8_t* d = (uint8_t*)dst;
        while (n) {
            if (at == block.size()) {
                do {
                    if (!next_block()) return false;
                } while (block.empty());
            }
            const size_t k = std::min(n, block.size() - at);
            memcpy(d, block.data() + at, k);
            d += k; at += k; n -= k;
        }
        return true;
    }
};

inline int32_t le32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t le16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

uint64_t name_id(const char* s, size_t n) {  // FNV-1a, 64 bit: mates share the name, so they share the id
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 1099511628211ull; }
    return h;
}

int aux_size(char t) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'd': return 8;
        default: return 0;
    }
}

struct Cols {
    std::vector<int32_t> tid, pos, mtid, mpos, isize, qlen, lib;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> bdqual;
    std::vector<uint64_t> name;
    void clear() { tid.clear(); pos.clear(); mtid.clear(); mpos.clear(); isize.clear(); qlen.clear(); lib.clear(); flag.clear(); bdqual.clear(); name.clear(); }
};

struct Loaded {
    std::vector<std::string> targets;
    Cols c;
};

// one decode pass over the file; only_tid >= 0 keeps that tid only (the "-o" reader, io/RegionLimitedBamReader.hpp:63-71)
bool decode_once(const char* path, void* oracle, int only_tid, Loaded& out, std::string& err) {
    Bgzf z;
    z.f = fopen(path, "rb");
    if (!z.f) { err = std::string("cannot open ") + path; return false; }
    std::vector<char> big(1 << 20);
    setvbuf(z.f, big.data(), _IOFBF, big.size());
    auto fail = [&](const std::string& m) { err = m.empty() ? "unexpected end of file" : m; fclose(z.f); return false; };
    uint8_t w[8];
    if (!z.read(w, 8) || memcmp(w, "BAM\1", 4) != 0) return fail(z.err.empty() ? "not a BAM file" : z.err);
    std::string text((size_t)le32(w + 4), '\0');
    if (!text.empty() && !z.read(&text[0], text.size())) return fail(z.err);
    if (!z.read(w, 4)) return fail(z.err);
    const int nref = le32(w);
    out.targets.clear();
    for (int i = 0; i < nref; ++i) {
        if (!z.read(w, 4)) return fail(z.err);
        std::string nm((size_t)le32(w), '\0');
        if (!z.read(&nm[0], nm.size()) || !z.read(w, 4)) return fail(z.err);
        if (!nm.empty() && nm.back() == '\0') nm.pop_back();
        out.targets.push_back(nm);
    }
    out.c.clear();
    std::vector<uint8_t> rec;  // (bam1_t::data: the record is copied out of the block buffer, as bam_read1 does)
    std::string rg;
    for (;;) {
        if (!z.read(w, 4)) {
            if (z.eof && z.err.empty()) break;
            return fail(z.err);
        }
        const int32_t bs = le32(w);
        if (bs < 32) return fail("bad record size");
        rec.resize((size_t)bs);
        if (!z.read(rec.data(), rec.size())) return fail(z.err);
        const uint8_t* p = rec.data();
        const int32_t tid = le32(p), pos = le32(p + 4);
        const unsigned l_qname = p[8], mapq = p[9];
        const unsigned n_cigar = le16(p + 12), flag = le16(p + 14);
        const int32_t l_qseq = le32(p + 16), mtid = le32(p + 20), mpos = le32(p + 24), isize = le32(p + 28);
        if ((flag & (0x100 | 0x800)) || tid < 0) continue;
        if (only_tid >= 0 && tid != only_tid) continue;
        const char* qname = (const char*)p + 32;
        const uint8_t* aux = p + 32 + l_qname + 4 * (size_t)n_cigar + ((size_t)l_qseq + 1) / 2 + (size_t)l_qseq;
        const uint8_t* end = p + bs;
        // bam_aux_get-style linear scans for "AM" (Alignment.cpp:12-23) and "RG" (:25-29)
        int bdqual = (int)mapq;
        rg.clear();
        for (const uint8_t* q = aux; q + 3 <= end;) {
            const char t0 = (char)q[0], t1 = (char)q[1], ty = (char)q[2];
            q += 3;
            if (ty == 'Z' || ty == 'H') {
                const uint8_t* e = (const uint8_t*)memchr(q, 0, (size_t)(end - q));
                if (!e) break;
                if (t0 == 'R' && t1 == 'G' && ty == 'Z') rg.assign((const char*)q, (size_t)(e - q));
                q = e + 1;
            } else if (ty == 'B') {
                if (q + 5 > end) break;
                const int es = aux_size((char)q[0]);
                const int32_t cnt = le32(q + 1);
                if (!es || cnt < 0 || (size_t)cnt * es > (size_t)(end - q - 5)) break;
                q += 5 + (size_t)cnt * es;
            } else {
                const int sz = aux_size(ty);
                if (!sz || q + sz > end) break;
                if (t0 == 'A' && t1 == 'M') {  // bam_aux2i, then the uint8_t truncation of determine_bdqual
                    int v = 0;
                    switch (ty) {
                        case 'c': v = (int8_t)q[0]; break;
                        case 'C': v = q[0]; break;
                        case 's': v = (int16_t)le16(q); break;
                        case 'S': v = le16(q); break;
                        case 'i': case 'I': v = le32(q); break;
                        default: v = 0;
                    }
                    bdqual = v & 0xFF;
                }
                q += sz;
            }
        }
        const size_t ln = l_qname ? strnlen(qname, l_qname) : 0;
        out.c.tid.push_back(tid); out.c.pos.push_back(pos); out.c.mtid.push_back(mtid); out.c.mpos.push_back(mpos);
        out.c.isize.push_back(isize); out.c.qlen.push_back(l_qseq); out.c.flag.push_back((uint16_t)flag);
        out.c.bdqual.push_back((uint8_t)bdqual);
        out.c.lib.push_back(oracle ? bdo_lib_of_readgroup(oracle, rg.c_str()) : 0);
        out.c.name.push_back(name_id(qname, ln));
    }
    fclose(z.f);
    return true;
}

thread_local std::string g_err;

}  // namespace

extern "C" {

const char* bdo_bam_error() { return g_err.c_str(); }

// Decode `path` `passes` times on the calling thread and hand the records of the last pass to the oracle as the stream
// of physical file `bam` (targets are set from this file's header when set_targets is non-zero: BamMerger.cpp:78 takes
// the first reader's).  seconds[0] = wall time of all decode passes.  Returns the number of records, -1 on error.
int64_t bdo_bam_load(void* oracle, int bam, const char* path, int only_tid, int passes, int set_targets, double* seconds) {
    Loaded L;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < (passes < 1 ? 1 : passes); ++i)
        if (!decode_once(path, oracle, only_tid, L, g_err)) return -1;
    if (seconds) seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (oracle) {
        if (set_targets) {
            std::vector<const char*> nm;
            for (auto const& s : L.targets) nm.push_back(s.c_str());
            bdo_set_targets(oracle, (int)nm.size(), nm.data());
        }
        bdo_set_stream(oracle, bam, (int64_t)L.c.tid.size(), L.c.tid.data(), L.c.pos.data(), L.c.mtid.data(), L.c.mpos.data(),
                       L.c.isize.data(), L.c.flag.data(), L.c.qlen.data(), L.c.bdqual.data(), L.c.lib.data(), L.c.name.data());
    }
    return (int64_t)L.c.tid.size();
}

// index of a target name in the file's header (-1 if absent), for "-o <chr>"
int bdo_bam_tid(const char* path, const char* name) {
    Bgzf z;
    z.f = fopen(path, "rb");
    if (!z.f) return -1;
    uint8_t w[8];
    int found = -1;
    if (z.read(w, 8) && memcmp(w, "BAM\1", 4) == 0) {
        std::string text((size_t)le32(w + 4), '\0');
        if ((text.empty() || z.read(&text[0], text.size())) && z.read(w, 4)) {
            const int nref = le32(w);
            for (int i = 0; i < nref && found < 0; ++i) {
                if (!z.read(w, 4)) break;
                std::string nm((size_t)le32(w), '\0');
                if (!z.read(&nm[0], nm.size()) || !z.read(w, 4)) break;
                if (!nm.empty() && nm.back() == '\0') nm.pop_back();
                if (nm == name) found = i;
            }
        }
    }
    fclose(z.f);
    return found;
}

}  // extern "C"
