#include "producer.h"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>
#include <memory>
#include <queue>
#include <stdexcept>
#include <unordered_map>

#include "bam_reader.h"

namespace bdhost {

bdx_batch ReadStream::batch() const {
    bdx_batch b;
    b.tid = tid.data(); b.pos = pos.data(); b.mtid = mtid.data(); b.mpos = mpos.data(); b.isize = isize.data();
    b.flag = flag.data(); b.qlen = qlen.data(); b.mapq = mapq.data(); b.lib = lib.data(); b.bam = bam.data();
    b.name_key = name_key.data();
    b.n = size();
    return b;
}

namespace {

// samtools region strings as the reference accepts them for -o (bam_aux.c:107-160 bam_parse_region): "name",
// "name:beg" or "name:beg-end" (1-based, commas allowed); a name that itself contains ':' is tried as a whole
bool parse_region(const BamReader& rd, const std::string& str, int& tid, int& beg, int& end) {
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

void produce(const BamConfig& cfg, const std::string& chr, int threads, ReadStream& out) {
    {   // reserve from the compressed sizes (~60-100 B per record) so that the column vectors do not regrow all the way
        size_t bytes = 0;
        for (auto const& f : cfg.bam_files()) {
            struct stat st;
            if (stat(f.c_str(), &st) == 0) bytes += (size_t)st.st_size;
        }
        const size_t guess = bytes / 100 + 1024;
        out.tid.reserve(guess); out.pos.reserve(guess); out.mtid.reserve(guess); out.mpos.reserve(guess); out.isize.reserve(guess);
        out.flag.reserve(guess); out.qlen.reserve(guess); out.mapq.reserve(guess); out.lib.reserve(guess); out.bam.reserve(guess);
        out.name_key.reserve(guess);
    }
    merge_streams(cfg, chr, threads, &out.targets, [&](uint64_t, const BamRecord& r, int bam_index, uint8_t lib) {
        out.tid.push_back(r.tid); out.pos.push_back(r.pos); out.mtid.push_back(r.mtid); out.mpos.push_back(r.mpos);
        out.isize.push_back(r.isize); out.flag.push_back(r.flag);
        out.qlen.push_back((uint16_t)(r.l_qseq > 65535 ? 65535 : (r.l_qseq < 0 ? 0 : r.l_qseq)));
        out.mapq.push_back(r.bdqual);
        out.lib.push_back(lib);
        out.bam.push_back((uint8_t)bam_index);
        out.name_key.push_back(r.name_key);
    });
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
