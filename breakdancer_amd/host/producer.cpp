#include "producer.h"

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

struct Stream {
    std::unique_ptr<BamReader> rd;
    int bam_index = 0;
    int only_tid = -1;
    BamRecord cur{};
    bool valid = false;
    // reader filter of the reference: primary (not secondary / supplementary) and tid >= 0
    // (io/AlignmentFilter.hpp:24-34, io/BamIo.cpp:11-18); -o keeps one tid (RegionLimitedBamReader.hpp:63-71)
    bool advance() {
        while (rd->next(cur)) {
            if (cur.flag & (0x100 | 0x800)) continue;
            if (cur.tid < 0) continue;
            if (only_tid >= 0 && cur.tid != only_tid) continue;
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

}  // namespace

void produce(const BamConfig& cfg, const std::string& chr, int threads, ReadStream& out) {
    std::vector<std::unique_ptr<Stream>> streams;
    for (size_t b = 0; b < cfg.num_bams(); ++b) {
        std::unique_ptr<Stream> s(new Stream);
        s->rd.reset(new BamReader(cfg.bam_files()[b], threads));
        s->bam_index = (int)b;
        if (!chr.empty()) {
            s->only_tid = s->rd->tid_of(chr);
            if (s->only_tid < 0)
                throw std::runtime_error("Failed to parse bam region '" + chr + "' in file " + cfg.bam_files()[b] + ". ");
        }
        streams.push_back(std::move(s));
    }
    if (streams.empty()) throw std::runtime_error("BamMerger created with no input streams!");
    out.targets = streams[0]->rd->target_names();

    // read-group string -> library index (io/BamConfig.hpp:62-72), cached per distinct RG value
    std::unordered_map<std::string, uint8_t> rg_cache;
    for (auto const& kv : cfg.readgroup_index()) rg_cache[kv.first] = (uint8_t)kv.second;
    const uint8_t fallback = (uint8_t)cfg.fallback_library();

    std::priority_queue<Stream*, std::vector<Stream*>, StreamGreater> pq;
    for (auto& s : streams)
        if (s->advance()) pq.push(s.get());
    std::string rgtmp;
    while (!pq.empty()) {
        Stream* s = pq.top();
        pq.pop();
        const BamRecord& r = s->cur;
        out.tid.push_back(r.tid); out.pos.push_back(r.pos); out.mtid.push_back(r.mtid); out.mpos.push_back(r.mpos);
        out.isize.push_back(r.isize); out.flag.push_back(r.flag);
        out.qlen.push_back((uint16_t)(r.l_qseq > 65535 ? 65535 : (r.l_qseq < 0 ? 0 : r.l_qseq)));
        out.mapq.push_back(r.bdqual);
        uint8_t lib = fallback;
        rgtmp.assign(r.rg ? r.rg : "", r.rg ? r.l_rg : 0);
        auto it = rg_cache.find(rgtmp);
        if (it != rg_cache.end()) lib = it->second;
        out.lib.push_back(lib);
        out.bam.push_back((uint8_t)s->bam_index);
        out.name_key.push_back(hash_name(r.qname, r.l_qname));
        if (s->advance()) pq.push(s);
    }
}

}  // namespace bdhost
