// H2 -- read-level host replay of the region graph, for inputs in which a read name occurs more than twice.
//
// The GPU path (K4 / K6) and the aggregate walk (H1, bdx_walk.cpp) rest on "a name is seen at most twice": a pair either
// exists (both mates in accepted regions) or it does not, and a pair group is consumed whole.  The reference makes no such
// assumption -- it keeps appending region ids to the name's list and only the list reaching size two adds an edge
// (ReadRegionData.cpp:108-113); SvBuilder pairs whatever reads of the two regions share a name, first come first paired
// (SvBuilder.cpp:101-118); reads whose name is still waiting for a mate stay in their region and can be paired again
// (BreakDancer.cpp:357-368); cleared regions shorten the lists so that later sightings form new edges
// (ReadRegionData.cpp:152-175).  Merged BAMs with clashing read names are enough to get there.  K4 notices a third
// sighting (StageCounts::irregular) and bdx_run then replays everything behind the region cut here, one read at a time,
// with the reference's containers' semantics.  The region table, the prefix samples and the pass-1 statistics are the
// GPU's (K1-K3 do not depend on names); the SV assembly behind the pairing is shared with H1 (emit_sv).
//
// Replays: ReadRegionData.cpp:89-124 (add_region), :126-142 (is_region_final), :152-175 (clear_region), :177-199
// (collapse), :201-205 (region_reads_range); BreakDancer.cpp:254-259, 266-346 (flush cadence, build_connection), :348-368
// and :510-511 (process_sv around the shared part); SvBuilder.cpp:18-34, 101-118.
#include <algorithm>
#include <map>
#include <unordered_map>

#include "bdx_dev.h"
#include "bdx_walk.h"

namespace bdx {

namespace {

struct NameLists {  // _read_regions: name -> region ids in sighting order
    std::unordered_map<uint64_t, std::vector<int32_t>> m;
    bool has(uint64_t k) const { return m.find(k) != m.end(); }
};

struct ReadReplay {
    const ReadWalkInput& in;
    WalkResult& out;
    const WalkInput& w;
    NameLists names;
    std::map<int, std::map<int, int>> graph;        // UndirectedWeightedGraph<int,int>: ascending vertices, ascending neighbours
    std::vector<std::vector<uint32_t>> reads;        // per region: compact indices of the reads it still holds
    std::vector<uint8_t> exists;                     // region not cleared yet
    int64_t n_added = 0;                             // regions registered so far (ids 0 .. n_added-1)
    int max_readlen = 0;
    uint64_t seq = 0;                                // emission counter: this walk's output is already in the reference's order
    // scratch of process_sv
    std::unordered_map<uint64_t, uint32_t> waiting;  // SvBuilder::observed_reads: name -> read still waiting for its mate
    std::vector<uint64_t> to_free;
    std::vector<LibAcc> la;

    ReadReplay(const ReadWalkInput& i, WalkResult& o) : in(i), out(o), w(i.base) {}

    int flag_of(uint32_t j) const { return meta_flag(in.meta[j]); }

    void add_region(int r, uint32_t first, uint32_t n) {
        const HostRegion& R = w.regions[r];
        exists[r] = 1;
        for (uint32_t j = first; j < first + n; ++j) {
            std::vector<int32_t>& v = names.m[in.key[j]];
            v.push_back(r);
            if (v.size() == 2) {  // Graph.hpp:41-46
                ++graph[v[0]][v[1]];
                if (v[0] != v[1]) ++graph[v[1]][v[0]];
            }
        }
        const int valid = w.opts.chr_restricted ? (int)R.nonctx : (int)R.n;
        if (valid >= w.opts.min_read_pair) {
            reads[r].resize(n);
            for (uint32_t k = 0; k < n; ++k) reads[r][k] = first + k;
        }
        n_added = r + 1;
    }

    bool region_final(int r) const {
        if (!exists[r] || r == n_added - 1) return false;
        for (uint32_t j : reads[r]) {
            if (w.opts.chr_restricted && flag_of(j) == BDX_ARP_CTX) continue;
            auto f = names.m.find(in.key[j]);
            if (f == names.m.end() || f->second.size() != 2) return false;
        }
        return true;
    }

    void clear_region(int r) {
        for (uint32_t j : reads[r]) {
            auto f = names.m.find(in.key[j]);
            if (f == names.m.end()) continue;
            std::vector<int32_t>& v = f->second;
            v.erase(std::remove(v.begin(), v.end(), (int32_t)r), v.end());
            if (v.empty()) names.m.erase(f);
        }
        std::vector<uint32_t>().swap(reads[r]);
        exists[r] = 0;
    }

    void process_sv(int A, int B) {
        const int n = B >= 0 ? 2 : 1;
        const int nodes[2] = {A, B};
        int flag_counts[BDX_NUM_FLAGS] = {0};
        int num_pairs = 0;
        // per (flag, library): pairs and span sum, taken from the second-observed mate of every pair
        struct Acc { int flag, lib, rc, span; };
        std::vector<Acc> acc;
        waiting.clear();
        to_free.clear();
        const size_t sup0 = in.support ? in.support->size() : 0;
        for (int i = 0; i < n; ++i) {
            for (uint32_t j : reads[nodes[i]]) {
                const uint64_t k = in.key[j];
                if (!names.has(k)) continue;  // region_reads_range only hands out reads whose name is still known
                auto ins = waiting.emplace(k, j);
                if (ins.second) continue;
                const int f = flag_of(j), lib = meta_lib(in.meta[j]);
                ++flag_counts[f];
                ++num_pairs;
                size_t q = 0;
                while (q < acc.size() && !(acc[q].flag == f && acc[q].lib == lib)) ++q;
                if (q == acc.size()) acc.push_back(Acc{f, lib, 0, 0});
                ++acc[q].rc;
                acc[q].span += in.isize[j];
                to_free.push_back(k);
                if (in.support) { in.support->push_back(j); in.support->push_back(ins.first->second); }
                waiting.erase(ins.first);
            }
        }
        // reads whose name is no longer waiting (paired here, or unknown) leave their region (BreakDancer.cpp:357-368)
        for (int i = 0; i < n; ++i) {
            std::vector<uint32_t>& v = reads[nodes[i]];
            size_t wr = 0;
            for (uint32_t j : v)
                if (names.has(in.key[j]) && waiting.find(in.key[j]) != waiting.end()) v[wr++] = j;
            v.resize(wr);
        }
        auto reject = [&] { if (in.support) in.support->resize(sup0); };
        if (num_pairs < w.opts.min_read_pair) return reject();
        int flag = BDX_NA;
        {
            int best = 0;
            for (int f = 0; f < BDX_NUM_FLAGS; ++f)
                if (flag_counts[f] > flag_counts[best]) best = f;
            if (flag_counts[best] > 0) flag = best;
        }
        if (flag_counts[flag] < w.opts.min_read_pair) return reject();
        la.clear();
        for (const Acc& a : acc)
            if (a.flag == flag) la.push_back(LibAcc{a.lib, a.rc, a.span});
        std::sort(la.begin(), la.end(), [](const LibAcc& x, const LibAcc& y) { return x.lib < y.lib; });
        emit_sv(w, out, A, B, flag_counts, flag, la.data(), (int)la.size(), max_readlen, 0u, seq++);
        if (in.support_off) in.support_off->push_back((uint32_t)in.support->size());
        for (uint64_t k : to_free) names.m.erase(k);  // BreakDancer.cpp:510-511 (only a candidate that passed the gates gets here)
    }

    void build_connection() {
        std::vector<int> active;
        for (auto const& kv : graph) active.push_back(kv.first);
        std::vector<int> tails, newtails;
        auto ii = graph.begin();
        while (ii != graph.end()) {
            tails.assign(1, ii->first);
            bool need_inc = true;
            while (!tails.empty()) {
                newtails.clear();
                for (int tail : tails) {
                    if (!exists[tail]) continue;
                    auto found = graph.find(tail);
                    if (found == graph.end()) continue;
                    std::map<int, int>& nb = found->second;
                    for (auto it = nb.begin(); it != nb.end();) {
                        const int s1 = it->first, weight = it->second;
                        it = nb.erase(it);
                        if (weight < w.opts.min_read_pair || !exists[s1]) continue;
                        if (tail != s1) {
                            auto back = graph.find(s1);
                            if (back != graph.end()) back->second.erase(tail);
                            newtails.push_back(s1);
                            process_sv(std::min(s1, tail), std::max(s1, tail));
                        } else {
                            newtails.push_back(s1);
                            process_sv(s1, -1);
                        }
                    }
                    // once the start vertex is gone the reference compares against end() (undefined behaviour); like H1 and the
                    // oracle this takes the outcome "not equal"
                    if (ii != graph.end() && tail == ii->first) {
                        ii = graph.erase(ii);
                        need_inc = false;
                    } else {
                        graph.erase(tail);
                    }
                }
                tails.swap(newtails);
            }
            if (need_inc) ++ii;
        }
        for (int r : active)
            if (region_final(r)) clear_region(r);
        graph.clear();
    }

    void run() {
        const int64_t NR = (int64_t)w.nregions;
        reads.assign((size_t)NR, {});
        exists.assign((size_t)NR, 0);
        if (!w.any_anomalous) return;
        const int64_t period = std::max<int64_t>(1, (int64_t)w.opts.buffer_size + 1);
        auto registered = [&](int r) {  // flush cadence: every buffer_size+1 accepted regions (BreakDancer.cpp:254-259)
            if ((int64_t)(r + 1) % period == 0) {
                max_readlen = w.regions[r].maxq;  // the stale _max_readlen of the candidate that closes at this flush
                build_connection();
            }
        };
        if (in.phantom) {  // the read-less region 0 of a negative -s
            exists[0] = 1;
            n_added = 1;
            registered(0);
        }
        uint32_t j = 0;
        while (j < in.n_reads) {
            const int32_t r = in.region_of[j];
            if (r < 0) {  // a read of a rejected candidate: its name is forgotten (ReadRegionData.cpp:177-199)
                names.m.erase(in.key[j]);
                ++j;
                continue;
            }
            const uint32_t n = w.regions[r].n;
            add_region(r, j, n);
            registered(r);
            j += n;
        }
        max_readlen = w.last_maxq;
        build_connection();
    }
};

}  // namespace

void read_level_walk(const ReadWalkInput& in, WalkResult& out) {
    if (in.support_off) in.support_off->assign(1, 0u);
    if (in.support) in.support->clear();
    ReadReplay r(in, out);
    r.run();
    out.n_groups = 0;
}

}  // namespace bdx
