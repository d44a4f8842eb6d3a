// H1 -- host replay of build_connection / process_sv over aggregated pair groups.
//
// Replaces the control flow of (reference file:line under src/lib/breakdancer):
//   BreakDancer.cpp:254-259, 536-541   flush cadence (every buffer_size+1 accepted regions, then the end)
//   BreakDancer.cpp:266-346            build_connection: ascending vertices, BFS frontier, edge consumption
//   BreakDancer.cpp:348-497            process_sv: gates, breakpoints, copy number, size, score inputs
//   SvBuilder.cpp:18-118               pairing / dominant flag / positions / copy number / allele frequency
//   ReadRegionData.cpp:70-78           accumulate_reads_between_regions (telescoped to two prefix samples)
//
// Why aggregates suffice: process_sv({A,B}) pairs *all* still-present reads of A then B by name, so it
// consumes exactly the pair groups (A,A), (A,B), (B,B) that are still alive, and a group is always consumed
// whole.  The scalar float32 arithmetic below keeps the reference's operation order.
#include "bdx_walk.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <unordered_map>

namespace bdx {

namespace {

struct Group {
    uint32_t lo, hi;
    uint32_t weight = 0;  // pairs = the reference's edge weight
    bool alive = true;
    std::vector<GroupPart> parts;
};

struct Walker {
    const WalkInput& in;
    WalkResult& out;
    std::vector<Group> groups;
    std::unordered_map<uint64_t, uint32_t> gindex;
    int max_readlen = 0;

    Walker(const WalkInput& i, WalkResult& o) : in(i), out(o) {}

    static uint64_t gkey(uint32_t lo, uint32_t hi) { return ((uint64_t)lo << 32) | hi; }

    bool stored(uint32_t r) const {  // ReadRegionData.cpp:118-121
        const HostRegion& R = (*in.regions)[r];
        const int valid = in.opts.chr_restricted ? (int)R.nonctx : (int)R.n;
        return valid >= in.opts.min_read_pair;
    }

    void build_groups() {
        for (const GroupPart& p : *in.parts) {
            const uint64_t k = gkey(p.lo, p.hi);
            auto it = gindex.find(k);
            uint32_t gi;
            if (it == gindex.end()) {
                gi = (uint32_t)groups.size();
                gindex.emplace(k, gi);
                groups.emplace_back();
                groups.back().lo = p.lo;
                groups.back().hi = p.hi;
            } else {
                gi = it->second;
            }
            Group& g = groups[gi];
            g.weight += p.pairs;
            bool merged = false;
            for (GroupPart& q : g.parts)
                if (q.flag == p.flag && q.lib == p.lib) { q.pairs += p.pairs; q.sum_isize += p.sum_isize; merged = true; break; }
            if (!merged) g.parts.push_back(p);
        }
        out.n_groups = (uint32_t)groups.size();
    }

    Group* alive_group(uint32_t lo, uint32_t hi) {
        auto it = gindex.find(gkey(lo, hi));
        if (it == gindex.end()) return nullptr;
        Group& g = groups[it->second];
        if (!g.alive) return nullptr;
        if (!stored(lo) || !stored(hi)) return nullptr;  // mates of an unstored region never complete a pair
        return &g;
    }

    void process_sv(const int* snodes, int n) {
        const std::vector<HostRegion>& R = *in.regions;
        const bdx_opts& o = in.opts;
        const int A = snodes[0], B = n == 2 ? snodes[1] : -1;
        int num_pairs = 0;
        int flag_counts[BDX_NUM_FLAGS] = {0};
        std::map<int, int> rc[BDX_NUM_FLAGS], span[BDX_NUM_FLAGS];
        Group* gs[3] = {alive_group(A, A), n == 2 ? alive_group(A, B) : nullptr, n == 2 ? alive_group(B, B) : nullptr};
        for (Group* g : gs) {
            if (!g) continue;
            for (const GroupPart& p : g->parts) {
                flag_counts[p.flag] += (int)p.pairs;
                rc[p.flag][p.lib] += (int)p.pairs;
                span[p.flag][p.lib] += (int)p.sum_isize;
                num_pairs += (int)p.pairs;
            }
            g->alive = false;  // paired reads leave both regions before any gate (BreakDancer.cpp:363-368)
        }
        if (num_pairs < o.min_read_pair) return;
        int flag = BDX_NA;
        {
            int best = 0;
            for (int f = 0; f < BDX_NUM_FLAGS; ++f)
                if (flag_counts[f] > flag_counts[best]) best = f;
            if (flag_counts[best] > 0) flag = best;
        }
        if (flag_counts[flag] < o.min_read_pair) return;

        int chr[2], pos[2], fwd[2], rev[2];
        const HostRegion& ra = R[A];
        chr[0] = ra.tid; pos[0] = ra.start; pos[1] = ra.end;
        fwd[0] = (int)(ra.n - ra.rev); rev[0] = (int)ra.rev;
        if (n == 2) {
            const HostRegion& rb = R[B];
            fwd[1] = (int)(rb.n - rb.rev); rev[1] = (int)rb.rev;
            if (flag == BDX_ARP_RF) pos[1] = rb.end + max_readlen - 5;
            else if (flag == BDX_ARP_FF) { pos[0] = pos[1]; pos[1] = rb.end + max_readlen - 5; }
            else if (flag == BDX_ARP_RR) pos[1] = rb.start;
            else { pos[0] = pos[1]; pos[1] = rb.start; }
            chr[1] = rb.tid;
        } else {
            fwd[1] = fwd[0]; rev[1] = rev[0]; chr[1] = ra.tid; pos[1] = ra.end;
        }

        // normal reads between the regions: proper reads after A's last read up to and including B's first
        const int cn_begin = (int)out.cn_key.size();
        float cn_sum = 0.0f;
        int nkeys_present = 0;
        if (n == 2) {
            for (int k = 0; k < in.nkeys; ++k) {
                const uint32_t cnt = in.r_pk[(size_t)B * 2 * in.nkeys + k] - in.r_pk[(size_t)A * 2 * in.nkeys + in.nkeys + k];
                if (cnt == 0) continue;
                const float cn = cnt / (in.key_density[k] * float(pos[1] - pos[0])) * 2.0f;
                out.cn_key.push_back(k);
                out.cn_value.push_back(cn);
                cn_sum += cn;
                ++nkeys_present;
            }
        }
        cn_sum /= 2.0f * (size_t)nkeys_present;
        const float allele_frequency = 1 - cn_sum;

        if (flag != BDX_ARP_RF && flag != BDX_ARP_RR && pos[0] + max_readlen - 5 < pos[1]) pos[0] += max_readlen - 5;

        float diff = 0;
        for (auto const& kv : rc[flag])
            diff += float(span[flag][kv.first]) - float(kv.second) * in.libs[kv.first].mean_insertsize;
        const int diffspan = int(diff / float(flag_counts[flag]) + 0.5);

        int total_region_size = ra.end - ra.start + 1;
        if (n == 2) total_region_size += R[B].end - R[B].start + 1;

        HostSv hs;
        bdx_sv& sv = hs.sv;
        for (int i = 0; i < 2; ++i) { sv.chr[i] = chr[i]; sv.pos[i] = pos[i] + 1; sv.fwd[i] = fwd[i]; sv.rev[i] = rev[i]; }
        sv.flag = flag; sv.size = diffspan; sv.score = 0; sv.num_reads = flag_counts[flag]; sv.printed = 0;
        sv.region[0] = A; sv.region[1] = B;
        sv.lib_begin = (int)out.lib_index.size(); sv.lib_count = (int)rc[flag].size();
        sv.cn_begin = cn_begin; sv.cn_count = nkeys_present;
        sv.allele_frequency = allele_frequency; sv.logp = 0;
        hs.term_begin = (uint32_t)out.terms.size();
        hs.term_count = (uint32_t)rc[flag].size();
        for (auto const& kv : rc[flag]) {
            out.lib_index.push_back(kv.first);
            out.lib_pairs.push_back(kv.second);
            const uint32_t nflag = in.hist[(size_t)kv.first * BDX_NUM_FLAGS + flag];
            double lambda = double(total_region_size) * (double(nflag) / double(in.covered_ref_len));
            lambda = std::max(1.0e-10, lambda);
            out.terms.push_back(SvTerm{lambda, kv.second});
        }
        out.svs.push_back(hs);
    }

    // BreakDancer.cpp:266-346 over the edges whose later region was added since the previous flush
    void flush(std::map<int, std::map<int, int>>& graph) {
        const int mrp = in.opts.min_read_pair;
        auto ii = graph.begin();
        while (ii != graph.end()) {
            std::vector<int> tails{ii->first};
            bool need_inc = true;
            while (!tails.empty()) {
                std::vector<int> newtails;
                for (int tail : tails) {
                    auto found = graph.find(tail);
                    if (found == graph.end()) continue;
                    auto& gt = found->second;
                    auto it = gt.begin();
                    while (it != gt.end()) {
                        const int s1 = it->first, nlinks = it->second;
                        gt.erase(it++);
                        if (nlinks < mrp) continue;
                        int snodes[2];
                        int n;
                        if (tail != s1) {
                            auto a = graph.find(s1);
                            if (a != graph.end()) a->second.erase(tail);
                            snodes[0] = std::min(s1, tail);
                            snodes[1] = std::max(s1, tail);
                            n = 2;
                        } else {
                            snodes[0] = s1;
                            n = 1;
                        }
                        newtails.push_back(s1);
                        process_sv(snodes, n);
                    }
                    // `ii` may already be end() here; the reference dereferences it anyway (UB that in practice
                    // compares against a non-vertex word), so it is treated as "not the start vertex"
                    if (ii != graph.end() && tail == ii->first) {
                        graph.erase(ii++);
                        need_inc = false;
                    } else {
                        graph.erase(tail);
                    }
                }
                tails.swap(newtails);
            }
            if (need_inc) ++ii;
        }
        graph.clear();
    }

    void run() {
        build_groups();
        const std::vector<HostRegion>& R = *in.regions;
        const uint32_t NR = (uint32_t)R.size();
        if (!in.any_anomalous) return;
        // groups ordered by the region that completed them
        std::vector<uint32_t> order(groups.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return groups[a].hi != groups[b].hi ? groups[a].hi < groups[b].hi : groups[a].lo < groups[b].lo;
        });
        const int64_t period = std::max<int64_t>(1, (int64_t)in.opts.buffer_size + 1);
        size_t next = 0;
        std::map<int, std::map<int, int>> graph;
        auto add_edges_upto = [&](uint32_t last) {
            while (next < order.size() && groups[order[next]].hi <= last) {
                const Group& g = groups[order[next]];
                graph[(int)g.lo][(int)g.hi] += (int)g.weight;
                if (g.lo != g.hi) graph[(int)g.hi][(int)g.lo] += (int)g.weight;
                ++next;
            }
        };
        for (uint32_t r = 0; r < NR; ++r) {
            if ((int64_t)(r + 1) % period != 0) continue;
            add_edges_upto(r);
            max_readlen = R[r].maxq;  // stale _max_readlen: the value of the candidate closing at this flush (Q5)
            flush(graph);
        }
        if (NR) add_edges_upto(NR - 1);
        max_readlen = in.last_maxq;
        flush(graph);
    }
};

double chisq_upper_tail_int(int half_df, double x) {  // Q(n, x/2 -> here x already halved) = e^-x sum_{i<n} x^i / i!
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < half_df; ++i) {
        term *= x / i;
        sum += term;
    }
    return std::exp(-x) * sum;
}

}  // namespace

void greedy_walk(const WalkInput& in, WalkResult& out) {
    Walker w(in, out);
    w.run();
}

// BreakDancer.cpp:56-84 (Kahan-compensated sum of the per-library log tails, optional Fisher) and :459-465
void finish_scores(const WalkInput& in, const std::vector<double>& log_tail, WalkResult& out, uint32_t* n_printed) {
    uint32_t printed = 0;
    for (HostSv& hs : out.svs) {
        double logpvalue = 0.0, err = 0.0;
        for (uint32_t i = 0; i < hs.term_count; ++i) {
            const double tmp_a = log_tail[hs.term_begin + i] - err;
            const double tmp_b = logpvalue + tmp_a;
            err = (tmp_b - logpvalue) - tmp_a;
            logpvalue = tmp_b;
        }
        if (in.opts.fisher && logpvalue < 0) {
            const double x = -2 * logpvalue;
            if (std::isfinite(x)) {  // Boost's chi_squared cdf throws on a non-finite argument; the reference keeps logp
                const double fisherP = chisq_upper_tail_int((int)hs.term_count, x / 2);
                logpvalue = fisherP > std::exp(-99.0) ? std::log(fisherP) : -99;
            }
        }
        const double phred_tmp = -10 * logpvalue / std::log(10);
        // int(NaN) is INT_MIN on x86-64 (cvttsd2si): the silent multi-library drop the reference exhibits (Q15)
        int phred;
        if (phred_tmp > 99) phred = 99;
        else if (std::isnan(phred_tmp + 0.5) || phred_tmp + 0.5 >= 2147483648.0 || phred_tmp + 0.5 <= -2147483649.0) phred = INT32_MIN;
        else phred = int(phred_tmp + 0.5);
        hs.sv.logp = logpvalue;
        hs.sv.score = phred;
        hs.sv.printed = phred > in.opts.score_threshold;
        printed += hs.sv.printed;
    }
    *n_printed = printed;
}

}  // namespace bdx
