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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace bdx {

// Flat, allocation-light replay.  Groups are kept sorted by (hi, lo); a flush covers the groups whose
// later region `hi` was added since the previous flush, so a vertex's adjacency in ascending neighbour order
// is [groups with hi == v, ascending lo (the self group last)] followed by [groups with lo == v, ascending hi],
// the latter threaded through `next_fwd` in insertion order.  "Erased from the graph" == visited.
struct Group {
    uint32_t lo, hi, weight;
    uint32_t pbeg, pcnt;
    int32_t next_fwd;
    bool alive, edge_done;
};

struct OldVertex {  // endpoint from an earlier flush: only has edges to regions of the current window
    uint32_t id;
    int32_t head, tail;
    bool visited;
};

struct KeyIdx {  // sort key (hi, lo, flag, lib) of a part and where it stands in the input
    uint64_t key;
    uint32_t idx;
};

struct WalkScratch {
    std::vector<KeyIdx> ka, kb;
    std::vector<GroupPart> parts;  // sorted by (hi, lo, flag, lib), duplicates merged
    std::vector<Group> groups;
    std::vector<uint32_t> ghi;     // groups with hi == r are [ghi[r], ghi[r+1])
    std::vector<uint8_t> stored_;
    std::vector<uint32_t> cnt, cur;
    std::vector<LibAcc> lib_acc_;
    std::vector<int32_t> win_head, win_tail;
    std::vector<uint8_t> win_visited;
    std::vector<OldVertex> olds;
    std::vector<int> tails, newtails;
};
WalkScratch* walk_scratch_new() { return new WalkScratch; }
void walk_scratch_free(WalkScratch* s) { delete s; }

// Second half of process_sv (BreakDancer.cpp:370-497) and of the SvBuilder constructor (SvBuilder.cpp:36-66): everything that
// follows the pairing and the two support gates.  flag_counts / la (pairs and span sums of the dominant flag per library,
// ascending library index) come from the pair-group aggregates (H1) or from the reads themselves (bdx_walk_reads.cpp).
void emit_sv(const WalkInput& in, WalkResult& out, int A, int B, const int* flag_counts, int flag, const LibAcc* la, int nacc,
             int max_readlen, uint32_t grp_mask, uint64_t cur_key) {
    const HostRegion* R = in.regions;
    const int n = B >= 0 ? 2 : 1;

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
    for (int i = 0; i < nacc; ++i) diff += float(la[i].span) - float(la[i].rc) * in.libs[la[i].lib].mean_insertsize;
    const int diffspan = int(diff / float(flag_counts[flag]) + 0.5);

    int total_region_size = ra.end - ra.start + 1;
    if (n == 2) total_region_size += R[B].end - R[B].start + 1;

    HostSv hs;
    bdx_sv& sv = hs.sv;
    for (int i = 0; i < 2; ++i) { sv.chr[i] = chr[i]; sv.pos[i] = pos[i] + 1; sv.fwd[i] = fwd[i]; sv.rev[i] = rev[i]; }
    sv.flag = flag; sv.size = diffspan; sv.score = 0; sv.num_reads = flag_counts[flag]; sv.printed = 0;
    sv.region[0] = A; sv.region[1] = B;
    sv.lib_begin = (int)out.lib_index.size(); sv.lib_count = nacc;
    sv.cn_begin = cn_begin; sv.cn_count = nkeys_present;
    sv.allele_frequency = allele_frequency; sv.logp = 0;
    hs.grp_mask = grp_mask;
    hs.start = (uint32_t)(cur_key & 0xffffffffu);
    for (int i = 0; i < nacc; ++i) {
        out.lib_index.push_back(la[i].lib);
        out.lib_pairs.push_back(la[i].rc);
        const uint32_t nflag = in.hist[(size_t)la[i].lib * BDX_NUM_FLAGS + flag];
        double lambda = double(total_region_size) * (double(nflag) / double(in.covered_ref_len));
        lambda = std::max(1.0e-10, lambda);
        out.terms.push_back(SvTerm{lambda, la[i].rc});
    }
    out.svs.push_back(hs);
    out.sv_key.push_back(cur_key);
}

namespace {

struct Walker {
    const WalkInput& in;
    WalkResult& out;
    WalkScratch& S;
    std::vector<GroupPart>& parts;
    std::vector<Group>& groups;
    std::vector<uint32_t>& ghi;
    std::vector<uint8_t>& stored_;
    std::vector<LibAcc>& lib_acc_;
    std::vector<int32_t>&win_head, &win_tail;
    std::vector<uint8_t>& win_visited;
    std::vector<OldVertex>& olds;
    std::vector<int>&tails, &newtails;
    int max_readlen = 0;
    uint64_t cur_key = 0;  // order key of the traversal in progress

    Walker(const WalkInput& i, WalkScratch& s, WalkResult& o)
        : in(i), out(o), S(s), parts(s.parts), groups(s.groups), ghi(s.ghi), stored_(s.stored_), lib_acc_(s.lib_acc_),
          win_head(s.win_head), win_tail(s.win_tail), win_visited(s.win_visited), olds(s.olds), tails(s.tails), newtails(s.newtails) {}

    void build_groups() {
        static const bool prof = getenv("BDX_WALK_PROFILE") != nullptr;   // (tracing on stderr; read once per process)
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        const auto b0 = tnow();
        const std::vector<GroupPart>& src = *in.parts;
        const uint32_t NR = (uint32_t)in.nregions;
        auto less = [](const GroupPart& a, const GroupPart& b) {
            if (a.lo != b.lo) return a.lo < b.lo;
            if (a.flag != b.flag) return a.flag < b.flag;
            return a.lib < b.lib;
        };
        auto b1 = b0;
        if (src.size() * 8 < (size_t)NR && src.size() < 64) {
            // few parts (the usual share of the host: a handful of components): a comparison sort; the passes over all regions
            // below cost ~10 us at 12 k regions, on the path between the device's walk and the launch of the table kernels
            parts.assign(src.begin(), src.end());
            std::sort(parts.begin(), parts.end(), [&](const GroupPart& a, const GroupPart& b) { return a.hi != b.hi ? a.hi < b.hi : less(a, b); });
            b1 = tnow();
        } else if (src.size() * 8 < (size_t)NR) {
            // some thousand parts among many more regions (rank 0's share of a sharded genome: the components that span ranks): an LSD
            // radix sort of (packed key, index) pairs, a byte at a time, bytes in which no two keys differ skipped -- half the time
            // of the comparison sort at 13 k parts.  (Parts with equal keys are merged below: their order does not matter.)
            const size_t n = src.size();
            std::vector<KeyIdx>&ka = S.ka, &kb = S.kb;
            ka.resize(n); kb.resize(n);
            uint64_t all_or = 0, all_and = ~0ull;
            for (size_t i = 0; i < n; ++i) {
                const GroupPart& p = src[i];
                const uint64_t k = ((uint64_t)p.hi << 38) | ((uint64_t)p.lo << 12) | ((uint64_t)p.flag << 8) | (uint64_t)p.lib;   // (region ids are 26-bit)
                ka[i] = KeyIdx{k, (uint32_t)i};
                all_or |= k; all_and &= k;
            }
            const uint64_t varying = all_or ^ all_and;
            KeyIdx *from = ka.data(), *to = kb.data();
            for (int shift = 0; shift < 64; shift += 8) {
                if (!((varying >> shift) & 0xFFu)) continue;
                uint32_t c[257] = {0};
                for (size_t i = 0; i < n; ++i) ++c[((from[i].key >> shift) & 0xFFu) + 1];
                for (int dg = 0; dg < 256; ++dg) c[dg + 1] += c[dg];
                for (size_t i = 0; i < n; ++i) to[c[(from[i].key >> shift) & 0xFFu]++] = from[i];
                std::swap(from, to);
            }
            parts.resize(n);
            for (size_t i = 0; i < n; ++i) parts[i] = src[from[i].idx];
            b1 = tnow();
        } else {
            // counting sort by hi, then tiny insertion sorts inside each hi bucket
            std::vector<uint32_t>& cnt = S.cnt;
            cnt.assign(NR + 2, 0);
            for (const GroupPart& p : src) ++cnt[p.hi + 1];
            for (uint32_t r = 0; r <= NR; ++r) cnt[r + 1] += cnt[r];
            parts.resize(src.size());
            {
                std::vector<uint32_t>& cur = S.cur;
                cur.assign(cnt.begin(), cnt.end() - 1);
                for (const GroupPart& p : src) parts[cur[p.hi]++] = p;
            }
            b1 = tnow();
            for (uint32_t r = 0; r < NR; ++r) {
                const uint32_t b = cnt[r], e = cnt[r + 1];
                for (uint32_t i = b + 1; i < e; ++i) {
                    GroupPart x = parts[i];
                    uint32_t j = i;
                    while (j > b && less(x, parts[j - 1])) { parts[j] = parts[j - 1]; --j; }
                    parts[j] = x;
                }
            }
        }
        const auto b2 = tnow();
        // merge duplicates (a group can straddle two K4 workgroups) and cut into groups
        ghi.resize(NR + 1);
        uint32_t next_hi = 0;  // ghi[0..next_hi) already filled
        size_t w = 0;
        groups.clear();
        groups.reserve(parts.size());
        for (size_t i = 0; i < parts.size(); ++i) {
            const GroupPart p = parts[i];
            if (w && parts[w - 1].hi == p.hi && parts[w - 1].lo == p.lo && parts[w - 1].flag == p.flag && parts[w - 1].lib == p.lib) {
                parts[w - 1].pairs += p.pairs;
                parts[w - 1].sum_isize += p.sum_isize;
                groups.back().weight += p.pairs;
                continue;
            }
            if (groups.empty() || groups.back().hi != p.hi || groups.back().lo != p.lo) {
                while (next_hi <= p.hi) ghi[next_hi++] = (uint32_t)groups.size();
                Group g;
                g.lo = p.lo; g.hi = p.hi; g.weight = 0; g.pbeg = (uint32_t)w; g.pcnt = 0; g.next_fwd = -1;
                g.alive = true; g.edge_done = false;
                groups.push_back(g);
            }
            parts[w++] = p;
            groups.back().weight += p.pairs;
            groups.back().pcnt++;
        }
        parts.resize(w);
        const auto b3 = tnow();
        while (next_hi <= NR) ghi[next_hi++] = (uint32_t)groups.size();
        out.n_groups = (uint32_t)groups.size();
        if (prof) fprintf(stderr, "[walk] build: count+scatter %.1f, insertion %.1f, merge %.1f, index+stored %.1f us\n", us(b0, b1), us(b1, b2),
                          us(b2, b3), us(b3, tnow()));
    }

    bool stored(uint32_t r) const {  // ReadRegionData.cpp:118-121
        const HostRegion& R = in.regions[r];
        const int valid = in.opts.chr_restricted ? (int)R.nonctx : (int)R.n;
        return valid >= in.opts.min_read_pair;
    }

    Group* alive_group(uint32_t lo, uint32_t hi) {
        for (uint32_t g = ghi[hi]; g < ghi[hi + 1]; ++g) {
            if (groups[g].lo != lo) continue;
            if (!groups[g].alive) return nullptr;
            if (!stored(lo) || !stored(hi)) return nullptr;  // mates of an unstored region never complete a pair
            return &groups[g];
        }
        return nullptr;
    }

    void process_sv(const int* snodes, int n) {
        const bdx_opts& o = in.opts;
        const int A = snodes[0], B = n == 2 ? snodes[1] : -1;
        int num_pairs = 0;
        int flag_counts[BDX_NUM_FLAGS] = {0};
        Group* gs[3] = {alive_group(A, A), n == 2 ? alive_group(A, B) : nullptr, n == 2 ? alive_group(B, B) : nullptr};
        for (Group* g : gs) {
            if (!g) continue;
            for (uint32_t i = 0; i < g->pcnt; ++i) {
                const GroupPart& p = parts[g->pbeg + i];
                flag_counts[p.flag] += (int)p.pairs;
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
        // per-library pairs / spans of the dominant flag, ascending library index (std::map order in the reference)
        std::vector<LibAcc>& la = lib_acc_;
        la.clear();
        for (Group* g : gs) {
            if (!g) continue;
            for (uint32_t i = 0; i < g->pcnt; ++i) {
                const GroupPart& p = parts[g->pbeg + i];
                if (p.flag != flag) continue;
                size_t k = 0;
                while (k < la.size() && la[k].lib < (int)p.lib) ++k;
                if (k < la.size() && la[k].lib == (int)p.lib) { la[k].rc += (int)p.pairs; la[k].span += (int)p.sum_isize; }
                else la.insert(la.begin() + k, LibAcc{(int)p.lib, (int)p.pairs, (int)p.sum_isize});
            }
        }
        const uint32_t grp_mask = (gs[0] ? 1u : 0u) | (gs[1] ? 2u : 0u) | (gs[2] ? 4u : 0u);
        emit_sv(in, out, A, B, flag_counts, flag, la.data(), (int)la.size(), max_readlen, grp_mask, cur_key);
    }

    // ---- one flush (BreakDancer.cpp:266-346) over the groups with hi in (prev, last] ----------------------------
    OldVertex* find_old(uint32_t id) {
        for (OldVertex& o : olds)
            if (o.id == id) return &o;
        return nullptr;
    }

    void try_edge(uint32_t gi, int tail, int s1) {
        Group& g = groups[gi];
        // an entry below the gate is skipped every time it is met (from either endpoint); one above it is
        // consumed from the side that reaches it first (erase_edge removes the reverse entry)
        if (g.edge_done || (int)g.weight < in.opts.min_read_pair) return;
        g.edge_done = true;
        int snodes[2];
        int n;
        if (tail != s1) { snodes[0] = std::min(s1, tail); snodes[1] = std::max(s1, tail); n = 2; }
        else { snodes[0] = s1; n = 1; }
        newtails.push_back(s1);
        process_sv(snodes, n);
    }

    void visit(int tail, int64_t prev) {
        if (tail > prev) {
            for (uint32_t gi = ghi[tail]; gi < ghi[tail + 1]; ++gi) try_edge(gi, tail, (int)groups[gi].lo);
            for (int32_t gi = win_head[tail - prev - 1]; gi >= 0; gi = groups[gi].next_fwd) try_edge((uint32_t)gi, tail, (int)groups[gi].hi);
            win_visited[tail - prev - 1] = 1;
        } else {
            OldVertex* ov = find_old((uint32_t)tail);
            for (int32_t gi = ov->head; gi >= 0; gi = groups[gi].next_fwd) try_edge((uint32_t)gi, tail, (int)groups[gi].hi);
            ov->visited = true;
        }
    }

    bool is_visited(int v, int64_t prev) {
        if (v > prev) return win_visited[v - prev - 1];
        OldVertex* ov = find_old((uint32_t)v);
        return !ov || ov->visited;
    }

    size_t prof_hist[8] = {0}, prof_old = 0;
    bool prof_on = false;
    void bfs_from(int v, int64_t prev) {
        tails.clear();
        tails.push_back(v);
        size_t nv = 0;
        const size_t sv0 = out.svs.size();
        while (!tails.empty()) {
            newtails.clear();
            for (size_t i = 0; i < tails.size(); ++i) {
                const int tail = tails[i];
                if (is_visited(tail, prev)) continue;
                visit(tail, prev);
                ++nv;
            }
            tails.swap(newtails);
        }
        if (prof_on && out.svs.size() > sv0) { prof_hist[std::min<size_t>(nv, 7)] += out.svs.size() - sv0; if (v <= prev) prof_old += out.svs.size() - sv0; }
    }

    void flush(int64_t prev, int64_t last) {
        if (last <= prev) return;
        const uint32_t g0 = ghi[prev + 1], g1 = ghi[last + 1];
        if (g0 == g1) return;
        const size_t wn = (size_t)(last - prev);
        win_head.assign(wn, -1);
        win_tail.assign(wn, -1);
        win_visited.assign(wn, 0);
        olds.clear();
        for (uint32_t gi = g0; gi < g1; ++gi) {
            Group& g = groups[gi];
            g.next_fwd = -1;
            if (g.lo == g.hi) continue;
            if ((int64_t)g.lo > prev) {
                const size_t i = (size_t)(g.lo - prev - 1);
                if (win_tail[i] < 0) win_head[i] = (int32_t)gi; else groups[win_tail[i]].next_fwd = (int32_t)gi;
                win_tail[i] = (int32_t)gi;
            } else {
                OldVertex* ov = find_old(g.lo);
                if (!ov) olds.push_back(OldVertex{g.lo, (int32_t)gi, (int32_t)gi, false});
                else { groups[ov->tail].next_fwd = (int32_t)gi; ov->tail = (int32_t)gi; }
            }
        }
        std::sort(olds.begin(), olds.end(), [](const OldVertex& a, const OldVertex& b) { return a.id < b.id; });
        const uint64_t window = (uint64_t)(prev + 1) / (uint64_t)std::max<int64_t>(1, (int64_t)in.opts.buffer_size + 1);
        for (size_t i = 0; i < olds.size(); ++i)
            if (!olds[i].visited) {
                cur_key = sv_order_key(window, true, olds[i].id);
                bfs_from((int)olds[i].id, prev);
            }
        for (int64_t v = prev + 1; v <= last; ++v) {
            const size_t i = (size_t)(v - prev - 1);
            if (win_visited[i]) continue;
            if (ghi[v] == ghi[v + 1] && win_head[i] < 0) continue;  // not a vertex of this flush's graph
            cur_key = sv_order_key(window, false, (uint32_t)v);
            bfs_from((int)v, prev);
        }
    }

    void run() {
        static const bool prof = getenv("BDX_WALK_PROFILE") != nullptr;
        prof_on = prof;
        const auto tp0 = std::chrono::steady_clock::now();
        build_groups();
        if (prof) fprintf(stderr, "[walk] build_groups %.1f us (parts %zu groups %zu regions %zu)\n",
                          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp0).count(), in.parts->size(),
                          groups.size(), in.nregions);
        const HostRegion* R = in.regions;
        const int64_t NR = (int64_t)in.nregions;
        if (!in.any_anomalous) return;
        // The region table usually sits where the device has just written it (pinned memory, or a fresh copy): every record the walk
        // touches is a cache miss of its own, and the walk's chain of look-ups pays them one after the other -- 0.4 us per SV when rank 0
        // walks the cross-rank components of a sharded run.  The records (and prefix samples) of every group's two regions are requested
        // up front, many at a time.
        if (parts.size() <= (1u << 20)) {
            const size_t row = (size_t)2 * in.nkeys;
            for (const GroupPart& p : parts) {
                __builtin_prefetch(&R[p.lo]); __builtin_prefetch(&R[p.hi]);
                if (row) { __builtin_prefetch(&in.r_pk[(size_t)p.lo * row]); __builtin_prefetch(&in.r_pk[(size_t)p.hi * row]); }
            }
        }
        const int64_t period = std::max<int64_t>(1, (int64_t)in.opts.buffer_size + 1);
        int64_t prev = -1;
        for (int64_t r = period - 1; r < NR; r += period) {
            if (ghi[prev + 1] != ghi[r + 1]) {   // (a window without groups is not looked at: its closing record stays where it is)
                max_readlen = R[r].maxq;  // stale _max_readlen: the value of the candidate closing at this flush (Q5)
                flush(prev, r);
            }
            prev = r;
        }
        max_readlen = in.last_maxq;
        flush(prev, NR - 1);
        if (prof) fprintf(stderr, "[walk] SVs by vertices visited in their traversal: 1:%zu 2:%zu 3:%zu 4:%zu 5:%zu 6:%zu 7+:%zu (from old vertices %zu)\n",
                          prof_hist[1], prof_hist[2], prof_hist[3], prof_hist[4], prof_hist[5], prof_hist[6], prof_hist[7], prof_old);
        if (prof) {  // sizes of the connected components over the gate-passing groups the host was handed
            std::vector<uint32_t> parent((size_t)NR);
            for (int64_t i = 0; i < NR; ++i) parent[(size_t)i] = (uint32_t)i;
            auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
            std::vector<uint8_t> seen((size_t)NR, 0);
            for (const Group& g : groups) {
                if ((int)g.weight < in.opts.min_read_pair && g.lo != g.hi) continue;
                seen[g.lo] = seen[g.hi] = 1;
                const uint32_t a = find(g.lo), b = find(g.hi);
                if (a != b) parent[std::max(a, b)] = std::min(a, b);
            }
            std::vector<uint32_t> size((size_t)NR, 0);
            for (int64_t i = 0; i < NR; ++i) if (seen[(size_t)i]) ++size[find((uint32_t)i)];
            size_t h[8] = {0};  // 1-4, 5-8, 9-16, 17-32, 33-64, 65-256, 257+
            size_t regions_in[8] = {0};
            for (int64_t i = 0; i < NR; ++i) {
                const uint32_t z = size[(size_t)i];
                if (!z) continue;
                const int b = z <= 4 ? 0 : z <= 8 ? 1 : z <= 16 ? 2 : z <= 32 ? 3 : z <= 64 ? 4 : z <= 256 ? 5 : 6;
                ++h[b]; regions_in[b] += z;
            }
            {   // why they are here: components holding a region of more than 64 reads / with more than 3 incoming groups
                std::vector<uint32_t> indeg((size_t)NR, 0);
                for (const Group& g : groups)
                    if (g.lo != g.hi && (int)g.weight >= in.opts.min_read_pair) ++indeg[g.hi];
                std::vector<uint8_t> why((size_t)NR, 0);
                for (int64_t i = 0; i < NR; ++i) {
                    if (!seen[(size_t)i]) continue;
                    const uint32_t root = find((uint32_t)i);
                    if (R[i].n > 64) why[root] |= 1;
                    if (indeg[(size_t)i] > 3) why[root] |= 2;
                }
                size_t nb = 0, nh = 0, nboth = 0, nnone = 0;
                for (int64_t i = 0; i < NR; ++i) {
                    if (!size[(size_t)i]) continue;
                    if (why[(size_t)i] == 1) ++nb; else if (why[(size_t)i] == 2) ++nh; else if (why[(size_t)i] == 3) ++nboth; else ++nnone;
                }
                fprintf(stderr, "[walk] host components with a region of > 64 reads: %zu, with > 3 incoming groups: %zu, both: %zu, neither: %zu\n", nb, nh, nboth, nnone);
            }
            fprintf(stderr, "[walk] host components by regions: <=4:%zu 5-8:%zu 9-16:%zu 17-32:%zu 33-64:%zu 65-256:%zu 257+:%zu; regions in them: %zu %zu %zu %zu %zu %zu %zu\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], regions_in[0], regions_in[1], regions_in[2], regions_in[3], regions_in[4], regions_in[5], regions_in[6]);
        }
        if (prof) fprintf(stderr, "[walk] total %.1f us, svs %zu\n",
                          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp0).count(), out.svs.size());
    }
};

double chisq_upper_tail_int(int half_df, double x) {  // Q(n, x) = e^-x sum_{i<n} x^i / i!  (x already halved)
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < half_df; ++i) {
        term *= x / i;
        sum += term;
    }
    return std::exp(-x) * sum;
}

}  // namespace

void greedy_walk(const WalkInput& in, WalkScratch* scratch, WalkResult& out) {
    Walker w(in, *scratch, out);
    w.run();
}

// BreakDancer.cpp:56-84 (Kahan-compensated sum of the per-library log tails, optional Fisher) and :459-465
void finish_scores(const bdx_opts& opts, const double* log_tail, HostSv* svs, size_t nsvs, uint32_t* n_printed) {
    uint32_t printed = 0;
    const double ln10 = std::log(10);
    for (size_t q = 0; q < nsvs; ++q) {
        HostSv& hs = svs[q];
        double logpvalue = 0.0, err = 0.0;
        for (int32_t i = 0; i < hs.sv.lib_count; ++i) {
            const double tmp_a = log_tail[hs.sv.lib_begin + i] - err;
            const double tmp_b = logpvalue + tmp_a;
            err = (tmp_b - logpvalue) - tmp_a;
            logpvalue = tmp_b;
        }
        if (opts.fisher && logpvalue < 0) {
            const double x = -2 * logpvalue;
            if (std::isfinite(x)) {  // Boost's chi_squared cdf throws on a non-finite argument; the reference keeps logp
                const double fisherP = chisq_upper_tail_int((int)hs.sv.lib_count, x / 2);
                logpvalue = fisherP > std::exp(-99.0) ? std::log(fisherP) : -99;
            }
        }
        const double phred_tmp = -10 * logpvalue / ln10;
        // int(NaN) is INT_MIN on x86-64 (cvttsd2si): the silent multi-library drop the reference exhibits (Q15)
        int phred;
        if (phred_tmp > 99) phred = 99;
        else if (std::isnan(phred_tmp + 0.5) || phred_tmp + 0.5 >= 2147483648.0 || phred_tmp + 0.5 <= -2147483649.0) phred = INT32_MIN;
        else phred = int(phred_tmp + 0.5);
        hs.sv.logp = logpvalue;
        hs.sv.score = phred;
        hs.sv.printed = phred > opts.score_threshold;
        printed += hs.sv.printed;
    }
    *n_printed = printed;
}

}  // namespace bdx
