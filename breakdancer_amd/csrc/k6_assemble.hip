// K6 -- region x region pair groups, component analysis and SV assembly on the device.
//
// Replaces, for the small components of the region graph (reference file:line under src/lib/breakdancer):
//   ReadRegionData.cpp:108-113   edge weights: pairs per (region, region)
//   BreakDancer.cpp:266-346      build_connection: ascending start vertices, BFS frontier, neighbours in ascending
//                                order, every edge consumed by the side that reaches it first, edges below the
//                                weight gate (-r) skipped each time they are met
//   BreakDancer.cpp:348-497      process_sv: gates, breakpoints, copy number, size, score inputs
//   SvBuilder.cpp:18-118         dominant flag, per-library counts of the second-observed mates, positions
//
// Region ids increase with the stream, and a pair is keyed by the region of its second-observed mate, so the
// pairs of region r sit in r's own slice of the compact read list: one wave sorts and run-length merges them
// in registers (k6_pairs_kernel).  Groups below the weight gate are inert in the reference's walk (never traversed,
// never consumed), so connectivity is taken over the gate-passing groups only: min-label propagation plus a closure
// check finds the components of at most kK6MaxMembers regions that lie inside one flush window, and one thread
// replays the walk of such a component from its smallest region (k6_walk_kernel).  Everything else -- larger or
// window-crossing components, regions with more reads than a wave sorts at once -- is handed to the host walk
// (bdx_walk.cpp) as a list of pair groups.
// float32 / float64 operations are spelled with the round-to-nearest intrinsics so that no fused multiply-add can
// change a result against the host code (and the oracle).
#include <algorithm>
#include <cstdint>

#include "bdx_k3.h"

#include "bdx_poisson.h"
#include "bdx_scan.h"

namespace bdx {

namespace {

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// wave-aggregated bump allocation: every lane asks for `need` entries, one atomic per wave
__device__ __forceinline__ uint32_t wave_reserve(uint32_t need, uint32_t* counter) {
    const uint32_t inc = wave_incl_scan(need);
    const uint32_t tot = (uint32_t)__shfl((int)inc, 63);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 63 && tot) base = atomicAdd(counter, tot);
    base = (uint32_t)__shfl((int)base, 63);
    return base + inc - need;
}

struct GrpRange {
    uint32_t beg, cnt;  // cnt == 0: group absent, already consumed, or its reads were never stored
};

__device__ __forceinline__ bool region_stored(const RegionRec& R, const K6Arrays& a) {  // ReadRegionData.cpp:118-121
    const int valid = a.chr_restricted ? (int)R.nonctx : (int)R.n;
    return valid >= a.min_read_pair;
}

}  // namespace

// Sort-and-merge of up to 64 items (one per lane) by their 64-bit key, in registers: equal keys become one part (the lane
// with the lowest index keeps it: `leader`), `rank` is the part's place in key order.  An item is a read (weight 1, its
// insert size) or, kWeighted, the partial part of a chunk of reads (its pairs, its insert-size sum).
struct Merged {
    uint32_t cnt, sum;    // pairs and insert-size sum of my part (all items with my key)
    uint32_t gw;          // pairs of my (lo, r) group = its weight as a connection
    uint32_t rank, gparts;  // my part's place in key order; parts of my group
    uint32_t total, wself;  // pairs of all items / of the items whose first mate is in region r itself
    uint64_t lmask;       // lanes holding a part
    bool leader, gleader; // gleader: one lane per (lo, r) group, holding its first part
};

template <bool kWeighted>
__device__ __forceinline__ Merged merge_items(uint64_t key, uint32_t wc, uint32_t ws, bool has, uint32_t r, int lane) {
    Merged m{};
    const uint64_t hasmask = __ballot(has);
    uint32_t eq_before = 0;
    for (uint64_t mm = hasmask; mm; mm &= mm - 1) {
        const int t = __builtin_ctzll(mm);
        const uint64_t kt = readlane64(key, t);
        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)ws, t);
        const uint32_t ct = kWeighted ? (uint32_t)__builtin_amdgcn_readlane((int)wc, t) : 1u;
        const bool eq = kt == key;
        m.cnt += eq ? ct : 0u;
        m.sum += eq ? st : 0u;
        eq_before += (eq && t < lane) ? 1u : 0u;
        m.gw += ((kt >> 12) == (key >> 12)) ? ct : 0u;
        if (kWeighted) {
            m.total += ct;
            m.wself += (uint32_t)(kt >> 12) == r ? ct : 0u;
        }
    }
    if (!kWeighted) {
        m.total = (uint32_t)__popcll(hasmask);
        m.wself = (uint32_t)__popcll(__ballot(has && (uint32_t)(key >> 12) == r));
    }
    m.leader = has && eq_before == 0;
    m.lmask = __ballot(m.leader);
    bool lo_first = true;
    for (uint64_t mm = m.lmask; mm; mm &= mm - 1) {
        const int t = __builtin_ctzll(mm);
        const uint64_t kt = readlane64(key, t);
        const bool same_lo = (kt >> 12) == (key >> 12);
        m.gparts += same_lo ? 1u : 0u;
        if (kt < key) {
            ++m.rank;
            if (same_lo) lo_first = false;
        }
    }
    m.gleader = m.leader && lo_first;
    return m;
}

// One wave per accepted region: its pairs (second-observed mate in the region) -> sorted, merged (lo, flag, lib) parts.
// A region of more than 64 reads is merged 64 reads at a time; the chunks' parts (at most 64 in total, else the host
// gets them) are merged once more.
__global__ __launch_bounds__(256) void k6_pairs_kernel(K6Arrays a) {
    __shared__ PartRec s_stash[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * 4;
    // the kernels before this one are complete (kernel boundary): tell the polling host, as a stream write-value command
    // -- itself a one-thread kernel on this runtime -- would, without its launch
    if (a.flag_regions && blockIdx.x == 0 && threadIdx.x == 0) {
        __threadfence_system();
        *(volatile uint32_t*)a.flag_regions = a.flag_value;
    }
    if (blockIdx.x == 0 && a.emit_part) a.emit_part[threadIdx.x] = 0;   // (k6_emit_kernel's partial totals: 64 slots x 4 counters)
    const uint32_t NR = a.counts->n_regions;
    const uint32_t mrp = (uint32_t)max(a.min_read_pair, 0);
    for (uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < NR; r += nwaves) {
        const RegionRec rr = a.r_rec[r];
        const uint32_t first = a.first_of ? a.first_of[r] : rr.first, n = rr.n;
        RegSum rs{};
        // the lane's read of chunk c0: key = (region of the first-observed mate, library, flag) if it is the second-observed
        // mate of a pair (SvBuilder.cpp:101-118)
        auto load_read = [&](uint32_t c0, uint64_t& key, uint32_t& is) -> bool {
            const uint32_t i = c0 + lane;
            key = ~0ull;
            is = 0;
            if (i >= n) return false;
            const uint32_t j = first + i;
            int32_t plo;
            if (a.pair_lo) {
                plo = a.pair_lo[j];
            } else {
                const int32_t p = a.partner[j];
                plo = (p >= 0 && (uint32_t)p < j) ? a.region_of[p] : -1;
            }
            if (plo < 0) return false;
            const uint32_t m = a.meta[j];
            key = ((uint64_t)(uint32_t)plo << 12) | ((uint64_t)meta_lib(m) << 4) | (uint64_t)meta_flag(m);
            is = (uint32_t)a.isize[j];
            return true;
        };
        uint64_t key = ~0ull;
        uint32_t is = 0;
        bool has = false, merged = false;
        Merged m{};
        if (a.in_groups) {
            // sharded run: the region's groups were aggregated on the ranks that joined their pairs; the same weighted merge
            // that folds the chunks of a large region folds them (a group can arrive in several partial aggregates)
            const uint32_t g0 = a.in_goff[r], ng = a.in_goff[r + 1] - g0;
            auto load_group = [&](uint32_t c0, uint64_t& k, uint32_t& cnt, uint32_t& sm) -> bool {
                const uint32_t i = c0 + lane;
                k = ~0ull; cnt = 0; sm = 0;
                if (i >= ng) return false;
                const GroupRec g = a.in_groups[g0 + i];
                k = ((g.key >> 38) << 12) | (g.key & 0xFFFull);  // (lo, library, flag): the part key
                cnt = g.pairs; sm = g.sum_isize;
                return true;
            };
            if (ng <= 64) {
                uint32_t cnt;
                has = load_group(0, key, cnt, is);
                if (__ballot(has)) {
                    m = merge_items<true>(key, cnt, is, has, r, lane);
                    merged = true;
                }
            } else {  // more aggregates than one wave merges at once: they go to the host as they are
                rs.big = 1u;
                for (uint32_t c0 = 0; c0 < ng; c0 += 64) {
                    uint64_t ck;
                    uint32_t cc, ci;
                    const bool ch = load_group(c0, ck, cc, ci);
                    const Merged cm = merge_items<true>(ck, cc, ci, ch, r, lane);
                    const uint32_t clo = (uint32_t)(ck >> 12);
                    rs.n_pairs += cm.total;
                    rs.w_self += cm.wself;
                    if (cm.gleader && clo < r) {
                        atomicAdd(&a.out_deg[clo], 1u);
                        a.bad_v[clo] = 1u;
                        if (a.taint && a.r_rec[clo].n == 0) { a.taint[clo] = 1; a.taint[r] = 1; }  // (another rank's region)
                    }
                    if (lane == 0) a.bad_v[r] = 1u;
                    const uint32_t nl = (uint32_t)__popcll(cm.lmask);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&a.counts->n_groups, nl);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (cm.leader) {
                        const uint32_t o = base + cm.rank;
                        if (o < a.g_cap) {
                            GroupRec g;
                            g.key = group_pack(clo, r, (uint32_t)((ck >> 4) & 255), (uint32_t)(ck & 15));
                            g.pairs = cm.cnt;
                            g.sum_isize = cm.sum;
                            a.g_rec[o] = g;
                        } else {
                            a.counts->overflow = 1;
                        }
                    }
                }
            }
        } else if (n <= 64) {
            has = load_read(0, key, is);
            if (__ballot(has)) {
                m = merge_items<false>(key, 1u, is, has, r, lane);
                merged = true;
            }
        } else {
            // chunks of 64 reads -> their parts, stashed in LDS in any order
            uint32_t np = 0;
            bool fits = true;
            for (uint32_t c0 = 0; c0 < n && fits; c0 += 64) {
                uint64_t ck;
                uint32_t ci;
                const bool ch = load_read(c0, ck, ci);
                if (!__ballot(ch)) continue;
                const Merged cm = merge_items<false>(ck, 1u, ci, ch, r, lane);
                const uint32_t nl = (uint32_t)__popcll(cm.lmask);
                if (np + nl > 64u) { fits = false; break; }
                if (cm.leader) s_stash[w][np + cm.rank] = PartRec{ck, cm.cnt, cm.sum};
                np += nl;
            }
            __builtin_amdgcn_wave_barrier();
            if (fits) {
                if (np) {
                    has = (uint32_t)lane < np;
                    const PartRec q = has ? s_stash[w][lane] : PartRec{~0ull, 0u, 0u};
                    key = q.key;
                    m = merge_items<true>(key, q.pairs, q.sum, has, r, lane);
                    merged = true;
                }
            } else {
                // more than 64 distinct parts: every chunk's parts go to the host, which merges them; every group counts
                // as a connection (a chunk cannot know the whole group's weight)
                rs.big = 1u;
                for (uint32_t c0 = 0; c0 < n; c0 += 64) {
                    uint64_t ck;
                    uint32_t ci;
                    const bool ch = load_read(c0, ck, ci);
                    if (!__ballot(ch)) continue;
                    const Merged cm = merge_items<false>(ck, 1u, ci, ch, r, lane);
                    const uint32_t clo = (uint32_t)(ck >> 12);
                    rs.n_pairs += cm.total;
                    rs.w_self += cm.wself;
                    if (cm.gleader && clo < r) {
                        atomicAdd(&a.out_deg[clo], 1u);
                        a.bad_v[clo] = 1u;
                        if (a.taint && a.r_rec[clo].n == 0) { a.taint[clo] = 1; a.taint[r] = 1; }  // (another rank's region)
                    }
                    if (lane == 0) a.bad_v[r] = 1u;
                    const uint32_t nl = (uint32_t)__popcll(cm.lmask);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&a.counts->n_groups, nl);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (cm.leader) {
                        const uint32_t o = base + cm.rank;
                        if (o < a.g_cap) {
                            GroupRec g;
                            g.key = group_pack(clo, r, (uint32_t)((ck >> 4) & 255), (uint32_t)(ck & 15));
                            g.pairs = cm.cnt;
                            g.sum_isize = cm.sum;
                            a.g_rec[o] = g;
                        } else {
                            a.counts->overflow = 1;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // the stash is reused by the wave's next region
        }
        if (merged) {
            const uint32_t lo = (uint32_t)(key >> 12);
            rs.n_pairs = m.total;
            rs.w_self = m.wself;
            // groups below the weight gate are skipped by every try_edge and never reach process_sv: inert
            const bool strong = m.gw >= mrp;
            const bool in_edge = m.gleader && lo < r && strong;
            const uint64_t gl_in = __ballot(in_edge);
            if (in_edge) atomicAdd(&a.out_deg[lo], 1u);
            // sharded run: the earlier region of a gate-passing group is another rank's (its place in this rank's table is empty): the
            // component spans ranks -- both regions are tainted, every rank learns it, and rank 0 walks the component
            if (a.taint && in_edge && a.r_rec[lo].n == 0) { a.taint[lo] = 1; a.taint[r] = 1; }
            rs.np_all = (uint32_t)__popcll(m.lmask);
            rs.np_self = (uint32_t)__popcll(__ballot(m.leader && lo == r));
            rs.np_emit = (uint32_t)__popcll(__ballot(m.leader && (lo == r || strong)));
            rs.n_weak = (uint32_t)__popcll(__ballot(m.gleader && lo < r && !strong));
            rs.n_in = (uint32_t)__popcll(gl_in);
            if (rs.n_in > (uint32_t)kK6MaxIn) {  // too many connections for a device-walked component
                if (in_edge) a.bad_v[lo] = 1u;
                if (lane == 0) a.bad_v[r] = 1u;
            } else {
                // (first round of the min-label propagation: each gate-passing group joins its two regions)
                if (in_edge && !a.force_host) {
                    const uint32_t lr = a.label[r], ll = a.label[lo];
                    if (lr < ll) atomicMin(&a.label[lo], lr);
                    else if (ll < lr) atomicMin(&a.label[r], ll);
                }
                uint64_t mm = gl_in;
#pragma unroll
                for (int e = 0; e < kK6MaxIn; ++e) {  // (static indices: the record stays in registers)
                    if (mm) {
                        const int t = __builtin_ctzll(mm);
                        mm &= mm - 1;
                        rs.e_lo[e] = (uint32_t)__builtin_amdgcn_readlane((int)lo, t);
                        rs.e_w[e] = (uint32_t)__builtin_amdgcn_readlane((int)m.gw, t);
                        rs.e_off[e] = (uint32_t)__builtin_amdgcn_readlane((int)m.rank, t);
                        rs.e_cnt[e] = (uint32_t)__builtin_amdgcn_readlane((int)m.gparts, t);
                    }
                }
            }
            if (m.leader) a.parts[first + m.rank] = PartRec{(lo < r && !strong) ? (key | kWeakPart) : key, m.cnt, m.sum};
        }
        if (lane == 0) a.rs[r] = rs;
    }
}

// one round of min-label propagation over the gate-passing groups (each is stored with its later region)
__global__ __launch_bounds__(256) void k6_label_kernel(K6Arrays a) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.counts->n_regions) return;
    const RegSum* s = &a.rs[r];
    const uint32_t n_in = s->n_in;
    if (s->big || n_in == 0 || n_in > (uint32_t)kK6MaxIn) return;
    for (uint32_t e = 0; e < n_in; ++e) {
        const uint32_t lo = s->e_lo[e];
        const uint32_t lr = a.label[r], ll = a.label[lo];
        if (lr < ll) atomicMin(&a.label[lo], lr);
        else if (ll < lr) atomicMin(&a.label[r], ll);
    }
    // pointer jumping: adopt the label of the region this one points at (chains of many regions converge in a few rounds;
    // whatever has not converged fails k6_classify_kernel's closure check and goes to the host)
    const uint32_t l = a.label[r], l2 = a.label[l];
    if (l2 < l) atomicMin(&a.label[r], l2);
}

// closure check and member registration: a label whose members only have groups among themselves, all inside one
// flush window, is a complete component
__global__ __launch_bounds__(256) void k6_classify_kernel(K6Arrays a) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.counts->n_regions) return;
    const RegSum* s = &a.rs[r];
    const uint32_t n_in = s->n_in;
    const bool hub = s->big || n_in > (uint32_t)kK6MaxIn;
    if (!hub && n_in == 0 && a.out_deg[r] == 0 && s->np_self == 0) return;  // no group that could ever be consumed
    const uint32_t L = a.label[r];
    if (hub || a.bad_v[r] || (a.taint && a.taint[r])) a.bad[L] = 1u;
    if (!hub) {
#pragma unroll
        for (uint32_t e = 0; e < (uint32_t)kK6MaxIn; ++e) {  // (static trip count: no private array behind the loop)
            if (e < n_in) {
                const uint32_t ll = a.label[s->e_lo[e]];
                if (ll != L) { a.bad[L] = 1u; a.bad[ll] = 1u; }
            }
        }
    }
    const uint32_t slot = atomicAdd(&a.mcount[L], 1u);
    if (a.big_walk && slot < (uint32_t)kK6BigMembers) a.member_ids[(size_t)L * kK6BigMembers + slot] = r;
    if (slot < (uint32_t)kK6MaxMembers) {
        // The member record leaves as seven 16-byte stores (k6_walk_kernel fetches it as seven 16-byte words): written field by field it was
        // 28 scattered 4-byte stores per region, each a 32-byte write at the memory side -- 138 MB for the 130 k regions of a genome share
        // (profiles/r05_pmc_genome.txt).  Every word is named (no struct on the stack: a private copy would go through scratch memory).
        static_assert(sizeof(MemberInfo) == 7 * 16 && sizeof(RegionRec) == 36 && offsetof(MemberInfo, rec) == 4 && offsetof(MemberInfo, np_all) == 40 &&
                          offsetof(MemberInfo, e_lo) == 56 && offsetof(MemberInfo, stored) == 104, "a member's record is seven 16-byte words");
        const uint32_t* rw = (const uint32_t*)&a.r_rec[r];
        const uint32_t r0 = rw[0], r1 = rw[1], r2 = rw[2], r3 = rw[3], r4 = rw[4], r5 = rw[5], r6 = rw[6], r7 = rw[7], r8 = a.first_of ? a.first_of[r] : rw[8];   // (word 8: `first`)
        const uint32_t stored = (int)(a.chr_restricted ? r5 : r3) >= a.min_read_pair ? 1u : 0u;  // region_stored(): nonctx is word 5, n word 3
        uint4* dst = (uint4*)&a.members[(size_t)L * kK6MaxMembers + slot];
        dst[0] = make_uint4(r, r0, r1, r2);
        dst[1] = make_uint4(r3, r4, r5, r6);
        dst[2] = make_uint4(r7, r8, s->np_all, s->np_self);
        dst[3] = make_uint4(s->w_self, hub ? 0u : n_in, s->e_lo[0], s->e_lo[1]);
        dst[4] = make_uint4(s->e_lo[2], s->e_w[0], s->e_w[1], s->e_w[2]);
        dst[5] = make_uint4(s->e_off[0], s->e_off[1], s->e_off[2], s->e_cnt[0]);
        dst[6] = make_uint4(s->e_cnt[1], s->e_cnt[2], stored, 0u);
    }
    if (s->np_emit) atomicAdd(&a.pcount[L], s->np_emit);
}

namespace {

// process_sv (BreakDancer.cpp:348-497) + SvBuilder for region A (and B when B >= 0) over the alive groups gs[0..2] =
// (A,A), (A,B), (B,B).  Returns false when a gate rejects the candidate; otherwise the record and its entries are
// written to the staging slot (by the lane with `store` set).
struct RunConst {  // run constants of process_sv, wherever the caller keeps them (HBM, or LDS when they are few)
    const float* lib_mean;     // [nlibs]
    const uint32_t* hist;      // [nlibs][kNumFlags]
    const float* key_density;  // [nkeys]
    uint32_t covered;          // covered_ref_len
};

constexpr int kFewLibs = 4;   // libraries up to which assemble_sv adds the parts up per library in registers
__device__ bool assemble_sv(const K6Arrays& a, const RunConst& rc_, const PartRec* P, uint32_t A, int32_t B, const RegionRec& ra, const RegionRec& rb,
                            const uint32_t* pk_last_a, const uint32_t* pk_first_b, const GrpRange (&gs)[3], int max_readlen,
                            uint32_t slot, uint32_t start, bool store, uint32_t* nacc_out, uint32_t* ncn_out, uint32_t kprow = ~0u) {
    (void)kprow;   // (measurement build: the row of clocks of this call, tools/kprof.py)
    KPROF(kprow, 0);
    const uint32_t lib_room = a.lib_stride;
    const int mrp = a.min_read_pair;
    int cv[kNumFlags];  // pairs per flag: registers, every index static (a counter array in LDS costs a dependent round trip per update)
#pragma unroll
    for (int f = 0; f < kNumFlags; ++f) cv[f] = 0;
    int num_pairs = 0;
    // (four parts of a group requested at once: a loop that fetches one part per turn waits for every one of them in turn -- ~1.5 us each
    // with sixteen waves of scattered requests on the compute unit -- and a group has up to flags x libraries parts)
    const PartRec none_part{0ull, 0u, 0u};
#pragma unroll
    for (int g = 0; g < 3; ++g)
        for (uint32_t i = 0; i < gs[g].cnt; i += 4) {
            PartRec q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = i + u < gs[g].cnt ? P[gs[g].beg + i + u] : none_part;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = (int)(q[u].key & 15);
                const int pr = (int)q[u].pairs;   // (0 for the padding: it counts for nothing)
#pragma unroll
                for (int ff = 0; ff < kNumFlags; ++ff) cv[ff] += f == ff ? pr : 0;
                num_pairs += pr;
            }
        }
    if (num_pairs != 0x7FFFFFFF) KPROF(kprow, 1);   // (the groups' parts have arrived)
    if (num_pairs < mrp) return false;
    int flag = BDX_NA, flag_count;
    {   // the dominant flag (lowest one on ties)
        int best = 0, bestv = cv[0];
#pragma unroll
        for (int f = 1; f < kNumFlags; ++f)
            if (cv[f] > bestv) { best = f; bestv = cv[f]; }
        if (bestv > 0) flag = best;
        flag_count = bestv > 0 ? bestv : cv[BDX_NA];
    }
    if (flag_count < mrp) return false;

    int chr[2], pos[2], fwd[2], rev[2];
    chr[0] = ra.tid; pos[0] = ra.start; pos[1] = ra.end;
    fwd[0] = (int)(ra.n - ra.rev); rev[0] = (int)ra.rev;
    int total_region_size = ra.end - ra.start + 1;
    if (B >= 0) {
        fwd[1] = (int)(rb.n - rb.rev); rev[1] = (int)rb.rev;
        if (flag == BDX_ARP_RF) pos[1] = rb.end + max_readlen - 5;
        else if (flag == BDX_ARP_FF) { pos[0] = pos[1]; pos[1] = rb.end + max_readlen - 5; }
        else if (flag == BDX_ARP_RR) pos[1] = rb.start;
        else { pos[0] = pos[1]; pos[1] = rb.start; }
        chr[1] = rb.tid;
        total_region_size += rb.end - rb.start + 1;
    } else {
        fwd[1] = fwd[0]; rev[1] = rev[0]; chr[1] = ra.tid; pos[1] = ra.end;
    }

    LibStage* ls = a.lib_stage + (size_t)slot * a.lib_stride;
    uint32_t nacc = 0;
    float diff = 0.0f;
    if (a.nlibs >= 2 && a.nlibs <= kFewLibs && !a.asm_plain) {   // (one library: the merge below is one short loop -- 0.2492 against 0.2512 ms per configs[1] step)
        // Few libraries (a run has one to four): ONE more pass over the groups' parts -- the loads of the first pass again, independent of each
        // other, out of the cache -- adds the dominant flag's pairs and spans up per library in registers (static indices), then the libraries
        // in ascending order.  The three-way merge below walks the parts by dependent loads -- P[beg + idx] decides the next idx --: 28 of a
        // call's 45 us at the median at a genome share (in-kernel clocks, profiles/r06_walk_kernel.txt).  Integer sums: the order of the
        // parts does not matter; the float accumulation runs over the libraries in ascending order as the merge does.
        int rcL[kFewLibs], spL[kFewLibs];
#pragma unroll
        for (int l = 0; l < kFewLibs; ++l) { rcL[l] = 0; spL[l] = 0; }
#pragma unroll
        for (int g = 0; g < 3; ++g)
            for (uint32_t i = 0; i < gs[g].cnt; i += 4) {
                PartRec q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = i + u < gs[g].cnt ? P[gs[g].beg + i + u] : none_part;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool mine = (int)(q[u].key & 15) == flag;   // (the padding's pairs and sum are 0)
                    const int lib = (int)((q[u].key >> 4) & 255);
#pragma unroll
                    for (int l = 0; l < kFewLibs; ++l) {
                        rcL[l] += (mine && lib == l) ? (int)q[u].pairs : 0;
                        spL[l] += (mine && lib == l) ? (int)q[u].sum : 0;
                    }
                }
            }
#pragma unroll
        for (int l = 0; l < kFewLibs; ++l) {
            if (rcL[l] == 0) continue;   // (a part holds at least one pair)
            diff = __fadd_rn(diff, __fsub_rn((float)spL[l], __fmul_rn((float)rcL[l], rc_.lib_mean[l])));
            const uint32_t nflag = rc_.hist[(size_t)l * kNumFlags + flag];
            double lambda = __dmul_rn((double)total_region_size, __ddiv_rn((double)nflag, (double)rc_.covered));
            lambda = (1.0e-10 < lambda) ? lambda : 1.0e-10;
            if (store && nacc < lib_room) ls[nacc] = LibStage{l, rcL[l], lambda};
            ++nacc;
        }
    } else {
    // per-library pairs / spans of the dominant flag in ascending library order: a three-way merge, the parts of a
    // group being sorted by (flag, library)
    uint32_t idx[3] = {0, 0, 0};  // (every loop over g below is unrolled: static indices, registers)
#pragma unroll
    for (int g = 0; g < 3; ++g)
        while (idx[g] < gs[g].cnt && (int)(P[gs[g].beg + idx[g]].key & 15) != flag) ++idx[g];
    while (true) {
        int best = 256;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            if (idx[g] < gs[g].cnt) best = min(best, (int)((P[gs[g].beg + idx[g]].key >> 4) & 255));
        if (best == 256) break;
        int rc = 0, span = 0;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            if (idx[g] < gs[g].cnt && (int)((P[gs[g].beg + idx[g]].key >> 4) & 255) == best) {
                rc += (int)P[gs[g].beg + idx[g]].pairs;
                span += (int)P[gs[g].beg + idx[g]].sum;
                ++idx[g];
                while (idx[g] < gs[g].cnt && (int)(P[gs[g].beg + idx[g]].key & 15) != flag) ++idx[g];
            }
        diff = __fadd_rn(diff, __fsub_rn((float)span, __fmul_rn((float)rc, rc_.lib_mean[best])));
        const uint32_t nflag = rc_.hist[(size_t)best * kNumFlags + flag];
        double lambda = __dmul_rn((double)total_region_size, __ddiv_rn((double)nflag, (double)rc_.covered));
        lambda = (1.0e-10 < lambda) ? lambda : 1.0e-10;
        if (store && nacc < lib_room) ls[nacc] = LibStage{best, rc, lambda};
        ++nacc;
    }
    }
    if (nacc != 0x7FFFFFFFu) KPROF(kprow, 2);   // (per-library merge done)
    if (nacc > lib_room) { a.counts->overflow = 3; return false; }  // cannot happen: distinct libraries <= min(nlibs, parts)

    // normal reads between the regions: proper reads after A's last read up to and including B's first
    CnStage* cs = a.cn_stage + (size_t)slot * a.nkeys;
    uint32_t ncn = 0;
    float cn_sum = 0.0f;
    if (B >= 0) {
        const int nk = a.nkeys;
        const float span = (float)(pos[1] - pos[0]);
        for (int k = 0; k < nk; ++k) {
            const uint32_t cnt = pk_first_b[k] - pk_last_a[k];
            if (cnt == 0) continue;
            const float cn = __fmul_rn(__fdiv_rn((float)cnt, __fmul_rn(rc_.key_density[k], span)), 2.0f);
            if (store) cs[ncn] = CnStage{k, cn};
            ++ncn;
            cn_sum = __fadd_rn(cn_sum, cn);
        }
    }
    // without normal reads between the regions the reference divides 0 by 0: x86's default NaN has the sign bit set and
    // survives "1 - x" unchanged, which is what its formatter prints as -nan
    const float allele_frequency =
        ncn ? __fsub_rn(1.0f, __fdiv_rn(cn_sum, __fmul_rn(2.0f, (float)ncn))) : __uint_as_float(0xFFC00000u);

    if (ncn != 0x7FFFFFFFu) KPROF(kprow, 3);   // (the proper-read samples have arrived: copy numbers done)
    if (flag != BDX_ARP_RF && flag != BDX_ARP_RR && pos[0] + max_readlen - 5 < pos[1]) pos[0] += max_readlen - 5;
    const int diffspan = (int)((double)__fdiv_rn(diff, (float)flag_count) + 0.5);

    SvOut o;
    for (int i = 0; i < 2; ++i) { o.sv.chr[i] = chr[i]; o.sv.pos[i] = pos[i] + 1; o.sv.fwd[i] = fwd[i]; o.sv.rev[i] = rev[i]; }
    o.sv.flag = flag; o.sv.size = diffspan; o.sv.score = 0; o.sv.num_reads = flag_count; o.sv.printed = 0;
    o.sv.region[0] = (int32_t)A; o.sv.region[1] = B;
    o.sv.lib_begin = 0; o.sv.lib_count = (int32_t)nacc;
    o.sv.cn_begin = 0; o.sv.cn_count = (int32_t)ncn;
    o.sv.allele_frequency = allele_frequency; o.sv.logp = 0.0;
    o.grp_mask = (gs[0].cnt ? 1u : 0u) | (gs[1].cnt ? 2u : 0u) | (gs[2].cnt ? 4u : 0u);
    o.start = start;
    if (store) a.sv_stage[slot] = o;
    KPROF(kprow, 4);
    *nacc_out = nacc;
    *ncn_out = ncn;
    return true;
}

__device__ __forceinline__ int pair_index(int x, int y) {  // x < y < 4 -> 0..5
    return x == 0 ? y - 1 : (x == 1 ? y + 1 : 5);
}

}  // namespace

namespace {

// is the component with label L walked on the device?  (evaluated identically by every member and by the walk)
__device__ __forceinline__ bool component_on_device(const K6Arrays& a, uint32_t L) {
    return !a.force_host && !a.bad[L] && a.mcount[L] <= (uint32_t)(a.big_walk ? kK6BigMembers : kK6MaxMembers) &&
           min((uint32_t)a.nlibs, a.pcount[L]) <= a.lib_stride;
}

}  // namespace

// One thread per region: the groups of the components that are not walked here go to the host list; totals; the last
// workgroup mirrors the counters into pinned host memory, so the host's share is complete when this kernel ends.
__global__ __launch_bounds__(256) void k6_emit_kernel(K6Arrays a) {
    const int lane = threadIdx.x & 63;
    const uint32_t NR = a.counts->n_regions;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < NR;
    RegSum s{};
    uint32_t od = 0, L = 0;
    if (active) { s = a.rs[r]; od = a.out_deg[r]; L = a.label[r]; a.own_nsv[r] = 0; a.own_nacc[r] = 0; a.own_ncn[r] = 0; }
    const bool hub = s.big || s.n_in > (uint32_t)kK6MaxIn;
    const bool registered = active && (hub || s.n_in > 0 || od > 0 || s.np_self > 0);
    const bool covered = registered && component_on_device(a, L);
    {
        const uint32_t need = (registered && !covered && !s.big) ? s.np_emit : 0u;
        uint32_t o = wave_reserve(need, &a.counts->n_groups);
        if (need) {
            const uint32_t first = a.first_of ? a.first_of[r] : a.r_rec[r].first;
            for (uint32_t i = 0; i < s.np_all; ++i) {
                const PartRec q = a.parts[first + i];
                const uint64_t key = q.key;
                if (key & kWeakPart) continue;
                if (o < a.g_cap) {
                    GroupRec g;
                    g.key = ((key >> 12) << 38) | ((uint64_t)r << 12) | (key & 0xFFFull);
                    g.pairs = q.pairs;
                    g.sum_isize = q.sum;
                    a.g_rec[o] = g;
                } else {
                    a.counts->overflow = 1;
                }
                ++o;
            }
        }
    }
    {   // the walk kernels' work: the smallest region of every device-walked component
        const bool owner = covered && L == r;
        const bool small = owner && a.mcount[L] <= (uint32_t)kK6MaxMembers;
        if (active) a.owners[r] = small ? a.mcount[L] : 0u;  // (the walk kernel takes one lane per region: no list, no indirection)
        const uint32_t no = (uint32_t)__popcll(__ballot(small));
        // (this kernel's four totals go to one of 64 slots, summed by whoever mirrors the counters: as atomics on the counter record itself they were
        // ~8,000 read-modify-writes of ONE cache line for a genome share's 130 k regions -- the kernel's 76 us)
        uint32_t* part = a.emit_part ? a.emit_part + (blockIdx.x & 63u) * 4 : nullptr;
        if (lane == 0 && no) atomicAdd(part ? &part[0] : &a.counts->n_owners, no);
        const uint32_t ob = wave_reserve(owner && !small ? 1u : 0u, &a.counts->n_owners_big);
        if (owner && !small) a.owners_big[ob] = r;
    }
    {   // pairs of all regions; connections: inert ones everywhere, the others where the component is walked on the device
        const uint32_t pairs = active ? s.n_pairs : 0u;
        const uint32_t grp = (active ? s.n_weak : 0u) + (covered ? s.n_in + (s.np_self ? 1u : 0u) : 0u);
        const uint32_t gb = (covered && a.mcount[L] > (uint32_t)kK6MaxMembers) ? s.n_in + (s.np_self ? 1u : 0u) : 0u;
        const uint32_t tp = wave_sum_u32(pairs), tg = wave_sum_u32(grp), tb = wave_sum_u32(gb);
        if (lane == 0) {
            uint32_t* part = a.emit_part ? a.emit_part + (blockIdx.x & 63u) * 4 : nullptr;
            if (tp) atomicAdd(part ? &part[1] : &a.counts->n_pairs, tp);
            if (tg) atomicAdd(part ? &part[2] : &a.counts->n_groups_dev, tg);
            if (tb) atomicAdd(part ? &part[3] : &a.counts->n_groups_big, tb);
        }
    }
}

namespace {
// the counter record as the host gets it: k6_emit_kernel's partial totals (64 slots of {owners, pairs, groups on the device, groups of large
// components}) added to their words.  All 64 lanes of a wave call it; lane l returns word l (lanes past the record: 0)
__device__ __forceinline__ uint32_t mirrored_count_word(const K6Arrays& a, int lane) {
    constexpr int kWords = (int)(sizeof(StageCounts) / 4);
    uint32_t v = lane < kWords ? ((const uint32_t*)a.counts)[lane] : 0u;
    if (a.emit_part) {
        uint32_t p0 = a.emit_part[lane * 4], p1 = a.emit_part[lane * 4 + 1], p2 = a.emit_part[lane * 4 + 2], p3 = a.emit_part[lane * 4 + 3];
        p0 = wave_sum_u32(p0); p1 = wave_sum_u32(p1); p2 = wave_sum_u32(p2); p3 = wave_sum_u32(p3);
        constexpr int iOwners = (int)(offsetof(StageCounts, n_owners) / 4), iPairs = (int)(offsetof(StageCounts, n_pairs) / 4),
                      iDev = (int)(offsetof(StageCounts, n_groups_dev) / 4), iBig = (int)(offsetof(StageCounts, n_groups_big) / 4);
        v += lane == iOwners ? p0 : lane == iPairs ? p1 : lane == iDev ? p2 : lane == iBig ? p3 : 0u;
    }
    return v;
}
}  // namespace

// the counters after k6_emit_kernel, mirrored into pinned host memory (a per-workgroup fence + last-block copy inside
// that kernel costs an L2 write-back per workgroup on this multi-die part; one tiny launch does not)
__global__ __launch_bounds__(64) void k6_mirror_kernel(K6Arrays a) {
    const uint32_t word = mirrored_count_word(a, (int)threadIdx.x);
    if (threadIdx.x < sizeof(StageCounts) / 4) ((uint32_t*)a.counts_host)[threadIdx.x] = word;
    if (a.flag_groups) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *(volatile uint32_t*)a.flag_groups = a.flag_value;
    }
}

// order key of a candidate that is placed by key (see K6Arrays)
constexpr int kKeyShiftT = 34, kKeyShiftStart = 7;
constexpr uint32_t kKeySeqMask = 127u;

// The general path of the walk: a component of up to kK6BigMembers (64) regions, its members' records in LDS and the groups
// found through each member's list of incoming groups instead of a table of all pairs.  Same replay as the main path
// of k6_walk_kernel below (which keeps the components of up to kK6MaxMembers regions: one coalesced fetch, pair table).
struct BigTab {
    uint32_t rid[kK6BigMembers];
    RegionRec rec[kK6BigMembers];
    RegSum rs[kK6BigMembers];
    uint8_t elo[kK6BigMembers][4];   // member index of each incoming group's earlier region
    int tails[4 * kK6BigMembers], newtails[4 * kK6BigMembers];
};

__device__ __forceinline__ void walk_big(const K6Arrays& a, BigTab& B, uint32_t L, int lane, uint32_t NR) {
    const int k = (int)a.mcount[L];
    const int mrp = a.min_read_pair, nk = a.nkeys;
    const uint32_t period = (uint32_t)a.period;
    const RunConst rc_{a.lib_mean, a.hist, a.key_density, a.p1->covered_ref_len};
    {   // member ids in ascending order: rank by counting (they all differ)
        const uint32_t id = lane < k ? a.member_ids[(size_t)L * kK6BigMembers + lane] : 0xffffffffu;
        uint32_t rank = 0;
        for (int q = 0; q < k; ++q) rank += (uint32_t)__shfl((int)id, q) < id ? 1u : 0u;
        if (lane < k) B.rid[rank] = id;
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int kRecW = sizeof(RegionRec) / 4, kRsW = sizeof(RegSum) / 4;
    for (int i = lane; i < k * (kRecW + kRsW); i += 64) {
        const int m = i / (kRecW + kRsW), wd = i - m * (kRecW + kRsW);
        const uint32_t r = B.rid[m];
        if (wd < kRecW) ((uint32_t*)&B.rec[m])[wd] = (wd == kRecW - 1 && a.first_of) ? a.first_of[r] : ((const uint32_t*)&a.r_rec[r])[wd];   // (the last word is `first`)
        else ((uint32_t*)&B.rs[m])[wd - kRecW] = ((const uint32_t*)&a.rs[r])[wd - kRecW];
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < k * kK6MaxIn; i += 64) {
        const int m = i / kK6MaxIn, e = i - m * kK6MaxIn;
        uint32_t x = 255u;
        if ((uint32_t)e < B.rs[m].n_in)
            for (int q = 0; q < m; ++q)
                if (B.rid[q] == B.rs[m].e_lo[e]) x = (uint32_t)q;  // the closure check guarantees it is a member
        B.elo[m][e] = (uint8_t)x;
    }
    __builtin_amdgcn_wave_barrier();
    // per-member state as 64-bit masks; a group (earlier region, m) is bit m of the mask of its place e in m's list
    uint64_t stored = 0, self_alive = 0, self_done = 0;
    uint64_t ea0 = 0, ea1 = 0, ea2 = 0, ed0 = 0, ed1 = 0, ed2 = 0;  // groups alive / done, by place in the list
    for (int m = 0; m < k; ++m) {
        const uint64_t bm = 1ull << m;
        if (region_stored(B.rec[m], a)) stored |= bm;
        if (B.rs[m].np_self) self_alive |= bm;
        const uint32_t ni = B.rs[m].n_in;
        if (ni > 0 && B.rs[m].e_cnt[0] && B.elo[m][0] != 255u) ea0 |= bm;
        if (ni > 1 && B.rs[m].e_cnt[1] && B.elo[m][1] != 255u) ea1 |= bm;
        if (ni > 2 && B.rs[m].e_cnt[2] && B.elo[m][2] != 255u) ea2 |= bm;
    }
    const GrpRange none{0, 0};
    int* const tails = B.tails;
    int* const newtails = B.newtails;
    for (int f = 0; f < k;) {
        const uint32_t W = B.rid[f] / period;
        int fe = f + 1;
        while (fe < k && B.rid[fe] / period == W) ++fe;
        const uint32_t rl = (W + 1) * period - 1;
        const int max_readlen = rl < NR ? a.r_rec[rl].maxq : a.counts->last_maxq;
        uint64_t visited = 0;
        uint32_t nseq = 0;
        for (int sv = 0; sv < fe; ++sv) {
            if (visited & (1ull << sv)) continue;
            const bool from_old = sv < f;
            const uint32_t start = B.rid[sv];
            uint32_t nsv = 0, nacc_tot = 0, ncn_tot = 0, prev_slot = 0;
            int nt = 1, nn = 0;
            tails[0] = sv;
            while (nt) {
                nn = 0;
                for (int ti = 0; ti < nt; ++ti) {
                    const int tail = tails[ti];
                    if (visited & (1ull << tail)) continue;
                    for (int nb = 0; nb < fe; ++nb) {  // neighbours in ascending order, the vertex itself at its own place
                        int A, Bm, eidx = -1;
                        uint32_t slot;
                        if (nb == tail) {
                            if (tail < f) continue;  // its self group belonged to an earlier flush
                            if (!B.rs[tail].np_self || (self_done & (1ull << tail)) || (int)B.rs[tail].w_self < mrp) continue;
                            self_done |= 1ull << tail;
                            A = tail; Bm = -1;
                            slot = B.rec[tail].first + B.rs[tail].n_in;
                        } else {
                            const int x = min(nb, tail), y = max(nb, tail);
                            if (y < f) continue;     // a group of an earlier flush
                            for (uint32_t e = 0; e < B.rs[y].n_in; ++e)
                                if (B.elo[y][e] == (uint8_t)x) eidx = (int)e;
                            if (eidx < 0) continue;
                            const uint64_t bit = 1ull << y;
                            const uint64_t done = eidx == 0 ? ed0 : (eidx == 1 ? ed1 : ed2);
                            if ((done & bit) || (int)B.rs[y].e_w[eidx] < mrp) continue;
                            if (eidx == 0) ed0 |= bit; else if (eidx == 1) ed1 |= bit; else ed2 |= bit;
                            A = x; Bm = y;
                            slot = B.rec[y].first + (uint32_t)eidx;
                        }
                        if (nn < 4 * kK6BigMembers) newtails[nn++] = nb;
                        const int Bi = Bm >= 0 ? Bm : A;
                        GrpRange gs[3] = {none, none, none};
                        const bool stA = (stored >> A) & 1ull, stB = (stored >> Bi) & 1ull;
                        if ((self_alive & (1ull << A)) && stA) gs[0] = GrpRange{B.rec[A].first + B.rs[A].np_all - B.rs[A].np_self, B.rs[A].np_self};
                        if (Bm >= 0) {
                            const uint64_t bit = 1ull << Bm;
                            const uint64_t alive = eidx == 0 ? ea0 : (eidx == 1 ? ea1 : ea2);
                            if ((alive & bit) && stA && stB) gs[1] = GrpRange{B.rec[Bm].first + B.rs[Bm].e_off[eidx], B.rs[Bm].e_cnt[eidx]};
                            if ((self_alive & bit) && stB) gs[2] = GrpRange{B.rec[Bm].first + B.rs[Bm].np_all - B.rs[Bm].np_self, B.rs[Bm].np_self};
                            if (gs[1].cnt) { if (eidx == 0) ea0 &= ~bit; else if (eidx == 1) ea1 &= ~bit; else ea2 &= ~bit; }
                        }
                        // paired reads leave their regions before any gate (BreakDancer.cpp:363-368)
                        if (gs[0].cnt) self_alive &= ~(1ull << A);
                        if (gs[2].cnt) self_alive &= ~(1ull << Bm);
                        const uint32_t* pkA = a.r_pk + (size_t)B.rid[A] * 2 * nk + nk;
                        const uint32_t* pkB = a.r_pk + (size_t)B.rid[Bi] * 2 * nk;
                        uint32_t nacc = 0, ncn = 0;
                        if (assemble_sv(a, rc_, a.parts, B.rid[A], Bm >= 0 ? (int32_t)B.rid[Bm] : -1, B.rec[A], B.rec[Bi], pkA, pkB, gs, max_readlen, slot,
                                        start, lane == 0, &nacc, &ncn)) {
                            if (from_old) {
                                if (lane == 0) {
                                    const uint32_t q = atomicAdd(&a.counts->n_old, 1u);
                                    a.old_key[q] = ((uint64_t)(W * period) << kKeyShiftT) | ((uint64_t)start << kKeyShiftStart) | (uint64_t)(nseq & kKeySeqMask);
                                    a.old_slot[q] = slot;
                                }
                                ++nseq;
                            } else if (lane == 0) {
                                if (nsv == 0) a.own_first[start] = slot; else a.slot_next[prev_slot] = slot;
                            }
                            prev_slot = slot;
                            ++nsv;
                            nacc_tot += nacc;
                            ncn_tot += ncn;
                        }
                    }
                    visited |= 1ull << tail;
                }
                nt = nn;
                for (int i = 0; i < nn; ++i) tails[i] = newtails[i];
            }
            if (!from_old && nsv && lane == 0) { a.own_nsv[start] = nsv; a.own_nacc[start] = nacc_tot; a.own_ncn[start] = ncn_tot; }
        }
        f = fe;
    }
}

// The walk's small tables are indexed by run-time values: as private arrays they would live in scratch memory.  Every lane
// owns one column of a wave-wide LDS table instead: field f of lane l at word f * 64 + l (no bank conflict when the lanes
// touch the same field, which they mostly do).
struct WalkTab {
    uint32_t* p;  // this lane's column
    static constexpr int kSbeg = 0, kScnt = 4, kEbeg = 8, kEcnt = 14, kSw = 20, kEw = 24, kSslot = 30, kEslot = 34, kOrd = 40, kRid = 44,
                         kCalls = 48, kRtid = 58, kRstart = 62, kRend = 66, kRn = 70, kRrev = 74, kFields = 78;
    static constexpr int kStride = 32;   // lanes of a wave that walk (K6Arrays::walk_lanes <= this): field f of lane l at word f * kStride + l
    __device__ __forceinline__ uint32_t& at(int f) const { return p[f * kStride]; }
    __device__ __forceinline__ uint32_t& Sbeg(int i) const { return at(kSbeg + i); }
    __device__ __forceinline__ uint32_t& Scnt(int i) const { return at(kScnt + i); }
    __device__ __forceinline__ uint32_t& Ebeg(int e) const { return at(kEbeg + e); }
    __device__ __forceinline__ uint32_t& Ecnt(int e) const { return at(kEcnt + e); }
    __device__ __forceinline__ uint32_t& Sw(int i) const { return at(kSw + i); }
    __device__ __forceinline__ uint32_t& Ew(int e) const { return at(kEw + e); }
    __device__ __forceinline__ uint32_t& Sslot(int i) const { return at(kSslot + i); }
    __device__ __forceinline__ uint32_t& Eslot(int e) const { return at(kEslot + e); }
    __device__ __forceinline__ uint32_t& rid(int i) const { return at(kRid + i); }    // region of the member at place i of the ascending order
    __device__ __forceinline__ uint32_t& Rtid(int i) const { return at(kRtid + i); }    // what process_sv reads of the region's record
    __device__ __forceinline__ uint32_t& Rstart(int i) const { return at(kRstart + i); }
    __device__ __forceinline__ uint32_t& Rend(int i) const { return at(kRend + i); }
    __device__ __forceinline__ uint32_t& Rn(int i) const { return at(kRn + i); }
    __device__ __forceinline__ uint32_t& Rrev(int i) const { return at(kRrev + i); }
    __device__ __forceinline__ uint32_t& call(int c) const { return at(kCalls + c); } // the walk's process_sv calls, in order (every group makes at most one)
};
static_assert(kK6MaxMembers == 4 && kK6MaxIn == 3, "WalkTab layout");

// One LANE per component of at most kK6MaxMembers regions, in three converged phases:
//   0  the component's description (k6_classify_kernel wrote it next to its label) -> the lane's tables, every load of it
//      issued at once (static indices, registers are free: the kernel needs ~140 waves);
//   1  the traversal itself -- build_connection's flushes, start vertices, frontier, neighbours in ascending order, every
//      group consumed by the side that reaches it first -- in registers only (gates and liveness as bit masks, the frontier a
//      packed list: its data-dependent control flow diverges between the lanes, but no memory or LDS latency sits inside
//      it); every process_sv call it would make is recorded instead (which groups are alive at that moment decides what
//      the call sees);
//   2  the recorded calls, in order: call c of all 64 components side by side (assemble_sv with its dependent loads runs
//      converged), then the bookkeeping of the traversal the call belongs to.
// (History: one wave per component, 64 lanes running the same scalar code, took ~8,900 waves of 127 registers for configs[1] --
// three rounds of ~7 us on the 4,096 wave slots, 37 us; one lane per component with the calls made inside the traversal
// loops, 40 us: 64 walks reach their calls at different iterations and every one of those pays its own round trips.)
constexpr int kWalkLdsLibs = 16, kWalkLdsKeys = 16;  // run constants of up to this many libraries / counter keys live in LDS

__global__ __launch_bounds__(64) void k6_walk_kernel(K6Arrays a) {
    // (32 walking lanes' columns, not 64: with 20 KB per one-wave workgroup a compute unit held 7 waves of this kernel -- 123 registers allow 16 --
    // and the 8,192 waves of a genome share entered over 130 us, profiles/r06_kprof_genome.txt)
    __shared__ uint32_t s_tab[WalkTab::kFields * WalkTab::kStride];
    __shared__ uint32_t s_const[kWalkLdsLibs * (1 + kNumFlags) + kWalkLdsKeys];
    const int lane = threadIdx.x;
    if (a.mirror_in_walk && blockIdx.x == 0) {
        // k6_mirror_kernel's job, done by the first wave of the kernel that follows k6_emit_kernel anyway: the counters are
        // final (kernel boundary), they go to the host's record and the word the host polls is set behind them
        const uint32_t word = mirrored_count_word(a, lane);
        if (lane < (int)(sizeof(StageCounts) / 4)) ((uint32_t*)a.counts_host)[lane] = word;
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && a.flag_groups) *(volatile uint32_t*)a.flag_groups = a.flag_value;
    }
    const WalkTab T{s_tab + lane};
    const int mrp = a.min_read_pair;
    const int nk = a.nkeys;
    const uint32_t period = (uint32_t)a.period;
    KPROF(blockIdx.x, 0);
    // (requested before the constants below are waited for: one round trip for both)
    const uint32_t NR = a.counts->n_regions;
    const int mq_last = a.counts->last_maxq;
    // run constants: in LDS when they are few (every call reads them behind a load it depends on)
    RunConst rc_{a.lib_mean, a.hist, a.key_density, a.p1->covered_ref_len};
    if (a.nlibs <= kWalkLdsLibs) {
        for (int i = lane; i < a.nlibs; i += 64) s_const[i] = __float_as_uint(a.lib_mean[i]);
        for (int i = lane; i < a.nlibs * kNumFlags; i += 64) s_const[kWalkLdsLibs + i] = a.hist[i];
        rc_.lib_mean = (const float*)s_const;
        rc_.hist = s_const + kWalkLdsLibs;
    }
    if (nk <= kWalkLdsKeys) {
        for (int i = lane; i < nk; i += 64) s_const[kWalkLdsLibs * (1 + kNumFlags) + i] = __float_as_uint(a.key_density[i]);
        rc_.key_density = (const float*)(s_const + kWalkLdsLibs * (1 + kNumFlags));
    }
    __builtin_amdgcn_wave_barrier();
    if (s_const[0] != 0x7FFFFFFFu) KPROF(blockIdx.x, 1);
    const PartRec* const P = a.parts;
    if (NR != 0xFFFFFFFFu) KPROF(blockIdx.x, 2);
    // One lane per REGION; the smallest region of a device-walked component (k6_emit_kernel marked it with the component's
    // size) walks it.  Everything the first phase needs is requested at once, the description (indexed by label = this region)
    // before it is known whether the region is such an owner: a dependent round trip costs ~1.3 us here, bytes cost nothing.
    const uint32_t wl = (uint32_t)min(a.walk_lanes, WalkTab::kStride);  // components per wave (the other lanes stay idle: fewer addresses per memory instruction)
    if ((uint32_t)lane >= wl) return;
    for (uint32_t r = blockIdx.x * wl + lane; r < NR; r += gridDim.x * wl) {
        const MemberInfo* D = a.members + (size_t)r * kK6MaxMembers;
        uint4 raw[kK6MaxMembers][7];
#pragma unroll
        for (int i = 0; i < kK6MaxMembers; ++i) {
            const uint4* src = (const uint4*)(D + i);
#pragma unroll
            for (int j = 0; j < 7; ++j) raw[i][j] = src[j];
        }
        const uint32_t ow = a.owners[r];
        // _max_readlen at the flush of this region's window: the value of the candidate that closes there (BreakDancer.cpp:254-259)
        const uint32_t rl0 = (r / period + 1) * period - 1;
        const int mq0 = a.r_rec[min(rl0, a.cap - 1)].maxq;
        const int k = (int)ow;
        if (raw[0][0].x != 0xFFFFFFFEu && raw[3][6].x != 0xFFFFFFFEu) KPROF(blockIdx.x, 3);
        if (k == 0) continue;
        const int mrl0 = rl0 < NR ? mq0 : mq_last;
        uint32_t touch = 0;  // the first words of what the calls will read, requested before the traversal and waited for after it
        // ---- phase 0 ----------------------------------------------------------------------------------------------------
        uint32_t stored = 0;  // bit i: the reads of the member at place i are stored (ReadRegionData.cpp:118-121)
        uint32_t self_has = 0, self_ok = 0, edge_has = 0, edge_ok = 0;  // bit per place / per pair: the group exists; ... and passes the weight gate (-r)
        uint32_t win4 = 0;      // flush window of the member at place i, as the place of the window's first member (2 bits each)
        {
            uint32_t mr[kK6MaxMembers], first[kK6MaxMembers], n_in[kK6MaxMembers], np_all[kK6MaxMembers], np_self[kK6MaxMembers],
                w_self[kK6MaxMembers], st[kK6MaxMembers];
            uint32_t e_lo[kK6MaxMembers][kK6MaxIn], e_w[kK6MaxMembers][kK6MaxIn], e_off[kK6MaxMembers][kK6MaxIn], e_cnt[kK6MaxMembers][kK6MaxIn];
            int32_t rtid[kK6MaxMembers], rstart[kK6MaxMembers], rend[kK6MaxMembers];
            uint32_t rn[kK6MaxMembers], rrev[kK6MaxMembers];
            static_assert(sizeof(MemberInfo) == 7 * 16 && offsetof(MemberInfo, rec) == 4 && offsetof(MemberInfo, np_all) == 40 &&
                              offsetof(MemberInfo, e_lo) == 56 && offsetof(MemberInfo, stored) == 104 && offsetof(RegionRec, first) == 32,
                          "a member's record is fetched as seven 16-byte words");
#pragma unroll
            for (int i = 0; i < kK6MaxMembers; ++i) {
                const uint4 (&q)[7] = raw[i];  // (members past the component's k hold whatever the table held: never used)
                mr[i] = i < k ? q[0].x : 0xFFFFFFFFu;
                rtid[i] = (int32_t)q[0].y; rstart[i] = (int32_t)q[0].z; rend[i] = (int32_t)q[0].w;
                rn[i] = q[1].x; rrev[i] = q[1].y;                      // (q[1].z nonctx, q[1].w nnormal, q[2].x maxq)
                first[i] = q[2].y; np_all[i] = q[2].z; np_self[i] = q[2].w;
                w_self[i] = q[3].x; n_in[i] = i < k ? q[3].y : 0u;
                e_lo[i][0] = q[3].z; e_lo[i][1] = q[3].w; e_lo[i][2] = q[4].x;
                e_w[i][0] = q[4].y; e_w[i][1] = q[4].z; e_w[i][2] = q[4].w;
                e_off[i][0] = q[5].x; e_off[i][1] = q[5].y; e_off[i][2] = q[5].z;
                e_cnt[i][0] = q[5].w; e_cnt[i][1] = q[6].x; e_cnt[i][2] = q[6].y;
                st[i] = q[6].z;
            }
            // members in ascending region order: a member's place is the number of members with a smaller region
            int place[kK6MaxMembers];
#pragma unroll
            for (int i = 0; i < kK6MaxMembers; ++i) {
                place[i] = 0;
#pragma unroll
                for (int q = 0; q < kK6MaxMembers; ++q) place[i] += mr[q] < mr[i] ? 1 : 0;  // (regions differ)
            }
            for (int e = 0; e < 6; ++e) { T.Ebeg(e) = 0; T.Ecnt(e) = 0; T.Ew(e) = 0; T.Eslot(e) = 0; }
#pragma unroll
            for (int i = 0; i < kK6MaxMembers; ++i) {
                if (i < k) {
                    const int pl = place[i];
                    T.rid(pl) = mr[i];
                    T.Sbeg(pl) = first[i] + np_all[i] - np_self[i];
                    T.Scnt(pl) = np_self[i];
                    T.Sw(pl) = w_self[i];
                    T.Sslot(pl) = first[i] + n_in[i];
                    T.Rtid(pl) = (uint32_t)rtid[i]; T.Rstart(pl) = (uint32_t)rstart[i]; T.Rend(pl) = (uint32_t)rend[i]; T.Rn(pl) = rn[i]; T.Rrev(pl) = rrev[i];
                    stored |= (st[i] ? 1u : 0u) << pl;
                    if (np_self[i]) { self_has |= 1u << pl; if ((int)w_self[i] >= mrp) self_ok |= 1u << pl; }
#pragma unroll
                    for (int e = 0; e < kK6MaxIn; ++e) {
                        if ((uint32_t)e < n_in[i]) {
                            int x = -1;  // place of the member that is the group's earlier region (the closure check guarantees there is one)
#pragma unroll
                            for (int q = 0; q < kK6MaxMembers; ++q)
                                if (mr[q] == e_lo[i][e]) x = place[q];
                            if (x >= 0 && x < pl) {
                                const int pi = pair_index(x, pl);
                                T.Ebeg(pi) = first[i] + e_off[i][e];
                                T.Ecnt(pi) = e_cnt[i][e];
                                T.Ew(pi) = e_w[i][e];
                                T.Eslot(pi) = first[i] + (uint32_t)e;
                                if (e_cnt[i][e]) { edge_has |= 1u << pi; if ((int)e_w[i][e] >= mrp) edge_ok |= 1u << pi; }
                            }
                        }
                    }
                }
            }
            // flush windows: members in ascending order, a window = period consecutive region ids
            uint32_t wq[kK6MaxMembers];
#pragma unroll
            for (int pl = 0; pl < kK6MaxMembers; ++pl) {
                wq[pl] = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < kK6MaxMembers; ++i)
                    if (i < k && place[i] == pl) wq[pl] = mr[i] / period;
            }
            int firstp = 0;
#pragma unroll
            for (int pl = 0; pl < kK6MaxMembers; ++pl) {
                if (pl > 0 && wq[pl] != wq[pl - 1]) firstp = pl;
                win4 |= (uint32_t)firstp << (2 * pl);
            }
#pragma unroll
            for (int i = 0; i < kK6MaxMembers; ++i)
                if (i < k) {
                    if (np_all[i]) touch ^= (uint32_t)P[first[i]].key ^ (uint32_t)P[first[i] + np_all[i] - 1].key;
                    touch ^= a.r_pk[(size_t)mr[i] * 2 * nk] ^ a.r_pk[(size_t)mr[i] * 2 * nk + 2 * nk - 1];
                }
        }
        KPROF(blockIdx.x, 4);
        // ---- phase 1 ----------------------------------------------------------------------------------------------------
        int ncalls = 0;
        {
            // registers only: the gates are bit masks, the frontier a packed list (2 bits per vertex, 12 entries as in the general walk)
            uint32_t self_done = 0, edge_done = 0, self_alive = self_has, edge_alive = edge_has;
            // One flush (BreakDancer.cpp:266-346) per window that holds members, in ascending order.  A group is part of
            // the flush of its later region's window; the flush starts traversals first from the members of earlier windows
            // that have a group in it, then from the window's own members, each in ascending order.
            for (int f = 0; f < k;) {
                int fe = f + 1;
                while (fe < k && (int)((win4 >> (2 * fe)) & 3u) == f) ++fe;
                uint32_t visited = 0;
                for (int sv = 0; sv < fe; ++sv) {
                    if (visited & (1u << sv)) continue;
                    int nt = 1;
                    uint32_t tails = (uint32_t)sv;
                    while (nt) {
                        int nn = 0;
                        uint32_t newtails = 0;
                        for (int ti = 0; ti < nt; ++ti) {
                            const int tail = (int)((tails >> (2 * ti)) & 3u);
                            if (visited & (1u << tail)) continue;
                            for (int nb = 0; nb < fe; ++nb) {  // neighbours in ascending order, the vertex itself at its own place
                                int A, B;
                                if (nb == tail) {
                                    if (tail < f) continue;  // its self group belonged to an earlier flush
                                    if (!((self_ok & ~self_done) & (1u << tail))) continue;
                                    self_done |= 1u << tail;
                                    A = tail; B = -1;
                                } else {
                                    const int x = min(nb, tail), y = max(nb, tail), pi = pair_index(x, y);
                                    if (y < f) continue;     // a group of an earlier flush
                                    if (!((edge_ok & ~edge_done) & (1u << pi))) continue;
                                    edge_done |= 1u << pi;
                                    A = x; B = y;
                                }
                                if (nn < 12) { newtails |= (uint32_t)nb << (2 * nn); ++nn; }
                                // the groups the call sees: (A,A), (A,B), (B,B) as far as they are alive and their reads stored;
                                // paired reads leave their regions before any gate (BreakDancer.cpp:363-368)
                                const bool stA = (stored >> A) & 1u, stB = (stored >> (B >= 0 ? B : A)) & 1u;
                                uint32_t g = 0;
                                if ((self_alive & (1u << A)) && stA) { g |= 1u; self_alive &= ~(1u << A); }
                                if (B >= 0) {
                                    const int pi = pair_index(A, B);
                                    if ((edge_alive & (1u << pi)) && stA && stB) { g |= 2u; edge_alive &= ~(1u << pi); }
                                    if ((self_alive & (1u << B)) && stB) { g |= 4u; self_alive &= ~(1u << B); }
                                }
                                // (a call that sees no group fails process_sv's first gate -- no pairs -- before it has any effect)
                                if (g != 0 || mrp <= 0) T.call(ncalls++) = (uint32_t)A | ((uint32_t)(B + 1) << 2) | (g << 5) | ((uint32_t)sv << 8) | ((uint32_t)f << 10);
                            }
                            visited |= 1u << tail;
                        }
                        nt = nn;
                        tails = newtails;
                    }
                }
                f = fe;
            }
        }
        if (nk < 0) a.counts->overflow = touch;  // (never: the requested words only have to arrive)
        KPROF(blockIdx.x, 5);
        // ---- phase 2 ----------------------------------------------------------------------------------------------------
        {
            const GrpRange none{0, 0};
            int cur_f = -1, cur_sv = -1, max_readlen = 0;
            uint32_t W = 0, start = 0, nseq = 0, nsv = 0, nacc_tot = 0, ncn_tot = 0, prev_slot = 0;
            bool from_old = false;
            for (int c = 0; c < ncalls; ++c) {
                const uint32_t d = T.call(c);
                const int A = (int)(d & 3u), B = (int)((d >> 2) & 7u) - 1, sv = (int)((d >> 8) & 3u), f = (int)((d >> 10) & 3u);
                const uint32_t g = (d >> 5) & 7u;
                if (f != cur_f) {
                    // _max_readlen at this window's flush: the value of the candidate that closes there (BreakDancer.cpp:254-259)
                    W = T.rid(f) / period;
                    const uint32_t rl = (W + 1) * period - 1;
                    max_readlen = f == 0 ? mrl0 : (rl < NR ? a.r_rec[rl].maxq : mq_last);
                    nseq = 0;
                }
                if (f != cur_f || sv != cur_sv) {  // the first call of another traversal
                    if (cur_f >= 0 && !from_old && nsv) { a.own_nsv[start] = nsv; a.own_nacc[start] = nacc_tot; a.own_ncn[start] = ncn_tot; }
                    cur_f = f; cur_sv = sv;
                    start = T.rid(sv);
                    from_old = sv < f;
                    nsv = 0; nacc_tot = 0; ncn_tot = 0; prev_slot = 0;
                }
                const uint32_t slot = B >= 0 ? T.Eslot(pair_index(A, B)) : T.Sslot(A);
                const int Bi = B >= 0 ? B : A;
                GrpRange gs[3] = {none, none, none};
                if (g & 1u) gs[0] = GrpRange{T.Sbeg(A), T.Scnt(A)};
                if (g & 2u) gs[1] = GrpRange{T.Ebeg(pair_index(A, Bi)), T.Ecnt(pair_index(A, Bi))};
                if (g & 4u) gs[2] = GrpRange{T.Sbeg(Bi), T.Scnt(Bi)};
                RegionRec recA{}, recB{};  // (the fields assemble_sv reads)
                recA.tid = (int32_t)T.Rtid(A); recA.start = (int32_t)T.Rstart(A); recA.end = (int32_t)T.Rend(A); recA.n = T.Rn(A); recA.rev = T.Rrev(A);
                recB.tid = (int32_t)T.Rtid(Bi); recB.start = (int32_t)T.Rstart(Bi); recB.end = (int32_t)T.Rend(Bi); recB.n = T.Rn(Bi); recB.rev = T.Rrev(Bi);
                const uint32_t rA = T.rid(A), rB = T.rid(Bi);
                // proper-read samples of the two regions (first read: nkeys words, last read: nkeys words)
                const uint32_t* pkA = a.r_pk + (size_t)rA * 2 * nk + nk;
                const uint32_t* pkB = a.r_pk + (size_t)rB * 2 * nk;
                uint32_t nacc = 0, ncn = 0;
                if (assemble_sv(a, rc_, P, rA, B >= 0 ? (int32_t)rB : -1, recA, recB, pkA, pkB, gs, max_readlen, slot, start, true, &nacc, &ncn,
                                c == 0 ? 16384u + blockIdx.x : ~0u)) {
                    if (from_old) {  // placed by its order key: after the earlier windows, before this window's own
                        const uint32_t q = atomicAdd(&a.counts->n_old, 1u);
                        a.old_key[q] = ((uint64_t)(W * period) << kKeyShiftT) | ((uint64_t)start << kKeyShiftStart) | (uint64_t)(nseq & kKeySeqMask);
                        a.old_slot[q] = slot;
                        ++nseq;
                    } else {
                        if (nsv == 0) a.own_first[start] = slot; else a.slot_next[prev_slot] = slot;
                    }
                    prev_slot = slot;
                    ++nsv;
                    nacc_tot += nacc;
                    ncn_tot += ncn;
                }
                if (c == 0) KPROF(blockIdx.x, 6);
            }
            if (cur_f >= 0 && !from_old && nsv) { a.own_nsv[start] = nsv; a.own_nacc[start] = nacc_tot; a.own_ncn[start] = ncn_tot; }
        }
    }
    KPROF(blockIdx.x, 7);
}

// the components of more than kK6MaxMembers regions (its own launch: inside the kernel above its registers and LDS
// would cost the common case a third of its occupancy)
__global__ __launch_bounds__(256) void k6_walk_big_kernel(K6Arrays a) {
    __shared__ BigTab s_big[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t NR = a.counts->n_regions, n = a.counts->n_owners_big;
    for (uint32_t oi = blockIdx.x * 4 + w; oi < n; oi += gridDim.x * 4) {
        walk_big(a, s_big[w], a.owners_big[oi], lane, NR);
        __builtin_amdgcn_wave_barrier();
    }
}

namespace {
__device__ __forceinline__ uint32_t count_below(const uint32_t* v, uint32_t n, uint32_t x) {  // #elements < x, v ascending
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (v[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t count_below64(const uint64_t* v, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (v[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
}  // namespace

// The candidates that are placed by order key -- the host walk's list (sorted, pinned host memory) and the device's
// candidates from traversals started at a vertex of an earlier window (k6_walk_kernel's list, any order) -- merged into
// one list sorted by key, with the running totals of their list entries.  One workgroup of 1,024 threads (k6_insert_kernel): sort the
// device's list (by rank counting up to 1,024 entries, bitonic in LDS up to 2,048; a longer list arrives with its ranks counted by the
// whole GPU, k6_ranksort_kernel, launched behind the device walk so that it runs beside the host's), then every entry finds its place by a
// binary search in the other list (no key occurs in both: a start vertex belongs to one component, and that is walked either here or by the host).
// Round 6: lists of 1,025..8,192 device entries (a GPU's share of a genome has ~6 k) are ranked inside this launch through BUCKETS of the
// keys' flush threshold -- a histogram in LDS, its running sums, every entry compared with its own bucket's few others: 127 us of rank-sort
// launch + one workgroup adding up ten partial ranks per entry became one pass over keys that stay in LDS (profiles/r06_insert_buckets.txt).
// A list whose keys crowd into one bucket (kInsBucketDepth) takes the old bitonic sort.
constexpr uint32_t kInsLds = 2048;
constexpr uint32_t kInsThreads = 1024;
constexpr uint32_t kInsBucketMax = 8192;     // device entries the bucket path takes
constexpr uint32_t kInsCntLds = 10240;       // merged entries whose counts stay in LDS
constexpr uint32_t kInsBuckets = 2048;
constexpr uint32_t kInsBucketDepth = 128;    // entries of one bucket beyond which the bucket path gives up
constexpr uint32_t kInsPerBucketPath = kInsBucketMax / kInsThreads;
constexpr uint32_t kRankGrid = 256;          // workgroups of k6_ranksort_kernel
// how k6_ranksort_kernel cuts its work for a list of nd entries: chunks of 256 entries x slices of the list they are compared against
__device__ __forceinline__ void rank_layout(uint32_t nd, uint32_t* nchunk, uint32_t* nslice, uint32_t* per) {
    *nchunk = (nd + kScanBlock - 1) / kScanBlock;
    *nslice = min(kK6RankSlices, max(1u, kRankGrid / *nchunk));
    *per = (nd + *nslice - 1) / *nslice;
}

struct InsertJob {
    K6Arrays a;
    __device__ void operator()() const;
};

__device__ void InsertJob::operator()() const {
    constexpr uint32_t kThreads = kInsThreads;
    constexpr uint32_t kPer = kInsCntLds / kThreads;  // list entries per thread in the LDS prefix pass
    // One workgroup, nothing else on its compute unit: 130 of gfx950's 160 KB of LDS.  s_ord: the device's keys bucket by bucket (bucket
    // path); the other paths keep their keys, slots and rank-counting copy in it
    __shared__ uint64_t s_ord[kInsBucketMax];
    __shared__ uint32_t s_cnt[kInsCntLds];        // lib_count | cn_count << 16 of the merged list
    __shared__ uint64_t s_hk[kInsLds];
    __shared__ uint32_t s_hist[kInsBuckets + 1];
    __shared__ uint32_t s_ws[2][kThreads / 64];
    __shared__ uint32_t s_crowded;
    uint64_t* const s_dk = s_ord;
    uint32_t* const s_dv = (uint32_t*)(s_ord + kInsLds);
    uint64_t* const s_tmp = s_ord + 2 * kInsLds;
    static_assert(2 * kInsLds + kThreads <= kInsBucketMax, "keys, slots and the rank-counting copy of the smaller paths fit s_ord");
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const uint32_t nh = a.nh;
    const uint32_t nd = a.counts->n_old;
    // the host's keys and counts come over the link: ask for them before anything waits
    uint64_t* hk = nh <= kInsLds ? s_hk : a.hs_key_dev;
    for (uint32_t i = tid; i < nh; i += kThreads) hk[i] = a.hs_key[i];
    uint32_t hc[kInsLds / kThreads];   // (the counts of this thread's host entries on the bucket path: the same round trip over the link as the keys')
#pragma unroll
    for (uint32_t q = 0; q < kInsLds / kThreads; ++q) hc[q] = (nh <= kInsLds && tid + q * kThreads < nh) ? a.hs_cnt[tid + q * kThreads] : 0u;
    const uint32_t n = nh + nd;
    if (n > a.sv_cap) {  // cannot happen: every candidate consumes at least one read pair
        if (tid == 0) { a.counts->overflow = 1; a.counts->n_ins = 0; a.ins_pre_l[0] = 0; a.ins_pre_c[0] = 0; }
        return;
    }
    if (tid == 0) a.counts->n_ins = n;
    if (n == 0) {
        if (tid == 0) { a.ins_pre_l[0] = 0; a.ins_pre_c[0] = 0; }
        return;
    }
    const bool by_rank = nd <= kThreads;
    bool by_bucket = !by_rank && nd <= kInsBucketMax && a.ins_plain != 1;
    const bool small = n <= kInsCntLds;  // the merged list's counts stay in LDS for the running totals
    uint64_t* dk = nullptr;
    uint32_t* dv = nullptr;
    uint32_t my_cnt = 0;              // rank-sort path: lib_count | cn_count << 16 of this thread's device entry
    if (by_bucket) {
        // An entry's bucket is a monotone function of its key (so a bucket's entries lie between its neighbours') that spreads what the keys
        // crowd around: the device's entries all carry a flush threshold T = window x period and differ in their start vertex -- mostly one of
        // the window before (local links), any earlier one for translocations.  Windows x 64 sub-slots by the start vertex, scaled to kInsBuckets.
        const uint32_t period = (uint32_t)max(a.period, 1);
        const uint32_t nwin = a.counts->n_regions / period + 2;
        constexpr uint32_t kSub = 64;
        const uint64_t span = (uint64_t)nwin * kSub + 1;
        auto bucket_of = [&](uint64_t key) {
            const uint32_t T = (uint32_t)(key >> kKeyShiftT), own = (uint32_t)(key >> 33) & 1u, start = (uint32_t)(key >> kKeyShiftStart) & 0x3FFFFFFu;
            const uint32_t W = T / period, r = T - W * period;
            uint32_t sub = kSub;   // (a threshold inside a window, or a candidate of its own window: behind every entry of that flush)
            if (!r && !own) {
                if (start >= T) sub = kSub - 1;
                else if (start + period >= T) sub = kSub / 2 + min(kSub / 2 - 1, (uint32_t)(((uint64_t)(start + period - T) * (kSub / 2)) / period));
                else sub = (uint32_t)(((uint64_t)start * (kSub / 2)) / T);
            }
            return (uint32_t)min((uint64_t)(kInsBuckets - 1), (((uint64_t)W * kSub + sub) * kInsBuckets) / span);
        };
        for (uint32_t i = tid; i <= kInsBuckets; i += kThreads) s_hist[i] = 0;
        if (tid == 0) s_crowded = a.ins_plain == 2 ? 1u : 0u;
        __syncthreads();
        uint64_t key[kInsPerBucketPath];
        uint32_t slot[kInsPerBucketPath], bk[kInsPerBucketPath], at[kInsPerBucketPath], cnt[kInsPerBucketPath];
#pragma unroll
        for (uint32_t k = 0; k < kInsPerBucketPath; ++k) {
            const uint32_t d = tid + k * kThreads;
            key[k] = d < nd ? a.old_key[d] : ~0ull;
            slot[k] = d < nd ? a.old_slot[d] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kInsPerBucketPath; ++k) {
            const uint32_t d = tid + k * kThreads;
            bk[k] = 0; at[k] = 0; cnt[k] = 0;
            if (d < nd) {
                cnt[k] = (uint32_t)a.sv_stage[slot[k]].sv.lib_count | ((uint32_t)a.sv_stage[slot[k]].sv.cn_count << 16);
                bk[k] = bucket_of(key[k]);
                at[k] = atomicAdd(&s_hist[bk[k]], 1u);
                if (at[k] >= kInsBucketDepth) s_crowded = 1u;
            }
        }
        __syncthreads();
        by_bucket = s_crowded == 0;   // (the same answer in every thread)
        if (by_bucket) {
            // running sums of the histogram: s_hist[b] = first place of bucket b in the bucket-ordered list, s_hist[kInsBuckets] = nd
            static_assert(kInsBuckets == 2 * kThreads, "two buckets per thread");
            const uint32_t h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
            const uint32_t inc = wave_incl_scan_t(h0 + h1);
            if (lane == 63) s_ws[0][w] = inc;
            __syncthreads();
            uint32_t off = inc - (h0 + h1);
            for (int q = 0; q < w; ++q) off += s_ws[0][q];
            s_hist[2 * tid] = off; s_hist[2 * tid + 1] = off + h0;
            if (tid == 0) s_hist[kInsBuckets] = nd;
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < kInsPerBucketPath; ++k)
                if (tid + k * kThreads < nd) s_ord[s_hist[bk[k]] + at[k]] = key[k];
            __syncthreads();
            auto below = [&](uint64_t x, uint32_t b) {   // keys of the device's list below x, b = x's bucket
                const uint32_t lo = s_hist[b], hi = s_hist[b + 1];
                uint32_t r = lo;
                for (uint32_t j = lo; j < hi; ++j) r += s_ord[j] < x ? 1u : 0u;
                return r;
            };
#pragma unroll
            for (uint32_t k = 0; k < kInsPerBucketPath; ++k)
                if (tid + k * kThreads < nd) {
                    const uint32_t pos = below(key[k], bk[k]) + count_below64(hk, nh, key[k]);
                    a.ins_T[pos] = (uint32_t)(key[k] >> kKeyShiftT);
                    a.ins_src[pos] = slot[k];
                    if (small) s_cnt[pos] = cnt[k];
                    else { a.ins_pre_l[pos] = cnt[k] & 0xffffu; a.ins_pre_c[pos] = cnt[k] >> 16; }
                }
            static_assert(kInsLds / kThreads == 2, "two prefetched host counts per thread");
            for (uint32_t j = tid; j < nh; j += kThreads) {
                const uint64_t x = hk[j];
                const uint32_t pos = j + below(x, bucket_of(x)), c = nh <= kInsLds ? (j < kThreads ? hc[0] : hc[1]) : a.hs_cnt[j];
                a.ins_T[pos] = (uint32_t)(x >> kKeyShiftT);
                a.ins_src[pos] = 0x80000000u | j;
                if (small) s_cnt[pos] = c;
                else { a.ins_pre_l[pos] = c & 0xffffu; a.ins_pre_c[pos] = c >> 16; }
            }
        }
        __syncthreads();   // (s_ord is the other paths' from here)
    }
    if (by_bucket) {
        // (placed above, the host's entries too)
    } else if (by_rank) {
        // one entry per thread: its rank is the number of smaller keys (all keys differ)
        dk = s_dk; dv = s_dv;
        const uint64_t key = tid < nd ? a.old_key[tid] : ~0ull;
        const uint32_t slot = tid < nd ? a.old_slot[tid] : 0u;
        if (tid < nd) my_cnt = (uint32_t)a.sv_stage[slot].sv.lib_count | ((uint32_t)a.sv_stage[slot].sv.cn_count << 16);
        s_tmp[tid] = key;
        __syncthreads();
        uint32_t rank = 0;
        for (uint32_t q = 0; q < nd; ++q) rank += s_tmp[q] < key ? 1u : 0u;
        if (tid < nd) { dk[rank] = key; dv[rank] = slot; }
        __syncthreads();
        if (tid < nd) {
            const uint32_t pos = rank + count_below64(hk, nh, key);
            a.ins_T[pos] = (uint32_t)(key >> kKeyShiftT);
            a.ins_src[pos] = slot;
            if (small) s_cnt[pos] = my_cnt;
            else { a.ins_pre_l[pos] = my_cnt & 0xffffu; a.ins_pre_c[pos] = my_cnt >> 16; }
        }
    } else if (a.rank_part && nd > kInsLds && nd <= kK6RankSortMax && (nd > kInsBucketMax || a.ins_plain == 1)) {
        // (ranked by the whole GPU in the launch before this one: a bitonic sort of this many keys in HBM by ONE workgroup took half a millisecond,
        // one workgroup per 256 entries counting against the whole list 0.4 ms at 6 k entries -- a wave's compare loop is instruction-bound)
        uint32_t nchunk, nslice, per;
        rank_layout(nd, &nchunk, &nslice, &per);
        for (uint32_t d = tid; d < nd; d += kThreads) {
            uint32_t rank = 0;
            for (uint32_t sl = 0; sl < nslice; ++sl) rank += a.rank_part[(size_t)sl * nd + d];
            a.sorted_key[rank] = a.old_key[d];
            a.sorted_slot[rank] = a.old_slot[d];
        }
        __threadfence();
        __syncthreads();
        dk = a.sorted_key; dv = a.sorted_slot;
        for (uint32_t d = tid; d < nd; d += kThreads) {
            const uint64_t key = dk[d];
            const uint32_t pos = d + count_below64(hk, nh, key), slot = dv[d];
            const uint32_t cnt = (uint32_t)a.sv_stage[slot].sv.lib_count | ((uint32_t)a.sv_stage[slot].sv.cn_count << 16);
            a.ins_T[pos] = (uint32_t)(key >> kKeyShiftT);
            a.ins_src[pos] = slot;
            if (small) s_cnt[pos] = cnt;
            else { a.ins_pre_l[pos] = cnt & 0xffffu; a.ins_pre_c[pos] = cnt >> 16; }
        }
    } else {
        uint32_t m = 1;
        while (m < nd) m <<= 1;
        if (m <= kInsLds) {
            dk = s_dk; dv = s_dv;
            for (uint32_t i = tid; i < m; i += kThreads) { dk[i] = i < nd ? a.old_key[i] : ~0ull; dv[i] = i < nd ? a.old_slot[i] : 0u; }
        } else {
            dk = a.old_key; dv = a.old_slot;
            for (uint32_t i = nd + tid; i < m; i += kThreads) dk[i] = ~0ull;
        }
        __syncthreads();
        for (uint32_t ksz = 2; ksz <= m; ksz <<= 1)
            for (uint32_t j = ksz >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < m / 2; t += kThreads) {
                    const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    const bool up = (lo & ksz) == 0;
                    const uint64_t x = dk[lo], y = dk[hi];
                    if ((x > y) == up) {
                        dk[lo] = y; dk[hi] = x;
                        const uint32_t vx = dv[lo];
                        dv[lo] = dv[hi]; dv[hi] = vx;
                    }
                }
                __syncthreads();
            }
        for (uint32_t d = tid; d < nd; d += kThreads) {
            const uint64_t key = dk[d];
            const uint32_t pos = d + count_below64(hk, nh, key), slot = dv[d];
            const uint32_t cnt = (uint32_t)a.sv_stage[slot].sv.lib_count | ((uint32_t)a.sv_stage[slot].sv.cn_count << 16);
            a.ins_T[pos] = (uint32_t)(key >> kKeyShiftT);
            a.ins_src[pos] = slot;
            if (small) s_cnt[pos] = cnt;
            else { a.ins_pre_l[pos] = cnt & 0xffffu; a.ins_pre_c[pos] = cnt >> 16; }
        }
    }
    for (uint32_t j = tid; j < nh && !by_bucket; j += kThreads) {
        const uint64_t key = hk[j];
        const uint32_t pos = j + count_below64(dk, nd, key), cnt = a.hs_cnt[j];
        a.ins_T[pos] = (uint32_t)(key >> kKeyShiftT);
        a.ins_src[pos] = 0x80000000u | j;
        if (small) s_cnt[pos] = cnt;
        else { a.ins_pre_l[pos] = cnt & 0xffffu; a.ins_pre_c[pos] = cnt >> 16; }
    }
    __syncthreads();
    // exclusive running totals, the grand totals at [n]
    if (small) {
        uint32_t l[kPer], c[kPer], sl = 0, sc = 0;
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const uint32_t i = tid * kPer + q;
            const uint32_t v = i < n ? s_cnt[i] : 0u;
            l[q] = v & 0xffffu; c[q] = v >> 16;
            sl += l[q]; sc += c[q];
        }
        const uint32_t il = wave_incl_scan_t(sl), ic = wave_incl_scan_t(sc);
        if (lane == 63) { s_ws[0][w] = il; s_ws[1][w] = ic; }
        __syncthreads();
        uint32_t ol = il - sl, oc = ic - sc, tl = 0, tc = 0;
        for (int q = 0; q < (int)kThreads / 64; ++q) {
            if (q < w) { ol += s_ws[0][q]; oc += s_ws[1][q]; }
            tl += s_ws[0][q]; tc += s_ws[1][q];
        }
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const uint32_t i = tid * kPer + q;
            if (i < n) { a.ins_pre_l[i] = ol; a.ins_pre_c[i] = oc; }
            ol += l[q]; oc += c[q];
        }
        if (tid == 0) { a.ins_pre_l[n] = tl; a.ins_pre_c[n] = tc; }
        return;
    }
    uint32_t carry_l = 0, carry_c = 0;
    for (uint32_t base = 0; base < n; base += kThreads) {
        const uint32_t i = base + tid;
        const uint32_t l = i < n ? a.ins_pre_l[i] : 0u, c = i < n ? a.ins_pre_c[i] : 0u;
        const uint32_t il = wave_incl_scan_t(l), ic = wave_incl_scan_t(c);
        if (lane == 63) { s_ws[0][w] = il; s_ws[1][w] = ic; }
        __syncthreads();
        uint32_t ol = carry_l, oc = carry_c, tl = 0, tc = 0;
        for (int q = 0; q < (int)kThreads / 64; ++q) {
            if (q < w) { ol += s_ws[0][q]; oc += s_ws[1][q]; }
            tl += s_ws[0][q]; tc += s_ws[1][q];
        }
        if (i < n) { a.ins_pre_l[i] = ol + il - l; a.ins_pre_c[i] = oc + ic - c; }
        carry_l += tl; carry_c += tc;
        __syncthreads();
    }
    if (tid == 0) { a.ins_pre_l[n] = carry_l; a.ins_pre_c[n] = carry_c; }
}

// The final table, one launch: a decoupled look-back scan over the start vertices gives every workgroup the place of its
// vertices' candidates in the table and in the two flat lists; the workgroup then finishes exactly that slice of the table --
// K5 (the Poisson log tail of every (candidate, library) term, bdx_poisson.h), their Kahan-compensated sum
// (BreakDancer.cpp:56-69), PhredQ = min(99, int(-10 logp / ln 10 + 0.5)) (:459-465, NaN -> INT_MIN as cvttsd2si does),
// printed = PhredQ > -y -- and writes it to pinned host memory.  Vertex i first places the candidates of the inserted list
// whose threshold is i, then those of the traversal that started at i in i's own window.
// A wave takes 64 candidates at a time: their records go through LDS, get their scores there, and leave as one contiguous
// 6 KiB write (single scattered stores over PCIe are several times slower); the list entries of a wave's candidates are
// neighbours too.  (Three launches before: block sums, rescan + placement into HBM lists, score kernel; 52 us -> see DESIGN.md.)
constexpr int kSvWords = sizeof(SvOut) / 4;
constexpr uint32_t kFinT = 1024;     // thresholds of the inserted list kept in LDS (more: searched in HBM)

// The device's insertion list (order keys of the candidates whose traversal started in an earlier flush window, any order) ranked by
// ALL of the GPU: every entry's rank is the number of keys below its own.  n^2 comparisons -- 3.5e7 at the 6 k entries of a GPU's share of
// a genome, microseconds on 256 compute units IF they are spread: a workgroup takes 256 entries and ONE SLICE of the list (up to 16 slices
// while there are fewer chunks than workgroups), tiles of the slice pass through LDS, a broadcast read per comparison, and leaves the
// partial ranks in rank_part[slice][entry]; k6_insert_kernel adds the slices up and places the entries.  Its own launch, enqueued behind the
// device walk: it runs while the host walks its share (until round 5 these workgroups rode in k6_insert_kernel's launch, whose last
// workgroup spun until they were through: one slice per entry, 0.4 ms, on the critical path -- and a wait HIP does not promise to end).
constexpr uint32_t kRankTile = 2048;
__global__ __launch_bounds__(kScanBlock) void k6_ranksort_kernel(K6Arrays a) {
    __shared__ uint64_t s_k[kRankTile];
    const uint32_t nd = a.counts->n_old;
    if (nd <= kInsLds || nd > kK6RankSortMax) return;
    if (nd <= kInsBucketMax && a.ins_plain != 1) return;   // (k6_insert_kernel ranks these through its buckets)
    uint32_t nchunk, nslice, per;
    rank_layout(nd, &nchunk, &nslice, &per);
    for (uint32_t item = blockIdx.x; item < nchunk * nslice; item += gridDim.x) {   // (whole workgroups stay in the loop: barriers inside)
        const uint32_t chunk = item % nchunk, slice = item / nchunk;
        const uint32_t i = chunk * kScanBlock + threadIdx.x;
        const uint64_t key = i < nd ? a.old_key[i] : ~0ull;
        const uint32_t k0 = min(nd, slice * per), k1 = min(nd, k0 + per);
        uint32_t rank = 0;
        for (uint32_t t0 = k0; t0 < k1; t0 += kRankTile) {
            const uint32_t cnt = min(kRankTile, k1 - t0);
            __syncthreads();
            for (uint32_t q = threadIdx.x; q < cnt; q += kScanBlock) s_k[q] = a.old_key[t0 + q];
            __syncthreads();
            for (uint32_t q = 0; q < cnt; ++q) {
                const uint64_t k = s_k[q];
                rank += (k < key || (k == key && t0 + q < i)) ? 1u : 0u;   // (keys of one run differ; the index settles it if they ever do not)
            }
        }
        if (i < nd) a.rank_part[(size_t)slice * nd + i] = rank;
    }
}

__global__ __launch_bounds__(kInsThreads) void k6_insert_kernel(K6Arrays a) { InsertJob{a}(); }

// Round 6: two launches again.  The single launch gave every workgroup the candidates of ITS 256 vertices to finish: at a GPU's share of a
// genome the densest workgroups had three 64-candidate passes per wave to work through, one after the other -- gather, K5, scores, ~25 us a
// pass -- while most had none left: 110-125 us for 55 k candidates, 75 us for the 6.5 k of a -t run (profiles/r06_table_split.txt).
//   k6_place_kernel   the look-back scan over the start vertices; every vertex writes where its candidates come from (staging slot or host
//                     entry), their first entries in the two flat lists and, for the merge of a sharded run's tables, the vertex itself,
//                     at the candidates' places in the table
//   k6_score_kernel   a wave per 64 CONSECUTIVE candidates of the table, whichever vertices they belong to: records through LDS, K5,
//                     scores, one contiguous write
__global__ __launch_bounds__(kScanBlock) void k6_place_kernel(K6Arrays a) {
    __shared__ U4 s_ws[kScanBlock / 64];
    __shared__ uint32_t s_prefix[4];
    __shared__ uint32_t s_T[kFinT];
    __shared__ uint32_t s_ex[3][kScanBlock];
    __shared__ uint32_t s_lohi[2];
    const uint32_t tid = threadIdx.x;
    const int w = tid >> 6;
    const uint32_t n = a.counts->n_regions;
    const uint32_t bid = blockIdx.x, base = bid * kScanBlock;
    if (w == 0 && bid < 16384u) KPROF(32768u + bid, 0);
    if (bid == 0 && tid == 0 && (uint64_t)gridDim.x * kScanBlock < n) {   // (the host sized the launch for fewer regions than there are: cannot happen, and must not pass)
        a.counts->overflow = 1;
        if (a.counts_host2) a.counts_host2->overflow = 1;
    }
    if (base >= n) return;
    const uint32_t j = base + tid;
    const bool valid = j < n;
    const U4 e = valid ? U4{a.own_nsv[j], a.own_nacc[j], a.own_ncn[j], 0u} : U4{0u, 0u, 0u, 0u};
    const uint32_t slot0 = valid ? a.own_first[j] : 0u;  // (meaningful when the vertex has candidates)
    const uint32_t nh = a.counts->n_ins;
    const uint32_t h_l = a.ins_pre_l[nh], h_c = a.ins_pre_c[nh];
    const bool t_lds = nh <= kFinT;
    if (t_lds)
        for (uint32_t i = tid; i < nh; i += kScanBlock) s_T[i] = a.ins_T[i];
    U4 tot;
    const U4 inc_local = block_incl_scan(e, s_ws, &tot);  // (its barriers also publish s_T)
    if (w == 0 && bid < 16384u) KPROF(32768u + bid, 1);
    const U4 carry = lookback_exclusive<U4>(tot, a.lb_state, a.lb_stamp, bid, s_prefix);
    if (w == 0 && bid < 16384u) KPROF(32768u + bid, 2);
    const U4 inc = carry + inc_local;
    const uint32_t* Tt = t_lds ? s_T : a.ins_T;
    uint32_t hb0 = 0, hb1 = 0;
    if (valid && nh) {
        hb0 = count_below(Tt, nh, j);
        hb1 = hb0;  // (entries with threshold j follow each other: a short look ahead, then a second search -- a -t run's vertices have dozens)
        for (int g = 0; hb1 < nh && Tt[hb1] == j; ++g) {
            ++hb1;
            if (g == 3) { hb1 = count_below(Tt, nh, j + 1); break; }
        }
    }
    const uint32_t ex_sv = inc.x - e.x, ex_l = inc.y - e.y, ex_c = inc.z - e.z;
    if (valid && j == n - 1) {
        const uint32_t tsv = inc.x + nh, tl = inc.y + h_l, tc = inc.z + h_c;
        a.counts->n_sv_dev = tsv; a.counts->n_terms_dev = tl; a.counts->n_cn_dev = tc;
        if (a.counts_host2) { a.counts_host2->n_sv_dev = tsv; a.counts_host2->n_terms_dev = tl; a.counts_host2->n_cn_dev = tc; a.counts_host2->n_old = a.counts->n_old; }
        if (tsv > a.sv_cap || tl > a.term_cap || tc > a.cn_cap) {
            a.counts->overflow = 1;
            if (a.counts_host2) a.counts_host2->overflow = 1;
        }
    }
    // (nothing is placed past a capacity: the last vertex reports that, and the score kernel stays away from the table)
    const bool wg_ok = !(carry.x + tot.x + nh > a.sv_cap || carry.y + tot.y + h_l > a.term_cap || carry.z + tot.z + h_c > a.cn_cap);
    if (!wg_ok) return;   // (the same answer in every thread of the workgroup)
    // The inserted candidates that come right before a vertex's own: the workgroup's vertices' entries of the inserted list lie together,
    // [lo, hi), and are placed by ALL its threads -- a -t run hangs every candidate of a flush on ONE vertex (~50 at a genome share), and that
    // vertex's thread placed them one dependent load after the other: 52 us of a 75 us launch
    s_ex[0][tid] = ex_sv; s_ex[1][tid] = ex_l; s_ex[2][tid] = ex_c;
    if (tid == 0) { s_lohi[0] = nh ? count_below(Tt, nh, base) : 0u; s_lohi[1] = nh ? count_below(Tt, nh, min(n, base + kScanBlock)) : 0u; }
    __syncthreads();
    for (uint32_t jj = s_lohi[0] + tid; jj < s_lohi[1]; jj += kScanBlock) {
        const uint32_t v = Tt[jj], vt = v - base;   // (its vertex: one of this workgroup's)
        const uint32_t pos = s_ex[0][vt] + jj;
        a.sv_src[pos] = a.ins_src[jj];
        a.sv_begin[pos] = make_uint2(s_ex[1][vt] + a.ins_pre_l[jj], s_ex[2][vt] + a.ins_pre_c[jj]);
        if (a.sv_vx) a.sv_vx[pos] = v;
    }
    if (!valid || !e.x) return;
    uint32_t lb = ex_l + a.ins_pre_l[hb1], cb = ex_c + a.ins_pre_c[hb1], slot = slot0;
    for (uint32_t q = 0; q < e.x; ++q) {
        const uint32_t pos = ex_sv + hb1 + q;
        a.sv_src[pos] = slot;
        a.sv_begin[pos] = make_uint2(lb, cb);
        if (a.sv_vx) a.sv_vx[pos] = j;
        if (q + 1 < e.x) {  // (a single candidate's entries are the vertex's totals: nothing to look up)
            lb += (uint32_t)a.sv_stage[slot].sv.lib_count;
            cb += (uint32_t)a.sv_stage[slot].sv.cn_count;
            slot = a.slot_next[slot];
        }
    }
}

__global__ __launch_bounds__(kScanBlock) void k6_score_kernel(K6Arrays a, double ln10, int score_threshold, int with_scores) {
    __shared__ uint32_t s_rec[kScanBlock / 64][64 * kSvWords];
    __shared__ uint32_t s_fromw[kScanBlock / 64][64];
    __shared__ uint32_t s_printed;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const uint32_t bid = blockIdx.x;
    const uint32_t kp = 32768u + 16384u + (bid * 4 + w < 16384u ? bid * 4 + w : 16383u);
    (void)kp;
    KPROF(kp, 0);
    if (tid == 0) s_printed = 0;
    __syncthreads();
    const uint32_t n_sv = a.counts->overflow ? 0u : a.counts->n_sv_dev;
    for (uint32_t c0 = (bid * (kScanBlock / 64) + (uint32_t)w) * 64u; c0 < n_sv; c0 += gridDim.x * kScanBlock) {
        const uint32_t cnt = min(64u, n_sv - c0);
        const bool act = (uint32_t)lane < cnt;
        uint32_t* rec = s_rec[w];
        const uint32_t from = act ? a.sv_src[c0 + lane] : 0u;
        const uint2 bg = act ? a.sv_begin[c0 + lane] : make_uint2(0u, 0u);
        s_fromw[w][lane] = from;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < cnt * kSvWords; i += 64) {  // gather the records from the staging slots / the host's list
            const uint32_t sv = i / kSvWords, wd = i - sv * kSvWords;
            const uint32_t f = s_fromw[w][sv];
            const uint32_t* src = (f & 0x80000000u) ? (const uint32_t*)(a.hs_rec + (f & 0x7FFFFFFFu)) : (const uint32_t*)(a.sv_stage + f);
            rec[i] = src[wd];
        }
        __builtin_amdgcn_wave_barrier();
        KPROF(kp, 3);
        SvOut* o = (SvOut*)rec + (act ? lane : 0);
        const bool hosted = (from & 0x80000000u) != 0;
        const int32_t nl = act ? o->sv.lib_count : 0, ncn = act ? o->sv.cn_count : 0;
        const int32_t l0 = o->sv.lib_begin, k0 = o->sv.cn_begin;  // (a host candidate's entries in the host's lists)
        if (act) { o->sv.lib_begin = (int32_t)bg.x; o->sv.cn_begin = (int32_t)bg.y; }
        int32_t max_nl = nl, max_ncn = ncn;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { max_nl = max(max_nl, __shfl_xor(max_nl, off)); max_ncn = max(max_ncn, __shfl_xor(max_ncn, off)); }
        // one candidate per lane: K5 for its terms (all 64 lanes take part: a term with a long series is summed by the whole
        // wave); the entries of the two flat lists go straight to the host's arrays
        double logp = 0.0, err = 0.0;
        for (int32_t q = 0; q < max_nl; ++q) {
            const bool a2 = q < nl;
            double lam = 1.0;
            int32_t k = 0, li = 0;
            if (a2) {
                if (hosted) { lam = a.hs_lambda[l0 + q]; k = a.hs_lib_pairs[l0 + q]; li = a.hs_lib_index[l0 + q]; }
                else { const LibStage t = a.lib_stage[(size_t)from * a.lib_stride + q]; lam = t.lambda; k = t.rc; li = t.lib; }
            }
            const double lt = poisson_term(lam, k, a2, lane);
            if (a2) {
                a.lib_index[bg.x + q] = li;
                a.lib_pairs[bg.x + q] = k;
                if (a.ltail_host) a.ltail_host[bg.x + q] = lt;
                const double tmp_a = __dsub_rn(lt, err);
                const double tmp_b = __dadd_rn(logp, tmp_a);
                err = __dsub_rn(__dsub_rn(tmp_b, logp), tmp_a);
                logp = tmp_b;
            }
        }
        for (int32_t t = 0; t < max_ncn; ++t) {
            if (t < ncn) {
                int32_t key;
                float value;
                if (hosted) { key = a.hs_cn_key[k0 + t]; value = a.hs_cn_value[k0 + t]; }
                else { const CnStage cn = a.cn_stage[(size_t)from * a.nkeys + t]; key = cn.key; value = cn.value; }
                a.cn_key[bg.y + t] = key;
                a.cn_value[bg.y + t] = value;
            }
        }
        bool pr = false;
        if (act && with_scores) {
            const double phred_tmp = __ddiv_rn(__dmul_rn(-10.0, logp), ln10);
            const double r = __dadd_rn(phred_tmp, 0.5);
            int phred;
            if (phred_tmp > 99.0) phred = 99;
            else if (r != r || r >= 2147483648.0 || r <= -2147483649.0) phred = INT32_MIN;
            else phred = (int)r;
            pr = phred > score_threshold;
            o->sv.logp = logp;
            o->sv.score = phred;
            o->sv.printed = pr ? 1 : 0;
        }
        if (a.sv_key && act) {   // placed at vertex T: before T's own candidates unless the traversal started at T itself
            const unsigned long long T = a.sv_vx[c0 + lane], st = o->start;
            a.sv_key[c0 + lane] = (T << kKeyShiftT) | (T == st ? 1ull << 33 : 0ull) | (st << kKeyShiftStart);
        }
        const uint32_t npr = (uint32_t)__popcll(__ballot(pr));
        if (lane == 0 && npr) atomicAdd(&s_printed, npr);
        KPROF(kp, 4);
        __builtin_amdgcn_wave_barrier();
        if (a.wire_rows) {
            // the row as it crosses the link: 12 words (SvWire).  Every lane takes its own record out of the slice, then the packed rows
            // go into the slice's front and leave as one contiguous block
            SvWire wr{};
            if (act) {
                wr.pos[0] = o->sv.pos[0]; wr.pos[1] = o->sv.pos[1]; wr.region[0] = o->sv.region[0]; wr.region[1] = o->sv.region[1];
                wr.size = o->sv.size; wr.score = o->sv.score; wr.num_reads = o->sv.num_reads; wr.allele_frequency = o->sv.allele_frequency;
                wr.logp = o->sv.logp; wr.start = o->start;
                wr.bits = ((uint32_t)o->sv.lib_count & 255u) | (((uint32_t)o->sv.cn_count & 255u) << 8) | (((uint32_t)o->sv.flag & 15u) << 16) |
                          ((o->grp_mask & 7u) << 20) | ((o->sv.printed ? 1u : 0u) << 23);
            }
            __builtin_amdgcn_wave_barrier();
            if (act) {
                uint32_t* q = rec + lane * kWireWords;
                const uint32_t* src = (const uint32_t*)&wr;
#pragma unroll
                for (int k = 0; k < kWireWords; ++k) q[k] = src[k];
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t* dst = (uint32_t*)a.sv_out + (size_t)c0 * kWireWords;
            for (uint32_t i = lane; i < cnt * kWireWords; i += 64) dst[i] = rec[i];
        } else {
            uint32_t* dst = (uint32_t*)(a.sv_out + c0);
            for (uint32_t i = lane; i < cnt * kSvWords; i += 64) dst[i] = rec[i];
        }
        __builtin_amdgcn_wave_barrier();  // (the wave's next 64 candidates reuse the slice)
    }
    __syncthreads();
    if (tid == 0) {
        if (s_printed) atomicAdd(&a.counts->n_printed, s_printed);
        // every workgroup leaves its count in the host's array (one plain store; the host adds them up): no follow-up launch
        // that copies the total (system-scope atomics on host memory are ~1 us each and serialise)
        if (a.printed_host) a.printed_host[bid] = s_printed;
    }
    KPROF(kp, 5);
}

// the end of the run: printed count and the word the host polls
__global__ __launch_bounds__(64) void k6_done_kernel(K6Arrays a) {
    if (threadIdx.x == 0) {
        if (a.counts_host2) a.counts_host2->n_printed = a.counts->n_printed;
        if (a.flag_done) {
            __threadfence_system();
            *(volatile uint32_t*)a.flag_done = a.flag_value;
        }
    }
}

__global__ __launch_bounds__(256) void k6_scratch_init_kernel(uint32_t* out_deg, uint32_t cap) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < cap; j += gridDim.x * 256) {
        out_deg[j] = 0;
        out_deg[(size_t)cap + j] = j;
        out_deg[2 * (size_t)cap + j] = 0;
        out_deg[3 * (size_t)cap + j] = 0;
        out_deg[4 * (size_t)cap + j] = 0;
        out_deg[5 * (size_t)cap + j] = 0;
    }
}

void launch_k6_scratch_init(uint32_t* out_deg, uint32_t cap, hipStream_t s) {
    if (!cap) return;
    hipLaunchKernelGGL(k6_scratch_init_kernel, dim3(std::min<uint32_t>((cap + 255) / 256, 4096u)), dim3(256), 0, s, out_deg, cap);
}

void launch_k6_pairs(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s) {
    if (n_anom_host == 0) return;
    // regions <= anomalous reads, typically a tenth of them: about one wave per region, a grid-stride loop for the rest
    const uint32_t gp = std::min<uint32_t>((n_anom_host / 8 + 3) / 4 + 1, 16384u);
    hipLaunchKernelGGL(k6_pairs_kernel, dim3(gp), dim3(256), 0, s, a);
}

void launch_k6_components(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s) {
    if (n_anom_host == 0) return;
    // one thread per region, a wave per workgroup: the regions are about a tenth of the grid's upper bound, and with 256 of them per
    // workgroup the ~12 k regions of configs[1] would keep 47 of the 256 compute units busy -- each with four waves' worth of
    // scattered requests (~64 address cycles per memory instruction) through one address pipeline
    constexpr uint32_t kRegionThreads = 64;  // (whole waves: the emit step's reservations are wave-aggregated; measured 256 / 128 / 64: step 0.2749 / 0.2730 / 0.2711 ms)
    const uint32_t grs = (n_anom_host + kRegionThreads - 1) / kRegionThreads;
    if (!a.force_host)
        for (int i = 1; i < a.label_rounds; ++i) hipLaunchKernelGGL(k6_label_kernel, dim3(grs), dim3(kRegionThreads), 0, s, a);  // round 1: k6_pairs
    hipLaunchKernelGGL(k6_classify_kernel, dim3(grs), dim3(kRegionThreads), 0, s, a);
    hipLaunchKernelGGL(k6_emit_kernel, dim3(grs), dim3(kRegionThreads), 0, s, a);
    if (a.counts_host && !a.mirror_in_walk) hipLaunchKernelGGL(k6_mirror_kernel, dim3(1), dim3(64), 0, s, a);
}

void launch_k6_groups(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s) {
    launch_k6_pairs(a, n_anom_host, s);
    launch_k6_components(a, n_anom_host, s);
}

void launch_k6_walk(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s) {
    if (n_anom_host == 0 || a.force_host) return;
    const uint32_t gp = std::min<uint32_t>((n_anom_host / 8 + 3) / 4 + 1, 16384u);
    hipLaunchKernelGGL(k6_walk_kernel, dim3(std::min<uint32_t>(n_anom_host / (uint32_t)a.walk_lanes / 4 + 1, 8192u)), dim3(64), 0, s, a);  // a lane per region, grid-stride: regions are typically a tenth of the reads
    if (a.big_walk) hipLaunchKernelGGL(k6_walk_big_kernel, dim3(gp / 8 + 1), dim3(256), 0, s, a);
    // (the ranks of the device's insertion list, should it be long: beside the host's walk.  Small inputs do without the launch)
    if (a.rank_part) hipLaunchKernelGGL(k6_ranksort_kernel, dim3(kRankGrid), dim3(kScanBlock), 0, s, a);
}

// workgroups of k6_score_kernel (a wave per 64 candidates, grid-stride: the host does not know how many there are) == entries of
// K6Arrays::printed_host
uint32_t k6_score_grid(const K6Arrays& a) { return std::max(1u, std::min(1024u, (a.sv_cap + kScanBlock - 1) / kScanBlock)); }

// the merged list of the candidates that are placed by key, then the table itself
void launch_k6_table(const K6Arrays& a, uint32_t n_anom_host, double ln10, int score_threshold, int with_scores, hipStream_t s) {
    if (n_anom_host) {
        hipLaunchKernelGGL(k6_insert_kernel, dim3(1), dim3(kInsThreads), 0, s, a);
        // (the placement's launch: for the regions the pair groups' report named where the host has read it, for their upper bound otherwise)
        hipLaunchKernelGGL(k6_place_kernel, dim3(scan_grid(a.fin_regions ? a.fin_regions : a.cap, 1)), dim3(kScanBlock), 0, s, a);
        hipLaunchKernelGGL(k6_score_kernel, dim3(k6_score_grid(a)), dim3(kScanBlock), 0, s, a, ln10, score_threshold, with_scores);
    }
    if (!a.printed_host || a.flag_done) hipLaunchKernelGGL(k6_done_kernel, dim3(1), dim3(64), 0, s, a);
}

}  // namespace bdx

// (bdx_warm_up: the HIP runtime loads a translation unit's device code at the first launch of any of its kernels)
__global__ void k6_noop_kernel() {}
namespace bdx { void warm_k6(hipStream_t s) { hipLaunchKernelGGL(k6_noop_kernel, dim3(1), dim3(64), 0, s); } }

#ifdef BDX_KPROF
extern "C" int bdx_debug_kprof(unsigned long long* out, size_t n) {
    const int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bdx::g_kprof), n * sizeof(unsigned long long));
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(bdx::g_kprof)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 8 * 65536);  // next run starts clean
    return rc;
}
#endif
