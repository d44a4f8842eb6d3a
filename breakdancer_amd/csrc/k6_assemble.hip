// K6 -- region x region pair groups, component classification and SV assembly on the device.
//
// Replaces, for the components of the region graph that need no real traversal (reference file:line under
// src/lib/breakdancer):
//   ReadRegionData.cpp:108-113   edge weights: pairs per (region, region)
//   BreakDancer.cpp:266-346      build_connection: a single region with a self edge, or two regions of one flush
//                                window joined by one edge, are visited in a fixed order (A's self edge, the edge
//                                A-B, then B's self edge), so the whole component is one thread's straight-line code
//   BreakDancer.cpp:348-497      process_sv: gates, breakpoints, copy number, size, score inputs
//   SvBuilder.cpp:18-118         dominant flag, per-library counts of the second-observed mates, positions
//
// Region ids increase with the stream, and a pair is keyed by the region of its second-observed mate, so the
// pairs of region r sit in r's own slice of the compact read list: one wave sorts and run-length merges them
// in registers (k6_pairs_kernel).  Every component that is not of the two shapes above -- and any region with more
// reads than a wave sorts at once -- is handed to the host walk (bdx_walk.cpp) as a list of pair groups.
// float32 / float64 operations are spelled with the round-to-nearest intrinsics so that no fused multiply-add can
// change a result against the host code (and the oracle).
#include <algorithm>

#include "bdx_k3.h"

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

struct GrpRange {
    uint32_t beg, cnt;  // cnt == 0: group absent, already consumed, or its reads were never stored
};

__device__ __forceinline__ bool region_stored(const RegionRec& R, const K6Arrays& a) {  // ReadRegionData.cpp:118-121
    const int valid = a.chr_restricted ? (int)R.nonctx : (int)R.n;
    return valid >= a.min_read_pair;
}

}  // namespace

// One wave per accepted region: its pairs (second-observed mate in the region) -> sorted, merged (lo, flag, lib) parts.
__global__ __launch_bounds__(256) void k6_pairs_kernel(K6Arrays a) {
    const int lane = threadIdx.x & 63;
    const uint32_t nwaves = gridDim.x * 4;
    const uint32_t NR = a.counts->n_regions;
    const uint32_t mrp = (uint32_t)max(a.min_read_pair, 0);
    for (uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < NR; r += nwaves) {
        const RegionRec rr = a.r_rec[r];
        const uint32_t first = rr.first, n = rr.n;
        const bool big = n > 64;
        RegSum rs{};
        rs.big = big ? 1u : 0u;
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const uint32_t i = c0 + lane;
            uint64_t key = ~0ull;
            uint32_t is = 0, lo = 0;
            bool has = false;
            if (i < n) {
                const uint32_t j = first + i;
                const int32_t p = a.partner[j];
                if (p >= 0 && (uint32_t)p < j) {  // j is the second-observed mate (SvBuilder.cpp:101-118)
                    lo = (uint32_t)a.region_of[p];
                    const uint32_t m = a.meta[j];
                    key = ((uint64_t)lo << 12) | ((uint64_t)meta_lib(m) << 4) | (uint64_t)meta_flag(m);
                    is = (uint32_t)a.isize[j];
                    has = true;
                }
            }
            const uint64_t hasmask = __ballot(has);
            if (!hasmask) continue;
            rs.n_pairs += (uint32_t)__popcll(hasmask);
            uint32_t eq_before = 0, cnt = 0, sum = 0, gw = 0;  // gw: pairs of my (lo, r) group = its edge weight
            for (uint64_t mm = hasmask; mm; mm &= mm - 1) {
                const int t = __builtin_ctzll(mm);
                const uint64_t kt = readlane64(key, t);
                const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)is, t);
                const bool eq = kt == key;
                cnt += eq ? 1u : 0u;
                sum += eq ? it : 0u;
                eq_before += (eq && t < lane) ? 1u : 0u;
                gw += ((kt >> 12) == (key >> 12)) ? 1u : 0u;
            }
            const bool leader = has && eq_before == 0;
            const uint64_t lmask = __ballot(leader);
            uint32_t rank = 0;
            bool lo_first = true;
            for (uint64_t mm = lmask; mm; mm &= mm - 1) {
                const int t = __builtin_ctzll(mm);
                const uint64_t kt = readlane64(key, t);
                if (kt < key) {
                    ++rank;
                    if ((kt >> 12) == (key >> 12)) lo_first = false;
                }
            }
            const bool gleader = leader && lo_first;  // one lane per (lo, r) group
            rs.w_self += (uint32_t)__popcll(__ballot(has && lo == r));
            if (!big) {
                // groups below the weight gate are skipped by every try_edge and never reach process_sv: inert
                const bool strong = gw >= mrp;
                const uint64_t gl_in = __ballot(gleader && lo < r && strong);
                if (gleader && lo < r && strong) {
                    atomicAdd(&a.out_deg[lo], 1u);
                    a.out_hi[lo] = r;
                }
                rs.np_all = (uint32_t)__popcll(lmask);
                rs.np_self = (uint32_t)__popcll(__ballot(leader && lo == r));
                rs.np_emit = (uint32_t)__popcll(__ballot(leader && (lo == r || strong)));
                rs.n_weak = (uint32_t)__popcll(__ballot(gleader && lo < r && !strong));
                rs.n_in = (uint32_t)__popcll(gl_in);
                if (gl_in) {
                    const int t = __builtin_ctzll(gl_in);
                    rs.in_lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, t);
                    rs.w_in = (uint32_t)__builtin_amdgcn_readlane((int)gw, t);
                    rs.in_off = (uint32_t)__builtin_amdgcn_readlane((int)rank, t);
                    rs.np_in = (uint32_t)__popcll(__ballot(leader && lo == rs.in_lo));
                }
                if (leader) {
                    a.p_key[first + rank] = (lo < r && !strong) ? (key | kWeakPart) : key;
                    a.p_pairs[first + rank] = cnt;
                    a.p_sum[first + rank] = sum;
                }
            } else {
                // too many reads for one in-register sort: the chunk's partial aggregates go to the host, which merges
                // them; every group counts as a connection (a chunk cannot know the whole group's weight)
                const uint64_t gl_in = __ballot(gleader && lo < r);
                if (gleader && lo < r) {
                    atomicAdd(&a.out_deg[lo], 1u);
                    a.out_hi[lo] = r;
                }
                rs.n_in += (uint32_t)__popcll(gl_in);
                const uint32_t nl = (uint32_t)__popcll(lmask);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&a.counts->n_groups, nl);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (leader) {
                    const uint32_t o = base + rank;
                    if (o < a.g_cap) {
                        GroupRec g;
                        g.key = group_pack(lo, r, (uint32_t)((key >> 4) & 255), (uint32_t)(key & 15));
                        g.pairs = cnt;
                        g.sum_isize = sum;
                        a.g_rec[o] = g;
                    } else {
                        a.counts->overflow = 1;
                    }
                }
            }
        }
        if (lane == 0) a.rs[r] = rs;
    }
}

namespace {

// process_sv (BreakDancer.cpp:348-497) + SvBuilder for region A (and B when B >= 0) over the alive groups gs[0..2] =
// (A,A), (A,B), (B,B).  Writes the slot's staging record; slot_info stays 0 when a gate rejects the candidate.
__device__ void assemble_sv(const K6Arrays& a, uint32_t A, int32_t B, const GrpRange (&gs)[3], int max_readlen, uint32_t slot,
                            uint32_t start) {
    const int mrp = a.min_read_pair;
    int flag_counts[kNumFlags];
#pragma unroll
    for (int f = 0; f < kNumFlags; ++f) flag_counts[f] = 0;
    int num_pairs = 0;
    for (int g = 0; g < 3; ++g)
        for (uint32_t i = 0; i < gs[g].cnt; ++i) {
            const int f = (int)(a.p_key[gs[g].beg + i] & 15);
            const int pr = (int)a.p_pairs[gs[g].beg + i];
            if (f < kNumFlags) flag_counts[f] += pr;
            num_pairs += pr;
        }
    if (num_pairs < mrp) return;
    int flag = BDX_NA;
    {
        int best = 0;
        for (int f = 0; f < kNumFlags; ++f)
            if (flag_counts[f] > flag_counts[best]) best = f;
        if (flag_counts[best] > 0) flag = best;
    }
    if (flag_counts[flag] < mrp) return;

    const RegionRec ra = a.r_rec[A];
    int chr[2], pos[2], fwd[2], rev[2];
    chr[0] = ra.tid; pos[0] = ra.start; pos[1] = ra.end;
    fwd[0] = (int)(ra.n - ra.rev); rev[0] = (int)ra.rev;
    int total_region_size = ra.end - ra.start + 1;
    if (B >= 0) {
        const RegionRec rb = a.r_rec[B];
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

    // per-library pairs / spans of the dominant flag in ascending library order: a three-way merge, the parts of a
    // group being sorted by (flag, library)
    uint32_t idx[3] = {0, 0, 0};
    auto skip = [&](int g) {
        while (idx[g] < gs[g].cnt && (int)(a.p_key[gs[g].beg + idx[g]] & 15) != flag) ++idx[g];
    };
    skip(0); skip(1); skip(2);
    LibStage* ls = a.lib_stage + (size_t)slot * a.acc_stride;
    uint32_t nacc = 0;
    float diff = 0.0f;
    while (true) {
        int best = 256;
        for (int g = 0; g < 3; ++g)
            if (idx[g] < gs[g].cnt) best = min(best, (int)((a.p_key[gs[g].beg + idx[g]] >> 4) & 255));
        if (best == 256) break;
        int rc = 0, span = 0;
        for (int g = 0; g < 3; ++g)
            if (idx[g] < gs[g].cnt && (int)((a.p_key[gs[g].beg + idx[g]] >> 4) & 255) == best) {
                rc += (int)a.p_pairs[gs[g].beg + idx[g]];
                span += (int)a.p_sum[gs[g].beg + idx[g]];
                ++idx[g];
                skip(g);
            }
        diff = __fadd_rn(diff, __fsub_rn((float)span, __fmul_rn((float)rc, a.lib_mean[best])));
        const uint32_t nflag = a.hist[(size_t)best * kNumFlags + flag];
        double lambda = __dmul_rn((double)total_region_size, __ddiv_rn((double)nflag, (double)a.covered_ref_len));
        lambda = (1.0e-10 < lambda) ? lambda : 1.0e-10;
        if (nacc < a.acc_stride) ls[nacc] = LibStage{best, rc, lambda};
        ++nacc;
    }
    if (nacc > a.acc_stride) { a.counts->overflow = 3; return; }  // cannot happen: distinct libraries <= min(nlibs, parts)

    // normal reads between the regions: proper reads after A's last read up to and including B's first
    CnStage* cs = a.cn_stage + (size_t)slot * a.nkeys;
    uint32_t ncn = 0;
    float cn_sum = 0.0f;
    if (B >= 0) {
        const int nk = a.nkeys;
        const float span = (float)(pos[1] - pos[0]);
        for (int k = 0; k < nk; ++k) {
            const uint32_t cnt = a.r_pk[(size_t)B * 2 * nk + k] - a.r_pk[(size_t)A * 2 * nk + nk + k];
            if (cnt == 0) continue;
            const float cn = __fmul_rn(__fdiv_rn((float)cnt, __fmul_rn(a.key_density[k], span)), 2.0f);
            cs[ncn] = CnStage{k, cn};
            ++ncn;
            cn_sum = __fadd_rn(cn_sum, cn);
        }
    }
    // without normal reads between the regions the reference divides 0 by 0: x86's default NaN has the sign bit set and
    // survives "1 - x" unchanged, which is what its formatter prints as -nan
    const float allele_frequency =
        ncn ? __fsub_rn(1.0f, __fdiv_rn(cn_sum, __fmul_rn(2.0f, (float)ncn))) : __uint_as_float(0xFFC00000u);

    if (flag != BDX_ARP_RF && flag != BDX_ARP_RR && pos[0] + max_readlen - 5 < pos[1]) pos[0] += max_readlen - 5;
    const int diffspan = (int)((double)__fdiv_rn(diff, (float)flag_counts[flag]) + 0.5);

    SvOut o;
    for (int i = 0; i < 2; ++i) { o.sv.chr[i] = chr[i]; o.sv.pos[i] = pos[i] + 1; o.sv.fwd[i] = fwd[i]; o.sv.rev[i] = rev[i]; }
    o.sv.flag = flag; o.sv.size = diffspan; o.sv.score = 0; o.sv.num_reads = flag_counts[flag]; o.sv.printed = 0;
    o.sv.region[0] = (int32_t)A; o.sv.region[1] = B;
    o.sv.lib_begin = 0; o.sv.lib_count = (int32_t)nacc;
    o.sv.cn_begin = 0; o.sv.cn_count = (int32_t)ncn;
    o.sv.allele_frequency = allele_frequency; o.sv.logp = 0.0;
    o.grp_mask = (gs[0].cnt ? 1u : 0u) | (gs[1].cnt ? 2u : 0u) | (gs[2].cnt ? 4u : 0u);
    o.start = start;
    a.slot[slot] = o;
    a.slot_info[slot] = 1u | (nacc << 1) | (ncn << 8);
}

}  // namespace

// One thread per region: decide who handles its component, hand foreign components to the host, walk its own.
__global__ __launch_bounds__(256) void k6_sv_kernel(K6Arrays a) {
    const int lane = threadIdx.x & 63;
    const uint32_t NR = a.counts->n_regions;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    const bool active = r < NR;
    const int mrp = a.min_read_pair;
    const uint32_t period = (uint32_t)a.period;
    RegSum s{};
    uint32_t od = 0;
    if (active) { s = a.rs[r]; od = a.out_deg[r]; }
    bool single = false, pairA = false, pairB = false;
    uint32_t B = 0;
    RegSum sB{};
    if (active && !a.force_host && !s.big) {
        if (s.n_in == 0 && od == 0) {
            single = s.np_self > 0 && s.np_self <= (uint32_t)kK6MaxParts;
        } else if (s.n_in == 0 && od == 1) {
            B = a.out_hi[r];
            sB = a.rs[B];
            pairA = !sB.big && sB.n_in == 1 && a.out_deg[B] == 0 && B / period == r / period &&
                    s.np_self + sB.np_in + sB.np_self <= (uint32_t)kK6MaxParts;
        } else if (s.n_in == 1 && od == 0) {
            const uint32_t A = s.in_lo;
            const RegSum sA = a.rs[A];
            pairB = !sA.big && sA.n_in == 0 && a.out_deg[A] == 1 && A / period == r / period &&
                    sA.np_self + s.np_in + s.np_self <= (uint32_t)kK6MaxParts;
        }
    }
    const bool covered = single || pairA || pairB;

    // groups of the components this kernel does not walk -> host list (one reservation per wave)
    {
        const uint32_t need = (active && !covered && !s.big) ? s.np_emit : 0u;
        const uint32_t inc = wave_incl_scan(need);
        const uint32_t tot = (uint32_t)__shfl((int)inc, 63);
        uint32_t base = 0;
        if (lane == 63 && tot) base = atomicAdd(&a.counts->n_groups, tot);
        base = (uint32_t)__shfl((int)base, 63);
        if (need) {
            const uint32_t first = a.r_rec[r].first;
            uint32_t o = base + inc - need;
            for (uint32_t i = 0; i < s.np_all; ++i) {
                const uint64_t key = a.p_key[first + i];
                if (key & kWeakPart) continue;
                if (o < a.g_cap) {
                    GroupRec g;
                    g.key = ((key >> 12) << 38) | ((uint64_t)r << 12) | (key & 0xFFFull);
                    g.pairs = a.p_pairs[first + i];
                    g.sum_isize = a.p_sum[first + i];
                    a.g_rec[o] = g;
                } else {
                    a.counts->overflow = 1;
                }
                ++o;
            }
        }
    }
    // totals: pairs of all regions, groups of the components walked here
    {
        const uint32_t pairs = active ? s.n_pairs : 0u;
        uint32_t grp = active ? s.n_weak : 0u;  // inert connections are counted but never listed
        if (single) grp += 1;
        else if (pairA) grp += 1u + (s.np_self ? 1u : 0u) + (sB.np_self ? 1u : 0u);
        const uint32_t tp = wave_sum_u32(pairs), tg = wave_sum_u32(grp);
        if (lane == 0) {
            if (tp) atomicAdd(&a.counts->n_pairs, tp);
            if (tg) atomicAdd(&a.counts->n_groups_dev, tg);
        }
    }
    if (r == 0) a.counts->n_slots = 3 * NR;
    if (!active || pairB) return;  // the slots of a pair's second region are written by the thread of the first

    // _max_readlen at this window's flush: the value of the candidate that closes there (BreakDancer.cpp:254-259)
    const uint32_t rl = (r / period + 1) * period - 1;
    const int max_readlen = rl < NR ? a.r_rec[rl].maxq : a.counts->last_maxq;
    a.slot_info[3 * r] = 0; a.slot_info[3 * r + 1] = 0; a.slot_info[3 * r + 2] = 0;
    const GrpRange none{0, 0};
    if (single) {
        const RegionRec RA = a.r_rec[r];
        if ((int)s.w_self >= mrp) {
            const GrpRange gs[3] = {region_stored(RA, a) ? GrpRange{RA.first + s.np_all - s.np_self, s.np_self} : none, none, none};
            assemble_sv(a, r, -1, gs, max_readlen, 3 * r, r);
        }
    } else if (pairA) {
        a.slot_info[3 * B] = 0; a.slot_info[3 * B + 1] = 0; a.slot_info[3 * B + 2] = 0;
        const RegionRec RA = a.r_rec[r], RB = a.r_rec[B];
        const bool stA = region_stored(RA, a), stB = region_stored(RB, a);
        const GrpRange AA{RA.first + s.np_all - s.np_self, s.np_self}, AB{RB.first + sB.in_off, sB.np_in},
            BB{RB.first + sB.np_all - sB.np_self, sB.np_self};
        bool aliveAA = AA.cnt > 0, aliveBB = BB.cnt > 0;
        if (AA.cnt > 0 && (int)s.w_self >= mrp) {  // visit(A): its self edge first
            const GrpRange gs[3] = {stA ? AA : none, none, none};
            assemble_sv(a, r, -1, gs, max_readlen, 3 * r, r);
            if (stA) aliveAA = false;
        }
        const bool passed = (int)sB.w_in >= mrp;  // then the edge A-B: B joins the frontier whatever process_sv decides
        if (passed) {
            const GrpRange gs[3] = {(aliveAA && stA) ? AA : none, (stA && stB) ? AB : none, (aliveBB && stB) ? BB : none};
            assemble_sv(a, r, (int32_t)B, gs, max_readlen, 3 * r + 1, r);
            if (stA) aliveAA = false;
            if (stB) aliveBB = false;
        }
        if (BB.cnt > 0 && (int)sB.w_self >= mrp) {  // visit(B): from A's frontier, or later as its own start vertex
            const GrpRange gs[3] = {(aliveBB && stB) ? BB : none, none, none};
            assemble_sv(a, B, -1, gs, max_readlen, passed ? 3 * r + 2 : 3 * B, passed ? r : B);
        }
    }
}

struct SlotIn {
    const uint32_t* info;
    __device__ U4 operator()(uint32_t i, uint32_t) const {
        const uint32_t v = info[i];
        return U4{v & 1u, (v >> 1) & 127u, (v >> 8) & 127u, 0u};
    }
};

struct SlotOut {
    K6Arrays a;
    __device__ void operator()(uint32_t i, uint32_t n, const U4& inc, const U4& e) const {
        if (i == n - 1) { a.counts->n_sv_dev = inc.x; a.counts->n_terms_dev = inc.y; a.counts->n_cn_dev = inc.z; }
        if (!e.x) return;
        const uint32_t d = inc.x - 1, lb = inc.y - e.y, cb = inc.z - e.z;
        if (d >= a.sv_cap || lb + e.y > a.term_cap || cb + e.z > a.cn_cap) { a.counts->overflow = 1; return; }
        SvOut o = a.slot[i];
        o.sv.lib_begin = (int32_t)lb;
        o.sv.cn_begin = (int32_t)cb;
        a.sv_out[d] = o;
        const LibStage* ls = a.lib_stage + (size_t)i * a.acc_stride;
        for (uint32_t q = 0; q < e.y; ++q) {
            const LibStage l = ls[q];
            a.lib_index[lb + q] = l.lib;
            a.lib_pairs[lb + q] = l.rc;
            a.t_lambda[lb + q] = l.lambda;
            a.t_k[lb + q] = l.rc;
        }
        const CnStage* cs = a.cn_stage + (size_t)i * a.nkeys;
        for (uint32_t q = 0; q < e.z; ++q) {
            const CnStage cn = cs[q];
            a.cn_key[cb + q] = cn.key;
            a.cn_value[cb + q] = cn.value;
        }
    }
};

void launch_k6_groups(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s) {
    if (n_anom_host == 0) return;
    const uint32_t gp = std::min<uint32_t>((n_anom_host + 3) / 4, 1024u);
    hipLaunchKernelGGL(k6_pairs_kernel, dim3(gp), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k6_sv_kernel, dim3((n_anom_host + 255) / 256), dim3(256), 0, s, a);
}

void launch_k6_compact(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s) {
    if (n_anom_host == 0) return;
    SlotIn in{a.slot_info};
    SlotOut out{a};
    scan_launch<U4>(in, out, &a.counts->n_slots, 3 * n_anom_host, a.ws_u4, a.total_u4, s);
}

}  // namespace bdx
