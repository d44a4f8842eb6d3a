// K1 -- per-read classification, pass-1 statistics and per-tile totals, one streaming pass.
//
// Replaces (reference file:line under src/lib):
//   io/IlluminaPEReadClassifier.cpp:13-101   pe_classify / classify
//   io/BamSummary.cpp:68-114                 pass-1 body (ref_len, proper counts, flag histogram)
//   breakdancer/BreakDancer.cpp:155-207      filter chain, -l remaps, RR->FF fold, normal-read tests
//
// HBM-bound: 25 B/read in (5 x i32, u16 flag, u8 mapq/lib/bam; 24 / 23 B with one library and / or one source file, whose
// index columns are not read), 1 B/read out (class byte).
// Layout: a tile is 256 consecutive reads = one wave64, lane l owns reads [4l, 4l+4), so every column is
// fetched with one 16/8/4-byte load per lane (1 KiB per wave-instruction for the i32 columns).  Waves are
// independent: there is NO workgroup barrier inside the tile loop and NO global atomic at all.  Per-tile
// totals come from ballots + popcounts, the pass-1 counters are privatised in LDS and written once per
// workgroup, and the per-file reference-length monoid of each tile goes to a plain per-tile table that a tiny
// follow-up kernel folds in order.
//
// Three tile bodies, chosen per tile by two wave-wide tests: the usual tile of one library and one file (library record in scalar
// registers), the tile of SEVERAL LIBRARIES IN ONE FILE (a genome whose read groups are different libraries: the record per read from LDS,
// the same masks; 0.626 -> 0.527 ms for 116 M records of four libraries, 6.2 TB/s by 28 B/read), and the general body for everything
// else (ragged end, several files in a tile, more than kMixedLibs libraries).
//
// Where the time goes (configs[1], 15 M reads, 400 MB): a kernel that only reads the nine columns in this shape takes 57 us, with
// the 1 B/read class-byte store 64-66 us (tools/stream_probe.hip: next to the read stream a byte written costs about four read);
// this kernel takes 68 us without and 73-75 us with the ready-made records for K2.  The instruction stream is not what bounds it:
// the usual tile's path below has about half the instructions of the general body it was split from, for the same time.
#include <cstddef>

#include <hip/hip_ext.h>

#include "bdx_dev.h"
#include "bdx_finalize.h"

#include <limits.h>

#ifdef BDX_KPROF
#include "bdx_scan.h"  // in-kernel clocks of a measurement build
#else
#define KPROF(row, col) do {} while (0)
#endif

namespace bdx {

constexpr int kMixedLibs = 8;   // the several-libraries tile body counts proper reads per library with one ballot per library and slot

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned short v4us __attribute__((ext_vector_type(4)));
typedef unsigned char v4uc __attribute__((ext_vector_type(4)));
// streaming (non-temporal) loads: every input byte is used exactly once
__device__ __forceinline__ int4 ldnt(const int32_t* p) { const v4i v = __builtin_nontemporal_load((const v4i*)p); return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ ushort4 ldnt(const uint16_t* p) { const v4us v = __builtin_nontemporal_load((const v4us*)p); return make_ushort4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uchar4 ldnt(const uint8_t* p) { const v4uc v = __builtin_nontemporal_load((const v4uc*)p); return make_uchar4(v.x, v.y, v.z, v.w); }

// (selects, innermost case first: as nested ifs this compiled to five levels of exec-mask branches per read)
__device__ __forceinline__ int classify_read(unsigned sam, int tid, int mtid, int pos, int mpos, int ai, float upper,
                                             float lower) {
    const bool rr = sam & 0x10u, mr = sam & 0x20u;
    const float fi = (float)ai;  // the reference compares int against the float cutoffs
    int f = fi < lower ? F_SMALL : F_NORMAL_FR;
    f = fi > upper ? F_LARGE : f;
    f = ((pos < mpos) == rr) ? F_RF : f;
    f = (rr == mr) ? (rr ? F_RR : F_FF) : f;
    f = tid != mtid ? F_CTX : f;
    f = (sam & 0x8u) ? F_MATE_UNMAPPED : f;
    f = (sam & 0x4u) ? F_UNMAPPED : f;
    f = ((sam & 0x400u) || !(sam & 0x1u)) ? F_NA : f;
    return f;
}

__device__ __forceinline__ int remap_long_insert(int f, int ai, float upper, float lower) {
    const float fi = (float)ai;
    if (fi > upper && f == F_NORMAL_RF) f = F_RF;
    if (fi < upper && f == F_RF) f = F_NORMAL_RF;
    if (fi < lower && f == F_NORMAL_RF) f = F_SMALL;
    return f;
}


__device__ __forceinline__ unsigned lane0_of(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// The usual tile: full, every record of one library and one source file.  Same results as the general body in the kernel
// below, written for the instruction stream: the library record and both indices are wave-uniform, every condition stays a
// lane mask (no bool crosses a branch), the short-circuit operators of the filter chain are plain ANDs, and the class bytes
// leave as one word.  Returns the tile's totals through the masks.
struct TileMasks {
    uint64_t ba[4], bn[4], bp[4];  // anomalous / normal-leftmost / proper-and-passing, per slot
    unsigned c1;                   // proper reads above the mapping-quality cutoff (BamSummary.cpp:100-108)
};

template <bool kLongInsert>
__device__ __forceinline__ unsigned classify_uniform_tile(const K1Params& p, const DevLib dl, uint32_t* s_hist /* of library L0 */,
                                                          const int (&tid)[4], const int (&pos)[4], const int (&mtid)[4],
                                                          const int (&mpos)[4], const int (&isz)[4], const unsigned (&sam)[4],
                                                          unsigned mqp, TileMasks& tm) {
    unsigned word = 0;
    tm.c1 = 0;
    const bool opt_t = p.opt_t != 0;
    const uint64_t m_opt_t = opt_t ? ~0ull : 0ull;
    int hist_f[4];
    bool hist_on[4];
    // (one straight block for the four slots: the comparisons, their ballots and the selects interleave freely; a ballot that
    // sits in another block than its comparison goes through a 0/1 value per lane and a second compare)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned s = sam[r];
        const int ai = abs(isz[r]);
        const float fi = (float)ai;  // the reference compares int against the float cutoffs
        const bool rr = (s & 0x10u) != 0;
        const bool same_strand = ((s ^ (s >> 1)) & 0x10u) == 0;
        const bool left = pos[r] < mpos[r];
        const bool same_tid = tid[r] == mtid[r];
        int f = fi < dl.lower ? F_SMALL : F_NORMAL_FR;
        f = fi > dl.upper ? F_LARGE : f;
        f = (left == rr) ? F_RF : f;
        const int ss = rr ? F_RR : F_FF;
        f = same_strand ? ss : f;
        f = same_tid ? f : F_CTX;
        f = (s & 0x8u) ? F_MATE_UNMAPPED : f;
        f = (s & 0x4u) ? F_UNMAPPED : f;
        f = ((s & 0x401u) != 0x1u) ? F_NA : f;
        const bool mq_ok = (int)((mqp >> (8 * r)) & 0xffu) > dl.min_mapq;
        const bool proper = (s & 0x40Fu) == 0x3u;
        const bool plain = (s & 0x40Du) == 0x1u;  // f != F_NA and neither end unmapped: one test of the flag word
        const bool h_ok = mq_ok & plain & !(opt_t & same_tid);
        const int f1 = kLongInsert ? remap_long_insert(f, ai, dl.upper, dl.lower) : f;
        const bool normal1 = (f1 & 14) == F_NORMAL_FR;  // F_NORMAL_FR (6) or F_NORMAL_RF (7)
        hist_f[r] = f1; hist_on[r] = h_ok & !normal1;
        const bool is_ctx = f == F_CTX, near = ai <= p.max_sd;
        const bool pass = h_ok & (is_ctx | near);
        const int f2 = (f1 == F_RR) ? F_FF : f1;
        const bool nl = pass & normal1 & left;
        // the wave's masks, combined as 64-bit scalars from the masks of the single comparisons
        const uint64_t m_mq = ballot64(mq_ok), m_prop = ballot64(proper), m_norm = ballot64(normal1);
        const uint64_t m_hok = m_mq & ballot64(plain) & ~(m_opt_t & ballot64(same_tid));
        const uint64_t m_pass = m_hok & (ballot64(is_ctx) | ballot64(near));
        tm.ba[r] = m_pass & ~m_norm; tm.bn[r] = m_pass & m_norm & ballot64(left); tm.bp[r] = m_pass & m_prop;
        tm.c1 += popc64(m_mq & m_prop);
        const unsigned hi = 0x10u | (proper ? 0x20u : 0u) | (nl ? 0x40u : 0u);
        const unsigned byte = pass ? ((unsigned)f2 | hi) : (unsigned)f;
        word |= byte << (8 * r);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (hist_on[r]) atomicAdd(&s_hist[hist_f[r]], 1u);
    return word;
}

// The usual tile of a genome of several libraries in one file (configs[2]-[3]: every tile holds reads of all four read groups): full,
// one source file, a library per READ.  The body of classify_uniform_tile with the library record fetched per slot from LDS (one
// 16-byte read each) instead of held in scalar registers; the masks and the class bytes are the same.  Leaves, beside the totals'
// masks, the proper-and-well-mapped mask per slot (the per-library counts of BamSummary.cpp:100-108 are taken from it by the caller)
// and whether every read's library counts for the counter key k0.
struct MixedMasks {
    uint64_t bq[4];   // proper reads above their library's mapping-quality cutoff, per slot
    bool one_key;     // (wave-uniform) every library in the tile has the counter key k0
};

template <bool kLongInsert>
__device__ __forceinline__ unsigned classify_mixed_tile(const K1Params& p, const DevLib* s_lib, uint32_t* s_cnt, const unsigned (&L)[4], int k0,
                                                        const int (&tid)[4], const int (&pos)[4], const int (&mtid)[4], const int (&mpos)[4],
                                                        const int (&isz)[4], const unsigned (&sam)[4], unsigned mqp, TileMasks& tm, MixedMasks& mx) {
    unsigned word = 0;
    tm.c1 = 0;
    const bool opt_t = p.opt_t != 0;
    const uint64_t m_opt_t = opt_t ? ~0ull : 0ull;
    int hist_i[4];
    bool hist_on[4];
    bool same_key = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const DevLib dl = s_lib[L[r]];
        same_key = same_key && dl.key == k0;
        const unsigned s = sam[r];
        const int ai = abs(isz[r]);
        const float fi = (float)ai;  // the reference compares int against the float cutoffs
        const bool rr = (s & 0x10u) != 0;
        const bool same_strand = ((s ^ (s >> 1)) & 0x10u) == 0;
        const bool left = pos[r] < mpos[r];
        const bool same_tid = tid[r] == mtid[r];
        int f = fi < dl.lower ? F_SMALL : F_NORMAL_FR;
        f = fi > dl.upper ? F_LARGE : f;
        f = (left == rr) ? F_RF : f;
        const int ss = rr ? F_RR : F_FF;
        f = same_strand ? ss : f;
        f = same_tid ? f : F_CTX;
        f = (s & 0x8u) ? F_MATE_UNMAPPED : f;
        f = (s & 0x4u) ? F_UNMAPPED : f;
        f = ((s & 0x401u) != 0x1u) ? F_NA : f;
        const bool mq_ok = (int)((mqp >> (8 * r)) & 0xffu) > dl.min_mapq;
        const bool proper = (s & 0x40Fu) == 0x3u;
        const bool plain = (s & 0x40Du) == 0x1u;
        const bool h_ok = mq_ok & plain & !(opt_t & same_tid);
        const int f1 = kLongInsert ? remap_long_insert(f, ai, dl.upper, dl.lower) : f;
        const bool normal1 = (f1 & 14) == F_NORMAL_FR;
        hist_i[r] = (int)L[r] * kNumFlags + f1; hist_on[r] = h_ok & !normal1;
        const bool is_ctx = f == F_CTX, near = ai <= p.max_sd;
        const bool pass = h_ok & (is_ctx | near);
        const int f2 = (f1 == F_RR) ? F_FF : f1;
        const bool nl = pass & normal1 & left;
        const uint64_t m_mq = ballot64(mq_ok), m_prop = ballot64(proper), m_norm = ballot64(normal1);
        const uint64_t m_hok = m_mq & ballot64(plain) & ~(m_opt_t & ballot64(same_tid));
        const uint64_t m_pass = m_hok & (ballot64(is_ctx) | ballot64(near));
        tm.ba[r] = m_pass & ~m_norm; tm.bn[r] = m_pass & m_norm & ballot64(left); tm.bp[r] = m_pass & m_prop;
        mx.bq[r] = m_mq & m_prop;
        tm.c1 += popc64(mx.bq[r]);
        const unsigned hi = 0x10u | (proper ? 0x20u : 0u) | (nl ? 0x40u : 0u);
        const unsigned byte = pass ? ((unsigned)f2 | hi) : (unsigned)f;
        word |= byte << (8 * r);
    }
    mx.one_key = __all(same_key);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (hist_on[r]) atomicAdd(&s_cnt[hist_i[r]], 1u);
    return word;
}

size_t k1_lds_bytes(int nlibs, int nbams, int nkeys) {
    size_t b = (size_t)nlibs * sizeof(DevLib);
    b += (size_t)(nlibs * kNumFlags + nlibs + nbams) * 4;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)kWaves * nbams * sizeof(MonoRec);  // lane-0 state of the exact (tid-boundary) path
    (void)nkeys;
    return (b + 15) & ~(size_t)15;
}

// kLongInsert: the -l remaps of BreakDancer.cpp:175-186 (p.opt_l), decided at the launch
template <bool kLongInsert>
__global__ __launch_bounds__(kBlock, 6) void k1_classify_kernel(const K1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nlibs = p.nlibs, nbams = p.nbams, nkeys = p.nkeys;
    const int ncnt = nlibs * kNumFlags + nlibs + nbams;
    const int ncols = 2 + nkeys;
    DevLib* s_lib = (DevLib*)smem;
    uint32_t* s_cnt = (uint32_t*)(s_lib + nlibs);
    size_t off = (size_t)nlibs * sizeof(DevLib) + (size_t)ncnt * 4;
    off = (off + 15) & ~(size_t)15;
    MonoRec* s_mono = (MonoRec*)(smem + off);

    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    __shared__ v4u_t s_stash[kWaves][2 * kStashCap];  // a wave's ready-made records of one tile, before they leave
    static_assert(kStashCap + 64 <= 2 * kStashCap * 4, "the general tile body keeps 16 rank slots and the lanes' 64 class words in a wave's slice");
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < nlibs; i += kBlock) s_lib[i] = p.libs[i];
    for (int i = t; i < ncnt; i += kBlock) s_cnt[i] = 0;
    uint32_t* s_libcnt = s_cnt + nlibs * kNumFlags;
    uint32_t* s_bamcnt = s_libcnt + nlibs;
    MonoRec* my_mono = s_mono + (size_t)w * nbams;
    __syncthreads();

    const uint32_t nwaves = gridDim.x * kWaves;
    for (uint32_t tile = p.tile0 + blockIdx.x * kWaves + w; tile < p.ntiles; tile += nwaves) {
        const uint64_t base = (uint64_t)tile * kTile + (uint64_t)lane * 4;
        int nvalid = 0;
        int tid[4], pos[4], mtid[4], mpos[4], isz[4];
        unsigned sam[4];
        unsigned mqp, libp, bamp;  // the four slots' bytes, slot r in byte r
        if (base + 4 <= p.n) {
            nvalid = 4;
            const int4 a = ldnt(p.r.tid + base);
            const int4 b = ldnt(p.r.pos + base);
            const int4 c = ldnt(p.r.mtid + base);
            const int4 d = ldnt(p.r.mpos + base);
            const int4 e = ldnt(p.r.isize + base);
            const ushort4 f = ldnt(p.r.flag + base);
            mqp = __builtin_nontemporal_load((const uint32_t*)(p.r.mapq + base));
            // with one library / one source file every index counts as 0 (as an index out of range does): the column is not read
            libp = nlibs > 1 ? __builtin_nontemporal_load((const uint32_t*)(p.r.lib + base)) : 0u;
            bamp = nbams > 1 ? __builtin_nontemporal_load((const uint32_t*)(p.r.bam + base)) : 0u;
            tid[0] = a.x; tid[1] = a.y; tid[2] = a.z; tid[3] = a.w;
            pos[0] = b.x; pos[1] = b.y; pos[2] = b.z; pos[3] = b.w;
            mtid[0] = c.x; mtid[1] = c.y; mtid[2] = c.z; mtid[3] = c.w;
            mpos[0] = d.x; mpos[1] = d.y; mpos[2] = d.z; mpos[3] = d.w;
            isz[0] = e.x; isz[1] = e.y; isz[2] = e.z; isz[3] = e.w;
            sam[0] = f.x; sam[1] = f.y; sam[2] = f.z; sam[3] = f.w;
        } else {
            mqp = libp = bamp = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint64_t i = base + r;
                const bool v = i < p.n;
                nvalid += v;
                tid[r] = v ? p.r.tid[i] : 0; pos[r] = v ? p.r.pos[i] : 0; mtid[r] = v ? p.r.mtid[i] : 0;
                mpos[r] = v ? p.r.mpos[i] : 0; isz[r] = v ? p.r.isize[i] : 0; sam[r] = v ? p.r.flag[i] : 0;
                mqp |= (v ? (unsigned)p.r.mapq[i] : 0u) << (8 * r);
                libp |= (v && nlibs > 1 ? (unsigned)p.r.lib[i] : 0u) << (8 * r);
                bamp |= (v && nbams > 1 ? (unsigned)p.r.bam[i] : 0u) << (8 * r);
            }
        }

        bool one_file = false;   // (wave-uniform) every record of the tile comes from one source file ...
        unsigned file0 = 0;      // ... this one
        const unsigned l0 = lane0_of(libp) & 0xffu, b0 = lane0_of(bamp) & 0xffu;
        if (__all(nvalid == 4 && libp == l0 * 0x01010101u && bamp == b0 * 0x01010101u)) {
            // ---- the usual tile -----------------------------------------------------------------------------------------
            const unsigned L0 = l0 < (unsigned)nlibs ? l0 : 0u, B0 = b0 < (unsigned)nbams ? b0 : 0u;
            const DevLib dl = s_lib[L0];
            TileMasks tm;
            const unsigned word = classify_uniform_tile<kLongInsert>(p, dl, s_cnt + L0 * kNumFlags, tid, pos, mtid, mpos, isz, sam, mqp, tm);
            unsigned na = 0, nn = 0, ck = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { na += popc64(tm.ba[r]); nn += popc64(tm.bn[r]); ck += popc64(tm.bp[r]); }
            *(uint32_t*)(p.cls + base) = word;
            const int k0 = dl.key;
            const unsigned colval = lane == kColAnom ? na : (lane == kColNormal ? nn : (lane == kColKey0 + k0 ? ck : 0u));
            if (lane < ncols) p.tile_tot[(uint32_t)lane * p.tstride + tile] = colval;  // (32 bits hold it: < 62 columns of < 2^24 tiles)
            if (lane == 0 && tm.c1) { atomicAdd(&s_libcnt[L0], tm.c1); atomicAdd(&s_bamcnt[B0], tm.c1); }
            one_file = true; file0 = B0;
            if (p.stash && na) {  // (wave-uniform; about a fifth of the tiles of a 1 % discordant genome)
                // Ready-made records for K2 (StashRec): a read's rank and prefix counts are the mask bits of the lower lanes plus
                // the lane's own earlier slots.  Only tiles with anomalous reads pay (half of the tiles of a 1 % discordant
                // chromosome: its discordant pairs come in clusters).  The records cost this kernel 6 us -- not for their ~250
                // instructions (halving the instructions of the whole kernel did not change its time) but for their 6 MB of
                // stores: next to a streaming read a byte written costs about four read (tools/stream_probe.hip).
                unsigned ra = 0, rn = 0, rk = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ra = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm.ba[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm.ba[r], ra));
                    rn = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm.bn[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm.bn[r], rn));
                    rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm.bp[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm.bp[r], rk));
                }
                // The records are put together in the wave's LDS slice and leave as ONE store of consecutive 16-byte pieces, in
                // order (written from the lanes that own the reads -- two stores per record, a few lanes per instruction -- they
                // took the same time).
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                v4u* mine = s_stash[w];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned cb = (word >> (8 * r)) & 0xffu;
                    const bool pass = (cb & 0x10u) != 0;
                    const bool an = pass && (cb & 14u) != (unsigned)F_NORMAL_FR;
                    rk += (cb & 0x30u) == 0x30u ? 1u : 0u;
                    if (an && ra < (unsigned)kStashCap) {
                        const v4u x0 = {(uint32_t)tid[r], (uint32_t)pos[r], (uint32_t)abs(isz[r]), (cb & 15u) | (((sam[r] >> 4) & 1u) << 4) | (L0 << 8)};
                        const v4u x1 = {(uint32_t)(lane * 4 + r) | (rn << 8) | ((uint32_t)k0 << 20), rk, 0u, 0u};
                        mine[2 * ra] = x0;
                        mine[2 * ra + 1] = x1;
                    }
                    ra += an ? 1u : 0u;
                    rn += (cb & 0x40u) ? 1u : 0u;
                }
                __builtin_amdgcn_wave_barrier();  // (LDS serves a wave's accesses in order; this keeps the compiler from reordering them)
                const unsigned pieces = 2u * (na < (unsigned)kStashCap ? na : (unsigned)kStashCap);
                if ((unsigned)lane < pieces) *((v4u*)(p.stash + (size_t)tile * kStashCap) + lane) = mine[lane];  // (plain store: K2 finds some of it in L2)
                __builtin_amdgcn_wave_barrier();
            }
        } else if (nlibs <= kMixedLibs && __all(nvalid == 4 && bamp == b0 * 0x01010101u)) {
            // ---- full, one source file, several libraries: the tiles of a genome whose read groups are different libraries ----------
            const unsigned B0 = b0 < (unsigned)nbams ? b0 : 0u;
            unsigned L[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned x = (libp >> (8 * r)) & 0xffu; L[r] = x < (unsigned)nlibs ? x : 0u; }
            const int k0 = __builtin_amdgcn_readfirstlane(s_lib[L[0]].key);
            TileMasks tm;
            MixedMasks mx;
            const unsigned word = classify_mixed_tile<kLongInsert>(p, s_lib, s_cnt, L, k0, tid, pos, mtid, mpos, isz, sam, mqp, tm, mx);
            unsigned na = 0, nn = 0, ck = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { na += popc64(tm.ba[r]); nn += popc64(tm.bn[r]); ck += popc64(tm.bp[r]); }
            *(uint32_t*)(p.cls + base) = word;
            unsigned colval = lane == kColAnom ? na : (lane == kColNormal ? nn : 0u);
            if (mx.one_key) {
                if (lane == kColKey0 + k0) colval = ck;
            } else {
                for (int k = 0; k < nkeys; ++k) {   // (-a: a counter key per library)
                    unsigned c = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c += popc64(tm.bp[r] & ballot64(s_lib[L[r]].key == k));
                    if (lane == kColKey0 + k) colval = c;
                }
            }
            if (lane < ncols) p.tile_tot[(uint32_t)lane * p.tstride + tile] = colval;
            if (tm.c1) {   // (wave-uniform) proper reads per library: one ballot per library and slot
                for (int v = 0; v < nlibs; ++v) {
                    unsigned c = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c += popc64(mx.bq[r] & ballot64(L[r] == (unsigned)v));
                    if (lane == 0 && c) atomicAdd(&s_libcnt[v], c);
                }
                if (lane == 0) atomicAdd(&s_bamcnt[B0], tm.c1);
            }
            one_file = true; file0 = B0;
            if (p.stash && na) {
                if (mx.one_key) {   // ready-made records for K2, as in the one-library tile; a record carries its own library
                    unsigned ra = 0, rn = 0, rk = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ra = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm.ba[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm.ba[r], ra));
                        rn = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm.bn[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm.bn[r], rn));
                        rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm.bp[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm.bp[r], rk));
                    }
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    v4u* mine = s_stash[w];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned cb = (word >> (8 * r)) & 0xffu;
                        const bool pass = (cb & 0x10u) != 0;
                        const bool an = pass && (cb & 14u) != (unsigned)F_NORMAL_FR;
                        rk += (cb & 0x30u) == 0x30u ? 1u : 0u;
                        if (an && ra < (unsigned)kStashCap) {
                            const v4u x0 = {(uint32_t)tid[r], (uint32_t)pos[r], (uint32_t)abs(isz[r]), (cb & 15u) | (((sam[r] >> 4) & 1u) << 4) | (L[r] << 8)};
                            const v4u x1 = {(uint32_t)(lane * 4 + r) | (rn << 8) | ((uint32_t)k0 << 20), rk, 0u, 0u};
                            mine[2 * ra] = x0;
                            mine[2 * ra + 1] = x1;
                        }
                        ra += an ? 1u : 0u;
                        rn += (cb & 0x40u) ? 1u : 0u;
                    }
                    __builtin_amdgcn_wave_barrier();
                    const unsigned pieces = 2u * (na < (unsigned)kStashCap ? na : (unsigned)kStashCap);
                    if ((unsigned)lane < pieces) *((v4u*)(p.stash + (size_t)tile * kStashCap) + lane) = mine[lane];
                    __builtin_amdgcn_wave_barrier();
                } else {
                    int lane_o = lane;   // (opaque: the slot's address is worked out here, not kept in registers across the tile loop)
                    asm volatile("" : "+v"(lane_o));
                    if ((unsigned)lane_o < (na < (unsigned)kStashCap ? na : (unsigned)kStashCap))
                        p.stash[(size_t)tile * kStashCap + lane_o].where = 0xFFFFFFFFu;  // (several counter keys in the tile: K2 compacts it from the columns)
                }
            }
        } else {
        // ---- any other tile: ragged end of the input, several libraries or files in it ----------------------------------
        // (opaque copies of the loop invariants this rarely taken body works with: what the compiler derives from them is then
        // computed here, when needed, instead of before the tile loop, where it took registers from every tile and spilled)
        int lane_r = lane, nlibs_r = nlibs, nbams_r = nbams, nkeys_r = nkeys;
        asm volatile("" : "+v"(lane_r), "+s"(nlibs_r), "+s"(nbams_r), "+s"(nkeys_r));
        const int ncols_r = 2 + nkeys_r;
        unsigned mq[4], lib[4], bam[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { mq[r] = (mqp >> (8 * r)) & 0xffu; lib[r] = (libp >> (8 * r)) & 0xffu; bam[r] = (bamp >> (8 * r)) & 0xffu; }
        unsigned cls4[4];
        bool p1c[4], pk[4], anom[4], nleft[4];
        int key[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = r < nvalid;
            const unsigned L = lib[r] < (unsigned)nlibs_r ? lib[r] : 0u;
            lib[r] = L;
            if (bam[r] >= (unsigned)nbams_r) bam[r] = 0;
            const DevLib dl = s_lib[L];
            const int ai = abs(isz[r]);
            const int f = classify_read(sam[r], tid[r], mtid[r], pos[r], mpos[r], ai, dl.upper, dl.lower);
            const bool mq_ok = (int)mq[r] > dl.min_mapq;
            const bool proper = (sam[r] & 0x40Fu) == 0x3u;
            p1c[r] = valid && mq_ok && proper;
            const bool h_ok = valid && mq_ok && f != F_NA && !(sam[r] & 0xCu) && !(p.opt_t && tid[r] == mtid[r]);
            const int f1 = kLongInsert ? remap_long_insert(f, ai, dl.upper, dl.lower) : f;
            if (h_ok && f1 != F_NORMAL_FR && f1 != F_NORMAL_RF) atomicAdd(&s_cnt[L * kNumFlags + f1], 1u);
            const bool pass = h_ok && !(f != F_CTX && ai > p.max_sd);
            const int f2 = (f1 == F_RR) ? F_FF : f1;
            const bool normal = (f2 == F_NORMAL_FR) || (f2 == F_NORMAL_RF);
            anom[r] = pass && !normal;
            nleft[r] = pass && normal && pos[r] < mpos[r];
            pk[r] = pass && proper;
            key[r] = dl.key;
            cls4[r] = pass ? ((unsigned)f2 | 0x10u | (proper ? 0x20u : 0u) | (nleft[r] ? 0x40u : 0u)) : (unsigned)f;
        }
        // (the lane's four class bytes also go to the wave's LDS slice, words [16, 80): the record stash below takes a read's byte from there)
        const unsigned clsp = cls4[0] | (cls4[1] << 8) | (cls4[2] << 16) | (cls4[3] << 24);
        if (p.stash) ((uint32_t*)s_stash[w])[kStashCap + lane_r] = clsp;
        if (nvalid == 4) {
            *(unsigned*)(p.cls + base) = clsp;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nvalid) p.cls[base + r] = (uint8_t)cls4[r];
        }

        // ---- per-tile totals (lane c keeps column c) and workgroup counters: ballots + popcounts --------------
        {
            unsigned na = 0, nn = 0;
            uint64_t ba[4], bn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { ba[r] = ballot64(anom[r]); bn[r] = ballot64(nleft[r]); na += popc64(ba[r]); nn += popc64(bn[r]); }
            unsigned colval = lane_r == kColAnom ? na : (lane_r == kColNormal ? nn : 0u);
            // one library and one source file (a ragged last tile usually): one count per key
            const unsigned L0 = __shfl(lib[0], 0), B0 = __shfl(bam[0], 0);
            const bool uni = __all(lib[0] == L0 && lib[1] == L0 && lib[2] == L0 && lib[3] == L0 && bam[0] == B0 &&
                                   bam[1] == B0 && bam[2] == B0 && bam[3] == B0);
            one_file = uni; file0 = B0;
            // Ready-made records for K2 from this body too when the tile is full and its reads share ONE counter key -- the tiles of a genome
            // of several libraries in one file (configs[2]-[3]: every tile is "mixed", and until round 5 K2 then re-read the class bytes of all
            // reads and gathered nine columns per anomalous read: 613 MB fetched for ~120 MB of payload at a GPU's share of a genome).  A
            // record carries its own library; the prefix counts are those of the uniform body.  Otherwise the slots say that there are none
            // and K2 compacts the tile from the columns.
            const int k_first = __shfl(key[0], 0);
            const bool one_key = __all(nvalid == 4 && key[0] == k_first && key[1] == k_first && key[2] == k_first && key[3] == k_first);
            if (uni) {
                unsigned c1 = 0, ck = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { c1 += popc64(ballot64(p1c[r])); ck += popc64(ballot64(pk[r])); }
                const int k0 = __shfl(key[0], 0);
                if (lane_r == kColKey0 + k0) colval = ck;
                if (lane_r == 0 && c1) { atomicAdd(&s_libcnt[L0], c1); atomicAdd(&s_bamcnt[B0], c1); }
            } else if (nlibs_r <= 8 && nbams_r <= 8) {
                // mixed wave, few libraries / files: one ballot per (value, slot), no divergence
                for (int v = 0; v < nlibs_r; ++v) {
                    unsigned c = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c += popc64(ballot64(p1c[r] && lib[r] == (unsigned)v));
                    if (lane_r == 0 && c) atomicAdd(&s_libcnt[v], c);
                }
                for (int v = 0; v < nbams_r; ++v) {
                    unsigned c = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c += popc64(ballot64(p1c[r] && bam[r] == (unsigned)v));
                    if (lane_r == 0 && c) atomicAdd(&s_bamcnt[v], c);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint64_t todo = ballot64(p1c[r]);
                    while (todo) {  // peel distinct libraries
                        const int ld = __ffsll((long long)todo) - 1;
                        const unsigned v = __shfl(lib[r], ld);
                        const uint64_t m = ballot64(p1c[r] && lib[r] == v);
                        if (lane_r == 0) atomicAdd(&s_libcnt[v], (unsigned)popc64(m));
                        todo &= ~m;
                    }
                    todo = ballot64(p1c[r]);
                    while (todo) {  // peel distinct source files
                        const int ld = __ffsll((long long)todo) - 1;
                        const unsigned v = __shfl(bam[r], ld);
                        const uint64_t m = ballot64(p1c[r] && bam[r] == v);
                        if (lane_r == 0) atomicAdd(&s_bamcnt[v], (unsigned)popc64(m));
                        todo &= ~m;
                    }
                }
            }
            if (!uni) {
                for (int k = 0; k < nkeys_r; ++k) {  // mixed wave: one ballot per key and slot
                    unsigned ck = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ck += popc64(ballot64(pk[r] && key[r] == k));
                    if (lane_r == kColKey0 + k) colval = ck;  // ncols <= 62 (bdx_create limits nkeys to 60)
                }
            }
            if (lane_r < ncols_r) p.tile_tot[(size_t)lane_r * p.tstride + tile] = colval;
            // (here, behind the totals: the classification's per-library temporaries are dead)
            if (p.stash && na && one_key) {
                // (written so that it needs few registers beside the classification's -- built like the uniform body's, from the lanes that own
                // the reads, this body spilled 73 registers and the kernel took three times as long: an owning lane only leaves its reads'
                // in-tile indices AND CLASS BYTES in the wave's LDS slice, by rank; lane j then puts record j together itself -- its other
                // fields gathered from the columns the tile has just been loaded from (input columns, nobody writes them), its prefix counts
                // from the ballots, which are scalar registers.  The class byte does not come back from p.cls: another lane stored it with a
                // plain store a moment ago, and nothing orders that store before this lane's load -- ADVICE r5)
                unsigned ra = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) ra = __builtin_amdgcn_mbcnt_hi((uint32_t)(ba[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ba[r], ra));
                uint32_t* idx = (uint32_t*)s_stash[w];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (anom[r] && ra < (unsigned)kStashCap) idx[ra] = (uint32_t)(lane_r * 4 + r);
                    ra += anom[r] ? 1u : 0u;
                }
                __builtin_amdgcn_wave_barrier();
                const unsigned nrec = na < (unsigned)kStashCap ? na : (unsigned)kStashCap;
                const uint32_t where = idx[(unsigned)lane_r < nrec ? lane_r : 0];
                const unsigned l = where >> 2, rr = where & 3u;
                const uint64_t below = (1ull << l) - 1ull;
                unsigned rn = 0, rk = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // (every ballot is used up at once: scalar registers are as scarce here as vector ones)
                    const uint64_t mn = ballot64(nleft[r]), mp = ballot64(pk[r]);
                    rn += popc64(mn & below) + ((unsigned)r < rr ? (unsigned)((mn >> l) & 1ull) : 0u);
                    rk += popc64(mp & below) + ((unsigned)r <= rr ? (unsigned)((mp >> l) & 1ull) : 0u);   // (up to and including the read itself)
                }
                if ((unsigned)lane_r < nrec) {
                    const uint64_t i = (uint64_t)tile * kTile + where;
                    const unsigned cb = (idx[kStashCap + l] >> (8 * rr)) & 0xffu;   // (the owning lane's word, written at the classification)
                    unsigned L = nlibs_r > 1 ? (unsigned)p.r.lib[i] : 0u;
                    if (L >= (unsigned)nlibs_r) L = 0;
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    const v4u x0 = {(uint32_t)p.r.tid[i], (uint32_t)p.r.pos[i], (uint32_t)abs(p.r.isize[i]), (cb & 15u) | ((((unsigned)p.r.flag[i] >> 4) & 1u) << 4) | (L << 8)};
                    const v4u x1 = {where | (rn << 8) | ((uint32_t)k_first << 20), rk, 0u, 0u};
                    v4u* dst = (v4u*)(p.stash + (size_t)tile * kStashCap + lane_r);
                    dst[0] = x0;
                    dst[1] = x1;
                }
                __builtin_amdgcn_wave_barrier();
            } else if (p.stash && (unsigned)lane_r < (na < (unsigned)kStashCap ? na : (unsigned)kStashCap)) {
                p.stash[(size_t)tile * kStashCap + lane_r].where = 0xFFFFFFFFu;  // (every slot K2 would read)
            }
        }
        }

        // ---- reference-length monoid per source file (BamSummary.cpp:70-74) ---------------------------------------
        {
            const int tw = __builtin_amdgcn_readfirstlane(tid[0]);
            bool same = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) same = same && (r >= nvalid || tid[r] == tw);
            const bool one_tid = __all(same);
            if (one_tid && one_file && __all(nvalid == 4)) {
                // the usual tile: one tid, one file, full -- the differences telescope to (last record) - (first record)
                const int first = __builtin_amdgcn_readfirstlane(pos[0]), last = __builtin_amdgcn_readlane(pos[3], 63);
                if (lane == 0) {
                    MonoRec m;
                    m.ft = tw; m.fp = first; m.lt = tw; m.lp = last; m.sum = (long long)last - (long long)first;
                    p.tile_mono[(size_t)file0 * p.tstride + tile] = m;
                }
            } else {
            int lane_m = lane, nbams_m = nbams;  // (opaque, as above)
            asm volatile("" : "+v"(lane_m), "+s"(nbams_m));
            unsigned bam[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned v = (bamp >> (8 * r)) & 0xffu;
                bam[r] = v < (unsigned)nbams_m ? v : 0u;
            }
            if (one_tid && nbams_m <= 64) {
                // all records of the tile share one tid: consecutive same-file differences telescope to
                // last - first, found with ballots; lane v keeps the record of file v
                int my_first = 0, my_last = 0;
                bool present = false;
                unsigned pending = (1u << nvalid) - 1u;
                while (true) {
                    const uint64_t any = ballot64(pending != 0);
                    if (!any) break;
                    const int ld = __ffsll((long long)any) - 1;
                    const int lowp = pending ? __ffs(pending) - 1 : 0;
                    const unsigned v = __shfl(bam[lowp], ld);
                    unsigned mine = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (((pending >> r) & 1u) && bam[r] == v) mine |= 1u << r;
                    pending &= ~mine;
                    const uint64_t M = ballot64(mine != 0);
                    const int fl = __ffsll((long long)M) - 1, ll = 63 - __clzll((long long)M);
                    const int fs = mine ? __ffs(mine) - 1 : 0, ls = mine ? 31 - __clz(mine) : 0;
                    const int first = __shfl(pos[fs], fl), last = __shfl(pos[ls], ll);
                    if (lane_m == (int)v) { present = true; my_first = first; my_last = last; }
                }
                if (present) {
                    MonoRec m;
                    m.ft = tw; m.fp = my_first; m.lt = tw; m.lp = my_last; m.sum = (long long)my_last - (long long)my_first;
                    p.tile_mono[(size_t)lane_m * p.tstride + tile] = m;
                }
            } else {
                // a tid boundary falls inside the tile (or > 64 files): replay the reference's recurrence exactly,
                // records in order, state held by lane 0 in its wave-private LDS slice
                if (lane_m == 0)
                    for (int b = 0; b < nbams_m; ++b) my_mono[b].ft = -1;
                for (int L = 0; L < 64; ++L) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nv = __shfl(nvalid, L);
                        if (r >= nv) continue;
                        const int rt = __shfl(tid[r], L), rp = __shfl(pos[r], L);
                        const unsigned rb = __shfl(bam[r], L);
                        if (lane_m == 0) {
                            MonoRec m = my_mono[rb];
                            if (m.ft == -1) { m.ft = rt; m.fp = rp; m.sum = 0; }
                            else if (m.lt == rt) m.sum += (long long)rp - (long long)m.lp;
                            m.lt = rt; m.lp = rp;
                            my_mono[rb] = m;
                        }
                    }
                }
                if (lane_m == 0)
                    for (int b = 0; b < nbams_m; ++b)
                        if (my_mono[b].ft != -1) p.tile_mono[(size_t)b * p.tstride + tile] = my_mono[b];
            }
            }
        }
    }
    __syncthreads();
    // flush: kCntCopies-way replicated global counters keep same-address atomics to gridDim/kCntCopies per word
    uint32_t* out = p.blk_cnt + (size_t)(blockIdx.x & (kCntCopies - 1)) * ncnt;
    for (int i = t; i < ncnt; i += kBlock) {
        const uint32_t v = s_cnt[i];
        if (v) atomicAdd(&out[i], v);
    }
}

// start / stop: events that take the kernel's own begin and end timestamps (what rocprofv3 reports for it); an event recorded
// on the stream before and after the launch also clocks the dispatch around it (+4 us)
void launch_k1(const K1Params& p, int grid, size_t lds, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    auto kernel = p.opt_l ? k1_classify_kernel<true> : k1_classify_kernel<false>;
    if (start && stop) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), (uint32_t)lds, s, start, stop, 0u, p);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, s, p);
}

// -------------------------------------------------------------------------------------------------------
// finalize + tile scan.  grid = ncols + 1 workgroups of 1024 threads:
//   workgroup c < ncols : exclusive scan of tile_tot[c][*] -> tile_pre[c][*], column total -> Pass1
//   workgroup ncols     : reduce the per-workgroup counters, fold the per-tile reference-length monoids,
//                         covered_ref_len (BamSummary.cpp:123-126) and the final window
//                         (BreakDancerMax.cpp:109-116)
// -------------------------------------------------------------------------------------------------------
constexpr int kFinBlock = 1024;

// The second level of the finalisation needs every workgroup's results.  When nothing else is enqueued behind this kernel
// that could take it along (K2 does on enqueue-ahead runs), the workgroup that finishes LAST runs it: a dozen or so
// workgroups publish their results with a device-scope fence and take a ticket -- cheaper than one more launch (~7 us).
__device__ __forceinline__ void finalize_tail(const FinalizeParams& p) {
    if (!p.done) return;
    __shared__ uint32_t s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(p.done, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x >= 256) return;  // (finalize2_body is written for four waves; the other waves of this workgroup leave)
    finalize2_body(p);
}

// workgroups [0, ncols): tile scans; workgroups [ncols, ncols + nfold): ordered partial folds of the monoid table
__global__ __launch_bounds__(kFinBlock) void finalize_kernel(const FinalizeParams p) {
    __shared__ uint32_t s_carry;
    __shared__ MonoRec s_mono[kFinBlock / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    KPROF(blockIdx.x * 16 + w, 0);

    const uint32_t nscan = (uint32_t)p.ncols * p.nchunk;
    if (blockIdx.x < nscan) {
        // column c, chunk g: the super tiles [g * chunk_super, (g + 1) * chunk_super) -- one workgroup scanning a whole column
        // of configs[1] (58.6 k totals) was busy for 10 us, longer than anything else in this kernel
        const int c = (int)(blockIdx.x / p.nchunk);
        const uint32_t g = blockIdx.x % p.nchunk;
        const uint32_t tile_lo = g * p.chunk_super * kK2TilesPerWave;
        const uint32_t tile_hi = min(p.ntiles, (g + 1) * p.chunk_super * kK2TilesPerWave);
        const uint32_t* in = p.tile_tot + (size_t)c * p.tstride;
        uint32_t* out = p.tile_pre + (size_t)c * p.tstride;
        if (t == 0) s_carry = 0;
        __syncthreads();
        // K2 takes one wave per kK2TilesPerWave tiles and only needs the prefix at those boundaries.  Thread t fetches the
        // four totals of super tile q * 1024 + t for q = 0..3 of a round (16-byte loads, each one contiguous across the
        // workgroup: with 64 consecutive totals per thread instead, the 16 k scattered line requests of a workgroup took 6-9 us),
        // so a thread's super tiles lie 1024 apart: four wave scans side by side, a 4 x 16 table of wave totals scanned by
        // the first wave, and every thread stores four prefixes, again contiguous across the workgroup.
        constexpr int kQ = 4;   // (a chunk is a whole number of such rounds: 4096 super tiles)
        static_assert(kK2TilesPerWave == 4 && kFinBlock == 1024, "one 16-byte load per super tile");
        __shared__ uint32_t s_wt[kQ * (kFinBlock / 64)];
        __shared__ uint32_t s_round;
        const uint32_t row_hi = min(p.tstride, (g + 1) * p.chunk_super * kK2TilesPerWave);  // (rows are padded to a multiple of 16 and zero-filled)
        for (uint32_t base = tile_lo; base < tile_hi; base += kFinBlock * kQ * 4) {
            uint32_t sum[kQ], inc[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t i = base + ((uint32_t)q * kFinBlock + t) * 4;
                uint4 x = make_uint4(0, 0, 0, 0);
                if (i < row_hi) x = *(const uint4*)(in + i);
                sum[q] = x.x + x.y + x.z + x.w;
            }
            if (sum[0] != 0xFFFFFFFFu) KPROF(blockIdx.x * 16 + w, 1);
#pragma unroll
            for (int q = 0; q < kQ; ++q) inc[q] = wave_incl_scan(sum[q]);
            if (lane == 63) {
#pragma unroll
                for (int q = 0; q < kQ; ++q) s_wt[q * (kFinBlock / 64) + w] = inc[q];
            }
            __syncthreads();
            if (w == 0) {  // exclusive scan of the 64 wave totals in (q, wave) order
                static_assert(kQ * (kFinBlock / 64) == 64, "one wave total per lane");
                const uint32_t v = s_wt[lane];
                const uint32_t li = wave_incl_scan(v);
                s_wt[lane] = li - v;
                if (lane == 63) s_round = li;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t i = base + ((uint32_t)q * kFinBlock + t) * 4;
                if (i < row_hi) out[i / 4] = s_carry + s_wt[q * (kFinBlock / 64) + w] + inc[q] - sum[q];
            }
            __syncthreads();
            if (t == 0) s_carry += s_round;
            __syncthreads();
        }
        if (t == 0) p.chunk_tot[(size_t)c * kMaxChunks + g] = s_carry;  // (the column totals: second level)
        KPROF(blockIdx.x * 16 + w, 2);
        finalize_tail(p);
        return;
    }

    // reference-length monoids, in tile order (associative, not commutative): this workgroup folds the tiles
    // [fb * chunk, (fb+1) * chunk) of every source file into one partial record
    const uint32_t fb = blockIdx.x - nscan;
    const uint32_t chunk = (p.ntiles + p.nfold - 1) / p.nfold;
    const uint32_t c0 = fb * chunk, c1 = min(c0 + chunk, p.ntiles);
    const uint32_t per = (chunk + kFinBlock - 1) / kFinBlock;
    for (int b = 0; b < p.nbams; ++b) {
        const MonoRec* mo = p.tile_mono + (size_t)b * p.tstride;
        MonoRec acc;
        acc.ft = -1; acc.fp = 0; acc.lt = 0; acc.lp = 0; acc.sum = 0;
        const uint32_t t0 = min(c0 + (uint32_t)t * per, c1), t1 = min(t0 + per, c1);
        for (uint32_t i = t0; i < t1; i += 8) {  // batches of 8 independent loads, then the ordered fold
            MonoRec e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (i + q < t1) e[q] = mo[i + q];
                else e[q].ft = -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = mono_combine(acc, e[q]);
        }
        if (acc.ft != -12345) KPROF(blockIdx.x * 16 + w, 1);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const MonoRec other = mono_shfl_down(acc, o);
            if (lane + o < 64 && ((lane & (2 * o - 1)) == 0)) acc = mono_combine(acc, other);
        }
        if (lane == 0) s_mono[w] = acc;
        __syncthreads();
        if (t == 0) {
            MonoRec tot = s_mono[0];
            for (int k = 1; k < kFinBlock / 64; ++k) tot = mono_combine(tot, s_mono[k]);
            p.fold_part[(size_t)b * p.nfold + fb] = tot;
        }
        __syncthreads();
    }
    KPROF(blockIdx.x * 16 + w, 2);
    finalize_tail(p);
}

__global__ __launch_bounds__(256) void finalize2_kernel(const FinalizeParams p) { finalize2_body(p); }

__global__ __launch_bounds__(256) void k0_init_kernel(const InitList l) {
    for (int f = 0; f < l.n; ++f)
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < l.words[f]; i += gridDim.x * 256) l.ptr[f][i] = l.value[f];
}

void launch_init(const InitList& l, hipStream_t s) {
    uint32_t mx = 0;
    for (int f = 0; f < l.n; ++f) mx = l.words[f] > mx ? l.words[f] : mx;
    if (!mx) return;
    const uint32_t g = (mx + 1023) / 1024;
    hipLaunchKernelGGL(k0_init_kernel, dim3(g < 1024u ? g : 1024u), dim3(256), 0, s, l);
}

void launch_finalize(const FinalizeParams& p, hipStream_t s, bool second_level) {
    FinalizeParams q = p;
    if (!second_level) q.done = nullptr;  // (a workgroup of K2 runs the second level)
    hipLaunchKernelGGL(finalize_kernel, dim3(p.ncols * p.nchunk + p.nfold), dim3(kFinBlock), 0, s, q);
    if (second_level && !p.done) hipLaunchKernelGGL(finalize2_kernel, dim3(1), dim3(256), 0, s, p);
}

void launch_finalize2_only(const FinalizeParams& p, hipStream_t s) { hipLaunchKernelGGL(finalize2_kernel, dim3(1), dim3(256), 0, s, p); }

}  // namespace bdx

// (bdx_warm_up: the HIP runtime loads a translation unit's device code at the first launch of any of its kernels)
__global__ void k1_noop_kernel() {}
namespace bdx { void warm_k1(hipStream_t s) { hipLaunchKernelGGL(k1_noop_kernel, dim3(1), dim3(64), 0, s); } }

#ifdef BDX_KPROF
extern "C" int bdx_debug_kprof1(unsigned long long* out, size_t n) {
    const int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bdx::g_kprof), n * sizeof(unsigned long long));
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(bdx::g_kprof)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 8 * 65536);
    return rc;
}
#endif
