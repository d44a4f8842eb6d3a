// K1 -- per-read classification, pass-1 statistics and per-tile totals, one streaming pass.
//
// Replaces (reference file:line under src/lib):
//   io/IlluminaPEReadClassifier.cpp:13-101   pe_classify / classify
//   io/BamSummary.cpp:68-114                 pass-1 body (ref_len, proper counts, flag histogram)
//   breakdancer/BreakDancer.cpp:155-207      filter chain, -l remaps, RR->FF fold, normal-read tests
//
// HBM-bound: 25 B/read in (5 x i32, u16 flag, u8 mapq/lib/bam), 1 B/read out (class byte).
// Layout: a tile is 1024 consecutive reads; thread t of the 256-thread workgroup owns reads
// [4t, 4t+4) so every array is fetched with one 16/8/4-byte load per lane (1 KiB per wave-instruction
// for the i32 arrays).  No atomics leave the CU: counters are privatised in LDS and written per
// workgroup; per-tile totals and per-tile reference-length monoids go to plain per-tile tables that a
// tiny follow-up kernel scans/reduces.
#include "bdx_dev.h"

#include <limits.h>

namespace bdx {

__device__ __forceinline__ int classify_read(unsigned sam, int tid, int mtid, int pos, int mpos, int ai, float upper,
                                             float lower) {
    if ((sam & 0x400u) || !(sam & 0x1u)) return F_NA;
    if (sam & 0x4u) return F_UNMAPPED;
    if (sam & 0x8u) return F_MATE_UNMAPPED;
    if (tid != mtid) return F_CTX;
    const bool rr = sam & 0x10u, mr = sam & 0x20u;
    if (rr == mr) return rr ? F_RR : F_FF;
    if ((pos < mpos) == rr) return F_RF;
    const float fi = (float)ai;  // the reference compares int against the float cutoffs
    if (fi > upper) return F_LARGE;
    if (fi < lower) return F_SMALL;
    return F_NORMAL_FR;
}

__device__ __forceinline__ int remap_long_insert(int f, int ai, float upper, float lower) {
    const float fi = (float)ai;
    if (fi > upper && f == F_NORMAL_RF) f = F_RF;
    if (fi < upper && f == F_RF) f = F_NORMAL_RF;
    if (fi < lower && f == F_NORMAL_RF) f = F_SMALL;
    return f;
}

size_t k1_lds_bytes(int nlibs, int nbams, int nkeys) {
    size_t b = 0;
    b += (size_t)nlibs * sizeof(DevLib);
    b += (size_t)(nlibs * kNumFlags + nlibs + nbams) * 4;
    b += (size_t)(2 + nkeys) * 4;
    b += 16 * 4;                          // per-wave tid / uniform / valid words
    b += (size_t)kWaves * nbams * 3 * 4;  // per-wave per-bam (present, first_pos, last_pos)
    b = (b + 15) & ~(size_t)15;
    b += kTile * 4 * 2 + kTile;           // general-path staging: tid, pos, bam
    b = (b + 15) & ~(size_t)15;
    b += (size_t)nbams * (8 + 4 * 4);     // general-path per-bam sum, first(tid,pos), first idx, last idx
    return (b + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(kBlock) void k1_classify_kernel(const K1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nlibs = p.nlibs, nbams = p.nbams, nkeys = p.nkeys;
    const int ncnt = nlibs * kNumFlags + nlibs + nbams;
    const int ncols = 2 + nkeys;
    DevLib* s_lib = (DevLib*)smem;
    uint32_t* s_cnt = (uint32_t*)(s_lib + nlibs);
    uint32_t* s_tile = s_cnt + ncnt;
    int32_t* s_wave = (int32_t*)(s_tile + ncols);  // [16]
    int32_t* s_mw = s_wave + 16;                   // [kWaves][nbams][3]
    size_t off = (size_t)((unsigned char*)(s_mw + kWaves * nbams * 3) - smem);
    off = (off + 15) & ~(size_t)15;
    int32_t* g_tid = (int32_t*)(smem + off);
    int32_t* g_pos = g_tid + kTile;
    uint8_t* g_bam = (uint8_t*)(g_pos + kTile);
    off += kTile * 9;
    off = (off + 15) & ~(size_t)15;
    long long* g_sum = (long long*)(smem + off);
    int32_t* g_first = (int32_t*)(g_sum + nbams);  // [nbams][2]
    int32_t* g_fidx = g_first + 2 * nbams;
    int32_t* g_lidx = g_fidx + nbams;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < nlibs; i += kBlock) s_lib[i] = p.libs[i];
    for (int i = t; i < ncnt; i += kBlock) s_cnt[i] = 0;
    uint32_t* s_libcnt = s_cnt + nlibs * kNumFlags;
    uint32_t* s_bamcnt = s_libcnt + nlibs;
    __syncthreads();

    for (uint32_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        for (int i = t; i < ncols; i += kBlock) s_tile[i] = 0;
        for (int i = t; i < kWaves * nbams; i += kBlock) s_mw[i * 3] = 0;
        __syncthreads();

        const uint64_t base = (uint64_t)tile * kTile + (uint64_t)t * 4;
        int nvalid = 0;
        int tid[4], pos[4], mtid[4], mpos[4], isz[4];
        unsigned sam[4], mq[4], lib[4], bam[4];
        if (base + 4 <= p.n) {
            nvalid = 4;
            const int4 a = *(const int4*)(p.r.tid + base);
            const int4 b = *(const int4*)(p.r.pos + base);
            const int4 c = *(const int4*)(p.r.mtid + base);
            const int4 d = *(const int4*)(p.r.mpos + base);
            const int4 e = *(const int4*)(p.r.isize + base);
            const ushort4 f = *(const ushort4*)(p.r.flag + base);
            const uchar4 q = *(const uchar4*)(p.r.mapq + base);
            const uchar4 l = *(const uchar4*)(p.r.lib + base);
            const uchar4 m = *(const uchar4*)(p.r.bam + base);
            tid[0] = a.x; tid[1] = a.y; tid[2] = a.z; tid[3] = a.w;
            pos[0] = b.x; pos[1] = b.y; pos[2] = b.z; pos[3] = b.w;
            mtid[0] = c.x; mtid[1] = c.y; mtid[2] = c.z; mtid[3] = c.w;
            mpos[0] = d.x; mpos[1] = d.y; mpos[2] = d.z; mpos[3] = d.w;
            isz[0] = e.x; isz[1] = e.y; isz[2] = e.z; isz[3] = e.w;
            sam[0] = f.x; sam[1] = f.y; sam[2] = f.z; sam[3] = f.w;
            mq[0] = q.x; mq[1] = q.y; mq[2] = q.z; mq[3] = q.w;
            lib[0] = l.x; lib[1] = l.y; lib[2] = l.z; lib[3] = l.w;
            bam[0] = m.x; bam[1] = m.y; bam[2] = m.z; bam[3] = m.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint64_t i = base + r;
                const bool v = i < p.n;
                nvalid += v;
                tid[r] = v ? p.r.tid[i] : 0; pos[r] = v ? p.r.pos[i] : 0; mtid[r] = v ? p.r.mtid[i] : 0;
                mpos[r] = v ? p.r.mpos[i] : 0; isz[r] = v ? p.r.isize[i] : 0; sam[r] = v ? p.r.flag[i] : 0;
                mq[r] = v ? p.r.mapq[i] : 0; lib[r] = v ? p.r.lib[i] : 0; bam[r] = v ? p.r.bam[i] : 0;
            }
        }

        unsigned cls4[4];
        bool p1c[4], pk[4], anom[4], nleft[4];
        int key[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = r < nvalid;
            const unsigned L = lib[r] < (unsigned)nlibs ? lib[r] : 0u;
            lib[r] = L;
            if (bam[r] >= (unsigned)nbams) bam[r] = 0;
            const DevLib dl = s_lib[L];
            const int ai = abs(isz[r]);
            const int f = classify_read(sam[r], tid[r], mtid[r], pos[r], mpos[r], ai, dl.upper, dl.lower);
            const bool mq_ok = (int)mq[r] > dl.min_mapq;
            const bool proper = (sam[r] & 0x40Fu) == 0x3u;
            p1c[r] = valid && mq_ok && proper;
            const bool h_ok = valid && mq_ok && f != F_NA && !(sam[r] & 0xCu) && !(p.opt_t && tid[r] == mtid[r]);
            const int f1 = p.opt_l ? remap_long_insert(f, ai, dl.upper, dl.lower) : f;
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
        if (nvalid == 4) {
            *(uchar4*)(p.cls + base) = make_uchar4(cls4[0], cls4[1], cls4[2], cls4[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nvalid) p.cls[base + r] = (uint8_t)cls4[r];
        }

        // ---- per-tile totals and block counters: ballots + popcounts, one LDS add per wave -------------
        {
            unsigned na = 0, nn = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { na += popc64(ballot64(anom[r])); nn += popc64(ballot64(nleft[r])); }
            if (lane == 0) {
                if (na) atomicAdd(&s_tile[kColAnom], na);
                if (nn) atomicAdd(&s_tile[kColNormal], nn);
            }
            // uniform fast path: every lane/slot has the same library and source file (the usual case)
            const unsigned L0 = __shfl(lib[0], 0), B0 = __shfl(bam[0], 0);
            const bool uni = __all(lib[0] == L0 && lib[1] == L0 && lib[2] == L0 && lib[3] == L0 && bam[0] == B0 &&
                                   bam[1] == B0 && bam[2] == B0 && bam[3] == B0);
            if (uni) {
                unsigned c1 = 0, ck = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { c1 += popc64(ballot64(p1c[r])); ck += popc64(ballot64(pk[r])); }
                if (lane == 0) {
                    if (c1) { atomicAdd(&s_libcnt[L0], c1); atomicAdd(&s_bamcnt[B0], c1); }
                    if (ck) atomicAdd(&s_tile[kColKey0 + key[0]], ck);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint64_t todo = ballot64(p1c[r]);
                    while (todo) {  // peel distinct libraries
                        const int ld = __ffsll((long long)todo) - 1;
                        const unsigned v = __shfl(lib[r], ld);
                        const uint64_t m = ballot64(p1c[r] && lib[r] == v);
                        if (lane == 0) atomicAdd(&s_libcnt[v], (unsigned)popc64(m));
                        todo &= ~m;
                    }
                    todo = ballot64(p1c[r]);
                    while (todo) {  // peel distinct source files
                        const int ld = __ffsll((long long)todo) - 1;
                        const unsigned v = __shfl(bam[r], ld);
                        const uint64_t m = ballot64(p1c[r] && bam[r] == v);
                        if (lane == 0) atomicAdd(&s_bamcnt[v], (unsigned)popc64(m));
                        todo &= ~m;
                    }
                    todo = ballot64(pk[r]);
                    while (todo) {  // peel distinct normal-read keys
                        const int ld = __ffsll((long long)todo) - 1;
                        const int v = __shfl(key[r], ld);
                        const uint64_t m = ballot64(pk[r] && key[r] == v);
                        if (lane == 0) atomicAdd(&s_tile[kColKey0 + v], (unsigned)popc64(m));
                        todo &= ~m;
                    }
                }
            }
        }

        // ---- reference-length monoid per source file (BamSummary.cpp:70-74), wave part -------------------
        {
            const uint64_t vm = ballot64(nvalid > 0);
            if (vm) {
                const int tw = __shfl(tid[0], 0);
                bool same = true;
#pragma unroll
                for (int r = 0; r < 4; ++r) same = same && (r >= nvalid || tid[r] == tw);
                const bool wuni = __all(same);
                if (lane == 0) { s_wave[w * 3] = tw; s_wave[w * 3 + 1] = wuni; s_wave[w * 3 + 2] = 1; }
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
                    if (lane == 0) { int32_t* e = s_mw + (w * nbams + v) * 3; e[0] = 1; e[1] = first; e[2] = last; }
                }
            } else if (lane == 0) {
                s_wave[w * 3 + 2] = 0;
            }
        }
        __syncthreads();

        for (int c = t; c < ncols; c += kBlock) p.tile_tot[(size_t)c * p.tstride + tile] = s_tile[c];

        bool tile_uniform = true;
        int tile_tid = 0;
        {
            bool have = false;
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) {
                if (!s_wave[ww * 3 + 2]) continue;
                if (!s_wave[ww * 3 + 1]) tile_uniform = false;
                if (!have) { tile_tid = s_wave[ww * 3]; have = true; }
                else if (s_wave[ww * 3] != tile_tid) tile_uniform = false;
            }
        }
        if (tile_uniform) {
            // all records of the tile share one tid: consecutive same-file differences telescope exactly
            for (int b = t; b < nbams; b += kBlock) {
                bool have = false;
                int first = 0, last = 0;
#pragma unroll
                for (int ww = 0; ww < kWaves; ++ww) {
                    const int32_t* e = s_mw + (ww * nbams + b) * 3;
                    if (!e[0]) continue;
                    if (!have) { first = e[1]; have = true; }
                    last = e[2];
                }
                int32_t* mo = p.tile_mono + (size_t)b * 4 * p.tstride + tile;
                mo[0] = have ? tile_tid : INT_MIN;
                mo[p.tstride] = first;
                mo[2 * (size_t)p.tstride] = tile_tid;
                mo[3 * (size_t)p.tstride] = last;
                p.tile_mono_sum[(size_t)b * p.tstride + tile] = have ? (long long)last - (long long)first : 0;
            }
        } else {
            // general path (a tid boundary falls inside the tile): exact recurrence via LDS staging
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                g_tid[t * 4 + r] = tid[r];
                g_pos[t * 4 + r] = pos[r];
                g_bam[t * 4 + r] = r < nvalid ? (uint8_t)bam[r] : (uint8_t)255;
            }
            for (int b = t; b < nbams; b += kBlock) { g_sum[b] = 0; g_fidx[b] = INT_MAX; g_lidx[b] = -1; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r >= nvalid) continue;
                const int q = t * 4 + r;
                const unsigned v = bam[r];
                int qq = q - 1;
                while (qq >= 0 && g_bam[qq] != v) --qq;
                if (qq >= 0) {
                    if (g_tid[qq] == tid[r]) atomicAdd((unsigned long long*)&g_sum[v], (unsigned long long)((long long)pos[r] - (long long)g_pos[qq]));
                } else {
                    g_fidx[v] = q;
                }
                atomicMax(&g_lidx[v], q);
            }
            __syncthreads();
            for (int b = t; b < nbams; b += kBlock) {
                const bool have = g_lidx[b] >= 0;
                int32_t* mo = p.tile_mono + (size_t)b * 4 * p.tstride + tile;
                mo[0] = have ? g_tid[g_fidx[b]] : INT_MIN;
                mo[p.tstride] = have ? g_pos[g_fidx[b]] : 0;
                mo[2 * (size_t)p.tstride] = have ? g_tid[g_lidx[b]] : 0;
                mo[3 * (size_t)p.tstride] = have ? g_pos[g_lidx[b]] : 0;
                p.tile_mono_sum[(size_t)b * p.tstride + tile] = have ? g_sum[b] : 0;
            }
        }
        __syncthreads();
    }

    uint32_t* out = p.blk_cnt + (size_t)blockIdx.x * ncnt;
    for (int i = t; i < ncnt; i += kBlock) out[i] = s_cnt[i];
}

void launch_k1(const K1Params& p, int grid, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(k1_classify_kernel, dim3(grid), dim3(kBlock), lds, s, p);
}

// -------------------------------------------------------------------------------------------------------
// finalize + tile scan.  grid = ncols + 1 workgroups of 1024 threads:
//   workgroup c < ncols : exclusive scan of tile_tot[c][*] -> tile_pre[c][*], column total -> Pass1
//   workgroup ncols     : reduce the per-workgroup counters, fold the per-tile reference-length monoids,
//                         covered_ref_len (BamSummary.cpp:123-126) and the final window
//                         (BreakDancerMax.cpp:109-116)
// -------------------------------------------------------------------------------------------------------
constexpr int kFinBlock = 1024;

struct Mono {
    int ft, fp, lt, lp;
    long long sum;
};
__device__ __forceinline__ Mono mono_combine(const Mono& a, const Mono& b) {
    if (a.ft == INT_MIN) return b;
    if (b.ft == INT_MIN) return a;
    Mono r;
    r.ft = a.ft; r.fp = a.fp; r.lt = b.lt; r.lp = b.lp;
    r.sum = a.sum + b.sum + (a.lt == b.ft ? (long long)b.fp - (long long)a.lp : 0ll);
    return r;
}
__device__ __forceinline__ Mono mono_shfl_down(const Mono& a, int o) {
    Mono r;
    r.ft = __shfl_down(a.ft, o); r.fp = __shfl_down(a.fp, o); r.lt = __shfl_down(a.lt, o); r.lp = __shfl_down(a.lp, o);
    r.sum = __shfl_down(a.sum, o);
    return r;
}

__global__ __launch_bounds__(kFinBlock) void finalize_kernel(const FinalizeParams p) {
    __shared__ uint32_t s_ws[kFinBlock / 64];
    __shared__ uint32_t s_carry;
    __shared__ Mono s_mono[kFinBlock / 64];
    __shared__ unsigned long long s_ref[256];
    __shared__ uint32_t s_acc[255 * 12 + 256];  // nlibs*11 + nlibs + nbams at the documented limits
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    if ((int)blockIdx.x < p.ncols) {
        const int c = blockIdx.x;
        const uint32_t* in = p.tile_tot + (size_t)c * p.tstride;
        uint32_t* out = p.tile_pre + (size_t)c * p.tstride;
        if (t == 0) s_carry = 0;
        __syncthreads();
        for (uint32_t base = 0; base < p.ntiles; base += kFinBlock * 4) {
            const uint32_t i = base + t * 4;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i < p.ntiles) v = *(const uint4*)(in + i);  // rows are padded to a multiple of 4 and zero-filled
            const uint32_t s1 = v.x, s2 = s1 + v.y, s3 = s2 + v.z, s4 = s3 + v.w;
            const uint32_t inc = wave_incl_scan(s4);
            if (lane == 63) s_ws[w] = inc;
            __syncthreads();
            uint32_t woff = 0;
            for (int k = 0; k < w; ++k) woff += s_ws[k];
            const uint32_t ex = s_carry + woff + inc - s4;
            if (i < p.ntiles) *(uint4*)(out + i) = make_uint4(ex, ex + s1, ex + s2, ex + s3);
            __syncthreads();
            if (t == kFinBlock - 1) s_carry = ex + s4;
            __syncthreads();
        }
        if (t == 0) {
            if (c == kColAnom) p.p1->n_anom = s_carry;
            else if (c == kColNormal) p.p1->n_normal = s_carry;
            else p.p1->key_tot[c - kColKey0] = s_carry;
        }
        return;
    }

    // counters: [nblk][ncnt] -> [ncnt]; coalesced sweep over the whole table, LDS atomics do the transpose
    for (int i = t; i < p.ncnt; i += kFinBlock) s_acc[i] = 0;
    __syncthreads();
    {
        const uint32_t total = p.nblk * (uint32_t)p.ncnt;
        for (uint32_t i = t; i < total; i += kFinBlock) {
            const uint32_t v = p.blk_cnt[i];
            if (v) atomicAdd(&s_acc[i % (uint32_t)p.ncnt], v);
        }
    }
    __syncthreads();
    for (int i = t; i < p.ncnt; i += kFinBlock) p.cnt[i] = s_acc[i];
    // reference-length monoids, in tile order (associative, not commutative)
    const uint32_t per = (p.ntiles + kFinBlock - 1) / kFinBlock;
    for (int b = 0; b < p.nbams; ++b) {
        const int32_t* mo = p.tile_mono + (size_t)b * 4 * p.tstride;
        const long long* ms = p.tile_mono_sum + (size_t)b * p.tstride;
        Mono acc;
        acc.ft = INT_MIN; acc.fp = 0; acc.lt = 0; acc.lp = 0; acc.sum = 0;
        const uint32_t t0 = (uint32_t)t * per, t1 = min(t0 + per, p.ntiles);
        for (uint32_t i = t0; i < t1; ++i) {
            Mono e;
            e.ft = mo[i]; e.fp = mo[p.tstride + i]; e.lt = mo[2 * (size_t)p.tstride + i]; e.lp = mo[3 * (size_t)p.tstride + i];
            e.sum = ms[i];
            acc = mono_combine(acc, e);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const Mono other = mono_shfl_down(acc, o);
            if (lane + o < 64 && ((lane & (2 * o - 1)) == 0)) acc = mono_combine(acc, other);
        }
        if (lane == 0) s_mono[w] = acc;
        __syncthreads();
        if (t == 0) {
            Mono tot = s_mono[0];
            for (int k = 1; k < kFinBlock / 64; ++k) tot = mono_combine(tot, s_mono[k]);
            if (b < 256) s_ref[b] = (unsigned long long)tot.sum;  // size_t ref_len, wraps like the reference
        }
        __syncthreads();
    }
    if (t == 0) {
        uint32_t covered = 0;
        for (int b = 0; b < p.nbams && b < 256; ++b)
            if ((unsigned long long)covered < s_ref[b]) covered = (uint32_t)s_ref[b];
        p.p1->covered_ref_len = covered;
    }
    __syncthreads();
    if (t == 0) {
        // p.cnt was written by other threads of this workgroup above; __syncthreads() orders it
        int W = p.w0;
        const uint32_t covered = p.p1->covered_ref_len;
        for (int i = 0; i < p.nlibs; ++i) {
            const int nd = (int)(p.cnt[i * kNumFlags + F_LARGE] + p.cnt[i * kNumFlags + F_SMALL]);
            const int tmp = nd > 0 ? (int)__fdiv_rn((float)covered, (float)nd) : 50;
            W = min(W, tmp);
        }
        p.p1->window = W;
    }
}

void launch_finalize(const FinalizeParams& p, hipStream_t s) {
    hipLaunchKernelGGL(finalize_kernel, dim3(p.ncols + 1), dim3(kFinBlock), 0, s, p);
}

}  // namespace bdx
