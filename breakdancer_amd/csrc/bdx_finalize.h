// Second level of the pass-1 finalisation, as a device function: it runs as its own one-workgroup kernel
// (finalize2_kernel) or as one extra workgroup of K2 when that kernel is enqueued behind it anyway (enqueue-ahead runs).
#pragma once
#include <cstddef>

#include "bdx_dev.h"

namespace bdx {

__device__ __forceinline__ MonoRec mono_combine(const MonoRec& a, const MonoRec& b) {
    if (a.ft == -1) return b;
    if (b.ft == -1) return a;
    MonoRec r;
    r.ft = a.ft; r.fp = a.fp; r.lt = b.lt; r.lp = b.lp;
    r.sum = a.sum + b.sum + (a.lt == b.ft ? (long long)b.fp - (long long)a.lp : 0ll);
    return r;
}
__device__ __forceinline__ MonoRec mono_shfl_down(const MonoRec& a, int o) {
    MonoRec r;
    r.ft = __shfl_down(a.ft, o); r.fp = __shfl_down(a.fp, o); r.lt = __shfl_down(a.lt, o); r.lp = __shfl_down(a.lp, o);
    r.sum = __shfl_down(a.sum, o);
    return r;
}

// one small workgroup: counters, the last level of the monoid fold, covered_ref_len (BamSummary.cpp:123-126) and the
// final window (BreakDancerMax.cpp:109-116)
__device__ __forceinline__ void finalize2_body(const FinalizeParams& p) {
    __shared__ uint32_t s_acc[255 * 12 + 256];  // nlibs*11 + nlibs + nbams at the documented limits
    __shared__ unsigned long long s_ref[256];
    __shared__ uint32_t s_head[2];  // covered_ref_len, window
    __shared__ uint32_t s_col[64];  // totals of the tile-total columns: n_anom, n_normal, key_tot[]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // counters: [kCntCopies][ncnt] -> [ncnt].  Every word is requested at once (contiguous across the workgroup) and added into
    // LDS; a loop of one reduction per counter waited for a round trip per iteration
    for (int i = t; i < p.ncnt; i += 256) s_acc[i] = 0;
    __syncthreads();
    for (int idx = t; idx < kCntCopies * p.ncnt; idx += 256) {
        const uint32_t v = p.blk_cnt[idx];
        if (v) atomicAdd(&s_acc[idx % p.ncnt], v);
    }
    __syncthreads();
    for (int i = t; i < p.ncnt; i += 256) {
        const uint32_t v = s_acc[i];
        p.cnt[i] = v;
        if (p.cnt_host) p.cnt_host[i] = v;
    }
    for (int c0 = 0; c0 < p.ncols; c0 += 4) {  // one wave per column: its chunks' totals
        const int c = c0 + w;
        uint32_t v = (c < p.ncols && (uint32_t)lane < p.nchunk) ? p.chunk_tot[(size_t)c * kMaxChunks + lane] : 0u;
        {   // what K2 adds to a chunk-local prefix: the totals of the chunks before (exclusive scan over the lanes)
            uint32_t inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(inc, o);
                if (lane >= o) inc += u;
            }
            if (c < p.ncols) p.chunk_base[(size_t)c * kMaxChunks + lane] = inc - v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && c < p.ncols) {
            s_col[c] = v;
            if (c == kColAnom) p.p1->n_anom = v;
            else if (c == kColNormal) p.p1->n_normal = v;
            else p.p1->key_tot[c - kColKey0] = v;
        }
    }
    for (int b0 = 0; b0 < p.nbams; b0 += 4) {  // one wave per source file: ordered tree fold of its <= 64 partial folds
        const int b = b0 + w;
        MonoRec acc;
        acc.ft = -1; acc.fp = 0; acc.lt = 0; acc.lp = 0; acc.sum = 0;
        if (b < p.nbams && (uint32_t)lane < p.nfold) acc = p.fold_part[(size_t)b * p.nfold + lane];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const MonoRec other = mono_shfl_down(acc, o);
            if (lane + o < 64 && ((lane & (2 * o - 1)) == 0)) acc = mono_combine(acc, other);
        }
        if (lane == 0 && b < p.nbams) {
            const unsigned long long r = acc.ft == -1 ? 0ull : (unsigned long long)acc.sum;  // size_t ref_len, wraps like the reference
            s_ref[b] = r;
            p.p1->ref_len[b] = r;
        }
    }
    __syncthreads();
    if (t == 0) {
        uint32_t covered = 0;
        for (int b = 0; b < p.nbams; ++b)
            if ((unsigned long long)covered < s_ref[b]) covered = (uint32_t)s_ref[b];
        p.p1->covered_ref_len = covered;
        int W = p.w0;
        for (int i = 0; i < p.nlibs; ++i) {
            const int nd = (int)(s_acc[i * kNumFlags + F_LARGE] + s_acc[i * kNumFlags + F_SMALL]);
            const int tmp = nd > 0 ? (int)__fdiv_rn((float)covered, (float)nd) : 50;
            W = min(W, tmp);
        }
        p.p1->window = W;
        s_head[0] = covered; s_head[1] = (uint32_t)W;
        if (p.key_density) {  // same float32 expressions as the host side (set_pass1 in bdx_api.hip)
            const uint32_t* lib_cnt = s_acc + p.nlibs * kNumFlags;
            const uint32_t* bam_cnt = lib_cnt + p.nlibs;
            for (int k = 0; k < p.nkeys; ++k) p.key_density[k] = 0.000001f;
            for (int i = 0; i < p.nlibs; ++i) {
                const int key = p.libs[i].key;  // library index with -a, else the library's source file
                float dens = 0.000001f;
                if (p.cn_lib) {
                    if (lib_cnt[i] != 0) dens = __fdiv_rn((float)lib_cnt[i], (float)covered);
                } else {
                    dens = __fdiv_rn((float)bam_cnt[key], (float)covered);
                }
                p.key_density[key] = dens;
            }
        }
    }
    if (p.p1_host) {
        // mirror the finished record into pinned host memory, from LDS: reading it back would cost a fence and a round trip on
        // the path the host waits for
        __syncthreads();
        uint32_t* dst = (uint32_t*)p.p1_host;
        static_assert(offsetof(Pass1, covered_ref_len) == 0 && offsetof(Pass1, window) == 4 && offsetof(Pass1, n_anom) == 8 &&
                          offsetof(Pass1, n_normal) == 12 && offsetof(Pass1, key_tot) == 16 && offsetof(Pass1, ref_len) % 8 == 0, "Pass1 mirror");
        static_assert(kColAnom == 0 && kColNormal == 1 && kColKey0 == 2, "the record's words 2.. are the column totals in column order");
        constexpr int kRefWord = (int)(offsetof(Pass1, ref_len) / 4);
        const int words = kRefWord + 2 * p.nbams;
        for (int i = t; i < words; i += 256) {
            uint32_t v;
            if (i < 2) v = s_head[i];
            else if (i < kRefWord) v = i - 2 < p.ncols ? s_col[i - 2] : 0u;
            else v = (uint32_t)(s_ref[(i - kRefWord) >> 1] >> (32 * ((i - kRefWord) & 1)));
            dst[i] = v;
        }
        if (p.flag_host) {
            __threadfence_system();
            __syncthreads();
            if (t == 0) *(volatile uint32_t*)p.flag_host = p.flag_value;
        }
    }
    if (p.na_cap) {
        __syncthreads();
        if (t == 0 && s_col[kColAnom] > p.na_cap) p.p1->n_anom = 0;
    }
}

}  // namespace bdx
