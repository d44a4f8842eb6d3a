// K9 -- what makes ONE context that holds several (not necessarily neighbouring) chromosomes of a genome behave like its share
// of a single run over the whole genome (bdx_shard.h, bdx_dist_impl.h).
//
// The reference runs one record stream through one set of counters (BreakDancer.cpp:147-242): normal read pairs and proper reads
// per library / file seen so far, the accepted regions so far, the read that closes a candidate region.  Here every rank runs the
// single-context kernels (K1-K6) once over all of its chromosomes, and the few quantities that cross a chromosome boundary come
// from small tables indexed by chromosome:
//   k9_tid_table     where each chromosome starts in the context's stream and what the counters read there (from K1's per-tile
//                    totals: no second pass over the reads)
//   k9_rebase        compact records: counters shifted by what the chromosomes in front (anybody's) have counted
//   k9_tid_regions   regions per chromosome
//   k9_globalize     region ids -> genome-wide ids, the region table laid out by genome-wide id
//   k9_window_*      the read length the walk uses at a flush belongs to one region per window, wherever it lives
//   k9_merge_*       rank 0: the ranks' SV tables, each sorted by order key, into one
#include "bdx_shard.h"

#include <cstddef>

#include "bdx_scan.h"

namespace bdx {

namespace {

__device__ __forceinline__ int lib_index_of(const uint8_t* lib, int nlibs, uint64_t i) {   // as K1 / K2 take it
    if (nlibs <= 1) return 0;
    const int l = lib[i];
    return l < nlibs ? l : 0;
}

}  // namespace

// one wave per chromosome boundary
__global__ __launch_bounds__(64) void k9_tid_table_kernel(TidTableParams p) {
    const int t = blockIdx.x;   // 0 .. ntids
    const int lane = threadIdx.x;
    const int W = 1 + p.ncols;
    uint64_t lo = 0, hi = p.n;   // first read with tid >= t (the stream is sorted by tid)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (p.tid[mid] < t) lo = mid + 1; else hi = mid;
    }
    const uint64_t s = lo;
    uint32_t* o = p.out + (size_t)t * W;
    uint32_t* err = p.out + (size_t)(p.ntids + 1) * W;
    if (lane == 0) {
        o[0] = (uint32_t)s;
        if (t == 0) err[0] = s != 0 ? 1u : 0u;              // a record with a negative reference id
        if (t == p.ntids) err[1] = s != p.n ? 1u : 0u;      // a record with a reference id beyond the header's sequences
    }
    if (s >= p.n) {   // the end of the stream: the columns' totals
        for (int c = lane; c < p.ncols; c += 64) o[1 + c] = c == kColAnom ? p.p1->n_anom : (c == kColNormal ? p.p1->n_normal : p.p1->key_tot[c - kColKey0]);
        return;
    }
    const uint32_t tile = (uint32_t)(s / kTile), T2 = tile / kK2TilesPerWave, chunk = T2 / p.chunk_super;
    unsigned cb[4];
    int key[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint64_t i = (uint64_t)tile * kTile + 4 * (uint64_t)lane + r;
        const bool in = i < s;
        cb[r] = in ? p.cls[i] : 0u;
        key[r] = (in && p.nkeys > 1) ? (p.libs[lib_index_of(p.lib, p.nlibs, i)].key & 255) : 0;
    }
    for (int c = 0; c < p.ncols; ++c) {
        uint32_t part = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned f = cb[r] & 15u;
            const bool pass = (cb[r] & 0x10u) != 0, normal = f == F_NORMAL_FR || f == F_NORMAL_RF;
            const bool hit = c == kColAnom ? (pass && !normal) : (c == kColNormal ? (pass && (cb[r] & 0x40u)) : (pass && (cb[r] & 0x20u) && key[r] == c - kColKey0));
            part += (uint32_t)__popcll(__ballot(hit));
        }
        if (lane == 0) {
            uint32_t base = p.chunk_base[(size_t)c * kMaxChunks + chunk] + p.tile_pre[(size_t)c * p.tstride + T2];
            for (uint32_t q = T2 * kK2TilesPerWave; q < tile; ++q) base += p.tile_tot[(size_t)c * p.tstride + q];
            o[1 + c] = base + part;
        }
    }
}

// Is the reference-id column what k9_tid_table_kernel's binary searches take it for -- ascending, every id within [0, ntids)?  One pass over
// the column (464 MB for a GPU's share of a genome: ~80 us); bdx_dist_prepare / the first bdx_dist_run on a set of reads run it once.
__global__ __launch_bounds__(256) void k9_check_sorted_kernel(const int32_t* __restrict__ tid, uint64_t n, int ntids, uint32_t* err) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const int32_t t = tid[i];
        bad |= t < 0 || t >= ntids || (i + 1 < n && tid[i + 1] < t);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *err = 1u;
}

void launch_k9_check_sorted(const int32_t* tid, uint64_t n, int ntids, uint32_t* err, hipStream_t s) {
    if (!n) return;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k9_check_sorted_kernel, dim3(grid), dim3(256), 0, s, tid, n, ntids, err);
}

void launch_k9_tid_table(const TidTableParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k9_tid_table_kernel, dim3((uint32_t)p.ntids + 1), dim3(64), 0, s, p);
}

__global__ __launch_bounds__(64) void k9_signal_kernel(uint32_t* flag, uint32_t value) {
    if (threadIdx.x == 0) {
        __threadfence_system();
        *(volatile uint32_t*)flag = value;
    }
}
void launch_k9_signal(uint32_t* flag, uint32_t value, hipStream_t s) { hipLaunchKernelGGL(k9_signal_kernel, dim3(1), dim3(64), 0, s, flag, value); }

__global__ __launch_bounds__(256) void k9_upload_kernel(const UploadList l) {
    for (int f = 0; f < l.n; ++f) {
        const uint32_t* src = l.src[f];
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < l.words[f]; i += gridDim.x * 256) l.dst[f][i] = src ? src[i] : l.value[f];
    }
}
void launch_k9_upload(const UploadList& l, hipStream_t s) {
    uint32_t mx = 0;
    for (int f = 0; f < l.n; ++f) mx = l.words[f] > mx ? l.words[f] : mx;
    if (!mx) return;
    const uint32_t g = (mx + 255) / 256;
    hipLaunchKernelGGL(k9_upload_kernel, dim3(g < 64u ? g : 64u), dim3(256), 0, s, l);
}

// a few words from HBM into the host's report area, then the ready word (one small launch instead of a copy command and a launch)
__global__ __launch_bounds__(64) void k9_report_kernel(const uint32_t* src, uint32_t* dst, uint32_t n, uint32_t* flag, uint32_t value) {
    for (uint32_t i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *(volatile uint32_t*)flag = value;
}
void launch_k9_report(const uint32_t* src, uint32_t* dst, uint32_t n, uint32_t* flag, uint32_t value, hipStream_t s) {
    hipLaunchKernelGGL(k9_report_kernel, dim3(1), dim3(64), 0, s, src, dst, n, flag, value);
}

__global__ __launch_bounds__(256) void k9_rebase_kernel(Compact cp, const uint32_t* n_ptr, int nkeys, const uint32_t* tid_off, uint32_t* first_tab) {
    const uint32_t n = *n_ptr;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int32_t t = cp.tid[j];
        const uint32_t* o = tid_off + (size_t)t * (1 + nkeys);
        const uint32_t nn = cp.nn[j] + o[0];
        cp.nn[j] = nn;
        for (int k = 0; k < nkeys; ++k) cp.pk[(size_t)k * cp.cap + j] += o[1 + k];
        if (first_tab && (j == 0 || cp.tid[j - 1] != t)) {   // the chromosome's first anomalous read: it closes the last candidate of the chromosome before
            first_tab[4 * (size_t)t] = 1u;
            first_tab[4 * (size_t)t + 1] = (uint32_t)meta_qlen(cp.meta[j]);
            first_tab[4 * (size_t)t + 2] = nn;
        }
    }
}
void launch_k9_rebase(const Compact& cp, const uint32_t* n_ptr, uint32_t n_upper, int nkeys, const uint32_t* tid_off, uint32_t* first_tab, hipStream_t s) {
    if (!n_upper) return;
    hipLaunchKernelGGL(k9_rebase_kernel, dim3(std::min<uint32_t>((n_upper + 255) / 256, 4096u)), dim3(256), 0, s, cp, n_ptr, nkeys, tid_off, first_tab);
}

__global__ __launch_bounds__(256) void k9_first_counts_kernel(FirstCountsParams p) {
    __shared__ uint32_t s_own[kMaxRanks];
    const Compact& cp = p.cp;
    for (uint32_t d = threadIdx.x; d < p.world; d += 256) s_own[d] = 0;
    __syncthreads();
    const uint32_t n = *p.n_ptr;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int32_t t = cp.tid[j];
        const uint32_t m = cp.meta[j];
        if (j == 0 || cp.tid[j - 1] != t) {
            p.first_tab[4 * (size_t)t] = 1u;
            p.first_tab[4 * (size_t)t + 1] = (uint32_t)meta_qlen(m);
            p.first_tab[4 * (size_t)t + 2] = cp.nn[j];
        }
        if (p.world > 1) {
            if (meta_flag(m) == F_CTX) {   // (rare: a global atomic each)
                const int32_t mt = p.mtid_col[cp.idx[j]];
                if (mt > t && mt < p.ntids) atomicAdd(&p.cnt_mtid[mt], 1u);
            }
            atomicAdd(&s_own[exchange_owner(cp.key[j], p.world)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < p.world; d += 256)
        if (p.world > 1 && s_own[d]) atomicAdd(&p.cnt_owner[d], s_own[d]);
}
void launch_k9_first_counts(const FirstCountsParams& p, uint32_t n_upper, hipStream_t s) {
    if (!n_upper) return;
    hipLaunchKernelGGL(k9_first_counts_kernel, dim3(std::min<uint32_t>((n_upper + 255) / 256, 2048u)), dim3(256), 0, s, p);
}

// window read lengths: pack into the other ranks' blocks of the all-to-all, unpack from the blocks received
__global__ __launch_bounds__(256) void k9_window_pack_kernel(const RegionRec* rg_rec, uint32_t nw, uint32_t period, WindowDst wd, uint32_t n_expected, uint32_t* cursor) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    const RegionRec* r = &rg_rec[(size_t)(w + 1) * period - 1];
    if (!r->n) return;   // (another rank's region: its place in this rank's table is empty)
    const uint32_t i = atomicAdd(cursor, 1u);
    if (i >= n_expected) return;   // (cannot happen: the host counted this rank's windows from the chromosomes' region ranges)
    const unsigned long long e = ((unsigned long long)w << 32) | (unsigned long long)(uint32_t)r->maxq;
    for (int q = 0; q < wd.world; ++q)
        if (wd.dst[q]) wd.dst[q][i] = e;
}
__global__ __launch_bounds__(256) void k9_window_unpack_kernel(RegionRec* rg_rec, uint32_t nw, uint32_t period, const unsigned long long* base, SegList sg, uint32_t n, uint32_t* err) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int q = sg.seg_of(j);
    const unsigned long long e = base[sg.off[q] + (j - sg.start[q])];
    const uint32_t w = (uint32_t)(e >> 32);
    if (w >= nw) { *err = 1u; return; }
    RegionRec* r = &rg_rec[(size_t)(w + 1) * period - 1];
    if (r->n) { *err = 1u; return; }   // (a window's last region has one owner)
    r->maxq = (int32_t)(uint32_t)e;
}
void launch_k9_window_pack(const RegionRec* rg_rec, uint32_t nr, uint32_t period, const WindowDst& wd, uint32_t n_expected, uint32_t* cursor, hipStream_t s) {
    const uint32_t nw = nr / period;
    if (nw && n_expected) hipLaunchKernelGGL(k9_window_pack_kernel, dim3((nw + 255) / 256), dim3(256), 0, s, rg_rec, nw, period, wd, n_expected, cursor);
}
void launch_k9_window_unpack(RegionRec* rg_rec, uint32_t nr, uint32_t period, const unsigned long long* base, const SegList& sg, uint32_t n, uint32_t* err, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k9_window_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rg_rec, nr / period, period, base, sg, n, err);
}

__global__ __launch_bounds__(64) void k9_tid_regions_kernel(const RegionRec* r_rec, const StageCounts* counts, int ntids, uint32_t* out) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    const uint32_t nr = counts->n_regions;
    if (t == 0) { out[ntids + 1] = nr; out[ntids + 2] = (uint32_t)counts->last_maxq; }
    if (t > ntids) return;
    uint32_t lo = 0, hi = nr;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (r_rec[mid].tid < t) lo = mid + 1; else hi = mid;
    }
    out[t] = lo;
}
void launch_k9_tid_regions(const RegionRec* r_rec, const StageCounts* counts, int ntids, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k9_tid_regions_kernel, dim3((uint32_t)(ntids + 1 + 63) / 64), dim3(64), 0, s, r_rec, counts, ntids, out);
}

__global__ __launch_bounds__(256) void k9_globalize_kernel(GlobalizeParams p) {
    const uint32_t i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    const uint32_t na = *p.n_ptr;
    for (uint32_t j = i0; j < na; j += stride) {
        const int32_t r = p.region_of[j];
        if (r >= 0) p.region_of[j] = r + (int32_t)p.roff[p.tid[j]];
    }
    for (uint32_t r = i0; r < p.nr_local; r += stride) {
        const RegionRec rec = p.r_rec[r];
        const size_t g = (size_t)r + p.roff[rec.tid];
        p.rg_rec[g] = rec;
        for (int k = 0; k < p.nkeys2; ++k) p.rg_pk[g * p.nkeys2 + k] = p.r_pk[(size_t)r * p.nkeys2 + k];
    }
    const size_t cap = p.cap;
    for (uint32_t q = i0; q < p.cap; q += stride) {   // out_deg, label, bad_v, bad, mcount, pcount
        p.scratch[q] = 0; p.scratch[cap + q] = q; p.scratch[2 * cap + q] = 0; p.scratch[3 * cap + q] = 0; p.scratch[4 * cap + q] = 0; p.scratch[5 * cap + q] = 0;
    }
    if (i0 == 0) { p.counts->n_regions = p.nr_global; p.counts->last_maxq = p.last_maxq; }
}
void launch_k9_globalize(const GlobalizeParams& p, uint32_t n_upper, hipStream_t s) {
    const uint32_t n = std::max(std::max(n_upper, p.cap), 1u);
    hipLaunchKernelGGL(k9_globalize_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 8192u)), dim3(256), 0, s, p);
}

__global__ __launch_bounds__(256) void k9_window_collect_kernel(const RegionRec* rg_rec, uint32_t nw, uint32_t period, unsigned long long* win) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    const RegionRec* r = &rg_rec[(size_t)(w + 1) * period - 1];
    win[w] = r->n ? (unsigned long long)(uint32_t)r->maxq : 0ull;
}
__global__ __launch_bounds__(256) void k9_window_apply_kernel(RegionRec* rg_rec, uint32_t nw, uint32_t period, const unsigned long long* win) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    RegionRec* r = &rg_rec[(size_t)(w + 1) * period - 1];
    if (!r->n) r->maxq = (int32_t)(uint32_t)win[w];
}
void launch_k9_window_collect(const RegionRec* rg_rec, uint32_t nr, uint32_t period, unsigned long long* win, hipStream_t s) {
    const uint32_t nw = nr / period;
    if (nw) hipLaunchKernelGGL(k9_window_collect_kernel, dim3((nw + 255) / 256), dim3(256), 0, s, rg_rec, nw, period, win);
}
void launch_k9_window_apply(RegionRec* rg_rec, uint32_t nr, uint32_t period, const unsigned long long* win, hipStream_t s) {
    const uint32_t nw = nr / period;
    if (nw) hipLaunchKernelGGL(k9_window_apply_kernel, dim3((nw + 255) / 256), dim3(256), 0, s, rg_rec, nw, period, win);
}

__global__ __launch_bounds__(256) void k9_pack_replay_kernel(Compact cp, const int32_t* region_of, const uint32_t* n_ptr, const uint32_t* tid_start,
                                                             unsigned long long* out) {
    const uint32_t n = *n_ptr;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int32_t t = cp.tid[j], r = region_of[j];
        unsigned long long* o = out + 5 * (size_t)j;
        o[0] = cp.key[j];
        o[1] = (unsigned long long)(r < 0 ? 0xFFFFFFFFu : (uint32_t)r) | ((unsigned long long)cp.meta[j] << 32);
        o[2] = (unsigned long long)(uint32_t)cp.isize[j] | ((unsigned long long)(uint32_t)t << 32);
        o[3] = cp.check ? cp.check[j] : 0ull;
        o[4] = (unsigned long long)(cp.idx[j] - tid_start[t]);
    }
}
void launch_k9_pack_replay(const Compact& cp, const int32_t* region_of, const uint32_t* n_ptr, uint32_t n_upper, const uint32_t* tid_start,
                           unsigned long long* out, hipStream_t s) {
    if (!n_upper) return;
    hipLaunchKernelGGL(k9_pack_replay_kernel, dim3(std::min<uint32_t>((n_upper + 255) / 256, 4096u)), dim3(256), 0, s, cp, region_of, n_ptr, tid_start, out);
}

// ---- rank 0: merge of the ranks' tables ------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ uint32_t keys_below(const unsigned long long* k, uint32_t n, unsigned long long key, bool or_equal) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const unsigned long long v = k[mid];
        if (v < key || (or_equal && v == key)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct BeginsIn {
    const uint2* b;
    __device__ U4 operator()(uint32_t j, uint32_t) const { const uint2 v = b[j]; return U4{v.x, v.y, 0u, 0u}; }
};
struct BeginsOut {
    uint2* b;
    __device__ void operator()(uint32_t j, uint32_t, const U4& inc, const U4& e) const { b[j] = make_uint2(inc.x - e.x, inc.y - e.y); }
};

}  // namespace

// a row's place: its index + the rows of the other tables in front of it (equal keys cannot come from two ranks; were they to, the
// lower rank goes first)
__global__ __launch_bounds__(256) void k9_merge_rank_kernel(const char* all, const TableDesc* Dp, uint32_t* src, uint2* begins) {
    const TableDesc& D = *Dp;   // (in device memory: 64 packages are more than a kernel takes as arguments)
    const int q = blockIdx.y;
    const TablePackage P = D.p[q];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n_sv) return;
    const unsigned long long key = ((const unsigned long long*)(all + P.keys_off))[i];
    uint32_t pos = i;
    for (int o = 0; o < D.world; ++o)
        if (o != q && D.p[o].n_sv) pos += keys_below((const unsigned long long*)(all + D.p[o].keys_off), D.p[o].n_sv, key, o < q);
    const SvOut* row = (const SvOut*)(all + P.rows_off) + i;
    src[pos] = ((uint32_t)q << 26) | i;
    begins[pos] = make_uint2((uint32_t)row->sv.lib_count, (uint32_t)row->sv.cn_count);
}

// One wave writes 64 consecutive rows of the merged table as one contiguous block (scattered single stores over PCIe are several
// times slower); the list entries go with their rows.
__global__ __launch_bounds__(256) void k9_merge_place_kernel(const char* all, const TableDesc* Dp, uint32_t n_total, const uint32_t* src, const uint2* begins,
                                                             MergeOut out) {
    constexpr int kLibBegin = (int)(offsetof(bdx_sv, lib_begin) / 4), kCnBegin = (int)(offsetof(bdx_sv, cn_begin) / 4);
    const int lane = threadIdx.x & 63;
    const uint32_t pos0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (pos0 >= n_total) return;
    const uint32_t cnt = min(64u, n_total - pos0);
    bool act = (uint32_t)lane < cnt;
    uint32_t sr = act ? src[pos0 + lane] : 0u;
    if (sr == 0xFFFFFFFFu) { sr = 0; act = false; }   // (a place no row claimed: cannot happen with distinct keys; the row stays as it is)
    const TablePackage P = Dp->p[act ? (sr >> 26) : 0];
    const uint32_t i = sr & ((1u << 26) - 1);
    const SvOut* row = (const SvOut*)(all + P.rows_off) + i;
    const uint2 bg = act ? begins[pos0 + lane] : make_uint2(0u, 0u);
    if (act) {
        const int32_t l0 = row->sv.lib_begin, nl = row->sv.lib_count, c0 = row->sv.cn_begin, nc = row->sv.cn_count;
        const int32_t* li = (const int32_t*)(all + P.lib_index_off);
        const int32_t* lp = (const int32_t*)(all + P.lib_pairs_off);
        const double* lt = (const double*)(all + P.ltail_off);
        for (int32_t e = 0; e < nl; ++e) {
            out.lib_index[bg.x + e] = li[l0 + e]; out.lib_pairs[bg.x + e] = lp[l0 + e];
            if (out.ltail) out.ltail[bg.x + e] = lt[l0 + e];   // (the terms' log tails: for Fisher's combination on the host only)
        }
        const int32_t* ck = (const int32_t*)(all + P.cn_key_off);
        const float* cv = (const float*)(all + P.cn_value_off);
        for (int32_t e = 0; e < nc; ++e) { out.cn_key[bg.y + e] = ck[c0 + e]; out.cn_value[bg.y + e] = cv[c0 + e]; }
    }
    const unsigned long long rp = (unsigned long long)(uintptr_t)row;
    // The rows cross PCIe as SvWire (48 bytes, bdx_k3.h) like a single context's: what is left out -- chromosomes, strand counts, list
    // offsets -- the host puts back from the region table and the counts (materialize, bdx_api.hip).  Wire word k of a row is word
    // kSrc[k] of the full row; the last one is put together from five of them.
    static_assert(offsetof(bdx_sv, pos) == 8 && offsetof(bdx_sv, flag) == 32 && offsetof(bdx_sv, size) == 36 && offsetof(bdx_sv, score) == 40 &&
                      offsetof(bdx_sv, num_reads) == 44 && offsetof(bdx_sv, printed) == 48 && offsetof(bdx_sv, region) == 52 && offsetof(bdx_sv, lib_count) == 64 &&
                      offsetof(bdx_sv, cn_count) == 72 && offsetof(bdx_sv, allele_frequency) == 76 && offsetof(bdx_sv, logp) == 80 && offsetof(SvOut, grp_mask) == 88 &&
                      offsetof(SvOut, start) == 92 && kWireWords == 12, "SvWire from SvOut, word by word");
    uint32_t* dst = (uint32_t*)out.sv_out + (size_t)pos0 * kWireWords;
    // (every lane takes part in every shuffle: a lane that has left the loop would be read as zero)
    const uint64_t actmask = __ballot(act);
    (void)kLibBegin; (void)kCnBegin; (void)bg;
    for (uint32_t w0 = 0; w0 < cnt * kWireWords; w0 += 64) {
        const uint32_t w = w0 + lane;
        const bool in = w < cnt * kWireWords;
        const int rr = in ? (int)(w / kWireWords) : 0, wd = in ? (int)(w - (uint32_t)rr * kWireWords) : 0;
        const uint32_t plo = (uint32_t)__shfl((int)(uint32_t)rp, rr), phi = (uint32_t)__shfl((int)(uint32_t)(rp >> 32), rr);
        if (!in || !((actmask >> rr) & 1ull)) continue;
        const uint32_t* sp = (const uint32_t*)(uintptr_t)(((unsigned long long)phi << 32) | plo);
        // pos[0] pos[1] region[0] region[1] size score num_reads allele_frequency logp(lo) logp(hi) start | bits
        const int src = wd == 0 ? 2 : wd == 1 ? 3 : wd == 2 ? 13 : wd == 3 ? 14 : wd == 4 ? 9 : wd == 5 ? 10 : wd == 6 ? 11 : wd == 7 ? 19 : wd == 8 ? 20 : wd == 9 ? 21 : 23;
        uint32_t v;
        if (wd < 11) v = sp[src];
        else v = (sp[16] & 255u) | ((sp[18] & 255u) << 8) | ((sp[8] & 15u) << 16) | ((sp[22] & 7u) << 20) | ((sp[12] ? 1u : 0u) << 23);
        dst[w] = v;
    }
}

void launch_k9_merge_tables(const char* all, const TableDesc* D, int world, uint32_t n_total, uint32_t max_n, uint32_t* src, uint2* begins, uint32_t* ws,
                            const uint32_t* n_dev, const MergeOut& out, hipStream_t s) {
    if (!n_total) return;
    hipLaunchKernelGGL(k9_merge_rank_kernel, dim3((max_n + 255) / 256, (uint32_t)world), dim3(256), 0, s, all, D, src, begins);
    U4* wsu = (U4*)ws;
    scan_launch<U4>(BeginsIn{begins}, BeginsOut{begins}, n_dev, n_total, wsu + 1, wsu, s);
    hipLaunchKernelGGL(k9_merge_place_kernel, dim3((n_total + 255) / 256), dim3(256), 0, s, all, D, n_total, src, begins, out);
}

// ---- rank 0: gathered pair groups -> buckets by later region ------------------------------------------------------------
__device__ __forceinline__ GroupRec seg_group(const unsigned long long* base, const SegList& sg, uint32_t i) {
    const int q = sg.seg_of(i);
    return ((const GroupRec*)(base + sg.off[q]))[i - sg.start[q]];
}
__global__ __launch_bounds__(256) void k9_group_count_kernel(const unsigned long long* __restrict__ base, SegList sg, uint32_t n, uint32_t nr, uint32_t* cnt, uint32_t* err) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t hi = (uint32_t)((seg_group(base, sg, i).key >> 12) & ((1u << 26) - 1));
    if (hi < nr) atomicAdd(&cnt[hi], 1u); else *err = 1u;
}
struct BucketIn {
    const uint32_t* cnt;
    __device__ uint32_t operator()(uint32_t j, uint32_t) const { return cnt[j]; }
};
struct BucketOut {
    uint32_t* goff;
    __device__ void operator()(uint32_t j, uint32_t, uint32_t inc, uint32_t e) const { goff[j] = inc - e; }
};
__global__ __launch_bounds__(256) void k9_group_scatter_kernel(const unsigned long long* __restrict__ base, SegList sg, uint32_t n, uint32_t nr, const uint32_t* goff, uint32_t* cur,
                                                               GroupRec* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const GroupRec g = seg_group(base, sg, i);
    const uint32_t hi = (uint32_t)((g.key >> 12) & ((1u << 26) - 1));
    if (hi < nr) out[goff[hi] + atomicAdd(&cur[hi], 1u)] = g;
}
void launch_k9_bucket_groups(const unsigned long long* base, const SegList& sg, uint32_t n, uint32_t nr, uint32_t* cnt, uint32_t* goff, uint32_t* cur, GroupRec* out,
                             uint32_t* scan_ws, uint32_t* n_words, uint32_t* err, hipStream_t s) {
    UploadList ul{};
    ul.fill(cnt, 0u, (size_t)nr + 1);
    ul.fill(cur, 0u, (size_t)nr + 1);
    ul.fill(n_words, nr + 1, 1);
    ul.fill(err, 0u, 1);
    launch_k9_upload(ul, s);
    if (n) hipLaunchKernelGGL(k9_group_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, base, sg, n, nr, cnt, err);
    scan_launch<uint32_t>(BucketIn{cnt}, BucketOut{goff}, n_words, nr + 1, scan_ws + 2, scan_ws, s);
    if (n) hipLaunchKernelGGL(k9_group_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, base, sg, n, nr, goff, cur, out);
}

// The proper-read samples (2 x nkeys words per region) of the regions the HOST's share of rank 0's walk touches -- the two regions of every
// pair group it was handed -- copied from HBM into the full-size pinned table at their own places.  The rest of that table never crosses the
// link: at a whole genome it is 40 MB of the 76 MB the region table would take (the host walks dozens of groups, the samples serve nothing else).
__global__ __launch_bounds__(256) void k9_pk_rows_kernel(const GroupRec* __restrict__ groups, uint32_t ng, const uint32_t* __restrict__ pk_dev, uint32_t* pk_host,
                                                         uint32_t row_words, uint32_t nr) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * ng) return;
    const uint64_t k = groups[i >> 1].key;
    const uint32_t r = (i & 1u) ? (uint32_t)((k >> 12) & ((1u << 26) - 1)) : (uint32_t)(k >> 38);
    if (r >= nr) return;
    for (uint32_t w = 0; w < row_words; ++w) pk_host[(size_t)r * row_words + w] = pk_dev[(size_t)r * row_words + w];
}
void launch_k9_pk_rows(const GroupRec* groups, uint32_t ng, const uint32_t* pk_dev, uint32_t* pk_host, uint32_t row_words, uint32_t nr, hipStream_t s) {
    if (!ng || !row_words) return;
    hipLaunchKernelGGL(k9_pk_rows_kernel, dim3((2 * ng + 255) / 256), dim3(256), 0, s, groups, ng, pk_dev, pk_host, row_words, nr);
}

}  // namespace bdx

// (bdx_warm_up: the HIP runtime loads a translation unit's device code at the first launch of any of its kernels)
__global__ void k9_noop_kernel() {}
namespace bdx { void warm_k9(hipStream_t s) { hipLaunchKernelGGL(k9_noop_kernel, dim3(1), dim3(64), 0, s); } }
