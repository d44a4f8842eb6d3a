// K2 -- stream compaction of the reads that enter the region accumulator.
//
// Replaces the bookkeeping half of BreakDancer::push_read (breakdancer/BreakDancer.cpp:172-175, 202-212,
// 233-241) and ReadRegionData::incr_normal_read_count (ReadRegionData.hpp:158-162): instead of bumping
// string-keyed maps once per read, the running counts the reference samples at region boundaries
// (normal-read pairs, per-key proper reads) become prefix sums that are *sampled* at the anomalous reads.
//
// Input: K1's ready-made records of the anomalous reads (up to kStashCap per tile) + exclusive per-tile prefixes; for tiles with
// more, and for runs with more than kStashKeys counter keys, the class bytes and the columns.  Output: one compact record per
// anomalous read, in stream order.  One wave per FOUR of K1's tiles (1024 reads, 16 consecutive class bytes = one 16-byte load
// per lane): the kernel is bound by dependent round trips per wave (class bytes -> gather -> store), not by bytes, so
// fewer, fatter waves finish sooner.  Per-lane state is three 16-bit masks; in-wave scans are shuffles on 16-bit
// packed counters, no workgroup barrier; super tiles without an anomalous read are skipped after one ballot; the
// anomalous slots are compacted through a wave-private LDS slice (256 at a time) so that the gather runs with dense
// lanes.  Measured: one wave per 256 reads 41 us, per 1024 reads 26 us, per 2048 reads 50 us (15 M reads).
// HBM traffic: 1-2 B per read plus a gather of ~35 B per anomalous read.
#include <cstdlib>

#include "bdx_dev.h"
#include "bdx_finalize.h"

namespace bdx {

size_t k2_lds_bytes(int) { return 0; }

constexpr int kSub = kK2TilesPerWave;   // K1 tiles per wave (finalize_kernel writes the prefixes for these boundaries)
constexpr int kPerLane = 4 * kSub;      // consecutive reads per lane (<= 32: the per-lane state is 32-bit masks)
constexpr int kTile2 = kTile * kSub;    // reads per wave
constexpr int kSlice = 256;             // anomalous reads compacted through LDS at a time (more in one super tile: several rounds)
constexpr int kOffBits = 10;            // bits of a read's offset in its super tile
static_assert(kPerLane <= 32 && kTile2 <= (1 << kOffBits), "K2 super tile");

// class byte r of the lane's 4 * kSub packed bytes (selects, no run-time array index)
__device__ __forceinline__ unsigned class_byte(const uint64_t (&cq)[kSub / 2], int r) {
    uint64_t v = cq[0];
#pragma unroll
    for (int q = 1; q < kSub / 2; ++q) v = (r >> 3) == q ? cq[q] : v;
    return (unsigned)((v >> (8 * (r & 7))) & 255u);
}

// name key and length of read i: from the resident columns, or -- when bdx_push left them in the caller's pinned host
// memory -- from the segment (one per pushed batch) that holds them; only anomalous reads (about 1 %) get here
__device__ __forceinline__ void key_and_qlen_of(const K2Params& p, uint64_t i, uint64_t& key, int& qlen, uint64_t& check) {
    if (p.nseg == 0) {
        key = p.r.key[i];
        qlen = (int)p.r.qlen[i];
        if (p.c.check) check = p.r.check[i];
        return;
    }
    int lo = 0, hi = p.nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (p.seg_begin[mid] <= i) lo = mid; else hi = mid - 1;
    }
    key = p.seg_ptr[lo][i];
    qlen = (int)p.seg_qlen[lo][i];
    if (p.c.check) check = p.seg_check[lo][i];
}

// finalize_kernel scanned a tile-total column in chunks: the exclusive prefix at a super tile is its chunk-local prefix plus the
// totals of the chunks before its own -- lane g holds chunk g's total (every lane loads its word: the row has kMaxChunks entries)
__device__ __forceinline__ uint32_t wave_sum_below(uint32_t v, uint32_t chunk, int lane) {
    v = (uint32_t)lane < chunk ? v : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// a read's library index as K1 takes it: out of range, or with one library (the column is then not even filled), 0
__device__ __forceinline__ int lib_of(const K2Params& p, uint64_t i) {
    if (p.nlibs <= 1) return 0;
    const int l = p.r.lib[i];
    return l < p.nlibs ? l : 0;
}

// kBases: the second level of the finalisation ran before this kernel and left the sums (else it runs beside it)
template <bool kBases> __device__ __forceinline__ uint32_t chunk_word(const K2Params& p, int col, uint32_t chunk, int lane) {
    return kBases ? p.chunk_base[(size_t)col * kMaxChunks + chunk] : p.chunk_tot[(size_t)col * kMaxChunks + lane];
}
template <bool kBases> __device__ __forceinline__ uint32_t chunk_sum(uint32_t word, uint32_t chunk, int lane) {
    return kBases ? word : wave_sum_below(word, chunk, lane);
}
template <bool kBases> __device__ __forceinline__ uint32_t col_prefix(const K2Params& p, int col, uint32_t tile2, uint32_t chunk, int lane) {
    return p.tile_pre[(size_t)col * p.tstride + tile2] + chunk_sum<kBases>(chunk_word<kBases>(p, col, chunk, lane), chunk, lane);
}

// nblk: workgroups that compact (the launch can hold one more, see k2_compact_side_kernel)
template <bool kBases> __device__ __forceinline__ void k2_body(const K2Params& p, uint32_t nblk) {
    __shared__ uint32_t s_src[kWaves * kSlice];  // per wave: offset in super tile | class byte << kOffBits, by in-tile rank
    __shared__ uint32_t s_nn[kWaves * kSlice];
    const int nkeys = p.nkeys;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t nwaves = nblk * kWaves;
    const uint32_t ntiles2 = (p.ntiles + kSub - 1) / kSub;
    for (uint32_t tile2 = blockIdx.x * kWaves + w; tile2 < ntiles2; tile2 += nwaves) {
        const uint64_t base = (uint64_t)tile2 * kTile2 + (uint64_t)lane * kPerLane;
        // the prefix bases (chunk-local prefix + the totals of the chunks before) are fetched together with the class bytes
        // (one round trip instead of two)
        // (requested here, added up -- wave_sum_below -- only once the super tile turns out to hold anomalous reads)
        const uint32_t chunk = tile2 / p.chunk_super;
        const uint32_t l_norm = p.tile_pre[(size_t)kColNormal * p.tstride + tile2], c_norm = chunk_word<kBases>(p, kColNormal, chunk, lane);
        const uint32_t l_anom = p.tile_pre[(size_t)kColAnom * p.tstride + tile2], c_anom = chunk_word<kBases>(p, kColAnom, chunk, lane);
        const uint32_t l_k0 = p.tile_pre[(size_t)kColKey0 * p.tstride + tile2], c_k0 = chunk_word<kBases>(p, kColKey0, chunk, lane);
        const uint32_t l_k1 = nkeys > 1 ? p.tile_pre[(size_t)(kColKey0 + 1) * p.tstride + tile2] : 0u;
        const uint32_t c_k1 = nkeys > 1 ? chunk_word<kBases>(p, kColKey0 + 1, chunk, lane) : 0u;
        if (p.stash) {
            // K1 left the anomalous reads of a tile ready-made (up to kStashCap of them): their records are copied to their places,
            // shifted by the prefixes -- no class bytes, no column gather except name key and read length.  Lane c * 4 + t holds
            // column c's total of the super tile's tile t.
            static_assert(kSub == 4 && kStashKeys == 2, "lane layout of the totals");
            const int t = lane & 3, col = lane >> 2;
            uint32_t v = 0;
            if (col < 2 + nkeys && tile2 * kSub + t < p.ntiles) v = p.tile_tot[(size_t)col * p.tstride + tile2 * kSub + t];
            const uint32_t a0 = __shfl(v, 0), a1 = __shfl(v, 1), a2 = __shfl(v, 2), a3 = __shfl(v, 3);
            const uint32_t A = a0 + a1 + a2 + a3;
            if (A == 0) continue;  // (wave-uniform)
            if (max(max(a0, a1), max(a2, a3)) <= (uint32_t)kStashCap) {
                const uint32_t pre_norm = l_norm + chunk_sum<kBases>(c_norm, chunk, lane), rank0 = l_anom + chunk_sum<kBases>(c_anom, chunk, lane);
                const uint32_t pre_k0 = l_k0 + chunk_sum<kBases>(c_k0, chunk, lane), pre_k1 = l_k1 + chunk_sum<kBases>(c_k1, chunk, lane);
                const uint32_t q = (uint32_t)lane;  // A <= 64: one read per lane
                const uint32_t tq = (q >= a0 ? 1u : 0u) + (q >= a0 + a1 ? 1u : 0u) + (q >= a0 + a1 + a2 ? 1u : 0u);
                const uint32_t before = tq == 0 ? 0u : (tq == 1 ? a0 : (tq == 2 ? a0 + a1 : a0 + a1 + a2));
                // exclusive in-super-tile prefixes of the other columns at tile tq
                uint32_t e[3];
#pragma unroll
                for (int c = 1; c <= 3; ++c) {
                    const uint32_t x0 = __shfl(v, c * 4), x1 = __shfl(v, c * 4 + 1), x2 = __shfl(v, c * 4 + 2);
                    e[c - 1] = tq == 0 ? 0u : (tq == 1 ? x0 : (tq == 2 ? x0 + x1 : x0 + x1 + x2));
                }
                const uint32_t j = rank0 + q;
                const uint32_t tile = tile2 * kSub + min(tq, 3u);
                uint4 r0 = make_uint4(0u, 0u, 0u, 0u);
                uint2 r1 = make_uint2(0u, 0u);
                if (q < A) {
                    const StashRec* src = p.stash + (size_t)tile * kStashCap + (q - before);
                    r0 = *(const uint4*)src;
                    r1 = *(const uint2*)&src->where;
                }
                if (!__any(r1.x == 0xFFFFFFFFu)) {  // no mixed tile among them (K1 marks every slot of one; else: from the columns, below)
                    if (q < A && j < p.c.cap) {  // (the capacity can be a guess of an enqueue-ahead run)
                        const uint64_t i = (uint64_t)tile * kTile + (r1.x & 255u);
                        const uint32_t k0 = (r1.x >> 20) & 63u;
                        uint64_t key, check = 0;
                        int qlen;
                        key_and_qlen_of(p, i, key, qlen, check);
                        p.c.tid[j] = (int32_t)r0.x;
                        p.c.pos[j] = (int32_t)r0.y;
                        p.c.isize[j] = (int32_t)r0.z;
                        p.c.meta[j] = r0.w | ((uint32_t)qlen << 16);
                        p.c.key[j] = key;
                        if (p.c.check) p.c.check[j] = check;
                        p.c.idx[j] = (uint32_t)i;
                        p.c.nn[j] = p.nn_base + pre_norm + e[0] + ((r1.x >> 8) & 511u);
                        p.c.pk[j] = p.pk_base[0] + pre_k0 + e[1] + (k0 == 0 ? r1.y : 0u);
                        if (nkeys > 1) p.c.pk[(size_t)p.c.cap + j] = p.pk_base[1] + pre_k1 + e[2] + (k0 == 1 ? r1.y : 0u);
                    }
                    continue;
                }
            }
            // a tile with more anomalous reads than K1 keeps: the super tile is compacted from the columns (below)
        }
        uint64_t cq[kSub / 2];  // the lane's class bytes, 8 per word
#pragma unroll
        for (int q = 0; q < kSub / 2; ++q) cq[q] = 0;
        int nvalid = 0;
        if (base + kPerLane <= p.n) {
            nvalid = kPerLane;
#pragma unroll
            for (int q = 0; q < kSub / 4; ++q) {
                const uint4 x = *(const uint4*)(p.cls + base + 16 * q);
                cq[2 * q] = (uint64_t)x.x | ((uint64_t)x.y << 32);
                cq[2 * q + 1] = (uint64_t)x.z | ((uint64_t)x.w << 32);
            }
        } else {
#pragma unroll
            for (int r = 0; r < kPerLane; ++r)
                if (base + r < p.n) { ++nvalid; cq[r >> 3] |= (uint64_t)p.cls[base + r] << (8 * (r & 7)); }
        }
        uint32_t m_anom = 0, m_nleft = 0, m_pk = 0;
#pragma unroll
        for (int r = 0; r < kPerLane; ++r) {
            const unsigned c = (unsigned)((cq[r >> 3] >> (8 * (r & 7))) & 255u);
            const unsigned f = c & 15u;
            const bool pass = r < nvalid && (c & 0x10u);
            const bool normal = f == F_NORMAL_FR || f == F_NORMAL_RF;
            m_anom |= (pass && !normal) ? 1u << r : 0u;
            m_nleft |= (pass && (c & 0x40u)) ? 1u << r : 0u;
            m_pk |= (pass && (c & 0x20u)) ? 1u << r : 0u;
        }
        if (!__any(m_anom != 0)) continue;  // wave-uniform
        const uint32_t pre_norm = l_norm + chunk_sum<kBases>(c_norm, chunk, lane), rank0 = l_anom + chunk_sum<kBases>(c_anom, chunk, lane);
        const uint32_t pre_k0 = l_k0 + chunk_sum<kBases>(c_k0, chunk, lane), pre_k1 = l_k1 + chunk_sum<kBases>(c_k1, chunk, lane);

        const uint32_t tot = (uint32_t)__popc(m_anom) + ((uint32_t)__popc(m_nleft) << 16);
        const uint32_t ex0 = wave_incl_scan(tot) - tot;
        const uint32_t nn0 = p.nn_base + pre_norm + (ex0 >> 16);
        const uint32_t local0 = ex0 & 0xFFFFu;
        // Wave-level compaction before the gather: every anomalous slot drops (offset in tile, class byte, nn) into the
        // wave's LDS slice at its in-tile rank; then lanes 0..cnt-1 each fetch ONE whole record, so the column
        // gathers are issued with all lanes busy and the compact stores are contiguous.
        const uint32_t cnt = __shfl((ex0 + tot) & 0xFFFFu, 63);
        for (uint32_t win = 0; win < cnt; win += kSlice) {
            {
                uint32_t local = local0;
                for (uint32_t mm = m_anom; mm; mm &= mm - 1, ++local) {
                    if (local < win || local >= win + kSlice) continue;
                    const int r = __builtin_ctz(mm);
                    s_src[w * kSlice + (local - win)] = (uint32_t)(lane * kPerLane + r) | (class_byte(cq, r) << kOffBits);
                    s_nn[w * kSlice + (local - win)] = nn0 + (uint32_t)__popc(m_nleft & ((1u << r) - 1u));
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t wcnt = min((uint32_t)kSlice, cnt - win);
            for (uint32_t b = 0; b < wcnt; b += 64) {
                const uint32_t q = b + lane;
                if (q < wcnt && rank0 + win + q < p.c.cap) {  // (the capacity can be a guess of an enqueue-ahead run)
                    const uint32_t src = s_src[w * kSlice + q];
                    const uint64_t i = (uint64_t)tile2 * kTile2 + (src & ((1u << kOffBits) - 1u));
                    const uint32_t j = rank0 + win + q;
                    const unsigned sam = p.r.flag[i];
                    p.c.tid[j] = p.r.tid[i];
                    p.c.pos[j] = p.r.pos[i];
                    p.c.isize[j] = abs(p.r.isize[i]);
                    uint64_t key, check = 0;
                    int qlen;
                    key_and_qlen_of(p, i, key, qlen, check);
                    p.c.meta[j] = meta_pack((int)((src >> kOffBits) & 15u), (sam >> 4) & 1u, lib_of(p, i), qlen);
                    p.c.key[j] = key;
                    if (p.c.check) p.c.check[j] = check;
                    p.c.idx[j] = (uint32_t)i;
                    p.c.nn[j] = s_nn[w * kSlice + q];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // per-key proper-read prefix counts (inclusive of the read itself), two keys per packed scan
        uint64_t kq[kSub / 2];  // counter key of each read (a byte each; all 0 with one key)
#pragma unroll
        for (int q = 0; q < kSub / 2; ++q) kq[q] = 0;
        if (nkeys > 1) {
#pragma unroll
            for (int r = 0; r < kPerLane; ++r)
                if (r < nvalid) kq[r >> 3] |= (uint64_t)(p.libs[lib_of(p, base + r)].key & 255) << (8 * (r & 7));
        }
        for (int k0 = 0; k0 < nkeys; k0 += 2) {
            uint32_t mk0 = 0, mk1 = 0;
            if (nkeys > 1) {
#pragma unroll
                for (int r = 0; r < kPerLane; ++r) {
                    const int key = (int)((kq[r >> 3] >> (8 * (r & 7))) & 255u);
                    if ((m_pk >> r) & 1u) { mk0 |= key == k0 ? 1u << r : 0u; mk1 |= key == k0 + 1 ? 1u << r : 0u; }
                }
            } else {
                mk0 = m_pk;
            }
            const uint32_t v = (uint32_t)__popc(mk0) + ((uint32_t)__popc(mk1) << 16);
            const uint32_t ex = wave_incl_scan(v) - v;
            const uint32_t b0 = p.pk_base[k0] + (k0 == 0 ? pre_k0 : col_prefix<kBases>(p, kColKey0 + k0, tile2, chunk, lane));
            const uint32_t b1 = k0 + 1 < nkeys ? p.pk_base[k0 + 1] + (k0 == 0 ? pre_k1 : col_prefix<kBases>(p, kColKey0 + k0 + 1, tile2, chunk, lane)) : 0u;
            uint32_t j = rank0 + local0;
            for (uint32_t mm = m_anom; mm; mm &= mm - 1, ++j) {
                if (j >= p.c.cap) break;
                const int r = __builtin_ctz(mm);
                const uint32_t upto = (2u << r) - 1u;  // reads 0..r of this lane
                p.c.pk[(size_t)k0 * p.c.cap + j] = b0 + (ex & 0xFFFFu) + (uint32_t)__popc(mk0 & upto);
                if (k0 + 1 < nkeys) p.c.pk[(size_t)(k0 + 1) * p.c.cap + j] = b1 + (ex >> 16) + (uint32_t)__popc(mk1 & upto);
            }
        }
    }
}

__device__ __forceinline__ void k2_fills(const K2Params& p) {  // (every workgroup of the launch takes part)
#pragma unroll
    for (int f = 0; f < 4; ++f)
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < p.fill_words[f]; i += gridDim.x * kBlock) p.fill_ptr[f][i] = p.fill_value[f];
}

__global__ __launch_bounds__(kBlock) void k2_compact_kernel(const K2Params p) {
    k2_fills(p);
    k2_body<true>(p, gridDim.x);
}

// the same with one more workgroup, the last, that runs the second level of the pass-1 finalisation: when K2 is enqueued
// without waiting for the pass-1 record (enqueue-ahead) that one-workgroup kernel would only sit between two launches
__global__ __launch_bounds__(kBlock) void k2_compact_side_kernel(const K2Params p, const FinalizeParams fp) {
    k2_fills(p);
    if (blockIdx.x == gridDim.x - 1) {
        finalize2_body(fp);
        return;
    }
    k2_body<false>(p, gridDim.x - 1);
}

void launch_k2(const K2Params& p, size_t lds, hipStream_t s, const FinalizeParams* side) {
    const uint32_t ntiles2 = (p.ntiles + kSub - 1) / kSub;
    const uint32_t nblk = (ntiles2 + kWaves - 1) / kWaves;
    // measured on MI355X at 58.6 k tiles: 2048 workgroups 54 us, 4096 46 us, 8192 43 us, one tile per wave (14.6 k) 45 us -- the
    // kernel is bound by its scattered 32-byte sector gathers (8 columns per anomalous read), not by wave count
    const uint32_t cap = 8192u;
    const uint32_t grid = nblk < cap ? nblk : cap;
    if (side) hipLaunchKernelGGL(k2_compact_side_kernel, dim3(grid + 1), dim3(kBlock), lds, s, p, *side);
    else hipLaunchKernelGGL(k2_compact_kernel, dim3(grid), dim3(kBlock), lds, s, p);
}

}  // namespace bdx

// (bdx_warm_up: the HIP runtime loads a translation unit's device code at the first launch of any of its kernels)
__global__ void k2_noop_kernel() {}
namespace bdx { void warm_k2(hipStream_t s) { hipLaunchKernelGGL(k2_noop_kernel, dim3(1), dim3(64), 0, s); } }
