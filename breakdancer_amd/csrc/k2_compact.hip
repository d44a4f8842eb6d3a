// K2 -- stream compaction of the reads that enter the region accumulator.
//
// Replaces the bookkeeping half of BreakDancer::push_read (breakdancer/BreakDancer.cpp:172-175, 202-212,
// 233-241) and ReadRegionData::incr_normal_read_count (ReadRegionData.hpp:158-162): instead of bumping
// string-keyed maps once per read, the running counts the reference samples at region boundaries
// (normal-read pairs, per-key proper reads) become prefix sums that are *sampled* at the anomalous reads.
//
// Input: class bytes from K1 + exclusive per-tile prefixes.  Output: one compact record per anomalous
// read, in stream order.  One wave per 256-read tile, in-tile scans are wave shuffles on 16-bit packed
// counters, no workgroup barrier; tiles without an anomalous read are skipped after one ballot; the anomalous
// slots of a tile are compacted through a wave-private LDS slice so that the gather runs with dense lanes.
// HBM traffic: 1-2 B per read plus a gather of ~35 B per anomalous read.
#include <cstdlib>

#include "bdx_dev.h"

namespace bdx {

size_t k2_lds_bytes(int) { return 0; }

__global__ __launch_bounds__(kBlock) void k2_compact_kernel(const K2Params p) {
    __shared__ uint32_t s_src[kWaves * kTile];  // per wave: offset in tile | class byte << 8, by in-tile rank
    __shared__ uint32_t s_nn[kWaves * kTile];
    const int nkeys = p.nkeys;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * kWaves;
#pragma unroll
    for (int f = 0; f < 4; ++f)
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < p.fill_words[f]; i += gridDim.x * kBlock) p.fill_ptr[f][i] = p.fill_value[f];
    for (uint32_t tile = blockIdx.x * kWaves + w; tile < p.ntiles; tile += nwaves) {
        const uint64_t base = (uint64_t)tile * kTile + (uint64_t)lane * 4;
        // the tile's prefix bases are fetched together with its class bytes (one round trip instead of two; the 7 % of
        // tiles without an anomalous read pay two wasted loads)
        const uint32_t pre_norm = p.tile_pre[(size_t)kColNormal * p.tstride + tile];
        const uint32_t rank0 = p.tile_pre[(size_t)kColAnom * p.tstride + tile];
        const uint32_t pre_k0 = p.tile_pre[(size_t)kColKey0 * p.tstride + tile];
        const uint32_t pre_k1 = nkeys > 1 ? p.tile_pre[(size_t)(kColKey0 + 1) * p.tstride + tile] : 0u;
        unsigned c[4] = {0, 0, 0, 0}, lib[4] = {0, 0, 0, 0};
        int nvalid = 0;
        if (base + 4 <= p.n) {
            nvalid = 4;
            const uchar4 q = *(const uchar4*)(p.cls + base);
            c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (base + r < p.n) { ++nvalid; c[r] = p.cls[base + r]; }
        }
        bool anom[4], nleft[4], pk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = r < nvalid;
            const unsigned f = c[r] & 15u;
            const bool pass = valid && (c[r] & 0x10u);
            const bool normal = f == F_NORMAL_FR || f == F_NORMAL_RF;
            anom[r] = pass && !normal;
            nleft[r] = pass && (c[r] & 0x40u);
            pk[r] = pass && (c[r] & 0x20u);
        }
        if (!__any(anom[0] || anom[1] || anom[2] || anom[3])) continue;  // wave-uniform

        uint32_t tot = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) tot += (anom[r] ? 1u : 0u) + (nleft[r] ? 0x10000u : 0u);
        const uint32_t ex0 = wave_incl_scan(tot) - tot;
        uint32_t nn = p.nn_base + pre_norm + (ex0 >> 16);
        uint32_t jj[4] = {0, 0, 0, 0};
        int key[4] = {0, 0, 0, 0};
        if (nkeys > 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nvalid) { lib[r] = p.r.lib[base + r]; key[r] = p.libs[lib[r]].key; }
        }
        // Wave-level compaction before the gather: every anomalous slot drops (offset in tile, class byte, nn) into the
        // wave's LDS slice at its in-tile rank; then lanes 0..cnt-1 each fetch ONE whole record, so the eight column
        // gathers are issued once per wave with all lanes busy and the compact stores are contiguous.
        const uint32_t cnt = __shfl((ex0 + tot) & 0xFFFFu, 63);
        {
            uint32_t local = ex0 & 0xFFFFu;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (anom[r]) {
                    jj[r] = rank0 + local;
                    s_src[w * kTile + local] = (uint32_t)(lane * 4 + r) | (c[r] << 8);
                    s_nn[w * kTile + local] = nn;
                    ++local;
                }
                if (nleft[r]) ++nn;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t b = 0; b < cnt; b += 64) {
            const uint32_t q = b + lane;
            if (q < cnt && rank0 + q < p.c.cap) {  // (the capacity can be a guess of an enqueue-ahead run)
                const uint32_t src = s_src[w * kTile + q];
                const uint64_t i = (uint64_t)tile * kTile + (src & 255u);
                const uint32_t j = rank0 + q;
                const unsigned sam = p.r.flag[i];
                p.c.tid[j] = p.r.tid[i];
                p.c.pos[j] = p.r.pos[i];
                p.c.isize[j] = abs(p.r.isize[i]);
                p.c.meta[j] = meta_pack((int)((src >> 8) & 15u), (sam >> 4) & 1u, (int)p.r.lib[i], (int)p.r.qlen[i]);
                p.c.key[j] = p.r.key[i];
                p.c.idx[j] = (uint32_t)i;
                p.c.nn[j] = s_nn[w * kTile + q];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // per-key proper-read prefix counts (inclusive of the read itself), two keys per packed scan
        for (int k0 = 0; k0 < nkeys; k0 += 2) {
            uint32_t v = 0, inc4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (pk[r]) v += (key[r] == k0 ? 1u : 0u) + (key[r] == k0 + 1 ? 0x10000u : 0u);
                inc4[r] = v;
            }
            const uint32_t ex = wave_incl_scan(v) - v;
            const uint32_t b0 = p.pk_base[k0] + (k0 == 0 ? pre_k0 : p.tile_pre[(size_t)(kColKey0 + k0) * p.tstride + tile]);
            const uint32_t b1 = k0 + 1 < nkeys ? p.pk_base[k0 + 1] + (k0 == 0 ? pre_k1 : p.tile_pre[(size_t)(kColKey0 + k0 + 1) * p.tstride + tile]) : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!anom[r] || jj[r] >= p.c.cap) continue;
                const uint32_t s = ex + inc4[r];
                p.c.pk[(size_t)k0 * p.c.cap + jj[r]] = b0 + (s & 0xFFFFu);
                if (k0 + 1 < nkeys) p.c.pk[(size_t)(k0 + 1) * p.c.cap + jj[r]] = b1 + (s >> 16);
            }
        }
    }
}

void launch_k2(const K2Params& p, size_t lds, hipStream_t s) {
    const uint32_t nblk = (p.ntiles + kWaves - 1) / kWaves;
    // measured on MI355X at 58.6 k tiles: 2048 workgroups 54 us, 4096 46 us, 8192 43 us, one tile per wave (14.6 k) 45 us -- the
    // kernel is bound by its scattered 32-byte sector gathers (8 columns per anomalous read), not by wave count
    static const uint32_t cap = getenv("BDX_K2_GRID") ? (uint32_t)atoi(getenv("BDX_K2_GRID")) : 8192u;
    const uint32_t grid = nblk < cap ? nblk : cap;
    hipLaunchKernelGGL(k2_compact_kernel, dim3(grid), dim3(kBlock), lds, s, p);
}

}  // namespace bdx
