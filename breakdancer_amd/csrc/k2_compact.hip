// K2 -- stream compaction of the reads that enter the region accumulator.
//
// Replaces the bookkeeping half of BreakDancer::push_read (breakdancer/BreakDancer.cpp:172-175, 202-212,
// 233-241) and ReadRegionData::incr_normal_read_count (ReadRegionData.hpp:158-162): instead of bumping
// string-keyed maps once per read, the running counts the reference samples at region boundaries
// (normal-read pairs, per-key proper reads) become prefix sums that are *sampled* at the anomalous reads.
//
// Input: class bytes from K1 + exclusive per-tile prefixes.  Output: one compact record per anomalous
// read, in stream order.  HBM traffic: 1-2 B per read plus a gather of ~35 B per anomalous read.
#include "bdx_dev.h"

namespace bdx {

size_t k2_lds_bytes(int nkeys) { return (size_t)(1 + (nkeys + 1) / 2) * kBlock * 4; }

__global__ __launch_bounds__(kBlock) void k2_compact_kernel(const K2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s_tt = (uint32_t*)smem;  // [words][256] packed 16-bit counters
    const int nkeys = p.nkeys;
    const int words = 1 + (nkeys + 1) / 2;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    for (uint32_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const uint64_t base = (uint64_t)tile * kTile + (uint64_t)t * 4;
        unsigned c[4] = {0, 0, 0, 0}, lib[4] = {0, 0, 0, 0};
        int nvalid = 0;
        if (base + 4 <= p.n) {
            nvalid = 4;
            const uchar4 q = *(const uchar4*)(p.cls + base);
            c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
            if (nkeys > 1) {
                const uchar4 l = *(const uchar4*)(p.r.lib + base);
                lib[0] = l.x; lib[1] = l.y; lib[2] = l.z; lib[3] = l.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (base + r < p.n) { ++nvalid; c[r] = p.cls[base + r]; lib[r] = nkeys > 1 ? p.r.lib[base + r] : 0; }
        }
        bool anom[4], nleft[4], pk[4];
        int key[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = r < nvalid;
            const unsigned f = c[r] & 15u;
            const bool pass = valid && (c[r] & 0x10u);
            const bool normal = f == F_NORMAL_FR || f == F_NORMAL_RF;
            anom[r] = pass && !normal;
            nleft[r] = pass && (c[r] & 0x40u);
            pk[r] = pass && (c[r] & 0x20u);
            key[r] = nkeys > 1 ? p.libs[lib[r]].key : 0;
        }
        // packed per-thread totals -> LDS
        {
            uint32_t tot = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) tot += (anom[r] ? 1u : 0u) + (nleft[r] ? 0x10000u : 0u);
            s_tt[t] = tot;
            for (int wd = 1; wd < words; ++wd) {
                const int k0 = (wd - 1) * 2;
                uint32_t v = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (pk[r]) v += (key[r] == k0 ? 1u : 0u) + (key[r] == k0 + 1 ? 0x10000u : 0u);
                s_tt[wd * kBlock + t] = v;
            }
        }
        __syncthreads();
        // each wave turns whole rows into exclusive prefixes (256 entries = 4 per lane)
        for (int wd = w; wd < words; wd += kWaves) {
            uint4 v = *(uint4*)(s_tt + wd * kBlock + lane * 4);
            const uint32_t s4 = v.x + v.y + v.z + v.w;
            const uint32_t ex = wave_incl_scan(s4) - s4;
            *(uint4*)(s_tt + wd * kBlock + lane * 4) = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        }
        __syncthreads();
        const bool any = anom[0] || anom[1] || anom[2] || anom[3];
        if (any) {
            const uint32_t ex0 = s_tt[t];
            uint32_t rank = p.tile_pre[(size_t)kColAnom * p.tstride + tile] + (ex0 & 0xFFFFu);
            uint32_t nn = p.tile_pre[(size_t)kColNormal * p.tstride + tile] + (ex0 >> 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (anom[r]) {
                    const uint64_t i = base + r;
                    const uint32_t j = rank;
                    const unsigned sam = p.r.flag[i];
                    const int isz = p.r.isize[i];
                    const unsigned L = nkeys > 1 ? lib[r] : p.r.lib[i];
                    p.c.tid[j] = p.r.tid[i];
                    p.c.pos[j] = p.r.pos[i];
                    p.c.isize[j] = abs(isz);
                    p.c.meta[j] = meta_pack((int)(c[r] & 15u), (sam >> 4) & 1u, (int)L, (int)p.r.qlen[i]);
                    p.c.key[j] = p.r.key[i];
                    p.c.nn[j] = nn;
                    for (int k = 0; k < nkeys; ++k) {
                        const uint32_t pw = s_tt[(1 + k / 2) * kBlock + t];
                        uint32_t v = p.tile_pre[(size_t)(kColKey0 + k) * p.tstride + tile] + ((pw >> (16 * (k & 1))) & 0xFFFFu);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            if (rr <= r && pk[rr] && key[rr] == k) ++v;
                        p.c.pk[(size_t)k * p.c.cap + j] = v;
                    }
                    ++rank;
                }
                if (nleft[r]) ++nn;
            }
        }
        __syncthreads();
    }
}

void launch_k2(const K2Params& p, size_t lds, hipStream_t s) {
    const uint32_t grid = p.ntiles < 2048u ? p.ntiles : 2048u;
    hipLaunchKernelGGL(k2_compact_kernel, dim3(grid), dim3(kBlock), lds, s, p);
}

}  // namespace bdx
