// K7 -- the exchange steps of a chromosome-sharded run.
//
// Regions never span chromosomes (BreakDancer.cpp:216), so with the chromosomes spread over GPUs every read pair whose
// mates lie on one chromosome is joined where it is.  Only the reads classified ARP_CTX (tid != mtid,
// IlluminaPEReadClassifier.cpp:78-80) have their mate elsewhere.  The pair is observed when its SECOND mate shows up
// (SvBuilder.cpp:101-118), and the stream is position sorted, so the second mate is the one on the later chromosome: a CTX read
// whose mate's chromosome comes later and belongs to another rank sends {name key, second hash, genome-wide region id} there --
// ONE all-to-all over RCCL -- and joins that rank's own reads as a "foreign" entry (Entries::n_local, k4_join.hip); the rank of
// the later chromosome then has everything its pair groups need where its reads are.  CTX pairs between two chromosomes of one
// rank never leave it.
//
// The reference joins on the read NAME alone (ReadRegionData.cpp:108-113), whatever the records' mtid fields say, and appends every
// sighting.  Routing by chromosome is only right for names that behave: one sighting, two on one chromosome, or two CTX reads that
// name each other's chromosome.  The name census checks exactly that: the name key of EVERY anomalous read also travels to
// owner(key) -- 16 bytes {key, tid, mtid, not CTX} -- where a table counts sightings; any other name makes the run replay read by
// read on rank 0 (bdx_dist_impl.h), which is the reference's behaviour for any input.
#include "bdx_k3.h"
#include "bdx_shard.h"

namespace bdx {

// the rank a CTX read's join record travels to, or -1: its mate's chromosome comes later in the stream and belongs to another rank
__device__ __forceinline__ int ctx_destination(const ExchangeSrc& x, uint32_t j, uint32_t m) {
    if (meta_flag(m) != F_CTX) return -1;
    const int32_t t = x.tid[j], mt = x.mtid_col[x.idx[j]];
    if (mt <= t || mt >= x.ntids) return -1;
    const int32_t o = x.owner_of_tid[mt];
    return (o < 0 || o == x.me) ? -1 : o;
}

// cnt[0 .. world): CTX records per destination; cnt[world .. 2 world): census records per owner(key)
__global__ __launch_bounds__(256) void k7_count_kernel(ExchangeSrc x, uint32_t* cnt) {
    __shared__ uint32_t s_cnt[2 * kMaxRanks];
    for (uint32_t d = threadIdx.x; d < 2 * x.world; d += 256) s_cnt[d] = 0;
    __syncthreads();
    const uint32_t n = *x.n_ptr;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int d = ctx_destination(x, j, x.meta[j]);
        if (d >= 0) atomicAdd(&s_cnt[d], 1u);
        atomicAdd(&s_cnt[x.world + exchange_owner(x.key[j], x.world)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < 2 * x.world; d += 256)
        if (s_cnt[d]) atomicAdd(&cnt[d], s_cnt[d]);
}

// cursor[d] / cursor[world + d] start at the destination's offset in the respective send buffer (entries).
// A workgroup takes kScatterChunk consecutive reads: it counts them per destination in LDS, reserves its places with ONE atomic per
// destination (with one rank every census record has the same destination: an atomic per wave on that one word -- 23 k of them at
// 1.5 M records -- took a quarter of a millisecond), and hands the places out with LDS atomics.  The order within a destination is free.
// (with a second name hash in the stream the census counts (key, check) pairs: its word is a mix of the two, so two names whose keys
// collide are two names here as well -- the joins tell them apart by the check -- and an equal mix of different pairs only costs a replay)
constexpr uint32_t kScatterChunk = 4096;

__global__ __launch_bounds__(256) void k7_scatter_kernel(ExchangeSrc x, uint32_t* cursor, ExchangeEntry* out, unsigned long long* names_out) {
    __shared__ uint32_t s_cnt[2 * kMaxRanks], s_base[2 * kMaxRanks];
    const uint32_t n = *x.n_ptr;
    for (uint32_t c0 = blockIdx.x * kScatterChunk; c0 < n; c0 += gridDim.x * kScatterChunk) {
        for (uint32_t d = threadIdx.x; d < 2 * x.world; d += 256) s_cnt[d] = 0;
        __syncthreads();
        int dest[kScatterChunk / 256];
        uint32_t own[kScatterChunk / 256];
#pragma unroll
        for (uint32_t it = 0; it < kScatterChunk / 256; ++it) {
            const uint32_t j = c0 + it * 256 + threadIdx.x;
            dest[it] = -1; own[it] = 0;
            if (j < n) {
                dest[it] = ctx_destination(x, j, x.meta[j]);
                own[it] = exchange_owner(x.key[j], x.world);
                if (dest[it] >= 0) atomicAdd(&s_cnt[dest[it]], 1u);
                atomicAdd(&s_cnt[x.world + own[it]], 1u);
            }
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < 2 * x.world; d += 256) {
            const uint32_t c = s_cnt[d];
            s_base[d] = c ? atomicAdd(&cursor[d], c) : 0u;
            s_cnt[d] = 0;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t it = 0; it < kScatterChunk / 256; ++it) {
            const uint32_t j = c0 + it * 256 + threadIdx.x;
            if (j >= n) continue;
            const uint32_t m = x.meta[j];
            const uint64_t k = x.key[j];
            const uint64_t c = x.check ? x.check[j] : 0ull;
            if (dest[it] >= 0) {
                const uint32_t slot = s_base[dest[it]] + atomicAdd(&s_cnt[dest[it]], 1u);
                ExchangeEntry e;
                e.key = k; e.order = 0; e.region = x.region_of[j]; e.meta = m; e.isize = 0; e.check = c;
                out[slot] = e;
                // (the record's region is an end of a group another rank will form: whatever component it belongs to spans ranks -- it is walked
                // where all of it comes together, on rank 0.  Said here, by the sender: the receiver's verdict on ITS end needs no exchange either)
                if (x.taint && e.region >= 0) x.taint[e.region] = 1;
            }
            const uint32_t nslot = s_base[x.world + own[it]] + atomicAdd(&s_cnt[x.world + own[it]], 1u);
            unsigned long long w = k;
            if (x.check) w = (k * 0x9E3779B97F4A7C15ull) ^ ((c << 31) | (c >> 33)) ^ (c * 0xC2B2AE3D27D4EB4Full);
            if (w == ~0ull) w = 0;   // (all ones marks an empty slot of the census table)
            const bool is_ctx = meta_flag(m) == F_CTX;
            // (the mate's chromosome matters for an inter-chromosomal sighting only: the column is not touched for the others)
            const uint32_t t1 = (uint32_t)(x.tid[j] + 1) & 0xFFFFFFu, mt1 = is_ctx ? (uint32_t)(x.mtid_col[x.idx[j]] + 1) & 0xFFFFFFu : 0u;
            names_out[2 * (size_t)nslot] = w;
            names_out[2 * (size_t)nslot + 1] = (is_ctx ? 0ull : 1ull) | ((unsigned long long)t1 << 8) | ((unsigned long long)mt1 << 32);
        }
        __syncthreads();
    }
}

// the foreign entries of the join, and its entry count (the context's own anomalous reads + these)
__global__ __launch_bounds__(256) void k7_unpack_kernel(const ExchangeEntry* in, uint32_t n, uint64_t* key, uint64_t* check, int32_t* region,
                                                        const uint32_t* n_local, uint32_t* n_total) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) *n_total = *n_local + n;
    if (j >= n) return;
    const ExchangeEntry e = in[j];
    key[j] = e.key; region[j] = e.region;
    if (check) check[j] = e.check;
}

__global__ __launch_bounds__(256) void k7_unpack_seg_kernel(const unsigned long long* base, SegList sg, uint32_t n, uint64_t* key, uint64_t* check, int32_t* region,
                                                            const uint32_t* n_local, uint32_t* n_total) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) *n_total = *n_local + n;
    if (j >= n) return;
    const int q = sg.seg_of(j);
    const ExchangeEntry e = ((const ExchangeEntry*)(base + sg.off[q]))[j - sg.start[q]];
    key[j] = e.key; region[j] = e.region;
    if (check) check[j] = e.check;
}
void launch_k7_unpack_seg(const unsigned long long* base, const SegList& sg, uint32_t n, uint64_t* key, uint64_t* check, int32_t* region, const uint32_t* n_local,
                          uint32_t* n_total, hipStream_t s) {
    hipLaunchKernelGGL(k7_unpack_seg_kernel, dim3(std::max(1u, (n + 255) / 256)), dim3(256), 0, s, base, sg, n, key, check, region, n_local, n_total);
}

// ---- name census ----
// One 16-byte slot per name: {key word (all ones = empty), info}.  info: bits 0-1 sightings (saturating at 3), bit 2 a sighting that is
// not a CTX read, bit 3 sightings on different chromosomes, bit 4 two CTX sightings that do not name each other's chromosome,
// bits 8-31 the first sighting's chromosome + 1, bits 32-55 its mate chromosome + 1.  One line is dirtied per insert (three arrays
// with one scattered atomic each wrote 7x the records' bytes: profiles/r03_pmc_all_kernels.txt).
__device__ __forceinline__ void names_insert(unsigned long long k, unsigned long long w, unsigned long long* slots, uint32_t nslots);
__global__ __launch_bounds__(256) void k7_names_insert_kernel(const unsigned long long* in, uint32_t n, unsigned long long* slots, uint32_t nslots) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    names_insert(in[2 * (size_t)j], in[2 * (size_t)j + 1], slots, nslots);
}
__global__ __launch_bounds__(256) void k7_names_insert_seg_kernel(const unsigned long long* base, SegList sg, uint32_t n, unsigned long long* slots, uint32_t nslots) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int q = sg.seg_of(j);
    const unsigned long long* in = base + sg.off[q] + 2 * (size_t)(j - sg.start[q]);
    names_insert(in[0], in[1], slots, nslots);
}
__device__ __forceinline__ void names_insert(unsigned long long k, unsigned long long w, unsigned long long* slots, uint32_t nslots) {
    const unsigned long long nonctx = w & 1ull, t1 = (w >> 8) & 0xFFFFFFull, mt1 = (w >> 32) & 0xFFFFFFull;
    uint32_t s = (uint32_t)(((k ^ (k >> 31)) * 0x9E3779B97F4A7C15ull) >> 32) % nslots;   // (not the owner's hash: the keys of one owner share that)
    for (;;) {
        const unsigned long long old = atomicCAS(&slots[2 * (size_t)s], ~0ull, k);
        if (old == ~0ull || old == k) break;
        s = s + 1 == nslots ? 0 : s + 1;   // (the table has 1.5 slots per record: at most two names in three slots)
    }
    unsigned long long* info = &slots[2 * (size_t)s + 1];
    unsigned long long seen = *(volatile unsigned long long*)info;
    for (;;) {
        unsigned long long next;
        const unsigned long long cnt = seen & 3ull;
        if (cnt == 0) {
            next = 1ull | (nonctx << 2) | (t1 << 8) | (mt1 << 32);
        } else {
            const unsigned long long ft = (seen >> 8) & 0xFFFFFFull, fm = (seen >> 32) & 0xFFFFFFull;
            next = (seen & ~3ull) | (cnt < 3 ? cnt + 1 : 3ull) | (nonctx << 2);
            if (ft != t1) next |= 1ull << 3;
            if (!(ft == mt1 && fm == t1)) next |= 1ull << 4;
        }
        const unsigned long long got = atomicCAS(info, seen, next);
        if (got == seen) break;
        seen = got;
    }
}

__global__ __launch_bounds__(256) void k7_names_verdict_kernel(const unsigned long long* slots, uint32_t nslots, uint32_t* irregular) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots || slots[2 * (size_t)s] == ~0ull) return;
    const unsigned long long v = slots[2 * (size_t)s + 1];
    const uint32_t count = (uint32_t)(v & 3ull);
    const bool nonctx = (v >> 2) & 1ull, spread = (v >> 3) & 1ull, strangers = (v >> 4) & 1ull;
    // regular: one sighting; two on one chromosome; two CTX reads on two chromosomes that name each other's
    if (count > 2 || (count == 2 && spread && (nonctx || strangers))) *irregular = 1;
}

void launch_k7_count(const ExchangeSrc& x, uint32_t n_upper, uint32_t* cnt, hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + 255) / 256, 2048u);
    hipLaunchKernelGGL(k7_count_kernel, dim3(g), dim3(256), 0, s, x, cnt);
}

void launch_k7_scatter(const ExchangeSrc& x, uint32_t n_upper, uint32_t* cursor, ExchangeEntry* out, unsigned long long* names_out, hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + kScatterChunk - 1) / kScatterChunk, 4096u);
    hipLaunchKernelGGL(k7_scatter_kernel, dim3(g), dim3(256), 0, s, x, cursor, out, names_out);
}

void launch_k7_unpack(const ExchangeEntry* in, uint32_t n, uint64_t* key, uint64_t* check, int32_t* region, const uint32_t* n_local, uint32_t* n_total,
                      hipStream_t s) {
    hipLaunchKernelGGL(k7_unpack_kernel, dim3(std::max(1u, (n + 255) / 256)), dim3(256), 0, s, in, n, key, check, region, n_local, n_total);
}

// slots: [2 nslots] words, key words all ones and info words zero on entry
void launch_k7_names_census(const unsigned long long* in, uint32_t n, unsigned long long* slots, uint32_t nslots, uint32_t* irregular, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k7_names_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, n, slots, nslots);
    hipLaunchKernelGGL(k7_names_verdict_kernel, dim3((nslots + 255) / 256), dim3(256), 0, s, slots, nslots, irregular);
}
void launch_k7_names_census_seg(const unsigned long long* base, const SegList& sg, uint32_t n, unsigned long long* slots, uint32_t nslots, uint32_t* irregular, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k7_names_insert_seg_kernel, dim3((n + 255) / 256), dim3(256), 0, s, base, sg, n, slots, nslots);
    hipLaunchKernelGGL(k7_names_verdict_kernel, dim3((nslots + 255) / 256), dim3(256), 0, s, slots, nslots, irregular);
}
uint32_t k7_names_slots(size_t records) { return (uint32_t)std::max<size_t>(1024, records + records / 2); }

// key words all ones, info words zero
__global__ __launch_bounds__(256) void k7_names_clear_kernel(unsigned long long* slots, uint32_t nslots, uint32_t* irregular) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *irregular = 0;
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < nslots; s += gridDim.x * 256) {
        slots[2 * (size_t)s] = ~0ull;
        slots[2 * (size_t)s + 1] = 0ull;
    }
}
void launch_k7_names_clear(unsigned long long* slots, uint32_t nslots, uint32_t* irregular, hipStream_t s) {
    hipLaunchKernelGGL(k7_names_clear_kernel, dim3(std::min<uint32_t>((nslots + 255) / 256, 4096u)), dim3(256), 0, s, slots, nslots, irregular);
}

}  // namespace bdx

// ---- rank 0 of a sharded run: the gathered packages of region records -> ONE region table (the packages arrive in HBM) ----
#include "bdx_scan.h"

namespace bdx {

// Every region record of a package goes to rbase[tid] + its rank among the package's records of that chromosome.  A package
// holds its rank's chromosomes in ascending order, each chromosome's regions in order, so that rank is the distance to the
// first record with the same tid (binary search over the records before it).
__global__ __launch_bounds__(256) void k8_place_regions_kernel(const char* all, GatherDesc D, const uint64_t* rbase, int ntids, int nkeys2,
                                                               RegionRec* r_rec, uint32_t* r_pk, uint32_t* err) {
    const GatherPackage P = D.p[blockIdx.y];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.nr) return;
    const RegionRec* rr = (const RegionRec*)(all + P.regions_off);
    const RegionRec me = rr[i];
    const int t = me.tid;
    if (t < 0 || t >= ntids) { *err = 1; return; }
    uint32_t lo = 0, hi = i;   // first record with tid >= t (record i has tid t)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rr[mid].tid < t) lo = mid + 1; else hi = mid;
    }
    const uint64_t g = rbase[t] + (i - lo);
    if (g >= rbase[t + 1]) { *err = 1; return; }
    r_rec[g] = me;
    const uint32_t* src = (const uint32_t*)(all + P.pk_off) + (size_t)i * nkeys2;
    for (int k = 0; k < nkeys2; ++k) r_pk[g * nkeys2 + k] = src[k];
}

void launch_k8_place_regions(const char* all, const GatherDesc& D, uint32_t max_nr, const uint64_t* rbase, int ntids, int nkeys2, RegionRec* r_rec,
                             uint32_t* r_pk, uint32_t* err, hipStream_t s) {
    if (!max_nr) return;
    hipLaunchKernelGGL(k8_place_regions_kernel, dim3((max_nr + 255) / 256, D.world), dim3(256), 0, s, all, D, rbase, ntids, nkeys2, r_rec, r_pk, err);
}

}  // namespace bdx

// (bdx_warm_up: the HIP runtime loads a translation unit's device code at the first launch of any of its kernels)
__global__ void k7_noop_kernel() {}
namespace bdx { void warm_k7(hipStream_t s) { hipLaunchKernelGGL(k7_noop_kernel, dim3(1), dim3(64), 0, s); } }
