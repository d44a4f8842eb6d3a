// K7 -- the exchange step of a chromosome-sharded run: inter-chromosomal (CTX) mate records to the rank that joins them.
//
// Regions never span chromosomes (BreakDancer.cpp:216), so with the chromosomes spread over GPUs every read pair whose
// mates lie on one chromosome is joined where it is (K4 on the chromosome's own context).  Only the reads classified
// ARP_CTX (tid != mtid, IlluminaPEReadClassifier.cpp:78-80) have their mate elsewhere: their join records
// {name key, stream order, global region id, meta, |isize|} go to owner(name key), one all-to-all over RCCL, and meet
// there (ReadRegionData.cpp:108-113 joins on the read name only).  These kernels pack the records by destination rank
// on the sending side and unpack them into the join's SoA layout on the receiving side; everything stays in HBM.
#include "bdx_k3.h"

namespace bdx {

__global__ __launch_bounds__(256) void k7_count_kernel(const uint64_t* key, const uint32_t* meta, const uint32_t* n_ptr, uint32_t world,
                                                       uint32_t* cnt) {
    __shared__ uint32_t s_cnt[kMaxRanks];
    for (uint32_t d = threadIdx.x; d < world; d += 256) s_cnt[d] = 0;
    __syncthreads();
    const uint32_t n = *n_ptr;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256)
        if (meta_flag(meta[j]) == F_CTX) atomicAdd(&s_cnt[exchange_owner(key[j], world)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < world; d += 256)
        if (s_cnt[d]) atomicAdd(&cnt[d], s_cnt[d]);
}

// A place in destination d's part of the send buffer for every active lane: ONE atomic per wave and destination (lanes with the same
// destination share it) -- with one rank every record of a chromosome has the same destination, and 250 k atomics on one word took a millisecond.
__device__ __forceinline__ uint32_t wave_slots(uint32_t* cursor, uint32_t dest, bool active) {
    const int lane = threadIdx.x & 63;
    uint32_t slot = 0;
    uint64_t todo = __ballot(active);
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t d = (uint32_t)__shfl((int)dest, leader);
        const uint64_t same = __ballot(active && dest == d);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&cursor[d], (uint32_t)__builtin_popcountll(same));
        base = (uint32_t)__shfl((int)base, leader);
        if (active && dest == d) slot = base + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

// cursor[d] starts at the destination's offset in the send buffer (entries)
__global__ __launch_bounds__(256) void k7_scatter_kernel(const uint64_t* key, const uint64_t* check, const int32_t* region_of, const uint32_t* meta,
                                                         const int32_t* isize, const uint32_t* n_ptr, uint32_t world, uint32_t order_base,
                                                         int32_t region_base, uint32_t* cursor, ExchangeEntry* out) {
    const uint32_t n = *n_ptr;
    for (uint32_t j0 = blockIdx.x * 256; j0 < n; j0 += gridDim.x * 256) {   // (whole waves stay in the loop: the slots are handed out wave-wide)
        const uint32_t j = j0 + threadIdx.x;
        const uint32_t m = j < n ? meta[j] : 0u;
        const bool ctx = j < n && meta_flag(m) == F_CTX;
        const uint64_t k = ctx ? key[j] : 0ull;
        const uint32_t slot = wave_slots(cursor, ctx ? exchange_owner(k, world) : 0u, ctx);
        if (!ctx) continue;
        const int32_t r = region_of[j];
        ExchangeEntry e;
        e.key = k; e.order = order_base + j; e.region = r < 0 ? -1 : r + region_base; e.meta = m; e.isize = isize[j];
        e.check = check ? check[j] : 0ull;
        out[slot] = e;
    }
}

__global__ __launch_bounds__(256) void k7_unpack_kernel(const ExchangeEntry* in, uint32_t n, uint64_t* key, uint64_t* check, uint32_t* order,
                                                        int32_t* region, uint32_t* meta, int32_t* isize) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const ExchangeEntry e = in[j];
    key[j] = e.key; order[j] = e.order; region[j] = e.region; meta[j] = e.meta; isize[j] = e.isize;
    if (check) check[j] = e.check;
}

// ---- name census: is any read name met more than twice, or twice on two chromosomes without being an inter-chromosomal pair? ----
// Every rank's joins see only its own chromosomes (and the CTX records it owns), but the reference keys its name map on the
// whole genome (ReadRegionData.cpp:108-113): merged files with clashing read names put sightings of one name on several
// chromosomes.  So the name key of EVERY anomalous read travels to owner(key) as well -- 16 bytes {key, tid << 1 | not CTX},
// no join, only a census: a table of keys with a count, the first chromosome seen and whether another one followed.  A name is
// regular if it has one sighting, two on one chromosome, or two CTX reads on two chromosomes; anything else makes the run
// replay read by read on rank 0 (bdx_dist_impl.h).
__global__ __launch_bounds__(256) void k7_names_count_kernel(const uint64_t* key, const uint32_t* n_ptr, uint32_t world, uint32_t* cnt) {
    __shared__ uint32_t s_cnt[kMaxRanks];
    for (uint32_t d = threadIdx.x; d < world; d += 256) s_cnt[d] = 0;
    __syncthreads();
    const uint32_t n = *n_ptr;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) atomicAdd(&s_cnt[exchange_owner(key[j], world)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < world; d += 256)
        if (s_cnt[d]) atomicAdd(&cnt[d], s_cnt[d]);
}

// (with a second name hash in the stream the census counts (key, check) pairs: its word is a mix of the two, so two names whose keys
// collide are two names here as well -- the joins tell them apart by the check -- and an equal mix of different pairs only costs a replay)
__global__ __launch_bounds__(256) void k7_names_scatter_kernel(const uint64_t* key, const uint64_t* check, const uint32_t* meta, const uint32_t* n_ptr,
                                                               uint32_t world, uint32_t tid, uint32_t* cursor, unsigned long long* out) {
    const uint32_t n = *n_ptr;
    for (uint32_t j0 = blockIdx.x * 256; j0 < n; j0 += gridDim.x * 256) {
        const uint32_t j = j0 + threadIdx.x;
        const bool in = j < n;
        const uint64_t k = in ? key[j] : 0ull;
        const uint32_t slot = wave_slots(cursor, in ? exchange_owner(k, world) : 0u, in);
        if (!in) continue;
        unsigned long long w = k;
        if (check) {
            const uint64_t c = check[j];
            w = (k * 0x9E3779B97F4A7C15ull) ^ ((c << 31) | (c >> 33)) ^ (c * 0xC2B2AE3D27D4EB4Full);
            if (w == ~0ull) w = 0;   // (all ones marks an empty slot of the census table)
        }
        out[2 * (size_t)slot] = w;
        out[2 * (size_t)slot + 1] = ((unsigned long long)tid << 1) | (meta_flag(meta[j]) != F_CTX ? 1ull : 0ull);
    }
}

// table[mask + 1] keys (all ones = empty), info[mask + 1] = count | not-CTX sightings << 16 | (another chromosome followed) << 32,
// first_tid[mask + 1] (all ones = none yet); all three start out as 0xFF bytes except info, which starts at zero
__global__ __launch_bounds__(256) void k7_names_insert_kernel(const unsigned long long* in, uint32_t n, unsigned long long* table,
                                                              unsigned long long* info, uint32_t* first_tid, uint32_t mask) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const unsigned long long k = in[2 * (size_t)j], w = in[2 * (size_t)j + 1];
    const uint32_t tid = (uint32_t)(w >> 1), nonctx = (uint32_t)(w & 1);
    uint32_t s = (uint32_t)(((k ^ (k >> 31)) * 0x9E3779B97F4A7C15ull) >> 24) & mask;   // (not the owner's hash: the keys of one owner share that)
    for (;;) {
        const unsigned long long old = atomicCAS(&table[s], ~0ull, k);
        if (old == ~0ull || old == k) break;
        s = (s + 1) & mask;   // (the table has at least twice as many slots as there are records)
    }
    unsigned long long add = 1ull | ((unsigned long long)nonctx << 16);
    const uint32_t ft = atomicCAS(&first_tid[s], 0xFFFFFFFFu, tid);
    if (ft != 0xFFFFFFFFu && ft != tid) add |= 1ull << 32;   // (more than one such sighting only makes the field non-zero)
    atomicAdd(&info[s], add);
}

__global__ __launch_bounds__(256) void k7_names_verdict_kernel(const unsigned long long* table, const unsigned long long* info, uint32_t slots,
                                                               uint32_t* irregular) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= slots || table[s] == ~0ull) return;
    const unsigned long long v = info[s];
    const uint32_t count = (uint32_t)(v & 0xFFFF), nonctx = (uint32_t)((v >> 16) & 0xFFFF);
    const bool spread = (v >> 32) != 0;
    if (count > 2 || (count == 2 && spread && nonctx)) *irregular = 1;
}

void launch_k7_names_count(const uint64_t* key, const uint32_t* n_ptr, uint32_t n_upper, uint32_t world, uint32_t* cnt, hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + 255) / 256, 2048u);
    hipLaunchKernelGGL(k7_names_count_kernel, dim3(g), dim3(256), 0, s, key, n_ptr, world, cnt);
}

void launch_k7_names_scatter(const uint64_t* key, const uint64_t* check, const uint32_t* meta, const uint32_t* n_ptr, uint32_t n_upper, uint32_t world,
                             uint32_t tid, uint32_t* cursor, unsigned long long* out, hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + 255) / 256, 2048u);
    hipLaunchKernelGGL(k7_names_scatter_kernel, dim3(g), dim3(256), 0, s, key, check, meta, n_ptr, world, tid, cursor, out);
}

void launch_k7_names_census(const unsigned long long* in, uint32_t n, unsigned long long* table, unsigned long long* info, uint32_t* first_tid,
                            uint32_t mask, uint32_t* irregular, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k7_names_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, n, table, info, first_tid, mask);
    hipLaunchKernelGGL(k7_names_verdict_kernel, dim3((mask + 256) / 256), dim3(256), 0, s, table, info, mask + 1, irregular);
}

void launch_k7_count(const uint64_t* key, const uint32_t* meta, const uint32_t* n_ptr, uint32_t n_upper, uint32_t world, uint32_t* cnt,
                     hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + 255) / 256, 2048u);
    hipLaunchKernelGGL(k7_count_kernel, dim3(g), dim3(256), 0, s, key, meta, n_ptr, world, cnt);
}

void launch_k7_scatter(const uint64_t* key, const uint64_t* check, const int32_t* region_of, const uint32_t* meta, const int32_t* isize, const uint32_t* n_ptr,
                       uint32_t n_upper, uint32_t world, uint32_t order_base, int32_t region_base, uint32_t* cursor, ExchangeEntry* out,
                       hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + 255) / 256, 2048u);
    hipLaunchKernelGGL(k7_scatter_kernel, dim3(g), dim3(256), 0, s, key, check, region_of, meta, isize, n_ptr, world, order_base, region_base,
                       cursor, out);
}

void launch_k7_unpack(const ExchangeEntry* in, uint32_t n, uint64_t* key, uint64_t* check, uint32_t* order, int32_t* region, uint32_t* meta,
                      int32_t* isize, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k7_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, n, key, check, order, region, meta, isize);
}

}  // namespace bdx

// ---- rank 0 of a sharded run: the gathered packages -> ONE region table and the pair groups bucketed by their later region ----
// (what the host did with memcpy loops and a counting sort until round 3; the packages arrive in HBM and K6 reads its input from HBM)
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

__device__ __forceinline__ uint32_t later_region(const GroupRec& g) { return (uint32_t)((g.key >> 12) & ((1u << 26) - 1)); }
// (a package's groups follow an odd or even number of 36-byte region records: 4-byte aligned only)
__device__ __forceinline__ GroupRec load_group(const char* base, uint32_t i) {
    const uint32_t* w = (const uint32_t*)base + 4 * (size_t)i;
    GroupRec g;
    g.key = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    g.pairs = w[2];
    g.sum_isize = w[3];
    return g;
}

__global__ __launch_bounds__(256) void k8_group_count_kernel(const char* all, GatherDesc D, uint32_t nregions, uint32_t* cnt, uint32_t* err) {
    const GatherPackage P = D.p[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < P.ng; i += gridDim.x * 256) {
        const uint32_t r = later_region(load_group(all + P.groups_off, i));
        if (r >= nregions) { *err = 1; continue; }
        atomicAdd(&cnt[r], 1u);
    }
}

// (the order of a region's groups is free: K6 and the host merge them by key, sums of integers)
__global__ __launch_bounds__(256) void k8_group_scatter_kernel(const char* all, GatherDesc D, uint32_t nregions, uint32_t* cur, GroupRec* out) {
    const GatherPackage P = D.p[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < P.ng; i += gridDim.x * 256) {
        const GroupRec g = load_group(all + P.groups_off, i);
        const uint32_t r = later_region(g);
        if (r < nregions) out[atomicAdd(&cur[r], 1u)] = g;
    }
}

struct BucketIn {
    const uint32_t* cnt;
    __device__ uint32_t operator()(uint32_t j, uint32_t) const { return cnt[j]; }
};
struct BucketOut {   // goff[j + 1] = groups of the regions 0 .. j; the counter becomes the bucket's scatter cursor
    uint32_t* goff;
    uint32_t* cur;
    __device__ void operator()(uint32_t j, uint32_t, uint32_t inc, uint32_t e) const {
        if (j == 0) goff[0] = 0;
        goff[j + 1] = inc;
        cur[j] = inc - e;
    }
};
struct SlotIn {
    const RegionRec* r;
    __device__ uint32_t operator()(uint32_t j, uint32_t) const { return r[j].n; }
};
struct SlotOut {     // K6's slot space: the regions laid end to end
    RegionRec* r;
    uint32_t* total;
    __device__ void operator()(uint32_t j, uint32_t n, uint32_t inc, uint32_t e) const {
        r[j].first = inc - e;
        if (j + 1 == n) *total = inc;
    }
};

void launch_k8_place_regions(const char* all, const GatherDesc& D, uint32_t max_nr, const uint64_t* rbase, int ntids, int nkeys2, RegionRec* r_rec,
                             uint32_t* r_pk, uint32_t* err, hipStream_t s) {
    if (!max_nr) return;
    hipLaunchKernelGGL(k8_place_regions_kernel, dim3((max_nr + 255) / 256, D.world), dim3(256), 0, s, all, D, rbase, ntids, nkeys2, r_rec, r_pk, err);
}

// cnt[nregions] zero on entry; ws: scan_grid(nregions) + 1 words; n_dev: device word holding nregions
void launch_k8_bucket_groups(const char* all, const GatherDesc& D, uint32_t max_ng, uint32_t nregions, const uint32_t* n_dev, uint32_t* cnt,
                             uint32_t* goff, GroupRec* out, uint32_t* ws, uint32_t* err, hipStream_t s) {
    if (!nregions) return;
    const uint32_t g = std::max(1u, std::min((max_ng + 255) / 256, 1024u));
    if (max_ng) hipLaunchKernelGGL(k8_group_count_kernel, dim3(g, D.world), dim3(256), 0, s, all, D, nregions, cnt, err);
    scan_launch<uint32_t>(BucketIn{cnt}, BucketOut{goff, cnt}, n_dev, nregions, ws + 1, ws, s);
    if (max_ng) hipLaunchKernelGGL(k8_group_scatter_kernel, dim3(g, D.world), dim3(256), 0, s, all, D, nregions, cnt, out);
}

void launch_k8_slot_space(RegionRec* r_rec, uint32_t nregions, const uint32_t* n_dev, uint32_t* total, uint32_t* ws, hipStream_t s) {
    if (!nregions) return;
    scan_launch<uint32_t>(SlotIn{r_rec}, SlotOut{r_rec, total}, n_dev, nregions, ws + 1, ws, s);
}

}  // namespace bdx
