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

// cursor[d] starts at the destination's offset in the send buffer (entries)
__global__ __launch_bounds__(256) void k7_scatter_kernel(const uint64_t* key, const int32_t* region_of, const uint32_t* meta,
                                                         const int32_t* isize, const uint32_t* n_ptr, uint32_t world, uint32_t order_base,
                                                         int32_t region_base, uint32_t* cursor, ExchangeEntry* out) {
    const uint32_t n = *n_ptr;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const uint32_t m = meta[j];
        if (meta_flag(m) != F_CTX) continue;
        const uint64_t k = key[j];
        const uint32_t slot = atomicAdd(&cursor[exchange_owner(k, world)], 1u);
        const int32_t r = region_of[j];
        ExchangeEntry e;
        e.key = k; e.order = order_base + j; e.region = r < 0 ? -1 : r + region_base; e.meta = m; e.isize = isize[j];
        out[slot] = e;
    }
}

__global__ __launch_bounds__(256) void k7_unpack_kernel(const ExchangeEntry* in, uint32_t n, uint64_t* key, uint32_t* order, int32_t* region,
                                                        uint32_t* meta, int32_t* isize) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const ExchangeEntry e = in[j];
    key[j] = e.key; order[j] = e.order; region[j] = e.region; meta[j] = e.meta; isize[j] = e.isize;
}

void launch_k7_count(const uint64_t* key, const uint32_t* meta, const uint32_t* n_ptr, uint32_t n_upper, uint32_t world, uint32_t* cnt,
                     hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + 255) / 256, 2048u);
    hipLaunchKernelGGL(k7_count_kernel, dim3(g), dim3(256), 0, s, key, meta, n_ptr, world, cnt);
}

void launch_k7_scatter(const uint64_t* key, const int32_t* region_of, const uint32_t* meta, const int32_t* isize, const uint32_t* n_ptr,
                       uint32_t n_upper, uint32_t world, uint32_t order_base, int32_t region_base, uint32_t* cursor, ExchangeEntry* out,
                       hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = std::min<uint32_t>((n_upper + 255) / 256, 2048u);
    hipLaunchKernelGGL(k7_scatter_kernel, dim3(g), dim3(256), 0, s, key, region_of, meta, isize, n_ptr, world, order_base, region_base,
                       cursor, out);
}

void launch_k7_unpack(const ExchangeEntry* in, uint32_t n, uint64_t* key, uint32_t* order, int32_t* region, uint32_t* meta, int32_t* isize,
                      hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k7_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, n, key, order, region, meta, isize);
}

}  // namespace bdx
