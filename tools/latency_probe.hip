// Probe: what a dependent load costs in the situations the kernels behind K2 are in.
//   A  data written by the kernel before (the usual case: every first touch of a kernel of the tail)
//   B  the same lines again (now in the reading die's L2 / the CU's L1)
//   C  data written earlier in the SAME kernel by a workgroup of the SAME compute die (blockIdx % 8 equal), read with
//      device-scope loads behind a flag: what a single-launch tail confined to one die could count on
//   D  ... by a workgroup of ANOTHER die
// One lane chases a random cycle for kHops steps and clocks itself with wall_clock64() (100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/latency_probe.hip -o bin/latency_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

constexpr int kHops = 64;

__global__ void produce(uint32_t* dst, const uint32_t* src, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// out[0] = ticks of the first chase, out[1] = of the second over the same lines
__global__ void chase(const uint32_t* next, uint32_t start, unsigned long long* out, uint32_t* sink) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int rep = 0; rep < 2; ++rep) {
        uint32_t i = start;
        const unsigned long long t0 = wall_clock64();
        for (int h = 0; h < kHops; ++h) i = next[i];
        const unsigned long long t1 = wall_clock64();
        out[rep] = t1 - t0;
        if (i == 0xFFFFFFFFu) *sink = i;
    }
}

// workgroup 0 writes the cycle and raises a flag; workgroup `reader` (same die when reader % 8 == 0) waits and chases with
// device-scope loads
__global__ void produce_then_chase(uint32_t* dst, const uint32_t* src, uint32_t n, uint32_t start, uint32_t reader, unsigned long long* flag,
                                   unsigned long long stamp, unsigned long long* out, uint32_t* sink) {
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        __threadfence();
        if (threadIdx.x == 0) __hip_atomic_store(flag, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (blockIdx.x != reader || threadIdx.x != 0) return;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != stamp) {}
    for (int rep = 0; rep < 2; ++rep) {
        uint32_t i = start;
        const unsigned long long t0 = wall_clock64();
        for (int h = 0; h < kHops; ++h) i = __hip_atomic_load(&dst[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t1 = wall_clock64();
        out[rep] = t1 - t0;
        if (i == 0xFFFFFFFFu) *sink = i;
    }
}

// E / F: every hop in another page: dst is `pages` pages of `page_words` words, hop h reads word 0 of page order[h] (the links are
// laid out by the host); out[0] = ticks of the first hop alone, out[1] = of all kHops hops, out[2] = of the same hops again
__global__ void chase_pages(const uint32_t* next, uint32_t start, unsigned long long* out, uint32_t* sink) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int rep = 0; rep < 2; ++rep) {
        uint32_t i = start;
        const unsigned long long t0 = wall_clock64();
        i = next[i];
        const unsigned long long t1 = wall_clock64() + (i == 0xFFFFFFFFu ? 1 : 0);
        for (int h = 1; h < kHops; ++h) i = next[i];
        const unsigned long long t2 = wall_clock64() + (i == 0xFFFFFFFFu ? 1 : 0);
        if (rep == 0) { out[0] = t1 - t0; out[1] = t2 - t0; } else { out[2] = t2 - t0; }
        if (i == 0xFFFFFFFFu) *sink = i;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    const uint32_t n = 1u << 20;  // 4 MB of 4-byte links, one hop per 128-byte line at least
    std::vector<uint32_t> perm(n), next(n);
    std::iota(perm.begin(), perm.end(), 0u);
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = n - 1; i > 0; --i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        std::swap(perm[i], perm[(uint32_t)(s % (i + 1))]);
    }
    for (uint32_t i = 0; i < n; ++i) next[perm[i]] = perm[(i + 1) % n];
    uint32_t *d_src, *d_dst, *d_sink;
    unsigned long long *d_out, *d_flag;
    CK(hipMalloc(&d_src, n * 4)); CK(hipMalloc(&d_dst, n * 4)); CK(hipMalloc(&d_sink, 4)); CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_flag, 8));
    CK(hipMemcpy(d_src, next.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_flag, 0, 8));
    unsigned long long out[2];
    auto us = [](unsigned long long t) { return t / 100.0 / kHops; };
    for (int round = 0; round < 3; ++round) {
        hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, d_dst, d_src, n);
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d_dst, perm[round * 1000], d_out, d_sink);
        CK(hipMemcpy(out, d_out, 16, hipMemcpyDeviceToHost));
        printf("A written by the previous kernel: %.2f us per dependent load;  B same lines again: %.2f us\n", us(out[0]), us(out[1]));
    }
    for (uint32_t reader : {8u, 16u, 1u, 3u}) {
        for (int round = 0; round < 2; ++round) {
            const unsigned long long stamp = 1000ull * reader + round + 1;
            hipLaunchKernelGGL(produce_then_chase, dim3(32), dim3(256), 0, 0, d_dst, d_src, n, perm[round * 777 + reader], reader, d_flag, stamp, d_out, d_sink);
            CK(hipMemcpy(out, d_out, 16, hipMemcpyDeviceToHost));
            printf("%s same kernel, writer workgroup 0, reader workgroup %u (%s die): %.2f us per device-scope dependent load; again: %.2f us\n",
                   reader % 8 == 0 ? "C" : "D", reader, reader % 8 == 0 ? "same" : "another", us(out[0]), us(out[1]));
        }
    }
    // one hop per page: page sizes 4 KB .. 2 MB, 64 pages visited in a scrambled order
    for (uint32_t page_bytes : {4096u, 65536u, 2097152u}) {
        const uint32_t pw = page_bytes / 4, pages = 128;
        std::vector<uint32_t> order(pages);
        std::iota(order.begin(), order.end(), 0u);
        for (uint32_t i = pages - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(order[i], order[(uint32_t)(s % (i + 1))]); }
        std::vector<uint32_t> links((size_t)pages * pw, 0u);
        for (uint32_t i = 0; i < pages; ++i) links[(size_t)order[i] * pw] = order[(i + 1) % pages] * pw;
        uint32_t *d_big, *d_big2;
        CK(hipMalloc(&d_big, links.size() * 4)); CK(hipMalloc(&d_big2, links.size() * 4));
        CK(hipMemcpy(d_big2, links.data(), links.size() * 4, hipMemcpyHostToDevice));
        unsigned long long o3[3];
        for (int round = 0; round < 2; ++round) {
            hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, d_big, d_big2, (uint32_t)links.size());
            hipLaunchKernelGGL(chase_pages, dim3(1), dim3(64), 0, 0, d_big, order[0] * pw, d_out, d_sink);
            CK(hipMemcpy(o3, d_out, 24, hipMemcpyDeviceToHost));
            printf("E one hop per %7u-byte page, written by the previous kernel: first hop %.2f us, %.2f us per hop over %d pages; again: %.2f us per hop\n",
                   page_bytes, o3[0] / 100.0, o3[1] / 100.0 / kHops, kHops, o3[2] / 100.0 / kHops);
        }
        CK(hipFree(d_big)); CK(hipFree(d_big2));
    }
    return 0;
}
