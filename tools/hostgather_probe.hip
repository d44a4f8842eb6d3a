// Probe: how fast can the GPU fetch 150 k scattered 8-byte + 2-byte values from pinned HOST memory (the name keys and read lengths of
// the anomalous reads that bdx_push leaves in the caller's arrays)?  One lane per value, all in flight at once.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hostgather_probe.hip -o bin/hostgather_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void gather(const uint64_t* key, const uint16_t* qlen, const uint32_t* idx, uint32_t n, uint64_t* ok, uint16_t* oq) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    ok[i] = key[j];
    oq[i] = qlen[j];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    const uint32_t N = 15000000, clusters = 6250 * 2, per = 12;
    uint64_t* hk; uint16_t* hq;
    CK(hipHostMalloc(&hk, (size_t)N * 8)); CK(hipHostMalloc(&hq, (size_t)N * 2));
    for (uint32_t i = 0; i < N; ++i) { hk[i] = i * 0x9E3779B97F4A7C15ull; hq[i] = 100; }
    std::vector<uint32_t> idx;
    uint64_t s = 88172645463325252ull;
    for (uint32_t c = 0; c < clusters; ++c) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t base = (uint32_t)(s % (N - 200));
        for (uint32_t k = 0; k < per; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; idx.push_back(base + (uint32_t)(s % 90)); }
    }
    const uint32_t n = (uint32_t)idx.size();
    uint32_t* di; uint64_t* ok; uint16_t* oq;
    CK(hipMalloc(&di, n * 4)); CK(hipMalloc(&ok, n * 8)); CK(hipMalloc(&oq, n * 2));
    CK(hipMemcpy(di, idx.data(), n * 4, hipMemcpyHostToDevice));
    uint64_t* dk; uint16_t* dq;
    CK(hipHostGetDevicePointer((void**)&dk, hk, 0)); CK(hipHostGetDevicePointer((void**)&dq, hq, 0));
    for (int block : {64, 256, 1024}) {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(gather, dim3((n + block - 1) / block), dim3(block), 0, 0, dk, dq, di, n, ok, oq);
            CK(hipDeviceSynchronize());
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("%u scattered (key, length) pairs from pinned host memory, %d threads per workgroup: %.3f ms\n", n, block, best);
    }
    // for comparison: copying both whole columns (150 MB)
    uint64_t* ck; uint16_t* cq;
    CK(hipMalloc(&ck, (size_t)N * 8)); CK(hipMalloc(&cq, (size_t)N * 2));
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(ck, hk, (size_t)N * 8, hipMemcpyHostToDevice, 0));
        CK(hipMemcpyAsync(cq, hq, (size_t)N * 2, hipMemcpyHostToDevice, 0));
        CK(hipDeviceSynchronize());
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("copying both columns whole (150 MB): %.3f ms\n", best);
    return 0;
}
