// What FETCH_SIZE / WRITE_SIZE count for SCATTERED access on gfx950 (the guide's x2 correction of FETCH_SIZE is calibrated for wide streaming
// loads only): kernels that move a KNOWN number of bytes -- streams of 16-byte loads / stores, gathers of 8 and 16 bytes from random 64-byte
// lines, scattered 4- and 8-byte stores, device-scope atomics on random words -- for tools/pmc_scatter.sh to run under rocprofv3 --pmc.
//   every kernel touches n = 2^22 elements spread over a 1 GiB buffer (far beyond L2 + Infinity Cache per launch of random lines)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
constexpr size_t kBytes = (size_t)1 << 30;
constexpr uint32_t kN = 1u << 22;
__global__ void stream_read16(const uint4* p, uint32_t n, uint32_t* sink) { uint32_t acc = 0; for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.w; } if (acc == 0x12345) *sink = acc; }
__global__ void stream_write16(uint4* p, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = make_uint4(i, i, i, i); }
__global__ void gather8(const uint64_t* p, uint32_t n, uint32_t* sink) { uint32_t acc = 0; for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += (uint32_t)p[(mix(i) % (kBytes / 64)) * 8 + (i & 7)]; if (acc == 0x12345) *sink = acc; }
__global__ void gather16(const uint4* p, uint32_t n, uint32_t* sink) { uint32_t acc = 0; for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += p[(mix(i) % (kBytes / 64)) * 4 + (i & 3)].x; if (acc == 0x12345) *sink = acc; }
__global__ void scatter4(uint32_t* p, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[(mix(i) % (kBytes / 64)) * 16 + (i & 15)] = i; }
__global__ void scatter8(uint64_t* p, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[(mix(i) % (kBytes / 64)) * 8 + (i & 7)] = i; }
__global__ void atomic_add4(uint32_t* p, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(p + (mix(i) % (kBytes / 64)) * 16 + (i & 15), 1u); }
__global__ void atomic_cas8(unsigned long long* p, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicCAS(p + (mix(i) % (kBytes / 64)) * 8 + (i & 7), 0ull, (unsigned long long)i); }
// the same atomics on a table that stays in the caches (4 MiB): what the join's table and the census pay when their lines are resident
__global__ void atomic_cas8_small(unsigned long long* p, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicCAS(p + (mix(i) % ((4u << 20) / 8)), 0ull, (unsigned long long)i); }
int main() {
    void* buf;
    uint32_t* sink;
    CK(hipMalloc(&buf, kBytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, kBytes));
    const dim3 g(4096), b(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_read16, g, b, 0, 0, (const uint4*)buf, (uint32_t)(kBytes / 16), sink);
        hipLaunchKernelGGL(stream_write16, g, b, 0, 0, (uint4*)buf, (uint32_t)(kBytes / 16));
        hipLaunchKernelGGL(gather8, g, b, 0, 0, (const uint64_t*)buf, kN, sink);
        hipLaunchKernelGGL(gather16, g, b, 0, 0, (const uint4*)buf, kN, sink);
        hipLaunchKernelGGL(scatter4, g, b, 0, 0, (uint32_t*)buf, kN);
        hipLaunchKernelGGL(scatter8, g, b, 0, 0, (uint64_t*)buf, kN);
        hipLaunchKernelGGL(atomic_add4, g, b, 0, 0, (uint32_t*)buf, kN);
        hipLaunchKernelGGL(atomic_cas8, g, b, 0, 0, (unsigned long long*)buf, kN);
        hipLaunchKernelGGL(atomic_cas8_small, g, b, 0, 0, (unsigned long long*)buf, kN);
        CK(hipDeviceSynchronize());
    }
    printf("n = %u elements per scattered kernel, %zu bytes streamed by the two streaming kernels\n", kN, kBytes);
    return 0;
}
