// Probe: what a streaming pass gets out of HBM on this part, in the shape K1 reads its input -- several columns, one 16-byte
// (or 8 / 4-byte) non-temporal load per lane and column, a wave per 256-element tile, nothing reused -- against one flat array
// read the same way.  The sum of the loaded words goes to one atomic per workgroup so that nothing is optimised away.
//   flat     one array of 16-byte loads, grid-stride
//   columns  K1's nine columns (5 x i32, 1 x u16, 3 x u8 per element) + its 1-byte store per element
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_probe.hip -o bin/stream_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int kUnroll>
__global__ __launch_bounds__(256) void flat_read(const v4u* __restrict__ src, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
        v4u v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const v4u v = __builtin_nontemporal_load(src + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) atomicAdd(sink, acc);
}

struct Cols {
    const int32_t* c32[5];
    const uint16_t* c16;
    const uint8_t* c8[3];
    uint8_t* out;
    uint8_t* extra;
};

template <int kStore>  // 0: no store, 1: one byte per element (a word per lane), 2: also 128 bytes per fifth tile, 3: as 1 with a non-temporal store
__global__ __launch_bounds__(256) void column_read(Cols c, uint32_t ntiles, unsigned* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned acc = 0;
    const uint32_t nwaves = gridDim.x * 4;
    for (uint32_t tile = blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
        const size_t base = (size_t)tile * 256 + (size_t)lane * 4;
        v4u a[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] = __builtin_nontemporal_load((const v4u*)(c.c32[k] + base));
        const v2u f = __builtin_nontemporal_load((const v2u*)(c.c16 + base));
        unsigned b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) b[k] = __builtin_nontemporal_load((const unsigned*)(c.c8[k] + base));
        unsigned x = f.x ^ f.y ^ b[0] ^ b[1] ^ b[2];
#pragma unroll
        for (int k = 0; k < 5; ++k) x ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
        if (kStore == 1 || kStore == 2) *(unsigned*)(c.out + base) = x;
        if (kStore == 3) __builtin_nontemporal_store(x, (unsigned*)(c.out + base));
        if (kStore == 2 && tile % 5 == 0 && lane < 8) __builtin_nontemporal_store(a[0], (v4u*)(c.extra + (size_t)tile * 512) + lane);
        acc += x;
    }
    if (acc == 0x12345678u) atomicAdd(sink, acc);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 15000000ull;  // elements (reads) of the column case
    const uint32_t ntiles = (uint32_t)(n / 256);
    const size_t ne = (size_t)ntiles * 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* sink;
    CK(hipMalloc(&sink, 4));
    // flat: as many bytes as the column case moves
    const size_t flat_bytes = ne * 26;
    v4u* flat;
    CK(hipMalloc(&flat, flat_bytes));
    CK(hipMemset(flat, 1, flat_bytes));
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        float best = 1e9f;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(flat_read<4>, dim3(grid), dim3(256), 0, 0, flat, flat_bytes / 16, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) best = ms;
        }
        printf("flat read    %7.1f MB, grid %5d x 256, 4 loads in flight per lane: %6.1f us  %6.0f GB/s\n", flat_bytes / 1e6, grid, best * 1e3, flat_bytes / best / 1e6);
    }
    Cols c;
    for (int k = 0; k < 5; ++k) { void* p; CK(hipMalloc(&p, ne * 4)); CK(hipMemset(p, k + 1, ne * 4)); c.c32[k] = (const int32_t*)p; }
    { void* p; CK(hipMalloc(&p, ne * 2)); CK(hipMemset(p, 7, ne * 2)); c.c16 = (const uint16_t*)p; }
    for (int k = 0; k < 3; ++k) { void* p; CK(hipMalloc(&p, ne)); CK(hipMemset(p, k + 9, ne)); c.c8[k] = (const uint8_t*)p; }
    { void* p; CK(hipMalloc(&p, ne)); c.out = (uint8_t*)p; }
    { void* p; CK(hipMalloc(&p, (size_t)ntiles * 512)); c.extra = (uint8_t*)p; }
    for (int store = 0; store < 4; ++store)
        for (int grid : {2048, 8192}) {
            float best = 1e9f;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                if (store == 0) hipLaunchKernelGGL(column_read<0>, dim3(grid), dim3(256), 0, 0, c, ntiles, sink);
                else if (store == 1) hipLaunchKernelGGL(column_read<1>, dim3(grid), dim3(256), 0, 0, c, ntiles, sink);
                else if (store == 2) hipLaunchKernelGGL(column_read<2>, dim3(grid), dim3(256), 0, 0, c, ntiles, sink);
                else hipLaunchKernelGGL(column_read<3>, dim3(grid), dim3(256), 0, 0, c, ntiles, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            const size_t bytes = ne * (25 + (store >= 1 ? 1 : 0)) + (store == 2 ? (size_t)(ntiles / 5) * 128 : 0);
            printf("nine columns, %zu elements, 25 B in%s, grid %5d x 256: %6.1f us  %6.0f GB/s\n", ne,
                   store == 0 ? "                                     " : (store == 1 ? " + 1 B out per element               " : (store == 2 ? " + 1 B out + 128 B per fifth tile out" : " + 1 B out per element, non-temporal ")), grid, best * 1e3,
                   bytes / best / 1e6);
        }
    return 0;
}
