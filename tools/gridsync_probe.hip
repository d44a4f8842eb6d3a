// Probe: what a grid-wide barrier inside one kernel costs against a kernel boundary (launch + start-up + end).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gridsync_probe.hip -o bin/gridsync_probe
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

namespace cg = cooperative_groups;

__global__ void coop(unsigned* data, int nsync) {
    cg::grid_group g = cg::this_grid();
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int s = 0; s < nsync; ++s) {
        data[i] += 1;
        g.sync();
    }
}

// a hand-made barrier: relaxed device-scope ticket per round, no fence (the data exchanged between phases would use device-scope
// loads / stores itself)
__global__ void manual(unsigned* data, int nsync, unsigned* ticket) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int s = 0; s < nsync; ++s) {
        data[i] += 1;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ticket + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ticket + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {}
        }
        __syncthreads();
    }
}

__global__ void plain(unsigned* data) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    data[i] += 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    for (int grid : {64, 256, 586, 1024}) {
        unsigned *d, *t;
        CK(hipMalloc(&d, (size_t)grid * 256 * 4)); CK(hipMalloc(&t, 4096));
        CK(hipMemset(d, 0, (size_t)grid * 256 * 4));
        for (int nsync : {1, 11}) {
            double best[3] = {1e9, 1e9, 1e9};
            for (int rep = 0; rep < 20; ++rep) {
                void* args[] = {&d, &nsync};
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                CK(hipLaunchCooperativeKernel((void*)coop, dim3(grid), dim3(256), args, 0, 0));
                CK(hipDeviceSynchronize());
                auto t1 = std::chrono::steady_clock::now();
                CK(hipMemsetAsync(t, 0, 4096, 0));
                CK(hipDeviceSynchronize());
                auto t2 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(manual, dim3(grid), dim3(256), 0, 0, d, nsync, t);
                CK(hipDeviceSynchronize());
                auto t3 = std::chrono::steady_clock::now();
                for (int s = 0; s < nsync; ++s) hipLaunchKernelGGL(plain, dim3(grid), dim3(256), 0, 0, d);
                CK(hipDeviceSynchronize());
                auto t4 = std::chrono::steady_clock::now();
                best[0] = std::min(best[0], std::chrono::duration<double, std::micro>(t1 - t0).count());
                best[1] = std::min(best[1], std::chrono::duration<double, std::micro>(t3 - t2).count());
                best[2] = std::min(best[2], std::chrono::duration<double, std::micro>(t4 - t3).count());
            }
            printf("grid %4d x 256, %2d phases: cooperative grid.sync %.1f us, hand-made ticket barrier %.1f us, %d separate launches %.1f us (host clock, launch + sync included)\n",
                   grid, nsync, best[0], best[1], nsync, best[2]);
        }
        CK(hipFree(d)); CK(hipFree(t));
    }
    return 0;
}
