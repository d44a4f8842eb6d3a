// what the HIP runtime's start-up consists of on the box (tools, not the product): hipcc --offload-arch=gfx950 -O2 tools/init_probe.hip -o /tmp/init_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int* p) { *p = 1; }
int main() {
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    hipInit(0);                    printf("hipInit            %7.2f ms\n", ms());
    int n = 0; hipGetDeviceCount(&n); printf("hipGetDeviceCount  %7.2f ms (%d)\n", ms(), n);
    hipSetDevice(0);               printf("hipSetDevice       %7.2f ms\n", ms());
    hipFree(nullptr);              printf("hipFree(0)         %7.2f ms\n", ms());
    int* d = nullptr; hipMalloc(&d, 4); printf("first hipMalloc    %7.2f ms\n", ms());
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); printf("first stream       %7.2f ms\n", ms());
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); printf("second stream      %7.2f ms\n", ms());
    void* h = nullptr; hipHostMalloc(&h, 1 << 20, 0); printf("first hipHostMalloc %6.2f ms\n", ms());
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s, d); hipStreamSynchronize(s); printf("first launch+sync  %7.2f ms\n", ms());
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s2, d); hipStreamSynchronize(s2); printf("launch on stream 2 %7.2f ms\n", ms());
    return 0;
}
