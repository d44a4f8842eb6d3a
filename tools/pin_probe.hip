// Pinned host memory the GPU writes results into: what it costs to get and to give back, by how it is obtained --
// hipHostMalloc, or anonymous memory (transparent huge pages asked for) registered with hipHostRegister.
//   pin_probe <method 0|1|2> <GB>     0 hipHostMalloc, 1 mmap + MADV_HUGEPAGE + touch + hipHostRegister, 2 the same without MADV_HUGEPAGE
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(uint32_t* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }
int main(int argc, char** argv) {
    const int method = argc > 1 ? atoi(argv[1]) : 0;
    const size_t bytes = (size_t)((argc > 2 ? atof(argv[2]) : 0.5) * (1 << 30)) & ~(size_t)((2 << 20) - 1);
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    const double t0 = now();
    void* h = nullptr; void* dptr = nullptr;
    double t_touch = 0;
    if (method == 0) { CK(hipHostMalloc(&h, bytes)); dptr = h; }
    else {
        h = mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        h = (void*)(((uintptr_t)h + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
        if (method == 1) madvise(h, bytes, MADV_HUGEPAGE);
        const double a = now();
        for (size_t o = 0; o < bytes; o += 4096) ((volatile char*)h)[o] = 0;
        t_touch = now() - a;
        CK(hipHostRegister(h, bytes, hipHostRegisterMapped));
        CK(hipHostGetDevicePointer(&dptr, h, 0));
    }
    const double t1 = now();
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (uint32_t*)dptr, bytes / 4);
    CK(hipDeviceSynchronize());
    const double t2 = now();
    const bool ok = ((uint32_t*)h)[12345] == 12345u && ((uint32_t*)h)[bytes / 4 - 1] == (uint32_t)(bytes / 4 - 1);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    printf("method %d, %.2f GB: obtained in %.3f s (touching %.3f), same pointer on the device: %s, GPU wrote it in %.3f s (%s) exit_at %.6f\n", method, bytes / 1073741824.0, t1 - t0, t_touch,
           dptr == h ? "yes" : "no", t2 - t1, ok ? "read back ok" : "WRONG", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
    fflush(stdout);
    _exit(0);
}
