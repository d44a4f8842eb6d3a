// What a process pays at its end for the memory it holds: device memory never touched / written once, pinned host memory.
// usage: exit_cost_probe <GB device untouched> <GB device written> <GB pinned> [free]   (prints the time of its own allocations; time the process from outside)
#include <hip/hip_runtime.h>
#include <chrono>
#include <time.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const double t0 = now();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    const double t1 = now();
    const double gu = argc > 1 ? atof(argv[1]) : 0, gw = argc > 2 ? atof(argv[2]) : 0, gp = argc > 3 ? atof(argv[3]) : 0;
    std::vector<void*> ptrs;
    for (double g = 0; g < gu; g += 2) { void* p; CK(hipMalloc(&p, (size_t)2 << 30)); ptrs.push_back(p); }
    const double t2 = now();
    for (double g = 0; g < gw; g += 2) { void* p; CK(hipMalloc(&p, (size_t)2 << 30)); CK(hipMemsetAsync(p, 1, (size_t)2 << 30, 0)); ptrs.push_back(p); }
    CK(hipDeviceSynchronize());
    const double t3 = now();
    void* hp = nullptr;
    if (gp > 0) CK(hipHostMalloc(&hp, (size_t)(gp * (1 << 30))));
    const double t4 = now();
    if (argc > 4) { for (void* p : ptrs) CK(hipFree(p)); if (hp) CK(hipHostFree(hp)); }
    const double t5 = now();
    printf("runtime up %.3f s; %.0f GB untouched %.3f s; %.0f GB written %.3f s; %.1f GB pinned %.3f s; freeing %.3f s; ", t1 - t0, gu, t2 - t1, gw, t3 - t2, gp, t4 - t3, t5 - t4);
    {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        printf("exit_at %.6f ", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
    }
    fflush(stdout);
    _exit(0);
}
