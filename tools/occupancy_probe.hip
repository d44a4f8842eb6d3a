// How many single-wave workgroups does a CU hold at once, by LDS per workgroup?  (8,192 workgroups of 64 threads that each sleep 5 ms:
// if all are resident they all start within microseconds; a second round starts 5 ms later.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(64, 8) void nap(unsigned long long ticks, unsigned long long* when) {
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { when[blockIdx.x] = t0; lds[0] = 1; }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int main() {
    const int n = 256 * 32;
    unsigned long long* t;
    CK(hipHostMalloc(&t, n * 8));
    for (int lds : {0, 1024, 2048, 4096, 5000, 5120, 5888, 8192}) {
        hipLaunchKernelGGL(nap, dim3(n), dim3(64), lds, 0, 500000ull, t);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> v(t, t + n);
        std::sort(v.begin(), v.end());
        int first_round = 0;
        while (first_round < n && v[first_round] - v[0] < 250000ull) ++first_round;
        printf("LDS %5d B per workgroup: %d of %d workgroups started in the first round (%.1f per CU), the last one %.3f ms after the first\n", lds, first_round, n,
               first_round / 256.0, (v[n - 1] - v[0]) / 100000.0);
    }
    return 0;
}
