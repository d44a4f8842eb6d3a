// A long-running kernel that leaves wave slots free (28 single-wave workgroups per CU, ~5 KB of LDS each, all resident): do other streams'
// kernels get dispatched into the free slots while it runs, and what workgroup shapes fit?  (cf. tools/cumask_probe.hip: a kernel with
// workgroups still waiting for dispatch keeps every other queue's kernels out, and a CU-masked queue held only 16 waves per CU.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int kThreads, int kMin>
__global__ __launch_bounds__(kThreads, kMin) void nap(unsigned long long ticks, unsigned long long* when) {
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { when[blockIdx.x] = t0; lds[0] = 1; }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int main() {
    unsigned long long *ta, *tb;
    CK(hipHostMalloc(&ta, 8192 * 8)); CK(hipHostMalloc(&tb, 8192 * 8));
    hipStream_t st[6];
    for (auto& x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int pair = 1; pair < 6; ++pair) {   // (HIP spreads a process's streams over a few hardware queues: two streams on one queue run in order)
        hipStream_t sa = st[0], sb = st[pair];
        const int per_cu = pair == 5 ? 32 : 28;
        for (int shape = 0; shape < 4; ++shape) {
            const int threads = shape == 0 ? 64 : shape == 1 ? 256 : shape == 2 ? 256 : 1024;
            const int lds = shape == 0 ? 2048 : shape == 1 ? 4096 : shape == 2 ? 32768 : 4096;
            const int nb = 2048;
            for (int i = 0; i < 8192; ++i) ta[i] = tb[i] = 0;
            hipLaunchKernelGGL((nap<64, 8>), dim3(256 * per_cu), dim3(64), 5000, sa, 2000000ull, ta);   // 20 ms
            const auto h0 = std::chrono::steady_clock::now();
            if (pair & 1) while (ta[0] == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() < 1.0) {}   // (even pairs: enqueued back to back)
            if (threads == 64) hipLaunchKernelGGL((nap<64, 1>), dim3(nb), dim3(64), lds, sb, 10000ull, tb);   // 0.1 ms each
            else if (threads == 256) hipLaunchKernelGGL((nap<256, 1>), dim3(nb), dim3(256), lds, sb, 10000ull, tb);
            else hipLaunchKernelGGL((nap<1024, 1>), dim3(nb), dim3(1024), lds, sb, 10000ull, tb);
            CK(hipStreamSynchronize(sb));
            CK(hipStreamSynchronize(sa));
            unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
            for (int i = 0; i < 256 * per_cu; ++i) { a0 = std::min(a0, ta[i]); a1 = std::max(a1, ta[i]); }
            for (int i = 0; i < nb; ++i) { b0 = std::min(b0, tb[i]); b1 = std::max(b1, tb[i]); }
            printf("streams 0 and %d: long kernel %d waves per CU (all started within %.3f ms); beside it %d workgroups of %4d threads, %5d B LDS, 0.1 ms each: first started %.3f ms "
                   "after the long kernel's first, last started %.3f ms after it\n", pair, per_cu, (a1 - a0) / 1e5, nb, threads, lds, ((double)b0 - (double)a0) / 1e5, ((double)b1 - (double)a0) / 1e5);
        }
    }
    return 0;
}
