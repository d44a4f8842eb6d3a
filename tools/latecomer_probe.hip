// A kernel submitted to an idle stream WHILE another stream's kernel is running: when does it start?  By the running kernel's shape.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int kThreads, int kMin>
__global__ __launch_bounds__(kThreads, kMin) void nap(unsigned long long ticks, unsigned long long* when) {
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { when[blockIdx.x] = t0; lds[0] = 1; }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 1: the other stream ran a kernel before and was NOT waited for by the host since; 2: waited for through an event of another stream instead
    unsigned long long *ta, *tb;
    CK(hipHostMalloc(&ta, 8192 * 8)); CK(hipHostMalloc(&tb, 8192 * 8));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    struct Case { int wgs, lds; const char* what; } cases[] = {
        {1, 0, "ONE wave, no LDS"}, {256, 0, "one wave per CU, no LDS"}, {256 * 8, 0, "8 waves per CU, no LDS"}, {256 * 8, 5000, "8 waves per CU, 5 KB LDS each"},
        {256 * 16, 5000, "16 waves per CU, 5 KB LDS each"}, {256 * 28, 0, "28 waves per CU, no LDS"}, {256 * 28, 5000, "28 waves per CU, 5 KB LDS each"}};
    unsigned long long* tc;
    CK(hipHostMalloc(&tc, 8192 * 8));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int rep = 0; rep < 2; ++rep)
        for (auto& c : cases) {
            for (int i = 0; i < 8192; ++i) ta[i] = tb[i] = 0;
            if (mode) {   // a kernel on the other stream, finished long before (its end observed through the pinned word it writes, not through the runtime)
                tc[0] = 0;
                hipLaunchKernelGGL((nap<64, 1>), dim3(1), dim3(64), 0, sb, 1000ull, tc);
                const auto w0 = std::chrono::steady_clock::now();
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.005) {}
            }
            hipLaunchKernelGGL((nap<64, 8>), dim3(c.wgs), dim3(64), c.lds, sa, 2000000ull, ta);   // 20 ms
            const auto h0 = std::chrono::steady_clock::now();
            while (ta[0] == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() < 1.0) {}
            hipLaunchKernelGGL((nap<64, 1>), dim3(512), dim3(64), 0, sb, 10000ull, tb);
            if (mode) {   // (the host learns that everything is through without touching the other stream)
                const auto w0 = std::chrono::steady_clock::now();
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.05) {}
                CK(hipStreamSynchronize(sa));
            } else {
                CK(hipStreamSynchronize(sb));
                CK(hipStreamSynchronize(sa));
            }
            unsigned long long a0 = ~0ull, b0 = ~0ull, b1 = 0;
            for (int i = 0; i < c.wgs; ++i) a0 = std::min(a0, ta[i]);
            for (int i = 0; i < 512; ++i) { b0 = std::min(b0, tb[i]); b1 = std::max(b1, tb[i]); }
            printf("running: %-34s -> 512 one-wave workgroups submitted to the other stream a moment later started %.3f .. %.3f ms after it\n", c.what,
                   ((double)b0 - (double)a0) / 1e5, ((double)b1 - (double)a0) / 1e5);
        }
    return 0;
}
