// Beside a long-running kernel (the persistent inflate kernel's shape: 28 single-wave workgroups per CU, all resident), can another stream
// be kept going?  tools/coresident_probe.hip: a kernel SUBMITTED to an idle stream while the long kernel runs starts only when that one ends.
// Here the second stream is never idle: a one-wave kernel spins at its tail waiting for a flag, work is enqueued behind it, then the flag is set.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int kThreads, int kMin>
__global__ __launch_bounds__(kThreads, kMin) void nap(unsigned long long ticks, unsigned long long* when) {
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { when[blockIdx.x] = t0; lds[0] = 1; }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void wait_flag(const uint32_t* flag, uint32_t want, unsigned long long* when, unsigned long long limit_ticks) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want && wall_clock64() - t0 < limit_ticks) __builtin_amdgcn_s_sleep(64);
    when[0] = wall_clock64();
}
static double ms_since(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main(int argc, char** argv) {
    const bool device_flag = argc > 1;   // the flag in device memory, set by a 4-byte H2D copy on a third stream (else: pinned host memory, set by the CPU)
    unsigned long long *ta, *tb, *tw;
    uint32_t* hflag;
    CK(hipHostMalloc(&ta, 8192 * 8)); CK(hipHostMalloc(&tb, 3 * 2048 * 8)); CK(hipHostMalloc(&tw, 64)); CK(hipHostMalloc(&hflag, 64));
    uint32_t* dflag;
    CK(hipMalloc(&dflag, 64)); CK(hipMemset(dflag, 0, 64));
    char *hsrc, *ddst;
    CK(hipHostMalloc(&hsrc, 32 << 20)); CK(hipMalloc(&ddst, 32 << 20));
    hipStream_t sa, sb, sc;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int i = 0; i < 8192; ++i) ta[i] = 0;
    hflag[0] = 0;
    const uint32_t* flag = device_flag ? dflag : hflag;
    const unsigned long long limit = 200ull * 100000;   // no wait outlives 200 ms
    hipLaunchKernelGGL((nap<64, 8>), dim3(256 * 28), dim3(64), 5000, sa, 60ull * 100000, ta);   // 60 ms
    hipLaunchKernelGGL(wait_flag, dim3(1), dim3(64), 0, sb, flag, 1u, tw + 0, limit);
    const auto h0 = std::chrono::steady_clock::now();
    while (ta[0] == 0 && ms_since(h0) < 1000) {}
    for (int round = 1; round <= 3; ++round) {
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
        hipLaunchKernelGGL((nap<256, 1>), dim3(2048), dim3(256), 4096, sb, 10000ull, tb + (round - 1) * 2048);
        hipLaunchKernelGGL(wait_flag, dim3(1), dim3(64), 0, sb, flag, (uint32_t)round + 1, tw + round, limit);
        // an 8 MiB H2D copy on the third stream, then the flag
        const auto c0 = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(ddst, hsrc, 8 << 20, hipMemcpyHostToDevice, sc));
        hflag[8] = (uint32_t)round;
        if (device_flag) CK(hipMemcpyAsync(dflag, hflag + 8, 4, hipMemcpyHostToDevice, sc));
        CK(hipEventRecord(ev, sc));
        CK(hipEventSynchronize(ev));
        const double copy_ms = ms_since(c0);
        if (!device_flag) __atomic_store_n(hflag, (uint32_t)round, __ATOMIC_RELEASE);
        printf("round %d: flag set %.2f ms after the long kernel started (the copies before it took %.3f ms)\n", round, ms_since(h0), copy_ms);
    }
    CK(hipStreamSynchronize(sb));
    if (device_flag) { hflag[8] = 9; CK(hipMemcpyAsync(dflag, hflag + 8, 4, hipMemcpyHostToDevice, sc)); } else __atomic_store_n(hflag, 9u, __ATOMIC_RELEASE);
    CK(hipDeviceSynchronize());
    unsigned long long a0 = ~0ull;
    for (int i = 0; i < 256 * 28; ++i) a0 = std::min(a0, ta[i]);
    for (int round = 1; round <= 3; ++round) {
        unsigned long long b0 = ~0ull, b1 = 0;
        for (int i = 0; i < 2048; ++i) { b0 = std::min(b0, tb[(round - 1) * 2048 + i]); b1 = std::max(b1, tb[(round - 1) * 2048 + i]); }
        printf("round %d (%s flag): the wait in front ended %.3f ms after the long kernel's start; the 2048 workgroups behind it started %.3f .. %.3f ms\n", round,
               device_flag ? "device" : "pinned", ((double)tw[round - 1] - (double)a0) / 1e5, ((double)b0 - (double)a0) / 1e5, ((double)b1 - (double)a0) / 1e5);
    }
    return 0;
}
