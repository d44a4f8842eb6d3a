// Does a CU-masked stream keep a kernel off the masked-out compute units, and can another stream's kernels run there meanwhile?
// (The persistent inflate kernel is meant to fill "its" CUs' wave slots for the whole decode; record stages and the classifier need room beside it.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask_probe tools/cumask_probe.hip && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kThreads>
__global__ __launch_bounds__(kThreads) void occupy(unsigned long long ticks, uint32_t* where, unsigned long long* when) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[2 * blockIdx.x] = hw; where[2 * blockIdx.x + 1] = xcc;
        when[blockIdx.x] = t0;
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    printf("%d CUs, wall clock %d kHz\n", ncu, pr.clockRate);
    int rate_khz = 100000;   // wall_clock64: 100 MHz
    for (int variant = 0; variant < 5; ++variant) {
        // variant 0: the last 16 CUs off; 1: every 16th CU off; 2: the first 16 off
        std::vector<uint32_t> mask((ncu + 31) / 32, 0);
        int on = 0;
        for (int i = 0; i < ncu; ++i) {
            const bool off = variant >= 3 ? false : variant == 0 ? i >= ncu - 16 : variant == 1 ? (i % 16 == 15) : i < 16;   // 3: a full mask; 4: no mask at all
            if (!off) { mask[i / 32] |= 1u << (i % 32); ++on; }
        }
        hipStream_t sm, sb;
        if (variant == 4) CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking)); else CK(hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data()));
        CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        unsigned flags = 99;
        CK(hipStreamGetFlags(sm, &flags));
        const int na = ncu * 32, nb = 64;
        uint32_t *wa, *wb;
        unsigned long long *ta, *tb;
        CK(hipHostMalloc(&wa, na * 8)); CK(hipHostMalloc(&wb, nb * 8)); CK(hipHostMalloc(&ta, na * 8)); CK(hipHostMalloc(&tb, nb * 8));
        for (int i = 0; i < na; ++i) ta[i] = 0;
        const auto h0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(occupy<64>, dim3((variant >= 3 ? 240 : on) * 32), dim3(64), 4096, sm, 20ull * rate_khz, wa, ta);   // 20 ms, 5 KB of LDS per wave like the inflate kernel
        // a moment later: 1024-thread workgroups on the other stream
        while (ta[0] == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() < 1.0) {}
        hipLaunchKernelGGL(occupy<1024>, dim3(nb), dim3(1024), 16384, sb, 1ull * rate_khz, wb, tb);
        CK(hipStreamSynchronize(sb));
        const double tb_done = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        // a blocking call on the null stream while the masked stream is busy: does it wait for it?
        void* dummy;
        CK(hipMalloc(&dummy, 1 << 20));
        char hbuf[64];
        CK(hipMemcpy(hbuf, dummy, 64, hipMemcpyDeviceToHost));
        const double t_null = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        CK(hipStreamSynchronize(sm));
        const double ta_done = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        std::map<uint32_t, int> cus_a, cus_b;
        unsigned long long a_first = ~0ull, a_last = 0;
        for (int i = 0; i < (variant >= 3 ? 240 : on) * 32; ++i) { cus_a[(wa[2 * i + 1] << 16) | ((wa[2 * i] >> 8) & 0xFFF)]++; a_first = std::min(a_first, ta[i]); a_last = std::max(a_last, ta[i]); }
        for (int i = 0; i < nb; ++i) cus_b[(wb[2 * i + 1] << 16) | ((wb[2 * i] >> 8) & 0xFFF)]++;
        int shared = 0;
        for (auto& kv : cus_b) if (cus_a.count(kv.first)) ++shared;
        int per_xcc_a[8] = {0}, per_xcc_b[8] = {0};
        for (auto& kv : cus_a) per_xcc_a[(kv.first >> 16) & 7]++;
        for (auto& kv : cus_b) per_xcc_b[(kv.first >> 16) & 7]++;
        printf("variant %d: %d CUs on; masked stream flags %u; its %d waves ran on %zu distinct CUs (per XCD:", variant, on, flags, on * 32, cus_a.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc_a[x]);
        printf("), all started within %.3f ms; the other stream's %d workgroups of 1024 (first start %.3f ms after the masked kernel's first) ran on %zu CUs (per XCD:", (a_last - a_first) / (double)rate_khz, nb, ((double)tb[0] - (double)a_first) / (double)rate_khz, cus_b.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc_b[x]);
        printf("), %d of them also used by the masked kernel; other stream done after %.2f ms, a blocking null-stream copy returned after %.2f ms, masked kernel done after %.2f ms\n",
               shared, tb_done, t_null, ta_done);
        CK(hipStreamDestroy(sm)); CK(hipStreamDestroy(sb));
    }
    return 0;
}
