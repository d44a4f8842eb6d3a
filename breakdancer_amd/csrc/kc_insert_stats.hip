// KC -- bam2cfg's per-library insert-size statistics on the device.
//
// Replaces, for `bam2cfg --device` (reference file:line under perl/):
//   bam2cfg.pl:153-166   mean and standard deviation of a library's insert sizes (Statistics::Descriptive: n - 1), the observations
//                        more than five standard deviations above the mean dropped, mean and standard deviation again
//   bam2cfg.pl:180-197   the one-sided deviations: the observations above the mean and those at or below it, each with n - 1
// One thread per library, every sum in the order the script (and host/bam2cfg_main.cpp) adds them up and every operation spelled with
// a round-to-nearest intrinsic: the figures are bit-for-bit the CPU tool's, so the configuration lines are (they print two decimals --
// a reduction tree's last bits could flip one).  A library is at most -n observations (10,000): four passes of that many dependent
// additions, a fraction of a millisecond.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/bdx.h"

namespace {

__global__ __launch_bounds__(64) void kc_insert_stats_kernel(const double* __restrict__ x, const uint32_t* __restrict__ off, int nlibs, bdx_insert_stats* out) {
    const int lib = blockIdx.x * 64 + threadIdx.x;
    if (lib >= nlibs) return;
    const double* v = x + off[lib];
    const uint32_t n = off[lib + 1] - off[lib];
    bdx_insert_stats r{};
    double s = 0.0;
    for (uint32_t i = 0; i < n; ++i) s = __dadd_rn(s, v[i]);
    const double mean0 = n ? __ddiv_rn(s, (double)n) : 0.0;
    double sd0 = 0.0;
    if (n >= 2) {
        double q = 0.0;
        for (uint32_t i = 0; i < n; ++i) { const double d = __dsub_rn(v[i], mean0); q = __dadd_rn(q, __dmul_rn(d, d)); }
        sd0 = __dsqrt_rn(__ddiv_rn(q, (double)(n - 1)));
    }
    r.mean_all = mean0; r.sd_all = sd0;
    const double cut = __dadd_rn(mean0, __dmul_rn(5.0, sd0));
    uint64_t nk = 0;
    s = 0.0;
    for (uint32_t i = 0; i < n; ++i)
        if (!(v[i] > cut)) { s = __dadd_rn(s, v[i]); ++nk; }
    const double mean = nk ? __ddiv_rn(s, (double)nk) : 0.0;
    double sd = 0.0;
    if (nk >= 2) {
        double q = 0.0;
        for (uint32_t i = 0; i < n; ++i)
            if (!(v[i] > cut)) { const double d = __dsub_rn(v[i], mean); q = __dadd_rn(q, __dmul_rn(d, d)); }
        sd = __dsqrt_rn(__ddiv_rn(q, (double)(nk - 1)));
    }
    r.mean = mean; r.sd = sd; r.n_kept = nk;
    double sm = 0.0, sp = 0.0;
    uint64_t nm = 0, np = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (v[i] > cut) continue;
        const double d = __dsub_rn(v[i], mean), dd = __dmul_rn(d, d);
        if (v[i] > mean) { sp = __dadd_rn(sp, dd); ++np; } else { sm = __dadd_rn(sm, dd); ++nm; }
    }
    r.n_minus = nm; r.n_plus = np;
    // (n - 1 of 0 or 1 observations: the script divides by zero or by -0 there too; the caller drops libraries of fewer than 100 observations before it looks)
    r.sd_minus = __dsqrt_rn(__ddiv_rn(sm, (double)((int64_t)nm - 1)));
    r.sd_plus = __dsqrt_rn(__ddiv_rn(sp, (double)((int64_t)np - 1)));
    out[lib] = r;
}

}  // namespace

extern "C" int bdx_insert_size_stats(int device, const double* x, const uint32_t* offsets, int nlibs, bdx_insert_stats* out) {
    if (!x || !offsets || !out || nlibs < 1) return BDX_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return BDX_EHIP;
    const size_t n = offsets[nlibs];
    double* dx = nullptr;
    uint32_t* doff = nullptr;
    bdx_insert_stats* dout = nullptr;
    int rc = BDX_OK;
    if (hipMalloc(&dx, std::max<size_t>(n, 1) * 8) != hipSuccess || hipMalloc(&doff, ((size_t)nlibs + 1) * 4) != hipSuccess ||
        hipMalloc(&dout, (size_t)nlibs * sizeof(bdx_insert_stats)) != hipSuccess)
        rc = BDX_ENOMEM;
    if (rc == BDX_OK && ((n && hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice) != hipSuccess) ||
                         hipMemcpy(doff, offsets, ((size_t)nlibs + 1) * 4, hipMemcpyHostToDevice) != hipSuccess))
        rc = BDX_EHIP;
    if (rc == BDX_OK) {
        hipLaunchKernelGGL(kc_insert_stats_kernel, dim3((nlibs + 63) / 64), dim3(64), 0, nullptr, dx, doff, nlibs, dout);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess ||
            hipMemcpy(out, dout, (size_t)nlibs * sizeof(bdx_insert_stats), hipMemcpyDeviceToHost) != hipSuccess)
            rc = BDX_EHIP;
    }
    if (dx) (void)hipFree(dx);
    if (doff) (void)hipFree(doff);
    if (dout) (void)hipFree(dout);
    return rc;
}
