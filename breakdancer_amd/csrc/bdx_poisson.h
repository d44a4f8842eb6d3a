// Device functions of K5, the Poisson upper-tail score of one (SV, library) term -- shared by the stand-alone kernels
// (k5_poisson.hip) and the score kernel of K6, which evaluates the terms of its candidates itself.
//
// Replaces log(cdf(complement(poisson_distribution(lambda), k))) at breakdancer/BreakDancer.cpp:64-65,
// i.e. log P(X > k) = log P(k+1, lambda) with P the regularised lower incomplete gamma function
// (Boost.Math 1.54 poisson.hpp -> gamma_p).  FP64 throughout; see k5_poisson.hip for the series.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bdx {

__device__ __forceinline__ double wave_prod_scan(double v) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o);
        if (l >= o) v *= t;
    }
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// log P(X > k) in three pieces, so that the table kernel of K6 holds ONE copy of the closing expression's exp / log / lgamma bodies (two
// copies of lgamma alone -- one per series flavour -- cost it 116 registers: 290 in all, one wave per SIMD):
//   poisson_wants_series  lambda > 0 and k > 0: the incomplete gamma series (the other cases are closed forms)
//   poisson_series_*      its sum: the lower series when lambda < k + 2, the upper one otherwise
//   poisson_close         exp(-lambda + b log(lambda) - lgamma(b + 1)) x sum with b = k + 1 (lower) or k (upper), then log / log1p
__device__ __forceinline__ bool poisson_wants_series(double lam, int k) { return lam > 0.0 && k != 0; }
__device__ __forceinline__ bool poisson_lower_series(double lam, int k) { return lam < ((double)k + 1.0) + 1.0; }

// the series summed by the whole wave (all lanes call it with the same lambda and k)
__device__ __forceinline__ double poisson_series_wave(double lam, int k, int lane) {
    const double a = (double)k + 1.0;
    double sum = 1.0, carry = 1.0;
    if (poisson_lower_series(lam, k)) {
        for (int c = 0; c < 1 << 20; ++c) {
            const double r = lam / (a + (double)(64 * c + lane + 1));
            const double pr = wave_prod_scan(r);
            const double step = wave_sum(carry * pr);
            const double nsum = sum + step;
            carry *= __shfl(pr, 63);
            if (nsum == sum) break;
            sum = nsum;
        }
    } else {
        for (int c = 0; c < 1 << 20; ++c) {
            const double num = a - (double)(64 * c + lane + 1);  // (a-1) - m + 1 with m = 64c+lane+1
            const double r = num > 0.0 ? num / lam : 0.0;
            const double pr = wave_prod_scan(r);
            const double step = wave_sum(carry * pr);
            const double nsum = sum + step;
            carry *= __shfl(pr, 63);
            if (nsum == sum || carry == 0.0) { sum = nsum; break; }
            sum = nsum;
        }
    }
    return sum;
}

// the series summed serially by one lane (throughput of short series)
__device__ __forceinline__ double poisson_series_lane(double lam, int k) {
    const double a = (double)k + 1.0;
    double sum = 1.0, term = 1.0;
    if (poisson_lower_series(lam, k)) {
        for (int m = 1; m < (1 << 26); ++m) {
            term *= lam / (a + (double)m);
            const double nsum = sum + term;
            if (nsum == sum) break;
            sum = nsum;
        }
    } else {
        for (int m = 1; m < (1 << 26); ++m) {
            const double num = a - (double)m;
            if (!(num > 0.0)) break;
            term *= num / lam;
            const double nsum = sum + term;
            if (nsum == sum) break;
            sum = nsum;
        }
    }
    return sum;
}

// (NOT inlined: the double-precision exp / log / lgamma / log1p bodies need ~170 registers of their own; inlined into K6's table kernel they
// sat on top of its 116 -- 290, one wave per SIMD -- while as a call the kernel stays at 134 and three waves share a SIMD)
__device__ __attribute__((noinline)) double poisson_close(double lam, int k, double sum) {
    if (!(lam > 0.0)) return log(0.0);  // cdf complement of a zero-mean Poisson is 0
    if (k == 0) return log(-expm1(-lam));
    const double a = (double)k + 1.0;
    const bool lower = poisson_lower_series(lam, k);
    const double b = lower ? a : a - 1.0;
    const double pre = exp(-lam + b * log(lam) - lgamma(b + 1.0));
    const double ps = pre * sum;
    return lower ? log(ps) : log1p(-ps);
}

__device__ __forceinline__ double poisson_log_upper_tail(double lam, int k, int lane) {
    return poisson_close(lam, k, poisson_wants_series(lam, k) ? poisson_series_wave(lam, k, lane) : 1.0);
}
__device__ __forceinline__ double poisson_log_upper_tail_lane(double lam, int k) {
    return poisson_close(lam, k, poisson_wants_series(lam, k) ? poisson_series_lane(lam, k) : 1.0);
}

constexpr double kLaneSeriesLimit = 4096.0;  // above: the series is summed by the whole wave

// shared body: `active` lanes hold one term each; all 64 lanes take part in the wave-parallel evaluation of the long ones
__device__ __forceinline__ double poisson_term(double lam, int k, bool active, int lane) {
    const bool series = active && poisson_wants_series(lam, k);
    const bool longs = series && ((double)k > kLaneSeriesLimit || lam > kLaneSeriesLimit);
    double sum = (series && !longs) ? poisson_series_lane(lam, k) : 1.0;
    for (uint64_t mm = __ballot(longs); mm; mm &= mm - 1) {
        const int t = __builtin_ctzll(mm);
        const double r = poisson_series_wave(__shfl(lam, t), __shfl(k, t), lane);
        if (lane == t) sum = r;
    }
    return active ? poisson_close(lam, k, sum) : 0.0;
}

}  // namespace bdx
