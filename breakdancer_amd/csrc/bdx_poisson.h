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

__device__ __forceinline__ double poisson_log_upper_tail(double lam, int k, int lane) {
    double result;
    if (!(lam > 0.0)) {
        result = log(0.0);  // cdf complement of a zero-mean Poisson is 0
    } else if (k == 0) {
        result = log(-expm1(-lam));
    } else {
        const double a = (double)k + 1.0;
        double sum = 1.0, carry = 1.0;
        if (lam < a + 1.0) {
            for (int c = 0; c < 1 << 20; ++c) {
                const double r = lam / (a + (double)(64 * c + lane + 1));
                const double pr = wave_prod_scan(r);
                const double step = wave_sum(carry * pr);
                const double nsum = sum + step;
                carry *= __shfl(pr, 63);
                if (nsum == sum) break;
                sum = nsum;
            }
            const double pre = exp(-lam + a * log(lam) - lgamma(a + 1.0));
            result = log(pre * sum);
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
            const double pre = exp(-lam + (a - 1.0) * log(lam) - lgamma(a));
            result = log1p(-(pre * sum));
        }
    }
    return result;
}

// one term per LANE: the series is summed serially by its lane (experiment: throughput of short series)
__device__ __forceinline__ double poisson_log_upper_tail_lane(double lam, int k) {
    if (!(lam > 0.0)) return log(0.0);
    if (k == 0) return log(-expm1(-lam));
    const double a = (double)k + 1.0;
    if (lam < a + 1.0) {
        double sum = 1.0, term = 1.0;
        for (int m = 1; m < (1 << 26); ++m) {
            term *= lam / (a + (double)m);
            const double nsum = sum + term;
            if (nsum == sum) break;
            sum = nsum;
        }
        const double pre = exp(-lam + a * log(lam) - lgamma(a + 1.0));
        return log(pre * sum);
    }
    double sum = 1.0, term = 1.0;
    for (int m = 1; m < (1 << 26); ++m) {
        const double num = a - (double)m;
        if (!(num > 0.0)) break;
        term *= num / lam;
        const double nsum = sum + term;
        if (nsum == sum) break;
        sum = nsum;
    }
    const double pre = exp(-lam + (a - 1.0) * log(lam) - lgamma(a));
    return log1p(-(pre * sum));
}

constexpr double kLaneSeriesLimit = 4096.0;  // above: the series is summed by the whole wave

// shared body: `active` lanes hold one term each; all 64 lanes take part in the wave-parallel evaluation of the long ones
__device__ __forceinline__ double poisson_term(double lam, int k, bool active, int lane) {
    const bool longs = active && ((double)k > kLaneSeriesLimit || lam > kLaneSeriesLimit);
    double result = (active && !longs) ? poisson_log_upper_tail_lane(lam, k) : 0.0;
    for (uint64_t mm = __ballot(longs); mm; mm &= mm - 1) {
        const int t = __builtin_ctzll(mm);
        const double r = poisson_log_upper_tail(__shfl(lam, t), __shfl(k, t), lane);
        if (lane == t) result = r;
    }
    return result;
}

}  // namespace bdx
