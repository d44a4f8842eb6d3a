// K5 -- Poisson upper-tail score of every (SV, library) term.
//
// Replaces log(cdf(complement(poisson_distribution(lambda), k))) at breakdancer/BreakDancer.cpp:64-65,
// i.e. log P(X > k) = log P(k+1, lambda) with P the regularised lower incomplete gamma function
// (Boost.Math 1.54 poisson.hpp -> gamma_p).  FP64 throughout.
//
//   lambda <  k+1 : P = t_a * sum_{m>=0} lambda^m / ((a+1)...(a+m)),  a = k+1, t_a = e^-lambda lambda^a / a!
//   lambda >= k+1 : P = 1 - Q,  Q = t_{a-1} * sum_{m=0}^{a-1} ((a-1)(a-2)...(a-m)) / lambda^m
// One term per lane: the series of a realistic term (a few to a few hundred summands: k = supporting pairs of one
// library, lambda = expected pairs in the regions) is summed serially by its lane -- 6 us for 6.6 k terms, against 18 us
// with one wavefront per term.  A term whose series can run to thousands of summands (k or lambda above 4096) is handed
// to the whole wave afterwards: the 64 lanes evaluate 64 consecutive summands per step (a multiplicative wave scan of
// the ratios gives every lane its summand, a wave sum reduces the step) until a step no longer changes the sum.
#include <cstdlib>

#include "bdx_k3.h"

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

__global__ __launch_bounds__(256) void k5_poisson_kernel(const double* __restrict__ lambda, const int32_t* __restrict__ kk,
                                                         double* __restrict__ out, uint32_t n) {
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= n) return;
    const double result = poisson_log_upper_tail(lambda[item], kk[item], lane);
    if (lane == 0) out[item] = result;
}

// same, with the term count in device memory and the waves striding over the terms
__global__ __launch_bounds__(256) void k5_poisson_dev_kernel(const double* __restrict__ lambda, const int32_t* __restrict__ kk,
                                                             double* __restrict__ out, double* __restrict__ out2,
                                                             const uint32_t* __restrict__ n_ptr) {
    const uint32_t n = *n_ptr;
    const int lane = threadIdx.x & 63;
    for (uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6); item < n; item += gridDim.x * 4) {
        const double result = poisson_log_upper_tail(lambda[item], kk[item], lane);
        if (lane == 0) {
            out[item] = result;
            if (out2) out2[item] = result;
        }
    }
}

__global__ __launch_bounds__(256) void k5_poisson_lane_n_kernel(const double* __restrict__ lambda, const int32_t* __restrict__ kk,
                                                                double* __restrict__ out, uint32_t n);

void launch_k5(const double* lambda, const int32_t* k, double* out, uint32_t n, hipStream_t s) {
    if (!n) return;
    if (!getenv("BDX_K5_WAVE")) {
        hipLaunchKernelGGL(k5_poisson_lane_n_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lambda, k, out, n);
        return;
    }
    hipLaunchKernelGGL(k5_poisson_kernel, dim3((n + 3) / 4), dim3(256), 0, s, lambda, k, out, n);
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

__global__ __launch_bounds__(256) void k5_poisson_lane_n_kernel(const double* __restrict__ lambda, const int32_t* __restrict__ kk,
                                                                double* __restrict__ out, uint32_t n) {
    const uint32_t item = blockIdx.x * 256 + threadIdx.x;
    const bool active = item < n;
    const double r = poisson_term(active ? lambda[item] : 1.0, active ? kk[item] : 0, active, threadIdx.x & 63);
    if (active) out[item] = r;
}

__global__ __launch_bounds__(256) void k5_poisson_lane_kernel(const double* __restrict__ lambda, const int32_t* __restrict__ kk,
                                                              double* __restrict__ out, double* __restrict__ out2,
                                                              const uint32_t* __restrict__ n_ptr) {
    const uint32_t n = *n_ptr;
    for (uint32_t base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {  // (uniform trip count per workgroup)
        const uint32_t item = base + threadIdx.x;
        const bool active = item < n;
        const double r = poisson_term(active ? lambda[item] : 1.0, active ? kk[item] : 0, active, threadIdx.x & 63);
        if (active) {
            out[item] = r;
            if (out2) out2[item] = r;
        }
    }
}

void launch_k5_dev(const double* lambda, const int32_t* k, double* out, double* out2, const uint32_t* n_ptr, uint32_t n_upper,
                   hipStream_t s) {
    if (!getenv("BDX_K5_WAVE")) {
        if (!n_upper) return;
        const uint32_t g = (n_upper + 255) / 256;
        hipLaunchKernelGGL(k5_poisson_lane_kernel, dim3(g < 1024u ? g : 1024u), dim3(256), 0, s, lambda, k, out, out2, n_ptr);
        return;
    }
    if (!n_upper) return;
    const uint32_t g = (n_upper + 3) / 4;
    hipLaunchKernelGGL(k5_poisson_dev_kernel, dim3(g < 2048u ? g : 2048u), dim3(256), 0, s, lambda, k, out, out2, n_ptr);
}

}  // namespace bdx
