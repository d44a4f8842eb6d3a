// K5 -- Poisson upper-tail score, one wavefront per (SV, library) term.
//
// Replaces log(cdf(complement(poisson_distribution(lambda), k))) at breakdancer/BreakDancer.cpp:64-65,
// i.e. log P(X > k) = log P(k+1, lambda) with P the regularised lower incomplete gamma function
// (Boost.Math 1.54 poisson.hpp -> gamma_p).  FP64 throughout.
//
//   lambda <  k+1 : P = t_a * sum_{m>=0} lambda^m / ((a+1)...(a+m)),  a = k+1, t_a = e^-lambda lambda^a / a!
//   lambda >= k+1 : P = 1 - Q,  Q = t_{a-1} * sum_{m=0}^{a-1} ((a-1)(a-2)...(a-m)) / lambda^m
// The 64 lanes evaluate 64 consecutive terms per step: a multiplicative wave scan of the term ratios gives
// every lane its term, a wave sum reduces the step, and the loop ends when a step no longer changes the sum.
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

void launch_k5(const double* lambda, const int32_t* k, double* out, uint32_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k5_poisson_kernel, dim3((n + 3) / 4), dim3(256), 0, s, lambda, k, out, n);
}

void launch_k5_dev(const double* lambda, const int32_t* k, double* out, double* out2, const uint32_t* n_ptr, uint32_t n_upper,
                   hipStream_t s) {
    if (!n_upper) return;
    const uint32_t g = (n_upper + 3) / 4;
    hipLaunchKernelGGL(k5_poisson_dev_kernel, dim3(g < 2048u ? g : 2048u), dim3(256), 0, s, lambda, k, out, out2, n_ptr);
}

}  // namespace bdx
