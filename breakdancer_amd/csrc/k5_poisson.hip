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
#include "bdx_poisson.h"

namespace bdx {

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
    if (true) {   // (one lane per term; the wave-per-term kernels below are kept for the comparison in DESIGN.md)
        hipLaunchKernelGGL(k5_poisson_lane_n_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lambda, k, out, n);
        return;
    }
    hipLaunchKernelGGL(k5_poisson_kernel, dim3((n + 3) / 4), dim3(256), 0, s, lambda, k, out, n);
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
    if (true) {   // (one lane per term; the wave-per-term kernels below are kept for the comparison in DESIGN.md)
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
