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

__global__ __launch_bounds__(256) void k5_poisson_lane_n_kernel(const double* __restrict__ lambda, const int32_t* __restrict__ kk,
                                                                double* __restrict__ out, uint32_t n) {
    const uint32_t item = blockIdx.x * 256 + threadIdx.x;
    const bool active = item < n;
    const double r = poisson_term(active ? lambda[item] : 1.0, active ? kk[item] : 0, active, threadIdx.x & 63);
    if (active) out[item] = r;
}

void launch_k5(const double* lambda, const int32_t* k, double* out, uint32_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k5_poisson_lane_n_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lambda, k, out, n);
}

}  // namespace bdx

// (bdx_warm_up: the HIP runtime loads a translation unit's device code at the first launch of any of its kernels)
__global__ void k5_noop_kernel() {}
namespace bdx { void warm_k5(hipStream_t s) { hipLaunchKernelGGL(k5_noop_kernel, dim3(1), dim3(64), 0, s); } }
