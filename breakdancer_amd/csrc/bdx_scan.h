// Device-wide inclusive scan in three launches (block sums -> scan of sums -> rescan with offsets; two when the
// workgroups are few enough to add up the block sums themselves),
// generic over the element type (uint32_t or a 4-column struct) and over load/store functors so the
// callers fuse their own per-element work into phase 1 / phase 3.  Element count may live on the device
// (n_ptr) so no host round trip is needed between dependent stages.
#pragma once
#include "bdx_dev.h"

namespace bdx {

// in-kernel clocks of a measurement build (-DBDX_KPROF, tools/kprof.py): one table per translation unit
#ifdef BDX_KPROF
static __device__ unsigned long long g_kprof[8 * 65536];
#define KPROF(row, col) do { if ((threadIdx.x & 63) == 0 && (row) < 65536u) g_kprof[(size_t)(row) * 8 + (col)] = wall_clock64(); } while (0)
#else
#define KPROF(row, col) do {} while (0)
#endif

struct U4 {
    uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ U4 operator+(const U4& a, const U4& b) { return U4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__host__ __device__ __forceinline__ U4 operator-(const U4& a, const U4& b) { return U4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

template <class T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ uint32_t zero_of<uint32_t>() { return 0u; }
template <> __device__ __forceinline__ U4 zero_of<U4>() { return U4{0, 0, 0, 0}; }

__device__ __forceinline__ uint32_t shfl_up_t(uint32_t v, int o) { return __shfl_up(v, o); }
__device__ __forceinline__ U4 shfl_up_t(const U4& v, int o) {
    return U4{(uint32_t)__shfl_up(v.x, o), (uint32_t)__shfl_up(v.y, o), (uint32_t)__shfl_up(v.z, o), (uint32_t)__shfl_up(v.w, o)};
}

template <class T> __device__ __forceinline__ T wave_incl_scan_t(T v) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T t = shfl_up_t(v, o);
        if (l >= o) v = v + t;
    }
    return v;
}

constexpr int kScanBlock = 256;
constexpr int kScanIters = 2;   // default: 512 elements per workgroup (~300 workgroups at 150 k elements, enough to cover 256 CUs);
                                // scans over few elements with a heavy output functor take 1 (template parameter kIters)

// block-wide inclusive scan of one element per thread; returns the inclusive value, *total = block sum
template <class T> __device__ __forceinline__ T block_incl_scan(T v, T* s_ws /*[4]*/, T* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T inc = wave_incl_scan_t(v);
    if (lane == 63) s_ws[w] = inc;
    __syncthreads();
    T off = zero_of<T>();
    T tot = zero_of<T>();
#pragma unroll
    for (int k = 0; k < kScanBlock / 64; ++k) {
        if (k < w) off = off + s_ws[k];
        tot = tot + s_ws[k];
    }
    __syncthreads();
    *total = tot;
    return off + inc;
}

template <class T, class In, int kIters> __device__ __forceinline__ void scan_phase1_body(const In& in, const uint32_t* n_ptr, T* blk) {
    constexpr int kScanChunk = kScanBlock * kIters;
    __shared__ T s_ws[kScanBlock / 64];
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * kScanChunk;
    if (base >= n) return;
    T acc = zero_of<T>();
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const uint32_t j = base + it * kScanBlock + threadIdx.x;
        if (j < n) acc = acc + in(j, n);
    }
    T tot;
    block_incl_scan(acc, s_ws, &tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

template <class T, class In, int kIters> __global__ __launch_bounds__(kScanBlock) void scan_phase1(In in, const uint32_t* n_ptr, T* blk) {
    scan_phase1_body<T, In, kIters>(in, n_ptr, blk);
}

// phase 1 with one extra workgroup (the last) that runs an independent side job of the caller
template <class T, class In, class Side, int kIters>
__global__ __launch_bounds__(kScanBlock) void scan_phase1_side(In in, Side side, const uint32_t* n_ptr, T* blk) {
    if (blockIdx.x == gridDim.x - 1) side();
    else scan_phase1_body<T, In, kIters>(in, n_ptr, blk);
}

// single workgroup: in-place exclusive scan of the block sums, grand total -> *total
template <class T, int kIters> __global__ __launch_bounds__(1024) void scan_phase2(T* blk, const uint32_t* n_ptr, T* total) {
    constexpr int kScanChunk = kScanBlock * kIters;
    __shared__ T s_ws[16];
    __shared__ T s_carry;
    const uint32_t n = *n_ptr;
    const uint32_t nb = (n + kScanChunk - 1) / kScanChunk;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = zero_of<T>();
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        T v = i < nb ? blk[i] : zero_of<T>();
        T inc = wave_incl_scan_t(v);
        if (lane == 63) s_ws[w] = inc;
        __syncthreads();
        T off = s_carry;
        for (int k = 0; k < w; ++k) off = off + s_ws[k];
        if (i < nb) blk[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// kSelfSum: blk holds phase 1's raw block sums and every workgroup adds up the ones before its own (few workgroups:
// cheaper than the extra launch of phase 2)
template <class T, class In, class Out, int kIters, bool kSelfSum>
__global__ __launch_bounds__(kScanBlock) void scan_phase3(In in, Out out, const uint32_t* n_ptr, const T* blk) {
    constexpr int kScanChunk = kScanBlock * kIters;
    __shared__ T s_ws[kScanBlock / 64];
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * kScanChunk;
    if (base >= n) return;
    T carry;
    if (kSelfSum) {
        T acc = zero_of<T>();
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kScanBlock) acc = acc + blk[i];
        block_incl_scan(acc, s_ws, &carry);
    } else {
        carry = blk[blockIdx.x];
    }
#pragma unroll 1
    for (int it = 0; it < kIters; ++it) {
        const uint32_t j = base + it * kScanBlock + threadIdx.x;
        if (base + it * kScanBlock >= n) break;
        T e = j < n ? in(j, n) : zero_of<T>();
        T tot;
        T inc = carry + block_incl_scan(e, s_ws, &tot);
        if (j < n) out(j, n, inc, e);
        carry = carry + tot;
    }
}

// host helper: grid for an upper bound on the element count
inline uint32_t scan_grid(uint32_t n_upper, int iters = kScanIters) {
    const uint32_t chunk = (uint32_t)(kScanBlock * iters);
    return n_upper ? (n_upper + chunk - 1) / chunk : 1;
}

constexpr uint32_t kScanSelfSumMax = 1024;  // workgroups up to which phase 3 sums the block totals itself

template <class T, int kIters = kScanIters, class In, class Out>
void scan_launch(In in, Out out, const uint32_t* n_ptr, uint32_t n_upper, T* blk_ws, T* total, hipStream_t s) {
    const uint32_t g = scan_grid(n_upper, kIters);
    hipLaunchKernelGGL((scan_phase1<T, In, kIters>), dim3(g), dim3(kScanBlock), 0, s, in, n_ptr, blk_ws);
    if (g <= kScanSelfSumMax) {
        hipLaunchKernelGGL((scan_phase3<T, In, Out, kIters, true>), dim3(g), dim3(kScanBlock), 0, s, in, out, n_ptr, blk_ws);
        return;
    }
    hipLaunchKernelGGL((scan_phase2<T, kIters>), dim3(1), dim3(1024), 0, s, blk_ws, n_ptr, total);
    hipLaunchKernelGGL((scan_phase3<T, In, Out, kIters, false>), dim3(g), dim3(kScanBlock), 0, s, in, out, n_ptr, blk_ws);
}

// ---- the same scan in ONE launch: decoupled look-back ------------------------------------------------------------
// Every workgroup scans its own chunk, publishes the chunk's aggregate, and finds its exclusive prefix by walking back over
// its predecessors' published words until it meets one that already carries an inclusive prefix.  A published word is
// self-contained -- value (32 bits) | run stamp (30 bits) | status (2 bits: 1 aggregate, 2 inclusive prefix) in one 64-bit
// relaxed device-scope atomic -- so no fence is needed (a device-scope fence per workgroup costs an L2 write-back on this
// multi-die part), and the state array never has to be cleared: words of earlier runs carry another stamp.  Columns of a
// multi-column element are independent scans that share the walk: the 64 lanes of the first wave look at 64 / columns
// predecessors at a time (one wave per column, 64 predecessors a step, measured slower: 18.4 against 15.8 us for the head
// scan).  Workgroups wait only for lower-numbered ones, which the dispatcher started earlier.
template <class T> struct ScanCols;
template <> struct ScanCols<uint32_t> {
    static constexpr int n = 1;
    __device__ static __forceinline__ uint32_t get(const uint32_t& v, int) { return v; }
    __device__ static __forceinline__ void set(uint32_t& v, int, uint32_t x) { v = x; }
};
template <> struct ScanCols<U4> {
    static constexpr int n = 4;
    __device__ static __forceinline__ uint32_t get(const U4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
    __device__ static __forceinline__ void set(U4& v, int c, uint32_t x) { if (c == 0) v.x = x; else if (c == 1) v.y = x; else if (c == 2) v.z = x; else v.w = x; }
};

// The walk itself, for kernels that do more with their prefix than one output per element: every thread of workgroup `bid`
// calls it with the workgroup's total; it returns the workgroup's exclusive prefix (the same value in all threads).
template <class T>
__device__ __forceinline__ T lookback_exclusive(const T& tot, unsigned long long* state, uint32_t stamp, uint32_t bid, uint32_t* s_prefix /* LDS [columns] */) {
    constexpr int kCols = ScanCols<T>::n;
    constexpr int kWin = 64 / kCols;  // predecessors per look-back step
    const unsigned long long stamp_bits = (unsigned long long)(stamp & 0x3FFFFFFFu) << 32;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane < kCols)  // own aggregate first (the first workgroup's is its inclusive prefix already)
            __hip_atomic_store(&state[(size_t)bid * kCols + lane],
                               (unsigned long long)ScanCols<T>::get(tot, lane) | stamp_bits | ((bid == 0 ? 2ull : 1ull) << 62), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        const int col = lane / kWin, k = lane % kWin;
        uint32_t acc = 0;     // this lane's column: aggregates walked over so far (the same in all lanes of the column)
        bool frozen = false;  // the column has met an inclusive prefix
        // kAhead windows are requested at once (the chain of dependent round trips through a few hundred workgroups is what the
        // scan costs: 586 workgroups in steps of 16 took 37 of them), then evaluated nearest first
        constexpr int kAhead = 4;
        for (uint32_t back = 0;; back += kWin * kAhead) {
            unsigned long long w[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int64_t pred = (int64_t)bid - 1 - (int64_t)back - (int64_t)u * kWin - k;
                w[u] = (2ull << 62) | stamp_bits;  // past the first workgroup: an inclusive prefix of zero
                if (!frozen && pred >= 0) w[u] = __hip_atomic_load(&state[(size_t)pred * kCols + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool done = false;
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                if (done) break;  // (wave-uniform)
                const int64_t pred = (int64_t)bid - 1 - (int64_t)back - (int64_t)u * kWin - k;
                unsigned long long x = w[u];
                if (!frozen && pred >= 0)
                    while ((x & (0x3FFFFFFFull << 32)) != stamp_bits || (x >> 62) == 0)  // not published yet: ask again
                        x = __hip_atomic_load(&state[(size_t)pred * kCols + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t val = (uint32_t)x, st = (uint32_t)(x >> 62);
                // the nearest predecessor (smallest k) of the column that carries an inclusive prefix ends the column's walk
                const uint64_t incl = __ballot(!frozen && st == 2);
                const uint64_t mine = kWin == 64 ? incl : ((incl >> (col * kWin)) & ((1ull << kWin) - 1));
                const int stop = mine ? __builtin_ctzll(mine) : kWin;
                uint32_t part = (!frozen && k <= stop) ? val : 0u;
#pragma unroll
                for (int o = 1; o < kWin; o <<= 1) part += __shfl_xor(part, o);
                acc += part;
                if (mine) frozen = true;
                if (!__ballot(!frozen)) done = true;
            }
            if (done) break;
        }
        if (k == 0) s_prefix[col] = acc;
    }
    __syncthreads();
    T carry = zero_of<T>();
#pragma unroll
    for (int c = 0; c < kCols; ++c) ScanCols<T>::set(carry, c, s_prefix[c]);
    if (threadIdx.x < kCols && bid != 0)  // inclusive prefix for the successors
        __hip_atomic_store(&state[(size_t)bid * kCols + threadIdx.x],
                           (unsigned long long)(s_prefix[threadIdx.x] + ScanCols<T>::get(tot, threadIdx.x)) | stamp_bits | (2ull << 62), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    return carry;
}

// kIters elements per thread (element it of thread t: base + it * kScanBlock + t): fewer workgroups and fewer steps of the walk, but
// the input functors' dependent loads run one iteration after the other -- measured slower for both of K3's scans (head scan 16 -> 20 us)
template <class T, class In, class Out, int kIters>
__global__ __launch_bounds__(kScanBlock) void scan_lookback(In in, Out out, const uint32_t* n_ptr, unsigned long long* state, uint32_t stamp) {
    __shared__ T s_ws[kScanBlock / 64];
    __shared__ uint32_t s_prefix[ScanCols<T>::n];
    const uint32_t n = *n_ptr;
    const uint32_t bid = blockIdx.x;
    const uint32_t base = bid * kScanBlock * kIters;
    if (base >= n) return;
    constexpr uint32_t kRow0 = ScanCols<T>::n == 4 ? 0u : 32768u;  // (clocks: the 4-column scan and the 1-column scan of a translation unit)
    (void)kRow0;
    KPROF(kRow0 + bid * 4 + (threadIdx.x >> 6), 0);
    T e[kIters], inc[kIters];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {  // (all inputs requested before the first is used)
        const uint32_t j = base + it * kScanBlock + threadIdx.x;
        e[it] = j < n ? in(j, n) : zero_of<T>();
    }
    T tot = zero_of<T>();
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        T t;
        inc[it] = tot + block_incl_scan(e[it], s_ws, &t);
        tot = tot + t;
    }
    KPROF(kRow0 + bid * 4 + (threadIdx.x >> 6), 1);
    const T carry = lookback_exclusive<T>(tot, state, stamp, bid, s_prefix);
    KPROF(kRow0 + bid * 4 + (threadIdx.x >> 6), 2);
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const uint32_t j = base + it * kScanBlock + threadIdx.x;
        if (j < n) out(j, n, carry + inc[it], e[it]);
    }
    KPROF(kRow0 + bid * 4 + (threadIdx.x >> 6), 3);
}

// host side: one launch; state = [scan_grid(n_upper, 1)][columns] 64-bit words (enough for any kIters), zero once at allocation;
// stamp != 0 and different from run to run (words of the previous run are then simply "not there yet")
template <class T, int kIters = 1, class In, class Out>
void scan_launch_lb(In in, Out out, const uint32_t* n_ptr, uint32_t n_upper, unsigned long long* state, uint32_t stamp, hipStream_t s) {
    const uint32_t g = scan_grid(n_upper, kIters);
    hipLaunchKernelGGL((scan_lookback<T, In, Out, kIters>), dim3(g), dim3(kScanBlock), 0, s, in, out, n_ptr, state, stamp);
}

// the same scan with a side job (a device functor run by one extra workgroup of kScanBlock threads during phase 1; its
// results are complete before phase 3's output functor runs)
template <class T, int kIters = kScanIters, class In, class Out, class Side>
void scan_launch_side(In in, Out out, Side side, const uint32_t* n_ptr, uint32_t n_upper, T* blk_ws, T* total, hipStream_t s) {
    const uint32_t g = scan_grid(n_upper, kIters);
    hipLaunchKernelGGL((scan_phase1_side<T, In, Side, kIters>), dim3(g + 1), dim3(kScanBlock), 0, s, in, side, n_ptr, blk_ws);
    if (g <= kScanSelfSumMax) {
        hipLaunchKernelGGL((scan_phase3<T, In, Out, kIters, true>), dim3(g), dim3(kScanBlock), 0, s, in, out, n_ptr, blk_ws);
        return;
    }
    hipLaunchKernelGGL((scan_phase2<T, kIters>), dim3(1), dim3(1024), 0, s, blk_ws, n_ptr, total);
    hipLaunchKernelGGL((scan_phase3<T, In, Out, kIters, false>), dim3(g), dim3(kScanBlock), 0, s, in, out, n_ptr, blk_ws);
}

}  // namespace bdx
