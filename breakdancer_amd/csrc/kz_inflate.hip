// KZ -- DEFLATE (RFC 1951) on the GPU: one wavefront inflates one BGZF block.
//
// Stands where the host producer has zlib / host/fast_inflate.cpp (and the reference samtools' bgzf.c:inflate_block behind
// io/BamReader.hpp:62-70).  A BAM of N records is ~N/330 independent deflate streams of <= 64 KiB, tens of thousands for a
// chromosome: enough to fill 256 compute units with one stream per wave, and the file then crosses PCIe compressed.
//
// What a wave does with its 64 lanes.  Huffman decoding is a serial chain -- a symbol's position is known only once its
// predecessor's length is -- so a CPU decoder pays one table look-up latency per symbol.  Here the look-ups of a whole
// 64-bit window run at once: lane j takes the bits at offset (cursor + j) and looks ITS candidate symbol up in the LDS table
// (one ds_read for 64 candidates) -- a literal, or a whole length / distance pair with its extra bits; the chain of real
// tokens is then followed through the lanes' results (the walk: one v_readlane, one v_writelane and four scalar instructions per
// token), the literal lanes on the chain store their bytes at their places, the matches on it are copied in stream order by all
// lanes (lane i copies byte i, i mod distance for overlapping copies; sources beyond the window are fetched from HBM by their
// own lanes, all of a step's at once), and the wave moves on by up to 64 + 35 bits.  Codes longer than the primary table's index
// (rare symbols by construction) are decoded canonically for every length at once -- lane l holds where the codes of length <= l
// end among the 15-bit prefixes: one compare and one ballot -- no second-level tables.
//
// Everything a step touches is in LDS: the compressed input is staged through a 1 KiB ring (512 bytes per refill, one
// coalesced load), the last 1 KiB of output live in a window that literals and near matches never leave (flushed 256 bytes at
// a time, one 4-byte store per lane; a match further back reads HBM if its whole source has been written back, the window otherwise),
// tables are 16 bits per entry; 5.0 KB per wave: 32 waves per CU with 64 vector registers.
//
// What bounds the kernel is the number of INSTRUCTIONS a step issues, of whatever kind: a wave's step is one long dependent
// stream, eight of them share a SIMD, and builds that traded vector instructions for more scalar ones and branches ran slower
// (tools/kz_ab.sh; profiles/r04_inflate_ab.txt, r04_inflate_ab2.txt): round 4 took the walk from ~18 instructions per token to
// 6, the lane predicates from ballots to scalar masks, the long codes from a bit-by-bit loop to one compare -- 4,880 -> 3,420
// cycles per step, 38 -> 57 GB/s of inflated bytes in launches of <= 1.5 GB (43 -> 62 in one launch).  Half the waves
// take 2,900 cycles per step: the rest is the latency of the step's chain (five dependent LDS look-ups, the walk, a round trip to
// memory for the far matches in two steps out of three); leaving the far matches' bytes pending across steps was built and
// measured -- its bookkeeping cost more than the wait it hid.
//
// Tables are built per deflate block by the wave itself: the code lengths read through a 7-bit table of the code-length code
// (19 symbols of <= 7 bits), canonical codes from a per-length scan, table filled symbol by symbol across the lanes.
#include <hip/hip_runtime.h>

#include "bdx_bam_dev.h"

namespace bdx {

namespace {

constexpr int kLB = 9;             // primary literal/length table: 2^9 entries (10 bits resolved 12 % fewer codes' worth of slow path and cost four waves per CU)
constexpr int kDB = 8;             // primary distance table
constexpr int kMaxBits = 15;
// 16-bit table entry: bits 0-3 code length (0: not in the table), 4-5 kind, 6-13 literal byte / length symbol / distance symbol
constexpr uint32_t T_SLOW = 0, T_LIT = 1, T_SYM = 2, T_EOB = 3;
constexpr uint32_t kOB = 1024, kOBM = kOB - 1;      // output window (1 KiB + 64 VGPRs: 32 waves per CU; what lies further back is fetched from HBM, a step's far matches all at once)
constexpr uint32_t kFlush = 256;                    // bytes written to HBM at a time (64 lanes x 4 bytes)
constexpr uint32_t kStepCap = 128;                  // a step's chain ends once it has produced this much: a step writes < 128 + 258 bytes (a power of two: the walk tests it with one AND)
constexpr uint32_t kWalkStop = 1u << 30;            // walk word of a token the tables do not resolve
constexpr uint32_t kWalkExit = 0xC0u | ((kStepCap | (kStepCap << 1)) << 8) | kWalkStop;   // cursor >= 64 | output >= kStepCap (< 4 kStepCap) | unresolved
constexpr uint32_t kNearDist = kOB - (kStepCap + 258) - 64;   // matches up to this distance are copied inside the window: their
                                                    // source cannot be overwritten by anything the step writes (positions are mod kOB)
// The window's invariants.  A step (and the token it stopped at) starts with less than kFlush bytes not yet written back -- KZ_FLUSH runs
// behind every step, BEFORE the token the chain stopped at and before a deflate block ends -- and writes at most kStepMax bytes.
constexpr uint32_t kStepMax = kStepCap - 1 + 258, kPending = kFlush - 1;
static_assert(kPending + kStepMax < kOB, "a step's writes never wrap onto bytes that are not in HBM yet");
static_assert(kNearDist + kStepMax < kOB, "a near match's source is not overwritten by anything its step writes");
// A match further back than kNearDist reads HBM only if its whole source has been written back (source end <= flushed: decided per
// match, not by the distance); one whose source reaches into [flushed, outpos) is served from the window, which still holds it:
static_assert(kPending + 258 + kStepMax < kOB, "a source that ends above `flushed` starts inside the window");
static_assert(kNearDist >= kPending + (kStepCap - 1) + 8, "the far matches of <= 8 bytes fetched by their own lanes lie below `flushed`");
constexpr uint32_t kIB = 1024, kIBM = kIB - 1;      // input ring
constexpr uint32_t kRefill = 512;

__device__ __forceinline__ uint32_t t_len(uint32_t e) { return e & 15u; }
__device__ __forceinline__ uint32_t t_kind(uint32_t e) { return (e >> 4) & 3u; }
__device__ __forceinline__ uint32_t t_value(uint32_t e) { return (e >> 6) & 255u; }
__device__ __forceinline__ uint16_t t_make(uint32_t kind, uint32_t len, uint32_t value) { return (uint16_t)(len | (kind << 4) | (value << 6)); }

// length symbol (0..28 = litlen symbols 257..285) -> base length, extra bits
__device__ __forceinline__ void length_of(uint32_t s, uint32_t* base, uint32_t* extra) {
    if (s < 8) { *base = 3 + s; *extra = 0; }
    else if (s == 28) { *base = 258; *extra = 0; }
    else { const uint32_t x = (s >> 2) - 1; *extra = x; *base = 3 + ((4 + (s & 3)) << x); }
}
__device__ __forceinline__ uint32_t length_extra(uint32_t s) { return (s < 8 || s == 28) ? 0u : (s >> 2) - 1; }
// distance symbol (0..29) -> base distance, extra bits
__device__ __forceinline__ void distance_of(uint32_t s, uint32_t* base, uint32_t* extra) {
    if (s < 4) { *base = 1 + s; *extra = 0; }
    else { const uint32_t x = (s >> 1) - 1; *extra = x; *base = 1 + ((2 + (s & 1)) << x); }
}

struct __attribute__((aligned(16))) Lds {
    uint32_t ibuf[kIB / 4 + 2];   // compressed bytes: position p at p mod 1024; the first 8 bytes once more behind the end (a step's three words need no wrap)
    uint8_t obuf[kOB + 64];    // the last 1 KiB of output: position p at p mod 1024; 64 bytes behind it take masked-off stores
    uint16_t lit[1 << kLB];
    uint16_t dist[1 << kDB];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t lit_cnt[16], dist_cnt[16];
    uint16_t first_code[16], first_index[16];   // of the alphabet being built
    // literal/length alphabet, for codes the primary table does not hold: lit_limit[l] = end of the codes of length <= l among the 15-bit
    // prefixes (first bit on top), lit_cf[l] = first index - first code of length l (mod 2^16): one compare per length, all lengths at once
    uint16_t lit_limit[16], lit_cf[16], dist_limit[16], dist_cf[16];   // (the same for the distance alphabet)
    uint8_t lens[320];         // code lengths: literal/length alphabet, then distances
    // length symbol -> base length | extra bits << 12; distance symbol -> base distance | extra bits << 16 (RFC 1951 3.2.5).  Two
    // look-ups instead of two dozen vector instructions per step: the vector pipe is what this kernel runs out of
    uint16_t lbx[32];
    uint32_t dbx[32];
};

// Lane predicates as scalar masks.  All 64 lanes are active wherever these are used (the step loop's control flow is uniform), so a
// vector compare's result IS the ballot; written as such the compare stays one instruction (the ballot builtin costs a v_cndmask and
// a second compare on this compiler), and a mask goes back to a lane predicate without any (inverse ballot: the mask becomes EXEC).
__device__ __forceinline__ uint64_t mask_eq(uint32_t a, uint32_t b) { uint64_t r; asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint64_t mask_gt(uint32_t a, uint32_t b) { uint64_t r; asm("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint64_t mask_gt_s(uint32_t a, uint32_t sb) { uint64_t r; asm("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(r) : "v"(a), "s"(sb)); return r; }
__device__ __forceinline__ uint64_t mask_le_s(uint32_t a, uint32_t sb) { uint64_t r; asm("v_cmp_le_u32_e64 %0, %1, %2" : "=s"(r) : "v"(a), "s"(sb)); return r; }   // (a constant in a scalar register: as a "v" operand it cost a v_mov per step)
__device__ __forceinline__ bool lanes_of(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// LDS accesses of a wave are executed in program order; this only keeps the compiler from moving them across
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// 64 bits of the stream starting at bit p, from the input ring (the caller has made sure the ring holds them)
__device__ __forceinline__ uint64_t ring_peek(const uint32_t* ibuf, uint32_t p) {
    const uint32_t byte = p >> 3;
    const uint32_t i0 = (byte >> 2) & (kIB / 4 - 1);
    const uint32_t d0 = ibuf[i0], d1 = ibuf[(i0 + 1) & (kIB / 4 - 1)], d2 = ibuf[(i0 + 2) & (kIB / 4 - 1)];
    const uint32_t sh = ((byte & 3u) << 3) | (p & 7u);   // 0..31
    const uint64_t lo = ((uint64_t)d1 << 32) | d0;
    return (lo >> sh) | (((uint64_t)d2 << 1) << (63 - sh));
}

// bytes [from, from + 512) of the member's payload into the ring (from is a multiple of 512; what lies behind the payload
// -- the footer, the next member, the buffer's padding -- is loaded too and never looked at)
// the same for the step loop: the two halves straight from v_alignbit (which shifts by the low five bits of p itself)
__device__ __forceinline__ void ring_peek2(const uint32_t* ibuf, uint32_t p, uint32_t& lo, uint32_t& hi) {
    const uint32_t i0 = (p >> 5) & (kIB / 4 - 1);
    const uint32_t d0 = ibuf[i0], d1 = ibuf[i0 + 1], d2 = ibuf[i0 + 2];   // (ibuf[256], ibuf[257] mirror ibuf[0], ibuf[1]: ring_load)
    lo = __builtin_amdgcn_alignbit(d1, d0, p);
    hi = __builtin_amdgcn_alignbit(d2, d1, p);
}

__device__ __forceinline__ void ring_load(uint32_t* ibuf, const uint8_t* in, uint32_t from, uint32_t lane) {
    uint64_t v;
    __builtin_memcpy(&v, in + from + 8u * lane, 8);
    const uint32_t at = (from + 8u * lane) & kIBM;
    *(uint64_t*)((uint8_t*)ibuf + at) = v;
    if (at == 0) *(uint64_t*)((uint8_t*)ibuf + kIB) = v;   // the mirror of the ring's first 8 bytes
}

__device__ __forceinline__ uint32_t bitrev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32 - len); }

// lens[0, n) -> cnt[16], sorted[], and (tbits > 0) the primary table.  All 64 lanes.  An incomplete code is accepted only
// if it consists of one code of length 1 (zlib's rule, inflate_table: "max != 1"), and never for the code-length code.
// kAlphabet: 0 literal/length, 1 distance, 2 code lengths.  Returns false for an over-subscribed or incomplete set of lengths.
template <int kAlphabet>
__device__ bool build_tables(Lds& L, const uint8_t* lens, uint32_t n, uint16_t* table, int tbits, uint16_t* cnt, uint16_t* sorted) {
    const uint32_t lane = threadIdx.x;
    // lane l counts the codes of length l
    uint32_t mine = 0;
    if (lane >= 1 && lane <= (uint32_t)kMaxBits)
        for (uint32_t s = 0; s < n; ++s) mine += lens[s] == lane;
    if (lane < 16) cnt[lane] = (uint16_t)(lane == 0 ? 0 : mine);
    __syncthreads();
    // Kraft sum, first code and first index of every length (the same in all lanes)
    int left = 1;
    uint32_t first_code = 0, first_index = 0, my_index = 0, used = 0, maxlen = 0;
    bool over = false;
    for (uint32_t l = 1; l <= (uint32_t)kMaxBits; ++l) {
        const uint32_t c = cnt[l];
        left <<= 1;
        left -= (int)c;
        if (left < 0) over = true;
        if (l == lane) { my_index = first_index; L.first_code[l] = (uint16_t)first_code; L.first_index[l] = (uint16_t)first_index; }
        if (kAlphabet != 2 && l == lane) {
            (kAlphabet == 0 ? L.lit_limit : L.dist_limit)[l] = (uint16_t)((first_code + c) << (kMaxBits - l));
            (kAlphabet == 0 ? L.lit_cf : L.dist_cf)[l] = (uint16_t)(first_index - first_code);
        }
        first_code = (first_code + c) << 1;
        first_index += c;
        used += c;
        if (c) maxlen = l;
    }
    if (over) return false;
    if (left > 0 && (kAlphabet == 2 || !(used <= 1 && maxlen <= 1))) return false;   // incomplete
    if (used == 0) {  // (an alphabet without codes is legal; meeting one of its symbols is the error)
        if (tbits > 0)
            for (uint32_t i = lane; i < (1u << tbits); i += 64) table[i] = t_make(T_SLOW, 0, 0);
        __syncthreads();
        return true;
    }
    // lane l: symbols of length l in symbol order -> sorted[]
    if (lane >= 1 && lane <= (uint32_t)kMaxBits && mine) {
        uint32_t k = 0;
        for (uint32_t s = 0; s < n; ++s)
            if (lens[s] == lane) sorted[my_index + k++] = (uint16_t)s;
    }
    if (tbits > 0) {
        const uint32_t tsize = 1u << tbits;
        for (uint32_t i = lane; i < tsize; i += 64) table[i] = t_make(T_SLOW, 0, 0);   // longer codes: canonical decode
        __syncthreads();
        // entry idx of the sorted list is the symbol with code first_code[l] + (idx - first_index[l])
        for (uint32_t idx = lane; idx < used; idx += 64) {
            const uint32_t s = sorted[idx];
            const uint32_t l = lens[s];
            if (l > (uint32_t)tbits) continue;
            uint16_t e;
            if (kAlphabet == 0) {
                if (s < 256) e = t_make(T_LIT, l, s);
                else if (s == 256) e = t_make(T_EOB, l, 0);
                else if (s <= 285) e = t_make(T_SYM, l, s - 257);
                else continue;   // 286 / 287: not valid symbols; left to the canonical decoder, which reports them
            } else {
                if (s > 29) continue;
                e = t_make(T_SYM, l, s);
            }
            const uint32_t code = (uint32_t)L.first_code[l] + (idx - (uint32_t)L.first_index[l]);
            for (uint32_t i = bitrev(code, l); i < tsize; i += 1u << l) table[i] = e;
        }
    }
    __syncthreads();
    return true;
}

// kProf: the measurement build (its counters cost every step half a dozen instructions: they live in spilled scalar registers)
template <bool kProf>
__global__ __launch_bounds__(64, 8) void kz_inflate_kernel(const uint8_t* __restrict__ in_all, const BgzfBlock* __restrict__ blocks, uint32_t nblk,
                                                        uint8_t* out_all, uint32_t* __restrict__ status, unsigned long long* __restrict__ prof_) {
    unsigned long long* const prof = kProf ? prof_ : nullptr;
    __shared__ Lds L;
    // measurement hook (BDX_KZ_PROF, tools/bamdec_probe.py): per member {cycles in all, in headers + tables, steps, matches, slow codes, deflate blocks, far matches fetched by their lanes, far matches copied from HBM by the wave}
    unsigned long long t_begin = 0, t_tables = 0;
    uint32_t n_steps = 0, n_match = 0, n_slow = 0, n_dblk = 0, n_far_lane = 0, n_far_loop = 0;   // (matches: all / short far ones fetched from HBM by their own lanes / far ones copied from HBM by the wave)
    if (prof) t_begin = __builtin_readcyclecounter();
    const uint32_t b = blockIdx.x;
    if (b >= nblk) return;
    const uint32_t lane = threadIdx.x;
    const BgzfBlock blk = blocks[b];
    const uint8_t* in = in_all + blk.in_off;
    uint8_t* out = out_all + blk.out_off;
    const uint32_t clen = blk.in_len, ulen = blk.out_len;
    const uint32_t bit_limit = clen * 8u;
    uint32_t bitpos = 0, outpos = 0, err = KZ_OK;
    uint32_t flushed = 0;      // output bytes already in HBM; [flushed, outpos) sit in the LDS window only
    uint32_t loaded_end = 0;   // the input ring holds the bytes [loaded_end - 1024, loaded_end) (those it has loaded)
    bool last = false;
    // A SIMD serves its oldest wave first: left alone, the eight members of a SIMD finish one after the other and the SIMD runs emptier
    // and emptier (a step is latency-bound: fewer waves, less throughput) -- a launch of one round of the wave slots lasted 24 M cycles
    // for members that take 18.8 M when the slots stay full (profiles/r04_inflate_member_clocks.txt).  So a wave gives way as it gets
    // ahead: priority 3 for its first 16 KiB of output, 2, 1, 0 for the next ones (s_setprio ranks above age).
    uint32_t prio = 3;
    asm volatile("s_setprio 3");
    if (lane < 32) {
        const uint32_t lx0 = length_extra(lane);
        L.lbx[lane] = (uint16_t)((lane < 8 ? 3 + lane : (lane == 28 ? 258u : 3 + ((4 + (lane & 3)) << lx0))) | (lx0 << 12));
        uint32_t dbase0, dx0;
        distance_of(lane, &dbase0, &dx0);
        L.dbx[lane] = dbase0 | (dx0 << 16);
    }
    __syncthreads();

    // the ring holds everything up to 96 bytes behind byte `at`
#define KZ_ENSURE(at)                                                                   \
    do {                                                                                \
        while ((at) + 96u > loaded_end) { ring_load(L.ibuf, in, loaded_end, lane); loaded_end += kRefill; } \
        lds_order();                                                                    \
    } while (0)
    // [flushed, flushed + 256) of the window to HBM
#define KZ_FLUSH()                                                                      \
    do {                                                                                \
        lds_order();                                                                    \
        if (outpos - flushed >= kFlush) {                                               \
            const uint32_t want_ = 3u - (outpos >> 14 > 3u ? 3u : outpos >> 14);        \
            if (want_ != prio) {                                                        \
                prio = want_;                                                           \
                if (want_ == 2) asm volatile("s_setprio 2");                            \
                else if (want_ == 1) asm volatile("s_setprio 1");                       \
                else asm volatile("s_setprio 0");                                       \
            }                                                                           \
        }                                                                               \
        while (outpos - flushed >= kFlush) {                                            \
            const uint32_t at_ = flushed + 4u * lane;                                   \
            uint32_t v_;                                                                \
            if ((flushed & 3u) == 0) v_ = *(const uint32_t*)(L.obuf + (at_ & kOBM));    \
            else v_ = (uint32_t)L.obuf[at_ & kOBM] | ((uint32_t)L.obuf[(at_ + 1) & kOBM] << 8) | ((uint32_t)L.obuf[(at_ + 2) & kOBM] << 16) | \
                      ((uint32_t)L.obuf[(at_ + 3) & kOBM] << 24);                       \
            __builtin_memcpy(out + at_, &v_, 4);                                        \
            flushed += kFlush;                                                          \
        }                                                                               \
    } while (0)

    while (!last && err == KZ_OK) {
        const unsigned long long t_hdr = prof ? __builtin_readcyclecounter() : 0;
        if (kProf) ++n_dblk;
        if (bitpos + 3 > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
        KZ_ENSURE(bitpos >> 3);
        uint64_t w = ring_peek(L.ibuf, bitpos);
        // (header fields are the same in all lanes; saying so keeps the loops they bound scalar)
        last = uni((uint32_t)w & 1u) != 0;
        const uint32_t type = uni((uint32_t)(w >> 1) & 3u);
        bitpos += 3;
        if (type == 0) {  // stored: to the byte boundary, LEN / NLEN, raw bytes (read from HBM directly)
            bitpos = (bitpos + 7u) & ~7u;
            const uint32_t at = bitpos >> 3;
            if (at + 4 > clen) { err = KZ_BAD_STORED; break; }
            const uint32_t len = uni(in[at] | ((uint32_t)in[at + 1] << 8)), nlen = uni(in[at + 2] | ((uint32_t)in[at + 3] << 8));
            if ((len ^ 0xFFFFu) != nlen || at + 4 + len > clen) { err = KZ_BAD_STORED; break; }
            if (len > ulen - outpos) { err = KZ_OUTPUT_OVERRUN; break; }
            lds_order();
            for (uint32_t i = flushed + lane; i < outpos; i += 64) out[i] = L.obuf[i & kOBM];   // what the window still holds
            for (uint32_t i = lane; i < len; i += 64) out[outpos + i] = in[at + 4 + i];
            for (uint32_t i = (len > kOB ? len - kOB : 0) + lane; i < len; i += 64) L.obuf[(outpos + i) & kOBM] = in[at + 4 + i];   // the window follows
            lds_order();
            outpos += len;
            flushed = outpos;
            bitpos = (at + 4 + len) * 8u;
            loaded_end = (bitpos >> 3) & ~(kRefill - 1);   // the ring starts over behind the stored bytes
            continue;
        }
        if (type == 3) { err = KZ_BAD_BLOCK_TYPE; break; }
        uint32_t hlit, hdist;
        if (type == 1) {  // fixed code
            for (uint32_t s = lane; s < 288; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) L.lens[288 + lane] = 5;   // (30 and 31 never occur in valid data)
            hlit = 288; hdist = 32;
            __syncthreads();
        } else {
            w >>= 3;
            hlit = uni((uint32_t)(w & 31) + 257);
            hdist = uni((uint32_t)((w >> 5) & 31) + 1);
            const uint32_t hclen = uni((uint32_t)((w >> 10) & 15) + 4);
            bitpos += 14;
            if (hlit > 286 || hdist > 30) { err = KZ_BAD_LENGTHS; break; }
            // the code-length code: 19 x 3 bits in a fixed order
            if (lane < 19) L.lens[lane] = 0;
            __syncthreads();
            {
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                KZ_ENSURE((bitpos >> 3) + 8u);
                if (lane < hclen) L.lens[order[lane]] = (uint8_t)(ring_peek(L.ibuf, bitpos + 3u * lane) & 7u);
            }
            bitpos += 3u * hclen;
            __syncthreads();
            // (the code-length code's 7-bit table, its counts and its sorted symbols live where the distance alphabet's will be built)
            if (!build_tables<2>(L, L.lens, 19, L.dist, 7, L.dist_cnt, L.dist_sorted)) { err = KZ_BAD_LENGTHS; break; }
            // the two alphabets' code lengths, run-length coded (every lane runs the same sequence)
            const uint32_t total = hlit + hdist;
            uint32_t n = 0, prev = 0;
            while (n < total) {
                if (bitpos > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
                KZ_ENSURE(bitpos >> 3);
                const uint64_t ww = ring_peek(L.ibuf, bitpos);
                const uint32_t pe = uni(L.dist[(uint32_t)ww & 127u]);   // (codes of at most 7 bits: one look-up)
                const uint32_t l = t_len(pe), sym = t_value(pe);
                if (l == 0) { err = KZ_BAD_LENGTHS; break; }
                uint32_t used = l, rep = 1, val = sym;
                if (sym == 16) {
                    if (n == 0) { err = KZ_BAD_LENGTHS; break; }
                    val = prev;
                    rep = 3 + (uint32_t)((ww >> l) & 3);
                    used += 2;
                } else if (sym == 17) {
                    val = 0;
                    rep = 3 + (uint32_t)((ww >> l) & 7);
                    used += 3;
                } else if (sym == 18) {
                    val = 0;
                    rep = 11 + (uint32_t)((ww >> l) & 127);
                    used += 7;
                }
                rep = uni(rep);
                if (n + rep > total) { err = KZ_BAD_LENGTHS; break; }
                if (lane < rep) L.lens[n + lane] = (uint8_t)val;           // rep <= 138: up to three rounds
                if (lane + 64 < rep) L.lens[n + 64 + lane] = (uint8_t)val;
                if (lane + 128 < rep) L.lens[n + 128 + lane] = (uint8_t)val;
                n += rep;
                prev = val;
                bitpos += used;
            }
            if (err != KZ_OK) break;
            __syncthreads();
            if (L.lens[256] == 0) { err = KZ_BAD_LENGTHS; break; }   // no end-of-block code
        }
        if (!build_tables<0>(L, L.lens, hlit, L.lit, kLB, L.lit_cnt, L.lit_sorted)) { err = KZ_BAD_LENGTHS; break; }
        if (!build_tables<1>(L, L.lens + hlit, hdist, L.dist, kDB, L.dist_cnt, L.dist_sorted)) { err = KZ_BAD_LENGTHS; break; }
        if (prof) t_tables += __builtin_readcyclecounter() - t_hdr;

        // ---- the block's symbols ----
        for (;;) {
            if (kProf) ++n_steps;
            if (bitpos > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
            KZ_ENSURE(bitpos >> 3);
            // Every lane decodes the token that would start at its bit: a literal, or a whole length/distance pair (length
            // code, its extra bits, the distance code from the table, its extra bits -- 36 bits at most, all within the 64 the
            // lane holds).  Tokens the tables do not resolve (codes longer than their index, the end-of-block code) stop the chain.
            uint32_t w_lo, w_hi;
            ring_peek2(L.ibuf, bitpos + lane, w_lo, w_hi);
            const uint32_t e = L.lit[w_lo & ((1u << kLB) - 1)];
            const uint32_t l1 = t_len(e), ek = t_kind(e), ev = t_value(e);
            const uint32_t lb = L.lbx[ev & 31u];
            const uint32_t lx = lb >> 12, lbase = lb & 0xFFFu;
            const uint32_t a1 = __builtin_amdgcn_alignbit(w_hi, w_lo, l1);            // the bits behind the length code (l1 <= 10)
            const uint32_t mlen = lbase + (a1 & ((1u << lx) - 1));
            const uint32_t de = L.dist[(a1 >> lx) & ((1u << kDB) - 1)];
            const uint32_t dl = t_len(de);
            const uint32_t db = L.dbx[t_value(de) & 31u];
            const uint32_t dx = db >> 16, dbase = db & 0xFFFFu;
            const uint32_t a2 = __builtin_amdgcn_alignbit(w_hi, w_lo, l1 + lx + dl);  // the distance's extra bits (l1 + lx + dl <= 23)
            const uint32_t mdist = dbase + (a2 & ((1u << dx) - 1));
            const uint64_t m_lit = mask_eq(ek, T_LIT), m_match = mask_eq(ek, T_SYM) & mask_eq(t_kind(de), T_SYM);
            const bool is_lit = lanes_of(m_lit);
            const bool is_match = lanes_of(m_match);
            const uint32_t olen = is_lit ? 1u : mlen;                                   // bytes the token produces
            // matches of the usual kind -- at most 64 bytes, source not overlapping the destination, inside the window -- are copied
            // without a branch; pk carries what the copy needs in one register
            const bool easy = is_match && mlen <= 64 && mdist >= mlen && mdist <= kNearDist;
            const uint32_t pk = mdist | (mlen << 16) | (easy ? 0x80000000u : 0u);
            // The chain of tokens that starts at the cursor, followed through the lanes' results.  What bounds this kernel is the number of
            // INSTRUCTIONS a step issues, of whatever kind (each of a CU's four SIMDs takes one instruction per wave every fourth cycle and
            // shares its fetch with 31 other waves that are somewhere else in this loop; builds that traded vector instructions for more
            // scalar ones ran slower: tools/kz_ab.sh, profiles/r04_inflate_ab.txt), so the walk is six instructions per token: one register
            // per lane holds the token's bits in its low byte and its output length above it (a token the tables do not resolve: bit 30
            // alone), and ONE scalar word carries cursor and output count the same way -- the token's register is added to it, and one
            // AND tells whether the walk goes on: not once the cursor has left the 64-bit window, kStepCap bytes have been produced or
            // the chain met a token the tables do not resolve (the step then stops in front of it).  The word as it stands before a token
            // is left in the token's lane (v_writelane): where its output begins.  v_readlane, v_writelane and s_bitset1 take the
            // cursor from the word's low six bits as it is.  Four tokens per turn of the loop: three of four branches fall through (63.4 against
            // 62.7 GB/s; eight per turn the same).
            const uint32_t walk = lanes_of(m_lit | m_match) ? ((is_lit ? l1 : l1 + lx + dl + dx) | (olen << 8)) : kWalkStop;
            uint32_t co = 0, wtok, wtest, offs = 0;
            uint64_t mask = 0;
            asm volatile(
                "1:\n\t"
                "v_readlane_b32 %[w], %[walk], %[co]\n\t"
                "v_writelane_b32 %[offs], %[co], %[co]\n\t"
                "s_bitset1_b64 %[mask], %[co]\n\t"
                "s_add_u32 %[co], %[co], %[w]\n\t"
                "s_and_b32 %[t], %[co], %[exit]\n\t"
                "s_cbranch_scc1 2f\n\t"
                "v_readlane_b32 %[w], %[walk], %[co]\n\t"
                "v_writelane_b32 %[offs], %[co], %[co]\n\t"
                "s_bitset1_b64 %[mask], %[co]\n\t"
                "s_add_u32 %[co], %[co], %[w]\n\t"
                "s_and_b32 %[t], %[co], %[exit]\n\t"
                "s_cbranch_scc1 2f\n\t"
                "v_readlane_b32 %[w], %[walk], %[co]\n\t"
                "v_writelane_b32 %[offs], %[co], %[co]\n\t"
                "s_bitset1_b64 %[mask], %[co]\n\t"
                "s_add_u32 %[co], %[co], %[w]\n\t"
                "s_and_b32 %[t], %[co], %[exit]\n\t"
                "s_cbranch_scc1 2f\n\t"
                "v_readlane_b32 %[w], %[walk], %[co]\n\t"
                "v_writelane_b32 %[offs], %[co], %[co]\n\t"
                "s_bitset1_b64 %[mask], %[co]\n\t"
                "s_add_u32 %[co], %[co], %[w]\n\t"
                "s_and_b32 %[t], %[co], %[exit]\n\t"
                "s_cbranch_scc0 1b\n\t"
                "2:"
                : [w] "=&s"(wtok), [t] "=&s"(wtest), [mask] "+s"(mask), [co] "+s"(co), [offs] "+v"(offs)
                : [walk] "v"(walk), [exit] "s"(kWalkExit)
                : "scc");
            const bool stopped = (co & kWalkStop) != 0;
            mask ^= (uint64_t)((co >> 30) & 1u) << (co & 63u);   // (the unresolved token was marked before its register was seen; its bits and bytes are 0)
            const uint32_t o = (co >> 8) & 0x1FFu, cur = co & 0xFFu;
            // where each taken token's output begins, relative to outpos: an exclusive prefix sum of the output lengths over the chain
            const int offv = (int)((offs >> 8) & 0x1FFu);   // (the walk left its word -- cursor and output count before the token -- in the token's lane)
            const uint32_t pos = cur;
            if (mask) {
                if (o > ulen - outpos) { err = KZ_OUTPUT_OVERRUN; break; }
                if (lanes_of(mask & m_lit)) L.obuf[(outpos + (uint32_t)offv) & kOBM] = (uint8_t)ev;
                // the matches, in stream order (a later one may copy what an earlier one, or a literal of this step, produced)
                uint64_t mm = mask & m_match;
                if (mm) {   // (one step in six takes literals only: it skips the distances' arithmetic and the far-match tests)
                if (mm & mask_gt(mdist, outpos + (uint32_t)offv)) { err = KZ_BAD_DISTANCE; break; }   // a source before the member's first byte
                if (kProf) n_match += (uint32_t)__builtin_popcountll(mm);
                lds_order();
                // Far matches first, all at once.  A source beyond the window is in HBM (written back at least two flushes ago) and depends on
                // nothing this step produces, while in the loop below every far match costs the wave a round trip to memory of its own --
                // and two matches in three of configs[1]'s members are far (level-1 deflate of random fields: distances all over the 32 KiB
                // window, tools/deflate_tokens.py).  So every lane whose token is a far match of at most 8 bytes fetches its own source -- ONE
                // load instruction for all of them -- and stores the bytes itself; the loop keeps the rest.  (31.9 -> 35.5 GB/s on the 49 k-member
                // file.  The same for near matches whose source ends before the step's first byte was measured too: 33.8 -- a single LDS
                // round trip saved does not pay for the instructions every step then carries.)
                {
                    const uint32_t dstf = outpos + (uint32_t)offv;
                    const uint64_t fmask = mm & mask_gt_s(mdist, kNearDist) & mask_le_s(mlen, 8u) & mask_le_s(dstf & kOBM, kOB - 8u);
                    const bool farm = lanes_of(fmask);
                    if (kProf) n_far_lane += (uint32_t)__builtin_popcountll(fmask);
                    if (fmask) {
                        if (farm) {
                            uint64_t v;
                            const uint32_t srco = dstf - mdist;   // (offset in the member's output, below 64 Ki: the member's base stays in scalar registers)
                            // (behind the write-backs issued so far; past this CU's L1, which may hold the line from before them)
                            asm volatile("s_waitcnt vmcnt(0)\n\tglobal_load_dwordx2 %0, %1, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(srco), "s"(out) : "memory");
                            uint8_t* q = L.obuf + (dstf & kOBM);
                            if (mlen == 8) {
                                __builtin_memcpy(q, &v, 8);
                            } else {
                                if (mlen & 4) { const uint32_t x = (uint32_t)v; __builtin_memcpy(q, &x, 4); q += 4; v >>= 32; }
                                if (mlen & 2) { const uint16_t x = (uint16_t)v; __builtin_memcpy(q, &x, 2); q += 2; v >>= 16; }
                                if (mlen & 1) *q = (uint8_t)v;
                            }
                        }
                        mm &= ~fmask;
                        lds_order();
                    }
                }
                while (mm) {
                    const uint32_t m = (uint32_t)__builtin_ctzll(mm);
                    mm &= mm - 1;
                    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)m);
                    const uint32_t dst = outpos + (uint32_t)__builtin_amdgcn_readlane(offv, (int)m);
                    const uint32_t length = (k >> 16) & 0x1FFu, dist = k & 0xFFFFu;
                    if (k >> 31) {
                        const uint8_t v = L.obuf[(dst - dist + lane) & kOBM];
                        L.obuf[lane < length ? ((dst + lane) & kOBM) : kOB + lane] = v;   // (lanes beyond the match write to a dump area)
                    } else if (dist <= kNearDist || dst - dist + length > flushed) {   // inside the window (the second case: a source that is not in HBM yet)
                        if (dist >= length) {
                            for (uint32_t i = lane; i < length; i += 64) L.obuf[(dst + i) & kOBM] = L.obuf[(dst - dist + i) & kOBM];
                        } else {
                            for (uint32_t i = lane; i < length; i += 64) L.obuf[(dst + i) & kOBM] = L.obuf[(dst - dist + i % dist) & kOBM];
                        }
                    } else {
                        if (kProf) ++n_far_loop;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        const uint8_t* src = out + dst - dist;
                        for (uint32_t i = lane; i < length; i += 64)
                            L.obuf[(dst + i) & kOBM] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    lds_order();
                }
                }
                outpos += o;
            }
            bitpos += pos;
            KZ_FLUSH();   // (before the token the chain stopped at too: that one, and the next deflate block's first step, count on < kFlush pending bytes)
            if (!stopped) continue;
            // the symbol the chain stopped at, with the bits (and the distance entry) its lane holds
            uint64_t ws = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)w_hi, (int)pos) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane((int)w_lo, (int)pos);
            const uint32_t se = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)pos);
            uint32_t sde = (uint32_t)__builtin_amdgcn_readlane((int)de, (int)pos);
            uint32_t kind = t_kind(se), used = t_len(se), lsym = t_value(se);
            if (kind == T_SLOW) {  // a code longer than the table's index (or none at all)
                if (kProf) ++n_slow;
                // canonical decoding, every length at once: lane l holds where the codes of length <= l end among the 15-bit prefixes
                const uint32_t code15 = __builtin_bitreverse32((uint32_t)ws) >> 17;   // the next 15 bits, the first on top
                const uint64_t shorter = mask_gt_s(L.lit_limit[lane & 15u], code15) & 0xFFFEull;
                if (!shorter) { err = KZ_BAD_CODE; break; }
                const uint32_t l = (uint32_t)__builtin_ctzll(shorter);
                const uint32_t sym = uni(L.lit_sorted[((code15 >> (kMaxBits - l)) + L.lit_cf[l]) & 0xFFFFu]);
                used = l;
                if (sym < 256) {
                    if (outpos >= ulen) { err = KZ_OUTPUT_OVERRUN; break; }
                    if (lane == 0) L.obuf[outpos & kOBM] = (uint8_t)sym;
                    ++outpos;
                    bitpos += used;
                    KZ_FLUSH();
                    continue;
                }
                if (sym == 256) kind = T_EOB;
                else if (sym <= 285) {
                    kind = T_SYM;
                    lsym = sym - 257;
                    sde = L.dist[(uint32_t)(ws >> (used + length_extra(lsym))) & ((1u << kDB) - 1)];
                    sde = uni(sde);
                } else { err = KZ_BAD_CODE; break; }
            }
            ws >>= used;
            if (kind == T_EOB) { bitpos += used; break; }
            if (kProf) ++n_match;
            uint32_t sbase, sx;
            length_of(lsym, &sbase, &sx);
            const uint32_t length = sbase + ((uint32_t)ws & ((1u << sx) - 1));
            ws >>= sx;
            used += sx;
            uint32_t dsym = t_value(sde), sdl = t_len(sde);
            if (t_kind(sde) != T_SYM) {
                const uint32_t dcode15 = __builtin_bitreverse32((uint32_t)ws) >> 17;
                const uint64_t dshorter = mask_gt_s(L.dist_limit[lane & 15u], dcode15) & 0xFFFEull;
                if (!dshorter) { err = KZ_BAD_CODE; break; }
                sdl = (uint32_t)__builtin_ctzll(dshorter);
                dsym = uni(L.dist_sorted[(((dcode15 >> (kMaxBits - sdl)) + L.dist_cf[sdl]) & 0xFFFFu) & 31u]);
                if (dsym > 29) { err = KZ_BAD_CODE; break; }
            }
            ws >>= sdl;
            used += sdl;
            uint32_t sdbase, sdx;
            distance_of(dsym, &sdbase, &sdx);
            const uint32_t dist = sdbase + ((uint32_t)ws & ((1u << sdx) - 1));
            used += sdx;
            bitpos += used;
            if (dist > outpos) { err = KZ_BAD_DISTANCE; break; }
            if (length > ulen - outpos) { err = KZ_OUTPUT_OVERRUN; break; }
            lds_order();
            if (dist <= kNearDist || outpos - dist + length > flushed) {   // inside the window (sources lie below outpos, destinations at or above it: no overlap)
                if (dist >= length) {
                    for (uint32_t i = lane; i < length; i += 64) L.obuf[(outpos + i) & kOBM] = L.obuf[(outpos - dist + i) & kOBM];
                } else {  // overlapping: the pattern of `dist` bytes repeats
                    for (uint32_t i = lane; i < length; i += 64) L.obuf[(outpos + i) & kOBM] = L.obuf[(outpos - dist + i % dist) & kOBM];
                }
            } else {   // far back: the whole source is in HBM
                if (kProf) ++n_far_loop;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint8_t* src = out + outpos - dist;
                for (uint32_t i = lane; i < length; i += 64)
                    L.obuf[(outpos + i) & kOBM] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            outpos += length;
            KZ_FLUSH();
        }
    }
    lds_order();
    for (uint32_t i = flushed + lane; i < outpos; i += 64) out[i] = L.obuf[i & kOBM];   // the rest of the window
    if (err == KZ_OK && outpos != ulen) err = KZ_SIZE_MISMATCH;
    if (err == KZ_OK && ((bitpos + 7) >> 3) > clen) err = KZ_INPUT_OVERRUN;
    if (lane == 0) status[b] = err;
    if (prof && lane == 0) {
        unsigned long long* q = prof + (size_t)b * 8;
        q[0] = __builtin_readcyclecounter() - t_begin; q[1] = t_tables; q[2] = n_steps; q[3] = n_match; q[4] = n_slow; q[5] = n_dblk; q[6] = n_far_lane; q[7] = n_far_loop;
    }
#undef KZ_ENSURE
#undef KZ_FLUSH
}

}  // namespace

void launch_kz_inflate(const uint8_t* in, const BgzfBlock* blocks, uint32_t nblk, uint8_t* out, uint32_t* status, hipStream_t s, unsigned long long* prof) {
    if (!nblk) return;
    if (prof) hipLaunchKernelGGL(kz_inflate_kernel<true>, dim3(nblk), dim3(64), 0, s, in, blocks, nblk, out, status, prof);
    else hipLaunchKernelGGL(kz_inflate_kernel<false>, dim3(nblk), dim3(64), 0, s, in, blocks, nblk, out, status, prof);
}

}  // namespace bdx
