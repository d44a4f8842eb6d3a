// KZ -- DEFLATE (RFC 1951) on the GPU: one wavefront inflates one BGZF block.
//
// Stands where the host producer has zlib / host/fast_inflate.cpp (and the reference samtools' bgzf.c:inflate_block behind
// io/BamReader.hpp:62-70).  A BAM of N records is ~N/330 independent deflate streams of <= 64 KiB, tens of thousands for a
// chromosome: enough to fill 256 compute units with one stream per wave, and the file then crosses PCIe compressed.
//
// What a wave does with its 64 lanes.  Huffman decoding is a serial chain -- a symbol's position is known only once its
// predecessor's length is -- so a CPU decoder pays one table look-up latency per symbol.  Here the look-ups of a whole
// 64-bit window run at once: lane j fetches the bits at offset (cursor + j) and looks ITS candidate symbol up in the
// LDS table (one ds_read for 64 candidates); the chain of real symbols is then followed through the lanes' results with
// v_readlane (scalar, ~8 instructions per symbol, no memory access), the literal lanes that lie on the chain store their
// bytes side by side (ballot rank), and the wave moves on by up to 64 + 14 bits.  Literal-heavy streams -- BAM's base and
// quality bytes -- take ~9 symbols per step.  A length/distance pair ends the step: its extra bits and distance code are in
// the 57 bits the stopping lane already holds, and the copy is done by all lanes (lane i copies byte i, i mod distance for
// overlapping copies).  Codes longer than the primary table's index (rare symbols by construction) are decoded canonically,
// bit by bit, from the count / sorted-symbol arrays (puff-style): no second-level tables.
//
// Tables are built per deflate block by the wave itself: lengths read with the same canonical decoder (the code-length
// code has 19 symbols of <= 7 bits), canonical codes by a per-length scan, table filled symbol by symbol across the lanes.
//
// Memory: the last 2 KiB of output live in an LDS window; literals and matches up to ~1.7 KiB back never touch HBM, and the
// window is written out 512 bytes at a time (one 8-byte store per lane).  This matters more than anything else here: with
// one byte store per step, every step's input load queued behind the previous step's store and waited for its write
// acknowledgement (loads and stores return in order on this part) -- 1.9 us per step, 24 GB/s in all.  Matches further back
// read HBM, where those bytes have been for at least one chunk.  LDS per wave ~9 KB -> 17 waves per compute unit.
#include <hip/hip_runtime.h>

#include "bdx_bam_dev.h"

namespace bdx {

namespace {

constexpr int kLB = 10;            // primary literal/length table: 2^10 entries
constexpr int kDB = 8;             // primary distance table
constexpr int kMaxBits = 15;
constexpr uint32_t K_SLOW = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_DIST = 4, K_BAD = 5;

// table entry: bits 0-3 code length, 4-6 kind, 8-11 extra bits, 16-31 value (literal byte / base length / base distance)
__device__ __forceinline__ uint32_t mk(uint32_t kind, uint32_t len, uint32_t extra, uint32_t value) {
    return len | (kind << 4) | (extra << 8) | (value << 16);
}
__device__ __forceinline__ uint32_t e_len(uint32_t e) { return e & 15u; }
__device__ __forceinline__ uint32_t e_kind(uint32_t e) { return (e >> 4) & 7u; }
__device__ __forceinline__ uint32_t e_extra(uint32_t e) { return (e >> 8) & 15u; }
__device__ __forceinline__ uint32_t e_value(uint32_t e) { return e >> 16; }

__device__ __forceinline__ uint32_t litlen_entry(uint32_t sym, uint32_t len) {
    if (sym < 256) return mk(K_LIT, len, 0, sym);
    if (sym == 256) return mk(K_EOB, len, 0, 0);
    if (sym > 285) return mk(K_BAD, len, 0, 0);
    const uint32_t s = sym - 257;
    if (s < 8) return mk(K_LEN, len, 0, 3 + s);
    if (s == 28) return mk(K_LEN, len, 0, 258);
    const uint32_t x = (s >> 2) - 1;
    return mk(K_LEN, len, x, 3 + ((4 + (s & 3)) << x));
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len) {
    if (sym > 29) return mk(K_BAD, len, 0, 0);
    if (sym < 4) return mk(K_DIST, len, 0, 1 + sym);
    const uint32_t x = (sym >> 1) - 1;
    return mk(K_DIST, len, x, 1 + ((2 + (sym & 1)) << x));
}

struct __attribute__((aligned(16))) Lds {
    uint32_t lit[1 << kLB];
    uint32_t dist[1 << kDB];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t lit_cnt[16];
    uint16_t dist_cnt[16];
    uint16_t pre_sorted[20];
    uint16_t pre_cnt[16];
    uint16_t codes[320];   // canonical code of every symbol (table build)
    uint8_t lens[320];     // code lengths: literal/length alphabet, then distances
    uint8_t obuf[2048];    // the last 2 KiB of output (position p at p mod 2048): literals and near matches never touch HBM
};
constexpr uint32_t kOB = 2048, kOBM = kOB - 1;
constexpr uint32_t kFlush = 512;                    // bytes written to HBM at a time (64 lanes x 8 bytes)
constexpr uint32_t kNearDist = kOB - 258 - 64;      // matches up to this distance are copied inside the window

// 64 bits of the stream starting at bit `bitpos` (>= 57 of them valid)
__device__ __forceinline__ uint64_t peek(const uint8_t* in, uint32_t bitpos) {
    uint64_t w;
    __builtin_memcpy(&w, in + (bitpos >> 3), 8);
    return w >> (bitpos & 7);
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// canonical decode, one bit at a time (codes are packed starting with their most significant bit): returns the code's
// length and its symbol, 0 if the bits are no code.  cnt[l] = codes of length l, sorted = symbols by (length, value).
__device__ __forceinline__ uint32_t canon_decode(uint64_t w, const uint16_t* cnt, const uint16_t* sorted, uint32_t* sym) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= kMaxBits; ++l) {
        code |= (int)(w & 1);
        w >>= 1;
        const int count = cnt[l];
        if (code - count < first) {
            *sym = sorted[index + (code - first)];
            return (uint32_t)l;
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return 0;
}

__device__ __forceinline__ uint32_t bitrev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32 - len); }

// lens[0, n) -> cnt[16], sorted[], and (tbits > 0) the primary table.  All 64 lanes.  allow_single: an incomplete code is
// accepted if it consists of one code of length 1 (DEFLATE's single distance code), as zlib does.  Returns false for an
// over-subscribed or incomplete set of lengths.
template <bool kDist>
__device__ bool build_tables(const uint8_t* lens, uint32_t n, uint32_t* table, int tbits, uint16_t* cnt, uint16_t* sorted, uint16_t* codes,
                             bool allow_single) {
    const uint32_t lane = threadIdx.x;
    // lane L counts the codes of length L
    uint32_t mine = 0;
    if (lane >= 1 && lane <= (uint32_t)kMaxBits)
        for (uint32_t s = 0; s < n; ++s) mine += lens[s] == lane;
    if (lane < 16) cnt[lane] = (uint16_t)(lane == 0 ? 0 : mine);
    __syncthreads();
    // Kraft sum, first code and first index of every length (the same in all lanes)
    int left = 1;
    uint32_t first_code = 0, first_index = 0, my_first = 0, my_index = 0, used = 0, maxlen = 0;
    bool over = false;
    for (uint32_t l = 1; l <= (uint32_t)kMaxBits; ++l) {
        const uint32_t c = cnt[l];
        left <<= 1;
        left -= (int)c;
        if (left < 0) over = true;
        if (l == lane) { my_first = first_code; my_index = first_index; }
        first_code = (first_code + c) << 1;
        first_index += c;
        used += c;
        if (c) maxlen = l;
    }
    if (over) return false;
    if (left > 0 && !(allow_single && used <= 1 && maxlen <= 1)) return false;   // incomplete (zlib: inflate_table returns -1)
    if (used == 0) {  // (a distance alphabet without codes: legal, every distance symbol is then an error)
        if (tbits > 0)
            for (uint32_t i = lane; i < (1u << tbits); i += 64) table[i] = mk(K_BAD, 1, 0, 0);
        __syncthreads();
        return true;
    }
    // lane L: symbols of length L in symbol order -> sorted[], and their codes
    if (lane >= 1 && lane <= (uint32_t)kMaxBits && mine) {
        uint32_t k = 0;
        for (uint32_t s = 0; s < n; ++s)
            if (lens[s] == lane) {
                sorted[my_index + k] = (uint16_t)s;
                codes[s] = (uint16_t)(my_first + k);
                ++k;
            }
    }
    if (tbits > 0) {
        const uint32_t tsize = 1u << tbits;
        for (uint32_t i = lane; i < tsize; i += 64) table[i] = mk(K_SLOW, 0, 0, 0);   // longer codes: canonical decode
        __syncthreads();
        for (uint32_t s = lane; s < n; s += 64) {
            const uint32_t l = lens[s];
            if (l == 0 || l > (uint32_t)tbits) continue;
            const uint32_t e = kDist ? dist_entry(s, l) : litlen_entry(s, l);
            for (uint32_t i = bitrev(codes[s], l); i < tsize; i += 1u << l) table[i] = e;
        }
    }
    __syncthreads();
    return true;
}

// LDS accesses of a wave are executed in program order; this only keeps the compiler from moving them across
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// 512 bytes of the window, [flushed, flushed + 512), to HBM: lane l takes 8 of them
__device__ __forceinline__ void flush_chunk(const uint8_t* obuf, uint8_t* out, uint32_t flushed, uint32_t lane) {
    const uint32_t at = flushed + 8u * lane;
    uint64_t v;
    if ((flushed & 7u) == 0) {
        v = *(const uint64_t*)(obuf + (at & kOBM));
    } else {
        v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) v |= (uint64_t)obuf[(at + k) & kOBM] << (8 * k);
    }
    __builtin_memcpy(out + at, &v, 8);
}

__global__ __launch_bounds__(64) void kz_inflate_kernel(const uint8_t* __restrict__ in_all, const BgzfBlock* __restrict__ blocks, uint32_t nblk,
                                                        uint8_t* out_all, uint32_t* __restrict__ status, unsigned long long* __restrict__ prof) {
    __shared__ Lds L;
    // measurement hook (tools/bamdec_probe.py --prof): per member {cycles in all, in headers + tables, steps, matches, slow codes, deflate blocks}
    unsigned long long t_begin = 0, t_tables = 0, n_steps = 0, n_match = 0, n_slow = 0, n_dblk = 0;
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
    uint32_t flushed = 0;   // output bytes already in HBM; [flushed, outpos) sit in the LDS window only
    bool last = false;

    while (!last && err == KZ_OK) {
        const unsigned long long t_hdr = prof ? __builtin_readcyclecounter() : 0;
        ++n_dblk;
        if (bitpos + 3 > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
        uint64_t w = peek(in, bitpos);
        // (header fields are the same in all lanes; saying so keeps the loops they bound scalar)
        last = uni((uint32_t)w & 1u) != 0;
        const uint32_t type = uni((uint32_t)(w >> 1) & 3u);
        bitpos += 3;
        if (type == 0) {  // stored: to the byte boundary, LEN / NLEN, raw bytes
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
            continue;
        }
        if (type == 3) { err = KZ_BAD_BLOCK_TYPE; break; }
        uint32_t hlit, hdist;
        if (type == 1) {  // fixed code
            for (uint32_t s = lane; s < 288; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) L.lens[288 + lane] = 5;   // (30 and 31 never occur in valid data: they decode to K_BAD)
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
                if (lane < hclen) {
                    const uint32_t p = bitpos + 3u * lane;
                    uint32_t v;
                    __builtin_memcpy(&v, in + (p >> 3), 4);
                    L.lens[order[lane]] = (uint8_t)((v >> (p & 7)) & 7u);
                }
            }
            bitpos += 3u * hclen;
            __syncthreads();
            if (!build_tables<false>(L.lens, 19, nullptr, 0, L.pre_cnt, L.pre_sorted, L.codes, false)) { err = KZ_BAD_LENGTHS; break; }
            // the two alphabets' code lengths, run-length coded (every lane runs the same sequence; lane 0 stores)
            const uint32_t total = hlit + hdist;
            uint32_t n = 0, prev = 0;
            // (L.lens is both the code-length code's input above and the output here: the build above is complete, and its
            // results live in pre_cnt / pre_sorted)
            __syncthreads();
            while (n < total) {
                if (bitpos > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
                const uint64_t ww = peek(in, bitpos);
                uint32_t sym = 0;
                const uint32_t l = uni(canon_decode(ww, L.pre_cnt, L.pre_sorted, &sym));
                if (l == 0) { err = KZ_BAD_LENGTHS; break; }
                sym = uni(sym);
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
        if (!build_tables<false>(L.lens, hlit, L.lit, kLB, L.lit_cnt, L.lit_sorted, L.codes, true)) { err = KZ_BAD_LENGTHS; break; }
        if (!build_tables<true>(L.lens + hlit, hdist, L.dist, kDB, L.dist_cnt, L.dist_sorted, L.codes, true)) { err = KZ_BAD_LENGTHS; break; }

        // ---- the block's symbols ----
        if (prof) t_tables += __builtin_readcyclecounter() - t_hdr;
        const uint64_t lane_bit = 1ull << lane;
        for (;;) {
            ++n_steps;
            if (bitpos > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
            // every lane's candidate symbol
            const uint64_t wl = peek(in, bitpos + lane);
            const uint32_t e = L.lit[(uint32_t)wl & ((1u << kLB) - 1)];
            // Where the chain goes from each lane, two symbols at a time.  nxt: the offset behind a literal (1..78), 128 + lane
            // for anything else.  t2: where a chain that stands on this lane stands two literals later -- or 64..78: the window
            // is used up, go on there; >= 128: stopped at lane (t2 - 128), which is no literal.  pm: the literal lanes passed.
            const bool lit = e_kind(e) == K_LIT;
            const uint32_t nxt = lit ? lane + e_len(e) : 128u + lane;
            const uint32_t n2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nxt & 63u) << 2), (int)nxt);
            const uint32_t t2 = !lit ? nxt : (nxt >= 64 ? nxt : n2);
            const bool two = lit && nxt < 64 && n2 < 128;
            const uint64_t pm = lit ? (lane_bit | (two ? 1ull << nxt : 0ull)) : 0ull;
            const uint32_t pm_lo = (uint32_t)pm, pm_hi = (uint32_t)(pm >> 32);
            uint32_t cur = 0, v;
            uint64_t mask = 0;
            for (;;) {
                v = (uint32_t)__builtin_amdgcn_readlane((int)t2, (int)cur);
                mask |= ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)pm_hi, (int)cur) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)pm_lo, (int)cur);
                if (v >= 64) break;
                cur = v;
            }
            const bool stopped = v >= 128;
            const uint32_t pos = stopped ? v - 128 : v;
            uint32_t stop_e = 0;
            if (stopped) stop_e = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)pos);
            const uint32_t nlit = (uint32_t)__builtin_popcountll(mask);
            if (nlit) {
                if (nlit > ulen - outpos) { err = KZ_OUTPUT_OVERRUN; break; }
                if ((mask >> lane) & 1) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                    L.obuf[(outpos + rank) & kOBM] = (uint8_t)e_value(e);
                }
                outpos += nlit;
                lds_order();
                if (outpos - flushed >= kFlush) { flush_chunk(L.obuf, out, flushed, lane); flushed += kFlush; }
            }
            bitpos += pos;
            if (!stopped) continue;
            // the symbol the chain stopped at, with the bits its lane holds
            uint64_t ws = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wl >> 32), (int)pos) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wl, (int)pos);
            uint32_t se = stop_e;
            if (e_kind(se) == K_SLOW) {  // a code longer than the table's index
                ++n_slow;
                uint32_t sym = 0;
                const uint32_t l = uni(canon_decode(ws, L.lit_cnt, L.lit_sorted, &sym));
                if (l == 0) { err = KZ_BAD_CODE; break; }
                se = litlen_entry(uni(sym), l);
            }
            const uint32_t kind = e_kind(se);
            uint32_t used = e_len(se);
            ws >>= used;
            if (kind == K_LIT) {
                if (outpos >= ulen) { err = KZ_OUTPUT_OVERRUN; break; }
                if (lane == 0) L.obuf[outpos & kOBM] = (uint8_t)e_value(se);
                ++outpos;
                lds_order();
                bitpos += used;
                continue;
            }
            if (kind == K_EOB) { bitpos += used; break; }
            if (kind != K_LEN) { err = KZ_BAD_CODE; break; }
            ++n_match;
            const uint32_t lx = e_extra(se);
            const uint32_t length = e_value(se) + ((uint32_t)ws & ((1u << lx) - 1));
            ws >>= lx;
            used += lx;
            uint32_t de = uni(L.dist[(uint32_t)ws & ((1u << kDB) - 1)]);
            if (e_kind(de) == K_SLOW) {
                uint32_t sym = 0;
                const uint32_t l = uni(canon_decode(ws, L.dist_cnt, L.dist_sorted, &sym));
                if (l == 0) { err = KZ_BAD_CODE; break; }
                de = dist_entry(uni(sym), l);
            }
            if (e_kind(de) != K_DIST) { err = KZ_BAD_CODE; break; }
            ws >>= e_len(de);
            used += e_len(de);
            const uint32_t dx = e_extra(de);
            const uint32_t dist = e_value(de) + ((uint32_t)ws & ((1u << dx) - 1));
            used += dx;
            bitpos += used;
            if (dist > outpos) { err = KZ_BAD_DISTANCE; break; }
            if (length > ulen - outpos) { err = KZ_OUTPUT_OVERRUN; break; }
            if (dist <= kNearDist) {   // inside the window (sources lie below outpos, destinations at or above it: no overlap)
                lds_order();
                if (dist >= length) {
                    for (uint32_t i = lane; i < length; i += 64) L.obuf[(outpos + i) & kOBM] = L.obuf[(outpos - dist + i) & kOBM];
                } else {  // overlapping: the pattern of `dist` bytes repeats
                    for (uint32_t i = lane; i < length; i += 64) L.obuf[(outpos + i) & kOBM] = L.obuf[(outpos - dist + i % dist) & kOBM];
                }
                lds_order();
            } else {   // far back: those bytes left the window, but they were flushed at least one chunk ago
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint8_t* src = out + outpos - dist;
                for (uint32_t i = lane; i < length; i += 64)
                    L.obuf[(outpos + i) & kOBM] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lds_order();
            }
            outpos += length;
            if (outpos - flushed >= kFlush) { flush_chunk(L.obuf, out, flushed, lane); flushed += kFlush; }
        }
    }
    lds_order();
    for (uint32_t i = flushed + lane; i < outpos; i += 64) out[i] = L.obuf[i & kOBM];   // the rest of the window
    if (err == KZ_OK && outpos != ulen) err = KZ_SIZE_MISMATCH;
    if (err == KZ_OK && ((bitpos + 7) >> 3) > clen) err = KZ_INPUT_OVERRUN;
    if (lane == 0) status[b] = err;
    if (prof && lane == 0) {
        unsigned long long* q = prof + (size_t)b * 6;
        q[0] = __builtin_readcyclecounter() - t_begin; q[1] = t_tables; q[2] = n_steps; q[3] = n_match; q[4] = n_slow; q[5] = n_dblk;
    }
}

}  // namespace

void launch_kz_inflate(const uint8_t* in, const BgzfBlock* blocks, uint32_t nblk, uint8_t* out, uint32_t* status, hipStream_t s, unsigned long long* prof) {
    if (!nblk) return;
    hipLaunchKernelGGL(kz_inflate_kernel, dim3(nblk), dim3(64), 0, s, in, blocks, nblk, out, status, prof);
}

}  // namespace bdx
