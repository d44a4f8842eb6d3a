// KZ (second path, opt-in: BDX_KZ=lanes) -- DEFLATE (RFC 1951) on the GPU in two kernels: Huffman decoding with one LANE per BGZF
// block (a wavefront decodes 64 members at once), then the matches resolved with 16 lanes per member.
//
// kz_inflate.hip gives a member a whole wave and spends its 64 lanes on speculation inside ONE serial Huffman chain: 131
// vector instructions buy 65 bits of one stream, and the vector pipe is what it runs out of (profiles/r03_inflate_pmc.txt).
// A BAM of configs[1]'s size is 49 k independent streams -- so here every lane runs the plain sequential decoder of its own
// member (what zlib's inflate_fast does per symbol, what the reference reaches through samtools' bgzf.c:inflate_block behind
// io/BamReader.hpp:62-70), and an instruction serves 64 streams.
//
// Kernel A, kz_huffman_lanes_kernel.  What makes 64 decoders fit a wave:
//   * no look-up tables: canonical decoding by comparison.  A code's length is 1 + the number of lengths l whose
//     left-justified limit (first_code[l] + count[l]) << (15 - l) the next 15 bits (bit-reversed) reach -- 15 compares against
//     values held in REGISTERS (a lane of a one-wave-per-SIMD kernel has hundreds) -- then one LDS read for the length's
//     index base and one for the symbol;
//   * 480 bytes of LDS per member: symbols sorted by (length, value) as bytes (the ninth bit of a literal/length symbol is
//     "index >= where the length's symbols >= 256 start", kept with the base), the code lengths of the block being built as
//     nibbles, overlaid by the per-length entries once the symbols are sorted; everything interleaved by lane at dword
//     granularity (dword k of lane j at k*64 + j), so that no access pattern has a bank conflict;
//   * input through a 64-bit bit buffer per lane with the next dword prefetched one refill ahead;
//   * NO reads of the output.  A lane that copied its matches itself would wait for a load of bytes it stored a moment ago,
//     in every iteration (some lane always has a match), and a wave's waits are wave-wide: the kernel ran 3x slower for it
//     (107 ms against 38 for 19.7 k members).  So a literal goes to its place, and a match leaves a 3-byte note -- distance,
//     length -- in the first bytes of the hole it will fill, a bit in a map of the output, and a bit in a second map if its
//     source lies further back than kernel B's ring (the maps: 1/4 of the output's size together, zeroed per launch);
//   * one token per lane per iteration; deflate-block headers (the expensive part: ~10^4 instructions per table build) are
//     taken when no lane can decode or at least 16 lanes wait -- zlib closes a block every 16 k symbols, so the lanes of a
//     wave reach their headers in the same iteration.
// Kernel B, kz_resolve_kernel: 16 lanes (a quarter wave) per member walk its output 64 bytes at a time through a 2 KiB ring in
// LDS (20 waves per CU) -- bytes loaded where no match owns them, the matches that start in the 64 bytes (the map's bits, in
// order) copied inside the ring, lane j byte j of a 16-byte piece, patterns for distances below 16 -- and write it back.  In
// configs[1]'s members two matches in three reach further back than any ring that fits (level-1 deflate of random fields:
// distances spread over the whole 32 KiB window, tools/deflate_tokens.py): their sources are final in HBM long before, so
// they are requested one iteration ahead and put in place before the group's other matches are resolved.  Everything per
// lane, no scalar chain: four members per instruction.
//
// Measured (profiles/r04_inflate_probe.txt, 49,201 members = 3.2 GB in ONE launch): A 46.8 ms + B 23.4 ms = 45.5 GB/s of inflated
// bytes against the wave kernel's 35.8 (32.0 in the launches of <= 1.5 GB the probe used before).  Not the 55 asked for, and
// not the default: (1) a lane is a sequential decoder -- a member takes ~45 ms however few there are, and with one wave per
// SIMD (30 KB of LDS per wave; 49 k members are 769 waves for 1,024 SIMDs anyway) nothing hides a token's dependent chain:
// 203 vector instructions, four LDS round trips and a load per token run at 2,800 cycles, 29 % of them issuing; (2) the decoder
// (bdx_bamdec_*) feeds the GPU in launches of 2-5 k members as the file arrives, where the wave kernel finishes a launch in 5-10 ms;
// a launch large enough for this path would be most of the file, i.e. no overlap with reading it.  Kept as a measured
// alternative with the parity tests running both (tests/test_gpu_bamdec.py); larger inputs per launch favour it further
// (118 k members: 42.5 GB/s before kernel B's 2 KiB ring).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "bdx_bam_dev.h"

namespace bdx {

namespace {

constexpr uint32_t kLitSym = 0;        // 288 bytes: literal/length symbols (low 8 bits) sorted by (length, value)
constexpr uint32_t kDistSym = 72;      // 32 bytes: distance symbols sorted the same way
constexpr uint32_t kScr = 80;          // 40 dwords: code lengths as nibbles while a block's tables are built ...
constexpr uint32_t kLitEnt = 80;       // ... then: entry of length l at kLitEnt + l: (index base - first code) & 0xFFFF | first index of a symbol >= 256 << 16
constexpr uint32_t kDistEnt = 96;      //           and kDistEnt + l: (index base - first code) & 0xFFFF
constexpr uint32_t kLaneDw = 120;
constexpr uint32_t kHdrBatch = 16;     // headers are taken when this many lanes wait (or nobody can decode)

enum : uint32_t { M_HEADER = 0, M_DECODE = 1, M_DONE = 3 };

// kernel B's ring (a member's last bytes, positions mod kRing) and what it can serve: a source at most kNearRing back is still there
// while its match (<= 258 bytes) and the next two 64-byte groups are written.  Kernel A marks the matches that reach further.
constexpr uint32_t kRing = 2048, kRingM = kRing - 1;
constexpr uint32_t kNearRing = kRing - 258 - 192 - 64;

struct BitIn {
    const uint8_t* in;
    uint64_t bb;     // the stream's next bc bits
    uint32_t bc;
    uint32_t nx;     // the dword behind them (already loaded)
    uint32_t ip;     // byte offset of the dword behind nx
};

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void st16(uint8_t* p, uint16_t v) { __builtin_memcpy(p, &v, 2); }

__device__ __forceinline__ void bit_start(BitIn& s, const uint8_t* in, uint32_t at) {
    s.in = in; s.bb = 0; s.bc = 0; s.nx = ld32(in + at); s.ip = at + 4;
}
// more than 32 bits in the buffer afterwards
__device__ __forceinline__ void refill(BitIn& s) {
    if (s.bc <= 32) {
        s.bb |= (uint64_t)s.nx << s.bc;
        s.bc += 32;
        s.nx = ld32(s.in + s.ip);
        s.ip += 4;
    }
}
__device__ __forceinline__ uint32_t peek(const BitIn& s) { return (uint32_t)s.bb; }
__device__ __forceinline__ void drop(BitIn& s, uint32_t n) { s.bb >>= n; s.bc -= n; }
__device__ __forceinline__ uint32_t bitpos(const BitIn& s) { return (s.ip - 4) * 8u - s.bc; }

__device__ __forceinline__ void length_of(uint32_t s, uint32_t* base, uint32_t* extra) {   // RFC 1951 3.2.5, symbols 257..285 as 0..28
    if (s < 8) { *base = 3 + s; *extra = 0; }
    else if (s == 28) { *base = 258; *extra = 0; }
    else { const uint32_t x = (s >> 2) - 1; *extra = x; *base = 3 + ((4 + (s & 3)) << x); }
}
__device__ __forceinline__ void distance_of(uint32_t s, uint32_t* base, uint32_t* extra) {
    if (s < 4) { *base = 1 + s; *extra = 0; }
    else { const uint32_t x = (s >> 1) - 1; *extra = x; *base = 1 + ((2 + (s & 1)) << x); }
}

// nibbles of x equal to l -> their top bits (exact: no carry crosses a nibble)
__device__ __forceinline__ uint32_t nib_eq(uint32_t x, uint32_t l) {
    const uint32_t y = x ^ (0x11111111u * l);
    return ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;
}
// top bits of the nibbles [0, n) of a dword (n may be <= 0 or >= 8)
__device__ __forceinline__ uint32_t nib_below(int n) { return n <= 0 ? 0u : n >= 8 ? 0x88888888u : (0x88888888u & ((1u << (4 * n)) - 1u)); }

// bitmap: pairs of dwords; bit (op & 31) of pair (out_off >> 5) - bm_origin + b + (op >> 5) is set in the pair's first dword where a
// match of member b starts at its byte op, and in the second too if the match's source lies further back than kernel B's ring
// (the + b keeps two members' pairs apart: out_off ascends by at least out_len from member to member)
__global__ __launch_bounds__(64) void kz_huffman_lanes_kernel(const uint8_t* __restrict__ in_all, const BgzfBlock* __restrict__ blocks, uint32_t nblk,
                                                           uint8_t* __restrict__ out_all, uint32_t* __restrict__ status, uint32_t* __restrict__ bitmap,
                                                           uint64_t bm_origin) {
    __shared__ uint32_t S[kLaneDw * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x * 64u + lane;
    const bool have = b < nblk;
    uint8_t* const Sb = (uint8_t*)S;
#define LDW(k) S[(k) * 64u + lane]
#define LBYTE(A, i) Sb[((((A) + ((i) >> 2)) * 64u + lane) << 2) + ((i) & 3u)]
    BgzfBlock blk{0, 0, 0, 0};
    if (have) blk = blocks[b];
    const uint8_t* in = in_all + blk.in_off;
    uint8_t* __restrict__ out = out_all + blk.out_off;
    uint32_t* __restrict__ bm = bitmap + 2 * ((blk.out_off >> 5) - bm_origin + b);
    const uint32_t ulen = blk.out_len;
    const uint32_t bit_limit = blk.in_len * 8u;
    uint32_t mode = have ? M_HEADER : M_DONE, err = KZ_OK, op = 0, last = 0;
    uint32_t bm_acc = 0, fm_acc = 0, bm_w = 0;   // the map's pair being filled
    BitIn s{};
    if (have) bit_start(s, in, 0);
    uint32_t lim[16], dlim[16];   // left-justified (15-bit) limits of the two codes; registers (every loop over them is unrolled)
#pragma unroll
    for (int l = 0; l < 16; ++l) { lim[l] = 0; dlim[l] = 0; }

    for (;;) {
        const bool hdr = mode == M_HEADER;
        const uint64_t m_hdr = __ballot(hdr), m_busy = __ballot(mode == M_DECODE);
        if (!m_hdr && !m_busy) break;
        if (m_hdr && (!m_busy || (uint32_t)__popcll(m_hdr) >= kHdrBatch)) {
            if (hdr) {
                // ---- a deflate block's header ----
                do {
                    if (bitpos(s) + 3 > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
                    refill(s);
                    uint32_t w = peek(s);
                    last = w & 1u;
                    const uint32_t type = (w >> 1) & 3u;
                    drop(s, 3);
                    if (type == 3) { err = KZ_BAD_BLOCK_TYPE; break; }
                    if (type == 0) {   // stored: to the byte boundary, LEN / NLEN, raw bytes
                        drop(s, s.bc & 7u);
                        refill(s);
                        w = peek(s);
                        const uint32_t len = w & 0xFFFFu, nlen = w >> 16;
                        const uint32_t at = (bitpos(s) >> 3) + 4;   // first raw byte
                        if ((bitpos(s) >> 3) + 4 > blk.in_len || (len ^ 0xFFFFu) != nlen || at + len > blk.in_len) { err = KZ_BAD_STORED; break; }
                        if (len > ulen - op) { err = KZ_OUTPUT_OVERRUN; break; }
                        uint32_t i = 0;
                        for (; i + 8 <= len; i += 8) st64(out + op + i, ld64(in + at + i));
                        for (; i < len; ++i) out[op + i] = in[at + i];
                        op += len;
                        bit_start(s, in, at + len);
                        if (last) mode = M_DONE;
                        break;
                    }
                    uint32_t hlit, hdist;
#pragma unroll
                    for (uint32_t k = 0; k < 40; ++k) LDW(kScr + k) = 0;
                    if (type == 1) {   // the fixed code: 0-143 8 bits, 144-255 9, 256-279 7, 280-287 8; 32 distance codes of 5 bits
                        hlit = 288; hdist = 32;
#pragma unroll
                        for (uint32_t k = 0; k < 40; ++k) LDW(kScr + k) = k < 18 ? 0x88888888u : k < 32 ? 0x99999999u : k < 35 ? 0x77777777u : k == 35 ? 0x88888888u : 0x55555555u;
                    } else {
                        hlit = (w >> 3 & 31u) + 257;
                        hdist = (w >> 8 & 31u) + 1;
                        const uint32_t hclen = (w >> 13 & 15u) + 4;
                        drop(s, 14);
                        if (hlit > 286 || hdist > 30) { err = KZ_BAD_LENGTHS; break; }
                        // the code-length code: 19 x 3 bits in a fixed order -> three bits per symbol in one register pair
                        uint64_t plen = 0;
                        {
                            const uint32_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
                            for (uint32_t i = 0; i < 19; ++i) {
                                if (i % 5 == 0) refill(s);
                                if (i < hclen) { plen |= (uint64_t)(peek(s) & 7u) << (3 * order[i]); drop(s, 3); }
                            }
                        }
                        constexpr uint64_t K3 = 0x0249249249249249ull;   // bit 3i, i < 19
                        uint32_t pc[8], plim[8];
                        uint64_t ps0 = 0, ps1 = 0;   // the code-length symbols sorted by (length, value), 5 bits each, 12 per register
                        {
                            int left = 1;
                            uint32_t first = 0, pos = 0;
                            bool bad = false;
                            pc[0] = 0; plim[0] = 0;
#pragma unroll
                            for (uint32_t l = 1; l <= 7; ++l) {
                                const uint64_t y = plen ^ (K3 * l);
                                uint64_t z = ~(((y & (K3 * 3)) + (K3 * 3)) | y) & (K3 * 4);
                                pc[l] = (uint32_t)__popcll(z);
                                left = (left << 1) - (int)pc[l];
                                if (left < 0) bad = true;
                                first = (first + pc[l - 1]) << 1;
                                plim[l] = (first + pc[l]) << (7 - l);
                                while (z) {
                                    const uint32_t sym = ((uint32_t)__builtin_ctzll(z) * 171u) >> 9;
                                    z &= z - 1;
                                    if (pos < 12) ps0 |= (uint64_t)sym << (5 * pos); else ps1 |= (uint64_t)sym << (5 * (pos - 12));
                                    ++pos;
                                }
                            }
                            if (bad || left != 0) { err = KZ_BAD_LENGTHS; break; }   // the code-length code must be complete
                        }
                        // the two alphabets' code lengths, run-length coded
                        const uint32_t total = hlit + hdist;
                        uint32_t n = 0, prev = 0, acc = 0, kcur = 0;
                        while (n < total) {
                            if (bitpos(s) > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
                            refill(s);
                            w = peek(s);
                            const uint32_t c7 = __builtin_bitreverse32(w) >> 25;
                            uint32_t L = 1, lo = 0, fi = 0;
#pragma unroll
                            for (uint32_t l = 1; l <= 7; ++l) {
                                const bool t = c7 >= plim[l];
                                L += t;
                                lo = t ? plim[l] : lo;
                                fi += t ? pc[l] : 0u;
                            }
                            if (L > 7) { err = KZ_BAD_LENGTHS; break; }
                            const uint32_t idx = fi + ((c7 - lo) >> (7 - L));
                            const uint32_t sym = idx < 12 ? (uint32_t)(ps0 >> (5 * idx)) & 31u : (uint32_t)(ps1 >> (5 * ((idx - 12) & 7u))) & 31u;
                            drop(s, L);
                            w = peek(s);
                            uint32_t rep = 1, val = sym;
                            if (sym == 16) {
                                if (n == 0) { err = KZ_BAD_LENGTHS; break; }
                                val = prev; rep = 3 + (w & 3u); drop(s, 2);
                            } else if (sym == 17) {
                                val = 0; rep = 3 + (w & 7u); drop(s, 3);
                            } else if (sym == 18) {
                                val = 0; rep = 11 + (w & 127u); drop(s, 7);
                            }
                            if (n + rep > total) { err = KZ_BAD_LENGTHS; break; }
                            if (val) {   // (zeros are what the area holds already)
                                for (uint32_t r = 0; r < rep; ++r) {
                                    const uint32_t k = (n + r) >> 3;
                                    if (k != kcur) { if (acc) LDW(kScr + kcur) = acc; acc = 0; kcur = k; }
                                    acc |= val << (((n + r) & 7u) * 4);
                                }
                            }
                            n += rep;
                            prev = val;
                        }
                        if (err != KZ_OK) break;
                        if (acc) LDW(kScr + kcur) = acc;
                        if ((LDW(kScr + 32) & 15u) == 0) { err = KZ_BAD_LENGTHS; break; }   // no end-of-block code
                    }
                    // ---- the block's two codes: counts per length, (symbols sorted by length), limits and entries ----
                    uint32_t cnt[16], dcnt[16], nlo[16];
#pragma unroll
                    for (int l = 0; l < 16; ++l) { cnt[l] = 0; dcnt[l] = 0; nlo[l] = 0; }
                    const uint32_t total = hlit + hdist;
                    for (uint32_t k = 0; k < 40; ++k) {
                        const uint32_t x = LDW(kScr + k);
                        const uint32_t ml = nib_below((int)hlit - (int)(8 * k));
                        const uint32_t md = nib_below((int)total - (int)(8 * k)) & ~ml;
                        const uint32_t mlo = k < 32 ? ml : 0u;
#pragma unroll
                        for (uint32_t l = 1; l <= 15; ++l) {
                            const uint32_t z = nib_eq(x, l);
                            cnt[l] += (uint32_t)__popc(z & ml);
                            dcnt[l] += (uint32_t)__popc(z & md);
                            nlo[l] += (uint32_t)__popc(z & mlo);
                        }
                    }
                    {   // over-subscribed, or incomplete other than by a single one-bit code (zlib's inflate_table rule)
                        int left = 1, dleft = 1;
                        uint32_t used = 0, dused = 0, maxlen = 0, dmax = 0;
                        bool bad = false;
#pragma unroll
                        for (uint32_t l = 1; l <= 15; ++l) {
                            left = (left << 1) - (int)cnt[l];
                            dleft = (dleft << 1) - (int)dcnt[l];
                            if (left < 0 || dleft < 0) bad = true;
                            used += cnt[l]; dused += dcnt[l];
                            if (cnt[l]) maxlen = l;
                            if (dcnt[l]) dmax = l;
                        }
                        if (bad || (left > 0 && !(used <= 1 && maxlen <= 1)) || (dleft > 0 && !(dused <= 1 && dmax <= 1))) { err = KZ_BAD_LENGTHS; break; }
                    }
                    {   // symbols in (length, value) order
                        uint32_t posl = 0, posd = 0;
                        for (uint32_t l = 1; l <= 15; ++l) {
                            for (uint32_t k0 = 0; k0 < 40; k0 += 4) {
                                uint32_t x[4];
#pragma unroll
                                for (uint32_t q = 0; q < 4; ++q) x[q] = LDW(kScr + k0 + q);
#pragma unroll
                                for (uint32_t q = 0; q < 4; ++q) {
                                    const uint32_t k = k0 + q;
                                    const uint32_t z = nib_eq(x[q], l);
                                    const uint32_t ml = nib_below((int)hlit - (int)(8 * k));
                                    uint32_t zl = z & ml, zd = z & nib_below((int)total - (int)(8 * k)) & ~ml;
                                    while (zl) {
                                        const uint32_t sym = 8 * k + ((uint32_t)__builtin_ctz(zl) >> 2);
                                        zl &= zl - 1;
                                        LBYTE(kLitSym, posl) = (uint8_t)sym;
                                        ++posl;
                                    }
                                    while (zd) {
                                        const uint32_t sym = 8 * k + ((uint32_t)__builtin_ctz(zd) >> 2) - hlit;
                                        zd &= zd - 1;
                                        LBYTE(kDistSym, posd) = (uint8_t)sym;
                                        ++posd;
                                    }
                                }
                            }
                        }
                    }
                    {   // (the nibbles are not needed any more: the entries take their place)
                        uint32_t first = 0, fi = 0, dfirst = 0, dfi = 0;
                        cnt[0] = 0; dcnt[0] = 0;
#pragma unroll
                        for (uint32_t l = 1; l <= 15; ++l) {
                            first = (first + cnt[l - 1]) << 1;
                            dfirst = (dfirst + dcnt[l - 1]) << 1;
                            lim[l] = (first + cnt[l]) << (15 - l);
                            dlim[l] = (dfirst + dcnt[l]) << (15 - l);
                            LDW(kLitEnt + l) = ((fi - first) & 0xFFFFu) | ((fi + nlo[l]) << 16);
                            LDW(kDistEnt + l) = (dfi - dfirst) & 0xFFFFu;
                            fi += cnt[l];
                            dfi += dcnt[l];
                        }
                    }
                    mode = M_DECODE;
                } while (false);
                if (err != KZ_OK) mode = M_DONE;
            }
            continue;
        }
        if (mode == M_DECODE) {
            // ---- one token ----
            do {
                if (bitpos(s) > bit_limit) { err = KZ_INPUT_OVERRUN; break; }
                refill(s);
                uint32_t w = peek(s);
                const uint32_t c15 = __builtin_bitreverse32(w) >> 17;
                uint32_t L = 1;
#pragma unroll
                for (uint32_t l = 1; l <= 15; ++l) L += c15 >= lim[l];
                if (L > 15) { err = KZ_BAD_CODE; break; }
                const uint32_t ent = LDW(kLitEnt + L);
                uint32_t idx = ((c15 >> (15 - L)) + ent) & 0xFFFFu;
                const bool hi = idx >= (ent >> 16);
                idx = idx < 287 ? idx : 287;
                const uint32_t symb = LBYTE(kLitSym, idx);
                drop(s, L);
                if (!hi) {
                    if (op >= ulen) { err = KZ_OUTPUT_OVERRUN; break; }
                    out[op] = (uint8_t)symb;
                    ++op;
                    break;
                }
                if (symb == 0) {   // end of block
                    mode = last ? M_DONE : M_HEADER;
                    break;
                }
                const uint32_t ls = symb - 1;
                if (ls > 28) { err = KZ_BAD_CODE; break; }
                uint32_t lbase, lx;
                length_of(ls, &lbase, &lx);
                const uint32_t len = lbase + (peek(s) & ((1u << lx) - 1u));
                drop(s, lx);
                refill(s);
                w = peek(s);
                const uint32_t d15 = __builtin_bitreverse32(w) >> 17;
                uint32_t dL = 1;
#pragma unroll
                for (uint32_t l = 1; l <= 15; ++l) dL += d15 >= dlim[l];
                if (dL > 15) { err = KZ_BAD_CODE; break; }
                uint32_t didx = ((d15 >> (15 - dL)) + LDW(kDistEnt + dL)) & 0xFFFFu;
                didx = didx < 31 ? didx : 31;
                const uint32_t dsym = LBYTE(kDistSym, didx);
                if (dsym > 29) { err = KZ_BAD_CODE; break; }
                drop(s, dL);
                uint32_t dbase, dx;
                distance_of(dsym, &dbase, &dx);
                const uint32_t dist = dbase + (peek(s) & ((1u << dx) - 1u));
                drop(s, dx);
                if (dist > op) { err = KZ_BAD_DISTANCE; break; }
                if (len > ulen - op) { err = KZ_OUTPUT_OVERRUN; break; }
                {   // the note in the hole's first three bytes, the bit(s) in the map
                    const uint32_t rec = (dist - 1u) | ((len - 3u) << 15);
                    st16(out + op, (uint16_t)rec);
                    out[op + 2] = (uint8_t)(rec >> 16);
                    const uint32_t wd = op >> 5;
                    if (wd != bm_w) {
                        if (bm_acc) { bm[2 * bm_w] = bm_acc; if (fm_acc) bm[2 * bm_w + 1] = fm_acc; }
                        bm_acc = 0; fm_acc = 0; bm_w = wd;
                    }
                    bm_acc |= 1u << (op & 31u);
                    if (dist > kNearRing) fm_acc |= 1u << (op & 31u);
                    op += len;
                }
            } while (false);
            if (err != KZ_OK) mode = M_DONE;
        }
    }
    if (have) {
        if (bm_acc) { bm[2 * bm_w] = bm_acc; if (fm_acc) bm[2 * bm_w + 1] = fm_acc; }
        if (err == KZ_OK && op != ulen) err = KZ_SIZE_MISMATCH;
        if (err == KZ_OK && ((bitpos(s) + 7) >> 3) > blk.in_len) err = KZ_INPUT_OVERRUN;
        status[b] = err;
    }
#undef LDW
#undef LBYTE
}


// ---- kernel B: the matches ----
constexpr uint32_t kRowsPerWave = 4;

__device__ __forceinline__ void lds_fence() { asm volatile("" ::: "memory"); }   // (a wave's LDS accesses execute in program order)

__global__ __launch_bounds__(64) void kz_resolve_kernel(const BgzfBlock* __restrict__ blocks, uint32_t nblk, uint8_t* out_all, const uint32_t* __restrict__ status,
                                                     const uint32_t* __restrict__ bitmap, uint64_t bm_origin) {
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[kRowsPerWave * kRing];
    const uint32_t lane = threadIdx.x, j = lane & 15u, row = lane >> 4;
    const uint32_t b = blockIdx.x * kRowsPerWave + row;
    uint8_t* const ring = ring_all + row * kRing;
    BgzfBlock blk{0, 0, 0, 0};
    if (b < nblk && status[b] == KZ_OK) blk = blocks[b];
    uint8_t* const out = out_all + blk.out_off;
    const uint32_t* const bm = bitmap + 2 * ((blk.out_off >> 5) - bm_origin + b);
    const uint32_t ulen = blk.out_len;
    const uint32_t ngroups = (ulen + 63u) >> 6;
    uint32_t filled = 0;   // where the last match met so far ends: bytes before it are the ring's, not HBM's
    // Everything a step needs from HBM is requested one iteration before it is used -- a group's bytes two groups ahead of the one
    // being resolved, the map's words two ahead, the first piece of up to two far matches of the next group -- always from a valid
    // address (lanes behind the member's end read its last dword instead and drop it), and a group is written back one
    // iteration after it was resolved: the one wait of an iteration, at its top, finds all of it done.
    auto group_addr = [&](uint32_t q, uint32_t* sh) -> const uint8_t* {
        const uint32_t p4 = q * 64u + 4u * j;
        uint32_t at = p4;
        *sh = 0;
        if (p4 + 4 > ulen) { at = ulen >= 4 ? ulen - 4 : 0; *sh = p4 < ulen ? 8u * (p4 - at) : 0u; }
        return out + at;
    };
    // bytes [64q + 4j, + 4) of the member into the ring, except what a match has produced already
    auto merge_group = [&](uint32_t q, uint32_t v) {
        const uint32_t p4 = q * 64u + 4u * j;
        if (p4 >= ulen) return;
        const uint32_t nb = filled > p4 ? filled - p4 : 0u;
        if (nb >= 4) return;
        uint32_t* const slot = (uint32_t*)(ring + (p4 & kRingM));
        const uint32_t m = nb ? (1u << (8 * nb)) - 1u : 0u;
        *slot = (*slot & m) | (v & ~m);
    };
    auto flush_group = [&](uint32_t q) {
        const uint32_t p4 = q * 64u + 4u * j;
        if (p4 >= ulen) return;
        const uint32_t v = *(const uint32_t*)(ring + (p4 & kRingM));
        if (p4 + 4 <= ulen) st32(out + p4, v);
        else {
            out[p4] = (uint8_t)v;
            if (p4 + 1 < ulen) out[p4 + 1] = (uint8_t)(v >> 8);
            if (p4 + 2 < ulen) out[p4 + 2] = (uint8_t)(v >> 16);
        }
    };
    // the map's two pairs of group q: match starts, far match starts (bits behind the member's end belong to its neighbour)
    auto map_of = [&](uint32_t q, uint64_t* starts, uint64_t* far) {
        uint32_t w[4];
        __builtin_memcpy(w, bm + 4 * q, 16);
        *starts = (uint64_t)w[0] | ((uint64_t)w[2] << 32);
        *far = (uint64_t)w[1] | ((uint64_t)w[3] << 32);
    };
    auto clip = [&](uint64_t v, uint32_t q) -> uint64_t {
        const uint32_t base = q * 64u;
        return base >= ulen ? 0ull : ulen - base < 64 ? v & ((1ull << (ulen - base)) - 1ull) : v;
    };
    auto note_at = [&](uint32_t p, uint32_t* dist, uint32_t* len) {
        const uint32_t rec = (uint32_t)ring[p & kRingM] | ((uint32_t)ring[(p + 1) & kRingM] << 8) | ((uint32_t)ring[(p + 2) & kRingM] << 16);
        *dist = (rec & 0x7FFFu) + 1u;
        *len = (rec >> 15) + 3u;
    };
    uint32_t sh_b, v_b;                 // group g + 1's bytes (in flight during iteration g - 1)
    uint64_t bits, far, bits_b, far_b;  // the map of group g, of group g + 1
    {
        uint32_t sh0;
        const uint32_t v0 = ld32(group_addr(0, &sh0));
        v_b = ld32(group_addr(1, &sh_b));
        map_of(0, &bits, &far);
        map_of(1, &bits_b, &far_b);
        merge_group(0, v0 >> sh0);
    }
    // Far matches (source further back than the ring; two in three of configs[1]'s) do not wait for their turn: the source is
    // final in HBM long before, so the first 16-byte piece of up to kFar of a group's far matches is requested one iteration
    // ahead and put in its place at the top of the group's own iteration, before any of the group's matches is resolved.  Only
    // matches of at most 16 bytes (97 %) are treated so (a longer one's note must survive until its turn).
    constexpr int kFar = 10;
    uint32_t pf_p[kFar], pf_len[kFar], pf_v[kFar];   // group g's: position (0xFFFFFFFF: none), length, lane j's byte
#pragma unroll
    for (int k = 0; k < kFar; ++k) { pf_p[k] = 0xFFFFFFFFu; pf_len[k] = 0; pf_v[k] = 0; }
    for (uint32_t g = 0; g < ngroups; ++g) {
        const uint32_t base = g * 64u;
        // all this wave has asked of memory in earlier iterations is done: the loads below have arrived, the groups written back so
        // far are in HBM (where a far match's source was written >= 50 iterations ago)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t todo = clip(bits, g);
#pragma unroll
        for (int k = 0; k < kFar; ++k) {
            if (pf_p[k] == 0xFFFFFFFFu) continue;
            if (j < pf_len[k]) ring[(pf_p[k] + j) & kRingM] = (uint8_t)pf_v[k];
            todo &= ~(1ull << (pf_p[k] - base));
            filled = filled > pf_p[k] + pf_len[k] ? filled : pf_p[k] + pf_len[k];
        }
        lds_fence();
        merge_group(g + 1, v_b >> sh_b);   // (a note in this group's last bytes runs into the next one)
        lds_fence();
        if (g) flush_group(g - 1);
        uint32_t sh_c;
        const uint32_t v_c = ld32(group_addr(g + 2, &sh_c));
        uint64_t bits_c, far_c;
        map_of(g + 2, &bits_c, &far_c);
        // the next group's far matches: their notes are in the ring now (unless one straddles into the group behind)
        uint32_t pn_p[kFar], pn_len[kFar], pn_v[kFar];
        {
            uint64_t fb = clip(far_b, g + 1);
#pragma unroll
            for (int k = 0; k < kFar; ++k) {
                pn_p[k] = 0xFFFFFFFFu; pn_len[k] = 0; pn_v[k] = 0;
                if (!fb) continue;
                const uint32_t bit = (uint32_t)__builtin_ctzll(fb);
                const uint32_t p = base + 64u + bit;
                fb &= fb - 1;
                if (bit > 61) continue;
                uint32_t dist, len;
                note_at(p, &dist, &len);
                if (len > 16) continue;
                pn_p[k] = p; pn_len[k] = len;
                pn_v[k] = __hip_atomic_load(out + p - dist + (j < len ? j : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        lds_fence();
        while (todo) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(todo);
            const uint32_t p = base + bit;
            todo &= todo - 1;
            uint32_t dist, len;
            note_at(p, &dist, &len);
            lds_fence();
            if (dist <= kNearRing) {
                if (dist >= 16) {   // 16-byte pieces in order: a piece's source lies before its destination
                    for (uint32_t i = j; i < len; i += 16) {
                        const uint8_t v = ring[(p + i - dist) & kRingM];
                        lds_fence();
                        ring[(p + i) & kRingM] = v;
                        lds_fence();
                    }
                } else {            // the pattern of dist bytes
                    for (uint32_t i = j; i < len; i += 16) ring[(p + i) & kRingM] = ring[(p - dist + i % dist) & kRingM];
                }
            } else {   // what is left of the far ones -- the ninth of a group, the long ones -- from HBM here
                       // (read past this CU's L1, which may hold the bytes of before the write-back)
                const uint8_t* src = out + p - dist;
                for (uint32_t i = j; i < len; i += 16)
                    ring[(p + i) & kRingM] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            lds_fence();
            filled = filled > p + len ? filled : p + len;
        }
        v_b = v_c; sh_b = sh_c;
        bits = bits_b; far = far_b; bits_b = bits_c; far_b = far_c;
#pragma unroll
        for (int k = 0; k < kFar; ++k) { pf_p[k] = pn_p[k]; pf_len[k] = pn_len[k]; pf_v[k] = pn_v[k]; }
    }
    lds_fence();
    if (ngroups) flush_group(ngroups - 1);
}

}  // namespace

bool kz_pick_lanes(size_t nblk, size_t comp_bytes, size_t infl_bytes) {
    (void)nblk; (void)comp_bytes;
    // Measured (profiles/r04_inflate_probe.txt): this path does not beat the wave kernel at configs[1]'s 49 k members, so it is
    // taken only on request -- BDX_KZ=lanes (the parity tests, tools/bamdec_probe.py, tools/kz_stats.sh) -- and only for launches of
    // at most 256 MiB of output: round 5's determinism probe met a memory fault in a launch of 0.8 GB (12 k members; 0.5 GB and less
    // decode byte for byte), and a path that lost the measurement is not worth chasing it -- larger launches run the wave kernel.
    // (The CLI's decoder never takes this path: bdx_bamdec_impl.h.)
    const char* e = getenv("BDX_KZ");
    return e && !strcmp(e, "lanes") && infl_bytes <= ((size_t)256 << 20);
}

size_t kz_bitmap_words(size_t out_span_bytes, size_t nblk) { return 2 * ((out_span_bytes >> 5) + nblk + 8); }

void launch_kz_inflate_lanes(const uint8_t* in, const BgzfBlock* blocks, uint32_t nblk, uint8_t* out, uint32_t* status, uint32_t* bitmap, size_t bitmap_words,
                             uint64_t out_origin, hipStream_t s) {
    if (!nblk) return;
    (void)hipMemsetAsync(bitmap, 0, bitmap_words * 4, s);
    hipLaunchKernelGGL(kz_huffman_lanes_kernel, dim3((nblk + 63) / 64), dim3(64), 0, s, in, blocks, nblk, out, status, bitmap, (uint64_t)(out_origin >> 5));
    hipLaunchKernelGGL(kz_resolve_kernel, dim3((nblk + kRowsPerWave - 1) / kRowsPerWave), dim3(64), 0, s, blocks, nblk, out, status, bitmap, (uint64_t)(out_origin >> 5));
}

}  // namespace bdx
