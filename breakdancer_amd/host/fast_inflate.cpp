// A DEFLATE (RFC 1951) decoder for BGZF blocks, written for throughput: the producer is bound by inflating ~3 GB of
// record bytes per 15 M reads on the few CPUs a container grants, and zlib's inflate manages
// ~375 MB/s per core on these literal-heavy blocks.
//
// What makes it faster than a bit-at-a-time or zlib-style decoder:
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load (one refill per literal/length + distance pair:
//     15 + 5 + 15 + 13 bits fit),
//   * Huffman decode tables with an 11-bit (literal/length) and 8-bit (distance) first level, so nearly every code is
//     one lookup; longer codes go through a second-level table,
//   * up to three literals decoded per loop iteration before the next refill (a table that yields two literals per lookup
//     was tried: read data's literals have 6-7 bit codes, two rarely fit a 12-bit index, and the table build per deflate
//     block ate what was left),
//   * matches copied eight bytes at a time (the output buffer has slack behind the block; short distances are widened
//     first),
//   * the whole block is one call with a known output size: no streaming state machine.
// Anything unexpected (bad code lengths, a distance before the start, output that does not come out at exactly the
// size the BGZF footer states) makes it return false; the caller then lets zlib decide whether the block is corrupt.
#include "fast_inflate.h"

#include <cstring>

namespace bdhost {

namespace {

constexpr int kLitBits = 11, kDistBits = 8, kMaxLen = 15;
constexpr int kNumLit = 288, kNumDist = 32, kNumPre = 19;
enum : uint8_t { K_LITERAL = 0, K_LENGTH = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };

struct Entry {
    uint16_t value;  // literal byte / base length / base distance / first index of the second-level table
    uint8_t len;     // bits of the code covered by this lookup (second level: the remaining bits)
    uint8_t kx;      // kind << 5 | number of extra bits (second-level pointer: its index width)
};
inline Entry make(uint16_t v, int len, int kind, int extra) { return Entry{v, (uint8_t)len, (uint8_t)((kind << 5) | extra)}; }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// symbol -> table entry for the three alphabets
inline Entry litlen_entry(int sym, int len) {
    if (sym < 256) return make((uint16_t)sym, len, K_LITERAL, 0);
    if (sym == 256) return make(0, len, K_EOB, 0);
    if (sym > 285) return make(0, len, K_BAD, 0);
    return make(kLenBase[sym - 257], len, K_LENGTH, kLenExtra[sym - 257]);
}
inline Entry dist_entry(int sym, int len) {
    if (sym > 29) return make(0, len, K_BAD, 0);
    return make(kDistBase[sym], len, K_LENGTH, kDistExtra[sym]);
}
inline Entry pre_entry(int sym, int len) { return make((uint16_t)sym, len, K_LITERAL, 0); }

// Canonical Huffman code -> two-level decode table.  table must hold (1 << tbits) + room for the second level
// (<= 2 * nsyms entries suffice for 15-bit codes over these alphabets).  Returns false for an over-subscribed code, or an
// incomplete one unless allow_incomplete (DEFLATE permits a single distance code).
template <class MakeEntry>
bool build_table(const uint8_t* lens, int nsyms, int tbits, Entry* table, int table_cap, MakeEntry mk, bool allow_incomplete) {
    int count[kMaxLen + 1] = {0};
    for (int s = 0; s < nsyms; ++s) ++count[lens[s]];
    count[0] = 0;
    uint32_t code = 0, first[kMaxLen + 2] = {0};
    int64_t left = 1;  // Kraft: codes still available
    for (int l = 1; l <= kMaxLen; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
        code = (code + (uint32_t)count[l - 1]) << 1;
        first[l] = code;
    }
    int used = 0;
    for (int l = 1; l <= kMaxLen; ++l) used += count[l];
    if (left > 0 && !(allow_incomplete && used <= 1)) return false;
    const int tsize = 1 << tbits;
    for (int i = 0; i < tsize; ++i) table[i] = make(0, 1, K_BAD, 0);
    if (used == 0) return true;
    // second-level tables: for every first-level prefix that long codes share, the longest of them decides the width
    uint8_t sub_bits[1 << kLitBits];
    memset(sub_bits, 0, (size_t)tsize);
    uint32_t next[kMaxLen + 2];
    memcpy(next, first, sizeof(next));
    // pass 1: widths of the second-level tables
    {
        uint32_t nx[kMaxLen + 2];
        memcpy(nx, first, sizeof(nx));
        for (int s = 0; s < nsyms; ++s) {
            const int l = lens[s];
            if (l <= tbits) { if (l) ++nx[l]; continue; }
            const uint32_t rev = reverse_bits(nx[l]++, l);
            const uint32_t pre = rev & (uint32_t)(tsize - 1);
            if (l - tbits > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - tbits);
        }
    }
    int pos = tsize;
    for (int i = 0; i < tsize; ++i) {
        if (!sub_bits[i]) continue;
        const int sz = 1 << sub_bits[i];
        if (pos + sz > table_cap) return false;
        table[i] = make((uint16_t)pos, tbits, K_SUB, sub_bits[i]);
        for (int k = 0; k < sz; ++k) table[pos + k] = make(0, 1, K_BAD, 0);
        pos += sz;
    }
    // pass 2: fill
    for (int s = 0; s < nsyms; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = reverse_bits(next[l]++, l);
        if (l <= tbits) {
            const Entry e = mk(s, l);
            for (uint32_t i = rev; i < (uint32_t)tsize; i += 1u << l) table[i] = e;
        } else {
            const uint32_t pre = rev & (uint32_t)(tsize - 1);
            const int sb = sub_bits[pre];
            const int base = table[pre].value;
            const Entry e = mk(s, l - tbits);
            for (uint32_t i = rev >> tbits; i < (1u << sb); i += 1u << (l - tbits)) table[base + i] = e;
        }
    }
    return true;
}

struct Tables {
    Entry lit[(1 << kLitBits) + 2 * kNumLit];
    Entry dist[(1 << kDistBits) + 256];  // (the worst 15-bit code over 30 symbols needs 146 second-level entries)
};

inline uint64_t load64(const uint8_t* p) { uint64_t w; memcpy(&w, p, 8); return w; }
inline void store64(uint8_t* p, uint64_t w) { memcpy(p, &w, 8); }

}  // namespace

bool fast_inflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, size_t out_slack) {
    const uint8_t* const in_begin = in;
    const uint8_t* const in_end = in + in_len;
    uint8_t* const out_begin = out;
    uint8_t* const out_end = out + out_len;
    uint64_t bitbuf = 0;
    unsigned bitsleft = 0;
    Tables T;
    bool fixed_built = false;
    Tables F;  // (fixed-code tables, built at most once per call)

#define BDX_REFILL()                                              \
    do {                                                          \
        bitbuf |= load64(in) << bitsleft;                         \
        in += (63 - bitsleft) >> 3;                               \
        bitsleft |= 56;                                           \
    } while (0)
#define BDX_BITS(n) ((uint32_t)(bitbuf & ((1ull << (n)) - 1)))
#define BDX_DROP(n) do { bitbuf >>= (n); bitsleft -= (n); } while (0)

    for (;;) {
        if (in > in_end + 8) return false;
        BDX_REFILL();
        const uint32_t final = BDX_BITS(1);
        const uint32_t type = (uint32_t)((bitbuf >> 1) & 3);
        BDX_DROP(3);
        const Tables* tb = nullptr;
        if (type == 0) {  // stored: byte-align, LEN / NLEN, raw bytes
            BDX_DROP(bitsleft & 7);
            // bytes already in the bit buffer belong to the input again
            in -= bitsleft >> 3;
            bitbuf = 0; bitsleft = 0;
            if (in + 4 > in_end) return false;
            const uint32_t len = in[0] | (in[1] << 8), nlen = in[2] | (in[3] << 8);
            in += 4;
            if ((len ^ 0xFFFFu) != nlen || in + len > in_end || len > (size_t)(out_end - out)) return false;
            memcpy(out, in, len);
            in += len; out += len;
            if (final) break;
            continue;
        }
        if (type == 3) return false;
        if (type == 1) {
            if (!fixed_built) {
                uint8_t l[kNumLit];
                for (int i = 0; i < 144; ++i) l[i] = 8;
                for (int i = 144; i < 256; ++i) l[i] = 9;
                for (int i = 256; i < 280; ++i) l[i] = 7;
                for (int i = 280; i < 288; ++i) l[i] = 8;
                uint8_t d[kNumDist];
                for (int i = 0; i < 32; ++i) d[i] = 5;
                if (!build_table(l, kNumLit, kLitBits, F.lit, (int)(sizeof(F.lit) / sizeof(Entry)), litlen_entry, false)) return false;
                if (!build_table(d, kNumDist, kDistBits, F.dist, (int)(sizeof(F.dist) / sizeof(Entry)), dist_entry, false)) return false;
                fixed_built = true;
            }
            tb = &F;
        } else {
            // dynamic: HLIT, HDIST, HCLEN, the code-length code, then the two alphabets' lengths
            const uint32_t hlit = BDX_BITS(5) + 257;
            BDX_DROP(5);
            const uint32_t hdist = BDX_BITS(5) + 1;
            BDX_DROP(5);
            const uint32_t hclen = BDX_BITS(4) + 4;
            BDX_DROP(4);
            if (hlit > 286 || hdist > 30) return false;
            uint8_t pl[kNumPre] = {0};
            for (uint32_t i = 0; i < hclen; ++i) {
                if (bitsleft < 3) BDX_REFILL();
                pl[kPreOrder[i]] = (uint8_t)BDX_BITS(3);
                BDX_DROP(3);
            }
            Entry pre[(1 << 7) + 2 * kNumPre];
            if (!build_table(pl, kNumPre, 7, pre, (int)(sizeof(pre) / sizeof(Entry)), pre_entry, false)) return false;
            uint8_t lens[kNumLit + kNumDist + 140];
            uint32_t n = 0;
            const uint32_t total = hlit + hdist;
            while (n < total) {
                if (in > in_end + 8) return false;
                BDX_REFILL();
                const Entry e = pre[BDX_BITS(7)];
                if ((e.kx >> 5) != K_LITERAL) return false;
                BDX_DROP(e.len);
                const uint32_t sym = e.value;
                if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (n == 0) return false;
                    val = lens[n - 1];
                    rep = 3 + BDX_BITS(2);
                    BDX_DROP(2);
                } else if (sym == 17) {
                    rep = 3 + BDX_BITS(3);
                    BDX_DROP(3);
                } else {
                    rep = 11 + BDX_BITS(7);
                    BDX_DROP(7);
                }
                if (n + rep > total) return false;
                memset(lens + n, (int)val, rep);
                n += rep;
            }
            if (lens[256] == 0) return false;  // no end-of-block code
            uint8_t ll[kNumLit] = {0}, dl[kNumDist] = {0};
            memcpy(ll, lens, hlit);
            memcpy(dl, lens + hlit, hdist);
            if (!build_table(ll, kNumLit, kLitBits, T.lit, (int)(sizeof(T.lit) / sizeof(Entry)), litlen_entry, false)) return false;
            if (!build_table(dl, kNumDist, kDistBits, T.dist, (int)(sizeof(T.dist) / sizeof(Entry)), dist_entry, true)) return false;
            tb = &T;
        }

        // ---- the block's symbols ----
        const Entry* const lt = tb->lit;
        const Entry* const dt = tb->dist;
        for (;;) {
            if (in > in_end + 8) return false;
            BDX_REFILL();
            Entry e = lt[BDX_BITS(kLitBits)];
            if ((e.kx >> 5) == K_SUB) {
                BDX_DROP(e.len);
                e = lt[e.value + BDX_BITS(e.kx & 31)];
            }
            BDX_DROP(e.len);
            unsigned kind = e.kx >> 5;
            if (kind == K_LITERAL) {
                if (out >= out_end) return false;
                *out++ = (uint8_t)e.value;
                // up to two more literals on the bits already in the buffer (>= 56 - 15 left)
                e = lt[BDX_BITS(kLitBits)];
                if ((e.kx >> 5) == K_LITERAL && out < out_end) {
                    BDX_DROP(e.len);
                    *out++ = (uint8_t)e.value;
                    e = lt[BDX_BITS(kLitBits)];
                    if ((e.kx >> 5) == K_LITERAL && out < out_end) {
                        BDX_DROP(e.len);
                        *out++ = (uint8_t)e.value;
                    }
                }
                continue;
            }
            if (kind == K_EOB) break;
            if (kind != K_LENGTH) return false;
            const unsigned lx = e.kx & 31;
            const uint32_t length = e.value + BDX_BITS(lx);
            BDX_DROP(lx);
            Entry d = dt[BDX_BITS(kDistBits)];
            if ((d.kx >> 5) == K_SUB) {
                BDX_DROP(d.len);
                d = dt[d.value + BDX_BITS(d.kx & 31)];
            }
            if ((d.kx >> 5) != K_LENGTH) return false;
            BDX_DROP(d.len);
            const unsigned dx = d.kx & 31;
            const uint32_t dist = d.value + BDX_BITS(dx);
            BDX_DROP(dx);
            if (dist > (size_t)(out - out_begin) || length > (size_t)(out_end - out)) return false;
            const uint8_t* src = out - dist;
            uint8_t* dst = out;
            out += length;
            if ((size_t)(out_end - dst) + out_slack >= length + 8 && dist >= 8) {
                // eight bytes at a time; may write up to seven bytes past the match (slack, or bytes decoded next)
                uint8_t* const stop = dst + length;
                do {
                    store64(dst, load64(src));
                    dst += 8; src += 8;
                } while (dst < stop);
            } else if (dist == 1 && (size_t)(out_end - dst) + out_slack >= length + 8) {
                const uint64_t v = 0x0101010101010101ull * src[0];
                uint8_t* const stop = dst + length;
                do {
                    store64(dst, v);
                    dst += 8;
                } while (dst < stop);
            } else {
                for (uint32_t i = 0; i < length; ++i) dst[i] = src[i];
            }
        }
        if (final) break;
    }
#undef BDX_REFILL
#undef BDX_BITS
#undef BDX_DROP
    // exactly the stated output, and no more input consumed than there was
    if (out != out_end) return false;
    const size_t consumed_bits = (size_t)(in - in_begin) * 8 - bitsleft;
    return consumed_bits <= in_len * 8;
}


namespace {
struct CrcTables {
    uint32_t t[16][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 16; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFFu];
    }
};
const CrcTables& crc_tables() { static const CrcTables t; return t; }
}  // namespace

uint32_t crc32_fast(const uint8_t* p, size_t n) {
    const CrcTables& T = crc_tables();
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 16) {
        uint32_t a, b, d, e;
        memcpy(&a, p, 4); memcpy(&b, p + 4, 4); memcpy(&d, p + 8, 4); memcpy(&e, p + 12, 4);
        a ^= c;
        c = T.t[15][a & 0xFFu] ^ T.t[14][(a >> 8) & 0xFFu] ^ T.t[13][(a >> 16) & 0xFFu] ^ T.t[12][a >> 24] ^
            T.t[11][b & 0xFFu] ^ T.t[10][(b >> 8) & 0xFFu] ^ T.t[9][(b >> 16) & 0xFFu] ^ T.t[8][b >> 24] ^
            T.t[7][d & 0xFFu] ^ T.t[6][(d >> 8) & 0xFFu] ^ T.t[5][(d >> 16) & 0xFFu] ^ T.t[4][d >> 24] ^
            T.t[3][e & 0xFFu] ^ T.t[2][(e >> 8) & 0xFFu] ^ T.t[1][(e >> 16) & 0xFFu] ^ T.t[0][e >> 24];
        p += 16; n -= 16;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

}  // namespace bdhost
