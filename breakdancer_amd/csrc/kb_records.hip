// KB -- BAM records on the GPU: where the records of the inflated stream start, and their fields as SoA columns.
//
// Stands where the host producer has column_reader.cpp (decode_piece) + bam_reader.cpp (parse_record,
// bam_guess_record_start), and the reference bam_read1 + Alignment's constructor + the RG -> library lookup
// (io/Alignment.cpp:12-29,45-64, io/AlignmentSource.hpp:48-65, io/BamConfig.hpp:62-72, io/AlignmentFilter.hpp:24-34).
//
// A record's position follows from the sizes of all records before it -- a serial chain over the whole file.  It is cut at
// the BGZF block boundaries the way the host reader cuts it at piece boundaries: every block GUESSES where its first record
// starts (the first offset at which three records in a row have plausible fields, a fitting size equation and a printable,
// NUL-terminated name -- 64 offsets tested at a time, one per lane), walks the chain of size words from there, and a
// stitching pass compares every block's guess with the end of its predecessor's chain; a block whose guess is wrong or
// missing is walked again from the true boundary.  The result is exact whatever the guesses were; they only decide
// how much is walked twice.  Then one lane per record reads the fields, hashes the name, scans the aux block for RG / AM
// and applies the reader's filter, and an ordered compaction appends the kept records to the destination columns.
#include <hip/hip_runtime.h>

#include "bdx_bam_dev.h"

namespace bdx {

namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// Does a BAM record start at u[p]?  (host/bam_reader.cpp plausible_record; the size bound is the device path's)
__device__ __forceinline__ bool plausible_record(const uint8_t* u, uint64_t p, uint64_t avail_end, int32_t n_targets, uint64_t* next) {
    if (p + 36 > avail_end) return false;
    const uint8_t* r = u + p;
    const uint32_t bs = ld32(r);
    if (bs < 32 || bs > kMaxDeviceRecord) return false;
    const int32_t tid = (int32_t)ld32(r + 4), rpos = (int32_t)ld32(r + 8);
    const uint32_t l_name = r[12], n_cigar = ld16(r + 16);
    const int32_t l_seq = (int32_t)ld32(r + 20), mtid = (int32_t)ld32(r + 24), mpos = (int32_t)ld32(r + 28);
    if (tid < -1 || tid >= n_targets || mtid < -1 || mtid >= n_targets || rpos < -1 || mpos < -1 || l_seq < 0 || l_name < 1) return false;
    const uint64_t need = 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > bs) return false;
    if (p + 36 + l_name > avail_end) return false;
    const uint8_t* nm = r + 36;
    if (nm[l_name - 1] != 0) return false;
    for (uint32_t i = 0; i + 1 < l_name; ++i)
        if (nm[i] < 33 || nm[i] > 126) return false;
    *next = p + 4 + (uint64_t)bs;
    return true;
}

// the chain of size words from block-relative offset `from`: record starts -> offs[], until the block's end
__device__ __forceinline__ void walk_block(const uint8_t* u, uint64_t ubeg, uint32_t ulen, uint64_t avail_end, uint32_t from, uint16_t* offs,
                                           bool store, ChainBlock* out) {
    uint64_t p = from;
    uint32_t k = 0, bad = 0;
    while (p < ulen) {
        if (ubeg + p + 4 > avail_end) { bad = 2; break; }   // the size word is not there (yet): too long a record, or a truncated file
        const uint32_t bs = ld32(u + ubeg + p);
        if (bs < 32) { bad = 1; break; }
        if (bs > kMaxDeviceRecord) { bad = 3; break; }   // (from a true boundary: a record longer than this path spans -- the host reader's case)
        if (store) offs[k] = (uint16_t)p;
        ++k;
        p += 4 + (uint64_t)bs;
    }
    out->count = k;
    out->end = (uint32_t)(p > 0xFFFFFFFFull ? 0xFFFFFFFFull : p);
    out->bad = bad;
}

// one wave per BGZF block: guess, then walk
__global__ __launch_bounds__(64) void kb_chain_kernel(const uint8_t* __restrict__ u, const BgzfBlock* __restrict__ blocks, uint32_t nblk,
                                                      uint64_t avail_end, int32_t n_targets, ChainBlock* __restrict__ cb, uint16_t* __restrict__ offs) {
    const uint32_t b = blockIdx.x;
    if (b >= nblk) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t ubeg = blocks[b].out_off;
    const uint32_t ulen = blocks[b].out_len;
    uint32_t guess = kNoGuess;
    for (uint32_t base = 0; base < ulen; base += 64) {
        const uint32_t o = base + lane;
        bool ok = false;
        if (o < ulen) {
            uint64_t a, c, d;
            ok = plausible_record(u, ubeg + o, avail_end, n_targets, &a) && plausible_record(u, a, avail_end, n_targets, &c) &&
                 plausible_record(u, c, avail_end, n_targets, &d);
        }
        const uint64_t m = __ballot(ok);
        if (m) { guess = base + (uint32_t)__builtin_ctzll(m); break; }
    }
    ChainBlock r{};
    r.guess = guess;
    if (guess != kNoGuess) walk_block(u, ubeg, ulen, avail_end, guess, offs + (size_t)b * kRecSlots, lane == 0, &r);
    r.guess = guess;
    if (lane == 0) cb[b] = r;
}

// One workgroup: the guesses against the chain of true boundaries, then the record numbering of the piece.
constexpr int kStitchThreads = 1024;
__global__ __launch_bounds__(kStitchThreads) void kb_stitch_kernel(const uint8_t* __restrict__ u, const BgzfBlock* __restrict__ blocks, uint32_t nblk,
                                                                   uint64_t avail_end, int is_last, ChainBlock* cb, uint16_t* offs, uint32_t* rec_base,
                                                                   PieceState* st, const uint32_t* __restrict__ inflate_status, uint64_t rebase_from,
                                                                   uint64_t rebase_to) {
    __shared__ uint32_t s_first_bad;
    __shared__ uint32_t s_inflate_bad;
    __shared__ uint32_t s_sum[kStitchThreads];
    const uint32_t t = threadIdx.x;
    if (t == 0) { s_first_bad = nblk; s_inflate_bad = 0; }
    __syncthreads();
    for (uint32_t b = t; b < nblk; b += kStitchThreads)
        if (inflate_status[b]) atomicOr(&s_inflate_bad, 1u);
    // (a piece that starts at the ring's front again: the carried boundary is an offset behind its predecessor's end)
    const uint64_t start = st->next_start - rebase_from + rebase_to;
    // block b is fine if its walk started where its predecessor's ended (by induction the whole prefix of fine blocks is exact)
    for (uint32_t b = t; b < nblk; b += kStitchThreads) {
        const ChainBlock c = cb[b];
        uint64_t expect;
        if (b == 0) expect = start;
        else expect = blocks[b - 1].out_off + (uint64_t)cb[b - 1].end;
        const bool fine = c.guess != kNoGuess && c.bad == 0 && blocks[b].out_off + c.guess == expect && (b == 0 || (cb[b - 1].guess != kNoGuess && cb[b - 1].bad == 0));
        if (!fine) atomicMin(&s_first_bad, b);
    }
    __syncthreads();
    const uint32_t first_bad = s_first_bad;
    if (t == 0) {
        uint64_t expect = first_bad == 0 ? start : blocks[first_bad - 1].out_off + (uint64_t)cb[first_bad - 1].end;
        uint32_t error = s_inflate_bad ? 5u : 0u, redo = 0;   // (a block that did not inflate: nothing behind it can be trusted)
        for (uint32_t b = first_bad; b < nblk && !error; ++b) {
            const uint64_t beg = blocks[b].out_off, end = beg + blocks[b].out_len;
            ChainBlock c = cb[b];
            if (expect >= end) {  // a record that began earlier covers the whole block
                c.count = 0; c.end = (uint32_t)(expect - beg); c.bad = 0; c.guess = kNoGuess;
                cb[b] = c;
                continue;
            }
            if (!(c.guess != kNoGuess && beg + c.guess == expect)) {
                c.guess = (uint32_t)(expect - beg);
                walk_block(u, beg, blocks[b].out_len, avail_end, c.guess, offs + (size_t)b * kRecSlots, true, &c);
                cb[b] = c;
                ++redo;
            }
            // from a TRUE boundary: a corrupt or truncated file, or too long a record -- unless the stream was cut here on purpose
            // (is_last == 2: the size word of the record behind the cut is simply not there)
            if (c.bad && !(is_last == 2 && c.bad == 2)) error = c.bad == 2 && is_last ? 4 : (c.bad == 1 ? 1 : 2);
            expect = beg + (uint64_t)c.end;
        }
        if (first_bad == nblk && nblk) expect = blocks[nblk - 1].out_off + (uint64_t)cb[nblk - 1].end;
        if (nblk == 0) expect = start;
        // (a fine prefix may still end in a walk that ran out of bytes)
        st->next_start = expect;
        st->redo += redo;
        if (is_last == 2 && !error && expect > avail_end) {
            // The caller stopped reading behind its region (a .bai seek): the record that begins in the last member handed over and ends
            // behind it is not part of the result -- writers that do not align records to members (htsjdk, sambamba) produce one at
            // nearly every cut.  It is the last record counted: taken back, no error.
            for (uint32_t b = nblk; b-- > 0;)
                if (cb[b].count) { cb[b].count -= 1; break; }
            expect = avail_end;
            st->next_start = expect;
        }
        if (is_last == 1 && !error && expect != avail_end) error = 4;   // the last record is cut off
        // (a record that begins in this batch and ends behind its successor: its tail -- the tags -- is not inflated yet.  Batches of the
        // default size hold a hundred times the longest record this path takes; a caller's tiny batches can be outgrown)
        if (!is_last && !error && expect > avail_end) error = 2;
        if (error && !st->error) st->error = error;
    }
    __syncthreads();
    // exclusive prefix of the blocks' record counts
    const uint32_t per = (nblk + kStitchThreads - 1) / kStitchThreads;
    const uint32_t lo = t * per, hi = min(nblk, lo + per);
    uint32_t mine = 0;
    for (uint32_t b = lo; b < hi; ++b) mine += cb[b].count;
    s_sum[t] = mine;
    __syncthreads();
    for (int off = 1; off < kStitchThreads; off <<= 1) {
        const uint32_t v = t >= (uint32_t)off ? s_sum[t - off] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[t] - mine;
    for (uint32_t b = lo; b < hi; ++b) { rec_base[b] = run; run += cb[b].count; }
    if (t == kStitchThreads - 1) {
        rec_base[nblk] = s_sum[t];
        st->piece_raw = s_sum[t];
    }
}

// ---- fields ----
__device__ __forceinline__ uint64_t hash_bytes(const uint8_t* s, uint32_t n) {   // host/bam_reader.cpp hash_name
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) h = name_hash_step(h, ld64(s + i));
    uint64_t w = 0;
    for (uint32_t k = 0; i + k < n; ++k) w |= (uint64_t)s[i + k] << (8 * k);
    return name_hash_finish(h, w);
}

// key and second hash of a read name in one pass over its bytes
__device__ __forceinline__ void hash_name_pair(const uint8_t* s, uint32_t n, uint64_t& key, uint64_t& check) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n, g = name_check_seed(n);
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const uint64_t w = ld64(s + i);
        h = name_hash_step(h, w);
        g = name_check_step(g, w);
    }
    uint64_t w = 0;
    for (uint32_t k = 0; i + k < n; ++k) w |= (uint64_t)s[i + k] << (8 * k);
    key = name_hash_finish(h, w);
    check = name_check_finish(g, w);
}

__device__ __forceinline__ uint8_t resolve_library(const RgTable& rg, const uint8_t* s, uint32_t n, bool have) {
    if (!have) return rg.missing;   // no RG tag
    const uint64_t h = hash_bytes(s, n);
    for (uint32_t i = 0; i < rg.n; ++i) {
        if (rg.hash[i] != h) continue;
        const uint32_t o = rg.off[i], l = rg.off[i + 1] - o;
        if (l != n) continue;
        bool same = true;
        for (uint32_t k = 0; k < n && same; ++k) same = (uint8_t)rg.chars[o + k] == s[k];
        if (same) return rg.lib[i];
    }
    return rg.fallback;
}

// one wave per BGZF block, one lane per record
__global__ __launch_bounds__(64) void kb_extract_kernel(const uint8_t* __restrict__ u, const BgzfBlock* __restrict__ blocks, uint32_t nblk,
                                                        const ChainBlock* __restrict__ cb, const uint16_t* __restrict__ offs,
                                                        const uint32_t* __restrict__ rec_base, RgTable rg, RecordFilterDev f, RawColumns raw,
                                                        PieceState* st) {
    const uint32_t b = blockIdx.x;
    if (b >= nblk) return;
    if (st->error) return;
    const uint32_t count = cb[b].count;
    const uint64_t ubeg = blocks[b].out_off;
    const uint32_t base = rec_base[b];
    for (uint32_t k = threadIdx.x; k < count; k += 64) {
        const uint8_t* rec = u + ubeg + offs[(size_t)b * kRecSlots + k];
        const uint32_t bs = ld32(rec);
        const uint8_t* p = rec + 4;
        const uint8_t* end = p + bs;
        const int32_t tid = (int32_t)ld32(p), pos = (int32_t)ld32(p + 4);
        const uint32_t l_read_name = p[8];
        const uint32_t mapq = p[9];
        const uint32_t n_cigar = ld16(p + 12), flag = ld16(p + 14);
        const int32_t l_qseq = (int32_t)ld32(p + 16);
        const int32_t mtid = (int32_t)ld32(p + 20), mpos = (int32_t)ld32(p + 24), isize = (int32_t)ld32(p + 28);
        const uint32_t r = base + k;
        if (l_qseq < 0 || 32ull + l_read_name + 4ull * n_cigar + ((uint64_t)l_qseq + 1) / 2 + (uint64_t)l_qseq > bs) {
            atomicMax(&st->error, 1u);   // "corrupt BAM record" (column_reader.cpp decode_piece)
            raw.keep[r] = 0;
            continue;
        }
        // reader filter: primary, placed (io/AlignmentFilter.hpp:24-34, io/BamIo.cpp:11-18); -o keeps one region
        bool keep = f.keep_all || (!(flag & (0x100u | 0x800u)) && tid >= 0);
        if (f.only_tid >= 0) {
            if (tid < 0 || tid > f.only_tid || (tid == f.only_tid && pos >= f.end)) st->past_region = 1;   // (benign race: all write 1)
            if (keep) {
                int32_t endp = pos;   // samtools bam_calend (bam.c:17-45)
                const uint8_t* cg = p + 32 + l_read_name;
                for (uint32_t c = 0; c < n_cigar; ++c) {
                    const uint32_t w = ld32(cg + 4 * c), op = w & 15, len = w >> 4;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) endp += (int32_t)len;
                }
                if (!n_cigar) endp = pos + 1;
                keep = tid == f.only_tid && (uint32_t)endp > (uint32_t)f.beg && (uint32_t)pos < (uint32_t)f.end;
            }
        }
        raw.keep[r] = keep ? 1 : 0;
        if (!keep) continue;
        // aux sweep: RG:Z and AM:<int> (bam_aux_get / bam_aux2i; host/bam_reader.cpp parse_record)
        const uint8_t* q = p + 32 + l_read_name + 4 * (size_t)n_cigar + ((size_t)l_qseq + 1) / 2 + (size_t)l_qseq;
        const uint8_t* rgp = nullptr;
        uint32_t rgl = 0;
        uint32_t bdqual = mapq;
        bool have_am = false;
        while (q + 3 <= end) {
            const uint8_t t0 = q[0], t1 = q[1], ty = q[2];
            q += 3;
            const size_t left = (size_t)(end - q);
            size_t sz = 0;
            long ival = 0;
            bool is_int = false;
            bool fixed = true;
            switch (ty) {
                case 'A': sz = 1; break;
                case 'c': sz = 1; if (left >= 1) { ival = (int8_t)q[0]; is_int = true; } break;
                case 'C': sz = 1; if (left >= 1) { ival = q[0]; is_int = true; } break;
                case 's': sz = 2; if (left >= 2) { ival = (int16_t)ld16(q); is_int = true; } break;
                case 'S': sz = 2; if (left >= 2) { ival = (long)ld16(q); is_int = true; } break;
                case 'i': sz = 4; if (left >= 4) { ival = (int32_t)ld32(q); is_int = true; } break;
                case 'I': sz = 4; if (left >= 4) { ival = (long)ld32(q); is_int = true; } break;
                case 'f': sz = 4; break;
                case 'd': sz = 8; break;
                case 'Z':
                case 'H': {
                    fixed = false;
                    const uint8_t* z = q;
                    while (z < end && *z) ++z;
                    if (z >= end) { q = end; break; }
                    if (t0 == 'R' && t1 == 'G' && ty == 'Z' && !rgp) { rgp = q; rgl = (uint32_t)(z - q); }
                    q = z + 1;
                    break;
                }
                case 'B': {
                    fixed = false;
                    if (q + 5 > end) { q = end; break; }
                    const uint8_t sub = q[0];
                    const uint32_t cnt = ld32(q + 1);
                    const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    const size_t bytes = (size_t)cnt * es;
                    q = bytes > (size_t)(end - q - 5) ? end : q + 5 + bytes;
                    break;
                }
                default: fixed = false; q = end; break;
            }
            if (!fixed) continue;
            if (sz > left) { q = end; continue; }
            if (t0 == 'A' && t1 == 'M' && !have_am) { if (!f.mapq_only) bdqual = (uint8_t)(is_int ? ival : 0); have_am = true; }
            q += sz;
        }
        raw.tid[r] = tid; raw.pos[r] = pos; raw.mtid[r] = mtid; raw.mpos[r] = mpos; raw.isize[r] = isize;
        raw.flag[r] = (uint16_t)flag;
        raw.qlen[r] = (uint16_t)(l_qseq > 65535 ? 65535 : l_qseq);
        raw.mapq[r] = (uint8_t)bdqual;
        raw.lib[r] = resolve_library(rg, rgp, rgl, rgp != nullptr);
        uint64_t nk, nc;
        hash_name_pair(p + 32, l_read_name ? l_read_name - 1 : 0, nk, nc);
        raw.key[r] = nk;
        raw.check[r] = nc;
    }
}

// ---- ordered compaction of the kept records ----
constexpr int kCompactThreads = 256;
__global__ __launch_bounds__(kCompactThreads) void kb_compact_count_kernel(const uint8_t* __restrict__ keep, const PieceState* __restrict__ st, uint32_t* wg_cnt) {
    __shared__ uint32_t s_cnt;
    const uint32_t n = st->piece_raw;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * kCompactThreads + threadIdx.x;
    const bool k = i < n && keep[i];
    const uint64_t m = __ballot(k);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
    __syncthreads();
    if (threadIdx.x == 0) wg_cnt[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(1024) void kb_compact_scan_kernel(uint32_t* wg_cnt, uint32_t nwg_cap, uint64_t dst_cap, PieceState* st,
                                                               volatile uint64_t* progress, uint64_t sequence) {
    __shared__ uint32_t s_sum[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t n = st->piece_raw;
    const uint32_t nwg = min(nwg_cap, (n + kCompactThreads - 1) / kCompactThreads);
    const uint32_t per = (nwg + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(nwg, lo + per);
    uint32_t mine = 0;
    for (uint32_t b = lo; b < hi; ++b) mine += wg_cnt[b];
    s_sum[t] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= (uint32_t)off ? s_sum[t - off] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[t] - mine;
    for (uint32_t b = lo; b < hi; ++b) { const uint32_t c = wg_cnt[b]; wg_cnt[b] = run; run += c; }
    if (t == 1023) {
        const uint32_t total = s_sum[t];
        wg_cnt[nwg_cap] = total;                       // kept records of the piece
        const uint64_t before = st->n_kept;
        wg_cnt[nwg_cap + 1] = (uint32_t)(before & 0xFFFFFFFFu);
        wg_cnt[nwg_cap + 2] = (uint32_t)(before >> 32);
        if (!st->error && before + total > dst_cap) st->error = 3;
        wg_cnt[nwg_cap + 3] = st->error;
    }
}

__global__ __launch_bounds__(kCompactThreads) void kb_compact_scatter_kernel(RawColumns raw, DstColumns dst, uint8_t bam_index, const uint32_t* __restrict__ wg_cnt,
                                                                              uint32_t nwg_cap, const PieceState* __restrict__ st) {
    __shared__ uint32_t s_wave[kCompactThreads / 64];
    if (wg_cnt[nwg_cap + 3]) return;   // an error: nothing is appended
    const uint32_t n = st->piece_raw;
    const uint32_t i = blockIdx.x * kCompactThreads + threadIdx.x;
    const bool k = i < n && raw.keep[i];
    const uint64_t m = __ballot(k);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_wave[wave] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    if (!k) return;
    uint32_t off = wg_cnt[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) off += s_wave[w];
    off += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    const uint64_t before = (uint64_t)wg_cnt[nwg_cap + 1] | ((uint64_t)wg_cnt[nwg_cap + 2] << 32);
    const uint64_t d = before + off;
    dst.tid[d] = raw.tid[i]; dst.pos[d] = raw.pos[i]; dst.mtid[d] = raw.mtid[i]; dst.mpos[d] = raw.mpos[i]; dst.isize[d] = raw.isize[i];
    dst.flag[d] = raw.flag[i]; dst.qlen[d] = raw.qlen[i]; dst.mapq[d] = raw.mapq[i]; dst.lib[d] = raw.lib[i]; dst.bam[d] = bam_index;
    dst.key[d] = raw.key[i];
    if (dst.check) dst.check[d] = raw.check[i];
}

__global__ void kb_compact_finish_kernel(const uint32_t* __restrict__ wg_cnt, uint32_t nwg_cap, PieceState* st, volatile uint64_t* progress, uint64_t sequence) {
    const uint32_t err = wg_cnt[nwg_cap + 3];
    if (!err) st->n_kept += wg_cnt[nwg_cap];
    st->n_raw += st->piece_raw;
    if (progress) {
        progress[0] = st->n_kept;
        progress[1] = (uint64_t)st->error | ((uint64_t)st->past_region << 8) | ((uint64_t)st->redo << 32);
        progress[2] = st->n_raw;
        __threadfence_system();
        progress[3] = sequence;
    }
}

}  // namespace

// ---- merge of several files' records: a gather by the permutation the caller computed (the reference's merge order, ties included,
// is the host's business: BamMerger.cpp:40-61 through a priority queue) ----
__global__ __launch_bounds__(256) void kb_gather_kernel(GatherSources src, const uint8_t* __restrict__ src_file, const uint32_t* __restrict__ src_index, uint64_t n,
                                                        DstColumns dst, uint32_t* err) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int f = src_file[i];
    if (f >= src.k) { *err = 1; return; }
    // (the sources' pointers live in kernel arguments: a run-time index into them would go through scratch memory)
    GatherSource g = src.s[0];
#pragma unroll
    for (int q = 1; q < kMaxGatherSources; ++q)
        if (q == f) g = src.s[q];
    const uint64_t j = src_index[i];
    if (j >= g.n) { *err = 1; return; }
    dst.tid[i] = g.tid[j]; dst.pos[i] = g.pos[j]; dst.mtid[i] = g.mtid[j]; dst.mpos[i] = g.mpos[j]; dst.isize[i] = g.isize[j];
    dst.flag[i] = g.flag[j]; dst.qlen[i] = g.qlen[j]; dst.mapq[i] = g.mapq[j]; dst.lib[i] = g.lib[j]; dst.bam[i] = g.bam[j];
    dst.key[i] = g.key[j];
    if (dst.check) dst.check[i] = g.check[j];
}

void launch_kb_gather(const GatherSources& src, const uint8_t* src_file, const uint32_t* src_index, uint64_t n, DstColumns dst, uint32_t* err, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(kb_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_file, src_index, n, dst, err);
}

void launch_kb_chain(const uint8_t* u, const BgzfBlock* blocks, uint32_t nblk, uint64_t avail_end, int32_t n_targets, ChainBlock* cb,
                     uint16_t* offs, hipStream_t s) {
    if (!nblk) return;
    hipLaunchKernelGGL(kb_chain_kernel, dim3(nblk), dim3(64), 0, s, u, blocks, nblk, avail_end, n_targets, cb, offs);
}

void launch_kb_stitch(const uint8_t* u, const BgzfBlock* blocks, uint32_t nblk, uint64_t avail_end, int is_last, ChainBlock* cb, uint16_t* offs,
                      uint32_t* rec_base, PieceState* st, const uint32_t* inflate_status, uint64_t rebase_from, uint64_t rebase_to, hipStream_t s) {
    hipLaunchKernelGGL(kb_stitch_kernel, dim3(1), dim3(kStitchThreads), 0, s, u, blocks, nblk, avail_end, is_last, cb, offs, rec_base, st, inflate_status,
                       rebase_from, rebase_to);
}

void launch_kb_extract(const uint8_t* u, const BgzfBlock* blocks, uint32_t nblk, const ChainBlock* cb, const uint16_t* offs,
                       const uint32_t* rec_base, RgTable rg, RecordFilterDev f, RawColumns raw, PieceState* st, hipStream_t s) {
    if (!nblk) return;
    hipLaunchKernelGGL(kb_extract_kernel, dim3(nblk), dim3(64), 0, s, u, blocks, nblk, cb, offs, rec_base, rg, f, raw, st);
}

void launch_kb_compact(RawColumns raw, uint32_t raw_cap, DstColumns dst, uint64_t dst_cap, uint8_t bam_index, uint32_t* scan_ws, PieceState* st,
                       volatile uint64_t* progress, uint64_t sequence, hipStream_t s) {
    const uint32_t nwg = (raw_cap + kCompactThreads - 1) / kCompactThreads;   // scan_ws holds nwg + 4 words
    if (nwg) hipLaunchKernelGGL(kb_compact_count_kernel, dim3(nwg), dim3(kCompactThreads), 0, s, raw.keep, st, scan_ws);
    hipLaunchKernelGGL(kb_compact_scan_kernel, dim3(1), dim3(1024), 0, s, scan_ws, nwg, dst_cap, st, progress, sequence);
    if (nwg) hipLaunchKernelGGL(kb_compact_scatter_kernel, dim3(nwg), dim3(kCompactThreads), 0, s, raw, dst, bam_index, scan_ws, nwg, st);
    hipLaunchKernelGGL(kb_compact_finish_kernel, dim3(1), dim3(1), 0, s, scan_ws, nwg, st, progress, sequence);
}

}  // namespace bdx
