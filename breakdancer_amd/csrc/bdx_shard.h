// Device side of a chromosome-sharded run (bdx_dist_*): ONE context per rank holds all of the rank's chromosomes, in ascending
// order, and runs the single-context launch sequence over them.  What differs from a run over the whole genome is small and
// per chromosome -- the running counters a chromosome starts with, the read that closes its last candidate region, its first
// region's genome-wide id -- and lives in tables indexed by tid that these kernels build and apply (k9_shard.hip), plus the
// exchange of the inter-chromosomal join records and the name census (k7_exchange.hip).
#pragma once
#include "bdx_k3.h"

namespace bdx {

// ---- per-chromosome tables ------------------------------------------------------------------------------------------
// where each chromosome starts in the context's stream and what the tile-total columns (anomalous, normal-leftmost, proper per key)
// have counted up to there: out[t][0] = first read with tid >= t, out[t][1 + c] = column c's count of the reads before it,
// t = 0 .. ntids; out[(ntids + 1) * (1 + ncols)] = error bits (1: a negative tid, 2: a tid >= ntids)
struct TidTableParams {
    const int32_t* tid;
    const uint8_t* lib;
    const uint8_t* cls;
    uint64_t n;
    uint32_t ntiles, tstride;
    int ntids, nkeys, nlibs, ncols;
    const DevLib* libs;
    const uint32_t *tile_tot, *tile_pre, *chunk_base;
    uint32_t chunk_super;
    const Pass1* p1;
    uint32_t* out;
};
void launch_k9_tid_table(const TidTableParams& p, hipStream_t s);
void launch_k9_check_sorted(const int32_t* tid, uint64_t n, int ntids, uint32_t* err, hipStream_t s);
// one thread behind a kernel boundary: *flag = value (the host polls the pinned word)
void launch_k9_signal(uint32_t* flag, uint32_t value, hipStream_t s);
// A handful of small tables from the rank's pinned report area into HBM (and small fills) in ONE launch: each of them as a copy or fill
// command of its own is a 5 us blit kernel with a gap in front of it (six of them between pass 1 and the compaction of a sharded run).
struct UploadList {
    uint32_t* dst[8];
    const uint32_t* src[8];   // pinned host memory, read by the kernel (null: fill with `value`)
    uint32_t words[8];
    uint32_t value[8];
    int n;
    void copy(void* d, const void* s_, size_t w) { dst[n] = (uint32_t*)d; src[n] = (const uint32_t*)s_; words[n] = (uint32_t)w; value[n] = 0; ++n; }
    void fill(void* d, uint32_t v, size_t w) { dst[n] = (uint32_t*)d; src[n] = nullptr; words[n] = (uint32_t)w; value[n] = v; ++n; }
};
void launch_k9_upload(const UploadList& l, hipStream_t s);
// ... after n words of device memory have been copied into the (pinned) report area
void launch_k9_report(const uint32_t* src, uint32_t* dst, uint32_t n, uint32_t* flag, uint32_t value, hipStream_t s);
// compact records: the counters of chromosome t's reads get tid_off[t][0] (normal pairs) and tid_off[t][1 + k] (proper reads of key k)
// added -- what the chromosomes in front of t (anybody's) have counted, minus what this context's own have; first_tab[t] = {1, read
// length, normal-pair count} of chromosome t's first anomalous read (pinned host memory, zero on entry)
void launch_k9_rebase(const Compact& cp, const uint32_t* n_ptr, uint32_t n_upper, int nkeys, const uint32_t* tid_off, uint32_t* first_tab, hipStream_t s);   // (first_tab may be null)
// What a rank can say about its compact records BEFORE it knows anything of the other ranks (it rides in the run's first all-reduce):
// first_tab[t] = {1, read length, normal-pair count in THIS context's stream} of chromosome t's first anomalous read (pinned host memory,
// zero on entry); cnt_mtid[mt] += inter-chromosomal reads whose mate lies on the LATER chromosome mt (whoever owns it: the host sorts them
// by owner once the owners are known); cnt_owner[q] += anomalous reads whose name key belongs to rank q's census (world > 1 only)
struct FirstCountsParams {
    Compact cp;
    const int32_t* mtid_col;
    const uint32_t* n_ptr;
    int32_t ntids;
    uint32_t world;          // 1: only first_tab is filled
    uint32_t* first_tab;
    uint32_t* cnt_mtid;      // [ntids], zero on entry
    uint32_t* cnt_owner;     // [world], zero on entry
};
void launch_k9_first_counts(const FirstCountsParams& p, uint32_t n_upper, hipStream_t s);
// Records that arrive in per-rank blocks of one buffer (an all-to-all's receive buffer, a gather's): segment q holds records
// [start[q], start[q + 1]) at base + off[q] (off in 64-bit words); a kernel finds record j's segment by a short search
struct SegList {
    uint64_t off[kMaxRanks];
    uint32_t start[kMaxRanks + 1];
    int n;
    __host__ __device__ __forceinline__ int seg_of(uint32_t j) const {
        int q = 0;
        while (q + 1 < n && j >= start[q + 1]) ++q;
        return q;
    }
};
// out[t] = first of the context's regions with tid >= t (t = 0 .. ntids), out[ntids + 1] = its region count, out[ntids + 2] = last_maxq
void launch_k9_tid_regions(const RegionRec* r_rec, const StageCounts* counts, int ntids, uint32_t* out, hipStream_t s);

// the context's regions take their genome-wide ids: region_of[j] += roff[tid]; record r goes to rg_rec[r + roff[tid]] (rg_rec zero
// elsewhere: a region with n == 0 is another rank's); K6's per-region scratch and the taint bytes get their start values over [0, cap)
struct GlobalizeParams {
    const int32_t* tid;         // compact reads
    int32_t* region_of;
    const uint32_t* n_ptr;
    const RegionRec* r_rec;     // the context's dense table
    const uint32_t* r_pk;
    uint32_t nr_local;
    const uint32_t* roff;       // [ntids]
    RegionRec* rg_rec;
    uint32_t* rg_pk;
    int nkeys2;
    uint32_t* scratch;          // [6][cap]
    uint32_t cap;
    StageCounts* counts;
    uint32_t nr_global;
    int32_t last_maxq;
};
void launch_k9_globalize(const GlobalizeParams& p, uint32_t n_upper, hipStream_t s);
// the read length _max_readlen holds at the flush of window w is that of region (w + 1) period - 1 (BreakDancer.cpp:254-259), whoever
// owns it: every rank enters its own into win[w] (else 0), the table is all-reduced, and the other ranks' values are entered into the
// empty places of rg_rec afterwards
void launch_k9_window_collect(const RegionRec* rg_rec, uint32_t nr, uint32_t period, unsigned long long* win, hipStream_t s);
void launch_k9_window_apply(RegionRec* rg_rec, uint32_t nr, uint32_t period, const unsigned long long* win, hipStream_t s);
// replay / support route: the compact records as five 64-bit words each {key, region | meta << 32, |isize| | tid << 32, check,
// index of the read in its chromosome's stream}
void launch_k9_pack_replay(const Compact& cp, const int32_t* region_of, const uint32_t* n_ptr, uint32_t n_upper, const uint32_t* tid_start,
                           unsigned long long* out, hipStream_t s);

// ---- the exchange (k7_exchange.hip) ----------------------------------------------------------------------------------
struct ExchangeSrc {
    const uint64_t* key;
    const uint64_t* check;      // may be null
    const uint32_t* meta;
    const int32_t* tid;         // compact
    const uint32_t* idx;        // compact: index in the resident stream
    const int32_t* mtid_col;    // the resident mtid column
    const int32_t* region_of;   // genome-wide ids
    const uint32_t* n_ptr;
    const int32_t* owner_of_tid;  // [ntids] rank that holds the chromosome, -1: nobody (it has no reads)
    int32_t ntids, me;
    uint32_t world;
    uint8_t* taint;             // [regions] a region that sends a record to another rank is part of a component that spans ranks (may be null)
};
void launch_k7_count(const ExchangeSrc& x, uint32_t n_upper, uint32_t* cnt, hipStream_t s);
void launch_k7_scatter(const ExchangeSrc& x, uint32_t n_upper, uint32_t* cursor, ExchangeEntry* out, unsigned long long* names_out, hipStream_t s);
void launch_k7_unpack(const ExchangeEntry* in, uint32_t n, uint64_t* key, uint64_t* check, int32_t* region, const uint32_t* n_local, uint32_t* n_total,
                      hipStream_t s);
// the same from the per-rank blocks of ONE all-to-all's receive buffer (base: 64-bit words; the segments' records are ExchangeEntry)
void launch_k7_unpack_seg(const unsigned long long* base, const SegList& sg, uint32_t n, uint64_t* key, uint64_t* check, int32_t* region, const uint32_t* n_local,
                          uint32_t* n_total, hipStream_t s);
void launch_k7_names_census_seg(const unsigned long long* base, const SegList& sg, uint32_t n, unsigned long long* slots, uint32_t nslots, uint32_t* irregular, hipStream_t s);
// window read lengths through the same all-to-all: every rank appends {window << 32 | read length} of the windows whose last region is its
// own to every other rank's block (k9_window_pack: dst[q] = the place in block q, null for the rank itself; n_expected entries each), and
// enters what it receives into the empty places of its region table (k9_window_unpack)
struct WindowDst { unsigned long long* dst[kMaxRanks]; int world; };
void launch_k9_window_pack(const RegionRec* rg_rec, uint32_t nr, uint32_t period, const WindowDst& wd, uint32_t n_expected, uint32_t* cursor, hipStream_t s);
void launch_k9_window_unpack(RegionRec* rg_rec, uint32_t nr, uint32_t period, const unsigned long long* base, const SegList& sg, uint32_t n, uint32_t* err, hipStream_t s);
void launch_k7_names_clear(unsigned long long* slots, uint32_t nslots, uint32_t* irregular, hipStream_t s);
void launch_k7_names_census(const unsigned long long* in, uint32_t n, unsigned long long* slots, uint32_t nslots, uint32_t* irregular, hipStream_t s);
uint32_t k7_names_slots(size_t records);   // the census table for so many sightings (1.5 slots each)

// ---- rank 0: the ranks' SV tables -> one table --------------------------------------------------------------------------
// Every rank's table is sorted by order key (K6Arrays::sv_key) and two ranks never hold the same key (a key names the traversal's
// start vertex, which one rank walked), so a row's place in the merged table is its index plus the rows with smaller keys in the
// other tables.  A package (one per rank, byte offsets into the gather buffer): rows, keys, then the flat lists.
struct TablePackage { uint64_t rows_off, keys_off, lib_index_off, lib_pairs_off, ltail_off, cn_key_off, cn_value_off; uint32_t n_sv, n_terms, n_cn; };
struct TableDesc { TablePackage p[kMaxRanks]; int world; };
struct MergeOut {
    SvOut* sv_out;            // pinned host, like a single-context run's
    int32_t* lib_index;
    int32_t* lib_pairs;
    double* ltail;
    int32_t* cn_key;
    float* cn_value;
};
// D: the descriptor in device memory; ws: scan workspace words (device); src: [n_total] (rank << 26 | row) by final position
void launch_k9_merge_tables(const char* all, const TableDesc* D, int world, uint32_t n_total, uint32_t max_n, uint32_t* src, uint2* begins, uint32_t* ws,
                            const uint32_t* n_dev, const MergeOut& out, hipStream_t s);

// ---- rank 0: the pair groups of the components no rank could walk alone -> K6's `in_groups` input --------------------------
// The gathered groups (any order) bucketed by their later region: goff[r] .. goff[r + 1] are region r's places in `out` -- and its
// place in K6's slot space: in a context that takes its pair groups as aggregates a region's parts and staging slots live in a slot
// space of one slot per group (K6Arrays::in_groups / first_of), not in the compact read list of the rank that cut the region.
// cnt / cur: [nr + 1] words each, scan_ws: 2 * (scan_grid(nr + 1) + 2) words; n_words[0] receives nr + 1; err: set to 1 if a group names
// a region outside [0, nr)
void launch_k9_pk_rows(const GroupRec* groups, uint32_t ng, const uint32_t* pk_dev, uint32_t* pk_host, uint32_t row_words, uint32_t nr, hipStream_t s);
void launch_k9_bucket_groups(const unsigned long long* base, const SegList& sg, uint32_t n, uint32_t nr, uint32_t* cnt, uint32_t* goff, uint32_t* cur, GroupRec* out,
                             uint32_t* scan_ws, uint32_t* n_words, uint32_t* err, hipStream_t s);   // (in: the ranks' blocks of one gather buffer)

// one no-op launch per translation unit (see bdx_warm_up)
void warm_k1(hipStream_t s); void warm_k2(hipStream_t s); void warm_k3(hipStream_t s); void warm_k4(hipStream_t s);
void warm_k5(hipStream_t s); void warm_k6(hipStream_t s); void warm_k7(hipStream_t s); void warm_k9(hipStream_t s);

}  // namespace bdx
