// Device arrays of the region-cut (K3), mate-join (K4) and scoring (K5) stages.
#pragma once
#include "../../include/bdx.h"
#include "bdx_dev.h"

namespace bdx {

struct U4;

struct StageCounts {
    uint32_t n_cand;
    uint32_t n_regions;
    int32_t last_maxq;   // max read length of the last candidate: BreakDancer::_max_readlen at the final flush
    uint32_t n_pairs;
    uint32_t n_groups;   // partial (flag, lib) aggregates written by K4
    uint32_t n_entries;  // reads of accepted regions entering the join
    uint32_t overflow;   // set if an output list ran out of capacity
    uint32_t n_slots;    // K6: 3 x n_regions SV slots
    uint32_t n_sv_dev;   // K6: SV candidates assembled on the device
    uint32_t n_terms_dev;  // K6: their (library, pairs) entries == Poisson terms
    uint32_t n_cn_dev;     // K6: their copy-number entries
    uint32_t n_groups_dev; // K6: region x region groups of the components handled on the device
    uint32_t pad[4];
};

struct RegionRec {
    int32_t tid, start, end;
    uint32_t n, rev, nonctx, nnormal;
    int32_t maxq;
    uint32_t first;  // compact index of the region's first read (its reads are [first, first + n))
};

struct GroupRec {  // partial aggregate of one (region_lo, region_hi, flag, lib) group
    uint64_t key;
    uint32_t pairs;
    uint32_t sum_isize;
};

struct K3Arrays {
    uint32_t cap;  // capacity of every array below (>= number of anomalous reads)
    // per compact read
    int32_t* cand;
    uint32_t *pre_q, *pre_rev, *pre_nonctx;
    // per candidate
    uint32_t* c_first;
    int32_t* c_maxq;
    uint32_t *c_accept, *c_n, *c_rev, *c_nonctx, *c_nnormal;
    int32_t* c_rid;
    int32_t* region_of;  // per compact read: accepted region id or -1
    // per accepted region (array-of-structs so one copy brings the table to the host)
    RegionRec* r_rec;
    uint32_t* r_pk;  // [cap][2*nkeys]: proper-read prefix counts at the region's first read (nkeys), then last read (nkeys)
    RegionRec* r_rec_dev;  // device-resident copies for K6 (r_rec / r_pk live in pinned host memory); may be null
    uint32_t* r_pk_dev;
    uint32_t* out_deg;     // [cap] K6 out-degree counters, zeroed by k3_region_of_kernel; may be null
    // scan workspace and totals
    U4* ws_u4;
    U4* head_total;
    uint32_t* ws_u32;
    uint32_t* acc_total;
    StageCounts* counts;
};

// the read that closes the last candidate when the stream continues in another context (next chromosome)
struct K3Tail {
    int has_next;
    int32_t qlen;
    uint32_t nn;
};

void launch_k3(const K3Arrays& a, const Compact& cp, const Pass1* p1, uint32_t n_anom_host, int min_len, int seq_coverage_lim,
               int nkeys, uint32_t nn_base, K3Tail tail, hipStream_t s);

// ---- K4 ---------------------------------------------------------------------------------------------
constexpr int kMaxBuckets = 8192;
constexpr int kJoinLdsSlots = 8192;  // 8192 x (8 B key + 4 B index) = 96 KiB of the CU's 160 KiB LDS
constexpr uint32_t kDirectJoinMax = 1u << 22;  // entries up to which the direct table (<= 128 MiB) is used
inline uint32_t direct_join_slots(uint32_t n) {
    uint32_t slots = 1024;
    while (slots < 4 * (uint64_t)n) slots <<= 1;
    return slots;
}
constexpr int kAggSlots = 4096;      // block-local group table: 4096 x 16 B = 64 KiB

struct K4Arrays {
    uint32_t nbuckets;   // power of two
    uint32_t log2b;
    uint32_t* bcnt;      // [nbuckets]
    uint32_t* boff;      // [nbuckets + 1]
    uint32_t* bcur;      // [nbuckets]
    uint64_t* e_key;     // [cap] bucketed entries
    uint32_t* e_idx;     // [cap]
    int32_t* partner;    // [cap] compact index of the mate, -1 if none
    uint64_t* t_key;     // [2*cap] global fallback table of the bucketed path; [t_mask + 1] table of the direct path
    int32_t* t_idx;
    uint32_t direct;     // 1: one open-addressing table for all entries (it stays in L2 / Infinity Cache), no partitioning
    uint32_t t_mask;     // direct path: slots - 1 (slots = power of two >= 4 x entries) of 64-bit words in t_key, all ones on
                         // entry; partner[] must be -1 on entry
    // output: partial aggregates of (r_lo, r_hi, flag, lib) -> (pairs, sum |isize|)
    GroupRec* g_rec;
    uint32_t g_cap;
};

__host__ __device__ __forceinline__ uint64_t group_pack(uint32_t rlo, uint32_t rhi, uint32_t lib, uint32_t flag) {
    return ((uint64_t)rlo << 38) | ((uint64_t)rhi << 12) | ((uint64_t)lib << 4) | (uint64_t)flag;
}
constexpr uint32_t kMaxRegions = (1u << 26) - 2;

// join input: one entry per anomalous read (region < 0: not in an accepted region, skipped)
struct Entries {
    const uint64_t* key;
    const int32_t* region;   // region id (global ids when the entries come from several shards)
    const uint32_t* order;   // position in the merged stream order; nullptr = the entry index itself
    const uint32_t* meta;    // flag | rev<<4 | lib<<8 | qlen<<16
    const int32_t* isize;    // |isize|
};

void launch_k4(const K4Arrays& k4, const Entries& e, const uint32_t* n_ptr, uint32_t n_upper, StageCounts* counts, hipStream_t s);

void launch_k4_join_only(const K4Arrays& k4, const Entries& e, const uint32_t* n_ptr, uint32_t n_upper, StageCounts* counts, hipStream_t s);

// ---- K5 ---------------------------------------------------------------------------------------------
void launch_k5(const double* lambda, const int32_t* k, double* out, uint32_t n, hipStream_t s);
// term count read from device memory (the SV assembly of K6 decides it); n_upper only sizes the grid
void launch_k5_dev(const double* lambda, const int32_t* k, double* out, const uint32_t* n_ptr, uint32_t n_upper, hipStream_t s);

// ---- K6: pair groups per region, component classification, SV assembly on the device ----------------------
// A connected component of the region graph that is a single region, or two regions of the same flush window joined
// by one group, is walked by one thread exactly as build_connection / process_sv would; every other component is
// handed to the host walk (bdx_walk.cpp) as a list of pair groups.
struct RegSum {        // per accepted region r, written by k6_pairs_kernel
    uint32_t np_all;   // sorted, merged (lo, flag, lib) parts of the pairs whose second mate is in r, at parts[first ..)
    uint32_t np_emit;  // those the host walk would need: the self group and every group that can pass the weight gate
    uint32_t in_off;   // with n_in == 1: offset and count of the parts of that one incoming group
    uint32_t np_in;
    uint32_t np_self;  // parts with lo == r: the last np_self of the np_all
    uint32_t n_in;     // groups (lo, r), lo < r, whose weight passes the gate (-r); lighter ones are never traversed
                       // or consumed by build_connection, so they do not connect anything
    uint32_t in_lo;    // that lo when n_in == 1
    uint32_t w_in;     // its weight
    uint32_t w_self;   // pairs with lo == r (weight of the self edge)
    uint32_t n_pairs;  // all pairs whose second mate is in r
    uint32_t n_weak;   // groups (lo, r), lo < r, below the gate
    uint32_t big;      // more reads than one wave sorts at once: its parts went straight to the host list
};
constexpr uint64_t kWeakPart = 1ull << 63;  // flag on a part key: belongs to a group below the weight gate

struct SvOut {         // == HostSv (bdx_walk.h)
    bdx_sv sv;
    uint32_t grp_mask; // consumed groups: bit 0 (A,A), bit 1 (A,B), bit 2 (B,B)
    uint32_t start;    // start vertex of the traversal that emitted it (output order key)
};

struct LibStage { int32_t lib, rc; double lambda; };
struct CnStage { int32_t key; float value; };

struct K6Arrays {
    uint32_t cap;                  // number of anomalous reads (capacity of the per-read / per-region arrays)
    const RegionRec* r_rec;        // device copies
    const uint32_t* r_pk;
    const int32_t* region_of;
    const int32_t* partner;
    const uint32_t* meta;
    const int32_t* isize;
    uint64_t* p_key;               // [cap] sorted parts of region r at [first_r, ...): lo << 12 | lib << 4 | flag
    uint32_t* p_pairs;
    uint32_t* p_sum;
    RegSum* rs;                    // [cap]
    uint32_t* out_deg;             // [cap] groups (r, hi) with hi > r
    uint32_t* out_hi;              // [cap] the hi of one of them
    // SV slots: 3 per region (start vertex, sequence number)
    SvOut* slot;                   // [3 cap]
    uint32_t* slot_info;           // [3 cap] valid | nacc << 1 | ncn << 8
    LibStage* lib_stage;           // [3 cap][acc_stride]
    CnStage* cn_stage;             // [3 cap][nkeys]
    uint32_t acc_stride;
    // dense outputs
    SvOut* sv_out;                 // pinned host
    int32_t* lib_index;            // pinned host
    int32_t* lib_pairs;            // pinned host
    int32_t* cn_key;               // pinned host
    float* cn_value;               // pinned host
    double* t_lambda;              // device
    int32_t* t_k;                  // device
    uint32_t sv_cap, term_cap, cn_cap;
    // groups of the components left to the host
    GroupRec* g_rec;               // pinned host
    uint32_t g_cap;
    U4* ws_u4;
    U4* total_u4;
    StageCounts* counts;
    // run constants
    const uint32_t* hist;          // [nlibs][11] adopted flag histogram
    const float* key_density;      // [nkeys]
    const float* lib_mean;         // [nlibs]
    uint32_t covered_ref_len;
    int nkeys, min_read_pair, chr_restricted, period, force_host;
};

constexpr int kK6MaxParts = 32;
void launch_k6_groups(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s);   // pair groups, host list, SV slots
void launch_k6_compact(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s);  // slots -> dense SV records and lists

}  // namespace bdx
