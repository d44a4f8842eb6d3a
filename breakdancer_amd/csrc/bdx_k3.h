// Device arrays of the region-cut (K3), mate-join (K4) and scoring (K5) stages.
#pragma once
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
    uint32_t pad;
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
    uint64_t* t_key;     // [2*cap] global fallback table
    int32_t* t_idx;      // [2*cap]
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

// ---- K5 ---------------------------------------------------------------------------------------------
void launch_k5(const double* lambda, const int32_t* k, double* out, uint32_t n, hipStream_t s);

}  // namespace bdx
