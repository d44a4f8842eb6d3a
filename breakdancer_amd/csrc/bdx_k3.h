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
    uint32_t n_printed;  // K6: SV candidates with score > opts.score_threshold
    uint32_t n_sv_dev;   // K6: SV candidates assembled on the device
    uint32_t n_terms_dev;  // K6: their (library, pairs) entries == Poisson terms
    uint32_t n_cn_dev;     // K6: their copy-number entries
    uint32_t n_groups_dev; // K6: region x region groups of the components handled on the device
    uint32_t stage_sv;     // K6: bump allocators of the staging lists
    uint32_t stage_lib;
    uint32_t stage_cn;
    uint32_t n_owners;     // K6: device-walked components (entries of K6Arrays::owners)
    uint32_t n_owners_big; // K6: entries of K6Arrays::owners_big
    uint32_t n_groups_big; // K6: groups of those components
    uint32_t n_old;        // K6: device candidates of traversals started from a vertex of an earlier flush window
    uint32_t n_ins;        // K6: candidates the compaction inserts by order key (the host walk's + n_old)
    uint32_t sort_done;    // K6: workgroups of the insertion list's rank sort that have finished (k6_insert_kernel's last workgroup waits for them)
    uint32_t irregular;    // K4: some read name was seen more than twice among the anomalous reads (ReadRegionData.cpp:108-113
                           // keeps appending): the pair model does not hold, the run is replayed read by read on the host
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
    int32_t* c_rid;
    int32_t* region_of;  // per compact read: accepted region id or -1
    // per accepted region (array-of-structs so one copy brings the table to the host)
    RegionRec* r_rec;
    uint32_t* r_pk;  // [cap][2*nkeys]: proper-read prefix counts at the region's first read (nkeys), then last read (nkeys)
    RegionRec* r_rec_dev;  // device-resident copies for K6 (r_rec / r_pk live in pinned host memory); may be null
    uint32_t* r_pk_dev;
    int host_copy_later;   // 1: write only the device copies here; the join kernel forwards them to r_rec / r_pk
    uint32_t* out_deg;     // K6 per-region scratch reset by k3_region_of_kernel: [6][cap] = out_deg, label (= index),
                           // bad_v, bad, mcount, pcount; may be null
    // scan workspace and totals
    U4* ws_u4;
    U4* head_total;
    uint32_t* ws_u32;
    uint32_t* acc_total;
    unsigned long long* lb_state;  // look-back words of the two scans: [5][scan_grid(cap, 1)], zero at allocation; null: three-launch scans
    uint32_t lb_stamp;             // run stamp of those words (never 0, changes every run)
    StageCounts* counts;
    StageCounts* counts_host;  // pinned host mirror: n_cand / n_regions / last_maxq are stored there as well (may be null)
    uint32_t* flag_host;       // pinned word set to flag_value by k3_region_of_kernel: the region table is complete
    uint32_t flag_value;
};

// the read that closes the last candidate when the stream continues in another context (next chromosome)
struct K3Tail {
    int has_next;
    int32_t qlen;
    uint32_t nn;
    // a context that holds SEVERAL chromosomes of a sharded run, not necessarily neighbours in the genome: the read that closes the
    // last candidate of chromosome t is the first anomalous read of the next chromosome that has any -- wherever it lives --, from
    // this table of four words per chromosome: {has_next, its read length, the normal-pair count its candidate ends with (the next
    // read's, or the genome's total), 0}.  Null: the compact list's own next read closes (single-context runs)
    const uint32_t* tid_tail;
};

void launch_k3(const K3Arrays& a, const Compact& cp, const Pass1* p1, uint32_t n_anom_host, int min_len, int seq_coverage_lim,
               int nkeys, uint32_t nn_base, K3Tail tail, bool region_of_launch, hipStream_t s);

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
    int32_t* partner;    // [cap] compact index of the mate; -1 name seen once; -2 name seen twice but a mate sits in a rejected
                         // candidate region (ReadRegionData.cpp:177-199 forgets the name: no pair)
    int32_t* pair_lo;    // [cap] direct join of a single-context run: at the second-observed mate of a pair, the region of the
                         // first-observed one (else -1, preset); saves K6 a dependent load per read.  May be null
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
constexpr uint32_t kMaxRegions = (1u << 26) - 2;      // accepted regions of one result (26-bit region ids in the packed group key)
constexpr uint32_t kMaxAnomalous = 0x7FFFFF00u;       // anomalous reads of one context / one sharded run (32-bit signed read indices)

// join input: one entry per anomalous read (region < 0: in a rejected candidate region -- it still enters the table so that
// a name seen three times is noticed wherever its reads lie, but it never forms a pair)
struct Entries {
    const uint64_t* key;
    const uint64_t* check;   // second name hash: two entries are mates only if it agrees as well (nullptr: the key alone decides)
    const int32_t* region;   // region id (global ids when the entries come from several shards)
    const uint32_t* order;   // position in the merged stream order; nullptr = the entry index itself
    const uint32_t* meta;    // flag | rev<<4 | lib<<8 | qlen<<16
    const int32_t* isize;    // |isize|
    int32_t region_base;     // added to both region ids of a pair group (a chromosome's local ids -> genome-wide ids)
    // single-context runs with the direct join: the join kernel derives the region itself (region = c_rid[cand[j]]),
    // stores it into region_out and resets K6's per-region scratch -- k3_region_of_kernel's work without its launch
    const int32_t* cand;
    const int32_t* c_rid;
    int32_t* region_out;
    // ... and copies the region table from HBM into pinned host memory (K3 then writes the HBM copy only: the PCIe writes
    // overlap this latency-bound kernel instead of stretching the scan's last pass)
    const RegionRec* r_rec_dev;
    const uint32_t* r_pk_dev;
    RegionRec* r_rec_host;
    uint32_t* r_pk_host;
    const StageCounts* counts;
    int nkeys2;              // 2 x nkeys words of proper-read samples per region
    uint32_t* k6_scratch;    // [6][scratch_cap]
    uint32_t scratch_cap;
    uint32_t* flag_host;     // pinned word: the region table (written by the kernel before) is complete
    uint32_t flag_value;
    uint32_t fwd_blocks;     // workgroups in front of the joining ones that forward the region table (0: kJoinForwardBlocks; ~0: every wave forwards, A/B)
    // sharded runs: entries n_local .. n - 1 are FOREIGN -- inter-chromosomal reads of chromosomes another rank owns whose mates lie on
    // a (later) chromosome of this rank (k7_exchange.hip) -- with their own arrays; they come first in stream order, are never the
    // second-observed mate, and have no partner[] / pair_lo[] entry.  n_local null: every entry is the context's own
    const uint32_t* n_local;
    const uint64_t* fkey;
    const uint64_t* fcheck;
    const int32_t* fregion;
    int want_pair_lo;        // fill K4Arrays::pair_lo from `region` (single-context runs derive it from c_rid)
};

void launch_k4(const K4Arrays& k4, const Entries& e, const uint32_t* n_ptr, uint32_t n_upper, StageCounts* counts, hipStream_t s);

// ---- K7: exchange of inter-chromosomal mate records between the GPUs of a chromosome-sharded run ------------------
constexpr int kMaxRanks = 64;
struct ExchangeEntry {  // one CTX read on its way to the rank that joins its name
    uint64_t key;
    uint32_t order;     // position among the anomalous reads of the whole genome (stream order: decides the second-observed mate)
    int32_t region;     // global region id, -1: the read sits in a rejected candidate region
    uint32_t meta;
    int32_t isize;
    uint64_t check;     // second name hash (0 when the streams carry none)
};
static_assert(sizeof(ExchangeEntry) == 32, "exchange entries travel as four 64-bit words");
// the rank that joins a name key: a mixed hash, so that the two mates of a pair (same key) meet on one rank
__host__ __device__ __forceinline__ uint32_t exchange_owner(uint64_t k, uint32_t world) {
    k = (k ^ (k >> 33)) * 0xff51afd7ed558ccdull;
    k ^= k >> 29;
    return (uint32_t)(k % world);
}
// rank 0: the packages of the gather (byte offsets into the gather buffer; a package = its rank's region records and their prefix samples)
struct GatherPackage { uint64_t regions_off, pk_off, groups_off; uint32_t nr, ng; };
struct GatherDesc { GatherPackage p[kMaxRanks]; int world; };
void launch_k8_place_regions(const char* all, const GatherDesc& D, uint32_t max_nr, const uint64_t* rbase, int ntids, int nkeys2, RegionRec* r_rec,
                             uint32_t* r_pk, uint32_t* err, hipStream_t s);

void launch_k4_join_only(const K4Arrays& k4, const Entries& e, const uint32_t* n_ptr, uint32_t n_upper, StageCounts* counts, hipStream_t s);

// ---- K5 ---------------------------------------------------------------------------------------------
void launch_k5(const double* lambda, const int32_t* k, double* out, uint32_t n, hipStream_t s);

// ---- K6: pair groups per region, small components walked on the device -----------------------------------
// Connectivity of the region graph comes from the groups whose weight passes the gate (-r): lighter groups are
// skipped by every try_edge of build_connection and never reach process_sv.  A connected component of at most
// kK6MaxMembers regions inside one flush window is walked by ONE thread exactly as build_connection / process_sv
// would (BreakDancer.cpp:266-497); every other component is handed to the host walk (bdx_walk.cpp) as a list of
// pair groups.
constexpr int kK6MaxMembers = 4;   // regions per device-walked component
constexpr int kK6MaxIn = 3;        // incoming gate-passing groups per region (a member of such a component has <= 3)
constexpr int kK6BigMembers = 64;  // regions per component walked by the general device path (one wave, member lists in LDS)
constexpr int kK6LibStride = 16;   // staged (library, pairs) entries per candidate; components that could need more go to the host
constexpr int kK6LabelRoundsBig = 8; // ... with the general walk (components of up to kK6BigMembers regions) enabled
constexpr uint32_t kK6RankSlices = 16;       // the list is compared against in up to this many slices (k6_ranksort_kernel)
constexpr uint32_t kK6RankSortMax = 1u << 17;  // entries of the insertion list the all-pairs rank sort takes (whole GPU: microseconds)
constexpr int kK6LabelRounds = 2;  // min-label propagation rounds (the first inside k6_pairs_kernel, the others with pointer jumping):
                                   // two settle chains of four regions in practice; a component that has not converged fails
                                   // the closure check and goes to the host

struct RegSum {        // per accepted region r, written by k6_pairs_kernel
    uint32_t np_all;   // sorted, merged (lo, flag, lib) parts of the pairs whose second mate is in r, at parts[first ..)
    uint32_t np_emit;  // those the host walk would need: the self group and every group that passes the gate
    uint32_t np_self;  // parts with lo == r: the last np_self of the np_all
    uint32_t w_self;   // pairs with lo == r (weight of the self edge)
    uint32_t n_in;     // groups (lo, r), lo < r, passing the gate
    uint32_t n_pairs;  // all pairs whose second mate is in r
    uint32_t n_weak;   // groups (lo, r), lo < r, below the gate
    uint32_t big;      // more reads than one wave sorts at once: its parts went straight to the host list
    uint32_t e_lo[kK6MaxIn];   // the first kK6MaxIn incoming gate-passing groups: region, weight, parts [off, off + cnt)
    uint32_t e_w[kK6MaxIn];
    uint32_t e_off[kK6MaxIn];
    uint32_t e_cnt[kK6MaxIn];
};
constexpr uint64_t kWeakPart = 1ull << 63;  // flag on a part key: belongs to a group below the weight gate

struct PartRec {       // one (lo, flag, lib) slice of the pairs whose second-observed mate is in a region
    uint64_t key;      // lo << 12 | lib << 4 | flag (| kWeakPart)
    uint32_t pairs;
    uint32_t sum;      // sum of |isize| of the second-observed mates
};

struct MemberInfo {    // what the walk needs to know about one region of a component, stored with the component's label
    uint32_t r;
    RegionRec rec;
    uint32_t np_all, np_self, w_self, n_in;
    uint32_t e_lo[kK6MaxIn], e_w[kK6MaxIn], e_off[kK6MaxIn], e_cnt[kK6MaxIn];
    uint32_t stored;   // ReadRegionData.cpp:118-121
    uint32_t pad;
};
constexpr int kMemberWords = sizeof(MemberInfo) / 4;

struct SvOut {         // == HostSv (bdx_walk.h)
    bdx_sv sv;
    uint32_t grp_mask; // consumed groups: bit 0 (A,A), bit 1 (A,B), bit 2 (B,B)
    uint32_t start;    // start vertex of the traversal that emitted it (output order key)
};

// A row of the final table as it crosses PCIe (single-context runs): 48 bytes instead of SvOut's 96.  What is left out the host has
// already -- chromosomes and read counts by strand are the regions' (the region table is in host memory before the walk starts), the
// list offsets are the running sums of the counts in table order -- and materialize() (bdx_api.hip) puts it back.  The table kernel's
// time is its bytes over the link (profiles/r06_kprof_genome.txt: every wave is through by 150 us of 175, 6.6 MB at 44 GB/s).
struct SvWire {
    int32_t pos[2], region[2];
    int32_t size, score, num_reads;
    float allele_frequency;
    double logp;
    uint32_t start;
    uint32_t bits;   // lib_count (8) | cn_count << 8 (8) | flag << 16 (4) | grp_mask << 20 (3) | printed << 23
};
static_assert(sizeof(SvWire) == 48, "twelve 32-bit words");
constexpr int kWireWords = sizeof(SvWire) / 4;

struct LibStage { int32_t lib, rc; double lambda; };
struct CnStage { int32_t key; float value; };

struct K6Arrays {
    uint32_t cap;                  // number of anomalous reads (capacity of the per-read / per-region arrays)
    const RegionRec* r_rec;        // device copies
    const uint32_t* r_pk;
    const int32_t* region_of;
    const int32_t* partner;
    const int32_t* pair_lo;        // see K4Arrays::pair_lo; null: derive it from partner and region_of
    const uint32_t* meta;
    const int32_t* isize;
    // sharded runs (rank 0): the pair groups arrive as aggregates gathered from the ranks, bucketed by their later region --
    // region r's are in_groups[in_goff[r] .. in_goff[r+1]) -- instead of being counted from the reads (partner / meta / isize
    // are then unused, and a region's `first` is its place in a slot space laid out by the host)
    const GroupRec* in_groups;
    const uint32_t* in_goff;
    const uint32_t* first_of;      // [regions] a region's place in that slot space, instead of RegionRec::first (null: the record's own)
    PartRec* parts;                // [cap] sorted parts of region r at [first_r, ...)
    RegSum* rs;                    // [cap]
    // component analysis; the six arrays below are reset by k3_region_of_kernel (label[r] = r, the others 0)
    uint32_t* out_deg;             // [cap] gate-passing groups (r, hi) with hi > r
    uint32_t* label;               // [cap] smallest region id seen so far in r's component
    uint32_t* bad_v;               // [cap] r itself cannot be walked on the device (too large, too many incoming groups)
    uint32_t* bad;                 // [cap] by label: the component goes to the host
    uint32_t* mcount;              // [cap] by label: members registered
    uint32_t* pcount;              // [cap] by label: parts its walk can touch (bounds the libraries of one SV candidate)
    MemberInfo* members;           // [cap][kK6MaxMembers] by label
    uint32_t* owners;              // [cap] smallest regions of the device-walked components (k6_emit_kernel), any order
    // SV candidates of the device-walked components.  Staging slots need no allocation: the candidate that consumes the
    // group (A, B) sits at first_B + (index of that group among B's incoming ones), the one of B's self group right
    // after them -- every group owns at least one read of its later region, so the slots exist and are distinct.
    uint32_t* own_nsv;             // [cap] per start vertex (smallest region of a component): candidates emitted ...
    uint32_t* own_nacc;            // [cap] ... the sum of their (library, pairs) entries ...
    uint32_t* own_ncn;             // [cap] ... and of their copy-number entries
    uint32_t* own_first;           // [cap] staging slot of the first of them ...
    uint32_t* slot_next;           // [cap] ... and, by staging slot, the slot of the next one in emission order
    uint32_t* emit_part;           // [64][4] k6_emit_kernel's totals by slot (owners, pairs, groups on the device, groups of large components), zeroed by k6_pairs_kernel
    uint32_t* owners_big;          // [cap] smallest regions of the device-walked components of more than kK6MaxMembers regions
    uint32_t* member_ids;          // [cap][kK6BigMembers] by label: the regions of a component (k6_classify_kernel, any order)
    SvOut* sv_stage;               // [cap]
    LibStage* lib_stage;           // [cap][lib_stride]
    CnStage* cn_stage;             // [cap][nkeys]
    uint32_t lib_stride;           // min(nlibs, kK6LibStride)
    // final table: built densely in HBM by the compaction, scored, then written to pinned host memory in one coalesced
    // pass by k6_score_kernel
    uint32_t* sv_src;              // device [sv_cap]: staging slot of the candidate at each final position, or 0x80000000 | j for
                                   // the host walk's candidate j
    uint2* sv_begin;               // device [sv_cap]: its first entries in the two flat lists
    uint32_t* sv_vx;               // device [sv_cap]: the vertex it is placed at (with sv_key: the order keys for the merge of a sharded run's tables); else null
    int32_t* d_lib_index;          // device [term_cap]
    int32_t* d_cn_key;             // device [cn_cap]
    float* d_cn_value;             // device [cn_cap]
    double* t_lambda;              // device [term_cap]
    int32_t* t_k;                  // device [term_cap] (= library pair counts)
    SvOut* sv_out;                 // pinned host
    int32_t* lib_index;            // pinned host
    int32_t* lib_pairs;            // pinned host
    int32_t* cn_key;               // pinned host
    float* cn_value;               // pinned host
    uint32_t sv_cap, term_cap, cn_cap;
    double* ltail;                 // device [term_cap]: log tails of the terms ...
    uint32_t* printed_host;        // pinned host [k6_score_grid()]: printed candidates per workgroup of k6_score_kernel; may be null
    double* ltail_host;            // ... and their copy in pinned host memory (both written by k6_score_kernel)
    // Candidates that are placed by their order key instead of by their start vertex: the host walk's (pinned host
    // memory) and the device's own whose traversal started from a vertex of an earlier flush window.  k6_insert_kernel
    // merges the two lists by key; entry j of the merged list precedes the candidates of start vertex ins_T[j] and after.
    // Order key: (T << 34) | (started at a vertex of its own window ? 1 << 33 : 0) | (start vertex << 7) | sequence number,
    // T = the start vertex, or the first vertex of the flush window for a traversal started from an earlier window's vertex.
    const SvOut* hs_rec;           // [nh] lib_begin / cn_begin index the host lists below
    const uint64_t* hs_key;        // [nh] ascending order keys
    const uint32_t* hs_cnt;        // [nh] lib_count | cn_count << 16
    uint64_t* old_key;             // device [pow2 >= sv_cap] order keys of the device's list (k6_walk_kernel, any order) ...
    uint32_t* old_slot;            // ... and their staging slots
    uint64_t* sorted_key;          // device [kK6RankSortMax]: the device's list sorted by key when it has more entries than one workgroup sorts
    uint32_t* sorted_slot;         // quickly (and at most kK6RankSortMax): placed by k6_insert_kernel from k6_ranksort_kernel's ranks; null: k6_insert_kernel sorts by itself
    uint32_t* rank_part;           // device [kK6RankSlices][n_old]: every entry's number of smaller keys within a slice of the list (k6_ranksort_kernel)
    uint64_t* hs_key_dev;          // device [sv_cap] copy of hs_key
    uint32_t* ins_T;               // device [sv_cap] merged list: threshold vertex ...
    uint32_t* ins_src;             // ... staging slot, or 0x80000000 | j for the host walk's candidate j
    uint32_t* ins_pre_l;           // [sv_cap + 1] entries of the list's candidates before j in the (library, pairs) lists ...
    uint32_t* ins_pre_c;           // [sv_cap + 1] ... and in the copy-number lists
    const int32_t* hs_lib_index;
    const int32_t* hs_lib_pairs;
    const double* hs_lambda;
    const int32_t* hs_cn_key;
    const float* hs_cn_value;
    uint32_t nh;
    // groups of the components left to the host
    GroupRec* g_rec;               // pinned host
    uint32_t g_cap;
    unsigned long long* lb_state;  // look-back words of k6_place_kernel's scan: [scan_grid(cap, 1)][4], zero at allocation
    uint32_t lb_stamp;             // run stamp of those words (never 0, changes every run)
    StageCounts* counts;
    StageCounts* counts_host;      // pinned: all counters after k6_emit_kernel (k6_mirror_kernel, or the first wave of k6_walk_kernel)
    StageCounts* counts_host2;     // pinned: n_sv_dev / n_terms_dev / n_cn_dev / overflow after the compaction
    uint32_t* flag_regions;        // pinned word set to flag_value by k6_pairs_kernel's first thread: everything enqueued before that kernel
                                   // (the join, which forwards the region table to the host) has completed; may be null
    uint32_t* flag_groups;         // pinned word set to flag_value behind counts_host: the host's share of the groups is complete; may be null
    uint32_t* flag_done;           // pinned word set by k6_done_kernel when the final table is complete; null when a stream command does it
    uint32_t flag_value;
    // run constants
    const uint32_t* hist;          // [nlibs][11] adopted flag histogram
    const float* key_density;      // [nkeys] read density per counter key (finalize2_kernel)
    const float* lib_mean;         // [nlibs]
    const Pass1* p1;               // covered_ref_len is read from the device's pass-1 record
    int nlibs, nkeys, min_read_pair, chr_restricted, period, force_host;
    int mirror_in_walk;            // the first wave of k6_walk_kernel mirrors the counters and sets flag_groups (no k6_mirror_kernel launch)
    int big_walk;                  // components of up to kK6BigMembers regions are walked on the device (k6_walk_big_kernel); 0: up to kK6MaxMembers
    uint32_t fin_regions;          // regions k6_place_kernel's launch is sized for (the host's count once it has read it; 0: cap)
    int walk_lanes;                // regions per wave of k6_walk_kernel (<= 64)
    int asm_plain;                 // test switch (bdx_set_debug "asm_plain"): the candidate assembly merges its parts by the three-way merge whatever the number of libraries
    int ins_plain;                 // test switch (bdx_set_debug "ins_plain"): 1 = the insertion list is ranked as before round 6 (k6_ranksort_kernel / LDS bitonic),
                                   // 2 = the bucket path declares its list crowded (the bitonic sort takes it)
    int label_rounds;              // min-label propagation rounds incl. the one inside k6_pairs_kernel (default kK6LabelRounds)
    // sharded runs: region ids are genome-wide, the table holds this rank's regions at their genome-wide places and n == 0 everywhere
    // else.  A gate-passing group whose earlier region is another rank's makes both of its regions `tainted` (bytes, all-reduced over
    // the ranks between k6_pairs_kernel and k6_classify_kernel): their components go to the host list, i.e. to rank 0's walk
    uint8_t* taint;                // [cap]; null: single-context run
    int wire_rows;                 // sv_out takes SvWire rows (the table goes to pinned host memory: single-context runs)
    unsigned long long* sv_key;    // [sv_cap] order key of every row of the final table (T << 34 | own << 33 | start << 7): what rank 0 merges
                                   // the ranks' tables by; null: not wanted
};

void launch_k6_groups(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s);   // pair groups, components, the host's list
void launch_k6_pairs(const K6Arrays& a, uint32_t n_upper, hipStream_t s);        // ... in two parts: the pair groups (sharded runs exchange
void launch_k6_components(const K6Arrays& a, uint32_t n_upper, hipStream_t s);   // the taint bytes in between), then the components
// start values of the per-region scratch (out_deg, label = index, bad_v, bad, mcount, pcount) when no join kernel has set them
void launch_k6_scratch_init(uint32_t* out_deg, uint32_t cap, hipStream_t s);
void launch_k6_walk(const K6Arrays& a, uint32_t n_anom_host, hipStream_t s);     // device-walked components -> SV staging
// staging + host candidates -> final table in pinned host memory, scored (k6_insert_kernel, k6_place_kernel, k6_score_kernel)
void launch_k6_table(const K6Arrays& a, uint32_t n_anom_host, double ln10, int score_threshold, int with_scores, hipStream_t s);
// ComputeProbScore's combination (BreakDancer.cpp:56-69) + PhredQ (:459-465) for every candidate of the final table
uint32_t k6_score_grid(const K6Arrays& a);  // workgroups of k6_score_kernel == entries of K6Arrays::printed_host

}  // namespace bdx
