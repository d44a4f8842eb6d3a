// H1 -- host replay of the reference's order-dependent greedy graph walk over *aggregated* pair groups.
// Everything per read / per pair happened on the GPU; what is left is O(#regions + #groups).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/bdx.h"

namespace bdx {

struct HostRegion {  // same layout as the device's RegionRec (bdx_k3.h): the walk reads the table the kernels wrote
    int32_t tid, start, end;
    uint32_t n, rev, nonctx, nnormal;
    int32_t maxq;
    uint32_t first;   // compact index of the region's first read
};

struct GroupPart {  // one (flag, lib) slice of a region x region group, as emitted by K4
    uint32_t lo, hi;
    uint8_t flag, lib;
    uint32_t pairs;
    uint32_t sum_isize;
};

struct SvTerm {  // one Poisson term to be scored by K5
    double lambda;
    int32_t k;
};

struct HostSv {         // same layout as the device's SvOut (bdx_k3.h)
    bdx_sv sv;            // its Poisson terms are WalkResult::terms[sv.lib_begin .. + sv.lib_count)
    uint32_t grp_mask;    // pair groups this SV consumed: bit 0 (A,A), bit 1 (A,B), bit 2 (B,B) with A, B = sv.region[0..1]
    uint32_t start;       // start vertex of the traversal that emitted it
};

// output order of the reference's walk: flush windows ascending; inside a window first the traversals started from
// vertices of earlier windows (ascending id), then the window's own vertices ascending
inline uint64_t sv_order_key(uint64_t window, bool old_vertex, uint32_t start) {
    return (window << 33) | (old_vertex ? 0ull : 1ull << 32) | start;
}

struct WalkInput {
    bdx_opts opts;
    const bdx_lib* libs;
    int nlibs, nbams, nkeys;
    const uint32_t* hist;        // [nlibs][11]
    uint32_t covered_ref_len;
    const float* key_density;    // [nkeys] read density per counter key
    const HostRegion* regions;
    size_t nregions;
    const uint32_t* r_pk;        // [nregions][2*nkeys]: prefix counts at first read (nkeys), at last read (nkeys)
    const std::vector<GroupPart>* parts;
    int32_t last_maxq;           // _max_readlen at the final flush
    bool any_anomalous;
};

struct WalkResult {
    std::vector<HostSv> svs;
    std::vector<uint64_t> sv_key;  // sv_order_key of every entry of svs
    std::vector<SvTerm> terms;
    std::vector<int32_t> lib_index, lib_pairs;
    std::vector<int32_t> cn_key;
    std::vector<float> cn_value;
    uint32_t n_groups = 0;
    void clear() {
        svs.clear(); sv_key.clear(); terms.clear(); lib_index.clear(); lib_pairs.clear(); cn_key.clear(); cn_value.clear();
        n_groups = 0;
    }
};

struct LibAcc {  // pairs and |isize| sum of one library for the dominant flag of an SV candidate
    int lib, rc, span;
};
void emit_sv(const WalkInput& in, WalkResult& out, int A, int B, const int* flag_counts, int flag, const LibAcc* la, int nacc,
             int max_readlen, uint32_t grp_mask, uint64_t cur_key);

// H2 (bdx_walk_reads.cpp): the same replay one read at a time, for inputs in which a read name occurs more than twice.
// base.parts is unused; base.regions / r_pk already carry the phantom shift (region_of too).
struct ReadWalkInput {
    WalkInput base;
    uint32_t n_reads;          // anomalous (compact) reads in stream order
    const uint64_t* key;       // name key
    const int32_t* region_of;  // accepted region id, or -1 for a read of a rejected candidate
    const uint32_t* meta;      // flag | rev<<4 | lib<<8 | qlen<<16
    const int32_t* isize;      // |isize|
    uint32_t phantom;          // 1: region 0 is the read-less region a negative -s registers
    std::vector<uint32_t>* support_off;  // optional: [n_svs + 1] offsets into support
    std::vector<uint32_t>* support;      // optional: compact indices of the supporting reads, SvBuilder order
};
void read_level_walk(const ReadWalkInput& in, WalkResult& out);

// scratch vectors of the walk, kept by the context so that steady-state runs do not allocate
struct WalkScratch;
WalkScratch* walk_scratch_new();
void walk_scratch_free(WalkScratch* s);

void greedy_walk(const WalkInput& in, WalkScratch* scratch, WalkResult& out);

// combine the per-library log tails into the final score exactly as ComputeProbScore does
void finish_scores(const bdx_opts& opts, const double* log_tail, HostSv* svs, size_t nsvs, uint32_t* n_printed);

}  // namespace bdx
