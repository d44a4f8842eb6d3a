// H1 -- host replay of the reference's order-dependent greedy graph walk over *aggregated* pair groups.
// Everything per read / per pair happened on the GPU; what is left is O(#regions + #groups).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/bdx.h"

namespace bdx {

struct HostRegion {
    int32_t tid, start, end;
    uint32_t n, rev, nonctx, nnormal;
    int32_t maxq;
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

struct HostSv {
    bdx_sv sv;
    uint32_t term_begin, term_count;  // into WalkResult::terms
    uint32_t ngrp;                    // pair groups this SV consumed, (lo, hi) region ids
    uint32_t grp_lo[3], grp_hi[3];
};

struct WalkInput {
    bdx_opts opts;
    const bdx_lib* libs;
    int nlibs, nbams, nkeys;
    const uint32_t* hist;        // [nlibs][11]
    uint32_t covered_ref_len;
    const float* key_density;    // [nkeys] read density per counter key
    const std::vector<HostRegion>* regions;
    const uint32_t* r_pk;        // [nregions][2*nkeys]: prefix counts at first read (nkeys), at last read (nkeys)
    const std::vector<GroupPart>* parts;
    int32_t last_maxq;           // _max_readlen at the final flush
    bool any_anomalous;
};

struct WalkResult {
    std::vector<HostSv> svs;
    std::vector<SvTerm> terms;
    std::vector<int32_t> lib_index, lib_pairs;
    std::vector<int32_t> cn_key;
    std::vector<float> cn_value;
    uint32_t n_groups = 0;
    void clear() {
        svs.clear(); terms.clear(); lib_index.clear(); lib_pairs.clear(); cn_key.clear(); cn_value.clear();
        n_groups = 0;
    }
};

// scratch vectors of the walk, kept by the context so that steady-state runs do not allocate
struct WalkScratch;
WalkScratch* walk_scratch_new();
void walk_scratch_free(WalkScratch* s);

void greedy_walk(const WalkInput& in, WalkScratch* scratch, WalkResult& out);

// combine the per-library log tails into the final score exactly as ComputeProbScore does
void finish_scores(const WalkInput& in, const std::vector<double>& log_tail, WalkResult& out, uint32_t* n_printed);

}  // namespace bdx
