// Shared device-side declarations of libbdx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bdx {

constexpr int kTile = 256;      // reads per tile = one wave64 x 4 consecutive reads per lane (no workgroup barrier per tile)
constexpr int kBlock = 256;     // threads per workgroup (4 wave64)
constexpr int kWaves = kBlock / 64;
constexpr int kNumFlags = 11;
constexpr int kK2TilesPerWave = 4; // K2 compacts four of K1's tiles per wave and needs the prefixes at those boundaries only
constexpr int kCntCopies = 64;   // replication factor of the global pass-1 counters (power of two)
constexpr int kMaxChunks = 64;   // chunks a tile-total column is scanned in (one workgroup of finalize_kernel each)
constexpr int kStashCap = 16;    // anomalous reads per tile that K1 leaves ready-made for K2 (a tile with more is compacted from the columns)
constexpr int kStashKeys = 2;    // ... when there are at most this many normal-read counter keys (K2's fast path holds their totals in 16 lanes)

// What K1 leaves for K2 about an anomalous read (slot `rank in tile` of the tile's kStashCap slots): everything K2 would
// otherwise gather from seven columns -- K1 has it in registers anyway -- plus the read's in-tile prefix counts, so that K2
// needs neither the class bytes nor the columns of a tile whose anomalous reads fit (name key and read length are fetched by
// K2: K1 does not load those columns).  Only tiles whose reads share one library and source file are served (their normal
// reads count for one key); the slots K2 would read of any other tile with anomalous reads say so: where == 0xFFFFFFFF.
struct StashRec {
    int32_t tid, pos, isize;   // isize = |isize|
    uint32_t meta;             // flag | rev << 4 | lib << 8 (read length: K2)
    uint32_t where;            // offset in tile (8 bits) | normal-leftmost reads before it in the tile << 8 | the tile's counter key << 20
    uint32_t proper;           // proper reads (of that key) up to and including it in the tile
    uint32_t pad[2];
};
static_assert(sizeof(StashRec) == 32, "two 16-byte stores");

enum : int { F_NA = 0, F_FF = 1, F_LARGE = 2, F_SMALL = 3, F_RF = 4, F_RR = 5, F_NORMAL_FR = 6, F_NORMAL_RF = 7,
             F_CTX = 8, F_MATE_UNMAPPED = 9, F_UNMAPPED = 10 };

struct DevLib {
    float upper, lower;
    int32_t min_mapq;  // already resolved against opts.min_map_qual
    int32_t key;       // normal-read counter key: library index (-a) or the library's BAM index
};

struct ReadsSoA {
    const int32_t *tid, *pos, *mtid, *mpos, *isize;
    const uint16_t *flag, *qlen;
    const uint8_t *mapq, *lib, *bam;
    const uint64_t* key;
    const uint64_t* check;   // second, independent hash of the read name (nullptr: names are compared by key alone)
};

// reference-length monoid of one (tile, source file): BamSummary.cpp:70-74 adds pos - last_pos for consecutive
// same-tid records of a file; (first, last, interior sum) composes associatively across tiles
struct MonoRec {
    int32_t ft, fp, lt, lp;
    long long sum;
};

// tile-total columns: 0 anomalous, 1 normal leftmost, 2.. per normal-read key
constexpr int kColAnom = 0, kColNormal = 1, kColKey0 = 2;

struct K1Params {
    ReadsSoA r;
    uint64_t n;
    uint32_t ntiles, tstride;  // tstride = padded row length of the per-tile tables
    uint32_t tile0;            // first tile of this launch (streamed input: tiles are classified as their reads arrive)
    int nlibs, nbams, nkeys;
    int max_sd, opt_t, opt_l;
    const DevLib* libs;
    uint8_t* cls;              // [n]
    uint32_t* tile_tot;        // [2+nkeys][tstride]
    MonoRec* tile_mono;        // [nbams][tstride], pre-set to 0xFF (ft == -1: file absent from the tile)
    uint32_t* blk_cnt;         // [kCntCopies][ncnt] zero-filled, ncnt = nlibs*11 + nlibs + nbams
    StashRec* stash;           // [tiles][kStashCap]; null: more than kStashKeys counter keys (K2 gathers everything itself)
};

// results of pass 1, produced on the device and mirrored to the host
struct Pass1 {
    uint32_t covered_ref_len;
    int32_t window;
    uint32_t n_anom;      // total anomalous reads
    uint32_t n_normal;    // total normal-leftmost reads
    uint32_t key_tot[60]; // proper read totals per key
    unsigned long long ref_len[256];  // per source file: sum of pos - last_pos (BamSummary.cpp:70-74)
};

struct FinalizeParams {
    uint32_t ntiles, tstride, nblk;
    uint32_t nfold;             // workgroups folding the monoid table (first level)
    MonoRec* fold_part;         // [nbams][nfold]
    int nlibs, nbams, nkeys, ncols, ncnt;
    int w0;
    const uint32_t* tile_tot;
    uint32_t* tile_pre;         // [ncols][tstride]: exclusive scan of tile_tot per column at every kK2TilesPerWave-th tile (entry tile /
                                // kK2TilesPerWave), restarting at every chunk of chunk_super such entries ...
    uint32_t* chunk_tot;        // [ncols][kMaxChunks]: ... whose totals are here ...
    uint32_t* chunk_base;       // [ncols][kMaxChunks]: ... and, from the second level, the totals of the chunks before each
    uint32_t chunk_super, nchunk;
    const MonoRec* tile_mono;
    const uint32_t* blk_cnt;
    uint32_t* cnt;              // [ncnt] reduced counters
    Pass1* p1;
    uint32_t* cnt_host;         // pinned host mirrors written by the last kernel (no copy command on the stream); may be null
    Pass1* p1_host;
    const DevLib* libs;         // for the read densities per counter key (BreakDancerMax.cpp:94-107)
    int cn_lib;
    float* key_density;         // [nkeys]; may be null
    uint32_t* flag_host;        // pinned word set to flag_value once the mirrors are written (the host polls it); may be null
    uint32_t flag_value;
    uint32_t* done;             // zeroed ticket counter: the workgroup of finalize_kernel that finishes last runs the second level
                                // itself (null: the second level is its own launch, or rides on K2)
    uint32_t na_cap;            // 0, or the capacity the later stages were already enqueued with: if there are more anomalous
                                // reads, the device copy of n_anom is zeroed (they then do nothing) and the host, which still
                                // gets the true count, runs them again
};

// one launch that sets several scratch buffers to their start values (replaces a chain of small fill commands)
struct InitList {
    uint32_t* ptr[6];
    uint32_t words[6];
    uint32_t value[6];
    int n;
};
void launch_init(const InitList& l, hipStream_t s);

// compact anomalous-read records (one per read entering the region accumulator)
struct Compact {
    int32_t* tid;
    int32_t* pos;
    int32_t* isize;     // |isize|
    uint32_t* meta;     // flag (4) | rev<<4 | lib<<8 | qlen<<16
    uint64_t* key;
    uint64_t* check;    // nullptr: the stream has no second name hash
    uint32_t* idx;      // index of the read in the resident stream
    uint32_t* nn;       // normal-leftmost reads seen before this read (stream order)
    uint32_t* pk;       // [nkeys][cap]: proper reads of key k seen up to and including this read
    uint32_t cap;
};

struct K2Params {
    ReadsSoA r;
    uint64_t n;
    uint32_t ntiles, tstride;
    int nkeys, nlibs;
    const DevLib* libs;
    const uint8_t* cls;
    const uint32_t* tile_pre;  // chunk-local prefixes and the chunks' totals (FinalizeParams)
    const uint32_t* chunk_tot;  // (K2 adds up the chunks before its own when it runs beside the second level ...
    const uint32_t* chunk_base; // ... and reads the sum otherwise)
    uint32_t chunk_super;
    const uint32_t* tile_tot;  // K1's per-tile totals and its ready-made records (stash null: not available)
    const StashRec* stash;
    Compact c;
    uint32_t nn_base;      // normal read pairs / proper reads of earlier shards (0 for a single context)
    uint32_t pk_base[60];
    // initialisation of later stages' scratch folded into this launch (its grid is large and mostly idle)
    uint32_t* fill_ptr[4];
    uint32_t fill_words[4];
    uint32_t fill_value[4];
    // name keys and read lengths left in the caller's pinned host memory (bdx_push): read i's key is seg_ptr[s][i], its
    // length seg_qlen[s][i], for the segment s with seg_begin[s] <= i < seg_begin[s+1]; only the ~1 % anomalous reads ever
    // fetch theirs (over PCIe).  nseg = 0: r.key / r.qlen
    int nseg;
    const uint64_t* seg_begin;          // [nseg + 1]
    const uint64_t* const* seg_ptr;     // [nseg] device-visible, biased by -seg_begin[s]
    const uint16_t* const* seg_qlen;    // [nseg] likewise
    const uint64_t* const* seg_check;   // [nseg] likewise (only read when c.check is set)
};

__device__ __forceinline__ uint32_t meta_pack(int flag, int rev, int lib, int qlen) {
    return (uint32_t)flag | ((uint32_t)rev << 4) | ((uint32_t)lib << 8) | ((uint32_t)qlen << 16);
}
__host__ __device__ __forceinline__ int meta_flag(uint32_t m) { return m & 15; }
__host__ __device__ __forceinline__ int meta_rev(uint32_t m) { return (m >> 4) & 1; }
__host__ __device__ __forceinline__ int meta_lib(uint32_t m) { return (m >> 8) & 255; }
__host__ __device__ __forceinline__ int meta_qlen(uint32_t m) { return m >> 16; }

// ---- wave helpers (wave64) ----------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
// (the builtin keeps the condition a lane mask; __ballot() goes through a 0/1 integer per lane and a second compare)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }

// inclusive wave scan of a u32 (6 shuffle steps)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (l >= o) v += t;
    }
    return v;
}

// launchers (host side, defined in the .hip files)
void launch_k1(const K1Params& p, int grid, size_t lds, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
size_t k1_lds_bytes(int nlibs, int nbams, int nkeys);
void launch_finalize(const FinalizeParams& p, hipStream_t s, bool second_level = true);
void launch_finalize2_only(const FinalizeParams& p, hipStream_t s);
void launch_k2(const K2Params& p, size_t lds, hipStream_t s, const FinalizeParams* side = nullptr);  // side: finalize2_body as an extra workgroup
size_t k2_lds_bytes(int nkeys);

}  // namespace bdx
