// Device-side BAM decode: BGZF blocks inflated by the GPU (kz_inflate.hip), BAM records found and turned into the SoA
// columns of the record stream (kb_records.hip).  Stands where the host producer has zlib / fast_inflate + column_reader.cpp
// (and the reference samtools' bgzf.c + bam_read1 behind io/BamReader.hpp:62-70, io/Alignment.cpp:12-64): the compressed
// file crosses PCIe once (about 0.7 x the bytes of the inflated records) and everything per byte and per record happens in HBM.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bdx {

// one BGZF member: deflate payload in the compressed piece, destination in the inflated ring
struct BgzfBlock {
    uint64_t in_off;    // payload offset in the compressed buffer
    uint64_t out_off;   // destination offset in the inflated buffer
    uint32_t in_len;    // payload bytes (BSIZE - XLEN - 19)
    uint32_t out_len;   // ISIZE
};

// per-block status of the inflate kernel
enum : uint32_t {
    KZ_OK = 0,
    KZ_BAD_BLOCK_TYPE = 1, KZ_BAD_STORED = 2, KZ_BAD_LENGTHS = 3, KZ_BAD_CODE = 4, KZ_BAD_DISTANCE = 5, KZ_OUTPUT_OVERRUN = 6,
    KZ_INPUT_OVERRUN = 7, KZ_SIZE_MISMATCH = 8
};

// in: compressed bytes (at least 16 readable bytes behind the last payload), out: inflated bytes; status[nblk]
void launch_kz_inflate(const uint8_t* in, const BgzfBlock* blocks, uint32_t nblk, uint8_t* out, uint32_t* status, hipStream_t s,
                       unsigned long long* prof = nullptr);
// ---- records ----
constexpr uint32_t kRecSlots = 2048;        // record starts a BGZF block can hold at most ((65536 / 36) + 1 < 2048)
constexpr uint32_t kNoGuess = 0xFFFFFFFFu;
constexpr uint32_t kMaxDeviceRecord = 4u << 20;  // larger records (ultra-long reads) are left to the host reader

struct ChainBlock {        // what kb_chain learns about one BGZF block (offsets relative to the block's first byte)
    uint32_t guess;        // first record start the block was walked from (kNoGuess: none recognised)
    uint32_t count;        // records that start in the block
    uint32_t end;          // where the record behind the last one starts (>= the block's length unless the walk broke off)
    uint32_t bad;          // 1: the walk met a size word that cannot be one (a wrong guess, or a corrupt file)
};

struct RgTable {           // read-group ids of the configuration, for the RG -> library lookup (io/BamConfig.hpp:62-72)
    const uint64_t* hash;  // hash_name(id)
    const uint32_t* off;   // [n + 1] into chars
    const char* chars;
    const uint8_t* lib;
    uint32_t n;
    uint8_t fallback;      // library of unknown / missing read groups (io/AlignmentSource.hpp:57-62)
    uint8_t missing;       // ... of records WITHOUT a read-group tag (= fallback unless the caller tells them apart: bam2cfg does, perl/bam2cfg.pl:93-104)
};

struct RecordFilterDev {   // -o <region> (io/RegionLimitedBamReader.hpp:63-71, bam_index.c:571-576): only_tid < 0: everything
    int32_t only_tid, beg, end;
    int32_t n_targets;
    int32_t keep_all;      // 1: every record passes (no reader filter: secondary / supplementary / unplaced records too -- bam2cfg reads what `samtools view` prints)
    int32_t mapq_only;     // 1: the quality column is MAPQ whether or not the record carries an AM tag (bam2cfg -m)
};

struct RawColumns {        // columns of every record of a piece, before the reader filter (piece-local index)
    int32_t *tid, *pos, *mtid, *mpos, *isize;
    uint16_t *flag, *qlen;
    uint8_t *mapq, *lib, *keep;
    uint64_t* key;
    uint64_t* check;
};

struct DstColumns {        // where the kept records go (the context's resident store, or the decoder's own buffers)
    int32_t *tid, *pos, *mtid, *mpos, *isize;
    uint16_t *flag, *qlen;
    uint8_t *mapq, *lib, *bam;
    uint64_t* key;
    uint64_t* check;       // nullptr: the destination keeps no second name hash
};

struct PieceState {        // running state of one file's decode, in device memory; one instance per decoder
    uint64_t next_start;   // ring offset where the first record of the next piece starts (the chain's carry)
    uint64_t n_raw;        // records seen so far
    uint64_t n_kept;       // records that passed the filter so far (= records appended to the destination)
    uint32_t error;        // 0 ok; 1 corrupt record chain; 2 record larger than kMaxDeviceRecord; 3 destination full; 4 truncated; 5 a block did not inflate
    uint32_t past_region;  // a record behind the -o region was met (sorted file: nothing of it follows)
    uint32_t piece_raw;    // records of the piece being processed
    uint32_t redo;         // blocks whose guessed start was wrong (statistics)
};

// u: the inflated ring; blocks: the piece's BGZF blocks (out_off = ring offsets, ascending and contiguous);
// avail_end: ring offset up to which inflated bytes are valid behind the piece; first_known: the first block starts the walk
// at st->next_start (always true except for ... nothing: the chain is carried from piece to piece)
void launch_kb_chain(const uint8_t* u, const BgzfBlock* blocks, uint32_t nblk, uint64_t avail_end, int32_t n_targets, ChainBlock* cb,
                     uint16_t* offs, hipStream_t s);
// checks the guesses against the chain of true boundaries (walking blocks again where they disagree), numbers the records:
// rec_base[nblk + 1].  rebase_from / rebase_to: the carried boundary st->next_start is taken as next_start - rebase_from + rebase_to
// (a piece that starts at the ring's front while its predecessor ended at rebase_from)
void launch_kb_stitch(const uint8_t* u, const BgzfBlock* blocks, uint32_t nblk, uint64_t avail_end, int is_last, ChainBlock* cb, uint16_t* offs,
                      uint32_t* rec_base, PieceState* st, const uint32_t* inflate_status, uint64_t rebase_from, uint64_t rebase_to, hipStream_t s);
void launch_kb_extract(const uint8_t* u, const BgzfBlock* blocks, uint32_t nblk, const ChainBlock* cb, const uint16_t* offs,
                       const uint32_t* rec_base, RgTable rg, RecordFilterDev f, RawColumns raw, PieceState* st, hipStream_t s);
// kept records of the piece -> dst at st->n_kept, in order; advances st->n_kept / n_raw; progress (pinned host memory, may be
// null) receives {n_kept, error, past_region, sequence}
void launch_kb_compact(RawColumns raw, uint32_t raw_cap, DstColumns dst, uint64_t dst_cap, uint8_t bam_index, uint32_t* scan_ws, PieceState* st,
                       volatile uint64_t* progress, uint64_t sequence, hipStream_t s);

// ---- several files: their decoded columns gathered into one store in the caller's merge order ----
constexpr int kMaxGatherSources = 16;
struct GatherSource {      // a decoder's own columns
    const int32_t *tid, *pos, *mtid, *mpos, *isize;
    const uint16_t *flag, *qlen;
    const uint8_t *mapq, *lib, *bam;
    const uint64_t *key, *check;
    uint64_t n;
};
struct GatherSources { GatherSource s[kMaxGatherSources]; int k; };
// record i of the destination = record src_index[i] of source src_file[i]; err is raised by an index out of range
void launch_kb_gather(const GatherSources& src, const uint8_t* src_file, const uint32_t* src_index, uint64_t n, DstColumns dst, uint32_t* err, hipStream_t s);

// the 64-bit name key of the host producer (host/bam_reader.cpp hash_name), same function on both sides
__host__ __device__ inline uint64_t name_hash_step(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0xff51afd7ed558ccdull;
    return h ^ (h >> 32);
}
__host__ __device__ inline uint64_t name_hash_finish(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    return h ^ (h >> 32);
}

// the second name hash (bdx_batch::name_check; host/bam_reader.cpp check_name): other constants, a rotation and an addition per
// word instead of the key's xor-shift, seeded by the length differently -- names whose keys collide do not collide here as well
__host__ __device__ inline uint64_t name_check_seed(uint64_t n) { return 0xD6E8FEB86659FD93ull + n * 0x9FB21C651E98DF25ull; }
__host__ __device__ inline uint64_t name_check_step(uint64_t h, uint64_t w) {
    h ^= w;
    h = (h << 27) | (h >> 37);
    return h * 0x9FB21C651E98DF25ull + 0x52DCE729ull;
}
__host__ __device__ inline uint64_t name_check_finish(uint64_t h, uint64_t w) {
    h = name_check_step(h, w);
    h ^= h >> 33;
    h *= 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    h *= 0x165667B19E3779F9ull;
    return h ^ (h >> 32);
}

}  // namespace bdx
