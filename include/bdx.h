/* bdx.h -- C ABI of libbdx, the MI355X-native anomalous read-pair clustering path of BreakDancerMax.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  The reference has no
 * FFI for this path; what it has are C++ seams, and each entry point below names the one it replaces
 * (file:line under the reference's src/).  A maintainer's binding is shown in INTEGRATION.md.
 *
 * Data contract: the caller (the BAM producer) hands over the *merged, position-sorted* stream of
 * records that survive the reference's reader filter (primary, tid >= 0; io/AlignmentFilter.hpp:24-34,
 * io/BamIo.cpp:11-18), as structure-of-arrays batches.  With "-o <chr>" the stream is the records of
 * that tid only (io/RegionLimitedBamReader.hpp:63-71) and opts.chr_restricted is 1.
 */
#ifndef BDX_H
#define BDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bdx_ctx bdx_ctx;

/* status codes (the reference throws C++ exceptions caught in main(), BreakDancerMax.cpp:157-160) */
enum {
    BDX_OK = 0,
    BDX_EINVAL = 1,    /* bad argument */
    BDX_ENOMEM = 2,    /* host or device allocation failed */
    BDX_EHIP = 3,      /* HIP runtime error (no device, launch failure ...) */
    BDX_ESTATE = 4,    /* call out of order (e.g. results before bdx_run) */
    BDX_ELIMIT = 5,    /* input exceeds a documented limit */
    BDX_EINTERNAL = 6
};

/* ReadFlag, common/ReadFlags.hpp:14-27 (values are part of the output contract) */
enum {
    BDX_NA = 0, BDX_ARP_FF = 1, BDX_ARP_LARGE_INSERT = 2, BDX_ARP_SMALL_INSERT = 3, BDX_ARP_RF = 4, BDX_ARP_RR = 5,
    BDX_NORMAL_FR = 6, BDX_NORMAL_RF = 7, BDX_ARP_CTX = 8, BDX_MATE_UNMAPPED = 9, BDX_UNMAPPED = 10, BDX_NUM_FLAGS = 11
};

/* per-read class byte written by the classifier kernel (bdx_get_read_class / bdx_classify):
 *   bits 0-3  ReadFlag after the pass-2 remaps (-l, RR->FF) when the read passes the filters,
 *             the raw classifier flag otherwise
 *   bit 4     read passes the pass-2 filter chain (breakdancer/BreakDancer.cpp:159-167)
 *   bit 5     pass && Alignment::proper_pair() (io/Alignment.hpp:144-148)
 *   bit 6     pass && NORMAL_* && Alignment::leftmost() (counted as a normal read pair, BreakDancer.cpp:202-206) */
#define BDX_CLS_FLAG(c) ((c) & 15)
#define BDX_CLS_PASS 0x10
#define BDX_CLS_PROPER 0x20
#define BDX_CLS_NORMAL_LEFT 0x40

/* Options, common/Options.hpp:24-45 / defaults Options.cpp:27-41 */
typedef struct bdx_opts {
    int32_t min_len;               /* -s */
    int32_t cut_sd;                /* -c (consumed by the config parser) */
    int32_t max_sd;                /* -m */
    int32_t min_map_qual;          /* -q */
    int32_t min_read_pair;         /* -r */
    int32_t seq_coverage_lim;      /* -x */
    int32_t buffer_size;           /* -b */
    int32_t transchr_rearrange;    /* -t */
    int32_t fisher;                /* -f */
    int32_t illumina_long_insert;  /* -l */
    int32_t cn_lib;                /* -a */
    int32_t print_af;              /* -h */
    int32_t score_threshold;       /* -y */
    int32_t chr_restricted;        /* 1 when -o was given (opts.chr non-empty) */
} bdx_opts;

void bdx_opts_default(bdx_opts* o);

/* LibraryConfig, io/LibraryConfig.hpp:11-24; libraries are indexed in sorted-name order
 * (io/BamConfig.cpp:97-101); bam_index indexes the sorted BAM path list (:103-119). */
typedef struct bdx_lib {
    float mean_insertsize, std_insertsize, uppercutoff, lowercutoff, readlens;
    int32_t min_mapping_quality; /* -1: use opts.min_map_qual */
    int32_t bam_index;
} bdx_lib;

/* One batch of records (what io/AlignmentSource.hpp:48-65 + io/Alignment.cpp:45-64 produce per read).
 *   isize     raw BAM isize (the kernels take |isize|)
 *   flag      SAM flag bits
 *   qlen      l_qseq
 *   mapq      "bdqual": AM aux tag if present else MAPQ, as uint8 (io/Alignment.cpp:12-23)
 *   lib       library index resolved from the RG tag (io/BamConfig.hpp:62-72 fallback included)
 *   bam       index of the physical file the record came from (pass-1 counters are per file,
 *             io/BamSummary.cpp:123,135-138)
 *             (an index out of range counts as 0; a context with one library / one file therefore never looks at the
 *             respective array -- it still has to be there)
 *   name_key  64-bit key of the read name; mates share it (ReadRegionData.cpp:109 joins on qname)
 *   name_check a second, independent 64-bit hash of the read name (after bdx_use_name_check; ignored and may be NULL otherwise):
 *             two reads are taken for one name only if key and check both agree, so that two names whose keys collide are
 *             not joined (the reference compares the names themselves, ReadRegionData.cpp:109, SvBuilder.cpp:101-118) */
typedef struct bdx_batch {
    const int32_t *tid, *pos, *mtid, *mpos, *isize;
    const uint16_t *flag, *qlen;
    const uint8_t *mapq, *lib, *bam;
    const uint64_t* name_key;
    size_t n;
    const uint64_t* name_check;
} bdx_batch;

/* Replaces: ConfigLoader + BreakDancer construction (io/ConfigLoader.cpp:18-44, BreakDancer.cpp:87-128).
 * max_read_window_size0 is BamConfig::max_read_window_size() (io/BamConfig.cpp:92-93,121). */
int bdx_create(bdx_ctx** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids,
               int max_read_window_size0, int device);
void bdx_destroy(bdx_ctx* ctx);
const char* bdx_strerror(int code);
const char* bdx_last_error(const bdx_ctx* ctx);

/* Replaces: AlignmentSource::next feeding BamSummary::_analyze_bam and BreakDancer::push_read
 * (io/AlignmentSource.hpp:48-65: one record stream, consumed as it is produced).  Three ways in:
 *
 *   bdx_acquire_batch / bdx_submit_batch   the streaming producer's path.  acquire hands out the columns of a pinned staging
 *       buffer owned by the context (a ring of four; it blocks only while all four are still being copied), the producer
 *       fills the first n records and submits them: the H2D copies run on the context's copy stream and the classifier
 *       (K1) follows on the tiles the batch completed, so decode, PCIe and pass 1 overlap and host memory stays bounded
 *       by the ring.  The buffer belongs to the context again as soon as bdx_submit_batch returns.
 *   bdx_push             a batch in the caller's own memory.  The copies are asynchronous: every array of the batch must
 *       stay valid and unchanged until the next bdx_run on this context has returned.  Arrays in pinned (page-locked)
 *       memory are copied at PCIe speed; pinned name_key and qlen arrays are not copied at all -- only the anomalous reads
 *       (about 1 %) need them and the compaction kernel fetches those straight from the caller's arrays (25 instead
 *       of 35 bytes per read cross the bus; 23 with one library and one BAM, whose index columns are not copied).
 *   bdx_set_device_reads adopts arrays that already live in HBM (no copy; must stay valid until bdx_destroy or
 *       bdx_reset_reads; every array base 16-byte aligned).
 *
 * bdx_reserve sizes the resident store up front (growing it later re-lays the per-tile tables: the classifier then starts
 * over in bdx_run) and, from 2^20 reads on, the buffers of the later stages for the prior a first run goes by (1/32 of the reads
 * anomalous): a context's ~60 device and pinned allocations then happen here, while the caller still decodes or copies, and not
 * inside its first bdx_run.  bdx_reset_reads empties the store (capacity is kept) so that one context can take the next
 * chromosome.  One context holds at most 2^32 - 1 reads. */
typedef struct bdx_batch_buf {
    int32_t *tid, *pos, *mtid, *mpos, *isize;
    uint16_t *flag, *qlen;
    uint8_t *mapq, *lib, *bam;
    uint64_t* name_key;
    size_t capacity;   /* records the buffer holds (>= the capacity asked for) */
    uint64_t* name_check;   /* read by bdx_submit_batch after bdx_use_name_check(ctx, 1) only */
} bdx_batch_buf;
/* Declares that every batch of this context (pushed, staged, adopted, or decoded by a bdx_bamdec) carries name_check.  Only the
 * anomalous reads' values are ever looked at: the mate join compares them on every key match, across GPUs as well (bdx_dist), and
 * the read-level replay tells names apart by the pair.  To be called while the context holds no reads; default off. */
int bdx_use_name_check(bdx_ctx* ctx, int on);
int bdx_reserve(bdx_ctx* ctx, size_t n_reads);
int bdx_push(bdx_ctx* ctx, const bdx_batch* host_batch);
int bdx_acquire_batch(bdx_ctx* ctx, size_t capacity, bdx_batch_buf* out);
int bdx_submit_batch(bdx_ctx* ctx, size_t n);
int bdx_set_device_reads(bdx_ctx* ctx, const bdx_batch* device_batch);
int bdx_reset_reads(bdx_ctx* ctx);

/* Replaces: BamSummary::_analyze_bams (io/BamSummary.cpp:129-150), main()'s density/window block
 * (BreakDancerMax.cpp:83-116) and BreakDancer::run (BreakDancer.cpp:131-144) up to the scored SV list. */
int bdx_run(bdx_ctx* ctx);

/* Replaces: a restored BamSummary (io/ConfigLoader.cpp:19-23, the -R cache).  counters = [nlibs*11 flag histogram |
 * nlibs library read counts | nbams file read counts] in bdx_get_counters' layout, e.g. those of an earlier run over the
 * whole genome; the following bdx_run calls derive window, read densities, lambda and the reported statistics from them
 * instead of from their own reads.  NULL switches back to the run's own statistics. */
int bdx_set_pass1_statistics(bdx_ctx* ctx, const uint32_t* counters, uint32_t covered_ref_len);

/* Replaces: the reference run once per chromosome (`-o <chr>`, README:73; one process each) when the caller holds one context per
 * chromosome on one GPU: bdx_run on every context of the list, at most in_flight of them at a time -- one host thread per context
 * in flight, each context on its own streams, so that one context's latency-bound tail (region cut, join, walk) runs beside
 * another's classifier.  The contexts must be distinct; results through each context's getters.  Returns the first error. */
int bdx_run_many(bdx_ctx* const* ctxs, size_t n, int in_flight);

typedef struct bdx_summary {
    uint64_t n_reads;
    uint64_t n_anomalous;        /* reads entering the region accumulator */
    uint32_t covered_ref_len;    /* BamSummary::covered_reference_length() */
    int32_t window;              /* final max_read_window_size (BreakDancerMax.cpp:109-116) */
    uint32_t n_candidates;       /* candidate regions cut by the sliding window */
    uint32_t n_regions;          /* accepted regions (ReadRegionData::add_region calls) */
    uint32_t n_pairs;            /* mate pairs with both reads in accepted regions */
    uint32_t n_groups;           /* distinct region x region connections */
    uint32_t n_svs;              /* SV candidates that reached scoring */
    uint32_t n_svs_printed;      /* ... with score > opts.score_threshold */
} bdx_summary;
int bdx_get_summary(const bdx_ctx* ctx, bdx_summary* out);

/* pass-1 counters: lib_read_count[nlibs], bam_read_count[nbams], flag_hist[nlibs*11] (BamSummary /
 * LibraryFlagDistribution), seqcov[nlibs] (BamSummary.cpp:140-149), density[nlibs] (read density of the
 * library's key, BreakDancerMax.cpp:94-107).  Any pointer may be NULL. */
int bdx_get_counters(const bdx_ctx* ctx, uint32_t* lib_read_count, uint32_t* bam_read_count, uint32_t* flag_hist,
                     float* seqcov, float* density);

/* BasicRegion (breakdancer/BasicRegion.hpp:24-46) as created by add_region */
typedef struct bdx_region {
    int32_t tid, start, end, normal_read_pairs, fwd_read_count, rev_read_count;
    int32_t n_reads;  /* anomalous reads in the region */
    int32_t stored;   /* reads kept for SV building (ReadRegionData.cpp:118-121) */
    int32_t max_qlen; /* BreakDancer::_max_readlen when the region closed */
} bdx_region;
int bdx_get_regions(const bdx_ctx* ctx, bdx_region* out, size_t cap);

/* One SV candidate = one process_sv call that reached the score (BreakDancer.cpp:348-497).
 * pos[] are 1-based as printed.  lib_* / cn_* index the flat lists below. */
typedef struct bdx_sv {
    int32_t chr[2], pos[2], fwd[2], rev[2];
    int32_t flag;       /* dominant ReadFlag (SvBuilder::choose_sv_flag) */
    int32_t size;       /* diffspan */
    int32_t score;      /* PhredQ */
    int32_t num_reads;  /* flag_counts[flag] */
    int32_t printed;    /* score > opts.score_threshold */
    int32_t region[2];  /* region ids; region[1] = -1 for a single-region SV */
    int32_t lib_begin, lib_count; /* (library, pairs) of the dominant flag, ascending library index */
    int32_t cn_begin, cn_count;   /* (key, copy number); key = library index with -a else BAM index */
    float allele_frequency;
    double logp;        /* ComputeProbScore result (BreakDancer.cpp:44-84) */
} bdx_sv;
int bdx_get_svs(const bdx_ctx* ctx, bdx_sv* out, size_t cap);
int bdx_get_sv_lists(const bdx_ctx* ctx, int32_t* lib_index, int32_t* lib_pairs, size_t lib_cap, int32_t* cn_key,
                     float* cn_value, size_t cn_cap);

/* Supporting reads of every SV candidate (SvBuilder::support_reads, SvBuilder.cpp:101-118), for the -g BED and
 * -d FASTQ dumps (BedWriter.cpp:21-56, BreakDancer.cpp:514-534).  Call bdx_set_collect_support(ctx, 1) before
 * bdx_run; afterwards sv_offsets[i]..sv_offsets[i+1] delimit SV i's reads in read_index (index into the pushed
 * stream) / read_flag (the read's ReadFlag after the pass-2 remaps), in the reference's order: per pair the
 * second-observed mate, then its mate; pairs in observation order.  Single-context runs only. */
int bdx_set_collect_support(bdx_ctx* ctx, int on);
int bdx_get_sv_support(const bdx_ctx* ctx, uint32_t* sv_offsets, uint64_t* read_index, uint8_t* read_flag, size_t cap,
                       size_t* n_total);

/* debug / parity: per-read class byte of the last bdx_run (n_reads bytes) */
int bdx_get_read_class(const bdx_ctx* ctx, uint8_t* out, size_t cap);

/* stage timings of the last bdx_run in milliseconds (HIP events on the context's stream):
 * [0] classify kernel (its own begin-to-end time from kernel-level start/stop events, taken on every 4th run by default,
 *     BDX_K1_EVENT_PERIOD=n; the latest measurement), [1] compaction, [2] region cut, [3] mate join + grouping + SV assembly + scores on the device,
 * [4] host: wait for the host's share of the groups, [5] host walk of that share, [6] host: final wait, merge, score
 * combination, [7] whole run; [8]-[10] split [6] into the final wait, the merge of the device's and the host's SV lists,
 * and the score combination.  Returns the number written. */
int bdx_get_timings(const bdx_ctx* ctx, float* ms, int cap);
/* [1]-[3] need HIP events between the stages, which idle the GPU for a few microseconds each: off by default */
int bdx_set_stage_timing(bdx_ctx* ctx, int on);
/* Enqueue-ahead: the stages behind pass 1 are launched before the pass-1 record is back on the host (which would leave
 * the GPU idle for a host round trip), sized by a guess of the number of anomalous reads; if the guess was too small the
 * device skips them and the host runs them again with the true count.
 *   2 (default)  the guess is the previous run's count where this context has just run an input of the same size,
 *                otherwise a prior of 1/32 of the reads (inputs of a million reads or more)
 *   1            always the prior: every bdx_run behaves like a first run of its input (what bench.py times)
 *   0            off: wait for the pass-1 record, then size exactly */
int bdx_set_enqueue_ahead(bdx_ctx* ctx, int on);

/* Where the SV candidates of the last bdx_run were assembled.  Components of the region graph that are one region, or
 * two regions of one flush window joined by one connection, are walked on the device (build_connection /
 * process_sv, BreakDancer.cpp:266-497); every other component goes through the host walk.  bdx_set_host_walk(ctx, 1)
 * sends everything through the host walk (same results; used by the parity tests).  Any pointer may be NULL. */
int bdx_set_host_walk(bdx_ctx* ctx, int on);
/* Test and measurement switches, by name (they used to be environment variables read inside the library): "no_stash",
 * "max_chunks", "spec_test", "big_walk", "bucketed_join", "no_poll", "finalize2_fold", "no_forward", "scan3", "label_rounds",
 * "k1_grid", "end_write_value", "k1_event_period", "pin_noncoherent", and from round 6: "walk_lanes" (regions per wave of the walk kernel),
 * "ins_plain" (1: the insertion list ranked by the rank-sort launch, 2: by the bitonic fall-back), "gather_walk" (sharded runs, rank 0's walk
 * of the gathered components: 1 device, 2 host), "region_dma" (the region table fetched by copy commands instead of forwarded by the join
 * kernel), "join_fwd" (-1: every joining wave forwards its share; n: that many forwarding workgroups), "regions_copy" (1: the host copies the
 * region table before its share of the walk, 2: never), "asm_plain" (the walk's candidate assembly merges its parts by the three-way merge).  Every switch selects another route to the same results (the parity tests force each
 * route); none is needed in production.  BDX_EINVAL for an unknown name. */
int bdx_set_debug(bdx_ctx* ctx, const char* name, int value);
int bdx_get_walk_split(const bdx_ctx* ctx, uint32_t* n_sv_device, uint32_t* n_sv_host, uint32_t* n_groups_host);
/* After a run, once the caller has what it wants: the result tables are copied out of the pinned host buffers the device assembled them
 * in (every getter keeps working, from the copies) and those buffers -- sized for the worst case of a prior, hundreds of MB for a genome --
 * are handed back.  A process pays for pinned memory when it ends (0.15 s per GB on this platform: tools/exit_cost_probe.hip); the CLI calls
 * this on a thread of its own while it prints the table.  The reference's counterpart is the end of BreakDancer::run (BreakDancer.cpp:131-144):
 * nothing is kept.  A later bdx_run allocates them again. */
int bdx_trim_results(bdx_ctx* ctx);
/* Of the device-assembled candidates, those whose traversal started from a region of an earlier flush window (the
 * reference's flush cadence, BreakDancer.cpp:254-264; they are placed in the output by order key, not by position). */
int bdx_get_cross_window_svs(const bdx_ctx* ctx, uint32_t* n_sv_device);
/* 1 if the last bdx_run met a read name more than twice among the anomalous reads (e.g. merged BAMs with clashing read
 * names) and therefore replayed everything behind the region cut one read at a time on the host, with the reference's
 * semantics for such names (ReadRegionData.cpp:108-113,152-175, SvBuilder.cpp:101-118, BreakDancer.cpp:357-368); 0 otherwise. */
int bdx_was_replayed(const bdx_ctx* ctx);

/* The HIP runtime loads a translation unit's device code at the first launch of one of its kernels (0.4-0.8 ms each, inside whatever
 * run comes first in the process).  bdx_warm_up launches one no-op kernel per translation unit of the library and waits: a process
 * that calls it while it still reads its input has its first run at the speed of the later ones.  bdx_dist_prepare calls it. */
int bdx_warm_up(int device);
/* Process-wide test / measurement switches (the library reads no environment variable for a behaviour switch; BDX_*_TRACE / BDX_*_PROF
 * variables only add output).  "pin_malloc" 1: every pinned buffer comes from hipHostMalloc (default: large ones are registered
 * huge pages).  Takes effect for buffers allocated afterwards.  BDX_EINVAL for an unknown name. */
int bdx_set_process_option(const char* name, int value);

/* Kernel-level entry points for parity tests.
 * bdx_classify replaces IAlignmentClassifier::classify (io/IlluminaPEReadClassifier.cpp:59-101) plus the
 * filter/remap half of push_read: host arrays in, class bytes out.
 * bdx_poisson_log_upper_tail replaces log(cdf(complement(poisson(lambda), k))) (BreakDancer.cpp:64-65). */
int bdx_classify(const bdx_opts* opts, const bdx_lib* libs, int nlibs, const bdx_batch* host_batch, uint8_t* cls_out,
                 int device);
int bdx_poisson_log_upper_tail(const double* lambda, const int32_t* k, double* out, size_t n, int device);

/* ---- staged execution: several contexts (one per chromosome, possibly on several GPUs) sharing one result ----
 * The single-context bdx_run is pass1 -> adopt own statistics -> regions -> join -> walk.  With chromosomes spread
 * over contexts the same stages run per context and the caller exchanges the few global quantities in between
 * (breakdancer_amd/shard.py does it with torch.distributed: all-reduce of the pass-1 counters, all-gather of the
 * per-chromosome totals, all-to-all of the join entries over RCCL, gather of regions/groups to the walking rank):
 *   bdx_stage_pass1          K1 + finalize on this context's reads
 *   bdx_get_pass1_local      counters [nlibs*11 + nlibs + nbams], per-file reference length sums, totals
 *                            [n_anomalous, n_normal_pairs, proper reads per key...]
 *   bdx_set_pass1_global     adopt (all-reduced) counters, covered_ref_len and window (window < 0: derive it)
 *   bdx_stage_compact        K2; nn_base / pk_base = totals of the chromosomes that precede this one; returns the
 *                            qlen / normal-pair count of this chromosome's first anomalous read (it closes the
 *                            previous chromosome's last candidate region, BreakDancer.cpp:202-231)
 *   bdx_stage_regions        K3; has_next / next_qlen / next_nn describe that closing read of the NEXT chromosome
 *   bdx_get_region_records   accepted regions of this context (local ids) + their prefix-count samples
 *   bdx_get_compact          per anomalous read: name key, local region id (-1 none), meta, |isize|
 *   bdx_join_entries         K4 on caller-supplied entries with global region ids / global stream order
 *   bdx_stage_walk           H1 walk + K5 over caller-supplied (global) regions and groups; results through
 *                            bdx_get_summary / bdx_get_svs as usual
 * Not supported in staged runs: negative -s. */
typedef struct bdx_region_rec {
    int32_t tid, start, end;
    uint32_t n_reads, rev_reads, nonctx_reads, normal_read_pairs;
    int32_t max_qlen;
    uint32_t first_read;  /* context-local index of the region's first anomalous read */
} bdx_region_rec;
typedef struct bdx_group {  /* partial aggregate of one (region_lo, region_hi, flag, lib) connection group */
    uint64_t key;           /* region_lo << 38 | region_hi << 12 | lib << 4 | flag */
    uint32_t pairs;
    uint32_t sum_isize;
} bdx_group;
int bdx_stage_pass1(bdx_ctx* ctx);
int bdx_get_pass1_local(const bdx_ctx* ctx, uint32_t* counters, uint64_t* ref_len_per_bam, uint32_t* totals);
int bdx_set_pass1_global(bdx_ctx* ctx, const uint32_t* counters, uint32_t covered_ref_len, int32_t window);
int bdx_stage_compact(bdx_ctx* ctx, uint32_t nn_base, const uint32_t* pk_base, int32_t* first_qlen, uint32_t* first_nn);
int bdx_stage_regions(bdx_ctx* ctx, int has_next, int32_t next_qlen, uint32_t next_nn);
int bdx_get_stage_regions(const bdx_ctx* ctx, uint32_t* n_regions, uint32_t* n_anomalous, int32_t* last_maxq);
int bdx_get_region_records(const bdx_ctx* ctx, bdx_region_rec* out, uint32_t* pk, size_t cap);
int bdx_get_compact(const bdx_ctx* ctx, uint64_t* key, int32_t* region, uint32_t* meta, int32_t* isize, size_t cap);
int bdx_join_entries(bdx_ctx* ctx, size_t n, const uint64_t* key, const uint32_t* order, const int32_t* region,
                     const uint32_t* meta, const int32_t* isize, bdx_group* out, size_t cap, uint32_t* n_groups,
                     uint32_t* n_pairs);
int bdx_stage_walk(bdx_ctx* ctx, size_t nregions, const bdx_region_rec* regions, const uint32_t* pk, size_t ngroups,
                   const bdx_group* groups, int32_t last_maxq, int any_anomalous);

/* ---- chromosome-sharded runs on several GPUs: ONE whole-genome result (what a single `breakdancer-max cfg` run prints,
 * -t included) with the chromosomes spread over ranks, one rank per GPU.  Regions never span chromosomes
 * (breakdancer/BreakDancer.cpp:216), so every rank runs classify / compact / region cut / mate join on its own chromosomes;
 * the ranks exchange the global pass-1 statistics and per-chromosome totals (all-reduces of a few KB), the inter-chromosomal
 * (ARP_CTX) join records in ONE all-to-all to owner(name key) -- pairs with both mates on one chromosome never leave their
 * GPU --, and finally gather region tables and pair groups on rank 0, which walks the region graph and holds the result.
 * The collectives run on device buffers over RCCL (xGMI inside a node).
 *
 *   bdx_dist_unique_id       rank 0 obtains the communicator id (ncclGetUniqueId); the caller hands the 128 bytes to every
 *                            rank by whatever means it has (torch.distributed, MPI, a file)
 *   bdx_dist_create          one per process: joins the communicator (ncclCommInitRank) -- collective
 *   bdx_dist_create_threads  the same ranks as threads of ONE process (out[world], one per device in `devices`, which may
 *                            repeat a device); every out[r] is then driven by its own thread.  Collectives are
 *                            device-to-device copies around a barrier
 *   bdx_dist_chromosome      the context that takes chromosome `tid`'s records: ONE context per rank holds all of the rank's
 *                            chromosomes (the same handle for every tid), fed with bdx_push / bdx_acquire_batch as usual -- a rank's
 *                            chromosomes in ASCENDING order, each chromosome's records together --; do NOT call bdx_run on it.
 *                            ntids = number of reference sequences; every chromosome that has reads is fed to exactly one rank
 *   bdx_dist_prepare         optional, after the chromosomes are loaded: sizes the buffers of the later stages for the prior a
 *                            first run goes by (what bdx_reserve does for a single context), outside the run
 *   bdx_dist_run             collective: all ranks call it once their chromosomes are loaded
 *   bdx_dist_result          rank 0 after bdx_dist_run: a context whose getters (bdx_get_summary, bdx_get_counters,
 *                            bdx_get_regions, bdx_get_svs, bdx_get_sv_lists) return the whole-genome result; NULL on other ranks
 *   bdx_dist_owner           the rank that takes the census of a name key (the routing rule of the name census; the inter-chromosomal
 *                            join records travel to the rank that holds the LATER of their two chromosomes)
 *   bdx_dist_plan            chromosomes -> ranks by longest-processing-time packing on `weight` (reads or length)
 * The -g/-d support lists (bdx_dist_set_collect_support), read names that occur more than twice and a negative -s (the read-less region 0
 * of BreakDancer.cpp:244-264, registered once for the genome) are served by the read-level walk: the compact records are gathered and
 * rank 0 walks them read by read. */
typedef struct bdx_dist bdx_dist;
typedef struct bdx_unique_id { char internal[128]; } bdx_unique_id;
int bdx_dist_unique_id(bdx_unique_id* out);
int bdx_dist_create(bdx_dist** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids,
                    int max_read_window_size0, int device, int rank, int world, const bdx_unique_id* id);
int bdx_dist_create_threads(bdx_dist** out, const bdx_opts* opts, const bdx_lib* libs, int nlibs, int nbams, int ntids,
                            int max_read_window_size0, const int* devices, int world);
void bdx_dist_destroy(bdx_dist* d);
const char* bdx_dist_last_error(const bdx_dist* d);
int bdx_dist_rank(const bdx_dist* d);
int bdx_dist_world(const bdx_dist* d);
bdx_ctx* bdx_dist_chromosome(bdx_dist* d, int tid);
int bdx_dist_prepare(bdx_dist* d);
int bdx_dist_reset_reads(bdx_dist* d);   /* empties the rank's store (capacity kept): the chromosomes are fed again */
int bdx_dist_run(bdx_dist* d);
bdx_ctx* bdx_dist_result(bdx_dist* d);
/* before bdx_dist_run, on every rank alike: the result context also holds the supporting reads of every SV (bdx_get_sv_support on
 * bdx_dist_result; read_index = position in the merged stream of the whole run, chromosomes ascending) -- the input of the
 * reference's -g / -d dumps (BreakDancer.cpp:514-534).  The compact records of all chromosomes are then gathered and rank 0 walks
 * them read by read, as for read names seen more than twice. */
int bdx_dist_set_collect_support(bdx_dist* d, int on);
/* after bdx_dist_run: CTX join records this rank sent / received in the all-to-all, bytes gathered on rank 0, wall time
 * of the whole run and of the exchange + CTX join (ms).  Any pointer may be NULL. */
int bdx_dist_get_exchange(const bdx_dist* d, uint64_t* ctx_records_sent, uint64_t* ctx_records_received, uint64_t* gathered_bytes,
                          float* ms_total, float* ms_exchange);
/* collectives this rank entered in the last bdx_dist_run: out[0] all-reduces, out[1] all-to-alls, out[2] gathers; the library that
 * carried them ("threads" for the in-process backend, else the path of the librccl that was loaded) and its ncclGetVersion code (0: none) */
int bdx_dist_get_collectives(const bdx_dist* d, uint32_t out[3], const char** backend, int* rccl_version);
/* bdx_set_debug for this rank's contexts -- its own and, on rank 0, the result context ("gather_walk": 1 = rank 0 walks the gathered
 * components on its device whatever their number, 2 = on its host; 0 = by their number).  Routes to the same results; tests force each */
int bdx_dist_set_debug(bdx_dist* d, const char* name, int value);
/* this rank's milliseconds of the last bdx_dist_run, phase by phase: local phases and the collectives behind them alternate
 * (bdx_dist_phase_name(i) names entry i; a collective's figure includes waiting for the slowest rank), then what only rank 0 does:
 * the merge of the ranks' tables, its walk of the gathered components on the device (K6 on the result context) and what of it its host takes. */
int bdx_dist_get_phase_ms(const bdx_dist* d, float* out, int n);
const char* bdx_dist_phase_name(int i);
int bdx_dist_owner(uint64_t name_key, int world);
int bdx_dist_plan(const uint64_t* weight, int ntids, int world, int* rank_of_tid);

/* ---- BAM decode on the device: BGZF inflate + record fields in HBM ----
 * Replaces, for one BAM file, what the host producer does per byte and per record: bgzf inflate (samtools bgzf.c behind
 * io/BamReader.hpp:62-70), bam_read1, Alignment's constructor (io/Alignment.cpp:12-29,45-64: core fields, bdqual from the AM tag,
 * RG), the RG -> library lookup with its fallback (io/BamConfig.hpp:62-72, io/AlignmentSource.hpp:57-62) and the reader filter
 * (primary, placed: io/AlignmentFilter.hpp:24-34, io/BamIo.cpp:11-18; -o region: io/RegionLimitedBamReader.hpp:63-71).  The caller
 * reads the file, finds the BGZF members (18-byte headers, 8-byte footers) and hands the bytes over as they are, in pieces of whole
 * members; everything else happens on the GPU, and the file crosses PCIe compressed.
 * Integrity: the inflate kernel rejects what zlib rejects (invalid codes, distances before the window, a member that does not end
 * at its ISIZE) and the record stage rejects chains that do not add up; the members' CRC-32 words are NOT checked on the device --
 * the host reader (host/column_reader.cpp, BDX_DECODE=host) checks every member's, as htslib does.
 *
 *   bdx_bamdec_create    sink != NULL: the decoded records are appended to that context's resident store (one file: the stream
 *                        IS the file's record order) and the classifier follows them; bdx_run afterwards as usual.  sink == NULL:
 *                        the decoder keeps the columns in HBM and bdx_bamdec_fetch copies them out (tests; callers that merge
 *                        several files themselves).  first_record_offset: offset of the first record in the inflated stream,
 *                        counted from the first member that will be submitted (the caller has parsed the BAM header).
 *   bdx_bamdec_acquire   a pinned staging buffer for `bytes` of file and a table of max_blocks members, both owned by the decoder
 *                        (a ring of six; blocks only while the next one's copy is in flight).  A caller may acquire several before it
 *                        submits -- reading the file ahead of the piece it is cutting into members -- up to all six; bdx_bamdec_submit
 *                        takes them in the order they were acquired, bdx_bamdec_finish drops what was acquired and never submitted
 *   bdx_bamdec_submit    the first `bytes` of the buffer are whole members, described by the first nblocks table entries
 *                        (offset = start of the member's deflate payload in the buffer); last != 0 with the file's final piece.
 *                        Asynchronous: the copy is enqueued at once; inflate and record kernels run per BATCH of pieces (one member is
 *                        one wavefront, so a launch wants thousands of them), the record kernels one batch behind
 *   bdx_bamdec_progress  non-blocking: records appended so far, records seen, whether a record behind the -o region was met
 *                        (a caller that seeked through the index may stop there), device-side error code
 *   bdx_bamdec_finish    waits for everything; errors of the decode (corrupt block, corrupt or truncated record chain, a record
 *                        beyond the device path's 4 MiB) surface here.  Without a `last` piece it ends the stream where it is
 *   bdx_inflate_blocks   kernel-level entry point for parity tests: members inflated by the GPU, host buffers in and out;
 *                        status[i] != 0: member i was rejected (the caller lets zlib judge it) */
typedef struct bdx_bamdec bdx_bamdec;
typedef struct bdx_bgzf_block {
    uint64_t offset;        /* of the member's deflate payload within the submitted bytes */
    uint32_t payload_len;   /* BSIZE + 1 - XLEN - 20 */
    uint32_t inflated_len;  /* ISIZE, at most 65536 */
} bdx_bgzf_block;
typedef struct bdx_bamdec_params {
    int32_t device;               /* used when there is no sink */
    int32_t n_targets;            /* reference sequences of the file's header */
    int32_t bam_index;            /* index of the file among the configuration's BAMs (the records' `bam` column) */
    int32_t only_tid, region_beg, region_end;  /* -o region; only_tid < 0: every placed record */
    uint32_t n_read_groups;       /* read-group ids of the configuration and their library indices */
    const char* const* rg_ids;
    const uint8_t* rg_lib;
    uint8_t fallback_lib;         /* library of records without (or with an unknown) read group */
    uint64_t first_record_offset;
    size_t ring_bytes;            /* inflated bytes kept in flight (at least four batches' worth), 0: 3 GiB */
    size_t batch_bytes;           /* compressed bytes and members after which the pieces gathered so far are launched as one batch */
    size_t batch_blocks;          /* (0: 256 MiB / 8192 members -- more members than the GPU has wave slots for the inflate kernel) */
    size_t expected_bytes;        /* compressed bytes the caller is going to submit (the file's size), 0: unknown.  With a sink whose
                                     store is still empty the decoder sizes the store from it, and once the first batch has shown how
                                     many records the bytes hold, the buffers of the stages behind pass 1 (what bdx_reserve does up
                                     front) -- while the GPU inflates, not in front of it */
    size_t piece_bytes, piece_blocks;  /* the sizes bdx_bamdec_acquire will be called with, 0: unknown.  Known: the staging buffers are
                                     pinned by threads of their own while the caller reads its first piece (pinning costs ~0.2 ms per
                                     MiB, and the first pieces would otherwise wait for it one after the other) */
    int32_t batch_rounds;         /* rounds of the GPU's wave slots an inflate launch takes (7,680 members each), 1..16; 0: from expected_bytes
                                     (one round per 2.5 GB, at most four).  Ignored when batch_blocks is given */
    int32_t stream_mode;          /* 0: inflate launches and record stages take turns on one stream (the product's arrangement);
                                     1: the inflate launches on a stream of their own beside the record stages (round 4's arrangement);
                                     2: as 1, with queue priorities.  1 and 2 are measurement / test arrangements */
    int32_t record_mode;          /* bits: 1 = no reader filter (secondary / supplementary / unplaced records are kept too), 2 = the quality column is
                                     MAPQ whether or not a record carries an AM tag.  0 for breakdancer-max's reader; bam2cfg sets them */
    int32_t missing_lib_plus1;    /* library of records WITHOUT a read-group tag, plus one; 0: fallback_lib, like an unknown read group */
    int32_t time_kernels;         /* != 0: a HIP event pair around every inflate launch; bdx_bamdec_host_ms [12] = their sum in ms, [13] = launches
                                     (a measurement: each pair idles the GPU for a few microseconds) */
} bdx_bamdec_params;
int bdx_bamdec_create(bdx_bamdec** out, bdx_ctx* sink, const bdx_bamdec_params* p);
void bdx_bamdec_destroy(bdx_bamdec* d);
const char* bdx_bamdec_last_error(const bdx_bamdec* d);
int bdx_bamdec_acquire(bdx_bamdec* d, size_t bytes, size_t max_blocks, void** buf, bdx_bgzf_block** blocks);
int bdx_bamdec_submit(bdx_bamdec* d, size_t bytes, size_t nblocks, int last);
int bdx_bamdec_progress(bdx_bamdec* d, uint64_t* n_records, uint64_t* n_raw, int* past_region, uint32_t* error);
int bdx_bamdec_finish(bdx_bamdec* d, uint64_t* n_records);
/* a finished decoder takes another stretch of the same file (another sequence's range, found through the index): buffers, streams and
 * events are kept, the region filter and the record chain start over; with a sink the records go behind what its store holds */
int bdx_bamdec_rearm(bdx_bamdec* d, int32_t only_tid, int32_t region_beg, int32_t region_end, uint64_t first_record_offset, size_t expected_bytes);
int bdx_bamdec_fetch(bdx_bamdec* d, uint64_t first, uint64_t n, const bdx_batch_buf* out);
int bdx_bamdec_stats(const bdx_bamdec* d, uint64_t* compressed_bytes, uint64_t* inflated_bytes, uint64_t* pieces, uint64_t* blocks_walked_twice);
/* Several BAMs decoded on the GPU, each by a decoder of its own (no sink: the records stay in the decoder's columns, bdx_bamdec_fetch
 * returns the columns the caller gives room for -- tid, pos and flag are what BamMerger's order looks at, io/BamMerger.cpp:40-61).  The
 * caller works out the merged order and hands it over as a permutation: record i of the context's store = record src_index[i] of
 * decoder src_file[i].  The gather runs in HBM; the context must hold no reads yet and then stands as after bdx_push of the merged stream. */
int bdx_merge_decoded(bdx_ctx* ctx, bdx_bamdec* const* decs, int k, const uint8_t* src_file, const uint32_t* src_index, uint64_t n);
/* The same gather BEHIND what the context's store holds (a rank of a sharded run takes its chromosomes one after the other: every chromosome
 * of a several-BAM configuration is decoded per file through the files' indexes, io/RegionLimitedBamReader.hpp:43-71, and merged in
 * BamMerger's order, io/BamMerger.cpp:40-126).  The store keeps its records; it must not hold batches whose name keys are still the caller's. */
int bdx_append_decoded(bdx_ctx* ctx, bdx_bamdec* const* decs, int k, const uint8_t* src_file, const uint32_t* src_index, uint64_t n);
/* milliseconds the feeding thread spent inside the decoder so far, by cause: [0] waiting for a staging buffer's copy, [1] pinning
 * staging memory, [2] waiting for a batch slot, [3] sizing a slot's buffers, [4] the pieces' copy calls, [5] launching batches
 * (includes [6]), [6] launching record stages, [7] feeding the classifier; and two marks, ms after the decoder's creation (or its last
 * bdx_bamdec_rearm): [8] the first inflate launch, [9] bdx_bamdec_finish's return; [10] / [11] of [7]: sizing the later stages' buffers,
 * classifier launches; [12] / [13] the inflate kernel's own milliseconds by HIP events and its launches (bdx_bamdec_params::time_kernels) */
int bdx_bamdec_host_ms(const bdx_bamdec* d, float* out, int n);
int bdx_inflate_blocks(int device, const void* compressed, size_t bytes, const bdx_bgzf_block* blocks, size_t nblocks, void* out,
                       size_t out_bytes, uint32_t* status, float* kernel_ms);

/* bam2cfg's insert-size statistics of nlibs libraries on the GPU (perl/bam2cfg.pl:153-197; `bam2cfg --device`): library i's observations
 * are x[offsets[i] .. offsets[i + 1]).  mean_all / sd_all over all of them (n - 1); mean / sd over the n_kept that are not more than five
 * standard deviations above mean_all; sd_minus / sd_plus the one-sided deviations around mean (n_minus observations <= mean, n_plus above,
 * each with n - 1).  Summed in the script's order with round-to-nearest operations: the figures equal the CPU tool's bit for bit. */
typedef struct bdx_insert_stats {
    double mean_all, sd_all, mean, sd, sd_minus, sd_plus;
    uint64_t n_kept, n_minus, n_plus;
} bdx_insert_stats;
int bdx_insert_size_stats(int device, const double* x, const uint32_t* offsets, int nlibs, bdx_insert_stats* out);

/* device the context is bound to and the HIP stream it launches on (as void*), for callers that time it */
int bdx_device(const bdx_ctx* ctx);
void* bdx_stream(const bdx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* BDX_H */
