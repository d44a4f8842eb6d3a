// Host producer: BAM files -> one merged, position-sorted SoA record stream (the bdx_batch layout).
// Stands where the reference has AlignmentSource over BamMerger over BamReader<IsPrimary && IsAligned>
// (io/AlignmentSource.hpp:48-65, io/BamMerger.cpp:40-126, io/BamIo.cpp:6-31), decoding every BAM ONCE
// (the reference decodes each file twice: pass 1 per file, pass 2 merged).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "bdx.h"
#include "config.h"

namespace bdhost {

struct ReadStream {
    std::vector<int32_t> tid, pos, mtid, mpos, isize;
    std::vector<uint16_t> flag, qlen;
    std::vector<uint8_t> mapq, lib, bam;
    std::vector<uint64_t> name_key, name_check;
    std::vector<std::string> targets;  // sequence names of the first BAM (BamMerger.cpp:78)
    size_t size() const { return tid.size(); }
    bdx_batch batch() const;
};

// One supporting read kept for the -g BED / -d FASTQ dumps (what Alignment holds when seq_data is on,
// io/Alignment.cpp:45-64)
struct SupportRead {
    int32_t tid = 0, pos = 0, l_qseq = 0;
    uint8_t bdqual = 0, lib = 0;
    bool rev = false, has_qual = false;
    std::string name, bases, qual;  // bases already decoded to letters, qual = raw phred bytes
};

// Second decode pass for the dumps: replays the merge and keeps the records whose stream index is in `wanted`
// (sorted, unique); out[i] corresponds to wanted[i].
void collect_reads(const BamConfig& cfg, const std::string& chr, int threads, const std::vector<uint64_t>& wanted,
                   std::vector<SupportRead>& out);

// Where the producer puts the merged stream: batches of SoA columns.  acquire() returns columns with room for at least
// `capacity` records, submit(n) hands the first n back.  Two sinks exist: the GPU context's pinned staging ring
// (bdx_acquire_batch / bdx_submit_batch: copies and the classifier overlap the decoding of the next batch) and ReadStream.
struct BatchSink {
    virtual ~BatchSink() {}
    virtual bdx_batch_buf acquire(size_t capacity) = 0;
    virtual void submit(size_t n) = 0;
};

// chr: empty = all sequences, otherwise the -o region in samtools syntax ("name", "name:beg" or "name:beg-end").
// Every BAM is decoded once, by `threads` threads in all (column_reader.h); several files are merged in the reference's
// order (io/BamMerger.cpp:40-126).  Returns the number of records; targets receives the first file's sequence names.
size_t produce_stream(const BamConfig& cfg, const std::string& chr, int threads, std::vector<std::string>* targets, BatchSink& sink,
                      size_t batch_records = 1u << 20);
// The same stream for a configuration of ONE BAM, decoded on the GPU (bdx_bamdec_*, include/bdx.h): this side reads the file into
// the decoder's pinned staging buffers (several threads), finds the BGZF members and submits them; inflate, record boundaries,
// fields, RG -> library and the reader filter run in HBM, and the records land in ctx's resident store with the classifier
// behind them.  Returns the number of records appended.  Throws std::runtime_error; `unsupported` (may be null) is set instead
// when the file is one the device path leaves to the host reader (a record of more than 4 MiB), with nothing appended.
size_t produce_on_device(const BamConfig& cfg, const std::string& chr, int threads, std::vector<std::string>* targets, bdx_ctx* ctx,
                         bool* unsupported);
// One whole-genome run over several GPUs from ONE indexed BAM: rank r's thread decodes the chromosomes t with rank_of[t] == r on
// devices[r], into bdx_dist_chromosome(ranks[r], t).  *unsupported: no index / several files -- nothing was appended, or a file the
// device path gives up on (the caller then recreates the ranks and takes the host producer).
size_t produce_sharded_on_device(const BamConfig& cfg, int threads, std::vector<std::string>* targets, const std::vector<bdx_dist*>& ranks,
                                 const std::vector<int>& devices, const std::vector<int>& rank_of, bool* unsupported);
// the decoders produce_on_device used are kept (releasing them costs more than the run that follows): this releases them
void release_device_decoders();
// reference sequences of the first BAM of the configuration (names and lengths from its header; io/BamMerger.cpp:78)
void read_targets(const BamConfig& cfg, std::vector<std::string>& names, std::vector<uint32_t>& lengths);
void produce(const BamConfig& cfg, const std::string& chr, int threads, ReadStream& out);
// BamMerger's order worked out from the files' (tid, pos, flag) columns alone: entry i of the merged stream is record src_index[i] of
// file src_file[i] (what the device path hands to bdx_merge_decoded).  emitted_last (two files): the file that emitted the stream's last
// record in FRONT of these -- a tie at the very first position goes to the other one, which has been waiting (1: as a queue filled afresh)
void merge_order(const std::vector<const int32_t*>& tid, const std::vector<const int32_t*>& pos, const std::vector<const uint16_t*>& flag,
                 const std::vector<size_t>& n, std::vector<uint8_t>& src_file, std::vector<uint32_t>& src_index, int threads = 1, int emitted_last = 1);
// the same merged stream as produce(), by way of merge_order: every file decoded on its own, then permuted (bdx-dump-reads uses it to
// hold the two merges against each other)
void produce_merged_by_columns(const BamConfig& cfg, const std::string& chr, int threads, ReadStream& out);

}  // namespace bdhost
