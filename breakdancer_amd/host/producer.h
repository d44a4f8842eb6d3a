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
    std::vector<uint64_t> name_key;
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

// chr: empty = all sequences, otherwise the -o region in samtools syntax ("name", "name:beg" or "name:beg-end")
void produce(const BamConfig& cfg, const std::string& chr, int threads, ReadStream& out);

}  // namespace bdhost
