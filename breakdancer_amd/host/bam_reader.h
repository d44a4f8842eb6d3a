// Minimal BGZF/BAM record reader (zlib): the host-side producer of SoA read records.
// Stands where the reference uses samtools 0.1.19 (io/BamReader.hpp:62-70 -> samread); only the core fields,
// the read name and the RG / AM aux tags are extracted (io/Alignment.cpp:12-29,45-64).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace bdhost {

struct BamRecord {
    int32_t tid, pos, mtid, mpos, isize, l_qseq;
    uint16_t flag;
    uint8_t mapq;
    uint8_t bdqual;         // AM aux tag if present else MAPQ, truncated to uint8 like the reference
    const char* qname;      // points into the reader's buffer, valid until the next call
    uint32_t l_qname;       // without the trailing NUL
    const char* rg;         // RG:Z value or nullptr
    uint32_t l_rg;
    const uint8_t* seq;     // 4-bit packed bases, (l_qseq+1)/2 bytes
    const uint8_t* qual;    // l_qseq bytes
    int32_t end_pos;        // bam_calend: pos + reference bases consumed by the CIGAR (pos + 1 without a CIGAR)
};

class BamReader {
public:
    explicit BamReader(const std::string& path, int threads = 4);
    ~BamReader();
    BamReader(const BamReader&) = delete;
    BamReader& operator=(const BamReader&) = delete;

    const std::string& path() const { return path_; }
    const std::vector<std::string>& target_names() const { return targets_; }
    const std::string& header_text() const { return header_text_; }  // the SAM header (@HD/@SQ/@RG ... lines)
    int tid_of(const std::string& name) const;  // -1 if absent
    // next record of the file (no filtering); false at end of file
    bool next(BamRecord& r);

private:
    bool fill();                       // inflate the next batch of BGZF blocks; false at EOF
    bool ensure(size_t need);          // make `need` decompressed bytes available at cur_
    std::string path_;
    FILE* fp_ = nullptr;
    int threads_;
    std::vector<uint8_t> comp_;        // compressed bytes not yet consumed
    size_t comp_off_ = 0;
    bool eof_ = false;
    std::vector<uint8_t> buf_;         // decompressed bytes
    size_t cur_ = 0;
    std::vector<std::string> targets_;
    std::string header_text_;
};

uint64_t hash_name(const char* s, size_t n);  // 64-bit name key shared by the two mates of a pair

}  // namespace bdhost
