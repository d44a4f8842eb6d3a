// Minimal BGZF/BAM record reader (zlib): the host-side producer of SoA read records.
// Stands where the reference uses samtools 0.1.19 (io/BamReader.hpp:62-70 -> samread); only the core fields,
// the read name and the RG / AM aux tags are extracted (io/Alignment.cpp:12-29,45-64).
#pragma once
#include <cstdint>
#include <cstdio>
#include <future>
#include <memory>
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
    uint64_t name_key;      // hash_name(qname)
    uint64_t rg_key;        // hash_name(rg), 0 without an RG tag: lets a consumer recognise runs of one read group without
                            // touching the record's bytes (they were decoded on another core)
};

class BamReader {
public:
    // fill_blocks: BGZF blocks inflated per batch, 0 = the default (2048: <= 128 MiB).  A caller that stops after the first few
    // thousand records (bam2cfg) asks for small batches: the batch after the current one is always inflated ahead
    explicit BamReader(const std::string& path, int threads = 4, size_t fill_blocks = 0);
    ~BamReader();
    BamReader(const BamReader&) = delete;
    BamReader& operator=(const BamReader&) = delete;

    const std::string& path() const { return path_; }
    const std::vector<std::string>& target_names() const { return targets_; }
    const std::string& header_text() const { return header_text_; }  // the SAM header (@HD/@SQ/@RG ... lines)
    int tid_of(const std::string& name) const;  // -1 if absent
    // next record of the file (no filtering); false at end of file
    bool next(BamRecord& r);
    // decode the record whose block_size word is at p (thread-safe: touches nothing but its arguments)
    static void parse_record(const uint8_t* p, BamRecord& r);

private:
    // decompressed bytes of one batch of BGZF blocks: valid in [beg, end) of a buffer that is allocated once and never
    // zero-filled; the gap in front takes the unparsed tail of the previous batch (a record can straddle two batches)
    struct Chunk {
        std::unique_ptr<uint8_t[]> data;
        size_t cap = 0, beg = 0, end = 0;
        // the records that start in [beg, tail): decoded by several threads as soon as the batch is inflated, each from a
        // guessed (and then verified) record boundary -- see parse_chunk
        std::vector<std::vector<BamRecord>> parts;
        size_t tail = 0;               // first byte of the incomplete record at the end of the batch (== end if none)
        void reserve(size_t n);
    };
    bool fill(Chunk& c);               // inflate the next batch of BGZF blocks into c; false at EOF (runs on the helper thread)
    void attach_tail(Chunk& c, const uint8_t* src, size_t n);  // put n bytes in front of c's bytes
    void parse_chunk(Chunk& c);        // decode every complete record of c (several threads)
    bool fill_and_parse(Chunk& c, const Chunk* prev);  // the helper thread's job in record mode
    bool advance();                    // header mode: make the next batch current; false at EOF
    bool advance_records();            // record mode: take over the batch the helper thread prepared, start the next one
    bool ensure(size_t need);          // header mode: make `need` decompressed bytes available at cur_
    const uint8_t* at() const { return chunk_[cur_chunk_].data.get() + cur_; }
    std::string path_;
    const uint8_t* map_ = nullptr;     // the compressed file, memory-mapped
    size_t map_size_ = 0;
    int threads_;
    size_t fill_blocks_ = 0;
    size_t comp_off_ = 0;              // first compressed byte not yet consumed
    Chunk chunk_[2];                   // one being parsed, one being inflated
    int cur_chunk_ = 1;                // (the first batch lands in chunk 0)
    size_t cur_ = 0, end_ = 0;         // parse position / end of the valid bytes in the current chunk
    std::future<bool> next_ready_;     // the helper thread's fill_and_parse() of the other chunk
    bool records_mode_ = false;        // false while the header is read byte-wise
    size_t part_ = 0, rec_ = 0;        // next record to hand out: chunk_[cur_chunk_].parts[part_][rec_]
    std::vector<std::string> targets_;
    std::string header_text_;
};

uint64_t hash_name(const char* s, size_t n);  // 64-bit name key shared by the two mates of a pair
uint64_t check_name(const char* s, size_t n); // a second, independent hash of the name (bdx_batch::name_check)

// first position in [from, seg_end) of `data` where three BAM records in a row look valid (field ranges, size equation, read
// name; `end` bounds what may be read); seg_end if there is none.  A guess: callers verify it against the true boundary chain
size_t bam_guess_record_start(const uint8_t* data, size_t from, size_t seg_end, size_t end, int32_t n_targets);

}  // namespace bdhost
