// ColumnReader: one BAM file -> the SoA columns of its records that pass the reader filter, decoded by a pool of
// threads, handed out in file order.
//
// Stands where the reference has AlignmentSource over BamReader<IsPrimary && IsAligned> (io/AlignmentSource.hpp:48-65,
// io/BamIo.cpp:6-31) and, per record, Alignment's constructor (io/Alignment.cpp:12-29,45-64) and the RG -> library
// lookup (io/BamConfig.hpp:62-72).  Unlike bam_reader.h's BamReader (one record at a time, used for the -d/-g dumps and
// by bam2cfg) nothing here is per-record on the consuming thread: the file is cut into pieces of kPieceBlocks BGZF
// blocks, every piece is inflated, split into records, filtered and turned into columns by ONE worker, and the consumer
// only checks that consecutive pieces agree on their record boundaries and takes the columns.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "config.h"

namespace bdhost {

// read-group string -> library index with the reference's fallback (io/BamConfig.hpp:62-72, io/AlignmentSource.hpp:57-62);
// immutable after construction, shared by the decode threads
class LibraryResolver {
public:
    explicit LibraryResolver(const BamConfig& cfg);
    uint8_t of(const char* rg, uint32_t len) const;
    uint8_t fallback() const { return fallback_; }

private:
    std::unordered_map<std::string, uint8_t> by_rg_;
    uint8_t fallback_ = 0;
};

struct RecordFilter {  // -o <region>: one tid, records overlapping [beg, end) (bam_index.c:571-576 is_overlap); tid < 0: all
    int only_tid = -1;
    int beg = 0, end = 1 << 29;
};

struct ColumnChunk {
    std::vector<int32_t> tid, pos, mtid, mpos, isize;
    std::vector<uint16_t> flag, qlen;
    std::vector<uint8_t> mapq, lib;
    std::vector<uint64_t> name_key, name_check;
    size_t size() const { return tid.size(); }
    void clear() {
        tid.clear(); pos.clear(); mtid.clear(); mpos.clear(); isize.clear(); flag.clear(); qlen.clear(); mapq.clear(); lib.clear();
        name_key.clear(); name_check.clear();
    }
};

class ColumnReader {
public:
    ColumnReader(const std::string& path, int threads, const LibraryResolver* libs);
    ~ColumnReader();
    ColumnReader(const ColumnReader&) = delete;
    ColumnReader& operator=(const ColumnReader&) = delete;

    const std::string& path() const { return path_; }
    const std::vector<std::string>& target_names() const { return targets_; }
    const std::vector<uint32_t>& target_lengths() const { return target_len_; }
    int tid_of(const std::string& name) const;
    // start decoding; must be called once, before next()
    void start(const RecordFilter& f);
    // Where decoding would start for this filter, without starting it (the device-side decoder's feeder, producer.cpp): file
    // offset of the BGZF member that holds the first record to look at, the record's offset in that member's inflated bytes,
    // and whether the BAM index was used to get there (then the records behind the region may be left unread).
    void locate(const RecordFilter& f, size_t* member_offset, uint64_t* record_offset, bool* seeked);
    const uint8_t* mapped() const { return map_; }
    size_t mapped_size() const { return map_size_; }
    // <path>.bai: the range of the file that holds sequence tid's records -- [begin, end) in compressed bytes, member aligned at its
    // begin, a member's worth of slack at its end.  false: no index, or nothing indexed for the sequence (*empty tells which)
    bool index_span(int tid, size_t* begin, size_t* end, bool* empty) const;
    // the next piece's columns in file order, or nullptr at the end of the file; valid until the following call
    const ColumnChunk* next();

private:
    struct Block {
        size_t coff, clen;  // deflate payload within the mapped file
        uint32_t ulen;
        uint64_t uabs;      // offset of the block's first byte in the uncompressed stream
    };
    struct Piece {
        size_t index = 0, b0 = 0, b1 = 0;   // blocks [b0, b1)
        uint64_t abs_begin = 0, abs_end = 0; // their range of the uncompressed stream
        uint64_t first_abs = 0;              // where this piece's first record starts (guessed, except for the first piece)
        uint64_t next_abs = 0;               // where the first record of the following piece starts
        bool found_start = false;
        bool past_region = false;            // met a record behind the -o region (sorted file: nothing of it follows)
        ColumnChunk cols;
        std::string error;
        bool done = false;
    };

    struct Scratch {                         // a decode thread's inflate buffer
        std::vector<uint8_t> buf;            // inflated bytes of the piece from abs_begin on (plus what its last record needed beyond)
        size_t filled = 0;
        size_t extra_blocks = 0;             // blocks beyond b1 inflated for the last record
        size_t piece = 0;                    // index + 1 of the piece the buffer holds
        size_t first_block = 0;              // the piece's block descriptors, copied out of the shared index
        std::vector<Block> blocks;
    };

    void read_header();
    bool seek_with_index(int tid, int beg);   // <path>.bai: start at the first block that can hold a record of the region
    void scan_blocks();                       // scanner thread: the block index, then the pieces' queue
    void worker();
    void decode_piece(Scratch& sc, Piece& p, bool known_start, uint64_t start_abs);
    void inflate_into(Scratch& sc, const Piece& p, size_t block);  // append block `block` to the scratch buffer
    bool wait_for_block(size_t i);            // true once block i is indexed, false if the file has fewer blocks
    Piece* acquire_piece();
    void release_piece(Piece* p);

    std::string path_;
    const uint8_t* map_ = nullptr;
    size_t map_size_ = 0;
    int threads_;
    const LibraryResolver* libs_;
    RecordFilter filter_;
    std::vector<std::string> targets_;
    std::vector<uint32_t> target_len_;
    size_t first_block_coff_ = 0;             // compressed offset of the block holding the first record
    uint64_t first_rec_abs_ = 0;              // uncompressed offset of the first record, relative to that block's start

    std::mutex mu_;
    std::condition_variable cv_blocks_, cv_work_, cv_done_, cv_free_;
    std::deque<Block> blocks_;
    bool scan_done_ = false;
    size_t next_scan_ = 0;                    // compressed offset of the next BGZF member to index
    uint64_t total_ulen_ = 0;
    std::deque<Piece*> work_;                 // pieces waiting for a worker
    std::deque<Piece*> order_;                // pieces in file order, done or not (the consumer takes from the front)
    std::vector<std::unique_ptr<Piece>> pool_;
    std::vector<Piece*> free_;
    bool all_queued_ = false;
    bool stop_ = false;
    std::string scan_error_;
    std::vector<std::thread> threads_v_;
    std::thread scanner_;
    Piece* current_ = nullptr;
    Scratch redo_;                            // the consumer's own buffer for a piece it has to decode again
    uint64_t expected_abs_ = 0;
    bool started_ = false;
    bool seeked_ = false;                     // decoding starts in the middle of the file (-o with a BAM index)
    bool region_done_ = false;                // a piece ran past the region: the stream ends there
};

}  // namespace bdhost
