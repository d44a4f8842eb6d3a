// bam2cfg configuration parser with the semantics of io/BamConfig.cpp:19-122 and
// io/BamConfigEntry.cpp:31-86 (tab-separated key:value fields, legacy key patterns, libraries in sorted
// name order, BAMs in sorted path order, cutoffs from mean/std when missing, initial window W0).
#pragma once
#include <istream>
#include <map>
#include <string>
#include <vector>

#include "bdx.h"

namespace bdhost {

enum Field { BAM_FILE, LIBRARY_NAME, READ_GROUP, INSERT_SIZE_MEAN, INSERT_SIZE_STDDEV, READ_LENGTH, INSERT_SIZE_UPPER_CUTOFF,
             INSERT_SIZE_LOWER_CUTOFF, MIN_MAP_QUAL, SAMPLE_NAME, UNKNOWN_FIELD };

Field translate_token(const std::string& key);

struct LibraryConfig {
    size_t index = 0;
    std::string name;
    size_t bam_file_index = 0;
    std::string bam_file;
    float mean_insertsize = 0, std_insertsize = 0, uppercutoff = 0, lowercutoff = 0, readlens = 0;
    int min_mapping_quality = -1;
};

class BamConfig {
public:
    BamConfig(std::istream& in, int cutoff_sd);
    int max_read_window_size() const { return max_read_window_size_; }
    size_t num_libs() const { return libs_.size(); }
    size_t num_bams() const { return bam_files_.size(); }
    const std::vector<std::string>& bam_files() const { return bam_files_; }
    const LibraryConfig& library_config(size_t i) const { return libs_.at(i); }
    // library index for a read-group string, with the reference's fallback to the library of the
    // alphabetically first BAM (io/BamConfig.hpp:62-72, io/AlignmentSource.hpp:57-62)
    size_t library_of_readgroup(const std::string& rg) const;
    size_t fallback_library() const { return fallback_lib_; }
    const std::map<std::string, size_t>& readgroup_index() const { return rg_lib_; }
    std::vector<bdx_lib> abi_libs() const;

private:
    std::vector<std::string> bam_files_;
    std::vector<LibraryConfig> libs_;
    std::map<std::string, size_t> rg_lib_;
    size_t fallback_lib_ = 0;
    int max_read_window_size_ = 100000000;
};

}  // namespace bdhost
