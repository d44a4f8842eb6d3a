#include "config.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <sstream>
#include <stdexcept>

namespace bdhost {

namespace {

bool is_word(char c) { return isalnum((unsigned char)c) || c == '_'; }

bool ieq_at(const std::string& s, size_t p, const char* lit) {
    for (size_t i = 0; lit[i]; ++i) {
        if (p + i >= s.size()) return false;
        if (tolower((unsigned char)s[p + i]) != lit[i]) return false;
    }
    return true;
}
bool all_word(const std::string& s, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i)
        if (!is_word(s[i])) return false;
    return true;
}
// unanchored, case-insensitive "<lit>$"
bool ends_with(const std::string& s, const char* lit) {
    const size_t n = std::char_traits<char>::length(lit);
    return s.size() >= n && ieq_at(s, s.size() - n, lit);
}
// unanchored, case-insensitive "<lit>\w*$"
bool lit_then_words(const std::string& s, const char* lit) {
    const size_t n = std::char_traits<char>::length(lit);
    for (size_t p = 0; p + n <= s.size(); ++p)
        if (ieq_at(s, p, lit) && all_word(s, p + n, s.size())) return true;
    return false;
}
// "map\w*qual\w*$"
bool map_qual(const std::string& s) {
    for (size_t p = 0; p + 3 <= s.size(); ++p) {
        if (!ieq_at(s, p, "map")) continue;
        for (size_t q = p + 3; q + 4 <= s.size(); ++q) {
            if (!all_word(s, p + 3, q)) break;
            if (ieq_at(s, q, "qual") && all_word(s, q + 4, s.size())) return true;
        }
    }
    return false;
}

float to_float(const std::string& s) {
    char* end = nullptr;
    const float v = strtof(s.c_str(), &end);
    if (s.empty() || *end != 0) throw std::runtime_error("bad lexical cast: source type value could not be interpreted as target");
    return v;
}
int to_int(const std::string& s) {
    char* end = nullptr;
    const long v = strtol(s.c_str(), &end, 10);
    if (s.empty() || *end != 0) throw std::runtime_error("bad lexical cast: source type value could not be interpreted as target");
    return (int)v;
}

}  // namespace

// The reference keeps (pattern -> field) in a flat_map ordered by the pattern text and takes the first
// regex_search hit (BamConfigEntry.cpp:43-65); this is that order.
Field translate_token(const std::string& k) {
    if (ends_with(k, "group")) return READ_GROUP;                    // group$
    if (lit_then_words(k, "lib")) return LIBRARY_NAME;               // lib\w*$
    if (lit_then_words(k, "low")) return INSERT_SIZE_LOWER_CUTOFF;   // low\w*$
    if (ends_with(k, "map")) return BAM_FILE;                        // map$
    if (map_qual(k)) return MIN_MAP_QUAL;                            // map\w*qual\w*$
    if (lit_then_words(k, "mean")) return INSERT_SIZE_MEAN;          // mean\w*$
    if (lit_then_words(k, "readlen")) return READ_LENGTH;            // readlen\w*$
    if (lit_then_words(k, "samp")) return SAMPLE_NAME;               // samp\w*$
    if (lit_then_words(k, "std")) return INSERT_SIZE_STDDEV;         // std\w*$
    if (lit_then_words(k, "upp")) return INSERT_SIZE_UPPER_CUTOFF;   // upp\w*$
    return UNKNOWN_FIELD;
}

BamConfig::BamConfig(std::istream& in, int cutoff_sd) {
    std::map<std::string, LibraryConfig> tmp;
    std::map<std::string, std::string> readgroup_library, bam_library;
    std::string line;
    size_t line_num = 0;
    while (std::getline(in, line)) {
        ++line_num;
        if (line.empty()) break;
        std::map<Field, std::string> dir;
        size_t b = 0;
        while (true) {
            const size_t e = line.find('\t', b);
            const std::string f = line.substr(b, e == std::string::npos ? std::string::npos : e - b);
            const size_t colon = f.find(':');
            if (colon != std::string::npos) {
                const Field fn = translate_token(f.substr(0, colon));
                if (fn != UNKNOWN_FIELD) dir[fn] = f.substr(colon + 1);
            }
            if (e == std::string::npos) break;
            b = e + 1;
        }
        auto has = [&](Field f) { return dir.find(f) != dir.end(); };
        std::string fmap, lib, readgroup;
        float mean = 0, stddev = 0, readlen = 0, upper = 0, lower = 0;
        int mqual = -1;
        if (has(LIBRARY_NAME)) lib = dir[LIBRARY_NAME];
        else if (has(SAMPLE_NAME)) lib = dir[SAMPLE_NAME];
        if (!has(BAM_FILE)) {
            std::ostringstream m;
            m << "Required field 'map' not found in config at line " << line_num << "!";
            throw std::runtime_error(m.str());
        }
        fmap = dir[BAM_FILE];
        readgroup = has(READ_GROUP) ? dir[READ_GROUP] : lib;
        readgroup_library[readgroup] = lib;
        bam_library[fmap] = lib;
        if (has(READ_LENGTH)) readlen = to_float(dir[READ_LENGTH]);
        if (has(MIN_MAP_QUAL)) mqual = to_int(dir[MIN_MAP_QUAL]);
        const bool have_mean = has(INSERT_SIZE_MEAN), have_std = has(INSERT_SIZE_STDDEV);
        const bool have_lower = has(INSERT_SIZE_LOWER_CUTOFF), have_upper = has(INSERT_SIZE_UPPER_CUTOFF);
        if (have_mean) mean = to_float(dir[INSERT_SIZE_MEAN]);
        if (have_std) stddev = to_float(dir[INSERT_SIZE_STDDEV]);
        if (have_lower) lower = to_float(dir[INSERT_SIZE_LOWER_CUTOFF]);
        if (have_upper) upper = to_float(dir[INSERT_SIZE_UPPER_CUTOFF]);
        if (have_mean && have_std && (!have_upper || !have_lower)) {
            upper = mean + stddev * cutoff_sd;
            lower = mean - stddev * cutoff_sd;
            lower = lower > 0 ? lower : 0;
        }
        LibraryConfig lc;
        lc.name = lib; lc.bam_file = fmap; lc.min_mapping_quality = mqual;
        lc.mean_insertsize = mean; lc.std_insertsize = stddev; lc.uppercutoff = upper; lc.lowercutoff = lower; lc.readlens = readlen;
        auto ins = tmp.insert(std::make_pair(lib, lc));
        if (!ins.second) {
            const LibraryConfig& p = ins.first->second;
            const bool same = p.bam_file == lc.bam_file && p.mean_insertsize == lc.mean_insertsize && p.std_insertsize == lc.std_insertsize &&
                              p.uppercutoff == lc.uppercutoff && p.lowercutoff == lc.lowercutoff && p.readlens == lc.readlens &&
                              p.min_mapping_quality == lc.min_mapping_quality;
            if (!same) {
                fprintf(stderr, "WARNING: at line %zu, library %s overwritten!\n", line_num, lib.c_str());
                ins.first->second = lc;
            }
        }
        const int t = mean - readlen * 2;
        max_read_window_size_ = std::min(max_read_window_size_, t);
    }
    for (auto& kv : tmp) {
        kv.second.index = libs_.size();
        libs_.push_back(kv.second);
    }
    for (auto const& kv : bam_library) bam_files_.push_back(kv.first);
    std::map<std::string, size_t> lib_index;
    for (auto& l : libs_) {
        lib_index[l.name] = l.index;
        auto it = std::find(bam_files_.begin(), bam_files_.end(), l.bam_file);
        if (it == bam_files_.end())
            throw std::runtime_error("Bam file '" + l.bam_file + "' referenced by library '" + l.name + "' but not found in bam list!");
        l.bam_file_index = it - bam_files_.begin();
    }
    for (auto const& kv : readgroup_library) rg_lib_[kv.first] = lib_index.at(kv.second);
    if (!bam_library.empty()) fallback_lib_ = lib_index.at(bam_library.begin()->second);
    max_read_window_size_ = std::max(max_read_window_size_, 50);
}

size_t BamConfig::library_of_readgroup(const std::string& rg) const {
    auto it = rg_lib_.find(rg);
    return it != rg_lib_.end() ? it->second : fallback_lib_;
}

std::vector<bdx_lib> BamConfig::abi_libs() const {
    std::vector<bdx_lib> v(libs_.size());
    for (size_t i = 0; i < libs_.size(); ++i) {
        const LibraryConfig& l = libs_[i];
        v[i] = bdx_lib{l.mean_insertsize, l.std_insertsize, l.uppercutoff, l.lowercutoff, l.readlens, l.min_mapping_quality,
                       (int32_t)l.bam_file_index};
    }
    return v;
}

}  // namespace bdhost
