// -C / -R: the pass-1 cache of the reference (io/ConfigLoader.cpp:18-66) in an own, plain-text format.
//
// The reference serialises Options, BamConfig and BamSummary with Boost.Serialization; "-R <file>" (no other argument
// allowed, common/Options.cpp:47-53) then re-runs with exactly those options, that configuration and that summary
// instead of reading the config and scanning the BAMs for pass 1.  Same contract here: the cache holds the original
// command line, the text of the configuration file and the pass-1 counters; Boost's XML is not reproduced.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bdhost {

struct Pass1Cache {
    std::vector<std::string> argv;     // the command line of the run that wrote the cache (without -C <file>)
    std::string config_text;           // contents of its configuration file
    uint32_t covered_ref_len = 0;      // BamSummary::covered_reference_length()
    std::vector<uint32_t> counters;    // [nlibs*11 flag histogram | nlibs library read counts | nbams file read counts]
    int nlibs = 0, nbams = 0;
};

void write_cache(const std::string& path, const Pass1Cache& c);  // throws std::runtime_error
Pass1Cache read_cache(const std::string& path);                  // throws std::runtime_error

}  // namespace bdhost
