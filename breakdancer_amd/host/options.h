// Command line of breakdancer-max: same getopt string, defaults and usage text as the reference
// (common/Options.cpp:27-122), -C / -R (the pass-1 cache, cache.h) included.
#pragma once
#include <string>
#include <vector>

#include "bdx.h"

namespace bdhost {

struct Options {
    std::string chr;             // -o
    std::string cache_file;      // -C: write the pass-1 cache
    std::string restore_file;    // -R: run from a pass-1 cache (no other argument allowed)
    std::string bam_config_path;
    std::string prefix_fastq;    // -d
    std::string dump_BED;        // -g
    bdx_opts o;                  // numeric options in the C-ABI layout
    std::vector<std::string> orig_argv;
    int device = 0;              // env BDX_DEVICE

    // parses argv; prints usage to stderr and exits 1 like the reference when no config is given
    Options(int argc, char** argv);
    std::string sv_type(int flag) const;  // Options.cpp:105-119
};

}  // namespace bdhost
