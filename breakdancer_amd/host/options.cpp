#include "options.h"

#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace bdhost {

namespace {

// One row per command-line option: where its value goes and what the usage text says about it.  The getopt string and the
// usage block are generated from this table (letters, argument kinds, defaults and wording as the reference documents them,
// common/Options.cpp:27-122; -C / -R are accepted without being listed, as there).
enum Kind { kInt, kFlag, kString };
struct Row {
    char letter;
    Kind kind;
    int32_t bdx_opts::*num;        // kInt / kFlag
    std::string Options::*str;     // kString
    const char* help;              // nullptr: not listed in the usage text
    bool show_default;
};
const Row kRows[] = {
    {'o', kString, nullptr, &Options::chr, "operate on a single chromosome [all chromosome]", false},
    {'s', kInt, &bdx_opts::min_len, nullptr, "minimum length of a region", true},
    {'c', kInt, &bdx_opts::cut_sd, nullptr, "cutoff in unit of standard deviation", true},
    {'m', kInt, &bdx_opts::max_sd, nullptr, "maximum SV size", true},
    {'q', kInt, &bdx_opts::min_map_qual, nullptr, "minimum alternative mapping quality", true},
    {'r', kInt, &bdx_opts::min_read_pair, nullptr, "minimum number of read pairs required to establish a connection", true},
    {'x', kInt, &bdx_opts::seq_coverage_lim, nullptr, "maximum threshold of haploid sequence coverage for regions to be ignored", true},
    {'b', kInt, &bdx_opts::buffer_size, nullptr, "buffer size for building connection", true},
    {'t', kFlag, &bdx_opts::transchr_rearrange, nullptr, "only detect transchromosomal rearrangement, by default off", false},
    {'f', kFlag, &bdx_opts::fisher, nullptr, nullptr, false},
    {'d', kString, nullptr, &Options::prefix_fastq, "prefix of fastq files that SV supporting reads will be saved by library", false},
    {'g', kString, nullptr, &Options::dump_BED, "dump SVs and supporting reads in BED format for GBrowse", false},
    {'l', kFlag, &bdx_opts::illumina_long_insert, nullptr, "analyze Illumina long insert (mate-pair) library", false},
    {'a', kFlag, &bdx_opts::cn_lib, nullptr, "print out copy number and support reads per library rather than per bam, by default off", false},
    {'h', kFlag, &bdx_opts::print_af, nullptr, "print out Allele Frequency column, by default off", false},
    {'y', kInt, &bdx_opts::score_threshold, nullptr, "output score filter", true},
    {'C', kString, nullptr, &Options::cache_file, nullptr, false},
    {'R', kString, nullptr, &Options::restore_file, nullptr, false},
};

std::string getopt_string() {
    std::string g;
    for (const Row& r : kRows) {
        g += r.letter;
        if (r.kind != kFlag) g += ':';
    }
    return g;
}

void print_usage(const bdx_opts& o) {
    fprintf(stderr, "\nbreakdancer-max (MI355X-native clustering path, libbdx)\n\nUsage: breakdancer-max <analysis.config>\n\nOptions: \n");
    for (const Row& r : kRows) {
        if (!r.help) continue;
        const char* arg = r.kind == kInt ? "INT   " : r.kind == kString ? "STRING" : "      ";
        if (r.show_default) fprintf(stderr, "       -%c %s       %s [%d]\n", r.letter, arg, r.help, o.*(r.num));
        else fprintf(stderr, "       -%c %s       %s\n", r.letter, arg, r.help);
    }
    fprintf(stderr, "\n");
}

}  // namespace

Options::Options(int argc, char** argv) : orig_argv(argv, argv + argc) {
    bdx_opts_default(&o);
    const std::string spec = getopt_string();
    int c;
    while ((c = getopt(argc, argv, spec.c_str())) >= 0) {
        const Row* row = nullptr;
        for (const Row& r : kRows)
            if (r.letter == c) row = &r;
        if (!row) {
            fprintf(stderr, "Unrecognized option '-%c'.\n", c);
            exit(1);
        }
        switch (row->kind) {
            case kInt: o.*(row->num) = atoi(optarg); break;
            case kFlag: o.*(row->num) = 1; break;
            case kString: this->*(row->str) = optarg; break;
        }
        if (c == 'R') {  // a run from a pass-1 cache takes its options from the cache (ConfigLoader.cpp:19-23)
            if (argc != 3) throw std::runtime_error("When using -R, no other options are allowed");
            return;
        }
    }
    o.chr_restricted = chr.empty() ? 0 : 1;
    if (optind == argc) {
        print_usage(o);
        exit(1);
    }
    bam_config_path = argv[optind];
    if (const char* d = getenv("BDX_DEVICE")) device = atoi(d);
}

std::string Options::sv_type(int flag) const {
    if (o.illumina_long_insert) {
        switch (flag) {
            case BDX_ARP_FF: return "INV";
            case BDX_ARP_SMALL_INSERT: return "INS";
            case BDX_ARP_RF: return "DEL";
            case BDX_ARP_RR: return "INV";
            case BDX_ARP_CTX: return "CTX";
            default: return "";
        }
    }
    switch (flag) {
        case BDX_ARP_FF: return "INV";
        case BDX_ARP_LARGE_INSERT: return "DEL";
        case BDX_ARP_SMALL_INSERT: return "INS";
        case BDX_ARP_RF: return "ITX";
        case BDX_ARP_RR: return "INV";
        case BDX_ARP_CTX: return "CTX";
        default: return "";
    }
}

}  // namespace bdhost
