#include "options.h"

#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace bdhost {

Options::Options(int argc, char** argv) : orig_argv(argv, argv + argc) {
    bdx_opts_default(&o);
    int c;
    while ((c = getopt(argc, argv, "o:s:c:m:q:r:x:b:tfd:g:lahy:C:R:")) >= 0) {
        switch (c) {
            case 'C': cache_file = optarg; break;
            case 'R':
                if (argc != 3) throw std::runtime_error("When using -R, no other options are allowed");
                restore_file = optarg;
                return;
            case 'o': chr = optarg; break;
            case 's': o.min_len = atoi(optarg); break;
            case 'c': o.cut_sd = atoi(optarg); break;
            case 'm': o.max_sd = atoi(optarg); break;
            case 'q': o.min_map_qual = atoi(optarg); break;
            case 'r': o.min_read_pair = atoi(optarg); break;
            case 'x': o.seq_coverage_lim = atoi(optarg); break;
            case 'b': o.buffer_size = atoi(optarg); break;
            case 't': o.transchr_rearrange = 1; break;
            case 'f': o.fisher = 1; break;
            case 'd': prefix_fastq = optarg; break;
            case 'g': dump_BED = optarg; break;
            case 'l': o.illumina_long_insert = 1; break;
            case 'a': o.cn_lib = 1; break;
            case 'h': o.print_af = 1; break;
            case 'y': o.score_threshold = atoi(optarg); break;
            default:
                fprintf(stderr, "Unrecognized option '-%c'.\n", c);
                exit(1);
        }
    }
    o.chr_restricted = chr.empty() ? 0 : 1;
    if (optind == argc) {
        fprintf(stderr, "\nbreakdancer-max (MI355X-native clustering path, libbdx)\n\n");
        fprintf(stderr, "Usage: breakdancer-max <analysis.config>\n\n");
        fprintf(stderr, "Options: \n");
        fprintf(stderr, "       -o STRING       operate on a single chromosome [all chromosome]\n");
        fprintf(stderr, "       -s INT          minimum length of a region [%d]\n", o.min_len);
        fprintf(stderr, "       -c INT          cutoff in unit of standard deviation [%d]\n", o.cut_sd);
        fprintf(stderr, "       -m INT          maximum SV size [%d]\n", o.max_sd);
        fprintf(stderr, "       -q INT          minimum alternative mapping quality [%d]\n", o.min_map_qual);
        fprintf(stderr, "       -r INT          minimum number of read pairs required to establish a connection [%d]\n", o.min_read_pair);
        fprintf(stderr, "       -x INT          maximum threshold of haploid sequence coverage for regions to be ignored [%d]\n", o.seq_coverage_lim);
        fprintf(stderr, "       -b INT          buffer size for building connection [%d]\n", o.buffer_size);
        fprintf(stderr, "       -t              only detect transchromosomal rearrangement, by default off\n");
        fprintf(stderr, "       -d STRING       prefix of fastq files that SV supporting reads will be saved by library\n");
        fprintf(stderr, "       -g STRING       dump SVs and supporting reads in BED format for GBrowse\n");
        fprintf(stderr, "       -l              analyze Illumina long insert (mate-pair) library\n");
        fprintf(stderr, "       -a              print out copy number and support reads per library rather than per bam, by default off\n");
        fprintf(stderr, "       -h              print out Allele Frequency column, by default off\n");
        fprintf(stderr, "       -y INT          output score filter [%d]\n", o.score_threshold);
        fprintf(stderr, "\n");
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
