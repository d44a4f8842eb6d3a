// bdx-dump-reads: prints the producer's merged SoA stream (no GPU involved).  Used by the CPU tests to check the
// BGZF/BAM decoder, the reader filter, RG->library resolution and the merge order against an independent decode.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "config.h"
#include "options.h"
#include "producer.h"

int main(int argc, char** argv) {
    try {
        bdhost::Options opts(argc, argv);
        std::ifstream in(opts.bam_config_path.c_str());
        if (!in.is_open()) throw std::runtime_error("unable to open config file '" + opts.bam_config_path + "'");
        bdhost::BamConfig cfg(in, opts.o.cut_sd);
        bdhost::ReadStream rs;
        if (getenv("BDX_DUMP_MERGE")) bdhost::produce_merged_by_columns(cfg, opts.chr, 4, rs);   // (the device path's merge, on host-decoded columns)
        else bdhost::produce(cfg, opts.chr, 4, rs);
        printf("#w0=%d nlibs=%zu nbams=%zu n=%zu\n", cfg.max_read_window_size(), cfg.num_libs(), cfg.num_bams(), rs.size());
        for (size_t i = 0; i < cfg.num_libs(); ++i) {
            const bdhost::LibraryConfig& l = cfg.library_config(i);
            printf("#lib\t%zu\t%s\t%s\t%zu\t%.9g\t%.9g\t%.9g\t%.9g\t%.9g\t%d\n", i, l.name.c_str(), l.bam_file.c_str(), l.bam_file_index,
                   l.mean_insertsize, l.std_insertsize, l.uppercutoff, l.lowercutoff, l.readlens, l.min_mapping_quality);
        }
        for (size_t i = 0; i < rs.size(); ++i)
            printf("%d\t%d\t%d\t%d\t%d\t%u\t%u\t%u\t%u\t%u\t%llu\t%llu\n", rs.tid[i], rs.pos[i], rs.mtid[i], rs.mpos[i], rs.isize[i],
                   (unsigned)rs.flag[i], (unsigned)rs.qlen[i], (unsigned)rs.mapq[i], (unsigned)rs.lib[i], (unsigned)rs.bam[i],
                   (unsigned long long)rs.name_key[i], (unsigned long long)rs.name_check[i]);
    } catch (std::exception const& e) {
        std::cerr << "ERROR: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
