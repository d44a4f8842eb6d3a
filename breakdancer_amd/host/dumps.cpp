#include "dumps.h"

#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace bdhost {

BedDump::BedDump(const std::string& path, const BamConfig& cfg, const std::vector<std::string>& targets)
    : out_(path.c_str()), cfg_(cfg), targets_(targets) {}

void BedDump::write(const SvForDump& sv) {
    std::stringstream trackname;
    trackname << sv.chr0 << "_" << sv.pos0 << "_" << sv.type << "_" << sv.size;
    out_ << "track name=" << trackname.str() << "\tdescription=\"BreakDancer" << " " << sv.chr0 << " " << sv.pos0 << " " << sv.type
         << " " << sv.size << "\"\tuseScore=0\n";
    for (size_t i = 0; i < sv.reads.size(); ++i) {
        const SupportRead& y = *sv.reads[i];
        if (y.bases.empty() || y.l_qseq <= 0 || sv.read_flags[i] != sv.flag) continue;  // has_sequence() && bdflag == sv.flag
        const int aln_end = y.pos + y.l_qseq;
        const char* color = y.rev ? "255,0,0" : "0,0,255";
        if (strncmp("chr", sv.chr0.c_str(), 3) != 0) out_ << "chr";
        const std::string tname = (y.tid >= 0 && (size_t)y.tid < targets_.size()) ? targets_[y.tid] : std::to_string(y.tid);
        out_ << tname << "\t" << y.pos << "\t" << aln_end << "\t" << y.name << "|" << cfg_.library_config(y.lib).name << "\t"
             << y.bdqual * 10 << "\t" << (y.rev ? 1 : 0) << "\t" << y.pos << "\t" << aln_end << "\t" << color << "\n";
    }
}

FastqDump::FastqDump(const std::string& prefix, const BamConfig& cfg) : prefix_(prefix), cfg_(cfg) {
    for (size_t i = 0; i < cfg.num_libs(); ++i) {  // BreakDancer.cpp:112-119: both files of every library exist afterwards
        open(cfg.library_config(i).name, true);
        open(cfg.library_config(i).name, false);
    }
}

std::ofstream& FastqDump::open(const std::string& lib, bool is_read1) {
    const std::string path = prefix_ + "." + lib + "." + (is_read1 ? "1" : "2") + ".fastq";
    auto it = streams_.find(path);
    if (it == streams_.end())
        it = streams_.emplace(path, std::unique_ptr<std::ofstream>(new std::ofstream(path.c_str(), std::ofstream::app))).first;
    if (!*it->second) throw std::runtime_error("Failed to open fastq file '" + path + "' for writing");
    return *it->second;
}

void FastqDump::write(const SvForDump& sv) {
    std::map<std::string, int> pairing;
    for (size_t i = 0; i < sv.reads.size(); ++i) {
        const SupportRead& y = *sv.reads[i];
        if (y.bases.empty() || y.l_qseq <= 0 || sv.read_flags[i] != sv.flag) continue;
        // "Paradoxically, the first read seen is put in file 2 and the second in file 1" (BreakDancer.cpp:526-527)
        const bool is_read1 = pairing.count(y.name) != 0;
        std::ofstream& s = open(cfg_.library_config(y.lib).name, is_read1);
        s << "@" << y.name << "\n" << y.bases << "\n+\n";
        if (y.has_qual) {
            for (int q = 0; q < y.l_qseq; ++q) s << char((unsigned char)y.qual[q] + 33);
        } else {
            std::cerr << "Warning: no quality data for read " << y.name << "\n";
        }
        s << "\n";
        pairing[y.name] = 1;
    }
}

}  // namespace bdhost
