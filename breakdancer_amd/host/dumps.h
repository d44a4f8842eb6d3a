// -g BED and -d FASTQ dumps of the reads supporting each printed SV (side outputs of the reference:
// breakdancer/BedWriter.cpp:21-56, BreakDancer.cpp:514-534, io/FastqWriter.cpp:22-46, io/Alignment.cpp:66-84).
#pragma once
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "config.h"
#include "producer.h"

namespace bdhost {

struct SvForDump {
    std::string chr0;      // sequence name of the SV's first breakpoint
    int pos0;              // 1-based, as printed
    std::string type;      // INS / DEL / ...
    int size;              // diffspan
    int flag;              // dominant ReadFlag
    std::vector<const SupportRead*> reads;   // SvBuilder::support_reads order
    std::vector<uint8_t> read_flags;         // ReadFlag of each read after the pass-2 remaps
};

class BedDump {
public:
    BedDump(const std::string& path, const BamConfig& cfg, const std::vector<std::string>& targets);
    void write(const SvForDump& sv);

private:
    std::ofstream out_;
    const BamConfig& cfg_;
    const std::vector<std::string>& targets_;
};

class FastqDump {
public:
    FastqDump(const std::string& prefix, const BamConfig& cfg);  // creates <prefix>.<lib>.{1,2}.fastq for every library
    void write(const SvForDump& sv);

private:
    std::ofstream& open(const std::string& lib, bool is_read1);
    std::string prefix_;
    const BamConfig& cfg_;
    std::map<std::string, std::unique_ptr<std::ofstream>> streams_;
};

}  // namespace bdhost
