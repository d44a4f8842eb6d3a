#include "cache.h"

#include <fstream>
#include <sstream>
#include <stdexcept>

namespace bdhost {

namespace {
const char* kMagic = "bdx-pass1-cache 1";

void put_blob(std::ostream& o, const char* tag, const std::string& s) { o << tag << " " << s.size() << "\n" << s << "\n"; }

std::string get_blob(std::istream& in, const char* tag) {
    std::string t;
    size_t n = 0;
    if (!(in >> t >> n) || t != tag) throw std::runtime_error(std::string("Failed to load restore file: expected '") + tag + "'");
    in.get();  // the newline behind the length
    std::string s(n, '\0');
    if (n && !in.read(&s[0], (std::streamsize)n)) throw std::runtime_error("Failed to load restore file: truncated");
    in.get();
    return s;
}
}  // namespace

void write_cache(const std::string& path, const Pass1Cache& c) {
    std::ofstream o(path.c_str(), std::ios::binary);
    if (!o) throw std::runtime_error("Failed to open cache file for writing");
    o << kMagic << "\n";
    o << "argc " << c.argv.size() << "\n";
    for (auto const& a : c.argv) put_blob(o, "arg", a);
    put_blob(o, "config", c.config_text);
    o << "nlibs " << c.nlibs << "\nnbams " << c.nbams << "\ncovered_ref_len " << c.covered_ref_len << "\ncounters " << c.counters.size() << "\n";
    for (size_t i = 0; i < c.counters.size(); ++i) o << c.counters[i] << (i + 1 == c.counters.size() ? "\n" : " ");
    if (!o) throw std::runtime_error("Failed to write cache file");
}

Pass1Cache read_cache(const std::string& path) {
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) throw std::runtime_error("Failed to load restore file");
    std::string line;
    if (!std::getline(in, line) || line != kMagic) throw std::runtime_error("Failed to load restore file: not a bdx pass-1 cache");
    Pass1Cache c;
    std::string t;
    size_t n = 0;
    if (!(in >> t >> n) || t != "argc") throw std::runtime_error("Failed to load restore file: expected 'argc'");
    for (size_t i = 0; i < n; ++i) c.argv.push_back(get_blob(in, "arg"));
    c.config_text = get_blob(in, "config");
    size_t nc = 0;
    std::string a, b, d, e;
    if (!(in >> a >> c.nlibs >> b >> c.nbams >> d >> c.covered_ref_len >> e >> nc) || a != "nlibs" || b != "nbams" || d != "covered_ref_len" ||
        e != "counters" || nc != (size_t)c.nlibs * 12 + (size_t)c.nbams)
        throw std::runtime_error("Failed to load restore file: bad statistics block");
    c.counters.resize(nc);
    for (size_t i = 0; i < nc; ++i)
        if (!(in >> c.counters[i])) throw std::runtime_error("Failed to load restore file: truncated statistics");
    return c;
}

}  // namespace bdhost
