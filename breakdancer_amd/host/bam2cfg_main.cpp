// bam2cfg: the BreakDancer configuration file for a set of BAM files -- per read group: library, read length, and the
// insert-size mean / standard deviation / asymmetric cutoffs estimated from the first properly paired reads.
//
// Follows perl/bam2cfg.pl of the reference: option letters and defaults (:16-17), the per-record
// loop with its early exits (:67-146), the outlier trim, quality gates and one-sided deviations (:153-197), the output
// line (:199-262) and the Shapiro-Wilk normality figure (:306-497, Royston's AS R94 as transcribed there).  The record
// classification is AlnParser.pm:38-130 restricted to what the loop uses: "flag 18 or 20" = paired, not duplicate, both
// mates mapped to the same reference, proper-pair bit set.  Records come from this build's own BGZF/BAM reader instead
// of a `samtools view` pipe.  Output lines follow the order of the @RG header lines (the Perl script walks a hash).
// Not carried over: -h (PNG histograms through GD::Graph), -C (SOLiD orientation rules), MAQ-era Aq:i / MF:i tags.
#include <fcntl.h>
#include <getopt.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bdx.h"
#include "bam_reader.h"

namespace {

struct Opts {
    int q = 35;
    long n = 10000;
    double v = 1, c = 4, s = 50;
    bool use_mapq = false, flag_dist = false;
    int device = -1;   // --device[=N]: the records decoded and the statistics summed on GPU N
    std::string rg_lib_file;
};

double mean_of(const std::vector<double>& x) {
    double s = 0;
    for (double v : x) s += v;
    return x.empty() ? 0.0 : s / (double)x.size();
}
double sample_sd(const std::vector<double>& x, double m) {  // Statistics::Descriptive::standard_deviation (n - 1)
    if (x.size() < 2) return 0.0;
    double s = 0;
    for (double v : x) s += (v - m) * (v - m);
    return std::sqrt(s / (double)(x.size() - 1));
}

// ---- Shapiro-Wilk as in bam2cfg.pl:306-497 (AS R94) with its helper routines ppnd / alnorm / poly --------------------
double poly_(const double* c, int nord, double x) {  // :673-699
    double fn = c[0];
    if (nord == 1) return fn;
    double p = x * c[nord - 1];
    if (nord == 2) return fn + p;
    int j = nord - 2;
    for (int i = 1; i <= nord - 2; ++i) {
        p = (p + c[j]) * x;
        --j;
    }
    return fn + p;
}
double ppnd(double p) {  // :503-580
    const double split1 = 0.425, split2 = 5.0, const1 = 0.180625, const2 = 1.6;
    const double a0 = 3.3871327179, a1 = 5.0434271938 * 10, a2 = 1.5929113202 * 100, a3 = 5.9109374720 * 10;
    const double b1 = 1.7895169469 * 10, b2 = 7.8757757664 * 10, b3 = 6.7187563600 * 10;
    const double c0 = 1.4234372777, c1 = 2.7568153900, c2 = 1.3067284816, c3 = 1.7023821103e-1;
    const double d1 = 7.3700164250e-1, d2 = 1.2021132975e-1;
    const double e0 = 6.6579051150, e1 = 3.0812263860, e2 = 4.2868294337e-1, e3 = 1.7337203997e-2;
    const double f1 = 2.4197894225e-1, f2 = 1.2258202635e-2;
    const double q = p - 0.5;
    if (std::fabs(q) <= split1) {
        const double r = const1 - q * q;
        return q * (((a3 * r + a2) * r + a1) * r + a0) / (((b3 * r + b2) * r + b1) * r + 1.0);
    }
    double r = q < 0 ? p : 1.0 - p;
    if (r <= 0) return 0.0;
    r = std::sqrt(-std::log(r));
    double nd;
    if (r <= split2) {
        r -= const2;
        nd = (((c3 * r + c2) * r + c1) * r + c0) / ((d2 * r + d1) * r + 1.0);
    } else {
        r -= split2;
        nd = (((e3 * r + e2) * r + e1) * r + e0) / ((f2 * r + f1) * r + 1.0);
    }
    return q < 0 ? -nd : nd;
}
double alnorm(double x, bool upper) {  // :589-650
    const double ltone = 7.0, utzero = 18.66, con = 1.28;
    const double p = 0.398942280444, q = 0.39990348504, r = 0.398942280385;
    const double a1 = 5.75885480458, a2 = 2.62433121679, a3 = 5.92885724438, b1 = -29.8213557807, b2 = 48.6959930692;
    const double c1 = -3.8052e-8, c2 = 3.98064794e-4, c3 = -0.151679116635, c4 = 4.8385912808, c5 = 0.742380924027, c6 = 3.99019417011;
    const double d1 = 1.00000615302, d2 = 1.98615381364, d3 = 5.29330324926, d4 = -15.1508972451, d5 = 30.789933034;
    bool up = upper;
    double z = x;
    if (z < 0) { up = !up; z = -z; }
    double fn;
    if (z <= ltone || (up && z <= utzero)) {
        const double y = 0.5 * z * z;
        if (z > con) fn = r * std::exp(-y) / (z + c1 + d1 / (z + c2 + d2 / (z + c3 + d3 / (z + c4 + d4 / (z + c5 + d5 / (z + c6))))));
        else fn = 0.5 - z * (p - q * y / (y + a1 + b1 / (y + a2 + b2 / (y + a3))));
    } else {
        fn = 0.0;
    }
    return up ? fn : 1.0 - fn;
}
double shapiro_wilk(const std::vector<double>& x) {  // x sorted ascending; returns p, or the script's negative codes
    const double c1[] = {0.0, 0.221157, -0.147981, -2.07119, 4.434685, -2.706056};
    const double c2[] = {0.0, 0.042981, -0.293762, -1.752461, 5.682633, -3.582633};
    const double c3[] = {0.5440, -0.39978, 0.025054, -0.6714e-3};
    const double c4[] = {1.3822, -0.77857, 0.062767, -0.0020322};
    const double c5[] = {-1.5861, -0.31082, -0.083751, 0.0038915};
    const double c6[] = {-0.4803, -0.082676, 0.0030302};
    const double g[] = {-2.273, 0.459};
    const double sqrth = 0.70711, qtr = 0.25, th = 0.375, small = 1e-19, pi6 = 1.909859, stqr = 1.047198;
    const int n = (int)x.size();
    if (n < 3) return -1;
    const int nn2 = n / 2;
    const double an = n;
    std::vector<double> a((size_t)nn2 + 1, 0.0);
    if (n == 3) {
        a[0] = sqrth;
    } else {
        const double an25 = an + qtr;
        double summ2 = 0;
        for (int i = 1; i <= nn2; ++i) {
            a[i - 1] = ppnd((i - th) / an25);
            summ2 += a[i - 1] * a[i - 1];
        }
        summ2 *= 2.0;
        const double ssumm2 = std::sqrt(summ2), rsn = 1.0 / std::sqrt(an);
        const double a1 = poly_(c1, 6, rsn) - a[0] / ssumm2;
        int i1;
        double fac;
        if (n > 5) {
            i1 = 3;
            const double a2 = -a[1] / ssumm2 + poly_(c2, 6, rsn);
            fac = std::sqrt((summ2 - 2.0 * a[0] * a[0] - 2.0 * a[1] * a[1]) / (1.0 - 2.0 * a1 * a1 - 2.0 * a2 * a2));
            a[0] = a1;
            a[1] = a2;
        } else {
            i1 = 2;
            fac = std::sqrt((summ2 - 2.0 * a[0] * a[0]) / (1.0 - 2.0 * a1 * a1));
            a[0] = a1;
        }
        for (int i = i1; i <= nn2; ++i) a[i - 1] = -a[i - 1] / fac;
    }
    const double range = x[n - 1] - x[0];
    if (range < small) return -2.2;
    double xx = x[0] / range, sx = xx, sa = -a[0];
    auto sgn = [](int v) { return v >= 0 ? 1.0 : -1.0; };
    int j = n - 1;
    for (int i = 2; i <= n; ++i) {
        const double xi = x[i - 1] / range;
        if (xx - xi > small) return -2.3;
        sx += xi;
        if (i != j) sa += sgn(i - j) * a[std::min(i, j) - 1];
        xx = xi;
        --j;
    }
    sa /= n;
    sx /= n;
    double ssa = 0, ssx = 0, sax = 0;
    j = n;
    for (int i = 1; i <= n; ++i) {
        const double asa = i != j ? sgn(i - j) * a[std::min(i, j) - 1] - sa : -sa;
        const double xsx = x[i - 1] / range - sx;
        ssa += asa * asa;
        ssx += xsx * xsx;
        sax += asa * xsx;
        --j;
    }
    const double ssassx = std::sqrt(ssa * ssx);
    const double w1 = (ssassx - sax) * (ssassx + sax) / (ssa * ssx);
    const double w = 1.0 - w1;
    if (n == 3) return pi6 * (std::asin(std::sqrt(w)) - stqr);
    double y = std::log(w1);
    const double lx = std::log(an);
    double m, s;
    if (n <= 11) {
        const double gamma = poly_(g, 2, an);
        if (y >= gamma) return small;
        y = -std::log(gamma - y);
        m = poly_(c3, 4, an);
        s = std::exp(poly_(c4, 4, an));
    } else {
        m = poly_(c5, 4, lx);
        s = std::exp(poly_(c6, 3, lx));
    }
    return alnorm((y - m) / s, true);
}

std::string tag_of(const std::string& line, const char* tag) {  // value of "\tXX:" up to the next whitespace
    const std::string key = std::string(tag) + ":";
    size_t p = 0;
    while ((p = line.find(key, p)) != std::string::npos) {
        if (p > 0 && line[p - 1] == '\t') {
            size_t e = p + key.size();
            while (e < line.size() && !isspace((unsigned char)line[e])) ++e;
            return line.substr(p + key.size(), e - p - key.size());
        }
        p += key.size();
    }
    return "";
}

struct LibStat {
    std::vector<double> insert, readlen_unused;
    double readlen_sum = 0;
    long readlen_n = 0, libpos = 0;
    bool has_insert = false;
};


struct Rec {   // what the loop reads of a record
    int32_t tid, pos, mtid, mpos, isize, l_qseq;
    unsigned flag;
    int qual;   // -m: MAPQ, else the AM tag where there is one
};

// perl/bam2cfg.pl:40-146: the read groups of the header (and of -f), and the loop over the file's first records with its exits.
// The records come one at a time from either source (this build's reader, or the columns the GPU decoded): step()
struct Scan {
    enum { kGoOn = 0, kStop = 1, kUnsorted = 2 };
    const Opts* o;
    std::vector<std::string> rg_order;
    std::map<std::string, std::string> rg_lib, rg_platform;
    std::map<std::string, bool> libs;  // still collecting
    std::map<std::string, LibStat> st;
    std::map<std::string, std::map<int, long>> flag_hist;
    std::map<std::string, long> flag_all;
    long recordcounter = 0, expected_max = 0;
    int last_tid = -2;
    int32_t ppos = 0;

    Scan(const Opts& opts, const std::vector<std::pair<std::string, std::string>>& forced, const std::string& header_text) : o(&opts) {
        for (auto const& fl : forced) {
            if (!rg_lib.count(fl.first)) rg_order.push_back(fl.first);
            rg_lib[fl.first] = fl.second;
            libs[fl.second] = true;
        }
        std::istringstream hs(header_text);
        std::string line;
        while (std::getline(hs, line)) {
            if (line.compare(0, 3, "@RG") != 0) continue;
            const std::string id = tag_of(line, "ID"), lb = tag_of(line, "LB"), pl = tag_of(line, "PL");
            if (!rg_lib.count(id)) rg_order.push_back(id);
            libs[lb] = true;
            rg_lib[id] = lb;
            rg_platform[id] = pl;
        }
    }

    int step(const Rec& r, bool has_rg, const std::string& rg) {
        size_t nlibs_active = 0, nselected = 0;
        for (auto const& l : libs) nlibs_active += l.second ? 1 : 0;
        for (auto const& l : st) nselected += l.second.has_insert ? 1 : 0;
        if (nlibs_active == 0) {
            if (nselected > 0) return kStop;
            libs["NA"] = true;
            rg_lib["NA"] = "NA";
            rg_platform["NA"] = "illumina";
            if (std::find(rg_order.begin(), rg_order.end(), "NA") == rg_order.end()) rg_order.push_back("NA");
            nlibs_active = 1;
        }
        if (expected_max <= 0) expected_max = 3 * (long)nlibs_active * o->n;
        if (recordcounter > expected_max) return kStop;
        if (r.tid != last_tid) ppos = 0;
        last_tid = r.tid;
        if (r.pos + 1 < ppos) { fprintf(stderr, "Please sort bam by position\n"); return kUnsorted; }
        ppos = r.pos + 1;
        std::string lib;
        bool have_lib = false;
        if (has_rg) {
            auto it = rg_lib.find(rg);
            if (it != rg_lib.end()) { lib = it->second; have_lib = true; }
        } else {
            lib = "NA";
            have_lib = true;
        }
        if (!have_lib) return kGoOn;
        auto la = libs.find(lib);
        if (la == libs.end() || !la->second) return kGoOn;
        LibStat& L = st[lib];
        L.readlen_sum += (double)(r.l_qseq > 0 ? r.l_qseq : 1);  // length of the SEQ column ('*' counts 1)
        ++L.readlen_n;
        if (r.qual <= o->q) return kGoOn;
        ++recordcounter;
        ++L.libpos;
        // AlnParser.pm:57-126 for Illumina: which of the reads count as a normally oriented proper pair
        int flag = 0;
        const unsigned f = r.flag;
        if (f & 0x400) flag = 0;
        else if (f & 0x1) {
            if (f & 0x4) flag = 192;
            else if (f & 0x8) flag = 64;
            else if (r.mtid != r.tid) flag = 32;
            else if (f & 0x2) flag = (r.pos < r.mpos) == !(f & 0x10) ? 18 : 20;
            else {
                const bool rev = f & 0x10, mrev = f & 0x20;
                if (rev == mrev) flag = mrev ? 8 : 1;
                else if ((r.mpos > r.pos && rev) || (r.pos > r.mpos && !rev)) flag = 4;
                else flag = 2;
            }
        }
        if (has_rg) { ++flag_hist[rg][flag]; ++flag_all[rg]; }
        const double nreads = L.has_insert ? (double)L.insert.size() : 1.0;
        if (nreads / (double)L.libpos < 1e-4) {  // single-end lane
            libs[lib] = false;
            L.has_insert = false;
            L.insert.clear();
        }
        if (!((flag == 18 || flag == 20) && r.isize >= 0)) return kGoOn;
        L.has_insert = true;
        L.insert.push_back((double)r.isize);
        if ((long)L.insert.size() > o->n) libs[lib] = false;
        return kGoOn;
    }
};

// ---- --device: the file's first members inflated and their records decoded by the GPU (bdx_bamdec, include/bdx.h) --------------------
// The decoder's `library` column carries the read group's index in the header's order; records without an RG tag get kNoTag, records
// whose read group the header does not name kUnknownRg.  No reader filter (the script reads every line `samtools view` prints), and
// the quality column as -m asks.
constexpr uint8_t kNoTag = 254, kUnknownRg = 255;

struct DeviceRecords {
    std::vector<int32_t> tid, pos, mtid, mpos, isize;
    std::vector<uint16_t> flag, qlen;
    std::vector<uint8_t> qual, lib;
    uint64_t n = 0;
    bool whole_file = false;   // the stretch reached the end of the file
};

struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    int fd = -1;
    explicit Mapped(const std::string& path) {
        fd = open(path.c_str(), O_RDONLY);
        struct stat sb;
        if (fd < 0 || fstat(fd, &sb) != 0) {
            if (fd >= 0) close(fd);
            throw std::runtime_error("cannot open " + path);
        }
        n = (size_t)sb.st_size;
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                close(fd);
                throw std::runtime_error("cannot map " + path);
            }
            p = (const uint8_t*)m;
        }
    }
    ~Mapped() {
        if (p) munmap((void*)p, n);
        if (fd >= 0) close(fd);
    }
    Mapped(const Mapped&) = delete;
    Mapped& operator=(const Mapped&) = delete;
};

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }

struct Member { size_t off, total, payload_off, payload_len; uint32_t ulen; };
// the BGZF member at `off` (RFC 1952 header with the BC extra field, SAM specification 4.1); false at the end of the file
bool member_at(const Mapped& f, size_t off, Member& m, const std::string& path) {
    if (off >= f.n) return false;
    const uint8_t* h = f.p + off;
    if (off + 18 > f.n || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF file: " + path);
    const size_t xlen = rd16(h + 10);
    if (off + 12 + xlen > f.n) throw std::runtime_error("truncated BGZF file: " + path);
    int bsize = -1;
    for (size_t x = 12; x + 4 <= 12 + xlen;) {
        const size_t slen = rd16(h + x + 2);
        if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (int)rd16(h + x + 4);
        x += 4 + slen;
    }
    if (bsize < 0) throw std::runtime_error("BGZF block without BC field: " + path);
    m.off = off;
    m.total = (size_t)bsize + 1;
    if (m.total < 12 + xlen + 8 || off + m.total > f.n) throw std::runtime_error("truncated BGZF file: " + path);
    m.payload_off = off + 12 + xlen;
    m.payload_len = m.total - 12 - xlen - 8;
    m.ulen = rd32(h + m.total - 4);
    if (m.ulen > 65536) throw std::runtime_error("BGZF block larger than 64 KiB: " + path);
    return true;
}

// Where the first record lies: the member that holds it and its offset in that member's inflated bytes.  The header is inflated with
// zlib (a few members) and measured: magic, l_text, text, n_ref, and per reference l_name, name, l_ref (SAM specification 4.2)
void first_record(const Mapped& f, const std::string& path, size_t* member_off, uint64_t* rec_off) {
    std::vector<uint8_t> text;
    std::vector<size_t> starts, offs;   // inflated offset and file offset of every member read so far
    size_t off = 0;
    auto more = [&]() {
        Member m;
        if (!member_at(f, off, m, path)) throw std::runtime_error("truncated BAM header: " + path);
        starts.push_back(text.size());
        offs.push_back(off);
        const size_t at = text.size();
        text.resize(at + m.ulen);
        z_stream z{};
        if (inflateInit2(&z, -15) != Z_OK) throw std::runtime_error("zlib");
        z.next_in = const_cast<Bytef*>(f.p + m.payload_off);
        z.avail_in = (uInt)m.payload_len;
        z.next_out = text.data() + at;
        z.avail_out = m.ulen;
        const int rc = inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (rc != Z_STREAM_END || z.avail_out != 0) throw std::runtime_error("corrupt BGZF block in the header of " + path);
        off += m.total;
    };
    auto need = [&](size_t n) { while (text.size() < n) more(); };
    need(12);
    if (memcmp(text.data(), "BAM\1", 4) != 0) throw std::runtime_error("not a BAM file: " + path);
    size_t q = 8 + (size_t)rd32(text.data() + 4);
    need(q + 4);
    const uint32_t n_ref = rd32(text.data() + q);
    q += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        need(q + 4);
        q += 4 + (size_t)rd32(text.data() + q) + 4;
        need(q);
    }
    // (the header ends a member: the first record opens the next one)
    if (q == text.size()) { *member_off = off; *rec_off = 0; return; }
    size_t k = starts.size() - 1;
    while (starts[k] > q) --k;
    *member_off = offs[k];
    *rec_off = q - starts[k];
}

void decode_prefix(const std::string& path, int device, const std::vector<std::string>& rg_ids, int n_targets, bool mapq_only, size_t max_members,
                   DeviceRecords& out) {
    Mapped f(path);
    size_t off = 0;
    uint64_t rec_off = 0;
    first_record(f, path, &off, &rec_off);
    const size_t kPieceMembers = 1024, kPieceBytes = (size_t)4 << 20;
    std::vector<const char*> idp;
    std::vector<uint8_t> index;
    for (size_t i = 0; i < rg_ids.size(); ++i) { idp.push_back(rg_ids[i].c_str()); index.push_back((uint8_t)i); }
    bdx_bamdec_params p{};
    p.device = device;
    p.n_targets = n_targets;
    p.only_tid = -1;
    p.n_read_groups = (uint32_t)rg_ids.size();
    p.rg_ids = idp.empty() ? nullptr : idp.data();
    p.rg_lib = index.empty() ? nullptr : index.data();
    p.fallback_lib = kUnknownRg;
    p.missing_lib_plus1 = (int32_t)kNoTag + 1;
    p.record_mode = 1 | (mapq_only ? 2 : 0);
    p.first_record_offset = rec_off;
    p.batch_blocks = std::min<size_t>(std::max<size_t>(max_members, 64), 4096);
    p.ring_bytes = 4 * (p.batch_blocks + kPieceMembers + 64) * 65536;
    p.expected_bytes = std::min(f.n - std::min(f.n, off), max_members * 65536);
    p.piece_bytes = kPieceBytes + 65536;
    p.piece_blocks = kPieceMembers;
    bdx_bamdec* dec = nullptr;
    int rc = bdx_bamdec_create(&dec, nullptr, &p);
    if (rc != BDX_OK) throw std::runtime_error(std::string("bdx_bamdec_create: ") + bdx_strerror(rc));
    struct Guard { bdx_bamdec* d; ~Guard() { bdx_bamdec_destroy(d); } } guard{dec};
    auto check = [&](int r, const char* what) {
        if (r != BDX_OK) throw std::runtime_error(std::string(what) + ": " + bdx_strerror(r) + " (" + bdx_bamdec_last_error(dec) + ") in " + path);
    };
    size_t fed = 0;
    bool at_end = false;
    while (!at_end && fed < max_members) {
        // a piece: whole members, at most kPieceMembers of them and kPieceBytes
        std::vector<Member> ms;
        size_t bytes = 0;
        const size_t begin = off;
        while (ms.size() < kPieceMembers && fed + ms.size() < max_members) {
            Member m;
            if (!member_at(f, off, m, path)) { at_end = true; break; }
            if (!ms.empty() && bytes + m.total > kPieceBytes) break;
            ms.push_back(m);
            bytes += m.total;
            off += m.total;
        }
        if (!at_end && off >= f.n) at_end = true;
        void* buf = nullptr;
        bdx_bgzf_block* tab = nullptr;
        check(bdx_bamdec_acquire(dec, std::max<size_t>(bytes, 1), kPieceMembers, &buf, &tab), "bdx_bamdec_acquire");
        if (bytes) memcpy(buf, f.p + begin, bytes);
        size_t nb = 0;
        for (auto const& m : ms) {
            if (!m.ulen) continue;   // (the end-of-file marker, flush blocks)
            tab[nb].offset = m.payload_off - begin;
            tab[nb].payload_len = (uint32_t)m.payload_len;
            tab[nb].inflated_len = m.ulen;
            ++nb;
        }
        fed += ms.size();
        check(bdx_bamdec_submit(dec, bytes, nb, at_end ? 1 : 0), "bdx_bamdec_submit");
    }
    uint64_t n = 0;
    check(bdx_bamdec_finish(dec, &n), "bdx_bamdec_finish");
    out.n = n;
    out.whole_file = at_end;
    out.tid.resize(n); out.pos.resize(n); out.mtid.resize(n); out.mpos.resize(n); out.isize.resize(n);
    out.flag.resize(n); out.qlen.resize(n); out.qual.resize(n); out.lib.resize(n);
    if (n) {
        bdx_batch_buf b{};
        b.tid = out.tid.data(); b.pos = out.pos.data(); b.mtid = out.mtid.data(); b.mpos = out.mpos.data(); b.isize = out.isize.data();
        b.flag = out.flag.data(); b.qlen = out.qlen.data(); b.mapq = out.qual.data(); b.lib = out.lib.data();
        b.capacity = n;
        check(bdx_bamdec_fetch(dec, 0, n, &b), "bdx_bamdec_fetch");
    }
}

}  // namespace

int main(int argc, char** argv) {
    Opts o;
    int ch;
    static const option long_opts[] = {{"device", optional_argument, nullptr, 1000}, {nullptr, 0, nullptr, 0}};
    while ((ch = getopt_long(argc, argv, "q:n:c:b:p:s:hmf:gCv:", long_opts, nullptr)) != -1) {
        switch (ch) {
            case 1000: o.device = optarg ? atoi(optarg) : 0; break;
            case 'q': o.q = atoi(optarg); break;
            case 'n': o.n = atol(optarg); break;
            case 'c': o.c = atof(optarg); break;
            case 's': o.s = atof(optarg); break;
            case 'v': o.v = atof(optarg); break;
            case 'm': o.use_mapq = true; break;
            case 'g': o.flag_dist = true; break;
            case 'f': o.rg_lib_file = optarg; break;
            case 'b': case 'p': break;
            case 'h': fprintf(stderr, "bam2cfg: -h (histogram plots) is not supported\n"); break;
            case 'C': fprintf(stderr, "bam2cfg: -C (SOLiD) is not supported\n"); return 1;
            default: return 1;
        }
    }
    if (optind >= argc) {
        fprintf(stderr,
                "\nUsage:   bam2cfg <bam files>\nOptions:\n"
                "         -q INT    Minimum mapping quality [%d]\n"
                "         -m        Using mapping quality instead of alternative mapping quality\n"
                "         -s        Minimal mean insert size [%g]\n"
                "         -c FLOAT  Cutoff in unit of standard deviation [%g]\n"
                "         -n INT    Number of observation required to estimate mean and s.d. insert size [%ld]\n"
                "         -v FLOAT  Cutoff on coefficients of variation [%g]\n"
                "         -f STRING A two column tab-delimited text file (RG, LIB) specify the RG=>LIB mapping\n"
                "         -g        Output mapping flag distribution\n"
                "         --device[=N]  Inflate and decode the records and sum the statistics on GPU N [0]\n\n",
                o.q, o.s, o.c, o.n, o.v);
        return 1;
    }
    std::vector<std::pair<std::string, std::string>> forced;  // -f: (rg, lib) in file order
    if (!o.rg_lib_file.empty()) {
        std::ifstream f(o.rg_lib_file);
        if (!f) { fprintf(stderr, "unable to open %s\n", o.rg_lib_file.c_str()); return 1; }
        std::string rg, lib;
        while (f >> rg >> lib) forced.emplace_back(rg, lib);
    }
    try {
        for (int fi = optind; fi < argc; ++fi) {
            const std::string fbam = argv[fi];
            bdhost::BamReader rd(fbam, 4, 32);  // (the loop below ends after ~3 x libraries x n records: small batches, little inflated in vain)
            Scan sc(o, forced, rd.header_text());
            if (o.device < 0) {
                bdhost::BamRecord r;
                while (rd.next(r)) {
                    const Rec x{r.tid, r.pos, r.mtid, r.mpos, r.isize, r.l_qseq, r.flag, o.use_mapq ? (int)r.mapq : (int)r.bdqual};
                    const std::string rg = r.rg ? std::string(r.rg, r.l_rg) : std::string();
                    const int what = sc.step(x, r.rg != nullptr, rg);
                    if (what == Scan::kUnsorted) return 1;
                    if (what == Scan::kStop) break;
                }
            } else {
                // The records of the file's first members decoded on the GPU; the script's loop then walks the columns.  How many records the
                // loop wants is known only once it has seen them (reads of poor quality and of finished libraries do not count): a first
                // stretch sized for the expected number, four times as much if the loop runs off its end
                const std::vector<std::string> ids = sc.rg_order;
                if (ids.size() > 253) { fprintf(stderr, "bam2cfg --device: more than 253 read groups in %s\n", fbam.c_str()); return 1; }
                size_t members = std::max<size_t>(64, (size_t)(12.0 * (double)std::max<size_t>(1, sc.libs.size()) * (double)o.n * 256.0 / 65536.0) + 16);
                for (;;) {
                    DeviceRecords dr;
                    decode_prefix(fbam, o.device, ids, (int)rd.target_names().size(), o.use_mapq, members, dr);
                    Scan trial(o, forced, rd.header_text());
                    bool ended = false;
                    const std::string unknown("\x01");   // (a read group the header does not name: no library, the record only counts for the loop's exits)
                    for (uint64_t i = 0; i < dr.n && !ended; ++i) {
                        const Rec x{dr.tid[i], dr.pos[i], dr.mtid[i], dr.mpos[i], dr.isize[i], (int32_t)dr.qlen[i], dr.flag[i], (int)dr.qual[i]};
                        const uint8_t g = dr.lib[i];
                        const int what = trial.step(x, g != kNoTag, g < ids.size() ? ids[g] : unknown);
                        if (what == Scan::kUnsorted) return 1;
                        ended = what == Scan::kStop;
                    }
                    if (ended || dr.whole_file) { sc = std::move(trial); break; }
                    members *= 4;
                }
            }
            struct Final { std::vector<double> x; double mean, sd, stdm, stdp; };
            std::map<std::string, Final> fin;
            std::map<std::string, bdx_insert_stats> on_device;
            if (o.device >= 0) {   // perl/bam2cfg.pl:153-197's sums, one thread per library (csrc/kc_insert_stats.hip)
                std::vector<double> x;
                std::vector<uint32_t> off{0};
                std::vector<std::string> names;
                for (auto& kv : sc.st) {
                    if (!kv.second.has_insert) continue;
                    x.insert(x.end(), kv.second.insert.begin(), kv.second.insert.end());
                    off.push_back((uint32_t)x.size());
                    names.push_back(kv.first);
                }
                if (!names.empty()) {
                    std::vector<bdx_insert_stats> out(names.size());
                    const int rc = bdx_insert_size_stats(o.device, x.data(), off.data(), (int)names.size(), out.data());
                    if (rc != BDX_OK) throw std::runtime_error(std::string("bdx_insert_size_stats: ") + bdx_strerror(rc));
                    for (size_t i = 0; i < names.size(); ++i) on_device[names[i]] = out[i];
                }
            }
            for (auto& kv : sc.st) {
                LibStat& L = kv.second;
                if (!L.has_insert) continue;
                Final F;
                double mean, sd;
                if (o.device >= 0) {
                    const bdx_insert_stats& S = on_device[kv.first];
                    const double cut = S.mean_all + 5 * S.sd_all;
                    for (double x : L.insert)
                        if (!(x > cut)) F.x.push_back(x);
                    if (F.x.size() != S.n_kept) throw std::runtime_error("bdx_insert_size_stats: kept another number of observations than the host");
                    mean = S.mean;
                    sd = S.sd;
                    F.stdm = S.sd_minus;
                    F.stdp = S.sd_plus;
                } else {
                    mean = mean_of(L.insert);
                    sd = sample_sd(L.insert, mean);
                    for (double x : L.insert)
                        if (!(x > mean + 5 * sd)) F.x.push_back(x);
                    mean = mean_of(F.x);
                    sd = sample_sd(F.x, mean);
                }
                if (mean < o.s) continue;
                const double cv = sd / mean;
                if (cv >= o.v) {
                    fprintf(stderr, "Coefficient of variation %g in library %s is larger than the cutoff %g, poor quality data, excluding from further analysis.\n",
                            cv, kv.first.c_str(), o.v);
                    continue;
                }
                if (F.x.size() < 100) continue;
                if (o.device < 0) {
                    double sm = 0, sp = 0;
                    long nm = 0, np = 0;
                    for (double x : F.x) {
                        if (x > mean) { sp += (x - mean) * (x - mean); ++np; }
                        else { sm += (x - mean) * (x - mean); ++nm; }
                    }
                    F.stdm = std::sqrt(sm / (double)(nm - 1));
                    F.stdp = std::sqrt(sp / (double)(np - 1));
                }
                F.mean = mean; F.sd = sd;
                fin[kv.first] = std::move(F);
            }
            for (const std::string& rg : sc.rg_order) {
                const std::string& lib = sc.rg_lib[rg];
                auto it = fin.find(lib);
                if (it == fin.end()) continue;
                const Final& F = it->second;
                const LibStat& L = sc.st[lib];
                std::string platform = sc.rg_platform.count(rg) && !sc.rg_platform[rg].empty() ? sc.rg_platform[rg] : "illumina";
                const double readlen = L.readlen_n ? L.readlen_sum / (double)L.readlen_n : 0.0;
                double lower = F.mean - o.c * F.stdm;
                const double upper = F.mean + o.c * F.stdp;
                if (lower < 0) lower = 0;
                printf("readgroup:%s\tplatform:%s\tmap:%s\treadlen:%.2f\tlib:%s\tnum:%d", rg.c_str(), platform.c_str(), fbam.c_str(), readlen,
                       lib.c_str(), (int)F.x.size());
                printf("\tlower:%.2f\tupper:%.2f", lower, upper);
                printf("\tmean:%.2f\tstd:%.2f", F.mean, F.sd);
                std::vector<double> data = F.x;
                std::sort(data.begin(), data.end());
                const double pv = shapiro_wilk(data);
                if (pv > 0) printf("\tSWnormality:%.2f", std::log(pv) / std::log(10.0));
                else if (pv == -1) printf("\tSWnormality:data not qualified -1");
                else if (pv == -2.1) printf("\tSWnormality:data not qualified -2.1");
                else if (pv == -2.2) printf("\tSWnormality:data not qualified -2.2");
                else if (pv == -2.3) printf("\tSWnormality:data not qualified -2.3");
                else if (pv == 0) printf("\tSWnormality:minus infinity");
                if (o.flag_dist) {
                    printf("\tflag:");
                    std::vector<std::pair<std::string, long>> fs;  // the script sorts the flag codes as strings
                    for (auto const& fh : sc.flag_hist[rg]) fs.emplace_back(std::to_string(fh.first), fh.second);
                    std::sort(fs.begin(), fs.end());
                    const long all = sc.flag_all[rg];
                    for (auto const& fh : fs) printf("%s(%.2f%%)", fh.first.c_str(), (double)fh.second * 100 / (double)all);
                    printf("%ld", all);
                }
                printf("\texe:samtools view\n");
            }
        }
    } catch (std::exception const& e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
