// breakdancer-max: same command line, configuration format and output columns as the reference
// (exe/breakdancer-max/BreakDancerMax.cpp:38-163); the clustering path runs on the GPU through libbdx.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <fstream>
#include <future>
#include <iomanip>
#include <iostream>
#include <algorithm>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <time.h>
#include <cerrno>
#include <csignal>
#include <fcntl.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#include <getopt.h>
#include <cstring>

#include "bdx.h"
#include "cache.h"
#include "config.h"
#include "dumps.h"
#include "options.h"
#include "producer.h"

using namespace bdhost;

namespace {

const int kPrintedFlagValue[BDX_NUM_FLAGS] = {0, 1, 2, 3, 4, 8, 18, 20, 32, 64, 192};  // common/ReadFlags.cpp:4-14

void check(bdx_ctx* ctx, int rc, const char* what) {
    if (rc == BDX_OK) return;
    std::string msg = std::string(what) + ": " + bdx_strerror(rc);
    if (ctx && bdx_last_error(ctx)[0]) msg += std::string(" (") + bdx_last_error(ctx) + ")";
    throw std::runtime_error(msg);
}

// BDX_GPUS: "4" -> devices 0,1,2,3; "0,2,5" -> that list (a device may repeat: several ranks then share it)
std::vector<int> gpu_list() {
    std::vector<int> v;
    const char* e = getenv("BDX_GPUS");
    if (!e || !*e) return v;
    const std::string s(e);
    if (s.find(',') == std::string::npos) {
        for (int i = 0; i < atoi(e); ++i) v.push_back(i);
        return v;
    }
    size_t p = 0;
    while (p <= s.size()) {
        const size_t q = s.find(',', p);
        v.push_back(atoi(s.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    return v;
}

// Sharded runs: the merged stream is position sorted, so every chromosome is one run of records; each run goes into the
// staging ring of the context that owns the chromosome (bdx_dist_chromosome of its rank).
struct RoutingSink : BatchSink {
    std::vector<bdx_dist*>& ranks;
    const std::vector<int>& rank_of;
    std::vector<char> store;
    bdx_batch_buf cur{};
    RoutingSink(std::vector<bdx_dist*>& r, const std::vector<int>& ro) : ranks(r), rank_of(ro) {}
    bdx_batch_buf acquire(size_t capacity) override {
        const size_t K = (capacity + 63) / 64 * 64;
        store.resize(K * 43 + 64);
        char* p = store.data();
        cur.name_key = (uint64_t*)p; p += K * 8;
        cur.name_check = (uint64_t*)p; p += K * 8;
        cur.tid = (int32_t*)p; p += K * 4; cur.pos = (int32_t*)p; p += K * 4; cur.mtid = (int32_t*)p; p += K * 4;
        cur.mpos = (int32_t*)p; p += K * 4; cur.isize = (int32_t*)p; p += K * 4;
        cur.flag = (uint16_t*)p; p += K * 2; cur.qlen = (uint16_t*)p; p += K * 2;
        cur.mapq = (uint8_t*)p; p += K; cur.lib = (uint8_t*)p; p += K; cur.bam = (uint8_t*)p;
        cur.capacity = K;
        return cur;
    }
    void submit(size_t n) override {
        size_t lo = 0;
        while (lo < n) {
            const int tid = cur.tid[lo];
            size_t hi = lo + 1;
            while (hi < n && cur.tid[hi] == tid) ++hi;
            if (tid < 0 || (size_t)tid >= rank_of.size()) throw std::runtime_error("record with a reference id beyond the header's sequences");
            bdx_ctx* c = bdx_dist_chromosome(ranks[rank_of[tid]], tid);
            if (!c) throw std::runtime_error("bdx_dist_chromosome failed");
            check(c, bdx_use_name_check(c, 1), "bdx_use_name_check");
            const size_t m = hi - lo;
            bdx_batch_buf b{};
            check(c, bdx_acquire_batch(c, m, &b), "bdx_acquire_batch");
            memcpy(b.tid, cur.tid + lo, m * 4); memcpy(b.pos, cur.pos + lo, m * 4); memcpy(b.mtid, cur.mtid + lo, m * 4);
            memcpy(b.mpos, cur.mpos + lo, m * 4); memcpy(b.isize, cur.isize + lo, m * 4);
            memcpy(b.flag, cur.flag + lo, m * 2); memcpy(b.qlen, cur.qlen + lo, m * 2);
            memcpy(b.mapq, cur.mapq + lo, m); memcpy(b.lib, cur.lib + lo, m); memcpy(b.bam, cur.bam + lo, m);
            memcpy(b.name_key, cur.name_key + lo, m * 8);
            memcpy(b.name_check, cur.name_check + lo, m * 8);
            check(c, bdx_submit_batch(c, m), "bdx_submit_batch");
            lo = hi;
        }
    }
};

// CPUs this process can use: hardware threads, capped by the cgroup v2 / v1 CPU quota if one is set
unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64] = {0};
        long period = 0;
        if (fscanf(f, "%63s %ld", quota, &period) == 2 && period > 0 && quota[0] != 'm')
            n = std::min<unsigned>(n, (unsigned)std::max(1L, (atol(quota) + period - 1) / period));
        fclose(f);
    } else {
        long q = -1, per = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &q) != 1) q = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &per) != 1) per = 0; fclose(g); }
        if (q > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, (q + per - 1) / per));
    }
    return n;
}

// The producer's batches go straight into the context's pinned staging ring: while the decode threads work on the next
// pieces of the BAMs, the previous batch crosses PCIe and the classifier runs over it (AlignmentSource.hpp:48-65: one
// record stream, consumed as it is produced).  The context is created on a helper thread (the HIP runtime takes ~0.1 s
// to come up) and only waited for when the first batch is ready.
struct GpuSink : BatchSink {
    std::future<int>& ready;
    bdx_ctx*& ctx;
    size_t reserve;
    bool up = false;
    GpuSink(std::future<int>& r, bdx_ctx*& c, size_t res) : ready(r), ctx(c), reserve(res) {}
    void bring_up(bool with_store = true) {
        if (up) return;
        if (ready.valid()) check(nullptr, ready.get(), "bdx_create");
        if (!with_store) return;   // (the device-side decoder sizes the store itself, and the later stages once it knows the record density)
        check(ctx, bdx_reserve(ctx, reserve), "bdx_reserve");
        up = true;
    }
    bdx_batch_buf acquire(size_t capacity) override {
        bring_up();
        bdx_batch_buf b{};
        check(ctx, bdx_acquire_batch(ctx, capacity, &b), "bdx_acquire_batch");
        return b;
    }
    void submit(size_t n) override { check(ctx, bdx_submit_batch(ctx, n), "bdx_submit_batch"); }
};

// The command's result is complete (everything printed, the dump files closed): tell the process that was started, which returns
// at once; this one -- which owns the GPU context -- goes on to hand gigabytes of HBM and the pinned buffers back to the driver
// (~0.1 s for a chromosome), with its standard streams closed so that a reader of the pipe sees the end of the output now.
int g_report_fd = -1;
void report_result(int status) {
    std::cout.flush();
    fflush(nullptr);
    if (g_report_fd < 0) return;
    const ssize_t w = write(g_report_fd, &status, sizeof status);
    (void)w;
    close(g_report_fd);
    g_report_fd = -1;
    close(STDOUT_FILENO);
    close(STDERR_FILENO);
}

int run(int argc, char** argv);

}  // namespace

// Releasing a GPU context takes the driver longer than the whole GPU path runs, and a process cannot return before its resources
// are gone.  So the work is done by a child (forked before the HIP runtime is touched); the process the user started waits for the
// child's word that the table is written and returns with its status, while the child's exit takes its time in the background.
// BDX_FOREGROUND=1 (and BDX_CLEAN_EXIT=1, the leak checkers' mode) keep everything in the one process.
int main(int argc, char** argv) {
    if (getenv("BDX_TIMING")) {   // (for whoever times the process from outside: when main() was reached, on the wall clock)
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "[bdx timing] main() entered at %.6f (wall clock)\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
    }
    if (!getenv("BDX_FOREGROUND") && !getenv("BDX_CLEAN_EXIT")) {
        int pfd[2];
        if (pipe(pfd) == 0) {
            const pid_t parent = getpid();   // (may be 1: a container's entry point)
            const pid_t pid = fork();
            if (pid > 0) {
                close(pfd[1]);
                int status = 1;
                ssize_t r;
                do r = read(pfd[0], &status, sizeof status); while (r < 0 && errno == EINTR);
                if (r == (ssize_t)sizeof status) _exit(status);
                int ws = 0;   // the child ended without reporting: usage errors (exit inside the option parser), crashes
                while (waitpid(pid, &ws, 0) < 0 && errno == EINTR) {}
                if (WIFEXITED(ws)) _exit(WEXITSTATUS(ws));
                if (WIFSIGNALED(ws)) { signal(WTERMSIG(ws), SIG_DFL); raise(WTERMSIG(ws)); }
                _exit(1);
            }
            if (pid == 0) {
                close(pfd[0]);
                g_report_fd = pfd[1];
                fcntl(g_report_fd, F_SETFD, FD_CLOEXEC);
                // (killing the command kills the work: the child is told when the process that was started goes -- after the report
                // that only cuts its exit short)
                prctl(PR_SET_PDEATHSIG, SIGTERM);
                if (getppid() != parent) _exit(1);   // (the parent went before the signal was armed)
            } else {  // (no child: everything in this process)
                close(pfd[0]);
                close(pfd[1]);
            }
        }
    }
    const int rc = run(argc, argv);
    report_result(rc);
    return rc;
}

namespace {

int run(int argc, char** argv) {
    bdx_ctx* ctx = nullptr;
    bool ctx_owned = true;  // false: ctx is a sharded run's result context, which belongs to rank 0's bdx_dist
    std::thread trimmer;    // hands the pinned result tables back while the table is printed; joined before anything destroys the context
    try {
        std::unique_ptr<Options> opts_p(new Options(argc, argv));
        Pass1Cache cache;
        const bool restored = !opts_p->restore_file.empty();
        std::string config_text;
        if (restored) {
            // -R <cache>: options, configuration and pass-1 statistics of the run that wrote the cache (ConfigLoader.cpp:19-23)
            cache = read_cache(opts_p->restore_file);
            std::vector<char*> av;
            for (auto& a : cache.argv) av.push_back(&a[0]);
            optind = 1;
            opts_p.reset(new Options((int)av.size(), av.data()));
            config_text = cache.config_text;
        } else {
            std::ifstream cfg_stream(opts_p->bam_config_path.c_str());
            if (!cfg_stream.is_open()) throw std::runtime_error("unable to open config file '" + opts_p->bam_config_path + "'");
            std::stringstream ss;
            ss << cfg_stream.rdbuf();
            config_text = ss.str();
        }
        Options& opts = *opts_p;
        const bool want_dumps = !opts.prefix_fastq.empty() || !opts.dump_BED.empty();
        std::istringstream cfg_stream(config_text);
        BamConfig cfg(cfg_stream, opts.o.cut_sd);
        if (restored && ((int)cfg.num_libs() != cache.nlibs || (int)cfg.num_bams() != cache.nbams))
            throw std::runtime_error("Failed to load restore file: statistics do not fit its configuration");
        if (cfg.num_bams() == 0) {
            std::cout << "Error: no bams files in config file!\n";
            return 1;
        }
        // Decode threads: the work is CPU-bound on zlib (~375 MB/s of inflated bytes per core), so what counts is the number of
        // cores this process may really use -- the cgroup's CPU quota when there is one (a container with "16 CPUs" on a
        // 256-thread host: 16 threads 0.64 s, 32 0.55 s, 64 0.70 s for 15 M records), else the hardware threads.  Twice that
        // many threads keep the cores busy across their waits; BDX_THREADS overrides.
        const unsigned io_threads = getenv("BDX_THREADS") ? (unsigned)std::max(1, atoi(getenv("BDX_THREADS")))
                                                          : std::min(std::max(2 * usable_cpus(), 4u), 64u);
        std::vector<std::string> targets;
        const bool timing = getenv("BDX_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double>(b - a).count();
        };
        const auto t_start = now();
        // the GPU context (HIP runtime start-up, ~0.1 s) comes up on a helper thread while the BAMs are decoded
        const std::vector<bdx_lib> libs = cfg.abi_libs();
        const int nlibs = (int)libs.size(), nbams = (int)cfg.num_bams();
        // BDX_GPUS=N (devices 0..N-1) or a list "0,1,2,3": ONE whole-genome run with the chromosomes spread over several
        // GPUs (bdx_dist_*: one rank per device, here as threads of this process) -- same output as on one GPU
        const std::vector<int> devices = gpu_list();
        const bool sharded = devices.size() > 1 && opts.chr.empty();
        std::vector<bdx_dist*> ranks(sharded ? devices.size() : 0, nullptr);
        struct RanksGuard {
            std::vector<bdx_dist*>& r;
            ~RanksGuard() { for (bdx_dist* d : r) if (d) bdx_dist_destroy(d); }
        } ranks_guard{ranks};
        std::future<int> ctx_ready;
        if (!sharded)
            ctx_ready = std::async(std::launch::async, [&] {
                int rc = bdx_create(&ctx, &opts.o, libs.data(), nlibs, nbams, 0, cfg.max_read_window_size(), opts.device);
                if (rc == BDX_OK) rc = bdx_use_name_check(ctx, 1);   // mates are joined on two hashes of the read name
                return rc;
            });
        // resident store sized from the compressed files (a record takes 50-150 bytes of BAM; a store that is too small
        // grows, at the price of classifying from the first tile again)
        size_t reserve = 1 << 20;
        for (auto const& f : cfg.bam_files()) {
            struct stat st;
            if (stat(f.c_str(), &st) == 0) reserve += (size_t)st.st_size / 48;
        }
        reserve = std::min<size_t>(reserve, 0xFFFFFFFFull - 1024);
        size_t n_reads = 0;
        bool device_decoded = false, sharded_on_device = false;
        auto t_decoded = now();
        if (sharded) {
            if (restored) throw std::runtime_error("-R is not supported with BDX_GPUS");
            std::vector<std::string> names;
            std::vector<uint32_t> lengths;
            read_targets(cfg, names, lengths);
            const int ntids = (int)std::max<size_t>(names.size(), 1), world = (int)devices.size();
            const auto t_create = now();
            int rc = bdx_dist_create_threads(ranks.data(), &opts.o, libs.data(), nlibs, nbams, ntids, cfg.max_read_window_size(), devices.data(), world);
            if (rc != BDX_OK) throw std::runtime_error(std::string("bdx_dist_create_threads: ") + bdx_strerror(rc));
            const auto t_created = now();
            if (want_dumps)   // -g / -d: the supporting reads come with the result (gathered compact records, walked read by read on rank 0)
                for (bdx_dist* r : ranks) bdx_dist_set_collect_support(r, 1);
            std::vector<uint64_t> weight(lengths.begin(), lengths.end());  // chromosomes -> ranks by sequence length
            weight.resize(ntids, 0);
            std::vector<int> rank_of(ntids, 0);
            bdx_dist_plan(weight.data(), ntids, world, rank_of.data());
            // BAMs with their indexes: every rank pulls the BGZF ranges of ITS chromosomes and decodes them on ITS GPU (bdx_bamdec_*), as the
            // reference reads one chromosome through the index (io/RegionLimitedBamReader.hpp:43-71); several files are merged per
            // chromosome in the reference's order by a gather in HBM.  Without an index or with BDX_DECODE=host: the host producer
            // decodes everything once and routes the records to the ranks.
            const char* dm = getenv("BDX_DECODE");
            bool on_device = false;
            if (!(dm && !strcmp(dm, "host"))) {
                bool unsupported = true;
                n_reads = produce_sharded_on_device(cfg, (int)std::min(std::max(usable_cpus(), 2u), 32u), &targets, ranks, devices, rank_of, &unsupported);
                on_device = !unsupported;
                if (unsupported)   // (no index: nothing was appended; a file the device path gave up on half way: what it appended goes again)
                    for (bdx_dist* r : ranks)
                        if (bdx_dist_reset_reads(r) != BDX_OK) throw std::runtime_error(std::string("bdx_dist_reset_reads: ") + bdx_dist_last_error(r));
            }
            if (!on_device) {
                RoutingSink sink(ranks, rank_of);
                n_reads = produce_stream(cfg, opts.chr, (int)io_threads, &targets, sink);
            }
            if (timing)
                fprintf(stderr, "[bdx timing] sharded run over %d ranks: %s\n", world,
                        on_device ? "every rank decoded its chromosomes' BGZF ranges on its own GPU (through the BAM index)"
                                  : "records decoded by the host producer and routed to the ranks");
            const auto t_fed = now();
            {   // (device code loaded, the later stages' buffers sized: beside nothing, but outside the run; every rank on its own thread)
                std::vector<std::thread> pt;
                for (bdx_dist* r : ranks) pt.emplace_back([r] { (void)bdx_dist_prepare(r); });
                for (auto& t : pt) t.join();
            }
            t_decoded = now();
            if (timing)
                fprintf(stderr, "[bdx timing] sharded run: ranks created %.3f s after start, in %.3f s; reads on the ranks %.3f s later; prepared in %.3f s\n",
                        secs(t_start, t_create), secs(t_create, t_created), secs(t_created, t_fed), secs(t_fed, t_decoded));
            sharded_on_device = on_device;
            std::vector<int> rcs(world, BDX_OK);
            std::vector<std::thread> th;
            for (int r = 0; r < world; ++r) th.emplace_back([&, r] { rcs[r] = bdx_dist_run(ranks[r]); });
            for (auto& t : th) t.join();
            for (int r = 0; r < world; ++r)
                if (rcs[r] != BDX_OK)
                    throw std::runtime_error(std::string("bdx_dist_run: ") + bdx_strerror(rcs[r]) + " (" + bdx_dist_last_error(ranks[r]) + ")");
            ctx_owned = false;
            ctx = bdx_dist_result(ranks[0]);
            if (!ctx) throw std::runtime_error("bdx_dist_result: no result on rank 0");
        } else {
            GpuSink sink(ctx_ready, ctx, reserve);
            try {
                // One BAM: the file goes to the GPU as it is and is decoded there (bdx_bamdec_*: inflate, record boundaries, fields,
                // RG -> library, reader filter in HBM).  Several BAMs are merged in the reference's order by the host producer, which
                // also takes the files the device path does not (BDX_DECODE=host forces it).
                const char* dm = getenv("BDX_DECODE");
                bool decoded = false;
                if (cfg.num_bams() <= 16 && !(dm && !strcmp(dm, "host"))) {
                    const auto tb = now();
                    sink.bring_up(false);
                    if (timing) fprintf(stderr, "[bdx timing] GPU context + resident store ready %.3f s after start (waited %.3f s for it)\n", secs(t_start, now()), secs(tb, now()));
                    bool unsupported = false;
                    n_reads = produce_on_device(cfg, opts.chr, getenv("BDX_READ_THREADS") ? std::max(1, atoi(getenv("BDX_READ_THREADS"))) : (int)std::min(std::max(usable_cpus(), 2u), 16u), &targets, ctx, &unsupported);
                    if (unsupported) check(ctx, bdx_reset_reads(ctx), "bdx_reset_reads");
                    else { decoded = true; sink.up = true; }
                    device_decoded = decoded;
                }
                if (!decoded) {
                    n_reads = produce_stream(cfg, opts.chr, (int)io_threads, &targets, sink);
                    sink.bring_up();  // (an input without records never asked for a batch)
                }
            } catch (...) {
                if (ctx_ready.valid()) ctx_ready.wait();
                throw;
            }
            t_decoded = now();
            if (want_dumps) check(ctx, bdx_set_collect_support(ctx, 1), "bdx_set_collect_support");
            if (restored) check(ctx, bdx_set_pass1_statistics(ctx, cache.counters.data(), cache.covered_ref_len), "bdx_set_pass1_statistics");
            check(ctx, bdx_run(ctx), "bdx_run");
        }
        const auto t_ran = now();

        bdx_summary sum;
        check(ctx, bdx_get_summary(ctx, &sum), "bdx_get_summary");
        std::vector<uint32_t> lib_cnt(nlibs), bam_cnt(nbams), hist((size_t)nlibs * BDX_NUM_FLAGS);
        std::vector<float> seqcov(nlibs), dens(nlibs);
        check(ctx, bdx_get_counters(ctx, lib_cnt.data(), bam_cnt.data(), hist.data(), seqcov.data(), dens.data()), "bdx_get_counters");

        if (!opts.cache_file.empty()) {  // -C <cache>: what a later "-R <cache>" run needs (ConfigLoader.cpp:34-42)
            Pass1Cache w;
            for (size_t i = 0; i < opts.orig_argv.size(); ++i) {
                if (opts.orig_argv[i] == "-C" && i + 1 < opts.orig_argv.size()) { ++i; continue; }
                if (opts.orig_argv[i].compare(0, 2, "-C") == 0 && opts.orig_argv[i].size() > 2) continue;
                w.argv.push_back(opts.orig_argv[i]);
            }
            w.config_text = config_text;
            w.covered_ref_len = sum.covered_ref_len;
            w.nlibs = nlibs; w.nbams = nbams;
            w.counters = hist;
            w.counters.insert(w.counters.end(), lib_cnt.begin(), lib_cnt.end());
            w.counters.insert(w.counters.end(), bam_cnt.begin(), bam_cnt.end());
            write_cache(opts.cache_file, w);
        }
        using std::cout;
        cout << "#Software: breakdancer-max-mi355x (libbdx)" << std::endl;
        cout << "#Command: ";
        for (auto const& a : opts.orig_argv) cout << a << " ";
        cout << std::endl;
        cout << "#Library Statistics:" << std::endl;
        const uint32_t covered = sum.covered_ref_len;
        for (int i = 0; i < nlibs; ++i) {  // BreakDancerMax.cpp:83-137
            const LibraryConfig& lc = cfg.library_config(i);
            const float physical_coverage = float(lib_cnt[i] * lc.mean_insertsize) / covered / 2;
            cout << "#" << lc.bam_file << "\tmean:" << lc.mean_insertsize << "\tstd:" << lc.std_insertsize
                 << "\tuppercutoff:" << lc.uppercutoff << "\tlowercutoff:" << lc.lowercutoff << "\treadlen:" << lc.readlens
                 << "\tlibrary:" << lc.name << "\treflen:" << covered << "\tseqcov:" << seqcov[i] << "\tphycov:" << physical_coverage;
            for (int f = 0; f < BDX_NUM_FLAGS; ++f) {
                const uint32_t c = hist[(size_t)i * BDX_NUM_FLAGS + f];
                if (c) cout << "\t" << kPrintedFlagValue[f] << ":" << c;
            }
            cout << "\n";
        }
        cout << "#Chr1\tPos1\tOrientation1\tChr2\tPos2\tOrientation2\tType\tSize\tScore\tnum_Reads\tnum_Reads_lib";
        if (opts.o.print_af) cout << "\tAllele_frequency";
        if (!opts.o.cn_lib)
            for (auto const& b : cfg.bam_files()) {
                const size_t p = b.rfind("/");
                cout << "\t" << (p != std::string::npos ? b.substr(p + 1) : b);
            }
        cout << "\n";

        std::vector<bdx_sv> svs(sum.n_svs);
        check(ctx, bdx_get_svs(ctx, svs.data(), svs.size()), "bdx_get_svs");
        size_t nl = 0, nc = 0;
        for (auto const& s : svs) { nl = std::max<size_t>(nl, s.lib_begin + s.lib_count); nc = std::max<size_t>(nc, s.cn_begin + s.cn_count); }
        std::vector<int32_t> li(nl), lp(nl), ck(nc);
        std::vector<float> cv(nc);
        check(ctx, bdx_get_sv_lists(ctx, li.data(), lp.data(), nl, ck.data(), cv.data(), nc), "bdx_get_sv_lists");

        // (a large run's pinned result tables go back while the table is printed: the process's end is that much shorter.  BDX_TRIM=0: kept)
        if (n_reads > (size_t)(8u << 20) && !want_dumps && !getenv("BDX_CLEAN_EXIT") && !(getenv("BDX_TRIM") && !strcmp(getenv("BDX_TRIM"), "0"))) {
            bdx_ctx* tc = ctx;
            trimmer = std::thread([tc] { (void)bdx_trim_results(tc); });   // (only _exit may leave it behind: every path that destroys the context joins it first)
        }
        auto tname = [&](int t) { return (t >= 0 && (size_t)t < targets.size()) ? targets[t] : std::to_string(t); };

        // supporting reads of the printed SVs: a second decode pass fetches just those records (the reference keeps
        // sequence data for every anomalous read in memory instead)
        std::vector<uint32_t> sup_off;
        std::vector<uint64_t> sup_idx, wanted;
        std::vector<uint8_t> sup_flag;
        std::vector<SupportRead> sup_reads;
        std::unique_ptr<BedDump> bed;
        std::unique_ptr<FastqDump> fastq;
        if (want_dumps) {
            if (!opts.prefix_fastq.empty()) fastq.reset(new FastqDump(opts.prefix_fastq, cfg));
            if (!opts.dump_BED.empty()) bed.reset(new BedDump(opts.dump_BED, cfg, targets));
            size_t total = 0;
            sup_off.resize(svs.size() + 1);
            check(ctx, bdx_get_sv_support(ctx, sup_off.data(), nullptr, nullptr, 0, &total), "bdx_get_sv_support");
            sup_idx.resize(total);
            sup_flag.resize(total);
            check(ctx, bdx_get_sv_support(ctx, sup_off.data(), sup_idx.data(), sup_flag.data(), total, &total), "bdx_get_sv_support");
            for (size_t i = 0; i < svs.size(); ++i)
                if (svs[i].printed) wanted.insert(wanted.end(), sup_idx.begin() + sup_off[i], sup_idx.begin() + sup_off[i + 1]);
            std::sort(wanted.begin(), wanted.end());
            wanted.erase(std::unique(wanted.begin(), wanted.end()), wanted.end());
            collect_reads(cfg, opts.chr, (int)io_threads, wanted, sup_reads);
        }
        // one row of the table (BreakDancer.cpp:395-497) into `os`; returns nothing, the stream's manipulators carry over to the next row
        auto print_row = [&](std::ostream& os, const bdx_sv& s) {
            std::map<int, float> cn;  // key -> copy number
            for (int i = 0; i < s.cn_count; ++i) cn[ck[s.cn_begin + i]] = cv[s.cn_begin + i];
            std::string sptype;
            if (opts.o.cn_lib) {
                for (int i = 0; i < s.lib_count; ++i) {
                    const int lib = li[s.lib_begin + i];
                    std::string cn_str = "NA";
                    if (s.flag != BDX_ARP_CTX) {
                        auto f = cn.find(lib);
                        if (f != cn.end()) {
                            std::stringstream ss;
                            ss << std::fixed << std::setprecision(2) << f->second;
                            cn_str = ss.str();
                        }
                    }
                    if (!sptype.empty()) sptype += ":";
                    sptype += cfg.library_config(lib).name + "|" + std::to_string(lp[s.lib_begin + i]) + "," + cn_str;
                }
            } else {
                std::map<std::string, int> bam_rc;
                for (int i = 0; i < s.lib_count; ++i) bam_rc[cfg.library_config(li[s.lib_begin + i]).bam_file] += lp[s.lib_begin + i];
                for (auto const& kv : bam_rc) {
                    if (!sptype.empty()) sptype += ":";
                    sptype += kv.first + "|" + std::to_string(kv.second);
                }
                if (sptype.empty()) sptype = "NA";
            }
            os << tname(s.chr[0]) << "\t" << s.pos[0] << "\t" << s.fwd[0] << "+" << s.rev[0] << "-"
               << "\t" << tname(s.chr[1]) << "\t" << s.pos[1] << "\t" << s.fwd[1] << "+" << s.rev[1] << "-"
               << "\t" << opts.sv_type(s.flag) << "\t" << s.size << "\t" << s.score << "\t" << s.num_reads << "\t" << sptype;
            if (opts.o.print_af) os << "\t" << s.allele_frequency;
            if (!opts.o.cn_lib && s.flag != BDX_ARP_CTX) {
                for (size_t b = 0; b < cfg.num_bams(); ++b) {
                    auto f = cn.find((int)b);
                    if (f == cn.end()) os << "\tNA";
                    else {
                        // the reference never resets these manipulators on cout: later allele frequencies print
                        // fixed with two decimals as well (BreakDancer.cpp:492-493)
                        os << "\t";
                        os << std::fixed;
                        os << std::setprecision(2) << f->second;
                    }
                }
            }
            os << "\n";
        };
        // does this row leave the stream printing fixed with two decimals (see print_row)?
        auto sets_fixed = [&](const bdx_sv& s) {
            if (opts.o.cn_lib || s.flag == BDX_ARP_CTX) return false;
            for (int i = 0; i < s.cn_count; ++i)
                if (ck[s.cn_begin + i] >= 0 && (size_t)ck[s.cn_begin + i] < cfg.num_bams()) return true;
            return false;
        };
        // A large table (a genome's: tens of thousands of rows, 0.7 us each through the stream's formatting) is written by several threads,
        // each into its own string stream that starts in the state the sequential loop would have reached at its first row; the strings
        // leave in order.  The dumps (-g / -d) keep the sequential loop.
        std::vector<size_t> printed_rows;
        for (size_t i = 0; i < svs.size(); ++i)
            if (svs[i].printed) printed_rows.push_back(i);
        size_t fmt_threads = want_dumps || printed_rows.size() < 8192 ? 1 : std::min<size_t>(std::min<size_t>(io_threads, 8), printed_rows.size() / 2048);
        if (const char* ft = getenv("BDX_FORMAT_THREADS"))   // (tests, A/B: 1 = the sequential loop; n = n threads whatever the table's size)
            if (!want_dumps) fmt_threads = std::max<size_t>(1, std::min<size_t>((size_t)atoi(ft), std::max<size_t>(printed_rows.size(), 1)));
        if (fmt_threads > 1) {
            cout.flush();
            const size_t none = (size_t)-1;
            size_t flip_row = none;   // the first printed row that leaves the stream printing fixed (the rows behind it start that way)
            const bool fixed_already = (cout.flags() & std::ios_base::fixed) != 0;
            for (size_t k = 0; k < printed_rows.size() && !fixed_already; ++k)
                if (sets_fixed(svs[printed_rows[k]])) { flip_row = k; break; }
            std::vector<std::string> parts(fmt_threads);
            std::vector<std::thread> workers;
            for (size_t t = 0; t < fmt_threads; ++t)
                workers.emplace_back([&, t] {
                    const size_t k0 = printed_rows.size() * t / fmt_threads, k1 = printed_rows.size() * (t + 1) / fmt_threads;
                    std::ostringstream os;
                    os.flags(cout.flags());
                    os.precision(cout.precision());
                    if (flip_row != none && k0 > flip_row) { os << std::fixed; os << std::setprecision(2); }
                    for (size_t k = k0; k < k1; ++k) print_row(os, svs[printed_rows[k]]);
                    parts[t] = os.str();
                });
            for (auto& w : workers) w.join();
            for (auto const& part : parts) cout.write(part.data(), (std::streamsize)part.size());
            if (flip_row != none) { cout << std::fixed; cout << std::setprecision(2); }
        } else {
            size_t sv_i = 0;
            for (auto const& s : svs) {
                const size_t this_sv = sv_i++;
                if (!s.printed) continue;
                print_row(cout, s);
                if (want_dumps) {
                    SvForDump d;
                    d.chr0 = tname(s.chr[0]); d.pos0 = s.pos[0]; d.type = opts.sv_type(s.flag); d.size = s.size; d.flag = s.flag;
                    for (uint32_t k = sup_off[this_sv]; k < sup_off[this_sv + 1]; ++k) {
                        const size_t w = std::lower_bound(wanted.begin(), wanted.end(), sup_idx[k]) - wanted.begin();
                        d.reads.push_back(&sup_reads[w]);
                        d.read_flags.push_back(sup_flag[k]);
                    }
                    if (bed) bed->write(d);
                    if (fastq) fastq->write(d);
                }
            }
        }
        if (timing) {
            float ms[8] = {0};
            bdx_get_timings(ctx, ms, 8);
            fprintf(stderr, "[bdx timing] reads=%zu decode+merge+stream=%.3fs (%s, single pass, records classified "
                            "as they are produced) bdx_run=%.4fs format=%.3fs total=%.3fs\n",
                    n_reads, secs(t_start, t_decoded), device_decoded || sharded_on_device ? "BGZF inflate and record decode on the GPU" : "host decode threads",
                    secs(t_decoded, t_ran), secs(t_ran, now()), secs(t_start, now()));
            fprintf(stderr, "[bdx timing] inside bdx_run (ms): classify kernel %.3f, host waits for its share of the groups %.3f, host walk %.3f, "
                            "final wait + scores %.3f, whole call %.3f\n", ms[0], ms[4], ms[5], ms[6], ms[7]);
        }
        // Everything is printed: the process ends here.  Releasing tens of buffers, the pinned pages and the HIP runtime one by one
        // takes longer than the whole GPU path ran (0.03-0.05 s for a chromosome); the operating system does it in one go.
        // BDX_CLEAN_EXIT=1 walks the destructors instead (leak checkers).
        if (!getenv("BDX_CLEAN_EXIT")) {
            bed.reset();     // (the dump files are closed by their writers' destructors)
            fastq.reset();
            // (but not with work still queued on a device -- a record stage of the decoder's last batch, a copy nobody waits for: the
            // driver then takes 0.11-0.16 s to tear the queues down, against 0.015 s for an idle device.  bdx_warm_up ends in a device-wide wait)
            if (sharded) { for (int dv : devices) (void)bdx_warm_up(dv); }
            else (void)bdx_warm_up(bdx_device(ctx));
            report_result(0);
            if (timing) {
                struct timespec ts;
                clock_gettime(CLOCK_REALTIME, &ts);
                fprintf(stderr, "[bdx timing] _exit called at %.6f (wall clock)\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
            }
            _exit(0);
        }
        if (trimmer.joinable()) trimmer.join();
        release_device_decoders();   // (before their sink: a decoder borrows its context's streams)
        if (ctx_owned) bdx_destroy(ctx);  // (a sharded run's result context belongs to rank 0, released with the ranks)
        ctx = nullptr;
    } catch (std::exception const& e) {
        std::cerr << "ERROR: " << e.what() << "\n";
        if (trimmer.joinable()) trimmer.join();
        if (ctx && ctx_owned) bdx_destroy(ctx);
        return 1;
    }
    return 0;
}

}  // namespace
