// bin/bdx-feed-probe: the ceilings of getting a file to the GPU the way the decoder's feeder does (host/producer.cpp: ReadPool + bdx_bamdec_acquire /
// submit) -- page cache -> pinned staging buffers by reader threads (pread, 1 MiB slices), staging -> HBM by hipMemcpyAsync of 8 MiB pieces in order --
// each alone and both pipelined, on the first <GiB> of <file>.  Prints one JSON line; bench.py puts it beside the CLI's BAM -> table rate.
//   bdx-feed-probe <file> [GiB = 4] [threads = 16] [pieces in flight = 12]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// a copy whose stores bypass the caches: the destination is read next by the copy engine, not by a CPU
static void copy_streaming(uint8_t* dst, const uint8_t* src, size_t n) {
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
        _mm256_stream_si256((__m256i*)(dst + i), a);
        _mm256_stream_si256((__m256i*)(dst + i + 32), b);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bdx-feed-probe <file> [GiB] [threads] [pieces in flight]\n"); return 2; }
    const double gib = argc > 2 ? atof(argv[2]) : 4.0;
    const int threads = argc > 3 ? atoi(argv[3]) : 16, nbuf = argc > 4 ? atoi(argv[4]) : 12;
    const int how = argc > 5 ? atoi(argv[5]) : 0;   // 0 pread, 1 mmap + memcpy, 2 mmap + a copy with streaming stores
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror(argv[1]); return 2; }
    struct stat st;
    fstat(fd, &st);
    const size_t piece = (size_t)8 << 20, slice = (size_t)1 << 20;
    const uint8_t* map = how ? (const uint8_t*)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0) : nullptr;
    if (how && map == MAP_FAILED) { perror("mmap"); return 2; }
    const size_t npieces = std::min<size_t>((size_t)(gib * (1 << 30)), (size_t)st.st_size) / piece;
    if (!npieces) { fprintf(stderr, "file smaller than a piece\n"); return 2; }
    CK(hipSetDevice(0));
    std::vector<uint8_t*> hb(nbuf);
    for (auto& p : hb) CK(hipHostMalloc(&p, piece));
    uint8_t* dev;
    CK(hipMalloc(&dev, 4 * piece));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(nbuf);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // readers: slices of piece p into buffer p % nbuf, allowed once the copy that last used the buffer is through (ready[p % nbuf] >= p - nbuf + 1 pieces copied)
    auto run = [&](bool do_read, bool do_copy) {
        std::atomic<size_t> next_slice{0};
        std::vector<std::atomic<size_t>> left(npieces);
        for (auto& l : left) l.store(piece / slice);
        std::atomic<size_t> copied{0};   // pieces whose H2D copy has completed (their buffers are free again)
        const double t0 = now();
        std::vector<std::thread> th;
        if (do_read)
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t i = next_slice.fetch_add(1);
                        const size_t p = i / (piece / slice);
                        if (p >= npieces) break;
                        while (do_copy && p >= copied.load(std::memory_order_acquire) + (size_t)nbuf) std::this_thread::yield();
                        uint8_t* dst = hb[p % nbuf] + (i % (piece / slice)) * slice;
                        if (how == 1) memcpy(dst, map + i * slice, slice);
                        else if (how == 2) copy_streaming(dst, map + i * slice, slice);
                        else
                            for (size_t done = 0; done < slice;) {
                                const ssize_t r = pread(fd, dst + done, slice - done, (off_t)(i * slice + done));
                                if (r <= 0) exit(3);
                                done += (size_t)r;
                            }
                        left[p].fetch_sub(1, std::memory_order_release);
                    }
                });
        if (do_copy) {
            size_t waited = 0;
            for (size_t p = 0; p < npieces; ++p) {
                while (do_read && left[p].load(std::memory_order_acquire)) std::this_thread::yield();
                CK(hipMemcpyAsync(dev + (p % 4) * piece, hb[p % nbuf], piece, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(ev[p % nbuf], s));
                // (the oldest copies' completion: what frees buffers for the readers)
                while (waited + (size_t)nbuf / 2 <= p) { CK(hipEventSynchronize(ev[waited % nbuf])); ++waited; copied.store(waited, std::memory_order_release); }
            }
            CK(hipStreamSynchronize(s));
            copied.store(npieces, std::memory_order_release);
        }
        for (auto& x : th) x.join();
        return (double)npieces * piece / (now() - t0) / 1e9;
    };
    (void)run(true, false);   // (warm: page tables of the staging buffers, the runtime's copy path)
    (void)run(false, true);
    const double r = run(true, false), c = run(false, true), both = run(true, true);
    printf("{\"how\": \"%s\", \"bytes\": %zu, \"threads\": %d, \"pieces_in_flight\": %d, \"page_cache_to_pinned_gb_s\": %.2f, \"pinned_to_hbm_gb_s\": %.2f, \"both_pipelined_gb_s\": %.2f}\n",
           how == 0 ? "pread" : how == 1 ? "mmap + memcpy" : "mmap + streaming stores", npieces * piece, threads, nbuf, r, c, both);
    return 0;
}
