// Page cache -> pinned staging memory: what a CPU pays per byte, by method (the feeder's ceiling: 16 granted CPUs copy the whole file once).
//   readpath_probe <file> [GiB to move]      pread / mmap + memcpy (MADV_POPULATE_READ first) / mmap + memcpy (faults as it goes), 1, 8 and 16 threads
#include <hip/hip_runtime.h>
#include <fcntl.h>
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
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argv[1];
    const size_t want = (size_t)((argc > 2 ? atof(argv[2]) : 4.0) * (1 << 30));
    const int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const size_t total = std::min<size_t>(want, (size_t)st.st_size) & ~(((size_t)1 << 20) - 1);
    const size_t slice = (size_t)1 << 20, nbuf = 64;
    uint8_t* pinned;
    CK(hipHostMalloc(&pinned, nbuf * slice));   // 64 MiB of staging, written round and round
    uint8_t* plain = (uint8_t*)malloc(nbuf * slice);
    memset(plain, 1, nbuf * slice);
    const uint8_t* map = (const uint8_t*)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    for (int method = 0; method < 5; ++method)
        for (int threads : {1, 8, 16}) {
            if (method == 3 || method == 4) {   // a fresh mapping: page tables empty again
                munmap((void*)map, (size_t)st.st_size);
                map = (const uint8_t*)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
            }
            std::atomic<size_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t] {
                    for (;;) {
                        const size_t i = next.fetch_add(1);
                        if (i * slice >= total) break;
                        uint8_t* dst = (method == 1 ? plain : pinned) + (i % nbuf) * slice;
                        if (method <= 1) {
                            for (size_t done = 0; done < slice;) { const ssize_t r = pread(fd, dst + done, slice - done, (off_t)(i * slice + done)); if (r <= 0) exit(2); done += (size_t)r; }
                        } else {
                            if (method == 3) madvise((void*)(map + i * slice), slice, MADV_POPULATE_READ);
                            memcpy(dst, map + i * slice, slice);
                        }
                    }
                });
            for (auto& x : th) x.join();
            const double dt = now() - t0;
            static const char* names[] = {"pread -> pinned", "pread -> malloc", "mmap (mapped before) + memcpy -> pinned", "mmap + MADV_POPULATE_READ + memcpy -> pinned", "mmap (fresh) + memcpy -> pinned"};
            printf("%-46s %2d threads: %6.2f GB/s  (%.2f GB/s per thread)\n", names[method], threads, total / dt / 1e9, total / dt / 1e9 / threads);
        }
    return 0;
}
