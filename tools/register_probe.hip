// How a file that sits in the page cache gets to the GPU fastest on this box (the CLI's feeding path: 2 GB of BAM per run).
//   A  what the decoder does: threads pread() pieces into pinned staging (hipHostMalloc), one hipMemcpyAsync per piece
//   B  mmap the file, hipHostRegister every piece (threads), hipMemcpyAsync from the mapping: no CPU copy, the DMA reads the page cache
//   C  mmap the file, hipMemcpy from the mapping as it is (pageable: the runtime stages or pins)
//   D  mmap + one hipHostRegister of the whole mapping
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/register_probe tools/register_probe.hip -lpthread
// run:   /tmp/register_probe <file> [threads=16] [piece MiB=8]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    const size_t piece = (size_t)(argc > 3 ? atoi(argv[3]) : 8) << 20;
    const int fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st)) { perror("open"); return 1; }
    const size_t bytes = (size_t)st.st_size, npieces = (bytes + piece - 1) / piece;
    CK(hipSetDevice(0));
    void* dev = nullptr;
    CK(hipMalloc(&dev, bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    {   // warm the page cache (and find out whether it holds the file)
        std::vector<char> tmp(piece);
        const double t0 = now();
        for (size_t p = 0; p < npieces; ++p) (void)!pread(fd, tmp.data(), piece, (off_t)(p * piece));
        printf("file %.1f MB, %zu pieces of %zu MiB; one thread reading it through: %.3f s\n", bytes / 1e6, npieces, piece >> 20, now() - t0);
    }
    // ---- A ----
    {
        const int nstage = 4 * threads;
        std::vector<void*> stage(nstage);
        for (auto& p : stage) CK(hipHostMalloc(&p, piece, hipHostMallocDefault));
        for (int rep = 0; rep < 2; ++rep) {
            std::atomic<size_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t] {
                    for (;;) {
                        const size_t p = next.fetch_add(1);
                        if (p >= npieces) break;
                        (void)!pread(fd, stage[(p % nstage)], std::min(piece, bytes - p * piece), (off_t)(p * piece));
                    }
                });
            for (auto& t : th) t.join();
            const double t1 = now();
            printf("A%d  %d threads pread into pinned staging (no copies to the device): %.3f s -> %.1f GB/s\n", rep, threads, t1 - t0, bytes / (t1 - t0) / 1e9);
        }
        // with the copies, as a pipeline: reader threads fill, the main thread submits in order
        {
            std::vector<std::atomic<int>> ready(npieces);
            for (auto& r : ready) r = 0;
            std::vector<hipEvent_t> done(nstage);
            for (auto& e : done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            std::vector<std::atomic<int>> free_at(nstage);   // piece index whose copy must have completed before the slot is reused
            std::atomic<size_t> next{0}, submitted{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t p = next.fetch_add(1);
                        if (p >= npieces) break;
                        while (p >= submitted.load() + (size_t)nstage) std::this_thread::yield();   // slot still waiting for its copy to be submitted
                        if (p >= (size_t)nstage) (void)hipEventSynchronize(done[p % nstage]);
                        (void)!pread(fd, stage[p % nstage], std::min(piece, bytes - p * piece), (off_t)(p * piece));
                        ready[p] = 1;
                    }
                });
            for (size_t p = 0; p < npieces; ++p) {
                while (!ready[p].load()) std::this_thread::yield();
                CK(hipMemcpyAsync((char*)dev + p * piece, stage[p % nstage], std::min(piece, bytes - p * piece), hipMemcpyHostToDevice, s));
                CK(hipEventRecord(done[p % nstage], s));
                submitted = p + 1;
            }
            CK(hipStreamSynchronize(s));
            const double t1 = now();
            for (auto& t : th) t.join();
            printf("A   pread + hipMemcpyAsync pipeline: %.3f s -> %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
        }
        for (auto& p : stage) (void)hipHostFree(p);
    }
    // ---- B ----
    for (int rep = 0; rep < 2; ++rep) {
        char* map = (char*)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { perror("mmap"); return 1; }
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&] {
                (void)hipSetDevice(0);
                for (;;) {
                    const size_t p = next.fetch_add(1);
                    if (p >= npieces) break;
                    const hipError_t e = hipHostRegister(map + p * piece, std::min(piece, bytes - p * piece), hipHostRegisterDefault);
                    if (e != hipSuccess) { if (!failed.exchange(1)) printf("B   hipHostRegister(piece %zu): %s\n", p, hipGetErrorString(e)); break; }
                }
            });
        for (auto& t : th) t.join();
        const double t1 = now();
        if (!failed) {
            for (size_t p = 0; p < npieces; ++p) CK(hipMemcpyAsync((char*)dev + p * piece, map + p * piece, std::min(piece, bytes - p * piece), hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
        }
        const double t2 = now();
        for (size_t p = 0; p < npieces && !failed; ++p) (void)hipHostUnregister(map + p * piece);
        const double t3 = now();
        printf("B%d  mmap + hipHostRegister per piece on %d threads: register %.3f s, copies %.3f s (%.1f GB/s), unregister %.3f s%s\n", rep, threads, t1 - t0, t2 - t1,
               bytes / (t2 - t1) / 1e9, t3 - t2, failed ? "  [FAILED]" : "");
        munmap(map, bytes);
    }
    // ---- C ----
    {
        char* map = (char*)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
        const double t0 = now();
        const hipError_t e = hipMemcpy(dev, map, bytes, hipMemcpyHostToDevice);
        const double t1 = now();
        printf("C   hipMemcpy from the pageable mapping: %.3f s -> %.1f GB/s (%s)\n", t1 - t0, bytes / (t1 - t0) / 1e9, hipGetErrorString(e));
        munmap(map, bytes);
    }
    // ---- D ----
    {
        char* map = (char*)mmap(nullptr, bytes, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        const double t0 = now();
        const hipError_t e = hipHostRegister(map, bytes, hipHostRegisterDefault);
        const double t1 = now();
        double t2 = t1;
        if (e == hipSuccess) {
            CK(hipMemcpyAsync(dev, map, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            t2 = now();
            (void)hipHostUnregister(map);
        }
        printf("D   one hipHostRegister of the populated mapping: %.3f s (%s), copy %.3f s -> %.1f GB/s\n", t1 - t0, hipGetErrorString(e), t2 - t1, t2 > t1 ? bytes / (t2 - t1) / 1e9 : 0.0);
        munmap(map, bytes);
    }
    return 0;
}
