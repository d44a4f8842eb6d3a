// can the runtime's start-up be spread over threads? (tools, not the product)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    hipInit(0); hipSetDevice(0);   printf("mode %d: init %7.2f ms\n", mode, ms());
    hipStream_t s[4];
    if (mode == 0) {
        for (int i = 0; i < 4; ++i) { hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); printf("  stream %d at %7.2f ms\n", i, ms()); }
    } else if (mode == 1) {
        std::vector<std::thread> th;
        for (int i = 0; i < 4; ++i) th.emplace_back([&, i] { hipSetDevice(0); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); });
        for (auto& t : th) t.join();
        printf("  4 streams by 4 threads at %7.2f ms\n", ms());
    } else {
        void* h[4];
        std::vector<std::thread> th;
        for (int i = 0; i < 4; ++i) th.emplace_back([&, i] { hipSetDevice(0); if (i < 2) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); else hipHostMalloc(&h[i], 9 << 20, 0); });
        for (auto& t : th) t.join();
        printf("  2 streams + 2 x 9 MiB pinned by 4 threads at %7.2f ms\n", ms());
    }
    return 0;
}
