// bdx-inflate-check <bgzf file>...: every BGZF block of the files through fast_inflate and through zlib; reports
// mismatches (exit 1) and the two decoders' throughput.  Test tooling for fast_inflate.cpp.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fast_inflate.h"

static bool zlib_inflate(const uint8_t* src, size_t clen, uint8_t* dst, size_t ulen) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(src); zs.avail_in = (uInt)clen; zs.next_out = dst; zs.avail_out = (uInt)ulen;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0;
}

int main(int argc, char** argv) {
    size_t blocks = 0, bad = 0, rejected = 0, bytes = 0;
    double t_fast = 0, t_zlib = 0;
    std::vector<uint8_t> a(65536 + 64), b(65536 + 64);
    for (int f = 1; f < argc; ++f) {
        const int fd = open(argv[f], O_RDONLY);
        if (fd < 0) { fprintf(stderr, "cannot open %s\n", argv[f]); return 2; }
        struct stat st;
        fstat(fd, &st);
        const size_t size = (size_t)st.st_size;
        const uint8_t* m = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        size_t off = 0;
        while (off + 18 <= size) {
            const uint8_t* h = m + off;
            if (h[0] != 31 || h[1] != 139) { fprintf(stderr, "%s: not BGZF at %zu\n", argv[f], off); return 2; }
            const size_t xlen = h[10] | (h[11] << 8);
            const size_t total = (size_t)(h[16] | (h[17] << 8)) + 1;
            const size_t coff = off + 12 + xlen, clen = total - 12 - xlen - 8;
            const uint32_t ulen = h[total - 4] | (h[total - 3] << 8) | (h[total - 2] << 16) | ((uint32_t)h[total - 1] << 24);
            off += total;
            if (!ulen) continue;
            ++blocks;
            bytes += ulen;
            auto t0 = std::chrono::steady_clock::now();
            const bool okz = zlib_inflate(m + coff, clen, b.data(), ulen);
            auto t1 = std::chrono::steady_clock::now();
            bool okf = false;
            if (coff + clen + 32 <= size) okf = bdhost::fast_inflate(m + coff, clen, a.data(), ulen, 64);
            auto t2 = std::chrono::steady_clock::now();
            t_zlib += std::chrono::duration<double>(t1 - t0).count();
            t_fast += std::chrono::duration<double>(t2 - t1).count();
            if (!okz) { fprintf(stderr, "%s: zlib rejects the block at %zu\n", argv[f], coff); ++bad; continue; }
            if (!okf) { ++rejected; continue; }
            if (memcmp(a.data(), b.data(), ulen) != 0) { fprintf(stderr, "%s: MISMATCH in the block at %zu\n", argv[f], coff); ++bad; }
        }
        munmap((void*)m, size);
    }
    printf("blocks %zu bytes %zu mismatches %zu left_to_zlib %zu fast %.1f MB/s zlib %.1f MB/s\n", blocks, bytes, bad, rejected,
           t_fast > 0 ? bytes / t_fast / 1e6 : 0.0, t_zlib > 0 ? bytes / t_zlib / 1e6 : 0.0);
    return bad ? 1 : 0;
}
