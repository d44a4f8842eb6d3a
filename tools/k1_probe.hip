// Probe: K1 timing vs grid size against a pure streaming-read ceiling with the same column mix.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ibreakdancer_amd/csrc tools/k1_probe.hip -o bin/k1_probe
#include "../breakdancer_amd/csrc/k1_classify.hip"

#include <cstdio>
#include <vector>

using namespace bdx;

__global__ __launch_bounds__(256) void stream_read_kernel(ReadsSoA r, uint64_t n, uint32_t ntiles, uint8_t* cls, unsigned* sink) {
    unsigned acc = 0;
    for (uint32_t tile = blockIdx.x * kWaves + (threadIdx.x >> 6); tile < ntiles; tile += gridDim.x * kWaves) {
        const uint64_t base = (uint64_t)tile * kTile + (uint64_t)(threadIdx.x & 63) * 4;
        if (base + 4 > n) continue;
        const int4 a = *(const int4*)(r.tid + base);
        const int4 b = *(const int4*)(r.pos + base);
        const int4 c = *(const int4*)(r.mtid + base);
        const int4 d = *(const int4*)(r.mpos + base);
        const int4 e = *(const int4*)(r.isize + base);
        const ushort4 f = *(const ushort4*)(r.flag + base);
        const uchar4 q = *(const uchar4*)(r.mapq + base);
        const uchar4 l = *(const uchar4*)(r.lib + base);
        const uchar4 m = *(const uchar4*)(r.bam + base);
        const unsigned x = a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.x ^ q.x ^ l.y ^ m.z ^ a.w ^ b.x ^ c.y ^ d.z ^ e.w;
        acc += x;
        *(uchar4*)(cls + base) = make_uchar4(x & 7, (x >> 3) & 7, (x >> 6) & 7, (x >> 9) & 7);
    }
    if (acc == 0x12345678u) *sink = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 15000000ull;
    const uint32_t ntiles = (uint32_t)((n + kTile - 1) / kTile), tstride = (ntiles + 15) & ~15u;
    std::vector<int32_t> tid(n, 0), pos(n), mtid(n, 0), mpos(n), isz(n);
    std::vector<uint16_t> flag(n);
    std::vector<uint8_t> mq(n), lib(n, 0), bam(n, 0);
    uint64_t s = 88172645463325252ull;
    for (uint64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        pos[i] = (int32_t)(i * 3 + 1000);
        const bool rev = s & 1;
        const int ins = 300 + (int)((s >> 8) % 200) + (((s >> 20) % 100) == 0 ? 1200 : 0);
        mpos[i] = rev ? pos[i] - ins : pos[i] + ins;
        isz[i] = rev ? -ins : ins;
        flag[i] = (uint16_t)(0x1 | 0x2 | (rev ? 0x10 : 0x20) | ((s >> 4) & 1 ? 0x40 : 0x80));
        mq[i] = ((s >> 40) % 100) < 3 ? 20 : 60;
    }
    ReadsSoA r{};
    void* p;
#define UP(field, vec) CK(hipMalloc(&p, vec.size() * sizeof(vec[0]) + 64)); CK(hipMemcpy(p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice)); r.field = (decltype(r.field))p;
    UP(tid, tid) UP(pos, pos) UP(mtid, mtid) UP(mpos, mpos) UP(isize, isz) UP(flag, flag) UP(mapq, mq) UP(lib, lib) UP(bam, bam)
    const int nlibs = 1, nbams = 1, nkeys = 1, ncols = 3, ncnt = 13;
    DevLib dl{490.f, 310.f, 35, 0};
    K1Params k{};
    k.r = r; k.n = n; k.ntiles = ntiles; k.tstride = tstride; k.nlibs = nlibs; k.nbams = nbams; k.nkeys = nkeys; k.max_sd = 1000000000;
    CK(hipMalloc(&p, sizeof(dl))); CK(hipMemcpy(p, &dl, sizeof(dl), hipMemcpyHostToDevice)); k.libs = (DevLib*)p;
    CK(hipMalloc(&p, n + 64)); k.cls = (uint8_t*)p;
    CK(hipMalloc(&p, (size_t)ncols * tstride * 4)); k.tile_tot = (uint32_t*)p;
    CK(hipMalloc(&p, (size_t)nbams * tstride * sizeof(MonoRec))); k.tile_mono = (MonoRec*)p;
    CK(hipMemset(p, 0xFF, (size_t)nbams * tstride * sizeof(MonoRec)));
    CK(hipMalloc(&p, (size_t)32768 * ncnt * 4)); k.blk_cnt = (uint32_t*)p;
    unsigned* sink; CK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = k1_lds_bytes(nlibs, nbams, nkeys);
    const double bytes_algo = 28.0 * n, bytes_real = 26.0 * n;
    for (int grid : {1024, 2048, 4096, 8192, 16384, 32768}) {
        for (int which = 0; which < 2; ++which) {
            float best = 1e9;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, 0, r, n, ntiles, k.cls, sink);
                else launch_k1(k, grid, lds, 0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("grid %5d %-12s %8.1f us  algo %7.0f GB/s  real %7.0f GB/s\n", grid, which ? "k1" : "stream-only", best * 1e3,
                   bytes_algo / best / 1e6, bytes_real / best / 1e6);
        }
    }
    // finalize kernel: whole, without the monoid fold (nbams = 0), scan workgroups only (ncols blocks, via grid trick)
    {
        FinalizeParams fp{};
        fp.ntiles = ntiles; fp.tstride = tstride; fp.nblk = 0; fp.nlibs = nlibs; fp.nbams = nbams; fp.nkeys = nkeys; fp.ncols = ncols;
        fp.ncnt = ncnt; fp.w0 = 200; fp.tile_tot = k.tile_tot; fp.tile_mono = k.tile_mono; fp.blk_cnt = k.blk_cnt;
        CK(hipMalloc(&p, (size_t)ncols * tstride * 4)); fp.tile_pre = (uint32_t*)p;
        CK(hipMalloc(&p, 4096)); fp.cnt = (uint32_t*)p;
        CK(hipMalloc(&p, sizeof(Pass1))); fp.p1 = (Pass1*)p;
        fp.nfold = 58; CK(hipMalloc(&p, 64 * sizeof(MonoRec))); fp.fold_part = (MonoRec*)p;
        launch_k1(k, 8192, lds, 0);
        for (int variant = 0; variant < 2; ++variant) {
            FinalizeParams q = fp;
            if (variant == 1) q.nbams = 0;
            float best = 1e9;
            for (int rep = 0; rep < 10; ++rep) {
                CK(hipEventRecord(e0));
                launch_finalize(q, 0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("finalize %-22s %8.1f us (ntiles %u)\n", variant ? "without monoid fold" : "full", best * 1e3, ntiles);
        }
    }
    CK(hipGetLastError());
    return 0;
}
