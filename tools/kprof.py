"""In-kernel phase clocks of the walk and score kernels (a measurement build, not the product).

Build:  hipcc ... -DBDX_KPROF -c breakdancer_amd/csrc/k6_assemble.hip -o /tmp/k6_kprof.o, link the other objects of libbdx.so with it
        into variants/libbdx_kprof.so ("in-kernel clocks").
Run:    python tools/kprof.py [--length 50000000]
The kernels store wall_clock64() (100 MHz) at a few places; this script runs configs[1] a few times and prints the distribution
of the intervals of the LAST run.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", type=int, default=50_000_000)
    ap.add_argument("--lib", default=os.path.join(ROOT, "variants", "libbdx_kprof.so"))
    ap.add_argument("--genome", type=float, default=0.0, help="hg38 lengths x this, 4 libraries (1/8: a GPU's share of a 30x genome) instead of configs[1]")
    ap.add_argument("--set", action="append", default=[], help="name=value for bdx_set_debug")
    a = ap.parse_args()
    import breakdancer_amd._lib as lib
    lib.LIB_PATH = a.lib
    import torch
    import breakdancer_amd.api as bda
    from breakdancer_amd.api import LibraryConfig, Options
    from breakdancer_amd.synth import LIB_C2, make_chromosome
    if a.genome:
        from breakdancer_amd.bamwrite import HG38_MBP, LIBS4
        from breakdancer_amd.synth import make_genome
        lengths = [int(m * 1e6 * a.genome) for m in HG38_MBP]
        d = make_genome(lengths, coverage=30.0, seed=11, libs=LIBS4, lib_bam=(0, 0, 0, 0), n_translocations=5000)
        libs = [LibraryConfig(mean_insertsize=m, std_insertsize=sd, uppercutoff=m + 3 * sd, lowercutoff=m - 3 * sd, readlens=100.0, name="lib%d" % i) for i, (m, sd) in enumerate(LIBS4)]
        ntids = len(lengths)
    else:
        d = make_chromosome(length=a.length, seed=1)
        libs, ntids = [LibraryConfig(**LIB_C2)], 1
    n = len(d["pos"])
    dev = torch.device("cuda", 0)
    tens = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
    bd = bda.BreakDancer(Options(), libs, 1, ntids=ntids, max_read_window_size=200, device=0)
    for kv in a.set:
        k, v = kv.split("=")
        bd.set_debug(k, int(v))
    bd.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
    bd.set_enqueue_ahead(0)
    L = lib.load()
    buf = np.zeros(8 * 65536, dtype=np.uint64)
    L.bdx_debug_kprof.argtypes = [C.c_void_p, C.c_size_t]
    buf3 = np.zeros(8 * 65536, dtype=np.uint64)
    have4 = hasattr(L, "bdx_debug_kprof4")
    if have4:
        L.bdx_debug_kprof4.argtypes = [C.c_void_p, C.c_size_t]
        buf4 = np.zeros(65536 * 8, np.uint64)
    have3 = hasattr(L, "bdx_debug_kprof3")
    if have3:
        L.bdx_debug_kprof3.argtypes = [C.c_void_p, C.c_size_t]
    buf1 = np.zeros(8 * 65536, dtype=np.uint64)
    have1 = hasattr(L, "bdx_debug_kprof1")
    if have1:
        L.bdx_debug_kprof1.argtypes = [C.c_void_p, C.c_size_t]
    for _ in range(5):
        bd.run()
    torch.cuda.synchronize()
    if have1:
        L.bdx_debug_kprof1(buf1.ctypes.data_as(C.c_void_p), buf1.size)
    L.bdx_debug_kprof(buf.ctypes.data_as(C.c_void_p), buf.size)  # (clears the device buffer)
    if have3:
        L.bdx_debug_kprof3(buf3.ctypes.data_as(C.c_void_p), buf3.size)
    if have4:
        L.bdx_debug_kprof4(buf4.ctypes.data_as(C.c_void_p), buf4.size)
    bd.run()
    torch.cuda.synchronize()
    rc = L.bdx_debug_kprof(buf.ctypes.data_as(C.c_void_p), buf.size)
    assert rc == 0, rc
    t = buf.reshape(65536, 8).astype(np.int64)
    if have3:
        L.bdx_debug_kprof3(buf3.ctypes.data_as(C.c_void_p), buf3.size)
        t3 = buf3.reshape(65536, 8).astype(np.int64)
    split = bd.walk_split()
    print("reads", n, "svs", bd.summary().get("n_sv") if hasattr(bd.summary(), "get") else "", "walk split", split)

    def stats(name, x):
        x = np.asarray(x, dtype=np.float64) / 100.0  # us
        if len(x) == 0:
            print("  %-34s (none)" % name)
            return
        print("  %-34s n=%6d  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % (name, len(x), x.min(), np.percentile(x, 50), np.percentile(x, 90), x.max()))

    # walk kernel: one lane per component, rows = workgroup (clocks of its lane 0 / of the whole wave at the end)
    own = t[:16384]
    k0 = own[:, 0][own[:, 0] > 0].min()
    have = own[:, 4] > 0  # waves whose lane 0 walks a component
    print("walk kernel: %d waves, %d with components (clocks of lane 0)" % ((own[:, 7] > 0).sum(), have.sum()))
    stats("wave entry", own[have, 0] - k0)
    stats("run constants in LDS", own[have, 1] - own[have, 0])
    stats("n_regions known", own[have, 2] - own[have, 1])
    stats("description arrived", own[have, 3] - own[have, 2])
    stats("tables built (phase 0)", own[have, 4] - own[have, 3])
    stats("traversal (phase 1) + touches", own[have, 5] - own[have, 4])
    c1 = have & (own[:, 6] > 0)
    stats("first call", own[c1, 6] - own[c1, 5])
    stats("rest + end", own[c1, 7] - own[c1, 6])
    stats("wave end", own[have, 7] - k0)
    # the walk kernel's FIRST assemble_sv call of a wave (rows 16384 + workgroup, lane 0's clocks)
    ac = t[16384:16384 + 8192]
    ha = ac[:, 4] > 0
    if ha.any():
        print("walk kernel, first call of a wave (lane 0 has one in %d waves):" % ha.sum())
        stats("parts of the groups arrived", ac[ha, 1] - ac[ha, 0])
        stats("per-library merge", ac[ha, 2] - ac[ha, 1])
        stats("proper-read samples, copy numbers", ac[ha, 3] - ac[ha, 2])
        stats("record stored", ac[ha, 4] - ac[ha, 3])
    # table kernels (round 6: placement and scores are two launches).  Rows 32768 + workgroup: k6_place_kernel's wave 0; rows 49152 + wave:
    # k6_score_kernel
    pl = t[32768:49152]
    hp = pl[:, 2] > 0
    if hp.any():
        p0 = pl[:, 0][pl[:, 0] > 0].min()
        print("place kernel: %d workgroups entered, %d with regions" % ((pl[:, 0] > 0).sum(), hp.sum()))
        stats("workgroup entry", pl[:, 0][pl[:, 0] > 0] - p0)
        stats("loads + block scan", pl[hp, 1] - pl[hp, 0])
        stats("look-back", pl[hp, 2] - pl[hp, 1])
    sc = t[49152:]
    hs = sc[:, 5] > 0
    if hs.any():
        s0 = sc[:, 0][sc[:, 0] > 0].min()
        g = sc[:, 4] > 0
        print("score kernel: %d waves, %d with candidates (marks of a wave's LAST 64 candidates)" % (hs.sum(), g.sum()))
        stats("wave entry", sc[hs, 0] - s0)
        stats("entry -> records gathered", sc[g, 3] - sc[g, 0])
        stats("terms, K5, scores", sc[g, 4] - sc[g, 3])
        stats("records -> host, end", sc[g, 5] - sc[g, 4])
        stats("wave end", sc[hs, 5] - s0)
    if have1:
        L.bdx_debug_kprof1(buf1.ctypes.data_as(C.c_void_p), buf1.size)
        t1 = buf1.reshape(65536, 8).astype(np.int64)
        h = t1[:, 2] > 0
        z = t1[h, 0].min()
        print("finalize kernel: %d waves (rows: workgroup * 16 + wave)" % h.sum())
        nscan = 3 * 4  # workgroups that scan tile-total columns at configs[1]: 3 columns x 4 chunks (rows: workgroup * 16 + wave)
        for name, sel in (("column scans", np.arange(65536) < 16 * nscan), ("monoid folds", np.arange(65536) >= 16 * nscan)):
            m = h & sel
            if m.any():
                stats(name + ": entry", t1[m, 0] - z)
                stats(name + ": inputs arrived", t1[m, 1] - t1[m, 0])
                stats(name + ": done", t1[m, 2] - z)
    if have3:
        for name, rows in (("head scan (4 columns)", t3[:32768]), ("accept scan", t3[32768:])):
            h = rows[:, 3] > 0
            if not h.any():
                continue
            z = rows[h, 0].min()
            print("%s: %d waves" % (name, h.sum()))
            stats("wave entry", rows[h, 0] - z)
            stats("inputs + block scan", rows[h, 1] - rows[h, 0])
            stats("look-back", rows[h, 2] - rows[h, 1])
            stats("outputs issued", rows[h, 3] - rows[h, 2])
            stats("wave end", rows[h, 3] - z)
    if have4:
        # join kernel: rows = wave (the first 65,536), clocks of lane 0: entry, region table forwarded, key and region arrived, first
        # compare-and-swap back, (second mates) partner confirmed and exchanged
        L.bdx_debug_kprof4(buf4.ctypes.data_as(C.c_void_p), buf4.size)
        t4 = buf4.reshape(65536, 8).astype(np.int64)
        h = t4[:, 3] > 0
        if h.any():
            z = t4[t4[:, 0] > 0, 0].min()
            print("join kernel: %d waves entered, %d with entries, %d whose lane 0 held a second mate" % ((t4[:, 0] > 0).sum(), h.sum(), (t4[:, 4] > 0).sum()))
            stats("wave entry", t4[h, 0] - z)
            stats("region table forwarded (issue)", t4[h, 1] - t4[h, 0])
            stats("key + region arrived", t4[h, 2] - t4[h, 1])
            stats("first compare-and-swap back", t4[h, 3] - t4[h, 2])
            m = t4[:, 4] > 0
            stats("second mate: partner done", t4[m, 4] - t4[m, 3])
            stats("wave's lane 0 done", np.maximum(t4[h, 3], t4[h, 4]) - z)
    bd.close()


if __name__ == "__main__":
    main()
