"""Throughput probe of the device-side BAM decode on configs[1]-shaped data (tools, not the product):
  * the inflate kernel alone (bdx_inflate_blocks: members already in HBM, kernel time by HIP events),
  * the whole decode of a file into columns (read + pinned copy + H2D + inflate + record kernels), Python feeder.
usage: python tools/bamdec_probe.py [--mbp 10] [--level 1] [--piece-blocks 512] [--realistic] [--check-zlib]
  --realistic   bases from a shared random reference, binned qualities, level 6 (bamwrite.Reference): the token mix of a real 30x BAM
  --check-zlib  every inflated byte of the kernel-alone runs against zlib's (threads)"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=10.0)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--piece-blocks", type=int, default=512)
    ap.add_argument("--bam", default=None, help="decode this file instead of a synthetic one")
    ap.add_argument("--inflate-only", action="store_true")
    ap.add_argument("--realistic", action="store_true")
    ap.add_argument("--check-zlib", action="store_true")
    ap.add_argument("--slice-gb", type=float, default=1.5, help="inflated bytes per bdx_inflate_blocks call (the kernel-alone measurement)")
    a = ap.parse_args()
    from breakdancer_amd import bamdec
    from breakdancer_amd.bamwrite import write_bam
    from breakdancer_amd.synth import make_chromosome
    with tempfile.TemporaryDirectory(prefix="bdx_probe_") as td:
        if a.bam:
            bam = a.bam
        else:
            d = make_chromosome(length=int(a.mbp * 1e6), seed=1)
            bam = os.path.join(td, "syn.bam")
            write_bam(bam, d, ["chrS"], seed=3, level=6 if a.realistic else a.level, realistic=a.realistic)
        image = np.fromfile(bam, dtype=np.uint8)
        members = bamdec.scan_bgzf(image)
        data = members[members["inflated_len"] > 0]
        ulen = int(data["inflated_len"].astype(np.int64).sum())
        print("file %.1f MB, %d members, %.1f MB inflated (ratio %.2f)" % (image.size / 1e6, len(data), ulen / 1e6, ulen / image.size))
        # kernel alone, in slices of <= --slice-gb of output
        tot_ms, done = 0.0, 0
        i = 0
        while i < len(data):
            j = i
            acc = 0
            while j < len(data) and acc < int(a.slice_gb * 1e9):
                acc += int(data["inflated_len"][j])
                j += 1
            lo = int(data["member"][i])
            hi = int(data["payload"][j - 1]) + int(data["payload_len"][j - 1]) + 8
            sl = data[i:j].copy()
            sl["payload"] -= np.uint64(lo)
            sl["member"] -= np.uint64(lo)
            for rep in range(2):
                out, status, ms = bamdec.inflate_blocks(image[lo:hi], sl)
            assert not status.any()
            if a.check_zlib:
                import zlib
                from concurrent.futures import ThreadPoolExecutor
                raw = image[lo:hi].tobytes()

                def ref(ab):
                    return b"".join(zlib.decompress(raw[int(m["payload"]):int(m["payload"]) + int(m["payload_len"])], -15) for m in sl[ab[0]:ab[1]])
                cuts = list(range(0, len(sl), 256)) + [len(sl)]
                with ThreadPoolExecutor(max_workers=16) as ex:
                    want = b"".join(ex.map(ref, zip(cuts[:-1], cuts[1:])))
                same = out.tobytes() == want
                print("slice of %d members, %.1f MB inflated: byte-exact against zlib: %s" % (len(sl), len(want) / 1e6, same))
                assert same
            pf = os.environ.get("BDX_KZ_PROF")
            if pf and os.path.exists(pf):
                q = np.fromfile(pf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
                c0 = q[:, 0]
                print("kernel clocks per member: min %.0f, 10 %% %.0f, median %.0f, 90 %% %.0f, 99 %% %.0f, max %.0f cycles; steps min %.0f max %.0f; by launch order (eighths of the members, mean cycles): %s" %
                      (c0.min(), np.percentile(c0, 10), np.median(c0), np.percentile(c0, 90), np.percentile(c0, 99), c0.max(), q[:, 2].min(), q[:, 2].max(),
                       " ".join("%.1fM" % (x.mean() / 1e6) for x in np.array_split(c0, 8))))
                print("kernel clocks per member (mean): cycles %.0f, of which headers+tables %.0f; steps %.0f, matches %.0f, slow codes %.0f, deflate blocks %.1f; "
                      "cycles per step %.0f" % (q[:, 0].mean(), q[:, 1].mean(), q[:, 2].mean(), q[:, 3].mean(), q[:, 4].mean(), q[:, 5].mean(),
                                                 (q[:, 0] - q[:, 1]).sum() / q[:, 2].sum()))
                print("matches by where their source came from (sum over the slice): %d in all; %d far ones of <= 8 bytes fetched from HBM by their own lanes (%.1f %%), "
                      "%d far ones copied from HBM by the wave (%.1f %%), the rest (%.1f %%) out of the LDS window" %
                      (q[:, 3].sum(), q[:, 6].sum(), 100 * q[:, 6].sum() / max(1, q[:, 3].sum()), q[:, 7].sum(), 100 * q[:, 7].sum() / max(1, q[:, 3].sum()),
                       100 * (q[:, 3].sum() - q[:, 6].sum() - q[:, 7].sum()) / max(1, q[:, 3].sum())))
            tot_ms += ms
            done += acc
            i = j
        print("inflate kernel: %.2f ms for %.1f MB -> %.1f GB/s inflated, %.1f GB/s compressed" %
              (tot_ms, done / 1e6, done / tot_ms / 1e6, image.size / tot_ms / 1e6))
        for rep in range(0 if a.inflate_only else 3):
            t0 = time.perf_counter()
            cols, names, stats = bamdec.decode_file(bam, rg_ids=["rg1"], rg_lib=[0], piece_blocks=a.piece_blocks)
            dt = time.perf_counter() - t0
            print("decode_file: %.3f s, %d records, %.2f GB/s of file, %s" % (dt, len(cols["tid"]), image.size / dt / 1e9, stats))


if __name__ == "__main__":
    main()
