"""bin/bam2cfg --device: the records inflated and decoded by the GPU decoder (bdx_bamdec_*, no reader filter, the
quality column as -m asks, the read group's index in the library column) and the per-library sums in a kernel
(bdx_insert_size_stats) -- held to the same vectors as the CPU tool (tests/test_bam2cfg.py: the restated rules of
perl/bam2cfg.pl, and the outputs of the reference's own AlnParser.pm / ShapiroWilk sub), and to the CPU tool's output byte for byte."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT
from test_bam2cfg import BIN, GOLD, PERL_VECTORS, VECTORS, check

pytestmark = pytest.mark.gpu

FIXTURES = ["NA19240_chr21_del_inv.bam", "NA19238_chr21_del_inv.bam"]


def run_raw(paths, *args):
    p = subprocess.run([BIN, *args, *paths], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return p.stdout


def rows_of(text):
    return [dict(f.split(":", 1) for f in line.split("\t")) for line in text.strip().split("\n")]


def same_as_cpu(path, *args):
    dev = run_raw([path], "--device", *args)
    cpu = run_raw([path], *args)
    assert dev == cpu and dev.strip()
    return rows_of(dev)


def check_exact_rows(path, rows, expect):
    assert [r["readgroup"] for r in rows] == [e["readgroup"] for e in expect] and rows
    for row, e in zip(rows, expect):
        assert row["map"] == path and row["exe"] == "samtools view"
        for k in ("platform", "lib", "num", "readlen", "lower", "upper", "mean", "std", "SWnormality", "flag"):
            assert row.get(k) == e[k], (e["readgroup"], k, row.get(k), e[k])


@pytest.mark.parametrize("name", FIXTURES)
def test_chr21_fixtures_on_the_device(name):
    path = os.path.join(GOLD, name)
    check(path, same_as_cpu(path), VECTORS[name]["default"])
    check(path, same_as_cpu(path, "-q", "20", "-c", "3", "-n", "1200"), VECTORS[name]["-q 20 -c 3 -n 1200"])


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("args", ["-g", "-q 20 -c 3 -n 1200 -g", "-m -g"])
def test_chr21_fixtures_on_the_device_against_the_reference_perl(name, args):
    path = os.path.join(GOLD, name)
    check_exact_rows(path, same_as_cpu(path, *args.split()), PERL_VECTORS[name][args])


def test_two_libraries_on_the_device(tmp_path):
    sys.path.insert(0, GOLDEN)
    from make_bam2cfg_vectors import two_library_records
    from breakdancer_amd.bamwrite import write_bam_records
    recs, rgs = two_library_records()
    path = str(tmp_path / "two.bam")
    write_bam_records(path, recs, ["c1"], rgs=rgs)
    check(path, same_as_cpu(path, "-n", "1500"), VECTORS["two_libraries_synthetic"]["-n 1500"])
    check_exact_rows(path, same_as_cpu(path, "-n", "1500", "-g"), PERL_VECTORS["two_libraries_synthetic"]["-n 1500 -g"])


@pytest.mark.parametrize("args", ["-q 0 -g", "-q 35 -g", "-m -q 35 -g"])
def test_every_orientation_code_on_the_device(tmp_path, args):
    """secondary, supplementary, unplaced and duplicate records, AM tags, records without a read group: the decoder runs without its
    reader filter here, and the -g histogram is AlnParser's code by code"""
    sys.path.insert(0, GOLDEN)
    from make_bam2cfg_perl_vectors import orientation_records
    from breakdancer_amd.bamwrite import write_bam_records
    recs, rgs = orientation_records()
    path = str(tmp_path / "orientations.bam")
    write_bam_records(path, recs, ["c1", "c2"], rgs=rgs)
    check_exact_rows(path, same_as_cpu(path, *args.split()), PERL_VECTORS["orientations_synthetic"][args])


def test_the_loop_runs_off_the_first_stretch(tmp_path):
    """a file that opens with more poor-quality records than the first decoded stretch holds (64 members at least): the device path
    decodes four times as much and starts the loop over; and a file without read groups ("NA")"""
    from breakdancer_amd.bamwrite import write_bam_records
    rng = np.random.default_rng(11)
    recs = []
    pos = 100
    for i in range(60000):   # ~280 bytes each inflated: ~250 members of poor quality first
        pos += int(rng.integers(1, 5))
        recs.append(dict(tid=0, pos=pos, mtid=0, mpos=pos + 200, isize=300, flag=0x1 | 0x2 | 0x20 | 0x40, qlen=100, mapq=3, name="l%d" % i, rg="rgA"))
    for i in range(3000):
        pos += int(rng.integers(1, 9))
        ins = int(max(150, rng.normal(320, 25)))
        recs.append(dict(tid=0, pos=pos, mtid=0, mpos=pos + ins - 100, isize=ins, flag=0x1 | 0x2 | 0x20 | 0x40, qlen=100, mapq=60, name="g%d" % i,
                         rg="rgA" if i % 3 else "rgB"))
    path = str(tmp_path / "late.bam")
    write_bam_records(path, recs, ["c1"], rgs=[("rgA", "libA", "illumina"), ("rgB", "libB", "illumina")])
    rows = same_as_cpu(path, "-n", "500", "-g")
    assert [r["lib"] for r in rows] == ["libA", "libB"] and all(abs(float(r["mean"]) - 320) < 6 for r in rows)
    for r in recs:
        r["rg"] = ""
    path = str(tmp_path / "norg.bam")
    write_bam_records(path, recs, ["c1"])
    rows = same_as_cpu(path, "-n", "800")
    assert [r["readgroup"] for r in rows] == ["NA"] and rows[0]["lib"] == "NA"


def test_a_header_of_several_bgzf_members(tmp_path):
    """8,000 reference sequences: the BAM header spans several BGZF members and the first record lies in the middle of one -- the device
    path measures the header itself (magic, l_text, text, n_ref, names) to tell the decoder where the records start"""
    sys.path.insert(0, GOLDEN)
    from make_bam2cfg_vectors import two_library_records
    from breakdancer_amd.bamwrite import write_bam_records
    recs, rgs = two_library_records()
    targets = ["contig_%05d_of_a_fragmented_assembly" % i for i in range(8000)]
    for r in recs[len(recs) // 2:]:
        r["tid"] = r["mtid"] = 7999
    path = str(tmp_path / "many.bam")
    write_bam_records(path, recs, targets, rgs=rgs)
    assert os.path.getsize(path) > 100_000
    rows = same_as_cpu(path, "-n", "1500", "-g")
    assert [r["lib"] for r in rows] == ["libA", "libB"]


def test_insert_size_stats_kernel_bit_for_bit():
    """bdx_insert_size_stats against the same sums in numpy float64 scalar arithmetic, in the script's order (perl/bam2cfg.pl:153-197)"""
    from breakdancer_amd._lib import load

    class Stats(ctypes.Structure):
        _fields_ = [(k, ctypes.c_double) for k in ("mean_all", "sd_all", "mean", "sd", "sd_minus", "sd_plus")] + \
                   [(k, ctypes.c_uint64) for k in ("n_kept", "n_minus", "n_plus")]
    lib = load()
    rng = np.random.default_rng(3)
    sizes = [1, 2, 150, 10001, 4097]
    xs = [np.floor(np.abs(rng.normal(300 + 40 * i, 30 + 5 * i, n))).astype(np.float64) for i, n in enumerate(sizes)]
    xs[2][::17] = 9000.0
    xs[3][5] = 1e6
    x = np.concatenate(xs)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    out = (Stats * len(sizes))()
    lib.bdx_insert_size_stats.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert lib.bdx_insert_size_stats(0, x.ctypes.data, off.ctypes.data, len(sizes), ctypes.addressof(out)) == 0

    def seq_sum(v):
        s = np.float64(0)
        for t in v:
            s = s + t
        return s

    def sd_of(v, m):
        return np.sqrt(seq_sum((v - m) * (v - m)) / np.float64(len(v) - 1)) if len(v) >= 2 else np.float64(0)
    for v, o in zip(xs, out):
        m0 = seq_sum(v) / np.float64(len(v))
        s0 = sd_of(v, m0)
        kept = v[~(v > m0 + np.float64(5) * s0)]
        m = seq_sum(kept) / np.float64(len(kept))
        s = sd_of(kept, m)
        assert (o.mean_all, o.sd_all, o.mean, o.sd, o.n_kept) == (m0, s0, m, s, len(kept))
        up, dn = kept[kept > m], kept[~(kept > m)]
        assert (o.n_plus, o.n_minus) == (len(up), len(dn))
        with np.errstate(all="ignore"):
            for got, part in ((o.sd_plus, up), (o.sd_minus, dn)):
                want = np.sqrt(seq_sum((part - m) * (part - m)) / np.float64(len(part) - 1))
                assert got == want or (np.isnan(got) and np.isnan(want))
