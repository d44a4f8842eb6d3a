"""bin/bam2cfg against committed golden vectors (tests/golden/bam2cfg_vectors.json, written by
tests/golden/make_bam2cfg_vectors.py from the documented rules of perl/bam2cfg.pl and AlnParser.pm), on the reference's
chr21 fixtures and on a synthetic BAM that reaches the script's early exits.  The reference's own golden config
(test-data/inv_del_bam_config) was made from the full BAMs, of which the fixtures are excerpts, so it pins the format and
the plausibility of the figures, not their digits; the Perl script as a whole cannot run here (no samtools, no
Statistics::Descriptive), but its record classifier (AlnParser.pm) and its Shapiro-Wilk sub can: the second half of this file
holds the tool to their outputs, column by column (tests/golden/make_bam2cfg_perl_vectors.py).  These are the tool's CPU source and
sums; `--device` (records decoded and statistics summed on the GPU) is held to the same vectors in tests/test_gpu_bam2cfg.py."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

GOLD = os.path.join(GOLDEN, "chr21")
BIN = os.path.join(ROOT, "bin", "bam2cfg")
VECTORS = json.load(open(os.path.join(GOLDEN, "bam2cfg_vectors.json")))


def run_tool(paths, *args):
    p = subprocess.run([BIN, *args, *paths], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rows = []
    for line in p.stdout.strip().split("\n"):
        rows.append(dict(f.split(":", 1) for f in line.split("\t")))
    return rows


def check(path, rows, expect):
    assert len(rows) == len(expect) and rows
    for row, e in zip(rows, expect):
        assert row["readgroup"] == e["readgroup"] and row["platform"] == e["platform"] and row["lib"] == e["lib"] and row["map"] == path
        assert row["exe"] == "samtools view"
        for k in ("readlen", "lower", "upper", "mean", "std"):
            assert row[k] == e[k], (k, row[k], e[k])
        assert int(row["num"]) == e["num"]
        if e["sw_log10_p"] is not None:  # scipy's Shapiro-Wilk p-value (the tool carries AS R94 like the script)
            assert abs(float(row["SWnormality"]) - e["sw_log10_p"]) < 0.06, (row["SWnormality"], e["sw_log10_p"])
        else:  # above 5000 observations the approximation underflows for these shapes, as in the reference's own config
            assert row["SWnormality"] in ("minus infinity",) or float(row["SWnormality"]) < 0


@pytest.mark.parametrize("name", ["NA19240_chr21_del_inv.bam", "NA19238_chr21_del_inv.bam"])
def test_chr21_fixtures(name):
    path = os.path.join(GOLD, name)
    rows = run_tool([path])
    check(path, rows, VECTORS[name]["default"])
    assert len(rows) == 7 and all(r["readlen"] == "90.00" for r in rows)  # as in the reference's golden config
    golden = {}
    for line in open(os.path.join(GOLD, "inv_del_bam_config")):
        f = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t"))
        golden[f["readgroup"]] = f
    for r in rows:  # the excerpts give figures close to the full files'
        g = golden[r["readgroup"]]
        assert r["lib"] == g["lib"] and r["platform"] == g["platform"]
        assert abs(float(r["mean"]) - float(g["mean"])) < 15 and abs(float(r["upper"]) - float(g["upper"])) < 40
    check(path, run_tool([path], "-q", "20", "-c", "3", "-n", "1200"), VECTORS[name]["-q 20 -c 3 -n 1200"])


def test_generated_config_is_accepted_by_the_config_parser():
    from helpers import OracleRun, make_opts
    paths = [os.path.join(GOLD, n) for n in ("NA19240_chr21_del_inv.bam", "NA19238_chr21_del_inv.bam")]
    p = subprocess.run([BIN, *paths], capture_output=True, text=True)
    run = OracleRun(p.stdout, make_opts())
    assert run.nlibs == 2 and run.nbams == 2


def test_early_exit_two_libraries_and_quality_gate(tmp_path):
    """two libraries in one file: collection stops at n + 1 pairs per library; low-quality and improper reads are ignored"""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_bam2cfg_vectors import two_library_records
    from breakdancer_amd.bamwrite import write_bam_records
    recs, rgs = two_library_records()
    path = str(tmp_path / "two.bam")
    write_bam_records(path, recs, ["c1"], rgs=rgs)
    rows = run_tool([path], "-n", "1500")
    check(path, rows, VECTORS["two_libraries_synthetic"]["-n 1500"])
    assert [r["lib"] for r in rows] == ["libA", "libB"]
    assert all(1480 <= int(r["num"]) <= 1501 for r in rows)
    assert abs(float(rows[0]["mean"]) - 300) < 3 and abs(float(rows[1]["mean"]) - 450) < 4


# ---- pinned on the parts of the reference's Perl that run (tests/golden/make_bam2cfg_perl_vectors.py): AlnParser::in of
# perl/AlnParser.pm on every record (orientation code, quality, read length, insert size, read group) and the ShapiroWilk sub
# of perl/bam2cfg.pl:284-770 on the collected insert sizes -- every column of the output line, the -g histogram included ----
PERL_VECTORS = json.load(open(os.path.join(GOLDEN, "bam2cfg_perl_vectors.json")))


def check_exact(path, args, expect):
    rows = run_tool([path], *args)
    assert [r["readgroup"] for r in rows] == [e["readgroup"] for e in expect] and rows
    for row, e in zip(rows, expect):
        assert row["map"] == path and row["exe"] == "samtools view"
        for k in ("platform", "lib", "num", "readlen", "lower", "upper", "mean", "std", "SWnormality", "flag"):
            assert row.get(k) == e[k], (e["readgroup"], k, row.get(k), e[k])


@pytest.mark.parametrize("name", ["NA19240_chr21_del_inv.bam", "NA19238_chr21_del_inv.bam"])
@pytest.mark.parametrize("args", ["-g", "-q 20 -c 3 -n 1200 -g", "-m -g"])
def test_chr21_fixtures_against_the_reference_perl(name, args):
    check_exact(os.path.join(GOLD, name), args.split(), PERL_VECTORS[name][args])


def test_two_libraries_against_the_reference_perl(tmp_path):
    import sys
    sys.path.insert(0, GOLDEN)
    from make_bam2cfg_vectors import two_library_records
    from breakdancer_amd.bamwrite import write_bam_records
    recs, rgs = two_library_records()
    path = str(tmp_path / "two.bam")
    write_bam_records(path, recs, ["c1"], rgs=rgs)
    check_exact(path, ["-n", "1500", "-g"], PERL_VECTORS["two_libraries_synthetic"]["-n 1500 -g"])


@pytest.mark.parametrize("args", ["-q 0 -g", "-q 35 -g", "-m -q 35 -g"])
def test_every_orientation_code_against_alnparser(tmp_path, args):
    """every combination of the flag bits AlnParser.pm:57-126 looks at, the mate on either side / another chromosome, AM tags
    above and below the quality gate, records without a read group: the -g histogram is AlnParser's, code by code"""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_bam2cfg_perl_vectors import orientation_records
    from breakdancer_amd.bamwrite import write_bam_records
    recs, rgs = orientation_records()
    path = str(tmp_path / "orientations.bam")
    write_bam_records(path, recs, ["c1", "c2"], rgs=rgs)
    expect = PERL_VECTORS["orientations_synthetic"][args]
    assert len({c.split("(")[0] for c in expect[0]["flag"].split(")")[:-1]}) == 10   # all ten codes occur
    check_exact(path, args.split(), expect)


def test_device_source_fails_loudly_without_a_gpu():
    """`--device` is the GPU record source and nothing else: without a GPU it says so and prints no configuration (no quiet CPU fallback)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the device source works here (tests/test_gpu_bam2cfg.py)")
    p = subprocess.run([BIN, "--device", os.path.join(GOLD, "NA19240_chr21_del_inv.bam")], capture_output=True, text=True)
    assert p.returncode != 0 and p.stdout == "" and "ERROR" in p.stderr
