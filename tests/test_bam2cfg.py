"""bin/bam2cfg against an independent Python restatement of perl/bam2cfg.pl's estimator, on the
reference's chr21 fixtures and on a synthetic BAM that reaches the script's early exits.  The reference's own golden
config (test-data/inv_del_bam_config) was made from the full BAMs, of which the fixtures are excerpts, so it pins the
format and the plausibility of the figures, not their digits; the Perl script itself cannot run here (no samtools, no
Statistics::Descriptive).  The Shapiro-Wilk figure is cross-checked against scipy."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, read_bam

GOLD = os.path.join(GOLDEN, "chr21")

BIN = os.path.join(ROOT, "bin", "bam2cfg")


def restate(path, q=35, n=10000, c=4.0, s=50.0, v=1.0):
    """perl/bam2cfg.pl:48-262 + AlnParser.pm:38-130 (Illumina rules), record by record"""
    _, r = read_bam(path, keep_all=True)
    rg_lib, rg_order, libs = {}, [], {}
    for line in r["header"].split("\n"):
        if line.startswith("@RG"):
            f = dict(x.split(":", 1) for x in line.split("\t")[1:] if ":" in x)
            if f["ID"] not in rg_lib:
                rg_order.append(f["ID"])
            rg_lib[f["ID"]] = (f.get("LB", ""), f.get("PL", "") or "illumina")
            libs[f.get("LB", "")] = True
    ins, rl, libpos = {}, {}, {}
    counter, expected = 0, 0
    for i in range(len(r["tid"])):
        active = [k for k, on in libs.items() if on]
        if not active:
            if ins:
                break
            libs["NA"] = True; rg_lib["NA"] = ("NA", "illumina"); rg_order.append("NA"); active = ["NA"]
        if expected <= 0:
            expected = 3 * len(active) * n
        if counter > expected:
            break
        rg = r["rg"][i]
        lib = rg_lib[rg][0] if rg else "NA"
        if rg and rg not in rg_lib:
            continue
        if not libs.get(lib):
            continue
        rl.setdefault(lib, []).append(int(r["qlen"][i]) or 1)
        if int(r["bdqual"][i]) <= q:
            continue
        counter += 1
        libpos[lib] = libpos.get(lib, 0) + 1
        fl = int(r["flag"][i])
        code = 0
        if not (fl & 0x400) and (fl & 1):
            if fl & 4: code = 192
            elif fl & 8: code = 64
            elif r["mtid"][i] != r["tid"][i]: code = 32
            elif fl & 2: code = 18 if (r["pos"][i] < r["mpos"][i]) == (not (fl & 0x10)) else 20
            else: code = 1
        nreads = len(ins[lib]) if lib in ins else 1
        if nreads / libpos[lib] < 1e-4:
            libs[lib] = False; ins.pop(lib, None)
        if code not in (18, 20) or r["isize"][i] < 0:
            continue
        ins.setdefault(lib, []).append(float(r["isize"][i]))
        if len(ins[lib]) > n:
            libs[lib] = False
    out = {}
    for lib, x in ins.items():
        x = np.array(x)
        m, sd = x.mean(), x.std(ddof=1)
        x = x[~(x > m + 5 * sd)]
        m, sd = x.mean(), x.std(ddof=1)
        if m < s or sd / m >= v or len(x) < 100:
            continue
        up, lo = x[x > m], x[x <= m]
        sp = np.sqrt(((up - m) ** 2).sum() / (len(up) - 1)); sm = np.sqrt(((lo - m) ** 2).sum() / (len(lo) - 1))
        out[lib] = dict(num=len(x), mean=m, std=sd, lower=max(0.0, m - c * sm), upper=m + c * sp, readlen=float(np.mean(rl[lib])), data=np.sort(x))
    return [(rg, rg_lib[rg][1], rg_lib[rg][0], out[rg_lib[rg][0]]) for rg in rg_order if rg_lib[rg][0] in out]


def run_tool(paths, *args):
    p = subprocess.run([BIN, *args, *paths], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rows = []
    for line in p.stdout.strip().split("\n"):
        rows.append(dict(f.split(":", 1) for f in line.split("\t")))
    return rows


def check(path, rows, expect):
    from scipy import stats
    assert len(rows) == len(expect) and rows
    for row, (rg, pl, lib, e) in zip(rows, expect):
        assert row["readgroup"] == rg and row["platform"] == pl and row["lib"] == lib and row["map"] == path
        assert row["exe"] == "samtools view"
        for k in ("readlen", "lower", "upper", "mean", "std"):
            assert row[k] == "%.2f" % e[k], (k, row[k], e[k])
        assert int(row["num"]) == e["num"]
        if len(e["data"]) <= 5000:
            p = stats.shapiro(e["data"]).pvalue
            if p > 1e-300:
                assert abs(float(row["SWnormality"]) - np.log10(p)) < 0.06, (row["SWnormality"], np.log10(p))
        else:  # above 5000 observations the approximation underflows for these shapes, as in the reference's own config
            assert row["SWnormality"] in ("minus infinity",) or float(row["SWnormality"]) < 0


@pytest.mark.parametrize("name", ["NA19240_chr21_del_inv.bam", "NA19238_chr21_del_inv.bam"])
def test_chr21_fixtures(name):
    path = os.path.join(GOLD, name)
    rows = run_tool([path])
    exp = restate(path)
    check(path, rows, exp)
    assert len(rows) == 7 and all(r["readlen"] == "90.00" for r in rows)  # as in the reference's golden config
    golden = {}
    for line in open(os.path.join(GOLD, "inv_del_bam_config")):
        f = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t"))
        golden[f["readgroup"]] = f
    for r in rows:  # the excerpts give figures close to the full files'
        g = golden[r["readgroup"]]
        assert r["lib"] == g["lib"] and r["platform"] == g["platform"]
        assert abs(float(r["mean"]) - float(g["mean"])) < 15 and abs(float(r["upper"]) - float(g["upper"])) < 40


def test_generated_config_is_accepted_by_the_config_parser():
    from helpers import OracleRun, make_opts
    paths = [os.path.join(GOLD, n) for n in ("NA19240_chr21_del_inv.bam", "NA19238_chr21_del_inv.bam")]
    p = subprocess.run([BIN, *paths], capture_output=True, text=True)
    run = OracleRun(p.stdout, make_opts())
    assert run.nlibs == 2 and run.nbams == 2


def test_early_exit_two_libraries_and_quality_gate(tmp_path):
    """two libraries in one file: collection stops at n + 1 pairs per library; low-quality and improper reads are ignored"""
    from breakdancer_amd.bamwrite import write_bam_records
    rng = np.random.default_rng(5)
    recs = []
    pos = 1000
    for i in range(9000):
        lib = i % 2
        ins = int(max(120, rng.normal(300 if lib == 0 else 450, 20 if lib == 0 else 35)))
        if i % 97 == 0:
            ins = 5000  # an outlier beyond mean + 5 sd
        q = 20 if i % 11 == 0 else 60
        proper = 0 if i % 13 == 0 else 2
        pos += int(rng.integers(1, 30))
        recs.append(dict(tid=0, pos=pos, mtid=0, mpos=pos + ins - 100, isize=ins, flag=0x1 | proper | 0x20 | 0x40, qlen=100, mapq=q,
                         name="p%d" % i, rg="rgA" if lib == 0 else "rgB"))
    path = str(tmp_path / "two.bam")
    write_bam_records(path, recs, ["c1"], rgs=[("rgA", "libA", "illumina"), ("rgB", "libB", "illumina")])
    rows = run_tool([path], "-n", "1500")
    exp = restate(path, n=1500)
    check(path, rows, exp)
    assert [r["lib"] for r in rows] == ["libA", "libB"]
    assert all(1480 <= int(r["num"]) <= 1501 for r in rows)
    assert abs(float(rows[0]["mean"]) - 300) < 3 and abs(float(rows[1]["mean"]) - 450) < 4
