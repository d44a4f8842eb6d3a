#!/usr/bin/env python3
"""Golden vectors for bin/bam2cfg: the figures perl/bam2cfg.pl would print, derived from the script's
documented rules -- AlnParser.pm:38-130 (which records count, the pair orientation codes), bam2cfg.pl:48-262 (per-library
collection with its early exits, outlier trim at mean + 5 sd, one-sided standard deviations for the cutoffs, mean read
length) -- by a record-by-record Python evaluation over an independent BAM decode (tests/helpers.read_bam), plus scipy's
Shapiro-Wilk p-value for the normality column.  The Perl script itself cannot run in this image (Statistics::Descriptive,
GD::Graph and samtools are absent), so these vectors pin the C++ tool on the formulae, not on the script's output:
parity with bam2cfg.pl stays unpinned, and the tool stays a CPU program (a sequential early-exit scan of <= 3 n reads per
library has nothing for the GPU).

    python tests/golden/make_bam2cfg_vectors.py      # rewrites tests/golden/bam2cfg_vectors.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import read_bam  # noqa: E402


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




def two_library_records():
    """the synthetic case of tests/test_bam2cfg.py::test_early_exit_two_libraries_and_quality_gate"""
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
    return recs, [("rgA", "libA", "illumina"), ("rgB", "libB", "illumina")]


def rows_of(path, **kw):
    from scipy import stats
    rows = []
    for rg, pl, lib, e in restate(path, **kw):
        sw = None
        if len(e["data"]) <= 5000:
            p = stats.shapiro(e["data"]).pvalue
            sw = float(np.log10(p)) if p > 1e-300 else None
        rows.append(dict(readgroup=rg, platform=pl, lib=lib, num=int(e["num"]), sw_log10_p=sw,
                         **{k: "%.2f" % e[k] for k in ("readlen", "lower", "upper", "mean", "std")}))
    return rows


def main():
    import tempfile
    from breakdancer_amd.bamwrite import write_bam_records
    gd = os.path.join(HERE, "chr21")
    out = {}
    for name in ("NA19240_chr21_del_inv.bam", "NA19238_chr21_del_inv.bam"):
        out[name] = {"default": rows_of(os.path.join(gd, name)),
                     "-q 20 -c 3 -n 1200": rows_of(os.path.join(gd, name), q=20, c=3.0, n=1200)}
    with tempfile.TemporaryDirectory() as td:
        recs, rgs = two_library_records()
        path = os.path.join(td, "two.bam")
        write_bam_records(path, recs, ["c1"], rgs=rgs)
        out["two_libraries_synthetic"] = {"-n 1500": rows_of(path, n=1500)}
    json.dump(out, open(os.path.join(HERE, "bam2cfg_vectors.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "bam2cfg_vectors.json"))


if __name__ == "__main__":
    main()
