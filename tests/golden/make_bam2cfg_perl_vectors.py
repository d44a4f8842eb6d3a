#!/usr/bin/env python3
"""Golden vectors for bin/bam2cfg from the parts of the reference's Perl that RUN in this image.

perl/bam2cfg.pl as a whole cannot run here: its `use Statistics::Descriptive` / `use GD::Graph::histogram` lines name CPAN
modules the image lacks and its records come from a `samtools view` pipe.  Two pieces of it can, and they are what decide a
record's fate and the normality column:

  * perl/AlnParser.pm (no dependencies): `AlnParser::in` is run, unchanged, on the SAM text of every record -- the pair
    orientation code ("flag": 0 1 2 4 8 18 20 32 64 192), the quality it goes by (MAPQ or AM:i / Aq:i), read length,
    insert size, read group;
  * the numeric subs of bam2cfg.pl (ShapiroWilk with ppnd / alnorm / poly_, :284-770), evaluated from the script's text.

tests/golden/bam2cfg_reference_driver.pl does both (reading /root/reference at run time; nothing of it is stored here).
What is left is the script's top-level loop (:48-262): its per-record bookkeeping is restated below (`collect`), fed with the
Perl module's per-record results, and mean / standard deviation follow Statistics::Descriptive's definitions (n - 1).
The SAM text is rendered from an independent BAM decode (tests/helpers.read_bam) in the column layout of `samtools view`.

So bin/bam2cfg is pinned on the reference's own classification of every record and on its own Shapiro-Wilk figure; the loop
around them stays pinned on its rules only.

    python tests/golden/make_bam2cfg_perl_vectors.py      # rewrites tests/golden/bam2cfg_perl_vectors.json
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import read_bam  # noqa: E402

REF_PERL = "/root/reference/perl"
DRIVER = os.path.join(HERE, "bam2cfg_reference_driver.pl")
_BASES = "=ACMGRSVTWYHKDBN"


def sam_text(path):
    """(header text, list of SAM record lines) in `samtools view -h` layout: the eleven columns, then the integer and string tags"""
    targets, r = read_bam(path, keep_all=True)
    lines = []
    for i in range(len(r["tid"])):
        tid, mtid = int(r["tid"][i]), int(r["mtid"][i])
        L = int(r["qlen"][i])
        seq = "".join(_BASES[(r["seq"][i][k >> 1] >> (0 if k & 1 else 4)) & 15] for k in range(L)) or "*"
        qual = "".join(chr(min(q, 93) + 33) for q in r["qual"][i]) if L and r["qual"][i][:1] != b"\xff" else "*"
        rnext = "*" if mtid < 0 else ("=" if mtid == tid else targets[mtid])
        cols = [r["name"][i], str(int(r["flag"][i])), targets[tid] if tid >= 0 else "*", str(int(r["pos"][i]) + 1), str(int(r["mapq"][i])),
                ("%dM" % L) if L else "*", rnext, str(int(r["mpos"][i]) + 1), str(int(r["isize"][i])), seq, qual]
        for tag, (ty, val) in r["aux"][i].items():
            cols.append("%s:%s:%s" % (tag.decode(), "Z" if ty == b"Z" else ("H" if ty == b"H" else "i"), val))
        lines.append("\t".join(cols))
    return r["header"], lines


def perl_records(header, lines, alt):
    """AlnParser::in of the reference on every record line -> list of dicts"""
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as f:
        f.write(header if header.endswith("\n") or not header else header + "\n")
        f.write("\n".join(lines) + "\n")
    try:
        out = subprocess.run(["perl", DRIVER, REF_PERL, "aln", "1" if alt else "0", f.name], check=True, capture_output=True, text=True).stdout
    finally:
        os.unlink(f.name)
    recs = []
    for ln in out.splitlines():
        flag, qual, readlen, ori, dist, rg = ln.split("\t")
        recs.append(dict(flag=int(flag), qual=int(qual), readlen=int(readlen), ori=ori, dist=int(dist), rg=None if rg == "-" else rg))
    assert len(recs) == len(lines)
    return recs


def perl_shapiro_wilk(vectors):
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for v in vectors:
            f.write(" ".join("%d" % x for x in v) + "\n")
    try:
        out = subprocess.run(["perl", DRIVER, REF_PERL, "sw", f.name], check=True, capture_output=True, text=True).stdout
    finally:
        os.unlink(f.name)
    return [ln.split("\t")[1] for ln in out.splitlines()]


def collect(header, t, q=35, n=10000, c=4.0, s=50.0, v=1.0):
    """bam2cfg.pl:48-262 around the per-record results `t` of AlnParser::in: which records reach the insert-size and read-length
    samples and the flag histogram, the early exits, the trim and the gates.  Returns rows in @RG header order."""
    rg_lib, rg_pl, rg_order, libs = {}, {}, [], {}
    for line in header.split("\n"):
        if line.startswith("@RG"):
            f = dict(x.split(":", 1) for x in line.split("\t")[1:] if ":" in x)
            if f["ID"] not in rg_lib:
                rg_order.append(f["ID"])
            rg_lib[f["ID"]] = f.get("LB", "")
            rg_pl[f["ID"]] = f.get("PL", "")
            libs[f.get("LB", "")] = True
    ins, rl, libpos, hist = {}, {}, {}, {}
    counter, expected = 0, 0
    for r in t:
        active = [k for k, on in libs.items() if on]
        if not active:
            if ins:
                break
            libs["NA"] = True; rg_lib["NA"] = "NA"; rg_pl["NA"] = "illumina"; rg_order.append("NA"); active = ["NA"]
        if expected <= 0:
            expected = 3 * len(active) * n
        if counter > expected:
            break
        lib = rg_lib.get(r["rg"]) if r["rg"] is not None else "NA"
        if lib is None or not libs.get(lib):
            continue
        rl.setdefault(lib, []).append(r["readlen"])
        if r["qual"] <= q:
            continue
        counter += 1
        libpos[lib] = libpos.get(lib, 0) + 1
        if r["rg"] is not None:
            h = hist.setdefault(r["rg"], {})
            h[r["flag"]] = h.get(r["flag"], 0) + 1
        nreads = len(ins[lib]) if lib in ins else 1
        if nreads / libpos[lib] < 1e-4:
            libs[lib] = False; ins.pop(lib, None)
        if r["flag"] not in (18, 20) or r["dist"] < 0:
            continue
        ins.setdefault(lib, []).append(float(r["dist"]))
        if len(ins[lib]) > n:
            libs[lib] = False
    fin = {}
    for lib, x in ins.items():
        x = np.array(x)
        m, sd = x.mean(), x.std(ddof=1)
        x = x[~(x > m + 5 * sd)]
        m, sd = x.mean(), x.std(ddof=1)
        if m < s or sd / m >= v or len(x) < 100:
            continue
        up, lo = x[x > m], x[x <= m]
        sp = np.sqrt(((up - m) ** 2).sum() / (len(up) - 1)); sm = np.sqrt(((lo - m) ** 2).sum() / (len(lo) - 1))
        fin[lib] = dict(num=len(x), mean=m, std=sd, lower=max(0.0, m - c * sm), upper=m + c * sp, readlen=float(np.mean(rl[lib])), data=x)
    libs_out = [lib for lib in fin]
    sw = dict(zip(libs_out, perl_shapiro_wilk([fin[lib]["data"] for lib in libs_out]))) if libs_out else {}
    rows = []
    for rg in rg_order:
        lib = rg_lib[rg]
        if lib not in fin:
            continue
        e = fin[lib]
        h = hist.get(rg, {})
        total = sum(h.values())
        flagtext = "".join("%d(%.2f%%)" % (int(k), h[int(k)] * 100 / total) for k in sorted(str(k) for k in h)) + "%d" % total if total else None
        rows.append(dict(readgroup=rg, platform=rg_pl[rg] or "illumina", lib=lib, num=str(int(e["num"])), SWnormality=sw[lib], flag=flagtext,
                         **{k: "%.2f" % e[k] for k in ("readlen", "lower", "upper", "mean", "std")}))
    return rows


def orientation_records():
    """one library: 400 ordinary pairs for the statistics, then every combination of the flag bits AlnParser looks at, with the mate on
    either side, on the same or another chromosome, with and without an AM tag"""
    rng = np.random.default_rng(11)
    recs = []
    pos = 1000
    for i in range(400):
        ins = int(max(150, rng.normal(320, 25)))
        pos += int(rng.integers(1, 40))
        recs.append(dict(tid=0, pos=pos, mtid=0, mpos=pos + ins - 100, isize=ins, flag=0x1 | 0x2 | 0x20 | 0x40, qlen=100, mapq=60, name="p%d" % i, rg="rgO"))
    k = 0
    for bits in range(1 << 8):
        flag = 0
        for j, b in enumerate((0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x400)):
            if bits >> j & 1:
                flag |= b
        for side in (-1, 0, 1):
            for other in (0, 1):
                pos += 3
                k += 1
                isz = 250 * side if not other else 0
                recs.append(dict(tid=0, pos=pos, mtid=1 if other else 0, mpos=pos + 250 * side, isize=isz, flag=flag, qlen=50 + k % 3, mapq=60 if k % 5 else 30,
                                 am=(40 if k % 7 == 0 else (10 if k % 7 == 1 else None)), name="o%d" % k, rg="rgO" if k % 9 else ""))
    return recs, [("rgO", "libO", "ILLUMINA")]


def cases():
    from breakdancer_amd.bamwrite import write_bam_records
    from make_bam2cfg_vectors import two_library_records
    gd = os.path.join(HERE, "chr21")
    td = tempfile.mkdtemp(prefix="bam2cfg_perl_")
    recs, rgs = two_library_records()
    two = os.path.join(td, "two.bam")
    write_bam_records(two, recs, ["c1"], rgs=rgs)
    recs, rgs = orientation_records()
    ori = os.path.join(td, "orientations.bam")
    write_bam_records(ori, recs, ["c1", "c2"], rgs=rgs)
    return [
        ("NA19240_chr21_del_inv.bam", os.path.join(gd, "NA19240_chr21_del_inv.bam"), [[], ["-q", "20", "-c", "3", "-n", "1200"], ["-m"]]),
        ("NA19238_chr21_del_inv.bam", os.path.join(gd, "NA19238_chr21_del_inv.bam"), [[], ["-q", "20", "-c", "3", "-n", "1200"], ["-m"]]),
        ("two_libraries_synthetic", two, [["-n", "1500"]]),
        ("orientations_synthetic", ori, [["-q", "0"], ["-q", "35"], ["-m", "-q", "35"]]),
    ]


def options(args):
    kw, alt = {}, False
    it = iter(args)
    for a in it:
        if a == "-m":
            alt = True
        else:
            kw[{"-q": "q", "-n": "n", "-c": "c", "-s": "s", "-v": "v"}[a]] = float(next(it)) if a in ("-c", "-s", "-v") else int(next(it))
    return kw, alt


def main():
    out = {}
    for name, path, arg_sets in cases():
        header, lines = sam_text(path)
        per_alt = {}
        out[name] = {}
        for args in arg_sets:
            kw, alt = options(args)
            if alt not in per_alt:
                per_alt[alt] = perl_records(header, lines, alt)
            out[name][" ".join(args + ["-g"])] = collect(header, per_alt[alt], **kw)
        # the reference's own per-record classification, as a histogram over ALL records (what a reader of the fixture can check at a glance)
        codes = {}
        for r in per_alt[False if False in per_alt else True]:
            codes[str(r["flag"])] = codes.get(str(r["flag"]), 0) + 1
        out[name]["_alnparser_flag_counts_all_records"] = codes
    json.dump(out, open(os.path.join(HERE, "bam2cfg_perl_vectors.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "bam2cfg_perl_vectors.json"))


if __name__ == "__main__":
    main()
