"""Seeded generator of small, messy multi-BAM inputs for differential tests (oracle vs product).
Produces per-BAM record streams + a bam2cfg-format config.  Shape follows SURVEY.md B.14."""
import numpy as np


def make_case(seed, n_pairs=None, contigs=(30000, 30000, 20000)):
    rng = np.random.default_rng(seed)
    if n_pairs is None:
        n_pairs = int(rng.integers(300, 1500))
    # 3 read groups -> 3 libraries over 2 BAMs
    libs = [dict(name="libA", bam="a.bam", rg="rg1", mean=300 + int(rng.integers(0, 100)), std=25 + int(rng.integers(0, 15))),
            dict(name="libB", bam="a.bam", rg="rg2", mean=350 + int(rng.integers(0, 100)), std=30),
            dict(name="libC", bam="b.bam", rg="rg3", mean=250 + int(rng.integers(0, 100)), std=20)]
    lines = []
    for l in libs:
        l["readlen"] = int(rng.choice([50, 75, 100]))
        f = ["readgroup:%s" % l["rg"], "platform:illumina", "map:%s" % l["bam"], "readlen:%d.00" % l["readlen"],
             "lib:%s" % l["name"], "num:10001"]
        if rng.random() < 0.7:
            f += ["lower:%.2f" % (l["mean"] - 3 * l["std"]), "upper:%.2f" % (l["mean"] + 3 * l["std"])]
        f += ["mean:%.2f" % l["mean"], "std:%.2f" % l["std"]]
        if rng.random() < 0.3:
            f.append("mapqual:%d" % int(rng.choice([10, 30, 36])))
        lines.append("\t".join(f))
    config = "\n".join(lines) + "\n"
    bam_index = {"a.bam": 0, "b.bam": 1}
    centres = [rng.integers(2000, L - 3000, int(rng.integers(3, 13))) for L in contigs]
    recs = {0: [], 1: []}
    kinds = np.array(["normal", "large", "small", "ff", "rr", "rf", "ctx", "mate_unmapped", "dup", "single"])
    probs = np.array([60, 8, 5, 5, 3, 5, 6, 2, 3, 1], dtype=float)
    probs /= probs.sum()
    for pid in range(n_pairs):
        kind = str(rng.choice(kinds, p=probs))
        li = int(rng.integers(0, 3))
        l = libs[li]
        rl = l["readlen"]
        tid = int(rng.integers(0, len(contigs)))
        anomalous = kind not in ("normal", "dup")
        c = int(rng.choice(centres[tid]))
        p1 = c + int(rng.integers(-150, 150)) if anomalous else int(rng.integers(1000, contigs[tid] - 3000))
        mtid = tid
        if kind == "normal" or kind == "dup" or kind == "mate_unmapped" or kind == "single":
            ins = int(max(rl + 10, rng.normal(l["mean"], l["std"])))
        elif kind == "large":
            ins = int(l["mean"] + l["std"] * rng.uniform(4, 30))
        elif kind == "small":
            ins = int(max(rl + 1, l["mean"] - l["std"] * rng.uniform(4, 8)))
        else:
            ins = int(rng.integers(rl + 50, 1500))
        p2 = p1 + ins - rl
        if kind == "ctx":
            mtid = int((tid + 1 + rng.integers(0, len(contigs) - 1)) % len(contigs))
            p2 = int(rng.choice(centres[mtid])) + int(rng.integers(-150, 150))
        p1 = max(p1, 1)
        p2 = max(p2, 1)
        r1, r2 = False, True
        if kind == "ff":
            r1, r2 = False, False
        elif kind == "rr":
            r1, r2 = True, True
        elif kind == "rf":
            r1, r2 = True, False
        mqs = [60, 37, 36, 35, 20, 0]
        q1 = int(rng.choice(mqs, p=[0.6, 0.1, 0.1, 0.05, 0.1, 0.05]))
        q2 = q1 if rng.random() < 0.8 else int(rng.choice(mqs))
        am = rng.random() < 0.5  # AM tag present: bdqual = min of the two
        bq1, bq2 = (min(q1, q2), min(q1, q2)) if am else (q1, q2)
        proper = kind == "normal" or (anomalous and rng.random() < 0.15)
        u = rng.random()
        rg = l["rg"] if u > 0.06 else ("" if u > 0.03 else "rgUnknown")
        bam = bam_index[l["bam"]]
        qlen1, qlen2 = rl, rl if rng.random() < 0.9 else int(rng.choice([50, 75, 100]))
        base = 0x1 | (0x2 if proper else 0) | (0x400 if kind == "dup" else 0)
        isz = (p2 + qlen2 - p1) if mtid == tid else 0
        f1 = base | 0x40 | (0x10 if r1 else 0) | (0x20 if r2 else 0) | (0x8 if kind == "mate_unmapped" else 0)
        f2 = base | 0x80 | (0x10 if r2 else 0) | (0x20 if r1 else 0) | (0x4 if kind == "mate_unmapped" else 0)
        a = dict(tid=tid, pos=p1, mtid=mtid, mpos=p2, isize=isz, flag=f1, qlen=qlen1, bdqual=bq1, rg=rg, name=pid + 1)
        b = dict(tid=mtid, pos=p2, mtid=tid, mpos=p1, isize=-isz, flag=f2, qlen=qlen2, bdqual=bq2, rg=rg, name=pid + 1)
        if kind == "mate_unmapped":  # unmapped mate is placed at its mate's position (samtools convention)
            b["tid"], b["pos"] = tid, p1
            a["mtid"], a["mpos"] = tid, p1
        recs[bam].append(a)
        if kind != "single":
            recs[bam].append(b)
        if rng.random() < 0.01:  # a secondary record, dropped by the reader filter -> never reaches the streams
            pass
    streams = []
    for b in (0, 1):
        rr = sorted(recs[b], key=lambda r: (r["tid"], r["pos"]))
        d = {k: np.array([r[k] for r in rr], dtype=dt) for k, dt in
             (("tid", np.int32), ("pos", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32),
              ("flag", np.uint16), ("qlen", np.int32), ("bdqual", np.uint8))}
        d["rg"] = [r["rg"] for r in rr]
        d["name_id"] = np.array([r["name"] for r in rr], dtype=np.uint64)
        streams.append(d)
    targets = ["c%d" % (i + 1) for i in range(len(contigs))]
    return config, streams, targets


OPTION_SETS = [dict(), dict(cn_lib=1, print_af=1), dict(print_af=1), dict(transchr_rearrange=1), dict(chr_tid=0),
               dict(chr_tid=1, cn_lib=1), dict(min_read_pair=1), dict(min_read_pair=3), dict(min_len=0), dict(min_len=60),
               dict(buffer_size=1), dict(buffer_size=2), dict(buffer_size=5), dict(min_map_qual=0), dict(min_map_qual=36),
               dict(seq_coverage_lim=3), dict(max_sd=900), dict(cut_sd=2), dict(illumina_long_insert=1), dict(fisher=1),
               dict(transchr_rearrange=1, min_read_pair=1)]
