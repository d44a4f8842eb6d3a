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


def make_graph_case(seed, n_slots=240, sizes=(1, 2, 2, 3, 3, 3, 4, 4, 5)):
    """Many small, separate components of the region graph with every shape the walk distinguishes: "slots" 3 kb apart
    (far beyond the window) each become one region; groups of 1-5 (or `sizes`) consecutive slots get random connections of 1-4 pairs
    (around the -r gate), random self groups, mixed flags / libraries, some reaching into the other contig (CTX).
    Returns (config, streams, targets) like make_case."""
    rng = np.random.default_rng(10_000 + seed)
    libs = [dict(name="libA", bam="a.bam", rg="rgA", mean=300, std=20, readlen=50),
            dict(name="libB", bam="b.bam", rg="rgB", mean=380, std=25, readlen=50)]
    config = "".join("readgroup:%s\tplatform:illumina\tmap:%s\treadlen:%d.00\tlib:%s\tnum:10001\tlower:%.2f\tupper:%.2f\tmean:%.2f\tstd:%.2f\n"
                     % (l["rg"], l["bam"], l["readlen"], l["name"], l["mean"] - 3 * l["std"], l["mean"] + 3 * l["std"], l["mean"], l["std"])
                     for l in libs)
    contigs = (n_slots * 3000 // 2 + 20000, n_slots * 3000 // 2 + 20000)
    slot_tid = [0 if i < n_slots // 2 else 1 for i in range(n_slots)]
    slot_pos = [5000 + (i if i < n_slots // 2 else i - n_slots // 2) * 3000 for i in range(n_slots)]
    recs = {0: [], 1: []}
    pid = [0]

    def add_pair(sa, sb, kind, li):
        """one read pair: first mate in slot sa, second in slot sb (sa <= sb in stream order)"""
        l = libs[li]
        rl = l["readlen"]
        ta, tb = slot_tid[sa], slot_tid[sb]
        pa = slot_pos[sa] + int(rng.integers(0, 150))
        pb = slot_pos[sb] + int(rng.integers(0, 150))
        if sa == sb:
            if kind == "small":
                pb = pa + int(rng.integers(8, 60))
            else:
                pb = pa + int(rng.integers(8, 140))
        ra, rb = {"fr": (False, True), "small": (False, True), "ff": (False, False), "rr": (True, True), "rf": (True, False)}[kind]
        isz = (pb + rl - pa) if ta == tb else 0
        pid[0] += 1
        base = 0x1
        fa = base | 0x40 | (0x10 if ra else 0) | (0x20 if rb else 0)
        fb = base | 0x80 | (0x10 if rb else 0) | (0x20 if ra else 0)
        bam = 0 if l["bam"] == "a.bam" else 1
        recs[bam].append(dict(tid=ta, pos=pa, mtid=tb, mpos=pb, isize=isz, flag=fa, qlen=rl, bdqual=60, rg=l["rg"], name=pid[0]))
        recs[bam].append(dict(tid=tb, pos=pb, mtid=ta, mpos=pa, isize=-isz, flag=fb, qlen=rl, bdqual=60, rg=l["rg"], name=pid[0]))

    # background: properly paired normal reads so that the counters, densities and copy numbers are non-trivial
    for _ in range(600):
        li = int(rng.integers(0, 2))
        l = libs[li]
        tid = int(rng.integers(0, 2))
        p1 = int(rng.integers(1000, contigs[tid] - 2000))
        ins = int(max(l["readlen"] + 10, rng.normal(l["mean"], l["std"] / 2)))
        p2 = p1 + ins - l["readlen"]
        pid[0] += 1
        bam = 0 if l["bam"] == "a.bam" else 1
        recs[bam].append(dict(tid=tid, pos=p1, mtid=tid, mpos=p2, isize=ins, flag=0x1 | 0x2 | 0x40 | 0x20, qlen=l["readlen"], bdqual=60, rg=l["rg"], name=pid[0]))
        recs[bam].append(dict(tid=tid, pos=p2, mtid=tid, mpos=p1, isize=-ins, flag=0x1 | 0x2 | 0x80 | 0x10, qlen=l["readlen"], bdqual=60, rg=l["rg"], name=pid[0]))
    kinds = ["fr", "ff", "rr", "rf"]
    s = 0
    while s < n_slots:
        k = int(rng.choice(list(sizes)))
        members = list(range(s, min(s + k, n_slots)))
        for a in members:  # self groups
            if rng.random() < 0.5:
                for _ in range(int(rng.integers(1, 4))):
                    add_pair(a, a, str(rng.choice(["small", "ff", "rf"])), int(rng.integers(0, 2)))
        for i, a in enumerate(members):  # connections
            for b in members[i + 1:]:
                if rng.random() < (0.9 if b == a + 1 else (0.35 if k <= 5 else 0.1)):
                    kind = str(rng.choice(kinds))
                    for _ in range(int(rng.integers(1, 5))):
                        add_pair(a, b, kind if rng.random() < 0.8 else str(rng.choice(kinds)), int(rng.integers(0, 2)))
        if rng.random() < 0.08 and members[-1] < n_slots // 2:  # a link into the other contig
            far = int(rng.integers(n_slots // 2, n_slots))
            for _ in range(int(rng.integers(1, 4))):
                add_pair(members[0], far, "fr", int(rng.integers(0, 2)))
        for a in members:  # every slot needs two reads a few bases apart to become a region at all
            add_pair(a, a, "ff", int(rng.integers(0, 2))) if rng.random() < 0.3 else None
        s += k + (1 if rng.random() < 0.3 else 0)
    streams = []
    for b in (0, 1):
        rr = sorted(recs[b], key=lambda r: (r["tid"], r["pos"]))
        d = {k: np.array([r[k] for r in rr], dtype=dt) for k, dt in
             (("tid", np.int32), ("pos", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32),
              ("flag", np.uint16), ("qlen", np.int32), ("bdqual", np.uint8))}
        d["rg"] = [r["rg"] for r in rr]
        d["name_id"] = np.array([r["name"] for r in rr], dtype=np.uint64)
        streams.append(d)
    return config, streams, ["c1", "c2"]


GRAPH_OPTION_SETS = [dict(), dict(min_read_pair=1), dict(min_read_pair=3), dict(min_read_pair=4), dict(buffer_size=1), dict(buffer_size=2),
                     dict(buffer_size=3, min_read_pair=1), dict(buffer_size=7), dict(cn_lib=1, print_af=1), dict(chr_tid=0),
                     dict(chr_tid=0, min_read_pair=1), dict(transchr_rearrange=1, min_read_pair=1), dict(min_len=40), dict(fisher=1)]


def clash_names(streams, seed, frac=0.04):
    """Give a fraction of the records the name of another record, within and across the BAMs: names then occur three,
    four ... times (what merged BAMs with clashing read names look like).  The reference keeps running on such input
    (ReadRegionData.cpp:108-113 appends every sighting, SvBuilder.cpp:101-118 pairs first come first paired)."""
    rng = np.random.default_rng(77_000 + seed)
    out = [dict(s) for s in streams]
    allnames = np.concatenate([s["name_id"] for s in out])
    for s in out:
        n = len(s["name_id"])
        if n == 0 or len(allnames) == 0:
            continue
        ids = s["name_id"].copy()
        k = max(1, int(n * frac))
        victims = rng.choice(n, size=min(k, n), replace=False)
        ids[victims] = rng.choice(allnames, size=len(victims))
        s["name_id"] = ids
    return out
