"""Seeded synthetic read-pair generator (SoA batches in the layout of include/bdx.h).

Shape follows BASELINE.json configs[1] / SURVEY.md 8(d): one chromosome, 2x100 bp reads at a given coverage,
insert ~ N(400, 30) clipped at 200, FR orientation, ~1 % of the pairs discordant in clusters of 12
(DEL: insert 1500-1600 FR; INS: 201-240 FR; INV: FF; ITX: RF), 3 % of the pairs at MAPQ 20, proper-pair bit
only on normal pairs, records sorted by (pos, strand), one read group / library."""
import numpy as np

READLEN = 100
LIB_C2 = dict(mean_insertsize=400.0, std_insertsize=30.0, uppercutoff=490.0, lowercutoff=310.0, readlens=100.0)


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def make_chromosome(length=50_000_000, coverage=30.0, seed=1, tid=0, lib=0, bam=0, discordant=0.01, cluster=12,
                    name_base=0, mean=400.0, std=30.0, n_pairs=None):
    """Returns a dict of numpy arrays (one record per read, position sorted)."""
    rng = np.random.default_rng(seed)
    if n_pairs is None:
        n_pairs = int(length * coverage / (2 * READLEN))
    n_clusters = int(n_pairs * discordant / cluster)
    n_disc = n_clusters * cluster
    n_norm = n_pairs - n_disc
    hi = max(length - 5000, 2000)

    start = np.empty(n_pairs, np.int64)
    insert = np.empty(n_pairs, np.int64)
    kind = np.zeros(n_pairs, np.int8)  # 0 normal, 1 DEL, 2 INS, 3 INV(FF), 4 ITX(RF)
    start[:n_norm] = rng.integers(1000, hi, n_norm)
    insert[:n_norm] = np.maximum(np.rint(rng.normal(mean, std, n_norm)), 201).astype(np.int64)
    if n_disc:
        centres = rng.integers(1000, hi, n_clusters)
        ckind = rng.integers(1, 5, n_clusters).astype(np.int8)
        k = np.repeat(ckind, cluster)
        kind[n_norm:] = k
        start[n_norm:] = np.repeat(centres, cluster) + rng.integers(0, 200, n_disc)
        ins = np.empty(n_disc, np.int64)
        ins[k == 1] = rng.integers(1500, 1601, int((k == 1).sum()))
        ins[k == 2] = rng.integers(201, 241, int((k == 2).sum()))
        m = (k == 3) | (k == 4)
        ins[m] = rng.integers(900, 1101, int(m.sum()))
        insert[n_norm:] = ins
    mapq = np.where(rng.random(n_pairs) < 0.03, 20, 60).astype(np.uint8)

    lpos = start
    rpos = start + insert - READLEN
    # orientation of (left read, right read): normal/DEL/INS = F,R ; INV = F,F ; ITX = R,F
    lrev = kind == 4
    rrev = (kind != 3) & (kind != 4)
    proper = kind == 0
    pid = np.arange(n_pairs, dtype=np.uint64) + np.uint64(name_base)
    key = splitmix64(pid)

    def rec(pos, mpos, rev, mrev, isz, first):
        flag = np.full(n_pairs, 0x1, np.uint16)
        flag |= np.where(proper, 0x2, 0).astype(np.uint16)
        flag |= np.where(rev, 0x10, 0).astype(np.uint16)
        flag |= np.where(mrev, 0x20, 0).astype(np.uint16)
        flag |= np.uint16(0x40 if first else 0x80)
        return pos, mpos, isz, flag

    p1, m1, i1, f1 = rec(lpos, rpos, lrev, rrev, insert, True)
    p2, m2, i2, f2 = rec(rpos, lpos, rrev, lrev, -insert, False)
    pos = np.concatenate([p1, p2])
    mpos = np.concatenate([m1, m2])
    isz = np.concatenate([i1, i2])
    flag = np.concatenate([f1, f2])
    mq = np.concatenate([mapq, mapq])
    keys = np.concatenate([key, key])
    order = np.argsort(pos * 2 + ((flag >> 4) & 1), kind="stable")
    n = 2 * n_pairs
    return dict(
        tid=np.full(n, tid, np.int32), pos=pos[order].astype(np.int32), mtid=np.full(n, tid, np.int32),
        mpos=mpos[order].astype(np.int32), isize=isz[order].astype(np.int32), flag=flag[order],
        qlen=np.full(n, READLEN, np.uint16), mapq=mq[order], lib=np.full(n, lib, np.uint8),
        bam=np.full(n, bam, np.uint8), name_key=keys[order])


def concat(parts):
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


def make_genome(lengths, coverage=30.0, seed=1, libs=((400.0, 30.0),), lib_bam=(0,), n_translocations=0, ctx_pairs=15, threads=None,
                only_tids=None):
    """Multi-chromosome, multi-library synthetic input (configs[2]-[4] shapes, scaled by `lengths`).
    Every library contributes coverage/len(libs) (`coverage` may also be a sequence with one value per library, e.g. a 60x
    tumour and a 30x normal file); `lib_bam[i]` is the source file of library i.  Planted translocations add clusters of
    `ctx_pairs` inter-chromosomal pairs (both mates carry tid != mtid, isize 0).
    Returns the merged, (tid, pos, strand)-sorted SoA.  The (chromosome, library) parts are generated and the chromosomes
    sorted on `threads` threads (numpy releases the GIL), which is what makes full-size inputs practical in a test.
    only_tids: generate just these chromosomes' records of the same genome (a rank of a sharded run makes its own share)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(seed)
    cov = list(coverage) if hasattr(coverage, "__len__") else [coverage / len(libs)] * len(libs)
    jobs = []
    base = 0
    for tid, L in enumerate(lengths):
        for li, (mean, std) in enumerate(libs):
            if only_tids is None or tid in only_tids:
                jobs.append(dict(length=L, coverage=cov[li], seed=seed * 1000 + tid * 16 + li, tid=tid, lib=li, bam=lib_bam[li],
                                 name_base=base, mean=mean, std=std))
            base += 1 << 36
    if threads is None:
        threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(lambda kw: make_chromosome(**kw), jobs))
    per_tid = [[p for p, j in zip(parts, jobs) if j["tid"] == t] for t in range(len(lengths))]
    if n_translocations:
        n = n_translocations * ctx_pairs
        ta = rng.integers(0, len(lengths), n_translocations)
        tb = (ta + 1 + rng.integers(0, len(lengths) - 1, n_translocations)) % len(lengths)
        ca = np.array([rng.integers(1000, lengths[t] - 5000) for t in ta])
        cb = np.array([rng.integers(1000, lengths[t] - 5000) for t in tb])
        ta, tb, ca, cb = (np.repeat(x, ctx_pairs) for x in (ta, tb, ca, cb))
        pa = ca + rng.integers(0, 200, n)
        pb = cb + rng.integers(0, 200, n)
        li = rng.integers(0, len(libs), n)
        key = splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(base))
        mq = np.where(rng.random(n) < 0.03, 20, 60).astype(np.uint8)
        bam = np.asarray(lib_bam, np.uint8)[li]

        def rec(tid, pos, mtid, mpos, rev, mrev, first):
            flag = np.full(n, 0x1, np.uint16) | np.where(rev, 0x10, 0).astype(np.uint16) | np.where(mrev, 0x20, 0).astype(np.uint16)
            flag |= np.uint16(0x40 if first else 0x80)
            return dict(tid=tid.astype(np.int32), pos=pos.astype(np.int32), mtid=mtid.astype(np.int32), mpos=mpos.astype(np.int32),
                        isize=np.zeros(n, np.int32), flag=flag, qlen=np.full(n, READLEN, np.uint16), mapq=mq,
                        lib=li.astype(np.uint8), bam=bam, name_key=key)
        f = np.zeros(n, bool)
        for part in (rec(ta, pa, tb, pb, f, ~f, True), rec(tb, pb, ta, pa, ~f, f, False)):
            for t in range(len(lengths)):  # (part order inside a chromosome as in one global stable sort)
                if only_tids is not None and t not in only_tids:
                    continue
                m = part["tid"] == t
                if m.any():
                    per_tid[t].append({k: v[m] for k, v in part.items()})

    def sort_tid(ps):
        d = concat(ps)
        order = np.lexsort(((d["flag"] >> 4) & 1, d["pos"]))
        return {k: v[order] for k, v in d.items()}
    with ThreadPoolExecutor(max_workers=threads) as ex:
        chroms = list(ex.map(sort_tid, [ps for ps in per_tid if ps]))
    return concat(chroms)
