"""Minimal BGZF/BAM writer for synthetic inputs (numpy-vectorised fixed-size records).  Tooling for tests and
end-to-end measurements only: the product never writes BAM (BamWriter is out of scope)."""
import struct
import zlib

import numpy as np

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


_BIN_QUALS = np.array([37, 23, 12, 2], np.uint8)   # the four quality bins of current Illumina instruments


class Reference:
    """A seeded random reference, one sequence per chromosome, generated when first asked for (uint8 BAM base codes 1 / 2 / 4 / 8).
    Reads drawn from it share sequence with their neighbours in a position-sorted file -- what a real BAM's deflate streams are made of
    (matches between the ~30 records that cover a locus), and what records of independent random bases cannot show."""

    def __init__(self, seed=1234, pad=1024):
        import threading
        self.seed, self.pad = seed, pad
        self._seqs = {}
        self._lock = threading.Lock()

    def seq(self, tid, upto):
        with self._lock:
            s = self._seqs.get(tid)
            if s is None or len(s) < upto + self.pad:
                n = max(upto + self.pad, 2 * len(s) if s is not None else 0)
                codes = np.array([1, 2, 4, 8], np.uint8)
                s = codes[np.random.default_rng([self.seed, tid]).integers(0, 4, n, dtype=np.uint8)]   # (a prefix of the same stream: deterministic in tid)
                self._seqs[tid] = s
            return s


def _realistic_payload(soa, lo, hi, L, rng, ref):
    """(bases as packed nybbles, qualities) of records [lo, hi): bases = the reference at the read's position with 0.5 % substitutions,
    qualities in four bins -- a read starts in the top bin and steps down towards its 3' end, with a few isolated dips"""
    n = hi - lo
    nb = (L + 1) // 2
    tid = np.asarray(soa["tid"][lo:hi]).astype(np.int64)
    pos = np.maximum(np.asarray(soa["pos"][lo:hi]).astype(np.int64), 0)
    bases = np.empty((n, 2 * nb), np.uint8)
    bases[:, L:] = 0
    for t in np.unique(tid):
        m = tid == t
        s = ref.seq(int(t), int(pos[m].max()) + L)
        win = np.lib.stride_tricks.sliding_window_view(s, L)
        bases[m, :L] = win[pos[m]]
    nsub = int(n * L * 0.005)
    if nsub:
        codes = np.array([1, 2, 4, 8], np.uint8)
        bases[rng.integers(0, n, nsub), rng.integers(0, L, nsub)] = codes[rng.integers(0, 4, nsub)]
    packed = (bases[:, 0::2] << 4) | bases[:, 1::2]
    col = np.arange(L, dtype=np.int32)[None, :]
    k1 = rng.integers(int(L * 0.6), int(L * 1.6), n, dtype=np.int32)[:, None]          # where the read leaves the top bin (often: never)
    k2 = k1 + rng.integers(int(L * 0.05), int(L * 0.5), n, dtype=np.int32)[:, None]
    q = np.where(col < k1, 0, np.where(col < k2, 1, 2)).astype(np.uint8)
    ndip = int(n * L * 0.06)
    if ndip:
        q[rng.integers(0, n, ndip), rng.integers(0, L, ndip)] = rng.choice(np.array([1, 2, 3], np.uint8), ndip, p=[0.6, 0.3, 0.1])
    return packed, _BIN_QUALS[q]


def _fixed_records(soa, lo, hi, L, rg, rng, names=None, ref=None):
    """records [lo, hi) of the SoA as one uint8 matrix (fixed-size records: same name width, read length, aux block)"""
    n = hi - lo
    if names is None:  # mates share the name: derive it from the name key (16 hex digits)
        keys = soa["name_key"][lo:hi].astype(np.uint64)
        hexd = np.frombuffer(b"0123456789abcdef", np.uint8)
        nm = np.empty((n, 17), np.uint8)
        for i in range(16):
            nm[:, 15 - i] = hexd[((keys >> np.uint64(4 * i)) & np.uint64(15)).astype(np.int64)]
        nm[:, 16] = 0
    else:
        w = max(len(x) for x in names) + 1
        nm = np.zeros((n, w), np.uint8)
        for i, x in enumerate(names[lo:hi]):
            nm[i, :len(x)] = np.frombuffer(x.encode(), np.uint8)
    lname = nm.shape[1]
    rgs = [rg] if isinstance(rg, str) else list(rg)   # several read groups (ids of one width: records keep one size): record i carries rgs[soa["lib"][i]]
    assert len({len(r) for r in rgs}) == 1
    aux = b"RGZ" + rgs[0].encode() + b"\0"
    rec_len = 32 + lname + 4 + (L + 1) // 2 + L + len(aux)
    rec = np.zeros((n, 4 + rec_len), np.uint8)

    def put(col, arr, dt):
        a = np.ascontiguousarray(np.asarray(arr).astype(dt)).view(np.uint8).reshape(n, -1)
        rec[:, col:col + a.shape[1]] = a

    put(0, np.full(n, rec_len), "<i4")
    put(4, soa["tid"][lo:hi], "<i4")
    put(8, soa["pos"][lo:hi], "<i4")
    rec[:, 12] = lname
    rec[:, 13] = soa["mapq"][lo:hi]
    put(14, np.zeros(n), "<u2")
    put(16, np.ones(n), "<u2")
    put(18, soa["flag"][lo:hi], "<u2")
    put(20, np.full(n, L), "<i4")
    put(24, soa["mtid"][lo:hi], "<i4")
    put(28, soa["mpos"][lo:hi], "<i4")
    put(32, soa["isize"][lo:hi], "<i4")
    o = 36
    rec[:, o:o + lname] = nm
    o += lname
    put(o, np.full(n, (L << 4) | 0), "<u4")
    o += 4
    nb = (L + 1) // 2
    if ref is not None:
        packed, quals = _realistic_payload(soa, lo, hi, L, rng, ref)
        rec[:, o:o + nb] = packed
        o += nb
        rec[:, o:o + L] = quals
    else:
        codes = np.array([1, 2, 4, 8], np.uint8)
        rec[:, o:o + nb] = (codes[rng.integers(0, 4, (n, nb))] << 4) | codes[rng.integers(0, 4, (n, nb))]
        o += nb
        rec[:, o:o + L] = rng.integers(2, 41, (n, L), dtype=np.uint8)
    o += L
    if len(rgs) == 1:
        rec[:, o:o + len(aux)] = np.frombuffer(aux, np.uint8)
    else:
        table = np.stack([np.frombuffer(b"RGZ" + r.encode() + b"\0", np.uint8) for r in rgs])
        rec[:, o:o + len(aux)] = table[np.asarray(soa["lib"][lo:hi]).astype(np.int64)]
    return rec


def write_bam(path, soa, targets, rg="rg1", readlen=None, level=1, seed=0, names=None, threads=None, chunk=500_000, index=False, realistic=False):
    """soa: dict with tid,pos,mtid,mpos,isize,flag,qlen,mapq (+name_key used to derive the read names).
    index=True also writes <path>.bai: per sequence ONE chunk (first record .. behind the last, in the root bin -- legal, if coarser
    than samtools') and the linear index of 16 kb windows, vectorised (the records have one size, so every virtual offset follows
    from the compressed sizes of the blocks).
    Every record gets `readlen` random bases / qualities (default: soa['qlen'][0]), a 100M-style CIGAR and RG:Z:<rg>.
    realistic=True: bases drawn from a seeded random REFERENCE at the read's position (overlapping reads share sequence) and qualities in
    four bins (class Reference, _realistic_payload); with level=6 -- samtools' default, bgzf.c -- the file compresses like a real 30x BAM.
    Records are built and deflated `chunk` at a time on `threads` threads (numpy and zlib release the GIL), so a
    configs[1]-sized file (15 M records, ~3 GB of record bytes) takes seconds on a many-core host and bounded memory."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = len(soa["tid"])
    L = int(readlen if readlen is not None else (soa["qlen"][0] if n else 100))
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (t, 300000000) for t in targets) + \
           "".join("@RG\tID:%s\tLB:lib%d\tSM:s\n" % (r, i + 1) for i, r in enumerate([rg] if isinstance(rg, str) else rg))
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(targets))
    for t in targets:
        hdr += struct.pack("<i", len(t) + 1) + t.encode() + b"\0" + struct.pack("<i", 300000000)

    ref = Reference(seed=seed + 977) if realistic else None

    def piece(i):
        lo, hi = i * chunk, min(n, (i + 1) * chunk)
        raw = _fixed_records(soa, lo, hi, L, rg, np.random.default_rng([seed, i]), names, ref).tobytes()
        blocks = [_bgzf_block(raw[j:j + 65280], level) for j in range(0, len(raw), 65280)]
        return b"".join(blocks), [len(b) for b in blocks], len(raw)

    nchunks = (n + chunk - 1) // chunk
    if threads is None:
        threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    with open(path, "wb") as f:
        for j in range(0, len(hdr), 65280):
            f.write(_bgzf_block(hdr[j:j + 65280], level))
        chunk_coff, chunk_blocks, rec_bytes = [], [], 0   # per chunk: compressed offset of its first block, its blocks' compressed sizes
        with ThreadPoolExecutor(max_workers=threads) as ex:
            window = []  # at most 2 x threads chunks in flight: bounded memory, written in order
            nxt = 0
            while nxt < nchunks or window:
                while nxt < nchunks and len(window) < 2 * threads:
                    window.append(ex.submit(piece, nxt))
                    nxt += 1
                data, sizes, raw_len = window.pop(0).result()
                chunk_coff.append(f.tell())
                chunk_blocks.append(sizes)
                if raw_len:
                    rec_bytes = raw_len // (min(n, (len(chunk_coff)) * chunk) - (len(chunk_coff) - 1) * chunk)
                f.write(data)
        end_coff = f.tell()
        f.write(_EOF)
    if index and n:
        # virtual offset of every record: chunk c starts a fresh run of 65280-byte blocks; record k of the chunk begins at raw offset k * rec_bytes
        idx = np.arange(n, dtype=np.int64)
        c = idx // chunk
        raw_off = (idx - c * chunk) * rec_bytes
        blk = raw_off // 65280
        within = raw_off - blk * 65280
        starts = [np.concatenate([[0], np.cumsum(np.asarray(sz, np.int64))[:-1]]) + co for sz, co in zip(chunk_blocks, chunk_coff)]
        blk_coff = np.concatenate(starts)
        first_blk = np.concatenate([[0], np.cumsum([len(sz) for sz in chunk_blocks])[:-1]])
        voff = (blk_coff[first_blk[c] + blk].astype(np.uint64) << np.uint64(16)) | within.astype(np.uint64)
        tid = np.asarray(soa["tid"]).astype(np.int64)
        pos = np.asarray(soa["pos"]).astype(np.int64)
        out = bytearray(b"BAI\1" + struct.pack("<i", len(targets)))
        bounds = np.searchsorted(tid, np.arange(len(targets) + 1))
        for t in range(len(targets)):
            lo, hi = int(bounds[t]), int(bounds[t + 1])
            if hi <= lo:
                out += struct.pack("<ii", 0, 0)
                continue
            v_end = int(voff[hi]) if hi < n else (end_coff << 16)
            out += struct.pack("<i", 1) + struct.pack("<Ii", 0, 1) + struct.pack("<QQ", int(voff[lo]), v_end)
            # linear index: the first record that STARTS in the window before (reads are far shorter than a window: never later than
            # the first record that overlaps the window)
            w = np.maximum(pos[lo:hi], 0) >> 14
            n_intv = int(w.max()) + 1
            first = np.full(n_intv, -1, np.int64)
            uw, ui = np.unique(w, return_index=True)
            first[uw] = ui + lo
            lin = np.zeros(n_intv, np.uint64)
            last = int(voff[lo])
            prev = last
            for k in range(n_intv):
                cur = int(voff[first[k]]) if first[k] >= 0 else prev
                lin[k] = prev if k else cur   # window k: what started in window k - 1 may reach into it
                prev = cur if first[k] >= 0 else prev
            out += struct.pack("<i", n_intv) + lin.astype("<u8").tobytes()
        open(path + ".bai", "wb").write(bytes(out))
    return path


def _reg2bin(beg, end):
    """UCSC binning scheme of the BAM specification (section 5.3)"""
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _write_bai(path, n_ref, entries):
    """entries: (tid, beg, end, virtual offset of the record, virtual offset behind it) in file order, mapped records only"""
    bins = [dict() for _ in range(n_ref)]
    lin = [dict() for _ in range(n_ref)]
    for tid, beg, end, v0, v1 in entries:
        b = _reg2bin(beg, end)
        ch = bins[tid].setdefault(b, [])
        if ch and ch[-1][1] == v0:
            ch[-1][1] = v1
        else:
            ch.append([v0, v1])
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            if w not in lin[tid] or v0 < lin[tid][w]:
                lin[tid][w] = v0
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for t in range(n_ref):
        out += struct.pack("<i", len(bins[t]))
        for b, ch in sorted(bins[t].items()):
            out += struct.pack("<Ii", b, len(ch))
            for v0, v1 in ch:
                out += struct.pack("<QQ", v0, v1)
        n_intv = (max(lin[t]) + 1) if lin[t] else 0
        out += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):  # (windows without their own entry inherit the previous one, as samtools' index does)
            last = lin[t].get(w, last)
            out += struct.pack("<Q", last)
    open(path, "wb").write(bytes(out))


def write_bam_records(path, recs, targets, rgs=(), level=1, seed=0, index=False):
    """Generic (slow, per-record) writer for small test inputs.  recs: list of dicts with tid,pos,mtid,mpos,isize,flag,
    qlen,mapq,name and optional rg (str or ''), am (int or None); random bases/qualities of length qlen.
    index=True also writes <path>.bai (bins + linear index, BAM specification section 5.2)."""
    rng = np.random.default_rng(seed)
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (t, 300000000) for t in targets) + \
           "".join(("@RG\tID:%s\tLB:x\tSM:s\n" % r) if isinstance(r, str) else ("@RG\tID:%s\tPL:%s\tLB:%s\tSM:s\n" % (r[0], r[2], r[1]))
                   for r in rgs)  # (id, library, platform) tuples give full @RG lines
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(targets)))
    for t in targets:
        out += struct.pack("<i", len(t) + 1) + t.encode() + b"\0" + struct.pack("<i", 300000000)
    codes = np.array([1, 2, 4, 8], np.uint8)
    spans = []  # (tid, beg, end, raw offset of the record, raw offset behind it)
    for r in recs:
        L = int(r["qlen"])
        name = r["name"].encode() + b"\0"
        nb = (L + 1) // 2
        seq = ((codes[rng.integers(0, 4, nb)] << 4) | codes[rng.integers(0, 4, nb)]).astype(np.uint8).tobytes()
        qual = rng.integers(2, 41, L, dtype=np.uint8).tobytes()
        aux = b""
        if r.get("am") is not None:
            aux += b"AMC" + struct.pack("<B", int(r["am"]) & 0xFF)
        if r.get("rg"):
            aux += b"RGZ" + r["rg"].encode() + b"\0"
        aux += b"NMi" + struct.pack("<i", 1)
        ncig = 1 if L else 0
        body = struct.pack("<iiBBHHHiiii", int(r["tid"]), int(r["pos"]), len(name), int(r["mapq"]), 0, ncig, int(r["flag"]), L,
                           int(r["mtid"]), int(r["mpos"]), int(r["isize"])) + name + (struct.pack("<I", (L << 4) | 0) if ncig else b"") + \
            seq + qual + aux
        start = len(out)
        out += struct.pack("<i", len(body)) + body
        if int(r["tid"]) >= 0:
            spans.append((int(r["tid"]), int(r["pos"]), int(r["pos"]) + max(L, 1), start, len(out)))
    raw = bytes(out)
    coffs = []
    with open(path, "wb") as f:
        for i in range(0, len(raw), 65280):
            coffs.append(f.tell())
            f.write(_bgzf_block(raw[i:i + 65280], level))
        coffs.append(f.tell())
        f.write(_EOF)
    if index:
        def voff(o):  # raw offset -> BGZF virtual offset
            return (coffs[o // 65280] << 16) | (o % 65280) if o < len(raw) else (coffs[-1] << 16)
        _write_bai(path + ".bai", len(targets), [(t, b, e, voff(s0), voff(s1)) for t, b, e, s0, s1 in spans])
    return path


# hg38 primary assembly, chr1-22, X, Y (Mbp) and the four libraries of BASELINE.json configs[2]
HG38_MBP = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28, 114.36, 107.04,
            101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82, 156.04, 57.23]
LIBS4 = ((400.0, 30.0), (350.0, 40.0), (500.0, 50.0), (300.0, 25.0))


def write_genome_bam(td, fraction, only_tids=None, tag="genome", translocations=None, seed=11, realistic=False):
    """ONE indexed, position-sorted 24-chromosome BAM of the configs[2]/[3] genome at `fraction` of hg38's lengths -- 30x, 2x100 bp, four
    libraries as four read groups, planted translocations -- and its bam2cfg-format configuration (one line per read group), in directory
    td.  fraction 1/8: 116 M records, 15.9 GB, one GPU's share of the 8-GPU configurations.  only_tids: just these chromosomes' records of
    the same genome (the slice the CPU baseline is timed on).  realistic=True: the same records with bases from a random reference and
    binned qualities, deflated at level 6 (tag it differently: the file name says only tag and fraction).  Returns (bam, cfg, records);
    files that exist are kept."""
    import os
    from .synth import make_genome
    os.makedirs(td, exist_ok=True)
    bam = os.path.join(td, "%s_%g.bam" % (tag, fraction))
    cfg = os.path.join(td, "%s_%g.cfg" % (tag, fraction))
    if not (os.path.exists(bam) and os.path.exists(cfg) and os.path.exists(bam + ".n")):
        lengths = [int(m * 1e6 * fraction) for m in HG38_MBP]
        d = make_genome(lengths, coverage=30.0, seed=seed, libs=LIBS4, lib_bam=(0, 0, 0, 0),
                        n_translocations=max(20, int(5000 * fraction * 8)) if translocations is None else translocations, only_tids=only_tids)
        write_bam(bam, d, ["chr%d" % (i + 1) for i in range(len(lengths))], rg=["rg%d" % (i + 1) for i in range(len(LIBS4))], seed=5, index=True,
                  realistic=realistic, level=6 if realistic else 1)
        with open(cfg, "w") as f:
            for i, (m, sd) in enumerate(LIBS4):
                f.write("readgroup:rg%d\tplatform:illumina\tmap:%s\treadlen:100.00\tlib:lib%d\tlower:%.2f\tupper:%.2f\tmean:%.2f\tstd:%.2f\n"
                        % (i + 1, os.path.basename(bam), i + 1, m - 3 * sd, m + 3 * sd, m, sd))
        open(bam + ".n", "w").write(str(len(d["tid"])))
    return bam, cfg, int(open(bam + ".n").read())
