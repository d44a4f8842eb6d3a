"""GPU tests of the device-side BAM decode (include/bdx.h bdx_bamdec_*, csrc/kz_inflate.hip, csrc/kb_records.hip):
the inflate kernel against zlib byte for byte, the record columns against the independent pure-Python decode of
tests/helpers.py (and through it against what the host reader produces, tests/test_producer.py)."""
import os
import struct
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, read_bam

pytestmark = pytest.mark.gpu

CHR21 = os.path.join(GOLDEN, "chr21")
BAMS = ["NA19238_chr21_del_inv.bam", "NA19240_chr21_del_inv.bam"]


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    assert bsize <= 65535, "test payload does not fit a BGZF member"
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def payloads(rng):
    """(label, bytes) inputs that exercise stored / fixed / dynamic blocks, long and overlapping matches, long codes"""
    words = [bytes(rng.integers(97, 123, rng.integers(2, 9), dtype=np.uint8)) for _ in range(300)]
    text = b" ".join(words[i] for i in rng.integers(0, len(words), 12000))
    skew = rng.choice(np.arange(256, dtype=np.uint8), size=60000, p=np.r_[np.full(8, 0.11), np.full(248, 0.12 / 248)]).tobytes()
    out = [
        ("empty", b""), ("one", b"x"), ("two", b"ab"), ("zeros", bytes(65280)), ("ones-short", b"\x01" * 700),
        ("pattern3", (b"abc" * 30000)[:65280]), ("pattern7", (b"0123456" * 9400)[:65280]),
        ("random", rng.integers(0, 256, 60000, dtype=np.uint8).tobytes()), ("random-small", rng.integers(0, 256, 517, dtype=np.uint8).tobytes()),
        ("text", text[:65280]), ("skewed", skew),   # skewed: a few short codes, 248 long ones (beyond the primary table)
        ("nibbles", (rng.integers(0, 4, 65280, dtype=np.uint8) * 17).tobytes()),
        ("runs", np.repeat(rng.integers(0, 256, 400, dtype=np.uint8), rng.integers(1, 400, 400))[:65000].tobytes()),
    ]
    return out


KZ = pytest.mark.parametrize("kz", ["wave"])   # kz_inflate.hip (a wave per member); the lane-per-member pair of round 4 left the library in round 6


@KZ
def test_inflate_matches_zlib_on_synthetic_members(kz, monkeypatch):
    from breakdancer_amd import bamdec
    rng = np.random.default_rng(5)
    blobs, want, labels = [], [], []
    for label, data in payloads(rng):
        for level in (0, 1, 6, 9):
            for strategy, mem in ((zlib.Z_DEFAULT_STRATEGY, 8), (zlib.Z_FIXED, 8), (zlib.Z_HUFFMAN_ONLY, 8), (zlib.Z_RLE, 8), (zlib.Z_FILTERED, 1),
                                  (zlib.Z_DEFAULT_STRATEGY, 1)):
                blobs.append(member(data, level, strategy, mem))
                want.append(data)
                labels.append("%s/l%d/s%d/m%d" % (label, level, strategy, mem))
    image = b"".join(blobs)
    members = bamdec.scan_bgzf(image)
    assert len(members) == len(blobs)
    out, status, ms = bamdec.inflate_blocks(image, members)
    bad = [labels[i] for i in range(len(blobs)) if status[i] != 0]
    assert not bad, "rejected: %s" % bad[:10]
    o = 0
    for i, w in enumerate(want):
        got = out[o:o + len(w)].tobytes()
        assert got == w, "member %d (%s) differs at byte %d" % (i, labels[i], next(k for k in range(len(w)) if got[k] != w[k]))
        o += len(w)
    assert o == len(out)


class TokenDeflate:
    """a DEFLATE writer for chosen token sequences (RFC 1951): zlib picks its own matches, the window tests below need a match of a
    given length at a given distance right behind given tokens.  One block: the fixed code (3.2.6), or a dynamic block (3.2.7)
    with the code lengths the caller gives (complete codes; sent without run lengths)"""
    LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    LEXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
    DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
    DEXTRA = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]

    @staticmethod
    def canonical(lens):   # symbol -> (code, length), RFC 1951 3.2.2
        code, out = 0, {}
        for l in range(1, 16):
            for s, x in enumerate(lens):
                if x == l:
                    out[s] = (code, l)
                    code += 1
            code <<= 1
        return out

    def __init__(self, lit_lens=None, dist_lens=None):
        self.acc, self.nbits, self.out, self.data = 0, 0, bytearray(), bytearray()
        self.bits(1, 1)   # BFINAL
        if lit_lens is None:
            self.bits(1, 2)
            lit_lens = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
            dist_lens = [5] * 30
        else:
            assert len(lit_lens) == 286 and len(dist_lens) == 30
            assert sum(2.0 ** -l for l in lit_lens if l) == 1.0 and sum(2.0 ** -l for l in dist_lens if l) == 1.0
            self.bits(2, 2)
            self.bits(286 - 257, 5); self.bits(30 - 1, 5); self.bits(19 - 4, 4)
            for sym in (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15):   # the code-length code: 0..15 in four bits each
                self.bits(0 if sym >= 16 else 4, 3)
            for l in list(lit_lens) + list(dist_lens):
                self.code(l, 4)
        self.lit, self.dist = self.canonical(lit_lens), self.canonical(dist_lens)

    def bits(self, v, n):   # n bits of v, least significant first
        self.acc |= (v & ((1 << n) - 1)) << self.nbits
        self.nbits += n
        while self.nbits >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.nbits -= 8

    def code(self, c, n):   # a Huffman code: most significant bit first
        self.bits(int(format(c, "0%db" % n)[::-1], 2), n)

    def literal(self, b):
        self.code(*self.lit[b])
        self.data.append(b)

    def match(self, length, dist):
        assert 3 <= length <= 258 and 1 <= dist <= len(self.data) and dist <= 32768
        ls = max(i for i in range(29) if self.LBASE[i] <= length) if length < 258 else 28
        self.code(*self.lit[257 + ls])
        self.bits(length - self.LBASE[ls], self.LEXTRA[ls])
        ds = max(i for i in range(30) if self.DBASE[i] <= dist)
        self.code(*self.dist[ds])
        self.bits(dist - self.DBASE[ds], self.DEXTRA[ds])
        for _ in range(length):
            self.data.append(self.data[-dist])

    def finish(self):
        self.code(*self.lit[256])
        if self.nbits: self.bits(0, 8 - self.nbits)
        data, comp = bytes(self.data), bytes(self.out)
        assert zlib.decompress(comp, -15) == data
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))), data


# a dynamic code whose longest matches have codes beyond any primary table (length 258: 10 bits; lengths 3..10: 4..10 bits), a match
# of 99..114 bytes in two bits, literals in nine
LONG_258 = [9] * 256 + [3] + [4, 5, 6, 7, 8, 9, 10] + [0] * 15 + [2] + [0] * 5 + [10]
DIST_LENS = [4, 4] + [5] * 28


def window_members(rng):
    """members whose matches reach back about as far as the kernels' LDS windows do (with a 1 KiB window the source of a
    long match at a distance of 575..640 that follows ~100 bytes of output in its step is neither wholly inside the window nor wholly
    written back): a long match at every distance around the windows' edges right behind a match of 60..127 bytes (so that it starts
    well into its step) -- in the fixed code, where every token is resolved by the tables, and in a dynamic code that gives the long
    match a 10-bit code, the kernel's path for tokens its tables do not hold --, behind runs of literals, alone and in pairs; near-duplicate
    records of 560..1100 bytes (what a BAM's neighbouring records look like) through zlib at levels 6 and 9"""
    out = []
    for dynamic in (False, True):
        for lo, hi in ((540, 760), (760, 1100), (1400, 2300)):
            for rep in range(10):
                f = TokenDeflate(LONG_258, DIST_LENS) if dynamic else TokenDeflate()
                for b in rng.integers(0, 256, 2600, dtype=np.uint8): f.literal(int(b))
                dist = lo + int(rng.integers(0, hi - lo))
                while len(f.data) < 58000 and len(f.out) < 60000:
                    k = int(rng.integers(0, 6))
                    for b in rng.integers(0, 256, int(rng.integers(0, 6)) if k else int(rng.integers(0, 141)), dtype=np.uint8): f.literal(int(b))
                    if k >= 2:   # ~100 bytes of output in front of the long match, in the same step
                        f.match(int(rng.integers(99, 115)) if dynamic else int(rng.integers(60, 128)), int(rng.integers(1, 500)))
                        for b in rng.integers(0, 256, int(rng.integers(0, 3)), dtype=np.uint8): f.literal(int(b))
                    f.match(258 if dynamic else int(rng.choice([258, 258, 258, 200, 131, 67])), dist)
                    if k == 5: f.match(258, dist)
                    dist = lo + (dist - lo + 1) % (hi - lo)
                out.append(f.finish() + ("tokens/%s/%d-%d/%d" % ("dynamic" if dynamic else "fixed", lo, hi, rep),))
    for rec_len in list(range(560, 1100, 45)) + [1500, 1800, 2100]:
        base = rng.integers(0, 256, rec_len, dtype=np.uint8)
        recs = []
        while sum(len(r) for r in recs) < 64000:
            base = base.copy()
            base[rng.integers(0, rec_len, int(rng.integers(1, 4)))] = rng.integers(0, 256, 1, dtype=np.uint8)
            recs.append(base.tobytes())
        data = b"".join(recs)[:65000]
        for level in (6, 9):
            out.append((member(data, level), data, "records/%d/l%d" % (rec_len, level)))
    return out


@KZ
def test_inflate_matches_that_reach_the_edge_of_the_window(kz, monkeypatch):
    from breakdancer_amd import bamdec
    cases = window_members(np.random.default_rng(11))
    image = b"".join(c[0] for c in cases)
    members = bamdec.scan_bgzf(image)
    assert len(members) == len(cases)
    out, status, ms = bamdec.inflate_blocks(image, members)
    bad = [cases[i][2] for i in range(len(cases)) if status[i] != 0]
    assert not bad, "rejected: %s" % bad[:10]
    o = 0
    for i, (_, w, label) in enumerate(cases):
        got = out[o:o + len(w)].tobytes()
        assert got == w, "member %d (%s) differs at byte %d" % (i, label, next(k for k in range(len(w)) if got[k] != w[k]))
        o += len(w)
    assert o == len(out)


@KZ
def test_inflate_matches_zlib_on_the_reference_bams(kz, monkeypatch):
    from breakdancer_amd import bamdec
    for name in BAMS:
        image = open(os.path.join(CHR21, name), "rb").read()
        members = bamdec.scan_bgzf(image)
        out, status, ms = bamdec.inflate_blocks(image, members)
        assert not status.any()
        want = b"".join(zlib.decompress(image[int(m["payload"]):int(m["payload"]) + int(m["payload_len"])], -15) for m in members)
        assert out.tobytes() == want


def test_inflate_and_decode_a_bam_that_looks_like_one(tmp_path):
    """level-6 deflate of records whose bases come from a shared reference and whose qualities are binned (bamwrite.Reference): the token
    mix of a real 30x BAM -- long matches between the ~30 records that cover a locus, most of them far back in the 32 KiB window, few
    literals -- where the files of random bases the other tests use are mostly literals and short matches.  >= 256 MB of inflated
    bytes: every member byte for byte against zlib, then the record columns of the device decode against the generator's"""
    from concurrent.futures import ThreadPoolExecutor
    from breakdancer_amd import bamdec
    from breakdancer_amd.bamwrite import write_bam
    from breakdancer_amd.synth import make_chromosome
    d = make_chromosome(length=4_400_000, seed=31)
    n = len(d["tid"])
    assert n * 214 >= (256 << 20)
    bam = str(tmp_path / "real.bam")
    write_bam(bam, d, ["chrR"], seed=3, level=6, realistic=True)
    image = np.fromfile(bam, dtype=np.uint8)
    members = bamdec.scan_bgzf(image)
    data = members[members["inflated_len"] > 0]
    ulen = int(data["inflated_len"].astype(np.int64).sum())
    assert ulen >= (256 << 20) and ulen / image.size > 3.5, (ulen, image.size)   # compresses like a BAM, not like noise (1.57)
    out, status, ms = bamdec.inflate_blocks(image, data)
    assert not status.any()
    raw = image.tobytes()

    def ref(lo, hi):
        return b"".join(zlib.decompress(raw[int(m["payload"]):int(m["payload"]) + int(m["payload_len"])], -15) for m in data[lo:hi])
    cuts = list(range(0, len(data), 256)) + [len(data)]
    with ThreadPoolExecutor(max_workers=8) as ex:
        want = b"".join(ex.map(lambda ab: ref(*ab), zip(cuts[:-1], cuts[1:])))
    got = out.tobytes()
    assert len(got) == len(want)
    if got != want:
        k = next(i for i in range(0, len(want), 4096) if got[i:i + 4096] != want[i:i + 4096])
        k = next(i for i in range(k, k + 4096) if got[i] != want[i])
        raise AssertionError("inflated bytes differ from zlib's at byte %d" % k)
    cols, names, stats = bamdec.decode_file(bam, rg_ids=["rg1"], rg_lib=[0])
    assert len(cols["tid"]) == n
    for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq"):
        np.testing.assert_array_equal(cols[k], d[k], err_msg=k)


@KZ
def test_inflate_verdict_on_corrupted_members_agrees_with_zlib(kz, monkeypatch):
    """a member is accepted only if zlib accepts it with the same bytes; what zlib rejects is rejected"""
    from breakdancer_amd import bamdec
    rng = np.random.default_rng(9)
    base = []
    for label, data in payloads(rng)[3:]:
        base.append((member(data, 6), data))
        base.append((member(data, 1, zlib.Z_FIXED), data))
    blobs, verdicts = [], []
    for blob, data in base:
        for _ in range(6):
            b = bytearray(blob)
            k = int(rng.integers(18, len(b) - 8))
            b[k] ^= 1 << int(rng.integers(0, 8))
            payload = bytes(b[18:len(b) - 8])
            d = zlib.decompressobj(-15)
            try:
                got = d.decompress(payload, len(data) + 1)
                ok = d.eof and len(got) == len(data)
            except zlib.error:
                got, ok = b"", False
            blobs.append(bytes(b))
            verdicts.append((ok, got))
    image = b"".join(blobs)
    members = bamdec.scan_bgzf(image)
    out, status, ms = bamdec.inflate_blocks(image, members)
    o = 0
    stricter = 0
    for i, (ok, got) in enumerate(verdicts):
        n = int(members["inflated_len"][i])
        if status[i] == 0:
            assert ok, "member %d accepted by the GPU, rejected by zlib" % i
            assert out[o:o + n].tobytes() == got
        elif ok:
            stricter += 1
        o += n
    assert stricter <= len(verdicts) // 20   # (zlib tolerates garbage behind the final block; so does the kernel -- nearly always the same verdict)


# ---- records ----
from namehash import check_name, hash_name  # noqa: E402


def config_read_groups(path):
    """read group -> library index (libraries in sorted name order, io/BamConfig.cpp:97-101) of a bam2cfg configuration"""
    rows = []
    for line in open(path):
        f = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t") if ":" in x)
        rows.append((f["readgroup"], f["lib"], f["map"]))
    libs = sorted({r[1] for r in rows})
    return rows, libs


def check_columns(cols, recs, rg_to_lib, fallback, region=None):
    keep = np.ones(len(recs["tid"]), bool)
    if region is not None:
        tid, beg, end = region
        keep = (recs["tid"] == tid) & (recs["rend"].astype(np.int64) > beg) & (recs["pos"].astype(np.int64) < end)
    idx = np.nonzero(keep)[0]
    assert len(cols["tid"]) == len(idx)
    for k in ("tid", "pos", "mtid", "mpos", "isize", "flag"):
        np.testing.assert_array_equal(cols[k].astype(np.int64), recs[k][idx].astype(np.int64), err_msg=k)
    np.testing.assert_array_equal(cols["qlen"].astype(np.int64), np.minimum(recs["qlen"][idx].astype(np.int64), 65535))
    np.testing.assert_array_equal(cols["mapq"], recs["bdqual"][idx])
    want_lib = np.array([rg_to_lib.get(recs["rg"][i], fallback) for i in idx], dtype=np.uint8)
    np.testing.assert_array_equal(cols["lib"], want_lib)
    want_key = np.array([hash_name(recs["name"][i].encode()) for i in idx], dtype=np.uint64)
    np.testing.assert_array_equal(cols["name_key"], want_key)
    want_check = np.array([check_name(recs["name"][i].encode()) for i in idx], dtype=np.uint64)
    np.testing.assert_array_equal(cols["name_check"], want_check)


@pytest.mark.parametrize("piece_blocks,ring,batch", [(512, 0, 0), (1, 1 << 20, 1), (2, 1 << 20, 3), (2, 0, 5), (1, 0, 0)])
def test_reference_bams_columns(piece_blocks, ring, batch):
    from breakdancer_amd import bamdec
    rows, libs = config_read_groups(os.path.join(CHR21, "inv_del_bam_config"))
    rg_ids = [r[0] for r in rows]
    rg_lib = [libs.index(r[1]) for r in rows]
    rg_to_lib = dict(zip(rg_ids, rg_lib))
    for bi, name in enumerate(BAMS):
        path = os.path.join(CHR21, name)
        targets, recs = read_bam(path)
        cols, names, stats = bamdec.decode_file(path, rg_ids=rg_ids, rg_lib=rg_lib, fallback_lib=1, bam_index=bi, piece_blocks=piece_blocks, ring_bytes=ring,
                                                batch_blocks=batch)
        assert names == targets
        check_columns(cols, recs, rg_to_lib, 1)
        assert (cols["bam"] == bi).all()
        # -o 21 style region filter (bam_index.c:571-576 overlap rule)
        t21 = targets.index("21")
        cols, _, _ = bamdec.decode_file(path, rg_ids=rg_ids, rg_lib=rg_lib, fallback_lib=1, region=(t21, 14_500_000, 14_600_000), piece_blocks=piece_blocks, ring_bytes=ring,
                                        batch_blocks=batch)
        check_columns(cols, recs, rg_to_lib, 1, region=(t21, 14_500_000, 14_600_000))


@pytest.mark.parametrize("ahead,piece_blocks", [(2, 1), (4, 1), (4, 2), (3, 512)])
def test_pieces_acquired_ahead_of_the_one_submitted(ahead, piece_blocks):
    """bdx_bamdec_acquire several times before bdx_bamdec_submit (a caller that reads the file ahead, as the CLI's feeder does): the pieces are
    taken in the order they were acquired, a thirteenth acquisition without a submit is refused (the decoder has twelve staging buffers), what was acquired and never submitted is dropped
    by bdx_bamdec_finish"""
    import ctypes as C
    from breakdancer_amd import bamdec
    rows, libs = config_read_groups(os.path.join(CHR21, "inv_del_bam_config"))
    rg_ids = [r[0] for r in rows]
    rg_lib = [libs.index(r[1]) for r in rows]
    path = os.path.join(CHR21, BAMS[0])
    targets, recs = read_bam(path)
    cols, names, stats = bamdec.decode_file(path, rg_ids=rg_ids, rg_lib=rg_lib, fallback_lib=1, piece_blocks=piece_blocks, ahead=ahead, batch_blocks=3)
    assert names == targets
    check_columns(cols, recs, dict(zip(rg_ids, rg_lib)), 1)
    if ahead == 4 and piece_blocks == 1:   # the ring holds twelve: the thirteenth is refused, and finish drops the held ones
        data = np.fromfile(path, dtype=np.uint8)
        members = bamdec.scan_bgzf(data)
        _, _, k, off = bamdec.bam_header(data, members)
        m = members[k:][members[k:]["inflated_len"] > 0]
        d = bamdec.BamDecoder(len(names), rg_ids=rg_ids, rg_lib=rg_lib, fallback_lib=1, first_record_offset=off)
        try:
            held = [d.acquire_fill(data, m[i:i + 1]) for i in range(12)]
            buf, tab = C.c_void_p(), C.c_void_p()
            assert d.lib.bdx_bamdec_acquire(d.h, 100, 1, C.byref(buf), C.byref(tab)) != 0
            d.submit_held(held[0], False)
            d.submit_held(held[1], False)
            assert d.finish() > 0        # two pieces decoded as far as their bytes go, ten dropped
        finally:
            d.close()


def synthetic_records(n, rng, tids=3):
    recs = []
    pos = 0
    for i in range(n):
        pos += int(rng.integers(0, 40))
        L = int(rng.choice([0, 1, 36, 100, 101, 250]))
        r = dict(tid=int(rng.integers(0, tids)) if rng.random() < 0.97 else -1, pos=pos, mtid=int(rng.integers(-1, tids)), mpos=int(rng.integers(0, 1 << 20)),
                 isize=int(rng.integers(-2000, 2000)), flag=int(rng.integers(0, 1 << 12)), qlen=L, mapq=int(rng.integers(0, 61)),
                 name="r%d_%s" % (i, "x" * int(rng.integers(0, 40))), rg=str(rng.choice(["a", "bb", "unknown", ""])),
                 am=(int(rng.integers(0, 300)) if rng.random() < 0.3 else None))
        recs.append(r)
    recs.sort(key=lambda r: (r["tid"] if r["tid"] >= 0 else 1 << 30, r["pos"]))
    return recs


@pytest.mark.parametrize("seed", range(4))
def test_synthetic_records_with_odd_shapes(seed, tmp_path):
    """records of many sizes (no sequence, aux tags of several types, unknown and missing read groups, unplaced reads,
    secondary / supplementary alignments) cut at arbitrary places by the members' boundaries"""
    from breakdancer_amd import bamdec
    from breakdancer_amd.bamwrite import write_bam_records
    rng = np.random.default_rng(100 + seed)
    recs = synthetic_records([6000, 30000, 30000, 6000][seed], rng)   # (seeds 1 and 2: several laps round a 1 MiB ring)
    path = str(tmp_path / "odd.bam")
    write_bam_records(path, recs, ["c0", "c1", "c2"], rgs=("a", "bb"), level=[1, 6, 0, 9][seed], seed=seed)
    targets, want = read_bam(path)
    cols, names, stats = bamdec.decode_file(path, rg_ids=["a", "bb"], rg_lib=[0, 1], fallback_lib=1, piece_blocks=[512, 1, 2, 5][seed],
                                            ring_bytes=[0, 1 << 20, 1 << 20, 0][seed], batch_blocks=[0, 1, 3, 7][seed])
    assert names == targets
    check_columns(cols, want, {"a": 0, "bb": 1}, 1)
    if seed in (1, 2):
        assert stats["inflated_bytes"] > 3 * (1 << 20)


def test_truncated_and_corrupt_files_are_errors(tmp_path):
    from breakdancer_amd import bamdec
    from breakdancer_amd.bamwrite import write_bam_records
    rng = np.random.default_rng(3)
    recs = synthetic_records(3000, rng)
    path = str(tmp_path / "t.bam")
    write_bam_records(path, recs, ["c0", "c1", "c2"], rgs=("a", "bb"), level=1)
    image = np.fromfile(path, dtype=np.uint8)
    members = bamdec.scan_bgzf(image)
    names, lens, k, off = bamdec.bam_header(image, members)
    data = members[members["inflated_len"] > 0]
    # the last data member missing: the final record is cut off
    d = bamdec.BamDecoder(len(names), first_record_offset=off, batch_blocks=6)
    d.feed(image, data[k:-1], 4)
    with pytest.raises(RuntimeError, match="truncated|corrupt"):
        d.finish()
    d.close()
    # a flipped bit in a payload: the member does not inflate (or the chain breaks)
    bad = image.copy()
    bad[int(data["payload"][k + 2]) + 40] ^= 0x10
    d = bamdec.BamDecoder(len(names), first_record_offset=off, batch_blocks=6)
    d.feed(bad, data[k:], 4)
    with pytest.raises(RuntimeError):
        d.finish()
    d.close()


def test_two_decoders_merged_by_a_gather_equal_the_oracle_run():
    """the reference's two BAMs, each through a decoder of its own (no sink), the merged order of the oracle's stream as the
    permutation: bdx_merge_decoded fills the context's store, the run prints the reference's table; a permutation that does not
    cover the records, or names one that does not exist, is refused"""
    from breakdancer_amd import bamdec
    import breakdancer_amd as bda
    from helpers import load_chr21, make_opts
    from runner import compare, product_options
    run = load_chr21(make_opts()).run()
    rows, libs = config_read_groups(os.path.join(CHR21, "inv_del_bam_config"))
    rg_ids = [r[0] for r in rows]
    rg_lib = [libs.index(r[1]) for r in rows]
    decs = []
    for b, fn in enumerate(run.bam_names):
        data = np.fromfile(os.path.join(CHR21, fn), dtype=np.uint8)
        members = bamdec.scan_bgzf(data)
        names, lens, k, off = bamdec.bam_header(data, members)
        d = bamdec.BamDecoder(len(names), bam_index=b, rg_ids=rg_ids, rg_lib=rg_lib, fallback_lib=0, first_record_offset=off)
        d.feed(data, members[k:], 3)
        d.finish()
        decs.append(d)
    from breakdancer_amd.api import LibraryConfig
    lcfg = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]), bam_file_index=int(run.lib_i[i, 1]),
                          name=run.lib_names[i]) for i in range(run.nlibs)]

    def ctx():
        return bda.BreakDancer(product_options(run.opts), lcfg, run.nbams, ntids=0, max_read_window_size=run.w0)

    src_file = run.m_bam.astype(np.uint8)
    src_index = run.m_src.astype(np.uint32)
    bd = ctx()
    bamdec.merge_decoded(bd, decs, src_file, src_index)
    bd.run()
    compare(run, bd)
    bd.close()
    bd = ctx()
    with pytest.raises(RuntimeError):
        bamdec.merge_decoded(bd, decs, src_file[:-1], src_index[:-1])     # does not cover the records
    bad = src_index.copy()
    bad[5] = 1 << 30
    with pytest.raises(RuntimeError):
        bamdec.merge_decoded(bd, decs, src_file, bad)                    # names a record that does not exist
    bd.close()
    for d in decs:
        d.close()
