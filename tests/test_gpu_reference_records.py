"""The reference's own unit-level records for a1 (Alignment fields, bdqual, read group, FASTQ text), through BOTH readers of the
product -- the host reader (bin/bdx-dump-reads, host/column_reader.cpp) and the device reader (bdx_bamdec_*, kb_records.hip) -- and
through the CLI's -d dump:

* test/lib/io/TestAlignment.cpp:11-46   a hand-made bam1_t: tid 22, pos 29185299, flag 163, MAPQ 60, mate 22:29184911, isize -388,
  name "junk", CIGAR 2M, sequence CT, qualities HB, aux RG:Z:rg3 AM:C:37; expected: read group "rg3", FASTQ "@junk\\nCT\\n+\\nHB\\n"
  (:16-21), forward strand, not leftmost, |isize| 388 (:49-89).  bdqual = the AM tag (io/Alignment.cpp:12-23): 37, not 60.
* test/lib/io/TestBam.cpp:36-51,84-93   six SAM records (three pairs at 21:10 / 21:15); expected: reads 0, 2, 4 are leftmost.

The records below are these fixtures' DATA (field values and the 23 payload bytes), written as BAM by this file."""
import os
import struct
import subprocess

import numpy as np
import pytest

from breakdancer_amd.bamwrite import _EOF, _bgzf_block
from helpers import ROOT
from namehash import check_name, hash_name
from test_producer import dump

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "bin", "breakdancer-max")

# TestAlignment.cpp:11-14: read name, CIGAR, sequence, qualities, aux -- the bam1_t payload as the reference holds it
JUNK_DATA = bytes([0x6a, 0x75, 0x6e, 0x6b, 0x0, 0x20, 0x0, 0x0, 0x0, 0x28, 0x27, 0x21,
                   0x52, 0x47, 0x5a, 0x72, 0x67, 0x33, 0x0, 0x41, 0x4d, 0x43, 0x25])
JUNK_CORE = dict(tid=22, pos=29185299, bin=6462, mapq=60, l_qname=5, flag=163, n_cigar=1, l_qseq=2, mtid=22, mpos=29184911, isize=-388)
SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def raw_record(core, data):
    body = struct.pack("<iiBBHHHiiii", core["tid"], core["pos"], core["l_qname"], core["mapq"], core["bin"], core["n_cigar"], core["flag"],
                       core["l_qseq"], core["mtid"], core["mpos"], core["isize"]) + data
    return struct.pack("<i", len(body)) + body


def make_record(name, tid, pos, flag, mapq, mtid, mpos, isize, seq, qual, cigar, aux=b""):
    """a BAM record from SAM-style fields (cigar: list of (length, op index in MIDNSHP=X); qual: printable, Phred + 33)"""
    nm = name.encode() + b"\0"
    cg = b"".join(struct.pack("<I", (n << 4) | op) for n, op in cigar)
    sq = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        sq[i // 2] |= SEQ_CODE[c] << (4 if i % 2 == 0 else 0)
    ql = bytes(ord(c) - 33 for c in qual)
    core = dict(tid=tid, pos=pos, bin=4681, mapq=mapq, l_qname=len(nm), flag=flag, n_cigar=len(cigar), l_qseq=len(seq), mtid=mtid, mpos=mpos, isize=isize)
    return raw_record(core, nm + cg + bytes(sq) + ql + aux)


def write_raw_bam(path, targets, records, rgs=()):
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (t, 300000000) for t in targets) + \
           "".join("@RG\tID:%s\tLB:x\tSM:s\n" % r for r in rgs)
    out = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(targets))
    for t in targets:
        out += struct.pack("<i", len(t) + 1) + t.encode() + b"\0" + struct.pack("<i", 300000000)
    out += b"".join(records)
    with open(path, "wb") as f:
        f.write(_bgzf_block(out, 6))
        f.write(_EOF)


def both_readers(tmp_path, cfg_name, bam_name, rg_ids, rg_lib, fallback):
    """(host reader's rows [tid pos mtid mpos isize flag qlen bdqual lib bam], keys, checks), device reader's columns"""
    from breakdancer_amd import bamdec
    head, rows, keys = dump([cfg_name], str(tmp_path))
    checks = dump.checks
    cols, names, _ = bamdec.decode_file(str(tmp_path / bam_name), rg_ids=rg_ids, rg_lib=rg_lib, fallback_lib=fallback)
    return (rows, keys, checks), cols


def test_TestAlignment_record_through_both_readers_and_the_fastq_dump(tmp_path):
    targets = [str(i + 1) for i in range(22)] + ["X"]          # (tid 22 = the 23rd sequence)
    # the fixture's read is the second in pair, forward, mate reverse and UPSTREAM of it: an outward-facing (RF) pair.  Its mate and two
    # more such pairs make the smallest input in which it supports an SV, so that -d writes it
    recs = []
    for k, (name, dx) in enumerate((("p2", -40), ("junk", 0), ("p3", 35))):
        lo, hi = 29184911 + dx, 29185299 + dx
        aux = b"RGZrg3\0AMC" + bytes([37])
        mate = make_record(name, 22, lo, 83, 60, 22, hi, 388, "AG", "II", [(2, 0)], aux)
        if name == "junk":
            read = raw_record(JUNK_CORE, JUNK_DATA)
        else:
            read = make_record(name, 22, hi, 163, 60, 22, lo, -388, "CT", "HB", [(2, 0)], aux)
        recs.append((lo, mate))
        recs.append((hi, read))
    recs.sort(key=lambda r: r[0])
    write_raw_bam(str(tmp_path / "junk.bam"), targets, [r for _, r in recs], rgs=["rg3"])
    (tmp_path / "cfg").write_text("readgroup:rg3\tplatform:illumina\tmap:junk.bam\treadlen:2.00\tlib:libJ\tnum:10\tlower:100.00\tupper:300.00\tmean:200.00\tstd:30.00\n"
                                  "readgroup:other\tplatform:illumina\tmap:junk.bam\treadlen:2.00\tlib:libA\tnum:10\tlower:100.00\tupper:300.00\tmean:200.00\tstd:30.00\n")
    # libraries in sorted name order (io/BamConfig.cpp:97-101): libA = 0, libJ = 1; rg3 -> libJ
    (rows, keys, checks), cols = both_readers(tmp_path, "cfg", "junk.bam", ["rg3", "other"], [1, 0], 0)
    i = [k for k in range(len(rows)) if rows[k, 1] == 29185299 and rows[k, 5] == 163][0]
    want = [22, 29185299, 22, 29184911, -388, 163, 2, 37, 1, 0]   # bdqual 37 = AM:C:37, not MAPQ 60 (Alignment.cpp:12-23); library of rg3
    assert rows[i].tolist() == want
    assert int(keys[i]) == hash_name(b"junk") and int(checks[i]) == check_name(b"junk")
    j = [k for k in range(len(cols["tid"])) if cols["pos"][k] == 29185299 and cols["flag"][k] == 163][0]
    got = [int(cols[c][j]) for c in ("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "mapq", "lib", "bam")]
    assert got == want
    assert int(cols["name_key"][j]) == hash_name(b"junk") and int(cols["name_check"][j]) == check_name(b"junk")
    assert len(rows) == len(cols["tid"]) == 6
    # TestAlignment.cpp:49-89: forward strand, not leftmost, |isize| 388
    assert not (want[5] & 0x10) and not (want[1] < want[3]) and abs(want[4]) == 388
    # ... and its FASTQ text, as -d writes it for a supporting read (TestAlignment.cpp:16-21): both readers
    for env in ({}, {"BDX_DECODE": "host"}):
        prefix = str(tmp_path / ("fq_" + ("host" if env else "dev")))
        p = subprocess.run([EXE, "-q", "0", "-y", "0", "-d", prefix, "cfg"], cwd=str(tmp_path), env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        rows_out = [l for l in p.stdout.decode().splitlines() if l and not l.startswith("#")]
        assert len(rows_out) == 1 and "\tITX\t" in rows_out[0], p.stdout.decode()
        text = "".join(open("%s.libJ.%s.fastq" % (prefix, k)).read() for k in ("1", "2") if os.path.exists("%s.libJ.%s.fastq" % (prefix, k)))
        assert "@junk\nCT\n+\nHB\n" in text, text


def test_TestBam_inline_sam_records_through_both_readers(tmp_path):
    # TestBam.cpp:36-51: name, flag, 1-based pos, MAPQ, CIGAR, 1-based mate pos, isize, sequence, qualities
    sam = [("P1", 99, 10, 60, [(5, 0), (5, 4)], 15, 15, "GTTTTTTTTT"), ("P1", 147, 15, 60, [(10, 0)], 10, -15, "GCCCCTTTTT"),
           ("P2", 99, 10, 60, [(6, 0), (4, 4)], 15, 15, "TGTTTTTTTT"), ("P2", 147, 15, 60, [(10, 0)], 10, -15, "CGCCCTTTTT"),
           ("P3", 99, 10, 60, [(10, 0)], 15, 15, "TTGTTTTTTT"), ("P3", 147, 15, 60, [(10, 0)], 10, -15, "CCGCCTTTTT")]
    order = [0, 2, 4, 1, 3, 5]   # coordinate order (@HD SO:coordinate): the three reads at 10, then the three at 15 -- as samtools sorts them
    recs = [make_record(n, 0, pos - 1, flag, mq, 0, mpos - 1, isz, seq, "HHHHHHHHHH", cig) for n, flag, pos, mq, cig, mpos, isz, seq in (sam[i] for i in order)]
    write_raw_bam(str(tmp_path / "t.bam"), ["21"], recs)
    (tmp_path / "cfg").write_text("readgroup:rg1\tplatform:illumina\tmap:t.bam\treadlen:10.00\tlib:lib1\tnum:6\tlower:5.00\tupper:30.00\tmean:15.00\tstd:3.00\n")
    (rows, keys, checks), cols = both_readers(tmp_path, "cfg", "t.bam", ["rg1"], [0], 0)
    assert len(rows) == len(cols["tid"]) == 6
    for k, i in enumerate(order):
        n, flag, pos, mq, cig, mpos, isz, seq = sam[i]
        want = [0, pos - 1, 0, mpos - 1, isz, flag, 10, mq, 0, 0]   # (no RG tag: the library of the alphabetically first BAM; no AM tag: bdqual = MAPQ)
        assert rows[k].tolist() == want
        assert [int(cols[c][k]) for c in ("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "mapq", "lib", "bam")] == want
        assert int(keys[k]) == int(cols["name_key"][k]) == hash_name(n.encode())
        # TestBam.cpp:84-93: the first read of every pair is leftmost, its mate is not
        assert (want[1] < want[3]) == (i % 2 == 0)
    # mates share their key, pairs differ
    assert len({int(x) for x in keys}) == 3
