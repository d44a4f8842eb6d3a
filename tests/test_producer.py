"""CPU tests of the host producer (BGZF/BAM decode, reader filter, RG->library, merge order, config parser):
bin/bdx-dump-reads against the independent pure-Python decode + the oracle's priority-queue merge."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, load_chr21, make_opts

DUMP = os.path.join(ROOT, "bin", "bdx-dump-reads")


def dump(args, cwd, env=None):
    if not os.path.exists(DUMP):
        import __graft_entry__ as g
        g.build()
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([DUMP] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=e).stdout.decode()
    head = [l for l in out.splitlines() if l.startswith("#")]
    data = [l.split("\t") for l in out.splitlines() if not l.startswith("#")]
    rows = np.array([[int(x) for x in f[:10]] for f in data], dtype=np.int64).reshape(-1, 10)
    keys = np.array([int(f[10]) for f in data], dtype=np.uint64)
    dump.checks = np.array([int(f[11]) for f in data], dtype=np.uint64)   # (the second name hash of the same call)
    return head, rows, keys


@pytest.mark.parametrize("chr_args,chr_tid", [([], -1), (["-o", "21"], 22)])
def test_producer_stream_equals_independent_decode(chr_args, chr_tid):
    run = load_chr21(make_opts(chr_tid=chr_tid)).run()
    head, rows, keys = dump(chr_args + ["inv_del_bam_config"], os.path.join(GOLDEN, "chr21"))
    soa = run.merged_soa()
    assert len(rows) == run.n_merged == 5917
    for col, k in enumerate(("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "bdqual", "lib", "bam")):
        want = soa[k].astype(np.int64)
        got = rows[:, col].astype(np.int64)
        np.testing.assert_array_equal(got, want, err_msg=k)
    # the name key is a function of the read name: equal ids <-> equal keys
    key, nid = keys, soa["name_id"]
    m = {}
    for k, i in zip(key.tolist(), nid.tolist()):
        assert m.setdefault(i, k) == k
    assert len(set(m.values())) == len(m)
    # config-derived header: W0 and the library table
    assert head[0].startswith("#w0=%d nlibs=2 nbams=2" % run.w0)
    for i in range(run.nlibs):
        f = head[1 + i].split("\t")
        assert f[2] == run.lib_names[i] and int(f[4]) == run.lib_i[i, 1] and int(f[10]) == run.lib_i[i, 0]
        np.testing.assert_array_equal(np.array([float(x) for x in f[5:10]], dtype=np.float32), run.lib_f[i])


def test_unknown_chromosome_is_an_error():
    p = subprocess.run([DUMP, "-o", "nope", "inv_del_bam_config"], cwd=os.path.join(GOLDEN, "chr21"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Failed to parse bam region 'nope'" in p.stderr


def test_missing_map_field_is_an_error(tmp_path):
    cfg = tmp_path / "cfg"
    cfg.write_text("readgroup:rg1\tlib:l1\tmean:400\tstd:30\n")
    p = subprocess.run([DUMP, str(cfg)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Required field 'map' not found in config at line 1!" in p.stderr


@pytest.mark.parametrize("env", [{}, {"BDX_BAM_PIECE_BLOCKS": "1"}, {"BDX_BAM_PIECE_BLOCKS": "3"}, {"BDX_BAM_PIECE_BLOCKS": "7"}])
def test_producer_decodes_a_multi_block_synthetic_bam(tmp_path, env):
    """a few hundred BGZF blocks, records straddling block boundaries, pieces of the file inflated and decoded by several
    threads from guessed record boundaries (the knob shrinks the pieces to 1 / 3 / 7 blocks: hundreds of guesses and of
    records that run into the next piece)"""
    from breakdancer_amd.bamwrite import write_bam
    from breakdancer_amd.synth import make_chromosome
    d = make_chromosome(length=300000, seed=5)
    write_bam(str(tmp_path / "syn.bam"), d, ["chrS"], seed=1)
    (tmp_path / "cfg").write_text("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
    head, rows, keys = dump(["cfg"], str(tmp_path), env)
    n = len(d["tid"])
    assert len(rows) == n and n > 80000
    for col, k in enumerate(("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "mapq", "lib", "bam")):
        np.testing.assert_array_equal(rows[:, col], d[k].astype(np.int64), err_msg=k)
    # mates share a key, distinct pairs do not
    m = {}
    for k, i in zip(keys.tolist(), d["name_key"].tolist()):
        assert m.setdefault(i, k) == k
    assert len(set(m.values())) == len(m)


@pytest.mark.parametrize("region,beg,end", [("21:29,185,000-29,186,200", 29184999, 29186200), ("21:34809000", 34808999, 1 << 29),
                                            ("21", 0, 1 << 29)])
def test_region_strings_follow_samtools_semantics(region, beg, end):
    """-o accepts samtools region strings (bam_aux.c:107-160); records overlapping [beg, end) are kept
    (bam_index.c:571-576), end of a record = pos + reference length of its CIGAR"""
    from helpers import read_bam
    gd = os.path.join(GOLDEN, "chr21")
    head, rows, keys = dump(["-o", region, "inv_del_bam_config"], gd)
    want = 0
    for b in ("NA19238_chr21_del_inv.bam", "NA19240_chr21_del_inv.bam"):
        targets, recs = read_bam(os.path.join(gd, b))
        tid = targets.index("21")
        want += int(((recs["tid"] == tid) & (recs["rend"] > beg) & (recs["pos"] < end)).sum())
    assert len(rows) == want and want > 0
    assert (rows[:, 0] == 22).all() and (rows[:, 1] < end).all()


def test_both_name_hashes_follow_their_definitions(tmp_path):
    """name_key and name_check of the host reader against the Python restatements of the two functions (tests/namehash.py; the
    device-side decoder is held to the same restatements in test_gpu_bamdec.py) at name lengths that cover every tail of the
    8-byte word loop, and: the two are independent of each other"""
    from breakdancer_amd.bamwrite import write_bam
    from breakdancer_amd.synth import make_chromosome
    from namehash import check_name, hash_name
    d = make_chromosome(length=20000, seed=3)
    all_keys, all_checks, all_names = [], [], []
    for width in (1, 2, 3, 5, 7, 8, 9, 12, 15, 16, 17, 24, 31, 40):
        names = [("%032x" % (int(k) * 0x9E3779B97F4A7C15 % (1 << 128))).rjust(width, "g")[-width:] for k in d["name_key"].tolist()]
        write_bam(str(tmp_path / "syn.bam"), d, ["chrS"], seed=1, names=names)
        (tmp_path / "cfg").write_text("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
        head, rows, keys = dump(["cfg"], str(tmp_path))
        assert len(keys) == len(names)
        np.testing.assert_array_equal(keys, np.array([hash_name(n.encode()) for n in names], dtype=np.uint64), err_msg=str(width))
        np.testing.assert_array_equal(dump.checks, np.array([check_name(n.encode()) for n in names], dtype=np.uint64), err_msg=str(width))
        all_keys += keys.tolist(); all_checks += dump.checks.tolist(); all_names += names
    # no fixed relation between the two: the xor of key and check takes as many values as there are names
    assert len({k ^ c for k, c in zip(all_keys, all_checks)}) == len(set(all_names))


def test_read_names_the_boundary_guess_rejects_are_still_decoded(tmp_path):
    """read names with bytes outside the printable range defeat the record-boundary guess of every piece; the consumer then
    decodes the piece again from the true boundary -- same stream"""
    from breakdancer_amd.bamwrite import write_bam
    from breakdancer_amd.synth import make_chromosome
    d = make_chromosome(length=120000, seed=9)
    n = len(d["tid"])
    names = ["r\x01%07d" % int(k % 9999991) for k in d["name_key"]]
    write_bam(str(tmp_path / "syn.bam"), d, ["chrS"], seed=1, names=names)
    (tmp_path / "cfg").write_text("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
    for env in ({}, {"BDX_BAM_PIECE_BLOCKS": "2"}):
        head, rows, keys = dump(["cfg"], str(tmp_path), env)
        assert len(rows) == n
        for col, k in enumerate(("tid", "pos", "mtid", "mpos", "isize", "flag")):
            np.testing.assert_array_equal(rows[:, col], d[k].astype(np.int64), err_msg=k)


def test_two_files_merge_in_the_reference_order_at_every_piece_size():
    run = load_chr21(make_opts()).run()
    soa = run.merged_soa()
    for env in ({"BDX_BAM_PIECE_BLOCKS": "1"}, {"BDX_BAM_PIECE_BLOCKS": "2"}):
        head, rows, keys = dump(["inv_del_bam_config"], os.path.join(GOLDEN, "chr21"), env)
        assert len(rows) == run.n_merged
        for col, k in enumerate(("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "bdqual", "lib", "bam")):
            np.testing.assert_array_equal(rows[:, col], soa[k].astype(np.int64), err_msg=k)


@pytest.mark.parametrize("region", ["c2", "c2:9000-21000", "c3:15,000", "c1:1-4000"])
def test_region_through_the_bam_index_equals_the_full_scan(tmp_path, region):
    """-o with a .bai next to the BAMs: decoding starts at the region (linear index) and stops behind it; the stream must be
    the one the full scan filters out of the whole file, piece size 1 block or default"""
    from breakdancer_amd.bamwrite import write_bam_records
    from fuzzgen import make_case
    cfg, streams, targets = make_case(77, n_pairs=2500)
    for fn, st in zip(("a.bam", "b.bam"), streams):
        recs = [dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i], flag=st["flag"][i],
                     qlen=st["qlen"][i], mapq=int(st["bdqual"][i]), rg=st["rg"][i], name="read%d" % int(st["name_id"][i])) for i in range(len(st["tid"]))]
        write_bam_records(str(tmp_path / fn), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=3, index=True)
    (tmp_path / "cfg").write_text(cfg)
    _, full, kfull = dump(["-o", region, "cfg"], str(tmp_path), {"BDX_BAM_NO_INDEX": "1"})
    assert len(full) > 50
    for env in ({}, {"BDX_BAM_PIECE_BLOCKS": "1"}):
        _, rows, keys = dump(["-o", region, "cfg"], str(tmp_path), env)
        np.testing.assert_array_equal(rows, full)
        np.testing.assert_array_equal(keys, kfull)


def test_merge_order_over_key_columns_equals_the_streaming_merge():
    """the device path's merge (host/producer.cpp merge_order: the reference's priority queue over the files' (tid, pos, strand)
    columns, with runs below the queue's top taken without touching it) against the streaming merge of the host reader, on the
    reference's two BAMs -- whose records tie in position across the files -- through the tool that prints both"""
    head, rows, keys = dump(["inv_del_bam_config"], os.path.join(GOLDEN, "chr21"))
    assert len(rows) == 5917
    # the two-file rule in one piece, the same cut into up to 64 pieces merged by threads of their own, the priority queue itself
    for how in ("columns", "pieces", "queue"):
        head2, rows2, keys2 = dump(["inv_del_bam_config"], os.path.join(GOLDEN, "chr21"), {"BDX_DUMP_MERGE": how})
        np.testing.assert_array_equal(rows, rows2, err_msg=how)
        np.testing.assert_array_equal(keys, keys2, err_msg=how)
    # ties across the two files exist in this input (equal tid and pos, records of both files)
    t = rows[:, 0] * (1 << 32) + rows[:, 1]
    same = (t[1:] == t[:-1]) & (rows[1:, 9] != rows[:-1, 9])
    assert same.sum() > 10


@pytest.mark.parametrize("nfiles,seed", [(2, 1), (2, 7), (2, 8), (3, 2), (5, 3)])
def test_merge_order_with_heavy_ties_across_several_files(tmp_path, nfiles, seed):
    """2, 3 and 5 files whose records crowd on a few positions and both strands: the order among equal keys is whatever the
    reference's priority queue does with them -- merge_order must reproduce the streaming merge record for record"""
    from breakdancer_amd.bamwrite import write_bam_records
    rng = np.random.default_rng(seed)
    lines = []
    for b in range(nfiles):
        n = int(rng.integers(300, 900))
        tid = np.sort(rng.integers(0, 2, n))
        pos = rng.integers(0, 40, n) * 50
        order = np.lexsort((pos, tid))
        recs = [dict(tid=int(tid[i]), pos=int(pos[i]), mtid=int(tid[i]), mpos=int(pos[i]) + 300, isize=400, flag=int(rng.choice([99, 147, 83, 163, 97, 145])), qlen=50,
                     mapq=40, am=None, rg="g%d" % b, name="f%dr%d" % (b, i)) for i in order]
        write_bam_records(str(tmp_path / ("f%d.bam" % b)), recs, ["c1", "c2"], rgs=("g%d" % b,), seed=b)
        lines.append("readgroup:g%d\tplatform:illumina\tmap:f%d.bam\treadlen:50.00\tlib:lib%d\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n" % (b, b, b))
    (tmp_path / "cfg").write_text("".join(lines))
    head, rows, keys = dump(["cfg"], str(tmp_path))
    assert len(rows) > 300 * nfiles
    for how in ("columns", "pieces", "queue"):   # (two files: the one-bit rule, whole and in pieces; more: the queue either way)
        head2, rows2, keys2 = dump(["cfg"], str(tmp_path), {"BDX_DUMP_MERGE": how})
        np.testing.assert_array_equal(rows, rows2, err_msg=how)
        np.testing.assert_array_equal(keys, keys2, err_msg=how)
    t = rows[:, 0] * (1 << 32) + rows[:, 1]
    assert ((t[1:] == t[:-1]) & (rows[1:, 9] != rows[:-1, 9])).sum() > 100   # neighbours with one key from two files


def test_a_block_that_fails_its_crc_is_an_error(tmp_path):
    """a member whose stored CRC-32 does not match what it inflates to (here: the CRC word itself is changed; a flipped bit in a stored
    or literal-only block looks the same): the host reader reports a corrupt file instead of decoding on"""
    import shutil
    import struct
    src = os.path.join(GOLDEN, "chr21")
    for f in ("NA19238_chr21_del_inv.bam", "NA19240_chr21_del_inv.bam", "inv_del_bam_config"):
        shutil.copy(os.path.join(src, f), str(tmp_path / f))
    path = str(tmp_path / "NA19238_chr21_del_inv.bam")
    data = bytearray(open(path, "rb").read())
    off, k = 0, 0
    while off < len(data):   # the third member's footer
        bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
        if k == 2:
            data[off + bsize - 8] ^= 0x40
            break
        off += bsize
        k += 1
    open(path, "wb").write(bytes(data))
    p = subprocess.run([DUMP, "inv_del_bam_config"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"CRC-32" in p.stderr
    p = subprocess.run([DUMP, "inv_del_bam_config"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BDX_BAM_NO_CRC="1"))
    assert p.returncode == 0
