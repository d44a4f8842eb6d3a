"""GPU tests: multi-BAM fuzz inputs written as real BAM files, run through bin/breakdancer-max (BGZF decode, aux tags,
reader filter, RG->library fallback, k-way merge tie order, GPU path, formatter) and compared with the oracle's rendering
of the same records."""
import os
import subprocess

import numpy as np
import pytest

from fuzzgen import make_case
from helpers import ROOT, filter_cmd_lines, make_opts
from runner import oracle_case

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "bin", "breakdancer-max")

FLAGSETS = [([], dict()), (["-a", "-h"], dict(cn_lib=1, print_af=1)), (["-t"], dict(transchr_rearrange=1)),
            (["-o", "c2"], dict(chr_tid=1)), (["-b", "2", "-r", "1"], dict(buffer_size=2, min_read_pair=1)),
            (["-q", "36", "-s", "0"], dict(min_map_qual=36, min_len=0)), (["-l"], dict(illumina_long_insert=1)),
            (["-m", "900", "-x", "3", "-c", "2", "-f"], dict(max_sd=900, seq_coverage_lim=3, cut_sd=2, fisher=1))]


def write_case(tmp, streams, targets, rng, index=False):
    from breakdancer_amd.bamwrite import write_bam_records
    for b, (fn, st) in enumerate(zip(("a.bam", "b.bam"), streams)):
        recs = []
        for i in range(len(st["tid"])):
            bq, q = int(st["bdqual"][i]), int(st["bdqual"][i])
            am = None
            if rng.random() < 0.5:   # bdqual through the AM tag, MAPQ something else
                am, q = bq, int(rng.integers(0, 61))
            recs.append(dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i],
                             flag=st["flag"][i], qlen=st["qlen"][i], mapq=q, am=am, rg=st["rg"][i], name="read%d" % int(st["name_id"][i])))
            if rng.random() < 0.02:  # a secondary and a supplementary copy: dropped by the reader filter
                extra = dict(recs[-1])
                extra["flag"] = int(extra["flag"]) | (0x100 if rng.random() < 0.5 else 0x800)
                recs.append(extra)
        write_bam_records(os.path.join(tmp, fn), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=b, index=index)


@pytest.mark.parametrize("seed", range(10))
def test_cli_on_fuzz_bams_equals_oracle(seed, tmp_path):
    rng = np.random.default_rng(seed)
    cfg, streams, targets = make_case(500 + seed)
    write_case(str(tmp_path), streams, targets, rng)
    (tmp_path / "cfg").write_text(cfg)
    for args, kw in (FLAGSETS[seed % len(FLAGSETS)], FLAGSETS[(3 * seed + 1) % len(FLAGSETS)]):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **kw))
        # both files decoded on the GPU, merged by a gather in the order the host works out from (tid, pos, strand) -- and the host
        # reader's merge of the same files
        for label, env in (("device", dict(BDX_TIMING="1")), ("device-small-pieces", dict(BDX_TIMING="1", BDX_BAM_PIECE_BYTES="50000", BDX_BAM_BATCH_BLOCKS="2")),
                           ("host", dict(BDX_TIMING="1", BDX_DECODE="host"))):
            p = subprocess.run([EXE, "-y", "-1"] + args + ["cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert p.returncode == 0, (label, p.stderr.decode())
            assert ("2 files decoded on the GPU" in p.stderr.decode()) == label.startswith("device"), (label, p.stderr.decode())
            assert filter_cmd_lines(p.stdout.decode()) == filter_cmd_lines(run.text), (label, args, p.stderr.decode())
        if "-o" not in args:  # the same run with the chromosomes spread over ranks (here: threads sharing the one GPU)
            env = dict(os.environ, BDX_GPUS="0,0,0" if seed % 2 else "0,0")
            p = subprocess.run([EXE, "-y", "-1"] + args + ["cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 0, p.stderr.decode()
            assert filter_cmd_lines(p.stdout.decode()) == filter_cmd_lines(run.text), ("sharded", args, p.stderr.decode())


@pytest.mark.parametrize("seed", range(8))
def test_cli_one_bam_decoded_on_the_gpu_equals_oracle(seed, tmp_path):
    """a configuration of ONE BAM takes the device path (bdx_bamdec_*: BGZF inflate, record boundaries, fields, RG -> library and
    reader filter on the GPU); same text as the oracle's rendering and as the host reader's (BDX_DECODE=host), also with pieces of a
    few members and a ring that wraps, and the -g / -d dumps agree between the two readers"""
    rng = np.random.default_rng(900 + seed)
    cfg, streams, targets = make_case(700 + seed, n_pairs=int(rng.integers(800, 4000)) if seed else 30000)   # (seed 0: laps round the small ring)
    cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
    write_case(str(tmp_path), streams[:1], targets, rng)
    (tmp_path / "cfg").write_text(cfg1)
    for args, kw in (FLAGSETS[seed % len(FLAGSETS)], FLAGSETS[(3 * seed + 2) % len(FLAGSETS)]):
        run = oracle_case(cfg1, streams[:1], targets, make_opts(score_threshold=-1, **kw))
        texts = {}
        # (the stream arrangements of the decoder that are kept as switches -- inflate launches on a stream of their own, with and without
        # queue priorities -- decode the same records: small batches, so that inflate launches and record stages do run beside each other)
        for label, env in (("device", dict(BDX_TIMING="1")), ("device-small-pieces", dict(BDX_TIMING="1", BDX_BAM_PIECE_BYTES="100000", BDX_BAM_BATCH_BLOCKS="3", BDX_BAM_RING_BYTES="1048576")),
                           ("device-own-inflate-stream", dict(BDX_TIMING="1", BDX_KZ_STREAM="own", BDX_BAM_PIECE_BYTES="100000", BDX_BAM_BATCH_BLOCKS="3")),
                           ("device-stream-priorities", dict(BDX_TIMING="1", BDX_KZ_STREAM="prio", BDX_BAM_PIECE_BYTES="100000", BDX_BAM_BATCH_BLOCKS="3")),
                           ("host", dict(BDX_TIMING="1", BDX_DECODE="host"))):
            if label in ("device-own-inflate-stream", "device-stream-priorities") and seed > 1:
                continue
            p = subprocess.run([EXE, "-y", "-1"] + args + ["cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert p.returncode == 0, (label, p.stderr.decode())
            assert ("on the GPU" in p.stderr.decode()) == label.startswith("device"), p.stderr.decode()
            texts[label] = filter_cmd_lines(p.stdout.decode())
            assert texts[label] == filter_cmd_lines(run.text), (label, args, p.stderr.decode())
    # supporting reads of the SVs (BED and FASTQ dumps): the stream indices the dumps are fetched by are the same
    outs = {}
    for label, env in (("device", dict()), ("host", dict(BDX_DECODE="host"))):
        d = tmp_path / label
        d.mkdir()
        p = subprocess.run([EXE, "-y", "-1", "-g", str(d / "out.bed"), "-d", str(d / "fq"), "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()
        outs[label] = {f: open(os.path.join(str(d), f), "rb").read() for f in sorted(os.listdir(str(d)))}
    assert outs["device"] == outs["host"] and outs["device"]


def test_cli_two_bams_one_of_them_without_records(tmp_path):
    """a configuration whose second BAM holds a header and nothing else: decoded on the GPU like the other (zero records from it), merged,
    same text as the oracle and as the host reader"""
    rng = np.random.default_rng(77)
    cfg, streams, targets = make_case(577)
    empty = {k: (v[:0] if hasattr(v, "__len__") and not isinstance(v, str) else v) for k, v in streams[1].items()}
    streams = [streams[0], empty]
    write_case(str(tmp_path), streams, targets, rng)
    (tmp_path / "cfg").write_text(cfg)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1))
    for label, env in (("device", dict(BDX_TIMING="1")), ("host", dict(BDX_TIMING="1", BDX_DECODE="host"))):
        p = subprocess.run([EXE, "-y", "-1", "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode())
        assert ("2 files decoded on the GPU" in p.stderr.decode()) == (label == "device"), p.stderr.decode()
        assert filter_cmd_lines(p.stdout.decode()) == filter_cmd_lines(run.text), (label, p.stderr.decode())
    # the file without records as a configuration of its own: the statistics lines, no rows, on either reader
    cfg_b = "".join(l + "\n" for l in cfg.splitlines() if "map:b.bam" in l)
    (tmp_path / "cfgb").write_text(cfg_b)
    texts = []
    for env in (dict(), dict(BDX_DECODE="host")):
        p = subprocess.run([EXE, "-y", "-1", "cfgb"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()
        texts.append(filter_cmd_lines(p.stdout.decode()))
    assert texts[0] == texts[1] and not [l for l in texts[0].splitlines() if not l.startswith("#")]


def test_cli_record_larger_than_the_device_path_takes_goes_to_the_host_reader(tmp_path):
    """one read of 3 M bases (a 4.5 MB record: over the 4 MiB the device-side boundary search spans) among ordinary ones: the decoder
    reports it, the CLI hands the file to the host reader -- same table as with BDX_DECODE=host, no error"""
    from breakdancer_amd.bamwrite import write_bam_records
    rng = np.random.default_rng(5)
    cfg, streams, targets = make_case(590, n_pairs=600)
    cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
    st = streams[0]
    recs = [dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i], flag=st["flag"][i], qlen=st["qlen"][i],
                 mapq=int(st["bdqual"][i]), am=None, rg=st["rg"][i], name="read%d" % int(st["name_id"][i])) for i in range(len(st["tid"]))]
    big = dict(recs[len(recs) // 2])
    big["qlen"] = 3_000_000
    big["flag"] = int(big["flag"]) | 0x100   # (secondary: the reader filter drops it, so the table does not depend on it)
    recs.insert(len(recs) // 2, big)
    write_bam_records(str(tmp_path / "a.bam"), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=1)
    (tmp_path / "cfg").write_text(cfg1)
    out = {}
    for label, env in (("default", dict(BDX_TIMING="1")), ("host", dict(BDX_TIMING="1", BDX_DECODE="host"))):
        p = subprocess.run([EXE, "-y", "-1", "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode()[-600:])
        assert "host decode threads" in p.stderr.decode(), (label, p.stderr.decode()[-600:])
        out[label] = filter_cmd_lines(p.stdout.decode())
    assert out["default"] == out["host"] and [l for l in out["host"].splitlines() if not l.startswith("#")]


def test_cli_header_longer_than_a_bgzf_member(tmp_path):
    """6,000 reference sequences: the header fills three BGZF members, the first record starts in the middle of one -- the device path
    is told where (member and offset) by the host's header parse; same table as the oracle's and the host reader's"""
    rng = np.random.default_rng(9)
    cfg, streams, targets = make_case(595, n_pairs=1500)
    many = list(targets) + ["extra_contig_%05d" % i for i in range(6000)]
    cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
    write_case(str(tmp_path), streams[:1], many, rng)
    (tmp_path / "cfg").write_text(cfg1)
    assert os.path.getsize(str(tmp_path / "a.bam")) > 40000
    run = oracle_case(cfg1, streams[:1], many, make_opts(score_threshold=-1))
    for label, env in (("device", dict(BDX_TIMING="1")), ("device-small-pieces", dict(BDX_TIMING="1", BDX_BAM_PIECE_BYTES="70000", BDX_BAM_BATCH_BLOCKS="2")),
                       ("host", dict(BDX_TIMING="1", BDX_DECODE="host"))):
        p = subprocess.run([EXE, "-y", "-1", "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode()[-600:])
        assert ("on the GPU" in p.stderr.decode()) == label.startswith("device"), p.stderr.decode()[-600:]
        assert filter_cmd_lines(p.stdout.decode()) == filter_cmd_lines(run.text), (label, p.stderr.decode()[-600:])


def test_cli_records_spanning_several_bgzf_members(tmp_path):
    """reads of 100 k and 400 k bases (records of 150 KB and 600 KB: three and ten BGZF members each) among ordinary ones, at piece and
    batch sizes that put their pieces into different batches: decoded on the GPU, same table as the host reader's"""
    from breakdancer_amd.bamwrite import write_bam_records
    cfg, streams, targets = make_case(596, n_pairs=900)
    cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
    st = streams[0]
    recs = [dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i], flag=st["flag"][i], qlen=st["qlen"][i],
                 mapq=int(st["bdqual"][i]), am=None, rg=st["rg"][i], name="read%d" % int(st["name_id"][i])) for i in range(len(st["tid"]))]
    for at, ql in ((len(recs) // 3, 100_000), (2 * len(recs) // 3, 400_000)):
        big = dict(recs[at])
        big["qlen"] = ql
        big["flag"] = int(big["flag"]) | 0x100   # (secondary: dropped by the reader filter on both sides)
        recs.insert(at, big)
    write_bam_records(str(tmp_path / "a.bam"), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=2)
    (tmp_path / "cfg").write_text(cfg1)
    out = {}
    for label, env in (("device", dict(BDX_TIMING="1")), ("device-small-pieces", dict(BDX_TIMING="1", BDX_BAM_PIECE_BYTES="70000", BDX_BAM_BATCH_BLOCKS="16", BDX_BAM_RING_BYTES="8388608")),
                       ("host-after-all", dict(BDX_TIMING="1", BDX_BAM_PIECE_BYTES="70000", BDX_BAM_BATCH_BLOCKS="2", BDX_BAM_RING_BYTES="4194304")),
                       ("host", dict(BDX_TIMING="1", BDX_DECODE="host"))):
        p = subprocess.run([EXE, "-y", "-1", "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode()[-600:])
        # (batches of two members are shorter than the 600 KB record: the decoder says so and the host reader takes the file)
        assert ("on the GPU" in p.stderr.decode()) == label.startswith("device"), (label, p.stderr.decode()[-600:])
        out[label] = filter_cmd_lines(p.stdout.decode())
    assert out["device"] == out["host"] == out["device-small-pieces"] == out["host-after-all"] and [l for l in out["host"].splitlines() if not l.startswith("#")]


@pytest.mark.parametrize("region,seed", [("c2", 1), ("c2:9000-21000", 2), ("c1:1-6000", 3), ("c3:15,000", 4)])
def test_cli_region_through_the_index_on_the_device_reader(region, seed, tmp_path):
    """-o with a .bai beside the BAM, decoded on the GPU: decoding starts at the region and STOPS behind it -- the feeder hands over no
    further piece, and a record that begins in the last member handed over and ends behind it (records are not aligned to members in
    this file, as in htsjdk's / sambamba's output) is dropped, not an error that sends the file to the host reader (io/RegionLimitedBamReader.hpp:43-71)"""
    from breakdancer_amd.bamwrite import write_bam_records
    cfg, streams, targets = make_case(810 + seed, n_pairs=40000)
    cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
    st = streams[0]
    recs = [dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i], flag=st["flag"][i],
                 qlen=st["qlen"][i], mapq=int(st["bdqual"][i]), rg=st["rg"][i], name="read%d" % int(st["name_id"][i])) for i in range(len(st["tid"]))]
    write_bam_records(str(tmp_path / "a.bam"), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=seed, index=True)
    assert os.path.getsize(str(tmp_path / "a.bam")) > 8 * 65536 and os.path.exists(str(tmp_path / "a.bam.bai"))
    (tmp_path / "cfg").write_text(cfg1)
    texts = {}
    for label, env in (("device", dict(BDX_TIMING="1", BDX_BAM_PIECE_BYTES="131072", BDX_BAM_BATCH_BLOCKS="2")), ("device-default", dict(BDX_TIMING="1")),
                       ("host", dict(BDX_TIMING="1", BDX_DECODE="host")), ("host-full-scan", dict(BDX_TIMING="1", BDX_DECODE="host", BDX_BAM_NO_INDEX="1"))):
        p = subprocess.run([EXE, "-y", "-1", "-o", region, "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode())
        assert ("on the GPU" in p.stderr.decode()) == label.startswith("device"), (label, p.stderr.decode())   # (no silent hand-over to the host reader)
        texts[label] = filter_cmd_lines(p.stdout.decode())
    assert texts["device"] == texts["device-default"] == texts["host"] == texts["host-full-scan"]
    assert [l for l in texts["host"].splitlines() if not l.startswith("#")]


@pytest.mark.parametrize("seed", range(6))
def test_cli_sharded_run_decodes_every_ranks_chromosomes_on_its_own_gpu(seed, tmp_path):
    """BDX_GPUS with ONE indexed BAM: every rank pulls the BGZF ranges of its chromosomes through the .bai and decodes them on its GPU
    (bdx_bamdec_* with the rank's context as sink) -- same text as the oracle's whole-genome run, as the host producer's routing
    (no index / BDX_DECODE=host) and as the single-GPU run"""
    from breakdancer_amd.bamwrite import write_bam_records
    rng = np.random.default_rng(40 + seed)
    cfg, streams, targets = make_case(860 + seed, n_pairs=int(rng.integers(1500, 6000)))
    cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
    st = streams[0]
    recs = [dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i], flag=st["flag"][i],
                 qlen=st["qlen"][i], mapq=int(st["bdqual"][i]), rg=st["rg"][i], name="read%d" % int(st["name_id"][i])) for i in range(len(st["tid"]))]
    write_bam_records(str(tmp_path / "a.bam"), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=seed, index=True)
    (tmp_path / "cfg").write_text(cfg1)
    args, kw = FLAGSETS[(2 * seed) % len(FLAGSETS)]
    if "-o" in args:
        args, kw = [], dict()
    run = oracle_case(cfg1, streams[:1], targets, make_opts(score_threshold=-1, **kw))
    want = filter_cmd_lines(run.text)
    gpus = "0,0,0" if seed % 2 else "0,0"
    for label, env, marker in (("device", dict(BDX_GPUS=gpus, BDX_TIMING="1"), "on its own GPU"),
                               ("device-small-pieces", dict(BDX_GPUS=gpus, BDX_TIMING="1", BDX_BAM_PIECE_BYTES="100000", BDX_BAM_BATCH_BLOCKS="3"), "on its own GPU"),
                               ("host-routing", dict(BDX_GPUS=gpus, BDX_TIMING="1", BDX_DECODE="host"), "routed to the ranks"),
                               ("no-index", dict(BDX_GPUS=gpus, BDX_TIMING="1", BDX_BAM_NO_INDEX="1"), "routed to the ranks"),
                               ("one-gpu", dict(BDX_TIMING="1"), "on the GPU")):
        p = subprocess.run([EXE, "-y", "-1"] + args + ["cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode())
        assert marker in p.stderr.decode(), (label, p.stderr.decode())
        assert filter_cmd_lines(p.stdout.decode()) == want, (label, args, p.stderr.decode())
    # -g / -d through the sharded device reader: the dumps' stream indices are positions in the merged stream of the whole file
    outs = {}
    for label, env in (("sharded", dict(BDX_GPUS=gpus)), ("one", dict())):
        d = tmp_path / label
        d.mkdir()
        p = subprocess.run([EXE, "-y", "-1", "-g", str(d / "out.bed"), "-d", str(d / "fq"), "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()
        outs[label] = {f: open(os.path.join(str(d), f), "rb").read() for f in sorted(os.listdir(str(d)))}
    assert outs["sharded"] == outs["one"] and outs["one"]


@pytest.mark.parametrize("seed", range(6))
def test_cli_sharded_run_of_two_indexed_bams_decodes_and_merges_on_the_ranks_gpus(seed, tmp_path):
    """BDX_GPUS with TWO indexed BAMs (the tumour / normal shape of configs[4]): per rank and chromosome one decoder per file over that file's
    range of the .bai, BamMerger's order from three columns (ties between the files included: the fuzz cases put reads of both files on the same
    positions and strands), one gather behind what the rank's store holds -- the oracle's text, the single-GPU run's, the host producer's routing"""
    rng = np.random.default_rng(140 + seed)
    cfg, streams, targets = make_case(940 + seed, n_pairs=int(rng.integers(1500, 5000)))
    write_case(str(tmp_path), streams, targets, rng, index=True)
    assert os.path.exists(str(tmp_path / "a.bam.bai")) and os.path.exists(str(tmp_path / "b.bam.bai"))
    (tmp_path / "cfg").write_text(cfg)
    # (positions shared by the two files exist in this case: the order of ties is on trial)
    a, b = streams[0], streams[1]
    assert set(zip(a["tid"].tolist(), a["pos"].tolist())) & set(zip(b["tid"].tolist(), b["pos"].tolist()))
    gpus = "0,0,0" if seed % 2 else "0,0"
    for args, kw in (FLAGSETS[1], FLAGSETS[(2 * seed) % len(FLAGSETS)]):
        if "-o" in args:
            args, kw = [], dict()
        want = filter_cmd_lines(oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **kw)).text)
        for label, env, marker in (("device", dict(BDX_GPUS=gpus, BDX_TIMING="1"), "on its own GPU"),
                                   ("device-small-pieces", dict(BDX_GPUS=gpus, BDX_TIMING="1", BDX_BAM_PIECE_BYTES="100000", BDX_BAM_BATCH_BLOCKS="3"), "on its own GPU"),
                                   ("host-routing", dict(BDX_GPUS=gpus, BDX_TIMING="1", BDX_DECODE="host"), "routed to the ranks"),
                                   ("one-gpu", dict(BDX_TIMING="1"), "2 files decoded on the GPU")):
            p = subprocess.run([EXE, "-y", "-1"] + args + ["cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert p.returncode == 0, (label, p.stderr.decode())
            assert marker in p.stderr.decode(), (label, p.stderr.decode())
            if label.startswith("device"):
                assert "(2 files)" in p.stderr.decode(), p.stderr.decode()
            assert filter_cmd_lines(p.stdout.decode()) == want, (label, args, p.stderr.decode())
    # -g / -d: the supporting reads' stream indices are positions in the merged stream of both files
    outs = {}
    for label, env in (("sharded", dict(BDX_GPUS=gpus)), ("one", dict())):
        d = tmp_path / label
        d.mkdir()
        p = subprocess.run([EXE, "-y", "-1", "-g", str(d / "out.bed"), "-d", str(d / "fq"), "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()
        outs[label] = {f: open(os.path.join(str(d), f), "rb").read() for f in sorted(os.listdir(str(d)))}
    assert outs["sharded"] == outs["one"] and outs["one"]


def test_cli_file_of_tens_of_thousands_of_tiny_members(tmp_path):
    """a BAM re-blocked into BGZF members of 160 inflated bytes (33 k members in 6 MB: more members than a piece's table holds,
    512 bytes per member on average being the device path's assumption): the device path says so and the host reader takes the
    file -- the same table as from the ordinary blocking, no error"""
    import gzip
    import struct
    from breakdancer_amd.bamwrite import _bgzf_block, _EOF, write_bam_records
    cfg, streams, targets = make_case(905, n_pairs=17000)
    cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
    st = streams[0]
    recs = [dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i], flag=st["flag"][i],
                 qlen=st["qlen"][i], mapq=int(st["bdqual"][i]), rg=st["rg"][i], name="read%d" % int(st["name_id"][i])) for i in range(len(st["tid"]))]
    write_bam_records(str(tmp_path / "big.bam"), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=5)
    raw = gzip.decompress((tmp_path / "big.bam").read_bytes())
    with open(str(tmp_path / "a.bam"), "wb") as f:
        n_members = 0
        for i in range(0, len(raw), 160):
            f.write(_bgzf_block(raw[i:i + 160], 1))
            n_members += 1
        f.write(_EOF)
    assert n_members > 8 * 1024 * 1024 // 512 + 4096 and os.path.getsize(str(tmp_path / "a.bam")) < 8 * 1024 * 1024
    (tmp_path / "cfg").write_text(cfg1)
    (tmp_path / "cfg_big").write_text(cfg1.replace("map:a.bam", "map:big.bam"))
    texts = {}
    for label, cfgname, env in (("tiny", "cfg", dict(BDX_TIMING="1")), ("tiny-host", "cfg", dict(BDX_TIMING="1", BDX_DECODE="host")), ("ordinary", "cfg_big", dict(BDX_TIMING="1"))):
        p = subprocess.run([EXE, "-y", "-1", cfgname], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode())
        texts[label] = (filter_cmd_lines(p.stdout.decode()), p.stderr.decode())
    assert "host decode threads" in texts["tiny"][1] and "on the GPU" in texts["ordinary"][1]
    strip = lambda t: t.replace("big.bam", "a.bam")
    assert texts["tiny"][0] == texts["tiny-host"][0] == strip(texts["ordinary"][0])
    assert [l for l in texts["tiny"][0].splitlines() if not l.startswith("#")]


@pytest.mark.parametrize("fraction,min_rows", [(0.004, 1000), (0.02, 5000)])
def test_cli_prints_the_same_table_every_time(tmp_path, fraction, min_rows):
    """The same BAM through the CLI thirty times -- ten runs each on 1, 2 and 3 ranks (BDX_GPUS) -- prints ONE table.  Sizes: a file of fewer
    than four decoder batches (3.7 M records: the thread that sizes the later stages is started by the decoder's LAST feed -- round 5's
    sizing pass signalled through a flag on the context that a run beside it saw, and four runs in ten ended with an EMPTY table and no
    error, tools/determinism_probe.py) and one of four to eight batches (18 M records: the sizing thread runs beside the decode).  The
    genome share (116 M records) goes through the same probe outside the suite: profiles/r06_determinism_probe.txt."""
    from breakdancer_amd.bamwrite import write_genome_bam
    bam, cfg, n = write_genome_bam(str(tmp_path), fraction)
    assert n > (1 << 20)
    texts = {}
    for gpus in (None, "0,0", "0,0,0"):
        env = dict(os.environ, BDX_FOREGROUND="1")
        if gpus:
            env["BDX_GPUS"] = gpus
        for _ in range(10):
            p = subprocess.run([EXE, cfg], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 0, p.stderr.decode()
            texts.setdefault(filter_cmd_lines(p.stdout.decode()), []).append(gpus or "one GPU")
    assert len(texts) == 1, {k[:80]: v for k, v in texts.items()}
    rows = [l for l in next(iter(texts)).splitlines() if l and not l.startswith("#")]
    assert len(rows) > min_rows


@pytest.mark.parametrize("a_emits_last", [True, False])
def test_cli_sharded_two_bams_tie_at_a_chromosomes_first_position(tmp_path, a_emits_last):
    """The reference merges the files through ONE queue (io/BamMerger.cpp:40-126): where the two files' first records of a chromosome have
    the same position and strand, the file that has been waiting at the top wins -- the one that did NOT emit the genome's last record in
    front of the chromosome.  A sharded run merges chromosome by chromosome, on whatever rank owns each: the file that emitted last is worked
    out from the files' last records on the chromosome before (here another rank's).  The tie is made to matter: file a's first record on c2
    opens a region (an inter-chromosomal cluster's mates), file b's is a proper read at the same position and strand -- whether it is counted
    in front of the region's first read decides b.bam's copy number between the SV's regions, hence the row's allele frequency (-h).  Oracle (one queue) == sharded == one GPU, both ways."""
    from breakdancer_amd.bamwrite import write_bam_records
    targets = ["c1", "c2", "c3"]
    cfg = "".join("readgroup:%s\tplatform:illumina\tmap:%s\treadlen:100.00\tlib:%s\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n" % x
                  for x in (("rg1", "a.bam", "libA"), ("rg3", "b.bam", "libC")))
    rows = {0: [], 1: []}   # (tid, pos, mtid, mpos, isize, flag, name_id) per file
    nid = [0]

    def pair(f, tid, p1, mtid, p2, rev1, rev2, proper):
        nid[0] += 1
        ins = (p2 + 100 - p1) if tid == mtid else 0
        fl1 = 0x1 | (0x2 if proper else 0) | (0x10 if rev1 else 0) | (0x20 if rev2 else 0) | 0x40
        fl2 = 0x1 | (0x2 if proper else 0) | (0x10 if rev2 else 0) | (0x20 if rev1 else 0) | 0x80
        rows[f].append((tid, p1, mtid, p2, ins, fl1, nid[0]))
        rows[f].append((mtid, p2, tid, p1, -ins, fl2, nid[0]))
    rng = np.random.default_rng(77)
    for f in (0, 1):   # normal pairs everywhere (the statistics), none of them at the first positions of c2
        for tid in (0, 1, 2):
            for p in rng.integers(6000, 20000, 150):
                pair(f, tid, int(p), tid, int(p) + 300, False, True, True)
    # who emits the genome's last record in front of c2: the file with the later last record on c1
    pair(0, 0, 27000 if a_emits_last else 26000, 0, 27300 if a_emits_last else 26300, False, True, True)
    pair(1, 0, 26000 if a_emits_last else 27000, 0, 26300 if a_emits_last else 27300, False, True, True)
    for i in range(6):     # file a: an inter-chromosomal cluster c1:3000.. <-> c2:5000.. ; its first mate on c2 is c2's first record of file a
        pair(0, 0, 3000 + 7 * i, 1, 5000 + 7 * i, False, False, False)
    pair(1, 1, 5000, 1, 5300, False, True, True)   # file b: a proper read at c2:5000, forward -- ties with file a's first record there
    streams = []
    for f in (0, 1):
        r = sorted(rows[f], key=lambda x: (x[0], x[1], (x[5] >> 4) & 1))
        n = len(r)
        streams.append(dict(tid=np.array([x[0] for x in r], np.int32), pos=np.array([x[1] for x in r], np.int32), mtid=np.array([x[2] for x in r], np.int32),
                            mpos=np.array([x[3] for x in r], np.int32), isize=np.array([x[4] for x in r], np.int32), flag=np.array([x[5] for x in r], np.uint16),
                            qlen=np.full(n, 100, np.uint16), bdqual=np.full(n, 60, np.uint8), rg=np.array(["rg1" if f == 0 else "rg3"] * n),
                            name_id=np.array([x[6] for x in r], np.int64)))
        assert (streams[f]["tid"] == 1).any() and streams[f]["pos"][streams[f]["tid"] == 1][0] == 5000
    write_case(str(tmp_path), streams, targets, np.random.default_rng(5), index=True)
    (tmp_path / "cfg").write_text(cfg)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, min_read_pair=2, print_af=1))   # (-h: the allele frequency column is made of the copy numbers)
    want = filter_cmd_lines(run.text)
    ctx_rows = [l for l in want.splitlines() if l and not l.startswith("#") and "CTX" in l]
    assert ctx_rows, want
    texts = {}
    for label, env in (("two-ranks", dict(BDX_GPUS="0,0", BDX_TIMING="1")), ("three-ranks", dict(BDX_GPUS="0,0,0", BDX_TIMING="1")), ("one-gpu", dict(BDX_TIMING="1"))):
        p = subprocess.run([EXE, "-y", "-1", "-r", "2", "-h", "cfg"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, (label, p.stderr.decode())
        if label != "one-gpu":
            assert "(2 files)" in p.stderr.decode(), p.stderr.decode()
        texts[label] = filter_cmd_lines(p.stdout.decode())
        assert texts[label] == want, (label, a_emits_last, p.stderr.decode()[-600:])


def test_the_tie_of_the_previous_test_changes_the_table(tmp_path):
    """... and it does: the oracle's tables for the two ways differ in the allele frequency of the inter-chromosomal SV"""
    import importlib
    me = importlib.import_module("test_gpu_cli_fuzz")
    seen = []
    orig = me.oracle_case

    def spy(*a, **kw):
        r = orig(*a, **kw)
        seen.append(filter_cmd_lines(r.text))
        return r
    me.oracle_case = spy
    try:
        for way in (True, False):
            d = tmp_path / str(way)
            d.mkdir()
            me.test_cli_sharded_two_bams_tie_at_a_chromosomes_first_position(d, way)
    finally:
        me.oracle_case = orig
    strip = lambda t: "\n".join(l for l in t.splitlines() if not l.startswith("#"))
    assert len(seen) == 2 and strip(seen[0]) != strip(seen[1])
