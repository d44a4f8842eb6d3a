"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on identical inputs."""
import json
import math
import os

import numpy as np
import pytest

from fuzzgen import GRAPH_OPTION_SETS, OPTION_SETS, make_case, make_graph_case
from helpers import GOLDEN, OracleRun, load_chr21, make_opts
from runner import compare, compare_support, oracle_case, product_from_oracle

pytestmark = pytest.mark.gpu
TID21 = 22


def test_native_library_is_the_one_running():
    import breakdancer_amd._lib as L
    lib = L.load()
    assert os.path.samefile(lib._name, os.path.join(os.path.dirname(L.__file__), "libbdx.so"))


CHR21_SETS = [dict(chr_tid=TID21), dict(), dict(chr_tid=TID21, print_af=1), dict(chr_tid=TID21, cn_lib=1),
              dict(chr_tid=TID21, cn_lib=1, print_af=1), dict(transchr_rearrange=1), dict(illumina_long_insert=1),
              dict(buffer_size=1), dict(buffer_size=2), dict(min_read_pair=1, score_threshold=-1), dict(min_map_qual=0),
              dict(min_len=0, score_threshold=-1), dict(fisher=1), dict(max_sd=600), dict(seq_coverage_lim=1)]


@pytest.mark.parametrize("kw", CHR21_SETS)
def test_chr21_fixtures_match_oracle(kw):
    run = load_chr21(make_opts(**kw)).run()
    bd = product_from_oracle(run)
    s = compare(run, bd)
    if not kw or kw == dict(chr_tid=TID21):
        assert s["n_svs_printed"] == 4 and s["window"] == 287 and s["covered_ref_len"] == 5626088


@pytest.mark.parametrize("seed", range(40))
def test_differential_fuzz(seed):
    cfg, streams, targets = make_case(seed)
    for o in (OPTION_SETS[seed % len(OPTION_SETS)], OPTION_SETS[(seed * 7 + 3) % len(OPTION_SETS)]):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        bd = product_from_oracle(run)
        compare(run, bd)
        bd.close()


@pytest.mark.parametrize("seed", range(12))
def test_supporting_reads_match_svbuilder_order(seed):
    """the reads behind every SV (input of the -g BED / -d FASTQ dumps), in SvBuilder's observation order"""
    cfg, streams, targets = make_case(400 + seed)
    for o in (OPTION_SETS[seed % len(OPTION_SETS)], dict(min_read_pair=1, buffer_size=2), dict(min_len=-1, min_read_pair=1)):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        bd = product_from_oracle(run, support=True)
        compare(run, bd)
        compare_support(run, bd)
        bd.close()
    run = load_chr21(make_opts(chr_tid=TID21)).run()
    bd = product_from_oracle(run, support=True)
    compare_support(run, bd)
    assert len(bd.sv_support()[1]) == 60  # expected.bed shows 56 of them (4 have another flag than their SV)


def test_empty_and_tiny_inputs():
    cfg = "readgroup:rg1\tmap:a.bam\treadlen:100\tlib:l1\tmean:400\tstd:30\tlower:310\tupper:490\n"
    empty = dict(tid=np.zeros(0, np.int32), pos=np.zeros(0, np.int32), mtid=np.zeros(0, np.int32), mpos=np.zeros(0, np.int32),
                 isize=np.zeros(0, np.int32), flag=np.zeros(0, np.uint16), qlen=np.zeros(0, np.int32),
                 bdqual=np.zeros(0, np.uint8), rg=[], name_id=np.zeros(0, np.uint64))
    run = oracle_case(cfg, [empty], ["c1"], make_opts())
    bd = product_from_oracle(run)
    compare(run, bd)
    assert bd.summary()["window"] == 50
    # one pair, one anomalous read each: a single candidate region closed by the end of the stream
    one = dict(tid=np.array([0, 0], np.int32), pos=np.array([100, 5000], np.int32), mtid=np.array([0, 0], np.int32),
               mpos=np.array([5000, 100], np.int32), isize=np.array([5000, -5000], np.int32),
               flag=np.array([0x1 | 0x20 | 0x40, 0x1 | 0x10 | 0x80], np.uint16), qlen=np.array([100, 100], np.int32),
               bdqual=np.array([60, 60], np.uint8), rg=["rg1", "rg1"], name_id=np.array([7, 7], np.uint64))
    run = oracle_case(cfg, [one], ["c1"], make_opts(min_read_pair=1, min_len=-1, score_threshold=-1))
    compare(run, product_from_oracle(run))


def test_ragged_batch_is_rejected():
    import breakdancer_amd as bda
    from breakdancer_amd.api import LibraryConfig, Options
    bd = bda.BreakDancer(Options(), [LibraryConfig(400, 30, 490, 310, 100)], 1)
    from breakdancer_amd.synth import make_chromosome
    d = make_chromosome(length=200000, seed=3)
    d["pos"] = d["pos"][:-1]
    with pytest.raises(ValueError):
        bd.push_reads(d)


def test_poisson_kernel_against_mpmath_vectors():
    """north_star: Poisson scores within 1e-6; the kernel holds 1e-10 relative on log p"""
    from breakdancer_amd.api import poisson_log_upper_tail
    v = json.load(open(os.path.join(GOLDEN, "poisson_vectors.json")))["poisson"]
    lam = np.array([float(r["lambda"]) for r in v])
    k = np.array([r["k"] for r in v], np.int32)
    got = poisson_log_upper_tail(lam, k)
    for r, g in zip(v, got):
        if r["p_double_positive"]:
            want = float(r["logp"])
            assert abs(g - want) <= 1e-10 * max(1.0, abs(want)), (r, g)
        else:
            assert g == -math.inf or g < -700, (r, g)


def _synth_case(length, seed, **kw):
    from breakdancer_amd.synth import LIB_C2, make_chromosome
    d = make_chromosome(length=length, seed=seed, **kw)
    cfg = "readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n"
    st = dict(tid=d["tid"], pos=d["pos"], mtid=d["mtid"], mpos=d["mpos"], isize=d["isize"], flag=d["flag"],
              qlen=d["qlen"].astype(np.int32), bdqual=d["mapq"], rg=["rg1"] * len(d["tid"]), name_id=d["name_key"])
    return cfg, st


@pytest.mark.parametrize("opts", [dict(), dict(buffer_size=10), dict(cn_lib=1, print_af=1), dict(min_read_pair=5)])
def test_synthetic_10mbp_matches_oracle(opts):
    """scaled-down configs[1] (10 Mbp, 3 M reads): every intermediate and every SV row bit-exact vs the oracle"""
    cfg, st = _synth_case(10_000_000, seed=11)
    run = OracleRun(cfg, make_opts(**opts))
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    assert run.n_svs > 100
    compare(run, product_from_oracle(run))


def test_index_columns_of_a_single_library_and_file_are_not_read():
    """one library, one BAM: every read belongs to them whatever its lib / bam bytes say (an index out of range counts as 0), so
    the kernels do not read those columns and bdx_push does not copy them -- garbage in them must not show"""
    from runner import product_options
    import breakdancer_amd as bda
    from breakdancer_amd.api import LibraryConfig
    cfg, st = _synth_case(3_000_000, seed=5)
    run = OracleRun(cfg, make_opts())
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    assert run.nlibs == 1 and run.nbams == 1 and run.n_svs > 20
    libs = [LibraryConfig(*[float(x) for x in run.lib_f[0]], min_mapping_quality=int(run.lib_i[0, 0]), bam_file_index=0, name=run.lib_names[0])]
    bd = bda.BreakDancer(product_options(run.opts), libs, 1, ntids=0, max_read_window_size=run.w0, device=0)
    soa = run.merged_soa()
    rng = np.random.default_rng(7)
    soa["lib"] = rng.integers(0, 256, len(soa["tid"])).astype(soa["lib"].dtype)
    soa["bam"] = rng.integers(0, 256, len(soa["tid"])).astype(np.uint8)
    bd.push_reads(soa)
    bd.run()
    compare(run, bd)


@pytest.mark.parametrize("max_chunks", ["1", "2"])
def test_tile_total_columns_scanned_in_chunks_of_several_rounds(max_chunks, monkeypatch):
    """finalize_kernel scans a tile-total column in chunks of whole rounds (4096 boundaries each); with at most 64 chunks a chunk
    takes more than one round only beyond 268 M reads -- BDX_MAX_CHUNKS brings that down to test size (20 Mbp: 5.9 k boundaries
    = two rounds: one chunk of two rounds / two chunks with BDX_MAX_CHUNKS=1 / 2)"""
    monkeypatch.setenv("BDX_MAX_CHUNKS", max_chunks)
    cfg, st = _synth_case(20_000_000, seed=12)
    run = OracleRun(cfg, make_opts())
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    assert run.n_svs > 1000
    compare(run, product_from_oracle(run))


def test_full_size_properties():
    """configs[1] at full size (50 Mbp, 15 M reads): size-independent properties of the path."""
    import breakdancer_amd as bda
    from breakdancer_amd.api import LibraryConfig, Options
    from breakdancer_amd.synth import LIB_C2, make_chromosome
    d = make_chromosome(length=50_000_000, seed=1)
    n = len(d["tid"])

    def run(arrs, **ow):
        bd = bda.BreakDancer(Options(**ow), [LibraryConfig(**LIB_C2)], 1, max_read_window_size=200)
        bd.push_reads(arrs)
        return bd.run()

    a = run(d)
    sa = a.summary()
    # (1) classifier census against a numpy restatement of the predicates (linearity of the counters)
    sam = d["flag"].astype(np.int64)
    ai = np.abs(d["isize"])
    fr = ((sam & 0x10) != 0) != ((sam & 0x20) != 0)
    left = d["pos"] < d["mpos"]
    rf = fr & (left == ((sam & 0x10) != 0))
    large = fr & ~rf & (ai > 490)
    small = fr & ~rf & ~large & (ai < 310)
    anom_all = (~fr) | rf | large | small
    mq_ok = d["mapq"] > 35
    assert sa["n_anomalous"] == int((anom_all & mq_ok).sum())
    c = a.counters()
    assert c["flag_hist"][0, 2] == int((large & mq_ok).sum()) and c["flag_hist"][0, 3] == int((small & mq_ok).sum())
    assert c["lib_read_count"][0] == int(((sam & 0x2) != 0)[mq_ok].sum())
    assert sa["covered_ref_len"] == int(d["pos"].max() - d["pos"].min())
    # (2) idempotence / determinism: a second context over the same input gives identical results
    b = run(d)
    assert b.summary() == sa
    np.testing.assert_array_equal(a.regions(), b.regions())
    sv_a, la, ca = a.svs()
    sv_b, lb, cb = b.svs()
    assert sv_a.tobytes() == sv_b.tobytes()
    # (3) regions are sorted, disjoint and respect the window: consecutive regions are separated by > W or rejected reads
    r = a.regions()
    assert (r["start"][1:] > r["end"][:-1]).all() and (r["end"] >= r["start"]).all()
    assert int(r["n_reads"].sum()) <= sa["n_anomalous"]
    # (4) pairs: every pair has both mates in accepted regions, so 2*pairs <= reads in accepted regions
    assert 2 * sa["n_pairs"] <= int(r["n_reads"].sum())
    # (5) batching invariance: pushing the same stream in 7 ragged batches changes nothing
    bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, max_read_window_size=200)
    cuts = [0, 1, 1025, 300000, 4000001, 9999999, 12345678, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        bd.push_reads({k: v[lo:hi] for k, v in d.items()})
    bd.run()
    assert bd.summary() == sa
    assert bd.svs()[0].tobytes() == sv_a.tobytes()
    # (6) planted clusters are recovered: most DEL calls have ~12 supporting pairs and size ~ 1550-400
    dels = sv_a[(sv_a["flag"] == 2) & (sv_a["printed"] == 1)]
    assert len(dels) > 1000
    assert abs(np.median(dels["size"]) - 1150) < 40
    # (7) and the whole result bit for bit against ONE oracle run over the same 15 M records (the oracle takes under a second):
    # class bytes, counters, window, region table, every SV field
    from test_gpu_configs import cfg_line, oracle_from_soa
    ref = oracle_from_soa(d, cfg_line("rg1", "syn.bam", "lib1", 400.0, 30.0), ["syn.bam"], make_opts(), ["chrS"])
    assert ref.n_merged == n
    s = compare(ref, product_from_oracle(ref))
    assert s["n_svs"] == sa["n_svs"] and s["n_regions"] == sa["n_regions"]


def _raw_case(n, keys, seed=3):
    """n anomalous reads (FF, insert 5100) on one chromosome with the given name ids, as one BAM stream + config"""
    rng = np.random.default_rng(seed)
    pos = np.sort(rng.integers(1000, 200000, n)).astype(np.int32)
    st = dict(tid=np.zeros(n, np.int32), pos=pos, mtid=np.zeros(n, np.int32), mpos=pos + 5000, isize=np.full(n, 5100, np.int32),
              flag=np.full(n, 0x1 | 0x20 | 0x40, np.uint16), qlen=np.full(n, 100, np.int32), bdqual=np.full(n, 60, np.uint8),
              rg=["rg1"] * n, name_id=np.asarray(keys, np.uint64))
    cfg = "readgroup:rg1\tplatform:illumina\tmap:x.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n"
    return cfg, [st], ["c1"]


def test_degenerate_inputs_do_not_hang():
    """all reads anomalous (every tile overflows the lane count); one name shared by thousands of reads; names shared by
    three reads.  The reference runs on all of them (ReadRegionData.cpp:108-113 just keeps appending; only the list
    reaching two adds an edge), so the product must produce the oracle's output, not an error."""
    n = 20000
    pairs = np.arange(n, dtype=np.uint64) // 2 + 1
    triples = pairs.copy()
    triples[2::50] = triples[0::50]
    for keys, expect_replay in ((pairs, False), (np.full(n, 42, np.uint64), True), (triples, True)):
        cfg, streams, targets = _raw_case(n, keys)
        for o in (dict(), dict(min_read_pair=1, buffer_size=3)):
            run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
            bd = product_from_oracle(run, support=True)
            s = compare(run, bd)
            compare_support(run, bd)
            assert s["n_anomalous"] == n
            assert bd.was_replayed() == expect_replay
            bd.close()


@pytest.mark.parametrize("seed", range(24))
def test_read_names_seen_more_than_twice(seed):
    """merged BAMs with clashing read names: triples, quadruples, duplicates across files and across accepted / rejected
    candidate regions.  The product notices the third sighting in the join and replays the region graph read by read with
    the reference's semantics (bdx_walk_reads.cpp); output and supporting reads must equal the oracle's."""
    from fuzzgen import clash_names
    if seed % 2 == 0:
        cfg, streams, targets = make_case(900 + seed)
        osets = OPTION_SETS
    else:
        cfg, streams, targets = make_graph_case(900 + seed)
        osets = GRAPH_OPTION_SETS
    streams = clash_names(streams, seed, frac=0.02 + 0.02 * (seed % 4))
    replayed = 0
    for o in (osets[seed % len(osets)], osets[(seed * 5 + 2) % len(osets)], dict(min_read_pair=1, buffer_size=1)):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        bd = product_from_oracle(run, support=True)
        compare(run, bd)
        compare_support(run, bd)
        replayed += bd.was_replayed()
        bd.close()
    assert replayed > 0


@pytest.mark.parametrize("seed", [3, 17, 29])
def test_bucketed_join_path(seed, monkeypatch):
    """inputs above 4 M join entries are partitioned into LDS-sized buckets; force that path on small inputs"""
    monkeypatch.setenv("BDX_BUCKETED_JOIN", "1")
    cfg, streams, targets = make_case(seed)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1))
    bd = product_from_oracle(run, support=True)
    compare(run, bd)
    compare_support(run, bd)
    bd.close()
    cfg, st = _synth_case(4_000_000, seed=5)
    run = OracleRun(cfg, make_opts())
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    compare(run, product_from_oracle(run))


@pytest.mark.parametrize("mode", ["ahead", "retry", "off"])
def test_repeated_runs_on_one_context(mode, monkeypatch):
    """a context that runs again on an input of the same size enqueues the later stages before the pass-1 record is back,
    sized from the previous run ("ahead"); BDX_SPEC_TEST=1 makes that guess too small, so the stages are neutralised on
    the device and run again ("retry"); every run must give the oracle's table"""
    if mode == "retry":
        monkeypatch.setenv("BDX_SPEC_TEST", "1")
    if mode == "off":
        monkeypatch.setenv("BDX_NO_SPECULATE", "1")
    cfg, st = _synth_case(6_000_000, seed=23)
    run = OracleRun(cfg, make_opts())
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    bd = product_from_oracle(run)
    compare(run, bd)
    for _ in range(3):
        bd.run()
        compare(run, bd)
    # every run sized like a first run (the prior on the read count: 1.8 M reads here, so it applies), and none ahead at all
    for m in (1, 0, 2):
        bd.set_enqueue_ahead(m)
        for _ in range(2):
            bd.run()
            compare(run, bd)
    bd.close()


@pytest.mark.parametrize("seed", range(28))
def test_small_component_graphs(seed):
    """hundreds of separate 1-5 region components with random connections around the weight gate, self groups, mixed
    flags and libraries, links across contigs: the device walk (components of <= 4 regions) and the host walk (the
    rest) against the oracle, and the device walk against the host walk"""
    from fuzzgen import GRAPH_OPTION_SETS, make_graph_case
    cfg, streams, targets = make_graph_case(seed)
    o = GRAPH_OPTION_SETS[seed % len(GRAPH_OPTION_SETS)]
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
    few = bool(o.get("transchr_rearrange"))  # -t keeps the inter-chromosomal pairs only
    assert run.n_svs > (0 if few else 15), run.n_svs
    bd = product_from_oracle(run, support=True)
    compare(run, bd)
    compare_support(run, bd)
    n_dev, n_host, _ = bd.walk_split()
    assert (few or n_dev > 5) and n_dev + n_host == run.n_svs, (n_dev, n_host)
    bd.close()
    bh = product_from_oracle(run, host_walk=True)
    compare(run, bh)
    bh.close()


def test_long_insertion_list_from_both_walks(monkeypatch):
    """dense chains of clusters with one region per flush window: thousands of order-key candidates from the device AND
    tens of thousands from the host walk -- the insertion list's merge beyond its LDS tiles on both sides"""
    monkeypatch.setenv("BDX_BIG_WALK", "0")
    cfg, st = _synth_case(12_000_000, seed=37, discordant=0.08, cluster=3)
    run = OracleRun(cfg, make_opts(buffer_size=0, min_read_pair=2))
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    bd = product_from_oracle(run)
    compare(run, bd)
    n_dev, n_host, _ = bd.walk_split()
    assert bd.cross_window_svs() > 2048 and n_host > 2048, (bd.cross_window_svs(), n_dev, n_host)
    bd.close()
    monkeypatch.setenv("BDX_BIG_WALK", "1")  # the same with the general device walk taking most of the host's share
    bb = product_from_oracle(run)
    compare(run, bb)
    assert bb.walk_split()[1] < n_host
    bb.close()


@pytest.mark.parametrize("seed", range(16))
def test_medium_component_graphs(seed, monkeypatch):
    """components of 1-66 regions with the general device walk (member lists instead of a pair table, up to 64
    regions; BDX_BIG_WALK=1 turns it on whatever the input size) against the oracle, support lists included"""
    from fuzzgen import GRAPH_OPTION_SETS, make_graph_case
    monkeypatch.setenv("BDX_BIG_WALK", "1")
    cfg, streams, targets = make_graph_case(100 + seed, n_slots=400, sizes=(1, 3, 5, 6, 8, 10, 14, 20, 30, 45, 66))
    o = GRAPH_OPTION_SETS[seed % len(GRAPH_OPTION_SETS)]
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
    bd = product_from_oracle(run, support=True)
    compare(run, bd)
    compare_support(run, bd)
    n_dev, n_host, _ = bd.walk_split()
    assert n_dev + n_host == run.n_svs, (n_dev, n_host)
    bd.close()
    monkeypatch.setenv("BDX_BIG_WALK", "0")
    bs = product_from_oracle(run)
    compare(run, bs)
    assert bs.walk_split()[0] <= n_dev  # without the general walk those components go to the host
    bs.close()


def test_sv_table_larger_than_one_pass_of_the_score_kernel():
    """several hundred thousand SV candidates (tiny clusters of two pairs): the kernels that are launched with a capped
    grid have to stride over the whole table (the score kernel once wrote only its first 131072 records)"""
    cfg, st = _synth_case(32_000_000, seed=31, discordant=0.25, cluster=2)
    run = OracleRun(cfg, make_opts())
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    assert run.n_svs > 150_000, run.n_svs  # well beyond 131072
    bd = product_from_oracle(run)
    compare(run, bd)
    bd.close()


@pytest.mark.parametrize("buffer_size,min_cross", [(1, 2048), (3, 256), (0, 2048)])
def test_components_spanning_flush_windows_are_walked_on_the_device(buffer_size, min_cross):
    """small flush windows (-b): most two-region components have their regions in different windows, so their SVs come
    from traversals started at an earlier window's region and are placed by order key -- enough of them to take the
    insertion list through its sort beyond one LDS tile"""
    cfg, st = _synth_case(12_000_000, seed=37, discordant=0.02, cluster=6)
    run = OracleRun(cfg, make_opts(buffer_size=buffer_size, min_read_pair=2))
    run.set_targets(["chrS"])
    st = dict(st)
    st["lib"] = np.zeros(len(st["tid"]), np.int32)
    run.set_stream(0, st)
    run.run()
    bd = product_from_oracle(run)
    compare(run, bd)
    n_dev, n_host, _ = bd.walk_split()
    assert bd.cross_window_svs() > min_cross, (bd.cross_window_svs(), n_dev, n_host)
    assert n_dev > 10 * max(n_host, 1), (n_dev, n_host)
    bd.close()


def test_contexts_in_flight_under_the_native_driver():
    """bdx_run_many: a list of contexts (one per chromosome of a caller that holds them all), three in flight at a time on threads of
    the library -- every context's table equals its own oracle run; a list that names a context twice is refused"""
    import breakdancer_amd as bda
    from breakdancer_amd.api import run_many
    runs, ctxs = [], []
    for seed in range(9):
        cfg, streams, targets = make_case(1500 + seed)
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **OPTION_SETS[seed % len(OPTION_SETS)]))
        libs = [bda.LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                                  bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
        from runner import product_options
        bd = bda.BreakDancer(product_options(run.opts), libs, run.nbams, ntids=0, max_read_window_size=run.w0)
        if run.n_merged:
            bd.push_reads(run.merged_soa())
        runs.append(run)
        ctxs.append(bd)
    for in_flight in (3, 1, 16):
        run_many(ctxs, in_flight)
        for run, bd in zip(runs, ctxs):
            compare(run, bd)
    with pytest.raises(bda.BdxError):
        run_many([ctxs[0], ctxs[1], ctxs[0]], 2)
    for bd in ctxs:
        bd.close()
