"""GPU tests of the chromosome-sharded path (include/bdx.h bdx_dist_*, csrc/bdx_dist_impl.h) on one GPU: one context
per rank holding all of its chromosomes, the chromosomes dealt to 1, 2 or 3 ranks that run as threads of this process and go
through every exchange of the multi-GPU run (all-reduces, the CTX all-to-all, the name census, the taint exchange, the gathers
and the merge on rank 0); plus the RCCL backend itself with a communicator of one rank.  The sharded whole-genome run must equal ONE oracle run over all chromosomes -- including the
cross-chromosome effects (global window / lambda, prefix counters, the closing read of the next chromosome, CTX mates)."""
import numpy as np
import pytest

from fuzzgen import make_case
from helpers import load_chr21, make_opts
from runner import compare, expected_ctx_travel, oracle_case, product_options, sharded_from_oracle, split_by_tid

pytestmark = pytest.mark.gpu

WG_OPTS = [dict(), dict(transchr_rearrange=1), dict(cn_lib=1, print_af=1), dict(buffer_size=1), dict(buffer_size=3, min_read_pair=1),
           dict(min_map_qual=0, min_len=0), dict(transchr_rearrange=1, min_read_pair=1), dict(illumina_long_insert=1), dict(fisher=1),
           dict(seq_coverage_lim=3), dict(max_sd=900)]


dev_total = [0]


@pytest.mark.parametrize("seed", range(24))
def test_staged_whole_genome_equals_single_run(seed):
    cfg, streams, targets = make_case(100 + seed)
    for i, o in enumerate((WG_OPTS[seed % len(WG_OPTS)], WG_OPTS[(seed * 5 + 2) % len(WG_OPTS)])):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        # (rank 0 walks the components that span ranks: few gathered groups on its host, many on its device -- here the route is forced in turn)
        util = sharded_from_oracle(run, world=1 + (seed + i) % 3, result_debug={"gather_walk": 1 + (seed // 3 + i) % 2})
        compare(run, util, check_cls=False)
        n_dev, n_host, _ = util.walk_split()
        assert n_dev + n_host == run.n_svs
        dev_total[0] += n_dev
    if seed == 23:
        assert dev_total[0] > 100   # the components inside one rank go through the device walk (K6) where they live, not only rank 0's host walk


@pytest.mark.parametrize("seed", range(8))
def test_sharded_run_with_a_negative_min_len(seed):
    """-s -1: the genome's very first anomalous read registers a read-less region 0 (BreakDancer.cpp:244-264, the reference accepts any -s,
    common/Options.cpp:44-74) -- once, on whichever rank holds the first chromosome with anomalous reads; every real region id shifts by one
    and the flush cadence with it.  1-3 ranks against the oracle and against the single context"""
    cfg, streams, targets = make_case(1200 + seed)
    for o in (dict(min_len=-1), dict(min_len=-5, buffer_size=2), dict(min_len=-1, transchr_rearrange=1))[seed % 3:][:2]:
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        assert run.n_regions > 1
        for world in (1, 2, 3):
            util = sharded_from_oracle(run, world=world)
            compare(run, util, check_cls=False)


@pytest.mark.parametrize("seed", range(8))
def test_sharded_run_over_five_and_eight_ranks(seed):
    """the shape of the 8-GPU configurations on one GPU: 8 (and 5) ranks as threads -- more ranks than some cases have chromosomes (ranks
    without reads take part in every collective), LPT packing at 8 bins, most inter-chromosomal pairs crossing ranks, rank 0 handling the
    components that span ranks; -t, -a -h, small flush windows.  Against ONE oracle run"""
    from fuzzgen import GRAPH_OPTION_SETS, make_graph_case
    cfg, streams, targets = (make_case if seed % 2 == 0 else make_graph_case)(3100 + seed)
    osets = (WG_OPTS[seed % len(WG_OPTS)], dict(transchr_rearrange=1, min_read_pair=1), dict(cn_lib=1, print_af=1, buffer_size=2)) if seed % 2 == 0 else \
        (GRAPH_OPTION_SETS[seed % len(GRAPH_OPTION_SETS)], dict(transchr_rearrange=1, min_read_pair=1))
    for i, o in enumerate(osets):
        if o.get("min_len", 0) < 0:
            continue
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        for world in ((8, 5) if i == 0 else (8,)):
            keep = []
            util = sharded_from_oracle(run, world=world, keep=keep, result_debug={"gather_walk": (0, 1, 2)[(seed + i + world) % 3]})
            compare(run, util, check_cls=False)
            n_dev, n_host, _ = util.walk_split()
            assert n_dev + n_host == run.n_svs
            n_ctx, n_travel = expected_ctx_travel(run, world)
            ex = keep[0].exchange
            assert sum(e["ctx_records_sent"] for e in ex) == n_travel == sum(e["ctx_records_received"] for e in ex)


def test_staged_chr21_all_sequences():
    run = load_chr21(make_opts()).run()
    for world in (1, 2):
        util = sharded_from_oracle(run, world=world)
        s = compare(run, util, check_cls=False)
        assert s["n_svs_printed"] == 4


def test_only_inter_chromosomal_records_cross_ranks():
    """translocation-rich input over 3 ranks, with and without -t: an inter-chromosomal read travels exactly when its mate lies on a
    LATER chromosome that another rank holds (the pair is observed where its second mate is); nothing else crosses ranks on the
    data path, and the components that stay inside one rank are walked there on the device"""
    from test_gpu_configs import cfg_line, oracle_from_soa
    from breakdancer_amd.synth import make_genome
    d = make_genome([2_000_000, 1_500_000, 1_500_000, 1_000_000], coverage=15.0, seed=5, n_translocations=200)
    cfg = cfg_line("rg0", "wgs.bam", "lib0", 400.0, 30.0)
    for kw in (dict(transchr_rearrange=1), dict()):
        run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(**kw), ["c1", "c2", "c3", "c4"])
        for route in (1, 2):   # rank 0's walk of the gathered components: on its device, on its host
            util = sharded_from_oracle(run, world=3, result_debug={"gather_walk": route})
            compare(run, util, check_cls=False)
        keep = []
        util = sharded_from_oracle(run, world=3, keep=keep)
        compare(run, util, check_cls=False)
        n_dev, n_host, _ = util.walk_split()
        assert n_dev + n_host == run.n_svs
        ex = keep[0].exchange
        n_ctx, n_travel = expected_ctx_travel(run, 3)
        assert sum(e["ctx_records_sent"] for e in ex) == n_travel == sum(e["ctx_records_received"] for e in ex)
        assert n_ctx > 5000 and 0 < n_travel < n_ctx                      # at most one mate of a pair travels, none inside a rank
        assert n_ctx < 0.01 * run.n_merged or kw                          # ... a sliver of the reads
        if not kw:
            assert n_dev > n_host > 0   # most components lie inside one chromosome and are walked where they live, on the device; the
                                        # translocations between chromosomes of two ranks are rank 0's


def test_rccl_backend_with_a_communicator_of_one():
    """the RCCL code path itself (ncclCommInitRank, ncclAllReduce, ncclAllToAllv, grouped send / receive) on the one GPU at hand"""
    from breakdancer_amd import dist as D
    from breakdancer_amd.api import LibraryConfig
    cfg, streams, targets = make_case(131)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, transchr_rearrange=1, min_read_pair=1))
    libs = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                          bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
    uid = D.unique_id()
    assert len(uid) == 128
    d = D.DistRun.create(product_options(run.opts), libs, run.nbams, len(targets), run.w0, 0, 0, 1, uid)
    for tid, arrs in split_by_tid(run.merged_soa()).items():
        d.chromosome(tid).push_reads(arrs)
    d.run()
    compare(run, d.result(), check_cls=False)
    ex = d.exchange()
    assert ex["ctx_records_sent"] == ex["ctx_records_received"] == 0   # (one rank: every mate lives here, nothing travels; the census does)
    d.close()


@pytest.mark.parametrize("how", ["prepare", "run"])
def test_a_rank_whose_reads_are_not_in_reference_order_is_refused(how):
    """the chromosome table is found by binary searches in the reference-id column: a batch that holds ids out of order -- here two records of the
    first chromosome in the middle of the second's -- or an id beyond the header's sequences is an error of the caller's, said so before anything
    is searched (by bdx_dist_prepare, or by the first bdx_dist_run on reads that were not prepared), not a silently wrong table"""
    from breakdancer_amd import dist as D
    from breakdancer_amd.api import BdxError, LibraryConfig
    cfg, streams, targets = make_case(133)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1))
    libs = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                          bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
    parts = split_by_tid(run.merged_soa())
    tids = sorted(parts)
    assert len(tids) >= 2
    for mode in ("swapped", "beyond"):
        d = D.DistRun.create(product_options(run.opts), libs, run.nbams, len(targets), run.w0, 0, 0, 1, D.unique_id())
        for tid in tids:
            arrs = {k: np.array(v, copy=True) for k, v in parts[tid].items()}
            if tid == tids[1]:
                mid = len(arrs["tid"]) // 2
                arrs["tid"][mid:mid + 2] = tids[0] if mode == "swapped" else len(targets) + 3
            d.chromosome(tid).push_reads(arrs)
        with pytest.raises(BdxError) as e:
            d.prepare() if how == "prepare" else d.run()
        assert mode == "beyond" or "ascending order" in str(e.value), str(e.value)
        d.close()


@pytest.mark.parametrize("seed", range(6))
def test_per_chromosome_mode_equals_dash_o_runs(seed):
    """README:31 parallel mode: results of chromosome t == `breakdancer-max -o t`"""
    from breakdancer_amd.api import LibraryConfig
    from breakdancer_amd.shard import run_per_chromosome
    cfg, streams, targets = make_case(200 + seed)
    whole = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1))
    libs = [LibraryConfig(*[float(x) for x in whole.lib_f[i]], min_mapping_quality=int(whole.lib_i[i, 0]),
                          bam_file_index=int(whole.lib_i[i, 1])) for i in range(whole.nlibs)]
    chroms = split_by_tid(whole.merged_soa())
    opts = product_options(make_opts(score_threshold=-1, chr_tid=0))
    res = run_per_chromosome(opts, libs, whole.nbams, whole.w0, chroms)
    assert list(res) == sorted(chroms)
    for t, r in res.items():
        single = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, chr_tid=t))
        svs = r["svs"][0]
        assert len(svs) == single.n_svs
        if single.n_svs:
            np.testing.assert_array_equal(svs["pos"][:, 0], single.sv_i[:, 1])
            np.testing.assert_array_equal(svs["pos"][:, 1], single.sv_i[:, 5])
            np.testing.assert_array_equal(svs["flag"], single.sv_i[:, 8])
            np.testing.assert_array_equal(svs["score"], single.sv_i[:, 10])
            np.testing.assert_array_equal(svs["num_reads"], single.sv_i[:, 11])
        assert r["summary"]["window"] == single.W and r["summary"]["covered_ref_len"] == single.ref_len


@pytest.mark.parametrize("seed", range(8))
def test_sharded_run_with_read_names_seen_more_than_twice(seed):
    """clashing read names (triples, quadruples, across files, across chromosomes and therefore across ranks) in a sharded run:
    some rank's join notices a third sighting, all ranks agree on it with the next all-reduce, the compact records of every
    chromosome are gathered and rank 0 replays the run read by read -- same output as ONE oracle run, no error and no hang
    (ReadRegionData.cpp:108-113,152-175, SvBuilder.cpp:101-118)"""
    from fuzzgen import GRAPH_OPTION_SETS, OPTION_SETS, clash_names, make_graph_case
    if seed % 2 == 0:
        cfg, streams, targets = make_case(940 + seed)
        osets = OPTION_SETS
    else:
        cfg, streams, targets = make_graph_case(940 + seed)
        osets = GRAPH_OPTION_SETS
    streams = clash_names(streams, seed, frac=0.02 + 0.02 * (seed % 4))
    replayed = 0
    for i, o in enumerate((osets[seed % len(osets)], dict(transchr_rearrange=1, min_read_pair=1), dict(min_read_pair=1, buffer_size=1))):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        util = sharded_from_oracle(run, world=1 + (seed + i) % 3)
        compare(run, util, check_cls=False)
        replayed += util.was_replayed()
    assert replayed > 0


@pytest.mark.parametrize("seed", range(6))
def test_sharded_run_returns_the_supporting_reads(seed):
    """-g / -d in a sharded run: the supporting reads of every SV, as positions in the merged stream of the whole genome, in
    SvBuilder's observation order -- through the gather of the compact records and the read-level walk on rank 0
    (BreakDancer.cpp:514-534, SvBuilder.cpp:101-118)"""
    from fuzzgen import GRAPH_OPTION_SETS, OPTION_SETS, clash_names, make_graph_case
    from runner import compare_support
    cfg, streams, targets = (make_case if seed % 2 == 0 else make_graph_case)(1400 + seed)
    if seed >= 4:
        streams = clash_names(streams, seed, frac=0.03)
    osets = OPTION_SETS if seed % 2 == 0 else GRAPH_OPTION_SETS
    done = 0
    for i, o in enumerate((osets[seed % len(osets)], dict(transchr_rearrange=1, min_read_pair=1), dict(min_read_pair=1, buffer_size=2))):
        if o.get("min_len", 0) < 0:
            continue
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        util = sharded_from_oracle(run, world=1 + (seed + i) % 3, support=True, collide=3 if seed == 3 else 0)
        compare(run, util, check_cls=False)
        compare_support(run, util)
        done += 1
    assert done >= 2


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_run_with_supporting_reads_and_no_region(world):
    """-t on an input without inter-chromosomal pairs: no region, no SV -- and an EMPTY list of supporting reads, not a missing one
    (found by tools/extended_fuzz_sharded.py: the early exit for an empty table left the list unset and bdx_get_sv_support refused)"""
    from fuzzgen import make_graph_case
    from runner import compare_support
    cfg, streams, targets = make_graph_case(8011)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, transchr_rearrange=1, min_read_pair=1))
    assert len(run.sup_off) == 1 and len(run.sup_idx) == 0   # the oracle's table is empty
    util = sharded_from_oracle(run, world=world, support=True)
    compare(run, util, check_cls=False)
    compare_support(run, util)
