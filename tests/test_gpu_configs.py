"""GPU parity on scaled-down shapes of BASELINE.json configs[2]-[4]: multi-chromosome / 4 libraries (configs[2]),
planted translocations with -t (configs[3]), tumour/normal 2 BAMs with -a -h (configs[4]); single context and the
chromosome-sharded staged path, each against ONE oracle run."""
import numpy as np
import pytest

from helpers import OracleRun, make_opts
from runner import compare, compare_support, product_from_oracle, sharded_from_oracle

pytestmark = pytest.mark.gpu


def oracle_from_soa(d, cfg, bams, opts, targets):
    run = OracleRun(cfg, opts)
    run.set_targets(targets)
    assert run.bam_names == bams
    for b in range(len(bams)):
        m = d["bam"] == b
        st = {k: d[k][m] for k in ("tid", "pos", "mtid", "mpos", "isize", "flag")}
        st["qlen"] = d["qlen"][m].astype(np.int32)
        st["bdqual"] = d["mapq"][m]
        st["lib"] = d["lib"][m].astype(np.int32)
        st["name_id"] = d["name_key"][m]
        run.set_stream(b, st)
    return run.run()


def cfg_line(rg, bam, lib, mean, std):
    return "readgroup:%s\tplatform:illumina\tmap:%s\treadlen:100.00\tlib:%s\tlower:%.2f\tupper:%.2f\tmean:%.2f\tstd:%.2f\n" % (
        rg, bam, lib, mean - 3 * std, mean + 3 * std, mean, std)


LIBS4 = ((400.0, 30.0), (350.0, 40.0), (500.0, 50.0), (300.0, 25.0))


def test_config2_shape_four_libraries_three_chromosomes():
    from breakdancer_amd.synth import make_genome
    d = make_genome([4_000_000, 3_000_000, 2_000_000], coverage=20.0, seed=3, libs=LIBS4, lib_bam=(0, 0, 0, 0))
    cfg = "".join(cfg_line("rg%d" % i, "wgs.bam", "lib%d" % i, m, s) for i, (m, s) in enumerate(LIBS4))
    for kw in (dict(), dict(cn_lib=1, print_af=1)):
        run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(**kw), ["c1", "c2", "c3"])
        assert run.n_svs > 300
        compare(run, product_from_oracle(run))
        compare(run, sharded_from_oracle(run, world=3), check_cls=False)


def test_two_libraries_in_one_file_through_k1_s_several_libraries_tile_body():
    """every tile holds reads of both read groups of ONE file: K1's body for such tiles (library record per read) -- with one counter
    key for the tile (the ready-made records carry the read's own library), with a key per library (-a: two keys, the slots K2 would read
    say that the tile has several), with the -l remaps and with -t"""
    from breakdancer_amd.synth import make_genome
    libs = ((400.0, 30.0), (300.0, 25.0))
    d = make_genome([3_000_000, 2_000_000], coverage=20.0, seed=21, libs=libs, lib_bam=(0, 0), n_translocations=40)
    cfg = "".join(cfg_line("rg%d" % i, "wgs.bam", "lib%d" % i, m, s) for i, (m, s) in enumerate(libs))
    for kw in (dict(), dict(cn_lib=1, print_af=1), dict(illumina_long_insert=1), dict(transchr_rearrange=1, cn_lib=1), dict(min_map_qual=10)):
        run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(**kw), ["c1", "c2"])
        assert run.n_svs > 30
        compare(run, product_from_oracle(run))


def test_config3_shape_translocations_with_dash_t():
    from breakdancer_amd.synth import make_genome
    d = make_genome([3_000_000, 2_500_000, 2_000_000, 1_500_000], coverage=15.0, seed=5, n_translocations=300)
    cfg = cfg_line("rg0", "wgs.bam", "lib0", 400.0, 30.0)
    for kw in (dict(transchr_rearrange=1), dict()):
        run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(**kw), ["c1", "c2", "c3", "c4"])
        ctx_rows = int((run.sv_i[:, 8] == 8).sum())
        assert ctx_rows > 250, ctx_rows
        if kw:
            assert run.W == 50  # -t: no insert-size flags are counted, the window drops to 50 (BreakDancerMax.cpp:114)
        compare(run, product_from_oracle(run))
        compare(run, sharded_from_oracle(run, world=3), check_cls=False)


def test_config4_shape_tumour_normal_two_bams():
    from breakdancer_amd.synth import make_genome
    libs = ((400.0, 30.0), (420.0, 35.0), (380.0, 28.0))
    d = make_genome([5_000_000, 3_000_000], coverage=30.0, seed=9, libs=libs, lib_bam=(0, 0, 1))
    cfg = cfg_line("rgT1", "tumour.bam", "libT1", *libs[2]) + cfg_line("rgN1", "normal.bam", "libN1", *libs[0]) + \
        cfg_line("rgN2", "normal.bam", "libN2", *libs[1])
    # sorted names: libraries libN1, libN2, libT1 ; files normal.bam (0), tumour.bam (1)
    for kw in (dict(cn_lib=1, print_af=1), dict(print_af=1)):
        run = oracle_from_soa(d, cfg, ["normal.bam", "tumour.bam"], make_opts(**kw), ["c1", "c2"])
        assert run.lib_names == ["libN1", "libN2", "libT1"] and run.n_svs > 300
        compare(run, product_from_oracle(run))
        compare(run, sharded_from_oracle(run, world=3), check_cls=False)


def test_many_files_and_libraries_take_the_general_paths():
    """70 source files / 70 libraries: beyond the 64-lane monoid fast path and the small-count ballot paths of K1,
    and 70 counter keys without -a; interleaved at random so that every wave mixes files and libraries"""
    rng = np.random.default_rng(17)
    nb = 60
    from breakdancer_amd.synth import make_chromosome, concat
    parts, cfg = [], ""
    for i in range(nb):
        mean, std = 300.0 + 5 * i, 25.0
        parts.append(make_chromosome(length=600_000, coverage=1.2, seed=100 + i, lib=i, bam=i, name_base=i << 36, mean=mean, std=std,
                                     discordant=0.05))
        cfg += cfg_line("rg%02d" % i, "f%02d.bam" % i, "lib%02d" % i, mean, std)
    d = concat(parts)
    order = np.lexsort(((d["flag"] >> 4) & 1, d["pos"], d["tid"]))
    d = {k: v[order] for k, v in d.items()}
    bams = ["f%02d.bam" % i for i in range(nb)]
    for kw in (dict(min_read_pair=1), dict(cn_lib=1, print_af=1)):
        run = oracle_from_soa(d, cfg, bams, make_opts(**kw), ["c1"])
        assert run.nlibs == nb and run.nbams == nb and run.n_svs > 50
        compare(run, product_from_oracle(run))


@pytest.mark.parametrize("cluster,two_files,nostash", [(15, False, False), (17, False, False), (16, True, False), (33, True, False), (16, True, True)])
def test_ready_made_records_of_k1_and_their_fallbacks(cluster, two_files, nostash, monkeypatch):
    """K1 leaves the anomalous reads of a tile ready-made for K2 when there are at most 16 of them and the tile's reads share one
    library and file; tiles with more, mixed tiles (here: two files alternating in 40 kbp blocks, so that tiles inside a block
    are uniform and the ones at a block edge are not) and runs with BDX_NO_STASH take the column gather.  All must equal the
    oracle, with one counter key and with two."""
    from breakdancer_amd.synth import make_chromosome
    if nostash:
        monkeypatch.setenv("BDX_NO_STASH", "1")
    d = make_chromosome(length=2_000_000, coverage=30.0, seed=40 + cluster, cluster=cluster, discordant=0.02)
    cfg, bams = cfg_line("rg0", "a.bam", "lib0", 400.0, 30.0), ["a.bam"]
    if two_files:
        # both mates of a pair go to the file of the pair's left end (the classifier reads lib / file per read, the oracle per stream)
        left = np.minimum(d["pos"], d["mpos"])
        blk = ((left // 40_000) & 1).astype(np.uint8)
        d = dict(d)
        d["bam"] = blk
        d["lib"] = blk
        cfg += cfg_line("rg1", "b.bam", "lib1", 400.0, 30.0)
        bams = ["a.bam", "b.bam"]
    for kw in (dict(), dict(cn_lib=1, print_af=1, min_read_pair=3)):
        run = oracle_from_soa(d, cfg, bams, make_opts(**kw), ["c1"])
        assert run.n_svs > 100
        compare(run, product_from_oracle(run))


SPLIT_SETS = [dict(), dict(buffer_size=1), dict(buffer_size=2), dict(buffer_size=7, min_read_pair=1), dict(min_read_pair=3),
              dict(min_read_pair=4, cn_lib=1, print_af=1), dict(fisher=1), dict(chr_tid=1), dict(transchr_rearrange=1),
              dict(min_len=30, seq_coverage_lim=3), dict(max_sd=900), dict(illumina_long_insert=1)]


@pytest.mark.parametrize("kw", SPLIT_SETS)
def test_device_assembly_and_host_walk_agree_with_the_oracle(kw):
    """The SV candidates of simple components are assembled on the device (K6), the rest by the host walk; forcing
    everything through the host walk must give the same rows, and both must equal the oracle's."""
    from breakdancer_amd.synth import make_genome
    libs = ((400.0, 30.0), (330.0, 25.0), (480.0, 45.0))
    d = make_genome([2_500_000, 2_000_000, 1_500_000], coverage=24.0, seed=21, libs=libs, lib_bam=(0, 1, 1), n_translocations=120)
    cfg = cfg_line("rgA", "a.bam", "libA", *libs[0]) + cfg_line("rgB", "b.bam", "libB", *libs[1]) + cfg_line("rgC", "b.bam", "libC", *libs[2])
    run = oracle_from_soa(d, cfg, ["a.bam", "b.bam"], make_opts(score_threshold=-1, **kw), ["c1", "c2", "c3"])
    assert run.n_svs > 10
    bd = product_from_oracle(run)
    compare(run, bd)
    n_dev, n_host, _ = bd.walk_split()
    assert n_dev > 0 and n_dev + n_host == run.n_svs, (n_dev, n_host, run.n_svs)
    bd.close()
    bh = product_from_oracle(run, host_walk=True)
    compare(run, bh)
    assert bh.walk_split()[0] == 0
    bh.close()


def test_regions_larger_than_a_wave_are_merged_in_two_levels():
    """clusters of 90 pairs: a region with more than 64 reads is sorted and merged 64 reads at a time, and the chunks'
    parts once more -- still on the device"""
    from breakdancer_amd.synth import make_chromosome
    d = make_chromosome(length=3_000_000, coverage=30.0, seed=5, cluster=90, discordant=0.03)
    cfg = cfg_line("rg0", "wgs.bam", "lib0", 400.0, 30.0)
    run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(score_threshold=-1), ["c1"])
    assert run.n_svs > 50 and int(run.regions[:, 7].max()) > 64
    bd = product_from_oracle(run, support=True)
    compare(run, bd)
    compare_support(run, bd)
    n_dev, n_host, _ = bd.walk_split()
    assert n_dev > 10 * max(1, n_host), (n_dev, n_host)
    bd.close()


def test_regions_with_more_parts_than_a_wave_go_through_the_host_walk():
    """clusters of 250 pairs drawn from 120 libraries: more than 64 distinct (region, flag, library) parts in one region
    cannot be merged in registers; the chunks' parts reach the host walk in pieces and are merged there"""
    from breakdancer_amd.synth import make_chromosome
    nl = 120
    d = make_chromosome(length=3_000_000, coverage=30.0, seed=9, cluster=250, discordant=0.04)
    d = dict(d)
    d["lib"] = ((d["name_key"] * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)).astype(np.int64) % nl  # one library per pair
    d["lib"] = d["lib"].astype(np.uint8)
    cfg = "".join(cfg_line("rg%03d" % i, "wgs.bam", "lib%03d" % i, 400.0, 30.0) for i in range(nl))
    run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(score_threshold=-1), ["c1"])
    assert run.n_svs > 20 and int(run.regions[:, 7].max()) > 200
    bd = product_from_oracle(run)
    compare(run, bd)
    assert bd.walk_split()[1] > 0
    bd.close()
