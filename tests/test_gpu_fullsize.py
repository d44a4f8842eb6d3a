"""GPU parity at the sizes BASELINE.json states for configs[2]-[4], one GPU's share each (the path shards by chromosome:
8 GPUs -> 1/8 of the reads per GPU).

  configs[2]  hg38-shaped, 24 chromosomes, 4 libraries, 30x: 3.09 Gbp / 8 -> >= 116 M reads on this GPU
  configs[3]  the same share + 5,000 planted translocations, run with -t
  configs[4]  tumour 60x + normal 30x, 6 read groups -> 3 libraries over 2 files, -a -h: a 1/24 share (46 M reads) under both
              option sets, and the full per-GPU share of 1/8 (348 M reads, 12 GB of records in HBM) under -a -h

Every case is checked bit-exactly against ONE oracle run over the same records, plus size-independent properties that do
not need the oracle: class bytes against a vectorised numpy restatement of the classifier, device walk == host walk,
region-table invariants, and recovery of the planted variants."""
import numpy as np
import pytest

from helpers import make_opts
from runner import compare, product_from_oracle
from test_gpu_configs import LIBS4, cfg_line, oracle_from_soa

pytestmark = pytest.mark.gpu

# hg38 primary assembly, chr1-22, X, Y (Mbp)
HG38_MBP = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28, 114.36, 107.04,
            101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82, 156.04, 57.23]
TARGETS = ["chr%s" % c for c in list(range(1, 23)) + ["X", "Y"]]


def share_lengths(fraction):
    return [int(m * 1e6 * fraction) for m in HG38_MBP]


def numpy_class_bytes(d, upper, lower, min_mapq, opt_t=False, max_sd=1000000000):
    """io/IlluminaPEReadClassifier.cpp:13-101 + the filter chain of BreakDancer.cpp:159-206 as array arithmetic:
    (flag | pass << 4 | proper << 5) per record, the oracle's class byte"""
    sam = d["flag"].astype(np.int32)
    tid, mtid, pos, mpos = d["tid"], d["mtid"], d["pos"], d["mpos"]
    ai = np.abs(d["isize"].astype(np.int64))
    lib = d["lib"].astype(np.int64)
    up = np.asarray(upper, np.float32)[lib]
    lo = np.asarray(lower, np.float32)[lib]
    rev, mrev = (sam & 0x10) != 0, (sam & 0x20) != 0
    f = np.full(len(sam), 6, np.uint8)                                    # NORMAL_FR
    f[ai.astype(np.float32) < lo] = 3                                       # ARP_SMALL_INSERT
    f[ai.astype(np.float32) > up] = 2                                       # ARP_LARGE_INSERT
    f[(pos < mpos) == rev] = 4                                              # ARP_RF (leftmost read on the reverse strand)
    same = rev == mrev
    f[same & rev] = 5                                                       # ARP_RR
    f[same & ~rev] = 1                                                      # ARP_FF
    f[tid != mtid] = 8                                                      # ARP_CTX
    f[(sam & 0x8) != 0] = 9                                                 # MATE_UNMAPPED
    f[(sam & 0x4) != 0] = 10                                                # UNMAPPED
    f[((sam & 0x400) != 0) | ((sam & 0x1) == 0)] = 0                        # NA
    unm = (sam & 0xC) != 0
    ok = (f != 0) & ~unm & (d["mapq"].astype(np.int32) > np.asarray(min_mapq)[lib]) & ((f == 8) | (ai <= max_sd))
    if opt_t:
        ok &= tid != mtid
    proper = ok & ((sam & (0x2 | 0x4 | 0x8 | 0x1 | 0x400)) == 0x3)
    g = f.copy()
    g[ok & (f == 5)] = 1                                                    # RR clusters with FF (BreakDancer.cpp:196-197)
    return g | (ok.astype(np.uint8) << 4) | (proper.astype(np.uint8) << 5)


def tables_equal(a, b):
    sa, (la, pa), (ka, va) = a.svs()
    sb, (lb, pb), (kb, vb) = b.svs()
    assert len(sa) == len(sb)
    for f in sa.dtype.names:
        if f in ("allele_frequency", "logp"):
            np.testing.assert_array_equal(sa[f].view(np.uint32 if f == "allele_frequency" else np.uint64),
                                          sb[f].view(np.uint32 if f == "allele_frequency" else np.uint64), err_msg=f)
        elif f not in ("lib_begin", "cn_begin"):
            np.testing.assert_array_equal(sa[f], sb[f], err_msg=f)
    np.testing.assert_array_equal(la, lb); np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ka, kb); np.testing.assert_array_equal(va.view(np.uint32), vb.view(np.uint32))


def region_invariants(regs):
    t, s, e = regs["tid"].astype(np.int64), regs["start"].astype(np.int64), regs["end"].astype(np.int64)
    assert (e >= s).all()
    same = t[1:] == t[:-1]
    assert (t[1:] >= t[:-1]).all()
    assert (s[1:][same] > e[:-1][same]).all()           # regions of one chromosome are disjoint and in stream order
    assert (regs["n_reads"] == regs["fwd_read_count"] + regs["rev_read_count"]).all()


@pytest.fixture(scope="module")
def genome_share():
    from breakdancer_amd.synth import make_genome
    lengths = share_lengths(1 / 8 + 0.0005)
    d = make_genome(lengths, coverage=30.0, seed=11, libs=LIBS4, lib_bam=(0, 0, 0, 0), n_translocations=5000)
    assert len(d["tid"]) >= 116_000_000 and len(np.unique(d["tid"])) == 24
    cfg = "".join(cfg_line("rg%d" % i, "wgs.bam", "lib%d" % i, m, s) for i, (m, s) in enumerate(LIBS4))
    return d, cfg


def test_config2_one_gpu_share_of_the_genome(genome_share):
    """configs[2]: >= 116 M reads, 24 chromosomes, 4 libraries, default options.  (The 5,000 translocations of configs[3]
    are in the data as well; without -t they are simply more anomalous reads.)"""
    d, cfg = genome_share
    run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(), TARGETS)
    assert run.n_svs > 40000
    bd = product_from_oracle(run)
    compare(run, bd)
    # properties that need no oracle
    up = [m + 3 * s for m, s in LIBS4]
    lo = [m - 3 * s for m, s in LIBS4]
    np.testing.assert_array_equal(bd.read_class() & 0x3F, numpy_class_bytes(d, up, lo, [35] * 4))
    region_invariants(bd.regions())
    svs, _, _ = bd.svs()
    printed = svs[svs["printed"] == 1]
    n_clusters = sum(int(int(L * 7.5 / 200) * 0.01 / 12) for L in share_lengths(1 / 8 + 0.0005)) * 4  # per (chromosome, library)
    dels = printed[printed["flag"] == 2]                                    # planted DEL clusters: insert 1500-1600
    assert 0.8 * n_clusters / 4 < len(dels) < 1.25 * n_clusters / 4, (len(dels), n_clusters)
    assert 1000 < np.median(dels["size"]) < 1300, np.median(dels["size"])
    ins = printed[printed["flag"] == 3]                                     # planted INS clusters: insert 201-240
    assert len(ins) > 0.3 * n_clusters / 4 and np.median(ins["size"]) < -80, (len(ins), np.median(ins["size"]))
    n_dev, n_host, _ = bd.walk_split()
    assert n_dev > 20 * max(1, n_host)
    bh = product_from_oracle(run, host_walk=True)
    tables_equal(bd, bh)
    # the candidates placed by order key (traversals started in an earlier flush window) were ranked through k6_insert_kernel's buckets;
    # the same table from the rank sort over the whole GPU (1) and from the bitonic sort a crowded bucket falls back to (2)
    assert 1024 < bd.cross_window_svs() <= 8192
    for mode in (1, 2):
        bd.set_debug("ins_plain", mode)
        bd.run()
        tables_equal(bd, bh)
    bd.set_debug("ins_plain", 0)
    # the region table fetched by a copy command once the host knows its size, instead of forwarded by the join kernel (a measured route)
    bd.set_debug("region_dma", 1)
    bd.run()
    tables_equal(bd, bh)
    region_invariants(bd.regions())
    bd.set_debug("region_dma", 0)
    for mode in (1, 2):   # the host's share of the walk on its own copy of the region table (as before round 6) / on the table in pinned memory
        bd.set_debug("regions_copy", mode)
        bd.run()
        tables_equal(bd, bh)
        np.testing.assert_array_equal(bd.regions(), bh.regions())
    bd.set_debug("regions_copy", 0)
    bd.set_debug("asm_plain", 1)   # the walk's candidate assembly by its three-way merge (four libraries take per-library sums in registers by default)
    bd.run()
    tables_equal(bd, bh)
    bd.set_debug("asm_plain", 0)
    for fwd in (-1, 3):   # ... by every joining wave (as before round 6), by three workgroups in front of them (default: 32)
        bd.set_debug("join_fwd", fwd)
        bd.run()
        tables_equal(bd, bh)
        np.testing.assert_array_equal(bd.regions(), bh.regions())
    bd.close()
    bh.close()


def test_config3_five_thousand_translocations_with_dash_t(genome_share):
    """configs[3]: -t keeps only inter-chromosomal pairs; every planted translocation (15 pairs) must come out as one CTX row"""
    d, cfg = genome_share
    run = oracle_from_soa(d, cfg, ["wgs.bam"], make_opts(transchr_rearrange=1), TARGETS)
    assert run.W == 50
    bd = product_from_oracle(run)
    compare(run, bd)
    up = [m + 3 * s for m, s in LIBS4]
    lo = [m - 3 * s for m, s in LIBS4]
    np.testing.assert_array_equal(bd.read_class() & 0x3F, numpy_class_bytes(d, up, lo, [35] * 4, opt_t=True))
    svs, _, _ = bd.svs()
    ctx = svs[(svs["flag"] == 8) & (svs["printed"] == 1)]
    assert len(ctx) == len(svs[svs["printed"] == 1])
    assert 4800 <= len(ctx) <= 7500, len(ctx)   # (a cluster whose reads leave a gap > W = 50 splits into two regions)
    assert (ctx["chr"][:, 0] != ctx["chr"][:, 1]).all()
    assert np.median(ctx["num_reads"]) >= 13
    bd.close()
    # the same run with the 24 chromosomes spread over 4 ranks (threads sharing this GPU): the CTX reads whose mates lie on a later
    # chromosome of another rank cross in the one all-to-all, rank 0 walks the components that span ranks -- the multi-GPU shape of configs[3]
    from runner import expected_ctx_travel, sharded_from_oracle
    keep = []
    util = sharded_from_oracle(run, world=4, keep=keep, result_debug={"gather_walk": 1})   # (rank 0 walks the gathered components on its device; the 8-rank run below on its host)
    compare(run, util, check_cls=False)
    ex = keep[0].exchange
    n_ctx, n_travel = expected_ctx_travel(run, 4)
    assert n_ctx > 100_000 and sum(e["ctx_records_sent"] for e in ex) == n_travel == sum(e["ctx_records_received"] for e in ex)
    assert n_ctx // 4 < n_travel <= n_ctx // 2   # (one mate per pair at most; 3 of 4 pairs span two ranks)
    # ... and over 8 ranks, the configuration's own rank count: LPT over 24 chromosomes at 8 bins, 7 of 8 pairs crossing ranks
    keep = []
    util = sharded_from_oracle(run, world=8, keep=keep)
    compare(run, util, check_cls=False)
    ex = keep[0].exchange
    n_ctx, n_travel = expected_ctx_travel(run, 8)
    assert sum(e["ctx_records_sent"] for e in ex) == n_travel == sum(e["ctx_records_received"] for e in ex)
    assert 3 * (n_ctx // 8) < n_travel <= n_ctx // 2


@pytest.mark.parametrize("fraction,min_reads,option_sets", [(1 / 24, 46_000_000, (dict(cn_lib=1, print_af=1), dict(print_af=1))),
                                                            (1 / 8, 340_000_000, (dict(cn_lib=1, print_af=1),))])
def test_config4_tumour_normal_two_files_copy_number_and_allele_frequency(fraction, min_reads, option_sets):
    """configs[4]: tumour 60x (two libraries of 30x) + normal 30x, 6 read groups -> 3 libraries over 2 files, -a -h"""
    from breakdancer_amd.synth import make_genome
    libs = ((400.0, 30.0), (420.0, 35.0), (380.0, 28.0))   # libN1 (normal), libT1, libT2 (tumour)
    d = make_genome(share_lengths(fraction), coverage=(30.0, 30.0, 30.0), seed=13, libs=libs, lib_bam=(0, 1, 1), n_translocations=int(9600 * fraction))
    assert len(d["tid"]) >= min_reads
    cfg = ""
    for i, (lib, bam) in enumerate((("libN1", "normal.bam"), ("libT1", "tumour.bam"), ("libT2", "tumour.bam"))):
        for rg in ("a", "b"):
            cfg += cfg_line("rg%s%s" % (lib, rg), bam, lib, *libs[i])
    for kw in option_sets:
        run = oracle_from_soa(d, cfg, ["normal.bam", "tumour.bam"], make_opts(**kw), TARGETS)
        assert run.lib_names == ["libN1", "libT1", "libT2"] and run.n_svs > 15000 * (fraction * 24)
        bd = product_from_oracle(run)
        compare(run, bd)
        region_invariants(bd.regions())
        svs, _, (ck, cv) = bd.svs()
        two = svs[(svs["printed"] == 1) & (svs["region"][:, 1] >= 0) & (svs["flag"] != 8)]
        assert (two["cn_count"] == (3 if kw.get("cn_lib") else 2)).mean() > 0.8   # a copy number per library / per file
        bd.close()


def test_more_than_two_to_the_26_anomalous_reads_in_one_context():
    """The reference grows its containers as long as memory lasts (ReadRegionData.cpp:93).  Here read indices are 32-bit and only the
    REGION ids are 26-bit fields of the packed group key: a context takes 2^31 anomalous reads as long as they form at most
    2^26 - 2 accepted regions.  150 M reads of which half are discordant (75 M anomalous reads, ~3 M regions): no oracle run at
    this size, nor the host walk (19 minutes for these 75 M reads) -- region-table invariants, census of the anomalous reads against
    a numpy restatement, recovery of the planted clusters, and the same tables when the stream arrives in ragged batches."""
    import breakdancer_amd as bda
    from breakdancer_amd.api import LibraryConfig, Options
    from breakdancer_amd.synth import LIB_C2, make_chromosome
    d = make_chromosome(length=500_000_000, coverage=30.0, seed=5, discordant=0.5, cluster=24)
    n = len(d["tid"])
    mq_ok = d["mapq"] > 35
    sam = d["flag"].astype(np.int64)
    ai = np.abs(d["isize"])
    fr = ((sam & 0x10) != 0) != ((sam & 0x20) != 0)
    rf = fr & ((d["pos"] < d["mpos"]) == ((sam & 0x10) != 0))
    anom = ((~fr) | rf | (fr & ~rf & ((ai > 490) | (ai < 310)))) & mq_ok
    assert int(anom.sum()) > (1 << 26) + 5_000_000

    def run(cuts):
        bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, max_read_window_size=200)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            bd.push_reads({k: v[lo:hi] for k, v in d.items()})
        return bd.run()
    a = run([0, n])
    sa = a.summary()
    assert sa["n_reads"] == n and sa["n_anomalous"] == int(anom.sum())
    assert 1_000_000 < sa["n_regions"] < (1 << 26)
    region_invariants(a.regions())
    svs, _, _ = a.svs()
    dels = svs[(svs["flag"] == 2) & (svs["printed"] == 1)]
    assert len(dels) > 10_000 and abs(np.median(dels["size"]) - 1150) < 60   # (most clusters of this dense input score below the print threshold)
    b = run([0, 1, 70_000_001, 70_000_002, 140_000_000, n])
    assert b.summary() == sa
    tables_equal(a, b)
    a.close()
    b.close()
