"""Shared machinery of the parity tests: run the oracle, run the product on the same merged stream, compare."""
import numpy as np

from helpers import OracleRun, make_opts

import breakdancer_amd as bda
from breakdancer_amd.api import Options, LibraryConfig


def oracle_case(config, streams, targets, opts):
    run = OracleRun(config, opts)
    run.set_targets(targets)
    for b, st in enumerate(streams):
        d = dict(st)
        d["lib"] = np.array([run.lib_of_rg(g) for g in st["rg"]], dtype=np.int32)
        run.set_stream(b, d)
    return run.run()


def product_options(opts):
    return Options(min_len=opts["min_len"], cut_sd=opts["cut_sd"], max_sd=opts["max_sd"], min_map_qual=opts["min_map_qual"],
                   min_read_pair=opts["min_read_pair"], seq_coverage_lim=opts["seq_coverage_lim"],
                   buffer_size=opts["buffer_size"], transchr_rearrange=bool(opts["transchr_rearrange"]),
                   fisher=bool(opts["fisher"]), Illumina_long_insert=bool(opts["illumina_long_insert"]),
                   CN_lib=bool(opts["cn_lib"]), print_AF=bool(opts["print_af"]), score_threshold=opts["score_threshold"],
                   chr="x" if opts["chr_tid"] >= 0 else "")


# the tests select the product's alternative routes through bdx_set_debug; monkeypatch.setenv("BDX_<NAME>", value) in a test is
# only the way the choice reaches this helper.  libbdx.so reads no environment variable for a behaviour switch: they travel through
# bdx_set_debug, bdx_set_process_option and bdx_bamdec_params; what it does read only adds output (BDX_ALLOC_TRACE, BDX_DIST_TRACE,
# BDX_BAMDEC_TRACE, BDX_WALK_PROFILE on stderr, BDX_KZ_PROF=<file>) -- tests/test_abi.py greps csrc/ for anything else
_SWITCHES = {"BDX_NO_STASH": "no_stash", "BDX_MAX_CHUNKS": "max_chunks", "BDX_SPEC_TEST": "spec_test", "BDX_BIG_WALK": "big_walk",
             "BDX_BUCKETED_JOIN": "bucketed_join"}


def apply_test_switches(bd):
    import os
    for env, name in _SWITCHES.items():
        if env in os.environ:
            bd.set_debug(name, int(os.environ[env]))
    if os.environ.get("BDX_NO_SPECULATE") == "1":
        bd.set_enqueue_ahead(0)
    return bd


def colliding_names(soa, share):
    """the stream with name keys that `share` different read names have in common, and a second hash that tells them apart
    (the oracle keeps the exact names: with bdx_use_name_check the product must still agree with it)"""
    out = dict(soa)
    ids = np.asarray(soa["name_id"], dtype=np.uint64)
    out["name_id"] = ids // np.uint64(share)
    out["name_check"] = (ids * np.uint64(0x9E3779B97F4A7C15)) ^ (ids >> np.uint64(7))
    return out


def product_from_oracle(run, device=0, support=False, host_walk=False, collide=0, name_check=True):
    """Feed the product the exact merged stream the oracle consumed (the producer's job in the CLI)."""
    libs = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                          bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
    bd = apply_test_switches(bda.BreakDancer(product_options(run.opts), libs, run.nbams, ntids=0, max_read_window_size=run.w0, device=device))
    soa = run.merged_soa()
    if collide:
        soa = colliding_names(soa, collide)
        if name_check:
            bd.use_name_check()
    if support:
        bd.collect_support()
    if host_walk:
        bd.set_host_walk(True)
    if run.n_merged:
        bd.push_reads(soa)
    bd.run()
    return bd


def compare(run, bd, logp_tol=1e-9, check_cls=True):
    s = bd.summary()
    if check_cls:
        assert s["n_reads"] == run.n_merged
    assert s["covered_ref_len"] == run.ref_len, (s["covered_ref_len"], run.ref_len)
    assert s["window"] == run.W, (s["window"], run.W)
    c = bd.counters()
    np.testing.assert_array_equal(c["lib_read_count"], run.lib_cnt)
    np.testing.assert_array_equal(c["bam_read_count"], run.bam_cnt)
    np.testing.assert_array_equal(c["flag_hist"], run.hist)
    np.testing.assert_array_equal(c["seqcov"].view(np.uint32), run.seqcov.view(np.uint32))
    if run.n_merged and check_cls:
        cls = bd.read_class()
        np.testing.assert_array_equal(cls & 0x3F, run.cls)
    # regions as created by add_region
    regs = bd.regions()
    assert len(regs) == run.n_regions, (len(regs), run.n_regions)
    if run.n_regions:
        o = run.regions
        for name, col in (("tid", 1), ("start", 2), ("end", 3), ("normal_read_pairs", 4), ("fwd_read_count", 5),
                          ("rev_read_count", 6), ("n_reads", 7), ("stored", 8)):
            np.testing.assert_array_equal(regs[name], o[:, col], err_msg=name)
    svs, (li, lp), (ck, cv) = bd.svs()
    assert len(svs) == run.n_svs, (len(svs), run.n_svs)
    if run.n_svs:
        oi, od = run.sv_i, run.sv_d
        np.testing.assert_array_equal(svs["chr"][:, 0], oi[:, 0]); np.testing.assert_array_equal(svs["pos"][:, 0], oi[:, 1])
        np.testing.assert_array_equal(svs["fwd"][:, 0], oi[:, 2]); np.testing.assert_array_equal(svs["rev"][:, 0], oi[:, 3])
        np.testing.assert_array_equal(svs["chr"][:, 1], oi[:, 4]); np.testing.assert_array_equal(svs["pos"][:, 1], oi[:, 5])
        np.testing.assert_array_equal(svs["fwd"][:, 1], oi[:, 6]); np.testing.assert_array_equal(svs["rev"][:, 1], oi[:, 7])
        np.testing.assert_array_equal(svs["flag"], oi[:, 8], err_msg="flag")
        np.testing.assert_array_equal(svs["size"], oi[:, 9], err_msg="size")
        np.testing.assert_array_equal(svs["num_reads"], oi[:, 11], err_msg="num_reads")
        np.testing.assert_array_equal(svs["lib_count"], oi[:, 13]); np.testing.assert_array_equal(svs["cn_count"], oi[:, 14])
        # float32 copy numbers / allele frequency: bit-exact (NaNs compared by bit pattern up to sign/payload class)
        af_p, af_o = svs["allele_frequency"], od[:, 1].astype(np.float32)
        np.testing.assert_array_equal(np.isnan(af_p), np.isnan(af_o))
        m = ~np.isnan(af_o)
        np.testing.assert_array_equal(af_p[m].view(np.uint32), af_o[m].view(np.uint32))
        np.testing.assert_array_equal(li, run.sv_lib[:, 0]); np.testing.assert_array_equal(lp, run.sv_lib[:, 1])
        np.testing.assert_array_equal(ck, run.sv_cn_key)
        np.testing.assert_array_equal(cv.view(np.uint32), run.sv_cn_val.view(np.uint32))
        # Poisson log tail: tolerance stated by north_star is 1e-6; we hold 1e-9 relative
        lp_p, lp_o = svs["logp"], od[:, 0]
        np.testing.assert_array_equal(np.isnan(lp_p), np.isnan(lp_o))
        np.testing.assert_array_equal(np.isinf(lp_p), np.isinf(lp_o))
        f = np.isfinite(lp_o)
        if f.any():
            err = np.abs(lp_p[f] - lp_o[f]) / np.maximum(1.0, np.abs(lp_o[f]))
            assert err.max() < logp_tol, err.max()
        np.testing.assert_array_equal(svs["score"], oi[:, 10], err_msg="score")
        np.testing.assert_array_equal(svs["printed"], oi[:, 12], err_msg="printed")
    return s


def split_by_tid(soa):
    """merged SoA -> {tid: SoA of that chromosome} (stream order preserved)"""
    out = {}
    tids = soa["tid"]
    for t in np.unique(tids):
        m = tids == t
        out[int(t)] = {k: v[m] for k, v in soa.items()}
    return out


def sharded_from_oracle(run, comm=None, device=0, world=1, keep=None, collide=0, support=False, result_debug=None):
    """the same whole-genome input through the chromosome-sharded path: one context per chromosome, the chromosomes dealt
    to `world` ranks (threads of this process on one GPU when comm is None)"""
    from breakdancer_amd.shard import ShardedRun
    libs = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                          bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
    sr = ShardedRun(product_options(run.opts), libs, run.nbams, run.w0, comm=comm, device=device, world=world,
                    ntids=len(getattr(run, "targets", [])) or None, support=support)
    if result_debug:
        sr.result_debug.update(result_debug)
    soa = run.merged_soa()
    if collide:
        soa = colliding_names(soa, collide)
    for tid, arrs in split_by_tid(soa).items():
        sr.add_chromosome(tid, arrs)
    res = sr.run()
    res._sharded_run = sr  # the result is a view into rank 0's context
    if keep is not None:
        keep.append(sr)
    return res


def compare_support(run, bd):
    """supporting reads of every SV: same reads, same order, same per-read flags as SvBuilder::support_reads"""
    off, idx, flg = bd.sv_support()
    np.testing.assert_array_equal(off.astype(np.int64), run.sup_off)
    np.testing.assert_array_equal(idx.astype(np.int64), run.sup_idx)
    np.testing.assert_array_equal(flg, run.sup_flag)


def expected_ctx_travel(run, world):
    """(inter-chromosomal reads that pass the filters, those of them that must cross ranks in a sharded run over `world` ranks): a CTX
    read travels exactly when its mate lies on a LATER chromosome held by another rank (chromosomes dealt as ShardedRun deals them)"""
    from breakdancer_amd.shard import plan_chromosomes
    soa = run.merged_soa()
    tid, mtid = soa["tid"].astype(np.int64), soa["mtid"].astype(np.int64)
    nt = int(max(tid.max(), mtid.max())) + 1
    counts = {int(t): int(c) for t, c in zip(*np.unique(tid, return_counts=True))}
    ro = np.full(nt, -1, np.int64)
    for r, tids in enumerate(plan_chromosomes(counts, world)):
        for t in tids:
            ro[t] = r
    ctx = (run.cls & 0x1F) == (0x10 | 8)   # passing reads classified ARP_CTX
    mt = np.clip(mtid, 0, nt - 1)
    travels = ctx & (mtid > tid) & (ro[mt] >= 0) & (ro[mt] != ro[tid])
    return int(ctx.sum()), int(travels.sum())
