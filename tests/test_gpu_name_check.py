"""Mates are joined on two hashes of the read name (bdx_use_name_check): read names whose 64-bit keys collide are not taken
for one name.  The reference compares the names themselves (ReadRegionData.cpp:109 `_read_regions[qname]`,
SvBuilder.cpp:101-118 `_observe_read`); the oracle here keeps exact name ids, the product gets keys that several names share
plus the second hash, and must still produce the oracle's table -- in the direct join, the bucketed join, the read-level replay,
and across the ranks of a sharded run."""
import numpy as np
import pytest

from fuzzgen import GRAPH_OPTION_SETS, OPTION_SETS, clash_names, make_case, make_graph_case
from helpers import make_opts
from runner import compare, compare_support, oracle_case, product_from_oracle, sharded_from_oracle

pytestmark = pytest.mark.gpu


def _differs(run, bd):
    try:
        compare(run, bd)
    except AssertionError:
        return True
    return bd.was_replayed()   # (or the collisions looked like third sightings and sent the run through the replay)


@pytest.mark.parametrize("seed", range(10))
def test_names_with_equal_keys_are_not_joined(seed):
    if seed % 2 == 0:
        cfg, streams, targets = make_case(1200 + seed)
        osets = OPTION_SETS
    else:
        cfg, streams, targets = make_graph_case(1200 + seed)
        osets = GRAPH_OPTION_SETS
    share = (2, 3, 4, 7)[seed % 4]
    blind = 0
    for o in (osets[seed % len(osets)], dict(min_read_pair=1, buffer_size=1)):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        bd = product_from_oracle(run, support=True, collide=share)
        compare(run, bd)
        compare_support(run, bd)
        assert not bd.was_replayed()   # colliding keys are sorted out in the join itself, not by the replay
        bd.close()
        # the same keys without the second hash: the key alone joins the wrong reads (or takes them for a name seen three times)
        bd = product_from_oracle(run, collide=share, name_check=False)
        blind += _differs(run, bd)
        bd.close()
    assert blind > 0


@pytest.mark.parametrize("seed", [2, 5])
def test_bucketed_join_compares_the_second_hash(seed, monkeypatch):
    monkeypatch.setenv("BDX_BUCKETED_JOIN", "1")
    cfg, streams, targets = make_case(1230 + seed)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, min_read_pair=1))
    bd = product_from_oracle(run, support=True, collide=4)
    compare(run, bd)
    compare_support(run, bd)
    assert not bd.was_replayed()
    bd.close()


@pytest.mark.parametrize("seed", range(6))
def test_replay_tells_names_apart_by_both_hashes(seed):
    """names really seen three and four times (clashing files) AND keys shared by different names: the replay numbers the
    (key, check) pairs, so the reference's semantics for the true clashes apply to the true names only"""
    cfg, streams, targets = (make_case if seed % 2 == 0 else make_graph_case)(1260 + seed)
    streams = clash_names(streams, seed, frac=0.04)
    replayed = 0
    for o in (dict(min_read_pair=1), dict(min_read_pair=1, buffer_size=1)):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        bd = product_from_oracle(run, support=True, collide=3)
        compare(run, bd)
        compare_support(run, bd)
        replayed += bd.was_replayed()
        bd.close()
    assert replayed > 0


@pytest.mark.parametrize("seed", range(6))
def test_sharded_run_joins_inter_chromosomal_mates_on_both_hashes(seed):
    """the CTX join records carry the second hash across ranks (32-byte exchange entries); with true clashes on top the gathered
    replay on rank 0 numbers (key, check) pairs as well"""
    cfg, streams, targets = (make_case if seed % 2 == 0 else make_graph_case)(1290 + seed)
    if seed >= 4:
        streams = clash_names(streams, seed, frac=0.03)
    for i, o in enumerate((dict(min_read_pair=1), dict(transchr_rearrange=1, min_read_pair=1))):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        util = sharded_from_oracle(run, world=2 + (seed + i) % 2, collide=(2, 5)[i])
        compare(run, util, check_cls=False)
        if seed < 4:
            assert not util.was_replayed()


def test_second_hash_through_the_staging_ring_and_adopted_device_arrays():
    import ctypes as C
    import breakdancer_amd as bda
    from breakdancer_amd.api import BATCH_FIELDS, LibraryConfig
    from runner import colliding_names, product_options
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipHostFree.argtypes = [C.c_void_p]
    hip.hipFree.argtypes = [C.c_void_p]
    cfg, streams, targets = make_case(1333)
    run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, min_read_pair=1))
    libs = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                          bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
    soa = colliding_names(run.merged_soa(), 4)
    fields = list(BATCH_FIELDS) + [("name_check", np.uint64)]

    def column(k, dt):
        src = soa.get(k)
        if src is None:
            src = soa["bdqual"] if k == "mapq" else soa["name_id"]
        return np.ascontiguousarray(src, dtype=dt)

    def new_ctx():
        return bda.BreakDancer(product_options(run.opts), libs, run.nbams, ntids=0, max_read_window_size=run.w0).use_name_check()

    # (a) bdx_acquire_batch / bdx_submit_batch
    bd = new_ctx()
    bd.stream_reads(soa, batch=777)
    bd.run()
    compare(run, bd)
    bd.close()
    # (b) bdx_set_device_reads: arrays the caller owns in HBM
    dev, ptrs = [], {}
    for k, dt in fields:
        a = column(k, dt)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(a.nbytes, 16)) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        dev.append(p)
        ptrs[k] = p.value
    bd = new_ctx()
    bd.set_device_reads(ptrs, run.n_merged)
    bd.run()
    compare(run, bd)
    bd.close()
    for p in dev:
        hip.hipFree(p)
    # (c) pinned host arrays: keys, lengths and the second hash of the anomalous reads are fetched from there by K2
    pinned, hp = {}, []
    for k, dt in fields:
        a = column(k, dt)
        p = C.c_void_p()
        assert hip.hipHostMalloc(C.byref(p), max(a.nbytes, 16), 0) == 0
        hp.append(p)
        v = np.ctypeslib.as_array((C.c_uint8 * a.nbytes).from_address(p.value)).view(dt)
        v[:] = a
        pinned[k] = v
    bd = new_ctx()
    bd.push_reads(pinned)
    bd.run()
    compare(run, bd)
    bd.close()
    for p in hp:
        hip.hipHostFree(p)
    # a batch without the second hash is refused once it has been declared
    bd = new_ctx()
    with pytest.raises(bda.BdxError):
        bd.push_reads({k: v for k, v in soa.items() if k != "name_check"})
    bd.close()
