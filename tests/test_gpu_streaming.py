"""GPU tests of the three ways reads reach the resident store (include/bdx.h): bdx_push of caller memory in one or many
batches (pageable, or pinned with the name keys left on the host), the acquire/submit staging ring of the streaming
producer, and adopted device arrays -- all must give the result of the oracle on the same records, with the classifier
running behind the copies as the batches arrive."""
import numpy as np
import pytest

from helpers import make_opts
from runner import compare, product_options
from test_gpu_configs import cfg_line, oracle_from_soa
from test_gpu_fullsize import tables_equal

import breakdancer_amd as bda
from breakdancer_amd.api import BATCH_FIELDS, LibraryConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case():
    from breakdancer_amd.synth import make_genome
    libs = ((400.0, 30.0), (330.0, 25.0))
    d = make_genome([9_000_000, 6_000_000], coverage=30.0, seed=31, libs=libs, lib_bam=(0, 1), n_translocations=60)
    assert len(d["tid"]) > 4_000_000   # several classifier launches behind the batches (one per >= 1 M new reads)
    cfg = cfg_line("rgA", "a.bam", "libA", *libs[0]) + cfg_line("rgB", "b.bam", "libB", *libs[1])
    run = oracle_from_soa(d, cfg, ["a.bam", "b.bam"], make_opts(score_threshold=-1), ["c1", "c2"])
    m = run.merged_soa()   # the stream in the reference's merge order (ties between the two files), in the batch layout
    d = dict(tid=m["tid"], pos=m["pos"], mtid=m["mtid"], mpos=m["mpos"], isize=m["isize"], flag=m["flag"],
             qlen=m["qlen"].astype(np.uint16), mapq=m["bdqual"], lib=m["lib"].astype(np.uint8), bam=m["bam"], name_key=m["name_id"])
    return d, run


def new_ctx(run):
    libs = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                          bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
    return bda.BreakDancer(product_options(run.opts), libs, run.nbams, ntids=0, max_read_window_size=run.w0)


def batches(d, sizes):
    lo = 0
    n = len(d["tid"])
    i = 0
    while lo < n:
        m = min(sizes[i % len(sizes)], n - lo)
        yield {k: v[lo:lo + m] for k, v in d.items()}
        lo += m
        i += 1


def test_push_in_many_batches_equals_one_batch(case):
    d, run = case
    one = new_ctx(run)
    one.push_reads(d)
    compare(run, one.run())
    many = new_ctx(run)   # no reserve: the store grows, the per-tile tables are re-laid, the classifier starts over
    for b in batches(d, [700_001, 1_300_000, 17, 2_000_003]):
        many.push_reads(b)
    compare(run, many.run())
    tables_equal(one, many)
    reserved = new_ctx(run)
    reserved.lib.bdx_reserve(reserved.h, len(d["tid"]))
    for b in batches(d, [1_100_000, 999, 1_500_000]):
        reserved.push_reads(b)
    compare(run, reserved.run())
    compare(run, reserved.run())   # a repeated run classifies from the first tile again
    tables_equal(one, reserved)
    # bdx_reserve also sizes the later stages' buffers (for 1/32 of the reserved reads anomalous): a reservation far below the
    # input (every buffer has to grow inside the run) and one far above it
    for guess in (1 << 20, 40_000_000):
        other = new_ctx(run)
        other.lib.bdx_reserve(other.h, guess)
        other.push_reads(d)
        compare(run, other.run())
        tables_equal(one, other)
        other.close()
    for x in (one, many, reserved):
        x.close()


def test_staging_ring_of_the_streaming_producer(case):
    d, run = case
    ref = new_ctx(run)
    ref.push_reads(d)
    ref.run()
    for batch, reserve in ((1 << 20, True), (300_000, True), (1 << 19, False)):   # 5, 15 and 9 trips round the ring of four
        bd = new_ctx(run)
        if reserve:
            bd.lib.bdx_reserve(bd.h, len(d["tid"]))
        bd.stream_reads(d, batch=batch)
        compare(run, bd.run())
        tables_equal(ref, bd)
        bd.close()
    ref.close()


def test_pinned_batches_keep_their_name_keys_on_the_host(case):
    """bdx_push of pinned arrays: 25 of the 35 bytes per read are copied, the compaction kernel fetches the name keys and
    read lengths of the anomalous reads from the caller's memory"""
    import ctypes as C
    d, run = case
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    pinned, ptrs = {}, []
    for k, dt in BATCH_FIELDS:
        a = np.ascontiguousarray(d[k], dtype=dt)
        ptr = C.c_void_p()
        assert hip.hipHostMalloc(C.byref(ptr), a.nbytes, 0) == 0
        ptrs.append(ptr)
        v = np.ctypeslib.as_array((C.c_uint8 * a.nbytes).from_address(ptr.value)).view(dt)
        v[:] = a
        pinned[k] = v
    bd = new_ctx(run)
    bd.lib.bdx_reserve(bd.h, len(d["tid"]))
    for i, b in enumerate(batches(pinned, [1_200_000, 64, 900_000])):   # mixed: every third batch from pageable memory
        bd.push_reads({k: np.array(v) for k, v in b.items()} if i % 3 == 2 else b)
    compare(run, bd.run())
    ref = new_ctx(run)
    ref.push_reads(d)
    ref.run()
    tables_equal(ref, bd)
    bd.close()
    ref.close()
    for ptr in ptrs:
        hip.hipHostFree(ptr)


def test_one_context_takes_the_next_chromosome_after_a_reset(case):
    d, run = case
    m0 = d["tid"] == 0
    parts = [{k: v[m0] for k, v in d.items()}, {k: v[~m0] for k, v in d.items()}]
    bd = new_ctx(run)
    fresh = []
    for p in parts:
        f = new_ctx(run)
        f.push_reads(p)
        fresh.append(f.run())
    for i, p in enumerate(parts + parts[:1]):
        bd.reset_reads()
        bd.stream_reads(p, batch=1 << 20)
        bd.run()
        tables_equal(fresh[i % 2], bd)
        assert bd.summary()["n_reads"] == len(p["tid"])
    for x in fresh + [bd]:
        x.close()
