"""GPU tests of the chromosome-sharded paths (breakdancer_amd/shard.py) on one GPU: one context per chromosome,
LocalComm (world of one).  The staged whole-genome run must equal ONE oracle run over all chromosomes -- including the
cross-chromosome effects (global window / lambda, prefix counters, the closing read of the next chromosome, CTX mates)."""
import numpy as np
import pytest

from fuzzgen import make_case
from helpers import load_chr21, make_opts
from runner import compare, oracle_case, product_options, sharded_from_oracle, split_by_tid

pytestmark = pytest.mark.gpu

WG_OPTS = [dict(), dict(transchr_rearrange=1), dict(cn_lib=1, print_af=1), dict(buffer_size=1), dict(buffer_size=3, min_read_pair=1),
           dict(min_map_qual=0, min_len=0), dict(transchr_rearrange=1, min_read_pair=1), dict(illumina_long_insert=1), dict(fisher=1),
           dict(seq_coverage_lim=3), dict(max_sd=900)]


@pytest.mark.parametrize("seed", range(24))
def test_staged_whole_genome_equals_single_run(seed):
    cfg, streams, targets = make_case(100 + seed)
    for o in (WG_OPTS[seed % len(WG_OPTS)], WG_OPTS[(seed * 5 + 2) % len(WG_OPTS)]):
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        util = sharded_from_oracle(run)
        compare(run, util, check_cls=False)


def test_staged_chr21_all_sequences():
    run = load_chr21(make_opts()).run()
    util = sharded_from_oracle(run)
    s = compare(run, util, check_cls=False)
    assert s["n_svs_printed"] == 4


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
