"""CPU tests: the oracle against the reference's golden vectors, plus host-side logic."""
import json
import math
import os

import numpy as np
import pytest

from helpers import GOLDEN, OracleRun, filter_cmd_lines, load_chr21, make_opts, oracle_lib

TID21 = 22

CASES = [("expected_output", dict(chr_tid=TID21)), ("expected_output", dict()),
         ("expected_output.af", dict(chr_tid=TID21, print_af=1)), ("expected_output.af", dict(print_af=1)),
         ("expected_output.cn_per_lib", dict(chr_tid=TID21, cn_lib=1)),
         ("expected_output.cn_per_lib.af", dict(chr_tid=TID21, cn_lib=1, print_af=1))]


@pytest.mark.parametrize("fn,kw", CASES)
def test_oracle_reproduces_reference_golden_tables(fn, kw):
    """integration-test/breakdancer_test.py:28-104: stdout minus #Command/#Software == test-data/expected_output*"""
    run = load_chr21(make_opts(**kw)).run()
    exp = filter_cmd_lines(open(os.path.join(GOLDEN, "chr21", fn)).read())
    assert filter_cmd_lines(run.text) == exp


@pytest.mark.parametrize("fn,kw", CASES)
def test_oracle_bam_front_end_reproduces_the_golden_tables(fn, kw):
    """the same six invocations with the records decoded by oracle/bd_oracle_bam.cpp (the single-threaded, reference-shaped
    reader bench.py's cpu_baseline times) instead of the pure-Python decoder"""
    gd = os.path.join(GOLDEN, "chr21")
    run = OracleRun(open(os.path.join(gd, "inv_del_bam_config")).read(), make_opts(**kw))
    counts = []
    for b, name in enumerate(run.bam_names):
        n, _ = run.load_bam(b, os.path.join(gd, name), only_tid=kw.get("chr_tid", -1), passes=2, set_targets=(b == 0))
        counts.append(n)
    if "chr_tid" not in kw:
        assert counts == [3069, 2848]
    run.run()
    assert filter_cmd_lines(run.text) == filter_cmd_lines(open(os.path.join(gd, fn)).read())
    assert oracle_lib().bdo_bam_tid(os.path.join(gd, run.bam_names[0]).encode(), b"21") == TID21


def test_chr21_read_counts():
    """test-data/TestData.hpp.in:21-24: 3069 / 2848 primary aligned records"""
    run = load_chr21(make_opts())
    n = {name: len(s["tid"]) for name, s in zip(run.bam_names, run.streams)}
    assert n["NA19238_chr21_del_inv.bam"] == 3069 and n["NA19240_chr21_del_inv.bam"] == 2848


def test_chr21_worked_example():
    """W, ref_len, region table of the -o 21 run"""
    run = load_chr21(make_opts(chr_tid=TID21)).run()
    assert run.W == 287 and run.ref_len == 5626088 and run.n_merged == 5917
    assert run.regions[:, 2].tolist() == [29184970, 29186121, 29186694, 34807608, 34809176, 34810172, 34810880]
    assert run.regions[:, 3].tolist() == [29185376, 29186400, 29186880, 34808851, 34809713, 34810512, 34810985]
    assert run.regions[:, 4].tolist()[:6] == [65, 312, 241, 805, 509, 385]


def test_poisson_restatement_against_mpmath():
    v = json.load(open(os.path.join(GOLDEN, "poisson_vectors.json")))
    L = oracle_lib()
    for r in v["poisson"]:
        p = L.bdo_poisson_upper_tail(float(r["lambda"]), r["k"])
        if r["p_double_positive"]:
            want = float(r["logp"])
            assert abs(math.log(p) - want) <= 1e-12 * max(1.0, abs(want)), r
        else:
            assert p < 2.3e-308
    for r in v["chisq"]:
        q = L.bdo_chisq_upper_tail(float(r["df"]), float(r["x"]))
        want = float(r["logq"])
        assert abs(math.log(q) - want) <= 1e-12 * max(1.0, abs(want)), r


CFG = ("readgroup:rg1\tplatform:illumina\tmap:x.bam\treadlen:90.00\tlib:lib1\tnum:10001\tlower:277.03\tupper:525.50\tmean:467.59\tstd:31.91\tSWnormality:minus infinity\texe:samtools view\n"
       "readgroup:rg2\tplatform:illumina\tmap:x.bam\treadlen:90.00\tlib:lib1\tnum:10001\tlower:277.03\tupper:525.50\tmean:467.59\tstd:31.91\n"
       "readgroup:rg3\tmapqual:10\tplatform:illumina\tmap:y.bam\treadlen:90.00\tlib:lib2\tnum:10001\tlower:311.36\tupper:532.53\tmean:475.76\tstd:28.67\n"
       "readgroup:rg4\tmapqual:10\tplatform:illumina\tmap:y.bam\treadlen:90.00\tlib:lib2\tnum:10001\tlower:311.36\tupper:532.53\tmean:475.76\tstd:28.67\n")


def test_config_parser_legacy_cases():
    """test/lib/io/TestBamConfig.cpp:35-110 (legacyParse)"""
    r = OracleRun(CFG, make_opts())
    assert r.bam_names == ["x.bam", "y.bam"] and r.lib_names == ["lib1", "lib2"]
    assert [r.lib_of_rg(g) for g in ("rg1", "rg2", "rg3", "rg4")] == [0, 0, 1, 1]
    assert r.lib_i[:, 1].tolist() == [0, 1] and r.lib_i[:, 0].tolist() == [-1, 10]
    np.testing.assert_allclose(r.lib_f[0], [467.59, 31.91, 525.50, 277.03, 90.0], rtol=1e-6)
    np.testing.assert_allclose(r.lib_f[1], [475.76, 28.67, 532.53, 311.36, 90.0], rtol=1e-6)
    assert r.w0 == int(np.float32(467.59) - np.float32(90.0) * 2)
    assert r.lib_of_rg("nope") == 0  # falls back to the library of the alphabetically first BAM


def test_config_cutoffs_from_mean_std():
    """io/BamConfig.cpp:64-68: missing lower/upper come from mean +- cut_sd * std, lower clamped at 0"""
    r = OracleRun("map:a.bam\tlib:l\tmean:100\tstd:40\treadlen:50\n", make_opts(cut_sd=3))
    assert r.lib_f[0, 2] == np.float32(220.0) and r.lib_f[0, 3] == 0.0
    assert r.w0 == 50  # max(50, int(100 - 2*50))


def test_token_translation():
    """io/BamConfigEntry.cpp:43-54 / test/lib/io/TestBamConfigEntry.cpp:40-84"""
    L = oracle_lib()
    names = ["BAM_FILE", "LIBRARY_NAME", "READ_GROUP", "INSERT_SIZE_MEAN", "INSERT_SIZE_STDDEV", "READ_LENGTH",
             "INSERT_SIZE_UPPER_CUTOFF", "INSERT_SIZE_LOWER_CUTOFF", "MIN_MAP_QUAL", "SAMPLE_NAME", "UNKNOWN"]
    want = {"map": "BAM_FILE", "bam_map": "BAM_FILE", "lib": "LIBRARY_NAME", "library": "LIBRARY_NAME", "group": "READ_GROUP",
            "readgroup": "READ_GROUP", "mean": "INSERT_SIZE_MEAN", "mean_insertsize": "INSERT_SIZE_MEAN",
            "std": "INSERT_SIZE_STDDEV", "stddev": "INSERT_SIZE_STDDEV", "readlen": "READ_LENGTH", "readlength": "READ_LENGTH",
            "upper": "INSERT_SIZE_UPPER_CUTOFF", "uppercutoff": "INSERT_SIZE_UPPER_CUTOFF", "lower": "INSERT_SIZE_LOWER_CUTOFF",
            "mapqual": "MIN_MAP_QUAL", "map_qual": "MIN_MAP_QUAL", "mapping_quality": "MIN_MAP_QUAL", "samp": "SAMPLE_NAME",
            "sample": "SAMPLE_NAME", "platform": "UNKNOWN", "num": "UNKNOWN", "exe": "UNKNOWN", "MAP": "BAM_FILE"}
    for k, v in want.items():
        assert names[L.bdo_translate_token(k.encode())] == v, k


def test_classifier_truth_table():
    """test/lib/io/TestIlluminaPEReadClassifier.cpp prints this table without asserting; the order of the
    tests in IlluminaPEReadClassifier.cpp:59-101 is what is pinned here."""
    L = oracle_lib()
    NA, FF, LARGE, SMALL, RF, RR, NFR, NRF, CTX, MU, UN = range(11)
    cl = lambda sam, tid=0, mtid=0, pos=100, mpos=400, isz=400: L.bdo_classify(sam, tid, mtid, pos, mpos, isz, 490.0, 310.0)
    assert cl(0x1 | 0x400) == NA and cl(0x0) == NA
    assert cl(0x1 | 0x4) == UN and cl(0x1 | 0x8) == MU and cl(0x1 | 0x4 | 0x8) == UN
    assert cl(0x1, mtid=3) == CTX
    assert cl(0x1) == FF and cl(0x1 | 0x10 | 0x20) == RR
    assert cl(0x1 | 0x20) == NFR                      # leftmost forward, mate reverse
    assert cl(0x1 | 0x10, pos=400, mpos=100) == NFR   # rightmost reverse
    assert cl(0x1 | 0x10) == RF and cl(0x1 | 0x20, pos=400, mpos=100) == RF
    assert cl(0x1 | 0x20, isz=491) == LARGE and cl(0x1 | 0x20, isz=490) == NFR
    assert cl(0x1 | 0x20, isz=309) == SMALL and cl(0x1 | 0x20, isz=310) == NFR
    assert cl(0x1 | 0x20, pos=100, mpos=100) == RF    # leftmost is strict: equal positions count as not leftmost


def test_fuzz_cases_run_through_the_oracle():
    from fuzzgen import OPTION_SETS, make_case
    from runner import oracle_case
    nsv = 0
    for seed in range(6):
        cfg, streams, targets = make_case(seed)
        for o in (OPTION_SETS[seed % len(OPTION_SETS)], OPTION_SETS[(seed * 7 + 3) % len(OPTION_SETS)]):
            run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
            nsv += run.n_svs
            assert run.n_merged > 0
    assert nsv > 0
