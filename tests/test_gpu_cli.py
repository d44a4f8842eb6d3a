"""GPU tests: the breakdancer-max CLI (BAM -> producer -> libbdx -> formatter) against the reference's golden
files, the way integration-test/breakdancer_test.py does (stdout minus #Command / #Software lines)."""
import os
import subprocess

import pytest

from helpers import GOLDEN, ROOT, filter_cmd_lines, load_chr21, make_opts

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "bin", "breakdancer-max")
CWD = os.path.join(GOLDEN, "chr21")

# integration-test/breakdancer_test.py:32,46,60,74,88,102
CASES = [("expected_output.cn_per_lib", ["-a", "-o", "21"]), ("expected_output.cn_per_lib.af", ["-a", "-h", "-o", "21"]),
         ("expected_output.af", ["-h", "-o", "21"]), ("expected_output", ["-o", "21"]), ("expected_output", []),
         ("expected_output.af", ["-h"])]


def run_cli(args, env=None):
    p = subprocess.run([EXE] + args + ["inv_del_bam_config"], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, **env) if env else None)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


@pytest.mark.parametrize("fn,args", CASES)
def test_cli_reproduces_reference_golden_output(fn, args):
    got = filter_cmd_lines(run_cli(args))
    exp = filter_cmd_lines(open(os.path.join(CWD, fn)).read())
    assert got == exp


@pytest.mark.parametrize("fn,args", CASES)
@pytest.mark.parametrize("threads", ["2", "5"])
def test_cli_table_written_by_several_threads_is_the_same_text(fn, args, threads):
    """a genome's table (tens of thousands of rows) is formatted by several threads, each starting in the stream state the sequential
    loop would have reached (the reference never resets std::fixed / setprecision(2) on cout, BreakDancer.cpp:492-493: allele
    frequencies behind the first copy number print with two decimals); forced here on the golden cases"""
    got = filter_cmd_lines(run_cli(args, env={"BDX_FORMAT_THREADS": threads}))
    exp = filter_cmd_lines(open(os.path.join(CWD, fn)).read())
    assert got == exp


OPTSETS = [(["-t"], dict(transchr_rearrange=1)), (["-l", "-y", "-1"], dict(illumina_long_insert=1, score_threshold=-1)),
           (["-b", "1", "-r", "1", "-y", "-1"], dict(buffer_size=1, min_read_pair=1, score_threshold=-1)),
           (["-q", "0", "-s", "0", "-a", "-h"], dict(min_map_qual=0, min_len=0, cn_lib=1, print_af=1)),
           (["-f", "-m", "600", "-x", "2", "-c", "2"], dict(fisher=1, max_sd=600, seq_coverage_lim=2, cut_sd=2))]


@pytest.mark.parametrize("args,kw", OPTSETS)
def test_cli_text_equals_oracle_text_on_other_option_sets(args, kw):
    """option paths the golden files do not cover: the CLI's full stdout against the oracle's rendering"""
    run = load_chr21(make_opts(**kw)).run()
    assert filter_cmd_lines(run_cli(args)) == filter_cmd_lines(run.text)


def test_cli_bed_dump_matches_reference(tmp_path):
    """integration-test/breakdancer_test.py:117-131 (test_breakdancer_bed_dump)"""
    bed = str(tmp_path / "out.bed")
    out = run_cli(["-g", bed])
    assert filter_cmd_lines(out) == filter_cmd_lines(open(os.path.join(CWD, "expected_output")).read())
    assert open(bed).read() == open(os.path.join(CWD, "expected.bed")).read()


def test_cli_fastq_dump_matches_reference(tmp_path):
    """integration-test/breakdancer_test.py:133-146 (test_breakdancer_fastq_dump)"""
    prefix = str(tmp_path / "actual")
    out = run_cli(["-o", "21", "-d", prefix])
    assert filter_cmd_lines(out) == filter_cmd_lines(open(os.path.join(CWD, "expected_output")).read())
    for lib in ("H_IJ-NA19238-NA19238-extlibs", "H_IJ-NA19240-NA19240-extlibs"):
        for k in ("1", "2"):
            got = open("%s.%s.%s.fastq" % (prefix, lib, k)).read()
            exp = open(os.path.join(CWD, "expected.%s.%s.fastq" % (lib, k))).read()
            assert got == exp, (lib, k)


@pytest.mark.parametrize("args,fn", [(["-a", "-h", "-o", "21"], "expected_output.cn_per_lib.af"), ([], "expected_output")])
def test_cli_pass1_cache_write_and_restore(tmp_path, args, fn):
    """-C writes the pass-1 cache, -R re-runs from it alone: options, configuration and statistics of the cached run
    (io/ConfigLoader.cpp:18-44; common/Options.cpp:47-53 allows no other argument next to -R)"""
    cache = str(tmp_path / "pass1.cache")
    exp = filter_cmd_lines(open(os.path.join(CWD, fn)).read())
    assert filter_cmd_lines(run_cli(["-C", cache] + args)) == exp
    assert os.path.getsize(cache) > 100
    p = subprocess.run([EXE, "-R", cache], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert filter_cmd_lines(p.stdout.decode()) == exp
    p = subprocess.run([EXE, "-R", cache, "-o", "21"], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"When using -R, no other options are allowed" in p.stderr


def test_cli_restored_statistics_are_the_ones_used(tmp_path):
    """a cache whose counters were edited: the restored run must print and use THOSE statistics (the reference takes its
    BamSummary from the archive, whatever the BAMs hold)"""
    cache = str(tmp_path / "pass1.cache")
    base = run_cli(["-C", cache, "-o", "21"])
    lines = open(cache).read().split("\n")
    i = [k for k, l in enumerate(lines) if l.startswith("covered_ref_len ")][0]
    cov = int(lines[i].split()[1])
    lines[i] = "covered_ref_len %d" % (cov // 4)   # a quarter of the reference length: window, densities, lambda all change
    open(cache, "w").write("\n".join(lines))
    p = subprocess.run([EXE, "-R", cache], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    out = p.stdout.decode()
    assert ("reflen:%d" % (cov // 4)) in out and ("reflen:%d" % cov) in base
    assert filter_cmd_lines(out) != filter_cmd_lines(base)


@pytest.mark.parametrize("gpus", ["0,0", "0,0,0"])
@pytest.mark.parametrize("args,fn", [([], "expected_output"), (["-h"], "expected_output.af"), (["-a"], "expected_output.cn_per_lib"), (["-a", "-h"], "expected_output.cn_per_lib.af")])
def test_cli_sharded_over_ranks_reproduces_the_golden_output(gpus, args, fn, tmp_path):
    """BDX_GPUS: one whole-genome run with the chromosomes spread over several ranks (bdx_dist_*; here the ranks are
    threads that share the one GPU) must print what the single-GPU run prints -- the -g BED dump and the -d FASTQ dumps of
    the supporting reads included (integration-test/breakdancer_test.py:117-146)"""
    p = subprocess.run([EXE] + args + ["inv_del_bam_config"], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, BDX_GPUS=gpus, BDX_TIMING="1"))
    assert p.returncode == 0, p.stderr.decode()
    assert filter_cmd_lines(p.stdout.decode()) == filter_cmd_lines(open(os.path.join(CWD, fn)).read())
    # the two indexed BAMs of the reference's configuration: every rank decodes its chromosomes' ranges of BOTH files on its GPU and merges them there
    assert "on its own GPU" in p.stderr.decode() and "(2 files)" in p.stderr.decode(), p.stderr.decode()
    if args:
        return
    bed, prefix = str(tmp_path / "out.bed"), str(tmp_path / "actual")
    p = subprocess.run([EXE, "-g", bed, "-d", prefix, "inv_del_bam_config"], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, BDX_GPUS=gpus))
    assert p.returncode == 0, p.stderr.decode()
    assert filter_cmd_lines(p.stdout.decode()) == filter_cmd_lines(open(os.path.join(CWD, "expected_output")).read())
    assert open(bed).read() == open(os.path.join(CWD, "expected.bed")).read()
    for lib in ("H_IJ-NA19238-NA19238-extlibs", "H_IJ-NA19240-NA19240-extlibs"):
        for k in ("1", "2"):
            assert open("%s.%s.%s.fastq" % (prefix, lib, k)).read() == open(os.path.join(CWD, "expected.%s.%s.fastq" % (lib, k))).read(), (lib, k)


@pytest.mark.parametrize("env", [{"BDX_FOREGROUND": "1"}, {"BDX_CLEAN_EXIT": "1"}, {"BDX_DECODE": "host"}])
def test_cli_process_modes_print_the_same_table(env):
    """the default (GPU work in a child whose exit the command does not wait for), one process (BDX_FOREGROUND), one process walking
    its destructors (BDX_CLEAN_EXIT), the host reader: the reference's expected output every time, and the dumps complete when the
    command returns"""
    import tempfile
    exp = filter_cmd_lines(open(os.path.join(CWD, "expected_output")).read())
    with tempfile.TemporaryDirectory() as td:
        bed = os.path.join(td, "out.bed")
        p = subprocess.run([EXE, "-g", bed, "inv_del_bam_config"], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()
        assert filter_cmd_lines(p.stdout.decode()) == exp
        assert open(bed).read() == open(os.path.join(CWD, "expected.bed")).read()
        p = subprocess.run([EXE, "-g", bed, "inv_del_bam_config"], cwd=CWD, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0 and filter_cmd_lines(p.stdout.decode()) == exp
        assert open(bed).read() == open(os.path.join(CWD, "expected.bed")).read()   # (read the moment the command has returned)
