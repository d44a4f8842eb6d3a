#!/usr/bin/env python3
"""bench.py -- read-pairs/s of the anomalous read-pair clustering hot path on N MI355X (one rank per GPU).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script re-launches itself through torch.distributed.run (one
rank per GPU, rendezvous on 127.0.0.1) and fails loudly if fewer than N devices are visible; under an external
launcher WORLD_SIZE must equal --gpus.

Workload (BASELINE.json configs[1]): synthetic single chromosome, 50 Mbp, 30x, 2x100 bp, 1 library, ~1 % discordant
pairs -> 15 M records = 7.5 M read pairs per GPU, resident in HBM before the timed region.  A "step" is one full pass
of the hot path (bdx_run: classify -> compact -> region cut -> mate join -> pair groups -> component walk -> Poisson
scores -> final SV table in pinned host memory) over that batch, run the way a caller's FIRST run of an input goes:
nothing is enqueued ahead of the pass-1 read-back (bdx_set_enqueue_ahead mode 0) -- sizing the later stages from the
previous run of the same input (mode 2) or from a prior on the read count (mode 1) are reported as untimed extras.  With N > 1 every rank owns its own chromosome
(the path shards by chromosome, no data-path collective), so scaling is weak and `value` is the aggregate.

One JSON line on rank 0; see the task contract for the fields.
  roofline      dominant kernel (K1, the streaming classifier): algorithmic bytes = 28 B/read x reads per
                launch / the kernel's average duration measured with HIP events on the context's stream during the timed
                steps (bdx_get_timings; the kernel is bracketed on every 4th step, an event pair idles the GPU ~10 us).
  config.timings  the three timings of SURVEY.md 8(d) at the same 15 M records (N = 1 only): (i) HBM-resident = `value`,
                (ii) pinned host SoA -> SV table (adds PCIe), (iii) BAM -> SV table through bin/breakdancer-max.
  cpu_baseline  the reference-shaped CPU path from the same BAM on one host core: single-threaded BGZF inflate + record
                decode, twice (the reference decodes every file once per pass: io/BamSummary.cpp:129-150,
                breakdancer/BreakDancer.cpp:131-144), then the oracle (a sequential restatement of the reference's path).
                kind = "port": the reference itself needs Boost 1.54, which is absent, and cannot be built here.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_READ = 28          # 27 B SoA record read + 1 B class byte written
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec (MI355X_MICROARCH.md)
PATH_BYTES_PER_PAIR = 57.3        # whole path, per read pair, at 1 % discordant pairs
CHROM_LEN = 50_000_000
CFG_LINE = "readgroup:rg1\tplatform:illumina\tmap:%s\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--length", type=int, default=CHROM_LEN, help="chromosome length per GPU (default: configs[1])")
    ap.add_argument("--dry", action="store_true", help="rendezvous only (no GPU work): prints the world the ranks see")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip timings (ii) and (iii)")
    ap.add_argument("--no-exchange", "--no-genome", dest="no_exchange", action="store_true",
                    help="skip the sharded whole-genome runs over all ranks (config.genome)")
    ap.add_argument("--genome-fraction", type=float, default=None,
                    help="hg38 lengths x this for the genome leg (default N/8: 116 M records per GPU at every N -- the whole genome of configs[2] / [3] at N = 8 -- "
                         "plus, for N > 1, the fixed 1/8 spread over the N ranks)")
    ap.add_argument("--no-sharded-cli", action="store_true", help="skip config.timings.bam_to_table_sharded (BDX_GPUS on one indexed genome BAM)")
    ap.add_argument("--sharded-cli-fraction", type=float, default=None, help="hg38 lengths x this for that BAM (default N/64: 14.5 M records, ~2 GB of BAM, per rank)")
    ap.add_argument("--no-genome-bam", action="store_true", help="skip config.timings.bam_to_table_genome (one GPU's share of a 30x genome as ONE BAM through the CLI)")
    ap.add_argument("--genome-bam-fraction", type=float, default=1.0 / 8, help="hg38 lengths x this for that BAM (default 1/8: 116 M records, 15.9 GB)")
    ap.add_argument("--no-realistic-bam", action="store_true", help="skip config.timings.bam_to_table_genome_realistic (the same records as a level-6 BAM of reference-drawn bases)")
    ap.add_argument("--realistic-bam-fraction", type=float, default=1.0 / 8, help="hg38 lengths x this for the realistic BAM")
    ap.add_argument("--no-overlap", action="store_true", help="skip the three-contexts-in-flight measurement (config.overlapped_contexts)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc sub-run that measures roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-parallel", type=int, default=-1,
                    help="processes of the README's one-process-per-chromosome mode in cpu_baseline (default: min(usable CPUs, 24), "
                         "bounded by free memory; 0 = skip)")
    ap.add_argument("--overlap", action="store_true",
                    help="after the timed region, also measure the same steps with three contexts in flight (reported under "
                         "config.overlapped_contexts_untimed)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="contexts in flight per GPU (each with its own host thread and HIP stream, all reading the same resident "
                         "input); the default 1 is the plain sequence of steps the roofline figures refer to")
    ap.add_argument("--cpu-worker", nargs=2, metavar=("BAM", "PASSES"), help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# launch: --gpus N always means N ranks
# ---------------------------------------------------------------------------------------------------------------------
# Test hook (tests/test_gpu_bench.py): BDX_BENCH_TEST_SHARED_GPU=1 lets all ranks of an N > 1 run share device 0, with gloo
# as the process group (RCCL refuses two ranks on one device), so that the N > 1 control flow can be exercised on a
# one-GPU box.  Never set by the driver; the JSON line says so under config.test_hook.
SHARED_GPU_TEST = os.environ.get("BDX_BENCH_TEST_SHARED_GPU") == "1"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ensure_world(a):
    """Returns (rank, world, local_rank); re-launches through torch.distributed.run when --gpus asks for more ranks than
    the environment provides."""
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != a.gpus:
            sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
        return int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus == 1:
        return 0, 1, 0
    if not a.dry and not SHARED_GPU_TEST:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) are visible" % (a.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def init_group(a, world, local):
    import torch
    import torch.distributed as dist
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if SHARED_GPU_TEST or (a.dry and not torch.cuda.is_available()):
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return dist


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the reference-shaped path from BAM on host cores
# ---------------------------------------------------------------------------------------------------------------------
def cpu_worker(bam, passes):
    """one process of the CPU baseline: BAM -> (decode x passes) -> oracle -> SV table; prints one JSON line"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import OracleRun, make_opts
    t0 = time.perf_counter()
    cfg_path = os.environ.get("BDX_CPU_WORKER_CFG")   # (a configuration file beside the BAM: the genome slice's four read groups)
    run = OracleRun(open(cfg_path).read().replace(os.path.basename(bam), bam) if cfg_path else CFG_LINE % bam, make_opts())
    n, dec = run.load_bam(0, bam, passes=passes, set_targets=True)
    t1 = time.perf_counter()
    rc = run.L.bdo_run(run.h)
    t2 = time.perf_counter()
    assert rc == 0
    s = np.zeros(5, dtype=np.int64)
    run.L.bdo_summary(run.h, s.ctypes.data_as(__import__("ctypes").c_void_p))
    print(json.dumps({"reads": n, "decode_s": dec, "front_end_s": t1 - t0, "path_s": t2 - t1, "total_s": t2 - t0, "svs": int(s[4])}), flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """hardware threads capped by the cgroup CPU quota (a container may see 256 threads and own 16)"""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(bam, n_reads, want_parallel):
    me = [sys.executable, os.path.abspath(__file__), "--cpu-worker", bam]
    one = json.loads(subprocess.run(me + ["2"], check=True, stdout=subprocess.PIPE).stdout.decode().strip().splitlines()[-1])
    pairs = n_reads / 2
    nproc = os.cpu_count() or 1
    usable = usable_cpus()
    out = {"value": pairs / one["total_s"], "unit": "read-pairs/s", "cores": 1, "kind": "port",
           "cpu": cpu_model(), "host_threads": nproc, "usable_cpus": usable,
           "sample": "the full configs[1] chromosome as BAM (%d records, random bases and qualities): single-threaded BGZF inflate + record "
                     "decode x 2 passes (%.1f s) + the oracle's sequential path on the decoded records (%.2f s) = %.1f s on one core"
                     % (n_reads, one["decode_s"], one["path_s"], one["total_s"]),
           "compute_only": {"value": pairs / one["path_s"], "unit": "read-pairs/s",
                            "note": "oracle path alone on already-decoded records (what BENCH_r01 reported)"}}
    par = want_parallel
    if par < 0:
        par = min(usable, 24)   # one process per core this container may use (its CPU quota), at most one per chromosome
    par = int(min(par, max(1.0, mem_available_gb() * 0.5 / 3.0)))  # ~3 GB per process
    if par > 1:
        t0 = time.perf_counter()
        procs = [subprocess.Popen(me + ["2"], stdout=subprocess.PIPE) for _ in range(par)]
        outs = [p.communicate()[0] for p in procs]
        wall = time.perf_counter() - t0
        if all(p.returncode == 0 for p in procs):
            out["per_chromosome_mode"] = {
                "value": par * pairs / wall, "unit": "read-pairs/s", "cores": par,
                "note": "the README's parallel mode (one '-o <chr>' process per chromosome, README:31,73) as weak scaling: %d concurrent "
                        "single-threaded processes, each on its own copy of the configs[1] chromosome; wall %.1f s (process start to "
                        "SV table, slowest process)" % (par, wall)}
        del outs
    return out


# ---------------------------------------------------------------------------------------------------------------------
# timings (ii) and (iii) of SURVEY.md 8(d)
# ---------------------------------------------------------------------------------------------------------------------
def time_host_soa(bda, Options, LibraryConfig, LIB_C2, d, n, local, torch):
    """(ii) pinned host SoA -> SV table: bdx_push + bdx_run on a context whose read store is already reserved"""
    from breakdancer_amd.api import BATCH_FIELDS
    pinned, views = {}, {}
    for k, dt in list(BATCH_FIELDS) + [("name_check", np.uint64)]:
        arr = np.ascontiguousarray(d[k], dtype=dt)
        view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint64): np.int64}.get(arr.dtype)
        t = torch.from_numpy(arr.view(view) if view else arr).pin_memory()
        pinned[k] = t
        views[k] = t.numpy().view(dt)
    # bdx_reserve (outside the timed region, like the allocation of the resident store) also sizes the buffers of the later stages,
    # so a context's first push + run (`cold_context_seconds`) is within a few percent of the later ones; the same context then takes
    # the input again (bdx_reset_reads keeps the buffers) -- what a caller that streams one chromosome after the other through a
    # context sees.
    cold = None
    for fresh in range(2):  # (two fresh contexts, the better one: the process's very first push can pay one-time costs of the runtime)
        if fresh:
            bd.close()
        bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=local).use_name_check()
        bd.lib.bdx_reserve(bd.h, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bd.push_reads(views)
        bd.run()
        dt_ = time.perf_counter() - t0
        cold = dt_ if cold is None else min(cold, dt_)
    best = None
    for _ in range(4):
        bd.reset_reads()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bd.push_reads(views)
        bd.run()
        dt_ = time.perf_counter() - t0
        nsv = bd.summary()["n_svs_printed"]
        best = dt_ if best is None else min(best, dt_)
    bd.close()
    bytes_per_read = sum(np.dtype(dt).itemsize for _, dt in BATCH_FIELDS) + 8   # (+ the second name hash)
    lazy = 8 + sum(np.dtype(dt).itemsize for k, dt in BATCH_FIELDS if k in ("name_key", "qlen", "lib", "bam"))  # (one library, one file)
    return {"seconds": best, "value": (n / 2) / best, "unit": "read-pairs/s", "svs": nsv,
            "pcie_gb_per_s": (bytes_per_read - lazy) * n / best / 1e9, "cold_context_seconds": cold,
            "note": "bdx_push of %d pinned host records + bdx_run on a context that has run before, best of 4 (cold_context_seconds: the "
                    "first push + run of a context fresh from bdx_reserve, which sizes the later stages' buffers as well).  %d of the %d B/read "
                    "cross PCIe as copies (name key, second name hash and read length stay in the caller's pinned arrays; K2 fetches them for the ~1 %% anomalous "
                    "reads; the library and file index columns are not copied for a single library and file), so the rate is the "
                    "host-to-device bandwidth of the box (pcie_gb_per_s, run time included)"
                    % (n, bytes_per_read - lazy, bytes_per_read)}


def measure_k1_traffic(length):
    """HBM bytes of one launch of the dominant kernel, measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one
    counter per pass, as MI355X_MICROARCH.md prescribes) around a short child run of this script on the same workload.  gfx950
    correction from the same guide: bytes read = FETCH_SIZE x 2 (the counter counts 64-byte units of 128-byte requests), both
    counters in KB.  None if rocprofv3 is not there or a pass fails."""
    import csv
    import glob
    import shutil
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "rocprofv3 not found"
    vals = {}
    with tempfile.TemporaryDirectory(prefix="bdx_pmc_", dir="/tmp") as td:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            od = os.path.join(td, c)
            cmd = [rp, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", od, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end", "--no-genome", "--no-pmc", "--no-overlap", "--length", str(length)]
            try:
                p = subprocess.run(cmd, cwd=td, env=dict(os.environ, TMPDIR=td), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            except subprocess.TimeoutExpired:
                return None, "%s pass timed out" % c
            if p.returncode != 0:
                return None, "%s pass failed: %s" % (c, p.stderr.decode()[-200:])
            acc = []
            for f in glob.glob(os.path.join(od, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k1_classify_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                        acc.append(float(r["Counter_Value"]))
            if not acc:
                return None, "no %s rows for the classifier" % c
            vals[c] = float(np.mean(acc))
    return {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
            "hbm_bytes_per_launch": int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)}, None


def time_bam_cli(bam, cfg, n):
    """(iii) BAM -> SV table through the CLI (process start to exit, page cache warm), best of 3 per reader: the default (a one-BAM
    configuration is inflated and decoded on the GPU) and the host reader (BDX_DECODE=host) at the CPUs the container grants"""
    def run(env_extra, label, settle=0.5):
        env = dict(os.environ, BDX_TIMING="1", **env_extra)
        best = None
        for _ in range(3):
            if settle:
                time.sleep(settle)   # (untimed, default command only: the child of the run before is still handing its GPU context back to the driver)
            t0 = time.perf_counter()
            p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max"), cfg], cwd=os.path.dirname(cfg), env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stderr.decode()[-400:]}
            rows = sum(1 for line in p.stdout.splitlines() if line and not line.startswith(b"#"))
            tl = [x for x in p.stderr.decode().splitlines() if x.startswith("[bdx timing]")]
            if best is None or dt < best[0]:
                best = (dt, rows, tl)
        return {"seconds": best[0], "value": (n / 2) / best[0], "unit": "read-pairs/s", "sv_rows": best[1], "reader": label, "cli_breakdown": best[2][:-1]}
    dev = run({}, "device: BGZF inflate + record decode on the GPU (bdx_bamdec_*)")
    if "error" in dev:
        return dev
    cpus = usable_cpus()
    host = run({"BDX_DECODE": "host"}, "host: fast_inflate / zlib on %d decode threads (2 x the %d CPUs granted)" % (min(max(2 * cpus, 4), 64), cpus))
    inflated = None
    try:  # the file's inflated size (sum of the members' ISIZE words), for MB/s figures
        from breakdancer_amd import bamdec
        img = np.memmap(bam, dtype=np.uint8, mode="r")
        inflated = int(bamdec.scan_bgzf(img)["inflated_len"].astype(np.int64).sum())
    except Exception:  # noqa: BLE001
        pass
    time.sleep(0.5)
    fg = run({"BDX_FOREGROUND": "1"}, "device: BGZF inflate + record decode on the GPU (bdx_bamdec_*), one process, exit included", settle=0)
    out = dict(dev)
    if "seconds" in fg:
        # the block's own figure is the ONE-process run, the release of the GPU context included; the default command returns
        # earlier (the process that holds the GPU context is a child whose exit runs behind the command's return): kept beside it
        out = dict(fg)
        out["command_return"] = {"seconds": dev["seconds"], "value": dev["value"], "unit": "read-pairs/s", "cli_breakdown": dev.get("cli_breakdown"),
                                 "note": "the default command: returns when the table is written; the GPU context's release (~0.1 s) runs behind it in a child process"}
    out.update({"bam_bytes": os.path.getsize(bam), "inflated_bytes": inflated, "usable_cpus": cpus, "host_reader": host,
                "note": "bin/breakdancer-max <cfg> on the configs[1] chromosome as one BAM (%d records), from starting the command to its exit "
                        "with the whole table on stdout, best of 3; `seconds` / `value` are one process from start to exit (BDX_FOREGROUND=1), the "
                        "driver taking back ~6 GB of HBM and the pinned buffers included; `command_return` is the default command, whose GPU work "
                        "runs in a child that finishes its exit behind the command's return" % n})
    if inflated and "seconds" in host:
        out["host_reader"]["inflate_mb_per_s_per_cpu"] = inflated / 1e6 / host["seconds"] / cpus
    # bam2cfg on the same BAM: its CPU record source and `--device` (records decoded by the GPU decoder, statistics summed on
    # the GPU); the tool stops after ~3 x libraries x 10,000 records of good quality, so both are start-up costs more than anything
    try:
        b2c, texts = {}, []
        for label, extra in (("cpu", []), ("device", ["--device"])):
            best, text = None, None
            for _ in range(3):
                t0 = time.perf_counter()
                p = subprocess.run([os.path.join(ROOT, "bin", "bam2cfg"), *extra, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                dt = time.perf_counter() - t0
                if p.returncode != 0:
                    raise RuntimeError(p.stderr.decode()[-300:])
                if best is None or dt < best:
                    best, text = dt, p.stdout
            b2c[label] = {"seconds": best, "stdout_bytes": len(text)}
            texts.append(text)
        b2c["same_output"] = texts[0] == texts[1]
        out["bam2cfg"] = b2c
    except Exception as e:  # noqa: BLE001
        out["bam2cfg"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return out


def time_bam_cli_genome(fraction, n_gpu_visible, realistic=False):
    """BAM -> SV table at the size the metric is about: ONE indexed 24-chromosome, 4-library BAM of one GPU's share of a 30x genome
    (hg38 lengths x 1/8: 116 M records, 15.9 GB; smaller if memory does not hold it: stated), bin/breakdancer-max in one process from
    start to exit, page cache warm, best of 3.  Beside it, measured in this invocation on the same file: the ceilings of the three
    things the file has to get through -- page cache -> pinned -> HBM (bin/bdx-feed-probe), the inflate kernel alone on a slice's
    members in one launch -- and the reference-shaped CPU path on a stated slice (two chromosomes of the same genome as their own BAM).
    realistic=True: the same records as a BAM that compresses like one -- bases from a shared random reference (the ~30 reads that cover a
    locus share sequence), qualities in four bins, zlib level 6 (samtools' default; bamwrite.Reference) -- where the default file holds
    random bases and qualities at level 1 (ratio 1.57): the token mix decides the inflate kernel's speed."""
    from breakdancer_amd.bamwrite import write_genome_bam
    need_gb = 15.9 * fraction * 8
    avail = mem_available_gb()
    note = None
    while fraction > 1.0 / 512 and avail < 3.5 * need_gb + 8:   # (the file in tmpfs / page cache, the synthesis' columns, the pinned and mapped copies of the ceilings' slice)
        fraction /= 2
        need_gb /= 2
        note = "memory available %.0f GB: genome fraction reduced to %g" % (avail, fraction)
    td = tempfile.mkdtemp(prefix="bdx_genome_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp")
    try:
        t0 = time.perf_counter()
        bam, cfg, n = write_genome_bam(td, fraction, realistic=realistic, tag="realistic" if realistic else "genome")
        prep_s = time.perf_counter() - t0
        size = os.path.getsize(bam)
        best = None
        for _ in range(3):
            time.sleep(2.5)   # (untimed: the driver is still taking back the previous process's tens of GB of HBM)
            t0 = time.perf_counter()
            p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max"), cfg], cwd=td, env=dict(os.environ, BDX_TIMING="1", BDX_FOREGROUND="1"),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stderr.decode()[-400:]}
            rows = sum(1 for line in p.stdout.splitlines() if line and not line.startswith(b"#"))
            if best is None or dt < best[0]:
                best = (dt, rows, [x for x in p.stderr.decode().splitlines() if x.startswith("[bdx timing]")])
        out = {"seconds": best[0], "value": (n / 2) / best[0], "unit": "read-pairs/s", "file_gb_per_s": size / best[0] / 1e9, "sv_rows": best[1],
               "records": n, "bam_bytes": size, "genome_fraction": fraction, "synthesis_and_bam_write_seconds_untimed": prep_s, "cli_breakdown": best[2][:-1],
               "note": "bin/breakdancer-max <cfg> on ONE indexed 24-chromosome, 4-library BAM (hg38 x %g, 30x: one GPU's share of configs[2]), one process from "
                       "start to exit (BDX_FOREGROUND=1), best of 3, file in the page cache (tmpfs)%s%s" % (fraction, "; " + note if note else "",
                       "; bases from a shared random reference, binned qualities, zlib level 6" if realistic else "; random bases and qualities, zlib level 1")}
        import re
        for line in best[2]:
            m = re.search(r"steady state: ([0-9.]+) GB of BAM .* first inflate launch \(([0-9.]+) s after .* last record \(([0-9.]+) s\): ([0-9.]+) s, ([0-9.]+) GB/s", line)
            if m:
                out["steady_state_gb_s"] = float(m.group(5))
                out["steady_state"] = {"seconds": float(m.group(4)), "first_inflate_launch_s_after_decoder_setup": float(m.group(2)),
                                       "note": "file bytes / (last record decoded - first inflate launch): what a file of any size approaches"}
            m = re.search(r"total=([0-9.]+)s", line)
            if m:
                out["inside_the_process_seconds"] = float(m.group(1))
        # ceilings, on this file, now
        ceil = {}
        try:
            fp = subprocess.run([os.path.join(ROOT, "bin", "bdx-feed-probe"), bam, "6", str(min(usable_cpus(), 16)), "12", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            ceil["feed"] = json.loads(fp.stdout.decode().strip().splitlines()[-1]) if fp.returncode == 0 else {"error": fp.stderr.decode()[-200:]}
        except Exception as e:  # noqa: BLE001
            ceil["feed"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if not realistic:
            # the feed when EIGHT GPUs' feeders share this host (tools/feed_scaling.py, profiles/r06_feed_scaling.txt): 8 concurrent probes with the
            # usable CPUs divided between them; the leg that a node shares is page cache -> pinned (host memory bandwidth and cores)
            try:
                k = 8
                t = max(1, min(usable_cpus(), 16) // k)
                ps = [subprocess.Popen([os.path.join(ROOT, "bin", "bdx-feed-probe"), bam, "2", str(t), "12", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(k)]
                outs = [json.loads(p.communicate(timeout=180)[0].decode().strip().splitlines()[-1]) for p in ps]
                host = [o["page_cache_to_pinned_gb_s"] for o in outs]
                ceil["feed_with_8_feeders_on_this_host"] = {"instances": k, "threads_per_instance": t, "usable_cpus": usable_cpus(),
                                                            "page_cache_to_pinned_gb_s_per_instance": sum(host) / k, "aggregate_gb_s": sum(host),
                                                            "note": "8 concurrent bdx-feed-probe instances (mmap + memcpy into pinned staging): what each of eight GPUs' feeders "
                                                                    "gets of this host; the inflate kernel takes 44 GB/s of this file per GPU, 30 of a level-6 file"}
            except Exception as e:  # noqa: BLE001
                ceil["feed_with_8_feeders_on_this_host"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        try:
            from breakdancer_amd import bamdec
            img = np.memmap(bam, dtype=np.uint8, mode="r")
            members = bamdec.scan_bgzf(img)
            data = members[members["inflated_len"] > 0]
            out["inflated_bytes"] = int(data["inflated_len"].astype(np.int64).sum())
            k = min(len(data), 61440)   # eight rounds of the wave slots: ~2.5 GB of the file
            lo = int(data["member"][0])
            hi = int(data["payload"][k - 1]) + int(data["payload_len"][k - 1]) + 8
            sl = data[:k].copy()
            sl["payload"] -= np.uint64(lo)
            sl["member"] -= np.uint64(lo)
            piece = np.ascontiguousarray(img[lo:hi])
            ms = None
            for _ in range(2):
                _o, status, ms = bamdec.inflate_blocks(piece, sl)
            ulen = int(sl["inflated_len"].astype(np.int64).sum())
            ceil["inflate_kernel_alone"] = {"members": int(k), "launches": 1, "inflated_gb_s": ulen / ms / 1e6, "file_gb_s": (hi - lo) / ms / 1e6, "ok": not bool(status.any()),
                                            "note": "kz_inflate_kernel on the file's first %d members in ONE launch, bytes already in HBM (HIP events)" % k}
            del _o, piece
        except Exception as e:  # noqa: BLE001
            ceil["inflate_kernel_alone"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        out["ceilings"] = ceil
        lows = [v for v in (ceil.get("feed", {}).get("both_pipelined_gb_s"), ceil.get("inflate_kernel_alone", {}).get("file_gb_s")) if v]
        if lows and out.get("steady_state_gb_s"):
            out["steady_state_over_lowest_ceiling"] = out["steady_state_gb_s"] / min(lows)
        if out.get("inflated_bytes"):
            out["deflate_ratio"] = out["inflated_bytes"] / size
            for line in best[2]:
                m = re.search(r"inflate kernel in the pipeline: ([0-9.]+) ms in ([0-9]+) launches", line)
                if m:
                    out["inflate_in_situ"] = {"kernel_ms": float(m.group(1)), "launches": int(m.group(2)), "inflated_gb_s": out["inflated_bytes"] / float(m.group(1)) / 1e6,
                                              "file_gb_s": size / float(m.group(1)) / 1e6, "note": "kz_inflate_kernel inside the CLI's pipeline, HIP events around every launch"}
        if realistic:
            return out
        # the CPU path on a slice of the same genome: its two smallest chromosomes (chr21, chr22) as their own BAM, one core, two decode passes
        try:
            sbam, scfg, sn = write_genome_bam(td, fraction, only_tids=(20, 21), tag="slice", translocations=0)
            one = json.loads(subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", sbam, "2"], check=True, stdout=subprocess.PIPE,
                                            env=dict(os.environ, BDX_CPU_WORKER_CFG=scfg)).stdout.decode().strip().splitlines()[-1])
            out["cpu_port_on_a_slice"] = {"value": (sn / 2) / one["total_s"], "unit": "read-pairs/s", "cores": 1, "kind": "port", "seconds": one["total_s"], "records": sn,
                                          "sample": "chr21 + chr22 of the same genome (same seed, four libraries) as their own BAM: BGZF inflate + record decode x 2 passes "
                                                    "+ the oracle's sequential path, one core"}
            out["over_cpu_port"] = out["value"] / out["cpu_port_on_a_slice"]["value"]
        except Exception as e:  # noqa: BLE001
            out["cpu_port_on_a_slice"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        return out
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)


def time_bam_cli_sharded(td, n_gpus_visible, fraction=1.0 / 64):
    """BAM -> SV table with the chromosomes of ONE indexed BAM spread over ranks (BDX_GPUS): every rank decodes its chromosomes' BGZF
    ranges on its own GPU.  A 24-chromosome genome at 1/64 of hg38 (14.5 M records, ~2 GB of BAM), one library.  With one GPU
    visible the two ranks SHARE it (the path, not a scaling measurement); the same file on one GPU without BDX_GPUS beside it."""
    from breakdancer_amd.bamwrite import write_bam
    from breakdancer_amd.synth import make_genome
    lengths = [int(m * 1e6 * fraction) for m in HG38_MBP]
    tg = time.perf_counter()
    d = make_genome(lengths, coverage=30.0, seed=21, n_translocations=max(20, int(600 * fraction * 64)))
    n = len(d["tid"])
    bam = os.path.join(td, "genome.bam")
    write_bam(bam, d, ["chr%d" % (i + 1) for i in range(len(lengths))], seed=5, index=True)
    prep_s = time.perf_counter() - tg
    cfg = os.path.join(td, "gcfg")
    open(cfg, "w").write(CFG_LINE % "genome.bam")
    gpus = ",".join(str(i) for i in range(n_gpus_visible)) if n_gpus_visible > 1 and not SHARED_GPU_TEST else "0,0"

    def run(env_extra, cfg=cfg, n=n):
        env = dict(os.environ, BDX_TIMING="1", BDX_FOREGROUND="1", **env_extra)
        best = None
        for _ in range(3):
            time.sleep(1.0)   # (untimed: the driver is still reclaiming the previous process's HBM -- tens of GB of decoder rings in a sharded run)
            t0 = time.perf_counter()
            try:
                p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max"), cfg], cwd=td, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            except subprocess.TimeoutExpired:
                return {"error": "no result after 120 s (%s)" % env_extra}
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stderr.decode()[-400:]}
            rows = [line for line in p.stdout.splitlines() if line and not line.startswith(b"#")]
            if best is None or dt < best[0]:
                best = (dt, rows, p.stderr.decode())
        return {"seconds": best[0], "value": (n / 2) / best[0], "unit": "read-pairs/s", "sv_rows": len(best[1]), "_rows": best[1], "_err": best[2]}
    sh = run({"BDX_GPUS": gpus})
    one = run({})
    if "error" in sh or "error" in one:
        return {"error": sh.get("error") or one.get("error")}
    out = {"seconds": sh["seconds"], "value": sh["value"], "unit": "read-pairs/s", "sv_rows": sh["sv_rows"], "BDX_GPUS": gpus,
           "every_rank_decoded_on_its_gpu": "on its own GPU" in sh["_err"], "same_table_as_one_gpu": sh["_rows"] == one["_rows"],
           "one_gpu": {"seconds": one["seconds"], "value": one["value"], "unit": "read-pairs/s"},
           "records": n, "bam_bytes": os.path.getsize(bam), "synthesis_and_bam_write_seconds_untimed": prep_s,
           "note": "bin/breakdancer-max on ONE indexed 24-chromosome BAM, one process from start to exit (BDX_FOREGROUND=1), best of 3: the "
                   "chromosomes spread over the ranks of BDX_GPUS, each rank pulling its chromosomes' BGZF ranges through the .bai and decoding them "
                   "on its GPU" + ("; ONE GPU is visible here, the two ranks share it: this shows the path, not a speed-up" if n_gpus_visible <= 1 else "")}
    # the tumour / normal shape of configs[4]: TWO indexed BAMs (20x + 10x of the same genome, a library each) with -a -h -- per rank and chromosome
    # one decoder per file, BamMerger's order, one gather in HBM
    try:
        d2 = make_genome(lengths, coverage=(20.0, 10.0), seed=22, libs=((400.0, 30.0), (350.0, 40.0)), lib_bam=(0, 1), n_translocations=max(20, int(600 * fraction * 64)))
        names = ["chr%d" % (i + 1) for i in range(len(lengths))]
        sizes = 0
        for b, fn in enumerate(("tumour.bam", "normal.bam")):
            m = d2["bam"] == b
            write_bam(os.path.join(td, fn), {k: v[m] for k, v in d2.items()}, names, rg="rg%d" % (b + 1), seed=7 + b, index=True)
            sizes += os.path.getsize(os.path.join(td, fn))
        cfg2 = os.path.join(td, "gcfg2")
        with open(cfg2, "w") as f:
            for b, (fn, (mean, sd)) in enumerate(zip(("tumour.bam", "normal.bam"), ((400.0, 30.0), (350.0, 40.0)))):
                f.write("readgroup:rg%d\tplatform:illumina\tmap:%s\treadlen:100.00\tlib:lib%d\tlower:%.2f\tupper:%.2f\tmean:%.2f\tstd:%.2f\n"
                        % (b + 1, fn, b + 1, mean - 3 * sd, mean + 3 * sd, mean, sd))
        n2 = len(d2["tid"])

        def run2(env_extra):
            env = dict(os.environ, BDX_TIMING="1", BDX_FOREGROUND="1", **env_extra)
            best = None
            for _ in range(2):
                time.sleep(1.0)
                t0 = time.perf_counter()
                p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max"), "-a", "-h", cfg2], cwd=td, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
                dt = time.perf_counter() - t0
                if p.returncode != 0:
                    return {"error": p.stderr.decode()[-400:]}
                rows = [line for line in p.stdout.splitlines() if line and not line.startswith(b"#")]
                if best is None or dt < best[0]:
                    best = (dt, rows, p.stderr.decode())
            return {"seconds": best[0], "_rows": best[1], "_err": best[2]}
        sh2, one2 = run2({"BDX_GPUS": gpus}), run2({})
        if "error" in sh2 or "error" in one2:
            out["two_files"] = {"error": sh2.get("error") or one2.get("error")}
        else:
            out["two_files"] = {"seconds": sh2["seconds"], "value": (n2 / 2) / sh2["seconds"], "unit": "read-pairs/s", "sv_rows": len(sh2["_rows"]), "records": n2, "bam_bytes": sizes,
                                "every_rank_decoded_on_its_gpu": "on its own GPU" in sh2["_err"] and "(2 files)" in sh2["_err"], "same_table_as_one_gpu": sh2["_rows"] == one2["_rows"],
                                "one_gpu": {"seconds": one2["seconds"], "value": (n2 / 2) / one2["seconds"], "unit": "read-pairs/s"},
                                "note": "-a -h on two indexed BAMs (20x + 10x, a library each): every rank decodes its chromosomes' ranges of BOTH files on its GPU and merges them there"}
    except Exception as e:  # noqa: BLE001
        out["two_files"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# One whole-genome run over all ranks of this launch (bdx_dist_*): what sharding by chromosome costs and buys
# ---------------------------------------------------------------------------------------------------------------------
# hg38 primary assembly, chr1-22, X, Y (Mbp); the genome leg runs it at GENOME_FRACTION of its length: 1/8 = 387 Mbp,
# 116 M records at 30x -- one GPU's share of configs[2] when N = 1, a strong-scaling problem of fixed size for N > 1
HG38_MBP = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28, 114.36, 107.04,
            101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82, 156.04, 57.23]
GENOME_FRACTION = 1.0 / 8
LIBS4 = ((400.0, 30.0), (350.0, 40.0), (500.0, 50.0), (300.0, 25.0))


def genome_leg(rank, world, local, dist, out, fraction=None, translocations=5000, steps=0, warmup=0):
    """configs[2] and configs[3] as ONE sharded run each over the N ranks of this launch: hg38-shaped genome, 4 libraries, 30x,
    chromosomes dealt to the ranks by bdx_dist_plan (longest processing time first, on sequence length); every rank synthesises
    and loads its own chromosomes, then all call bdx_dist_run (csrc/bdx_dist_impl.h: one launch sequence per rank over all of its
    chromosomes, all-reduces of the statistics and per-chromosome tables, ONE all-to-all of the inter-chromosomal join records,
    components walked where they live, rank 0 walks what spans ranks and merges the ranks' tables).  Timed between barriers, max
    over ranks.  Sizes: `weak` = hg38 x N/8, i.e. 116 M records per GPU at every N and the whole genome of configs[2] / [3] at N = 8;
    `strong` (N > 1) = the fixed hg38 x 1/8 spread over the N ranks.  At N = 1 the same records also go through ONE context
    (bdx_run): `single_context`, the figure the sharded run is measured against.  Fills `out` on rank 0."""
    import torch
    from breakdancer_amd import dist as D
    from breakdancer_amd.api import LibraryConfig, Options
    from breakdancer_amd.synth import make_genome
    libs = [LibraryConfig(mean_insertsize=m, std_insertsize=sd, uppercutoff=m + 3 * sd, lowercutoff=m - 3 * sd, readlens=100.0, name="lib%d" % i)
            for i, (m, sd) in enumerate(LIBS4)]
    cpu_dev = torch.device("cpu") if (SHARED_GPU_TEST or dist is None) else torch.device("cuda", local)
    threads_backend = SHARED_GPU_TEST and world > 1   # (test hook: RCCL refuses two ranks on one device; rank 0 drives all ranks as threads)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def allsum(vals):
        t = torch.tensor(vals, dtype=torch.int64, device=cpu_dev)
        if dist is not None:
            dist.all_reduce(t)
        return [int(x) for x in t.tolist()]

    def allmax(val):
        t = torch.tensor([val], dtype=torch.float64, device=cpu_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def one_size(frac, scaling):
        lengths = [int(m * 1e6 * frac) for m in HG38_MBP]
        ntids = len(lengths)
        rank_of = D.plan(lengths, world)
        mine = set(t for t in range(ntids) if rank_of[t] == rank)
        tg = time.perf_counter()
        only = None if threads_backend else mine
        d = make_genome(lengths, coverage=30.0, seed=11, libs=LIBS4, lib_bam=(0, 0, 0, 0), n_translocations=translocations, only_tids=only) \
            if (not threads_backend or rank == 0) else None
        gen_s = time.perf_counter() - tg
        if threads_backend:
            per_rank = [0] * world
            if rank == 0:
                for t in range(ntids):
                    per_rank[rank_of[t]] += int((d["tid"] == t).sum())
            per_rank = allsum(per_rank)
        else:
            per_rank = [0] * world
            per_rank[rank] = len(d["tid"])
            per_rank = allsum(per_rank)
        total = sum(per_rank)
        legs = {}
        for label, opts in (("default_options", Options()), ("t_option", Options(transchr_rearrange=True))):
            if threads_backend and rank != 0:
                continue
            if threads_backend:
                ranks = D.DistRun.threads(opts, libs, 1, ntids, 200, [local] * world)
                owned = lambda r: [t for t in range(ntids) if rank_of[t] == r]   # noqa: E731
            elif dist is None:
                ranks = [D.DistRun.create(opts, libs, 1, ntids, 200, local, 0, 1, D.unique_id())]
                owned = lambda r: sorted(mine)   # noqa: E731
            else:
                ranks = [D.DistRun.from_process_group(opts, libs, 1, ntids, 200, local)]
                owned = lambda r: sorted(mine)   # noqa: E731
            tl = time.perf_counter()
            bounds = np.searchsorted(d["tid"], np.arange(ntids + 1))
            for r, run in enumerate(ranks):
                for t in owned(r):
                    lo, hi = int(bounds[t]), int(bounds[t + 1])
                    if hi > lo:
                        run.chromosome(t).push_reads({k: v[lo:hi] for k, v in d.items()})
                run.prepare()   # (outside the run, like bdx_reserve: buffers of the later stages sized for a first run, device code loaded)
            if not threads_backend:
                barrier()
            else:
                torch.cuda.synchronize()
            load_s = time.perf_counter() - tl
            import gc
            gc.collect()
            gc.disable()   # (a collection inside a 2 ms run would be most of it)
            times = []
            for it in range(2):   # the first run of the input, then the same handles once more
                t0 = time.perf_counter()
                if threads_backend:
                    res = D.run_threads(ranks)
                    times.append(time.perf_counter() - t0)
                else:
                    ranks[0].run(release=False)
                    barrier()
                    times.append(allmax(time.perf_counter() - t0))
                    res = ranks[0].result()
                if it == 0:
                    first_phases = ranks[0].phases()
                    first_ms_total = ranks[0].exchange()["ms_total"]
            timed = None
            if steps and label == "default_options":   # N > 1: the benchmark line's own K steps -- one step = one bdx_dist_run over the loaded genome
                for _ in range(warmup):
                    if threads_backend:
                        D.run_threads(ranks)
                    else:
                        ranks[0].run(release=False)
                if threads_backend:
                    torch.cuda.synchronize()
                else:
                    barrier()
                t0 = time.perf_counter()
                for _ in range(steps):
                    if threads_backend:
                        res = D.run_threads(ranks)
                    else:
                        ranks[0].run(release=False)
                if threads_backend:
                    torch.cuda.synchronize()
                    timed = time.perf_counter() - t0
                else:
                    barrier()
                    timed = allmax(time.perf_counter() - t0)
                    res = ranks[0].result()
            gc.enable()
            for run in ranks:
                run.release_inputs()   # (the Python side's references to the loaded arrays: not part of the run)
            ex = [r.exchange() for r in ranks]
            if threads_backend:
                sent, rank_ms = sum(e["ctx_records_sent"] for e in ex), [e["ms_total"] for e in ex]
            else:
                sent = allsum([ex[0]["ctx_records_sent"]])[0]
                rank_ms = [0] * world
                rank_ms[rank] = int(ex[0]["ms_total"] * 1000)
                rank_ms = [x / 1000.0 for x in allsum(rank_ms)]
            if rank == 0:
                sm = res.summary()
                ph = ranks[0].phases()
                # (as for single_context below: `seconds` is the run on handles that have run before; the first run is reported beside it --
                # 2.0-2.2 ms since the region table goes to the host through a kernel: the first device-to-host copy COMMAND of a process
                # sets up a copy-engine queue, 6 ms inside this process's first run until then)
                dt = times[1]
                r0 = ph.get("rank0_only_merge", 0.0) + ph.get("rank0_only_host_walk", 0.0) + ph.get("rank0_only_device_walk_of_gathered_groups", 0.0)
                legs[label] = {"seconds": dt, "value": total / 2 / dt, "unit": "read-pairs/s", "first_run_seconds": times[0], "second_run_seconds": times[1],
                               "hbm_roofline_frac_whole_path": total / 2 / dt / world * PATH_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS,
                               "svs_printed": sm["n_svs_printed"], "regions": sm["n_regions"], "sv_candidates_device_host": list(res.walk_split())[:2],
                               "ctx_records_exchanged": sent, "gathered_bytes_on_rank0": ex[0]["gathered_bytes"], "collectives_per_run": ranks[0].collectives(),
                               "bdx_dist_run_ms_per_rank": rank_ms, "rank0_bdx_dist_run_ms_first_run": first_ms_total, "rank0_phase_ms_first_run": first_phases, "rank0_phase_ms": ph,
                               "rank0_only": {"ms": r0, "share_of_run": r0 / (ex[0]["ms_total"] or 1.0),
                                              "note": "what only rank 0 does (second run): the merge of the ranks' tables, the device walk of the gathered "
                                                      "components that span ranks (its result context's K6) and the host walk of what that leaves"},
                               "load_and_prepare_seconds_untimed": load_s}
                if timed is not None:
                    legs[label]["timed_steps"] = {"steps": steps, "warmup": warmup, "seconds": timed, "ms_per_step": timed / steps * 1e3,
                                                  "value": total / 2 * steps / timed, "unit": "read-pairs/s",
                                                  "note": "K runs of bdx_dist_run on the loaded handles between two barriers, max over ranks: the N > 1 line's `value`"}
            for run in ranks:
                run.close()
        single = None
        if world == 1 and rank == 0:
            # the same records through ONE context: the reference point of the sharded run
            import breakdancer_amd as bda
            bd = bda.BreakDancer(Options(), libs, 1, ntids=ntids, max_read_window_size=200, device=local)
            bd.lib.bdx_reserve(bd.h, len(d["tid"]))
            bd.push_reads(d)
            torch.cuda.synchronize()
            ts = []
            for it in range(4):
                if it == 1:
                    bd.set_enqueue_ahead(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bd.run()
                ts.append(time.perf_counter() - t0)
            # (the same runs with the stages behind pass 1 enqueued ahead of its read-back, sized from the PRIOR on the read count -- mode 1: every run
            # behaves like a first run of its input, what the headline step of configs[1] times; mode 2 would size them from the previous run)
            bd.set_enqueue_ahead(1)
            tp = []
            for it in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bd.run()
                tp.append(time.perf_counter() - t0)
            bd.set_enqueue_ahead(0)
            single = {"first_run_seconds": ts[0], "seconds": min(ts[1:]), "value": total / 2 / min(ts[1:]), "unit": "read-pairs/s",
                      "hbm_roofline_frac_whole_path": total / 2 / min(ts[1:]) * PATH_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS,
                      "svs_printed": bd.summary()["n_svs_printed"],
                      "enqueued_ahead_from_the_prior": {"seconds": min(tp[1:]), "value": total / 2 / min(tp[1:]),
                                                        "hbm_roofline_frac_whole_path": total / 2 / min(tp[1:]) * PATH_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS,
                                                        "note": "bdx_set_enqueue_ahead(1), as the headline step: the later stages are launched while K1 runs, sized by the prior (no gain at this size: profiles/r06_genome_ab.txt)"},
                      "note": "bdx_run on ONE context holding all 24 chromosomes (default options; `seconds`: repeated runs WITHOUT enqueue-ahead -- the later stages are "
                              "sized after pass 1's read-back --, best of 3; first_run_seconds is NOT comparable: the records were classified tile by tile while they were pushed, so the context's first run finds K1's work done)"}
            # K1 on these records (every tile holds reads of the four libraries: its several-libraries tile body), by kernel-level HIP events
            bd.set_stage_timing(True)
            k1 = []
            for it in range(6):
                bd.run()
                k1.append(bd.timings()["classify"])
            bd.set_stage_timing(False)
            k1 = [x for x in k1 if x > 0]
            if k1:
                k1_ms = sum(k1) / len(k1)
                single["k1_classify"] = {"avg_kernel_ms": k1_ms, "achieved": total * 28 / (k1_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": total * 28 / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "note": "28 algorithmic bytes per read (SURVEY 8d), as in `roofline`; 4 libraries in one file"}
            bd.close()
        threads_entries = {}
        if world == 1 and rank == 0 and not SHARED_GPU_TEST:
            # The protocol of the 8-GPU configurations on the ONE GPU at hand: the same records dealt to 8 (and 2) ranks that run as threads of
            # this process and share the device -- every collective, the LPT packing of 24 chromosomes into 8 bins, 7 of 8 inter-chromosomal
            # pairs crossing ranks, rank 0's own part.  The ranks' kernels take turns on one GPU: a check of the path and of what is serial
            # in it, not a speed-up.
            for nr_threads in (8, 2):
                ent = {}
                plan_t = D.plan(lengths, nr_threads)
                per_t = [0] * nr_threads
                bounds = np.searchsorted(d["tid"], np.arange(ntids + 1))
                for t in range(ntids):
                    per_t[plan_t[t]] += int(bounds[t + 1] - bounds[t])
                for label, opts in (("default_options", Options()), ("t_option", Options(transchr_rearrange=True))):
                    try:
                        rk = D.DistRun.threads(opts, libs, 1, ntids, 200, [local] * nr_threads)
                        for r, run in enumerate(rk):
                            for t in range(ntids):
                                if plan_t[t] == r and bounds[t + 1] > bounds[t]:
                                    run.chromosome(t).push_reads({k: v[int(bounds[t]):int(bounds[t + 1])] for k, v in d.items()})
                            run.prepare()
                        torch.cuda.synchronize()
                        tt = []
                        for it in range(4):
                            t0 = time.perf_counter()
                            res = D.run_threads(rk)
                            tt.append(time.perf_counter() - t0)
                        ph = rk[0].phases()
                        exs = [r.exchange() for r in rk]
                        r0 = ph.get("rank0_only_merge", 0.0) + ph.get("rank0_only_host_walk", 0.0) + ph.get("rank0_only_device_walk_of_gathered_groups", 0.0)
                        coll_ms = {k: v for k, v in ph.items() if k.startswith(("allreduce", "alltoall", "gather"))}
                        sm = res.summary()
                        ent[label] = {"seconds": min(tt[1:]), "first_run_seconds": tt[0], "svs_printed": sm["n_svs_printed"], "regions": sm["n_regions"],
                                      "sv_candidates_device_host": list(res.walk_split())[:2],
                                      "ctx_records_travelled": sum(e["ctx_records_sent"] for e in exs), "gathered_bytes_on_rank0": exs[0]["gathered_bytes"],
                                      "collectives_per_run": rk[0].collectives(), "rank0_collective_ms": coll_ms, "rank0_phase_ms": ph,
                                      "bdx_dist_run_ms_per_rank": [round(e["ms_total"], 3) for e in exs],
                                      "rank0_only": {"ms": r0, "share_of_run": r0 / (exs[0]["ms_total"] or 1.0)}}
                        for run in rk:
                            run.release_inputs()
                            run.close()
                    except Exception as e:  # noqa: BLE001
                        ent[label] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                ent["reads_per_rank"] = per_t
                ent["lpt_imbalance_max_over_mean"] = max(per_t) / (sum(per_t) / nr_threads)
                ent["note"] = ("%d ranks as threads of one process sharing this GPU (bdx_dist_create_threads): the multi-GPU protocol end to end on one device; "
                               "seconds = best of 3 runs on handles that have run before" % nr_threads)
                threads_entries["ranks_as_threads_%d" % nr_threads] = ent
        if rank != 0:
            return None
        o = {"scaling": scaling, "genome_fraction": frac,
             "workload": "hg38-shaped genome at %.3g of its length (%d Mbp), 24 chromosomes, 4 libraries, 30x, %d planted translocations; "
                         "chromosomes -> ranks by bdx_dist_plan" % (frac, int(sum(lengths) / 1e6), translocations),
             "reads": total, "reads_per_rank": per_rank, "lpt_imbalance_max_over_mean": max(per_rank) / (total / world),
             "synthesis_seconds_untimed": gen_s, **legs}
        if single:
            o["single_context"] = single
        o.update(threads_entries)
        return o

    weak_frac = fraction if fraction is not None else world / 8.0
    sizes = [(weak_frac, "weak")]
    if fraction is None and world > 1:
        sizes.append((GENOME_FRACTION, "strong"))
    results = [one_size(f, sc) for f, sc in sizes]
    if rank == 0:
        first = results[0]
        out.update({"ranks": world,
                    "backend": ("ranks as threads of rank 0's process sharing one device (test hook)" if threads_backend else
                                "RCCL (ncclAllReduce, ncclAllToAllv, grouped ncclSend/ncclRecv on device buffers)"),
                    **first})
        if len(results) > 1:
            out["strong_scaling_fixed_size"] = results[1]


_LINE_FD = None


def emit(obj):
    """the benchmark line: ONE line of JSON on the stdout this process was started with"""
    sys.stdout.flush()
    os.write(_LINE_FD if _LINE_FD is not None else 1, (json.dumps(obj) + "\n").encode())


def main():
    global _LINE_FD
    a = parse()
    if a.cpu_worker:
        return cpu_worker(a.cpu_worker[0], int(a.cpu_worker[1]))
    rank, world, local = ensure_world(a)   # (--gpus N outside a launcher: starts the ranks and does not come back)
    # Libraries write to stdout behind Python's back (RCCL prints a version banner when a communicator is created): everything
    # that is not the line goes to stderr -- file descriptor 1 is pointed there, the line is written to a copy of the original
    _LINE_FD = os.dup(1)
    os.dup2(2, 1)
    if SHARED_GPU_TEST:
        local = 0
    import torch
    dist = init_group(a, world, local)
    # (a second, host-side group: ranks that wait while rank 0 runs the sharded CLI must not keep a barrier kernel spinning on their GPUs)
    host_group = None
    if dist is not None and dist.get_backend() != "gloo":
        try:
            host_group = dist.new_group(backend="gloo")
        except Exception:  # noqa: BLE001  (no gloo in this build: the waiting ranks use the default group's barrier)
            host_group = None
    if a.dry:
        ranks = [rank]
        if dist is not None:
            dev = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.zeros(world, dtype=torch.int64, device=dev)
            t[rank] = rank + 1
            dist.all_reduce(t)
            ranks = (t.cpu() - 1).tolist()
        if rank == 0:
            emit({"dry": True, "n_gpus": world, "ranks": ranks, "backend": dist.get_backend() if dist else None})
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
        sys.exit("bench.py: rank %d needs GPU %d but %d are visible (there is no CPU path)" %
                 (rank, local, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import breakdancer_amd as bda
    from breakdancer_amd.api import BATCH_FIELDS, LibraryConfig, Options
    from breakdancer_amd.synth import LIB_C2, make_chromosome

    d = make_chromosome(length=a.length, seed=1 + rank, name_base=rank << 40)
    n = len(d["tid"])
    # the second hash of the read name (bdx_use_name_check: mates are joined on both), as the readers compute it beside the key
    d["name_check"] = (d["name_key"].astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93)) ^ (d["name_key"].astype(np.uint64) >> np.uint64(29))
    # inputs resident in HBM before the timed region (torch owns the memory; libbdx adopts the pointers)
    tens = {}
    for k, dt in list(BATCH_FIELDS) + [("name_check", np.uint64)]:
        arr = np.ascontiguousarray(d[k], dtype=dt)
        view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint64): np.int64}.get(arr.dtype)
        tens[k] = torch.from_numpy(arr.view(view) if view else arr).to(dev)
    torch.cuda.synchronize()

    def new_ctx():
        x = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=local).use_name_check()
        x.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
        return x

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    bd = new_ctx()
    bd.set_enqueue_ahead(0)       # every timed step: pass 1, read-back, exact sizing of the later stages (nothing ahead)
    # --contexts > 1: further contexts on the same resident input; the timed steps are dealt out round-robin and run
    # concurrently (what a whole-genome caller does with one context per chromosome)
    ctxs = [bd]
    for _ in range(max(1, a.contexts) - 1):
        ctxs.append(new_ctx().set_enqueue_ahead(0))
    for _ in range(a.warmup):
        for x in ctxs:
            x.run()
    barrier()
    k1_ms = []
    t0 = time.perf_counter()
    if len(ctxs) == 1:
        for _ in range(a.steps):
            bd.run()
            k1_ms.append(bd.timings()["classify"])
    else:
        import threading

        def drive(i):
            for _ in range(i, a.steps, len(ctxs)):
                ctxs[i].run()
                k1_ms.append(ctxs[i].timings()["classify"])
        th = [threading.Thread(target=drive, args=(i,)) for i in range(len(ctxs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    barrier()
    dt = time.perf_counter() - t0
    summary = bd.summary()
    split = bd.walk_split() + (bd.cross_window_svs(),)

    # ---- untimed extras ---------------------------------------------------------------------------------------------
    # repeated runs of one context with enqueue-ahead (the later stages sized from the previous run's counts)
    rsteps = max(10, a.steps // 4)
    other_ms = {}
    for mode in (2, 1):
        bd.set_enqueue_ahead(mode)
        for _ in range(3):
            bd.run()
        torch.cuda.synchronize()
        tr = time.perf_counter()
        for _ in range(rsteps):
            bd.run()
        torch.cuda.synchronize()
        other_ms[mode] = (time.perf_counter() - tr) / rsteps * 1e3
    repeat_ms = other_ms[2]
    bd.set_enqueue_ahead(0)
    # per-stage device timings need HIP events between the stages, which idle the GPU: three extra steps
    bd.set_stage_timing(True)
    stage = {}
    for _ in range(3):
        bd.run()
        for k, v in bd.timings().items():
            stage[k] = stage.get(k, 0.0) + v
    bd.set_stage_timing(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if SHARED_GPU_TEST else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # The sharded whole-genome runs (config.genome): timed by their own barriers, outside the K timed steps above.  They run on a
    # helper thread under a watchdog: whatever happens to them -- RCCL with more than one rank runs here first -- the line is printed.
    exchange, exchange_hung = {}, False
    if not a.no_exchange and not a.pmc_child:
        import threading

        def guarded():
            try:
                genome_leg(rank, world, local, dist, exchange, fraction=a.genome_fraction, steps=a.steps if world > 1 else 0, warmup=a.warmup)
            except Exception as e:  # noqa: BLE001
                exchange["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        th = threading.Thread(target=guarded, daemon=True)
        th.start()
        th.join(360)
        if th.is_alive():
            exchange_hung = True
            exchange["error"] = "no result after 360 s"

    # N > 1: BAM -> table with the chromosomes of ONE indexed BAM spread over the N GPUs of this launch (rank 0 starts the one command,
    # BDX_GPUS=0..N-1; the other ranks wait on the host)
    sharded_cli = None
    hung_any = exchange_hung
    if world > 1:   # (every rank takes the same way from here: a genome leg that hung on ANY rank leaves its GPU in an unknown state)
        try:
            on_host = host_group is not None or dist.get_backend() == "gloo"
            flag = torch.tensor([1 if exchange_hung else 0], dtype=torch.int32, device=torch.device("cpu") if on_host else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=host_group)
            hung_any = bool(int(flag.item()))
        except Exception:  # noqa: BLE001
            hung_any = True
    if world > 1 and not a.no_end_to_end and not a.no_sharded_cli and not a.pmc_child and not hung_any:
        if rank == 0:
            try:
                with tempfile.TemporaryDirectory(prefix="bdx_bench_sh_") as td_sh:
                    sharded_cli = time_bam_cli_sharded(td_sh, world, a.sharded_cli_fraction if a.sharded_cli_fraction is not None else world / 64.0)
            except Exception as e:  # noqa: BLE001
                sharded_cli = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        dist.barrier(group=host_group) if host_group is not None else dist.barrier()

    # (behind the genome runs: two dozen contexts created, run and released leave the process in a state in which rank 0's part of a
    # sharded run -- its first, with all its allocations -- took three times as long)
    overlapped = None
    if world == 1 and len(ctxs) == 1 and not a.pmc_child and not a.no_overlap:   # (by default: three contexts in flight is how a whole-genome caller keeps one GPU busy)
        # the native driver (bdx_run_many): 24 contexts -- one per chromosome of a genome, here all on the same resident input --
        # of which three are in flight at a time
        from breakdancer_amd.api import run_many
        more = [bd] + [new_ctx().set_enqueue_ahead(0) for _ in range(23)]
        run_many(more, 3)
        run_many(more, 3)
        torch.cuda.synchronize()
        rounds = max(1, a.steps // 24)
        to = time.perf_counter()
        for _ in range(rounds):
            run_many(more, 3)
        torch.cuda.synchronize()
        do = time.perf_counter() - to
        overlapped = {"contexts_in_flight": 3, "contexts": len(more), "steps": rounds * len(more), "ms_per_step": do / (rounds * len(more)) * 1e3,
                      "value": (n // 2) * rounds * len(more) / do, "unit": "read-pairs/s", "driver": "bdx_run_many"}
        for x in more[1:]:
            x.close()
    if rank == 0:
        pairs = n // 2
        value = world * pairs * a.steps / dt
        k1_avg_ms = float(np.mean(k1_ms))
        achieved = ALGO_BYTES_PER_READ * n / (k1_avg_ms * 1e-3) / 1e9
        traffic, traffic_note, traffic_detail = None, None, None
        if world == 1 and not a.no_pmc and not a.pmc_child:
            traffic_detail, traffic_note = measure_k1_traffic(a.length)
            if traffic_detail:
                traffic = traffic_detail["hbm_bytes_per_launch"]
        n_ctxs_timed = len(ctxs)
        timings = {"hbm_resident": {"seconds": dt / a.steps, "value": value / world, "unit": "read-pairs/s",
                                    "note": "= `value` per GPU: one bdx_run on records already in HBM"}}
        cpu = None
        if sharded_cli is not None:
            timings["bam_to_table_sharded"] = sharded_cli
        if world == 1 and not (a.no_end_to_end and a.no_cpu_baseline):
            from breakdancer_amd.bamwrite import write_bam
            with tempfile.TemporaryDirectory(prefix="bdx_bench_") as td:
                if not a.no_end_to_end:
                    timings["pinned_host_soa"] = time_host_soa(bda, Options, LibraryConfig, LIB_C2, d, n, local, torch)
                bam = os.path.join(td, "syn.bam")
                tw = time.perf_counter()
                write_bam(bam, d, ["chrS"], seed=3)
                bam_write_s = time.perf_counter() - tw
                cfg = os.path.join(td, "cfg")
                open(cfg, "w").write(CFG_LINE % "syn.bam")
                if not a.no_end_to_end:
                    timings["bam_to_table"] = time_bam_cli(bam, cfg, n)
                    timings["bam_to_table"]["bam_write_s_untimed"] = bam_write_s
                    if not a.no_sharded_cli:
                        try:
                            timings["bam_to_table_sharded"] = time_bam_cli_sharded(td, torch.cuda.device_count(), a.sharded_cli_fraction if a.sharded_cli_fraction is not None else 1.0 / 64)
                        except Exception as e:  # noqa: BLE001
                            timings["bam_to_table_sharded"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                if not a.no_cpu_baseline:
                    cpu = cpu_baseline(bam, n, a.cpu_parallel)
            if not a.no_end_to_end and not a.no_genome_bam:
                n_ctxs_timed = len(ctxs)
                for x in ctxs:   # (the timed contexts' HBM is not needed any more: the CLI below reserves tens of GB)
                    x.close()
                ctxs = []
                try:
                    timings["bam_to_table_genome"] = time_bam_cli_genome(a.genome_bam_fraction, torch.cuda.device_count())
                except Exception as e:  # noqa: BLE001
                    timings["bam_to_table_genome"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                if not a.no_realistic_bam:
                    try:
                        timings["bam_to_table_genome_realistic"] = time_bam_cli_genome(a.realistic_bam_fraction, torch.cuda.device_count(), realistic=True)
                    except Exception as e:  # noqa: BLE001
                        timings["bam_to_table_genome_realistic"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        out = {
            "metric": "read-pairs/s, records resident in HBM -> scored SV table (SURVEY 8d timing i); end to end from BAM: config.timings.bam_to_table (vs_baseline is taken there)",
            "value": value, "unit": "read-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic single chromosome %d Mbp, 30x, 2x100 bp, 1 library, ~1%% discordant "
                                   "pairs; %d read pairs (%d records) per GPU, HBM-resident SoA; every timed step runs pass 1, reads its "
                                   "record back and sizes the later stages exactly (no enqueue-ahead)" % (a.length // 1000000, pairs, n),
                       "sharding": "one chromosome per GPU, no data-path collective", "contexts_in_flight": n_ctxs_timed,
                       "svs_per_gpu": summary["n_svs_printed"],
                       "timings": timings,
                       "repeat_run_enqueue_ahead": {"ms_per_step": repeat_ms, "value": pairs / (repeat_ms * 1e-3), "unit": "read-pairs/s",
                                                    "steps": rsteps, "note": "untimed extra: the same context re-running the same input, later stages "
                                                                             "sized from the previous run's count (BENCH_r01's figure)"},
                       "first_run_enqueue_ahead_on_prior": {"ms_per_step": other_ms[1], "value": pairs / (other_ms[1] * 1e-3), "unit": "read-pairs/s",
                                                            "steps": rsteps, "note": "untimed extra: later stages enqueued ahead sized by a prior of n/32 "
                                                                                     "anomalous reads (what a context's first run of an input does by default)"},
                       "stage_ms_profiled_steps": {k: v / 3 for k, v in stage.items()},
                       "sv_candidates": dict(zip(("assembled_on_device", "from_host_walk", "groups_to_host_walk", "device_placed_by_order_key"), split))},
            "roofline": {"bound": "hbm", "kernel": "k1_classify_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE sub-runs of this invocation (FETCH_SIZE x 2 on gfx950): %s" % json.dumps(traffic_detail)
                                            if traffic_detail else "not measured: %s" % (traffic_note or "skipped (--no-pmc, N > 1)")),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_READ * n,
                         "avg_kernel_ms": k1_avg_ms,
                         # the same kernel by the bytes it really moves (PMC, `traffic`): fewer than the algorithmic figure, which
                         # counts 2 B/read of read length the classifier never needs and, with one library and one file as here,
                         # 2 B/read of index columns it does not read either
                         "by_traffic": (None if not traffic else {"achieved": traffic / (k1_avg_ms * 1e-3) / 1e9,
                                                                  "frac": traffic / (k1_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}),
                         # SURVEY.md 8d's whole-path figure: 57.3 algorithmic bytes per read pair (28 B per read + 64 B per
                         # anomalous read at 1 % discordant) over the wall time of the step, not only the dominant kernel
                         "whole_path": {"algorithmic_bytes_per_read_pair": PATH_BYTES_PER_PAIR,
                                        "achieved": value / world * PATH_BYTES_PER_PAIR / 1e9,
                                        "frac": value / world * PATH_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS}},
        }
        ts = exchange.get("default_options", {}).get("timed_steps") if world > 1 else None
        if ts:
            # N > 1: the line is about the path that SHARDS -- configs[2], one whole-genome run over the N ranks (bdx_dist_run: chromosomes
            # dealt to the ranks, all-reduces of the statistics, the all-to-all of the inter-chromosomal join records), N/8 of the genome
            # so that every GPU holds what one of eight would (weak scaling); N replicas of configs[1] scale by construction and are kept as a note
            out["config"]["per_rank_replicas"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "unit": "read-pairs/s", "workload": out["config"]["workload"],
                                                  "note": "configs[1] replicated on every rank, no data-path collective (the N = 1 line's workload): linear by construction"}
            out["value"] = ts["value"]
            out["ms_per_step"] = ts["ms_per_step"]
            out["metric"] = "read-pairs/s, records resident in HBM -> scored SV table, ONE whole-genome run sharded by chromosome over the N GPUs (config.genome; per-GPU work fixed)"
            out["config"]["workload"] = ("configs[2]: " + exchange.get("workload", "") + "; %d records (%s per rank), HBM-resident; one timed step = one bdx_dist_run over all ranks"
                                         % (exchange.get("reads", 0), "/".join(str(x) for x in exchange.get("reads_per_rank", []))))
            out["config"]["sharding"] = "chromosomes -> ranks (longest first), RCCL all-reduces of the statistics and per-chromosome tables, one all-to-all of the inter-chromosomal join records"
            for k in ("roofline",):   # (the dominant kernel's figure was measured on the replicas' steps: it stays, and says so)
                out[k]["measured_on"] = "config.per_rank_replicas (configs[1] on every rank)"
            out["roofline"]["whole_path"] = {"algorithmic_bytes_per_read_pair": PATH_BYTES_PER_PAIR, "achieved": ts["value"] / world * PATH_BYTES_PER_PAIR / 1e9,
                                             "frac": ts["value"] / world * PATH_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS}
        # the quantity the metric's name ends with, at the top level: BAM -> SV table, one process, exit included
        g2 = timings.get("bam_to_table_genome", {})
        g1 = timings.get("bam_to_table", {})
        gs = timings.get("bam_to_table_sharded", {})
        if world == 1 and "seconds" in g2:
            out["end_to_end"] = {"seconds": g2["seconds"], "value": g2["value"], "unit": "read-pairs/s", "file_gb_per_s": g2.get("file_gb_per_s"),
                                 "workload": "one GPU's share of a 30x genome as ONE indexed 24-chromosome, 4-library BAM (%d records, %.1f GB): bin/breakdancer-max <cfg>, "
                                             "one process from start to exit, file in the page cache" % (g2.get("records", 0), g2.get("bam_bytes", 0) / 1e9),
                                 "configs1_as_one_bam": ({"seconds": g1["seconds"], "value": g1["value"]} if "seconds" in g1 else None),
                                 "source": "config.timings.bam_to_table_genome (configs[1] as one BAM: config.timings.bam_to_table, where vs_baseline is taken)"}
        elif world == 1 and "seconds" in g1:
            out["end_to_end"] = {"seconds": g1["seconds"], "value": g1["value"], "unit": "read-pairs/s", "workload": "configs[1] as one BAM through bin/breakdancer-max",
                                 "source": "config.timings.bam_to_table"}
        elif world > 1 and "seconds" in gs:
            out["end_to_end"] = {"seconds": gs["seconds"], "value": gs.get("value"), "unit": "read-pairs/s", "workload": "one indexed 24-chromosome BAM sharded over the ranks (BDX_GPUS)",
                                 "source": "config.timings.bam_to_table_sharded"}
        if overlapped:
            out["config"]["overlapped_contexts"] = overlapped
        if not a.no_exchange and not a.pmc_child:
            out["config"]["genome"] = exchange
        if SHARED_GPU_TEST:
            out["config"]["test_hook"] = "BDX_BENCH_TEST_SHARED_GPU: all ranks on device 0, gloo process group -- not a measurement"
        if cpu:
            out["cpu_baseline"] = cpu
            e2e = timings.get("bam_to_table", {})
            if "value" in e2e and cpu.get("value"):
                # like for like: BAM -> SV table on this box, GPU path (one process, exit included) over the reference-shaped CPU path on one core
                out["vs_baseline"] = e2e["value"] / cpu["value"]
                out["vs_baseline_is"] = "config.timings.bam_to_table.value / cpu_baseline.value (both from the same BAM to the SV table on this box; BASELINE.md holds no published number)"
                if e2e.get("host_reader", {}).get("value") and cpu.get("per_chromosome_mode", {}).get("value"):
                    out["config"]["timings"]["bam_to_table"]["over_cpu_per_chromosome_mode"] = e2e["value"] / cpu["per_chromosome_mode"]["value"]
            if cpu.get("compute_only", {}).get("value"):
                out["config"]["timings"]["hbm_resident"]["over_cpu_compute_only"] = (value / world) / cpu["compute_only"]["value"]
        emit(out)
    if exchange_hung or "error" in exchange:  # a rank stuck (or a peer lost) in a collective cannot be joined: leave as is
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        import threading
        threading.Timer(60, lambda: os._exit(0)).start()  # (peers that left the hard way would leave this barrier waiting)
        dist.barrier()
        dist.destroy_process_group()
        os._exit(0)


if __name__ == "__main__":
    main()
