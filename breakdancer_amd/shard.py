"""Chromosome sharding across GPUs (one process per GPU, torch.distributed; "nccl" is RCCL over xGMI on ROCm).

The path shards by chromosome: regions never span tids (breakdancer/BreakDancer.cpp:216).  Two modes:

* `run_per_chromosome`  -- the reference's documented parallel mode, one `-o <chr>` run per chromosome
  (README:31,73; pass 1 is restricted to the chromosome as well, io/BamSummary.cpp:136).  Chromosomes are
  independent units: no data-path collective, rank 0 only gathers the SV rows.
* `ShardedRun`          -- one whole-genome run (what a single `breakdancer-max cfg` prints, incl. `-t`), with the
  chromosomes spread over ranks: a thin front end of the native path (dist.py, csrc/bdx_dist_impl.h), where the
  orchestration and every exchange live -- every rank runs ONE launch sequence over all of its chromosomes with genome-wide
  region ids, all-reduces carry the pass-1 statistics and the per-chromosome tables, ONE all-to-all over RCCL takes an
  inter-chromosomal (CTX) read to the rank that holds its mate's LATER chromosome, components of the region graph are walked
  where they live, and rank 0 walks the ones that span ranks and merges the ranks' tables by order key.

The numpy helpers below (owner_of, plan_chromosomes, covered_from, prefix_bases, ctx_destination, route_entries, taint_regions,
merge_by_key, TorchComm) restate the routing and bookkeeping rules of the native path so that they can be exercised without a
GPU (tests/test_distributed.py).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .api import BATCH_FIELDS, BdxError, BreakDancer

ENTRY_DTYPE = np.dtype([("key", "<u8"), ("order", "<u4"), ("region", "<i4"), ("meta", "<u4"), ("isize", "<i4"), ("tid", "<i4"), ("mtid", "<i4")])


def plan_chromosomes(read_counts, world):
    """Longest-processing-time packing of chromosomes (tid -> #reads) onto `world` ranks.  Returns list of tid lists."""
    bins = [[] for _ in range(world)]
    load = [0] * world
    for tid, n in sorted(read_counts.items(), key=lambda kv: (-kv[1], kv[0])):
        r = min(range(world), key=lambda i: (load[i], i))
        bins[r].append(tid)
        load[r] += n
    return [sorted(b) for b in bins]


def owner_of(keys, world):
    """rank that joins a name key: a mixed hash so that both mates of a pair (same key) meet on one rank"""
    k = keys.astype(np.uint64)
    k = (k ^ (k >> np.uint64(33))) * np.uint64(0xff51afd7ed558ccd)
    k = k ^ (k >> np.uint64(29))
    return (k % np.uint64(world)).astype(np.int64)


class LocalComm:
    """world of one (also what the tests use to exercise the staged path on a single GPU)"""
    rank, world = 0, 1

    def allreduce_sum(self, a):
        return a.copy()

    def allgather_obj(self, o):
        return [o]

    def gather_obj(self, o, root=0):
        return [o]

    def alltoall_bytes(self, chunks):
        return [chunks[0]]


class TorchComm:
    """torch.distributed: backend "nccl" (= RCCL, device tensors) on GPUs, "gloo" (CPU tensors) in the CPU tests"""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else torch.device("cpu")

    def _t(self, a):
        return self.torch.from_numpy(a).to(self.device)

    def allreduce_sum(self, a):
        """exact for uint64 payloads < 2^63 (counters)"""
        t = self._t(np.ascontiguousarray(a).astype(np.int64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().astype(a.dtype)

    def allgather_obj(self, o):
        out = [None] * self.world
        self.dist.all_gather_object(out, o)
        return out

    def gather_obj(self, o, root=0):
        out = [None] * self.world if self.rank == root else None
        self.dist.gather_object(o, out, dst=root)
        return out

    def alltoall_bytes(self, chunks):
        """chunks[d]: uint8 numpy array for rank d.  One all-to-all for the sizes, one for the payload."""
        torch, dist = self.torch, self.dist
        sizes = torch.tensor([len(c) for c in chunks], dtype=torch.int64, device=self.device)
        rsizes = torch.empty_like(sizes)
        dist.all_to_all_single(rsizes, sizes)
        rs = [int(x) for x in rsizes.cpu().tolist()]
        send = self._t(np.concatenate(chunks) if sum(len(c) for c in chunks) else np.zeros(0, np.uint8))
        recv = torch.empty(sum(rs), dtype=torch.uint8, device=self.device)
        dist.all_to_all_single(recv, send, output_split_sizes=rs, input_split_sizes=[len(c) for c in chunks])
        flat = recv.cpu().numpy()
        out, o = [], 0
        for n in rs:
            out.append(flat[o:o + n])
            o += n
        return out


def ctx_destination(tid, mtid, owner_of_tid, me):
    """where an inter-chromosomal read's join record travels (k7_exchange.hip, ctx_destination): the rank that holds its mate's
    chromosome if that chromosome comes LATER in the stream and is another rank's -- the pair is observed where its second mate is --,
    else -1 (the record stays: its mate joins it here, or the mate's record comes to it)"""
    tid, mtid = np.asarray(tid, np.int64), np.asarray(mtid, np.int64)
    own = np.asarray(owner_of_tid, np.int64)
    ok = (mtid > tid) & (mtid < len(own))
    dest = np.where(ok, own[np.clip(mtid, 0, len(own) - 1)], -1)
    return np.where((dest >= 0) & (dest != me), dest, -1)


def route_entries(entries, owner_of_tid, me, world):
    """split a rank's CTX entries (ENTRY_DTYPE incl. tid / mtid) by destination -> (list of uint8 chunks per rank, entries that stay)"""
    dest = ctx_destination(entries["tid"], entries["mtid"], owner_of_tid, me)
    chunks = [np.ascontiguousarray(entries[dest == d]).view(np.uint8).reshape(-1) for d in range(world)]
    return chunks, entries[dest < 0]


def taint_regions(groups_lo_hi, owner_of_region, n_regions):
    """regions that a (gate-passing) pair group connects ACROSS ranks: both of its regions; every rank learns them through an
    all-reduce and leaves their components to rank 0's walk (K6Arrays::taint)"""
    t = np.zeros(n_regions, np.uint8)
    own = np.asarray(owner_of_region)
    for lo, hi in groups_lo_hi:
        if own[lo] != own[hi]:
            t[lo] = 1
            t[hi] = 1
    return t


def order_key(T, start):
    """order key of an SV row (k6_score_kernel): the vertex it is placed at, before that vertex's own rows unless the traversal
    started there, then the start vertex"""
    T, start = np.asarray(T, np.uint64), np.asarray(start, np.uint64)
    return (T << np.uint64(34)) | ((T == start).astype(np.uint64) << np.uint64(33)) | (start << np.uint64(7))


def merge_by_key(keys_per_rank):
    """rank 0's merge of the ranks' tables (k9_merge_rank_kernel): every table is sorted by key; row i of rank q goes to i + the
    rows of the other tables in front of it (equal keys: the lower rank first).  Returns [(rank, row)] in final order."""
    n = sum(len(k) for k in keys_per_rank)
    out = [None] * n
    for q, kq in enumerate(keys_per_rank):
        pos = np.arange(len(kq))
        for p, kp in enumerate(keys_per_rank):
            if p != q and len(kp):
                pos = pos + np.searchsorted(kp, kq, side="right" if p < q else "left")
        for i, at in enumerate(pos.tolist()):
            assert out[at] is None
            out[at] = (q, i)
    return out


def covered_from(ref_len_per_bam):
    """BamSummary.cpp:123-126: uint32 covered, compared against each file's size_t sum"""
    covered = 0
    for r in ref_len_per_bam.tolist():
        if covered < int(r):
            covered = int(r) & 0xFFFFFFFF
    return covered


def prefix_bases(per_tid_totals):
    """{tid: totals vector} -> {tid: exclusive prefix over ascending tid}"""
    bases, acc = {}, None
    for tid in sorted(per_tid_totals):
        t = np.asarray(per_tid_totals[tid], dtype=np.int64)
        if acc is None:
            acc = np.zeros_like(t)
        bases[tid] = acc.copy()
        acc = acc + t
    return bases


class ShardedRun:
    """Whole-genome-equivalent run with the chromosomes spread over ranks, on the native path (dist.py -> bdx_dist_*).

    comm=None: the ranks are threads of this process -- `world` of them, all on `device` (what the single-GPU tests use to
    run the whole multi-rank orchestration, including the CTX all-to-all); chromosomes are dealt to the ranks by
    longest-processing-time packing.  comm=TorchComm: one process per GPU over RCCL; add_chromosome only for the tids this
    rank owns (plan_chromosomes)."""

    def __init__(self, opts, libs, nbams, max_read_window_size, comm=None, device=0, ntids=None, world=1, support=False):
        from . import dist as D
        self.opts, self.libs, self.nbams, self.w0 = opts, list(libs), nbams, max_read_window_size
        self.comm, self.device, self.ntids, self.world = comm, device, ntids, (comm.world if comm else world)
        self.D = D
        self.support = support
        self.pending = {}
        self.result_debug = {}   # bdx_set_debug switches for rank 0's result context (tests: "gather_walk")

    def add_chromosome(self, tid, arrs):
        self.pending[int(tid)] = arrs

    def run(self):
        D = self.D
        ntids = self.ntids if self.ntids is not None else (max(self.pending) + 1 if self.pending else 1)
        if self.comm is not None and self.comm.world > 1:
            ntids = max(self.comm.allgather_obj(ntids))
            d = D.DistRun.from_process_group(self.opts, self.libs, self.nbams, ntids, self.w0, self.device)
            if self.support:
                d.collect_support()
            for tid, arrs in sorted(self.pending.items()):
                c = d.chromosome(tid)
                if len(arrs["tid"]):
                    if arrs.get("name_check") is not None:
                        c.use_name_check()
                    c.push_reads(arrs)
            d.run()
            self.exchange = d.exchange()
            self._ranks = [d]
            return d.result()
        ranks = D.DistRun.threads(self.opts, self.libs, self.nbams, ntids, self.w0, [self.device] * self.world)
        if self.support:
            for r in ranks:
                r.collect_support()
        counts = {t: len(a["tid"]) for t, a in self.pending.items()}
        for r, tids in enumerate(plan_chromosomes(counts, self.world)):
            for tid in tids:
                c = ranks[r].chromosome(tid)
                if len(self.pending[tid]["tid"]):
                    if self.pending[tid].get("name_check") is not None:
                        c.use_name_check()
                    c.push_reads(self.pending[tid])
        for k, v in self.result_debug.items():
            for r in ranks:
                r.set_debug(k, v)
        res = D.run_threads(ranks)
        self.exchange = [r.exchange() for r in ranks]
        self._ranks = ranks  # (the result lives in rank 0's context: keep the ranks alive with it)
        return res


def run_per_chromosome(opts, libs, nbams, max_read_window_size, chromosomes, comm=None, device=0, runner=None):
    """`-o` semantics: every chromosome is an independent unit (no data-path collective).  `chromosomes` maps
    tid -> SoA arrays for the tids THIS rank owns; `runner(tid, arrs)` -> result object (default: the GPU path, returning
    the structured SV arrays).  Rank 0 gets {tid: result} for all chromosomes, in tid order."""
    comm = comm or LocalComm()

    def gpu_runner(tid, arrs):
        bd = BreakDancer(opts, libs, nbams, 0, max_read_window_size, device)
        if len(arrs["tid"]):
            bd.push_reads(arrs)
        bd.run()
        res = dict(summary=bd.summary(), svs=bd.svs(), counters=bd.counters())
        bd.close()
        return res

    runner = runner or gpu_runner
    mine = {int(t): runner(int(t), a) for t, a in sorted(chromosomes.items())}
    gathered = comm.gather_obj(mine, root=0)
    if comm.rank != 0:
        return None
    out = {}
    for d in gathered:
        out.update(d)
    return dict(sorted(out.items()))
