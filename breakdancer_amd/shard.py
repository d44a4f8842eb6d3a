"""Chromosome sharding across GPUs (one process per GPU, torch.distributed; "nccl" is RCCL over xGMI on ROCm).

The path shards by chromosome: regions never span tids (breakdancer/BreakDancer.cpp:216).  Two modes:

* `run_per_chromosome`  -- the reference's documented parallel mode, one `-o <chr>` run per chromosome
  (README:31,73; pass 1 is restricted to the chromosome as well, io/BamSummary.cpp:136).  Chromosomes are
  independent units: no data-path collective, rank 0 only gathers the SV rows.
* `ShardedRun`          -- one whole-genome run (what a single `breakdancer-max cfg` prints, incl. `-t`), with the
  chromosomes spread over ranks.  Needs three small exchanges: an all-reduce of the pass-1 counters (window, lambda and
  densities are global), an all-gather of per-chromosome totals (bases of the prefix counters / region ids / stream
  order), and ONE all-to-all of the join entries {name key, stream order, region, meta, |isize|} to owner(hash(key)) so
  that inter-chromosomal mates meet -- the only real exchange step of the path.  Regions and pair groups are then
  gathered to rank 0 for the (inherently sequential) walk.  All payloads are KBs..MBs: latency-bound, one hop each.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .api import BATCH_FIELDS, BdxError, BreakDancer

ENTRY_DTYPE = np.dtype([("key", "<u8"), ("order", "<u4"), ("region", "<i4"), ("meta", "<u4"), ("isize", "<i4")])


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


def route_entries(entries, world):
    """split a structured ENTRY_DTYPE array by owner rank -> list of uint8 chunks (stable order within a chunk)"""
    if world == 1:
        return [entries.view(np.uint8).reshape(-1)]
    own = owner_of(entries["key"], world)
    return [np.ascontiguousarray(entries[own == d]).view(np.uint8).reshape(-1) for d in range(world)]


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


class _Ctx(BreakDancer):
    """BreakDancer plus the staged entry points"""

    def stage_pass1(self):
        self._chk(self.lib.bdx_stage_pass1(self.h), "bdx_stage_pass1")
        nkeys = self.nlibs if self.opts.CN_lib else self.nbams
        cnt = np.zeros(self.nlibs * 12 + self.nbams, np.uint32)
        ref = np.zeros(self.nbams, np.uint64)
        tot = np.zeros(2 + nkeys, np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_get_pass1_local(self.h, p(cnt), p(ref), p(tot)), "bdx_get_pass1_local")
        return cnt, ref, tot

    def set_global(self, cnt, covered, window=-1):
        cnt = np.ascontiguousarray(cnt, np.uint32)
        self._chk(self.lib.bdx_set_pass1_global(self.h, cnt.ctypes.data_as(C.c_void_p), int(covered), int(window)), "bdx_set_pass1_global")

    def stage_compact(self, nn_base, pk_base):
        pk = np.ascontiguousarray(pk_base, np.uint32)
        q, nn = C.c_int32(), C.c_uint32()
        self._chk(self.lib.bdx_stage_compact(self.h, int(nn_base), pk.ctypes.data_as(C.c_void_p), C.byref(q), C.byref(nn)),
                  "bdx_stage_compact")
        return q.value, nn.value

    def stage_regions(self, has_next, next_qlen, next_nn):
        self._chk(self.lib.bdx_stage_regions(self.h, int(has_next), int(next_qlen), int(next_nn)), "bdx_stage_regions")
        nr, na, lm = C.c_uint32(), C.c_uint32(), C.c_int32()
        self._chk(self.lib.bdx_get_stage_regions(self.h, C.byref(nr), C.byref(na), C.byref(lm)), "bdx_get_stage_regions")
        return nr.value, na.value, lm.value

    def region_records(self, nr):
        nkeys = self.nlibs if self.opts.CN_lib else self.nbams
        recs = np.zeros(nr, L.REGION_REC_DTYPE)
        pk = np.zeros((nr, 2 * nkeys), np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_get_region_records(self.h, p(recs), p(pk), nr), "bdx_get_region_records")
        return recs, pk

    def compact(self, na):
        key, region = np.zeros(na, np.uint64), np.zeros(na, np.int32)
        meta, isize = np.zeros(na, np.uint32), np.zeros(na, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_get_compact(self.h, p(key), p(region), p(meta), p(isize), na), "bdx_get_compact")
        return key, region, meta, isize

    def join(self, ent):
        n = len(ent)
        out = np.zeros(n // 2 + 1, L.GROUP_DTYPE)
        ng, npairs = C.c_uint32(), C.c_uint32()
        cols = [np.ascontiguousarray(ent[k]) for k in ("key", "order", "region", "meta", "isize")]
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_join_entries(self.h, n, *[p(c) for c in cols], p(out), len(out), C.byref(ng), C.byref(npairs)),
                  "bdx_join_entries")
        return out[:ng.value], npairs.value

    def walk(self, recs, pk, groups, last_maxq, any_anom):
        recs = np.ascontiguousarray(recs, L.REGION_REC_DTYPE)
        pk = np.ascontiguousarray(pk, np.uint32)
        groups = np.ascontiguousarray(groups, L.GROUP_DTYPE)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_stage_walk(self.h, len(recs), p(recs), p(pk), len(groups), p(groups), int(last_maxq), int(any_anom)),
                  "bdx_stage_walk")


class ShardedRun:
    """Whole-genome-equivalent run with chromosomes spread over ranks (see module docstring)."""

    def __init__(self, opts, libs, nbams, max_read_window_size, comm=None, device=0):
        if opts.min_len < 0:
            raise BdxError("staged runs do not support a negative -s")
        self.opts, self.libs, self.nbams, self.w0 = opts, list(libs), nbams, max_read_window_size
        self.comm = comm or LocalComm()
        self.device = device
        self.ctx = {}
        self.util = _Ctx(opts, libs, nbams, 0, max_read_window_size, device)  # joins / walks; owns no reads

    def add_chromosome(self, tid, arrs):
        c = _Ctx(self.opts, self.libs, self.nbams, 0, self.w0, self.device)
        if len(arrs["tid"]):
            c.push_reads(arrs)
        self.ctx[int(tid)] = c

    def run(self):
        comm, util = self.comm, self.util
        nkeys = len(self.libs) if self.opts.CN_lib else self.nbams
        ncnt = len(self.libs) * 12 + self.nbams
        # (1) pass 1 per chromosome, (C1) all-reduce of the counters, (C2) all-gather of per-chromosome totals
        cnt_sum, ref_sum, totals = np.zeros(ncnt, np.uint64), np.zeros(self.nbams, np.uint64), {}
        for tid in sorted(self.ctx):
            cnt, ref, tot = self.ctx[tid].stage_pass1()
            cnt_sum += cnt
            ref_sum += ref
            totals[tid] = tot.astype(np.int64)
        util.stage_pass1()
        red = comm.allreduce_sum(np.concatenate([cnt_sum, ref_sum]))
        cnt_g, ref_g = red[:ncnt].astype(np.uint32), red[ncnt:]
        covered = covered_from(ref_g)
        all_tot = {}
        for d in comm.allgather_obj(totals):
            all_tot.update(d)
        bases = prefix_bases(all_tot)
        # (2) compaction per chromosome with the bases of the preceding chromosomes; the first anomalous read of a
        #     chromosome closes the last candidate region of the previous one, so that read is all-gathered too
        firsts = {}
        for tid in sorted(self.ctx):
            c = self.ctx[tid]
            c.set_global(cnt_g, covered)
            b = bases[tid]
            firsts[tid] = c.stage_compact(b[1], b[2:])
        all_first = {}
        for d in comm.allgather_obj(firsts):
            all_first.update(d)
        anom_tids = [t for t in sorted(all_tot) if all_tot[t][0] > 0]
        nxt = {t: anom_tids[i + 1] for i, t in enumerate(anom_tids[:-1])}
        # (3) regions per chromosome with the global window
        nreg, last_maxq = {}, {}
        for tid in sorted(self.ctx):
            c = self.ctx[tid]
            if tid in nxt:
                q, nn = all_first[nxt[tid]]
                nr, na, lm = c.stage_regions(1, q, nn)
            else:
                nr, na, lm = c.stage_regions(0, 0, 0)
            nreg[tid] = nr
            last_maxq[tid] = lm
        util.set_global(cnt_g, covered)
        all_nreg, all_lm = {}, {}
        for d, e in comm.allgather_obj((nreg, last_maxq)):
            all_nreg.update(d)
            all_lm.update(e)
        rbase = prefix_bases({t: [n] for t, n in all_nreg.items()})
        # (4) join entries -> owner(hash(key)): the one real exchange step (C3)
        ents, reg_out = [], {}
        for tid in sorted(self.ctx):
            c = self.ctx[tid]
            na = int(all_tot[tid][0])
            key, region, meta, isize = c.compact(na)
            m = region >= 0
            e = np.zeros(int(m.sum()), ENTRY_DTYPE)
            e["key"], e["meta"], e["isize"] = key[m], meta[m], isize[m]
            e["region"] = region[m] + int(rbase[tid][0])
            e["order"] = (np.arange(na, dtype=np.int64)[m] + int(bases[tid][0])).astype(np.uint32)
            ents.append(e)
            reg_out[tid] = c.region_records(nreg[tid])
        mine = np.concatenate(ents) if ents else np.zeros(0, ENTRY_DTYPE)
        recv = comm.alltoall_bytes(route_entries(mine, comm.world))
        got = np.concatenate([np.frombuffer(r.tobytes(), ENTRY_DTYPE) for r in recv]) if recv else np.zeros(0, ENTRY_DTYPE)
        groups, npairs = util.join(got)
        # (5) regions + groups to rank 0, walk there
        gathered = comm.gather_obj((reg_out, groups, npairs), root=0)
        if comm.rank != 0:
            return None
        regs, allg, pairs = {}, [], 0
        for ro, g, npr in gathered:
            regs.update(ro)
            allg.append(g)
            pairs += npr
        tids = sorted(regs)
        recs = np.concatenate([regs[t][0] for t in tids]) if tids else np.zeros(0, L.REGION_REC_DTYPE)
        pk = np.concatenate([regs[t][1] for t in tids]) if tids else np.zeros((0, 2 * nkeys), np.uint32)
        allg = np.concatenate(allg) if allg else np.zeros(0, L.GROUP_DTYPE)
        with_anom = [t for t in sorted(all_tot) if all_tot[t][0] > 0]
        lm = all_lm[with_anom[-1]] if with_anom else 0
        util.walk(recs, pk, allg, lm, bool(with_anom))
        self.n_pairs = pairs
        return util


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
