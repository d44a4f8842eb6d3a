"""Chromosome-sharded runs over several GPUs: the Python mirror of include/bdx.h's bdx_dist_* entry points.

One whole-genome result (what a single `breakdancer-max cfg` run prints, -t included) with the chromosomes spread over
ranks, one rank per GPU.  All orchestration and every exchange live in libbdx (csrc/bdx_dist_impl.h): all-reduces of the
pass-1 statistics and per-chromosome totals, ONE all-to-all of the inter-chromosomal (ARP_CTX) join records over RCCL --
pairs with both mates on one chromosome never leave their GPU --, a gather of region tables and pair groups on rank 0.
Python only boots the communicator (rank 0's 128-byte id travels over whatever process group the launcher set up) and
feeds the chromosomes' contexts.

    one process per GPU (torchrun):   d = DistRun.from_process_group(opts, libs, nbams, ntids, w0, device=local_rank)
    ranks as threads of one process:  ranks = DistRun.threads(opts, libs, nbams, ntids, w0, devices=[0, 0, 0]); run_threads(ranks)
"""
import ctypes as C
import threading

import numpy as np

from . import _lib as L
from .api import BdxError, BreakDancer


def _lib():
    lib = L.load()
    if not getattr(lib, "_dist_bound", False):
        vp = C.c_void_p
        lib.bdx_dist_unique_id.argtypes = [vp]
        lib.bdx_dist_create.argtypes = [C.POINTER(vp), C.POINTER(L.bdx_opts), C.POINTER(L.bdx_lib), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, vp]
        lib.bdx_dist_create_threads.argtypes = [C.POINTER(vp), C.POINTER(L.bdx_opts), C.POINTER(L.bdx_lib), C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.POINTER(C.c_int), C.c_int]
        lib.bdx_dist_destroy.argtypes = [vp]
        lib.bdx_dist_destroy.restype = None
        lib.bdx_dist_last_error.argtypes = [vp]
        lib.bdx_dist_last_error.restype = C.c_char_p
        lib.bdx_dist_rank.argtypes = [vp]
        lib.bdx_dist_world.argtypes = [vp]
        lib.bdx_dist_chromosome.argtypes = [vp, C.c_int]
        lib.bdx_dist_chromosome.restype = vp
        lib.bdx_dist_run.argtypes = [vp]
        lib.bdx_dist_result.argtypes = [vp]
        lib.bdx_dist_result.restype = vp
        lib.bdx_dist_set_collect_support.argtypes = [vp, C.c_int]
        lib.bdx_dist_get_exchange.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.bdx_dist_get_collectives.argtypes = [vp, vp, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        lib.bdx_dist_owner.argtypes = [C.c_uint64, C.c_int]
        lib.bdx_dist_prepare.argtypes = [vp]
        lib.bdx_dist_phase_name.argtypes = [C.c_int]
        lib.bdx_dist_phase_name.restype = C.c_char_p
        lib.bdx_dist_plan.argtypes = [vp, C.c_int, C.c_int, vp]
        lib._dist_bound = True
    return lib


def unique_id():
    """rank 0: the communicator id (ncclGetUniqueId), 128 bytes"""
    buf = C.create_string_buffer(128)
    rc = _lib().bdx_dist_unique_id(buf)
    if rc != 0:
        raise BdxError("bdx_dist_unique_id: %s" % _lib().bdx_strerror(rc).decode())
    return buf.raw


def owner(key, world):
    """rank that joins a name key (the routing rule of the all-to-all)"""
    return _lib().bdx_dist_owner(int(key), int(world))


def plan(weights, world):
    """chromosomes -> ranks, longest-processing-time packing on `weights` (reads or sequence lengths); list of ranks per tid"""
    w = np.ascontiguousarray(weights, np.uint64)
    out = np.zeros(len(w), np.int32)
    rc = _lib().bdx_dist_plan(w.ctypes.data_as(C.c_void_p), len(w), int(world), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise BdxError("bdx_dist_plan: %s" % _lib().bdx_strerror(rc).decode())
    return out.tolist()


def _c_args(opts, libs):
    co = opts.to_c()
    arr = (L.bdx_lib * len(libs))()
    for i, l in enumerate(libs):
        arr[i].mean_insertsize, arr[i].std_insertsize = l.mean_insertsize, l.std_insertsize
        arr[i].uppercutoff, arr[i].lowercutoff, arr[i].readlens = l.uppercutoff, l.lowercutoff, l.readlens
        arr[i].min_mapping_quality, arr[i].bam_index = l.min_mapping_quality, l.bam_file_index
    return co, arr


class DistRun:
    """one rank of a chromosome-sharded run"""

    def __init__(self, handle, opts, libs, nbams):
        self.lib = _lib()
        self.h = C.c_void_p(handle)
        self.opts, self.libs, self.nbams = opts, list(libs), nbams
        self.rank = self.lib.bdx_dist_rank(self.h)
        self.world = self.lib.bdx_dist_world(self.h)
        self._chrom = {}

    @classmethod
    def create(cls, opts, libs, nbams, ntids, max_read_window_size, device, rank, world, uid):
        """one process per GPU: joins the RCCL communicator identified by `uid` (collective)"""
        lib = _lib()
        co, arr = _c_args(opts, libs)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), 128)
        rc = lib.bdx_dist_create(C.byref(h), C.byref(co), arr, len(libs), nbams, ntids, max_read_window_size, device, rank, world, buf)
        if rc != 0:
            raise BdxError("bdx_dist_create: %s" % lib.bdx_strerror(rc).decode())
        return cls(h.value, opts, libs, nbams)

    @classmethod
    def from_process_group(cls, opts, libs, nbams, ntids, max_read_window_size, device):
        """under torch.distributed (torchrun): rank 0's communicator id is broadcast over the existing process group"""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.create(opts, libs, nbams, ntids, max_read_window_size, device, rank, world, box[0])

    @classmethod
    def threads(cls, opts, libs, nbams, ntids, max_read_window_size, devices):
        """the ranks as threads of this process, rank r on devices[r] (a device may repeat); returns the list of ranks"""
        lib = _lib()
        co, arr = _c_args(opts, libs)
        world = len(devices)
        hs = (C.c_void_p * world)()
        dv = (C.c_int * world)(*devices)
        rc = lib.bdx_dist_create_threads(hs, C.byref(co), arr, len(libs), nbams, ntids, max_read_window_size, dv, world)
        if rc != 0:
            raise BdxError("bdx_dist_create_threads: %s" % lib.bdx_strerror(rc).decode())
        return [cls(hs[r], opts, libs, nbams) for r in range(world)]

    def _chk(self, rc, what):
        if rc != 0:
            raise BdxError("%s: %s (%s)" % (what, self.lib.bdx_strerror(rc).decode(), self.lib.bdx_dist_last_error(self.h).decode()))

    def chromosome(self, tid):
        """the context that takes chromosome `tid`'s records (ONE per rank: a rank's chromosomes are fed in ascending order):
        push_reads / stream_reads it, do not run it"""
        h = self.lib.bdx_dist_chromosome(self.h, int(tid))
        if not h:
            raise BdxError("bdx_dist_chromosome(%d) failed: %s" % (tid, self.lib.bdx_dist_last_error(self.h).decode()))
        if h not in self._chrom:
            self._chrom[h] = BreakDancer.borrow(h, self.opts, self.libs, self.nbams)
        return self._chrom[h]

    def collect_support(self, on=True):
        """the result also holds the supporting reads of every SV (every rank alike, before run)"""
        self._chk(self.lib.bdx_dist_set_collect_support(self.h, 1 if on else 0), "bdx_dist_set_collect_support")
        return self

    def run(self, release=True):
        self._chk(self.lib.bdx_dist_run(self.h), "bdx_dist_run")
        if release:   # (the pushed arrays were kept alive for the asynchronous copies; freeing gigabytes of them takes tens of milliseconds)
            self.release_inputs()
        return self

    def release_inputs(self):
        for c in self._chrom.values():
            c._keep.clear()

    def result(self):
        """rank 0: a BreakDancer whose getters return the whole-genome result (owned by this DistRun); None elsewhere"""
        h = self.lib.bdx_dist_result(self.h)
        return BreakDancer.borrow(h, self.opts, self.libs, self.nbams) if h else None

    def exchange(self):
        sent, recv, gathered = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ms_total, ms_x = C.c_float(), C.c_float()
        self._chk(self.lib.bdx_dist_get_exchange(self.h, C.byref(sent), C.byref(recv), C.byref(gathered), C.byref(ms_total), C.byref(ms_x)),
                  "bdx_dist_get_exchange")
        return dict(ctx_records_sent=sent.value, ctx_records_received=recv.value, gathered_bytes=gathered.value, ms_total=ms_total.value,
                    ms_exchange=ms_x.value)

    def set_debug(self, name, value=1):
        """a test / measurement switch for this rank's contexts, the result context included (bdx_dist_set_debug)"""
        self.lib.bdx_dist_set_debug.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self._chk(self.lib.bdx_dist_set_debug(self.h, name.encode(), int(value)), "bdx_dist_set_debug")
        return self

    def collectives(self):
        """collectives this rank entered in the last run, and what carried them"""
        out = (C.c_uint32 * 3)()
        name, ver = C.c_char_p(), C.c_int()
        self._chk(self.lib.bdx_dist_get_collectives(self.h, out, C.byref(name), C.byref(ver)), "bdx_dist_get_collectives")
        return dict(allreduce=int(out[0]), alltoall=int(out[1]), gather=int(out[2]), backend=(name.value or b"").decode(), rccl_version=int(ver.value))

    N_PHASES = 18

    def phase_ms(self):
        out = (C.c_float * self.N_PHASES)()
        self.lib.bdx_dist_get_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        self._chk(self.lib.bdx_dist_get_phase_ms(self.h, out, self.N_PHASES), "bdx_dist_get_phase_ms")
        return [float(x) for x in out]

    def phase_names(self):
        return [self.lib.bdx_dist_phase_name(i).decode() for i in range(self.N_PHASES)]

    def phases(self):
        """{phase name: ms} of the last run on this rank"""
        return {n: round(v, 3) for n, v in zip(self.phase_names(), self.phase_ms()) if n}

    def prepare(self):
        """after loading, outside the run: size the later stages' buffers for a first run (bdx_reserve's job for one context)"""
        self._chk(self.lib.bdx_dist_prepare(self.h), "bdx_dist_prepare")
        return self

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.bdx_dist_destroy(self.h)
            self.h = C.c_void_p()
            self._chrom = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_threads(ranks):
    """drive the ranks of DistRun.threads: one thread each (bdx_dist_run is collective and releases the GIL)"""
    errs = [None] * len(ranks)

    def go(i):
        try:
            ranks[i].run()
        except Exception as e:  # noqa: BLE001
            errs[i] = e
    th = [threading.Thread(target=go, args=(i,)) for i in range(len(ranks))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return ranks[0].result()
