"""Python mirror of the reference's interface for the clustering path.

Names follow the reference: `Options` (common/Options.hpp:24-45, same field names), `LibraryConfig`
(io/LibraryConfig.hpp:11-24) and `BreakDancer` (breakdancer/BreakDancer.hpp:30-99: construct, feed reads, run).
Errors surface as `BdxError` the way the reference's exceptions surface as "ERROR: ..." in main()."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L

FLAG_NAMES = ["NA", "ARP_FF", "ARP_LARGE_INSERT", "ARP_SMALL_INSERT", "ARP_RF", "ARP_RR", "NORMAL_FR", "NORMAL_RF",
              "ARP_CTX", "MATE_UNMAPPED", "UNMAPPED"]


class BdxError(RuntimeError):
    pass


@dataclass
class Options:
    """common/Options.cpp:27-41 defaults"""
    min_len: int = 7
    cut_sd: int = 3
    max_sd: int = 1000000000
    min_map_qual: int = 35
    min_read_pair: int = 2
    seq_coverage_lim: int = 1000
    buffer_size: int = 100
    transchr_rearrange: bool = False
    fisher: bool = False
    Illumina_long_insert: bool = False
    CN_lib: bool = False
    print_AF: bool = False
    score_threshold: int = 30
    chr: str = ""

    def to_c(self):
        o = L.bdx_opts()
        o.min_len, o.cut_sd, o.max_sd, o.min_map_qual = self.min_len, self.cut_sd, self.max_sd, self.min_map_qual
        o.min_read_pair, o.seq_coverage_lim, o.buffer_size = self.min_read_pair, self.seq_coverage_lim, self.buffer_size
        o.transchr_rearrange, o.fisher = int(self.transchr_rearrange), int(self.fisher)
        o.illumina_long_insert, o.cn_lib, o.print_af = int(self.Illumina_long_insert), int(self.CN_lib), int(self.print_AF)
        o.score_threshold = self.score_threshold
        o.chr_restricted = 1 if self.chr else 0
        return o


@dataclass
class LibraryConfig:
    mean_insertsize: float
    std_insertsize: float
    uppercutoff: float
    lowercutoff: float
    readlens: float
    min_mapping_quality: int = -1
    bam_file_index: int = 0
    name: str = ""


BATCH_FIELDS = (("tid", np.int32), ("pos", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32),
                ("flag", np.uint16), ("qlen", np.uint16), ("mapq", np.uint8), ("lib", np.uint8), ("bam", np.uint8),
                ("name_key", np.uint64))


def make_batch(arrs):
    """dict of numpy arrays -> (bdx_batch, keepalive list).  Accepts 'bdqual' for mapq and 'name_id' for name_key."""
    b = L.bdx_batch()
    keep = []
    n = None
    for k, dt in BATCH_FIELDS:
        src = arrs.get(k)
        if src is None and k == "mapq":
            src = arrs.get("bdqual")
        if src is None and k == "name_key":
            src = arrs.get("name_id")
        if src is None:
            raise KeyError(k)
        a = np.ascontiguousarray(src, dtype=dt)
        if n is None:
            n = len(a)
        elif len(a) != n:
            raise ValueError("ragged batch: %s has %d rows, expected %d" % (k, len(a), n))
        keep.append(a)
        setattr(b, k, a.ctypes.data_as(C.c_void_p))
    if arrs.get("name_check") is not None:  # the second name hash (looked at after use_name_check() only)
        a = np.ascontiguousarray(arrs["name_check"], dtype=np.uint64)
        if len(a) != (n or 0):
            raise ValueError("ragged batch: name_check has %d rows, expected %d" % (len(a), n or 0))
        keep.append(a)
        b.name_check = a.ctypes.data_as(C.c_void_p)
    b.n = n or 0
    return b, keep


def run_many(contexts, in_flight=3):
    """bdx_run_many: every context of the list runs, `in_flight` of them at a time (the native driver for one context per chromosome)"""
    lib = L.load()
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    rc = lib.bdx_run_many(arr, len(contexts), in_flight)
    if rc != 0:
        msgs = [lib.bdx_last_error(c.h).decode() for c in contexts]
        raise BdxError("bdx_run_many: %s (%s)" % (lib.bdx_strerror(rc).decode(), "; ".join(m for m in msgs if m)))
    for c in contexts:
        c._keep.clear()
    return contexts


class BreakDancer:
    """One clustering context on one GPU (reference: BreakDancer::BreakDancer / run, BreakDancer.cpp:87-144)."""

    def __init__(self, opts, libs, nbams, ntids=0, max_read_window_size=100000000, device=0):
        self.lib = L.load()
        self.opts = opts
        self.libs = list(libs)
        self.nlibs, self.nbams = len(self.libs), nbams
        co = opts.to_c()
        arr = (L.bdx_lib * self.nlibs)()
        for i, l in enumerate(self.libs):
            arr[i].mean_insertsize, arr[i].std_insertsize = l.mean_insertsize, l.std_insertsize
            arr[i].uppercutoff, arr[i].lowercutoff, arr[i].readlens = l.uppercutoff, l.lowercutoff, l.readlens
            arr[i].min_mapping_quality, arr[i].bam_index = l.min_mapping_quality, l.bam_file_index
        self.h = C.c_void_p()
        rc = self.lib.bdx_create(C.byref(self.h), C.byref(co), arr, self.nlibs, nbams, ntids, max_read_window_size, device)
        if rc != 0:
            self.h = C.c_void_p()
            raise BdxError("bdx_create: %s" % self.lib.bdx_strerror(rc).decode())
        self._keep = []

    @classmethod
    def borrow(cls, handle, opts, libs, nbams):
        """wrap a context that somebody else owns (a chromosome or the result of a sharded run, dist.py): never destroyed here"""
        self = cls.__new__(cls)
        self.lib = L.load()
        self.opts, self.libs = opts, list(libs)
        self.nlibs, self.nbams = len(self.libs), nbams
        self.h = C.c_void_p(handle)
        self._keep = []
        self._borrowed = True
        return self

    def _chk(self, rc, what):
        if rc != 0:
            raise BdxError("%s: %s (%s)" % (what, self.lib.bdx_strerror(rc).decode(),
                                           self.lib.bdx_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "_borrowed", False):
            self.h = C.c_void_p()
            return
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.bdx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_name_check(self, on=True):
        """every batch carries 'name_check', a second hash of the read name: mates must agree in it as well (bdx_use_name_check)"""
        self._chk(self.lib.bdx_use_name_check(self.h, 1 if on else 0), "bdx_use_name_check")
        return self

    def push_reads(self, arrs):
        b, keep = make_batch(arrs)
        self._keep.append(keep)  # the H2D copies are asynchronous: the arrays must outlive them (released by run())
        self._chk(self.lib.bdx_push(self.h, C.byref(b)), "bdx_push")

    def stream_reads(self, arrs, batch=1 << 20):
        """The streaming producer's path: fill the context's pinned staging buffers batch by batch (bdx_acquire_batch /
        bdx_submit_batch); copies and the classifier overlap the filling of the next buffer."""
        n = len(arrs["tid"])
        cols = {}
        for k, dt in BATCH_FIELDS:
            src = arrs.get(k)
            if src is None and k == "mapq":
                src = arrs.get("bdqual")
            if src is None and k == "name_key":
                src = arrs.get("name_id")
            cols[k] = np.ascontiguousarray(src, dtype=dt)
        fields = list(BATCH_FIELDS)
        if arrs.get("name_check") is not None:
            cols["name_check"] = np.ascontiguousarray(arrs["name_check"], dtype=np.uint64)
            fields.append(("name_check", np.uint64))
        for lo in range(0, n, batch):
            m = min(batch, n - lo)
            buf = L.bdx_batch_buf()
            self._chk(self.lib.bdx_acquire_batch(self.h, m, C.byref(buf)), "bdx_acquire_batch")
            for k, dt in fields:
                C.memmove(getattr(buf, k), cols[k][lo:lo + m].ctypes.data, m * np.dtype(dt).itemsize)
            self._chk(self.lib.bdx_submit_batch(self.h, m), "bdx_submit_batch")

    def reset_reads(self):
        self._chk(self.lib.bdx_reset_reads(self.h), "bdx_reset_reads")
        self._keep.clear()
        return self

    def set_device_reads(self, ptrs, n):
        """ptrs: dict field -> device pointer (int); arrays stay owned by the caller."""
        b = L.bdx_batch()
        for k, _ in BATCH_FIELDS:
            setattr(b, k, C.c_void_p(int(ptrs[k])))
        if ptrs.get("name_check"):
            b.name_check = C.c_void_p(int(ptrs["name_check"]))
        b.n = n
        self._chk(self.lib.bdx_set_device_reads(self.h, C.byref(b)), "bdx_set_device_reads")

    def run(self):
        self._chk(self.lib.bdx_run(self.h), "bdx_run")
        self._keep.clear()  # bdx_run has waited for the stream: every pushed batch is in HBM
        return self

    # ---- results ----------------------------------------------------------------------------------------
    def summary(self):
        s = L.bdx_summary()
        self._chk(self.lib.bdx_get_summary(self.h, C.byref(s)), "bdx_get_summary")
        return {k: getattr(s, k) for k, _ in s._fields_}

    def counters(self):
        lib_cnt = np.zeros(self.nlibs, np.uint32)
        bam_cnt = np.zeros(self.nbams, np.uint32)
        hist = np.zeros((self.nlibs, 11), np.uint32)
        seqcov = np.zeros(self.nlibs, np.float32)
        dens = np.zeros(self.nlibs, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_get_counters(self.h, p(lib_cnt), p(bam_cnt), p(hist), p(seqcov), p(dens)), "bdx_get_counters")
        return dict(lib_read_count=lib_cnt, bam_read_count=bam_cnt, flag_hist=hist, seqcov=seqcov, density=dens)

    def regions(self):
        n = self.summary()["n_regions"]
        out = np.zeros(n, dtype=L.REGION_DTYPE)
        self._chk(self.lib.bdx_get_regions(self.h, out.ctypes.data_as(C.c_void_p), n), "bdx_get_regions")
        return out

    def svs(self):
        n = self.summary()["n_svs"]
        out = np.zeros(n, dtype=L.SV_DTYPE)
        assert L.SV_DTYPE.itemsize == 88
        self._chk(self.lib.bdx_get_svs(self.h, out.ctypes.data_as(C.c_void_p), n), "bdx_get_svs")
        nl = int((out["lib_begin"] + out["lib_count"]).max()) if n else 0
        nc = int((out["cn_begin"] + out["cn_count"]).max()) if n else 0
        li, lp = np.zeros(nl, np.int32), np.zeros(nl, np.int32)
        ck, cv = np.zeros(nc, np.int32), np.zeros(nc, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_get_sv_lists(self.h, p(li), p(lp), nl, p(ck), p(cv), nc), "bdx_get_sv_lists")
        return out, (li, lp), (ck, cv)

    def collect_support(self, on=True):
        self._chk(self.lib.bdx_set_collect_support(self.h, int(on)), "bdx_set_collect_support")

    def sv_support(self):
        """(offsets[n_svs+1], read_index, read_flag): supporting reads per SV in the reference's order"""
        n = self.summary()["n_svs"]
        off = np.zeros(n + 1, np.uint32)
        tot = C.c_size_t()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.bdx_get_sv_support(self.h, p(off), None, None, 0, C.byref(tot)), "bdx_get_sv_support")
        idx, flg = np.zeros(tot.value, np.uint64), np.zeros(tot.value, np.uint8)
        self._chk(self.lib.bdx_get_sv_support(self.h, p(off), p(idx), p(flg), tot.value, C.byref(tot)), "bdx_get_sv_support")
        return off, idx, flg

    def read_class(self):
        n = self.summary()["n_reads"]
        out = np.zeros(n, np.uint8)
        self._chk(self.lib.bdx_get_read_class(self.h, out.ctypes.data_as(C.c_void_p), n), "bdx_get_read_class")
        return out

    def set_stage_timing(self, on=True):
        """HIP events between the stages (compact / regions / join timings); costs a few microseconds per event."""
        self._chk(self.lib.bdx_set_stage_timing(self.h, int(on)), "bdx_set_stage_timing")
        return self

    def set_enqueue_ahead(self, mode=2):
        """How the stages behind pass 1 are sized and launched (bdx_set_enqueue_ahead): 2 ahead of the pass-1 read-back, from
        the previous run of the same input or else a prior on the read count (default); 1 always from the prior (every run
        behaves like a first run); 0 only after the read-back.  True / False stand for 2 / 0."""
        mode = 2 if mode is True else (0 if mode is False else int(mode))
        self._chk(self.lib.bdx_set_enqueue_ahead(self.h, mode), "bdx_set_enqueue_ahead")
        return self

    def set_debug(self, name, value=1):
        """test / measurement switches by name (include/bdx.h bdx_set_debug); before run()"""
        self._chk(self.lib.bdx_set_debug(self.h, name.encode(), int(value)), "bdx_set_debug")
        return self

    def set_host_walk(self, on=True):
        """Send every component of the region graph through the host walk (same results as the device assembly)."""
        self._chk(self.lib.bdx_set_host_walk(self.h, int(on)), "bdx_set_host_walk")
        return self

    def walk_split(self):
        """(SVs assembled on the device, SVs from the host walk, pair groups handed to the host walk)"""
        v = [C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)]
        self._chk(self.lib.bdx_get_walk_split(self.h, C.byref(v[0]), C.byref(v[1]), C.byref(v[2])), "bdx_get_walk_split")
        return tuple(x.value for x in v)

    def was_replayed(self):
        """True if the last run met a read name more than twice and replayed the region graph read by read on the host"""
        return bool(self.lib.bdx_was_replayed(self.h))

    def cross_window_svs(self):
        """device-assembled SVs whose traversal started from a region of an earlier flush window"""
        v = C.c_uint32(0)
        self._chk(self.lib.bdx_get_cross_window_svs(self.h, C.byref(v)), "bdx_get_cross_window_svs")
        return v.value

    def timings(self):
        ms = np.zeros(12, np.float32)
        self.lib.bdx_get_timings(self.h, ms.ctypes.data_as(C.c_void_p), 12)
        return dict(zip(("classify", "compact", "regions", "join", "readback", "walk", "score", "total", "final_wait", "merge",
                         "combine_scores"), ms.tolist()))


def poisson_log_upper_tail(lam, k, device=0):
    lib = L.load()
    lam = np.ascontiguousarray(lam, np.float64)
    k = np.ascontiguousarray(k, np.int32)
    out = np.zeros(len(lam), np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.bdx_poisson_log_upper_tail(p(lam), p(k), p(out), len(lam), device)
    if rc != 0:
        raise BdxError("bdx_poisson_log_upper_tail: %s" % lib.bdx_strerror(rc).decode())
    return out
