"""ctypes mirror of include/bdx.h's bdx_bamdec_* / bdx_inflate_blocks: a BAM file decoded on the GPU (BGZF inflate,
record boundaries, record fields -> SoA columns).  For tests and bench.py; the CLI's feeder is host/device_bam.cpp.

The host side only looks at the BGZF members' 18-byte headers and 8-byte footers and at the BAM header (to know the
reference sequences and where the first record starts); everything per byte and per record happens in libbdx's kernels."""
import ctypes as C
import struct
import zlib

import numpy as np

from . import _lib as L


class bdx_bgzf_block(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("payload_len", C.c_uint32), ("inflated_len", C.c_uint32)]


class bdx_bamdec_params(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_targets", C.c_int32), ("bam_index", C.c_int32), ("only_tid", C.c_int32),
                ("region_beg", C.c_int32), ("region_end", C.c_int32), ("n_read_groups", C.c_uint32),
                ("rg_ids", C.POINTER(C.c_char_p)), ("rg_lib", C.c_void_p), ("fallback_lib", C.c_uint8),
                ("first_record_offset", C.c_uint64), ("ring_bytes", C.c_size_t), ("batch_bytes", C.c_size_t), ("batch_blocks", C.c_size_t), ("expected_bytes", C.c_size_t),
                ("piece_bytes", C.c_size_t), ("piece_blocks", C.c_size_t), ("batch_rounds", C.c_int32), ("stream_mode", C.c_int32), ("record_mode", C.c_int32), ("missing_lib_plus1", C.c_int32), ("time_kernels", C.c_int32)]


BLOCK_DTYPE = np.dtype([("offset", "<u8"), ("payload_len", "<u4"), ("inflated_len", "<u4")])


def _lib():
    lib = L.load()
    if not getattr(lib, "_bamdec_bound", False):
        vp = C.c_void_p
        lib.bdx_bamdec_create.argtypes = [C.POINTER(vp), vp, C.POINTER(bdx_bamdec_params)]
        lib.bdx_bamdec_destroy.argtypes = [vp]
        lib.bdx_bamdec_destroy.restype = None
        lib.bdx_bamdec_last_error.argtypes = [vp]
        lib.bdx_bamdec_last_error.restype = C.c_char_p
        lib.bdx_bamdec_acquire.argtypes = [vp, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(vp)]
        lib.bdx_bamdec_submit.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_int]
        lib.bdx_bamdec_progress.argtypes = [vp, vp, vp, vp, vp]
        lib.bdx_bamdec_finish.argtypes = [vp, vp]
        lib.bdx_bamdec_fetch.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(L.bdx_batch_buf)]
        lib.bdx_bamdec_stats.argtypes = [vp, vp, vp, vp, vp]
        lib.bdx_inflate_blocks.argtypes = [C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, vp]
        lib._bamdec_bound = True
    return lib


def scan_bgzf(data):
    """BGZF members of a file image (bytes / memoryview / uint8 array): structured array of
    (member offset, payload offset, payload length, inflated length); the EOF marker and other empty members included"""
    mv = memoryview(data)
    n = len(mv)
    out = []
    off = 0
    while off < n:
        if n - off < 18 or mv[off] != 31 or mv[off + 1] != 139:
            raise ValueError("not a BGZF member at %d" % off)
        xlen = mv[off + 10] | (mv[off + 11] << 8)
        bsize = None
        x = off + 12
        while x + 4 <= off + 12 + xlen:
            slen = mv[x + 2] | (mv[x + 3] << 8)
            if mv[x] == 66 and mv[x + 1] == 67 and slen == 2:
                bsize = mv[x + 4] | (mv[x + 5] << 8)
            x += 4 + slen
        if bsize is None:
            raise ValueError("BGZF member without BC field at %d" % off)
        total = bsize + 1
        isize = struct.unpack_from("<I", mv, off + total - 4)[0]
        out.append((off, off + 12 + xlen, total - 12 - xlen - 8, isize))
        off += total
    return np.array(out, dtype=[("member", "<u8"), ("payload", "<u8"), ("payload_len", "<u4"), ("inflated_len", "<u4")])


def inflate_blocks(data, members, device=0):
    """members of `data` (rows of scan_bgzf) through the GPU's inflate: (bytes, status per member, kernel ms)"""
    lib = _lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    blocks = np.zeros(len(members), dtype=BLOCK_DTYPE)
    blocks["offset"] = members["payload"]
    blocks["payload_len"] = members["payload_len"]
    blocks["inflated_len"] = members["inflated_len"]
    total = int(members["inflated_len"].astype(np.int64).sum())
    out = np.zeros(total + 64, dtype=np.uint8)
    status = np.zeros(max(len(members), 1), dtype=np.uint32)
    ms = C.c_float(0)
    rc = lib.bdx_inflate_blocks(device, buf.ctypes.data, buf.size, blocks.ctypes.data, len(members), out.ctypes.data, out.size,
                                status.ctypes.data, C.addressof(ms))
    if rc != 0:
        raise RuntimeError("bdx_inflate_blocks: %s" % lib.bdx_strerror(rc).decode())
    return out[:total], status[:len(members)], ms.value


def bam_header(data, members):
    """(target names, target lengths, index of the member that holds the first record, the record's offset in that member's
    inflated bytes) -- the header is inflated with zlib here, on the host: it is a few KB"""
    raw = bytearray()
    starts = []
    i = 0

    def need(k):
        nonlocal i
        while len(raw) < k:
            if i >= len(members):
                raise ValueError("truncated BAM header")
            m = members[i]
            starts.append(len(raw))
            if m["inflated_len"]:
                raw.extend(zlib.decompress(bytes(data[int(m["payload"]):int(m["payload"]) + int(m["payload_len"])]), -15))
            i += 1
    need(12)
    if raw[:4] != b"BAM\1":
        raise ValueError("not a BAM file")
    l_text = struct.unpack_from("<I", raw, 4)[0]
    p = 8 + l_text
    need(p + 4)
    n_ref = struct.unpack_from("<I", raw, p)[0]
    p += 4
    names, lens = [], []
    for _ in range(n_ref):
        need(p + 4)
        l = struct.unpack_from("<I", raw, p)[0]
        need(p + 4 + l + 4)
        names.append(bytes(raw[p + 4:p + 4 + l - 1]).decode())
        lens.append(struct.unpack_from("<I", raw, p + 4 + l)[0])
        p += 4 + l + 4
    # the member that holds byte p of the inflated stream (the next one if the header ends exactly at a member's end)
    if p < len(raw):
        k = max(j for j, s in enumerate(starts) if s <= p)
        return names, lens, k, p - starts[k]
    return names, lens, i, 0


class BamDecoder:
    """One BAM file through bdx_bamdec_*.  sink: a breakdancer_amd.api.BreakDancer whose store receives the records, or None
    (the columns stay in the decoder; fetch() copies them out)."""

    def __init__(self, n_targets, sink=None, device=0, bam_index=0, rg_ids=(), rg_lib=(), fallback_lib=0, region=None,
                 first_record_offset=0, ring_bytes=0, batch_bytes=0, batch_blocks=0):
        self.lib = _lib()
        p = bdx_bamdec_params()
        p.device = device
        p.n_targets = n_targets
        p.bam_index = bam_index
        p.only_tid, p.region_beg, p.region_end = region if region is not None else (-1, 0, 1 << 29)
        self._ids = (C.c_char_p * max(len(rg_ids), 1))(*[x.encode() for x in rg_ids])
        self._libs = np.asarray(list(rg_lib), dtype=np.uint8)
        p.n_read_groups = len(rg_ids)
        p.rg_ids = C.cast(self._ids, C.POINTER(C.c_char_p))
        p.rg_lib = self._libs.ctypes.data if len(rg_ids) else None
        p.fallback_lib = fallback_lib
        p.first_record_offset = first_record_offset
        p.ring_bytes = ring_bytes
        p.batch_bytes = batch_bytes
        p.batch_blocks = batch_blocks
        h = C.c_void_p()
        rc = self.lib.bdx_bamdec_create(C.byref(h), sink.h if sink is not None else None, C.byref(p))
        if rc != 0:
            raise RuntimeError("bdx_bamdec_create: %s" % self.lib.bdx_strerror(rc).decode())
        self.h = h
        self.n = None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s (%s)" % (what, self.lib.bdx_strerror(rc).decode(), self.lib.bdx_bamdec_last_error(self.h).decode()))

    def submit(self, data, members, last):
        """members: consecutive rows of scan_bgzf over `data`; their bytes are copied into the decoder's pinned staging buffer"""
        lo = int(members["member"][0])
        hi = int(members["payload"][-1]) + int(members["payload_len"][-1]) + 8
        nbytes = hi - lo
        buf, tab = C.c_void_p(), C.c_void_p()
        self._check(self.lib.bdx_bamdec_acquire(self.h, nbytes, len(members), C.byref(buf), C.byref(tab)), "bdx_bamdec_acquire")
        C.memmove(buf.value, np.frombuffer(data, dtype=np.uint8, count=nbytes, offset=lo).ctypes.data, nbytes)
        t = np.ctypeslib.as_array(C.cast(tab.value, C.POINTER(C.c_uint8)), shape=(len(members) * BLOCK_DTYPE.itemsize,)).view(BLOCK_DTYPE)
        t["offset"] = members["payload"] - np.uint64(lo)
        t["payload_len"] = members["payload_len"]
        t["inflated_len"] = members["inflated_len"]
        self._check(self.lib.bdx_bamdec_submit(self.h, nbytes, len(members), 1 if last else 0), "bdx_bamdec_submit")

    def acquire_fill(self, data, members):
        """first half of submit(): a staging buffer acquired and filled, NOT submitted yet (several may be held; submit_held takes them in order)"""
        lo = int(members["member"][0])
        hi = int(members["payload"][-1]) + int(members["payload_len"][-1]) + 8
        nbytes = hi - lo
        buf, tab = C.c_void_p(), C.c_void_p()
        self._check(self.lib.bdx_bamdec_acquire(self.h, nbytes, len(members), C.byref(buf), C.byref(tab)), "bdx_bamdec_acquire")
        C.memmove(buf.value, np.frombuffer(data, dtype=np.uint8, count=nbytes, offset=lo).ctypes.data, nbytes)
        t = np.ctypeslib.as_array(C.cast(tab.value, C.POINTER(C.c_uint8)), shape=(len(members) * BLOCK_DTYPE.itemsize,)).view(BLOCK_DTYPE)
        t["offset"] = members["payload"] - np.uint64(lo)
        t["payload_len"] = members["payload_len"]
        t["inflated_len"] = members["inflated_len"]
        return nbytes, len(members)

    def submit_held(self, held, last):
        self._check(self.lib.bdx_bamdec_submit(self.h, held[0], held[1], 1 if last else 0), "bdx_bamdec_submit")

    def feed(self, data, members, piece_blocks=512, ahead=1):
        """all of `members` (non-empty ones) in pieces of piece_blocks members; ahead > 1: that many pieces are acquired and filled before
        the oldest is submitted (what a caller that reads the file ahead does: bdx_bamdec_acquire several times, bdx_bamdec_submit in order)"""
        m = members[members["inflated_len"] > 0]
        if len(m) == 0:
            return
        # members must be contiguous in the file for one submit: an empty member in between ends a piece
        cuts = [0]
        for i in range(1, len(m)):
            contiguous = int(m["member"][i]) == int(m["payload"][i - 1]) + int(m["payload_len"][i - 1]) + 8
            if not contiguous or i - cuts[-1] >= piece_blocks:
                cuts.append(i)
        cuts.append(len(m))
        if ahead <= 1:
            for a, b in zip(cuts[:-1], cuts[1:]):
                self.submit(data, m[a:b], last=(b == len(m)))
            return
        held = []
        pieces = list(zip(cuts[:-1], cuts[1:]))
        nxt = 0
        while nxt < len(pieces) or held:
            while nxt < len(pieces) and len(held) < ahead:
                a, b = pieces[nxt]
                held.append((self.acquire_fill(data, m[a:b]), b == len(m)))
                nxt += 1
            h, last = held.pop(0)
            self.submit_held(h, last)

    def finish(self):
        n = C.c_uint64(0)
        self._check(self.lib.bdx_bamdec_finish(self.h, C.byref(n)), "bdx_bamdec_finish")
        self.n = n.value
        return self.n

    def progress(self):
        n, raw, past, err = C.c_uint64(0), C.c_uint64(0), C.c_int(0), C.c_uint32(0)
        self.lib.bdx_bamdec_progress(self.h, C.byref(n), C.byref(raw), C.byref(past), C.byref(err))
        return dict(records=n.value, raw=raw.value, past_region=past.value, error=err.value)

    def stats(self):
        a, b, c, d = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.lib.bdx_bamdec_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(compressed_bytes=a.value, inflated_bytes=b.value, pieces=c.value, blocks_walked_twice=d.value)

    def fetch(self):
        n = self.n
        cols = dict(tid=np.zeros(n, np.int32), pos=np.zeros(n, np.int32), mtid=np.zeros(n, np.int32), mpos=np.zeros(n, np.int32),
                    isize=np.zeros(n, np.int32), flag=np.zeros(n, np.uint16), qlen=np.zeros(n, np.uint16), mapq=np.zeros(n, np.uint8),
                    lib=np.zeros(n, np.uint8), bam=np.zeros(n, np.uint8), name_key=np.zeros(n, np.uint64),
                    name_check=np.zeros(n, np.uint64))
        b = L.bdx_batch_buf()
        for k, v in cols.items():
            setattr(b, k, v.ctypes.data)
        b.capacity = n
        self._check(self.lib.bdx_bamdec_fetch(self.h, 0, n, C.byref(b)), "bdx_bamdec_fetch")
        return cols

    def close(self):
        if self.h:
            self.lib.bdx_bamdec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_decoded(sink, decoders, src_file, src_index):
    """bdx_merge_decoded: the finished decoders' columns gathered into sink's store, record i = record src_index[i] of decoder src_file[i]"""
    lib = _lib()
    lib.bdx_merge_decoded.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    f = np.ascontiguousarray(src_file, dtype=np.uint8)
    i = np.ascontiguousarray(src_index, dtype=np.uint32)
    arr = (C.c_void_p * len(decoders))(*[d.h for d in decoders])
    rc = lib.bdx_merge_decoded(sink.h, arr, len(decoders), f.ctypes.data, i.ctypes.data, len(f))
    if rc != 0:
        raise RuntimeError("bdx_merge_decoded: %s (%s)" % (lib.bdx_strerror(rc).decode(), lib.bdx_last_error(sink.h).decode()))


def decode_file(path, rg_ids=(), rg_lib=(), fallback_lib=0, bam_index=0, region=None, piece_blocks=512, ring_bytes=0, sink=None, device=0,
                batch_blocks=0, ahead=1):
    """whole file -> (columns or None with a sink, target names, decoder statistics)"""
    data = np.fromfile(path, dtype=np.uint8)
    members = scan_bgzf(data)
    names, lens, k, off = bam_header(data, members)
    d = BamDecoder(len(names), sink=sink, device=device, bam_index=bam_index, rg_ids=rg_ids, rg_lib=rg_lib, fallback_lib=fallback_lib,
                   region=region, first_record_offset=off, ring_bytes=ring_bytes, batch_blocks=batch_blocks)
    try:
        d.feed(data, members[k:], piece_blocks, ahead=ahead)
        d.finish()
        cols = d.fetch() if sink is None else None
        return cols, names, d.stats()
    finally:
        d.close()
