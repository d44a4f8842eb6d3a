"""Test-side helpers: an independent pure-Python BAM decoder, the ctypes binding of the CPU oracle
(oracle/libbdoracle.so) and small utilities.  Nothing here is imported by the product."""
import ctypes as C
import gzip
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

OPT_FIELDS = ["min_len", "cut_sd", "max_sd", "min_map_qual", "min_read_pair", "seq_coverage_lim", "buffer_size",
              "transchr_rearrange", "fisher", "illumina_long_insert", "cn_lib", "print_af", "score_threshold", "chr_tid"]
# common/Options.cpp:27-41
OPT_DEFAULTS = dict(min_len=7, cut_sd=3, max_sd=1000000000, min_map_qual=35, min_read_pair=2, seq_coverage_lim=1000,
                    buffer_size=100, transchr_rearrange=0, fisher=0, illumina_long_insert=0, cn_lib=0, print_af=0,
                    score_threshold=30, chr_tid=-1)


def make_opts(**kw):
    d = dict(OPT_DEFAULTS)
    for k, v in kw.items():
        if k not in d:
            raise KeyError(k)
        d[k] = int(v)
    return d


def opts_array(d):
    return np.array([d[k] for k in OPT_FIELDS], dtype=np.int32)


# ------------------------------------------------------------------------------------------------
# Pure-Python BAM decoding (BGZF is a multi-member gzip stream): independent of the product's reader.
# ------------------------------------------------------------------------------------------------
_AUX_FIXED = {b"A": 1, b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4}
_AUX_FMT = {b"c": "<b", b"C": "<B", b"s": "<h", b"S": "<H", b"i": "<i", b"I": "<I"}


def _parse_aux(buf):
    """returns {tag: (type, value)} for the tags we care about (RG:Z, AM:int)"""
    out = {}
    o = 0
    n = len(buf)
    while o + 3 <= n:
        tag = buf[o:o + 2]
        ty = buf[o + 2:o + 3]
        o += 3
        if ty in _AUX_FIXED:
            sz = _AUX_FIXED[ty]
            if ty in _AUX_FMT:
                out[tag] = (ty, struct.unpack_from(_AUX_FMT[ty], buf, o)[0])
            o += sz
        elif ty in (b"Z", b"H"):
            e = buf.index(b"\0", o)
            out[tag] = (ty, buf[o:e].decode())
            o = e + 1
        elif ty == b"B":
            sub = buf[o:o + 1]
            cnt, = struct.unpack_from("<i", buf, o + 1)
            o += 5 + cnt * _AUX_FIXED[sub]
        else:
            raise ValueError("bad aux type %r" % ty)
    return out


def read_bam(path, keep_all=False):
    """Decode one BAM.  Returns (targets, recs) where recs holds numpy arrays over the records that pass
    the reference's reader filter (primary, tid >= 0: io/AlignmentFilter.hpp:24-34, io/BamIo.cpp:11-18);
    keep_all=True keeps every record and adds the SAM header text as recs["header"]."""
    d = gzip.decompress(open(path, "rb").read())
    assert d[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", d, 4)
    header_text = d[8:8 + l_text].decode(errors="replace")
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, o)
    o += 4
    targets = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", d, o)
        o += 4
        targets.append(d[o:o + l - 1].decode())
        o += l + 4
    cols = {k: [] for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "mapq", "bdqual", "rend")}
    names, rgs, seqs, quals, auxes = [], [], [], [], []
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o)
        o += 4
        tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, isize = struct.unpack_from("<iiBBHHHiiii", d, o)
        p = o + 32
        name = d[p:p + l_rn - 1].decode()
        rend = pos
        for k in range(n_cig):
            cg, = struct.unpack_from("<I", d, p + l_rn + 4 * k)
            if (cg & 15) in (0, 2, 3, 7, 8):  # M D N = X consume the reference (samtools bam_calend)
                rend += cg >> 4
        if n_cig == 0:
            rend = pos + 1
        p += l_rn + 4 * n_cig
        seq = d[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        qual = d[p:p + l_seq]
        p += l_seq
        aux = _parse_aux(d[p:o + bs])
        o += bs
        if not keep_all and ((flag & (0x100 | 0x800)) or tid < 0):
            continue
        am = aux.get(b"AM")
        bdqual = (am[1] & 0xFF) if am is not None else mapq  # io/Alignment.cpp:12-23 (uint8_t truncation)
        rg = aux.get(b"RG")
        for k, v in zip(cols, (tid, pos, mtid, mpos, isize, flag, l_seq, mapq, bdqual, rend)):
            cols[k].append(v)
        names.append(name)
        rgs.append(rg[1] if rg is not None and rg[0] == b"Z" else "")
        seqs.append(seq)
        quals.append(qual)
        if keep_all:
            auxes.append(aux)
    dt = dict(tid=np.int32, pos=np.int32, mtid=np.int32, mpos=np.int32, isize=np.int32, flag=np.uint16,
              qlen=np.int32, mapq=np.uint8, bdqual=np.uint8, rend=np.int32)
    recs = {k: np.array(v, dtype=dt[k]) for k, v in cols.items()}
    recs["name"] = names
    recs["rg"] = rgs
    recs["seq"] = seqs
    recs["qual"] = quals
    if keep_all:
        recs["header"] = header_text
        recs["aux"] = auxes   # {tag: (type, value)} per record: the integer and string tags
    return targets, recs


# ------------------------------------------------------------------------------------------------
# Oracle binding
# ------------------------------------------------------------------------------------------------
_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        so = os.path.join(ROOT, "oracle", "libbdoracle.so")
        src = os.path.join(ROOT, "oracle", "bd_oracle.cpp")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.bdo_new.restype = C.c_void_p
        L.bdo_new.argtypes = [C.c_char_p, C.c_void_p]
        L.bdo_error.restype = C.c_char_p
        L.bdo_lib_name.restype = C.c_char_p
        L.bdo_bam_name.restype = C.c_char_p
        L.bdo_text.restype = C.c_int64
        L.bdo_poisson_upper_tail.restype = C.c_double
        L.bdo_poisson_upper_tail.argtypes = [C.c_double, C.c_int]
        L.bdo_chisq_upper_tail.restype = C.c_double
        L.bdo_chisq_upper_tail.argtypes = [C.c_double, C.c_double]
        L.bdo_classify.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float]
        for f in ("bdo_free", "bdo_error", "bdo_nlibs", "bdo_nbams", "bdo_w0", "bdo_run"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.bdo_lib_name.argtypes = [C.c_void_p, C.c_int]
        L.bdo_bam_name.argtypes = [C.c_void_p, C.c_int]
        L.bdo_lib_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.bdo_lib_of_readgroup.argtypes = [C.c_void_p, C.c_char_p]
        L.bdo_set_targets.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.bdo_set_stream.argtypes = [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 10
        L.bdo_text.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bdo_summary.argtypes = [C.c_void_p, C.c_void_p]
        L.bdo_counters.argtypes = [C.c_void_p] * 5
        L.bdo_merged.argtypes = [C.c_void_p] * 4
        L.bdo_regions.argtypes = [C.c_void_p] * 2
        L.bdo_svs.argtypes = [C.c_void_p] * 3
        L.bdo_sv_lists.argtypes = [C.c_void_p] * 4
        L.bdo_translate_token.argtypes = [C.c_char_p]
        L.bdo_sv_support.restype = C.c_int64
        L.bdo_sv_support.argtypes = [C.c_void_p] * 4
        L.bdo_bam_load.restype = C.c_int64
        L.bdo_bam_load.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.bdo_bam_error.restype = C.c_char_p
        L.bdo_bam_tid.argtypes = [C.c_char_p, C.c_char_p]
        _oracle = L
    return _oracle


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleRun:
    """One oracle invocation: config text + options + per-BAM record streams -> everything the tests compare."""

    def __init__(self, config_text, opts):
        self.L = oracle_lib()
        self.opts = dict(opts)
        self._oa = opts_array(opts)
        self.h = self.L.bdo_new(config_text.encode(), _p(self._oa))
        err = self.L.bdo_error(self.h).decode()
        if err:
            raise RuntimeError(err)
        self.nlibs = self.L.bdo_nlibs(self.h)
        self.nbams = self.L.bdo_nbams(self.h)
        self.w0 = self.L.bdo_w0(self.h)
        self.lib_names = [self.L.bdo_lib_name(self.h, i).decode() for i in range(self.nlibs)]
        self.bam_names = [self.L.bdo_bam_name(self.h, i).decode() for i in range(self.nbams)]
        self.lib_f = np.zeros((self.nlibs, 5), dtype=np.float32)
        self.lib_i = np.zeros((self.nlibs, 2), dtype=np.int32)
        for i in range(self.nlibs):
            self.L.bdo_lib_params(self.h, i, _p(self.lib_f[i]), _p(self.lib_i[i]))
        self._keep = []
        self.streams = [None] * self.nbams

    def __del__(self):
        try:
            self.L.bdo_free(self.h)
        except Exception:
            pass

    def lib_of_rg(self, rg):
        return self.L.bdo_lib_of_readgroup(self.h, rg.encode())

    def set_targets(self, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._keep.append(arr)
        self.L.bdo_set_targets(self.h, len(names), arr)

    def set_stream(self, bam, recs):
        """recs: dict with tid,pos,mtid,mpos,isize,flag,qlen,bdqual,lib(int32),name_id(uint64)"""
        a = {k: np.ascontiguousarray(recs[k], dtype=t) for k, t in
             (("tid", np.int32), ("pos", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32),
              ("flag", np.uint16), ("qlen", np.int32), ("bdqual", np.uint8), ("lib", np.int32), ("name_id", np.uint64))}
        self._keep.append(a)
        self.streams[bam] = a
        n = len(a["tid"])
        self.L.bdo_set_stream(self.h, bam, n, *[_p(a[k]) for k in
                              ("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "bdqual", "lib", "name_id")])

    def load_bam(self, bam, path, only_tid=-1, passes=1, set_targets=False):
        """Decode `path` with the oracle's own single-threaded BGZF/BAM front end (oracle/bd_oracle_bam.cpp) `passes`
        times and use its records as the stream of physical file `bam`.  Returns (records, seconds spent decoding)."""
        sec = C.c_double(0)
        n = self.L.bdo_bam_load(self.h, bam, path.encode(), only_tid, passes, int(set_targets), C.byref(sec))
        if n < 0:
            raise RuntimeError(self.L.bdo_bam_error().decode())
        return int(n), sec.value

    def run(self):
        rc = self.L.bdo_run(self.h)
        if rc != 0:
            raise RuntimeError(self.L.bdo_error(self.h).decode())
        n = self.L.bdo_text(self.h, None, 0)
        buf = C.create_string_buffer(n)
        self.L.bdo_text(self.h, buf, n)
        self.text = buf.raw.decode()
        s = np.zeros(5, dtype=np.int64)
        self.L.bdo_summary(self.h, _p(s))
        self.ref_len, self.W, self.n_merged, self.n_regions, self.n_svs = [int(x) for x in s]
        self.lib_cnt = np.zeros(self.nlibs, dtype=np.uint32)
        self.bam_cnt = np.zeros(self.nbams, dtype=np.uint32)
        self.hist = np.zeros((self.nlibs, 11), dtype=np.uint32)
        self.seqcov = np.zeros(self.nlibs, dtype=np.float32)
        self.L.bdo_counters(self.h, _p(self.lib_cnt), _p(self.bam_cnt), _p(self.hist), _p(self.seqcov))
        self.m_bam = np.zeros(self.n_merged, dtype=np.int32)
        self.m_src = np.zeros(self.n_merged, dtype=np.int64)
        self.cls = np.zeros(self.n_merged, dtype=np.uint8)
        self.L.bdo_merged(self.h, _p(self.m_bam), _p(self.m_src), _p(self.cls))
        self.regions = np.zeros((self.n_regions, 9), dtype=np.int32)
        self.L.bdo_regions(self.h, _p(self.regions))
        self.sv_i = np.zeros((self.n_svs, 15), dtype=np.int32)
        self.sv_d = np.zeros((self.n_svs, 2), dtype=np.float64)
        self.L.bdo_svs(self.h, _p(self.sv_i), _p(self.sv_d))
        nl = int(self.sv_i[:, 13].sum()) if self.n_svs else 0
        nc = int(self.sv_i[:, 14].sum()) if self.n_svs else 0
        self.sv_lib = np.zeros((nl, 2), dtype=np.int32)
        self.sv_cn_key = np.zeros(nc, dtype=np.int32)
        self.sv_cn_val = np.zeros(nc, dtype=np.float32)
        self.L.bdo_sv_lists(self.h, _p(self.sv_lib), _p(self.sv_cn_key), _p(self.sv_cn_val))
        ns = self.L.bdo_sv_support(self.h, None, None, None)
        self.sup_off = np.zeros(self.n_svs + 1, dtype=np.int64)
        self.sup_idx = np.zeros(ns, dtype=np.int64)
        self.sup_flag = np.zeros(ns, dtype=np.uint8)
        self.L.bdo_sv_support(self.h, _p(self.sup_off), _p(self.sup_idx), _p(self.sup_flag))
        return self

    def merged_soa(self):
        """The merged, position-ordered record stream in the product's SoA batch layout (include/bdx.h)."""
        out = {}
        for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "bdqual", "lib", "name_id"):
            parts = np.empty(self.n_merged, dtype=self.streams[0][k].dtype if self.streams[0] is not None else np.int32)
            for b in range(self.nbams):
                if self.streams[b] is None:
                    continue
                m = self.m_bam == b
                parts[m] = self.streams[b][k][self.m_src[m]]
            out[k] = parts
        out["bam"] = self.m_bam.astype(np.uint8)
        return out


def name_ids(*name_lists):
    """Exact (collision-free) 64-bit ids for read names across several BAMs."""
    table = {}
    out = []
    for names in name_lists:
        ids = np.empty(len(names), dtype=np.uint64)
        for i, n in enumerate(names):
            ids[i] = table.setdefault(n, len(table) + 1)
        out.append(ids)
    return out


def filter_cmd_lines(text):
    """integration-test filter: drop '#Command' / '#Software' lines (build-common integrationtest.py:44-50)"""
    return "\n".join(l for l in text.splitlines() if not (l.startswith("#Command") or l.startswith("#Software")))


def load_chr21(opts, config_name="inv_del_bam_config"):
    """Feed the chr21 fixtures to a fresh OracleRun (python BAM decode -> oracle)."""
    gd = os.path.join(GOLDEN, "chr21")
    cfg_text = open(os.path.join(gd, config_name)).read()
    run = OracleRun(cfg_text, opts)
    decoded = [read_bam(os.path.join(gd, b)) for b in run.bam_names]
    ids = name_ids(*[d[1]["name"] for d in decoded])
    run.set_targets(decoded[0][0])
    for b, ((targets, recs), nid) in enumerate(zip(decoded, ids)):
        r = dict(recs)
        r["lib"] = np.array([run.lib_of_rg(g) for g in recs["rg"]], dtype=np.int32)
        r["name_id"] = nid
        run.set_stream(b, r)
    run.targets = decoded[0][0]
    run.decoded = decoded
    return run
