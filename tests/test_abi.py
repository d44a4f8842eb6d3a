"""CPU tests: the C-ABI library loads and exports every symbol include/bdx.h declares (no compute calls)."""
import os
import re

import pytest

import breakdancer_amd._lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bdx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bdx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = L.load()
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), "libbdx.so does not export %s" % s
    assert set(syms) == set(L.EXPORTS)


def test_struct_layouts_match_the_header():
    assert L.SV_DTYPE.itemsize == 88 and L.SV_DTYPE.fields["logp"][1] == 80
    assert L.REGION_DTYPE.itemsize == 36
    import ctypes as C
    assert C.sizeof(L.bdx_opts) == 56 and C.sizeof(L.bdx_lib) == 28 and C.sizeof(L.bdx_batch) == 104 and L.bdx_batch.name_check.offset == 96 and C.sizeof(L.bdx_batch_buf) == 104
    assert C.sizeof(L.bdx_summary) == 48


def test_no_gpu_means_loud_failure():
    """On a box without a GPU bdx_create must fail (no CPU fallback on the product path)."""
    import ctypes as C
    lib = L.load()
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    o = L.bdx_opts()
    lib.bdx_opts_default(C.byref(o))
    assert o.min_map_qual == 35 and o.score_threshold == 30 and o.buffer_size == 100
    libs = (L.bdx_lib * 1)()
    h = C.c_void_p()
    rc = lib.bdx_create(C.byref(h), C.byref(o), libs, 1, 1, 1, 100, 0)
    assert rc != 0
    assert lib.bdx_strerror(rc).decode() != "ok"


def test_the_library_reads_no_environment_variable_for_a_behaviour_switch():
    """csrc/ may call getenv for tracing / profiling output only: switches travel through bdx_set_debug,
    bdx_set_process_option and bdx_bamdec_params."""
    allowed = {"BDX_ALLOC_TRACE", "BDX_DIST_TRACE", "BDX_BAMDEC_TRACE", "BDX_WALK_PROFILE", "BDX_KZ_PROF"}
    csrc = os.path.join(ROOT, "breakdancer_amd", "csrc")
    seen = set()
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h", ".cpp")):
            continue
        for m in re.finditer(r'getenv\(\s*"([A-Z0-9_]+)"\s*\)', open(os.path.join(csrc, name)).read()):
            seen.add(m.group(1))
        assert "getenv(" not in re.sub(r'getenv\(\s*"[A-Z0-9_]+"\s*\)', "", open(os.path.join(csrc, name)).read()), name + ": getenv of a computed name"
    assert seen <= allowed, "behaviour switches read from the environment: %s" % sorted(seen - allowed)
    assert not os.path.exists(os.path.join(csrc, "kz_inflate_lanes.hip"))


def test_process_option_names():
    lib = L.load()
    lib.bdx_set_process_option.argtypes = [__import__("ctypes").c_char_p, __import__("ctypes").c_int]
    assert lib.bdx_set_process_option(b"pin_malloc", 0) == 0
    assert lib.bdx_set_process_option(b"no_such_option", 1) != 0
    assert lib.bdx_set_process_option(None, 1) != 0
