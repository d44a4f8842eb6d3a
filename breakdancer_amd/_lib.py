"""ctypes binding of libbdx.so (include/bdx.h).  There is no CPU fallback: if the HIP library is missing or no
GPU is visible the calls fail loudly."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbdx.so")

OPT_FIELDS = ["min_len", "cut_sd", "max_sd", "min_map_qual", "min_read_pair", "seq_coverage_lim", "buffer_size",
              "transchr_rearrange", "fisher", "illumina_long_insert", "cn_lib", "print_af", "score_threshold",
              "chr_restricted"]


class bdx_opts(C.Structure):
    _fields_ = [(k, C.c_int32) for k in OPT_FIELDS]


class bdx_lib(C.Structure):
    _fields_ = [("mean_insertsize", C.c_float), ("std_insertsize", C.c_float), ("uppercutoff", C.c_float),
                ("lowercutoff", C.c_float), ("readlens", C.c_float), ("min_mapping_quality", C.c_int32),
                ("bam_index", C.c_int32)]


class bdx_batch(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "mapq", "lib", "bam",
                                           "name_key")] + [("n", C.c_size_t), ("name_check", C.c_void_p)]


class bdx_batch_buf(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "qlen", "mapq", "lib", "bam",
                                           "name_key")] + [("capacity", C.c_size_t), ("name_check", C.c_void_p)]


class bdx_summary(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_anomalous", C.c_uint64), ("covered_ref_len", C.c_uint32),
                ("window", C.c_int32), ("n_candidates", C.c_uint32), ("n_regions", C.c_uint32), ("n_pairs", C.c_uint32),
                ("n_groups", C.c_uint32), ("n_svs", C.c_uint32), ("n_svs_printed", C.c_uint32)]


REGION_DTYPE = np.dtype([("tid", "<i4"), ("start", "<i4"), ("end", "<i4"), ("normal_read_pairs", "<i4"),
                         ("fwd_read_count", "<i4"), ("rev_read_count", "<i4"), ("n_reads", "<i4"), ("stored", "<i4"),
                         ("max_qlen", "<i4")])
SV_DTYPE = np.dtype([("chr", "<i4", 2), ("pos", "<i4", 2), ("fwd", "<i4", 2), ("rev", "<i4", 2), ("flag", "<i4"),
                     ("size", "<i4"), ("score", "<i4"), ("num_reads", "<i4"), ("printed", "<i4"), ("region", "<i4", 2),
                     ("lib_begin", "<i4"), ("lib_count", "<i4"), ("cn_begin", "<i4"), ("cn_count", "<i4"),
                     ("allele_frequency", "<f4"), ("logp", "<f8")])

EXPORTS = ["bdx_opts_default", "bdx_create", "bdx_destroy", "bdx_strerror", "bdx_last_error", "bdx_reserve", "bdx_push",
           "bdx_set_device_reads", "bdx_run", "bdx_get_summary", "bdx_get_counters", "bdx_get_regions", "bdx_get_svs",
           "bdx_get_sv_lists", "bdx_get_read_class", "bdx_get_timings", "bdx_classify", "bdx_poisson_log_upper_tail",
           "bdx_device", "bdx_stream", "bdx_stage_pass1", "bdx_get_pass1_local", "bdx_set_pass1_global", "bdx_stage_compact", "bdx_stage_regions",
           "bdx_get_stage_regions", "bdx_get_region_records", "bdx_get_compact", "bdx_join_entries", "bdx_stage_walk", "bdx_set_collect_support", "bdx_get_sv_support",
           "bdx_set_host_walk", "bdx_set_debug", "bdx_use_name_check", "bdx_run_many", "bdx_get_walk_split", "bdx_trim_results", "bdx_set_stage_timing", "bdx_get_cross_window_svs",
           "bdx_set_enqueue_ahead", "bdx_was_replayed", "bdx_acquire_batch", "bdx_submit_batch", "bdx_reset_reads", "bdx_set_pass1_statistics",
           "bdx_warm_up", "bdx_set_process_option", "bdx_dist_unique_id", "bdx_dist_create", "bdx_dist_create_threads", "bdx_dist_destroy", "bdx_dist_last_error", "bdx_dist_rank",
           "bdx_dist_world", "bdx_dist_chromosome", "bdx_dist_run", "bdx_dist_result", "bdx_dist_set_collect_support", "bdx_dist_get_phase_ms", "bdx_dist_phase_name", "bdx_dist_prepare", "bdx_dist_reset_reads", "bdx_dist_get_exchange", "bdx_dist_get_collectives", "bdx_dist_set_debug", "bdx_dist_owner", "bdx_dist_plan",
           "bdx_bamdec_create", "bdx_bamdec_destroy", "bdx_bamdec_last_error", "bdx_bamdec_acquire", "bdx_bamdec_submit", "bdx_bamdec_progress",
           "bdx_bamdec_finish", "bdx_bamdec_rearm", "bdx_bamdec_fetch", "bdx_bamdec_stats", "bdx_bamdec_host_ms", "bdx_merge_decoded", "bdx_append_decoded", "bdx_inflate_blocks", "bdx_insert_size_stats"]

REGION_REC_DTYPE = np.dtype([("tid", "<i4"), ("start", "<i4"), ("end", "<i4"), ("n_reads", "<u4"), ("rev_reads", "<u4"),
                             ("nonctx_reads", "<u4"), ("normal_read_pairs", "<u4"), ("max_qlen", "<i4"), ("first_read", "<u4")])
GROUP_DTYPE = np.dtype([("key", "<u8"), ("pairs", "<u4"), ("sum_isize", "<u4")])

_lib = None


def load():
    """Load libbdx.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libbdx.so is missing at %s: build it with `make` (hipcc, gfx950). "
                           "There is no CPU fallback for the clustering path." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.bdx_opts_default.argtypes = [C.POINTER(bdx_opts)]
    L.bdx_opts_default.restype = None
    L.bdx_create.argtypes = [C.POINTER(vp), C.POINTER(bdx_opts), C.POINTER(bdx_lib), C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int]
    L.bdx_destroy.argtypes = [vp]
    L.bdx_destroy.restype = None
    L.bdx_strerror.argtypes = [C.c_int]
    L.bdx_strerror.restype = C.c_char_p
    L.bdx_last_error.argtypes = [vp]
    L.bdx_last_error.restype = C.c_char_p
    L.bdx_reserve.argtypes = [vp, C.c_size_t]
    L.bdx_push.argtypes = [vp, C.POINTER(bdx_batch)]
    L.bdx_set_device_reads.argtypes = [vp, C.POINTER(bdx_batch)]
    L.bdx_run.argtypes = [vp]
    L.bdx_get_summary.argtypes = [vp, C.POINTER(bdx_summary)]
    L.bdx_get_counters.argtypes = [vp] + [vp] * 5
    L.bdx_get_regions.argtypes = [vp, vp, C.c_size_t]
    L.bdx_get_svs.argtypes = [vp, vp, C.c_size_t]
    L.bdx_get_sv_lists.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_size_t]
    L.bdx_get_read_class.argtypes = [vp, vp, C.c_size_t]
    L.bdx_get_timings.argtypes = [vp, vp, C.c_int]
    L.bdx_set_host_walk.argtypes = [vp, C.c_int]
    L.bdx_set_debug.argtypes = [vp, C.c_char_p, C.c_int]
    L.bdx_use_name_check.argtypes = [vp, C.c_int]
    L.bdx_run_many.argtypes = [C.POINTER(vp), C.c_size_t, C.c_int]
    L.bdx_set_stage_timing.argtypes = [vp, C.c_int]
    L.bdx_set_enqueue_ahead.argtypes = [vp, C.c_int]
    L.bdx_was_replayed.argtypes = [vp]
    L.bdx_acquire_batch.argtypes = [vp, C.c_size_t, C.POINTER(bdx_batch_buf)]
    L.bdx_submit_batch.argtypes = [vp, C.c_size_t]
    L.bdx_reset_reads.argtypes = [vp]
    L.bdx_set_pass1_statistics.argtypes = [vp, vp, C.c_uint32]
    L.bdx_get_walk_split.argtypes = [vp, vp, vp, vp]
    L.bdx_get_cross_window_svs.argtypes = [vp, vp]
    L.bdx_classify.argtypes = [C.POINTER(bdx_opts), C.POINTER(bdx_lib), C.c_int, C.POINTER(bdx_batch), vp, C.c_int]
    L.bdx_poisson_log_upper_tail.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
    L.bdx_device.argtypes = [vp]
    L.bdx_stream.argtypes = [vp]
    L.bdx_stream.restype = vp
    L.bdx_stage_pass1.argtypes = [vp]
    L.bdx_get_pass1_local.argtypes = [vp, vp, vp, vp]
    L.bdx_set_pass1_global.argtypes = [vp, vp, C.c_uint32, C.c_int32]
    L.bdx_stage_compact.argtypes = [vp, C.c_uint32, vp, vp, vp]
    L.bdx_stage_regions.argtypes = [vp, C.c_int, C.c_int32, C.c_uint32]
    L.bdx_get_stage_regions.argtypes = [vp, vp, vp, vp]
    L.bdx_get_region_records.argtypes = [vp, vp, vp, C.c_size_t]
    L.bdx_get_compact.argtypes = [vp, vp, vp, vp, vp, C.c_size_t]
    L.bdx_join_entries.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp]
    L.bdx_stage_walk.argtypes = [vp, C.c_size_t, vp, vp, C.c_size_t, vp, C.c_int32, C.c_int]
    L.bdx_set_collect_support.argtypes = [vp, C.c_int]
    L.bdx_get_sv_support.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    _lib = L
    return L
