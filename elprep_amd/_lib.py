"""ctypes loaders for libelprep_hip.so (device boundary, include/elprep_hip.h) and libelprep_host.so
(host float/report code, include/elprep_host.h).

There is no CPU fallback for the device library: loading fails loudly if the shared object is missing, and
`elp_create` fails if no gfx950 device is usable.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
# ELP_HIP_SO: another build of the same library (A/B timing of two builds on one box, tools/prof/path_ab.py); never a fallback
HIP_SO = os.environ.get("ELP_HIP_SO") or os.path.join(_PKG, "libelprep_hip.so")
HOST_SO = os.path.join(_PKG, "libelprep_host.so")

_hip: Optional[C.CDLL] = None
_host: Optional[C.CDLL] = None

HIP_SYMBOLS = [
    "elp_create", "elp_destroy", "elp_last_error", "elp_sync", "elp_stream", "elp_set_header", "elp_reserve", "elp_stage", "elp_reset",
    "elp_num_records", "elp_num_qual_bytes", "elp_num_sorted", "elp_sort_coordinate", "elp_sort_ahead", "elp_get_permutation", "elp_mark_duplicates", "elp_get_flags", "elp_get_adapted",
    "elp_dup_metrics", "elp_dup_metrics_hist", "elp_bqsr_set_reference", "elp_bqsr_set_known_sites", "elp_bqsr_gather", "elp_bqsr_apply", "elp_get_qual",
    "elp_bqsr_gather_device", "elp_bqsr_tables_fetch", "elp_bqsr_quals_counted", "elp_bqsr_tables_fetch_rows", "elp_bqsr_lut_upload_rows", "elp_group_unique_id", "elp_group_init", "elp_group_rank", "elp_group_size",
    "elp_bqsr_tables_add", "elp_bqsr_tables_allreduce", "elp_allreduce_i64",
    "elp_filter_records", "elp_clean_sam", "elp_split_classify", "elp_merge_spread",
    "elp_set_read_group_ids", "elp_pinned_alloc", "elp_pinned_free", "elp_stage_bam", "elp_emit_sorted_bam", "elp_stage_bgzf", "elp_emit_sorted_bgzf",
    "elp_set_header_columns", "elp_stage_columns", "elp_set_read_group_ids_flat", "elp_filter_records_flat", "elp_group_probe", "elp_group_init_transport", "elp_copy_records", "elp_exchange_records", "elp_group_set_p2p", "elp_group_share", "elp_emit_merged_bam", "elp_bqsr_lut_upload",
    "elp_snapshot", "elp_rollback", "elp_set_tuning", "elp_profile_enable", "elp_profile_reset", "elp_profile_count", "elp_profile_get", "elp_debug_check_guards",
]
HOST_SYMBOLS = [
    "elp_bqsr_tables_new", "elp_bqsr_tables_new_rows", "elp_bqsr_tables_free", "elp_bqsr_tables_merge", "elp_bqsr_tables_finalize", "elp_bqsr_tables_empirical",
    "elp_bqsr_tables_combined", "elp_bqsr_tables_quantize", "elp_bqsr_tables_build_lut", "elp_bqsr_tables_build_lut_rows", "elp_bqsr_tables_report", "elp_host_free",
    "elp_dup_derived", "elp_dup_metrics_report", "elp_dup_metrics_report_hist",
]


def build_native(force: bool = False) -> None:
    """Compile both shared objects in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "-s", "clean"])
        subprocess.check_call(["make", "-C", os.path.join(_PKG, "host"), "-s", "clean"])
    subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "-s", "-j8"])
    subprocess.check_call(["make", "-C", os.path.join(_PKG, "host"), "-s"])


def hip() -> C.CDLL:
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise RuntimeError(f"{HIP_SO} is missing: build it with `make -C elprep_amd/csrc` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the hot path")
        L = C.CDLL(HIP_SO)
        L.elp_last_error.restype = C.c_char_p
        L.elp_last_error.argtypes = [C.c_void_p]
        L.elp_num_records.restype = C.c_uint64
        L.elp_num_records.argtypes = [C.c_void_p]
        L.elp_num_qual_bytes.restype = C.c_uint64
        L.elp_num_qual_bytes.argtypes = [C.c_void_p]
        L.elp_num_sorted.restype = C.c_uint64
        L.elp_num_sorted.argtypes = [C.c_void_p]
        L.elp_stream.restype = C.c_void_p
        L.elp_stream.argtypes = [C.c_void_p]
        L.elp_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.elp_destroy.argtypes = [C.c_void_p]
        L.elp_destroy.restype = None
        for name in ("elp_sync", "elp_reset", "elp_sort_coordinate", "elp_profile_reset", "elp_profile_count", "elp_snapshot", "elp_rollback"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.elp_set_header.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_stage.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_reserve.argtypes = [C.c_void_p] + [C.c_uint64] * 5
        L.elp_get_permutation.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_mark_duplicates.argtypes = [C.c_void_p, C.c_int]
        L.elp_sort_ahead.argtypes = [C.c_void_p, C.c_int]
        L.elp_get_flags.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_get_adapted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_dup_metrics.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.elp_dup_metrics_hist.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.elp_bqsr_set_reference.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.elp_bqsr_set_known_sites.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.elp_bqsr_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_apply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.elp_get_qual.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_set_read_group_ids.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_pinned_alloc.restype = C.c_void_p
        L.elp_pinned_alloc.argtypes = [C.c_size_t]
        L.elp_pinned_free.restype = None
        L.elp_pinned_free.argtypes = [C.c_void_p]
        L.elp_stage_bam.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint16]
        L.elp_emit_sorted_bam.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.elp_emit_sorted_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.elp_stage_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint16]
        L.elp_filter_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_clean_sam.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.elp_split_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_merge_spread.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_gather_device.argtypes = [C.c_void_p, C.c_int]
        L.elp_bqsr_tables_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_group_unique_id.argtypes = [C.c_void_p]
        L.elp_group_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.elp_group_rank.argtypes = [C.c_void_p]
        L.elp_group_size.argtypes = [C.c_void_p]
        L.elp_bqsr_tables_add.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.elp_allreduce_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.elp_set_header_columns.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.elp_stage_columns.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 19
        L.elp_set_read_group_ids_flat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_filter_records_flat.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_group_probe.argtypes = []
        L.elp_bqsr_lut_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.elp_bqsr_lut_upload_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_quals_counted.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_fetch_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_emit_merged_bam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.elp_copy_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.elp_group_init_transport.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.elp_group_set_p2p.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_exchange_records.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.elp_group_share.argtypes = [C.c_void_p, C.c_void_p]
        L.elp_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.elp_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.elp_profile_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        _hip = L
    return _hip


def host() -> C.CDLL:
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise RuntimeError(f"{HOST_SO} is missing: build it with `make -C elprep_amd/host`")
        L = C.CDLL(HOST_SO)
        L.elp_bqsr_tables_new.restype = C.c_void_p
        L.elp_bqsr_tables_new.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_new_rows.restype = C.c_void_p
        L.elp_bqsr_tables_new_rows.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_build_lut_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_free.argtypes = [C.c_void_p]
        L.elp_bqsr_tables_free.restype = None
        L.elp_bqsr_tables_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_finalize.argtypes = [C.c_void_p]
        L.elp_bqsr_tables_empirical.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_combined.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.elp_bqsr_tables_quantize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_build_lut.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.elp_bqsr_tables_report.restype = C.c_void_p
        L.elp_bqsr_tables_report.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.elp_host_free.argtypes = [C.c_void_p]
        L.elp_host_free.restype = None
        L.elp_dup_derived.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.elp_dup_metrics_report.restype = C.c_void_p
        L.elp_dup_metrics_report.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
        L.elp_dup_metrics_report_hist.restype = C.c_void_p
        L.elp_dup_metrics_report_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p]
        _host = L
    return _host
