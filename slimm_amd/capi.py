"""ctypes binding of the C ABI in include/slimm_hip.h (slimm_amd/libslimm_hip.so).

This is the only way Python reaches the HIP path; there is no CPU fallback.  If the shared
library is missing the import of this module's `lib()` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SLIMM_HIP_LIB: another build of the same library, for tuning experiments -- scripts/build_variant.sh)
LIB_PATH = os.environ.get("SLIMM_HIP_LIB") or os.path.join(_HERE, "libslimm_hip.so")

OK = 0
E_INVALID = -1
E_HIP = -2
E_REF_RANGE = -3
E_RUN_LENGTH = -4
E_KEY_COLLISION = -5
E_REGROUP = -6
E_RETRY = 2
E_NO_HITS = 1

ORDER_GROUPED = 0
ORDER_ANY = 1


class SlimmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"slimm_hip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("n_refs", C.c_uint32), ("ref_len", C.c_void_p), ("lineage", C.c_void_p),
        ("bin_width", C.c_uint32), ("avg_read_len", C.c_uint32), ("min_reads", C.c_uint32),
        ("cov_cut_off", C.c_float), ("abundance_cut_off", C.c_float), ("rank", C.c_char_p),
        ("n_taxa", C.c_uint32), ("tax_id", C.c_void_p), ("tax_rank", C.c_void_p),
        ("tax_name", C.POINTER(C.c_char_p)), ("device", C.c_int), ("record_order", C.c_int),
    ]


class Partials(C.Structure):
    _fields_ = [
        ("n_refs", C.c_uint32), ("n_taxa_dense", C.c_uint32),
        ("uniq_reads_count2", C.c_void_p), ("lca_count", C.c_void_p), ("level_marks", C.c_void_p),
        ("pairs", C.c_void_p), ("n_pairs", C.c_uint32), ("scalars", C.c_uint32 * 4),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "hits_count", "matches_count", "uniq_matches_count", "uniq_hits_count", "uniq_matches_count2",
        "reference_count", "matched_ref_length", "failed_by_cov", "failed_by_uniq_cov", "failed_by_min_read",
        "n_valid", "bin_width", "min_reads", "avg_read_len", "profile_count", "profile_failed")] + [
        ("coverage_cut_off", C.c_float), ("uniq_coverage_cut_off", C.c_float), ("expected_coverage", C.c_float),
        ("n_records", C.c_uint64), ("n_targets", C.c_uint64), ("total_bins", C.c_uint64)]


class RefColumns(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "reads_count", "uniq_reads_count", "uniq_reads_count2", "nbins", "nz_cov", "nz_uniq_cov", "nz_uniq_cov2",
        "valid", "abundance", "uniq_abundance")]


# every symbol include/slimm_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("slimm_create", C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    ("slimm_destroy", None, [_P]),
    ("slimm_last_error", C.c_char_p, [_P]),
    ("slimm_reset", C.c_int, [_P]),
    ("slimm_reset_cutoffs", C.c_int, [_P]),
    ("slimm_get_cutoff_cache", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("slimm_set_cutoff_cache", C.c_int, [_P, C.c_float, C.c_float]),
    ("slimm_set_min_reads", C.c_int, [_P, C.c_uint32]),
    ("slimm_reserve", C.c_int, [_P, C.c_uint64]),
    ("slimm_check_grouping", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("slimm_push_records", C.c_int, [_P, _P, _P, _P, _P, C.c_uint64]),
    ("slimm_warm_up", C.c_int, [C.c_int]),
    ("slimm_group_create", C.c_int, [_P, _P, C.c_uint32, C.POINTER(_P)]),
    ("slimm_group_destroy", None, [_P]),
    ("slimm_group_last_error", C.c_char_p, [_P]),
    ("slimm_group_size", C.c_uint32, [_P]),
    ("slimm_group_context", _P, [_P, C.c_uint32]),
    ("slimm_group_uses_rccl", C.c_int, [_P]),
    ("slimm_group_reset", C.c_int, [_P]),
    ("slimm_group_push_records", C.c_int, [_P, _P, _P, _P, _P, C.c_uint64]),
    ("slimm_group_get_profiles", C.c_int, [_P, C.c_char_p]),
    ("slimm_group_push_records_checked", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint64]),
    ("slimm_group_push_records_packed", C.c_int, [_P, _P, _P, _P, C.c_uint64]),
    ("slimm_group_push_records_marked", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("slimm_mark_word", C.c_uint32, [C.c_int32, C.c_uint16, C.c_int]),
    ("slimm_mark_words", None, [_P, _P, _P, C.c_uint64, _P, _P]),
    ("slimm_push_records_marked", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("slimm_push_records_marked_async", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("slimm_set_records_device_marked", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("slimm_push_staged_marked_async", C.c_int, [_P, C.c_uint32, C.c_uint64]),
    ("slimm_group_set_exchange", C.c_int, [_P, C.c_int]),
    ("slimm_group_exchange", C.c_int, [_P]),
    ("slimm_uniq_cov2_buffer", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("slimm_filter_alignments_launch", C.c_int, [_P]),
    ("slimm_get_stream", C.c_int, [_P, C.POINTER(_P)]),
    ("slimm_set_stream_ordered", C.c_int, [_P, C.c_int]),
    ("slimm_push_records_checked", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint64]),
    ("slimm_push_records_async", C.c_int, [_P, _P, _P, _P, _P, C.c_uint64]),
    ("slimm_pack_key", C.c_uint64, [C.c_uint64, C.c_uint16]),
    ("slimm_pack_keys", None, [_P, _P, C.c_uint64, _P]),
    ("slimm_push_records_packed", C.c_int, [_P, _P, _P, _P, C.c_uint64]),
    ("slimm_push_records_packed_async", C.c_int, [_P, _P, _P, _P, C.c_uint64]),
    ("slimm_set_records_device_packed", C.c_int, [_P, _P, _P, _P, C.c_uint64]),
    ("slimm_push_staged_packed_async", C.c_int, [_P, C.c_uint32, C.c_uint64]),
    ("slimm_push_wait", C.c_int, [_P]),
    ("slimm_staging_buffers", C.c_int, [_P, C.c_uint32, C.c_uint64, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                        C.POINTER(_P)]),
    ("slimm_push_staged_async", C.c_int, [_P, C.c_uint32, C.c_uint64]),
    ("slimm_staging_wait", C.c_int, [_P, C.c_uint32]),
    ("slimm_set_records_device", C.c_int, [_P, _P, _P, _P, _P, C.c_uint64]),
    ("slimm_analyze_alignments", C.c_int, [_P]),
    ("slimm_coverage_buffer", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("slimm_prepare_summary", C.c_int, [_P, C.c_uint32]),
    ("slimm_keep_bins", C.c_int, [_P, C.c_int]),
    ("slimm_coverage_summary", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("slimm_finish_coverage_merged", C.c_int, [_P, _P, C.c_uint32]),
    ("slimm_merge_summary_slices", C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("slimm_finish_coverage_reduced", C.c_int, [_P]),
    ("slimm_finish_coverage", C.c_int, [_P]),
    ("slimm_set_coverage_columns", C.c_int, [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32]),
    ("slimm_filter_alignments", C.c_int, [_P]),
    ("slimm_dense_taxa", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(_P)]),
    ("slimm_get_partials", C.c_int, [_P, C.POINTER(Partials)]),
    ("slimm_set_partials", C.c_int, [_P, C.POINTER(Partials)]),
    ("slimm_partials_buffer", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("slimm_install_merged_partials", C.c_int, [_P, C.POINTER(C.c_uint32)]),
    ("slimm_get_reads_lca_count", C.c_int, [_P]),
    ("slimm_get_profiles", C.c_int, [_P, C.c_char_p]),
    ("slimm_write_abundance", C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]),
    ("slimm_write_abundance_file", C.c_int, [_P, C.c_char_p]),
    ("slimm_get_stats", C.c_int, [_P, C.POINTER(Stats)]),
    ("slimm_get_ref_columns", C.c_int, [_P, C.POINTER(RefColumns)]),
    ("slimm_get_bins", C.c_int, [_P, C.c_int, _P]),
    ("slimm_get_read_targets", C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("slimm_taxon_count_size", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32)]),
    ("slimm_get_taxon_counts", C.c_int, [_P, C.c_int, _P, _P]),
    ("slimm_children_pairs_size", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64)]),
    ("slimm_get_children_pairs", C.c_int, [_P, C.c_int, _P, _P]),
    ("slimm_enable_kernel_timing", C.c_int, [_P, C.c_int]),
    ("slimm_time_only_kernel", C.c_int, [_P, C.c_char_p]),
    ("slimm_kernel_times", C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint32),
                                     C.c_uint32, C.POINTER(C.c_uint32), C.c_int]),
    ("slimm_push_bam_bytes", C.c_int, [_P, _P, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
    ("slimm_push_bgzf_blocks", C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]),
    ("slimm_push_sam_bytes", C.c_int, [_P, _P, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
    ("slimm_set_reference_names", C.c_int, [_P, C.POINTER(C.c_char_p)]),
    ("slimm_bgzf_inflate", C.c_int, [C.c_int, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_char_p, C.c_uint64]),
    ("slimm_bgzf_inflate_with", C.c_int, [C.c_int, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_char_p,
                                          C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("slimm_records_device", C.c_int, [_P, _P, _P, _P, _P, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    ("slimm_pin_host_buffer", C.c_int, [_P, _P, C.c_uint64]),
    ("slimm_set_input_size_hint", C.c_int, [_P, C.c_uint64]),
    ("slimm_window_memory", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("slimm_device_memory", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("slimm_group_plan", None, [C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                C.POINTER(C.c_uint32)]),
    ("slimm_grouped_records", C.c_int, [_P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("slimm_host_avg_read_length", C.c_uint32, [_P, C.c_uint64, C.c_uint32]),
    ("slimm_host_quantile_cut_off", C.c_float, [_P, C.c_uint32, C.c_float]),
    ("slimm_host_bin_of", C.c_uint32, [C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("slimm_host_canonical_read_name", C.c_uint32, [C.c_char_p, C.c_uint32, C.c_uint16, C.c_void_p]),
    ("slimm_host_q18_note", None, [_P, C.c_int, C.c_int]),
    ("slimm_host_q18_regroup_needed", C.c_int, [_P]),
    ("slimm_get_q18_runs", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("slimm_shutdown", C.c_int, []),
    ("slimm_version", C.c_char_p, []),
]

_lib = None


def lib():
    """Load libslimm_hip.so (built by `make -C slimm_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C slimm_amd/csrc). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(ctx, rc: int) -> int:
    if rc < 0:
        msg = lib().slimm_last_error(ctx)
        raise SlimmError(rc, msg.decode() if msg else "")
    return rc
