"""ctypes bindings of the two in-tree shared libraries.

  libomnirecall_hip.so   include/omnirecall_hip.h   (gfx950 scorer, the product)
  libomnirecall_host.so  include/omnirecall_host.h  (host string semantics)

There is no fallback: a missing library raises at import of this module.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("ORR_HIP_LIB") or os.path.join(_PKG, "libomnirecall_hip.so")    # (the override: A/B builds of one kernel on one box)
HOST_LIB_PATH = os.path.join(_PKG, "libomnirecall_host.so")


class OrrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libomnirecall_hip status {code}: {message}")
        self.code = code


ORR_OK, ORR_EINVAL, ORR_ENOMEM, ORR_EDEVICE, ORR_ECOMM, ORR_EDIM, ORR_ESTATE = 0, -1, -2, -3, -4, -5, -6
ORR_CAND_TRAILER, ORR_CAND_DOT_EXACT, ORR_CAND_DEAD = 1, 2, 16


class OrrConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("dim", C.c_int32), ("flags", C.c_int32),
                ("capacity_rows", C.c_int64), ("row_base", C.c_int64)]


class OrrCandidate(C.Structure):
    _fields_ = [("approx_score", C.c_double), ("dot", C.c_double), ("norm_b", C.c_double),
                ("created_ticks", C.c_int64), ("row_id", C.c_int64), ("order_key", C.c_int64),
                ("matches", C.c_int32), ("flags", C.c_int32)]


class OrrKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("algo_bytes", C.c_double)]


class OrrSearchStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("searches", "queries", "passes", "requeried", "overflowed_queries", "buffer_growths",
                                         "exact_pass_queries", "survivors_total", "survivor_samples", "survivors_max",
                                         "survivor_capacity", "vocab_tokens", "kw_hits_total", "kw_passes", "pass_mode")] + [("reserved", C.c_int64 * 1)]


def _load(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                          "There is no CPU fallback for the scorer.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _load(HIP_LIB_PATH)
host = _load(HOST_LIB_PATH)

_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double

hip.orr_abi_version.restype = C.c_int
hip.orr_device_count.restype = C.c_int
hip.orr_last_error.restype = C.c_char_p
hip.orr_index_create.restype = C.c_int
hip.orr_index_create.argtypes = [C.POINTER(OrrConfig), C.POINTER(_vp)]
hip.orr_index_destroy.restype = None
hip.orr_index_destroy.argtypes = [_vp]
hip.orr_index_append.restype = C.c_int
hip.orr_index_append.argtypes = [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]
hip.orr_index_seal.restype = C.c_int
hip.orr_index_seal.argtypes = [_vp]
hip.orr_index_set_row_base.restype = C.c_int
hip.orr_index_set_row_base.argtypes = [_vp, _i64]
hip.orr_index_rows.restype = _i64
hip.orr_index_rows.argtypes = [_vp]
hip.orr_index_dim.restype = _i32
hip.orr_index_dim.argtypes = [_vp]
hip.orr_search_batch.restype = C.c_int
hip.orr_search_batch.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]
hip.orr_search_shard.restype = C.c_int
hip.orr_search_shard.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp]
hip.orr_search_shard_ex.restype = C.c_int
hip.orr_search_shard_ex.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _vp]
hip.orr_merge_candidates_ex.restype = C.c_int
hip.orr_merge_candidates_ex.argtypes = [_i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]
hip.orr_merge_candidates.restype = C.c_int
hip.orr_merge_candidates.argtypes = [_i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]
hip.orr_index_save.restype = C.c_int
hip.orr_index_save.argtypes = [_vp, C.c_char_p]
hip.orr_index_load.restype = C.c_int
hip.orr_index_load.argtypes = [C.POINTER(OrrConfig), C.c_char_p, C.POINTER(_vp)]
hip.orr_index_set_option.restype = C.c_int
hip.orr_index_set_option.argtypes = [_vp, C.c_char_p, _i64]
hip.orr_index_delete_rows.restype = C.c_int
hip.orr_index_delete_rows.argtypes = [_vp, _i64, _vp, _vp]
hip.orr_index_compact.restype = C.c_int
hip.orr_index_compact.argtypes = [_vp, _vp]
hip.orr_cluster_compact.restype = C.c_int
hip.orr_cluster_compact.argtypes = [_vp, _vp]
hip.orr_index_live_rows.restype = _i64
hip.orr_index_live_rows.argtypes = [_vp]
hip.orr_index_view.restype = C.c_int
hip.orr_index_view.argtypes = [_vp, C.POINTER(_vp)]
hip.orr_index_screen_dots.restype = C.c_int
hip.orr_index_screen_dots.argtypes = [_vp, _i32, _i32, _vp, _vp]
hip.orr_index_screen_i8_dots.restype = C.c_int
hip.orr_index_screen_i8_dots.argtypes = [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]
hip.orr_index_set_profiling.restype = C.c_int
hip.orr_index_set_profiling.argtypes = [_vp, _i32]
hip.orr_index_kernel_stats.restype = C.c_int
hip.orr_index_kernel_stats.argtypes = [_vp, _vp, _i32]
hip.orr_index_search_stats.restype = C.c_int
hip.orr_index_search_stats.argtypes = [_vp, C.POINTER(OrrSearchStats), _i32]
hip.orr_cluster_create.restype = C.c_int
hip.orr_cluster_create.argtypes = [_vp, _i32, _i32, _i64, C.POINTER(_vp)]
hip.orr_cluster_destroy.restype = None
hip.orr_cluster_destroy.argtypes = [_vp]
hip.orr_cluster_shards.restype = _i32
hip.orr_cluster_shards.argtypes = [_vp]
hip.orr_cluster_shard.restype = _vp
hip.orr_cluster_shard.argtypes = [_vp, _i32]
hip.orr_cluster_seal.restype = C.c_int
hip.orr_cluster_seal.argtypes = [_vp]
hip.orr_cluster_rows.restype = _i64
hip.orr_cluster_rows.argtypes = [_vp]
hip.orr_cluster_search_batch.restype = C.c_int
hip.orr_cluster_search_batch.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]
hip.orr_cluster_set_option.restype = C.c_int
hip.orr_cluster_set_option.argtypes = [_vp, C.c_char_p, _i64]
hip.orr_cluster_search_stats.restype = C.c_int
hip.orr_cluster_search_stats.argtypes = [_vp, C.POINTER(OrrSearchStats), _i32]

host.orrh_is_blank.restype = _i32
host.orrh_is_blank.argtypes = [C.c_char_p, _i64]
host.orrh_lower_invariant.restype = _i64
host.orrh_lower_invariant.argtypes = [C.c_char_p, _i64, _vp, _i64]
host.orrh_query_terms.restype = _i32
host.orrh_query_terms.argtypes = [C.c_char_p, _i64, _vp, _i64, _vp, _i32]
host.orrh_build_snippet.restype = _i64
host.orrh_build_snippet.argtypes = [C.c_char_p, _i64, _i32, _vp, _i64]
host.orrh_round4.restype = _dbl
host.orrh_round4.argtypes = [_dbl]
host.orrh_has_sufficient_evidence.restype = _i32
host.orrh_has_sufficient_evidence.argtypes = [_vp, _i32, _i32, _dbl]
host.orrh_format_score_f4.restype = _i32
host.orrh_format_score_f4.argtypes = [_dbl, _vp, _i32]

host.orrh_last_error.restype = C.c_char_p
host.orrh_store_create.restype = _vp
host.orrh_store_destroy.restype = None
host.orrh_store_destroy.argtypes = [_vp]
host.orrh_store_upsert_document.restype = C.c_int
host.orrh_store_upsert_document.argtypes = [_vp, C.c_char_p, C.c_char_p, _i64]
host.orrh_store_upsert_chunks.restype = C.c_int
host.orrh_store_upsert_chunks.argtypes = [_vp, C.c_char_p, _i32, _vp, _vp, _vp, _vp, _vp, _vp]
host.orrh_store_delete_document.restype = C.c_int
host.orrh_store_delete_document.argtypes = [_vp, C.c_char_p]
host.orrh_store_import_cosmos_json.restype = C.c_int
host.orrh_store_import_cosmos_json.argtypes = [_vp, C.c_char_p, _i64, _vp, _vp]
host.orrh_store_export_cosmos_json.restype = C.c_int
host.orrh_store_export_cosmos_json.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_i64)]
host.orrh_store_chunk_count.restype = _i64
host.orrh_store_chunk_count.argtypes = [_vp]
host.orrh_service_create.restype = _vp
host.orrh_service_create.argtypes = [_vp, _i32, _i64]
host.orrh_service_stats.restype = None
host.orrh_service_stats.argtypes = [_vp, _vp, _vp, _vp]
host.orrh_service_tombstoned_rows.restype = _i64
host.orrh_service_tombstoned_rows.argtypes = [_vp]
host.orrh_service_compactions.restype = _i64
host.orrh_service_compactions.argtypes = [_vp]
host.orrh_service_delta_merges.restype = _i64
host.orrh_service_delta_merges.argtypes = [_vp]
host.orrh_service_destroy.restype = None
host.orrh_service_destroy.argtypes = [_vp]
host.orrh_service_search_json.restype = C.c_int
host.orrh_service_search_json.argtypes = [_vp, C.c_char_p, _vp, _i32, _i32, _i64, C.POINTER(_vp), C.POINTER(_i64)]
host.orrh_free.restype = None
host.orrh_free.argtypes = [_vp]
host.orrh_batcher_create.restype = _vp
host.orrh_batcher_create.argtypes = [_vp, _i32, _i32]
host.orrh_batcher_destroy.restype = None
host.orrh_batcher_destroy.argtypes = [_vp]
host.orrh_batcher_search.restype = C.c_int
host.orrh_batcher_search.argtypes = [_vp, _i32, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp]
host.orrh_batcher_search_at.restype = C.c_int
host.orrh_batcher_search_at.argtypes = [_vp, _i32, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp, _vp]
host.orrh_batcher_stats.restype = None
host.orrh_batcher_stats.argtypes = [_vp, _vp, _vp, _vp]
host.orrh_batcher_create.restype = _vp
host.orrh_batcher_create.argtypes = [_vp, _i32, _i32]
host.orrh_batcher_destroy.restype = None
host.orrh_batcher_destroy.argtypes = [_vp]
host.orrh_batcher_search.restype = C.c_int
host.orrh_batcher_search.argtypes = [_vp, _i32, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp]
host.orrh_batcher_stats.restype = None
host.orrh_batcher_stats.argtypes = [_vp, _vp, _vp, _vp]

EXPORTED_HIP_SYMBOLS = [
    "orr_abi_version", "orr_device_count", "orr_last_error", "orr_index_create", "orr_index_destroy",
    "orr_index_append", "orr_index_seal", "orr_index_rows", "orr_index_dim", "orr_search_batch",
    "orr_search_shard", "orr_search_shard_ex", "orr_merge_candidates", "orr_merge_candidates_ex", "orr_index_set_profiling", "orr_index_kernel_stats",
    "orr_index_save", "orr_index_load", "orr_index_set_row_base", "orr_index_set_option", "orr_index_screen_dots", "orr_index_screen_i8_dots", "orr_index_view",
    "orr_index_delete_rows", "orr_index_live_rows", "orr_index_compact", "orr_cluster_compact", "orr_index_search_stats",
    "orr_cluster_create", "orr_cluster_destroy", "orr_cluster_shards", "orr_cluster_shard", "orr_cluster_seal", "orr_cluster_rows",
    "orr_cluster_search_batch", "orr_cluster_search_stats", "orr_cluster_set_option",
]
EXPORTED_HOST_SYMBOLS = ["orrh_is_blank", "orrh_lower_invariant", "orrh_query_terms", "orrh_build_snippet",
                         "orrh_round4", "orrh_has_sufficient_evidence", "orrh_format_score_f4", "orrh_last_error", "orrh_store_create", "orrh_store_destroy",
                         "orrh_store_upsert_document", "orrh_store_upsert_chunks", "orrh_store_delete_document",
                         "orrh_store_chunk_count", "orrh_store_import_cosmos_json", "orrh_store_export_cosmos_json", "orrh_service_create", "orrh_service_destroy",
                         "orrh_service_search_json", "orrh_service_stats", "orrh_service_tombstoned_rows", "orrh_service_compactions", "orrh_service_delta_merges", "orrh_free", "orrh_batcher_create", "orrh_batcher_destroy",
                         "orrh_batcher_search", "orrh_batcher_search_at", "orrh_batcher_stats"]


def check(status: int) -> None:
    if status != ORR_OK:
        raise OrrError(status, (hip.orr_last_error() or b"").decode("utf-8", "replace"))
