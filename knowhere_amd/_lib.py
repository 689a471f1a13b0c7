"""knowhere_amd/_lib.py -- ctypes binding of libknhip.so (include/knhip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C knowhere_amd/csrc``.
There is NO fallback: if the shared object is missing or a HIP call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KNHIP_LIB") or os.path.join(_HERE, "libknhip.so")  # KNHIP_LIB: experiments only

BRUTE_FORCE, IVF_FLAT, IVF_PQ, IVF_SQ8 = 0, 1, 2, 3
L2, IP = 0, 1
NSTAGE = 16
ABI_VERSION = 9  # KNHIP_ABI_VERSION of include/knhip.h
(STAGE_COARSE, STAGE_GROUP, STAGE_LUT, STAGE_SCAN, STAGE_MERGE, STAGE_OTHER, STAGE_SCAN_RANK0, STAGE_TABLES, STAGE_REFINE,
 STAGE_TIES) = range(10)
STAGE_NAMES = ["coarse", "group", "lut", "scan", "merge", "other", "scan_rank0", "tables", "refine", "ties"]


class KnhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"knhip error {code}: {msg}")
        self.code = code


class Desc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("metric", C.c_int32), ("dim", C.c_int32), ("device", C.c_int32),
                ("nlist", C.c_int64), ("pq_m", C.c_int32), ("pq_nbits", C.c_int32),
                ("precomputed_table_max_bytes", C.c_int64)]


class StageTimes(C.Structure):
    _fields_ = [("ms", C.c_float * NSTAGE), ("launches", C.c_int64 * NSTAGE), ("scan_bytes", C.c_double),
                ("coarse_flops", C.c_double), ("scan_items", C.c_int64), ("coarse_fallback_queries", C.c_int64), ("scan_bytes_rank0", C.c_double),
                ("mscan_queries", C.c_int64), ("mscan_overflow_queries", C.c_int64), ("mscan_candidates", C.c_int64), ("mscan_stream_bytes", C.c_double),
                ("mscan_recomputed", C.c_int64), ("pq_filter_form", C.c_int64), ("tie_queries", C.c_int64),
                ("tie_anomalies", C.c_int64)]


# every symbol include/knhip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "knhip_abi_version", "knhip_device_count", "knhip_last_error", "knhip_index_create",
    "knhip_index_destroy", "knhip_index_set_coarse", "knhip_index_set_pq", "knhip_index_set_sq", "knhip_index_set_row_scale", "knhip_index_add_assigned_by", "knhip_index_get_desc",
    "knhip_index_add_lists", "knhip_index_add_vectors", "knhip_index_set_coarse_device",
    "knhip_index_set_lists_device", "knhip_index_add_vectors_device", "knhip_index_count",
    "knhip_index_device_bytes", "knhip_index_uses_precomputed_table", "knhip_index_last_range_ranks", "knhip_search",
    "knhip_search_device", "knhip_coarse_search_device", "knhip_merge_topk_device",
    "knhip_search_canonical_device", "knhip_ties_rule_applies", "knhip_tie_flag_device", "knhip_tie_arrivals_device", "knhip_tie_resolve_device",
    "knhip_tie_flag_host", "knhip_tie_resolve_host", "knhip_refine_distances_device", "knhip_refine_rows_distances_device",
    "knhip_refine_combine_device", "knhip_refine_select_device", "knhip_refine_select_host",
    "knhip_merge_topk_host", "knhip_refine_device", "knhip_fvec_L2sqr_ny", "knhip_fvec_inner_products_ny",
    "knhip_fvec_norms_L2sqr", "knhip_fvec_madd", "knhip_int8_vec_L2sqr_ny",
    "knhip_int8_vec_inner_products_ny", "knhip_profile_enable", "knhip_profile_reset",
    "knhip_profile_get", "knhip_stage_kernel_name", "knhip_range_search", "knhip_range_search_ranked", "knhip_free", "knhip_search_preassigned_device",
    "knhip_kmeans_device", "knhip_index_train", "knhip_index_train_device", "knhip_index_add", "knhip_index_add_device",
    "knhip_index_encode_device", "knhip_index_get_coarse", "knhip_index_get_pq", "knhip_index_get_sq",
    "knhip_index_get_list_sizes", "knhip_index_get_lists", "knhip_index_get_vectors_device", "knhip_search_refine",
    "knhip_index_get_vectors", "knhip_index_find_vectors", "knhip_index_assign", "knhip_device_memory",
    "knhip_rows_create", "knhip_rows_destroy", "knhip_rows_train", "knhip_rows_train_uniform", "knhip_rows_set_trained", "knhip_rows_get_trained",
    "knhip_rows_add", "knhip_rows_add_codes", "knhip_rows_get_codes", "knhip_rows_count", "knhip_rows_code_size",
    "knhip_rows_device_bytes", "knhip_search_refine_rows", "knhip_refine_rows_device",
    "knhip_fvec_L1_ny", "knhip_fvec_Linf_ny", "knhip_fvec_norms_L2sqr_ref", "knhip_fvec_L2sqr_ny_transposed",
    "knhip_fvec_L2sqr_ny_nearest", "knhip_fvec_L2sqr_ny_nearest_y_transposed", "knhip_fvec_madd_and_argmin",
    "knhip_fvec_batch_4", "knhip_typed_vec_ny", "knhip_typed_vec_batch_4", "knhip_ivec_ny",
]


class TrainParams(C.Structure):
    _fields_ = [("niter", C.c_int32), ("max_points_per_centroid", C.c_int32), ("seed", C.c_int64),
                ("spherical", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def load():
    """Load libknhip.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C knowhere_amd/csrc). "
            "knowhere_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    if L.knhip_abi_version() != ABI_VERSION:  # the structures below mirror include/knhip.h at this version
        raise ImportError(f"{LIB_PATH} speaks ABI {L.knhip_abi_version()}, this binding {ABI_VERSION}: rebuild the library")
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    L.knhip_last_error.restype = C.c_char_p
    L.knhip_stage_kernel_name.restype = C.c_char_p
    L.knhip_index_count.restype = i64
    L.knhip_index_device_bytes.restype = i64
    L.knhip_index_create.argtypes = [C.POINTER(Desc), C.POINTER(vp)]
    L.knhip_index_destroy.argtypes = [vp]
    L.knhip_index_destroy.restype = None
    L.knhip_range_search.argtypes = [vp, vp, i64, C.c_float, i32, vp, i64, vp, C.POINTER(C.POINTER(C.c_int64)),
                                     C.POINTER(C.POINTER(C.c_float))]
    L.knhip_range_search_ranked.argtypes = [vp, vp, i64, C.c_float, vp, i64, vp, C.POINTER(C.POINTER(C.c_int64)),
                                            C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_int32))]
    L.knhip_free.argtypes = [vp]
    L.knhip_free.restype = None
    L.knhip_index_set_coarse.argtypes = [vp, vp]
    L.knhip_index_set_coarse_device.argtypes = [vp, vp]
    L.knhip_index_set_pq.argtypes = [vp, vp]
    L.knhip_index_set_sq.argtypes = [vp, vp, vp]
    L.knhip_index_set_row_scale.argtypes = [vp, vp, i32]
    L.knhip_index_add_assigned_by.argtypes = [vp, i64, vp, vp, vp]
    L.knhip_index_add_lists.argtypes = [vp, vp, vp, vp]
    L.knhip_index_set_lists_device.argtypes = [vp, vp, vp, vp]
    L.knhip_index_add_vectors.argtypes = [vp, i64, vp, vp, i64]
    L.knhip_index_add_vectors_device.argtypes = [vp, i64, vp, vp, i64]
    L.knhip_index_count.argtypes = [vp]
    L.knhip_index_device_bytes.argtypes = [vp]
    L.knhip_index_uses_precomputed_table.argtypes = [vp]
    L.knhip_index_last_range_ranks.argtypes = [vp]
    L.knhip_index_last_range_ranks.restype = C.c_int64
    L.knhip_search.argtypes = [vp, vp, i64, i32, i32, vp, i64, vp, vp]
    L.knhip_search_device.argtypes = [vp, vp, i64, i32, i32, vp, i64, vp, vp, vp]
    L.knhip_coarse_search_device.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.knhip_search_preassigned_device.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, i64, vp, vp, vp]
    L.knhip_merge_topk_device.argtypes = [i32, i64, i32, i32, vp, vp, vp, vp, vp]
    L.knhip_merge_topk_host.argtypes = [i32, i64, i32, i32, vp, vp, vp, vp]
    L.knhip_refine_device.argtypes = [i32, i32, vp, i64, i64, vp, i64, vp, i32, i32, vp, vp, vp]
    L.knhip_search_canonical_device.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, i64, vp, vp, vp]
    L.knhip_tie_flag_device.argtypes = [vp, vp, i64, i32, vp, vp, vp, C.POINTER(C.c_int32), vp]
    L.knhip_tie_arrivals_device.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp]
    L.knhip_tie_resolve_device.argtypes = [i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.knhip_tie_flag_host.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.knhip_tie_resolve_host.argtypes = [i32, i32, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.knhip_refine_distances_device.argtypes = [i32, i32, vp, i64, i64, vp, i64, vp, i32, vp, vp]
    L.knhip_refine_rows_distances_device.argtypes = [i32, vp, i64, vp, i64, vp, i32, vp, vp]
    L.knhip_refine_combine_device.argtypes = [i32, i64, vp, vp, vp]
    L.knhip_refine_select_device.argtypes = [i32, i64, vp, vp, i32, i32, vp, vp, vp]
    L.knhip_refine_select_host.argtypes = [i32, i64, vp, vp, i32, i32, vp, vp]
    for f in ("knhip_fvec_L2sqr_ny", "knhip_fvec_inner_products_ny", "knhip_int8_vec_L2sqr_ny",
              "knhip_int8_vec_inner_products_ny"):
        getattr(L, f).argtypes = [vp, vp, vp, i64, i64, vp]
    L.knhip_fvec_norms_L2sqr.argtypes = [vp, vp, i64, i64, vp]
    L.knhip_fvec_norms_L2sqr_ref.argtypes = [vp, vp, i64, i64, vp]
    L.knhip_fvec_L1_ny.argtypes = [vp, vp, vp, i64, i64, vp]
    L.knhip_fvec_Linf_ny.argtypes = [vp, vp, vp, i64, i64, vp]
    L.knhip_fvec_L2sqr_ny_transposed.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp]
    L.knhip_fvec_L2sqr_ny_nearest.argtypes = [vp, vp, vp, i64, i64, vp, vp]
    L.knhip_fvec_L2sqr_ny_nearest_y_transposed.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, vp]
    L.knhip_fvec_madd_and_argmin.argtypes = [i64, vp, C.c_float, vp, vp, vp, vp]
    L.knhip_fvec_batch_4.argtypes = [i32, vp, vp, vp, vp, vp, i64, vp, vp]
    L.knhip_typed_vec_ny.argtypes = [i32, i32, vp, vp, vp, i64, i64, vp]
    L.knhip_typed_vec_batch_4.argtypes = [i32, i32, vp, vp, vp, vp, vp, i64, vp, vp]
    L.knhip_ivec_ny.argtypes = [i32, vp, vp, vp, i64, i64, vp]
    L.knhip_fvec_madd.argtypes = [i64, vp, C.c_float, vp, vp, vp]
    tpp = C.POINTER(TrainParams)
    L.knhip_kmeans_device.argtypes = [i32, i32, i64, vp, i64, tpp, vp, i32]
    L.knhip_index_train.argtypes = [vp, i64, vp, tpp]
    L.knhip_index_train_device.argtypes = [vp, i64, vp, tpp]
    L.knhip_index_add.argtypes = [vp, i64, vp, vp]
    L.knhip_index_add_device.argtypes = [vp, i64, vp, vp]
    L.knhip_index_encode_device.argtypes = [vp, i64, vp, vp, vp, vp]
    L.knhip_index_get_coarse.argtypes = [vp, vp]
    L.knhip_index_get_pq.argtypes = [vp, vp]
    L.knhip_index_get_sq.argtypes = [vp, vp, vp]
    L.knhip_index_get_list_sizes.argtypes = [vp, vp]
    L.knhip_index_get_lists.argtypes = [vp, vp, vp]
    L.knhip_index_get_vectors_device.argtypes = [vp, C.POINTER(vp)]
    L.knhip_search_refine.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp]
    L.knhip_index_get_vectors.argtypes = [vp, i64, vp, vp]
    L.knhip_index_find_vectors.argtypes = [vp, i64, vp, vp, vp]
    L.knhip_index_assign.argtypes = [vp, i64, vp, vp]
    L.knhip_device_memory.argtypes = [i32, vp, vp]
    L.knhip_rows_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.knhip_rows_destroy.argtypes = [vp]
    L.knhip_rows_destroy.restype = None
    L.knhip_rows_train.argtypes = [vp, i64, vp]
    L.knhip_rows_train_uniform.argtypes = [vp, i64, vp, i32, C.c_float]
    L.knhip_rows_set_trained.argtypes = [vp, vp, vp]
    L.knhip_rows_get_trained.argtypes = [vp, vp, vp]
    L.knhip_rows_add.argtypes = [vp, i64, vp]
    L.knhip_rows_add_codes.argtypes = [vp, i64, vp]
    L.knhip_rows_get_codes.argtypes = [vp, vp]
    for f in ("knhip_rows_count", "knhip_rows_code_size", "knhip_rows_device_bytes"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = i64
    L.knhip_search_refine_rows.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp]
    L.knhip_refine_rows_device.argtypes = [i32, vp, i64, vp, i64, vp, i32, i32, vp, vp, vp]
    L.knhip_profile_enable.argtypes = [vp, C.c_int]
    L.knhip_profile_reset.argtypes = [vp]
    L.knhip_profile_get.argtypes = [vp, C.POINTER(StageTimes)]
    L.knhip_stage_kernel_name.argtypes = [C.c_int, C.c_int]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise KnhipError(rc, load().knhip_last_error().decode(errors="replace"))
