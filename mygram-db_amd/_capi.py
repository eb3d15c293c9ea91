"""ctypes bindings of libmygram_gpu.so (include/mygram_gpu.h, include/mygram_tools.h).

The library is the product; this module only declares its C ABI for Python callers (tests, bench, the multi-GPU
driver). It fails loudly if the library is missing: there is no Python or CPU fallback for any operator.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MGX_LIBRARY: tools only (e.g. the -DMGX_ABLATION build of `make ablation`); the product is libmygram_gpu.so
LIB_PATH = os.environ.get("MGX_LIBRARY") or os.path.join(_HERE, "libmygram_gpu.so")

ABI_VERSION = 3
GRAM_ABSENT = 0xFFFFFFFF
SORT_DOCID, SORT_SCORE = 0, 1

# every symbol include/mygram_gpu.h and include/mygram_tools.h declare
# int (*mgx_gather_fn)(void* user, const void* mine, void* all, uint64_t bytes, void* hip_stream)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)

EXPORTS = [
    "mgx_abi_version", "mgx_last_error", "mgx_free", "mgx_device_count",
    "mgx_columns_build", "mgx_columns_from_mgix", "mgx_dump_open", "mgx_dump_view_get", "mgx_dump_filter_column_get",
    "mgx_dump_take_columns", "mgx_dump_destroy", "mgx_columns_view_get", "mgx_columns_lookup", "mgx_columns_destroy",
    "mgx_index_create", "mgx_index_destroy", "mgx_posting_size", "mgx_index_memory_bytes",
    "mgx_index_add_filter_bitmap", "mgx_index_add_filter_column", "mgx_index_filter_compare", "mgx_facet_counts",
    "mgx_index_filter_column_read", "mgx_index_copy_text", "mgx_index_read_text", "mgx_index_filter_column_export", "mgx_index_set_live_bitmap", "mgx_index_clear_postings", "mgx_index_update_filter_bitmap", "mgx_index_set_doc_map",
    "mgx_index_invalidate_statistics", "mgx_index_synchronize", "mgx_batch_df_merge_local", "mgx_batch_merge_local",
    "mgx_index_set_batch_order", "mgx_index_attach_text", "mgx_batch_count_df", "mgx_batch_df_buffer", "mgx_score_documents_text",
    "mgx_batch_prepare", "mgx_batch_reset", "mgx_batch_stream", "mgx_batch_execute", "mgx_batch_fetch", "mgx_batch_export_topk",
    "mgx_batch_merge_shards", "mgx_batch_export_buffer", "mgx_comm_unique_id", "mgx_comm_create", "mgx_comm_destroy", "mgx_comm_abort",
    "mgx_batch_exchange", "mgx_batch_exchange_df", "mgx_batch_execute_sharded", "mgx_batch_execute_gather", "mgx_batch_algorithmic_bytes", "mgx_batch_kernel_time_ms", "mgx_batch_destroy",
    "mgx_and", "mgx_or", "mgx_not", "mgx_threshold", "mgx_retain", "mgx_score_documents", "mgx_sort_by_score",
    "mgxt_corpus_generate", "mgxt_corpus_view", "mgxt_corpus_destroy", "mgxt_measure_read_bandwidth", "mgxt_fail_device_allocs",
]


class BuildParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("version", C.c_uint32), ("ngram_size", C.c_int32),
                ("kanji_ngram_size", C.c_int32), ("cross_boundary_ngrams", C.c_int32), ("n_threads", C.c_int32)]


class MgixInfo(C.Structure):
    _fields_ = [("version", C.c_uint32), ("ngram_size", C.c_int32), ("kanji_ngram_size", C.c_int32),
                ("cross_boundary_ngrams", C.c_int32), ("normalize_nfkc", C.c_int32), ("normalize_lower", C.c_int32),
                ("normalize_width", C.c_char * 16), ("n_terms", C.c_uint64)]


class ColumnsView(C.Structure):
    _fields_ = [("n_grams", C.c_uint64), ("key_bytes", C.c_void_p), ("key_off", C.c_void_p),
                ("offsets", C.c_void_p), ("docids", C.c_void_p), ("tf", C.c_void_p), ("n_postings", C.c_uint64),
                ("first_doc_id", C.c_uint32), ("n_docs", C.c_uint64), ("doc_len", C.c_void_p),
                ("bm25_doc_count", C.c_uint64), ("bm25_total_len", C.c_uint64),
                ("tf_overflow_pos", C.c_void_p), ("tf_overflow_val", C.c_void_p), ("n_tf_overflow", C.c_uint64)]


class IndexDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("version", C.c_uint32), ("device", C.c_int32),
                ("tile_shift", C.c_uint32), ("first_doc_id", C.c_uint32), ("n_docs", C.c_uint64),
                ("n_grams", C.c_uint64), ("offsets", C.c_void_p), ("docids", C.c_void_p), ("tf", C.c_void_p),
                ("doc_len", C.c_void_p), ("dense_threshold", C.c_double),
                ("tf_overflow_pos", C.c_void_p), ("tf_overflow_val", C.c_void_p), ("n_tf_overflow", C.c_uint64)]


class Term(C.Structure):
    _fields_ = [("gram_ids", C.c_void_p), ("n_grams", C.c_uint32), ("threshold", C.c_uint32), ("idf", C.c_double),
                ("text", C.c_void_p), ("text_len", C.c_uint32)]


class Filter(C.Structure):
    _fields_ = [("bitmap_id", C.c_uint32), ("negate", C.c_uint32)]


class Query(C.Structure):
    _fields_ = [("terms", C.c_void_p), ("n_terms", C.c_uint32), ("not_terms", C.c_void_p),
                ("n_not_terms", C.c_uint32), ("filters", C.c_void_p), ("n_filters", C.c_uint32),
                ("sort", C.c_uint32), ("limit", C.c_uint32), ("offset", C.c_uint32), ("reverse", C.c_uint32),
                ("k1", C.c_double), ("b", C.c_double), ("total_docs", C.c_uint64), ("avg_doc_length", C.c_double),
                ("expr", C.c_void_p), ("n_expr", C.c_uint32), ("universe_first", C.c_uint32),
                ("universe_count", C.c_uint64), ("exact_text", C.c_uint32), ("score_terms", C.c_void_p),
                ("n_score_terms", C.c_uint32)]


class DumpView(C.Structure):
    _fields_ = [("table_name", C.c_char_p), ("index_info", MgixInfo), ("first_doc_id", C.c_uint32), ("n_docs", C.c_uint64),
                ("n_existing", C.c_uint64), ("exists", C.c_void_p), ("text_bytes", C.c_void_p), ("text_off", C.c_void_p),
                ("n_filter_columns", C.c_uint32), ("has_texts", C.c_int32)]


class DumpFilterColumn(C.Structure):
    _fields_ = [("name", C.c_char_p), ("value_type", C.c_uint32), ("values", C.c_void_p), ("is_null", C.c_void_p),
                ("string_bytes", C.c_void_p), ("string_off", C.c_void_p)]


class FilterColumnDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("version", C.c_uint32), ("value_class", C.c_uint32), ("n_values", C.c_uint32),
                ("values", C.c_void_p), ("is_null", C.c_void_p), ("value_ids", C.c_void_p)]


class ExprToken(C.Structure):
    _fields_ = [("op", C.c_uint32), ("arg", C.c_uint32)]


EXPR_TERM, EXPR_EMPTY, EXPR_AND, EXPR_OR, EXPR_NOT = 0, 1, 2, 3, 4


class QueryResult(C.Structure):
    _fields_ = [("total", C.c_uint64), ("total_candidates", C.c_uint64), ("after_intersection", C.c_uint64),
                ("after_not", C.c_uint64), ("after_filters", C.c_uint64), ("n_docs", C.c_uint32),
                ("docs_begin", C.c_uint32)]


class ResultView(C.Structure):
    _fields_ = [("n_queries", C.c_uint32), ("queries", C.POINTER(QueryResult)), ("docs", C.POINTER(C.c_uint32)),
                ("scores", C.POINTER(C.c_double))]


_lib = None


def load():
    """Loads libmygram_gpu.so. Raises (never falls back) when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make` (or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise ImportError("libmygram_gpu.so does not export %s" % name)
    vp, sz = C.c_void_p, C.c_size_t
    u32, u64, i32, f64 = C.c_uint32, C.c_uint64, C.c_int, C.c_double
    L.mgx_abi_version.restype = i32
    L.mgx_last_error.restype = C.c_char_p
    L.mgx_free.argtypes = [vp]
    L.mgx_free.restype = None
    L.mgx_device_count.restype = i32
    L.mgx_columns_build.argtypes = [C.POINTER(BuildParams), vp, vp, u32, u64, C.POINTER(vp)]
    L.mgx_columns_view_get.argtypes = [vp, C.POINTER(ColumnsView)]
    L.mgx_columns_lookup.argtypes = [vp, C.c_char_p, sz, C.POINTER(u32), C.POINTER(i32)]
    L.mgx_columns_destroy.argtypes = [vp]
    L.mgx_columns_destroy.restype = None
    L.mgx_index_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(vp)]
    L.mgx_index_destroy.argtypes = [vp]
    L.mgx_index_destroy.restype = None
    L.mgx_posting_size.argtypes = [vp, u32, C.POINTER(u64)]
    L.mgx_index_memory_bytes.argtypes = [vp, C.POINTER(u64)]
    L.mgx_columns_from_mgix.argtypes = [vp, u64, u32, u64, C.POINTER(vp), C.POINTER(MgixInfo)]
    L.mgx_index_add_filter_bitmap.argtypes = [vp, vp, u64, C.POINTER(u32)]
    L.mgx_batch_prepare.argtypes = [vp, C.POINTER(Query), u32, C.POINTER(vp)]
    L.mgx_batch_reset.argtypes = [vp, C.POINTER(Query), u32]
    L.mgx_batch_stream.argtypes = [vp, C.POINTER(vp)]
    L.mgx_batch_execute.argtypes = [vp, vp]
    L.mgx_batch_fetch.argtypes = [vp, C.POINTER(ResultView)]
    L.mgx_index_attach_text.argtypes = [vp, vp, vp]
    L.mgx_dump_open.argtypes = [C.c_char_p, u64, C.c_char_p, C.POINTER(vp)]
    L.mgx_dump_view_get.argtypes = [vp, C.POINTER(DumpView)]
    L.mgx_dump_filter_column_get.argtypes = [vp, u32, C.POINTER(DumpFilterColumn)]
    L.mgx_dump_take_columns.argtypes = [vp, C.POINTER(vp)]
    L.mgx_dump_destroy.argtypes = [vp]
    L.mgx_dump_destroy.restype = None
    L.mgx_index_set_batch_order.argtypes = [vp, C.c_uint32]
    L.mgx_index_add_filter_column.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    L.mgx_index_filter_compare.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, i32, i32, C.POINTER(C.c_uint32)]
    L.mgx_facet_counts.argtypes = [vp, vp, C.c_uint32, vp, C.POINTER(C.c_uint64)]
    L.mgx_batch_count_df.argtypes = [vp, vp]
    L.mgx_batch_df_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u32)]
    L.mgx_batch_export_topk.argtypes = [vp, vp, vp, C.POINTER(u32), vp]
    L.mgx_batch_export_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.mgx_batch_merge_shards.argtypes = [vp, u32, vp, u64, vp, u64, vp]
    L.mgx_comm_unique_id.argtypes = [vp]
    L.mgx_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.mgx_comm_destroy.argtypes = [vp]
    L.mgx_comm_destroy.restype = None
    L.mgx_batch_exchange.argtypes = [vp, vp, vp]
    L.mgx_batch_execute_sharded.argtypes = [vp, vp, vp]
    L.mgx_batch_execute_gather.argtypes = [vp, C.c_int, GATHER_FN, vp, vp]
    L.mgx_batch_exchange_df.argtypes = [vp, vp, vp]
    L.mgx_batch_algorithmic_bytes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.mgx_batch_kernel_time_ms.argtypes = [vp, C.POINTER(f64), C.POINTER(u32)]
    L.mgx_batch_destroy.argtypes = [vp]
    L.mgx_batch_destroy.restype = None
    pp32, p64 = C.POINTER(C.POINTER(u32)), C.POINTER(u64)
    L.mgx_and.argtypes = [vp, vp, u32, u64, i32, pp32, p64]
    L.mgx_or.argtypes = [vp, vp, u32, pp32, p64]
    L.mgx_not.argtypes = [vp, vp, u64, vp, u32, pp32, p64]
    L.mgx_threshold.argtypes = [vp, vp, u32, u32, pp32, p64]
    L.mgx_retain.argtypes = [vp, vp, u64, vp, u32, pp32, p64]
    L.mgx_score_documents.argtypes = [vp, vp, u64, vp, vp, u32, f64, f64, f64, vp]
    L.mgx_score_documents_text.argtypes = [vp, vp, u64, vp, vp, vp, u32, f64, f64, f64, vp]
    L.mgx_sort_by_score.argtypes = [vp, vp, vp, u64, i32, u32, u32, pp32, p64]
    L.mgxt_corpus_generate.argtypes = [u64, u64, u64, i32, C.POINTER(vp)]
    L.mgxt_corpus_view.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.mgxt_corpus_destroy.argtypes = [vp]
    L.mgxt_corpus_destroy.restype = None
    L.mgxt_measure_read_bandwidth.argtypes = [i32, u64, i32, C.POINTER(C.c_double)]
    L.mgxt_fail_device_allocs.argtypes = [i32, i32]
    L.mgxt_fail_device_allocs.restype = None
    _lib = L
    return L


class MgxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mgx error %d: %s" % (code, message))
        self.code = code


def check(rc):
    if rc != 0:
        raise MgxError(rc, load().mgx_last_error().decode("utf-8", "replace"))
