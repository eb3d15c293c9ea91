"""ctypes bindings of libmygram_shim.so (include/mygram_shim_c.h): the C face of the C++17 host layer
(mygram-db_amd/csrc/shim/). bench.py and the tests use it to run whole batches — planning, compilation, execution,
fetch — in C++, pipelined by search_pipeline::BatchExecutor. Fails loudly when the library is missing."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmygram_shim.so")
EXPORTS = ["mgxs_last_error", "mgxs_table_adopt", "mgxs_table_set_global_stats", "mgxs_table_destroy",
           "mgxs_table_set_normalization", "mgxs_table_set_absent_grams", "mgxs_normalize_uses_icu", "mgxs_normalize_text",
           "mgxs_executor_create", "mgxs_executor_create_sharded", "mgxs_executor_destroy", "mgxs_executor_warm",
           "mgxs_submit", "mgxs_wait", "mgxs_table_from_dump", "mgxs_table_add_filter_column", "mgxs_search", "mgxs_facet",
           "mgxs_table_add_document", "mgxs_table_update_document", "mgxs_table_remove_document", "mgxs_table_mutation_stats", "mgxs_table_compact", "mgxs_table_set_mutation_staleness", "mgxs_table_update_filters",
           "mgxs_batcher_create", "mgxs_batcher_destroy", "mgxs_batcher_search", "mgxs_batcher_stats"]
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make`" % LIB_PATH)
    from . import _capi
    _capi.load()  # libmygram_gpu.so first (the shim links against it)
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise ImportError("libmygram_shim.so does not export %s" % name)
    vp, u32, u64, i32, f64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    L.mgxs_last_error.restype = C.c_char_p
    L.mgxs_table_adopt.argtypes = [vp, vp, i32, i32, i32, C.POINTER(vp)]
    L.mgxs_table_set_global_stats.argtypes = [vp, u64, f64, vp, u64]
    L.mgxs_table_set_normalization.argtypes = [vp, i32, C.c_char_p, i32]
    L.mgxs_table_set_absent_grams.argtypes = [vp, u64, vp, vp]
    L.mgxs_normalize_uses_icu.restype = i32
    L.mgxs_normalize_text.argtypes = [C.c_char_p, C.c_size_t, i32, C.c_char_p, i32, vp, C.c_size_t,
                                      C.POINTER(C.c_size_t)]
    L.mgxs_table_destroy.argtypes = [vp]
    L.mgxs_table_destroy.restype = None
    L.mgxs_executor_create.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.mgxs_executor_create_sharded.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.mgxs_executor_destroy.argtypes = [vp]
    L.mgxs_executor_destroy.restype = None
    L.mgxs_submit.argtypes = [vp, u32, vp, vp, u32, u32, i32, i32, C.POINTER(u64)]
    L.mgxs_executor_warm.argtypes = [vp, u32, vp, vp, u32, u32, i32, i32, i32]
    L.mgxs_wait.argtypes = [vp, u64, vp, vp, vp, vp, vp]
    L.mgxs_table_add_filter_column.argtypes = [vp, C.c_char_p, i32, u64, vp, vp, vp]
    L.mgxs_table_from_dump.argtypes = [C.c_char_p, u64, C.c_char_p, i32, C.POINTER(vp)]
    L.mgxs_search.argtypes = [vp, u32, vp, u32, vp, u32, vp, vp, vp, i32, i32, u32, u32, C.POINTER(u64), C.POINTER(u32), vp, vp]
    L.mgxs_facet.argtypes = [vp, u32, vp, u32, vp, u32, vp, vp, vp, C.c_char_p, u32, u32, C.POINTER(u64), C.POINTER(u64),
                             C.POINTER(u32), vp, vp, C.c_size_t, vp]
    L.mgxs_table_add_document.argtypes = [vp, u32, C.c_char_p, C.c_size_t, u32, vp, vp, vp, vp]
    L.mgxs_table_update_document.argtypes = [vp, u32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, i32, u32, vp, vp, vp, vp]
    L.mgxs_table_remove_document.argtypes = [vp, u32, C.c_char_p, C.c_size_t]
    L.mgxs_table_compact.argtypes = [vp]
    L.mgxs_table_set_mutation_staleness.argtypes = [vp, u64]
    L.mgxs_table_update_filters.argtypes = [vp, u32, u32, vp, vp, vp, vp]
    L.mgxs_table_mutation_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.mgxs_batcher_create.argtypes = [vp, u32, u32, i32, i32, C.POINTER(vp)]
    L.mgxs_batcher_destroy.argtypes = [vp]
    L.mgxs_batcher_destroy.restype = None
    L.mgxs_batcher_search.argtypes = [vp, u32, vp, u32, u32, i32, i32, C.POINTER(u64), C.POINTER(u32), vp, vp]
    L.mgxs_batcher_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    _lib = L
    return L


class ShimError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise ShimError("mgxs error %d: %s" % (rc, load().mgxs_last_error().decode("utf-8", "replace")))


def normalize_uses_icu():
    """Which branch of mygram::utils::NormalizeText this build runs (ICU, or the reference's ASCII fallback)."""
    return bool(load().mgxs_normalize_uses_icu())


def normalize_text(text, nfkc=True, width="keep", lower=True):
    """mygram::utils::NormalizeText (src/utils/string_utils.cpp:295-380) of the C++ host layer. `text` is str or bytes
    (bytes may be invalid UTF-8: the answer is then b""); returns the same type."""
    raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    cap = len(raw) * 3 + 64
    n = C.c_size_t(0)
    buf = C.create_string_buffer(cap)
    rc = load().mgxs_normalize_text(raw, len(raw), int(nfkc), width.encode(), int(lower), buf, cap, C.byref(n))
    if rc == 3:  # the rare code point that expands further (U+FDFA: 1 -> 18)
        cap = n.value
        buf = C.create_string_buffer(cap)
        rc = load().mgxs_normalize_text(raw, len(raw), int(nfkc), width.encode(), int(lower), buf, cap, C.byref(n))
    _check(rc)
    out = buf.raw[: n.value]
    return out.decode("utf-8") if isinstance(text, str) else out


class QueryBatch:
    """A batch of conjunctive queries as the C arrays mgxs_submit takes (built once, outside any timed loop: these are
    the request strings a front end would hand over)."""

    def __init__(self, term_lists):
        self.n = len(term_lists)
        self.n_terms = np.asarray([len(t) for t in term_lists], dtype=np.uint32)
        flat = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for t in term_lists for s in t]
        self._bytes = flat  # keeps the strings alive
        self.terms = (C.c_char_p * max(len(flat), 1))(*flat)


class Table:
    """mygramdb::index::Index adopted from an engine.Index's handles (Index::Adopt)."""

    def __init__(self, index):
        self._index = index  # keeps columns + device index alive
        h = C.c_void_p()
        _check(load().mgxs_table_adopt(index.columns._h, index.device_index._h, index.ngram_size,
                                       index.kanji_ngram_size, int(index.cross_boundary), C.byref(h)))
        self._h = h
        _check(load().mgxs_table_set_normalization(self._h, int(index.normalize_nfkc), index.normalize_width.encode(),
                                                   int(index.normalize_lower)))
        if index._global_sizes is not None:
            sizes = np.ascontiguousarray(index._global_sizes, dtype=np.uint64)
            _check(load().mgxs_table_set_global_stats(self._h, int(index.total_docs), float(index.avg_doc_length),
                                                      sizes.ctypes.data, len(sizes)))
        if getattr(index, "_global_dict", None) is not None:
            # grams only other shards hold: known to the planner with their table-wide size (MGX_GRAM_ABSENT on this shard)
            absent = [(k, v) for k, v in index._global_dict.items() if v > 0 and index.columns.lookup(k) is None
                      and b"\x00" not in k]
            if absent:
                keys = (C.c_char_p * len(absent))(*[k for k, _ in absent])
                vals = np.asarray([v for _, v in absent], dtype=np.uint64)
                _check(load().mgxs_table_set_absent_grams(self._h, len(absent), C.cast(keys, C.c_void_p), vals.ctypes.data))

    @classmethod
    def from_dump(cls, data, table_name=None, device=0):
        """Index::FromDump: a table of a reference dump ("MGDB" v2) with its texts, id set and filter columns."""
        self = cls.__new__(cls)
        self._index = None
        raw = bytes(data)
        h = C.c_void_p()
        _check(load().mgxs_table_from_dump(raw, len(raw), table_name.encode() if table_name else None, device, C.byref(h)))
        self._h = h
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            load().mgxs_table_destroy(self._h)
            self._h = None

    # ---- FILTER conditions and FACET through the C++ planner -----------------------------------------------------
    VALUE_TYPES = {"bool": 1, "int8": 2, "uint8": 3, "int16": 4, "uint16": 5, "int32": 6, "uint32": 7, "int64": 8,
                   "uint64": 9, "time": 10, "string": 11, "double": 12}
    OPS = {"=": 0, "!=": 1, ">": 2, ">=": 3, "<": 4, "<=": 5}

    def add_filter_column(self, name, value_type, values, is_null=None):
        """Index::AddFilterColumn. values: one per doc slot (numbers, or str/bytes for "string"); is_null: bool mask."""
        t = self.VALUE_TYPES[value_type]
        n = len(values)
        nul = np.ascontiguousarray(is_null, dtype=np.uint8) if is_null is not None else None
        if t == 11:
            raw = [(v.encode("utf-8") if isinstance(v, str) else bytes(v)) for v in values]
            arr = (C.c_char_p * max(n, 1))(*raw)
            _check(load().mgxs_table_add_filter_column(self._h, name.encode(), t, n, None, C.cast(arr, C.c_void_p),
                                                       nul.ctypes.data if nul is not None else None))
            return
        dt = np.float64 if t == 12 else (np.uint64 if t in (3, 5, 7, 9) else np.int64)
        a = np.ascontiguousarray(values, dtype=dt)
        _check(load().mgxs_table_add_filter_column(self._h, name.encode(), t, n, a.ctypes.data, None,
                                                   nul.ctypes.data if nul is not None else None))

    # ---- mutable tables: the binlog applier's calls after the index was built ----------------------------------------
    def _filters(self, filters):
        """{name: (value_type, value)} -> the C arrays of mgxs_table_add_document (value None = NULL)."""
        items = list((filters or {}).items())
        n = len(items)
        names = (C.c_char_p * max(n, 1))(*[k.encode() for k, _ in items])
        types = np.zeros(max(n, 1), np.int32)
        vals = np.zeros(max(n, 1), np.uint64)
        strs = []
        for i, (_, (vt, v)) in enumerate(items):
            t = 0 if v is None else self.VALUE_TYPES[vt]
            types[i] = t
            strs.append(b"")
            if t == 11:
                strs[-1] = v.encode("utf-8") if isinstance(v, str) else bytes(v)
            elif t == 12:
                vals[i] = np.float64(v).view(np.uint64)
            elif t != 0:
                vals[i] = np.int64(v).view(np.uint64) if t not in (3, 5, 7, 9) else np.uint64(v)
        sarr = (C.c_char_p * max(n, 1))(*strs)
        return n, names, types, vals, sarr

    @staticmethod
    def _text(t):
        return t.encode("utf-8") if isinstance(t, str) else bytes(t)

    def add_document(self, doc_id, text, filters=None):
        """Index::AddDocument on a built index (normalized text): the document joins the delta index."""
        n, names, types, vals, sarr = self._filters(filters)
        raw = self._text(text)
        _check(load().mgxs_table_add_document(self._h, doc_id, raw, len(raw), n, C.cast(names, C.c_void_p), types.ctypes.data,
                                              vals.ctypes.data, C.cast(sarr, C.c_void_p)))

    def update_document(self, doc_id, old_text, new_text, filters=None):
        """Index::UpdateDocument; filters None: the document keeps its filter values."""
        n, names, types, vals, sarr = self._filters(filters)
        o, w = self._text(old_text), self._text(new_text)
        _check(load().mgxs_table_update_document(self._h, doc_id, o, len(o), w, len(w), int(filters is not None), n,
                                                 C.cast(names, C.c_void_p), types.ctypes.data, vals.ctypes.data,
                                                 C.cast(sarr, C.c_void_p)))

    def update_filters(self, doc_id, filters):
        """Index::UpdateFilters: new filter values for a document whose text stays."""
        n, names, types, vals, sarr = self._filters(filters)
        _check(load().mgxs_table_update_filters(self._h, doc_id, n, C.cast(names, C.c_void_p), types.ctypes.data,
                                                vals.ctypes.data, C.cast(sarr, C.c_void_p)))

    def remove_document(self, doc_id, text):
        """Index::RemoveDocument (text = the document's current normalized text)."""
        raw = self._text(text)
        _check(load().mgxs_table_remove_document(self._h, doc_id, raw, len(raw)))

    def set_mutation_staleness(self, seconds):
        """Index::SetMutationStaleness: how long recorded changes may wait before queries see them (0: not at all)."""
        _check(load().mgxs_table_set_mutation_staleness(self._h, int(seconds * 1e6)))

    def compact(self):
        """Index::Compact: the main index rebuilt from the current documents (delta and live row go)."""
        _check(load().mgxs_table_compact(self._h))

    def mutation_stats(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(load().mgxs_table_mutation_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"main_documents": a.value, "delta_documents": b.value, "removed_from_main": c.value, "epoch": d.value}

    @staticmethod
    def _strs(items):
        raw = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in items]
        return (C.c_char_p * max(len(raw), 1))(*raw), len(raw)

    def _conds(self, conditions):
        cols, _ = self._strs([c[0] for c in conditions])
        vals, n = self._strs([c[2] for c in conditions])
        ops = np.asarray([self.OPS[c[1]] for c in conditions] or [0], dtype=np.uint32)
        return cols, ops, vals, n

    def search(self, terms, not_terms=(), conditions=(), sort_by_score=False, descending=True, limit=100, offset=0):
        """One query with FILTER conditions [(column, op, literal)] -> (total, docs, scores)."""
        t, nt = self._strs(terms)
        x, nx = self._strs(not_terms)
        cols, ops, vals, nc = self._conds(list(conditions))
        total, n = C.c_uint64(), C.c_uint32()
        docs = np.zeros(max(limit, 1), np.uint32)
        scores = np.zeros(max(limit, 1), np.float64)
        _check(load().mgxs_search(self._h, nt, C.cast(t, C.c_void_p), nx, C.cast(x, C.c_void_p), nc,
                                  C.cast(cols, C.c_void_p), ops.ctypes.data, C.cast(vals, C.c_void_p), int(sort_by_score),
                                  int(descending), limit, offset, C.byref(total), C.byref(n), docs.ctypes.data,
                                  scores.ctypes.data))
        return int(total.value), docs[: n.value].copy(), scores[: n.value].copy()

    def facet(self, column, terms=(), not_terms=(), conditions=(), limit=100, offset=0):
        """ExecuteFacet -> (matched documents, total values, [(display value, count)] count-descending)."""
        t, nt = self._strs(terms)
        x, nx = self._strs(not_terms)
        cols, ops, vals, nc = self._conds(list(conditions))
        matched, total_values, n = C.c_uint64(), C.c_uint64(), C.c_uint32()
        counts = np.zeros(max(limit, 1), np.uint64)
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        off = np.zeros(max(limit, 1) + 1, np.uint32)
        _check(load().mgxs_facet(self._h, nt, C.cast(t, C.c_void_p), nx, C.cast(x, C.c_void_p), nc,
                                 C.cast(cols, C.c_void_p), ops.ctypes.data, C.cast(vals, C.c_void_p), column.encode(), limit,
                                 offset, C.byref(matched), C.byref(total_values), C.byref(n), counts.ctypes.data, buf, cap,
                                 off.ctypes.data))
        out = [(buf.raw[off[k]: off[k + 1]], int(counts[k])) for k in range(n.value)]
        return int(matched.value), int(total_values.value), out


class Executor:
    """search_pipeline::BatchExecutor: submit() plans + compiles + enqueues a fresh batch in C++, wait() fetches it."""

    def __init__(self, table, depth=2, planner_threads=4, comm=None):
        """comm: a dist.Comm (one rank of a doc-range-sharded table) — every batch's top-k is then all-gathered over RCCL
        and merged inside submit(); every rank submits the same batches in the same order."""
        self._table = table
        self._comm = comm  # kept alive
        h = C.c_void_p()
        if comm is None:
            _check(load().mgxs_executor_create(table._h, depth, planner_threads, C.byref(h)))
        else:
            _check(load().mgxs_executor_create_sharded(table._h, depth, planner_threads, comm._h, C.byref(h)))
        self._h = h
        self._shape = {}

    def __del__(self):
        if getattr(self, "_h", None):
            load().mgxs_executor_destroy(self._h)
            self._h = None

    def warm(self, qb, limit=10, offset=0, sort_by_score=True, descending=True, rounds=2):
        """BatchExecutor::Warm: setup outside any timed loop — every slot's arenas, pinned blocks, streams and helper
        threads are created by running `qb` through each slot `rounds` times; results are discarded."""
        _check(load().mgxs_executor_warm(self._h, qb.n, qb.n_terms.ctypes.data, C.cast(qb.terms, C.c_void_p), limit, offset,
                                         int(sort_by_score), int(descending), rounds))

    def submit(self, qb, limit=10, offset=0, sort_by_score=True, descending=True):
        t = C.c_uint64()
        _check(load().mgxs_submit(self._h, qb.n, qb.n_terms.ctypes.data, C.cast(qb.terms, C.c_void_p), limit, offset,
                                  int(sort_by_score), int(descending), C.byref(t)))
        self._shape[t.value] = (qb.n, limit)
        return t.value

    def wait(self, ticket, out=None):
        """-> (totals u64[n], n_docs u32[n], docs u32[n, limit], scores f64[n, limit], timing_ms f64[5]: plan, compile, enqueue, wait ms + device query count)."""
        n, limit = self._shape.pop(ticket)
        if out is None:
            out = (np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, max(limit, 1)), np.uint32),
                   np.zeros((n, max(limit, 1)), np.float64), np.zeros(5, np.float64))
        totals, n_docs, docs, scores, timing = out
        _check(load().mgxs_wait(self._h, ticket, totals.ctypes.data, n_docs.ctypes.data, docs.ctypes.data,
                                scores.ctypes.data, timing.ctypes.data))
        return out


class Batcher:
    """search_pipeline::MicroBatcher: search() is thread-safe and blocking (ctypes drops the GIL while it waits), queries
    from many threads share device batches."""

    def __init__(self, table, max_batch=1024, max_delay_us=200, depth=2, planner_threads=4):
        self._table = table
        h = C.c_void_p()
        _check(load().mgxs_batcher_create(table._h, max_batch, max_delay_us, depth, planner_threads, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            load().mgxs_batcher_destroy(self._h)
            self._h = None

    def search(self, terms, limit=10, offset=0, sort_by_score=True, descending=True):
        """-> (total, docs u32[n], scores f64[n])"""
        raw = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in terms]
        arr = (C.c_char_p * max(len(raw), 1))(*raw)
        total, n = C.c_uint64(), C.c_uint32()
        docs = np.zeros(max(limit, 1), dtype=np.uint32)
        scores = np.zeros(max(limit, 1), dtype=np.float64)
        _check(load().mgxs_batcher_search(self._h, len(raw), C.cast(arr, C.c_void_p), limit, offset, int(sort_by_score),
                                          int(descending), C.byref(total), C.byref(n), docs.ctypes.data,
                                          scores.ctypes.data))
        return int(total.value), docs[: n.value].copy(), scores[: n.value].copy()

    def stats(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(load().mgxs_batcher_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"batches": a.value, "queries": b.value, "closed_full": c.value, "closed_by_delay": d.value}
