"""Host-side mirror of the reference's operator interface over the C ABI of libmygram_gpu.so.

Names follow the reference (src/index/index.h, src/index/bm25_scorer.h, src/query/result_sorter.h,
src/server/search_pipeline.h) so the parity tests read like the reference's own tests. Everything that computes on
postings goes through the C ABI to the HIP kernels; what stays on the host is what the reference also does per query
on the host before touching posting lists: normalisation, n-gram generation, dictionary lookup, term ordering and
the empty/unknown-term rules.
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import check, load


# --------------------------------------------------------------------------------------------------------------------
# host-side string rules (src/utils/string_utils.cpp)
# --------------------------------------------------------------------------------------------------------------------

def _is_cjk_ideograph(cp):  # string_utils.cpp:441-448 — kana is NOT an ideograph here
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or
            0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0xF900 <= cp <= 0xFAFF)


def _generate_ngrams(text, n):  # string_utils.cpp:382-423
    if n <= 0 or len(text) < n:
        return []
    return [text[i:i + n] for i in range(len(text) - n + 1)]


def _generate_hybrid_ngrams(text, ascii_n, kanji_n, cross_boundary):  # string_utils.cpp:452-509
    out = []
    if ascii_n <= 0 or kanji_n <= 0:
        return out
    for i, ch in enumerate(text):
        cjk = _is_cjk_ideograph(ord(ch))
        n = kanji_n if cjk else ascii_n
        if i + n > len(text):
            continue
        if not cross_boundary and any(_is_cjk_ideograph(ord(c)) != cjk for c in text[i + 1:i + n]):
            continue
        out.append(text[i:i + n])
    return out


def generate_query_ngrams(normalized, ngram_size, kanji_ngram_size, cross_boundary=True):
    """GenerateQueryNgrams, string_utils.cpp:639-653 (returns str grams, in text order)."""
    if kanji_ngram_size > 0:
        return _generate_hybrid_ngrams(normalized, ngram_size if ngram_size > 0 else 2, kanji_ngram_size,
                                       cross_boundary)
    if ngram_size == 0:
        return _generate_hybrid_ngrams(normalized, 2, 1, True)
    return _generate_ngrams(normalized, ngram_size)


def normalize_text(text, nfkc=True, width="keep", lower=True):
    """mygram::utils::NormalizeText (string_utils.cpp:295-380) — the C++ host layer's implementation, so that the shim
    and this module normalise identically: ICU NFKC -> width -> lower when libmygram_shim.so was built with ICU
    (`normalize_uses_icu()`), otherwise the reference's non-ICU branch (ASCII lower-casing)."""
    from . import _shim_capi
    return _shim_capi.normalize_text(text, nfkc, width, lower)


def normalize_uses_icu():
    from . import _shim_capi
    return _shim_capi.normalize_uses_icu()


def _is_cjk_ideograph_pipeline(cp):
    """IsCjkIdeograph of src/server/search_pipeline.cpp:70-78."""
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF)


def has_uncovered_hybrid_fragment(normalized, ngram_size, kanji_ngram_size, cross_boundary):
    """HasUncoveredHybridFragment (src/server/search_pipeline.cpp:80-136): a term that mixes CJK ideographs with other
    code points and has a code point no query n-gram covers; such a query is post-filtered by exact text (:858-866)."""
    if not normalized or kanji_ngram_size <= 0:
        return False
    ascii_n = ngram_size if ngram_size > 0 else 2
    cps = [ord(c) for c in normalized]
    if len(cps) < 2:
        return False
    flags = [_is_cjk_ideograph_pipeline(c) for c in cps]
    if all(flags) or not any(flags):
        return False
    covered = [False] * len(cps)
    for i, start_is_cjk in enumerate(flags):
        n = kanji_ngram_size if start_is_cjk else ascii_n
        if n <= 0 or i + n > len(cps):
            continue
        if not cross_boundary and any(flags[i + j] != start_is_cjk for j in range(1, n)):
            continue
        for j in range(n):
            covered[i + j] = True
    return not all(covered)


def compute_idf(total_docs, doc_freq):
    """BM25Scorer::ComputeIDF, src/index/bm25_scorer.cpp:14-25 (host side: once per term per query)."""
    if total_docs == 0:
        return 0.0
    df = min(doc_freq, total_docs)
    return math.log((float(total_docs) - float(df) + 0.5) / (float(df) + 0.5) + 1.0)


def _np_view(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=int(count))


# --------------------------------------------------------------------------------------------------------------------
# corpus / columns / device index
# --------------------------------------------------------------------------------------------------------------------

class Corpus:
    """Normalized document texts as (text_bytes u8, text_off u64[n+1])."""

    def __init__(self, text_bytes, text_off, _owner=None):
        self.text_bytes, self.text_off, self._owner = text_bytes, text_off, _owner
        self.n_docs = len(text_off) - 1

    @classmethod
    def from_texts(cls, texts):
        bs = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        data = np.frombuffer(b"".join(bs) + b"\0" * 16, dtype=np.uint8).copy()
        return cls(data, off)

    @classmethod
    def synthetic(cls, n_docs, seed=42, global_first=0, n_threads=0):
        """The deterministic ASCII corpus of include/mygram_tools.h."""
        L = load()
        h = C.c_void_p()
        rc = L.mgxt_corpus_generate(seed, global_first, n_docs, n_threads, C.byref(h))
        if rc != 0:
            raise RuntimeError("mgxt_corpus_generate rc=%d" % rc)
        tb, to, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        L.mgxt_corpus_view(h, C.byref(tb), C.byref(to), C.byref(n))
        off = _np_view(to.value, n.value + 1, np.uint64)
        data = _np_view(tb.value, int(off[-1]) + 16, np.uint8)

        class _Owner:
            def __init__(self, handle):
                self.handle = handle

            def __del__(self):
                load().mgxt_corpus_destroy(self.handle)

        return cls(data, off, _Owner(h))

    def text(self, i):
        return bytes(self.text_bytes[int(self.text_off[i]):int(self.text_off[i + 1])])


class Columns:
    """mgx_columns: gram dictionary + CSR postings + tf + doc_len, built on the host by libmygram_gpu."""

    def __init__(self, corpus, first_doc_id=1, ngram_size=2, kanji_ngram_size=0, cross_boundary=True, n_threads=0):
        L = load()
        self.ngram_size, self.kanji_ngram_size, self.cross_boundary = ngram_size, kanji_ngram_size, cross_boundary
        bp = _capi.BuildParams(C.sizeof(_capi.BuildParams), _capi.ABI_VERSION, ngram_size, kanji_ngram_size,
                               int(cross_boundary), n_threads)
        h = C.c_void_p()
        check(L.mgx_columns_build(C.byref(bp), corpus.text_bytes.ctypes.data, corpus.text_off.ctypes.data,
                                  first_doc_id, corpus.n_docs, C.byref(h)))
        self._adopt(h)

    @classmethod
    def from_mgix(cls, data, first_doc_id=0, n_docs=0):
        """The reference's index dump (MGIX v1..v4, Index::SaveToStream) -> columns: doc ids only (tf / doc_len stay
        empty — BM25 needs the texts). n_docs=0: the doc range is the span of the ids in the dump."""
        L = load()
        raw = bytes(data)
        h, info = C.c_void_p(), _capi.MgixInfo()
        check(L.mgx_columns_from_mgix(raw, len(raw), first_doc_id, n_docs, C.byref(h), C.byref(info)))
        self = cls.__new__(cls)
        self.ngram_size, self.kanji_ngram_size = int(info.ngram_size), int(info.kanji_ngram_size)
        self.cross_boundary = bool(info.cross_boundary_ngrams)
        self.mgix = {"version": int(info.version), "normalize_nfkc": bool(info.normalize_nfkc),
                     "normalize_width": info.normalize_width.decode(), "normalize_lower": bool(info.normalize_lower),
                     "n_terms": int(info.n_terms)}
        self._adopt(h)
        return self

    def _adopt(self, h):
        L = load()
        self._h = h
        v = _capi.ColumnsView()
        check(L.mgx_columns_view_get(h, C.byref(v)))
        self.view = v
        self.n_grams = int(v.n_grams)
        self.n_postings = int(v.n_postings)
        self.first_doc_id = int(v.first_doc_id)
        self.n_docs = int(v.n_docs)
        self.bm25_doc_count = int(v.bm25_doc_count)
        self.bm25_total_len = int(v.bm25_total_len)
        self.key_off = _np_view(v.key_off, self.n_grams + 1, np.uint32)
        self.key_bytes = _np_view(v.key_bytes, int(self.key_off[-1]) if self.n_grams else 0, np.uint8)
        self.offsets = _np_view(v.offsets, self.n_grams + 1, np.uint64)
        self.docids = _np_view(v.docids, self.n_postings, np.uint32)
        self.tf = _np_view(v.tf, self.n_postings, np.uint8)
        self.doc_len = _np_view(v.doc_len, self.n_docs, np.uint32)
        self.tf_overflow_pos = _np_view(v.tf_overflow_pos, int(v.n_tf_overflow), np.uint64)
        self.tf_overflow_val = _np_view(v.tf_overflow_val, int(v.n_tf_overflow), np.uint32)

    def __del__(self):
        if getattr(self, "_h", None):
            load().mgx_columns_destroy(self._h)
            self._h = None

    def lookup(self, gram):
        """gram (str/bytes) -> gram id, or None when the gram is not in the index."""
        g = gram.encode("utf-8") if isinstance(gram, str) else bytes(gram)
        gid, found = C.c_uint32(), C.c_int()
        check(load().mgx_columns_lookup(self._h, g, len(g), C.byref(gid), C.byref(found)))
        return int(gid.value) if found.value else None

    def gram(self, gid):
        return bytes(self.key_bytes[int(self.key_off[gid]):int(self.key_off[gid + 1])])

    def avg_doc_length(self):  # BM25Stats::avg_doc_length, server_types.h:182-187
        return self.bm25_total_len / self.bm25_doc_count if self.bm25_doc_count else 0.0


class DeviceIndex:
    """mgx_index: one doc-range shard resident in HBM."""

    def __init__(self, columns, device=0, dense_threshold=0.0, with_scoring=True):
        L = load()
        v = columns.view
        d = _capi.IndexDesc(C.sizeof(_capi.IndexDesc), _capi.ABI_VERSION, device, 0, v.first_doc_id, v.n_docs,
                            v.n_grams, v.offsets, v.docids, v.tf if with_scoring else None,
                            v.doc_len if with_scoring else None, dense_threshold,
                            v.tf_overflow_pos if with_scoring else None, v.tf_overflow_val if with_scoring else None,
                            v.n_tf_overflow if with_scoring else 0)
        h = C.c_void_p()
        check(L.mgx_index_create(C.byref(d), C.byref(h)))
        self._h = h
        self.device = device

    def __del__(self):
        if getattr(self, "_h", None):
            load().mgx_index_destroy(self._h)
            self._h = None

    def memory_bytes(self):
        n = C.c_uint64()
        check(load().mgx_index_memory_bytes(self._h, C.byref(n)))
        return int(n.value)

    def set_batch_order(self, fifo=True):
        """mgx_index_set_batch_order: batches in flight run first-in-first-out (default) or time-slice the device."""
        check(load().mgx_index_set_batch_order(self._h, 0 if fifo else 1))

    def attach_text(self, corpus):
        """Normalized doc text into HBM (text-level BM25 terms: tf/df by text scan)."""
        check(load().mgx_index_attach_text(self._h, corpus.text_bytes.ctypes.data, corpus.text_off.ctypes.data))

    def add_filter_bitmap(self, docids):
        a = np.ascontiguousarray(docids, dtype=np.uint32)
        out = C.c_uint32()
        check(load().mgx_index_add_filter_bitmap(self._h, a.ctypes.data, len(a), C.byref(out)))
        return int(out.value)

    def add_filter_column(self, values, is_null=None, value_ids=None, n_values=0):
        """mgx_index_add_filter_column: a typed filter column by doc slot. `values`: int64 (signed class), uint64
        (unsigned class / string ranks) or float64 array."""
        v = np.ascontiguousarray(values)
        cls = {"i": 0, "u": 1, "f": 2}[v.dtype.kind]
        v = v.astype({0: np.int64, 1: np.uint64, 2: np.float64}[cls])
        keep = [v]
        nul = None
        if is_null is not None:
            nul = np.ascontiguousarray(is_null, dtype=np.uint8)
            keep.append(nul)
        ids = None
        if value_ids is not None:
            ids = np.ascontiguousarray(value_ids, dtype=np.uint32)
            keep.append(ids)
        d = _capi.FilterColumnDesc(C.sizeof(_capi.FilterColumnDesc), _capi.ABI_VERSION, cls, int(n_values), v.ctypes.data,
                                   nul.ctypes.data if nul is not None else None, ids.ctypes.data if ids is not None else None)
        out = C.c_uint32()
        check(load().mgx_index_add_filter_column(self._h, C.byref(d), C.byref(out)))
        return int(out.value)

    def filter_compare(self, column_id, op, literal, eq_epsilon=0.0, null_matches=False, never_matches=False):
        """mgx_index_filter_compare -> filter bitmap id. op: 0 = , 1 != , 2 < , 3 <= , 4 > , 5 >= ; literal: a python int or
        float in the column's class."""
        if isinstance(literal, float):
            bits = int(np.float64(literal).view(np.uint64))
        else:
            bits = int(literal) & 0xFFFFFFFFFFFFFFFF
        out = C.c_uint32()
        check(load().mgx_index_filter_compare(self._h, column_id, op, bits, float(eq_epsilon), int(null_matches),
                                              int(never_matches), C.byref(out)))
        return int(out.value)


# --------------------------------------------------------------------------------------------------------------------
# queries
# --------------------------------------------------------------------------------------------------------------------

class Query:
    """The parts of query::Query (src/query/query_parser.h:207-243) the hot path consumes."""

    def __init__(self, terms=(), not_terms=(), filters=(), sort_score=False, limit=100, offset=0, descending=True,
                 k1=1.2, b=0.75, fuzzy=0, expr=None, universe=None, verify_text=False):
        """`expr`: boolean tree over term strings instead of the plain AND of `terms` (query::QueryNode):
        "term" | ("and", e, ...) | ("or", e, ...) | ("not", e). `universe` = (first_doc_id, count) of the NOT universe
        (DocumentStore::GetAllDocIds); None = every slot of the index."""
        self.terms, self.not_terms, self.filters = list(terms), list(not_terms), list(filters)
        self.sort_score, self.limit, self.offset, self.descending = sort_score, limit, offset, descending
        self.k1, self.b, self.fuzzy = k1, b, fuzzy
        # verify_text: the caller's memory.verify_text decision (ShouldApplyVerifyText, search_pipeline.cpp:42-66);
        # queries with an uncovered mixed-script fragment are post-filtered by exact text regardless (:858-866)
        self.verify_text = verify_text
        self.expr, self.universe = expr, universe
        if expr is not None and not self.terms:
            seen = []

            def walk(e):
                if isinstance(e, str):
                    if e not in seen:
                        seen.append(e)
                else:
                    for c in e[1:]:
                        walk(c)
            walk(expr)
            self.terms = seen


class TermInfo:
    """search_pipeline::SearchTermInfo (src/server/search_pipeline.h:44-55)."""
    __slots__ = ("term", "normalized", "grams", "gram_ids", "estimated_size", "df", "threshold", "fuzzy_ids",
                 "fuzzy_empty")


class SearchResult:
    __slots__ = ("total", "docs", "scores", "total_candidates", "after_intersection", "after_not", "after_filters",
                 "empty_term_detected", "term_order")

    def __init__(self):
        self.total = 0
        self.docs = np.zeros(0, np.uint32)
        self.scores = np.zeros(0, np.float64)
        self.total_candidates = self.after_intersection = self.after_not = self.after_filters = 0
        self.empty_term_detected = False
        self.term_order = []


def _take_u32(ptr, n):
    out = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[: int(n)].copy() if n else np.zeros(0, np.uint32)
    load().mgx_free(ptr)
    return out.astype(np.uint32)


class PreparedBatch:
    """mgx_batch: a compiled, device-resident batch of queries that can be executed repeatedly."""

    def __init__(self, index, cqueries, keep, shells, orders):
        self.index, self._keep, self._shells, self._orders = index, keep, shells, orders
        self.n = len(cqueries)
        self._h = None
        if self.n:
            h = C.c_void_p()
            arr = (_capi.Query * self.n)(*cqueries)
            check(load().mgx_batch_prepare(index.device_index._h, arr, self.n, C.byref(h)))
            self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            load().mgx_batch_destroy(self._h)
            self._h = None

    def reset(self, cqueries, keep, shells, orders):
        """mgx_batch_reset: the same batch object (device arenas, pinned result block) for a new set of queries."""
        self._keep, self._shells, self._orders = keep, shells, orders
        self.n = len(cqueries)
        self._exchange = self._df_view = None  # (views of the old contents' device memory)
        arr = (_capi.Query * max(self.n, 1))(*cqueries)
        if self._h is None:
            h = C.c_void_p()
            check(load().mgx_batch_prepare(self.index.device_index._h, arr, self.n, C.byref(h)))
            self._h = h
        else:
            check(load().mgx_batch_reset(self._h, arr, self.n))

    def execute(self, stream=None):
        if self._h:
            check(load().mgx_batch_execute(self._h, stream))

    def execute_sharded(self, comm, stream=None):
        """mgx_batch_execute_sharded: execute on a shard of a table — the seeds' best keys of every rank are all-gathered
        (RCCL) and raise each query's pruning bound before the rest of the batch runs; follow with exchange()."""
        if self._h:
            check(load().mgx_batch_execute_sharded(self._h, comm._h, stream))

    def execute_gather(self, world, gather, stream=None):
        """mgx_batch_execute_gather: the same with the caller's all-gather, gather(mine_ptr, all_ptr, nbytes) -> None
        (device pointers; rank r's block goes to all_ptr + r * nbytes)."""
        if not self._h:
            return
        failure = []

        def _cb(_user, mine, all_, nbytes, _stream):
            try:
                gather(mine, all_, nbytes)
                return 0
            except BaseException as e:  # noqa: BLE001 (re-raised below: no exception may cross the C frame)
                failure.append(e)
                return 7
        fn = _capi.GATHER_FN(_cb)
        rc = load().mgx_batch_execute_gather(self._h, world, fn, None, stream)
        if failure:
            raise failure[0]
        check(rc)

    def fetch(self):
        """-> list[SearchResult] in the order the queries were given (host-resolved queries included)."""
        v = _capi.ResultView()
        if self._h:
            check(load().mgx_batch_fetch(self._h, C.byref(v)))
        out = []
        k = 0
        for qi, shell in enumerate(self._shells):
            if shell is not None:  # resolved on the host (empty / unknown term): never reached the device
                out.append(shell)
                continue
            r = v.queries[k]
            k += 1
            s = SearchResult()
            s.term_order = self._orders[qi]
            s.total = int(r.total)
            s.total_candidates, s.after_intersection = int(r.total_candidates), int(r.after_intersection)
            s.after_not, s.after_filters = int(r.after_not), int(r.after_filters)
            if r.n_docs:
                s.docs = np.ctypeslib.as_array(v.docs, shape=(r.docs_begin + r.n_docs,))[r.docs_begin:].copy()
                s.scores = np.ctypeslib.as_array(v.scores, shape=(r.docs_begin + r.n_docs,))[r.docs_begin:].copy()
            out.append(s)
        return out

    def kernel_time_ms(self):
        ms, n = C.c_double(), C.c_uint32()
        check(load().mgx_batch_kernel_time_ms(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def algorithmic_bytes(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(load().mgx_batch_algorithmic_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def fetch_raw(self):
        """mgx_batch_fetch without building Python objects (the timed path of bench.py)."""
        v = _capi.ResultView()
        check(load().mgx_batch_fetch(self._h, C.byref(v)))
        return v

    def exchange_df(self, comm, stream=None):
        """mgx_batch_exchange_df: table-wide df of the text-level terms (df pass + RCCL all-reduce); before execute."""
        check(load().mgx_batch_exchange_df(self._h, comm._h, stream))

    def exchange(self, comm, stream=None):
        """mgx_batch_exchange: RCCL all-gather of every shard's top-k + merge; after execute, same stream."""
        check(load().mgx_batch_exchange(self._h, comm._h, stream))

    def count_df(self, stream=None):
        """df pass of the text-level terms only (sharded tables reduce the counts before execute)."""
        check(load().mgx_batch_count_df(self._h, stream))

    def df_buffer(self):
        """-> (device pointer of the u64 df counts, number of text-level terms)."""
        p, n = C.c_void_p(), C.c_uint32()
        check(load().mgx_batch_df_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, int(n.value)

    def topk_stride(self):
        s = C.c_uint32()
        check(load().mgx_batch_export_topk(self._h, None, None, C.byref(s), None))
        return int(s.value)

    def export_topk(self, blob64_ptr, blob32_ptr, stream=None):
        s = C.c_uint32()
        check(load().mgx_batch_export_topk(self._h, blob64_ptr, blob32_ptr, C.byref(s), stream))
        return int(s.value)

    def export_buffer(self):
        """-> (device pointer, bytes, byte offset of the 32-bit part) of the library-owned exchange blob, or
        (None, 0, 0) when the batch has none (docid-ordered pages export by copy)."""
        p, nb, off = C.c_void_p(), C.c_uint64(), C.c_uint64()
        check(load().mgx_batch_export_buffer(self._h, C.byref(p), C.byref(nb), C.byref(off)))
        return p.value, int(nb.value), int(off.value)

    def merge_shards(self, n_shards, blob64_ptr, blob32_ptr, stream=None, pitch64=0, pitch32=0):
        check(load().mgx_batch_merge_shards(self._h, n_shards, blob64_ptr, pitch64, blob32_ptr, pitch32, stream))


class Index:
    """Read side of mygramdb::index::Index (src/index/index.h:46-300) over one device shard.

    `total_docs` / `avg_doc_length` are the table's BM25Stats (global over all shards when sharded)."""

    def __init__(self, corpus=None, texts=None, first_doc_id=1, ngram_size=2, kanji_ngram_size=0, cross_boundary=True,
                 device=0, dense_threshold=0.0, total_docs=None, avg_doc_length=None, global_posting_sizes=None,
                 n_threads=0, normalize_nfkc=True, normalize_width="keep", normalize_lower=True):
        if corpus is None:
            corpus = Corpus.from_texts(texts if texts is not None else [])
        self.ngram_size = ngram_size
        # the Index object keeps kanji = ngram when 0 (index.cpp:31); query n-grams use the table config value
        self.kanji_ngram_size = kanji_ngram_size
        self.cross_boundary = cross_boundary
        # index.h:58-60: how QUERY terms are normalised; document texts arrive normalised (as Index::AddDocument's do)
        self.normalize_nfkc, self.normalize_width, self.normalize_lower = normalize_nfkc, normalize_width, normalize_lower
        self.columns = Columns(corpus, first_doc_id, ngram_size, kanji_ngram_size, cross_boundary, n_threads)
        self.device_index = DeviceIndex(self.columns, device, dense_threshold)
        self.corpus = corpus        # the DocumentStore's normalized texts
        self._text_attached = False
        self.total_docs = self.columns.bm25_doc_count if total_docs is None else total_docs
        self.avg_doc_length = self.columns.avg_doc_length() if avg_doc_length is None else avg_doc_length
        # df source: posting sizes of the WHOLE table (a shard passes the global sizes so idf is identical everywhere)
        self._global_sizes = global_posting_sizes
        self._global_dict = None  # sharded tables: {gram bytes: table-wide posting size} over every shard's dictionary

    @classmethod
    def from_mgix(cls, data, device=0, dense_threshold=0.0, first_doc_id=0, n_docs=0):
        """Index::LoadFromData (src/index/index_serialization.cpp:260-420): an index over the postings of a reference
        dump, configured by the dump's own header. It holds doc ids only, so it answers everything but SORT _score."""
        self = cls.__new__(cls)
        cols = Columns.from_mgix(data, first_doc_id, n_docs)
        self.ngram_size, self.kanji_ngram_size, self.cross_boundary = cols.ngram_size, cols.kanji_ngram_size, cols.cross_boundary
        self.normalize_nfkc, self.normalize_width = cols.mgix["normalize_nfkc"], cols.mgix["normalize_width"]
        self.normalize_lower = cols.mgix["normalize_lower"]
        self.columns = cols
        self.device_index = DeviceIndex(cols, device, dense_threshold, with_scoring=False)
        self.corpus = None
        self._text_attached = False
        self.total_docs, self.avg_doc_length = 0, 0.0
        self._global_sizes = None
        self._global_dict = None
        return self

    @classmethod
    def from_dump(cls, data, table=None, device=0, dense_threshold=0.0):
        """A table of a reference dump (DUMP SAVE, "MGDB" v2: src/storage/dump_format_v2.cpp) as a ranking index: columns
        built from the dump's stored texts (and checked against its own MGIX index), texts attached, filter columns kept in
        `dump_filter_columns` [(name, value type 1..12, values, is_null, strings)]; `exists` marks the doc ids the store
        holds."""
        L = load()
        raw = bytes(data)
        h = C.c_void_p()
        check(L.mgx_dump_open(raw, len(raw), table.encode() if table else None, C.byref(h)))
        try:
            v = _capi.DumpView()
            check(L.mgx_dump_view_get(h, C.byref(v)))
            n = int(v.n_docs)
            info = v.index_info
            self = cls.__new__(cls)
            self.table_name = v.table_name.decode()
            self.ngram_size, self.kanji_ngram_size = int(info.ngram_size), int(info.kanji_ngram_size)
            self.cross_boundary = bool(info.cross_boundary_ngrams)
            self.normalize_nfkc, self.normalize_lower = bool(info.normalize_nfkc), bool(info.normalize_lower)
            self.normalize_width = info.normalize_width.decode()
            self.exists = np.ctypeslib.as_array(C.cast(v.exists, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy()
            toff = np.ctypeslib.as_array(C.cast(v.text_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            tb = np.ctypeslib.as_array(C.cast(v.text_bytes, C.POINTER(C.c_uint8)), shape=(int(toff[-1]) + 16,)).copy()
            self.corpus = Corpus(tb, toff)
            self.dump_filter_columns = []
            for i in range(int(v.n_filter_columns)):
                fc = _capi.DumpFilterColumn()
                check(L.mgx_dump_filter_column_get(h, i, C.byref(fc)))
                vals = np.ctypeslib.as_array(C.cast(fc.values, C.POINTER(C.c_uint64)), shape=(max(n, 1),))[:n].copy()
                nul = np.ctypeslib.as_array(C.cast(fc.is_null, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy()
                strings = None
                if fc.value_type == 11:
                    so = np.ctypeslib.as_array(C.cast(fc.string_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
                    sb = np.ctypeslib.as_array(C.cast(fc.string_bytes, C.POINTER(C.c_uint8)), shape=(int(so[-1]) + 1,)).tobytes()
                    strings = [sb[int(so[k]): int(so[k + 1])] for k in range(n)]
                self.dump_filter_columns.append((fc.name.decode(), int(fc.value_type), vals, nul.astype(bool), strings))
            ch = C.c_void_p()
            check(L.mgx_dump_take_columns(h, C.byref(ch)))
            cols = Columns.__new__(Columns)
            cols.ngram_size, cols.kanji_ngram_size, cols.cross_boundary = self.ngram_size, self.kanji_ngram_size, self.cross_boundary
            cols._adopt(ch)
            has_texts = bool(v.has_texts)
        finally:
            L.mgx_dump_destroy(h)
        self.columns = cols
        self.device_index = DeviceIndex(cols, device, dense_threshold, with_scoring=has_texts)
        self._text_attached = False
        self.total_docs = cols.bm25_doc_count if has_texts else 0
        self.avg_doc_length = cols.avg_doc_length() if has_texts else 0.0
        self._global_sizes = None
        self._global_dict = None
        return self

    # ---- dictionary -------------------------------------------------------------------------------------------
    def posting_size(self, gram):
        if self._global_dict is not None:
            g = gram.encode("utf-8") if isinstance(gram, str) else bytes(gram)
            return int(self._global_dict.get(g, 0))
        gid = self.columns.lookup(gram)
        if gid is None:
            return 0
        if self._global_sizes is not None:
            return int(self._global_sizes[gid])
        return int(self.columns.offsets[gid + 1] - self.columns.offsets[gid])

    estimate_posting_size = posting_size  # index.cpp:756-759

    def ensure_text(self):
        """Text-level BM25 terms scan doc text on the device: upload it on first use."""
        if not self._text_attached:
            self.device_index.attach_text(self.corpus)
            self._text_attached = True

    def _ids(self, terms):
        """gram strings -> (ids of known grams, saw_unknown)."""
        ids, unknown = [], False
        for t in terms:
            gid = self.columns.lookup(t)
            if gid is None:
                unknown = True
            else:
                ids.append(gid)
        return ids, unknown

    # ---- Index::Search* ---------------------------------------------------------------------------------------
    def search_and(self, terms, limit=0, reverse=False):
        """Index::SearchAnd, index.cpp:199-368."""
        if len(terms) == 0:
            return np.zeros(0, np.uint32)  # :203
        ids, unknown = self._ids(terms)
        if unknown:
            return np.zeros(0, np.uint32)  # :211-215
        a = np.asarray(ids, dtype=np.uint32)
        p, n = C.POINTER(C.c_uint32)(), C.c_uint64()
        check(load().mgx_and(self.device_index._h, a.ctypes.data, len(a), limit, int(reverse), C.byref(p),
                             C.byref(n)))
        return _take_u32(p, n.value)

    def search_or(self, terms):
        """Index::SearchOr, index.cpp:418-448 (unknown terms skipped)."""
        ids, _ = self._ids(terms)
        if not ids:
            return np.zeros(0, np.uint32)
        a = np.asarray(sorted(set(ids)), dtype=np.uint32)
        p, n = C.POINTER(C.c_uint32)(), C.c_uint64()
        check(load().mgx_or(self.device_index._h, a.ctypes.data, len(a), C.byref(p), C.byref(n)))
        return _take_u32(p, n.value)

    def search_not(self, all_docs, terms):
        """Index::SearchNot, index.cpp:450-486."""
        docs = np.ascontiguousarray(all_docs, dtype=np.uint32)
        ids, _ = self._ids(terms)
        if len(terms) == 0 or not ids:
            return docs.copy()  # :451-453 / nothing to exclude
        a = np.asarray(sorted(set(ids)), dtype=np.uint32)
        p, n = C.POINTER(C.c_uint32)(), C.c_uint64()
        check(load().mgx_not(self.device_index._h, docs.ctypes.data, len(docs), a.ctypes.data, len(a), C.byref(p),
                             C.byref(n)))
        return _take_u32(p, n.value)

    def search_by_threshold(self, terms, threshold):
        """Index::SearchByThreshold, index.cpp:488-578."""
        if len(terms) == 0 or threshold == 0:
            return np.zeros(0, np.uint32)
        uniq = sorted(set(t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in terms))  # :496-497
        if threshold > len(uniq):
            return np.zeros(0, np.uint32)
        if threshold == len(uniq):
            return self.search_and(uniq)  # :504-506
        ids, _ = self._ids(uniq)  # missing grams do not count (:512-518)
        if len(ids) < threshold:
            return np.zeros(0, np.uint32)
        a = np.asarray(ids, dtype=np.uint32)
        p, n = C.POINTER(C.c_uint32)(), C.c_uint64()
        check(load().mgx_threshold(self.device_index._h, a.ctypes.data, len(a), threshold, C.byref(p), C.byref(n)))
        return _take_u32(p, n.value)

    def filter_by_ngrams(self, candidates, terms):
        """Index::FilterByNgrams, index.cpp:370-416 (caller order and duplicates kept)."""
        cand = np.ascontiguousarray(candidates, dtype=np.uint32)
        if len(cand) == 0:
            return np.zeros(0, np.uint32)
        if len(terms) == 0:
            return cand.copy()
        ids, unknown = self._ids(terms)
        if unknown:
            return np.zeros(0, np.uint32)
        a = np.asarray(ids, dtype=np.uint32)
        p, n = C.POINTER(C.c_uint32)(), C.c_uint64()
        check(load().mgx_retain(self.device_index._h, cand.ctypes.data, len(cand), a.ctypes.data, len(a),
                                C.byref(p), C.byref(n)))
        return _take_u32(p, n.value)

    # ---- BM25Scorer / ResultSorter ----------------------------------------------------------------------------
    def score_documents(self, candidates, terms, dfs, total_docs, avg_doc_length, k1=1.2, b=0.75):
        """BM25Scorer::ScoreDocuments (bm25_scorer.cpp:47-99): tf columns when every term is exactly one n-gram, the
        doc text otherwise."""
        if len(terms) != len(dfs):
            raise _capi.MgxError(2, "BM25 search_terms and term_doc_freqs must have identical lengths")
        cand = np.ascontiguousarray(candidates, dtype=np.uint32)
        tis = [self.term_info(t) for t in terms]
        if any(len(ti.grams) != 1 or ti.grams[0] != ti.normalized.encode("utf-8") for ti in tis):
            # some term is not one n-gram: count every term in the doc text, like the reference
            self.ensure_text()
            tb = [ti.normalized.encode("utf-8") for ti in tis]
            off = np.zeros(len(tb) + 1, dtype=np.uint32)
            off[1:] = np.cumsum([len(x) for x in tb])
            data = np.frombuffer(b"".join(tb) + b"\0" * 16, dtype=np.uint8).copy()
            idfs = np.asarray([compute_idf(total_docs, d) for d in dfs], dtype=np.float64)
            out = np.zeros(max(len(cand), 1), dtype=np.float64)
            check(load().mgx_score_documents_text(self.device_index._h, cand.ctypes.data, len(cand), data.ctypes.data,
                                                  off.ctypes.data, idfs.ctypes.data, len(tb), float(avg_doc_length),
                                                  float(k1), float(b), out.ctypes.data))
            return out[: len(cand)]
        gids = []
        for t in terms:
            gid = self.columns.lookup(t)
            gids.append(0xFFFFFFFF if gid is None else gid)
        g = np.asarray(gids, dtype=np.uint32)
        idfs = np.asarray([compute_idf(total_docs, d) for d in dfs], dtype=np.float64)
        out = np.zeros(max(len(cand), 1), dtype=np.float64)
        check(load().mgx_score_documents(self.device_index._h, cand.ctypes.data, len(cand), g.ctypes.data,
                                         idfs.ctypes.data, len(g), float(avg_doc_length), float(k1), float(b),
                                         out.ctypes.data))
        return out[: len(cand)]

    def sort_by_score(self, results, scores, descending=True, limit=0, offset=0):
        """ResultSorter::SortByScore, result_sorter.cpp:661-716."""
        r = np.ascontiguousarray(results, dtype=np.uint32)
        s = np.ascontiguousarray(scores, dtype=np.float64)
        p, n = C.POINTER(C.c_uint32)(), C.c_uint64()
        check(load().mgx_sort_by_score(self.device_index._h, r.ctypes.data, s.ctypes.data, len(r), int(descending),
                                       limit, offset, C.byref(p), C.byref(n)))
        return _take_u32(p, n.value)

    # ---- search_pipeline ---------------------------------------------------------------------------------------
    def term_info(self, term, fuzzy=0):
        """GenerateTermInfos for one term (search_pipeline.cpp:569-603); df from posting sizes (single-gram terms)."""
        ti = TermInfo()
        ti.term = term
        ti.normalized = normalize_text(term, self.normalize_nfkc, self.normalize_width, self.normalize_lower)
        grams = generate_query_ngrams(ti.normalized, self.ngram_size, self.kanji_ngram_size, self.cross_boundary)
        ti.grams = sorted(set(g.encode("utf-8") for g in grams))  # DeduplicateSorted: bytewise
        ti.gram_ids = []
        size = None  # SIZE_MAX
        for g in ti.grams:
            ps = self.posting_size(g)
            if ps > 0:
                size = ps if size is None else min(size, ps)
                gid = self.columns.lookup(g)  # None: another shard has the gram, this one holds no posting for it
                ti.gram_ids.append(_capi.GRAM_ABSENT if gid is None else gid)
            else:
                size = 0
                break
        ti.estimated_size = size
        # a term that IS its single n-gram scores from the gram's tf column (df = posting count); any other term
        # needs the text: tf by CountTermOccurrences, df by PopulateTermDocumentFrequency (search_pipeline.cpp:542-565)
        ti.df = size if (len(ti.grams) == 1 and size and ti.grams[0] == ti.normalized.encode("utf-8")) else None
        ti.threshold = 0
        if fuzzy and ti.grams:
            # ExecuteWithFuzzy, search_pipeline.cpp:1685-1703: theta = |grams| - d * n_eff, at least 1; n_eff = the kanji
            # size when more than half of the grams are at most 3 bytes long
            n_eff = self.ngram_size if self.ngram_size > 0 else 2
            if self.kanji_ngram_size > 0 and sum(1 for g in ti.grams if len(g) <= 3) > len(ti.grams) // 2:
                n_eff = self.kanji_ngram_size
            drop = fuzzy * n_eff
            ti.threshold = len(ti.grams) - drop if len(ti.grams) > drop else 1
            # Index::SearchByThreshold (index.cpp:488-578) over the DISTINCT grams: theta == all of them -> SearchAnd
            # (an unknown gram empties it); otherwise unknown grams are skipped and fewer than theta known ones -> {}
            known = [g for g in ti.grams if self.posting_size(g) > 0]
            ti.fuzzy_ids = [(_capi.GRAM_ABSENT if self.columns.lookup(g) is None else self.columns.lookup(g)) for g in known]
            if ti.threshold == len(ti.grams):
                ti.fuzzy_empty = len(known) != len(ti.grams)
            else:
                ti.fuzzy_empty = len(known) < ti.threshold
        return ti

    def prepare(self, queries, into=None):
        """Compiles queries the way ExecuteFullPipeline's regular branch plans them (search_pipeline.cpp:2002-2030,
        Execute :795-869) and uploads them as one batch. `into`: a PreparedBatch to re-use (mgx_batch_reset)."""
        cqueries, keep, shells, orders = [], [], [], []
        for q in queries:
            tis = [self.term_info(t, q.fuzzy) for t in q.terms]
            if q.fuzzy:
                # ExecuteWithFuzzy (search_pipeline.cpp:1659-1744): terms in the order given (no size sort); a term
                # without n-grams ends the query; per term "at least theta of its known grams", AND across terms
                order = list(range(len(tis)))
                shell = SearchResult()
                shell.term_order = order
                orders.append(order)
                if not tis or any(not t.grams for t in tis):
                    shell.empty_term_detected = True
                    shells.append(shell)
                    continue
                cterms = (_capi.Term * len(tis))()
                scored = []
                for j, t in enumerate(tis):
                    ids_list = [_capi.GRAM_ABSENT] if (t.fuzzy_empty or not t.fuzzy_ids) else t.fuzzy_ids
                    ids = np.asarray(ids_list, dtype=np.uint32)
                    keep.append(ids)
                    thr = 0 if (t.fuzzy_empty or t.threshold >= len(ids)) else t.threshold
                    idf, text_ptr, text_len = 0.0, None, 0
                    # SORT _score: the EXACT term is scored (search_handler.cpp:428-456 over GenerateTermInfos' infos,
                    # search_pipeline.cpp:1895-1899). A term with an unknown n-gram occurs in no text: tf = df = 0, it
                    # adds nothing and is left out of the scored list.
                    if q.sort_score and len(t.fuzzy_ids) == len(t.grams) and not t.fuzzy_empty:
                        scored.append(j)
                        if t.df is None:  # text-level term
                            self.ensure_text()
                            tb = np.frombuffer(t.normalized.encode("utf-8"), dtype=np.uint8).copy()
                            keep.append(tb)
                            text_ptr, text_len = tb.ctypes.data, len(tb)
                        else:
                            idf = compute_idf(self.total_docs, t.df)
                    cterms[j] = _capi.Term(ids.ctypes.data, len(ids), thr, idf, text_ptr, text_len)
                nts = []
                for nt in q.not_terms:
                    ti = self.term_info(nt)
                    if not ti.grams:
                        raise _capi.MgxError(4, "NOT term shorter than one n-gram is not on the device path")
                    if ti.estimated_size == 0:
                        continue
                    nts.append(ti)
                cnots = (_capi.Term * max(len(nts), 1))()
                for j, t in enumerate(nts):
                    ids = np.asarray(t.gram_ids, dtype=np.uint32)
                    keep.append(ids)
                    cnots[j] = _capi.Term(ids.ctypes.data, len(ids), 0, 0.0, None, 0)
                cf = (_capi.Filter * max(len(q.filters), 1))()
                for j, (bid, negate) in enumerate(q.filters):
                    cf[j] = _capi.Filter(bid, int(negate))
                keep.extend([cterms, cnots, cf])
                cq = _capi.Query(C.cast(cterms, C.c_void_p), len(tis), C.cast(cnots, C.c_void_p), len(nts),
                                 C.cast(cf, C.c_void_p), len(q.filters),
                                 _capi.SORT_SCORE if q.sort_score else _capi.SORT_DOCID, q.limit, q.offset,
                                 int(q.descending), q.k1, q.b, self.total_docs, self.avg_doc_length)
                if q.sort_score:
                    sc = np.asarray(scored, dtype=np.uint32)
                    keep.append(sc)
                    cq.score_terms = sc.ctypes.data if len(sc) else C.cast(cterms, C.c_void_p).value  # (non-NULL: "none")
                    cq.n_score_terms = len(sc)
                cqueries.append(cq)
                shells.append(None)
                continue
            if q.expr is not None:
                order = list(range(len(tis)))  # expression leaves keep their first-use order
            else:
                order = sorted(range(len(tis)), key=lambda i: (float("inf") if tis[i].estimated_size is None
                                                               else tis[i].estimated_size))  # stable, :2012-2014
            tis = [tis[i] for i in order]
            shell = SearchResult()
            shell.term_order = order
            orders.append(order)
            expr_tokens = None
            if q.expr is not None:
                # ExecuteWithBooleanAst (search_pipeline.cpp:1408-1578): a term with an unknown gram is an empty
                # doc set inside the tree, not the end of the query
                toks = []

                def emit(e):
                    if isinstance(e, str):
                        i = q.terms.index(e)
                        toks.append((_capi.EXPR_EMPTY, 0) if tis[i].estimated_size in (0, None)
                                    else (_capi.EXPR_TERM, i))
                        return
                    for c in e[1:]:
                        emit(c)
                    op = {"and": _capi.EXPR_AND, "or": _capi.EXPR_OR, "not": _capi.EXPR_NOT}[e[0]]
                    toks.append((op, len(e) - 1))
                emit(q.expr)
                expr_tokens = (_capi.ExprToken * len(toks))(*[_capi.ExprToken(o, a) for o, a in toks])
                keep.append(expr_tokens)
                # SORT _score over an expression: the TERM leaves that are not under a NOT, in tree order, repeats kept
                # (CollectAstScoringTerms, search_pipeline.cpp:232-254); a leaf with an unknown gram occurs in no text
                # (tf = df = 0) and is left out
                expr_scored = []

                def collect(e, under_not):
                    if isinstance(e, str):
                        i = q.terms.index(e)
                        if not under_not and tis[i].estimated_size not in (0, None):
                            expr_scored.append(i)
                        return
                    for c in e[1:]:
                        collect(c, under_not or e[0] == "not")
                collect(q.expr, False)
                # placeholder grams for EMPTY leaves so that every mgx_term stays well-formed
                for t in tis:
                    if t.estimated_size in (0, None):
                        t.gram_ids = [0]
            elif any(t.estimated_size in (0, None) and (t.grams or not t.normalized) for t in tis):
                shell.empty_term_detected = True  # Execute :804-810
                shells.append(shell)
                continue
            if any(not t.grams for t in tis) and (q.expr is not None or q.fuzzy):
                raise _capi.MgxError(4, "a term shorter than one n-gram inside an expression / FUZZY query "
                                        "(SearchNormalizedSubstring) is not on the device path")
            exact = q.expr is None and (q.verify_text or any(
                has_uncovered_hybrid_fragment(t.normalized, self.ngram_size, self.kanji_ngram_size, self.cross_boundary)
                for t in tis))
            if exact:
                self.ensure_text()
            cterms = (_capi.Term * len(tis))()
            for j, t in enumerate(tis):
                ids = np.asarray(t.gram_ids, dtype=np.uint32)
                keep.append(ids)
                thr = t.threshold if (q.fuzzy and t.threshold < len(ids)) else 0
                idf, text_ptr, text_len = 0.0, None, 0
                if exact or not t.grams:
                    # (no grams at all: a term shorter than one n-gram — the device scans the texts for it,
                    # query::SearchNormalizedSubstring, src/query/substring_search.h:24-42)
                    self.ensure_text()
                    tb = np.frombuffer(t.normalized.encode("utf-8"), dtype=np.uint8).copy()
                    keep.append(tb)
                    text_ptr, text_len = tb.ctypes.data, len(tb)
                if q.sort_score:
                    if t.df is None:  # text-level term
                        self.ensure_text()
                        tb = np.frombuffer(t.normalized.encode("utf-8"), dtype=np.uint8).copy()
                        keep.append(tb)
                        text_ptr, text_len = tb.ctypes.data, len(tb)
                    else:
                        idf = compute_idf(self.total_docs, t.df)
                cterms[j] = _capi.Term(ids.ctypes.data if len(ids) else None, len(ids), thr, idf, text_ptr, text_len)
            nts = []
            for nt in q.not_terms:
                ti = self.term_info(nt)
                if not ti.grams:
                    if not ti.normalized:
                        continue  # "" is contained in nothing (SearchNormalizedSubstring returns {})
                    self.ensure_text()
                elif ti.estimated_size == 0:
                    continue  # an unknown gram: the NOT term matches nothing
                nts.append(ti)
            cnots = (_capi.Term * max(len(nts), 1))()
            for j, t in enumerate(nts):
                ids = np.asarray(t.gram_ids, dtype=np.uint32)
                keep.append(ids)
                text_ptr, text_len = None, 0
                if not t.grams:  # substring NOT term
                    tb = np.frombuffer(t.normalized.encode("utf-8"), dtype=np.uint8).copy()
                    keep.append(tb)
                    text_ptr, text_len = tb.ctypes.data, len(tb)
                cnots[j] = _capi.Term(ids.ctypes.data if len(ids) else None, len(ids), 0, 0.0, text_ptr, text_len)
            cf = (_capi.Filter * max(len(q.filters), 1))()
            for j, (bid, negate) in enumerate(q.filters):
                cf[j] = _capi.Filter(bid, int(negate))
            keep.extend([cterms, cnots, cf])
            cq = _capi.Query(C.cast(cterms, C.c_void_p), len(tis), C.cast(cnots, C.c_void_p), len(nts),
                             C.cast(cf, C.c_void_p), len(q.filters),
                             _capi.SORT_SCORE if q.sort_score else _capi.SORT_DOCID, q.limit, q.offset,
                             int(q.descending), q.k1, q.b, self.total_docs, self.avg_doc_length)
            cq.exact_text = int(exact)
            if expr_tokens is not None and q.sort_score:
                sc = np.asarray(expr_scored, dtype=np.uint32)
                keep.append(sc)
                cq.score_terms = sc.ctypes.data if len(sc) else C.cast(cterms, C.c_void_p).value  # (non-NULL: "none")
                cq.n_score_terms = len(sc)
            if expr_tokens is not None:
                cq.expr = C.cast(expr_tokens, C.c_void_p)
                cq.n_expr = len(expr_tokens)
                if q.universe is not None:
                    cq.universe_first, cq.universe_count = q.universe
            cqueries.append(cq)
            shells.append(None)
        if into == "raw":  # (facet: the compiled C queries themselves)
            return cqueries, keep, shells
        if into is not None:
            into.reset(cqueries, keep, shells, orders)
            return into
        return PreparedBatch(self, cqueries, keep, shells, orders)

    def facet_counts(self, query, column_id, n_values):
        """mgx_facet_counts: (matched documents, counts per value id) of the query's whole result set."""
        cqueries, keep, shells = self.prepare([query], into="raw")
        if shells[0] is not None:  # resolved on the host: an empty result
            return 0, np.zeros(n_values, np.uint64)
        counts = np.zeros(max(n_values, 1), np.uint64)
        matched = C.c_uint64()
        check(load().mgx_facet_counts(self.device_index._h, C.byref(cqueries[0]), column_id, counts.ctypes.data,
                                      C.byref(matched)))
        return int(matched.value), counts[:n_values]

    def search_batch(self, queries):
        b = self.prepare(queries)
        b.execute()
        return b.fetch()
