"""ctypes wrapper over oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Mirrors the reference's class/method names (mygramdb::index::Index, BM25Scorer, ResultSorter,
search_pipeline::Execute) so parity tests read like the reference's own gtest files. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so (gcc, seconds)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u8p, u32p, u64p, f64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    sz = C.c_size_t
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_count_code_points.restype = sz
    L.orc_count_code_points.argtypes = [C.c_char_p, sz]
    L.orc_count_term_occurrences.restype = C.c_uint32
    L.orc_count_term_occurrences.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
    L.orc_compute_idf.restype = C.c_double
    L.orc_compute_idf.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_index_create.restype = C.c_void_p
    L.orc_index_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_index_destroy.argtypes = [C.c_void_p]
    L.orc_index_add_document.restype = C.c_int
    L.orc_index_add_document.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, sz]
    L.orc_index_from_csr.restype = C.c_void_p
    L.orc_index_from_csr.argtypes = [C.c_int, C.c_int, C.c_int, sz, u8p, u32p, u64p, u32p]
    L.orc_index_posting_size.restype = C.c_uint64
    L.orc_index_posting_size.argtypes = [C.c_void_p, C.c_char_p, sz]
    L.orc_index_gram_count.restype = sz
    L.orc_index_gram_count.argtypes = [C.c_void_p]
    for name in ("orc_search_and",):
        getattr(L, name).restype = C.POINTER(C.c_uint32)
    L.orc_search_and.argtypes = [C.c_void_p, u8p, u32p, sz, sz, C.c_int, C.POINTER(sz)]
    L.orc_search_or.restype = C.POINTER(C.c_uint32)
    L.orc_search_or.argtypes = [C.c_void_p, u8p, u32p, sz, C.POINTER(sz)]
    L.orc_search_not.restype = C.POINTER(C.c_uint32)
    L.orc_search_not.argtypes = [C.c_void_p, u32p, sz, u8p, u32p, sz, C.POINTER(sz)]
    L.orc_search_by_threshold.restype = C.POINTER(C.c_uint32)
    L.orc_search_by_threshold.argtypes = [C.c_void_p, u8p, u32p, sz, sz, C.POINTER(sz)]
    L.orc_filter_by_ngrams.restype = C.POINTER(C.c_uint32)
    L.orc_filter_by_ngrams.argtypes = [C.c_void_p, u32p, sz, u8p, u32p, sz, C.POINTER(sz)]
    L.orc_docstore_create.restype = C.c_void_p
    L.orc_docstore_destroy.argtypes = [C.c_void_p]
    L.orc_docstore_add.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, sz, C.c_int]
    L.orc_docstore_from_arrays.restype = C.c_void_p
    L.orc_docstore_from_arrays.argtypes = [sz, u8p, u64p]
    L.orc_docstore_bm25_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_score_documents.restype = C.c_int
    L.orc_score_documents.argtypes = [C.c_void_p, u32p, sz, u8p, u32p, sz, u64p, sz, C.c_uint64, C.c_double,
                                      C.c_double, C.c_double, f64p]
    L.orc_sort_by_score.restype = C.POINTER(C.c_uint32)
    L.orc_sort_by_score.argtypes = [u32p, f64p, sz, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(sz)]
    L.orc_execute.restype = C.c_int
    L.orc_execute.argtypes = [C.c_void_p, C.c_void_p, u8p, u32p, sz, u8p, u32p, sz, C.c_void_p, sz, C.c_int, C.c_int,
                              C.c_int, sz, C.c_int, C.c_int, C.c_void_p]
    L.orc_contains_fuzzy_match.restype = C.c_int
    L.orc_contains_fuzzy_match.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.c_uint32]
    L.orc_execute_fuzzy.restype = C.c_int
    L.orc_execute_fuzzy.argtypes = [C.c_void_p, C.c_void_p, u8p, u32p, sz, C.c_uint32, u8p, u32p, sz, C.c_void_p, sz,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_post_filter_by_text.restype = C.POINTER(C.c_uint32)
    L.orc_post_filter_by_text.argtypes = [C.c_void_p, u32p, sz, u8p, u32p, sz, C.POINTER(sz)]
    L.orc_pipeline_result_free.argtypes = [C.c_void_p]
    L.orc_search_scored.restype = C.c_int
    L.orc_search_scored.argtypes = [C.c_void_p, C.c_void_p, u8p, u32p, sz, C.c_int, C.c_int, C.c_int, sz, C.c_uint64,
                                    C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_uint32, u32p, f64p,
                                    C.POINTER(sz), C.POINTER(C.c_uint64)]
    _LIB = L
    return L


class _StrList(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("off", C.POINTER(C.c_uint32)), ("count", C.c_size_t),
                ("cap_bytes", C.c_size_t), ("cap_count", C.c_size_t)]


class _PipelineResult(C.Structure):
    _fields_ = [("results", C.POINTER(C.c_uint32)), ("n_results", C.c_size_t), ("total_candidates", C.c_size_t),
                ("after_intersection", C.c_size_t), ("after_not", C.c_size_t), ("after_filters", C.c_size_t),
                ("empty_term_detected", C.c_int), ("n_terms", C.c_size_t), ("term_order", C.c_uint32 * 64),
                ("term_df", C.c_uint64 * 64), ("term_estimated_size", C.c_uint64 * 64),
                ("exact_text_applied", C.c_int)]


class _Filter(C.Structure):
    _fields_ = [("docs", C.c_void_p), ("n_docs", C.c_size_t), ("negate", C.c_int)]


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def pack_terms(terms):
    """terms -> (bytes buffer, uint32 offsets) as numpy arrays (kept alive by the caller)."""
    bs = [_b(t) for t in terms]
    off = np.zeros(len(bs) + 1, dtype=np.uint32)
    for i, t in enumerate(bs):
        off[i + 1] = off[i] + len(t)
    buf = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy()
    return buf, off


def _take(ptr, n):
    out = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[: int(n)].copy() if n else np.zeros(0, np.uint32)
    lib().orc_free(ptr)
    return out.astype(np.uint32)


def count_code_points(text):
    t = _b(text)
    return int(lib().orc_count_code_points(t, len(t)))


def count_term_occurrences(text, term):
    t, m = _b(text), _b(term)
    return int(lib().orc_count_term_occurrences(t, len(t), m, len(m)))


def normalize_text(text, nfkc=True, width="keep", lower=True):
    """mygram::utils::NormalizeText with ICU (src/utils/string_utils.cpp:307-380), restated with the interpreter's own
    Unicode tables (unicodedata: an implementation independent of ICU; the tables here are Unicode 13, ICU 70's are
    14 — identical on every code point assigned by 13): NFKC, then width folding, then full lower-casing. Width is
    restated for the ASCII block only — "narrow": U+FF01..FF5E -> U+0021..007E and U+3000 -> space, "wide" the inverse
    (ICU's Fullwidth-Halfwidth / Halfwidth-Fullwidth also fold kana and a few symbols: not restated, callers keep
    those out of "narrow"/"wide" inputs). Invalid UTF-8 (bytes input) -> "" (:363-366). Pinned by the reference's
    own NormalizeText vectors (tests/golden/normalize.json)."""
    import unicodedata
    if isinstance(text, (bytes, bytearray)):
        try:
            text = bytes(text).decode("utf-8")
        except UnicodeDecodeError:
            return ""
    if nfkc:
        text = unicodedata.normalize("NFKC", text)
    if width == "narrow":
        text = "".join(chr(ord(c) - 0xFEE0) if 0xFF01 <= ord(c) <= 0xFF5E else (" " if c == "\u3000" else c)
                       for c in text)
    elif width == "wide":
        text = "".join(chr(ord(c) + 0xFEE0) if 0x21 <= ord(c) <= 0x7E else ("\u3000" if c == " " else c)
                       for c in text)
    if lower:
        text = text.lower()
    return text


def compute_idf(total_docs, doc_freq):
    return float(lib().orc_compute_idf(int(total_docs), int(doc_freq)))


def _grams(fn, text, *args):
    L = lib()
    t = _b(text)
    sl = _StrList()
    getattr(L, fn).argtypes = None
    getattr(L, fn)(C.c_char_p(t), C.c_size_t(len(t)), *[C.c_int(a) for a in args], C.byref(sl))
    out = []
    if sl.count:
        raw = C.string_at(sl.bytes, sl.off[sl.count])
        out = [raw[sl.off[i]:sl.off[i + 1]] for i in range(sl.count)]
    L.orc_strlist_free(C.byref(sl))
    return out


def generate_ngrams(text, n):
    return _grams("orc_generate_ngrams", text, n)


def generate_hybrid_ngrams(text, ascii_n=2, kanji_n=1, cross_boundary=True):
    return _grams("orc_generate_hybrid_ngrams", text, ascii_n, kanji_n, int(cross_boundary))


def generate_query_ngrams(text, ngram_size, kanji_ngram_size, cross_boundary=True):
    return _grams("orc_generate_query_ngrams", text, ngram_size, kanji_ngram_size, int(cross_boundary))


class Index:
    """mygramdb::index::Index, read side + AddDocument (src/index/index.h:46-300)."""

    def __init__(self, ngram_size=2, kanji_ngram_size=1, cross_boundary=True, _handle=None, _keep=None):
        self.ngram_size, self.kanji_ngram_size, self.cross_boundary = ngram_size, kanji_ngram_size, cross_boundary
        self._keep = _keep
        self._h = _handle or lib().orc_index_create(ngram_size, kanji_ngram_size, int(cross_boundary))

    @classmethod
    def from_csr(cls, ngram_size, kanji_ngram_size, cross_boundary, key_bytes, key_off, offsets, docids):
        key_bytes = np.ascontiguousarray(key_bytes, dtype=np.uint8)
        key_off = np.ascontiguousarray(key_off, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        docids = np.ascontiguousarray(docids, dtype=np.uint32)
        h = lib().orc_index_from_csr(ngram_size, kanji_ngram_size, int(cross_boundary), len(key_off) - 1,
                                     key_bytes.ctypes.data, key_off.ctypes.data, offsets.ctypes.data,
                                     docids.ctypes.data)
        return cls(ngram_size, kanji_ngram_size, cross_boundary, _handle=h,
                   _keep=(key_bytes, key_off, offsets, docids))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_index_destroy(self._h)
            self._h = None

    def add_document(self, doc_id, text):
        t = _b(text)
        return bool(lib().orc_index_add_document(self._h, doc_id, t, len(t)))

    def posting_size(self, gram):
        g = _b(gram)
        return int(lib().orc_index_posting_size(self._h, g, len(g)))

    def gram_count(self):
        return int(lib().orc_index_gram_count(self._h))

    def search_and(self, terms, limit=0, reverse=False):
        buf, off = pack_terms(terms)
        n = C.c_size_t()
        p = lib().orc_search_and(self._h, buf.ctypes.data, off.ctypes.data, len(terms), limit, int(reverse),
                                 C.byref(n))
        return _take(p, n.value)

    def search_or(self, terms):
        buf, off = pack_terms(terms)
        n = C.c_size_t()
        p = lib().orc_search_or(self._h, buf.ctypes.data, off.ctypes.data, len(terms), C.byref(n))
        return _take(p, n.value)

    def search_not(self, all_docs, terms):
        a = np.ascontiguousarray(all_docs, dtype=np.uint32)
        buf, off = pack_terms(terms)
        n = C.c_size_t()
        p = lib().orc_search_not(self._h, a.ctypes.data, len(a), buf.ctypes.data, off.ctypes.data, len(terms),
                                 C.byref(n))
        return _take(p, n.value)

    def search_by_threshold(self, terms, threshold):
        buf, off = pack_terms(terms)
        n = C.c_size_t()
        p = lib().orc_search_by_threshold(self._h, buf.ctypes.data, off.ctypes.data, len(terms), threshold,
                                          C.byref(n))
        return _take(p, n.value)

    def filter_by_ngrams(self, candidates, terms):
        c = np.ascontiguousarray(candidates, dtype=np.uint32)
        buf, off = pack_terms(terms)
        n = C.c_size_t()
        p = lib().orc_filter_by_ngrams(self._h, c.ctypes.data, len(c), buf.ctypes.data, off.ctypes.data, len(terms),
                                       C.byref(n))
        return _take(p, n.value)


class DocumentStore:
    """The text side of storage::DocumentStore that BM25 reads (normalized text by docid)."""

    def __init__(self, _handle=None, _keep=None):
        self._keep = _keep
        self._h = _handle or lib().orc_docstore_create()

    @classmethod
    def from_arrays(cls, text_bytes, text_off):
        text_bytes = np.ascontiguousarray(text_bytes, dtype=np.uint8)
        text_off = np.ascontiguousarray(text_off, dtype=np.uint64)
        h = lib().orc_docstore_from_arrays(len(text_off) - 1, text_bytes.ctypes.data, text_off.ctypes.data)
        return cls(_handle=h, _keep=(text_bytes, text_off))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_docstore_destroy(self._h)
            self._h = None

    def add(self, doc_id, text=None):
        if text is None:
            lib().orc_docstore_add(self._h, doc_id, b"", 0, 0)
        else:
            t = _b(text)
            lib().orc_docstore_add(self._h, doc_id, t, len(t), 1)

    def bm25_stats(self):
        c, t = C.c_uint64(), C.c_uint64()
        lib().orc_docstore_bm25_stats(self._h, C.byref(c), C.byref(t))
        return int(c.value), int(t.value)


def score_documents(store, candidates, terms, dfs, total_docs, avg_doc_length, k1=1.2, b=0.75):
    """BM25Scorer::ScoreDocuments. Raises ValueError on the reference's kInvalidArgument case."""
    c = np.ascontiguousarray(candidates, dtype=np.uint32)
    buf, off = pack_terms(terms)
    d = np.ascontiguousarray(dfs, dtype=np.uint64)
    out = np.zeros(max(len(c), 1), dtype=np.float64)
    rc = lib().orc_score_documents(store._h, c.ctypes.data, len(c), buf.ctypes.data, off.ctypes.data, len(terms),
                                   d.ctypes.data, len(d), int(total_docs), float(avg_doc_length), float(k1), float(b),
                                   out.ctypes.data)
    if rc != 0:
        raise ValueError("BM25 search_terms and term_doc_freqs must have identical lengths")
    return out[: len(c)]


def sort_by_score(results, scores, descending=True, limit=0, offset=0):
    """ResultSorter::SortByScore."""
    r = np.ascontiguousarray(results, dtype=np.uint32)
    s = np.ascontiguousarray(scores, dtype=np.float64)
    n = C.c_size_t()
    p = lib().orc_sort_by_score(r.ctypes.data, s.ctypes.data, len(r), int(descending), limit, offset, C.byref(n))
    return _take(p, n.value)


def execute(index, store, terms, not_terms=(), filters=(), filter_threshold=1000, compute_df=False,
            ngram_size=None, kanji_ngram_size=None, cross_boundary=None, verify_text=False):
    """search_pipeline::GenerateTermInfos + sort + Execute. filters = [(sorted docids, negate)]."""
    tb, toff = pack_terms(terms)
    nb, noff = pack_terms(not_terms)
    keep = []
    farr = (_Filter * max(len(filters), 1))()
    for i, (docs, negate) in enumerate(filters):
        a = np.ascontiguousarray(docs, dtype=np.uint32)
        keep.append(a)
        farr[i].docs, farr[i].n_docs, farr[i].negate = a.ctypes.data, len(a), int(negate)
    pr = _PipelineResult()
    rc = lib().orc_execute(index._h, store._h if store is not None else None, tb.ctypes.data, toff.ctypes.data,
                           len(terms), nb.ctypes.data, noff.ctypes.data, len(not_terms), C.byref(farr), len(filters),
                           index.ngram_size if ngram_size is None else ngram_size,
                           index.kanji_ngram_size if kanji_ngram_size is None else kanji_ngram_size,
                           int(index.cross_boundary if cross_boundary is None else cross_boundary),
                           filter_threshold, int(compute_df), int(verify_text), C.byref(pr))
    if rc != 0:
        raise ValueError("orc_execute rc=%d" % rc)
    res = np.ctypeslib.as_array(pr.results, shape=(max(pr.n_results, 1),))[: pr.n_results].copy()
    out = {
        "results": res.astype(np.uint32),
        "total_candidates": pr.total_candidates, "after_intersection": pr.after_intersection,
        "after_not": pr.after_not, "after_filters": pr.after_filters,
        "empty_term_detected": bool(pr.empty_term_detected),
        "term_order": [int(pr.term_order[i]) for i in range(pr.n_terms)],
        "term_df": [int(pr.term_df[i]) for i in range(pr.n_terms)],
        "term_estimated_size": [int(pr.term_estimated_size[i]) for i in range(pr.n_terms)],
        "exact_text_applied": bool(pr.exact_text_applied),
    }
    lib().orc_pipeline_result_free(C.byref(pr))
    return out


def contains_fuzzy_match(text, term, max_distance):
    """ContainsFuzzyMatch, src/utils/edit_distance.cpp:199-296."""
    t = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    q = term.encode("utf-8") if isinstance(term, str) else bytes(term)
    return bool(lib().orc_contains_fuzzy_match(t, len(t), q, len(q), max_distance))


def execute_fuzzy(index, store, terms, max_distance, not_terms=(), filters=(), ngram_size=None, kanji_ngram_size=None,
                  cross_boundary=None, verify_text=False):
    """search_pipeline::ExecuteWithFuzzy (src/server/search_pipeline.cpp:1659-1744). -> dict like execute() plus
    "thetas": the SearchByThreshold threshold of every term."""
    tb, toff = pack_terms(terms)
    nb, noff = pack_terms(not_terms)
    keep = []
    farr = (_Filter * max(len(filters), 1))()
    for i, (docs, negate) in enumerate(filters):
        a = np.ascontiguousarray(docs, dtype=np.uint32)
        keep.append(a)
        farr[i].docs, farr[i].n_docs, farr[i].negate = a.ctypes.data, len(a), int(negate)
    pr = _PipelineResult()
    thetas = np.zeros(max(len(terms), 1), dtype=np.uint64)
    rc = lib().orc_execute_fuzzy(index._h, store._h if store is not None else None, tb.ctypes.data, toff.ctypes.data,
                                 len(terms), int(max_distance), nb.ctypes.data, noff.ctypes.data, len(not_terms),
                                 C.byref(farr), len(filters),
                                 index.ngram_size if ngram_size is None else ngram_size,
                                 index.kanji_ngram_size if kanji_ngram_size is None else kanji_ngram_size,
                                 int(index.cross_boundary if cross_boundary is None else cross_boundary),
                                 int(verify_text), C.byref(pr), thetas.ctypes.data)
    if rc != 0:
        raise ValueError("orc_execute_fuzzy rc=%d" % rc)
    res = np.ctypeslib.as_array(pr.results, shape=(max(pr.n_results, 1),))[: pr.n_results].copy()
    out = {"results": res.astype(np.uint32), "total_candidates": pr.total_candidates,
           "after_intersection": pr.after_intersection, "after_not": pr.after_not, "after_filters": pr.after_filters,
           "empty_term_detected": bool(pr.empty_term_detected), "exact_text_applied": bool(pr.exact_text_applied),
           "thetas": [int(t) for t in thetas[: len(terms)]]}
    lib().orc_pipeline_result_free(C.byref(pr))
    return out


def post_filter_by_text(store, candidates, normalized_terms):
    """PostFilterByText, search_pipeline.cpp:1239-1246."""
    cand = np.ascontiguousarray(candidates, dtype=np.uint32)
    tb, toff = pack_terms(normalized_terms)
    n = C.c_size_t()
    p = lib().orc_post_filter_by_text(store._h, cand.ctypes.data, len(cand),
                                      tb.ctypes.data, toff.ctypes.data, len(normalized_terms), C.byref(n))
    out = np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[: n.value].copy()
    lib().orc_free(p)
    return out.astype(np.uint32)


def has_uncovered_hybrid_fragment(term, ngram_size, kanji_ngram_size, cross_boundary):
    """search_pipeline.cpp:80-136 on a NORMALIZED term."""
    t = _b(term)
    return bool(lib().orc_has_uncovered_hybrid_fragment(t, len(t), ngram_size, kanji_ngram_size, int(cross_boundary)))


def search_scored(index, store, terms, total_docs, avg_doc_length, k1=1.2, b=0.75, descending=True, limit=10,
                  offset=0, filter_threshold=1000, max_results=None):
    """SEARCH ... SORT _score: (total, top docids, their scores)."""
    tb, toff = pack_terms(terms)
    cap = max_results if max_results is not None else (limit if limit > 0 else 1 << 24)
    docs = np.zeros(max(cap, 1), dtype=np.uint32)
    scores = np.zeros(max(cap, 1), dtype=np.float64)
    n, total = C.c_size_t(), C.c_uint64()
    rc = lib().orc_search_scored(index._h, store._h, tb.ctypes.data, toff.ctypes.data, len(terms), index.ngram_size,
                                 index.kanji_ngram_size, int(index.cross_boundary), filter_threshold, int(total_docs),
                                 float(avg_doc_length), float(k1), float(b), int(descending), limit, offset,
                                 docs.ctypes.data, scores.ctypes.data, C.byref(n), C.byref(total))
    if rc != 0:
        raise ValueError("orc_search_scored rc=%d" % rc)
    return int(total.value), docs[: n.value].copy(), scores[: n.value].copy()
