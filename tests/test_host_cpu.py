"""CPU-only checks: the C-ABI library loads and exports what include/*.h declares, and the host-side column builder
(product code) agrees with the oracle's independent text-level restatement. No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O
from pkg import ROOT, mg


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(mgxt?_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    L = mg._capi.load()
    declared = _declared("mygram_gpu.h") | _declared("mygram_tools.h")
    assert declared == set(mg._capi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.mgx_abi_version() == mg._capi.ABI_VERSION


def test_shim_library_exports_every_declared_symbol():
    from mygram_db_amd import _shim_capi as S
    L = S.load()
    src = open(os.path.join(ROOT, "include", "mygram_shim_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"\b(mgxs_[a-z0-9_]+)\s*\(", src))
    assert declared == set(S.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_no_cpu_fallback_without_device():
    L = mg._capi.load()
    if L.mgx_device_count() > 0:
        pytest.skip("a device is present")
    cols = mg.Columns(mg.Corpus.from_texts(["abc", "bcd"]), 1, 2, 0, True)
    with pytest.raises(mg._capi.MgxError) as e:
        mg.DeviceIndex(cols)
    assert e.value.code == 5 and "no CPU fallback" in str(e.value)


def _check_columns_against_oracle(texts, first_doc_id, ngram, kanji, cross):
    corpus = mg.Corpus.from_texts(texts)
    cols = mg.Columns(corpus, first_doc_id, ngram, kanji, cross, n_threads=3)
    oidx = O.Index(ngram, kanji, cross)
    store = O.DocumentStore()
    for i, t in enumerate(texts):
        oidx.add_document(first_doc_id + i, t)
        store.add(first_doc_id + i, t)
    assert cols.n_grams == oidx.gram_count()
    keys = [cols.gram(g) for g in range(cols.n_grams)]
    assert keys == sorted(keys)
    for g, key in enumerate(keys):
        lo, hi = int(cols.offsets[g]), int(cols.offsets[g + 1])
        want = oidx.search_and([key])
        assert cols.docids[lo:hi].tolist() == want.tolist(), key
        ovf = dict(zip(cols.tf_overflow_pos.tolist(), cols.tf_overflow_val.tolist()))
        for p in range(lo, hi):
            text = texts[int(cols.docids[p]) - first_doc_id]
            want = O.count_term_occurrences(text, key)
            assert cols.tf[p] == min(255, want), (key, text)
            assert (ovf.get(p) == want) if want >= 255 else (p not in ovf), (key, want)
        assert cols.lookup(key) == g
    assert cols.lookup("\x01\x02") is None
    for i, t in enumerate(texts):
        assert cols.doc_len[i] == O.count_code_points(t)
    assert (cols.bm25_doc_count, cols.bm25_total_len) == store.bm25_stats()


def test_columns_match_oracle_on_golden_corpora():
    for case in G.index_cases():
        docs = G.expand_docs(case["docs"])
        ids = [d for d, _ in docs]
        if len(docs) > 3000 or ids != list(range(ids[0], ids[0] + len(ids))):
            continue  # the builder takes dense id ranges; sparse-id cases are covered through the device tests
        ic = case["index"]
        _check_columns_against_oracle([t for _, t in docs], ids[0], ic["ngram"], ic.get("kanji", 0), True)


def test_columns_tf_is_non_overlapping_and_greedy():
    # bm25_scorer_test.cpp:69-73: "aa" in "aaa" -> 1, in "aaaa" -> 2
    _check_columns_against_oracle(["aaa", "aaaa", "aaaaa", "abababa", "", "a", "aa bb aa"], 1, 2, 0, True)


def test_columns_hybrid_cjk():
    texts = ["東京都", "東京は日本の首都です", "aあ東京b", "カタカナとひらがな", "abc東京def", "", "京"]
    _check_columns_against_oracle(texts, 10, 2, 1, True)
    _check_columns_against_oracle(texts, 10, 2, 1, False)
    _check_columns_against_oracle(texts, 10, 3, 3, True)
    _check_columns_against_oracle(texts, 10, 1, 0, True)


def test_columns_match_oracle_on_synthetic_corpus():
    corpus = mg.Corpus.synthetic(3000, seed=7)
    texts = [corpus.text(i).decode() for i in range(corpus.n_docs)]
    assert all(4 <= len(t.split(" ")) <= 16 for t in texts)
    _check_columns_against_oracle(texts, 1, 2, 0, True)


def test_synthetic_corpus_is_shard_independent():
    whole = mg.Corpus.synthetic(5000, seed=42, n_threads=2)
    part = mg.Corpus.synthetic(1200, seed=42, global_first=3100, n_threads=5)
    for i in range(1200):
        assert part.text(i) == whole.text(3100 + i)
    assert mg.Corpus.synthetic(50, seed=43).text(0) != whole.text(0)


def test_query_ngram_rules_match_oracle():
    for text in ["hello world", "東京ab", "aあ東", "x", "", "日本語のテキスト"]:
        for (n, k, cb) in [(2, 0, True), (2, 1, True), (2, 1, False), (3, 3, True), (0, 0, True), (1, 0, True)]:
            got = [g.encode() for g in mg.generate_query_ngrams(text, n, k, cb)]
            assert got == O.generate_query_ngrams(text, n, k, cb), (text, n, k, cb)


def test_compute_idf_matches_oracle():
    for n, df in [(100, 10), (100, 0), (0, 10), (10, 20), (10_000_000, 3_141_592)]:
        assert mg.compute_idf(n, df) == O.compute_idf(n, df)


def _raw_index_desc(first_doc_id, n_docs, offsets, docids):
    off = np.asarray(offsets, dtype=np.uint64)
    ids = np.asarray(docids, dtype=np.uint32)
    tf = np.ones(len(ids), dtype=np.uint8)
    dl = np.ones(n_docs, dtype=np.uint32)
    d = mg._capi.IndexDesc(C.sizeof(mg._capi.IndexDesc), mg._capi.ABI_VERSION, 0, 0, first_doc_id, n_docs,
                           len(off) - 1, off.ctypes.data, ids.ctypes.data, tf.ctypes.data, dl.ctypes.data, 0.0)
    return d, (off, ids, tf, dl)


@pytest.mark.parametrize("first,n_docs,offsets,docids,code,what", [
    (10, 5, [0, 2, 3], [10, 15, 11], 3, "outside"),       # 15 is one past the owned range [10, 15)
    (10, 5, [0, 2, 3], [9, 12, 11], 3, "outside"),        # below first_doc_id (a shard given the wrong base)
    (1, 8, [0, 3, 4], [2, 2, 5, 1], 2, "ascend"),         # duplicate inside a list
    (1, 8, [0, 3, 4], [3, 2, 5, 1], 2, "ascend"),         # descending inside a list
    (1, 8, [0, 3, 2], [1, 2, 3], 2, "non-decreasing"),    # offsets go backwards
])
def test_index_create_checks_its_contract_before_any_device_write(first, n_docs, offsets, docids, code, what):
    """The build kernels address bitmap bits / tf nibbles by doc id without a bounds guard: the descriptor's contract
    (ids inside the owned range, strictly ascending per gram, monotonic offsets) is validated on the host first — and
    before the device is even looked for, so this runs on the CPU box."""
    d, keep = _raw_index_desc(first, n_docs, offsets, docids)
    h = C.c_void_p()
    rc = mg._capi.load().mgx_index_create(C.byref(d), C.byref(h))
    assert rc == code and h.value is None
    assert what in mg._capi.load().mgx_last_error().decode()


def test_columns_tf_overflow_table():
    """CountTermOccurrences has no ceiling (bm25_scorer.cpp:27-45): counts of 255 and more keep their true value in the
    side table next to the saturated byte column."""
    texts = ["ab" * 300, "ab" * 255 + " x", "ab" * 254, "xyab", "a" * 700, ""]
    _check_columns_against_oracle(texts, 1, 2, 0, True)
    cols = mg.Columns(mg.Corpus.from_texts(texts), 1, 2, 0, True)
    # "ab" x255, "ba" inside "ab" x300, "ab" x300, "aa" in 700 a's
    assert sorted(cols.tf_overflow_val.tolist()) == [255, 299, 300, 350]
    assert np.all(np.diff(cols.tf_overflow_pos.astype(np.int64)) > 0)


def _normalize_vectors():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "normalize.json"), encoding="utf-8"))


def test_normalize_text_reference_vectors():
    """mygram::utils::NormalizeText of the host layer (csrc/shim/normalize.cpp) against the reference's own vectors;
    the vectors inside the reference's #ifdef USE_ICU apply when this build found ICU (the image has ICU 70)."""
    from mygram_db_amd import _shim_capi as S
    doc = _normalize_vectors()
    icu = S.normalize_uses_icu()
    ran = 0
    for v in doc["vectors"]:
        if v["needs_icu"] and not icu:
            continue
        assert S.normalize_text(v["text"], v["nfkc"], v["width"], v["lower"]) == v["expected"], v
        ran += 1
    assert ran >= (len(doc["vectors"]) if icu else 5)
    bad = doc["invalid_utf8"]  # fails closed
    assert S.normalize_text(bytes.fromhex(bad["hex"]), bad["nfkc"], bad["width"], bad["lower"]) == b""
    assert mg.engine.normalize_uses_icu() == icu


def test_normalize_text_against_oracle_restatement():
    """ICU (the product's host layer) vs the oracle's independent restatement (the interpreter's Unicode tables) on
    seeded strings over the blocks a CJK / Latin corpus meets: compatibility forms, width forms, case pairs."""
    from mygram_db_amd import _shim_capi as S
    if not S.normalize_uses_icu():
        pytest.skip("built without ICU: only the ASCII branch exists")
    import unicodedata
    rng = np.random.default_rng(77)
    blocks = [(0x20, 0x7E), (0xA0, 0x24F), (0x370, 0x3FF), (0x400, 0x4FF), (0x2100, 0x214F), (0x2460, 0x24FF),
              (0x3040, 0x30FF), (0x3300, 0x33FF), (0x4E00, 0x4FFF), (0xFB00, 0xFB06), (0xFF01, 0xFFEE), (0x1F600, 0x1F64F)]
    checked = 0
    for _ in range(400):
        cps = []
        for _ in range(int(rng.integers(1, 24))):
            lo, hi = blocks[int(rng.integers(len(blocks)))]
            c = int(rng.integers(lo, hi + 1))
            if unicodedata.category(chr(c)) == "Cn":  # unassigned in the interpreter's (older) tables: version-dependent
                continue
            cps.append(c)
        s = "".join(map(chr, cps))
        for nfkc in (True, False):
            for lower in (True, False):
                assert S.normalize_text(s, nfkc, "keep", lower) == O.normalize_text(s, nfkc, "keep", lower), \
                    ([hex(c) for c in cps], nfkc, lower)
                checked += 1
        ascii_only = "".join(chr(c) for c in cps if c < 0x7F or 0xFF01 <= c <= 0xFF5E)
        for w in ("narrow", "wide"):
            assert S.normalize_text(ascii_only, False, w, False) == O.normalize_text(ascii_only, False, w, False), \
                (ascii_only, w)
    assert checked == 1600


def test_index_normalizes_query_terms_like_the_reference():
    """Index::NormalizeText forwards its three settings (index.h:321-323); engine.Index plans with them."""
    from mygram_db_amd import _shim_capi as S
    if not S.normalize_uses_icu():
        pytest.skip("built without ICU")
    assert S.normalize_text("ﾗｲﾌﾞ ＡＢＣ") == "ライブ abc"
    assert S.normalize_text("ＡＢＣ", False, "keep", False) == "ＡＢＣ"
    assert S.normalize_text("ÀÉ Σ", True, "keep", True) == "àé σ"


# ---- MGIX (the reference's index dump) --------------------------------------------------------------------------------

def _mgix_expected():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "mgix_expected.json"), encoding="utf-8"))


@pytest.mark.parametrize("name", ["mgix_v4.bin", "mgix_v3.bin", "mgix_v2.bin", "mgix_v1.bin"])
def test_mgix_dump_to_columns(name):
    """mgx_columns_from_mgix over the committed dumps (tests/golden/make_mgix.py): header fields of every format
    version, delta lists, Roaring array / bitset / run containers, both cookies, with and without the offset header."""
    exp = _mgix_expected()
    f = exp["files"][name]
    data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert len(data) == f["bytes"]
    cols = mg.Columns.from_mgix(data)
    assert cols.mgix["version"] == f["version"] and cols.ngram_size == f["ngram_size"]
    assert cols.kanji_ngram_size == f["kanji_ngram_size"] and cols.cross_boundary == f["cross_boundary"]
    assert cols.mgix["normalize_width"] == f["normalize_width"] and cols.mgix["n_terms"] == len(f["terms"])
    assert cols.n_grams == len(f["terms"])
    keys = [cols.gram(g) for g in range(cols.n_grams)]
    assert keys == sorted(t.encode("utf-8") for t in f["terms"])  # the dictionary is sorted bytewise, whatever the file order
    lo, hi = 1 << 32, 0
    for t in f["terms"]:
        e = exp["terms"][t]
        gid = cols.lookup(t)
        ids = cols.docids[int(cols.offsets[gid]):int(cols.offsets[gid + 1])]
        assert len(ids) == e["count"] and int(ids[0]) == e["first"] and int(ids[-1]) == e["last"], t
        assert int(ids.astype(np.uint64).sum()) == e["sum"] and int(np.bitwise_xor.reduce(ids)) == e["xor"], t
        assert np.all(np.diff(ids.astype(np.int64)) > 0), t
        lo, hi = min(lo, e["first"]), max(hi, e["last"])
    assert cols.first_doc_id == lo and cols.n_docs == hi - lo + 1  # the span of the dump's ids
    assert len(cols.tf) == 0 and len(cols.doc_len) == 0            # a dump carries doc ids only


def test_mgix_dump_rejects_damage():
    data = bytearray(open(os.path.join(ROOT, "tests", "golden", "mgix_v4.bin"), "rb").read())
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x40  # a flipped bit anywhere fails the CRC32 trailer
    for blob, what in [(bytes(bad), "CRC32"), (bytes(data[:-9]), "CRC32"), (b"MGIY" + bytes(data[4:]), "magic"),
                       (bytes(data[:12]), "magic")]:
        with pytest.raises(mg._capi.MgxError) as e:
            mg.Columns.from_mgix(blob)
        assert what in str(e.value), (what, str(e.value))
    v9 = bytearray(data)
    v9[4] = 9
    with pytest.raises(mg._capi.MgxError) as e:
        mg.Columns.from_mgix(bytes(v9))
    assert e.value.code == 4
    with pytest.raises(mg._capi.MgxError) as e:  # a caller's range that does not hold the dump's ids
        mg.Columns.from_mgix(bytes(data), first_doc_id=1, n_docs=1000)
    assert e.value.code == 3


def test_sort_based_build_equals_the_dictionary_first_build(monkeypatch):
    """The column builder has two builds — dictionary first (small dictionaries: a bigram index) and sort-based (large ones:
    CJK trigrams, 10^7..10^8 grams; chosen from the first chunk's distinct grams). Both must produce identical columns, and
    the sort-based one is checked against the oracle like the other."""
    rng = np.random.default_rng(5)
    alphabet = [chr(0x4E00 + i) for i in range(400)] + [chr(0x3042 + i) for i in range(40)] + list("abc ")
    cjk = ["".join(alphabet[j] for j in rng.integers(0, len(alphabet), size=int(rng.integers(0, 40)))) for _ in range(6000)]
    ascii_texts = [mg.Corpus.synthetic(4000, seed=9).text(i).decode() for i in range(4000)] + ["ab" * 300, "", "a" * 700]

    def build(texts, ngram, kanji, mode, threads):
        monkeypatch.setenv("MGX_BUILD_SORTED", mode)
        c = mg.Columns(mg.Corpus.from_texts(texts), 7, ngram, kanji, True, n_threads=threads)
        return (c.n_grams, c.key_bytes.tobytes(), c.key_off.tolist(), c.offsets.tolist(), c.docids[: c.n_postings].tolist(),
                c.tf[: c.n_postings].tolist(), c.doc_len.tolist(), c.bm25_doc_count, c.bm25_total_len,
                c.tf_overflow_pos.tolist(), c.tf_overflow_val.tolist())
    for texts, ngram, kanji in ((cjk, 3, 3), (cjk, 2, 1), (ascii_texts, 2, 0)):
        a = build(texts, ngram, kanji, "0", 3)
        b = build(texts, ngram, kanji, "1", 5)
        assert a == b
    monkeypatch.setenv("MGX_BUILD_SORTED", "1")
    _check_columns_against_oracle(["東京都", "東京は日本の首都です", "aあ東京b", "ab" * 260, "", "京", "abababa"], 10, 2, 1, True)
    _check_columns_against_oracle([mg.Corpus.synthetic(800, seed=3).text(i).decode() for i in range(800)], 1, 2, 0, True)


def test_a_large_dictionary_is_recognised_whatever_the_chunk_size(monkeypatch):
    """The choice between the two builds is made from a fixed sample of documents. (It was "the first chunk": when chunks
    shrank to 512 documents a CJK trigram corpus showed too few distinct grams in it, was taken for a small dictionary, and
    the dictionary-first build of 18M grams never came back.) 40k CJK docs -> ~1M distinct trigrams must build in seconds."""
    import time
    monkeypatch.delenv("MGX_BUILD_SORTED", raising=False)
    rng = np.random.default_rng(3)
    n = 40_000
    cps = np.concatenate([0x4E00 + np.arange(3000), 0x3042 + np.arange(80)]).astype(np.uint32)
    wi = 1.0 / np.arange(1, 3001)
    p = np.concatenate([wi / wi.sum() * 0.9, np.full(80, 0.1 / 80)])
    lens = rng.integers(16, 49, size=n)
    flat = cps[rng.choice(len(cps), size=int(lens.sum()), p=p)]
    b = np.empty((len(flat), 3), np.uint8)
    b[:, 0] = 0xE0 | (flat >> 12)
    b[:, 1] = 0x80 | ((flat >> 6) & 0x3F)
    b[:, 2] = 0x80 | (flat & 0x3F)
    off = np.concatenate([[0], np.cumsum(lens.astype(np.uint64) * 3)]).astype(np.uint64)
    corpus = mg.Corpus(np.concatenate([b.reshape(-1), np.zeros(16, np.uint8)]), off)
    t0 = time.perf_counter()
    cols = mg.Columns(corpus, 1, 3, 3, True, n_threads=4)
    took = time.perf_counter() - t0
    assert cols.n_grams > 500_000 and cols.n_postings >= cols.n_grams
    assert took < 30.0, took


def test_dump_v2_table_to_columns_texts_and_filters():
    """mgx_dump_open: a DUMP SAVE file ("MGDB" v2) -> one table's columns (built from the stored texts, cross-checked with
    the dump's own MGIX index), normalized texts by slot, the store's id set and its filter values. The fixture is written
    by tests/golden/make_dump_v2.py from the reference's writer code (no dump written by the reference exists here: parity
    with a real DUMP SAVE is unpinned beyond that restatement)."""
    import json
    L = mg._capi.load()
    raw = open(os.path.join(ROOT, "tests", "golden", "dump_v2_small.bin"), "rb").read()
    exp = json.load(open(os.path.join(ROOT, "tests", "golden", "dump_v2_expected.json"), encoding="utf-8"))
    h = C.c_void_p()
    mg._capi.check(L.mgx_dump_open(raw, len(raw), b"app_db.articles", C.byref(h)))
    v = mg._capi.DumpView()
    mg._capi.check(L.mgx_dump_view_get(h, C.byref(v)))
    assert v.table_name == b"app_db.articles" and v.first_doc_id == 1 and v.n_docs == 40 and v.n_existing == len(exp["ids"])
    assert v.index_info.ngram_size == 2 and v.index_info.kanji_ngram_size == 2 and v.has_texts == 1
    exists = np.ctypeslib.as_array(C.cast(v.exists, C.POINTER(C.c_uint8)), shape=(40,))
    assert [i + 1 for i in np.nonzero(exists)[0]] == exp["ids"]
    off = np.ctypeslib.as_array(C.cast(v.text_off, C.POINTER(C.c_uint64)), shape=(41,))
    tb = np.ctypeslib.as_array(C.cast(v.text_bytes, C.POINTER(C.c_uint8)), shape=(int(off[40]) + 16,)).tobytes()
    texts = [tb[int(off[i]): int(off[i + 1])].decode() for i in range(40)]
    assert texts == [exp["texts"].get(str(i + 1), "") for i in range(40)]
    # filter columns: name, type, widened values / strings, NULLs (a NULL value and an absent column read the same)
    cols = {}
    for i in range(v.n_filter_columns):
        fc = mg._capi.DumpFilterColumn()
        mg._capi.check(L.mgx_dump_filter_column_get(h, i, C.byref(fc)))
        vals = np.ctypeslib.as_array(C.cast(fc.values, C.POINTER(C.c_uint64)), shape=(40,)).copy()
        nul = np.ctypeslib.as_array(C.cast(fc.is_null, C.POINTER(C.c_uint8)), shape=(40,)).copy()
        cols[fc.name.decode()] = (int(fc.value_type), vals, nul, fc)
    assert set(cols) == {"status", "category", "score", "flag"}
    assert cols["status"][0] == 6 and cols["category"][0] == 11 and cols["score"][0] == 12 and cols["flag"][0] == 1
    for d in exp["ids"]:
        f = exp["filters"][str(d)]
        if f.get("status", ["null"])[0] == "null":
            assert all(cols[c][2][d - 1] == 1 for c in cols)
            continue
        assert cols["status"][1][d - 1] == d % 3 and cols["status"][2][d - 1] == 0
        assert cols["score"][1][d - 1:d].view(np.float64)[0] == d / 4.0
        assert cols["flag"][1][d - 1] == d % 2
    fc = cols["category"][3]
    so = np.ctypeslib.as_array(C.cast(fc.string_off, C.POINTER(C.c_uint64)), shape=(41,))
    sb = np.ctypeslib.as_array(C.cast(fc.string_bytes, C.POINTER(C.c_uint8)), shape=(int(so[40]) + 1,)).tobytes()
    assert sb[int(so[0]): int(so[1])] == b"food" and sb[int(so[6]): int(so[7])] == b""  # doc 1: 1 % 3 -> food; doc 7: deleted
    # the columns equal a direct build from the same texts
    ch = C.c_void_p()
    mg._capi.check(L.mgx_dump_take_columns(h, C.byref(ch)))
    got = mg.Columns.__new__(mg.Columns)
    got.ngram_size, got.kanji_ngram_size, got.cross_boundary = 2, 2, True
    got._adopt(ch)
    want = mg.Columns(mg.Corpus.from_texts(texts), 1, 2, 2, True)
    assert got.n_grams == want.n_grams and got.offsets.tolist() == want.offsets.tolist()
    assert got.docids[: got.n_postings].tolist() == want.docids[: want.n_postings].tolist()
    assert got.tf[: got.n_postings].tolist() == want.tf[: want.n_postings].tolist() and got.doc_len.tolist() == want.doc_len.tolist()
    L.mgx_dump_destroy(h)
    # the first table of the file is the (empty) other one; an unknown table is an error
    h2 = C.c_void_p()
    mg._capi.check(L.mgx_dump_open(raw, len(raw), None, C.byref(h2)))
    v2 = mg._capi.DumpView()
    mg._capi.check(L.mgx_dump_view_get(h2, C.byref(v2)))
    assert v2.table_name == b"app_db.other" and v2.n_docs == 0
    L.mgx_dump_destroy(h2)
    assert L.mgx_dump_open(raw, len(raw), b"nosuch", C.byref(h2)) == 4000
    # damage: a flipped byte (file CRC), a truncated file, a wrong version
    bad = bytearray(raw)
    bad[len(bad) // 2] ^= 0x40
    assert L.mgx_dump_open(bytes(bad), len(bad), None, C.byref(h2)) == 2 and b"CRC32" in L.mgx_last_error()
    assert L.mgx_dump_open(raw[:-9], len(raw) - 9, None, C.byref(h2)) == 2
    v9 = bytearray(raw)
    v9[4] = 9
    assert L.mgx_dump_open(bytes(v9), len(v9), None, C.byref(h2)) == 4
