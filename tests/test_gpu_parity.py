"""-m gpu parity tests: the HIP path (through the C ABI of libmygram_gpu.so) against the CPU oracle and the
reference's own known-answer vectors. Integer results are compared bit-exactly; BM25 scores bit-exactly too
(the required bar is 1e-5 relative)."""
import os

import numpy as np
import pytest

import golden_util as G
from gpu_util import Pair, densify
from oracle import oracle as O
from pkg import mg

pytestmark = pytest.mark.gpu
Query = mg.engine.Query

DENSE = [pytest.param(0.0, id="dense-bitmaps"), pytest.param(4.0, id="lists-only")]


def _run_call(idx, call):
    op = call["op"]
    if op == "search_and":
        return idx.search_and(call["terms"], call.get("limit", 0), call.get("reverse", False))
    if op == "search_or":
        return idx.search_or(call["terms"])
    if op == "search_not":
        return idx.search_not(G.expand_ids(call["all_docs"]), call["terms"])
    if op == "search_by_threshold":
        return idx.search_by_threshold(call["terms"], call["threshold"])
    if op == "filter_by_ngrams":
        return idx.filter_by_ngrams(G.expand_ids(call["candidates"]), call["terms"])
    raise ValueError(op)


@pytest.mark.parametrize("dense", DENSE)
@pytest.mark.parametrize("case", G.index_cases(), ids=lambda c: c["id"])
def test_reference_known_answers(case, dense):
    """tests/index/{index_search,search_by_threshold,index_gettopn}_test.cpp vectors through Index::Search*."""
    first, texts = densify(G.expand_docs(case["docs"]))
    ic = case["index"]
    idx = mg.Index(texts=texts, first_doc_id=first, ngram_size=ic["ngram"], kanji_ngram_size=ic.get("kanji", 0),
                   dense_threshold=dense)
    for call in case["calls"]:
        got = _run_call(idx, call)
        assert got.tolist() == G.expand_ids(call["expect"]), (case["source"], call)


@pytest.fixture(scope="module", params=DENSE)
def pair60k(request):
    return Pair(corpus=mg.Corpus.synthetic(60_000, seed=42), dense_threshold=request.param)


def _letter_grams(pair):
    c = pair.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    g = [i for i in range(c.n_grams) if b" " not in c.gram(i)]
    return c, sizes, sorted(g, key=lambda i: -sizes[i])


def test_set_algebra_random_vs_oracle(pair60k):
    c, sizes, grams = _letter_grams(pair60k)
    rng = np.random.default_rng(1)
    pool = grams[:200]
    all_docs = np.arange(1, c.n_docs + 1, dtype=np.uint32)
    for it in range(30):
        k = int(rng.integers(1, 6))
        terms = [c.gram(int(g)).decode() for g in rng.choice(pool, size=k, replace=False)]
        if it % 5 == 0:
            terms.append("zq")  # maybe unknown / very rare
        for limit, reverse in [(0, False), (7, False), (7, True), (0, True)]:
            assert pair60k.dev.search_and(terms, limit, reverse).tolist() == \
                pair60k.oidx.search_and(terms, limit, reverse).tolist(), (terms, limit, reverse)
        assert pair60k.dev.search_or(terms).tolist() == pair60k.oidx.search_or(terms).tolist(), terms
        sub = all_docs[:: int(rng.integers(1, 5))]
        assert pair60k.dev.search_not(sub, terms).tolist() == pair60k.oidx.search_not(sub, terms).tolist(), terms
        for th in range(0, len(terms) + 2):
            assert pair60k.dev.search_by_threshold(terms + terms[:1], th).tolist() == \
                pair60k.oidx.search_by_threshold(terms + terms[:1], th).tolist(), (terms, th)
        cand = rng.integers(0, c.n_docs + 50, size=int(rng.integers(1, 1500))).astype(np.uint32)
        assert pair60k.dev.filter_by_ngrams(cand, terms[:2]).tolist() == \
            pair60k.oidx.filter_by_ngrams(cand, terms[:2]).tolist(), terms


@pytest.fixture(params=["wave-kernel", "block-kernel"])
def score_kernel(request, monkeypatch):
    """SORT _score batches run on the wave-autonomous kernel when every query is flat; the general workgroup
    kernel is kept covered by forcing it."""
    if request.param == "block-kernel":
        monkeypatch.setenv("MGX_FORCE_BLOCK_KERNEL", "1")
    return request.param


def test_scored_batch_bit_exact(pair60k, score_kernel):
    c, sizes, grams = _letter_grams(pair60k)
    rng = np.random.default_rng(2)
    queries = []
    for it in range(48):
        k = int(rng.integers(1, 5))
        pool = grams[:60] if it % 3 else grams[:400]
        terms = [c.gram(int(g)).decode() for g in rng.choice(pool, size=k, replace=False)]
        queries.append(Query(terms, sort_score=True, limit=int(rng.choice([1, 10, 37, 100])),
                             offset=int(rng.choice([0, 0, 3, 50])), descending=bool(it % 4 != 3)))
    queries.append(Query(["th", "zz"], sort_score=True, limit=10))       # unknown gram -> empty before the device
    queries.append(Query(["TH", "He", "IN"], sort_score=True, limit=10))  # normalisation: ASCII lower
    pair60k.check(queries)


def test_not_terms_filters_and_funnel(pair60k, score_kernel):
    c, sizes, grams = _letter_grams(pair60k)
    rng = np.random.default_rng(3)
    f_even = pair60k.add_filter(range(2, c.n_docs + 1, 2))
    f_mod5 = pair60k.add_filter(range(5, c.n_docs + 1, 5))
    f_none = pair60k.add_filter([])
    queries = []
    for it in range(40):
        terms = [c.gram(int(g)).decode() for g in rng.choice(grams[:80], size=int(rng.integers(1, 4)), replace=False)]
        nots = [c.gram(int(g)).decode() for g in rng.choice(grams[:300], size=int(rng.integers(0, 3)), replace=False)]
        if it % 7 == 0:
            nots.append("qj")  # unknown gram in a NOT term: dropped
        if it % 6 == 0:
            nots.append(terms[0] + "e")  # multi-gram NOT term
        filters = [[], [(f_even, False)], [(f_mod5, True)], [(f_even, False), (f_mod5, False)], [(f_none, True)],
                   [(f_none, False)]][it % 6]
        queries.append(Query(terms, nots, filters, sort_score=bool(it % 2), limit=int(rng.choice([5, 20, 0]) or 5)
                             if it % 2 else int(rng.choice([0, 5, 20])), descending=bool(it % 3)))
    pair60k.check(queries)


def test_multi_gram_terms_docid_order(pair60k):
    # config-3 style: terms longer than one n-gram are the AND of their grams (verify_text off), no scoring
    c = pair60k.dev.columns
    words = sorted({w for i in range(300) for w in pair60k.corpus.text(i).decode().split(" ") if len(w) >= 3})
    rng = np.random.default_rng(4)
    queries = [Query([str(w) for w in rng.choice(words, size=int(rng.integers(1, 4)), replace=False)],
                     limit=int(rng.choice([0, 10])), descending=bool(i % 2)) for i in range(30)]
    pair60k.check(queries)


def test_boolean_expressions(pair60k):
    """ExecuteWithBooleanAst / EvaluateBooleanAstExpanded shapes: nested AND/OR/NOT over terms, then NOT terms and
    filters (full-pipeline vectors of search_pipeline_test.cpp:1122-1348 in test_pipeline_fixture)."""
    c, sizes, grams = _letter_grams(pair60k)
    rng = np.random.default_rng(9)
    words = sorted({w for i in range(300) for w in pair60k.corpus.text(i).decode().split(" ") if len(w) >= 3})
    f_even = pair60k.add_filter(range(2, c.n_docs + 1, 2))

    def term():
        if rng.random() < 0.3:
            return str(rng.choice(words))
        if rng.random() < 0.1:
            return "zq"  # unknown gram: an empty set inside the tree
        return c.gram(int(rng.choice(grams[:150]))).decode()

    def tree(depth):
        r = rng.random()
        if depth == 0 or r < 0.25:
            return term()
        if r < 0.5:
            return ("not", tree(depth - 1))
        return (("and", "or")[int(rng.integers(0, 2))],) + tuple(tree(depth - 1) for _ in range(int(rng.integers(2, 4))))

    queries = []
    for it in range(40):
        nots = [c.gram(int(rng.choice(grams[:200]))).decode()] if it % 4 == 0 else []
        filters = [(f_even, bool(it % 2))] if it % 5 == 0 else []
        universe = None if it % 3 else (1, int(c.n_docs * 0.7))
        queries.append(Query(expr=tree(3), not_terms=nots, filters=filters, limit=int(rng.choice([0, 17])),
                             descending=bool(it % 2), universe=universe))
    pair60k.check(queries)


def test_pipeline_fixture_boolean_vectors():
    # tests/server/search_pipeline_test.cpp:916-951 fixture; :1122-1134 "basics OR cats" -> [d1,d3];
    # :1321-1334 "(basics OR cats)" + and_terms ["old"] -> [d3]; :1336-1348 "basics OR cats NOT old" -> [d1]
    p = Pair(docs=[(1, "machine learning basics"), (2, "deep learning techniques"), (3, "old article about cats")],
             ngram=2, kanji=0, cross=False)
    got = p.dev.search_batch([
        Query(expr=("or", "basics", "cats"), limit=0, descending=False),
        Query(expr=("and", ("or", "basics", "cats"), "old"), limit=0, descending=False),
        Query(expr=("or", "basics", ("and", "cats", ("not", "old"))), limit=0, descending=False),
    ])
    assert [g.docs.tolist() for g in got] == [[1, 3], [3], [1]]


def test_fuzzy_threshold_terms(pair60k):
    """ExecuteWithFuzzy against the oracle's restatement of search_pipeline.cpp:1659-1744 (pinned by the reference's
    FUZZY vectors in tests/golden/fuzzy.json): theta / n_eff come from the oracle, not from the code under test; typos,
    several terms in the given order (no size sort), unknown grams skipped, NOT terms, funnel counters."""
    words = sorted({w for i in range(400) for w in pair60k.corpus.text(i).decode().split(" ") if len(w) >= 5})
    rng = np.random.default_rng(21)

    def typo(w):
        k = int(rng.integers(0, len(w)))
        return w[:k] + "q" + w[k + 1:]

    cases = []
    for i, w in enumerate(words[:24]):
        terms = [w if i % 3 else typo(w)]
        if i % 4 == 0:
            terms.append(words[(i * 7 + 3) % len(words)])
        cases.append((terms, 1 + (i % 5 == 0), [words[(i + 11) % len(words)][:2]] if i % 6 == 1 else []))
    cases.append((["zzqqzz"], 1, []))         # no known gram at all
    cases.append((["thequick", "qq"], 2, []))
    queries = [Query(t, nt, limit=0, descending=False, fuzzy=d) for t, d, nt in cases]
    got = pair60k.dev.search_batch(queries)
    for (terms, d, nts), g in zip(cases, got):
        r = O.execute_fuzzy(pair60k.oidx, pair60k.ostore, terms, d, not_terms=nts, ngram_size=2, kanji_ngram_size=0,
                            cross_boundary=True)
        assert g.docs.tolist() == r["results"].tolist(), (terms, d, nts, r["thetas"])
        assert g.total == len(r["results"])
        if not r["empty_term_detected"]:
            for k in ("total_candidates", "after_intersection", "after_not", "after_filters"):
                assert getattr(g, k) == r[k], (terms, k)


def test_fuzzy_cjk_effective_ngram_size():
    """n_eff of a CJK-dominant term is the kanji size (search_pipeline.cpp:1685-1698): hybrid index (ascii 2, kanji 1),
    the reference's FuzzyCjk corpus shape."""
    texts = ["私は東京都に住む", "東京市役所", "大阪府大阪市", "東西南北", "tokyo tower 東京", "京都の市", "abc def"]
    p = Pair(docs=list(enumerate(texts, start=1)), ngram=2, kanji=1, cross=False)
    for terms, d in [(["東京市"], 1), (["東京都庁"], 1), (["大阪市"], 2), (["tokyo"], 1), (["東京", "市"], 1)]:
        got = p.dev.search_batch([Query(terms, limit=0, descending=False, fuzzy=d)])[0]
        r = O.execute_fuzzy(p.oidx, p.ostore, terms, d, ngram_size=2, kanji_ngram_size=1, cross_boundary=False)
        assert got.docs.tolist() == r["results"].tolist(), (terms, d, r["thetas"])


@pytest.mark.parametrize("ngram", [5, 2], ids=["tf-column", "text-level"])
def test_bm25_golden_vectors_on_device(ngram):
    bm = G.load("bm25.json")
    for v in bm["score_properties"]:
        first, texts = densify([(d, t) for d, t in v["docs"]])
        # ngram 5: "hello"/"alpha" are one 5-gram each (tf column); ngram 2: they span four bigrams (tf from the text)
        idx = mg.Index(texts=texts, first_doc_id=first, ngram_size=ngram)
        if v["assert"] == "raises":
            with pytest.raises(mg._capi.MgxError):
                idx.score_documents(v["candidates"], v["terms"], v["dfs"], v["N"], v["avgdl"], v["k1"], v["b"])
            continue
        score = idx.score_documents(v["candidates"], v["terms"], v["dfs"], v["N"], v["avgdl"], v["k1"], v["b"])
        assert eval(v["assert"], {"score": score, "abs": abs, "idf": O.compute_idf}), (v["id"], score)
    idx = mg.Index(texts=["x"], ngram_size=1)
    for v in bm["sort_by_score"]:
        got = idx.sort_by_score(v["results"], v["scores"], v["desc"], v["limit"], v["offset"])
        assert got.tolist() == v["expect"], v


def test_score_documents_unknown_and_textless_candidates():
    # bm25_scorer_test.cpp:194-230: unknown candidates and docs without the term score 0.0 and are kept, in order
    texts = ["alpha beta" if i % 3 == 0 else "gamma delta" for i in range(2051)]
    idx = mg.Index(texts=texts, ngram_size=5)
    cands = list(range(1, 2052)) + [9_000_000]
    score = idx.score_documents(cands, ["alpha"], [684], 2051, 2.0)
    assert len(score) == len(cands)
    assert all(score[i] > 0.0 for i in range(0, 2051, 3))
    assert score[1] == 0.0 and score[-1] == 0.0


def test_ties_break_on_docid_both_directions(score_kernel):
    # every doc identical => all scores tie; DESC keeps larger docids first, ASC smaller first (result_sorter.cpp:681-686)
    p = Pair(docs=[(i, "abab cd") for i in range(1, 40_001)])
    got = p.check([Query(["ab", "cd"], sort_score=True, limit=10, descending=True),
                   Query(["ab", "cd"], sort_score=True, limit=10, descending=False),
                   Query(["ab"], sort_score=True, limit=1000, offset=24, descending=True)])
    assert got[0].docs.tolist() == list(range(40_000, 39_990, -1))
    assert got[1].docs.tolist() == list(range(1, 11))


def test_spans_many_tiles_and_workgroups(score_kernel):
    # 1.2M docs = 74 tiles of 16384 = 2 workgroup items (64 tiles each) per query
    corpus = mg.Corpus.synthetic(1_200_000, seed=5)
    p = Pair(corpus=corpus)
    c, sizes, grams = _letter_grams(p)
    rng = np.random.default_rng(6)
    qs = [Query([c.gram(int(g)).decode() for g in rng.choice(grams[:40], size=3, replace=False)], sort_score=True,
                limit=10) for _ in range(6)]
    qs += [Query([c.gram(int(g)).decode() for g in rng.choice(grams[100:300], size=2, replace=False)],
                 limit=25, descending=True) for _ in range(4)]
    p.check(qs)


# ---- N1: text-level BM25 terms (terms that are not exactly one n-gram) ------------------------------------------------

def _words(pair, n_docs=400, min_len=3):
    return sorted({w for i in range(n_docs) for w in pair.corpus.text(i).decode().split(" ") if len(w) >= min_len})


def test_text_level_terms_scored_bit_exact(pair60k):
    # SEARCH "word" AND "other" SORT _score: candidates = AND of each term's grams (verify_text off), tf =
    # CountTermOccurrences over the text, df = candidates whose text contains the term (search_pipeline.cpp:542-565)
    c, sizes, grams = _letter_grams(pair60k)
    words = _words(pair60k)
    rng = np.random.default_rng(11)
    queries = []
    for it in range(36):
        k = int(rng.integers(1, 4))
        terms = [str(w) for w in rng.choice(words, size=k, replace=False)]
        if it % 3 == 0:  # mixed: one single-gram term scored from its tf column next to the text-level ones
            terms.append(c.gram(int(rng.choice(grams[:40]))).decode())
        if it % 5 == 0:  # a prefix of a word: n-gram candidates that are real, overlapping occurrences possible
            terms.append(terms[0][:3])
        queries.append(Query(terms, sort_score=True, limit=int(rng.choice([1, 10, 50])),
                             offset=int(rng.choice([0, 0, 2])), descending=bool(it % 4 != 3)))
    # n-gram false positives: "aaa" on a bigram index has the single gram "aa" but is not that gram
    queries.append(Query(["the", "and"], sort_score=True, limit=10))
    queries.append(Query(["THE", "And"], sort_score=True, limit=10))  # normalisation
    pair60k.check(queries)


def test_text_level_false_positives_and_overlaps():
    # "aba" (grams ab, ba): doc 3 has both grams but not the term -> stays in the result (verify_text off) with the
    # term's tf 0; "aa" inside "aaaa" counts 2 non-overlapping occurrences; repeated-gram term "aaa" -> gram "aa"
    docs = [(1, "aba cd"), (2, "ababa cd"), (3, "ab ba cd"), (4, "aaaa cd"), (5, "xaaax aba"), (6, "cd cd"),
            (7, "aaa aaa aba")]
    p = Pair(docs=docs)
    p.check([Query(["aba"], sort_score=True, limit=10),
             Query(["aba", "cd"], sort_score=True, limit=10),
             Query(["aaa"], sort_score=True, limit=10),
             Query(["aaa", "aba"], sort_score=True, limit=10, descending=False),
             Query(["aba"], ["cd"], sort_score=True, limit=10)])


def test_text_level_terms_cjk_trigram():
    # config-3 shape: 3-gram index over ideographs, terms of 3-6 code points
    rng = np.random.default_rng(12)
    chars = [chr(0x4E00 + i) for i in range(40)] + [chr(0x3042 + i) for i in range(8)]
    docs = [(i + 1, "".join(rng.choice(chars, size=int(rng.integers(8, 40))))) for i in range(3000)]
    p = Pair(docs=docs, ngram=3, kanji=3)
    qs = []
    for it in range(30):
        d = docs[int(rng.integers(0, len(docs)))][1]
        terms = []
        for _ in range(int(rng.integers(1, 3))):
            L = int(rng.integers(3, 7))
            s = int(rng.integers(0, max(1, len(d) - L)))
            terms.append(d[s:s + L])
        qs.append(Query(terms, sort_score=True, limit=10, descending=bool(it % 2)))
    p.check(qs)


def test_text_level_spans_many_tiles():
    corpus = mg.Corpus.synthetic(300_000, seed=9)
    p = Pair(corpus=corpus)
    words = _words(p, 200, 4)
    rng = np.random.default_rng(13)
    qs = [Query([str(w) for w in rng.choice(words, size=2, replace=False)], sort_score=True, limit=10)
          for _ in range(6)]
    p.check(qs)


# ---- docid-ordered pages (no SORT _score): count pass + page pass, no result bitmap ------------------------------------

def test_docid_pages_dense_and_sparse(score_kernel):
    # the reference's default order: primary key DESC with a limit (result_sorter.cpp:502-510). Dense queries have their
    # page inside one tile; sparse ones (rare grams, multi-gram words) spread it over hundreds of mostly empty tiles.
    corpus = mg.Corpus.synthetic(1_200_000, seed=7)
    p = Pair(corpus=corpus)
    c, sizes, grams = _letter_grams(p)
    words = _words(p, 300, 5)
    rng = np.random.default_rng(21)
    qs = []
    for it in range(10):
        qs.append(Query([c.gram(int(g)).decode() for g in rng.choice(grams[:30], size=2, replace=False)],
                        limit=int(rng.choice([1, 10, 100, 1000])), descending=bool(it % 2)))
    for it in range(10):
        qs.append(Query([c.gram(int(g)).decode() for g in rng.choice(grams[450:], size=2, replace=False)],
                        limit=int(rng.choice([3, 50, 500])), descending=bool(it % 2)))
    for it in range(8):
        qs.append(Query([str(w) for w in rng.choice(words, size=2, replace=False)], limit=20, descending=bool(it % 2)))
    qs.append(Query([c.gram(int(grams[0])).decode()], limit=16384, descending=True))    # largest page of the page path
    qs.append(Query([c.gram(int(grams[0])).decode()], limit=16385, descending=False))   # one more: materialising path
    qs.append(Query([c.gram(int(grams[600])).decode(), c.gram(int(grams[5])).decode()], limit=0))  # everything
    qs.append(Query(["th"], ["he"], limit=7, descending=True))
    qs.append(Query(expr=("and", ("or", "th", "qu"), ("not", "he")), limit=15))
    p.check(qs)


def test_saturated_tf_and_long_docs(score_kernel):
    # dense grams whose tf exceeds the 4-bit nibble (>= 15 -> exact posting lookup), docs longer than the 8-bit doc
    # length (>= 255 -> u32 doc_len) and than the BM25 table (direct formula), tf above the table's rows
    docs = []
    for i in range(1, 40_001):
        reps = i % 27 + 1                       # tf("ab") = 1..27
        pad = "x" * ((i % 9) * 45)              # doc lengths up to ~420
        docs.append((i, "ab" * reps + " cd " + ("ef " * (i % 4)) + pad))
    p = Pair(docs=docs)
    p.check([Query(["ab", "cd"], sort_score=True, limit=25),
             Query(["ab"], sort_score=True, limit=100, descending=False),
             Query(["cd", "ef", "ab"], sort_score=True, limit=10, offset=5),
             Query(["xx", "ab"], sort_score=True, limit=10)])


def test_sort_by_score_long_arrays():
    # ResultSorter::SortByScore over more entries than the rank-by-counting kernel takes: bounded pages go through the
    # per-wave top-k scan + merge, an unbounded sort through the full device sort
    idx = mg.Index(texts=["x"], ngram_size=1)
    rng = np.random.default_rng(31)
    n = 300_000
    docs = rng.permutation(np.arange(1, n + 1, dtype=np.uint32))
    scores = np.round(rng.random(n) * 50.0, 1)  # many ties: docid decides
    for desc, limit, offset in [(True, 10, 0), (False, 10, 0), (True, 100, 37), (False, 1000, 24), (True, 1, 0)]:
        got = idx.sort_by_score(docs, scores, desc, limit, offset)
        want = O.sort_by_score(docs, scores, desc, limit, offset)
        assert got.tolist() == want.tolist(), (desc, limit, offset)
    assert idx.sort_by_score(docs, scores, True, 0, 0).tolist() == O.sort_by_score(docs, scores, True, 0, 0).tolist()


# ---- exact-text post-filter (PostFilterByText) ------------------------------------------------------------------------

def test_pipeline_execute_vectors_on_device():
    # every Execute vector of tests/server/search_pipeline_test.cpp on the device, among them
    # MixedScriptBoundaryFragmentRequiresExactTextMatch (:1492-1512): "京タ" must not match "京都"
    for v in G.load("pipeline.json")["execute"]:
        ic, qp = v["index"], v["query_params"]
        # the device index is built with the index-side settings; the query side uses the query parameters
        p = Pair(docs=[(d, t) for d, t in v["docs"]], ngram=ic["ngram"], kanji=ic["kanji"], cross=True)
        p.dev.kanji_ngram_size, p.dev.cross_boundary = qp["kanji"], qp["cross_boundary"]
        filters = [(p.add_filter(f["docs"]), f["negate"]) for f in v["filters"]]
        for desc in (False, True):
            q = Query(v["terms"], v["not_terms"], filters, limit=0, descending=desc,
                      verify_text=v.get("verify_text", False))
            g = p.dev.search_batch([q])[0]
            want = v["expect_results"][::-1] if desc else v["expect_results"]
            if g.empty_term_detected:
                assert want == [], v["id"]
                continue
            assert g.docs.tolist() == want, (v["id"], g.docs.tolist())
            for k, w in v.get("expect_funnel", {}).items():
                assert getattr(g, k) == w, (v["id"], k)


def test_exact_text_filter_all_modes(score_kernel):
    # hybrid index (bigrams, kanji unigrams, no cross-boundary n-grams): a mixed-script term with an uncovered code
    # point is filtered by exact text in docid pages (count + page pass), full results and SORT _score alike
    rng = np.random.default_rng(41)
    kana, kanji = "タワーカナ", "京都東大阪"
    docs = []
    for i in range(1, 30_001):
        body = "".join(rng.choice(list(kanji + kana), size=int(rng.integers(2, 9))))
        docs.append((i, body + (" 京タワー" if i % 97 == 0 else "") + (" 京都" if i % 5 == 0 else "")))
    p = Pair(docs=docs, ngram=2, kanji=1, cross=False)
    qs = [Query(["京タ"], limit=10, descending=True), Query(["京タ"], limit=0), Query(["京タ", "都"], limit=50),
          Query(["京タ"], sort_score=True, limit=10), Query(["京タ", "ワー"], sort_score=True, limit=5, descending=False)]
    got = p.check(qs)
    assert all(g.total > 0 for g in got[:2])
    # the caller's verify_text decision: n-gram false positives ("ab" and "bc" present, "abc" absent) are dropped
    p2 = Pair(docs=[(1, "abc"), (2, "ab bc"), (3, "xabcx bc"), (4, "bcab")])
    g = p2.dev.search_batch([Query(["abc"], limit=0, descending=False),
                             Query(["abc"], limit=0, descending=False, verify_text=True),
                             Query(["abc", "bc"], limit=10, verify_text=True)])
    assert g[0].docs.tolist() == [1, 2, 3, 4] and g[1].docs.tolist() == [1, 3] and g[1].after_filters == 4
    assert g[2].docs.tolist() == [3, 1] and g[2].total == 2


def test_batch_object_reused_for_fresh_batches(pair60k):
    """mgx_batch_reset: a serving loop re-prepares ONE batch object for every new set of queries (arenas and pinned
    result blocks are kept, inputs travel with the execute). Shapes change from step to step — scored / docid pages /
    NOT terms, batch sizes up and down — and every step is checked against the oracle."""
    c, sizes, grams = _letter_grams(pair60k)
    rng = np.random.default_rng(77)
    batch = None
    for step in range(8):
        n = [5, 40, 3, 64, 1, 17, 33, 2][step]
        queries = []
        for i in range(n):
            k = int(rng.integers(1, 4))
            terms = [c.gram(int(g)).decode() for g in rng.choice(grams[:150], size=k, replace=False)]
            if (step + i) % 3 == 0:
                queries.append(Query(terms, limit=int(rng.choice([1, 10, 100])), descending=bool(i % 2)))
            elif (step + i) % 3 == 1:
                queries.append(Query(terms, sort_score=True, limit=int(rng.choice([1, 10, 50])), offset=i % 4))
            else:
                nt = [c.gram(int(rng.choice(grams[:300]))).decode()]
                queries.append(Query(terms, nt, sort_score=True, limit=10))
        batch = pair60k.dev.prepare(queries, into=batch)
        for _ in range(2):  # a prepared batch still executes repeatedly
            batch.execute()
            got = batch.fetch()
        for q, g in zip(queries, got):
            total, page, scores, r = pair60k.oracle_query(q)
            assert g.total == total and g.docs.tolist() == page.tolist(), (step, q.terms)
            if scores is not None:
                assert np.array_equal(g.scores, scores), (step, q.terms)


def test_search_and_limit_max_u32_does_not_hang(pair60k):
    """SearchAnd(terms, UINT32_MAX) must neither wrap the page size nor spin (index.cpp:356-367 semantics:
    a limit beyond the result size returns everything)."""
    c, sizes, grams = _letter_grams(pair60k)
    terms = [c.gram(int(grams[3])).decode(), c.gram(int(grams[9])).decode()]
    want = pair60k.oidx.search_and(terms, 0, False).tolist()
    assert pair60k.dev.search_and(terms, 0xFFFFFFFF, False).tolist() == want
    assert pair60k.dev.search_and(terms, 0xFFFFFFFF, True).tolist() == want[::-1]
    assert pair60k.dev.search_and(terms, 1 << 40, False).tolist() == want


def test_deep_offset_pages_full_sort(pair60k):
    """R15 beyond the fused top-k: OFFSET 10000 LIMIT 100 (the reference's own deep-paging benchmark,
    docs/releases/v1.3.5.md:238), offset+limit just past 1024, an unbounded page, ASC order — through the batched entry:
    every match scored, ResultSorter::SortByScore by a full device sort. Bit-exact docids and scores."""
    c, sizes, grams = _letter_grams(pair60k)
    top = [c.gram(int(g)).decode() for g in grams[:6]]
    queries = [
        Query([top[0], top[1]], sort_score=True, limit=100, offset=10000),
        Query([top[0], top[2]], sort_score=True, limit=100, offset=10000, descending=False),
        Query([top[1], top[3], top[4]], sort_score=True, limit=1000, offset=25),       # 1025 > kMaxNeeded
        Query([top[2], top[5]], sort_score=True, limit=0, offset=0),                   # every match, ranked
        Query([top[0]], [top[1]], sort_score=True, limit=50, offset=3000),             # with a NOT term
        Query([top[3], top[4]], sort_score=True, limit=10, offset=10_000_000),         # past the end: empty page
        Query([top[0], top[1]], sort_score=True, limit=10),                            # a fused page in the same batch
    ]
    got = pair60k.check(queries)
    assert len(got[0].docs) == 100 and got[0].total > 10100
    assert len(got[3].docs) == got[3].total
    assert len(got[5].docs) == 0 and got[5].total > 0


def test_sort_by_score_any_length(pair60k):
    """ResultSorter::SortByScore stand-alone (result_sorter.cpp:661-716) on arrays the old path refused: 200,000 and
    70,000 entries with many ties, deep offsets and unbounded pages, both orders."""
    rng = np.random.default_rng(12)
    for n in (200_000, 70_000, 5_000, 4_097):
        docs = rng.permutation(np.arange(1, n + 1, dtype=np.uint32))
        scores = np.round(rng.random(n) * 50) / 7.0  # ~50 distinct values: ties everywhere
        for desc, limit, offset in [(True, 100, 10000), (False, 0, 0), (True, 0, n - 10), (False, 1000, 1500),
                                    (True, 10, 0), (True, 5, n + 3)]:
            want = O.sort_by_score(docs, scores, desc, limit, offset)
            got = pair60k.dev.sort_by_score(docs, scores, desc, limit, offset)
            assert got.tolist() == want.tolist(), (n, desc, limit, offset)


@pytest.mark.parametrize("dense", DENSE)
def test_tf_beyond_the_byte_column(dense, monkeypatch):
    """A document that repeats an n-gram 300 times (tf byte saturated at 255, doc length past the dl8 escape) scores as
    the reference scores it: the true count comes from the overflow table — fused wave kernel, general kernel, stand-alone
    ScoreDocuments."""
    rng = np.random.default_rng(5)
    words = ["ab", "cd", "abcd", "xy", "cdab", "zz"]
    texts = [" ".join(rng.choice(words, size=int(rng.integers(2, 9)))) for _ in range(3000)]
    texts[17] = "ab" * 300 + " cd"
    texts[1234] = "cd " + "ab" * 255
    texts[2000] = "ab" * 254 + " cd cd"
    texts[2999] = "cd" * 400 + "ab"
    pair = Pair(docs=list(enumerate(texts, start=1)), dense_threshold=dense)
    qs = [Query(["ab", "cd"], sort_score=True, limit=20), Query(["ab"], sort_score=True, limit=5),
          Query(["cd", "ab"], sort_score=True, limit=20, descending=False), Query(["ab", "cd"], sort_score=True, limit=50, offset=2000)]
    got = pair.check(qs)
    assert {18, 1235, 2001, 3000} & set(got[2].docs.tolist())  # (the long docs rank last: they lead the ASC page)
    monkeypatch.setenv("MGX_FORCE_BLOCK_KERNEL", "1")
    pair.check(qs)
    monkeypatch.delenv("MGX_FORCE_BLOCK_KERNEL")
    cand = np.asarray([17, 18, 1235, 2001, 3000, 5], dtype=np.uint32)
    dfs = [pair.dev.posting_size("ab"), pair.dev.posting_size("cd")]
    want = O.score_documents(pair.ostore, cand, ["ab", "cd"], dfs, pair.N, pair.avgdl)
    assert np.array_equal(pair.dev.score_documents(cand, ["ab", "cd"], dfs, pair.N, pair.avgdl), want)


def test_many_operands_and_many_scored_terms(pair60k, monkeypatch):
    """The reference takes 64 AND terms / 64 NOT terms / 64 filters per query (query_parser.h:270-272) and terms of
    ~127 n-grams: a 60-gram term (one conjunction of 60 lists), 20 scored terms, 30 NOT terms. Only sorted-list and
    scored operands occupy LDS in the general kernel, so these shapes run instead of being refused."""
    c, sizes, grams = _letter_grams(pair60k)
    docs = [pair60k.corpus.text(i).decode() for i in range(0, 60_000, 601)]
    long_terms = [d[:61] for d in docs if len(d) >= 61][:3]          # 60 bigrams each, matches at least its doc
    many = [c.gram(int(g)).decode() for g in grams[:24]]
    qs = [Query([t], limit=10) for t in long_terms]
    qs += [Query([long_terms[0][:40]], sort_score=True, limit=5)]       # text-level term of 39 grams
    qs += [Query(many[:20], sort_score=True, limit=10),                 # 20 scored single-gram terms
           Query(many[:5], sort_score=True, limit=10),                  # 5 scored terms
           Query(many[:2], [c.gram(int(g)).decode() for g in grams[200:230]], limit=20),  # 30 NOT terms
           Query(many[:4], sort_score=True, limit=10)]
    pair60k.check(qs)
    monkeypatch.setenv("MGX_FAST_PATH", "1")   # the 4-term query on and_score_kernel<4>
    pair60k.check(qs[-3:])


def test_index_from_reference_dump_answers_like_the_postings():
    """N2: an index created from an MGIX dump (tests/golden/mgix_v4.bin: delta lists + every Roaring container kind)
    answers AND / OR / NOT and docid pages exactly like plain set algebra over the same posting lists."""
    import json
    root = os.path.dirname(os.path.abspath(__file__))
    exp = json.load(open(os.path.join(root, "golden", "mgix_expected.json"), encoding="utf-8"))
    idx = mg.Index.from_mgix(open(os.path.join(root, "golden", "mgix_v4.bin"), "rb").read())
    c = idx.columns
    sets = {}
    for t in exp["files"]["mgix_v4.bin"]["terms"]:
        gid = c.lookup(t)
        sets[t] = set(c.docids[int(c.offsets[gid]):int(c.offsets[gid + 1])].tolist())
    di = idx.device_index
    cases = [(["ab", "bc"], []), (["ab", "cd", "qr"], []), (["cd", "ef"], []), (["ab"], ["cd"]), (["bc", "qr"], ["ab"]),
             (["de"], []), (["zz", "ab"], [])]
    qs = [mg.engine.Query(a, n, limit=50) for a, n in cases] + [mg.engine.Query(a, n, limit=50, descending=False) for a, n in cases]
    got = idx.search_batch(qs)
    for q, g in zip(qs, got):
        want = set.intersection(*[sets[t] for t in q.terms])
        for t in q.not_terms:
            want -= sets[t]
        want = sorted(want, reverse=q.descending)
        assert g.total == len(want) and g.docs.tolist() == want[:50], (q.terms, q.not_terms, q.descending)
    union = mg.ops.search_or(di, [c.lookup("de"), c.lookup("zz"), c.lookup("bc")]) if hasattr(mg, "ops") else None
    if union is not None:
        assert union.tolist() == sorted(sets["de"] | sets["zz"] | sets["bc"])
    with pytest.raises(mg._capi.MgxError):  # no tf / doc_len in a dump: SORT _score is refused, not guessed
        idx.search_batch([mg.engine.Query(["ab", "bc"], sort_score=True, limit=10)])


def test_terms_shorter_than_one_ngram_take_the_substring_fallback():
    """SearchTermDocuments' other branch (search_pipeline.cpp:438-446): a term that yields no n-gram — one ASCII letter
    on a bigram index, one kana on a hybrid index — matches by substring over the stored texts
    (query::SearchNormalizedSubstring); positive terms (alone, first, later), NOT terms, scored (df 0), docid pages."""
    rng = np.random.default_rng(12)
    words = ["alpha", "beta", "gamma", "delta", "omega", "zeta", "x", "q", "か", "き", "東京", "京都", "大阪"]
    texts = [" ".join(str(w) for w in rng.choice(words, size=int(rng.integers(2, 9)))) for _ in range(6000)]
    p = Pair(docs=list(enumerate(texts, start=1)), ngram=2, kanji=1)
    Q = mg.engine.Query
    qs = [Q(["x"], limit=30), Q(["q", "alpha"], limit=30), Q(["alpha", "x"], limit=30), Q(["か"], limit=30),
          Q(["東京", "か"], limit=30), Q(["beta"], ["x"], limit=30), Q(["x"], ["q"], limit=30), Q(["x", "q", "か"], limit=30),
          Q(["alpha", "x"], sort_score=True, limit=15), Q(["x"], sort_score=True, limit=15),
          Q(["か", "東京"], sort_score=True, limit=15), Q(["x"], limit=25, descending=False), Q(["z"], limit=5),
          Q(["alpha"], ["w"], limit=30)]
    got = p.dev.search_batch(qs)
    hits = 0
    for q, g in zip(qs, got):
        total, page, scores, r = p.oracle_query(q)
        assert g.total == total and g.docs.tolist() == page.tolist(), (q.terms, q.not_terms)
        if q.sort_score:
            assert np.array_equal(g.scores, scores), q.terms
        elif not r["empty_term_detected"]:
            for k in ("total_candidates", "after_intersection", "after_not", "after_filters"):
                assert getattr(g, k) == r[k], (q.terms, k)
        hits += total > 0
    assert hits >= 11


def test_pruning_keeps_ties_and_page_boundaries_exact():
    """Block-max pruning under adversarial ties: 400k docs drawn from 40 distinct texts, so thousands of docs share every
    score and the page boundary falls inside a tie class (the docid decides); quarters whose bound EQUALS the k-th best
    score must be kept. Pages at several offsets, both orders, 1-3 scored terms, against the oracle."""
    rng = np.random.default_rng(77)
    words = ["alpha", "beta", "gamma", "delta", "kappa", "sigma", "omega"]
    protos = [" ".join(str(w) for w in rng.choice(words, size=int(rng.integers(1, 7)))) for _ in range(40)]
    texts = [protos[int(i)] for i in rng.integers(0, len(protos), size=400_000)]
    p = Pair(docs=list(enumerate(texts, start=1)), ngram=2, kanji=0)
    Q = mg.engine.Query
    qs = []
    for terms in (["al"], ["ma", "ga"], ["ta", "be", "et"], ["ka", "pp"], ["om", "eg", "ga"], ["si", "gm"]):
        for limit, offset in ((10, 0), (10, 5), (37, 0), (100, 11), (1, 0)):
            qs.append(Q(terms, sort_score=True, limit=limit, offset=offset))
        qs.append(Q(terms, sort_score=True, limit=10, descending=False))
    got = p.dev.search_batch(qs)
    for q, g in zip(qs, got):
        total, page, scores, _ = p.oracle_query(q)
        assert g.total == total and g.docs.tolist() == page.tolist(), (q.terms, q.limit, q.offset, q.descending)
        assert np.array_equal(g.scores, scores), (q.terms, q.limit, q.offset)
