"""Pins the CPU oracle (oracle/mygram_oracle.c) against the reference's own known-answer tests.

Every expectation here is data transcribed from /root/reference/tests (see the 'source' field of each
fixture). Runs on CPU only.
"""
import json
import math
import os

import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O


def _build(case):
    ic = case["index"]
    idx = O.Index(ic["ngram"], ic.get("kanji", 0), ic.get("cross_boundary", True))
    for doc_id, text in G.expand_docs(case["docs"]):
        idx.add_document(doc_id, text)
    return idx


def run_call(idx, call):
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


@pytest.mark.parametrize("case", G.index_cases(), ids=lambda c: c["id"])
def test_index_known_answers(case):
    idx = _build(case)
    for call in case["calls"]:
        got = run_call(idx, call)
        assert got.tolist() == G.expand_ids(call["expect"]), (case["source"], call)


def test_idf_known_answers():
    env = {"log": math.log}
    for v in G.load("bm25.json")["idf"]:
        got = O.compute_idf(v["N"], v["df"])
        if "same_as" in v:
            assert got == O.compute_idf(v["same_as"]["N"], v["same_as"]["df"])
        elif "expect_expr" in v:
            assert abs(got - eval(v["expect_expr"], env)) <= v["tol"], v
        else:
            assert got == v["expect"]
    assert O.compute_idf(1000, 1) > O.compute_idf(1000, 500)


def test_tf_known_answers():
    for v in G.load("bm25.json")["tf"]:
        assert O.count_term_occurrences(v["text"], v["term"]) == v["expect"], v


def test_sort_by_score_known_answers():
    for v in G.load("bm25.json")["sort_by_score"]:
        got = O.sort_by_score(v["results"], v["scores"], v["desc"], v["limit"], v["offset"])
        assert got.tolist() == v["expect"], v


@pytest.mark.parametrize("v", G.load("bm25.json")["score_properties"], ids=lambda v: v["id"])
def test_score_documents_properties(v):
    store = O.DocumentStore()
    for doc_id, text in v["docs"]:
        store.add(doc_id, text)
    if v["assert"] == "raises":
        with pytest.raises(ValueError):
            O.score_documents(store, v["candidates"], v["terms"], v["dfs"], v["N"], v["avgdl"], v["k1"], v["b"])
        return
    score = O.score_documents(store, v["candidates"], v["terms"], v["dfs"], v["N"], v["avgdl"], v["k1"], v["b"])
    assert eval(v["assert"], {"score": score, "abs": abs, "idf": O.compute_idf}), (v["id"], score)


def test_score_documents_chunk_boundary_order():
    cb = G.load("bm25.json")["chunk_boundary"]
    n = cb["chunk"] * 2 + 3
    store = O.DocumentStore()
    cands, scoring = [], []
    for i in range(n):
        store.add(i + 1, "alpha beta" if i % 3 == 0 else "gamma delta")
        cands.append(i + 1)
        if i % 3 == 0:
            scoring.append(i)
    cands.append(9_000_000)
    score = O.score_documents(store, cands, ["alpha"], [len(scoring)], n, 2.0)
    assert len(score) == len(cands)
    assert all(score[p] > 0.0 for p in scoring)
    assert score[1] == 0.0 and score[-1] == 0.0


def test_term_infos_and_df():
    for v in G.load("pipeline.json")["term_infos"]:
        ic = v["index"]
        idx = O.Index(ic["ngram"], ic["kanji"], True)
        store = O.DocumentStore()
        for doc_id, text in v["docs"]:
            idx.add_document(doc_id, text)
            store.add(doc_id, text)
        r = O.execute(idx, store, v["terms"], compute_df=v["compute_df"])
        for pos, exp in enumerate(v["expect"]):
            assert v["terms"][r["term_order"][pos]] == exp["term"], v["id"]
            assert r["term_df"][pos] == exp["df"], v["id"]
            if "estimated_size" in exp:
                assert r["term_estimated_size"][pos] == exp["estimated_size"], v["id"]


def test_execute_full_pipeline_fixture():
    for v in G.load("pipeline.json")["execute"]:
        ic, qp = v["index"], v["query_params"]
        idx = O.Index(ic["ngram"], ic["kanji"], True)
        store = O.DocumentStore()
        for doc_id, text in v["docs"]:
            idx.add_document(doc_id, text)
            store.add(doc_id, text)
        filters = [(f["docs"], f["negate"]) for f in v["filters"]]
        r = O.execute(idx, store, v["terms"], v["not_terms"], filters, ngram_size=qp["ngram"],
                      kanji_ngram_size=qp["kanji"], cross_boundary=qp["cross_boundary"],
                      verify_text=v.get("verify_text", False))
        assert r["results"].tolist() == v["expect_results"], v["id"]
        assert r["exact_text_applied"] == v.get("expect_exact_text", False), v["id"]
        for k, want in v.get("expect_funnel", {}).items():
            assert r[k] == want, (v["id"], k)


def test_post_filter_by_text_vectors():
    for v in G.load("pipeline.json")["post_filter_by_text"]:
        store = O.DocumentStore()
        for doc_id, text in v["docs"]:
            store.add(doc_id, text)
        assert O.post_filter_by_text(store, v["candidates"], v["terms"]).tolist() == v["expect"], v["id"]


def test_uncovered_hybrid_fragment_rule():
    # src/server/search_pipeline.cpp:80-136: only mixed-script terms with a code point outside every query n-gram
    f = O.has_uncovered_hybrid_fragment
    assert f("京タ", 2, 1, False)            # 京 is a unigram; タ is the last code point and cannot start a bigram
    assert f("京タ", 2, 1, True)             # ... whatever the cross-boundary setting
    assert f("タ京", 2, 1, False)            # the bigram タ京 crosses the script boundary: not generated
    assert not f("タ京", 2, 1, True)         # generated: both code points covered
    assert not f("京都", 2, 1, False)        # one script only
    assert not f("京タワ", 2, 1, False)      # タワ covers the kana
    assert not f("京タ", 2, 0, False)        # no hybrid mode
    assert not f("a", 2, 1, False)


def test_ngram_rules():
    # src/utils/string_utils.cpp:382-423,452-509 — code-point windows; hybrid sizes by the starting code point.
    assert O.generate_ngrams("abcd", 2) == [b"ab", b"bc", b"cd"]
    assert O.generate_ngrams("a", 2) == []
    assert O.generate_ngrams("東京都", 1) == ["東".encode(), "京".encode(), "都".encode()]
    # kana is not a CJK ideograph: bigram windows; kanji: unigram windows (ascii 2 / kanji 1)
    assert O.generate_hybrid_ngrams("東京ab", 2, 1, True) == ["東".encode(), "京".encode(), b"ab"]
    assert O.generate_hybrid_ngrams("aあ東", 2, 1, True) == ["aあ".encode(), "あ東".encode(), "東".encode()]
    assert O.generate_hybrid_ngrams("aあ東", 2, 1, False) == ["aあ".encode(), "東".encode()]
    # invalid bytes are skipped, not counted (string_utils.cpp:655-669)
    assert O.count_code_points(b"ab\xff\xfecd") == 4
    assert O.count_code_points("東京タワー") == 5


def test_search_scored_matches_manual_composition():
    docs = ["alpha beta", "beta gamma alpha alpha", "gamma", "alpha", "beta beta alpha"]
    idx, store = O.Index(2, 1, True), O.DocumentStore()
    for i, t in enumerate(docs):
        idx.add_document(i + 1, t)
        store.add(i + 1, t)
    n, total_len = store.bm25_stats()
    assert n == 5 and total_len == sum(len(t) for t in docs)
    total, top, scores = O.search_scored(idx, store, ["al", "be"], n, total_len / n, limit=3)
    r = O.execute(idx, store, ["al", "be"], compute_df=True)
    terms = [["al", "be"][i] for i in r["term_order"]]
    sc = O.score_documents(store, r["results"], terms, r["term_df"], n, total_len / n)
    want = O.sort_by_score(r["results"], sc, True, 3, 0)
    assert total == len(r["results"]) and top.tolist() == want.tolist()
    lookup = dict(zip(r["results"].tolist(), sc.tolist()))
    assert [lookup[d] for d in top.tolist()] == scores.tolist()


def test_fuzzy_contains_vectors():
    """tests/utils/edit_distance_test.cpp:86-204 ContainsFuzzyMatch known answers pin the oracle's Levenshtein /
    word-split / code-point-window restatement (src/utils/edit_distance.cpp)."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fuzzy.json")))
    for text, term, d, want in g["contains_fuzzy_match"]["cases"]:
        assert O.contains_fuzzy_match(text, term, d) == want, (text, term, d)


def test_fuzzy_pipeline_vectors():
    """tests/server/search_pipeline_test.cpp:2034-2078: ExecuteFullPipeline with FUZZY 1 and verify_text=all — the typo
    'learnig' finds the 'learning' docs; a CJK term is verified by code-point windows inside unspaced text."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fuzzy.json")))["pipeline"]
    for case in g["cases"]:
        docs = g["docs"] + case.get("extra_docs", [])
        kanji = case.get("query_kanji", g["index"]["kanji"])
        idx = O.Index(g["index"]["ngram"], kanji, g["index"]["cross_boundary"])
        store = O.DocumentStore()
        for d, t in docs:
            idx.add_document(d, t)
            store.add(d, t)
        r = O.execute_fuzzy(idx, store, case["terms"], case["distance"], verify_text=case["verify_text"])
        got = r["results"].tolist()
        if "expect" in case:
            assert got == case["expect"], case["id"]
        else:
            assert len(got) >= case["min_results"] and set(case["expect_contains"]) <= set(got), (case["id"], got)


def test_normalize_text_vectors():
    """The ICU branch of NormalizeText as the oracle restates it (unicodedata) against the reference's own vectors."""
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "normalize.json"), encoding="utf-8"))
    for v in doc["vectors"]:
        assert O.normalize_text(v["text"], v["nfkc"], v["width"], v["lower"]) == v["expected"], v
    bad = doc["invalid_utf8"]
    assert O.normalize_text(bytes.fromhex(bad["hex"]), bad["nfkc"], bad["width"], bad["lower"]) == bad["expected"]


def test_filter_oracle_against_the_reference_filter_parity_vectors():
    """oracle/filters.py (ApplyFilters / ApplyFiltersWithBitmap / facet counts restated) against the reference's own
    expectations: tests/server/search_pipeline_test.cpp:725-910, tests/server/facet_handler_test.cpp:203-258."""
    import json
    import os
    from oracle import filters as F
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "filters.json")))
    fx = g["fixture"]

    def col(name):
        vals = [tuple(v) if v is not None else None for v in fx[name]]
        return lambda d: vals[d]
    columns = {n: col(n) for n in ("status", "category", "score")}
    for c in g["cases"]:
        docs = c.get("docs", fx["docs"])
        conds = [tuple(x) for x in c["conditions"]]
        a = F.apply_filters_with_bitmap(docs, conds, columns)
        b = F.apply_filters(docs, conds, columns)
        assert a == b, c  # the reference asserts both paths agree on every one of these
        if "expect" in c:
            assert a == c["expect"], c
    for f in g["facet"]:
        lookup = lambda d, f=f: ("string", f["docs"][str(d)])
        counts = F.facet_counts(f["results"], lookup)
        shown = {k[1].decode(): v for k, v in counts.items()}
        if "counts" in f:
            assert shown == f["counts"], f
        if "page" in f:
            order = sorted(shown.items(), key=lambda kv: -kv[1])
            assert len(order) == f["total_values"]
            assert [list(x) for x in order[f["offset"]: f["offset"] + f["limit"]]] == f["page"]
    # ParseFilterValue corner cases (std::from_chars: whole string, no '+', no whitespace)
    assert F.parse_filter_value("80.0")["double"] == 80.0 and F.parse_filter_value("80.0")["int64"] is None
    assert F.parse_filter_value("+1")["int64"] is None and F.parse_filter_value(" 1")["double"] is None
    assert F.parse_filter_value("-5")["uint64"] is None and F.parse_filter_value("-5")["int64"] == -5
    assert F.parse_filter_value("1e999")["double"] is None and F.parse_filter_value("18446744073709551616")["uint64"] is None
