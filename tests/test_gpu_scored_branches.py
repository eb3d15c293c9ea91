"""-m gpu: SORT _score over the result sets of the boolean-expression and FUZZY branches. The reference scores whatever
branch produced the results (src/server/handlers/search_handler.cpp:405-470 after search_pipeline.cpp:1842-1954): the
scored terms of an expression are its TERM leaves that are not under a NOT, in tree order, repeats kept
(CollectAstScoringTerms :232-254; vector tests/server/search_pipeline_test.cpp:1350-1392: NOT terms are not scored); a
FUZZY query scores its exact terms. Expected values come from the oracle's own pieces: the branch's result set, df per
term as PopulateTermDocumentFrequency counts it, BM25Scorer::ScoreDocuments over those terms, ResultSorter::SortByScore."""
import numpy as np
import pytest

from gpu_util import Pair
from oracle import oracle as O
from pkg import mg

pytestmark = pytest.mark.gpu
Query = mg.engine.Query


def _positive_leaves(e, under_not=False, out=None):
    out = [] if out is None else out
    if isinstance(e, str):
        if not under_not:
            out.append(e)
        return out
    for c in e[1:]:
        _positive_leaves(c, under_not or e[0] == "not", out)
    return out


def _df(p, term):
    return O.execute(p.oidx, p.ostore, [term], compute_df=True, ngram_size=p.dev.ngram_size,
                     kanji_ngram_size=p.dev.kanji_ngram_size, cross_boundary=p.dev.cross_boundary)["term_df"][0]


def _expected_scored(p, res, terms, q):
    terms = [O.normalize_text(t) for t in terms]
    dfs = [_df(p, t) for t in terms]
    sc = O.score_documents(p.ostore, res, terms, dfs, p.N, p.avgdl, q.k1, q.b)
    page = O.sort_by_score(res, sc, q.descending, q.limit, q.offset)
    lookup = dict(zip(res.tolist(), sc.tolist()))
    return page, np.asarray([lookup[d] for d in page.tolist()])


def _check_expr(p, qs):
    got = p.dev.search_batch(qs)
    for q, g in zip(qs, got):
        total, _, _, _ = p.oracle_query(Query(expr=q.expr, not_terms=q.not_terms, filters=q.filters, limit=0))
        res = np.asarray(sorted(p.oracle_query(Query(expr=q.expr, not_terms=q.not_terms, filters=q.filters, limit=0,
                                                     descending=False))[1].tolist()), dtype=np.uint32)
        page, scores = _expected_scored(p, res, _positive_leaves(q.expr), q)
        assert g.total == total == len(res), q.expr
        assert g.docs.tolist() == page.tolist(), (q.expr, g.docs, page)
        assert np.array_equal(g.scores, scores), (q.expr, g.scores, scores)
    return got


def test_expression_results_are_scored_by_their_positive_terms_small():
    """The reference's pipeline fixture (search_pipeline_test.cpp:916-951) + ranking cases: OR branches (a doc may hold
    only one of the scored terms), a NOT-excluded term that must not be scored, a repeated leaf, an unknown leaf."""
    docs = [(1, "machine learning basics"), (2, "deep learning techniques"), (3, "old article about cats"),
            (4, "learning learning learning"), (5, "cats and machine cats"), (6, "basics of basics")]
    p = Pair(docs=docs, ngram=2, kanji=0)
    qs = [Query(expr=("or", "basics", "cats"), sort_score=True, limit=10),
          Query(expr=("and", ("or", "basics", "cats"), "machine"), sort_score=True, limit=10),
          Query(expr=("and", "learning", ("not", "deep")), sort_score=True, limit=10),
          Query(expr=("or", "learning", "learning"), sort_score=True, limit=10, descending=False),
          Query(expr=("or", "cats", "zebra"), sort_score=True, limit=3, offset=1),
          Query(expr=("or", "ca", ("and", "le", "ar")), sort_score=True, limit=10)]
    got = _check_expr(p, qs)
    assert got[0].total == 4 and got[2].total == 2


def test_expression_and_fuzzy_scoring_at_scale():
    """Random trees over single-gram and whole-word terms on a 150k-doc corpus (several tiles, text-level df/tf on the
    device), and FUZZY 1 queries scored by their exact terms."""
    p = Pair(corpus=mg.Corpus.synthetic(150_000, seed=17))
    rng = np.random.default_rng(18)
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = [c.gram(g).decode() for g in np.argsort(-sizes)[:150] if b" " not in c.gram(g)]
    words = sorted({w for i in range(0, 150_000, 97) for w in p.corpus.text(i).decode().split(" ") if 3 <= len(w) <= 8})

    def leaf():
        return grams[int(rng.integers(0, len(grams)))] if rng.random() < 0.6 else words[int(rng.integers(0, len(words)))]

    qs = []
    for i in range(60):
        a, b, cc = leaf(), leaf(), leaf()
        shape = i % 4
        e = (("and", ("or", a, b), cc), ("or", a, ("and", b, cc)), ("and", a, ("not", b)), ("or", a, b, cc))[shape]
        qs.append(Query(expr=e, sort_score=True, limit=int(rng.choice([1, 10, 50])), descending=i % 5 != 0))
    got = _check_expr(p, qs)
    assert sum(g.total > 0 for g in got) > 40
    # FUZZY 1 SORT _score: ExecuteWithFuzzy's result set, the exact terms scored
    fq = [Query([words[int(rng.integers(0, len(words)))] for _ in range(int(rng.integers(1, 3)))], fuzzy=1, sort_score=True,
                limit=10) for _ in range(40)]
    fg = p.dev.search_batch(fq)
    nz = 0
    for q, g in zip(fq, fg):
        r = O.execute_fuzzy(p.oidx, p.ostore, q.terms, 1, ngram_size=2, kanji_ngram_size=0, cross_boundary=True)
        res = r["results"]
        page, scores = _expected_scored(p, res, q.terms, q)
        assert g.total == len(res), q.terms
        assert g.docs.tolist() == page.tolist(), (q.terms, g.docs, page)
        assert np.array_equal(g.scores, scores), q.terms
        nz += len(res) > 0
    assert nz > 20
