"""-m gpu: selective queries — a sparse posting list among the operands — run candidate-driven (mgx::cand_kernel: the
smallest list is enumerated, every other operand probed per candidate; the device form of the reference's small-candidate
strategy, search_pipeline.cpp:828-829 -> Index::FilterByNgrams -> PostingList::RetainPresent posting_list.cpp:432-474, and
of SearchAnd's smallest-first early exit, index.cpp:228-240,338-351). Checked against the oracle (docids, totals, funnel
counters, scores bit for bit) and against the tile kernels on the same batch (MGX_CAND=0)."""
import os

import numpy as np
import pytest

from gpu_util import Pair
from pkg import mg

pytestmark = pytest.mark.gpu
Query = mg.engine.Query


@pytest.fixture(scope="module")
def pair():
    return Pair(corpus=mg.Corpus.synthetic(300_000, seed=7))


def _grams_by_size(p):
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = sorted((g for g in range(c.n_grams) if b" " not in c.gram(g)), key=lambda g: -sizes[g])
    return c, sizes, grams


def _zipf_queries(p, n, rng, n_terms=(2, 6), **kw):
    c, sizes, grams = _grams_by_size(p)
    w = 1.0 / np.arange(1, len(grams) + 1)
    w /= w.sum()
    qs = []
    for _ in range(n):
        k = int(rng.integers(*n_terms))
        pick = rng.choice(len(grams), size=k, replace=False, p=w)
        qs.append([c.gram(grams[j]).decode() for j in pick])
    return qs


def test_selective_pages_match_the_oracle(pair):
    """docid-ordered pages (both directions, limits 1..5000) of Zipf-by-rank conjunctions, with NOT terms and EQ/NE
    filters: totals, pages and all four funnel counters."""
    p = pair
    rng = np.random.default_rng(11)
    c, sizes, grams = _grams_by_size(p)
    cat = rng.integers(0, 4, size=c.n_docs)
    fids = [p.add_filter((np.nonzero(cat == v)[0] + 1).astype(np.uint32)) for v in range(4)]
    sparse = [g for g in grams if 0 < sizes[g] < c.n_docs // 300]
    assert len(sparse) > 50
    qs = []
    for i, terms in enumerate(_zipf_queries(p, 600, rng)):
        kw = {"limit": int(rng.choice([1, 10, 100, 5000])), "descending": bool(i % 2)}
        if i % 3 == 0:
            kw["filters"] = [(fids[int(rng.integers(0, 4))], i % 6 == 0)]
        not_terms = []
        if i % 4 == 1:  # a NOT term: dense or sparse
            src = grams[: 40] if i % 8 == 1 else sparse
            not_terms = [c.gram(src[int(rng.integers(0, len(src)))]).decode()]
        if i % 5 == 0:  # make sure a sparse gram is among the terms
            terms = terms + [c.gram(sparse[int(rng.integers(0, len(sparse)))]).decode()]
        qs.append(Query(terms, not_terms, **kw))
    got = p.check(qs)
    assert sum(g.total > 0 for g in got) > 30


def test_selective_scored_queries_match_the_oracle(pair):
    """SORT _score (DESC and ASC, top-1 .. top-200, offsets) with a sparse gram among the scored terms: tf from the
    driver's own column, from nibble rows and from other sparse lists; ranks and fp64 scores bit for bit."""
    p = pair
    rng = np.random.default_rng(12)
    c, sizes, grams = _grams_by_size(p)
    sparse = [g for g in grams if 50 < sizes[g] < c.n_docs // 300]
    dense = grams[:60]
    qs = []
    for i in range(400):
        terms = [c.gram(sparse[int(rng.integers(0, len(sparse)))]).decode()]
        for _ in range(int(rng.integers(1, 4))):
            src = sparse if rng.random() < 0.3 else dense
            terms.append(c.gram(src[int(rng.integers(0, len(src)))]).decode())
        terms = list(dict.fromkeys(terms))
        rng.shuffle(terms)
        kw = {"limit": int(rng.choice([1, 10, 100, 200])), "descending": i % 3 != 0, "sort_score": True}
        if i % 7 == 0:
            kw["offset"] = int(rng.integers(0, 20))
        not_terms = [c.gram(dense[int(rng.integers(0, len(dense)))]).decode()] if i % 5 == 0 else []
        qs.append(Query(terms, not_terms, **kw))
    got = p.check(qs)
    assert sum(g.total > 0 for g in got) > 50


def test_candidate_driven_equals_tile_scan(pair):
    """The same 2048-query batch (pages and scored queries mixed) with and without the candidate-driven path."""
    p = pair
    rng = np.random.default_rng(13)
    qs = []
    for i, terms in enumerate(_zipf_queries(p, 2048, rng, n_terms=(1, 6))):
        if i % 2:
            qs.append(Query(terms, sort_score=True, limit=10))
        else:
            qs.append(Query(terms, limit=100, descending=bool(i % 4)))
    a = p.dev.search_batch(qs)
    os.environ["MGX_CAND"] = "0"
    try:
        b = p.dev.search_batch(qs)
    finally:
        del os.environ["MGX_CAND"]
    for x, y, q in zip(a, b, qs):
        assert x.total == y.total, q.terms
        assert x.docs.tolist() == y.docs.tolist(), q.terms
        assert np.array_equal(x.scores, y.scores), q.terms
        for k in ("total_candidates", "after_intersection", "after_not", "after_filters"):
            assert getattr(x, k) == getattr(y, k), (q.terms, k)


@pytest.mark.parametrize("dense_threshold", [2.0, 0.18])
def test_long_posting_arrays_merge_with_carried_ranks(dense_threshold):
    """An index WITHOUT bitmaps (dense_threshold 2: every list a sorted u32 array) and one at the reference's
    roaring_threshold 0.18 (lists below 18 % density stay arrays, posting_list.cpp:800-834): SORT _score over LONG lists
    runs on mgx::merge_score_kernel — the smallest list drives, the others are staged per tile, a match's tf under a
    list-form term is read at its RANK in the tile segment. Oracle parity (ranks and fp64 scores bit for bit, funnel
    counters) and equality with the round-2 list path (MGX_MERGE=0)."""
    p = Pair(corpus=mg.Corpus.synthetic(200_000, seed=21), dense_threshold=dense_threshold)
    rng = np.random.default_rng(22)
    c, sizes, grams = _grams_by_size(p)
    cat = rng.integers(0, 3, size=c.n_docs)
    fids = [p.add_filter((np.nonzero(cat == v)[0] + 1).astype(np.uint32)) for v in range(3)]
    dense = grams[:120]
    w = sizes[dense].astype(np.float64)
    w /= w.sum()
    qs = []
    for i in range(300):
        k = int(rng.integers(1, 5))
        pick = rng.choice(len(dense), size=k, replace=False, p=w)
        terms = [c.gram(dense[j]).decode() for j in pick]
        kw = {"sort_score": True, "limit": int(rng.choice([1, 10, 100])), "descending": i % 4 != 0}
        if i % 6 == 0:
            kw["offset"] = int(rng.integers(0, 30))
        if i % 5 == 0:
            kw["filters"] = [(fids[int(rng.integers(0, 3))], i % 10 == 0)]
        not_terms = [c.gram(grams[int(rng.integers(0, 200))]).decode()] if i % 7 == 0 else []
        qs.append(Query(terms, not_terms, **kw))
    got = p.check(qs)
    assert sum(g.total > 0 for g in got) > 100
    os.environ["MGX_MERGE"] = "0"
    try:
        old = p.dev.search_batch(qs)
    finally:
        del os.environ["MGX_MERGE"]
    for x, y, q in zip(got, old, qs):
        assert x.total == y.total and x.docs.tolist() == y.docs.tolist() and np.array_equal(x.scores, y.scores), q.terms
