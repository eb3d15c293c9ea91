"""-m gpu: the batch SHAPES of BASELINE.json configs[2..4] at sizes the oracle checks in seconds — full-size batches
(4096 / 8192 / 1024 queries), a seeded sample of every batch compared with the oracle (docids, totals, funnel counters,
scores bit for bit). What these add over test_gpu_parity.py is scale: thousands of queries per launch, every query kind of
a config mixed in one batch, limit 100 pages, 1M-doc indexes spanning 60+ tiles."""
import numpy as np
import pytest

from gpu_util import Pair
from oracle import oracle as O
from pkg import mg

pytestmark = pytest.mark.gpu
Query = mg.engine.Query


def _cjk_texts(n_docs, seed):
    """configs[2] corpus: 3,000 ideographs (Zipf) + 80 kana, 16-48 code points per doc (NFKC-normal already)."""
    rng = np.random.default_rng(seed)
    alphabet = np.asarray([chr(0x4E00 + i) for i in range(3000)] + [chr(0x3042 + i) for i in range(80)])
    wi = 1.0 / np.arange(1, 3001)
    p = np.concatenate([wi / wi.sum() * 0.9, np.full(80, 0.1 / 80)])
    lens = rng.integers(16, 49, size=n_docs)
    flat = rng.choice(len(alphabet), size=int(lens.sum()), p=p)
    texts, at = [], 0
    for n in lens:
        texts.append("".join(alphabet[flat[at:at + n]]))
        at += n
    return texts


def test_config2_shape_cjk_trigram_mixed_batch_4096():
    """configs[2]: CJK trigram index, batch 4096 = 50% AND of 2-4 multi-gram terms, 20% (a OR b) AND c, 20% a AND NOT b,
    10% FUZZY 1; docid-DESC pages of 100 (the reference's default order without SORT, result_sorter.cpp:502-510)."""
    texts = _cjk_texts(120_000, seed=3)
    p = Pair(docs=list(enumerate(texts, start=1)), ngram=3, kanji=3)
    rng = np.random.default_rng(33)

    def term():
        d = texts[int(rng.integers(0, len(texts)))]
        n = int(rng.integers(3, 7))
        s = int(rng.integers(0, max(1, len(d) - n)))
        return d[s:s + n]

    qs = []
    for i in range(4096):
        r = i % 10
        if r < 5:
            qs.append(Query([term() for _ in range(int(rng.integers(2, 5)))], limit=100))
        elif r < 7:
            a, b, c = term(), term(), term()
            qs.append(Query(expr=("and", ("or", a, b), c), limit=100))
        elif r < 9:
            qs.append(Query([term()], [term()], limit=100))
        else:
            qs.append(Query([term()], fuzzy=1, limit=100))
    got = p.dev.search_batch(qs)
    assert len(got) == 4096
    sample = rng.choice(4096, size=400, replace=False)
    n_nonempty = 0
    for i in sample.tolist():
        q, g = qs[i], got[i]
        if q.fuzzy:
            r = O.execute_fuzzy(p.oidx, p.ostore, q.terms, q.fuzzy, ngram_size=3, kanji_ngram_size=3, cross_boundary=True)
            res = r["results"]
            page = res[::-1][:100]
            assert g.total == len(res) and g.docs.tolist() == page.tolist(), (i, q.terms)
        else:
            total, page, _, r = p.oracle_query(q)
            assert g.total == total and g.docs.tolist() == page.tolist(), (i, q.terms, q.not_terms, q.expr)
            if q.expr is None and not r["empty_term_detected"]:
                for k in ("total_candidates", "after_intersection", "after_not", "after_filters"):
                    assert getattr(g, k) == r[k], (i, k)
        n_nonempty += g.total > 0
    assert n_nonempty > 100  # (terms are cut out of documents: most queries match something)


@pytest.fixture(scope="module")
def pair1m():
    return Pair(corpus=mg.Corpus.synthetic(1_000_000, seed=42))


def test_config4_shape_5term_and_filter_batch_8192(pair1m):
    """configs[4] on one GPU's scale-down: 1M docs, batch 8192 x 5-term AND with terms Zipf-sampled BY RANK (lists from
    10^2 to 10^6 postings in one query) + FILTER category = x (EQ bitmap, 5 uniform values), docid-DESC limit 100."""
    p = pair1m
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = sorted((g for g in range(c.n_grams) if b" " not in c.gram(g)), key=lambda g: -sizes[g])
    rng = np.random.default_rng(5)
    cat = rng.integers(0, 5, size=c.n_docs)
    fids = [p.add_filter((np.nonzero(cat == v)[0] + 1).astype(np.uint32)) for v in range(5)]
    w = 1.0 / np.arange(1, len(grams) + 1)
    w /= w.sum()
    qs = []
    for i in range(8192):
        pick = rng.choice(len(grams), size=5, replace=False, p=w)
        qs.append(Query([c.gram(grams[j]).decode() for j in pick], filters=[(fids[int(rng.integers(0, 5))], i % 17 == 3)],
                        limit=100, descending=True))
    got = p.dev.search_batch(qs)
    for i in rng.choice(8192, size=160, replace=False).tolist():
        total, page, _, r = p.oracle_query(qs[i])
        g = got[i]
        assert g.total == total and g.docs.tolist() == page.tolist(), (i, qs[i].terms)
        for k in ("total_candidates", "after_intersection", "after_not", "after_filters"):
            assert getattr(g, k) == r[k], (i, k)


def test_config3_shape_3term_and_top100_batch_1024(pair1m):
    """configs[3] on one GPU's scale-down: 1M docs, batch 1024 x 3-term AND (bigrams sampled by df) + BM25 top-100
    (offset + limit = 100 entries per wave list, a merge of ~10 workgroup lists per query)."""
    p = pair1m
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    cand = [g for g in range(c.n_grams) if b" " not in c.gram(g)]
    w = sizes[cand].astype(np.float64)
    w /= w.sum()
    rng = np.random.default_rng(44)
    qs = []
    for _ in range(1024):
        pick = rng.choice(len(cand), size=3, replace=False, p=w)
        qs.append(Query([c.gram(cand[j]).decode() for j in pick], sort_score=True, limit=100))
    got = p.dev.search_batch(qs)
    n, avg = p.N, p.avgdl
    for i in rng.choice(1024, size=48, replace=False).tolist():
        total, docs, scores = O.search_scored(p.oidx, p.ostore, qs[i].terms, n, avg, limit=100)
        g = got[i]
        assert g.total == total, (i, qs[i].terms)
        assert g.docs.tolist() == docs.tolist(), (i, qs[i].terms)
        assert np.array_equal(g.scores, scores), (i, qs[i].terms)


def test_config2_raw_width_and_case_variants_are_normalised_like_the_reference():
    """configs[2] as a client sends it: half-width katakana, full-width Latin, upper case and compatibility forms in
    documents AND queries. Documents are normalised before Index::AddDocument (as the reference's loaders do with
    NormalizeText); query terms inside the planner (Index::NormalizeText, ICU NFKC + lower). Engine and the C++ executor
    against the oracle, whose normaliser is independent of ICU."""
    from mygram_db_amd import _shim_capi as S
    if not S.normalize_uses_icu():
        pytest.skip("built without ICU: queries must arrive NFKC-normal")
    rng = np.random.default_rng(91)
    half_kana = [chr(c) for c in range(0xFF66, 0xFF9E)] + ["ｶﾞ", "ｷﾞ", "ﾊﾟ", "ﾋﾞ", "ﾌﾞ"]
    full_latin = [chr(c) for c in range(0xFF21, 0xFF3B)] + [chr(c) for c in range(0xFF41, 0xFF5B)]
    kanji = [chr(0x4E00 + i) for i in range(400)]
    extra = ["ﬁ", "①", "㈱", "Å", "É", "Σ", "ß", "ǅ", "㌔", " "]
    pools = [half_kana, full_latin, kanji, [chr(c) for c in range(0x41, 0x5B)], extra]
    raw = []
    for _ in range(40_000):
        n = int(rng.integers(8, 33))
        raw.append("".join(str(rng.choice(pools[int(rng.choice(5, p=[.3, .2, .3, .15, .05]))])) for _ in range(n)))
    texts = [O.normalize_text(t) for t in raw]
    for i in rng.choice(len(raw), size=2000, replace=False).tolist():  # the product's normaliser says the same
        assert S.normalize_text(raw[i]) == texts[i], raw[i]
    p = Pair(docs=list(enumerate(texts, start=1)), ngram=2, kanji=1)

    def raw_term():
        d = raw[int(rng.integers(0, len(raw)))]
        n = int(rng.integers(2, 5))
        s = int(rng.integers(0, max(1, len(d) - n)))
        return d[s:s + n]

    qs = [Query([raw_term() for _ in range(int(rng.integers(1, 4)))], sort_score=True, limit=20) for _ in range(600)]
    # (a term that normalises to less than one n-gram takes the substring fallback: covered in test_gpu_parity.py)
    qs = [q for q in qs if all(" " not in O.normalize_text(t) and O.generate_query_ngrams(O.normalize_text(t), 2, 1, True)
                               for t in q.terms)]
    got = p.dev.search_batch(qs)
    hits = 0
    for q, g in zip(qs, got):
        total, page, scores, _ = p.oracle_query(q)
        assert g.total == total and g.docs.tolist() == page.tolist(), q.terms
        assert np.array_equal(g.scores, scores), q.terms
        hits += total > 0
    assert len(qs) > 200 and hits > 100
    # the same raw batch through search_pipeline::BatchExecutor (planned and normalised in C++)
    ex = S.Executor(S.Table(p.dev), depth=1, planner_threads=2)
    totals, n_docs, docs, scores, _ = ex.wait(ex.submit(S.QueryBatch([q.terms for q in qs]), limit=20))
    for qi, g in enumerate(got):
        assert totals[qi] == g.total and docs[qi, :n_docs[qi]].tolist() == g.docs.tolist(), qs[qi].terms
        assert np.array_equal(scores[qi, :n_docs[qi]], g.scores), qs[qi].terms


def test_pruned_fast_path_agrees_with_the_general_kernel_on_every_query(pair1m, monkeypatch):
    """Differential check at scale: 2048 SORT _score queries (1-5 terms sampled by df — the heaviest grams included —,
    pages of 10 / 37 / 100 with offsets, both orders) through bitmap_score_kernel (block-max pruning, re-masking, the pair
    queue) and again through the general workgroup kernel (no pruning, MGX_FORCE_BLOCK_KERNEL): totals, funnel counters,
    pages and scores of EVERY query must be identical; a seeded sample is also held against the oracle."""
    p = pair1m
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    cand = [g for g in range(c.n_grams) if b" " not in c.gram(g)]
    w = sizes[cand].astype(np.float64)
    w /= w.sum()
    rng = np.random.default_rng(2024)
    qs = []
    for i in range(2048):
        k = int(rng.integers(1, 6))
        pick = rng.choice(len(cand), size=k, replace=False, p=w)
        limit = int(rng.choice([10, 10, 37, 100]))
        qs.append(Query([c.gram(cand[j]).decode() for j in pick], sort_score=True, limit=limit,
                        offset=int(rng.choice([0, 0, 0, 7])), descending=bool(i % 5 != 4)))
    fast = p.dev.search_batch(qs)
    monkeypatch.setenv("MGX_FORCE_BLOCK_KERNEL", "1")
    general = p.dev.search_batch(qs)
    monkeypatch.delenv("MGX_FORCE_BLOCK_KERNEL")
    for i, (a, b) in enumerate(zip(fast, general)):
        assert a.total == b.total and a.docs.tolist() == b.docs.tolist(), (i, qs[i].terms, qs[i].limit, qs[i].offset)
        assert np.array_equal(a.scores, b.scores), (i, qs[i].terms)
        for k in ("total_candidates", "after_intersection", "after_not", "after_filters"):
            assert getattr(a, k) == getattr(b, k), (i, k)
    n, avg = p.N, p.avgdl
    for i in rng.choice(2048, size=24, replace=False).tolist():
        q = qs[i]
        total, page, scores, _ = p.oracle_query(q)
        assert fast[i].total == total and fast[i].docs.tolist() == page.tolist(), (i, q.terms)
        assert np.array_equal(fast[i].scores, scores), (i, q.terms)
