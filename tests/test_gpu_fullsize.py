"""-m gpu, BASELINE.json configs[1] at FULL size (10M docs, bigram index, batch 1024 x 3-term AND + BM25 top-10): properties
that need no oracle — the scored path against the intersection-only path (different kernels), page prefixes across page
sizes (different top-k capacities and item plans), order and tie rule inside every page, idempotence of a re-run — plus
the oracle itself on a seeded sample of the batch."""
import numpy as np
import pytest

from oracle import oracle as O
from pkg import mg

pytestmark = pytest.mark.gpu
Query = mg.engine.Query


@pytest.fixture(scope="module")
def table():
    corpus = mg.Corpus.synthetic(10_000_000, seed=42)
    return corpus, mg.Index(corpus=corpus, ngram_size=2)


def _batch(idx, n=1024, seed=42):
    c = idx.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    cand = [g for g in range(c.n_grams) if b" " not in c.gram(g)]
    w = sizes[cand].astype(np.float64)
    rng = np.random.default_rng(seed)
    return [[c.gram(cand[i]).decode() for i in rng.choice(len(cand), size=3, replace=False, p=w / w.sum())] for _ in range(n)]


def test_full_size_batch_properties_and_sampled_oracle(table):
    corpus, idx = table
    terms = _batch(idx)
    top10 = idx.search_batch([Query(t, sort_score=True, limit=10) for t in terms])
    top100 = idx.search_batch([Query(t, sort_score=True, limit=100) for t in terms])
    counts = idx.search_batch([Query(t, limit=10) for t in terms])          # docid order: wave_count / wave_page kernels
    again = idx.search_batch([Query(t, sort_score=True, limit=10) for t in terms])
    n_total = 0
    for t, a, b, c, d in zip(terms, top10, top100, counts, again):
        assert a.total == b.total == c.total, t                             # the pruned scored path counts every match
        assert a.docs.tolist() == b.docs[:10].tolist() and np.array_equal(a.scores, b.scores[:10]), t
        assert a.docs.tolist() == d.docs.tolist() and np.array_equal(a.scores, d.scores), t
        s, dd = b.scores, b.docs.astype(np.int64)
        assert np.all(s[:-1] >= s[1:]), t                                   # DESC by score ...
        ties = s[:-1] == s[1:]
        assert np.all(dd[:-1][ties] > dd[1:][ties]), t                      # ... ties by larger docid first
        assert len(b.docs) == min(100, b.total)
        n_total += a.total
    assert n_total > 100_000_000                                            # (the batch really is the heavy one)
    # the oracle on a sample (each query costs it ~1 s at this size)
    c = idx.columns
    oidx = O.Index.from_csr(2, 0, True, c.key_bytes, c.key_off, c.offsets, c.docids)
    ostore = O.DocumentStore.from_arrays(corpus.text_bytes, corpus.text_off)
    order = np.argsort([r.total for r in top10])
    sample = [int(order[i]) for i in (0, 1, 50, 200, 400, 512, 700, 900, 1000, 1023)]  # sparse ... dense
    for i in sample:
        total, docs, scores = O.search_scored(oidx, ostore, terms[i], c.bm25_doc_count, c.avg_doc_length(), limit=100)
        assert top100[i].total == total and top100[i].docs.tolist() == docs.tolist(), terms[i]
        assert np.array_equal(top100[i].scores, scores), terms[i]
