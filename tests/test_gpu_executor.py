"""-m gpu: search_pipeline::BatchExecutor (C++17, through libmygram_shim.so) — fresh batches planned, compiled, run and
fetched in C++, two in flight — against the oracle."""
import numpy as np
import pytest

from gpu_util import Pair
from oracle import oracle as O
from pkg import mg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    return Pair(corpus=mg.Corpus.synthetic(120_000, seed=9))


def _batches(pair, n_batches, batch, seed):
    c = pair.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    cand = [g for g in range(c.n_grams) if b" " not in c.gram(g) and sizes[g] > 0]
    w = sizes[cand].astype(np.float64)
    rng = np.random.default_rng(seed)
    out = []
    for b in range(n_batches):
        qs = []
        for i in range(batch):
            k = 3 if (b + i) % 4 else int(rng.integers(1, 5))
            pick = rng.choice(len(cand), size=k, replace=False, p=w / w.sum())
            terms = [c.gram(cand[j]).decode() for j in pick]
            if i % 37 == 5:
                terms[0] = terms[0].upper()  # normalisation happens in the C++ planner
            if i % 53 == 7:
                terms.append("q~")           # unknown gram: resolved on the host (empty_term_detected)
            qs.append(terms)
        out.append(qs)
    return out


def test_pipelined_fresh_batches_match_oracle(pair):
    from mygram_db_amd import _shim_capi as S
    table = S.Table(pair.dev)
    ex = S.Executor(table, depth=2, planner_threads=3)
    batches = _batches(pair, 6, 96, seed=4)
    qbs = [S.QueryBatch(b) for b in batches]
    limit = 10
    tickets = [ex.submit(qbs[0], limit=limit)]
    results = []
    for i in range(1, len(qbs) + 1):
        if i < len(qbs):
            tickets.append(ex.submit(qbs[i], limit=limit))  # batch i is planned while batch i-1 runs
        totals, n_docs, docs, scores, timing = ex.wait(tickets[i - 1])
        results.append((totals.copy(), n_docs.copy(), docs.copy(), scores.copy()))
        assert timing[0] >= 0 and timing[1] >= 0
    n, avg = pair.N, pair.avgdl
    for terms_list, (totals, n_docs, docs, scores) in zip(batches, results):
        for qi, terms in enumerate(terms_list):
            total, d, s = O.search_scored(pair.oidx, pair.ostore, [t.lower() for t in terms], n, avg, limit=limit)
            assert totals[qi] == total, terms
            assert docs[qi, :n_docs[qi]].tolist() == d.tolist(), terms
            assert np.array_equal(scores[qi, :n_docs[qi]], s), terms


def test_executor_refuses_more_batches_than_slots(pair):
    from mygram_db_amd import _shim_capi as S
    ex = S.Executor(S.Table(pair.dev), depth=1, planner_threads=1)
    qb = S.QueryBatch([["th", "he"]])
    t = ex.submit(qb)
    with pytest.raises(S.ShimError):
        ex.submit(qb)
    ex.wait(t)
    ex.wait(ex.submit(qb))
