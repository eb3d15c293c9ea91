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


def test_micro_batcher_many_client_threads(pair):
    """search_pipeline::MicroBatcher: 48 client threads, one blocking Search() each at a time; their queries share device
    batches (closed by size or by delay) and every answer equals the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    from mygram_db_amd import _shim_capi as S
    mb = S.Batcher(S.Table(pair.dev), max_batch=64, max_delay_us=300, depth=2, planner_threads=3)
    term_lists = [t for b in _batches(pair, 4, 96, seed=21) for t in b]

    def one(terms):
        return mb.search(terms, limit=10)

    with ThreadPoolExecutor(max_workers=48) as pool:
        got = list(pool.map(one, term_lists))
    n, avg = pair.N, pair.avgdl
    for terms, (total, docs, scores) in zip(term_lists, got):
        t, d, s = O.search_scored(pair.oidx, pair.ostore, [x.lower() for x in terms], n, avg, limit=10)
        assert total == t and docs.tolist() == d.tolist(), terms
        assert np.array_equal(scores, s), terms
    st = mb.stats()
    assert st["queries"] == len(term_lists)
    assert st["batches"] < len(term_lists) / 4  # (48 clients in flight: batches hold many queries, not one)
    # a lone query is answered after max_delay, not held for a full batch
    total, docs, scores = mb.search(term_lists[0], limit=10)
    assert total == got[0][0] and docs.tolist() == got[0][1].tolist()
    assert mb.stats()["closed_by_delay"] >= 1


def test_single_operators_from_many_threads_at_once():
    """Index::SearchAnd / SearchOr / SearchByThreshold / FilterByNgrams / ScoreDocuments / SortByScore are const and
    re-entrant in the reference — every worker thread of the server calls them concurrently under the table's shared lock
    (command_handler.cpp:66). Here each call leases its own arenas, pinned block and stream from the index: 12 threads x
    40 mixed calls against the oracle."""
    import threading
    import numpy as np
    from gpu_util import Pair
    from oracle import oracle as O
    p = Pair(corpus=mg.Corpus.synthetic(60_000, seed=23))
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = [c.gram(g).decode() for g in np.argsort(-sizes)[:60] if b" " not in c.gram(g)]
    errors = []

    def work(seed):
        rng = np.random.default_rng(seed)
        try:
            for k in range(40):
                terms = [grams[int(i)] for i in rng.choice(len(grams), size=int(rng.integers(1, 4)), replace=False)]
                kind = k % 5
                if kind == 0:
                    lim, rev = int(rng.choice([0, 10, 500])), bool(k % 2)
                    assert p.dev.search_and(terms, lim, rev).tolist() == p.oidx.search_and(terms, lim, rev).tolist()
                elif kind == 1:
                    assert p.dev.search_or(terms).tolist() == p.oidx.search_or(terms).tolist()
                elif kind == 2:
                    th = int(rng.integers(1, len(terms) + 1))
                    assert p.dev.search_by_threshold(terms, th).tolist() == p.oidx.search_by_threshold(terms, th).tolist()
                elif kind == 3:
                    cand = np.sort(rng.choice(60_000, size=300, replace=False)).astype(np.uint32) + 1
                    assert p.dev.filter_by_ngrams(cand, terms).tolist() == p.oidx.filter_by_ngrams(cand, terms).tolist()
                else:
                    res = p.oidx.search_and(terms)
                    if len(res) == 0:
                        continue
                    dfs = [p.oidx.posting_size(t) for t in terms]
                    want = O.score_documents(p.ostore, res, terms, dfs, p.N, p.avgdl)
                    got = p.dev.score_documents(res, terms, dfs, p.N, p.avgdl)
                    assert np.array_equal(got, want)
                    assert p.dev.sort_by_score(res, got, True, 20, 3).tolist() == O.sort_by_score(res, want, True, 20, 3).tolist()
        except Exception as e:  # noqa: BLE001 (reported by the main thread)
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(100 + i,)) for i in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
