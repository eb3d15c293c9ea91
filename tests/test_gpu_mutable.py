"""-m gpu: a table that changes after its index was built — Index::AddDocument / UpdateDocument /
RemoveDocument as the binlog applier calls them (src/mysql/binlog_event_processor.cpp:96,138,184,234,283; index.cpp:39-233)
through the C++ shim: a live-document row on the main index + a delta index, merged on the device. The bar: after any
sequence of changes, pages, totals and BM25 scores equal those of an index BUILT from the table's current documents
(what the reference's index holds after the same calls), bit for bit — checked against the oracle over such an index."""
import numpy as np
import pytest

from gpu_util import Pair
from oracle import filters as F
from pkg import mg

pytestmark = pytest.mark.gpu
Query = mg.engine.Query


def _shim():
    from mygram_db_amd import _shim_capi as S
    return S


class Mirror:
    """The table's current documents, changed alongside the shim table."""

    def __init__(self, corpus):
        self.docs = {i + 1: corpus.text(i) for i in range(corpus.n_docs)}

    def rebuilt(self):
        """Device + oracle pair over an index built from the current documents (ids kept)."""
        return Pair(docs=sorted(self.docs.items()), ngram=2, kanji=0)


def t_delta_docs(m, t):
    """Some documents to remove at the end of the test: the newest ids (they sit in the delta, or did before a compaction)."""
    return sorted(m.docs)[-120:]


def _queries(rng, grams, n):
    qs = []
    for i in range(n):
        k = int(rng.integers(1, 4))
        terms = [grams[int(rng.integers(0, len(grams)))] for _ in range(k)]
        nots = [grams[int(rng.integers(0, len(grams)))]] if i % 5 == 0 else []
        qs.append((terms, nots))
    return qs


def _check(t, want_pair, qs, executor=None):
    S = _shim()
    for terms, nots in qs:
        # SORT _score, both orders; docid pages, both orders
        for desc in (True, False):
            total, docs, scores = t.search(terms, not_terms=nots, sort_by_score=True, descending=desc, limit=10)
            wt, wp, ws, _ = want_pair.oracle_query(Query(terms, not_terms=nots, sort_score=True, descending=desc, limit=10))
            assert total == wt, (terms, nots, desc, total, wt)
            assert docs.tolist() == wp.tolist(), (terms, nots, desc)
            assert np.array_equal(scores, ws), (terms, nots, desc, scores, ws)
            total, docs, _ = t.search(terms, not_terms=nots, descending=desc, limit=7)
            wt, wp, _, _ = want_pair.oracle_query(Query(terms, not_terms=nots, descending=desc, limit=7))
            assert total == wt and docs.tolist() == wp.tolist(), (terms, nots, desc)
    if executor is not None:
        term_lists = [terms for terms, nots in qs if not nots]
        qb = S.QueryBatch(term_lists)
        t1 = executor.submit(qb, limit=10)
        t2 = executor.submit(qb, limit=5, sort_by_score=False, descending=True)
        totals, n_docs, docs, scores, _ = executor.wait(t1)
        for qi, terms in enumerate(term_lists):
            wt, wp, ws, _ = want_pair.oracle_query(Query(terms, sort_score=True, limit=10))
            assert totals[qi] == wt and docs[qi, :n_docs[qi]].tolist() == wp.tolist(), terms
            assert np.array_equal(scores[qi, :n_docs[qi]], ws), terms
        totals, n_docs, docs, scores, _ = executor.wait(t2)
        for qi, terms in enumerate(term_lists):
            wt, wp, _, _ = want_pair.oracle_query(Query(terms, limit=5))
            assert totals[qi] == wt and docs[qi, :n_docs[qi]].tolist() == wp.tolist(), terms


def test_changes_after_the_build_equal_an_index_built_from_the_current_documents():
    S = _shim()
    n = 40_000
    corpus = mg.Corpus.synthetic(n, seed=41)
    p = Pair(corpus=corpus)
    p.dev.ensure_text()  # (text-level BM25 terms scan the texts on the device)
    t = S.Table(p.dev)
    m = Mirror(corpus)
    rng = np.random.default_rng(42)
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = [c.gram(g).decode() for g in np.argsort(-sizes)[:60] if b" " not in c.gram(g)]
    # text-level terms (longer than one n-gram: df and tf are counted in the texts, on both indexes) and a word only
    # changed documents hold (a gram the main index's dictionary does not know)
    words = [w.decode() for w in corpus.text(0).split()[:3] if len(w) >= 3]
    qs = _queries(rng, grams, 24) + [([w], []) for w in words] + [(["zzqx"], []), ([grams[0], "zzqx"], [])]
    ex = S.Executor(t, depth=2, planner_threads=2)
    _check(t, p, qs[:6] + qs[24:26], ex)  # the static table first (the executor's slots are warm when the table starts to change)

    def change(n_remove, n_update, n_add, novel):
        live = sorted(m.docs)
        for d in rng.choice(live, n_remove, replace=False).tolist():
            t.remove_document(d, m.docs[d])
            del m.docs[d]
        live = sorted(m.docs)
        for k, d in enumerate(rng.choice(live, n_update, replace=False).tolist()):
            new = corpus.text(int(rng.integers(0, n)))
            if k < novel:
                new = new + b" zzqx"
            t.update_document(d, m.docs[d], new)
            m.docs[d] = new
        top = max(max(m.docs), n)
        for k in range(n_add):
            new = corpus.text(int(rng.integers(0, n))) + (b" zzqx tail" if k < novel else b"")
            t.add_document(top + 1 + k, new)
            m.docs[top + 1 + k] = new

    change(300, 400, 250, novel=5)
    st = t.mutation_stats()
    assert st["delta_documents"] == 650 and st["removed_from_main"] == 700, st
    want = m.rebuilt()
    _check(t, want, qs, ex)
    assert t.mutation_stats()["epoch"] == 1
    # second round: documents of the delta change and leave too; a removed id comes back
    gone = [d for d in range(1, n + 1) if d not in m.docs][:3]
    for d in gone:
        t.add_document(d, corpus.text(d - 1))
        m.docs[d] = corpus.text(d - 1)
    delta_docs = [d for d in m.docs if d > n][:40]
    for d in delta_docs[:20]:
        t.remove_document(d, m.docs[d])
        del m.docs[d]
    for d in delta_docs[20:]:
        new = corpus.text(int(rng.integers(0, n)))
        t.update_document(d, m.docs[d], new)
        m.docs[d] = new
    change(100, 100, 50, novel=2)
    want = m.rebuilt()
    _check(t, want, qs, ex)
    # compaction: the main index is rebuilt from the current documents (texts back from the device), the delta and the live
    # row go; the executor's batch objects move to the new device index; then the table changes again
    t.compact()
    st = t.mutation_stats()
    assert st["delta_documents"] == 0 and st["main_documents"] == len([d for d in m.docs.values() if d]), st
    _check(t, want, qs, ex)
    change(60, 80, 40, novel=2)
    want = m.rebuilt()
    _check(t, want, qs, ex)
    # removals only: the delta goes away when its last document does
    for d in t_delta_docs(m, t):
        t.remove_document(d, m.docs[d])
        del m.docs[d]
    want = m.rebuilt()
    _check(t, want, qs[:10], ex)
    # a live id cannot be added again
    with pytest.raises(S.ShimError):
        t.add_document(next(iter(m.docs)), b"again")


def test_filter_values_follow_a_changed_document_and_facets_count_both_indexes():
    S = _shim()
    n = 20_000
    corpus = mg.Corpus.synthetic(n, seed=51)
    p = Pair(corpus=corpus)
    p.dev.ensure_text()  # (compaction reads the main index's texts back from the device)
    t = S.Table(p.dev)
    rng = np.random.default_rng(52)
    status = rng.integers(0, 6, n)
    cat = ["cat%d" % k for k in rng.integers(0, 5, n)]
    nul = rng.random(n) < 0.05
    t.add_filter_column("status", "int32", status, nul)
    t.add_filter_column("category", "string", cat)
    m = Mirror(corpus)
    vals = {d: {"status": None if nul[d - 1] else ("int32", int(status[d - 1])), "category": ("string", cat[d - 1])}
            for d in m.docs}
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = [c.gram(g).decode() for g in np.argsort(-sizes)[:30] if b" " not in c.gram(g)]
    live = sorted(m.docs)
    upd = rng.choice(live, 300, replace=False).tolist()
    for k, d in enumerate(upd):
        new = corpus.text(int(rng.integers(0, n)))
        if k % 2 == 0:  # the document keeps its filter values (read back from the device columns)
            t.update_document(d, m.docs[d], new)
        else:
            f = {"status": ("int32", 7), "category": ("string", "fresh")}
            t.update_document(d, m.docs[d], new, filters=f)
            vals[d] = dict(f)
        m.docs[d] = new
    for d in rng.choice([x for x in live if x not in upd], 200, replace=False).tolist():
        t.remove_document(d, m.docs[d])
        del m.docs[d]
        del vals[d]
    for k in range(100):
        d = n + 1 + k
        new = corpus.text(int(rng.integers(0, n)))
        f = {"status": ("int32", int(k % 3)), "category": ("string", "cat%d" % (k % 7))}
        t.add_document(d, new, filters=f)
        m.docs[d] = new
        vals[d] = f
    # filter values alone change (DocumentStore::UpdateDocument(doc_id, filters)): a document of the main index, and one of
    # the delta
    still_main = [x for x in live if x in m.docs and x not in upd][:40]
    for d in still_main + [n + 1, n + 2]:
        f = {"status": ("int32", 9), "category": ("string", "moved")}
        t.update_filters(d, f)
        vals[d] = f
    with pytest.raises(S.ShimError):
        t.update_filters(10 * n, {"status": ("int32", 1)})
    conds_extra = [[("status", "=", "9")], [("category", "=", "moved")]]
    want = m.rebuilt()
    columns = {name: (lambda doc, name=name: vals.get(doc, {}).get(name)) for name in ("status", "category")}
    conds_list = [[("status", "=", "7")], [("status", "!=", "2")], [("category", "=", "fresh")], [("status", ">=", "3")],
                  [("category", "=", "cat6")], [("status", "<", "2"), ("category", "!=", "cat1")]]
    conds_list = conds_list + conds_extra
    for i, conds in enumerate(conds_list * 3):
        terms = [grams[int(rng.integers(0, len(grams)))]]
        base = want.oracle_query(Query(terms, limit=0, descending=False))[1].tolist()
        exp = F.apply_filters_with_bitmap(base, conds, columns)
        total, docs, _ = t.search(terms, conditions=conds, descending=False, limit=50)
        assert total == len(exp) and docs.tolist() == exp[:50], (terms, conds, total, len(exp))
        if i % 3 == 0:
            matched, n_values, page = t.facet("category", terms=terms, conditions=conds, limit=100)
            counts = {}
            for d in exp:
                v = vals[d]["category"]
                counts[v[1].encode()] = counts.get(v[1].encode(), 0) + 1
            assert matched == len(exp) and dict(page) == counts, (terms, conds)
        if i == len(conds_list):  # the rest of the loop runs on the compacted table: filter values moved with the documents
            t.compact()
            assert t.mutation_stats()["delta_documents"] == 0


def test_the_general_kernels_score_a_changed_table_too():
    """The same scenario with every query forced onto the general workgroup kernel and without the fast scoring path: after
    dead documents' bits left the bitmap form of their grams, tf must come by doc slot, not by popcount rank."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MGX_FORCE_BLOCK_KERNEL="1")
    here = os.path.dirname(os.path.abspath(__file__))
    for extra in ({}, {"MGX_FORCE_BLOCK_KERNEL": "0", "MGX_FAST_PATH": "0"}):
        e = dict(env, **extra)
        if e["MGX_FORCE_BLOCK_KERNEL"] == "0":
            del e["MGX_FORCE_BLOCK_KERNEL"]
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_gpu_mutable.py"),
                            "-k", "changes_after_the_build"], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_a_table_changes_while_a_micro_batcher_serves_it():
    """search_pipeline::MicroBatcher over a table that a writer thread keeps changing: the mutation calls only record (any
    thread), the batcher's submitting thread applies them between batches; searches never fail, and once the writer is done
    the answers are those of an index built from the final documents."""
    import threading
    S = _shim()
    n = 30_000
    corpus = mg.Corpus.synthetic(n, seed=61)
    p = Pair(corpus=corpus)
    p.dev.ensure_text()
    t = S.Table(p.dev)
    m = Mirror(corpus)
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = [c.gram(g).decode() for g in np.argsort(-sizes)[:40] if b" " not in c.gram(g)]
    b = S.Batcher(t, max_batch=64, max_delay_us=200, depth=2, planner_threads=2)
    errors, done = [], threading.Event()

    def searcher(seed):
        r = np.random.default_rng(seed)
        try:
            while not done.is_set():
                terms = [grams[int(r.integers(0, len(grams)))] for _ in range(int(r.integers(1, 4)))]
                total, docs, scores = b.search(terms, limit=10, sort_by_score=bool(r.integers(0, 2)))
                assert len(docs) <= 10 and total >= len(docs)
        except BaseException as e:  # noqa: BLE001 (reported by the main thread)
            errors.append(e)

    threads = [threading.Thread(target=searcher, args=(100 + k,)) for k in range(6)]
    for th in threads:
        th.start()
    rng = np.random.default_rng(62)
    try:
        for step in range(300):
            live = sorted(m.docs)
            d = int(live[int(rng.integers(0, len(live)))])
            kind = step % 3
            if kind == 0:
                t.remove_document(d, m.docs[d])
                del m.docs[d]
            elif kind == 1:
                new = corpus.text(int(rng.integers(0, n)))
                t.update_document(d, m.docs[d], new)
                m.docs[d] = new
            else:
                new_id = max(max(m.docs), n) + 1
                new = corpus.text(int(rng.integers(0, n)))
                t.add_document(new_id, new)
                m.docs[new_id] = new
            if step == 150:
                b.search([grams[0]], limit=5)  # (a search from the writer's thread too)
    finally:
        done.set()
        for th in threads:
            th.join()
    assert not errors, errors[:1]
    want = m.rebuilt()
    for terms in ([grams[0]], [grams[1], grams[2]], [grams[3], grams[4], grams[5]]):
        for by_score in (True, False):
            total, docs, scores = b.search(terms, limit=10, sort_by_score=by_score)
            wt, wp, ws, _ = want.oracle_query(Query(terms, sort_score=by_score, limit=10))
            assert total == wt and docs.tolist() == wp.tolist(), (terms, by_score)
            if by_score:
                assert np.array_equal(scores, ws), terms
    st = b.stats()
    assert st["queries"] >= 20 and st["batches"] >= 5, st  # (every batch formed while the writer ran applied changes first)


def test_changes_applied_in_the_background_under_a_staleness_bound():
    """Index::SetMutationStaleness: the delta of the recorded changes is built by a thread while queries run on the old state
    and installed by a later entry point — queries see the old table or the new one, never a mixture, and the new one within
    the bound (plus the build)."""
    import threading
    import time
    S = _shim()
    n = 30_000
    corpus = mg.Corpus.synthetic(n, seed=71)
    p = Pair(corpus=corpus)
    p.dev.ensure_text()
    t = S.Table(p.dev)
    m = Mirror(corpus)
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = [c.gram(g).decode() for g in np.argsort(-sizes)[:30] if b" " not in c.gram(g)]
    rng = np.random.default_rng(72)
    probe = [grams[0]]
    # the first application is synchronous (it creates the live row); from then on changes wait for the bound
    t.remove_document(1, m.docs[1])
    del m.docs[1]
    before = m.rebuilt()
    assert t.search(probe, limit=5)[0] == before.oracle_query(Query(probe, limit=5))[0]
    t.set_mutation_staleness(0.05)
    for k in range(400):
        d = int(rng.integers(2, n))
        if d not in m.docs:
            continue
        if k % 2:
            t.remove_document(d, m.docs[d])
            del m.docs[d]
        else:
            new = corpus.text(int(rng.integers(0, n)))
            t.update_document(d, m.docs[d], new)
            m.docs[d] = new
    after = m.rebuilt()
    want_before = before.oracle_query(Query(probe, sort_score=True, limit=10))
    want_after = after.oracle_query(Query(probe, sort_score=True, limit=10))
    assert want_before[0] != want_after[0]
    deadline = time.time() + 20.0
    seen_after = False
    while time.time() < deadline:
        total, docs, scores = t.search(probe, sort_by_score=True, limit=10)
        if total == want_after[0]:
            assert docs.tolist() == want_after[1].tolist() and np.array_equal(scores, want_after[2])
            seen_after = True
            break
        # not yet installed: exactly the old table
        assert total == want_before[0] and docs.tolist() == want_before[1].tolist() and np.array_equal(scores, want_before[2])
        time.sleep(0.01)
    assert seen_after, "the background application never became visible"
    # several queries against the final state, and a batcher with searching threads while more changes arrive
    for terms in ([grams[1], grams[2]], [grams[3]]):
        total, docs, scores = t.search(terms, sort_by_score=True, limit=10)
        wt, wp, ws, _ = after.oracle_query(Query(terms, sort_score=True, limit=10))
        assert total == wt and docs.tolist() == wp.tolist() and np.array_equal(scores, ws), terms
    b = S.Batcher(t, max_batch=64, max_delay_us=200, depth=2, planner_threads=2)
    errors, done = [], threading.Event()

    def searcher(seed):
        r = np.random.default_rng(seed)
        try:
            while not done.is_set():
                terms = [grams[int(r.integers(0, len(grams)))] for _ in range(int(r.integers(1, 3)))]
                total, docs, _ = b.search(terms, limit=10)
                assert total >= len(docs)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=searcher, args=(200 + k,)) for k in range(4)]
    for th in threads:
        th.start()
    try:
        for k in range(300):
            d = int(rng.integers(2, n))
            if d in m.docs:
                new = corpus.text(int(rng.integers(0, n)))
                t.update_document(d, m.docs[d], new)
                m.docs[d] = new
            time.sleep(0.001)
    finally:
        done.set()
        for th in threads:
            th.join()
    assert not errors, errors[:1]
    t.set_mutation_staleness(0.0)  # (every change recorded so far is visible to the next query)
    final = m.rebuilt()
    for terms in (probe, [grams[4], grams[5]]):
        total, docs, scores = b.search(terms, limit=10)
        wt, wp, ws, _ = final.oracle_query(Query(terms, sort_score=True, limit=10))
        assert total == wt and docs.tolist() == wp.tolist() and np.array_equal(scores, ws), terms
