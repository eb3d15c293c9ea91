"""One rank, RCCL for real (launched by tests/test_gpu_dist.py): the nccl branch of the sharded path — the collective
inside the library (mgx_batch_exchange / mgx_batch_exchange_df over an mgx_comm of world 1) through dist.ShardedTable.run
AND through the C++ executor (search_pipeline::BatchExecutor with Options::comm) — against the oracle. A one-rank
all-gather moves the shard's blob through RCCL into the gather buffer and the merge kernel runs on it, which is every
line the multi-rank path executes except the wire."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pkg import mg  # noqa: E402
from mygram_db_amd import dist as mdist  # noqa: E402
from mygram_db_amd import _shim_capi as S  # noqa: E402
from oracle import oracle as O  # noqa: E402
from dist_worker import queries_for  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ["MGX_FORCE_EXCHANGE"] = "1"  # one rank still takes the exchange path
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    corpus = mg.Corpus.synthetic(60_000, seed=11)
    table = mdist.ShardedTable(corpus, first_doc_id=1, device=0, ngram_size=2, kanji_ngram_size=0)
    assert table.force_exchange
    cols = table.index.columns
    oidx = O.Index.from_csr(2, 0, True, cols.key_bytes, cols.key_off, cols.offsets, cols.docids)
    ostore = O.DocumentStore.from_arrays(corpus.text_bytes, corpus.text_off)
    n, avg = cols.bm25_doc_count, cols.avg_doc_length()
    sizes = np.diff(cols.offsets.astype(np.int64))
    keys = [cols.gram(g) for g in range(cols.n_grams)]
    qs = queries_for(keys, sizes.tolist(), n=48, limit=10)
    qs.append(mg.engine.Query(["the", "an"], sort_score=True, limit=10))  # text-level terms: the df all-reduce
    # (1) dist.ShardedTable.run: nccl backend -> mgx_batch_exchange_df + execute + mgx_batch_exchange
    b = table.prepare(qs)
    table.run(b)
    got = b.fetch()
    for q, g in zip(qs, got):
        total, docs, scores = O.search_scored(oidx, ostore, q.terms, n, avg, limit=q.limit, offset=q.offset)
        assert g.total == total and g.docs.tolist() == docs.tolist(), q.terms
        assert np.array_equal(g.scores, scores), q.terms
    # docid-ordered pages take the export-by-copy branch of the exchange
    pq = [mg.engine.Query(q.terms, limit=7) for q in qs[:16]]
    b2 = table.prepare(pq)
    table.run(b2)
    for q, g in zip(pq, b2.fetch()):
        r = O.execute(oidx, ostore, q.terms, [], [], compute_df=False, ngram_size=2, kanji_ngram_size=0)
        res = r["results"]
        assert g.total == len(res) and g.docs.tolist() == res[::-1][:7].tolist(), q.terms
    # (2) the C++ executor with a communicator: two batches in flight, each exchanged on its own stream
    comm = mdist.Comm()
    ex = S.Executor(S.Table(table.index), depth=2, planner_threads=2, comm=comm)
    term_lists = [q.terms for q in qs if q.offset == 0]
    qb = S.QueryBatch(term_lists)
    t1, t2 = ex.submit(qb, limit=10), ex.submit(qb, limit=10)
    for t in (t1, t2):
        totals, n_docs, docs, scores, _ = ex.wait(t)
        for qi, terms in enumerate(term_lists):
            total, d, s = O.search_scored(oidx, ostore, terms, n, avg, limit=10)
            assert totals[qi] == total and docs[qi, :n_docs[qi]].tolist() == d.tolist(), terms
            assert np.array_equal(scores[qi, :n_docs[qi]], s), terms
    # (3) df cache and the all-reduce: a text-level term's df was cached by the batches above (the index keeps this shard's
    # LOCAL count); a later sharded batch must still contribute that count to the reduction — with the round-2 cache (the
    # reduced value stored, zero contributed) a warm rank would have scored with df 0
    warm = [mg.engine.Query(["the", "an"], sort_score=True, limit=10)]
    for _ in range(2):
        bw = table.prepare(warm)
        table.run(bw)
        g = bw.fetch()[0]
        total, d, s = O.search_scored(oidx, ostore, warm[0].terms, n, avg, limit=10)
        assert g.total == total and g.docs.tolist() == d.tolist() and np.array_equal(g.scores, s)
    # (4) a rank-local failure between compile and exchange poisons the communicator: that ticket fails, and every later
    # ticket fails at once instead of pairing its collectives with another batch's on the peers
    lib = mg._capi.load()
    ex2 = S.Executor(S.Table(table.index), depth=2, planner_threads=2, comm=mdist.Comm())
    t_ok = ex2.submit(qb, limit=10)
    ex2.wait(t_ok)
    lib.mgxt_fail_device_allocs(0, 1000)  # every device allocation of the next batch fails
    t_bad = ex2.submit(S.QueryBatch([t + ["zq"] for t in term_lists[:8]] + term_lists), limit=100)
    failed = False
    try:
        ex2.wait(t_bad)
    except S.ShimError:
        failed = True
    lib.mgxt_fail_device_allocs(0, 0)
    assert failed, "the injected allocation failure must surface as an error"
    t_after = ex2.submit(qb, limit=10)
    try:
        ex2.wait(t_after)
        raise AssertionError("a ticket after the failure must not run on an aborted communicator")
    except S.ShimError as e:
        assert "aborted" in str(e), str(e)
    dist.destroy_process_group()
    print("ok")


if __name__ == "__main__":
    main()
