"""Worker of the world_size-2 sharding tests (launched by tests/test_dist_cpu.py / tests/test_gpu_dist.py).

mode=cpu : gloo, no device. Checks the sharding design with host code only: shard ranges, the global statistics every
           rank must agree on (gram sizes aligned by gram bytes, BM25 N / total length), and that per-shard top-k
           lists scored with the GLOBAL idf merge into exactly the unsharded ranking (oracle as the checker).
mode=gpu : gloo for the exchange (both ranks share the one GPU of the box, which RCCL refuses), device kernels for
           everything else: ShardedTable.run -> export_topk -> all-gather -> merge_shards kernel -> fetch, compared
           with the oracle on the whole corpus.
"""
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
from oracle import oracle as O  # noqa: E402

N_DOCS = 40_000


def queries_for(keys, sizes, n=24, limit=10):
    rng = np.random.default_rng(5)
    cand = [k for k, s in zip(keys, sizes) if b" " not in k and s > 0]
    w = np.asarray([s for k, s in zip(keys, sizes) if b" " not in k and s > 0], dtype=np.float64)
    out = []
    for i in range(n):
        pick = rng.choice(len(cand), size=3, replace=False, p=w / w.sum())
        out.append(mg.engine.Query([cand[j].decode() for j in pick], sort_score=True, limit=limit,
                                   offset=[0, 2][i % 2]))
    return out


def table_texts():
    """The synthetic corpus plus a few grams that exist in ONE half of the doc range only (the synthetic vocabulary
    puts every letter bigram into every shard)."""
    base = mg.Corpus.synthetic(N_DOCS, seed=11)
    texts = [base.text(i) for i in range(N_DOCS)]
    for i in (100, 2_000, 2_001, 15_000):
        texts[i] = texts[i] + b" k9k"
    for i in (25_000, 31_000, 39_999):
        texts[i] = texts[i] + b" 7j7 the"
    return texts


def oracle_whole(texts):
    corpus = mg.Corpus.from_texts(texts)
    cols = mg.Columns(corpus, 1, 2, 0, True)
    oidx = O.Index.from_csr(2, 0, True, cols.key_bytes, cols.key_off, cols.offsets, cols.docids)
    ostore = O.DocumentStore.from_arrays(corpus.text_bytes, corpus.text_off)
    return corpus, cols, oidx, ostore


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    before, mine = mdist.shard_range(N_DOCS, rank, world)
    assert sum(mdist.shard_range(N_DOCS, r, world)[1] for r in range(world)) == N_DOCS
    texts = table_texts()
    shard_corpus = mg.Corpus.from_texts(texts[before:before + mine])
    corpus, cols, oidx, ostore = oracle_whole(texts)
    keys_w = [cols.gram(g) for g in range(cols.n_grams)]
    sizes_w = np.diff(cols.offsets.astype(np.int64))

    if mode == "cpu":
        scols = mg.Columns(shard_corpus, 1 + before, 2, 0, True)
        keys = [scols.gram(g) for g in range(scols.n_grams)]
        sizes = np.diff(scols.offsets.astype(np.int64))
        gsizes, n, total_len, gdict = mdist.global_table_stats(keys, sizes, scols.bm25_doc_count,
                                                               scols.bm25_total_len)
        whole = dict(zip(keys_w, sizes_w.tolist()))
        assert [whole[k] for k in keys] == gsizes.tolist()
        assert gdict == whole  # union of the shards' dictionaries with table-wide sizes
        assert (n, total_len) == (cols.bm25_doc_count, cols.bm25_total_len)
        # per-shard oracle top-k with GLOBAL statistics, gathered and merged == unsharded oracle ranking
        sidx = O.Index.from_csr(2, 0, True, scols.key_bytes, scols.key_off, scols.offsets, scols.docids)
        shard_texts = [shard_corpus.text(i) for i in range(mine)]
        sstore = O.DocumentStore()
        for i, t in enumerate(shard_texts):
            sstore.add(1 + before + i, t)
        avg = total_len / n
        for q in queries_for(keys_w, sizes_w):
            terms = sorted(q.terms, key=lambda t: whole[t.encode()])
            dfs = [whole[t.encode()] for t in terms]
            res = sidx.search_and(terms)
            sc = O.score_documents(sstore, res, terms, dfs, n, avg)
            top = O.sort_by_score(res, sc, True, q.offset + q.limit, 0)
            lookup = dict(zip(res.tolist(), sc.tolist()))
            mine_list = [(lookup[d], d) for d in top.tolist()]
            gathered = [None] * world
            dist.all_gather_object(gathered, (mine_list, len(res)))
            merged = sorted((x for lst, _ in gathered for x in lst), key=lambda x: (-x[0], -x[1]))
            page = merged[q.offset:q.offset + q.limit]
            total, docs, scores = O.search_scored(oidx, ostore, q.terms, n, avg, limit=q.limit, offset=q.offset)
            assert sum(c for _, c in gathered) == total
            assert [d for _, d in page] == docs.tolist()
            assert [s for s, _ in page] == scores.tolist()
    else:
        torch.cuda.set_device(0)
        table = mdist.ShardedTable(shard_corpus, first_doc_id=1 + before, device=0)
        assert table.index.total_docs == cols.bm25_doc_count
        qs = queries_for(keys_w, sizes_w)
        # grams that only ONE shard holds: the other shard must still run the query (as an empty operand) so that
        # both ranks exchange blobs of the same layout
        half = 1 + mdist.shard_range(N_DOCS, 1, world)[0]
        one_sided = []
        for g in range(cols.n_grams):
            ids = cols.docids[int(cols.offsets[g]):int(cols.offsets[g + 1])]
            if len(ids) and (ids.max() < half or ids.min() >= half):
                one_sided.append(cols.gram(g).decode())
        assert one_sided
        common = [k.decode() for k, sz in zip(keys_w, sizes_w) if sz > N_DOCS // 4 and b" " not in k][:3]
        assert {"k9", "7j"} <= set(one_sided)
        for i, g in enumerate(one_sided[:8]):
            qs.append(mg.engine.Query([g, common[i % len(common)]], sort_score=True, limit=10))
            qs.append(mg.engine.Query([g], sort_score=True, limit=5))
        # text-level terms (N1): df is summed over the shards before idf is taken
        words = sorted({w for i in range(0, N_DOCS, 997) for w in corpus.text(i).decode().split(" ") if len(w) >= 3})
        rng = np.random.default_rng(8)
        for i in range(8):
            terms = [str(w) for w in rng.choice(words, size=2, replace=False)]
            if i % 2:
                terms.append(common[0])
            qs.append(mg.engine.Query(terms, sort_score=True, limit=10, offset=i % 3))
        qs.append(mg.engine.Query(["k9k"], sort_score=True, limit=10))         # text-level, one shard only
        qs.append(mg.engine.Query(["7j7", "the"], sort_score=True, limit=10))  # ... mixed with a term every shard has
        batch = table.prepare(qs)
        for _ in range(2):  # a prepared batch can be run repeatedly
            table.run(batch)
            got = batch.fetch()
        n, avg = cols.bm25_doc_count, cols.avg_doc_length()
        for q, g in zip(qs, got):
            total, docs, scores = O.search_scored(oidx, ostore, q.terms, n, avg, limit=q.limit, offset=q.offset)
            assert g.total == total, (q.terms, g.total, total)
            assert g.docs.tolist() == docs.tolist(), q.terms
            assert np.array_equal(g.scores, scores), q.terms
        # The df cache must not change what a rank contributes to the df all-reduce: rank 0 ALONE first runs a text-level
        # term through a plain (unsharded) execute of its shard — its index now knows the term's LOCAL count — then both
        # ranks run the sharded batch, rank 1 counting for the first time. (The round-2 cache stored whatever count came
        # back — here rank 0's local one under the table-wide key — and a warm rank contributed 0 to the reduction.)
        cold_terms = [str(w) for w in words[-3:]]
        if rank == 0:
            lone = table.index.prepare([mg.engine.Query([cold_terms[0], cold_terms[1]], sort_score=True, limit=5)])
            lone.execute()
            lone.fetch()
        dist.barrier()
        cq = [mg.engine.Query([cold_terms[0], cold_terms[1]], sort_score=True, limit=10),
              mg.engine.Query([cold_terms[1], cold_terms[2], common[0]], sort_score=True, limit=10)]
        for _ in range(2):  # (second round: both ranks warm)
            cb = table.prepare(cq)
            table.run(cb)
            for q, g in zip(cq, cb.fetch()):
                total, docs, scores = O.search_scored(oidx, ostore, q.terms, n, avg, limit=q.limit)
                assert g.total == total and g.docs.tolist() == docs.tolist() and np.array_equal(g.scores, scores), q.terms
        # fresh batches of different shapes, each dropped before the next is prepared (a serving loop): the exchange
        # buffers belong to the batch, so a recycled Python id / device address must never leak a stale blob or layout
        for rnd in range(6):
            sub = qs[rnd::6][: 5 + 3 * rnd]
            lim = [10, 3, 25, 1, 50, 7][rnd]
            sub = [mg.engine.Query(q.terms, sort_score=True, limit=lim, offset=rnd % 3) for q in sub]
            fb = table.prepare(sub)
            table.run(fb)
            for q, g in zip(sub, fb.fetch()):
                total, docs, scores = O.search_scored(oidx, ostore, q.terms, n, avg, limit=q.limit, offset=q.offset)
                assert g.total == total and g.docs.tolist() == docs.tolist() and np.array_equal(g.scores, scores), \
                    (rnd, q.terms)
            del fb
        # the C++ planner (search_pipeline::BatchExecutor) on a shard: a gram only the OTHER shard holds must stay a device
        # query (MGX_GRAM_ABSENT, Index::SetAbsentGrams) so that every rank's batch — and therefore its collectives — has
        # the same shape; locally it simply matches nothing. (No communicator here: two ranks share one GPU, which RCCL
        # refuses; what is checked is the planning every rank must agree on.)
        from mygram_db_amd import _shim_capi as S
        ex = S.Executor(S.Table(table.index), depth=1, planner_threads=2)
        tl = [[g, common[0]] for g in one_sided[:6]] + [[common[0], common[1]], ["~#"]]
        totals, n_docs, docs, scores, timing = ex.wait(ex.submit(S.QueryBatch(tl), limit=10))
        counts = torch.tensor([int(timing[4])], dtype=torch.int64)
        gathered = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gathered, counts)
        assert all(int(x) == len(tl) - 1 for x in gathered), [int(x) for x in gathered]  # ("~#": unknown to the whole table)
        tot = torch.from_numpy(totals.astype(np.int64).copy())
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)  # shards are disjoint doc ranges: local totals add up to the table's
        for terms, t in zip(tl, tot.tolist()):
            want = O.search_scored(oidx, ostore, terms, cols.bm25_doc_count, cols.avg_doc_length(), limit=10)[0]
            assert t == want, (terms, t, want)
        # docid-ordered pages across the shards (the reference's default order): totals add up, pages merge by doc id
        pq = []
        for i in range(10):
            pick = [common[i % len(common)], common[(i + 1) % len(common)]]
            pq.append(mg.engine.Query(pick, limit=[1, 10, 100][i % 3], descending=bool(i % 2)))
        pq.append(mg.engine.Query(["k9"], limit=10, descending=True))              # only the first shard holds it
        pq.append(mg.engine.Query(["7j", common[0]], limit=10, descending=False))  # only the second
        pq.append(mg.engine.Query(["the"], ["an"], limit=50, descending=True))
        pbatch = table.prepare(pq)
        table.run(pbatch)
        for q, g in zip(pq, pbatch.fetch()):
            r = O.execute(oidx, ostore, q.terms, q.not_terms, [], compute_df=False, ngram_size=2, kanji_ngram_size=0,
                          cross_boundary=True)
            res = r["results"]
            page = (res[::-1] if q.descending else res)[: q.limit]
            assert g.total == len(res), (q.terms, g.total, len(res))
            assert g.docs.tolist() == page.tolist(), (q.terms, g.docs.tolist()[:5], page.tolist()[:5])
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
