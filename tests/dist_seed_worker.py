"""Worker of tests/test_gpu_dist.py::test_seed_bound_exchange_world2: two ranks (gloo) share the box's GPU, each holds a
1.2M-doc shard — large enough for seed items —, and run SORT _score batches through ShardedTable.run, i.e. through
mgx_batch_execute_gather: the ranks' seed keys are all-gathered and every query's pruning bound is raised to the
(offset+limit)-th best of the union before the rest of the batch runs. Checked against the UNSHARDED index of the same
table on the same device (that path is held against the oracle by the other GPU tests): totals, pages, scores of every
query identical."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402

mg = entry.load_package()
import bench as B  # noqa: E402
from mygram_db_amd import dist as mdist  # noqa: E402

N_DOCS = 2_400_000


def main():
    os.environ["MGX_VERBOSE"] = "1"
    os.environ["MGX_SEED_EXCHANGE"] = "1"  # (by default only tables cut eight ways and more exchange their seeds)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    before, mine = mdist.shard_range(N_DOCS, rank, world)
    corpus = mg.Corpus.synthetic(mine, seed=42, global_first=before)
    table = mdist.ShardedTable(corpus, first_doc_id=1 + before, device=0, ngram_size=2, kanji_ngram_size=0,
                               dense_threshold=0.0)
    terms = B.make_queries(mg, table, 1, 384)[0]
    Q = mg.engine.Query
    qs = []
    for i, t in enumerate(terms):
        if i % 7 == 3:
            qs.append(Q(t, sort_score=True, limit=100))
        elif i % 7 == 5:
            qs.append(Q(t[:2], sort_score=True, limit=10, offset=5))
        elif i % 11 == 0:
            qs.append(Q(t, sort_score=True, limit=10, descending=False))  # ASC: no pruning, no seed keys
        else:
            qs.append(Q(t, sort_score=True, limit=10))
    batch = table.prepare(qs)
    for _ in range(2):
        table.run(batch)
        got = batch.fetch()
    whole = mg.Index(corpus=mg.Corpus.synthetic(N_DOCS, seed=42), ngram_size=2, kanji_ngram_size=0)
    want = whole.search_batch(qs)
    hits = 0
    for q, g, w in zip(qs, got, want):
        assert g.total == w.total, (q.terms, g.total, w.total)
        assert g.docs.tolist() == w.docs.tolist(), (q.terms, q.limit, q.offset, q.descending)
        assert np.array_equal(g.scores, w.scores), q.terms
        hits += w.total > 0
    assert hits > 300
    print("ok")


if __name__ == "__main__":
    main()
