"""Replay-only timing of the score kernels on the benchmark batch (tools; not the headline):
python tools/kernel_sweep.py [n_docs]  — env knobs (MGX_*) and MGX_LIBRARY select what is measured."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as entry
mg = entry.load_package()
import bench as B
from mygram_db_amd import dist as mdist

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
corpus = mg.Corpus.synthetic(n_docs, seed=42)
table = mdist.ShardedTable(corpus, first_doc_id=1, device=0, ngram_size=2, kanji_ngram_size=0, dense_threshold=0.0)
tb = B.make_queries(mg, table, 4, 1024)
qs = [[mg.engine.Query(t, sort_score=True, limit=10) for t in b] for b in tb]
batches = [table.prepare(q) for q in qs]
for i in range(4):
    table.run(batches[i]); batches[i].fetch_raw()
for b in batches:
    b.kernel_time_ms()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 24
for i in range(N):
    table.run(batches[i % 4]); batches[i % 4].fetch_raw()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / N
k = [b.kernel_time_ms() for b in batches]
kms = sum(ms * n for ms, n in k) / max(1, sum(n for ms, n in k))
r = batches[0].fetch()
chk = int(sum(int(x.total) for x in r)) ^ int(sum(int(d) for x in r for d in x.docs.tolist()))
print(json.dumps({"lib": os.environ.get("MGX_LIBRARY", "default"), "env": {k: v for k, v in os.environ.items() if k.startswith("MGX_") and k != "MGX_LIBRARY"},
                  "kernel_ms": round(kms, 4), "step_ms": round(el * 1e3, 4), "check": chk}))
