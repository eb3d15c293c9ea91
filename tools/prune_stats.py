"""Debug (ablation build, MGX_DEBUG_SKIP=64): how many matches the score kernel actually scores after block-max pruning."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
mg = entry.load_package()
import bench as B
from mygram_db_amd import dist as mdist
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
corpus = mg.Corpus.synthetic(n_docs, seed=42)
table = mdist.ShardedTable(corpus, first_doc_id=1, device=0, ngram_size=2, kanji_ngram_size=0, dense_threshold=0.0)
tb = B.make_queries(mg, table, 1, 1024)
qs = [mg.engine.Query(t, sort_score=True, limit=10) for t in tb[0]]
r = table.index.search_batch(qs)
tot = np.asarray([x.total for x in r], dtype=np.float64)
sc = np.asarray([x.after_filters for x in r], dtype=np.float64) - tot  # (the counter holds matches + scored)
by = np.argsort(-sc)
cum = np.cumsum(sc[by]) / max(1.0, sc.sum())
dens = tot / n_docs
classes = {}
for lo, hi in ((0, .001), (.001, .005), (.005, .02), (.02, .05), (.05, .1), (.1, 1.1)):
    m = (dens >= lo) & (dens < hi)
    classes["%g-%g" % (lo, hi)] = (int(m.sum()), float(tot[m].sum()), float(sc[m].sum()))
order = np.argsort(-tot)
print(json.dumps({"matches": tot.sum(), "scored": sc.sum(), "frac": sc.sum() / tot.sum(),
                  "top10_by_matches": [(int(tot[i]), int(sc[i])) for i in order[:10]],
                  "median_query": (float(np.median(tot)), float(np.median(sc))),
                  "cum_share_of_scored_at": {k: round(float(cum[k - 1]), 3) for k in (10, 30, 100, 300, 600)},
                  "by_match_density(n_queries, matches, scored)": classes}))
