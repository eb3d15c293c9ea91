#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are parity-test shapes, not the bench line
(cut to one GPU's share). Prints one JSON line per config. Not used by the driver.

    python tools/bench_configs.py cfg5 [--docs 12500000] [--batch 8192] [--steps 10]
    python tools/bench_configs.py cfg3 [--docs 1000000]  [--batch 4096] [--steps 10]
    python tools/bench_configs.py cfg4 | cfg2doc | text2
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def cfg5(mg, args):
    """Zipf-by-rank 5-term AND + FILTER category = x, docid DESC page of 100 (no SORT _score)."""
    corpus = mg.Corpus.synthetic(args.docs, seed=42)
    idx = mg.Index(corpus=corpus, ngram_size=2)
    c = idx.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = sorted((g for g in range(c.n_grams) if b" " not in c.gram(g)), key=lambda g: -sizes[g])
    rng = np.random.default_rng(5)
    cat = rng.integers(0, 5, size=c.n_docs, dtype=np.int64)
    fids = [idx.device_index.add_filter_bitmap(np.nonzero(cat == v)[0].astype(np.uint32) + 1) for v in range(5)]
    w = 1.0 / np.arange(1, len(grams) + 1)
    w /= w.sum()
    qs = []
    for _ in range(args.batch):
        pick = rng.choice(len(grams), size=5, replace=False, p=w)
        qs.append(mg.engine.Query([c.gram(grams[i]).decode() for i in pick], filters=[(fids[int(rng.integers(0, 5))], False)],
                                  limit=100, descending=True))
    return idx, qs, {"workload": "cfg5 share: %d docs, bigram, batch %d x 5-term AND (Zipf by rank) + category filter, "
                                 "docid DESC limit 100" % (args.docs, args.batch)}


def cfg3(mg, args):
    """CJK trigram index; 50% AND of 2-4 multi-gram terms, 20% (a OR b) AND c, 20% a AND NOT b, 10% FUZZY 1.
    The corpus is made as UTF-8 bytes with numpy (10M docs as Python strings would take minutes): 3000 ideographs
    (Zipf) + 80 kana, 16-48 code points per doc, every code point 3 bytes."""
    rng = np.random.default_rng(3)
    cps = np.concatenate([0x4E00 + np.arange(3000), 0x3042 + np.arange(80)]).astype(np.uint32)
    wi = 1.0 / np.arange(1, 3001)
    p = np.concatenate([wi / wi.sum() * 0.9, np.full(80, 0.1 / 80)])
    lens = rng.integers(16, 49, size=args.docs)
    flat = cps[rng.choice(len(cps), size=int(lens.sum()), p=p)]
    b = np.empty((len(flat), 3), np.uint8)
    b[:, 0] = 0xE0 | (flat >> 12)
    b[:, 1] = 0x80 | ((flat >> 6) & 0x3F)
    b[:, 2] = 0x80 | (flat & 0x3F)
    del flat
    text_off = np.concatenate([[0], np.cumsum(lens.astype(np.uint64) * 3)]).astype(np.uint64)
    corpus = mg.Corpus(np.concatenate([b.reshape(-1), np.zeros(16, np.uint8)]), text_off)
    del b
    t0 = time.perf_counter()
    idx = mg.Index(corpus=corpus, ngram_size=3, kanji_ngram_size=3)
    build_s = time.perf_counter() - t0

    def term():
        d = int(rng.integers(0, args.docs))
        n = int(rng.integers(3, 7))
        s = int(rng.integers(0, max(1, int(lens[d]) - n)))
        a = int(text_off[d]) + 3 * s
        return corpus.text_bytes[a:a + 3 * n].tobytes().decode("utf-8")

    qs = []
    for i in range(args.batch):
        r = i % 10
        if r < 5:
            qs.append(mg.engine.Query([term() for _ in range(int(rng.integers(2, 5)))], limit=100))
        elif r < 7:
            a, b2, cc = term(), term(), term()
            qs.append(mg.engine.Query(expr=("and", ("or", a, b2), cc), limit=100))
        elif r < 9:
            qs.append(mg.engine.Query([term()], [term()], limit=100))
        else:
            qs.append(mg.engine.Query([term()], fuzzy=1, limit=100))
    return idx, qs, {"workload": "cfg3 share: %d CJK docs (3000 ideographs Zipf + 80 kana, 16-48 cp), trigram, batch %d "
                                 "mixed AND/OR/NOT/FUZZY, docid DESC limit 100" % (args.docs, args.batch),
                     "index_build_s": build_s, "_corpus": corpus}


def cfg2doc(mg, args):
    """The bench.py batch (3 bigrams sampled by df) in docid order: what the operand phase alone costs."""
    corpus = mg.Corpus.synthetic(args.docs, seed=42)
    idx = mg.Index(corpus=corpus, ngram_size=2)
    c = idx.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    cand = [g for g in range(c.n_grams) if b" " not in c.gram(g)]
    w = sizes[cand].astype(np.float64)
    w /= w.sum()
    rng = np.random.default_rng(42)
    qs = []
    for _ in range(args.batch):
        pick = rng.choice(len(cand), size=3, replace=False, p=w)
        qs.append(mg.engine.Query([c.gram(cand[i]).decode() for i in pick], limit=10, descending=True))
    return idx, qs, {"workload": "bench.py batch in docid order: %d docs, batch %d x 3-term AND, docid DESC limit 10"
                                 % (args.docs, args.batch)}


def text2(mg, args):
    """Text-level BM25: 2 whole words per query (each several bigrams long), SORT _score top-10; per
    execute: df pass (text scan of every term's candidates) + scoring with tf from the text."""
    corpus = mg.Corpus.synthetic(args.docs, seed=42)
    idx = mg.Index(corpus=corpus, ngram_size=2)
    rng = np.random.default_rng(9)
    words = sorted({w for i in range(0, args.docs, max(1, args.docs // 4000)) for w in corpus.text(i).decode().split(" ")
                    if len(w) >= 3})
    qs = [mg.engine.Query([str(w) for w in rng.choice(words, size=2, replace=False)], sort_score=True, limit=10)
          for _ in range(args.batch)]
    return idx, qs, {"workload": "text-level BM25: %d docs, bigram, batch %d x 2 whole-word terms, top-10" %
                                 (args.docs, args.batch), "vocabulary_sample": len(words)}


def cfg4(mg, args):
    """One GPU's share of config 4: 12.5M docs, 1024 x 3-term AND (bigrams sampled by df) + BM25 top-100."""
    corpus = mg.Corpus.synthetic(args.docs, seed=42)
    idx = mg.Index(corpus=corpus, ngram_size=2)
    c = idx.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    cand = [g for g in range(c.n_grams) if b" " not in c.gram(g)]
    w = sizes[cand].astype(np.float64)
    w /= w.sum()
    rng = np.random.default_rng(42)
    qs = []
    for _ in range(args.batch):
        pick = rng.choice(len(cand), size=3, replace=False, p=w)
        qs.append(mg.engine.Query([c.gram(cand[i]).decode() for i in pick], sort_score=True, limit=100))
    return idx, qs, {"workload": "cfg4 share: %d docs, bigram, batch %d x 3-term AND + BM25 top-100" % (args.docs, args.batch)}


def score5(mg, args):
    """The 4-5-scored-term path: 10M docs, 1024 x 5-term AND (bigrams sampled by df) + BM25 top-10
    (bitmap_score_kernel<5>; five tf gathers + five block-max rows per surviving match)."""
    corpus = mg.Corpus.synthetic(args.docs, seed=42)
    idx = mg.Index(corpus=corpus, ngram_size=2)
    c = idx.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    cand = [g for g in range(c.n_grams) if b" " not in c.gram(g)]
    w = sizes[cand].astype(np.float64)
    w /= w.sum()
    rng = np.random.default_rng(42)
    qs = []
    for i in range(args.batch):
        pick = rng.choice(len(cand), size=5 if i % 2 else 4, replace=False, p=w)
        qs.append(mg.engine.Query([c.gram(cand[i]).decode() for i in pick], sort_score=True, limit=10))
    return idx, qs, {"workload": "%d docs, bigram, batch %d x 4- and 5-term AND + BM25 top-10" % (args.docs, args.batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["cfg3", "cfg4", "cfg5", "cfg2doc", "text2", "score5"])
    ap.add_argument("--docs", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--check", type=int, default=0, help="sample this many queries of the batch against the CPU oracle")
    args = ap.parse_args()
    args.docs = args.docs or {"cfg3": 1_000_000, "cfg5": 12_500_000, "cfg2doc": 10_000_000, "text2": 2_000_000, "cfg4": 12_500_000, "score5": 10_000_000}[args.config]
    args.batch = args.batch or {"cfg3": 4096, "cfg5": 8192, "cfg2doc": 1024, "text2": 1024, "cfg4": 1024, "score5": 1024}[args.config]
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    mg = entry.load_package()
    t0 = time.perf_counter()
    idx, qs, cfg = {"cfg3": cfg3, "cfg5": cfg5, "cfg2doc": cfg2doc, "text2": text2, "cfg4": cfg4, "score5": score5}[args.config](mg, args)
    batch = idx.prepare(qs)
    if args.config == "text2":
        # a serving loop prepares a fresh batch per step: after the first one the df of every term it has seen is in the
        # index's cache (the index is static) and the df pass has nothing left to count
        batch.execute()
        batch.fetch_raw()
        batch = idx.prepare(qs)
    setup = time.perf_counter() - t0
    for _ in range(args.warmup):
        batch.execute()
        batch.fetch_raw()
    batch.kernel_time_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.execute()
        batch.fetch_raw()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms, k_n = batch.kernel_time_ms()
    res = batch.fetch()
    cfg.update({"index_bytes_hbm": idx.device_index.memory_bytes(), "grams": idx.columns.n_grams,
                "postings": idx.columns.n_postings, "setup_s": setup,
                "mean_total": float(np.mean([r.total for r in res])), "device_queries": batch.n})
    corpus_for_check = cfg.pop("_corpus", None)
    if args.check:
        # a seeded sample of the batch against the oracle at THIS size (docids, totals, funnel counters)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from gpu_util import Pair
        from oracle import oracle as O
        c = idx.columns
        pr = Pair.__new__(Pair)
        pr.dev, pr.corpus, pr.filters = idx, corpus_for_check, {}
        pr.oidx = O.Index.from_csr(idx.ngram_size, idx.kanji_ngram_size, idx.cross_boundary, c.key_bytes, c.key_off, c.offsets, c.docids)
        pr.ostore = O.DocumentStore.from_arrays(corpus_for_check.text_bytes, corpus_for_check.text_off) if corpus_for_check is not None else None
        pr.N, pr.total_len, pr.avgdl = c.bm25_doc_count, c.bm25_total_len, c.avg_doc_length()
        rng = np.random.default_rng(99)
        bad = 0
        picks = rng.choice(len(qs), size=min(args.check, len(qs)), replace=False).tolist()
        for i in picks:
            q, g = qs[i], res[i]
            if q.fuzzy:
                r = O.execute_fuzzy(pr.oidx, pr.ostore, q.terms, q.fuzzy, ngram_size=idx.ngram_size,
                                    kanji_ngram_size=idx.kanji_ngram_size, cross_boundary=idx.cross_boundary)
                want_total, want_page = len(r["results"]), r["results"][::-1][: q.limit]
            else:
                want_total, want_page, _, _ = pr.oracle_query(q)
            if g.total != want_total or g.docs.tolist() != want_page.tolist():
                bad += 1
        cfg["oracle_checked"], cfg["oracle_mismatches"] = len(picks), bad
    print(json.dumps({"metric": "queries/sec", "value": len(qs) * args.steps / dt, "unit": "queries/s",
                      "ms_per_step": 1e3 * dt / args.steps, "tile_kernel_ms": k_ms, "steps": args.steps, "config": cfg}))


if __name__ == "__main__":
    main()
