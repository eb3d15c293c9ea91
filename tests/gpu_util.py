"""Shared helpers of the -m gpu parity tests: build the same corpus on the device (through the C ABI) and in the
oracle, and run one query through both."""
import numpy as np

import golden_util as G
from oracle import oracle as O
from pkg import mg


def densify(docs):
    """[(doc_id, text)] with arbitrary ids -> (first_doc_id, texts over the dense id range; gaps = empty text,
    which produces no n-grams and no BM25 weight, i.e. the doc does not exist for the hot path)."""
    ids = [d for d, _ in docs]
    first, last = min(ids), max(ids)
    texts = [""] * (last - first + 1)
    for d, t in docs:
        texts[d - first] = t if t is not None else ""
    return first, texts


class Pair:
    """The same index on the device (mg.Index) and in the oracle."""

    def __init__(self, docs=None, corpus=None, first_doc_id=1, ngram=2, kanji=0, cross=True, dense_threshold=0.0):
        if corpus is None:
            first_doc_id, texts = densify(docs)
            corpus = mg.Corpus.from_texts(texts)
        self.corpus = corpus
        self.dev = mg.Index(corpus=corpus, first_doc_id=first_doc_id, ngram_size=ngram, kanji_ngram_size=kanji,
                            cross_boundary=cross, dense_threshold=dense_threshold)
        c = self.dev.columns
        # the oracle adopts the CSR arrays; tests/test_host_cpu.py pins those arrays to the oracle's own text-level build
        self.oidx = O.Index.from_csr(ngram, kanji, cross, c.key_bytes, c.key_off, c.offsets, c.docids)
        if first_doc_id == 1:
            self.ostore = O.DocumentStore.from_arrays(corpus.text_bytes, corpus.text_off)
        else:
            self.ostore = O.DocumentStore()
            for i in range(corpus.n_docs):
                t = corpus.text(i)
                if t:
                    self.ostore.add(first_doc_id + i, t)
        self.N, self.total_len = c.bm25_doc_count, c.bm25_total_len
        self.avgdl = c.avg_doc_length()
        self.filters = {}  # bitmap id -> sorted docids

    def add_filter(self, docids):
        docids = np.asarray(sorted(docids), dtype=np.uint32)
        bid = self.dev.device_index.add_filter_bitmap(docids)
        self.filters[bid] = docids
        return bid

    def oracle_expr(self, e, universe):
        """EvaluateBooleanAstExpanded (src/server/search_pipeline.cpp:327-378) with the oracle's set operations."""
        if isinstance(e, str):
            norm = O.normalize_text(e)
            grams = sorted(set(O.generate_query_ngrams(norm, self.dev.ngram_size, self.dev.kanji_ngram_size,
                                                      self.dev.cross_boundary)))
            return set(self.oidx.search_and(grams).tolist()) if grams else set()
        kids = [self.oracle_expr(c, universe) for c in e[1:]]
        if e[0] == "and":
            out = kids[0]
            for k in kids[1:]:
                out = out & k
            return out
        if e[0] == "or":
            out = set()
            for k in kids:
                out |= k
            return out
        return universe - kids[0]

    def oracle_query(self, q):
        """(total, page docids, page scores or None, funnel dict) the reference would produce."""
        filters = [(self.filters[bid], neg) for bid, neg in q.filters]
        if q.expr is not None:
            c = self.dev.columns
            first, count = q.universe if q.universe is not None else (c.first_doc_id, c.n_docs)
            universe = set(range(max(first, c.first_doc_id), min(first + count, c.first_doc_id + c.n_docs)))
            res = self.oracle_expr(q.expr, universe)
            for nt in q.not_terms:
                res -= self.oracle_expr(nt, universe)
            for docs, neg in filters:
                res = (res - set(docs.tolist())) if neg else (res & set(docs.tolist()))
            res = np.asarray(sorted(res), dtype=np.uint32)
            page = res[::-1] if q.descending else res
            if q.limit:
                page = page[: q.limit]
            return len(res), page, None, {"empty_term_detected": True}
        # the C oracle restates the non-ICU NormalizeText branch; the ICU branch is applied here (idempotent under it)
        q_terms = [O.normalize_text(t) for t in q.terms]
        q_not = [O.normalize_text(t) for t in q.not_terms]
        r = O.execute(self.oidx, self.ostore, q_terms, q_not, filters, compute_df=q.sort_score,
                      ngram_size=self.dev.ngram_size, kanji_ngram_size=self.dev.kanji_ngram_size,
                      cross_boundary=self.dev.cross_boundary, verify_text=q.verify_text)
        res = r["results"]
        if q.sort_score:
            terms = [q_terms[i] for i in r["term_order"]]
            sc = O.score_documents(self.ostore, res, terms, r["term_df"], self.N, self.avgdl, q.k1, q.b)
            page = O.sort_by_score(res, sc, q.descending, q.limit, q.offset)
            lookup = dict(zip(res.tolist(), sc.tolist()))
            return len(res), page, np.asarray([lookup[d] for d in page.tolist()]), r
        page = res[::-1] if q.descending else res
        if q.limit:
            page = page[: q.limit]
        return len(res), page, None, r

    def check(self, queries):
        got = self.dev.search_batch(queries)
        for q, g in zip(queries, got):
            total, page, scores, r = self.oracle_query(q)
            ctx = (q.terms, q.not_terms, q.filters, q.limit, q.offset, q.descending)
            assert g.total == total, ctx
            assert g.docs.tolist() == page.tolist(), ctx
            if scores is not None:
                # fp64 arithmetic in the reference's operation order: bit-exact, far inside the 1e-5 relative bar
                assert np.array_equal(g.scores, scores), (ctx, g.scores, scores)
            if q.expr is not None:
                continue
            if not r["empty_term_detected"]:
                for k in ("total_candidates", "after_intersection", "after_not", "after_filters"):
                    assert getattr(g, k) == r[k], (ctx, k)
            else:
                assert g.empty_term_detected
        return got
