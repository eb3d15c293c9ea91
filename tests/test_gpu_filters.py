"""-m gpu: FILTER conditions and FACET through the C++ planner (libmygram_shim.so) on typed device columns, against
oracle/filters.py — the restatement of ApplyFiltersWithBitmap / ApplyFilters / BuildTypeUnionBitmap
(src/server/search_pipeline.cpp:1021-1237) and of FilterIndex::GetColumnValueCountsFiltered (filter_index.cpp:284-312),
itself pinned by the reference's own vectors (tests/golden/filters.json, checked on the CPU)."""
import json
import os

import numpy as np
import pytest

from gpu_util import Pair
from oracle import filters as F
from pkg import mg

pytestmark = pytest.mark.gpu
Query = mg.engine.Query
HERE = os.path.dirname(os.path.abspath(__file__))


def _table(p):
    from mygram_db_amd import _shim_capi as S
    return S.Table(p.dev)


def test_reference_filter_parity_fixture_on_the_device():
    """tests/server/search_pipeline_test.cpp:725-910: status int64 / category string / score double, doc 4 all NULL."""
    g = json.load(open(os.path.join(HERE, "golden", "filters.json")))
    fx = g["fixture"]
    texts = ["text zero", "text one", "text two", "text three", "text four"]
    p = Pair(docs=[(i + 1, t) for i, t in enumerate(texts)], ngram=2, kanji=0)
    t = _table(p)
    nul = [v is None for v in fx["status"]]
    t.add_filter_column("status", "int64", [0 if v is None else v[1] for v in fx["status"]], nul)
    t.add_filter_column("category", "string", ["" if v is None else v[1] for v in fx["category"]], nul)
    t.add_filter_column("score", "double", [0.0 if v is None else v[1] for v in fx["score"]], nul)
    for c in g["cases"]:
        if "docs" in c:
            continue  # (a restricted candidate list: covered by the oracle-level vector)
        total, docs, _ = t.search(["text"], conditions=[tuple(x) for x in c["conditions"]], descending=False, limit=10)
        if "expect" in c:
            assert docs.tolist() == [d + 1 for d in c["expect"]] and total == len(c["expect"]), c
    # the column name resolves case-insensitively (:818-830); NE keeps the NULL document (:1151-1157)
    total, docs, _ = t.search(["text"], conditions=[("CATEGORY", "=", "tech")], descending=False, limit=10)
    assert docs.tolist() == [1, 3]
    total, docs, _ = t.search(["text"], conditions=[("status", "!=", "1")], descending=False, limit=10)
    assert docs.tolist() == [2, 4, 5]
    # FACET (tests/server/facet_handler_test.cpp:203-258): every document / a search / offset before limit
    matched, n_values, page = t.facet("category")
    assert matched == 5 and n_values == 3 and page[0] == (b"tech", 2) and dict(page) == {b"tech": 2, b"sports": 1, b"music": 1}
    matched, n_values, page = t.facet("Category", terms=["zero"])
    assert (matched, n_values, page) == (1, 1, [(b"tech", 1)])
    matched, n_values, page = t.facet("category", limit=1, offset=1)
    assert n_values == 3 and len(page) == 1 and page[0][1] == 1
    with pytest.raises(Exception):
        t.facet("no_such_column")


def test_random_typed_columns_conditions_and_facets_match_the_oracle():
    n = 120_000
    p = Pair(corpus=mg.Corpus.synthetic(n, seed=31))
    t = _table(p)
    rng = np.random.default_rng(32)
    cols = {}

    def add(name, vtype, values, null_frac):
        nul = rng.random(n) < null_frac
        t.add_filter_column(name, vtype, values, nul)
        vals = list(values)
        cols[name] = (vtype, vals, nul)

    add("status", "int32", rng.integers(-3, 10, n), 0.05)
    add("category", "string", [("cat%02d" % k) for k in rng.integers(0, 20, n)], 0.02)
    add("score", "double", rng.integers(0, 400, n) / 4.0, 0.1)
    add("small", "uint16", rng.integers(0, 60000, n), 0.0)
    add("flag", "bool", rng.integers(0, 2, n), 0.3)
    add("when", "time", rng.integers(-5000, 5000, n), 0.01)
    add("big", "uint64", rng.integers(0, 2 ** 62, n, dtype=np.uint64) * np.uint64(3), 0.0)
    add("tiny", "int8", rng.integers(-128, 128, n), 0.0)

    def lookup(name):
        vtype, vals, nul = cols[name]

        def f(doc):
            i = doc - 1
            if nul[i]:
                return None
            v = vals[i]
            return (vtype, v if vtype == "string" else (float(v) if vtype == "double" else int(v)))
        return f
    columns = {name: lookup(name) for name in cols}
    c = p.dev.columns
    sizes = np.diff(c.offsets.astype(np.int64))
    grams = [c.gram(g).decode() for g in np.argsort(-sizes)[:80] if b" " not in c.gram(g)]
    literals = {"status": ["1", "0", "-2", "9", "12", "3.0", "abc", "", "true", "+1", "2147483648"],
                "category": ["cat03", "cat19", "cat20", "", "cat", "cat1", "zzz", "CAT03"],
                "score": ["50", "50.0", "12.25", "99.75", "-1", "1e2", "abc", "0.0000000001", "nan"],
                "small": ["0", "59999", "30000", "70000", "-1", "3e4", "100"],
                "flag": ["1", "0", "true", "false", "yes", "2"],
                "when": ["0", "-4999", "4000", "1.5", "x"],
                "big": ["0", "18446744073709551615", "9223372036854775808", "-1"],
                "tiny": ["5", "-128", "127", "128", "300"],
                "missing": ["1", "x"]}
    ops = ["=", "!=", ">", ">=", "<", "<="]
    checked = nonempty = 0
    for i in range(120):
        terms = [grams[int(rng.integers(0, len(grams)))] for _ in range(int(rng.integers(1, 3)))]
        k = int(rng.integers(1, 4))
        conds = []
        for _ in range(k):
            col = list(literals)[int(rng.integers(0, len(literals)))]
            op = ops[int(rng.integers(0, 2 if i % 2 == 0 else 6))]  # even queries: EQ / NE only (the bitmap path)
            conds.append((col, op, literals[col][int(rng.integers(0, len(literals[col])))]))
        base = p.oracle_query(Query(terms, limit=0, descending=False))[1].tolist()
        want = F.apply_filters_with_bitmap(base, conds, columns)
        total, docs, _ = t.search(terms, conditions=conds, descending=False, limit=50)
        assert total == len(want), (terms, conds, total, len(want))
        assert docs.tolist() == want[:50], (terms, conds)
        checked += 1
        nonempty += len(want) > 0
        if i % 4 == 0:  # FACET of the same result set
            fcol = ("category", "status", "flag", "tiny")[i // 4 % 4]
            matched, n_values, page = t.facet(fcol, terms=terms, conditions=conds, limit=300)
            exp = {F.display_string((k0[0], k0[1])): v for k0, v in F.facet_counts(want, columns[fcol]).items()}
            assert matched == len(want) and n_values == len(exp), (terms, conds, fcol)
            assert dict(page) == exp, (terms, conds, fcol)
            assert all(page[j][1] >= page[j + 1][1] for j in range(len(page) - 1))
    assert checked == 120 and nonempty > 40
    # FACET without a search: every document (GetAllDocIds), narrowed by a NOT term and a condition
    matched, n_values, page = t.facet("category", not_terms=[grams[0]], conditions=[("flag", "=", "1")], limit=100)
    base = set(range(1, n + 1)) - set(p.oracle_query(Query([grams[0]], limit=0))[1].tolist())
    want = F.apply_filters_with_bitmap(sorted(base), [("flag", "=", "1")], columns)
    exp = {F.display_string(k0): v for k0, v in F.facet_counts(want, columns["category"]).items()}
    assert matched == len(want) and dict(page) == exp


def test_a_table_loaded_from_a_dump_v2_file_ranks_filters_and_facets():
    """Index::FromDump / engine.Index.from_dump on tests/golden/dump_v2_small.bin (a DUMP SAVE file written from the
    reference's writer code: 37 live docs of ids 1..40, bigram index, status / category / score / flag columns): SORT _score
    equals an index built from the same texts, NOT keeps to the ids the store holds, FILTER and FACET read the dump's
    filter values."""
    from mygram_db_amd import _shim_capi as S
    raw = open(os.path.join(HERE, "golden", "dump_v2_small.bin"), "rb").read()
    exp = json.load(open(os.path.join(HERE, "golden", "dump_v2_expected.json"), encoding="utf-8"))
    texts = [exp["texts"].get(str(i + 1), "") for i in range(40)]
    loaded = mg.Index.from_dump(raw, "app_db.articles")
    direct = mg.Index(texts=texts, first_doc_id=1, ngram_size=2, kanji_ngram_size=2)
    assert loaded.table_name == "app_db.articles" and loaded.total_docs == direct.total_docs
    qs = [Query(["alpha"], sort_score=True, limit=10), Query(["tokyo", "data"], sort_score=True, limit=5),
          Query(["東京"], sort_score=True, limit=10), Query(["search"], ["beta"], limit=20, descending=False)]
    for a, b in zip(loaded.search_batch(qs), direct.search_batch(qs)):
        assert a.total == b.total and a.docs.tolist() == b.docs.tolist() and np.array_equal(a.scores, b.scores)
    t = S.Table.from_dump(raw, "app_db.articles")
    live = set(exp["ids"])
    # a search + FILTER conditions on the dump's columns
    total, docs, _ = t.search(["alpha"], conditions=[("status", "=", "1")], descending=False, limit=40)
    want = [d for d in exp["ids"] if "alpha" in texts[d - 1] and d % 10 != 0 and d % 3 == 1]
    assert docs.tolist() == want and total == len(want)
    total, docs, _ = t.search(["a"], conditions=[("score", ">=", "5"), ("category", "!=", "tech")], descending=False, limit=40)
    want = [d for d in exp["ids"] if "a" in texts[d - 1] and d % 10 != 0 and d / 4.0 >= 5 and ["tech", "food", "music"][d % 3] != "tech"]
    assert docs.tolist() == want
    # FACET over every document of the store: deleted ids (7, 8, 23) are not documents; NULL values are not counted
    matched, n_values, page = t.facet("category")
    assert matched == len(live)
    cnt = {}
    for d in exp["ids"]:
        if d % 10 != 0:
            c = ["tech", "food", "music"][d % 3].encode()
            cnt[c] = cnt.get(c, 0) + 1
    assert dict(page) == cnt and n_values == 3
    # the loaded table changes like any other: a document leaves, one changes its text and keeps its dump
    # filter values, a deleted id comes back with values of its own; then the delta is folded back (texts and values come
    # from the device, where the loader put them)
    def check(tag):
        cur_ids = sorted(cur)
        total, docs, _ = t.search(["alpha"], conditions=[("status", "=", "1")], descending=False, limit=40)
        want = [d for d in cur_ids if "alpha" in cur[d] and status_of(d) == 1]
        assert docs.tolist() == want and total == len(want), (tag, docs.tolist(), want)
        matched, _, page = t.facet("category")
        cnt = {}
        for d in cur_ids:
            c = category_of(d)
            if c is not None:
                cnt[c.encode()] = cnt.get(c.encode(), 0) + 1
        assert matched == len(cur_ids) and dict(page) == cnt, tag
    cur = {d: texts[d - 1] for d in exp["ids"]}
    override = {}

    def status_of(d):
        return override[d][0] if d in override else (None if d % 10 == 0 else d % 3)

    def category_of(d):
        return override[d][1] if d in override else (None if d % 10 == 0 else ["tech", "food", "music"][d % 3])
    victim = next(d for d in exp["ids"] if "alpha" in texts[d - 1] and d % 10 != 0 and d % 3 == 1)
    t.remove_document(victim, cur[victim])
    del cur[victim]
    changed = next(d for d in exp["ids"] if d in cur and "alpha" not in texts[d - 1] and d % 10 != 0 and d % 3 == 1)
    t.update_document(changed, cur[changed], cur[changed] + " alpha")
    cur[changed] = cur[changed] + " alpha"
    t.add_document(7, "alpha returns", filters={"status": ("int32", 1), "category": ("string", "tech")})
    cur[7] = "alpha returns"
    override[7] = (1, "tech")
    check("changed")
    t.compact()
    assert t.mutation_stats()["delta_documents"] == 0
    check("compacted")
