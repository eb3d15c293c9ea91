"""Expands the compact doc/expectation rules used by tests/golden/*.json (fixtures are data only)."""
import json
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN_DIR, name), encoding="utf-8") as f:
        return json.load(f)


def expand_docs(docs):
    """-> list of (doc_id, text) in insertion order."""
    out = []
    for d in docs:
        if isinstance(d, list):
            out.append((int(d[0]), d[1]))
            continue
        lo, hi = d["range"]
        excl = set(d.get("exclude", ()))
        for i in range(lo, hi + 1):
            if i in excl:
                continue
            if "text" in d:
                out.append((i, d["text"]))
                continue
            for c in d["cases"]:
                if "mod" not in c or i % c["mod"] == c["eq"]:
                    out.append((i, c["text"]))
                    break
    return out


def expand_ids(spec):
    """Expected / candidate / all_docs id lists."""
    if isinstance(spec, list):
        return [int(x) for x in spec]
    if "range" in spec:
        a, b, s = spec["range"]
        return list(range(a, b + (1 if s > 0 else -1), s))
    if "filter_range" in spec or "filter_range_step" in spec:
        if "filter_range" in spec:
            a, b = spec["filter_range"]
            ids = range(a, b + 1)
        else:
            a, b, s = spec["filter_range_step"]
            ids = range(a, b + 1, s)
        if "any_mod_zero" in spec:
            return [i for i in ids if any(i % m == 0 for m in spec["any_mod_zero"])]
        if "all_mod_nonzero" in spec:
            return [i for i in ids if all(i % m != 0 for m in spec["all_mod_nonzero"])]
    raise ValueError("bad id spec %r" % (spec,))


def index_cases():
    """All (case, call) pairs over Index::Search* known answers."""
    out = []
    for name in ("index_search.json", "threshold.json", "gettopn.json"):
        for case in load(name)["cases"]:
            out.append(case)
    return out
