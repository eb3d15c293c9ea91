"""CPU oracle of the FILTER / FACET part of the hot path — TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's CPU leg
may import it). A plain-Python restatement of the reference's two filter paths and of its facet aggregation, each function
citing the lines it follows (paths relative to the reference tree):

  ParseFilterValue            src/server/search_pipeline.cpp:953-993
  ApplyFilters                src/server/search_pipeline.cpp:1098-1194   per-document comparison (the fallback)
  BuildTypeUnionBitmap        src/server/search_pipeline.cpp:1021-1094   every type interpretation of the literal, OR-ed
  ApplyFiltersWithBitmap      src/server/search_pipeline.cpp:1196-1237   EQ: AND the union, NE: ANDNOT it; else fallback
  CompareValues / CompareDoubleValues   src/utils/comparison_utils.h:29-70
  GetColumnValueCountsFiltered          src/storage/filter_index.cpp:284-312  (FACET)

Pinned by tests/golden/filters.json (the reference's SearchPipelineFilterParityTest fixture and expectations,
tests/server/search_pipeline_test.cpp:725-910, and FacetHandlerTest, tests/server/facet_handler_test.cpp:203-258).
A stored value is None (NULL) or (type, value) with type in TYPES — the alternatives of storage::FilterValue
(src/storage/document_store.h:73-87)."""
import math
import re
import struct

TYPES = ("bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "time", "string", "double")
SIGNED = {"int8": 8, "int16": 16, "int32": 32, "int64": 64}
UNSIGNED = {"uint8": 8, "uint16": 16, "uint32": 32, "uint64": 64}
EPSILON = 1e-9  # mygram::constants::kFilterValueEpsilon, src/utils/constants.h:104

_INT = re.compile(rb"-?[0-9]+\Z")
_UINT = re.compile(rb"[0-9]+\Z")
# std::from_chars(double), chars_format::general: no leading '+' or whitespace; inf / nan spelled out
_DEC = re.compile(rb"-?([0-9]+\.?[0-9]*|\.[0-9]+)([eE][+-]?[0-9]+)?\Z")
_SPECIAL = re.compile(rb"-?(inf|infinity|nan)\Z", re.IGNORECASE)


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def parse_filter_value(value):
    """ParseFilterValue (:953-993): the literal under every numeric reading, each valid only if the whole string parses."""
    v = _b(value)
    out = {"bool_val": v in (b"1", b"true"), "double": None, "int64": None, "uint64": None}
    if _DEC.match(v):
        d = float(v)
        if not math.isinf(d):  # from_chars reports result_out_of_range instead of returning infinity
            out["double"] = d
    elif _SPECIAL.match(v):
        out["double"] = float(v)
    if _INT.match(v):
        i = int(v)
        if -(1 << 63) <= i < (1 << 63):
            out["int64"] = i
    if _UINT.match(v):
        u = int(v)
        if u < (1 << 64):
            out["uint64"] = u
    return out


def _cmp(a, b, op):
    """CompareValues (comparison_utils.h:29-45)."""
    return {"=": a == b, "!=": a != b, "<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b}[op]


def _cmp_double(a, b, op):
    """CompareDoubleValues (comparison_utils.h:56-70)."""
    if op == "=":
        return abs(a - b) < EPSILON
    if op == "!=":
        return abs(a - b) >= EPSILON
    return _cmp(a, b, op)


def doc_matches(stored, op, value, parsed=None):
    """One condition on one document: the std::visit of ApplyFilters (:1149-1187)."""
    parsed = parsed or parse_filter_value(value)
    if stored is None:
        return op == "!="  # NULL: only != matches (:1151-1157)
    t, v = stored
    if t == "string":
        return _cmp(_b(v), _b(value), op)
    if t == "bool":
        return _cmp(bool(v), parsed["bool_val"], op)
    if t == "double":
        return parsed["double"] is not None and _cmp_double(float(v), parsed["double"], op)
    if t == "time" or t in SIGNED:
        return parsed["int64"] is not None and _cmp(int(v), parsed["int64"], op)
    return parsed["uint64"] is not None and _cmp(int(v), parsed["uint64"], op)


def apply_filters(results, conditions, columns):
    """ApplyFilters (:1098-1194). conditions: [(column, op, literal)]; columns: {name: [stored value per doc id index]} with
    `lookup(doc) -> stored`; results keep their order."""
    out = []
    parsed = [parse_filter_value(c[2]) for c in conditions]
    for d in results:
        ok = True
        for (col, op, value), p in zip(conditions, parsed):
            stored = columns[col](d) if col in columns else None
            if not doc_matches(stored, op, value, p):
                ok = False
                break
        if ok:
            out.append(d)
    return out


def literal_interpretations(value):
    """BuildTypeUnionBitmap (:1021-1094): the (type, value) keys the literal may be stored under."""
    v = _b(value)
    keys = [("string", v)]
    if v in (b"1", b"true"):
        keys.append(("bool", True))
    elif v in (b"0", b"false"):
        keys.append(("bool", False))
    p = parse_filter_value(v)
    if p["int64"] is not None:
        i = p["int64"]
        keys.append(("int64", i))
        for t, bits in (("int8", 8), ("int16", 16), ("int32", 32)):
            if -(1 << (bits - 1)) <= i < (1 << (bits - 1)):
                keys.append((t, i))
        keys.append(("time", i))
    if p["uint64"] is not None:
        u = p["uint64"]
        keys.append(("uint64", u))
        for t, bits in (("uint8", 8), ("uint16", 16), ("uint32", 32)):
            if u < (1 << bits):
                keys.append((t, u))
    if p["double"] is not None:
        keys.append(("double", struct.pack("<d", p["double"])))  # (keys compare as serialized bytes)
    return keys


def _key(stored):
    t, v = stored
    if t == "double":
        return (t, struct.pack("<d", float(v)))
    if t == "string":
        return (t, _b(v))
    if t == "bool":
        return (t, bool(v))
    return (t, int(v))


def apply_filters_with_bitmap(results, conditions, columns):
    """ApplyFiltersWithBitmap (:1196-1237): all EQ / NE -> AND / ANDNOT of the literal's type union; otherwise the
    per-document path for the whole list."""
    if any(op not in ("=", "!=") for _, op, _ in conditions):
        return apply_filters(results, conditions, columns)
    out = sorted(set(results))  # (the Roaring round trip returns ascending unique ids)
    for col, op, value in conditions:
        keys = set(literal_interpretations(value))
        member = []
        for d in out:
            stored = columns[col](d) if col in columns else None
            hit = stored is not None and _key(stored) in keys
            member.append(hit)
        out = [d for d, h in zip(out, member) if (h if op == "=" else not h)]
    return out


def display_string(stored):
    """DeserializeToDisplayString (filter_index.cpp:314-407)."""
    t, v = stored
    if t == "bool":
        return b"true" if v else b"false"
    if t == "string":
        return _b(v)
    if t == "double":
        return _b(repr(float(v))) if not float(v).is_integer() or abs(float(v)) >= 1e16 else _b(str(int(float(v))))
    return _b(str(int(v)))


def facet_counts(results, column_lookup):
    """GetColumnValueCountsFiltered (filter_index.cpp:284-312): {value key: count} over the result docs; NULLs have no
    bitmap and are not counted. The reference orders by count descending (ties in hash order: unspecified)."""
    counts = {}
    for d in results:
        stored = column_lookup(d)
        if stored is None:
            continue
        k = _key(stored)
        counts[k] = counts.get(k, 0) + 1
    return counts
