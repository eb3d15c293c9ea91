"""Writes tests/golden/dump_v2_small.bin + dump_v2_expected.json: a table dump in the reference's "MGDB" version-2 format,
made HERE from the format as the reference's writer defines it — WriteDumpV2 (src/storage/dump_format_v2.cpp:520-770:
fixed header, HeaderV2 :300-327, section envelopes dump_format.h:111-118, the table-data section :160-290) around
Index::SaveToStream (the MGIX stream of make_mgix.py) and DocumentStore::SerializeDocuments
(src/storage/document_store_persistence.cpp:59-175: "MGDS" v3). The reference cannot be built here, so no dump written BY
the reference exists: parity of the reader with a real DUMP SAVE is "unpinned" beyond this restatement.

    python tests/golden/make_dump_v2.py      # rewrites the fixture deterministically
"""
import json
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_mgix import mgix, posting_bytes  # noqa: E402


def wstr(s):
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    return struct.pack("<I", len(b)) + b


TYPE = {"null": 0, "bool": 1, "int8": 2, "uint8": 3, "int16": 4, "uint16": 5, "int32": 6, "uint32": 7, "int64": 8,
        "uint64": 9, "time": 10, "string": 11, "double": 12}
FMT = {1: "<B", 2: "<b", 3: "<B", 4: "<h", 5: "<H", 6: "<i", 7: "<I", 8: "<q", 9: "<Q", 10: "<q", 12: "<d"}


def filter_value(t, v):
    ti = TYPE[t]
    if ti == 0:
        return struct.pack("<B", 0)
    if ti == 11:
        return struct.pack("<B", 11) + wstr(v)
    return struct.pack("<B", ti) + struct.pack(FMT[ti], v)


def mgds(docs, next_doc_id):
    """docs: [(doc_id, pk, {column: (type, value)}, normalized text, original text)] in the store's (hash) order."""
    out = bytearray(b"MGDS" + struct.pack("<II", 3, next_doc_id) + wstr("") + struct.pack("<Q", len(docs)))
    for doc_id, pk, filters, text, orig in docs:
        out += struct.pack("<I", doc_id) + wstr(pk) + struct.pack("<I", len(filters))
        for name, (t, v) in filters.items():
            out += wstr(name) + filter_value(t, v)
        out += wstr(text) + wstr(orig)
    return bytes(out)


def section(stype, data):
    return struct.pack("<IIQ", stype, zlib.crc32(data) & 0xFFFFFFFF, len(data)) + data


def dump_v2(tables, gtid="uuid:1-5"):
    """tables: [(name, index stream, document stream)] -> file bytes."""
    sections = [section(1, b"\x00" * 12)]  # a config section (opaque to the hot path: skipped by type)
    for name, istream, dstream in tables:
        data = wstr(name) + struct.pack("<I", 0) + struct.pack("<Q", len(istream)) + istream + struct.pack("<Q", len(dstream)) + dstream
        sections.append(section(3, data))
    g = gtid.encode()
    header_size = 4 + 4 + 8 + 8 + 4 + 4 + 4 + len(g)
    body = b"".join(sections)
    total = 8 + header_size + len(body)
    head = b"MGDB" + struct.pack("<I", 2) + struct.pack("<IIQQII", header_size, 1, 1_700_000_000, total, 0, len(sections)) + wstr(g)
    blob = bytearray(head + body)
    crc = zlib.crc32(bytes(blob)) & 0xFFFFFFFF  # (the CRC field itself reads as zero)
    blob[32:36] = struct.pack("<I", crc)
    return bytes(blob)


def ngrams(text, n):
    cps = list(text)
    return {"".join(cps[i:i + n]) for i in range(len(cps) - n + 1)}


def main():
    rng = np.random.default_rng(77)
    words = ["alpha", "beta", "gamma", "delta", "tokyo", "kyoto", "osaka", "data", "base", "search", "東京", "京都", "検索"]
    docs, texts = [], {}
    ids = [i for i in range(1, 41) if i not in (7, 8, 23)]  # ids 7, 8, 23 were deleted: gaps in the store
    for d in ids:
        t = " ".join(words[int(k)] for k in rng.integers(0, len(words), size=int(rng.integers(1, 7))))
        texts[d] = t
        f = {"status": ("int32", int(d % 3)), "category": ("string", ["tech", "food", "music"][d % 3]),
             "score": ("double", float(d) / 4.0), "flag": ("bool", d % 2)}
        if d % 10 == 0:
            f = {"status": ("null", None)}  # a NULL value and absent columns
        docs.append((d, "pk%d" % d, f, t, t.upper()))
    order = rng.permutation(len(docs))  # the store writes in hash-map order
    docs = [docs[i] for i in order]
    # the index stream: bigram postings of the same texts (ASCII bigrams + CJK bigrams: kanji_ngram_size 2, cross off is
    # irrelevant for lists made here: grams are taken per whitespace-separated run like GenerateHybridNgrams over the text)
    post = {}
    for d in ids:
        for g in ngrams(texts[d], 2):
            post.setdefault(g, []).append(d)
    terms = []
    for i, (g, lst) in enumerate(sorted(post.items(), key=lambda kv: zlib.crc32(kv[0].encode()))):
        pb, _ = posting_bytes(np.asarray(sorted(lst), dtype=np.uint32), 1 if (i % 3 == 0 and len(lst) > 3) else 0)
        terms.append((g, pb))
    istream = mgix(4, 2, 2, True, True, "keep", True, terms)
    blob = dump_v2([("app_db.other", mgix(4, 2, 2, True, True, "keep", True, []), mgds([], 1)),
                    ("app_db.articles", istream, mgds(docs, 41))])
    open(os.path.join(HERE, "dump_v2_small.bin"), "wb").write(blob)
    json.dump({"table": "app_db.articles", "texts": {str(k): v for k, v in texts.items()}, "ids": ids,
               "filters": {str(d): {k: list(v) for k, v in f.items()} for d, _, f, _, _ in docs}, "bytes": len(blob)},
              open(os.path.join(HERE, "dump_v2_expected.json"), "w"), ensure_ascii=False, indent=1)
    print("dump_v2_small.bin", len(blob))


if __name__ == "__main__":
    main()
