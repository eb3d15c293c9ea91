"""Writes tests/golden/mgix_*.bin + mgix_expected.json: index dumps in the reference's MGIX format, made HERE from the
format as the reference's own writer defines it — Index::SaveToStream (src/index/index_serialization.cpp:113-194) and
PostingList::Serialize (src/index/posting_list.cpp:973-1023) — with the Roaring containers laid out per CRoaring's
published portable format (the library itself is not in the reference tree: an un-vendored dependency, so its bytes
are restated from the format specification: cookies 12346 / 12347, {key, cardinality-1} pairs, the offset header,
array / bitset / run containers). The reference cannot be built here (CRoaring, abseil, spdlog absent), so no dump
written BY the reference exists to test against: parity of the reader with a real dump is "unpinned" beyond this
restatement; what the fixtures pin is the reader against every container kind, both cookies, the CRC and every
format version.

    python tests/golden/make_mgix.py      # rewrites the fixtures deterministically
"""
import json
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def roaring_portable(docids, use_runs):
    """Sorted unique u32 -> Roaring portable bytes. use_runs: run containers where a chunk is made of long runs."""
    docids = np.asarray(docids, dtype=np.uint32)
    keys = (docids >> 16).astype(np.uint32)
    chunks = [(int(k), (docids[keys == k] & 0xFFFF).astype(np.uint32)) for k in np.unique(keys)]
    n = len(chunks)
    kinds, bodies = [], []
    for _, low in chunks:
        runs = []
        start = prev = int(low[0])
        for v in low[1:].tolist():
            if v != prev + 1:
                runs.append((start, prev - start))
                start = v
            prev = v
        runs.append((start, prev - start))
        if use_runs and 2 + 4 * len(runs) < min(2 * len(low), 8192):
            kinds.append("run")
            bodies.append(struct.pack("<H", len(runs)) + b"".join(struct.pack("<HH", s, l) for s, l in runs))
        elif len(low) <= 4096:
            kinds.append("array")
            bodies.append(low.astype("<u2").tobytes())
        else:
            kinds.append("bitset")
            words = np.zeros(1024, dtype=np.uint64)
            np.bitwise_or.at(words, low >> 6, np.uint64(1) << (low & 63).astype(np.uint64))
            bodies.append(words.astype("<u8").tobytes())
    has_runs = "run" in kinds
    out = bytearray()
    if has_runs:
        out += struct.pack("<I", 12347 | ((n - 1) << 16))
        flags = bytearray((n + 7) // 8)
        for i, k in enumerate(kinds):
            if k == "run":
                flags[i // 8] |= 1 << (i % 8)
        out += flags
    else:
        out += struct.pack("<II", 12346, n)
    for (k, low) in chunks:
        out += struct.pack("<HH", k, len(low) - 1)
    if not has_runs or n >= 4:
        at = len(out) + 4 * n
        for b in bodies:
            out += struct.pack("<I", at)
            at += len(b)
    for b in bodies:
        out += b
    return bytes(out), kinds


def posting_bytes(docids, strategy, use_runs=False):
    if strategy == 0:  # kFixedWidthDelta
        d = np.asarray(docids, dtype=np.uint32)
        enc = np.concatenate([d[:1], np.diff(d)]).astype("<u4") if len(d) else np.zeros(0, "<u4")
        return struct.pack("<BI", 0, len(enc)) + enc.tobytes(), ["delta"]
    body, kinds = roaring_portable(docids, use_runs)
    return struct.pack("<BI", 1, len(body)) + body, kinds


def mgix(version, ngram, kanji, cross, nfkc, width, lower, terms):
    out = bytearray(b"MGIX" + struct.pack("<II", version, ngram))
    if version >= 3:
        out += struct.pack("<IB", kanji, int(cross))
    if version >= 4:
        out += struct.pack("<BI", int(nfkc), len(width)) + width.encode() + struct.pack("<B", int(lower))
    out += struct.pack("<Q", len(terms))
    for term, pbytes in terms:
        t = term.encode("utf-8")
        out += struct.pack("<I", len(t)) + t + struct.pack("<Q", len(pbytes)) + pbytes
    if version >= 2:
        out += struct.pack("<I", zlib.crc32(bytes(out)) & 0xFFFFFFFF)
    return bytes(out)


def main():
    rng = np.random.default_rng(2026)
    n_docs = 300_000  # doc ids 1..300000: five 65536-id chunks
    lists = {
        "ab": np.sort(rng.choice(np.arange(1, n_docs + 1), size=180_000, replace=False)),        # bitset containers
        "bc": np.sort(rng.choice(np.arange(1, n_docs + 1), size=9_000, replace=False)),          # array containers
        "cd": np.concatenate([np.arange(10, 40_000), np.arange(70_000, 140_000), np.arange(200_001, 200_010)]),  # runs
        "de": np.asarray([5, 6, 7, 65_535, 65_536, 65_537, 299_999, 300_000]),                  # 3 tiny arrays, no offsets w/ runs
        "東京": np.sort(rng.choice(np.arange(1, n_docs + 1), size=700, replace=False)),            # delta list
        "zz": np.asarray([123_456]),                                                             # one posting
        "qr": np.sort(rng.choice(np.arange(100_000, 200_000), size=30_000, replace=False)),       # mixed, inside 2 chunks
        "ef": np.concatenate([np.arange(100, 50_001), np.arange(70_000, 90_001)]),               # 2 run containers: no offset header
    }
    strategies = {"ab": (1, False), "bc": (1, False), "cd": (1, True), "de": (1, True), "東京": (0, False), "zz": (0, False),
                  "qr": (1, True), "ef": (1, True)}
    expected = {"n_docs": n_docs, "first_doc_id": 1, "terms": {}, "files": {}}
    terms = []
    for term in ["qr", "zz", "ab", "東京", "cd", "bc", "de", "ef"]:  # file order is the writer's map order: not sorted
        ids = np.asarray(lists[term], dtype=np.uint32)
        st, runs = strategies[term]
        pb, kinds = posting_bytes(ids, st, runs)
        terms.append((term, pb))
        expected["terms"][term] = {"count": int(len(ids)), "first": int(ids[0]), "last": int(ids[-1]),
                                   "sum": int(ids.astype(np.uint64).sum()), "xor": int(np.bitwise_xor.reduce(ids)),
                                   "containers": kinds}
    configs = {
        "mgix_v4.bin": (4, 2, 1, True, True, "keep", True),
        "mgix_v3.bin": (3, 2, 2, False, True, "keep", True),
        "mgix_v2.bin": (2, 2, 0, True, True, "keep", True),
        "mgix_v1.bin": (1, 3, 0, True, True, "keep", True),
    }
    for name, (ver, ng, kj, cross, nfkc, width, lower) in configs.items():
        sub = terms if ver == 4 else terms[:3] + terms[3:5]  # the older versions carry fewer terms (smaller files)
        blob = mgix(ver, ng, kj, cross, nfkc, width, lower, sub)
        open(os.path.join(HERE, name), "wb").write(blob)
        expected["files"][name] = {"version": ver, "ngram_size": ng, "kanji_ngram_size": kj if ver >= 3 else 0,
                                   "cross_boundary": cross if ver >= 3 else True, "normalize_width": width,
                                   "terms": [t for t, _ in sub], "bytes": len(blob)}
    json.dump(expected, open(os.path.join(HERE, "mgix_expected.json"), "w"), ensure_ascii=False, indent=1)
    for name in configs:
        print(name, os.path.getsize(os.path.join(HERE, name)))


if __name__ == "__main__":
    main()
