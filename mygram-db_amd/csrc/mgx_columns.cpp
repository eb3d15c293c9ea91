// mgx_columns.cpp — host-side column builder: normalized texts -> (gram dictionary, CSR postings, tf, doc_len).
//
// Restates, as one multi-threaded bulk pass, what the reference does incrementally:
//   * n-gram generation: GenerateHybridNgrams, src/utils/string_utils.cpp:452-509 (window size chosen by the class of
//     the STARTING code point; IsCJKIdeograph :441-448 — kana is not CJK; cross_boundary=false drops mixed windows),
//     over code points decoded as Utf8ToCodepoints :199-218 does (invalid bytes skipped);
//   * Index::AddDocument, src/index/index.cpp:39-74: unique grams of a doc -> one posting each, docids ascending;
//   * BM25 ingest bookkeeping, src/mysql/binlog_event_processor.cpp:98-99: doc_len = CountCodePoints(text)
//     (string_utils.cpp:655-669), counted in N / total_len only when the text is non-empty;
//   * the tf column (build-owned): BM25Scorer::CountTermOccurrences (src/index/bm25_scorer.cpp:27-45) of the gram's
//     bytes in the doc text — non-overlapping, left-greedy. For valid UTF-8 every byte occurrence of a gram is one
//     of the doc's own windows, so tf is the greedy count over that gram's window positions.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mygram_gpu.h"
#include "mgx_host.hpp"
#include "mgx_text.hpp"

namespace mgx {

namespace {

// A gram key: up to 15 UTF-8 bytes packed big-endian into the top of a 128-bit integer, length in the low byte.
// Integer order == bytewise lexicographic order of the byte strings (shorter prefix first).
using Key = unsigned __int128;

inline Key MakeKey(const uint8_t* b, size_t n) {
  Key k = 0;
  for (size_t i = 0; i < n; ++i) k |= static_cast<Key>(b[i]) << (8 * (15 - i));
  return k | static_cast<Key>(n);
}

inline uint64_t HashKey(Key k) {
  uint64_t a = static_cast<uint64_t>(k >> 64), b = static_cast<uint64_t>(k);
  uint64_t h = (a ^ (b * 0x9E3779B97F4A7C15ull)) * 0xFF51AFD7ED558CCDull;
  return h ^ (h >> 32);
}

// Open-addressing set/map of keys (0 is never a valid key: every key has a non-zero length byte).
struct KeyTable {
  std::vector<Key> keys;
  std::vector<uint32_t> vals;
  size_t used = 0;
  explicit KeyTable(size_t cap = 1024) : keys(cap, 0), vals(cap, 0) {}
  void Grow() {
    KeyTable n(keys.size() * 2);
    for (size_t i = 0; i < keys.size(); ++i)
      if (keys[i]) n.Insert(keys[i], vals[i]);
    keys.swap(n.keys);
    vals.swap(n.vals);
  }
  void Insert(Key k, uint32_t v) {
    if ((used + 1) * 2 > keys.size()) Grow();
    size_t m = keys.size() - 1, i = HashKey(k) & m;
    while (keys[i] && keys[i] != k) i = (i + 1) & m;
    if (!keys[i]) {
      keys[i] = k;
      vals[i] = v;
      ++used;
    }
  }
  bool Find(Key k, uint32_t* v) const {
    size_t m = keys.size() - 1, i = HashKey(k) & m;
    while (keys[i] && keys[i] != k) i = (i + 1) & m;
    if (!keys[i]) return false;
    *v = vals[i];
    return true;
  }
};

using text::EncodeUtf8;
using text::IsCjkIdeograph;
using text::ParseUtf8;

struct DocScratch {
  std::vector<uint32_t> cps;
  std::vector<uint8_t> cplen;
  std::vector<Key> keys;       // window keys (0 = no window at this position)
  std::vector<uint8_t> wsize;  // window size in code points
  std::vector<uint64_t> sortbuf;
};

// Decodes the text and produces one key per window position. Returns false if a gram exceeds 15 bytes.
inline bool DocWindows(const uint8_t* text, size_t len, int ascii_n, int kanji_n, bool cross, DocScratch& s,
                       uint32_t* doc_len) {
  s.cplen.clear();
  text::Decode(text, len, &s.cps);
  const size_t n = s.cps.size();
  *doc_len = static_cast<uint32_t>(n);
  s.keys.assign(n, 0);
  s.wsize.assign(n, 0);
  if (ascii_n <= 0 || kanji_n <= 0) return true;
  uint8_t buf[64];
  for (size_t p = 0; p < n; ++p) {
    const int w = text::WindowAt(s.cps, p, ascii_n, kanji_n, cross);
    if (w == 0) continue;
    size_t nb = 0;
    for (int j = 0; j < w; ++j) {
      if (nb + 4 > sizeof(buf)) return false;
      nb += EncodeUtf8(s.cps[p + j], buf + nb);
    }
    if (nb > 15) return false;
    s.keys[p] = MakeKey(buf, nb);
    s.wsize[p] = static_cast<uint8_t>(w);
  }
  return true;
}

}  // namespace

struct Columns {
  mgx_build_params params;
  std::vector<uint8_t> key_bytes;
  std::vector<uint32_t> key_off;
  std::vector<Key> sorted_keys;
  std::vector<uint64_t> offsets;
  std::vector<uint32_t> docids;
  std::vector<uint8_t> tf;
  // postings whose count does not fit the byte column (tf >= 255): (posting index, true tf), ascending by index
  std::vector<uint64_t> tf_ovf_pos;
  std::vector<uint32_t> tf_ovf_val;
  std::vector<uint32_t> doc_len;
  uint32_t first_doc_id = 0;
  uint64_t n_docs = 0, bm25_doc_count = 0, bm25_total_len = 0;
};

int BuildColumns(const mgx_build_params& bp, const uint8_t* text_bytes, const uint64_t* text_off,
                 uint32_t first_doc_id, uint64_t n_docs, Columns** out, std::string* err) {
  const int ascii_n = bp.ngram_size;
  const int kanji_n = bp.kanji_ngram_size > 0 ? bp.kanji_ngram_size : bp.ngram_size;  // index.cpp:31
  const bool cross = bp.cross_boundary_ngrams != 0;
  unsigned hw = std::thread::hardware_concurrency();
  const unsigned n_threads = bp.n_threads > 0 ? static_cast<unsigned>(bp.n_threads) : (hw ? hw : 4);
  const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>((n_docs + 4095) / 4096, n_threads * 8ull));
  const uint64_t chunk_docs = (n_docs + n_chunks - 1) / n_chunks;

  auto cols = std::make_unique<Columns>();
  cols->params = bp;
  cols->first_doc_id = first_doc_id;
  cols->n_docs = n_docs;
  cols->doc_len.assign(n_docs, 0);

  std::atomic<uint64_t> next{0};
  std::atomic<int> failed{0};
  auto run_chunks = [&](auto&& fn) {
    next = 0;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n_threads; ++t) {
      th.emplace_back([&]() {
        DocScratch s;
        for (;;) {
          uint64_t c = next.fetch_add(1);
          if (c >= n_chunks || failed.load()) break;
          fn(c, s);
        }
      });
    }
    for (auto& t : th) t.join();
  };

  // ---- pass 1: dictionary of all grams, doc_len, BM25 stats ----------------------------------------------------
  std::vector<KeyTable> chunk_sets(n_chunks);
  std::vector<uint64_t> chunk_count(n_chunks, 0), chunk_total(n_chunks, 0);
  run_chunks([&](uint64_t c, DocScratch& s) {
    KeyTable& set = chunk_sets[c];
    const uint64_t d0 = c * chunk_docs, d1 = std::min(n_docs, d0 + chunk_docs);
    for (uint64_t d = d0; d < d1; ++d) {
      const uint8_t* t = text_bytes + text_off[d];
      const size_t len = static_cast<size_t>(text_off[d + 1] - text_off[d]);
      uint32_t dl = 0;
      if (!DocWindows(t, len, ascii_n, kanji_n, cross, s, &dl)) {
        failed = 1;
        return;
      }
      cols->doc_len[d] = dl;
      if (len > 0) {
        chunk_count[c]++;
        chunk_total[c] += dl;
      }
      for (Key k : s.keys)
        if (k) set.Insert(k, 0);
    }
  });
  if (failed.load()) {
    *err = "an n-gram longer than 15 UTF-8 bytes is not supported by the column builder";
    return MGX_ERR_NOT_IMPLEMENTED;
  }
  for (uint64_t c = 0; c < n_chunks; ++c) {
    cols->bm25_doc_count += chunk_count[c];
    cols->bm25_total_len += chunk_total[c];
  }
  {
    KeyTable all;
    for (auto& s : chunk_sets) {
      for (Key k : s.keys)
        if (k) all.Insert(k, 0);
      s = KeyTable(2);
    }
    cols->sorted_keys.reserve(all.used);
    for (Key k : all.keys)
      if (k) cols->sorted_keys.push_back(k);
    std::sort(cols->sorted_keys.begin(), cols->sorted_keys.end());
  }
  const uint64_t G = cols->sorted_keys.size();
  KeyTable dict(std::max<size_t>(1024, 1ull << (64 - __builtin_clzll(std::max<uint64_t>(G * 2, 2)))));
  cols->key_off.resize(G + 1);
  cols->key_off[0] = 0;
  for (uint64_t g = 0; g < G; ++g) {
    const Key k = cols->sorted_keys[g];
    dict.Insert(k, static_cast<uint32_t>(g));
    const size_t nb = static_cast<size_t>(k & 0xFF);
    for (size_t i = 0; i < nb; ++i) cols->key_bytes.push_back(static_cast<uint8_t>(k >> (8 * (15 - i))));
    cols->key_off[g + 1] = static_cast<uint32_t>(cols->key_bytes.size());
  }

  // ---- pass 2: per chunk, (gram id, tf) of every doc; per-chunk per-gram counts --------------------------------
  struct ChunkOut {
    std::vector<uint32_t> ids;      // concatenated per doc, ascending gram id
    std::vector<uint8_t> tfs;
    std::vector<uint32_t> big;      // true tf of the entries whose byte saturated (255), in entry order
    std::vector<std::pair<uint64_t, uint32_t>> ovf;  // (posting index, true tf) once the lists are laid out
    std::vector<uint32_t> per_doc;  // number of entries per doc
    std::vector<uint32_t> counts;   // per gram
  };
  std::vector<ChunkOut> outs(n_chunks);
  run_chunks([&](uint64_t c, DocScratch& s) {
    ChunkOut& o = outs[c];
    o.counts.assign(G, 0);
    const uint64_t d0 = c * chunk_docs, d1 = std::min(n_docs, d0 + chunk_docs);
    o.per_doc.reserve(d1 - d0);
    for (uint64_t d = d0; d < d1; ++d) {
      const uint8_t* t = text_bytes + text_off[d];
      const size_t len = static_cast<size_t>(text_off[d + 1] - text_off[d]);
      uint32_t dl = 0;
      DocWindows(t, len, ascii_n, kanji_n, cross, s, &dl);
      s.sortbuf.clear();
      for (size_t p = 0; p < s.keys.size(); ++p) {
        if (!s.keys[p]) continue;
        uint32_t id = 0;
        dict.Find(s.keys[p], &id);
        s.sortbuf.push_back((static_cast<uint64_t>(id) << 32) | p);
      }
      std::sort(s.sortbuf.begin(), s.sortbuf.end());
      uint32_t emitted = 0;
      for (size_t i = 0; i < s.sortbuf.size();) {
        const uint32_t id = static_cast<uint32_t>(s.sortbuf[i] >> 32);
        // greedy non-overlapping count over this gram's window positions (bm25_scorer.cpp:34-43)
        uint32_t tf = 0;
        uint64_t next_ok = 0;
        size_t j = i;
        for (; j < s.sortbuf.size() && static_cast<uint32_t>(s.sortbuf[j] >> 32) == id; ++j) {
          const uint64_t p = s.sortbuf[j] & 0xFFFFFFFFull;
          if (p >= next_ok) {
            ++tf;
            next_ok = p + s.wsize[p];
          }
        }
        o.ids.push_back(id);
        o.tfs.push_back(static_cast<uint8_t>(tf >= 255 ? 255 : tf));
        if (tf >= 255) o.big.push_back(tf);
        o.counts[id]++;
        ++emitted;
        i = j;
      }
      o.per_doc.push_back(emitted);
    }
  });

  // ---- CSR offsets, then fill (each chunk owns a disjoint range of every list) ---------------------------------
  cols->offsets.assign(G + 1, 0);
  for (uint64_t g = 0; g < G; ++g) {
    uint64_t tot = 0;
    for (uint64_t c = 0; c < n_chunks; ++c) {
      const uint32_t v = outs[c].counts[g];
      outs[c].counts[g] = static_cast<uint32_t>(tot);  // becomes this chunk's start within list g
      tot += v;
      if (tot > 0xFFFFFFFFull) {
        *err = "a posting list exceeds 2^32 entries";
        return MGX_ERR_OUT_OF_RANGE;
      }
    }
    cols->offsets[g + 1] = cols->offsets[g] + tot;
  }
  const uint64_t P = cols->offsets[G];
  cols->docids.assign(P + 4, 0xFFFFFFFFu);  // 4 ids of padding: device scatter loads are 16 B wide
  cols->tf.assign(P + 4, 0);
  run_chunks([&](uint64_t c, DocScratch&) {
    ChunkOut& o = outs[c];
    const uint64_t d0 = c * chunk_docs;
    size_t at = 0, big_at = 0;
    for (size_t di = 0; di < o.per_doc.size(); ++di) {
      const uint32_t doc = first_doc_id + static_cast<uint32_t>(d0 + di);
      for (uint32_t e = 0; e < o.per_doc[di]; ++e, ++at) {
        const uint32_t id = o.ids[at];
        const uint64_t pos = cols->offsets[id] + o.counts[id]++;
        cols->docids[pos] = doc;
        cols->tf[pos] = o.tfs[at];
        if (o.tfs[at] == 255) o.ovf.emplace_back(pos, o.big[big_at++]);
      }
    }
    ChunkOut().ids.swap(o.ids);
    ChunkOut().tfs.swap(o.tfs);
  });
  {
    std::vector<std::pair<uint64_t, uint32_t>> all;
    for (const ChunkOut& o : outs) all.insert(all.end(), o.ovf.begin(), o.ovf.end());
    std::sort(all.begin(), all.end());
    for (const auto& e : all) {
      cols->tf_ovf_pos.push_back(e.first);
      cols->tf_ovf_val.push_back(e.second);
    }
  }
  *out = cols.release();
  return MGX_OK;
}

void ColumnsView(const Columns* c, mgx_columns_view* v) {
  v->n_grams = c->sorted_keys.size();
  v->key_bytes = c->key_bytes.data();
  v->key_off = c->key_off.data();
  v->offsets = c->offsets.data();
  v->docids = c->docids.data();
  v->tf = c->tf.data();
  v->n_postings = c->offsets.empty() ? 0 : c->offsets.back();
  v->first_doc_id = c->first_doc_id;
  v->n_docs = c->n_docs;
  v->doc_len = c->doc_len.data();
  v->bm25_doc_count = c->bm25_doc_count;
  v->bm25_total_len = c->bm25_total_len;
  v->tf_overflow_pos = c->tf_ovf_pos.data();
  v->tf_overflow_val = c->tf_ovf_val.data();
  v->n_tf_overflow = c->tf_ovf_pos.size();
}

bool ColumnsLookup(const Columns* c, const uint8_t* gram, size_t len, uint32_t* id) {
  if (len == 0 || len > 15) return false;
  const Key k = MakeKey(gram, len);
  auto it = std::lower_bound(c->sorted_keys.begin(), c->sorted_keys.end(), k);
  if (it == c->sorted_keys.end() || *it != k) return false;
  *id = static_cast<uint32_t>(it - c->sorted_keys.begin());
  return true;
}

void DestroyColumns(Columns* c) { delete c; }

}  // namespace mgx
