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
#include <cstdlib>
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

// The build for LARGE dictionaries: no per-gram state while the docs are read. Every (gram key, doc, tf) posting becomes a
// 24-byte record; chunks of docs are turned into records in parallel and each chunk sorts its own by key; the key space
// is cut at splitters (quantiles of the first chunk's keys) into ranges, and every range is assembled on its own — the
// chunks' pieces concatenated in doc order, stably sorted by key — so gram ids (ranks of the keys), posting offsets and
// the posting arrays of a range are written without looking at any other. Memory: ~2 x 24 B per posting at the peak.
static int BuildColumnsSorted(const mgx_build_params& bp, const uint8_t* text_bytes, const uint64_t* text_off,
                              uint32_t first_doc_id, uint64_t n_docs, unsigned n_threads, Columns** out, std::string* err) {
  const int ascii_n = bp.ngram_size;
  const int kanji_n = bp.kanji_ngram_size > 0 ? bp.kanji_ngram_size : bp.ngram_size;
  const bool cross = bp.cross_boundary_ngrams != 0;
  struct Rec {
    Key key;
    uint32_t doc;
    uint32_t tf;
  };
  auto cols = std::make_unique<Columns>();
  cols->params = bp;
  cols->first_doc_id = first_doc_id;
  cols->n_docs = n_docs;
  cols->doc_len.assign(n_docs, 0);
  const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>((n_docs + 16383) / 16384, n_threads * 16ull));
  const uint64_t chunk_docs = (n_docs + n_chunks - 1) / n_chunks;
  std::vector<std::vector<Rec>> runs(n_chunks);
  std::vector<uint64_t> chunk_count(n_chunks, 0), chunk_total(n_chunks, 0);
  std::atomic<uint64_t> next{0};
  std::atomic<int> failed{0};
  auto parallel = [&](uint64_t n_jobs, auto&& fn) {
    next = 0;
    const unsigned workers = static_cast<unsigned>(std::min<uint64_t>(n_threads, n_jobs));
    auto body = [&]() {
      DocScratch s;
      for (;;) {
        const uint64_t c = next.fetch_add(1);
        if (c >= n_jobs || failed.load()) break;
        fn(c, s);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < workers; ++t) th.emplace_back(body);
    body();
    for (auto& t : th) t.join();
  };
  // ---- A: records of every chunk, sorted by key (docs ascend inside equal keys: the sort is stable) -------------------
  parallel(n_chunks, [&](uint64_t c, DocScratch& s) {
    std::vector<Rec>& run = runs[c];
    const uint64_t d0 = c * chunk_docs, d1 = std::min(n_docs, d0 + chunk_docs);
    std::vector<std::pair<Key, uint32_t>> wins;
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
      wins.clear();
      for (size_t p = 0; p < s.keys.size(); ++p)
        if (s.keys[p]) wins.emplace_back(s.keys[p], static_cast<uint32_t>(p));
      std::sort(wins.begin(), wins.end());
      for (size_t i = 0; i < wins.size();) {
        // greedy non-overlapping count over this gram's window positions (bm25_scorer.cpp:34-43)
        uint32_t tf = 0;
        uint64_t next_ok = 0;
        size_t j = i;
        for (; j < wins.size() && wins[j].first == wins[i].first; ++j)
          if (wins[j].second >= next_ok) {
            ++tf;
            next_ok = static_cast<uint64_t>(wins[j].second) + s.wsize[wins[j].second];
          }
        run.push_back(Rec{wins[i].first, first_doc_id + static_cast<uint32_t>(d), tf});
        i = j;
      }
    }
    std::stable_sort(run.begin(), run.end(), [](const Rec& a, const Rec& b) { return a.key < b.key; });
  });
  if (failed.load()) {
    *err = "an n-gram longer than 15 UTF-8 bytes is not supported by the column builder";
    return MGX_ERR_NOT_IMPLEMENTED;
  }
  for (uint64_t c = 0; c < n_chunks; ++c) {
    cols->bm25_doc_count += chunk_count[c];
    cols->bm25_total_len += chunk_total[c];
  }
  // ---- B: key ranges from the quantiles of a sample of every chunk's keys -----------------------------------------------
  const uint64_t n_ranges = std::max<uint64_t>(1, n_threads * 16ull);
  std::vector<Key> sample;
  for (const auto& run : runs)
    for (size_t i = 0; i < run.size(); i += 1024) sample.push_back(run[i].key);
  std::sort(sample.begin(), sample.end());
  std::vector<Key> split;  // range r = keys in [split[r-1], split[r])
  for (uint64_t r = 1; r < n_ranges && !sample.empty(); ++r) {
    const Key k = sample[sample.size() * r / n_ranges];
    if (split.empty() || k > split.back()) split.push_back(k);
  }
  const uint64_t R = split.size() + 1;
  // where every chunk's run crosses the splitters
  std::vector<std::vector<size_t>> cut(n_chunks, std::vector<size_t>(R + 1, 0));
  parallel(n_chunks, [&](uint64_t c, DocScratch&) {
    const auto& run = runs[c];
    for (uint64_t r = 1; r < R; ++r)
      cut[c][r] = static_cast<size_t>(std::lower_bound(run.begin(), run.end(), split[r - 1],
                                                        [](const Rec& a, const Key& k) { return a.key < k; }) - run.begin());
    cut[c][R] = run.size();
  });
  // ---- C: every range on its own: gather in chunk (= doc) order, stable sort by key, count keys and postings -------------
  std::vector<std::vector<Rec>> ranges(R);
  std::vector<uint64_t> range_keys(R, 0);
  parallel(R, [&](uint64_t r, DocScratch&) {
    std::vector<Rec>& v = ranges[r];
    size_t total = 0;
    for (uint64_t c = 0; c < n_chunks; ++c) total += cut[c][r + 1] - cut[c][r];
    v.reserve(total);
    for (uint64_t c = 0; c < n_chunks; ++c) v.insert(v.end(), runs[c].begin() + cut[c][r], runs[c].begin() + cut[c][r + 1]);
    std::stable_sort(v.begin(), v.end(), [](const Rec& a, const Rec& b) { return a.key < b.key; });
    uint64_t nk = 0;
    for (size_t i = 0; i < v.size(); ++i) nk += (i == 0 || v[i].key != v[i - 1].key) ? 1 : 0;
    range_keys[r] = nk;
  });
  std::vector<std::vector<Rec>>().swap(runs);
  std::vector<uint64_t> gbase(R + 1, 0), pbase(R + 1, 0);
  for (uint64_t r = 0; r < R; ++r) {
    gbase[r + 1] = gbase[r] + range_keys[r];
    pbase[r + 1] = pbase[r] + ranges[r].size();
  }
  const uint64_t G = gbase[R], P = pbase[R];
  if (G > 0xFFFFFFF0ull) {
    *err = "more than 2^32 distinct n-grams";
    return MGX_ERR_OUT_OF_RANGE;
  }
  cols->sorted_keys.resize(G);
  cols->offsets.assign(G + 1, 0);
  cols->docids.assign(P + 4, 0xFFFFFFFFu);
  cols->tf.assign(P + 4, 0);
  std::vector<std::vector<std::pair<uint64_t, uint32_t>>> ovf(R);
  parallel(R, [&](uint64_t r, DocScratch&) {
    std::vector<Rec>& v = ranges[r];
    uint64_t g = gbase[r], pos = pbase[r];
    for (size_t i = 0; i < v.size(); ++i, ++pos) {
      if (i == 0 || v[i].key != v[i - 1].key) {
        cols->sorted_keys[g] = v[i].key;
        cols->offsets[g] = pos;
        ++g;
      }
      cols->docids[pos] = v[i].doc;
      cols->tf[pos] = static_cast<uint8_t>(v[i].tf >= 255 ? 255 : v[i].tf);
      if (v[i].tf >= 255) ovf[r].emplace_back(pos, v[i].tf);
    }
    std::vector<Rec>().swap(v);
  });
  cols->offsets[G] = P;
  for (uint64_t g = 0; g < G; ++g)
    if (cols->offsets[g + 1] - cols->offsets[g] > 0xFFFFFFFFull) {
      *err = "a posting list exceeds 2^32 entries";
      return MGX_ERR_OUT_OF_RANGE;
    }
  for (const auto& o : ovf)
    for (const auto& e : o) {
      cols->tf_ovf_pos.push_back(e.first);
      cols->tf_ovf_val.push_back(e.second);
    }
  cols->key_off.resize(G + 1);
  cols->key_off[0] = 0;
  uint64_t nbytes = 0;
  for (uint64_t g = 0; g < G; ++g) nbytes += static_cast<uint64_t>(cols->sorted_keys[g] & 0xFF);
  if (nbytes > 0xFFFFFFF0ull) {
    *err = "the n-gram keys exceed 4 GiB";
    return MGX_ERR_OUT_OF_RANGE;
  }
  cols->key_bytes.resize(nbytes);
  uint64_t at = 0;
  for (uint64_t g = 0; g < G; ++g) {
    const Key k = cols->sorted_keys[g];
    const size_t nb = static_cast<size_t>(k & 0xFF);
    for (size_t i = 0; i < nb; ++i) cols->key_bytes[at + i] = static_cast<uint8_t>(k >> (8 * (15 - i)));
    at += nb;
    cols->key_off[g + 1] = static_cast<uint32_t>(at);
  }
  *out = cols.release();
  return MGX_OK;
}

int BuildColumns(const mgx_build_params& bp, const uint8_t* text_bytes, const uint64_t* text_off,
                 uint32_t first_doc_id, uint64_t n_docs, Columns** out, std::string* err) {
  const int ascii_n = bp.ngram_size;
  const int kanji_n = bp.kanji_ngram_size > 0 ? bp.kanji_ngram_size : bp.ngram_size;  // index.cpp:31
  const bool cross = bp.cross_boundary_ngrams != 0;
  unsigned hw = std::thread::hardware_concurrency();
  const unsigned n_threads = bp.n_threads > 0 ? static_cast<unsigned>(bp.n_threads) : (hw ? hw : 4);
  // (512 documents per chunk at least: a delta index of a few thousand documents still spreads over the cores)
  const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>((n_docs + 511) / 512, n_threads * 8ull));
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
    // (no more threads than chunks: a delta index of a few thousand documents is one or two chunks, and spawning one
    //  thread per hardware thread of a 256-way host for each of the three passes was 75 of its 80 ms; the caller works too)
    const unsigned workers = static_cast<unsigned>(std::min<uint64_t>(n_threads, n_chunks));
    auto body = [&]() {
      DocScratch s;
      for (;;) {
        uint64_t c = next.fetch_add(1);
        if (c >= n_chunks || failed.load()) break;
        fn(c, s);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < workers; ++t) th.emplace_back(body);
    body();
    for (auto& t : th) t.join();
  };

  // ---- which build: a small dictionary (a bigram index of an alphabetic corpus: hundreds of grams) or a large one
  // (CJK trigrams: 10^7..10^8 distinct grams)? The dictionary-first build below keeps a count per (chunk, gram): fine
  // for the former, 50 GB for the latter. The first chunk's distinct grams decide.
  {
    DocScratch s;
    KeyTable probe;
    // (a fixed sample, not "the first chunk": with 512-document chunks a CJK trigram corpus showed 15k distinct grams in its
    //  first chunk, was taken for a small dictionary, and the dictionary-first build never came back)
    const uint64_t d1 = std::min<uint64_t>(n_docs, std::max<uint64_t>(chunk_docs, 4096));
    bool big = false;
    for (uint64_t d = 0; d < d1 && !big; ++d) {
      uint32_t dl = 0;
      if (!DocWindows(text_bytes + text_off[d], static_cast<size_t>(text_off[d + 1] - text_off[d]), ascii_n, kanji_n, cross, s, &dl)) break;
      for (Key k : s.keys)
        if (k) probe.Insert(k, 0);
      big = probe.used > (1u << 16);
    }
    const char* force = std::getenv("MGX_BUILD_SORTED");  // 1: always the sort-based build, 0: never (tests)
    if (force ? atoi(force) != 0 : big)
      return BuildColumnsSorted(bp, text_bytes, text_off, first_doc_id, n_docs, n_threads, out, err);
  }

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
  v->tf = c->tf.empty() ? nullptr : c->tf.data();
  v->n_postings = c->offsets.empty() ? 0 : c->offsets.back();
  v->first_doc_id = c->first_doc_id;
  v->n_docs = c->n_docs;
  v->doc_len = c->doc_len.empty() ? nullptr : c->doc_len.data();
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

// ---------------------------------------------------------------------------------------------------------------
// MGIX: the reference's index dump (Index::SaveToStream / LoadFromData, src/index/index_serialization.cpp:113-194 and
// :260-420) -> CSR columns. Little-endian throughout:
//   "MGIX" | u32 version (1..4) | u32 ngram_size | [v3+: u32 kanji_ngram_size, u8 cross_boundary] |
//   [v4: u8 normalize_nfkc, u32 width_len, width bytes, u8 normalize_lower] | u64 term_count |
//   term_count x { u32 term_len, term bytes, u64 posting_size, posting bytes } | [v2+: u32 CRC32 (zlib) of all before]
// posting bytes (PostingList::Serialize, src/index/posting_list.cpp:973-1023): u8 strategy, u32 size, then
//   strategy 0 (kFixedWidthDelta): `size` u32 values — the first doc id, then deltas;
//   strategy 1 (kRoaringBitmap)  : `size` bytes of Roaring's PORTABLE format (CRoaring's published RoaringFormatSpec:
//     cookie 12347 | (n-1) << 16 + run-flag bitset, or cookie 12346 + u32 n; n x {u16 key, u16 cardinality-1};
//     n x u32 offsets unless (run cookie and n < 4); containers: sorted u16 array (cardinality <= 4096), 1024 x u64
//     bitset, or u16 n_runs + n_runs x {u16 start, u16 length-1}).
// A dump holds doc ids only: tf and doc lengths live in the reference's DocumentStore, so the columns come back without
// them (BM25 then needs the texts: mgx_columns_build).
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Reader {
  const uint8_t* p;
  uint64_t n, at = 0;
  bool ok = true;
  bool Need(uint64_t k) {
    if (!ok || k > n - at) ok = false;
    return ok;
  }
  uint8_t U8() { return Need(1) ? p[at++] : 0; }
  uint16_t U16() {
    if (!Need(2)) return 0;
    const uint16_t v = static_cast<uint16_t>(p[at] | (p[at + 1] << 8));
    at += 2;
    return v;
  }
  uint32_t U32() {
    if (!Need(4)) return 0;
    const uint32_t v = static_cast<uint32_t>(p[at]) | (static_cast<uint32_t>(p[at + 1]) << 8) |
                       (static_cast<uint32_t>(p[at + 2]) << 16) | (static_cast<uint32_t>(p[at + 3]) << 24);
    at += 4;
    return v;
  }
  uint64_t U64() {
    const uint64_t lo = U32(), hi = U32();
    return lo | (hi << 32);
  }
};

uint32_t Crc32(const uint8_t* d, uint64_t n) {  // zlib's crc32 (IEEE 802.3, reflected), as src/utils/crc32.h uses it
  static uint32_t table[256];
  static const bool init = [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    return true;
  }();
  (void)init;
  uint32_t c = 0xFFFFFFFFu;
  for (uint64_t i = 0; i < n; ++i) c = table[(c ^ d[i]) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// Roaring portable bytes -> ascending doc ids (appended). false: malformed, or more ids than `budget` allows (a run
// container is 4 bytes for up to 65536 ids: an untrusted dump of under a megabyte could otherwise expand to gigabytes
// before any range check). *budget is decremented by what was decoded.
bool DecodeRoaring(const uint8_t* data, uint64_t len, std::vector<uint32_t>* out, uint64_t* budget, bool* over_budget) {
  *over_budget = false;
  Reader r{data, len};
  const uint32_t cookie = r.U32();
  uint32_t n = 0;
  bool has_runs = false;
  const uint8_t* run_flags = nullptr;
  if ((cookie & 0xFFFFu) == 12347u) {
    has_runs = true;
    n = (cookie >> 16) + 1;
    const uint32_t fb = (n + 7) / 8;
    if (!r.Need(fb)) return false;
    run_flags = data + r.at;
    r.at += fb;
  } else if (cookie == 12346u) {
    n = r.U32();
  } else {
    return false;
  }
  if (!r.ok || n > 65536) return false;
  std::vector<uint16_t> keys(n);
  std::vector<uint32_t> cards(n);
  for (uint32_t i = 0; i < n; ++i) {
    keys[i] = r.U16();
    cards[i] = static_cast<uint32_t>(r.U16()) + 1;
  }
  if (!has_runs || n >= 4) {  // offset header (not needed for a sequential read)
    if (!r.Need(4ull * n)) return false;
    r.at += 4ull * n;
  }
  for (uint32_t i = 0; i < n && r.ok; ++i) {
    const uint32_t hi = static_cast<uint32_t>(keys[i]) << 16;
    if (i && keys[i] <= keys[i - 1]) return false;
    const bool is_run = has_runs && ((run_flags[i / 8] >> (i % 8)) & 1u);
    if (is_run) {
      const uint32_t n_runs = r.U16();
      uint32_t next_free = 0;  // runs ascend and do not touch: a container yields at most 65536 ids whatever its bytes say
      uint32_t decoded = 0;
      for (uint32_t k = 0; k < n_runs && r.ok; ++k) {
        const uint32_t start = r.U16(), last = start + r.U16();
        if (last > 0xFFFFu || start < next_free) return false;
        const uint32_t cnt = last - start + 1;
        if (cnt > *budget) { *over_budget = true; return false; }
        *budget -= cnt;
        decoded += cnt;
        for (uint32_t v = start; v <= last; ++v) out->push_back(hi | v);
        next_free = last + 1;
      }
      if (r.ok && decoded != cards[i]) return false;  // the header's cardinality must be what the runs hold
    } else if (cards[i] <= 4096) {
      if (cards[i] > *budget) { *over_budget = true; return false; }
      *budget -= cards[i];
      uint32_t prev = 0;
      for (uint32_t k = 0; k < cards[i] && r.ok; ++k) {
        const uint32_t v = r.U16();
        if (k && v <= prev) return false;  // (sorted, distinct)
        out->push_back(hi | v);
        prev = v;
      }
    } else {
      if (!r.Need(8192)) return false;
      uint32_t pop = 0;
      for (uint32_t w = 0; w < 1024; ++w) {
        uint64_t bits = 0;
        for (int b = 0; b < 8; ++b) bits |= static_cast<uint64_t>(data[r.at + w * 8 + b]) << (8 * b);
        pop += static_cast<uint32_t>(__builtin_popcountll(bits));
      }
      if (pop != cards[i]) return false;  // (the stated cardinality is checked, not trusted)
      if (pop > *budget) { *over_budget = true; return false; }
      *budget -= pop;
      for (uint32_t w = 0; w < 1024; ++w) {
        uint64_t bits = 0;
        for (int b = 0; b < 8; ++b) bits |= static_cast<uint64_t>(data[r.at + w * 8 + b]) << (8 * b);
        while (bits) {
          out->push_back(hi | (w * 64 + static_cast<uint32_t>(__builtin_ctzll(bits))));
          bits &= bits - 1;
        }
      }
      r.at += 8192;
    }
  }
  return r.ok;
}

}  // namespace

int ColumnsFromMgix(const uint8_t* data, uint64_t len, uint32_t first_doc_id, uint64_t n_docs, Columns** out,
                    mgx_mgix_info* info, std::string* err) {
  auto fail = [&](int code, const char* what) {
    *err = std::string("mgx_columns_from_mgix: ") + what;
    return code;
  };
  Reader r{data, len};
  if (len < 20 || std::memcmp(data, "MGIX", 4) != 0) return fail(MGX_ERR_INVALID_ARGUMENT, "not an MGIX dump (bad magic)");
  r.at = 4;
  const uint32_t version = r.U32();
  if (version < 1 || version > 4) return fail(MGX_ERR_NOT_IMPLEMENTED, "unsupported format version");
  if (version >= 2) {  // CRC32 trailer over everything before it (index_serialization.cpp:71-112)
    if (len < 24) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated");
    const uint64_t body = len - 4;
    const uint32_t want = static_cast<uint32_t>(data[body]) | (static_cast<uint32_t>(data[body + 1]) << 8) |
                          (static_cast<uint32_t>(data[body + 2]) << 16) | (static_cast<uint32_t>(data[body + 3]) << 24);
    if (Crc32(data, body) != want) return fail(MGX_ERR_INVALID_ARGUMENT, "CRC32 mismatch");
    r.n = body;
  }
  mgx_mgix_info mi{};
  mi.version = version;
  mi.ngram_size = static_cast<int32_t>(r.U32());
  mi.kanji_ngram_size = 0;
  mi.cross_boundary_ngrams = 1;
  mi.normalize_nfkc = 1;
  mi.normalize_lower = 1;
  std::strcpy(mi.normalize_width, "keep");
  if (version >= 3) {
    mi.kanji_ngram_size = static_cast<int32_t>(r.U32());
    mi.cross_boundary_ngrams = r.U8() ? 1 : 0;
  }
  if (version >= 4) {
    mi.normalize_nfkc = r.U8() ? 1 : 0;
    const uint32_t wl = r.U32();
    if (wl >= sizeof(mi.normalize_width) || !r.Need(wl)) return fail(MGX_ERR_INVALID_ARGUMENT, "bad normalize_width");
    std::memset(mi.normalize_width, 0, sizeof(mi.normalize_width));
    std::memcpy(mi.normalize_width, data + r.at, wl);
    r.at += wl;
    mi.normalize_lower = r.U8() ? 1 : 0;
  }
  const uint64_t n_terms = r.U64();
  if (!r.ok || n_terms > (r.n - r.at) / 17) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated header");
  mi.n_terms = n_terms;
  struct Term {
    Key key;
    std::vector<uint32_t> docs;
  };
  std::vector<Term> terms;
  terms.reserve(n_terms);
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  // Decoded-size caps of an untrusted dump: a posting list holds distinct doc ids, so at most n_docs of them when the
  // caller gave the range (2^32 otherwise); all lists together at most MGX_MGIX_MAX_POSTINGS (default 2^33 ids = 32 GiB
  // of host memory — a table beyond that is loaded shard by shard).
  static const uint64_t kTotalCap = std::getenv("MGX_MGIX_MAX_POSTINGS") ? static_cast<uint64_t>(atoll(std::getenv("MGX_MGIX_MAX_POSTINGS"))) : (1ull << 33);
  uint64_t total_budget = kTotalCap;
  for (uint64_t t = 0; t < n_terms; ++t) {
    const uint32_t tl = r.U32();
    if (!r.Need(tl)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated term");
    if (tl == 0 || tl > 15) return fail(MGX_ERR_NOT_IMPLEMENTED, "a term longer than 15 bytes (n-gram keys are at most 15)");
    Term term;
    term.key = MakeKey(data + r.at, tl);
    r.at += tl;
    const uint64_t ps = r.U64();
    if (!r.Need(ps) || ps < 5) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated posting list");
    Reader pr{data + r.at, ps};
    r.at += ps;
    const uint8_t strategy = pr.U8();
    const uint32_t size = pr.U32();
    if (strategy == 0) {
      if (size > (pr.n - pr.at) / 4) return fail(MGX_ERR_INVALID_ARGUMENT, "delta list longer than its bytes");
      if (size > total_budget) return fail(MGX_ERR_INVALID_ARGUMENT, "the dump decodes to more doc ids than MGX_MGIX_MAX_POSTINGS allows");
      total_budget -= size;
      term.docs.reserve(size);
      uint32_t prev = 0;
      for (uint32_t i = 0; i < size; ++i) {  // PostingList::DecodeDelta, posting_list.cpp:954-971
        const uint32_t v = pr.U32();
        prev = i ? prev + v : v;
        if (i && v == 0) return fail(MGX_ERR_INVALID_ARGUMENT, "delta list is not strictly ascending");
        term.docs.push_back(prev);
      }
    } else if (strategy == 1) {
      if (size > pr.n - pr.at) return fail(MGX_ERR_INVALID_ARGUMENT, "roaring bitmap longer than its bytes");
      uint64_t budget = std::min<uint64_t>(n_docs ? n_docs : (1ull << 32), total_budget);
      const uint64_t before = budget;
      bool over = false;
      if (!DecodeRoaring(pr.p + pr.at, size, &term.docs, &budget, &over)) {
        if (over && n_docs && before == n_docs)  // more distinct ids in one list than the range has slots
          return fail(MGX_ERR_OUT_OF_RANGE, "a doc id lies outside the given range");
        return fail(MGX_ERR_INVALID_ARGUMENT, over ? "the dump decodes to more doc ids than MGX_MGIX_MAX_POSTINGS allows"
                                                   : "malformed roaring bitmap");
      }
      total_budget -= before - budget;
    } else {
      return fail(MGX_ERR_INVALID_ARGUMENT, "unknown posting strategy");
    }
    if (!term.docs.empty()) {
      lo = std::min(lo, term.docs.front());
      hi = std::max(hi, term.docs.back());
    }
    terms.push_back(std::move(term));
  }
  if (!r.ok) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated");
  std::sort(terms.begin(), terms.end(), [](const Term& a, const Term& b) { return a.key < b.key; });
  for (size_t i = 1; i < terms.size(); ++i)
    if (terms[i].key == terms[i - 1].key) return fail(MGX_ERR_INVALID_ARGUMENT, "a term appears twice");
  auto cols = std::make_unique<Columns>();
  cols->params = mgx_build_params{sizeof(mgx_build_params), MGX_ABI_VERSION, mi.ngram_size, mi.kanji_ngram_size,
                                  mi.cross_boundary_ngrams, 0};
  if (n_docs == 0) {  // the range the dump's doc ids span
    first_doc_id = lo == 0xFFFFFFFFu ? 1u : lo;
    n_docs = lo == 0xFFFFFFFFu ? 0 : static_cast<uint64_t>(hi) - lo + 1;
  } else if (lo != 0xFFFFFFFFu && (lo < first_doc_id || static_cast<uint64_t>(hi) - first_doc_id >= n_docs)) {
    return fail(MGX_ERR_OUT_OF_RANGE, "a doc id lies outside the given range");
  }
  cols->first_doc_id = first_doc_id;
  cols->n_docs = n_docs;
  cols->key_off.push_back(0);
  cols->offsets.push_back(0);
  for (const Term& t : terms) {
    cols->sorted_keys.push_back(t.key);
    const size_t nb = static_cast<size_t>(t.key & 0xFF);
    for (size_t i = 0; i < nb; ++i) cols->key_bytes.push_back(static_cast<uint8_t>(t.key >> (8 * (15 - i))));
    cols->key_off.push_back(static_cast<uint32_t>(cols->key_bytes.size()));
    cols->docids.insert(cols->docids.end(), t.docs.begin(), t.docs.end());
    cols->offsets.push_back(cols->docids.size());
  }
  if (info) *info = mi;
  *out = cols.release();
  return MGX_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Dump v2 ("MGDB" version 2: what DUMP SAVE writes, src/storage/dump_format_v2.cpp:520-770) -> columns + texts + filter values
// ---------------------------------------------------------------------------------------------------------------
// "MGDB" | u32 version = 2 | HeaderV2 (dump_format_v2.h:71-80, written field by field :300-327): u32 header_size, u32 flags,
// u64 dump_timestamp, u64 total_file_size, u32 file_crc32 (zlib CRC of the whole file with these four bytes zeroed,
// dump_format_internal.cpp:98-133; at offset 32), u32 section_count, string gtid | sections, each
// {u32 type, u32 crc32 of the data, u64 data_length, data} (dump_format.h:111-118). Strings are u32 length + bytes.
// A kTableData section (type 3; :160-290): string table_name | u32 stats_len + stats | u64 index_len + the MGIX stream |
// u64 docs_len + the document store stream "MGDS" (document_store_persistence.cpp:59-175): u32 version (1..3),
// u32 next_doc_id, string gtid, u64 doc_count, then per document u32 doc_id, string primary key, u32 filter_count x
// {string column, u8 FilterValue alternative, value: nothing for NULL, u32 len + bytes for a string, the raw little-endian
// scalar otherwise}, v2+: string normalized text, v3+: string original text.
struct DumpData {
  std::string table;
  mgx_mgix_info info{};
  Columns* cols = nullptr;  // owned until mgx_dump_columns hands it over
  uint32_t first_doc_id = 1;
  uint64_t n_docs = 0, n_existing = 0;
  std::vector<uint8_t> exists, text_bytes;
  std::vector<uint64_t> text_off;
  struct Col {
    std::string name;
    uint32_t type = 0;
    std::vector<uint64_t> values;
    std::vector<uint8_t> is_null, str_bytes;
    std::vector<uint64_t> str_off;
  };
  std::vector<Col> filter_cols;
  bool has_scores = false;  // the texts were stored: tf / doc_len columns exist
  ~DumpData() { DestroyColumns(cols); }
};

namespace {
bool ReadStr(Reader& r, uint64_t max_len, std::string* out) {
  const uint32_t n = r.U32();
  if (!r.ok || n > max_len || !r.Need(n)) return false;
  out->assign(reinterpret_cast<const char*>(r.p + r.at), n);
  r.at += n;
  return true;
}
}  // namespace

int DumpOpen(const uint8_t* data, uint64_t len, const char* table, DumpData** out, std::string* err) {
  auto fail = [&](int code, const std::string& what) {
    *err = "mgx_dump_open: " + what;
    return code;
  };
  if (len < 44 || std::memcmp(data, "MGDB", 4) != 0) return fail(MGX_ERR_INVALID_ARGUMENT, "not a MygramDB dump (bad magic)");
  Reader r{data, len};
  r.at = 4;
  const uint32_t version = r.U32();
  if (version != 2) return fail(MGX_ERR_NOT_IMPLEMENTED, "only dump format version 2 is read (version " + std::to_string(version) + ")");
  (void)r.U32();  // header_size
  (void)r.U32();  // flags
  (void)r.U64();  // timestamp
  const uint64_t total = r.U64();
  const uint32_t file_crc = r.U32();
  const uint32_t n_sections = r.U32();
  std::string gtid;
  if (!ReadStr(r, 64 * 1024, &gtid)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated header");
  if (total != len) return fail(MGX_ERR_INVALID_ARGUMENT, "the file is not the size its header states (truncated?)");
  {  // whole-file CRC with its own four bytes (offset 32) read as zero
    std::vector<uint8_t> head(data, data + 36);
    std::memset(head.data() + 32, 0, 4);
    uint32_t c = Crc32(head.data(), 36);
    // (continue the running CRC over the rest: Crc32() finalises, so un-finalise / re-run through the table)
    uint32_t run = c ^ 0xFFFFFFFFu;
    static uint32_t table_[256];
    static const bool init = [] {
      for (uint32_t i = 0; i < 256; ++i) {
        uint32_t v = i;
        for (int k = 0; k < 8; ++k) v = (v & 1u) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
        table_[i] = v;
      }
      return true;
    }();
    (void)init;
    for (uint64_t i = 36; i < len; ++i) run = table_[(run ^ data[i]) & 0xFFu] ^ (run >> 8);
    if ((run ^ 0xFFFFFFFFu) != file_crc) return fail(MGX_ERR_INVALID_ARGUMENT, "file CRC32 mismatch");
  }
  const uint8_t* tdata = nullptr;
  uint64_t tlen = 0;
  std::string tname;
  for (uint32_t sct = 0; sct < n_sections; ++sct) {
    const uint32_t type = r.U32(), crc = r.U32();
    const uint64_t dl = r.U64();
    if (!r.Need(dl)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated section");
    const uint8_t* sd = data + r.at;
    r.at += dl;
    if (type != 3) continue;  // config / statistics / metadata sections: not on this path
    if (Crc32(sd, dl) != crc) return fail(MGX_ERR_INVALID_ARGUMENT, "section CRC32 mismatch");
    Reader tr{sd, dl};
    std::string name;
    if (!ReadStr(tr, 4096, &name)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated table section");
    if (table != nullptr && table[0] != '\0' && name != table) continue;
    if (tdata != nullptr) continue;  // (the first table, or the named one)
    tname = name;
    tdata = sd + tr.at;
    tlen = dl - tr.at;
  }
  if (tdata == nullptr) return fail(MGX_ERR_INDEX_NOT_FOUND, table && table[0] ? std::string("table \"") + table + "\" is not in the dump" : "the dump holds no table");
  Reader tr{tdata, tlen};
  const uint32_t stats_len = tr.U32();
  if (!tr.Need(stats_len)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated table statistics");
  tr.at += stats_len;
  const uint64_t ilen = tr.U64();
  if (!tr.Need(ilen)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated index stream");
  const uint8_t* istream = tdata + tr.at;
  tr.at += ilen;
  const uint64_t dlen = tr.U64();
  if (!tr.Need(dlen)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated document stream");
  Reader dr{tdata + tr.at, dlen};
  auto dump = std::make_unique<DumpData>();
  dump->table = tname;
  // ---- the index stream: n-gram parameters + doc-id postings (cross-checked against the texts below) -------------------
  Columns* from_index = nullptr;
  {
    const int rc = ColumnsFromMgix(istream, ilen, 1, 0, &from_index, &dump->info, err);
    if (rc != MGX_OK) return rc;
  }
  std::unique_ptr<Columns, void (*)(Columns*)> index_cols(from_index, DestroyColumns);
  // ---- the document stream ------------------------------------------------------------------------------------------------
  if (!dr.Need(4) || std::memcmp(dr.p, "MGDS", 4) != 0) return fail(MGX_ERR_INVALID_ARGUMENT, "bad document store magic");
  dr.at = 4;
  const uint32_t dsv = dr.U32();
  if (dsv < 1 || dsv > 3) return fail(MGX_ERR_NOT_IMPLEMENTED, "unsupported document store version");
  (void)dr.U32();  // next_doc_id
  std::string g2;
  if (!ReadStr(dr, 64 * 1024, &g2)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated document store header");
  const uint64_t doc_count = dr.U64();
  if (!dr.ok || doc_count > (dr.n - dr.at) / 12) return fail(MGX_ERR_INVALID_ARGUMENT, "document count exceeds the stream");
  struct Doc {
    uint32_t id;
    uint64_t text_at, text_len;
    std::vector<std::pair<uint32_t, std::pair<uint8_t, std::pair<uint64_t, uint64_t>>>> filters;  // column -> (type, (value | str at, str len))
  };
  std::vector<Doc> docs;
  docs.reserve(doc_count);
  std::vector<std::string> col_names;
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  for (uint64_t d = 0; d < doc_count; ++d) {
    Doc doc{};
    doc.id = dr.U32();
    std::string pk;
    if (!ReadStr(dr, 1u << 20, &pk)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated document");
    const uint32_t nf = dr.U32();
    if (!dr.ok || nf > 4096) return fail(MGX_ERR_INVALID_ARGUMENT, "implausible filter count");
    for (uint32_t f = 0; f < nf; ++f) {
      std::string cname;
      if (!ReadStr(dr, 4096, &cname)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated filter name");
      const uint8_t type = dr.U8();
      uint64_t v = 0, sl = 0;
      switch (type) {
        case 0: break;                                            // std::monostate
        case 1: case 2: case 3: v = dr.U8(); if (type == 2) v = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int8_t>(v))); break;
        case 4: v = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int16_t>(dr.U16()))); break;
        case 5: v = dr.U16(); break;
        case 6: v = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(dr.U32()))); break;
        case 7: v = dr.U32(); break;
        case 8: case 9: case 10: case 12: v = dr.U64(); break;    // int64, uint64, TimeValue seconds, double bits
        case 11: {
          sl = dr.U32();
          if (!dr.Need(sl)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated filter string");
          v = dr.at;  // (offset of the bytes inside the document stream)
          dr.at += sl;
          break;
        }
        default: return fail(MGX_ERR_INVALID_ARGUMENT, "unknown filter value type");
      }
      if (!dr.ok) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated filter value");
      uint32_t ci = 0;
      while (ci < col_names.size() && col_names[ci] != cname) ++ci;
      if (ci == col_names.size()) col_names.push_back(cname);
      doc.filters.push_back({ci, {type, {v, sl}}});
    }
    if (dsv >= 2) {
      const uint32_t tl = dr.U32();
      if (!dr.Need(tl)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated text");
      doc.text_at = dr.at;
      doc.text_len = tl;
      dr.at += tl;
    }
    if (dsv >= 3) {
      const uint32_t ol = dr.U32();
      if (!dr.Need(ol)) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated original text");
      dr.at += ol;
    }
    if (!dr.ok) return fail(MGX_ERR_INVALID_ARGUMENT, "truncated document");
    lo = std::min(lo, doc.id);
    hi = std::max(hi, doc.id);
    docs.push_back(std::move(doc));
  }
  // the doc-id range: everything the store or the index mentions
  if (index_cols->n_docs) {
    lo = std::min<uint32_t>(lo, index_cols->first_doc_id);
    hi = std::max<uint32_t>(hi, static_cast<uint32_t>(index_cols->first_doc_id + index_cols->n_docs - 1));
  }
  if (lo == 0xFFFFFFFFu) {
    lo = 1;
    hi = 0;
  }
  dump->first_doc_id = lo;
  dump->n_docs = hi >= lo ? static_cast<uint64_t>(hi) - lo + 1 : 0;
  dump->n_existing = docs.size();
  dump->exists.assign(dump->n_docs, 0);
  std::vector<const Doc*> by_slot(dump->n_docs, nullptr);
  for (const Doc& d : docs) {
    if (by_slot[d.id - lo] != nullptr) return fail(MGX_ERR_INVALID_ARGUMENT, "a doc id appears twice in the document store");
    by_slot[d.id - lo] = &d;
    dump->exists[d.id - lo] = 1;
  }
  dump->text_off.assign(dump->n_docs + 1, 0);
  uint64_t tbytes = 0;
  for (uint64_t sl = 0; sl < dump->n_docs; ++sl) {
    dump->text_off[sl] = tbytes;
    if (by_slot[sl]) tbytes += by_slot[sl]->text_len;
  }
  dump->text_off[dump->n_docs] = tbytes;
  dump->text_bytes.assign(tbytes + 16, 0);
  for (uint64_t sl = 0; sl < dump->n_docs; ++sl)
    if (by_slot[sl] && by_slot[sl]->text_len)
      std::memcpy(dump->text_bytes.data() + dump->text_off[sl], dr.p + by_slot[sl]->text_at, by_slot[sl]->text_len);
  // filter columns by slot
  dump->filter_cols.resize(col_names.size());
  for (size_t c = 0; c < col_names.size(); ++c) {
    DumpData::Col& col = dump->filter_cols[c];
    col.name = col_names[c];
    col.values.assign(dump->n_docs, 0);
    col.is_null.assign(dump->n_docs, 1);
  }
  for (const Doc& d : docs)
    for (const auto& f : d.filters) {
      DumpData::Col& col = dump->filter_cols[f.first];
      const uint8_t type = f.second.first;
      if (type == 0) continue;
      if (col.type == 0) col.type = type;
      if (col.type != type)
        return fail(MGX_ERR_NOT_IMPLEMENTED, "filter column \"" + col.name + "\" holds values of more than one type");
      col.is_null[d.id - lo] = 0;
      col.values[d.id - lo] = f.second.second.first;
    }
  for (DumpData::Col& col : dump->filter_cols) {
    if (col.type != 11) continue;  // strings: bytes by slot
    col.str_off.assign(dump->n_docs + 1, 0);
    uint64_t at = 0;
    std::vector<std::pair<uint64_t, uint64_t>> where(dump->n_docs, {0, 0});
    for (const Doc& d : docs)
      for (const auto& f : d.filters)
        if (&dump->filter_cols[f.first] == &col && f.second.first == 11) where[d.id - lo] = f.second.second;
    for (uint64_t sl = 0; sl < dump->n_docs; ++sl) {
      col.str_off[sl] = at;
      at += where[sl].second;
    }
    col.str_off[dump->n_docs] = at;
    col.str_bytes.assign(at + 1, 0);
    for (uint64_t sl = 0; sl < dump->n_docs; ++sl)
      if (where[sl].second) std::memcpy(col.str_bytes.data() + col.str_off[sl], dr.p + where[sl].first, where[sl].second);
  }
  // ---- columns: from the texts when the store kept them (tf and doc lengths need them), else the index's doc ids alone ----
  if (tbytes != 0) {
    mgx_build_params bp{sizeof(mgx_build_params), MGX_ABI_VERSION, dump->info.ngram_size, dump->info.kanji_ngram_size,
                        dump->info.cross_boundary_ngrams, 0};
    Columns* built = nullptr;
    const int rc = BuildColumns(bp, dump->text_bytes.data(), dump->text_off.data(), lo, dump->n_docs, &built, err);
    if (rc != MGX_OK) return rc;
    std::unique_ptr<Columns, void (*)(Columns*)> guard(built, DestroyColumns);
    // the dump's own index must say the same: same grams, same lists
    if (built->sorted_keys != index_cols->sorted_keys) return fail(MGX_ERR_INVALID_ARGUMENT, "the dump's index and its stored texts disagree (different n-grams)");
    const uint64_t P = built->offsets.back();
    if (index_cols->offsets != built->offsets || !std::equal(built->docids.begin(), built->docids.begin() + P, index_cols->docids.begin()))
      return fail(MGX_ERR_INVALID_ARGUMENT, "the dump's index and its stored texts disagree (different posting lists)");
    dump->cols = guard.release();
    dump->has_scores = true;
  } else {
    Columns* c = index_cols.release();
    // (re-based to the common doc range)
    c->first_doc_id = lo;
    c->n_docs = dump->n_docs;
    dump->cols = c;
  }
  *out = dump.release();
  return MGX_OK;
}

void DumpDestroy(DumpData* d) { delete d; }
void DumpView(const DumpData* d, mgx_dump_view* v) {
  v->table_name = d->table.c_str();
  v->index_info = d->info;
  v->first_doc_id = d->first_doc_id;
  v->n_docs = d->n_docs;
  v->n_existing = d->n_existing;
  v->exists = d->exists.data();
  v->text_bytes = d->text_bytes.data();
  v->text_off = d->text_off.data();
  v->n_filter_columns = static_cast<uint32_t>(d->filter_cols.size());
  v->has_texts = d->has_scores ? 1 : 0;
}
bool DumpFilterColumn(const DumpData* d, uint32_t i, mgx_dump_filter_column* o) {
  if (i >= d->filter_cols.size()) return false;
  const DumpData::Col& c = d->filter_cols[i];
  o->name = c.name.c_str();
  o->value_type = c.type;
  o->values = c.values.data();
  o->is_null = c.is_null.data();
  o->string_bytes = c.str_bytes.empty() ? nullptr : c.str_bytes.data();
  o->string_off = c.str_off.empty() ? nullptr : c.str_off.data();
  return true;
}
Columns* DumpTakeColumns(DumpData* d) {
  Columns* c = d->cols;
  d->cols = nullptr;
  return c;
}

}  // namespace mgx
