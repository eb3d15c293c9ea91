// mgx_tools.cpp — deterministic synthetic corpus generator (bench/test tooling, see include/mygram_tools.h).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mygram_tools.h"

namespace {

inline uint64_t SplitMix64(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Xoshiro {
  uint64_t s[4];
  explicit Xoshiro(uint64_t seed) {
    for (auto& v : s) v = SplitMix64(seed);
  }
  static inline uint64_t Rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  inline uint64_t Next() {
    const uint64_t r = Rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = Rotl(s[3], 45);
    return r;
  }
  inline double Uniform() { return static_cast<double>(Next() >> 11) * (1.0 / 9007199254740992.0); }
};

constexpr int kVocab = 20000;

struct Vocab {
  std::vector<std::string> words;
  std::vector<double> cdf;  // Zipf(s=1) over ranks 1..kVocab
  explicit Vocab(uint64_t seed) {
    static const char kLetters[] = "etaoinshrdlcumwfgypbvkjxqz";
    double lw[26], lsum = 0;
    for (int i = 0; i < 26; ++i) lsum += (lw[i] = 1.0 / (i + 1));
    double lcdf[26], acc = 0;
    for (int i = 0; i < 26; ++i) lcdf[i] = (acc += lw[i] / lsum);
    Xoshiro rng(seed ^ 0x5EEDC0DE5EEDC0DEull);
    words.reserve(kVocab);
    for (int w = 0; w < kVocab; ++w) {
      const int len = 2 + static_cast<int>(rng.Next() % 9);
      std::string s;
      for (int i = 0; i < len; ++i) {
        const double u = rng.Uniform();
        int k = 0;
        while (k < 25 && u > lcdf[k]) ++k;
        s.push_back(kLetters[k]);
      }
      words.push_back(std::move(s));
    }
    cdf.resize(kVocab);
    double zsum = 0;
    for (int r = 1; r <= kVocab; ++r) zsum += 1.0 / r;
    acc = 0;
    for (int r = 1; r <= kVocab; ++r) cdf[r - 1] = (acc += (1.0 / r) / zsum);
    cdf[kVocab - 1] = 1.0;
  }
};

}  // namespace

struct mgxt_corpus {
  std::vector<uint8_t> bytes;
  std::vector<uint64_t> off;
  uint64_t n_docs = 0;
};

extern "C" {

int mgxt_corpus_generate(uint64_t seed, uint64_t global_first, uint64_t n_docs, int n_threads, mgxt_corpus** out) {
  if (!out) return 2;
  *out = nullptr;
  try {
    const Vocab vocab(seed);
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = n_threads > 0 ? static_cast<unsigned>(n_threads) : (hw ? hw : 4);
    const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>((n_docs + 8191) / 8192, nt * 8ull));
    const uint64_t chunk = (n_docs + n_chunks - 1) / n_chunks;
    std::vector<std::vector<uint8_t>> cb(n_chunks);
    std::vector<std::vector<uint32_t>> cl(n_chunks);
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
      th.emplace_back([&]() {
        for (;;) {
          const uint64_t c = next.fetch_add(1);
          if (c >= n_chunks) break;
          const uint64_t d0 = c * chunk, d1 = std::min(n_docs, d0 + chunk);
          auto& b = cb[c];
          auto& l = cl[c];
          b.reserve((d1 - d0) * 72);
          l.reserve(d1 - d0);
          for (uint64_t d = d0; d < d1; ++d) {
            uint64_t s = seed * 0xD1342543DE82EF95ull + (global_first + d) * 0x9E3779B97F4A7C15ull + 1;
            Xoshiro rng(s);
            const int n_words = 4 + static_cast<int>(rng.Next() % 13);
            const size_t start = b.size();
            for (int w = 0; w < n_words; ++w) {
              const double u = rng.Uniform();
              const size_t r = std::lower_bound(vocab.cdf.begin(), vocab.cdf.end(), u) - vocab.cdf.begin();
              const std::string& word = vocab.words[std::min<size_t>(r, kVocab - 1)];
              if (w) b.push_back(' ');
              b.insert(b.end(), word.begin(), word.end());
            }
            l.push_back(static_cast<uint32_t>(b.size() - start));
          }
        }
      });
    }
    for (auto& t : th) t.join();
    auto c = new mgxt_corpus();
    c->n_docs = n_docs;
    c->off.resize(n_docs + 1);
    uint64_t total = 0;
    for (auto& b : cb) total += b.size();
    c->bytes.resize(total + 16);
    uint64_t at = 0, d = 0;
    c->off[0] = 0;
    for (uint64_t k = 0; k < n_chunks; ++k) {
      if (!cb[k].empty()) std::memcpy(c->bytes.data() + at, cb[k].data(), cb[k].size());
      uint64_t o = at;
      for (uint32_t len : cl[k]) {
        o += len;
        c->off[++d] = o;
      }
      at += cb[k].size();
      std::vector<uint8_t>().swap(cb[k]);
    }
    *out = c;
    return 0;
  } catch (...) {
    return 5;
  }
}

int mgxt_corpus_view(const mgxt_corpus* c, const uint8_t** text_bytes, const uint64_t** text_off, uint64_t* n_docs) {
  if (!c || !text_bytes || !text_off || !n_docs) return 2;
  *text_bytes = c->bytes.data();
  *text_off = c->off.data();
  *n_docs = c->n_docs;
  return 0;
}

void mgxt_corpus_destroy(mgxt_corpus* c) { delete c; }

}  // extern "C"
