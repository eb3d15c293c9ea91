#define _GNU_SOURCE /* memmem */
/*
 * mygram_oracle.c — CPU restatement of MygramDB's query hot path. TEST INFRASTRUCTURE ONLY (see mygram_oracle.h).
 *
 * Plain C11, single-threaded, written for clarity, not speed. Every function cites the reference lines it follows
 * (paths relative to /root/reference). Nothing here is reachable from the product library.
 */
#include "mygram_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ================================================================================================
 * small helpers
 * ============================================================================================== */

typedef struct {
  uint32_t* v;
  size_t n, cap;
} u32vec;

static void u32vec_push(u32vec* a, uint32_t x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 16;
    a->v = (uint32_t*)realloc(a->v, a->cap * sizeof(uint32_t));
  }
  a->v[a->n++] = x;
}

static uint32_t* u32_dup(const uint32_t* src, size_t n) {
  uint32_t* out = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
  if (n) memcpy(out, src, n * sizeof(uint32_t));
  return out;
}

void orc_free(void* p) { free(p); }

/* std::set_intersection on two ascending arrays. */
static u32vec set_intersection_u32(const uint32_t* a, size_t na, const uint32_t* b, size_t nb) {
  u32vec out = {0};
  size_t i = 0, j = 0;
  while (i < na && j < nb) {
    if (a[i] < b[j]) {
      ++i;
    } else if (b[j] < a[i]) {
      ++j;
    } else {
      u32vec_push(&out, a[i]);
      ++i;
      ++j;
    }
  }
  return out;
}

/* std::set_union. */
static u32vec set_union_u32(const uint32_t* a, size_t na, const uint32_t* b, size_t nb) {
  u32vec out = {0};
  size_t i = 0, j = 0;
  while (i < na && j < nb) {
    if (a[i] < b[j]) {
      u32vec_push(&out, a[i++]);
    } else if (b[j] < a[i]) {
      u32vec_push(&out, b[j++]);
    } else {
      u32vec_push(&out, a[i]);
      ++i;
      ++j;
    }
  }
  while (i < na) u32vec_push(&out, a[i++]);
  while (j < nb) u32vec_push(&out, b[j++]);
  return out;
}

/* std::set_difference. */
static u32vec set_difference_u32(const uint32_t* a, size_t na, const uint32_t* b, size_t nb) {
  u32vec out = {0};
  size_t i = 0, j = 0;
  while (i < na) {
    if (j == nb) {
      u32vec_push(&out, a[i++]);
    } else if (a[i] < b[j]) {
      u32vec_push(&out, a[i++]);
    } else if (b[j] < a[i]) {
      ++j;
    } else {
      ++i;
      ++j;
    }
  }
  return out;
}

static int bytes_cmp(const uint8_t* a, size_t na, const uint8_t* b, size_t nb) {
  size_t m = na < nb ? na : nb;
  int c = m ? memcmp(a, b, m) : 0;
  if (c != 0) return c;
  return (na > nb) - (na < nb);
}

/* ================================================================================================
 * text primitives
 * ============================================================================================== */

/* src/utils/string_utils.cpp:94-164 */
int orc_try_parse_utf8(const uint8_t* d, size_t avail, uint32_t* out_cp) {
  if (avail == 0) return -1;
  uint8_t b0 = d[0];
  if ((b0 & 0x80) == 0) {
    *out_cp = b0;
    return 1;
  }
  if ((b0 & 0xE0) == 0xC0) {
    if (b0 < 0xC2) return -1; /* overlong */
    if (avail < 2) return -1;
    if ((d[1] & 0xC0) != 0x80) return -1;
    *out_cp = ((uint32_t)(b0 & 0x1F) << 6) | (d[1] & 0x3F);
    return 2;
  }
  if ((b0 & 0xF0) == 0xE0) {
    if (avail < 3) return -1;
    if ((d[1] & 0xC0) != 0x80 || (d[2] & 0xC0) != 0x80) return -1;
    uint32_t cp = ((uint32_t)(b0 & 0x0F) << 12) | ((uint32_t)(d[1] & 0x3F) << 6) | (d[2] & 0x3F);
    if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) return -1;
    *out_cp = cp;
    return 3;
  }
  if ((b0 & 0xF8) == 0xF0) {
    if (b0 > 0xF4) return -1;
    if (avail < 4) return -1;
    if ((d[1] & 0xC0) != 0x80 || (d[2] & 0xC0) != 0x80 || (d[3] & 0xC0) != 0x80) return -1;
    uint32_t cp = ((uint32_t)(b0 & 0x07) << 18) | ((uint32_t)(d[1] & 0x3F) << 12) | ((uint32_t)(d[2] & 0x3F) << 6) |
                  (d[3] & 0x3F);
    if (cp < 0x10000 || cp > 0x10FFFF) return -1;
    *out_cp = cp;
    return 4;
  }
  return -1;
}

/* src/utils/string_utils.cpp:655-669 */
size_t orc_count_code_points(const uint8_t* text, size_t len) {
  size_t count = 0;
  for (size_t i = 0; i < len;) {
    uint32_t cp = 0;
    int n = orc_try_parse_utf8(text + i, len - i, &cp);
    if (n < 0) {
      ++i;
      continue;
    }
    i += (size_t)n;
    ++count;
  }
  return count;
}

/* std::string_view::find(term, pos): first index >= pos where term occurs, or (size_t)-1. */
static size_t bytes_find(const uint8_t* text, size_t text_len, const uint8_t* term, size_t term_len, size_t pos) {
  if (term_len == 0) return pos <= text_len ? pos : (size_t)-1;
  if (term_len > text_len || pos > text_len - term_len) return (size_t)-1;
  /* glibc memmem (two-way / SIMD first-byte scan): what libstdc++'s string_view::find does with memchr + memcmp */
  const uint8_t* hit = (const uint8_t*)memmem(text + pos, text_len - pos, term, term_len);
  return hit ? (size_t)(hit - text) : (size_t)-1;
}

/* src/index/bm25_scorer.cpp:27-45 */
uint32_t orc_count_term_occurrences(const uint8_t* text, size_t text_len, const uint8_t* term, size_t term_len) {
  if (text_len == 0 || term_len == 0) return 0;
  if (term_len > text_len) return 0;
  uint32_t count = 0;
  size_t pos = 0;
  while (pos <= text_len - term_len) {
    size_t found = bytes_find(text, text_len, term, term_len, pos);
    if (found == (size_t)-1) break;
    ++count;
    pos = found + term_len; /* non-overlapping */
  }
  return count;
}

/* src/index/bm25_scorer.cpp:14-25 */
double orc_compute_idf(uint64_t total_docs, uint64_t doc_freq) {
  if (total_docs == 0) return 0.0;
  if (doc_freq > total_docs) doc_freq = total_docs;
  double n = (double)total_docs;
  double df = (double)doc_freq;
  return log((n - df + 0.5) / (df + 0.5) + 1.0);
}

/* src/utils/string_utils.cpp:371-377 (non-ICU branch; ASCII range only — see header). */
void orc_normalize_ascii_lower(const uint8_t* text, size_t len, uint8_t* out) {
  for (size_t i = 0; i < len; ++i) {
    uint8_t c = text[i];
    out[i] = (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c;
  }
}

/* ---- string lists ---- */

static void strlist_push(orc_strlist* l, const uint8_t* s, size_t n) {
  if (l->off == NULL) {
    l->cap_count = 16;
    l->off = (uint32_t*)malloc((l->cap_count + 1) * sizeof(uint32_t));
    l->off[0] = 0;
  }
  if (l->count == l->cap_count) {
    l->cap_count *= 2;
    l->off = (uint32_t*)realloc(l->off, (l->cap_count + 1) * sizeof(uint32_t));
  }
  size_t cur = l->off[l->count];
  if (cur + n > l->cap_bytes) {
    l->cap_bytes = (cur + n) * 2 + 16;
    l->bytes = (uint8_t*)realloc(l->bytes, l->cap_bytes);
  }
  if (n) memcpy(l->bytes + cur, s, n);
  l->off[++l->count] = (uint32_t)(cur + n);
}

void orc_strlist_free(orc_strlist* l) {
  free(l->bytes);
  free(l->off);
  memset(l, 0, sizeof(*l));
}

/* src/utils/string_utils.cpp:199-218 Utf8ToCodepoints: invalid bytes skipped one at a time. */
static uint32_t* utf8_to_codepoints(const uint8_t* text, size_t len, size_t* out_n) {
  uint32_t* cps = (uint32_t*)malloc((len ? len : 1) * sizeof(uint32_t));
  size_t n = 0, i = 0;
  while (i < len) {
    uint32_t cp = 0;
    int k = orc_try_parse_utf8(text + i, len - i, &cp);
    if (k > 0) {
      cps[n++] = cp;
      i += (size_t)k;
    } else {
      ++i;
    }
  }
  *out_n = n;
  return cps;
}

/* src/utils/string_utils.cpp:241-272 CodepointsToUtf8 (surrogates and >0x10FFFF skipped). */
static size_t codepoints_to_utf8(const uint32_t* begin, const uint32_t* end, uint8_t* out) {
  size_t n = 0;
  for (const uint32_t* it = begin; it != end; ++it) {
    uint32_t cp = *it;
    if ((cp >= 0xD800 && cp <= 0xDFFF) || cp > 0x10FFFF) continue;
    if (cp <= 0x7F) {
      out[n++] = (uint8_t)cp;
    } else if (cp <= 0x7FF) {
      out[n++] = (uint8_t)(0xC0 | (cp >> 6));
      out[n++] = (uint8_t)(0x80 | (cp & 0x3F));
    } else if (cp <= 0xFFFF) {
      out[n++] = (uint8_t)(0xE0 | (cp >> 12));
      out[n++] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F));
      out[n++] = (uint8_t)(0x80 | (cp & 0x3F));
    } else {
      out[n++] = (uint8_t)(0xF0 | (cp >> 18));
      out[n++] = (uint8_t)(0x80 | ((cp >> 12) & 0x3F));
      out[n++] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F));
      out[n++] = (uint8_t)(0x80 | (cp & 0x3F));
    }
  }
  return n;
}

/* src/utils/string_utils.cpp:382-423 */
void orc_generate_ngrams(const uint8_t* text, size_t len, int n, orc_strlist* out) {
  memset(out, 0, sizeof(*out));
  size_t cpn = 0;
  uint32_t* cps = utf8_to_codepoints(text, len, &cpn);
  if (cpn == 0 || n <= 0 || cpn < (size_t)n) {
    free(cps);
    return;
  }
  uint8_t buf[4 * 64];
  uint8_t* tmp = (size_t)n * 4 <= sizeof(buf) ? buf : (uint8_t*)malloc((size_t)n * 4);
  for (size_t i = 0; i + (size_t)n <= cpn; ++i) {
    size_t k = codepoints_to_utf8(cps + i, cps + i + n, tmp);
    strlist_push(out, tmp, k);
  }
  if (tmp != buf) free(tmp);
  free(cps);
}

/* src/utils/string_utils.cpp:441-448 */
static int is_cjk_ideograph(uint32_t cp) {
  return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
         (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0xF900 && cp <= 0xFAFF);
}

/* src/server/search_pipeline.cpp:70-78 (this copy of the predicate knows one more CJK extension block than the one in
 * string_utils.cpp) */
static int pipeline_is_cjk_ideograph(uint32_t cp) {
  return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
         (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
         (cp >= 0xF900 && cp <= 0xFAFF);
}

/* src/server/search_pipeline.cpp:80-136 HasUncoveredHybridFragment: a normalized term that mixes CJK ideographs with
 * other code points and has a code point that no query n-gram (of the size its script starts) covers. */
int orc_has_uncovered_hybrid_fragment(const uint8_t* term, size_t len, int ngram_size, int kanji_ngram_size,
                                      int cross_boundary) {
  if (len == 0 || kanji_ngram_size <= 0) return 0;
  const int ascii_n = ngram_size > 0 ? ngram_size : 2;
  size_t cpn = 0;
  uint32_t* cps = utf8_to_codepoints(term, len, &cpn);
  if (cpn < 2) {
    free(cps);
    return 0;
  }
  int has_cjk = 0, has_other = 0;
  for (size_t i = 0; i < cpn; ++i) {
    if (pipeline_is_cjk_ideograph(cps[i])) has_cjk = 1; else has_other = 1;
  }
  if (!has_cjk || !has_other) {
    free(cps);
    return 0;
  }
  uint8_t* covered = (uint8_t*)calloc(cpn, 1);
  for (size_t i = 0; i < cpn; ++i) {
    const int start_is_cjk = pipeline_is_cjk_ideograph(cps[i]);
    const int n = start_is_cjk ? kanji_ngram_size : ascii_n;
    if (n <= 0 || i + (size_t)n > cpn) continue;
    if (!cross_boundary) {
      int crossed = 0;
      for (int j = 1; j < n; ++j) {
        if (pipeline_is_cjk_ideograph(cps[i + (size_t)j]) != start_is_cjk) {
          crossed = 1;
          break;
        }
      }
      if (crossed) continue;
    }
    for (int j = 0; j < n; ++j) covered[i + (size_t)j] = 1;
  }
  int uncovered = 0;
  for (size_t i = 0; i < cpn; ++i) uncovered = uncovered || !covered[i];
  free(covered);
  free(cps);
  return uncovered;
}

/* src/utils/string_utils.cpp:452-509 */
void orc_generate_hybrid_ngrams(const uint8_t* text, size_t len, int ascii_n, int kanji_n, int cross_boundary,
                                orc_strlist* out) {
  memset(out, 0, sizeof(*out));
  if (ascii_n <= 0 || kanji_n <= 0) return;
  size_t cpn = 0;
  uint32_t* cps = utf8_to_codepoints(text, len, &cpn);
  if (cpn == 0) {
    free(cps);
    return;
  }
  int maxn = ascii_n > kanji_n ? ascii_n : kanji_n;
  uint8_t* tmp = (uint8_t*)malloc((size_t)maxn * 4);
  for (size_t i = 0; i < cpn; ++i) {
    int start_is_cjk = is_cjk_ideograph(cps[i]);
    int n = start_is_cjk ? kanji_n : ascii_n;
    if (i + (size_t)n > cpn) continue;
    if (!cross_boundary) {
      int crossed = 0;
      for (int j = 1; j < n; ++j) {
        if (is_cjk_ideograph(cps[i + (size_t)j]) != start_is_cjk) {
          crossed = 1;
          break;
        }
      }
      if (crossed) continue;
    }
    size_t k = codepoints_to_utf8(cps + i, cps + i + n, tmp);
    strlist_push(out, tmp, k);
  }
  free(tmp);
  free(cps);
}

/* src/utils/string_utils.cpp:639-653 */
void orc_generate_query_ngrams(const uint8_t* text, size_t len, int ngram_size, int kanji_ngram_size,
                               int cross_boundary, orc_strlist* out) {
  if (kanji_ngram_size > 0) {
    int eff = ngram_size > 0 ? ngram_size : 2;
    orc_generate_hybrid_ngrams(text, len, eff, kanji_ngram_size, cross_boundary, out);
    return;
  }
  if (ngram_size == 0) {
    /* GenerateHybridNgrams(normalized) with its defaults (2, 1, cross_boundary=true): string_utils.h:81 */
    orc_generate_hybrid_ngrams(text, len, 2, 1, 1, out);
    return;
  }
  orc_generate_ngrams(text, len, ngram_size, out);
}

/* src/utils/string_utils.h:192-196: std::sort + std::unique on std::string (bytewise order). */
void orc_strlist_dedup_sorted(orc_strlist* l) {
  if (l->count < 2) return;
  size_t n = l->count;
  /* insertion sort of (off,len) pairs — lists are at most a few hundred grams */
  uint32_t* s = (uint32_t*)malloc(n * sizeof(uint32_t));
  uint32_t* e = (uint32_t*)malloc(n * sizeof(uint32_t));
  for (size_t i = 0; i < n; ++i) {
    s[i] = l->off[i];
    e[i] = l->off[i + 1];
  }
  for (size_t i = 1; i < n; ++i) {
    uint32_t cs = s[i], ce = e[i];
    size_t j = i;
    while (j > 0 && bytes_cmp(l->bytes + s[j - 1], e[j - 1] - s[j - 1], l->bytes + cs, ce - cs) > 0) {
      s[j] = s[j - 1];
      e[j] = e[j - 1];
      --j;
    }
    s[j] = cs;
    e[j] = ce;
  }
  orc_strlist outl;
  memset(&outl, 0, sizeof(outl));
  for (size_t i = 0; i < n; ++i) {
    if (i > 0 && bytes_cmp(l->bytes + s[i - 1], e[i - 1] - s[i - 1], l->bytes + s[i], e[i] - s[i]) == 0) continue;
    strlist_push(&outl, l->bytes + s[i], e[i] - s[i]);
  }
  free(s);
  free(e);
  orc_strlist_free(l);
  *l = outl;
}

/* ================================================================================================
 * index: gram -> ascending docid list
 * ============================================================================================== */

typedef struct {
  uint8_t* key;
  uint32_t key_len;
  u32vec docs; /* owned unless the index is CSR-adopted */
} posting;

struct orc_index {
  int ngram_size, kanji_ngram_size, cross_boundary;
  /* open-addressing hash table of postings (dynamic mode) */
  posting* slots;
  size_t n_slots, n_used;
  /* CSR-adopted mode */
  int csr;
  size_t n_grams;
  const uint8_t* key_bytes;
  const uint32_t* key_off;
  const uint64_t* offsets;
  const uint32_t* docids;
};

static uint64_t fnv1a(const uint8_t* s, size_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < n; ++i) {
    h ^= s[i];
    h *= 1099511628211ULL;
  }
  return h;
}

orc_index* orc_index_create(int ngram_size, int kanji_ngram_size, int cross_boundary) {
  orc_index* idx = (orc_index*)calloc(1, sizeof(orc_index));
  idx->ngram_size = ngram_size;
  idx->kanji_ngram_size = kanji_ngram_size > 0 ? kanji_ngram_size : ngram_size; /* index.cpp:31 */
  idx->cross_boundary = cross_boundary;
  idx->n_slots = 1024;
  idx->slots = (posting*)calloc(idx->n_slots, sizeof(posting));
  return idx;
}

orc_index* orc_index_from_csr(int ngram_size, int kanji_ngram_size, int cross_boundary, size_t n_grams,
                              const uint8_t* key_bytes, const uint32_t* key_off, const uint64_t* offsets,
                              const uint32_t* docids) {
  orc_index* idx = (orc_index*)calloc(1, sizeof(orc_index));
  idx->ngram_size = ngram_size;
  idx->kanji_ngram_size = kanji_ngram_size > 0 ? kanji_ngram_size : ngram_size;
  idx->cross_boundary = cross_boundary;
  idx->csr = 1;
  idx->n_grams = n_grams;
  idx->key_bytes = key_bytes;
  idx->key_off = key_off;
  idx->offsets = offsets;
  idx->docids = docids;
  return idx;
}

void orc_index_destroy(orc_index* idx) {
  if (!idx) return;
  if (!idx->csr) {
    for (size_t i = 0; i < idx->n_slots; ++i) {
      free(idx->slots[i].key);
      free(idx->slots[i].docs.v);
    }
    free(idx->slots);
  }
  free(idx);
}

static posting* table_find(posting* slots, size_t n_slots, const uint8_t* key, size_t len) {
  size_t i = (size_t)(fnv1a(key, len) & (n_slots - 1));
  for (;;) {
    posting* p = &slots[i];
    if (p->key == NULL) return p;
    if (p->key_len == len && memcmp(p->key, key, len) == 0) return p;
    i = (i + 1) & (n_slots - 1);
  }
}

static void table_grow(orc_index* idx) {
  size_t nn = idx->n_slots * 2;
  posting* ns = (posting*)calloc(nn, sizeof(posting));
  for (size_t i = 0; i < idx->n_slots; ++i) {
    posting* p = &idx->slots[i];
    if (p->key) *table_find(ns, nn, p->key, p->key_len) = *p;
  }
  free(idx->slots);
  idx->slots = ns;
  idx->n_slots = nn;
}

/* PostingList::Add (src/index/posting_list.cpp:239-288): O(1) append when ascending, else sorted insert, no dups. */
static void posting_add(u32vec* d, uint32_t doc_id) {
  if (d->n == 0 || d->v[d->n - 1] < doc_id) {
    u32vec_push(d, doc_id);
    return;
  }
  size_t lo = 0, hi = d->n;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (d->v[mid] < doc_id) lo = mid + 1; else hi = mid;
  }
  if (lo < d->n && d->v[lo] == doc_id) return;
  u32vec_push(d, 0);
  memmove(d->v + lo + 1, d->v + lo, (d->n - 1 - lo) * sizeof(uint32_t));
  d->v[lo] = doc_id;
}

/* src/index/index.cpp:39-74 */
int orc_index_add_document(orc_index* idx, uint32_t doc_id, const uint8_t* text, size_t len) {
  if (idx->csr) return 0;
  orc_strlist grams;
  orc_generate_hybrid_ngrams(text, len, idx->ngram_size, idx->kanji_ngram_size, idx->cross_boundary, &grams);
  orc_strlist_dedup_sorted(&grams);
  if (grams.count == 0) {
    orc_strlist_free(&grams);
    return 0;
  }
  for (size_t g = 0; g < grams.count; ++g) {
    const uint8_t* key = grams.bytes + grams.off[g];
    size_t klen = grams.off[g + 1] - grams.off[g];
    if ((idx->n_used + 1) * 2 > idx->n_slots) table_grow(idx);
    posting* p = table_find(idx->slots, idx->n_slots, key, klen);
    if (p->key == NULL) {
      p->key = (uint8_t*)malloc(klen ? klen : 1);
      memcpy(p->key, key, klen);
      p->key_len = (uint32_t)klen;
      idx->n_used++;
    }
    posting_add(&p->docs, doc_id);
  }
  orc_strlist_free(&grams);
  return 1;
}

/* Index::TakePostingSnapshot (src/index/index.cpp:728-747): list for a gram, or NULL/0 when unknown. */
static const uint32_t* index_list(const orc_index* idx, const uint8_t* gram, size_t len, size_t* out_n) {
  if (idx->csr) {
    size_t lo = 0, hi = idx->n_grams;
    while (lo < hi) {
      size_t mid = (lo + hi) / 2;
      int c = bytes_cmp(idx->key_bytes + idx->key_off[mid], idx->key_off[mid + 1] - idx->key_off[mid], gram, len);
      if (c < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < idx->n_grams &&
        bytes_cmp(idx->key_bytes + idx->key_off[lo], idx->key_off[lo + 1] - idx->key_off[lo], gram, len) == 0) {
      *out_n = (size_t)(idx->offsets[lo + 1] - idx->offsets[lo]);
      return idx->docids + idx->offsets[lo];
    }
    *out_n = 0;
    return NULL;
  }
  posting* p = table_find(idx->slots, idx->n_slots, gram, len);
  if (p->key == NULL) {
    *out_n = 0;
    return NULL;
  }
  *out_n = p->docs.n;
  return p->docs.v ? p->docs.v : (const uint32_t*)"";
}

uint64_t orc_index_posting_size(const orc_index* idx, const uint8_t* gram, size_t len) {
  size_t n = 0;
  const uint32_t* l = index_list(idx, gram, len, &n);
  return l ? (uint64_t)n : 0;
}

size_t orc_index_gram_count(const orc_index* idx) { return idx->csr ? idx->n_grams : idx->n_used; }

/* src/index/index.cpp:199-368. The reference has three execution strategies (Roaring chain :228-240, planner
 * :244-332, standard :338-351); all compute the same set, then order/limit identically:
 * ascending; reverse => descending; limit keeps the first `limit` of that order. */
uint32_t* orc_search_and(const orc_index* idx, const uint8_t* tb, const uint32_t* toff, size_t n_terms, size_t limit,
                         int reverse, size_t* out_n) {
  *out_n = 0;
  if (n_terms == 0) return u32_dup(NULL, 0); /* :203 */
  for (size_t i = 0; i < n_terms; ++i) {    /* :211-215 */
    size_t n;
    if (index_list(idx, tb + toff[i], toff[i + 1] - toff[i], &n) == NULL) return u32_dup(NULL, 0);
  }
  size_t n0;
  const uint32_t* l0 = index_list(idx, tb + toff[0], toff[1] - toff[0], &n0);
  u32vec result = {0};
  result.v = u32_dup(l0, n0);
  result.n = result.cap = n0;
  if (result.cap == 0) result.cap = 1;
  for (size_t i = 1; i < n_terms; ++i) { /* :341-351 */
    size_t n;
    const uint32_t* l = index_list(idx, tb + toff[i], toff[i + 1] - toff[i], &n);
    u32vec inter = set_intersection_u32(result.v, result.n, l, n);
    free(result.v);
    result = inter;
    if (result.n == 0) break;
  }
  /* :356-367 */
  if (limit > 0 && result.n > limit) {
    if (reverse) {
      memmove(result.v, result.v + (result.n - limit), limit * sizeof(uint32_t));
      result.n = limit;
      for (size_t i = 0; i < result.n / 2; ++i) {
        uint32_t t = result.v[i];
        result.v[i] = result.v[result.n - 1 - i];
        result.v[result.n - 1 - i] = t;
      }
    } else {
      result.n = limit;
    }
  } else if (reverse) {
    for (size_t i = 0; i < result.n / 2; ++i) {
      uint32_t t = result.v[i];
      result.v[i] = result.v[result.n - 1 - i];
      result.v[result.n - 1 - i] = t;
    }
  }
  *out_n = result.n;
  if (!result.v) result.v = u32_dup(NULL, 0);
  return result.v;
}

/* src/index/index.cpp:418-448 */
uint32_t* orc_search_or(const orc_index* idx, const uint8_t* tb, const uint32_t* toff, size_t n_terms, size_t* out_n) {
  u32vec result = {0};
  for (size_t i = 0; i < n_terms; ++i) {
    size_t n;
    const uint32_t* l = index_list(idx, tb + toff[i], toff[i + 1] - toff[i], &n);
    if (l == NULL) continue; /* unknown term skipped */
    u32vec u = set_union_u32(result.v, result.n, l, n);
    free(result.v);
    result = u;
  }
  *out_n = result.n;
  if (!result.v) result.v = u32_dup(NULL, 0);
  return result.v;
}

/* src/index/index.cpp:450-486 */
uint32_t* orc_search_not(const orc_index* idx, const uint32_t* all_docs, size_t n_all, const uint8_t* tb,
                         const uint32_t* toff, size_t n_terms, size_t* out_n) {
  if (n_terms == 0) {
    *out_n = n_all;
    return u32_dup(all_docs, n_all);
  }
  size_t n_ex = 0;
  uint32_t* excluded = orc_search_or(idx, tb, toff, n_terms, &n_ex);
  u32vec r = set_difference_u32(all_docs, n_all, excluded, n_ex);
  free(excluded);
  *out_n = r.n;
  if (!r.v) r.v = u32_dup(NULL, 0);
  return r.v;
}

/* src/index/index.cpp:488-578 */
uint32_t* orc_search_by_threshold(const orc_index* idx, const uint8_t* tb, const uint32_t* toff, size_t n_terms,
                                  size_t threshold, size_t* out_n) {
  *out_n = 0;
  if (n_terms == 0 || threshold == 0) return u32_dup(NULL, 0);
  orc_strlist uniq;
  memset(&uniq, 0, sizeof(uniq));
  for (size_t i = 0; i < n_terms; ++i) strlist_push(&uniq, tb + toff[i], toff[i + 1] - toff[i]);
  orc_strlist_dedup_sorted(&uniq); /* :496-497 */
  if (threshold > uniq.count) {    /* :499 */
    orc_strlist_free(&uniq);
    return u32_dup(NULL, 0);
  }
  if (threshold == uniq.count) { /* :504 */
    uint32_t* r = orc_search_and(idx, uniq.bytes, uniq.off, uniq.count, 0, 0, out_n);
    orc_strlist_free(&uniq);
    return r;
  }
  /* valid (known) lists only :512-523 */
  const uint32_t** lists = (const uint32_t**)malloc(uniq.count * sizeof(*lists));
  size_t* lens = (size_t*)malloc(uniq.count * sizeof(size_t));
  size_t* pos = (size_t*)calloc(uniq.count, sizeof(size_t));
  size_t nv = 0;
  for (size_t i = 0; i < uniq.count; ++i) {
    size_t n;
    const uint32_t* l = index_list(idx, uniq.bytes + uniq.off[i], uniq.off[i + 1] - uniq.off[i], &n);
    if (l) {
      lists[nv] = l;
      lens[nv] = n;
      nv++;
    }
  }
  u32vec result = {0};
  if (nv >= threshold) {
    /* k-way merge with counting (:532-575); a linear min-scan replaces the heap — same emitted sequence. */
    for (;;) {
      int have = 0;
      uint32_t cur = 0;
      for (size_t i = 0; i < nv; ++i) {
        if (pos[i] < lens[i] && (!have || lists[i][pos[i]] < cur)) {
          cur = lists[i][pos[i]];
          have = 1;
        }
      }
      if (!have) break;
      size_t cnt = 0;
      for (size_t i = 0; i < nv; ++i) {
        if (pos[i] < lens[i] && lists[i][pos[i]] == cur) {
          ++cnt;
          ++pos[i];
        }
      }
      if (cnt >= threshold) u32vec_push(&result, cur);
    }
  }
  free(lists);
  free(lens);
  free(pos);
  orc_strlist_free(&uniq);
  *out_n = result.n;
  if (!result.v) result.v = u32_dup(NULL, 0);
  return result.v;
}

/* PostingList::RetainPresent, src/index/posting_list.cpp:432-474: ascending candidates (duplicates allowed) kept iff
 * present in the list. */
static u32vec retain_present(const uint32_t* list, size_t n, const uint32_t* cand, size_t nc) {
  u32vec out = {0};
  size_t p = 0;
  for (size_t i = 0; i < nc; ++i) {
    while (p < n && list[p] < cand[i]) ++p;
    if (p == n) break;
    if (list[p] == cand[i]) u32vec_push(&out, cand[i]);
  }
  return out;
}

static int cmp_u32(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return (x > y) - (x < y);
}

/* src/index/index.cpp:370-416 */
uint32_t* orc_filter_by_ngrams(const orc_index* idx, const uint32_t* cand, size_t nc, const uint8_t* tb,
                               const uint32_t* toff, size_t n_terms, size_t* out_n) {
  *out_n = 0;
  if (nc == 0) return u32_dup(NULL, 0);
  if (n_terms == 0) { /* :378-380 */
    *out_n = nc;
    return u32_dup(cand, nc);
  }
  for (size_t i = 0; i < n_terms; ++i) { /* :381-385 */
    size_t n;
    if (index_list(idx, tb + toff[i], toff[i + 1] - toff[i], &n) == NULL) return u32_dup(NULL, 0);
  }
  int ascending = 1;
  for (size_t i = 1; i < nc; ++i) {
    if (cand[i] < cand[i - 1]) {
      ascending = 0;
      break;
    }
  }
  uint32_t* sorted = NULL;
  if (!ascending) {
    sorted = u32_dup(cand, nc);
    qsort(sorted, nc, sizeof(uint32_t), cmp_u32);
  }
  u32vec retained = {0};
  {
    size_t n;
    const uint32_t* l = index_list(idx, tb + toff[0], toff[1] - toff[0], &n);
    retained = retain_present(l, n, ascending ? cand : sorted, nc);
  }
  for (size_t i = 1; i < n_terms && retained.n > 0; ++i) {
    size_t n;
    const uint32_t* l = index_list(idx, tb + toff[i], toff[i + 1] - toff[i], &n);
    u32vec r2 = retain_present(l, n, retained.v, retained.n);
    free(retained.v);
    retained = r2;
  }
  free(sorted);
  if (ascending || retained.n == 0) {
    *out_n = retained.n;
    if (!retained.v) retained.v = u32_dup(NULL, 0);
    return retained.v;
  }
  /* :406-415 restore caller order, keeping repeated candidates */
  u32vec ordered = {0};
  for (size_t i = 0; i < nc; ++i) {
    if (bsearch(&cand[i], retained.v, retained.n, sizeof(uint32_t), cmp_u32)) u32vec_push(&ordered, cand[i]);
  }
  free(retained.v);
  *out_n = ordered.n;
  if (!ordered.v) ordered.v = u32_dup(NULL, 0);
  return ordered.v;
}

/* ================================================================================================
 * document store (texts) and BM25
 * ============================================================================================== */

typedef struct {
  uint32_t doc_id;
  int has_text;
  uint8_t* text;
  size_t len;
} doc_entry;

struct orc_docstore {
  doc_entry* docs; /* kept sorted by doc_id */
  size_t n, cap;
  int adopted;
  size_t n_adopted;
  const uint8_t* text_bytes;
  const uint64_t* text_off;
};

orc_docstore* orc_docstore_create(void) { return (orc_docstore*)calloc(1, sizeof(orc_docstore)); }

orc_docstore* orc_docstore_from_arrays(size_t n_docs, const uint8_t* text_bytes, const uint64_t* text_off) {
  orc_docstore* ds = (orc_docstore*)calloc(1, sizeof(orc_docstore));
  ds->adopted = 1;
  ds->n_adopted = n_docs;
  ds->text_bytes = text_bytes;
  ds->text_off = text_off;
  return ds;
}

void orc_docstore_destroy(orc_docstore* ds) {
  if (!ds) return;
  for (size_t i = 0; i < ds->n; ++i) free(ds->docs[i].text);
  free(ds->docs);
  free(ds);
}

void orc_docstore_add(orc_docstore* ds, uint32_t doc_id, const uint8_t* text, size_t len, int has_text) {
  if (ds->adopted) return;
  if (ds->n == ds->cap) {
    ds->cap = ds->cap ? ds->cap * 2 : 64;
    ds->docs = (doc_entry*)realloc(ds->docs, ds->cap * sizeof(doc_entry));
  }
  size_t pos = ds->n;
  while (pos > 0 && ds->docs[pos - 1].doc_id > doc_id) {
    ds->docs[pos] = ds->docs[pos - 1];
    --pos;
  }
  doc_entry* e = &ds->docs[pos];
  e->doc_id = doc_id;
  e->has_text = has_text;
  e->len = has_text ? len : 0;
  e->text = (uint8_t*)malloc(e->len ? e->len : 1);
  if (e->len) memcpy(e->text, text, e->len);
  ds->n++;
}

/* DocumentStore::VisitNormalizedTextsFor (src/storage/document_store_retrieval.cpp:289-322): text pointer for a doc,
 * NULL when the doc is unknown or has no stored text. */
static const uint8_t* docstore_text(const orc_docstore* ds, uint32_t doc_id, size_t* len) {
  *len = 0;
  if (ds->adopted) {
    if (doc_id == 0 || doc_id > ds->n_adopted) return NULL;
    *len = (size_t)(ds->text_off[doc_id] - ds->text_off[doc_id - 1]);
    return ds->text_bytes + ds->text_off[doc_id - 1];
  }
  size_t lo = 0, hi = ds->n;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (ds->docs[mid].doc_id < doc_id) lo = mid + 1; else hi = mid;
  }
  if (lo == ds->n || ds->docs[lo].doc_id != doc_id || !ds->docs[lo].has_text) return NULL;
  *len = ds->docs[lo].len;
  return ds->docs[lo].text;
}

/* src/server/server_types.h:157-193 (fed by ingest: src/mysql/binlog_event_processor.cpp:98-99). */
void orc_docstore_bm25_stats(const orc_docstore* ds, uint64_t* doc_count, uint64_t* total_len) {
  uint64_t c = 0, t = 0;
  size_t n = ds->adopted ? ds->n_adopted : ds->n;
  for (size_t i = 0; i < n; ++i) {
    size_t len;
    const uint8_t* tx = docstore_text(ds, ds->adopted ? (uint32_t)(i + 1) : ds->docs[i].doc_id, &len);
    if (tx && len > 0) {
      c++;
      t += orc_count_code_points(tx, len);
    }
  }
  *doc_count = c;
  *total_len = t;
}

/* src/index/bm25_scorer.cpp:47-99 — arithmetic kept in the reference's exact operation order. */
int orc_score_documents(const orc_docstore* ds, const uint32_t* cand, size_t nc, const uint8_t* tb,
                        const uint32_t* toff, size_t n_terms, const uint64_t* dfs, size_t n_dfs, uint64_t total_docs,
                        double avg_doc_length, double k1, double b, double* scores_out) {
  if (n_terms != n_dfs) return 11;
  double* idfs = (double*)malloc((n_terms ? n_terms : 1) * sizeof(double));
  for (size_t i = 0; i < n_terms; ++i) idfs[i] = orc_compute_idf(total_docs, dfs[i]);
  for (size_t c = 0; c < nc; ++c) {
    size_t len;
    const uint8_t* text = docstore_text(ds, cand[c], &len);
    double score = 0.0;
    if (text != NULL && len > 0) {
      double doc_length = (double)orc_count_code_points(text, len);
      for (size_t i = 0; i < n_terms; ++i) {
        double tf = (double)orc_count_term_occurrences(text, len, tb + toff[i], toff[i + 1] - toff[i]);
        if (tf > 0.0) {
          double length_norm = 1.0 - b + b * doc_length / (avg_doc_length > 1.0 ? avg_doc_length : 1.0);
          double numerator = tf * (k1 + 1.0);
          double denominator = tf + k1 * length_norm;
          score += idfs[i] * numerator / denominator;
        }
      }
    }
    scores_out[c] = score;
  }
  free(idfs);
  return 0;
}

/* src/query/result_sorter.cpp:661-716. The comparator is a strict total order on (score, doc_id) whenever doc ids
 * are distinct, so full sort and partial_sort give the same prefix; a full sort is used here. */
typedef struct {
  uint32_t doc_id;
  double score;
} score_entry;

static int score_cmp_desc(const void* pa, const void* pb) {
  const score_entry* a = (const score_entry*)pa;
  const score_entry* b = (const score_entry*)pb;
  if (a->score == b->score) {
    if (a->doc_id == b->doc_id) return 0;
    return a->doc_id > b->doc_id ? -1 : 1;
  }
  return a->score > b->score ? -1 : 1;
}

static int score_cmp_asc(const void* pa, const void* pb) {
  const score_entry* a = (const score_entry*)pa;
  const score_entry* b = (const score_entry*)pb;
  if (a->score == b->score) {
    if (a->doc_id == b->doc_id) return 0;
    return a->doc_id < b->doc_id ? -1 : 1;
  }
  return a->score < b->score ? -1 : 1;
}

uint32_t* orc_sort_by_score(const uint32_t* results, const double* scores, size_t n, int descending, uint32_t limit,
                            uint32_t offset, size_t* out_n) {
  *out_n = 0;
  if (n == 0) return u32_dup(NULL, 0);
  score_entry* e = (score_entry*)malloc(n * sizeof(score_entry));
  for (size_t i = 0; i < n; ++i) {
    e[i].doc_id = results[i];
    e[i].score = scores[i];
  }
  qsort(e, n, sizeof(score_entry), descending ? score_cmp_desc : score_cmp_asc);
  size_t start = offset < n ? offset : n;
  size_t end = limit == 0 ? n : (start + limit < n ? start + limit : n);
  uint32_t* out = (uint32_t*)malloc(((end - start) ? (end - start) : 1) * sizeof(uint32_t));
  for (size_t i = start; i < end; ++i) out[i - start] = e[i].doc_id;
  free(e);
  *out_n = end - start;
  return out;
}

/* ================================================================================================
 * conjunctive pipeline
 * ============================================================================================== */

typedef struct {
  orc_strlist ngrams;
  size_t estimated_size;
  uint64_t df;
  uint8_t* normalized;
  size_t normalized_len;
  uint32_t src_index;
} term_info;

/* query::SearchNormalizedSubstring (src/query/substring_search.h:24-42): docs whose stored text contains the term —
 * only reached for terms too short to yield an n-gram. */
static u32vec substring_docs(const orc_docstore* ds, const uint8_t* term, size_t len) {
  u32vec out = {0};
  if (!ds) return out;
  size_t n = ds->adopted ? ds->n_adopted : ds->n;
  for (size_t i = 0; i < n; ++i) {
    uint32_t id = ds->adopted ? (uint32_t)(i + 1) : ds->docs[i].doc_id;
    size_t tl;
    const uint8_t* tx = docstore_text(ds, id, &tl);
    if (tx && bytes_find(tx, tl, term, len, 0) != (size_t)-1) u32vec_push(&out, id);
  }
  return out;
}

/* SearchTermDocuments, src/server/search_pipeline.cpp:438-446 */
static u32vec search_term_documents(const orc_index* idx, const orc_docstore* ds, const term_info* ti) {
  u32vec out = {0};
  if (ti->ngrams.count == 0) return substring_docs(ds, ti->normalized, ti->normalized_len);
  out.v = orc_search_and(idx, ti->ngrams.bytes, ti->ngrams.off, ti->ngrams.count, 0, 0, &out.n);
  out.cap = out.n;
  return out;
}

/* src/server/search_pipeline.cpp:569-603 (+ :542-565) */
static void make_term_info(const orc_index* idx, const orc_docstore* ds, const uint8_t* term, size_t len,
                           int ngram_size, int kanji_ngram_size, int cross_boundary, int compute_df, term_info* ti) {
  memset(ti, 0, sizeof(*ti));
  ti->normalized = (uint8_t*)malloc(len ? len : 1);
  ti->normalized_len = len;
  orc_normalize_ascii_lower(term, len, ti->normalized);
  orc_generate_query_ngrams(ti->normalized, len, ngram_size, kanji_ngram_size, cross_boundary, &ti->ngrams);
  orc_strlist_dedup_sorted(&ti->ngrams);
  size_t min_size = (size_t)-1;
  for (size_t g = 0; g < ti->ngrams.count; ++g) {
    uint64_t ps = orc_index_posting_size(idx, ti->ngrams.bytes + ti->ngrams.off[g],
                                         ti->ngrams.off[g + 1] - ti->ngrams.off[g]);
    if (ps > 0) {
      if ((size_t)ps < min_size) min_size = (size_t)ps;
    } else {
      min_size = 0;
      break;
    }
  }
  ti->estimated_size = min_size;
  ti->df = 0;
  if (compute_df && ds != NULL && ti->ngrams.count > 0 && min_size != 0 && min_size != (size_t)-1) {
    size_t nc = 0;
    uint32_t* cand = orc_search_and(idx, ti->ngrams.bytes, ti->ngrams.off, ti->ngrams.count, 0, 0, &nc);
    uint64_t matching = 0;
    for (size_t i = 0; i < nc; ++i) {
      size_t tl;
      const uint8_t* tx = docstore_text(ds, cand[i], &tl);
      if (tx && bytes_find(tx, tl, ti->normalized, ti->normalized_len, 0) != (size_t)-1) ++matching;
    }
    free(cand);
    ti->df = matching;
  }
}

static void free_term_info(term_info* ti) {
  orc_strlist_free(&ti->ngrams);
  free(ti->normalized);
}

void orc_pipeline_result_free(orc_pipeline_result* r) {
  free(r->results);
  r->results = NULL;
}

int orc_execute(const orc_index* idx, const orc_docstore* ds, const uint8_t* tb, const uint32_t* toff, size_t n_terms,
                const uint8_t* nb, const uint32_t* noff, size_t n_not, const orc_filter* filters, size_t n_filters,
                int ngram_size, int kanji_ngram_size, int cross_boundary, size_t filter_threshold, int compute_df,
                int verify_text, orc_pipeline_result* out) {
  memset(out, 0, sizeof(*out));
  if (n_terms > 64) return 11;
  term_info* tis = (term_info*)calloc(n_terms ? n_terms : 1, sizeof(term_info));
  for (size_t i = 0; i < n_terms; ++i) {
    make_term_info(idx, ds, tb + toff[i], toff[i + 1] - toff[i], ngram_size, kanji_ngram_size, cross_boundary,
                   compute_df, &tis[i]);
    tis[i].src_index = (uint32_t)i;
  }
  /* :2012-2014 std::sort by estimated_size. libstdc++ sorts ranges of <=16 elements by insertion sort, which keeps
   * equal keys in input order; that is what is restated (queries carry <=64 terms, ties beyond 16 terms are
   * implementation-defined in the reference too). */
  for (size_t i = 1; i < n_terms; ++i) {
    term_info cur = tis[i];
    size_t j = i;
    while (j > 0 && cur.estimated_size < tis[j - 1].estimated_size) {
      tis[j] = tis[j - 1];
      --j;
    }
    tis[j] = cur;
  }
  out->n_terms = n_terms;
  for (size_t i = 0; i < n_terms; ++i) {
    out->term_order[i] = tis[i].src_index;
    out->term_df[i] = tis[i].df;
    out->term_estimated_size[i] = (uint64_t)tis[i].estimated_size;
  }

  u32vec results = {0};
  /* Execute :804-810 */
  for (size_t i = 0; i < n_terms; ++i) {
    if ((tis[i].estimated_size == 0 || tis[i].estimated_size == (size_t)-1) &&
        (tis[i].ngrams.count != 0 || tis[i].normalized_len == 0)) {
      out->empty_term_detected = 1;
      goto done;
    }
  }
  /* :813-839 */
  if (n_terms > 0) {
    results = search_term_documents(idx, ds, &tis[0]);
    out->total_candidates = results.n;
    for (size_t i = 1; i < n_terms && results.n > 0; ++i) {
      if (tis[i].ngrams.count == 0) {
        u32vec ar = search_term_documents(idx, ds, &tis[i]);
        u32vec inter = set_intersection_u32(results.v, results.n, ar.v, ar.n);
        free(ar.v);
        free(results.v);
        results = inter;
        continue;
      }
      if (results.n <= filter_threshold) {
        u32vec f = {0};
        f.v = orc_filter_by_ngrams(idx, results.v, results.n, tis[i].ngrams.bytes, tis[i].ngrams.off,
                                   tis[i].ngrams.count, &f.n);
        free(results.v);
        results = f;
      } else {
        u32vec ar = search_term_documents(idx, ds, &tis[i]);
        u32vec inter = set_intersection_u32(results.v, results.n, ar.v, ar.n);
        free(ar.v);
        free(results.v);
        results = inter;
      }
    }
  }
  out->after_intersection = results.n;

  /* ApplyNotFilter :871-932 (no synonym dictionary) */
  if (n_not > 0 && results.n > 0) {
    u32vec excluded = {0};
    for (size_t i = 0; i < n_not; ++i) {
      term_info nti;
      make_term_info(idx, ds, nb + noff[i], noff[i + 1] - noff[i], ngram_size, kanji_ngram_size, cross_boundary, 0,
                     &nti);
      u32vec td = search_term_documents(idx, ds, &nti);
      u32vec u = set_union_u32(excluded.v, excluded.n, td.v, td.n);
      free(td.v);
      free(excluded.v);
      excluded = u;
      free_term_info(&nti);
    }
    if (excluded.n > 0) {
      u32vec d = set_difference_u32(results.v, results.n, excluded.v, excluded.n);
      free(results.v);
      results = d;
    }
    free(excluded.v);
  }
  out->after_not = results.n;

  /* ApplyFiltersWithBitmap :1196-1237 — EQ => AND, NE => ANDNOT, applied in order */
  for (size_t f = 0; f < n_filters; ++f) {
    u32vec r2 = filters[f].negate ? set_difference_u32(results.v, results.n, filters[f].docs, filters[f].n_docs)
                                  : set_intersection_u32(results.v, results.n, filters[f].docs, filters[f].n_docs);
    free(results.v);
    results = r2;
  }
  out->after_filters = results.n;
  /* ApplyVerifyTextFilter :856-857 (:1248-1266) filters by exact text when memory.verify_text says so ("off" is the
   * default, src/config/config.h:317-330).
   * :858-866: a mixed-script term with a code point no query n-gram covers makes the n-gram AND too weak, so the
   * results are filtered by the exact text of ALL terms (PostFilterByText :1239-1246). */
  {
    int exact = verify_text != 0; /* the caller's ShouldApplyVerifyText(memory.verify_text, terms) :42-66 */
    for (size_t i = 0; i < n_terms; ++i)
      exact = exact || orc_has_uncovered_hybrid_fragment(tis[i].normalized, tis[i].normalized_len, ngram_size,
                                                         kanji_ngram_size, cross_boundary);
    out->exact_text_applied = exact;
    if (exact && results.n > 0 && ds != NULL) {
      size_t w = 0;
      for (size_t r = 0; r < results.n; ++r) {
        size_t tl;
        const uint8_t* tx = docstore_text(ds, results.v[r], &tl);
        int all = tx != NULL;
        for (size_t i = 0; all && i < n_terms; ++i)
          all = bytes_find(tx, tl, tis[i].normalized, tis[i].normalized_len, 0) != (size_t)-1;
        if (all) results.v[w++] = results.v[r];
      }
      results.n = w;
    }
  }

done:
  out->results = results.v ? results.v : u32_dup(NULL, 0);
  out->n_results = results.n;
  for (size_t i = 0; i < n_terms; ++i) free_term_info(&tis[i]);
  free(tis);
  return 0;
}

/* ---- fuzzy search (FUZZY d) ----------------------------------------------------------------------------------- */

/* src/utils/edit_distance.cpp:75-127 ComputeDistanceImpl over code points (bytes of ASCII strings are their code
 * points): Levenshtein distance, max_distance + 1 once it cannot come back under max_distance. */
static uint32_t levenshtein_cp(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb, uint32_t max_distance) {
  if (na > nb) {
    const uint32_t* t = a; a = b; b = t;
    uint32_t tn = na; na = nb; nb = tn;
  }
  if (nb - na > max_distance) return max_distance + 1;
  if (na == 0) return nb;
  uint32_t* dp = (uint32_t*)malloc((na + 1) * sizeof(uint32_t));
  for (uint32_t j = 0; j <= na; ++j) dp[j] = j;
  for (uint32_t i = 0; i < nb; ++i) {
    uint32_t prev = dp[0];
    dp[0] = i + 1;
    uint32_t row_min = dp[0];
    for (uint32_t j = 0; j < na; ++j) {
      uint32_t cost = a[j] == b[i] ? 0u : 1u;
      uint32_t ins = dp[j + 1] + 1, del = dp[j] + 1, rep = prev + cost;
      prev = dp[j + 1];
      uint32_t m = ins < del ? ins : del;
      dp[j + 1] = m < rep ? m : rep;
      if (dp[j + 1] < row_min) row_min = dp[j + 1];
    }
    if (row_min > max_distance) {
      free(dp);
      return max_distance + 1;
    }
  }
  uint32_t r = dp[na] <= max_distance ? dp[na] : max_distance + 1;
  free(dp);
  return r;
}

/* src/utils/edit_distance.cpp:129-156 ContainsFuzzyCodepointWindow */
static int contains_fuzzy_window(const uint32_t* word, uint32_t word_len, const uint32_t* term, uint32_t term_len,
                                 uint32_t max_distance) {
  uint64_t min_len = term_len > max_distance ? term_len - max_distance : 1;
  uint64_t max_len = (uint64_t)term_len + max_distance;
  if (max_len > word_len) max_len = word_len;
  if (min_len > max_len) return 0;
  for (uint32_t start = 0; start < word_len; ++start) {
    uint64_t remaining = word_len - start;
    uint64_t last = max_len < remaining ? max_len : remaining;
    for (uint64_t cl = min_len; cl <= last; ++cl)
      if (levenshtein_cp(word + start, (uint32_t)cl, term, term_len, max_distance) <= max_distance) return 1;
  }
  return 0;
}

static int all_ascii(const uint8_t* s, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (s[i] >= 0x80) return 0;
  return 1;
}

/* src/utils/edit_distance.cpp:199-296 ContainsFuzzyMatch: some whitespace-delimited word of the text (U+3000 and U+00A0
 * count as spaces) is within max_distance of the term; a non-ASCII word is also searched by term-sized code-point
 * windows, because CJK text has no spaces between words. */
int orc_contains_fuzzy_match(const uint8_t* text, size_t text_len, const uint8_t* term, size_t term_len,
                             uint32_t max_distance) {
  if (term_len == 0) return 1;
  if (text_len == 0) return 0;
  uint8_t* nt = (uint8_t*)malloc(text_len + 1); /* NormalizeUnicodeWhitespace :24-47 */
  size_t nl = 0;
  for (size_t i = 0; i < text_len;) {
    if (text[i] == 0xE3 && i + 2 < text_len && text[i + 1] == 0x80 && text[i + 2] == 0x80) {
      nt[nl++] = ' ';
      i += 3;
    } else if (text[i] == 0xC2 && i + 1 < text_len && text[i + 1] == 0xA0) {
      nt[nl++] = ' ';
      i += 2;
    } else {
      nt[nl++] = text[i++];
    }
  }
  size_t tn = 0;
  uint32_t* tcp = utf8_to_codepoints(term, term_len, &tn);
  int term_ascii = all_ascii(term, term_len);
  int found = 0;
  size_t pos = 0;
  while (!found && pos < nl) {
    while (pos < nl && (nt[pos] == ' ' || nt[pos] == '\t' || nt[pos] == '\r' || nt[pos] == '\n')) ++pos;
    if (pos >= nl) break;
    size_t end = pos;
    while (end < nl && !(nt[end] == ' ' || nt[end] == '\t' || nt[end] == '\r' || nt[end] == '\n')) ++end;
    size_t wn = 0;
    uint32_t* wcp = utf8_to_codepoints(nt + pos, end - pos, &wn);
    int word_ascii = all_ascii(nt + pos, end - pos);
    if (!word_ascii && !(term_ascii && word_ascii) &&
        contains_fuzzy_window(wcp, (uint32_t)wn, tcp, (uint32_t)tn, max_distance)) {
      found = 1;
    } else {
      uint32_t diff = wn > tn ? (uint32_t)(wn - tn) : (uint32_t)(tn - wn);
      if (diff <= max_distance && levenshtein_cp(wcp, (uint32_t)wn, tcp, (uint32_t)tn, max_distance) <= max_distance)
        found = 1;
    }
    free(wcp);
    pos = end;
  }
  free(tcp);
  free(nt);
  return found;
}

/* src/server/search_pipeline.cpp:1659-1744 ExecuteWithFuzzy (no synonyms): per term theta = |grams| - d * n_eff
 * (at least 1), n_eff = the kanji size when more than half of the term's grams are at most 3 bytes long;
 * Index::SearchByThreshold(grams, theta); AND across terms in the order given; NOT terms and filters; then — only
 * when the caller's verify_text decision says so (:1722-1732) — PostFilterByFuzzyText: every term fuzzily contained in
 * the text; then the exact-text rule for mixed-script fragments (:1733-1741). thetas[] (may be NULL) receives the
 * threshold of every term. */
int orc_execute_fuzzy(const orc_index* idx, const orc_docstore* ds, const uint8_t* tb, const uint32_t* toff,
                      size_t n_terms, uint32_t max_distance, const uint8_t* nb, const uint32_t* noff, size_t n_not,
                      const orc_filter* filters, size_t n_filters, int ngram_size, int kanji_ngram_size,
                      int cross_boundary, int verify_text, orc_pipeline_result* out, uint64_t* thetas) {
  memset(out, 0, sizeof(*out));
  if (n_terms > 64) return 11;
  u32vec results = {0};
  if (n_terms == 0) {
    out->empty_term_detected = 1;
    goto done;
  }
  term_info* tis = (term_info*)calloc(n_terms, sizeof(term_info));
  for (size_t i = 0; i < n_terms; ++i)
    make_term_info(idx, ds, tb + toff[i], toff[i + 1] - toff[i], ngram_size, kanji_ngram_size, cross_boundary, 0,
                   &tis[i]);
  int first = 1;
  for (size_t i = 0; i < n_terms; ++i) {
    const term_info* ti = &tis[i];
    if (ti->ngrams.count == 0) { /* :1676-1682 */
      free(results.v);
      results.v = NULL;
      results.n = 0;
      out->empty_term_detected = 1;
      first = 0;
      break;
    }
    int n_eff = ngram_size > 0 ? ngram_size : 2; /* :1685-1698 */
    if (kanji_ngram_size > 0) {
      size_t short_count = 0;
      for (size_t g = 0; g < ti->ngrams.count; ++g)
        if (ti->ngrams.off[g + 1] - ti->ngrams.off[g] <= 3) ++short_count;
      if (short_count > ti->ngrams.count / 2) n_eff = kanji_ngram_size;
    }
    size_t drop = (size_t)max_distance * (size_t)n_eff; /* :1700-1703 */
    size_t theta = ti->ngrams.count > drop ? ti->ngrams.count - drop : 1;
    if (thetas) thetas[i] = theta;
    u32vec tr = {0};
    tr.v = orc_search_by_threshold(idx, ti->ngrams.bytes, ti->ngrams.off, ti->ngrams.count, theta, &tr.n);
    if (first) { /* IntersectSorted(results, term_results, first_term) :421-436 */
      out->total_candidates = tr.n;
      results = tr;
      first = 0;
    } else {
      u32vec inter = set_intersection_u32(results.v, results.n, tr.v, tr.n);
      free(tr.v);
      free(results.v);
      results = inter;
    }
  }
  out->after_intersection = results.n;
  /* ApplyNotAndFilters */
  if (n_not > 0 && results.n > 0) {
    u32vec excluded = {0};
    for (size_t i = 0; i < n_not; ++i) {
      term_info nti;
      make_term_info(idx, ds, nb + noff[i], noff[i + 1] - noff[i], ngram_size, kanji_ngram_size, cross_boundary, 0,
                     &nti);
      u32vec td = search_term_documents(idx, ds, &nti);
      u32vec u = set_union_u32(excluded.v, excluded.n, td.v, td.n);
      free(td.v);
      free(excluded.v);
      excluded = u;
      free_term_info(&nti);
    }
    if (excluded.n > 0) {
      u32vec d = set_difference_u32(results.v, results.n, excluded.v, excluded.n);
      free(results.v);
      results = d;
    }
    free(excluded.v);
  }
  out->after_not = results.n;
  for (size_t f = 0; f < n_filters; ++f) {
    u32vec r2 = filters[f].negate ? set_difference_u32(results.v, results.n, filters[f].docs, filters[f].n_docs)
                                  : set_intersection_u32(results.v, results.n, filters[f].docs, filters[f].n_docs);
    free(results.v);
    results = r2;
  }
  out->after_filters = results.n;
  if (verify_text && results.n > 0 && ds != NULL) { /* PostFilterByFuzzyText :1746-1757 */
    size_t w = 0;
    for (size_t r = 0; r < results.n; ++r) {
      size_t tl;
      const uint8_t* tx = docstore_text(ds, results.v[r], &tl);
      int all = tx != NULL;
      for (size_t i = 0; all && i < n_terms; ++i)
        all = orc_contains_fuzzy_match(tx, tl, tis[i].normalized, tis[i].normalized_len, max_distance);
      if (all) results.v[w++] = results.v[r];
    }
    results.n = w;
  }
  {
    int exact = 0; /* RequiresExactTextForHybridFragments :1733-1741 */
    for (size_t i = 0; i < n_terms; ++i)
      exact = exact || orc_has_uncovered_hybrid_fragment(tis[i].normalized, tis[i].normalized_len, ngram_size,
                                                         kanji_ngram_size, cross_boundary);
    out->exact_text_applied = exact;
    if (exact && results.n > 0 && ds != NULL) {
      size_t w = 0;
      for (size_t r = 0; r < results.n; ++r) {
        size_t tl;
        const uint8_t* tx = docstore_text(ds, results.v[r], &tl);
        int all = tx != NULL;
        for (size_t i = 0; all && i < n_terms; ++i)
          all = bytes_find(tx, tl, tis[i].normalized, tis[i].normalized_len, 0) != (size_t)-1;
        if (all) results.v[w++] = results.v[r];
      }
      results.n = w;
    }
  }
  for (size_t i = 0; i < n_terms; ++i) free_term_info(&tis[i]);
  free(tis);
done:
  out->results = results.v ? results.v : u32_dup(NULL, 0);
  out->n_results = results.n;
  out->n_terms = n_terms;
  return 0;
}

/* PostFilterByText, src/server/search_pipeline.cpp:1239-1246: candidates whose stored text contains every
 * (already normalized) term; order kept. */
uint32_t* orc_post_filter_by_text(const orc_docstore* ds, const uint32_t* cand, size_t nc, const uint8_t* tb,
                                  const uint32_t* toff, size_t n_terms, size_t* out_n) {
  uint32_t* out = (uint32_t*)malloc((nc ? nc : 1) * sizeof(uint32_t));
  size_t w = 0;
  for (size_t c = 0; c < nc; ++c) {
    size_t tl;
    const uint8_t* tx = docstore_text(ds, cand[c], &tl);
    int all = tx != NULL;
    for (size_t i = 0; all && i < n_terms; ++i)
      all = bytes_find(tx, tl, tb + toff[i], toff[i + 1] - toff[i], 0) != (size_t)-1;
    if (all) out[w++] = cand[c];
  }
  *out_n = w;
  return out;
}

/* src/server/handlers/search_handler.cpp:405-470 on top of ExecuteFullPipeline's regular branch
 * (src/server/search_pipeline.cpp:2002-2030). */
int orc_search_scored(const orc_index* idx, const orc_docstore* ds, const uint8_t* tb, const uint32_t* toff,
                      size_t n_terms, int ngram_size, int kanji_ngram_size, int cross_boundary, size_t filter_threshold,
                      uint64_t total_docs, double avg_doc_length, double k1, double b, int descending, uint32_t limit,
                      uint32_t offset, uint32_t* top_docs, double* top_scores, size_t* n_top, uint64_t* total) {
  orc_pipeline_result pr;
  int rc = orc_execute(idx, ds, tb, toff, n_terms, NULL, NULL, 0, NULL, 0, ngram_size, kanji_ngram_size,
                       cross_boundary, filter_threshold, 1, 0, &pr);
  if (rc != 0) return rc;
  *total = pr.n_results;
  *n_top = 0;
  /* normalized terms in sorted order + their dfs (search_handler.cpp:440-452) */
  orc_strlist terms;
  memset(&terms, 0, sizeof(terms));
  uint8_t tmp[512];
  for (size_t i = 0; i < pr.n_terms; ++i) {
    uint32_t s = pr.term_order[i];
    size_t len = toff[s + 1] - toff[s];
    uint8_t* nbuf = len <= sizeof(tmp) ? tmp : (uint8_t*)malloc(len);
    orc_normalize_ascii_lower(tb + toff[s], len, nbuf);
    strlist_push(&terms, nbuf, len);
    if (nbuf != tmp) free(nbuf);
  }
  double* scores = (double*)malloc((pr.n_results ? pr.n_results : 1) * sizeof(double));
  rc = orc_score_documents(ds, pr.results, pr.n_results, terms.bytes, terms.off ? terms.off : (const uint32_t*)"\0\0\0\0",
                           pr.n_terms, pr.term_df, pr.n_terms, total_docs, avg_doc_length, k1, b, scores);
  if (rc == 0) {
    size_t ns = 0;
    uint32_t* sorted = orc_sort_by_score(pr.results, scores, pr.n_results, descending, limit, offset, &ns);
    for (size_t i = 0; i < ns; ++i) {
      top_docs[i] = sorted[i];
      /* score of that doc: results are ascending, binary search */
      size_t lo = 0, hi = pr.n_results;
      while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (pr.results[mid] < sorted[i]) lo = mid + 1; else hi = mid;
      }
      top_scores[i] = scores[lo];
    }
    *n_top = ns;
    free(sorted);
  }
  free(scores);
  orc_strlist_free(&terms);
  orc_pipeline_result_free(&pr);
  return rc;
}
