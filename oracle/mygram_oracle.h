/*
 * mygram_oracle.h — CPU restatement of MygramDB's query hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X build. It is a from-scratch plain-C restatement of the
 * reference algorithms, written from the reference's behaviour (files cited per function below, paths
 * relative to /root/reference). It is NOT part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it. The product library (libmygram_gpu.so) never links,
 * loads or calls anything in oracle/.
 *
 * Pinning: the reference's own hot-path translation units cannot be compiled in this image (they need
 * CRoaring 4.6.1, abseil and spdlog headers, none of which are present, and stand-ins are not allowed),
 * so this oracle is pinned by the reference's own known-answer tests, transcribed as data into
 * tests/golden/ (JSON files) (tests/index/index_search_test.cpp, search_by_threshold_test.cpp,
 * index_gettopn_test.cpp, bm25_scorer_test.cpp, tests/query/bm25_sort_test.cpp,
 * tests/server/search_pipeline_test.cpp). See tests/test_oracle_golden.py.
 *
 * Dense-list arithmetic in the reference goes through CRoaring v4.6.1 (third_party/CMakeLists.txt:105-112,
 * not vendored). Every CRoaring call on the path has pure sorted-unique-u32-set semantics
 * (and/or/andnot/contains/to_uint32_array/iterators), which is what is restated here.
 */
#ifndef MYGRAM_ORACLE_H_
#define MYGRAM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- text primitives -------------------------------------------------------------------------- */

/* src/utils/string_utils.cpp:94-164 TryParseUtf8Char: bytes consumed (1..4) or -1. */
int orc_try_parse_utf8(const uint8_t* data, size_t available, uint32_t* out_cp);
/* src/utils/string_utils.cpp:655-669 CountCodePoints (invalid bytes skipped, not counted). */
size_t orc_count_code_points(const uint8_t* text, size_t len);
/* src/index/bm25_scorer.cpp:27-45 CountTermOccurrences (non-overlapping, left-greedy byte substring). */
uint32_t orc_count_term_occurrences(const uint8_t* text, size_t text_len, const uint8_t* term, size_t term_len);
/* src/index/bm25_scorer.cpp:14-25 ComputeIDF. */
double orc_compute_idf(uint64_t total_docs, uint64_t doc_freq);
/* ASCII-only restatement of Index::NormalizeText with normalize_lower=true (src/utils/string_utils.cpp:362-380,
 * non-ICU branch: std::tolower per byte). Non-ASCII bytes pass through unchanged; ICU NFKC/width folding is
 * NOT restated (inputs to the oracle must already be NFKC-normalised). Writes len bytes to out. */
void orc_normalize_ascii_lower(const uint8_t* text, size_t len, uint8_t* out);

/* A list of byte strings held in one arena. */
typedef struct orc_strlist {
  uint8_t* bytes;    /* concatenated gram bytes */
  uint32_t* off;     /* count+1 offsets into bytes */
  size_t count;
  size_t cap_bytes, cap_count;
} orc_strlist;
void orc_strlist_free(orc_strlist* l);

/* src/utils/string_utils.cpp:382-423 GenerateNgrams (code-point windows of n). */
void orc_generate_ngrams(const uint8_t* text, size_t len, int n, orc_strlist* out);
/* src/utils/string_utils.cpp:452-509 GenerateHybridNgrams (window size by class of the STARTING code point;
 * kana is not CJK; cross_boundary=0 drops windows mixing classes). */
void orc_generate_hybrid_ngrams(const uint8_t* text, size_t len, int ascii_n, int kanji_n, int cross_boundary,
                                orc_strlist* out);
/* src/utils/string_utils.cpp:639-653 GenerateQueryNgrams. */
void orc_generate_query_ngrams(const uint8_t* text, size_t len, int ngram_size, int kanji_ngram_size,
                               int cross_boundary, orc_strlist* out);
/* src/utils/string_utils.h:192-196 DeduplicateSorted (bytewise sort + unique), in place. */
void orc_strlist_dedup_sorted(orc_strlist* l);

/* ---- index ------------------------------------------------------------------------------------ */

typedef struct orc_index orc_index;

/* src/index/index.cpp:28-36 Index::Index (kanji_ngram_size<=0 => ngram_size). */
orc_index* orc_index_create(int ngram_size, int kanji_ngram_size, int cross_boundary);
void orc_index_destroy(orc_index* idx);
/* src/index/index.cpp:39-74 Index::AddDocument: hybrid n-grams, dedupe, append docid to each list.
 * Returns 1 if indexed, 0 if the text produced no n-grams. */
int orc_index_add_document(orc_index* idx, uint32_t doc_id, const uint8_t* text, size_t len);
/* Adopt external CSR posting arrays (not copied; caller keeps them alive). keys: n_grams byte strings given by
 * key_bytes/key_off, sorted bytewise ascending; lists: offsets[n_grams+1], docids ascending per list. Used only at
 * sizes where rebuilding through orc_index_add_document would take minutes. */
orc_index* orc_index_from_csr(int ngram_size, int kanji_ngram_size, int cross_boundary, size_t n_grams,
                              const uint8_t* key_bytes, const uint32_t* key_off, const uint64_t* offsets,
                              const uint32_t* docids);
/* src/index/index.cpp:580-584,756-759 PostingSize / EstimatePostingSize (0 for an unknown gram). */
uint64_t orc_index_posting_size(const orc_index* idx, const uint8_t* gram, size_t len);
size_t orc_index_gram_count(const orc_index* idx);

/* Result vectors are malloc'd; free with orc_free. *out_n receives the length. */
void orc_free(void* p);

/* `terms` = n_terms byte strings: term i is term_bytes[term_off[i] .. term_off[i+1]). */
/* src/index/index.cpp:199-368 Index::SearchAnd. */
uint32_t* orc_search_and(const orc_index* idx, const uint8_t* term_bytes, const uint32_t* term_off, size_t n_terms,
                         size_t limit, int reverse, size_t* out_n);
/* src/index/index.cpp:418-448 Index::SearchOr. */
uint32_t* orc_search_or(const orc_index* idx, const uint8_t* term_bytes, const uint32_t* term_off, size_t n_terms,
                        size_t* out_n);
/* src/index/index.cpp:450-486 Index::SearchNot. */
uint32_t* orc_search_not(const orc_index* idx, const uint32_t* all_docs, size_t n_all, const uint8_t* term_bytes,
                         const uint32_t* term_off, size_t n_terms, size_t* out_n);
/* src/index/index.cpp:488-578 Index::SearchByThreshold. */
uint32_t* orc_search_by_threshold(const orc_index* idx, const uint8_t* term_bytes, const uint32_t* term_off,
                                  size_t n_terms, size_t threshold, size_t* out_n);
/* src/index/index.cpp:370-416 Index::FilterByNgrams (+ PostingList::RetainPresent posting_list.cpp:432-474). */
uint32_t* orc_filter_by_ngrams(const orc_index* idx, const uint32_t* candidates, size_t n_cand,
                               const uint8_t* term_bytes, const uint32_t* term_off, size_t n_terms, size_t* out_n);

/* ---- document store (text side of BM25) ------------------------------------------------------- */

typedef struct orc_docstore orc_docstore;
orc_docstore* orc_docstore_create(void);
void orc_docstore_destroy(orc_docstore* ds);
/* Store normalized text for doc_id (has_text=0 models a document stored without text). */
void orc_docstore_add(orc_docstore* ds, uint32_t doc_id, const uint8_t* text, size_t len, int has_text);
/* Adopt external text arrays for docids 1..n_docs: doc d's text is text_bytes[text_off[d-1] .. text_off[d]). */
orc_docstore* orc_docstore_from_arrays(size_t n_docs, const uint8_t* text_bytes, const uint64_t* text_off);
/* src/server/server_types.h:157-193 BM25Stats: docs with non-empty text, and their summed code-point length. */
void orc_docstore_bm25_stats(const orc_docstore* ds, uint64_t* doc_count, uint64_t* total_len);

/* src/index/bm25_scorer.cpp:47-99 BM25Scorer::ScoreDocuments. scores_out[n_cand]. Returns 0, or 11 (kInvalidArgument
 * in spirit) when n_terms != n_dfs. */
int orc_score_documents(const orc_docstore* ds, const uint32_t* candidates, size_t n_cand, const uint8_t* term_bytes,
                        const uint32_t* term_off, size_t n_terms, const uint64_t* dfs, size_t n_dfs,
                        uint64_t total_docs, double avg_doc_length, double k1, double b, double* scores_out);

/* src/query/result_sorter.cpp:661-716 ResultSorter::SortByScore. descending!=0 => SortOrder::DESC. */
uint32_t* orc_sort_by_score(const uint32_t* results, const double* scores, size_t n, int descending, uint32_t limit,
                            uint32_t offset, size_t* out_n);

/* ---- conjunctive pipeline --------------------------------------------------------------------- */

typedef struct orc_pipeline_result {
  uint32_t* results; /* malloc'd, ascending docids after AND/NOT/FILTER */
  size_t n_results;
  size_t total_candidates, after_intersection, after_not, after_filters;
  int empty_term_detected;
  /* per search term, in the order the pipeline sorted them (estimated_size ascending): */
  size_t n_terms;
  uint32_t term_order[64]; /* index into the caller's term array */
  uint64_t term_df[64];    /* text-verified document frequency (0 when not computed) */
  uint64_t term_estimated_size[64];
  int exact_text_applied; /* search_pipeline.cpp:858-866 fired: results were filtered by the terms' exact text */
} orc_pipeline_result;

/* One column filter already resolved to a sorted docid set (FilterIndex bitmap, src/storage/filter_index.h:39-124):
 * negate=0 => EQ (results AND set), negate=1 => NE (results ANDNOT set). src/server/search_pipeline.cpp:1196-1237. */
typedef struct orc_filter {
  const uint32_t* docs;
  size_t n_docs;
  int negate;
} orc_filter;

/* src/server/search_pipeline.cpp:569-603 GenerateTermInfos (+ :542-565 PopulateTermDocumentFrequency when
 * compute_df and ds != NULL), :2012-2014 sort by estimated_size, :795-869 Execute, :871-932 ApplyNotFilter.
 * Terms are raw search terms (normalised here with orc_normalize_ascii_lower). filter_threshold is
 * SearchHandler::filter_threshold_ (1000). verify_text = the caller's ShouldApplyVerifyText decision (:42-66). */
/* src/server/search_pipeline.cpp:80-136 */
int orc_has_uncovered_hybrid_fragment(const uint8_t* term, size_t len, int ngram_size, int kanji_ngram_size,
                                      int cross_boundary);
int orc_execute(const orc_index* idx, const orc_docstore* ds, const uint8_t* term_bytes, const uint32_t* term_off,
                size_t n_terms, const uint8_t* not_bytes, const uint32_t* not_off, size_t n_not,
                const orc_filter* filters, size_t n_filters, int ngram_size, int kanji_ngram_size, int cross_boundary,
                size_t filter_threshold, int compute_df, int verify_text, orc_pipeline_result* out);
/* src/utils/edit_distance.cpp:199-296 ContainsFuzzyMatch. */
int orc_contains_fuzzy_match(const uint8_t* text, size_t text_len, const uint8_t* term, size_t term_len,
                             uint32_t max_distance);
/* src/server/search_pipeline.cpp:1659-1744 ExecuteWithFuzzy: theta / n_eff per term, SearchByThreshold, AND across terms
 * in the order given, NOT terms, filters, PostFilterByFuzzyText when verify_text (the caller's ShouldApplyVerifyText
 * decision). thetas (may be NULL): the threshold of every term. */
int orc_execute_fuzzy(const orc_index* idx, const orc_docstore* ds, const uint8_t* term_bytes, const uint32_t* term_off,
                      size_t n_terms, uint32_t max_distance, const uint8_t* not_bytes, const uint32_t* not_off,
                      size_t n_not, const orc_filter* filters, size_t n_filters, int ngram_size, int kanji_ngram_size,
                      int cross_boundary, int verify_text, orc_pipeline_result* out, uint64_t* thetas);
/* PostFilterByText, src/server/search_pipeline.cpp:1239-1246 (terms already normalized); caller frees the result. */
uint32_t* orc_post_filter_by_text(const orc_docstore* ds, const uint32_t* cand, size_t n_cand,
                                  const uint8_t* term_bytes, const uint32_t* term_off, size_t n_terms, size_t* out_n);
void orc_pipeline_result_free(orc_pipeline_result* r);

/* The whole SEARCH ... SORT _score DESC LIMIT k path for one query: orc_execute(compute_df=1), then
 * ScoreDocuments with the sorted term order and SortByScore (src/server/handlers/search_handler.cpp:405-470).
 * top_docs/top_scores receive up to `limit` entries; *total receives results.size(). */
int orc_search_scored(const orc_index* idx, const orc_docstore* ds, const uint8_t* term_bytes, const uint32_t* term_off,
                      size_t n_terms, int ngram_size, int kanji_ngram_size, int cross_boundary, size_t filter_threshold,
                      uint64_t total_docs, double avg_doc_length, double k1, double b, int descending, uint32_t limit,
                      uint32_t offset, uint32_t* top_docs, double* top_scores, size_t* n_top, uint64_t* total);

#ifdef __cplusplus
}
#endif
#endif /* MYGRAM_ORACLE_H_ */
