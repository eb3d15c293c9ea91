/*
 * mygram_gpu.h — C ABI of libmygram_gpu.so: MygramDB's in-memory query hot path on AMD MI355X (gfx950).
 *
 * This is the ONLY boundary HIP code is reached through. There is no FFI around this path in the reference; the
 * seam it replaces is the set of public C++ methods below (paths relative to the reference tree). The C++17 shim in
 * mygram-db_amd/csrc/shim/ re-exposes them with the reference's exact signatures on top of this ABI.
 *
 *   Index::SearchAnd            src/index/index.h:127-128  (impl src/index/index.cpp:199-368)   -> mgx_and / mgx_batch_*
 *   Index::FilterByNgrams       src/index/index.h:138-139  (impl index.cpp:370-416)             -> mgx_retain
 *   Index::SearchOr             src/index/index.h:147      (impl index.cpp:418-448)             -> mgx_or
 *   Index::SearchNot            src/index/index.h:156-157  (impl index.cpp:450-486)             -> mgx_not
 *   Index::SearchByThreshold    src/index/index.h:172      (impl index.cpp:488-578)             -> mgx_threshold
 *   Index::EstimatePostingSize  src/index/index.h:286      (impl index.cpp:756-759)             -> mgx_posting_size
 *   BM25Scorer::ScoreDocuments  src/index/bm25_scorer.h:79-82 (impl bm25_scorer.cpp:47-99)      -> mgx_score_documents
 *   ResultSorter::SortByScore   src/query/result_sorter.h:75-76 (impl result_sorter.cpp:661-716)-> mgx_sort_by_score
 *   search_pipeline::Execute    src/server/search_pipeline.h:121-124 (impl search_pipeline.cpp:795-869),
 *     + BM25 glue src/server/handlers/search_handler.cpp:405-470                                -> mgx_batch_* (batched; new)
 *
 * House style follows the reference's own C API (src/client/mygramclient_c.h:5-16,49-60,186-223): every function
 * returns 0 on success and a non-zero mygram::utils::ErrorCode value (src/utils/error.h:35+) on failure;
 * mgx_last_error() returns a thread-local message; handles are opaque; descriptor structs carry
 * struct_size/version; library-allocated outputs are released with the matching free; on failure pointer
 * out-parameters are NULL and numeric ones zero; no C++ exception crosses this ABI.
 *
 * There is NO CPU fallback behind these entry points: if no gfx950 device is usable they fail with
 * MGX_ERR_INTERNAL and a message. Document ids are uint32 (src/types/doc_id.h:31); result lists are ascending and
 * unique unless stated otherwise.
 */
#ifndef MYGRAM_GPU_H_
#define MYGRAM_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_ABI_VERSION 3U

/* Values of mygram::utils::ErrorCode used on this path (src/utils/error.h:37-48,100). */
#define MGX_OK 0
#define MGX_ERR_INVALID_ARGUMENT 2
#define MGX_ERR_OUT_OF_RANGE 3
#define MGX_ERR_NOT_IMPLEMENTED 4
#define MGX_ERR_INTERNAL 5
#define MGX_ERR_INDEX_NOT_FOUND 4000

int mgx_abi_version(void);
/* Thread-local message of the last failing call on this thread; valid until that thread's next failing call. */
const char* mgx_last_error(void);
/* Releases any buffer this library returned through an out-pointer (docid/score arrays). NULL is allowed. */
void mgx_free(void* p);
/* Number of usable gfx950 devices (0 when there is none; never fails). */
int mgx_device_count(void);

/* =====================================================================================================
 * Host-side column builder: normalized texts -> the column arrays the device index is made of.
 * Restates what Index::AddDocument (index.cpp:39-74) + BM25 ingest bookkeeping
 * (src/mysql/binlog_event_processor.cpp:98-99) produce, plus the build-owned tf column that makes
 * BM25Scorer::CountTermOccurrences (bm25_scorer.cpp:27-45) a lookup for terms that are exactly one n-gram long.
 * Pure host code (multi-threaded); no device needed.
 * ===================================================================================================== */

typedef struct mgx_columns mgx_columns;

typedef struct mgx_build_params {
  uint32_t struct_size; /* sizeof(mgx_build_params) */
  uint32_t version;     /* MGX_ABI_VERSION */
  int32_t ngram_size;           /* Index(ngram_size, ...) */
  int32_t kanji_ngram_size;     /* <=0 => ngram_size (index.cpp:31) */
  int32_t cross_boundary_ngrams;
  int32_t n_threads;            /* 0 => hardware concurrency */
} mgx_build_params;

typedef struct mgx_columns_view {
  uint64_t n_grams;
  const uint8_t* key_bytes;  /* gram keys, concatenated, sorted bytewise ascending; gram id = rank */
  const uint32_t* key_off;   /* n_grams+1 */
  const uint64_t* offsets;   /* n_grams+1, CSR into docids/tf */
  const uint32_t* docids;    /* ascending per gram */
  const uint8_t* tf;         /* non-overlapping occurrence count of the gram's bytes in the doc text; 255 = "255 or more,
                              * see tf_overflow" */
  uint64_t n_postings;
  uint32_t first_doc_id;     /* doc d has local slot d - first_doc_id */
  uint64_t n_docs;           /* slots */
  const uint32_t* doc_len;   /* n_docs: CountCodePoints(text) (string_utils.cpp:655-669); 0 for empty text */
  uint64_t bm25_doc_count;   /* docs with non-empty text (server_types.h:157-193) */
  uint64_t bm25_total_len;   /* sum of their doc_len */
  /* postings whose count does not fit the byte column: their index into docids/tf (ascending) and the true count, so
   * that BM25 sees exactly what CountTermOccurrences (bm25_scorer.cpp:27-45) returns for any document */
  const uint64_t* tf_overflow_pos;
  const uint32_t* tf_overflow_val;
  uint64_t n_tf_overflow;
} mgx_columns_view;

/* Docs are first_doc_id, first_doc_id+1, ...; doc i's normalized text is text_bytes[text_off[i] .. text_off[i+1]). */
int mgx_columns_build(const mgx_build_params* params, const uint8_t* text_bytes, const uint64_t* text_off,
                      uint32_t first_doc_id, uint64_t n_docs, mgx_columns** out);
/* The reference's index dump ("MGIX" v1..v4: Index::SaveToStream / LoadFromData, src/index/index_serialization.cpp:113-194,
 * :260-420; posting lists as fixed-width deltas or Roaring portable bytes, src/index/posting_list.cpp:973-1073) ->
 * columns. The CRC32 trailer (v2+) is verified. A dump holds doc ids only: the view's tf / doc_len are NULL (an index
 * created from it answers everything but SORT _score). n_docs == 0: the doc range is the span of the ids found. */
typedef struct mgx_mgix_info {
  uint32_t version;
  int32_t ngram_size, kanji_ngram_size, cross_boundary_ngrams;
  int32_t normalize_nfkc, normalize_lower;
  char normalize_width[16];
  uint64_t n_terms;
} mgx_mgix_info;
int mgx_columns_from_mgix(const uint8_t* data, uint64_t len, uint32_t first_doc_id, uint64_t n_docs, mgx_columns** out,
                          mgx_mgix_info* info /* may be NULL */);
/* The reference's table dump ("MGDB" version 2: what DUMP SAVE writes, src/storage/dump_format_v2.cpp:520-770) -> one
 * table's columns, normalized texts and filter values: the MGIX index stream (as mgx_columns_from_mgix) AND the document
 * store stream ("MGDS", src/storage/document_store_persistence.cpp:59-175). File and section CRC32s are verified. When the
 * store kept the normalized texts (memory.verify_text on: what BM25 needs in the reference too) the columns are built from
 * them — tf and doc lengths included, so the loaded index ranks — and must agree with the dump's own index gram by gram and
 * list by list; otherwise the columns hold the index's doc ids alone. `table`: NULL or "" = the first table of the dump. */
typedef struct mgx_dump mgx_dump;
typedef struct mgx_dump_view {
  const char* table_name;
  mgx_mgix_info index_info;
  uint32_t first_doc_id; /* the range spans every doc id the store or the index mentions */
  uint64_t n_docs;
  uint64_t n_existing;   /* documents in the store (DocumentStore::Size) */
  const uint8_t* exists; /* n_docs: 1 = the store holds this id (DocumentStore::GetAllDocIds, the NOT universe) */
  const uint8_t* text_bytes; /* normalized texts by slot (mgx_index_attach_text takes them as they are) */
  const uint64_t* text_off;  /* n_docs + 1 */
  uint32_t n_filter_columns;
  int32_t has_texts;
} mgx_dump_view;
typedef struct mgx_dump_filter_column {
  const char* name;
  uint32_t value_type;   /* the storage::FilterValue alternative (src/storage/document_store.h:73-87): 1 bool .. 12 double */
  const uint64_t* values;     /* n_docs, widened as mgx_filter_column_desc wants them (strings: unused) */
  const uint8_t* is_null;     /* n_docs */
  const uint8_t* string_bytes; /* value_type 11: doc i's string is string_bytes[string_off[i] .. string_off[i+1]) */
  const uint64_t* string_off;
} mgx_dump_filter_column;
int mgx_dump_open(const uint8_t* data, uint64_t len, const char* table, mgx_dump** out);
int mgx_dump_view_get(const mgx_dump* dump, mgx_dump_view* out);
int mgx_dump_filter_column_get(const mgx_dump* dump, uint32_t i, mgx_dump_filter_column* out);
/* Hands the table's columns to the caller (release with mgx_columns_destroy); a second call returns NULL. */
int mgx_dump_take_columns(mgx_dump* dump, mgx_columns** out);
void mgx_dump_destroy(mgx_dump* dump);
int mgx_columns_view_get(const mgx_columns* cols, mgx_columns_view* out);
/* Gram dictionary lookup (host-side replacement of Index::TakePostingSnapshot's map lookup, index.cpp:728-747).
 * *found = 0 and *gram_id = 0 for an unknown gram. */
int mgx_columns_lookup(const mgx_columns* cols, const uint8_t* gram, size_t len, uint32_t* gram_id, int* found);
void mgx_columns_destroy(mgx_columns* cols);

/* =====================================================================================================
 * Device index
 * ===================================================================================================== */

typedef struct mgx_index mgx_index;

typedef struct mgx_index_desc {
  uint32_t struct_size; /* sizeof(mgx_index_desc) */
  uint32_t version;     /* MGX_ABI_VERSION */
  int32_t device;       /* HIP device ordinal */
  uint32_t tile_shift;  /* log2 docids per tile; 0 => default (14) */
  /* the doc-id range this index (or shard) owns: local slot = doc_id - first_doc_id, 0 <= slot < n_docs */
  uint32_t first_doc_id;
  uint64_t n_docs;
  uint64_t n_grams;
  const uint64_t* offsets; /* host, n_grams+1 */
  const uint32_t* docids;  /* host, offsets[n_grams]; every id inside the owned range, ascending per gram */
  const uint8_t* tf;       /* host, parallel to docids; NULL => BM25 scoring unavailable */
  const uint32_t* doc_len; /* host, n_docs; NULL => BM25 scoring unavailable */
  /* posting lists at least this dense (|L| / n_docs) are ALSO kept as precomputed bitmaps in HBM and read in that
   * form by the set-algebra kernels (what Roaring bitset containers are to the reference, posting_list.cpp:800-834).
   * 0 => default (1/256: at most 8x the bytes of the u32 list, bought for latency: bitmap operands need no scatter
   * and BM25 runs on the wave-autonomous kernel); >= 2 => disabled. */
  double dense_threshold;
  /* side table of the tf bytes that read 255 (mgx_columns_view.tf_overflow_*): posting indices ascending; may be
   * NULL / 0 (then 255 is taken literally) */
  const uint64_t* tf_overflow_pos;
  const uint32_t* tf_overflow_val;
  uint64_t n_tf_overflow;
} mgx_index_desc;

/* Host arrays are copied; the caller keeps ownership. */
int mgx_index_create(const mgx_index_desc* desc, mgx_index** out);
void mgx_index_destroy(mgx_index* idx);
int mgx_posting_size(const mgx_index* idx, uint32_t gram_id, uint64_t* out);
/* Device bytes held by the index (postings, tf, doc_len, skip tables, dense bitmaps). */
int mgx_index_memory_bytes(const mgx_index* idx, uint64_t* out);
/* Registers one FilterIndex (column,value) doc set (src/storage/filter_index.h:39-124) as a device bitmap usable
 * as a filter operand; docids ascending, inside the owned range. */
int mgx_index_add_filter_bitmap(mgx_index* idx, const uint32_t* docids, uint64_t n, uint32_t* out_bitmap_id);
/* Typed filter columns (DocumentStore filter values, src/storage/document_store.h:73-87): what FILTER conditions other
 * than a caller-resolved bitmap, and FACET, are evaluated on. Values by local doc slot, widened to 8 bytes:
 * MGX_FC_SIGNED (bool, int8..int64, TimeValue seconds) as int64, MGX_FC_UNSIGNED (uint8..uint64; strings as their rank in
 * the caller's bytewise-sorted dictionary) as uint64, MGX_FC_DOUBLE as IEEE bits. is_null (may be NULL: none) marks NULLs.
 * value_ids (may be NULL: FACET unavailable on this column): the dense id < n_values of every doc's value, 0xFFFFFFFF
 * for NULL — FilterIndex keeps one bitmap per distinct value (filter_index.h:39-124); the ids are their device form. */
#define MGX_FC_SIGNED 0u
#define MGX_FC_UNSIGNED 1u
#define MGX_FC_DOUBLE 2u
typedef struct mgx_filter_column_desc {
  uint32_t struct_size; /* sizeof(mgx_filter_column_desc) */
  uint32_t version;     /* MGX_ABI_VERSION */
  uint32_t value_class; /* MGX_FC_* */
  uint32_t n_values;    /* distinct non-NULL values (value_ids) */
  const void* values;   /* n_docs x 8 bytes */
  const uint8_t* is_null;
  const uint32_t* value_ids;
} mgx_filter_column_desc;
int mgx_index_add_filter_column(mgx_index* idx, const mgx_filter_column_desc* desc, uint32_t* out_column_id);
/* One FilterCondition (src/query/query_parser.h:123-127) on a typed column -> a filter bitmap usable like the ones of
 * mgx_index_add_filter_bitmap: the docs whose stored value compares true against the literal (given in the column's
 * class: literal_bits = the int64 / uint64 / double bit pattern). op: MGX_CMP_*. eq_epsilon > 0: = and != on doubles
 * test |stored - literal| against it (CompareDoubleValues, src/utils/comparison_utils.h:56-70: the per-document path of
 * ApplyFilters, search_pipeline.cpp:1098-1194); 0: exact (the FilterIndex bitmap path compares serialized keys,
 * :1021-1094). null_matches: NULL docs are members (the per-document path lets NULL pass != only, :1151-1157).
 * never_matches: the literal has no interpretation in the column's type — no non-NULL doc matches (:1167,1172,1177,1182). */
#define MGX_CMP_EQ 0u
#define MGX_CMP_NE 1u
#define MGX_CMP_LT 2u
#define MGX_CMP_LE 3u
#define MGX_CMP_GT 4u
#define MGX_CMP_GE 5u
int mgx_index_filter_compare(mgx_index* idx, uint32_t column_id, uint32_t op, uint64_t literal_bits, double eq_epsilon,
                             int null_matches, int never_matches, uint32_t* out_bitmap_id);
/* How batches that are in flight at the same time (one stream per batch object) share the device.
 * MGX_ORDER_FIFO (default): the main kernels of a batch start when those of the batch enqueued before it on this index
 * have finished — first in, first out, so a batch's latency is its own kernel time plus what was queued ahead of it;
 * uploads, clears, merges and result copies still overlap the previous batch. MGX_ORDER_CONCURRENT: no ordering — the
 * device time-slices all batches in flight (a few percent more throughput at four batches in flight, every batch
 * finishing after roughly depth x kernel time). */
#define MGX_ORDER_FIFO 0u
#define MGX_ORDER_CONCURRENT 1u
int mgx_index_set_batch_order(mgx_index* idx, uint32_t order);
/* Keeps the NORMALIZED text of the shard's docs resident in HBM (what DocumentStore::VisitNormalizedTextsFor hands
 * to BM25Scorer::ScoreDocuments, document_store_retrieval.cpp:289-322), by local slot: doc first_doc_id + i is
 * text_bytes[text_off[i] .. text_off[i+1]). Needed by text-level scored terms (mgx_term.text). Host arrays are copied. */
int mgx_index_attach_text(mgx_index* idx, const uint8_t* text_bytes, const uint64_t* text_off);

/* =====================================================================================================
 * Batched search (new: the reference executes one query per request, request_dispatcher.cpp:93-193)
 * ===================================================================================================== */

#define MGX_SORT_DOCID 0u /* no SORT _score: docid order; `reverse` selects descending (SearchAnd limit/reverse) */
#define MGX_SORT_SCORE 1u /* SORT _score: BM25 + ResultSorter::SortByScore */

#define MGX_MAX_TERMS 64u /* query_parser.h:270-272 */
/* Gram id of an n-gram the TABLE knows but this doc-range shard holds no posting for: an empty doc set. (A gram the
 * whole table lacks never reaches the device, see mgx_term.) Keeps the batch layout identical on every shard. */
#define MGX_GRAM_ABSENT 0xFFFFFFFFu

/* One search term = the AND of its (deduplicated) n-grams, as search_pipeline::GenerateTermInfos leaves it
 * (search_pipeline.cpp:569-603). Unknown grams must be resolved by the caller: a positive term with an unknown
 * gram makes the whole query empty before it reaches the device (Execute :804-810), a NOT term with one is dropped. */
typedef struct mgx_term {
  const uint32_t* gram_ids;
  uint32_t n_grams;
  /* FUZZY terms (search_pipeline.cpp:1697-1702): docs in at least `threshold` of the distinct grams;
   * 0 => plain AND of all grams. */
  uint32_t threshold;
  double idf; /* BM25Scorer::ComputeIDF(N, df) for MGX_SORT_SCORE when the term is exactly one n-gram (text == NULL):
               * its tf is the gram's tf column, its df the posting count */
  /* Text-level scored term (a search term that spans several n-grams, or differs from its single gram): the
   * normalized term bytes. In a MGX_SORT_SCORE query tf is then BM25Scorer::CountTermOccurrences over the doc text
   * (bm25_scorer.cpp:27-45) and df is counted per execute as PopulateTermDocumentFrequency does
   * (search_pipeline.cpp:542-565: docs of the gram AND whose text contains the term); idf = ComputeIDF(total_docs,
   * df) is evaluated by the library and the `idf` field is ignored. Needs mgx_index_attach_text. NULL otherwise.
   * n_grams == 0 with a text: a term shorter than one n-gram — its doc set is every doc whose stored text contains it
   * (query::SearchNormalizedSubstring, src/query/substring_search.h:24-42, reached from SearchTermDocuments,
   * search_pipeline.cpp:438-446), found by a text scan on the device; as a scored term its df is 0, as the reference's
   * is (PopulateTermDocumentFrequency returns before counting, search_pipeline.cpp:546-549). */
  const uint8_t* text;
  uint32_t text_len;
} mgx_term;

typedef struct mgx_filter {
  uint32_t bitmap_id; /* from mgx_index_add_filter_bitmap */
  uint32_t negate;    /* 0: EQ (AND), 1: NE (ANDNOT) — search_pipeline.cpp:1196-1237 */
} mgx_filter;

/* Boolean expression over the positive terms (query::QueryNode, src/query/query_ast.h:52-83), in POSTFIX order.
 * Evaluated like EvaluateBooleanAstExpanded (src/server/search_pipeline.cpp:327-378): TERM = the term's doc set,
 * AND = intersection of its children, OR = union, NOT = universe minus child, where the universe is
 * DocumentStore::GetAllDocIds() (document_store_retrieval.cpp:242-258). */
#define MGX_EXPR_TERM 0u  /* arg = index into mgx_query.terms */
#define MGX_EXPR_EMPTY 1u /* a term the caller already knows to match nothing (unknown gram) */
#define MGX_EXPR_AND 2u   /* arg = number of children (>= 1), already on the stack */
#define MGX_EXPR_OR 3u    /* arg = number of children (>= 1) */
#define MGX_EXPR_NOT 4u   /* one child */
typedef struct mgx_expr_token {
  uint32_t op, arg;
} mgx_expr_token;

typedef struct mgx_query {
  /* positive terms, ALREADY in the order search_pipeline.cpp:2012-2014 leaves them (estimated_size ascending):
   * BM25 sums term contributions in this order (bm25_scorer.cpp:77-86) */
  const mgx_term* terms;
  uint32_t n_terms;
  const mgx_term* not_terms; /* ApplyNotFilter, search_pipeline.cpp:871-932 */
  uint32_t n_not_terms;
  const mgx_filter* filters;
  uint32_t n_filters;
  uint32_t sort;    /* MGX_SORT_* */
  uint32_t limit;   /* 0 => all */
  uint32_t offset;  /* MGX_SORT_SCORE only */
  uint32_t reverse; /* MGX_SORT_DOCID: descending docids; MGX_SORT_SCORE: 1 => SortOrder::DESC */
  double k1, b;     /* BM25Params (bm25_scorer.h:20-23) */
  uint64_t total_docs;    /* BM25Stats::doc_count; only documents the caller's idf values already encode */
  double avg_doc_length;  /* BM25Stats::avg_doc_length() */
  /* Optional boolean expression (ExecuteWithBooleanAst, search_pipeline.cpp:1408-1578). When n_expr > 0 the positive
   * part of the query is the expression instead of the plain AND of `terms`; NOT terms and filters still apply to its
   * result. The NOT universe is the doc-id range [universe_first, universe_first + universe_count) clipped to the
   * index (universe_count 0 => every slot of the index). */
  const mgx_expr_token* expr;
  uint32_t n_expr;
  uint32_t universe_first;
  uint64_t universe_count;
  /* Exact-text post-filter (PostFilterByText, search_pipeline.cpp:1239-1246): after NOT terms and filters, keep only
   * the docs whose text contains EVERY positive term. The caller sets it when the reference would: memory.verify_text
   * says so (ShouldApplyVerifyText :42-66), or a mixed-script term has a code point no query n-gram covers
   * (RequiresExactTextForHybridFragments :138-150). Every positive term must then carry its normalized `text`;
   * needs mgx_index_attach_text. after_filters still counts the docs before this filter, `total` those after it. */
  uint32_t exact_text;
  /* MGX_SORT_SCORE: which terms BM25 sums, as indices into `terms`, in summation order (bm25_scorer.cpp:77-86); a term
   * may be listed more than once (ScoreDocuments adds a repeated term's contribution each time). NULL => every positive
   * term in order (the plain conjunctive query). An expression query lists the TERM leaves that are not under a NOT, in
   * tree order (CollectAstScoringTerms, search_pipeline.cpp:232-254; search_handler.cpp:428-456 scores exactly those:
   * reference vector tests/server/search_pipeline_test.cpp:1350-1392); a FUZZY query its terms in the order given. A
   * listed term may match none of a result's text (an OR branch, a fuzzy neighbour): its tf is then 0 and it adds
   * nothing, as in the reference. */
  const uint32_t* score_terms;
  uint32_t n_score_terms;
} mgx_query;

typedef struct mgx_batch mgx_batch;

typedef struct mgx_query_result {
  uint64_t total;             /* results.size() before pagination (search_handler.cpp:471) */
  uint64_t total_candidates;  /* SearchPipelineResult funnel (search_pipeline.h:58-65) */
  uint64_t after_intersection;
  uint64_t after_not;
  uint64_t after_filters;
  uint32_t n_docs;            /* entries returned for this query */
  uint32_t docs_begin;        /* index of its first entry in mgx_result_view.docs / .scores */
} mgx_query_result;

typedef struct mgx_result_view {
  uint32_t n_queries;
  const mgx_query_result* queries;
  const uint32_t* docs;   /* concatenated per query, in rank order */
  const double* scores;   /* parallel to docs for MGX_SORT_SCORE queries (0.0 otherwise) */
} mgx_result_view;

/* Compiles and uploads a batch; the batch can be executed any number of times. */
int mgx_batch_prepare(mgx_index* idx, const mgx_query* queries, uint32_t n_queries, mgx_batch** out);
/* Re-compiles `batch` in place for a NEW set of queries (a serving loop: every step is a fresh batch). Equivalent to
 * mgx_batch_destroy + mgx_batch_prepare on the same index, but the object keeps its device arenas, pinned result
 * block and events, so the steady state allocates nothing. The last execute of the old contents must have completed
 * (mgx_batch_fetch returned, or its stream was synchronized). On failure the batch is empty but still usable. */
int mgx_batch_reset(mgx_batch* batch, const mgx_query* queries, uint32_t n_queries);
/* A non-blocking stream the batch object owns (created on first request, kept across mgx_batch_reset): batch objects
 * are independent pipeline slots, and a host layer without HIP headers passes this to execute / export / merge. */
int mgx_batch_stream(mgx_batch* batch, void** hip_stream);
/* Enqueues the whole batch on `hip_stream` (a hipStream_t; NULL = the default stream). Asynchronous; ends with the copy
 * of the result block to pinned host memory behind an event, so mgx_batch_fetch waits for THIS batch only. The batch's
 * input arrays (built in pinned host memory by prepare/reset) are shipped by the first execute, on this stream. */
int mgx_batch_execute(mgx_batch* batch, void* hip_stream);
/* Waits for the last execute and copies the (small) results to host memory owned by the batch. */
int mgx_batch_fetch(mgx_batch* batch, mgx_result_view* out);
/* Multi-GPU exchange (one rank per doc-range shard). The batch must be all MGX_SORT_SCORE, or all docid-ordered
 * pages (MGX_SORT_DOCID with 0 < limit <= 16384): a page is a best-first list under the key "doc id" (or its
 * complement for ascending order), so the same two calls serve both.
 * Copies this shard's per-query top-(offset+limit) of the last execute into two CALLER-owned DEVICE blobs on
 * `hip_stream`, laid out so that ONE all-gather per blob moves everything:
 *   blob64[n_queries*stride + n_queries] : keys (order-preserving u64 of the fp64 score, best first, `stride` per
 *                                           query), then this shard's match count per query;
 *   blob32[n_queries*stride + n_queries] : doc ids parallel to the keys, then the number of valid entries per query.
 * Call with both blobs NULL to only learn *stride. */
int mgx_batch_export_topk(mgx_batch* batch, uint64_t* blob64, uint32_t* blob32, uint32_t* stride, void* hip_stream);
/* Zero-copy form of the export for MGX_SORT_SCORE batches: the library keeps the shard's result in the exchange layout
 * already; *device_blob is that buffer (blob64 at offset 0, blob32 at *offset32 bytes, *bytes in all, a multiple of 8),
 * valid once the execute it belongs to has completed on its stream. NULL for other batches (use
 * mgx_batch_export_topk). */
int mgx_batch_export_buffer(mgx_batch* batch, void** device_blob, uint64_t* bytes, uint64_t* offset32);
/* Merges the blobs of `n_shards` ranks into the final page and total of every query, on `hip_stream`; read the
 * result with mgx_batch_fetch. Rank r's 64-bit blob starts at blob64 + r*pitch64 (u64 elements), its 32-bit blob at
 * blob32 + r*pitch32 (u32 elements); pitch 0 = the blob's own size (blobs gathered separately, rank after rank).
 * With explicit pitches a rank can export both blobs into ONE buffer (blob32 right behind blob64) and a single
 * all-gather moves everything: pitch64 = bytes_per_rank/8, pitch32 = bytes_per_rank/4. */
int mgx_batch_merge_shards(mgx_batch* batch, uint32_t n_shards, const uint64_t* blob64, uint64_t pitch64,
                           const uint32_t* blob32, uint64_t pitch32, void* hip_stream);
/* The collective of the sharded path behind this ABI (RCCL, loaded on first use): one communicator per rank of a table
 * whose shards are doc ranges, one rank per GPU. Rank 0 draws the id (mgx_comm_unique_id), the host layer hands its
 * MGX_COMM_ID_BYTES bytes to every rank by whatever channel it has (the reference has none to replace: it is a single
 * process), every rank calls mgx_comm_create — collectively.
 * mgx_batch_exchange, after mgx_batch_execute on the same stream: all-gathers every shard's per-query top-(offset+limit)
 * (ONE ncclAllGather of the batch's exchange blob) and merges them (mgx_batch_merge_shards); mgx_batch_fetch then
 * returns the table-wide page and total on every rank. mgx_batch_exchange_df, BEFORE mgx_batch_execute: the df pass of
 * the batch's text-level terms + ONE ncclAllReduce of their counts (no-op without such terms). Every rank must call
 * the same sequence of exchanges on its communicator, in the same order. */
#define MGX_COMM_ID_BYTES 128
typedef struct mgx_comm mgx_comm;
int mgx_comm_unique_id(uint8_t* id /* [MGX_COMM_ID_BYTES] */);
int mgx_comm_create(const uint8_t* id, int rank, int world, int device, mgx_comm** out);
void mgx_comm_destroy(mgx_comm* comm);
/* ncclCommAbort: this rank leaves the communicator NOW — its enqueued collectives are cancelled instead of waited for, and
 * every later exchange on `comm` fails with MGX_ERR_INTERNAL. What a rank must do when it cannot issue a batch's
 * collectives (a local failure between compile and exchange): its peers have issued theirs and would otherwise wait for
 * ever; with this rank gone their collectives fail or time out under the job's own watchdog, and the process exits
 * non-zero. The handle stays valid for mgx_comm_destroy. */
int mgx_comm_abort(mgx_comm* comm);
int mgx_batch_exchange_df(mgx_batch* batch, mgx_comm* comm, void* hip_stream);
int mgx_batch_exchange(mgx_batch* batch, mgx_comm* comm, void* hip_stream);
/* mgx_batch_execute for a shard of a table (every rank calls it for the same batch, in the same order; follow it with
 * mgx_batch_exchange): SORT _score batches run their seed items first, all-gather the seeds' best keys (ONE small
 * ncclAllGather: offset+limit keys per query and rank) and raise every query's pruning bound to the (offset+limit)-th best
 * key of the union before the rest of the batch runs — a shard's own matches give a weak bound (top-k is a property of
 * the whole table). The shard's LOCAL page may then hold fewer than offset+limit docs: those it lacks are on no rank's
 * share of the table-wide page; totals and funnel counters are unaffected. Other batches: as mgx_batch_execute. The
 * exchange runs from four ranks up (MGX_SEED_EXCHANGE=0 / 1 in every rank's environment forces it off / on; otherwise
 * this call is mgx_batch_execute) and is entered by every rank or by none: the decision uses the queries and the world
 * size only (shards of a million docs and more, MGX_SEED_EXCHANGE_MIN_DOCS). */
int mgx_batch_execute_sharded(mgx_batch* batch, mgx_comm* comm, void* hip_stream);
/* The same with the caller's transport (a test's gloo all-gather, MPI, ...): `gather` must fill all[r * bytes ..] with
 * rank r's `mine` for every rank (both are DEVICE buffers of the batch; the call may enqueue on hip_stream or block) and
 * return MGX_OK. */
typedef int (*mgx_gather_fn)(void* user, const void* mine, void* all, uint64_t bytes, void* hip_stream);
int mgx_batch_execute_gather(mgx_batch* batch, int world, mgx_gather_fn gather, void* user, void* hip_stream);
/* Text-level terms across shards: df must be table-wide before idf is taken. mgx_batch_count_df enqueues only the df
 * pass of the batch's text-level terms; mgx_batch_df_buffer exposes the DEVICE array of their counts (u64 per DISTINCT
 * term of the batch, in order of first appearance — the same on every shard), which the caller sums over ranks in place (one RCCL all-reduce); the next
 * mgx_batch_execute then uses those counts instead of running its own df pass. *n == 0: nothing to exchange. */
int mgx_batch_count_df(mgx_batch* batch, void* hip_stream);
int mgx_batch_df_buffer(mgx_batch* batch, uint64_t** device_counts, uint32_t* n);
/* Algorithmic bytes of one execute (4*sum|L_i| + R*(T+4) + 12*min(k,R), summed over queries); R is
 * taken from the last fetched execute. */
int mgx_batch_algorithmic_bytes(mgx_batch* batch, uint64_t* list_bytes, uint64_t* score_bytes, uint64_t* topk_bytes);
/* Average duration (ms) of the dominant kernel over the executes since the last call, measured with HIP events on
 * the stream the kernel ran on; *n receives how many launches were averaged. Enables event recording from now on. */
int mgx_batch_kernel_time_ms(mgx_batch* batch, double* avg_ms, uint32_t* n);
void mgx_batch_destroy(mgx_batch* batch);

/* =====================================================================================================
 * Single operators (what the shim's Index / BM25Scorer / ResultSorter methods call)
 * ===================================================================================================== */

/* Index::SearchAnd(terms, limit, reverse): every gram known (the caller returns {} otherwise). */
int mgx_and(mgx_index* idx, const uint32_t* gram_ids, uint32_t n, uint64_t limit, int reverse, uint32_t** out_docs,
            uint64_t* out_n);
/* Index::SearchOr: unknown grams already dropped by the caller. */
int mgx_or(mgx_index* idx, const uint32_t* gram_ids, uint32_t n, uint32_t** out_docs, uint64_t* out_n);
/* Index::SearchNot(all_docs, terms): all_docs ascending. */
int mgx_not(mgx_index* idx, const uint32_t* all_docs, uint64_t n_all, const uint32_t* gram_ids, uint32_t n,
            uint32_t** out_docs, uint64_t* out_n);
/* Index::SearchByThreshold over DISTINCT known grams (the caller dedupes and handles the delegate/empty rules,
 * index.cpp:489-523). */
int mgx_threshold(mgx_index* idx, const uint32_t* gram_ids, uint32_t n, uint32_t threshold, uint32_t** out_docs,
                  uint64_t* out_n);
/* Index::FilterByNgrams: keeps caller order and duplicates. */
int mgx_retain(mgx_index* idx, const uint32_t* candidates, uint64_t n_cand, const uint32_t* gram_ids, uint32_t n,
               uint32_t** out_docs, uint64_t* out_n);
/* FACET (ExecuteFacetPipeline, src/server/search_pipeline.cpp:2061-2153): counts_out[v] = how many docs of the query's
 * result set hold value id v of the column (FilterIndex::GetColumnValueCountsFiltered, filter_index.cpp:284-312 — a
 * histogram of the column's value ids over the result bitmap); *matched = the size of the result set. `query` is the
 * search part (terms / expression, NOT terms, filters; sort, limit and offset are ignored: the whole result set counts);
 * the caller orders the values by count and pages them. counts_out has room for the column's n_values. */
int mgx_facet_counts(mgx_index* idx, const mgx_query* query, uint32_t column_id, uint64_t* counts_out, uint64_t* matched);

/* ---- mutable tables (the binlog applier's Index::AddDocument / UpdateDocument / RemoveDocument,
 * src/index/index.cpp:39-233, after the index was built) -------------------------------------------------------------
 * The column arrays of an mgx_index are immutable. A table that changes is TWO indexes on one device: the MAIN index as
 * built, with a LIVE row (a filter bitmap whose bits are cleared as documents are removed or superseded), and a small
 * DELTA index over the documents added or changed since, numbered 1..n in ascending order of their table ids with a doc
 * map back to those ids. A batch is compiled against both (same queries, table-wide statistics in both: the df the
 * planner passes, total_docs, avg_doc_length), both are executed on ONE stream, and the delta's pages are merged into
 * the main batch — the exchange of a sharded table without the wire. The host layer (mygram_shim: Index::AddDocument &
 * co.) owns the bookkeeping and rebuilds the delta when it changes; when the delta has grown it rebuilds the main index.
 *
 * mgx_index_set_live_bitmap: every batched query (and every df query of a text-level term) on `idx` is ANDed with this
 *   filter row right after its first term, so candidates, totals and funnel counters count live documents only; the
 *   single-operator entry points (mgx_and ...) do NOT apply it. enable = 0 turns it off.
 * mgx_index_update_filter_bitmap: sets / clears bits of a filter row in place (ids are table doc ids of this index's
 *   range). Not ordered against batches in flight: call it between batches (mgx_index_synchronize) when a batch must see
 *   either all or none of an update.
 * mgx_index_set_doc_map: table id of every doc slot (ascending) — mgx_batch_merge_local reports this index's documents
 *   under these ids.
 * mgx_index_invalidate_statistics: drops everything the index cached per (total_docs, avg_doc_length, idf): contribution
 *   tables, length norms, block-max bounds, df counts. Call when the table-wide statistics changed, with NO batch of this
 *   index compiled or in flight (compiled batches hold table addresses); it waits for the device.
 * mgx_batch_df_merge_local, before the executes: df pass of the text-level terms on every batch, summed, the sum handed
 *   to every batch (mgx_batch_exchange_df without the all-reduce).
 * mgx_batch_merge_local, after the executes, all on `hip_stream`: merges the others' top-(offset+limit) per query into
 *   `primary` (mgx_batch_exchange without the all-gather); mgx_batch_fetch(primary) then returns the table-wide page
 *   and total. Funnel counters stay per index: fetch the others too and add them. Every query needs a page
 *   bound (MGX_SORT_SCORE, or a docid-ordered page); unlike the exchange, a batch may mix the two (each group is merged
 *   on its own). The others must stay untouched until primary has been fetched. */
/* Compaction of a mutable table (the main index is rebuilt from the table's current documents): the texts the index holds
 * (mgx_index_attach_text) and a filter column's arrays back on the host. mgx_index_copy_text with both buffers NULL only
 * reports *total_bytes; text_off has room for n_docs + 1 entries (offsets start at 0). */
int mgx_index_copy_text(mgx_index* idx, uint8_t* text_bytes, uint64_t capacity, uint64_t* text_off, uint64_t* total_bytes);
int mgx_index_filter_column_export(mgx_index* idx, uint32_t column_id, uint64_t* values, uint8_t* is_null,
                                   uint32_t* value_ids /* may be NULL */);
/* One document's text (text NULL: only *len). */
int mgx_index_read_text(mgx_index* idx, uint32_t doc_id, uint8_t* text, uint64_t capacity, uint64_t* len);
/* One document's stored value of a filter column, as it was given to mgx_index_add_filter_column (a document that moves
 * to the delta index takes its filter values along). value_id may be NULL. */
int mgx_index_filter_column_read(mgx_index* idx, uint32_t column_id, uint32_t doc_id, uint64_t* value_bits, int* is_null,
                                 uint32_t* value_id);
/* enable: 0 off, 1 on, MGX_LIVE_BITMAPS_CLEAN: on, and the caller also takes every dead document out of the bitmap form of
 * its dense grams (mgx_index_clear_postings with the document's distinct n-grams, BEFORE clearing its live bit): a query
 * whose first term is made of bitmap-form grams only then holds live documents by construction and gets no live-row operand
 * (the benchmark's 3-term AND keeps its three operands). */
#define MGX_LIVE_BITMAPS_CLEAN 3
int mgx_index_set_live_bitmap(mgx_index* idx, uint32_t bitmap_id, int enable);
/* Clears posting (docids[i], gram_ids[i]) in the gram's bitmap form, if it has one (list-form grams are left alone: queries
 * that read them keep the live-row operand). Same ordering rule as mgx_index_update_filter_bitmap. */
int mgx_index_clear_postings(mgx_index* idx, const uint32_t* docids, const uint32_t* gram_ids, uint64_t n);
int mgx_index_update_filter_bitmap(mgx_index* idx, uint32_t bitmap_id, const uint32_t* set_docids, uint64_t n_set,
                                   const uint32_t* clear_docids, uint64_t n_clear);
int mgx_index_set_doc_map(mgx_index* idx, const uint32_t* table_ids, uint64_t n);
int mgx_index_invalidate_statistics(mgx_index* idx);
int mgx_index_synchronize(mgx_index* idx);
int mgx_batch_df_merge_local(mgx_batch* primary, mgx_batch* const* others, uint32_t n_others, void* hip_stream);
int mgx_batch_merge_local(mgx_batch* primary, mgx_batch* const* others, uint32_t n_others, void* hip_stream);

/* BM25Scorer::ScoreDocuments for single-gram terms: one score per candidate, in candidate order. A candidate
 * outside the index or with empty text scores 0.0 (bm25_scorer.cpp:73-89). */
int mgx_score_documents(mgx_index* idx, const uint32_t* candidates, uint64_t n_cand, const uint32_t* gram_ids,
                        const double* idfs, uint32_t n_terms, double avg_doc_length, double k1, double b,
                        double* scores_out);
/* BM25Scorer::ScoreDocuments for terms of any length: tf counted in the doc text as the reference does
 * (CountTermOccurrences, bm25_scorer.cpp:27-45). Terms are normalized bytes, term t = term_bytes[term_off[t] ..
 * term_off[t+1]). Needs mgx_index_attach_text. */
int mgx_score_documents_text(mgx_index* idx, const uint32_t* candidates, uint64_t n_cand, const uint8_t* term_bytes,
                             const uint32_t* term_off, const double* idfs, uint32_t n_terms, double avg_doc_length,
                             double k1, double b, double* scores_out);
/* ResultSorter::SortByScore, any length, any page (limit 0 = all): a bounded page (offset+limit <= 1024) of a long
 * array is selected by per-wave top-k scans + a merge, everything else by a full device sort of the (score, docid)
 * pairs (the reference is benchmarked with OFFSET 10000, docs/releases/v1.3.5.md:238). n < 2^32. */
int mgx_sort_by_score(mgx_index* idx, const uint32_t* results, const double* scores, uint64_t n, int descending,
                      uint32_t limit, uint32_t offset, uint32_t** out_docs, uint64_t* out_n);

#ifdef __cplusplus
}
#endif
#endif /* MYGRAM_GPU_H_ */
