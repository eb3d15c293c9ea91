/*
 * mygram_shim_c.h — C entry points of libmygram_shim.so: the host C++17 layer of this repository
 * (mygram-db_amd/csrc/shim/: mygramdb::index::Index, search_pipeline::ExecuteBatch, BatchExecutor) made callable from
 * a host language that is not C++ — this repository's bench.py and tests (ctypes); a cgo / JNI binding would use the
 * same functions. NOT a reference interface: the reference's seam for this path is C++; what these
 * wrap mirrors search_pipeline::ExecuteFullPipeline's regular branch (src/server/search_pipeline.cpp:2002-2030) +
 * the BM25 glue of SearchHandler::HandleSearch (src/server/handlers/search_handler.cpp:405-470) for a batch.
 * 0 = success, otherwise a mygram::utils::ErrorCode value; mgxs_last_error() is thread-local.
 */
#ifndef MYGRAM_SHIM_C_H_
#define MYGRAM_SHIM_C_H_
#include <stddef.h>
#include <stdint.h>

#include "mygram_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgxs_table mgxs_table;        /* a mygramdb::index::Index over adopted handles */
typedef struct mgxs_executor mgxs_executor;  /* a search_pipeline::BatchExecutor */

const char* mgxs_last_error(void);

/* Index::Adopt: the handles stay the caller's and must outlive the table. */
int mgxs_table_adopt(mgx_columns* columns, mgx_index* device_index, int ngram_size, int kanji_ngram_size,
                     int cross_boundary_ngrams, mgxs_table** out);
/* Index::FromDump: a table of a reference dump (DUMP SAVE, "MGDB" v2) as a searchable, ranking table — postings, texts, the
 * store's id set and its filter columns on the device. table_name NULL / "": the first table. */
int mgxs_table_from_dump(const uint8_t* data, uint64_t len, const char* table_name, int device, mgxs_table** out);
/* Doc-range shards: the table-wide BM25Stats and per-gram posting sizes (by this shard's gram ids) every rank must
 * agree on, so that idf — and every score — is identical on all ranks. */
int mgxs_table_set_global_stats(mgxs_table* table, uint64_t total_docs, double avg_doc_length,
                                const uint64_t* global_posting_sizes, uint64_t n_grams);
/* Index(normalize_nfkc, normalize_width, normalize_lower) of src/index/index.h:58-60 for an adopted table: how query
 * terms are normalised before n-gram generation (defaults: nfkc, "keep", lower). */
int mgxs_table_set_normalization(mgxs_table* table, int nfkc, const char* width, int lower);
/* Doc-range shards: grams (NUL-terminated UTF-8) of the table that this shard holds no posting for, with their table-wide
 * posting sizes — the planner then resolves them to MGX_GRAM_ABSENT instead of ending the query on this rank only. */
int mgxs_table_set_absent_grams(mgxs_table* table, uint64_t n, const char* const* grams, const uint64_t* sizes);
void mgxs_table_destroy(mgxs_table* table);

/* mygram::utils::NormalizeText (src/utils/string_utils.cpp:295-380): NFKC -> width ("narrow" | "wide" | "keep") ->
 * lower through ICU when this library was built with it (mgxs_normalize_uses_icu() == 1), otherwise the reference's
 * non-ICU branch (ASCII lower-casing). Invalid UTF-8 normalises to "" (fails closed, :363-366). *out_len receives the
 * byte length; MGX_ERR_OUT_OF_RANGE when it exceeds cap (nothing is written then). */
int mgxs_normalize_uses_icu(void);
int mgxs_normalize_text(const char* text, size_t len, int nfkc, const char* width, int lower, char* out, size_t cap,
                        size_t* out_len);

int mgxs_executor_create(mgxs_table* table, int depth, int planner_threads, mgxs_executor** out);
/* One rank of a table sharded by doc range: every batch's per-shard top-k is all-gathered over `comm` (mgx_comm_create,
 * RCCL) and merged inside mgxs_submit, on the batch's stream; mgxs_wait returns table-wide pages and totals on every
 * rank. Collective: every rank submits the same batches in the same order (BatchExecutor::Options::comm). */
int mgxs_executor_create_sharded(mgxs_table* table, int depth, int planner_threads, mgx_comm* comm, mgxs_executor** out);
void mgxs_executor_destroy(mgxs_executor* ex);

/* BatchExecutor::Warm — setup, outside any timed loop: runs the sample batch (same arguments as mgxs_submit) through
 * every slot `rounds` times and discards the results, so that the slots' device arenas, pinned blocks, streams, the
 * compile helper threads and the index's per-parameter tables exist before the first real batch. No batch may be in
 * flight. */
int mgxs_executor_warm(mgxs_executor* ex, uint32_t n_queries, const uint32_t* n_terms, const char* const* terms,
                       uint32_t limit, uint32_t offset, int sort_by_score, int descending, int rounds);

/* One batch of plain conjunctive queries (query::Query with search_text + and_terms): query i has n_terms[i] raw term
 * strings, taken in order from `terms` (NUL-terminated UTF-8). All queries share limit / offset / sort / order
 * (SORT _score DESC LIMIT 10 is the benchmark's shape). Plans on the host, compiles into a re-used batch object and
 * enqueues; *ticket identifies the batch for mgxs_wait. Fails when all `depth` slots hold unfetched batches. */
int mgxs_submit(mgxs_executor* ex, uint32_t n_queries, const uint32_t* n_terms, const char* const* terms,
                uint32_t limit, uint32_t offset, int sort_by_score, int descending, uint64_t* ticket);
/* Results of a submitted batch: per query its total (results.size() before pagination) and page length, pages packed
 * `limit` entries apart in docs / scores (scores may be NULL). timing_ms (may be NULL) receives
 * {plan, compile, enqueue, wait} host milliseconds of this batch and, fifth, how many of its queries ran on the device
 * (the rest were resolved by the planner: an unknown gram): 5 doubles. */
int mgxs_wait(mgxs_executor* ex, uint64_t ticket, uint64_t* totals, uint32_t* n_docs, uint32_t* docs, double* scores,
              double* timing_ms);

/* Index::AddFilterColumn: one filter column of the table (DocumentStore filter values, src/storage/document_store.h:73-87),
 * one value per doc slot. value_type = the FilterValue alternative every non-NULL value holds: 1 bool, 2 int8, 3 uint8,
 * 4 int16, 5 uint16, 6 int32, 7 uint32, 8 int64, 9 uint64, 10 TimeValue (seconds), 11 string, 12 double. `values`: n x 8
 * bytes (int64 for the signed types and bool, uint64 for the unsigned ones, double for 12; ignored for strings, which
 * come NUL-terminated in `strings`); is_null (may be NULL) marks NULLs. */
int mgxs_table_add_filter_column(mgxs_table* table, const char* name, int value_type, uint64_t n, const void* values,
                                 const char* const* strings, const uint8_t* is_null);
/* Mutable tables (Index::AddDocument / UpdateDocument / RemoveDocument after the index was built: the binlog applier's calls,
 * src/mysql/binlog_event_processor.cpp:96,138,184,234,283). Texts are normalized. A document's filter values travel as
 * n_filters (name, type, 8-byte value | string) tuples: types as in mgxs_table_add_filter_column, 0 = NULL. Changes take
 * effect at the next search / submit on the table. */
int mgxs_table_add_document(mgxs_table* table, uint32_t doc_id, const char* text, size_t len, uint32_t n_filters,
                            const char* const* names, const int* types, const void* values, const char* const* strings);
/* with_filters = 0: the document keeps its filter values. */
int mgxs_table_update_document(mgxs_table* table, uint32_t doc_id, const char* old_text, size_t old_len, const char* new_text,
                               size_t new_len, int with_filters, uint32_t n_filters, const char* const* names,
                               const int* types, const void* values, const char* const* strings);
int mgxs_table_remove_document(mgxs_table* table, uint32_t doc_id, const char* text, size_t len);
/* Index::UpdateFilters (DocumentStore::UpdateDocument(doc_id, filters)): the document's filter values change, its text stays. */
int mgxs_table_update_filters(mgxs_table* table, uint32_t doc_id, uint32_t n_filters, const char* const* names,
                              const int* types, const void* values, const char* const* strings);
/* Index::SetMutationStaleness: recorded changes become visible at most this long after they were made (0: at once). */
int mgxs_table_set_mutation_staleness(mgxs_table* table, uint64_t microseconds);
/* Index::Compact: the main index rebuilt from the table's current documents; the delta goes. */
int mgxs_table_compact(mgxs_table* table);
int mgxs_table_mutation_stats(mgxs_table* table, uint64_t* main_documents, uint64_t* delta_documents,
                              uint64_t* removed_from_main, uint64_t* epoch);
/* One query with parsed FILTER conditions (query::FilterCondition: column, op 0 EQ 1 NE 2 GT 3 GTE 4 LT 5 LTE as
 * query::FilterOp, literal) through search_pipeline::ExecuteBatch; the conditions are resolved like
 * ApplyFiltersWithBitmap (src/server/search_pipeline.cpp:1196-1237). docs / scores have room for `limit` entries. */
int mgxs_search(mgxs_table* table, uint32_t n_terms, const char* const* terms, uint32_t n_not, const char* const* not_terms,
                uint32_t n_cond, const char* const* cond_columns, const uint32_t* cond_ops, const char* const* cond_values,
                int sort_by_score, int descending, uint32_t limit, uint32_t offset, uint64_t* total, uint32_t* n_docs,
                uint32_t* docs, double* scores /* may be NULL */);
/* search_pipeline::ExecuteFacet (ExecuteFacetPipeline, src/server/search_pipeline.cpp:2061-2153): value counts of `column`
 * over the query's result set (no terms: every document), count descending, paged by offset then limit. counts has room
 * for `limit` entries, display_off for limit + 1; value k's printable form (DeserializeToDisplayString) is
 * display[display_off[k] .. display_off[k+1]). */
int mgxs_facet(mgxs_table* table, uint32_t n_terms, const char* const* terms, uint32_t n_not, const char* const* not_terms,
               uint32_t n_cond, const char* const* cond_columns, const uint32_t* cond_ops, const char* const* cond_values,
               const char* column, uint32_t limit, uint32_t offset, uint64_t* matched, uint64_t* total_values,
               uint32_t* n_out, uint64_t* counts, char* display, size_t display_cap, uint32_t* display_off);

/* search_pipeline::MicroBatcher: the thread-safe front end. mgxs_batcher_search may be called from any number of threads at
 * once and blocks until the query's batch has run; queries are gathered into batches of at most `max_batch`, a batch
 * closing when full or when its first query has waited `max_delay_us`. docs / scores have room for
 * `limit` entries. */
typedef struct mgxs_batcher mgxs_batcher;
int mgxs_batcher_create(mgxs_table* table, uint32_t max_batch, uint32_t max_delay_us, int depth, int planner_threads,
                        mgxs_batcher** out);
void mgxs_batcher_destroy(mgxs_batcher* b);
int mgxs_batcher_search(mgxs_batcher* b, uint32_t n_terms, const char* const* terms, uint32_t limit, uint32_t offset,
                        int sort_by_score, int descending, uint64_t* total, uint32_t* n_docs, uint32_t* docs,
                        double* scores /* may be NULL */);
int mgxs_batcher_stats(mgxs_batcher* b, uint64_t* batches, uint64_t* queries, uint64_t* closed_full,
                       uint64_t* closed_by_delay);

#ifdef __cplusplus
}
#endif
#endif
