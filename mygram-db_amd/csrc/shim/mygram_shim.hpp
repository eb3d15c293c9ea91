// mygram_shim.hpp — the reference's C++ operator surface for the query hot path, re-exposed over libmygram_gpu.so.
//
// A MygramDB maintainer swaps these classes in behind the call sites listed in INTEGRATION.md; signatures, argument
// meaning and error behaviour follow the reference (paths below are relative to the reference tree):
//   mygramdb::index::Index          src/index/index.h:46-300      read side: SearchAnd / SearchOr / SearchNot /
//                                                                 SearchByThreshold / FilterByNgrams / PostingSize /
//                                                                 EstimatePostingSize / Count / GetNgramSize / ...
//   mygramdb::index::BM25Scorer     src/index/bm25_scorer.h:43-83 ComputeIDF, ScoreDocuments
//   mygramdb::query::ResultSorter   src/query/result_sorter.h:75-76 SortByScore
//   mygramdb::search_pipeline::ExecuteBatch   NEW: N parsed queries at once (the reference runs one per request,
//                                             src/server/search_pipeline.cpp:1757); plans each like Execute :795-869
// Everything that touches postings runs on the device through the C ABI (include/mygram_gpu.h); the host keeps what
// the reference also does on the host per query: normalisation, n-gram generation, dictionary lookup, term
// ordering, the empty / unknown-term rules. No exception leaves these classes (request_dispatcher.cpp:188-192).
//
// Differences a maintainer must know (also listed in INTEGRATION.md):
//   * AddDocument() before the first search only records documents; the column arrays are then built and uploaded
//     once. Changes after that (the binlog applier's AddDocument / UpdateDocument / RemoveDocument) go to a delta index
//     beside the main one and a live-document row on it (see "mutable tables" below);
//   * BM25Scorer::ScoreDocuments takes the Index where the reference takes a DocumentStore: tf and doc length come
//     from the index's own columns, which is exact for search terms that are one n-gram long;
//   * NormalizeText runs ICU (NFKC / width / lower) when the build found ICU, else the reference's ASCII fallback;
//     mygram::utils::NormalizeTextUsesIcu() says which.
#pragma once

#include <chrono>
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <string_view>
#include <utility>
#include <variant>
#include <vector>

#include "../../../include/mygram_gpu.h"

namespace mygram::utils {

// values of src/utils/error.h:35+ that this path can return
enum class ErrorCode : std::uint16_t {
  kSuccess = 0,
  kInvalidArgument = 2,
  kOutOfRange = 3,
  kNotImplemented = 4,
  kInternalError = 5,
  kQueryInvalidSort = 3007,
  kIndexNotFound = 4000,
};

class Error {
 public:
  Error() = default;
  Error(ErrorCode code, std::string message) : code_(code), message_(std::move(message)) {}
  [[nodiscard]] ErrorCode code() const { return code_; }
  [[nodiscard]] const std::string& message() const { return message_; }

 private:
  ErrorCode code_ = ErrorCode::kSuccess;
  std::string message_;
};

inline Error MakeError(ErrorCode code, std::string message) { return Error(code, std::move(message)); }

template <typename E>
struct Unexpected {
  E error;
};
template <typename E>
Unexpected<std::decay_t<E>> MakeUnexpected(E&& e) {
  return Unexpected<std::decay_t<E>>{std::forward<E>(e)};
}

// Minimal value-or-error carrier with the interface the call sites use (has_value, operator bool, *, ->, error()).
template <typename T, typename E>
class Expected {
 public:
  Expected(T value) : v_(std::move(value)) {}                       // NOLINT(google-explicit-constructor)
  Expected(Unexpected<E> u) : v_(std::move(u.error)) {}             // NOLINT(google-explicit-constructor)
  [[nodiscard]] bool has_value() const { return v_.index() == 0; }
  explicit operator bool() const { return has_value(); }
  T& operator*() { return std::get<0>(v_); }
  const T& operator*() const { return std::get<0>(v_); }
  T* operator->() { return &std::get<0>(v_); }
  const T* operator->() const { return &std::get<0>(v_); }
  T& value() { return std::get<0>(v_); }
  const T& value() const { return std::get<0>(v_); }
  const E& error() const { return std::get<1>(v_); }

 private:
  std::variant<T, E> v_;
};

// src/utils/string_utils.h NormalizeText (string_utils.cpp:295-380): invalid UTF-8 fails closed ("" + the failure
// counter); with ICU (build flag MGX_USE_ICU, set by the Makefile when <unicode/unorm2.h> is present): NFKC ->
// width transliteration ("narrow" = Fullwidth-Halfwidth, "wide" = Halfwidth-Fullwidth, anything else keeps) -> full
// Unicode lower-casing; without ICU: ASCII lower-casing only (the reference's own non-ICU branch, :371-377).
[[nodiscard]] std::string NormalizeText(std::string_view text, bool nfkc, std::string_view width, bool lower);
[[nodiscard]] bool NormalizeTextUsesIcu();  // which of the two branches this build runs
[[nodiscard]] uint64_t GetTextNormalizationFailureCount();
void ResetTextNormalizationFailureCountForTesting();

}  // namespace mygram::utils

namespace mygramdb::storage {
using DocId = uint32_t;  // src/types/doc_id.h:31

// src/storage/document_store.h:44-87: the value a document holds in one filter column.
struct TimeValue {
  int64_t seconds;
  bool operator==(const TimeValue& o) const { return seconds == o.seconds; }
  bool operator<(const TimeValue& o) const { return seconds < o.seconds; }
};
using FilterValue = std::variant<std::monostate, bool, int8_t, uint8_t, int16_t, uint16_t, int32_t, uint32_t, int64_t,
                                 uint64_t, TimeValue, std::string, double>;
using FilterMap = std::unordered_map<std::string, FilterValue>;
// FilterIndex::SerializeFilterValue (src/storage/filter_index.cpp:177-258): type tag + little-endian value bytes — the key
// FACET reports a value under; DeserializeToDisplayString (:314-372) is its printable form.
std::string SerializeFilterValue(const FilterValue& value);
std::string FilterValueToDisplayString(const FilterValue& value);
}

namespace mygramdb::index {

using DocId = storage::DocId;

// Failure reporting for the methods that keep the reference's signatures (std::vector results, no Expected): a device
// failure cannot travel in the return value, and an empty vector would read as "no hits". Every such method clears
// this thread's error on entry and sets it on failure; a call site checks it after the call and treats the result as
// valid only if it is empty (SearchHandler would answer with an error instead of an empty page):
//     auto hits = index.SearchAnd(terms);
//     if (!mygramdb::index::LastDeviceError().empty()) return MakeUnexpected(MakeError(kInternalError, ...));
// (the reference has the same shape of contract around its Roaring fallback, posting_list.cpp:676-689)
const std::string& LastDeviceError();

class Index {
 public:
  // src/index/index.h:58-60 (roaring_threshold becomes the dense-bitmap density threshold of the device index;
  // 0 keeps the library default)
  explicit Index(int ngram_size = 2, int kanji_ngram_size = 1, double roaring_threshold = 0.0,
                 bool cross_boundary_ngrams = true, bool normalize_nfkc = true,
                 const std::string& normalize_width = "keep", bool normalize_lower = true, int device = 0);
  ~Index();
  Index(const Index&) = delete;
  Index& operator=(const Index&) = delete;

  struct DocumentItem {
    DocId doc_id;
    std::string text;  // normalized
  };

  // Recorded on the host until the first search. Returns whether the text yields any n-gram.
  bool AddDocument(DocId doc_id, std::string_view text);
  void AddDocumentBatch(const std::vector<DocumentItem>& documents);

  // ---- mutable tables: the binlog applier's calls after the index was built (src/index/index.h:88-117, call sites
  // src/mysql/binlog_event_processor.cpp:96,138,184,234,283) --------------------------------------------------------------
  // The device's column arrays are immutable, so a table that changes is two indexes: the MAIN index as built, whose
  // removed / superseded documents are cleared in a live-document row every query is ANDed with, and a DELTA index over
  // the documents added or changed since (rebuilt when it changed, before the next query). search_pipeline::ExecuteBatch,
  // BatchExecutor, MicroBatcher and ExecuteFacet run a query on both and merge the pages on the device
  // (mgx_batch_merge_local: the exchange of a sharded table without the wire); BM25 statistics (N, average length, df)
  // are the table's — live documents of both — so scores equal those of an index built from the current documents.
  // Texts are normalized (as for AddDocument); `text` / `old_text` of a removal must be the document's current text, as
  // in the reference (its n-grams are what is taken out of the posting sizes).
  //   AddDocument(doc, text[, filters]) after the first search: a NEW document (false + LastError() if the id is live);
  //   UpdateDocument: the document's text becomes new_text (index.cpp:199-233); its filter values stay, or become `filters`;
  //   RemoveDocument: the document leaves the table (index.cpp:148-197).
  // Calls are cheap (host bookkeeping); they take effect at the next query entry point, which must not overlap another
  // thread's query on the same Index while it applies them (one BatchExecutor / MicroBatcher per mutable table, or
  // external exclusion — the reference holds its index write lock for the same span).
  // Limits: BatchQuery::filters (raw bitmap ids) are per device index and are refused while a delta exists — use
  // filter_conditions; every query of a batch needs a page bound (SORT _score, or a docid page with 0 < limit <= 16384:
  // what the device merge takes; score and page queries may share a batch); sharded tables (BatchExecutor::Options::comm)
  // are static.
  void UpdateDocument(DocId doc_id, std::string_view old_text, std::string_view new_text);
  void UpdateDocument(DocId doc_id, std::string_view old_text, std::string_view new_text,
                      const storage::FilterMap& filters);
  void RemoveDocument(DocId doc_id, std::string_view text);
  // DocumentStore::UpdateDocument(doc_id, filters) (src/storage/document_store.h; binlog_event_processor.cpp:217): the
  // document's filter values become `filters`, its text stays. The typed device columns are immutable, so a document of
  // the main index moves to the delta with its text (read back from the device). False if the id is no live document.
  bool UpdateFilters(DocId doc_id, const storage::FilterMap& filters);
  struct MutationStats {
    uint64_t main_documents = 0;   // live documents of the main index
    uint64_t delta_documents = 0;  // documents of the delta index
    uint64_t removed_from_main = 0;
    uint64_t epoch = 0;            // times the device state was brought up to date
  };
  [[nodiscard]] MutationStats GetMutationStats() const;
  // Brings the device state up to date with the recorded changes (every query entry point calls it). "" or an error.
  // With a staleness bound set (below) the entry points' calls do nothing until the bound has passed since the last
  // application; force = true applies now.
  std::string ApplyMutations(bool force = false) const;
  // A table under a steady stream of changes would rebuild its delta index for every batch (0.07-0.2 s each). With a
  // staleness bound, recorded changes become visible to queries at most that long after they were made (they are applied
  // by the first query entry point after the bound has passed): the reference's binlog applier is asynchronous to
  // queries in the same way. Under a bound the delta of the recorded changes is built by a background thread while
  // queries run on the old state, and installed (a few milliseconds) by the next entry point that finds it ready.
  // 0 (default): every query sees every change recorded before it.
  void SetMutationStaleness(std::chrono::microseconds max_staleness);
  // Folds the delta back: the main index is rebuilt from the table's current documents (the texts of its live documents come
  // back from the device, where BM25 keeps them; filter values likewise) and the delta and the live row go. What the
  // reference's Optimize / a dump-and-reload do for a grown index. Same exclusion rule as the mutation-applying entry
  // points; an executor's batch objects move to the new device index by themselves. "" or an error.
  std::string Compact() const;

  [[nodiscard]] std::vector<DocId> SearchAnd(const std::vector<std::string>& terms, size_t limit = 0,
                                             bool reverse = false) const;
  [[nodiscard]] std::vector<DocId> FilterByNgrams(const std::vector<DocId>& candidates,
                                                  const std::vector<std::string>& terms) const;
  [[nodiscard]] std::vector<DocId> SearchOr(const std::vector<std::string>& terms) const;
  [[nodiscard]] std::vector<DocId> SearchNot(const std::vector<DocId>& all_docs,
                                             const std::vector<std::string>& terms) const;
  [[nodiscard]] std::vector<DocId> SearchByThreshold(const std::vector<std::string>& terms, size_t threshold) const;
  [[nodiscard]] uint64_t PostingSize(std::string_view term) const;
  [[nodiscard]] uint64_t EstimatePostingSize(std::string_view term) const;
  [[nodiscard]] uint64_t Count(std::string_view term) const { return PostingSize(term); }
  [[nodiscard]] int GetNgramSize() const { return ngram_size_; }
  [[nodiscard]] int GetKanjiNgramSize() const { return kanji_ngram_size_; }
  [[nodiscard]] bool GetCrossBoundaryNgrams() const { return cross_boundary_; }
  [[nodiscard]] bool GetNormalizeNfkc() const { return normalize_nfkc_; }                 // index.h:312-318
  [[nodiscard]] const std::string& GetNormalizeWidth() const { return normalize_width_; }
  [[nodiscard]] bool GetNormalizeLower() const { return normalize_lower_; }
  void SetNormalization(bool nfkc, const std::string& width, bool lower);  // for adopted handles (Adopt has no ctor args)
  [[nodiscard]] std::string NormalizeText(std::string_view text) const {                  // index.h:321-323
    return mygram::utils::NormalizeText(text, normalize_nfkc_, normalize_width_, normalize_lower_);
  }

  // BM25Stats of the table (src/server/server_types.h:157-193), computed when the index is finalised.
  [[nodiscard]] uint64_t Bm25DocCount() const;
  [[nodiscard]] double Bm25AvgDocLength() const;
  // FilterIndex (column,value) doc set -> device bitmap id usable in search_pipeline::BatchQuery::filters.
  [[nodiscard]] mygram::utils::Expected<uint32_t, mygram::utils::Error> AddFilterBitmap(
      const std::vector<DocId>& docs) const;
  // One filter column of the table (what DocumentStore::AddDocument's FilterMap holds per document,
  // src/storage/document_store.h:73-87): values[i] belongs to doc first_doc_id + i (std::monostate = NULL); every
  // non-NULL value must hold the same alternative (one MySQL column type). The column goes to the device by doc slot;
  // BatchQuery::filter_conditions and ExecuteFacet resolve against it. Returns "" or an error message.
  std::string AddFilterColumn(const std::string& name, const std::vector<storage::FilterValue>& values) const;
  // AddDocument with the document's filter values (recorded until the first search, like the text)
  bool AddDocument(DocId doc_id, std::string_view text, const storage::FilterMap& filters);

  // An Index over column arrays and a device index that already exist (built through the C ABI by the caller, e.g. a
  // loader that read them from a dump): nothing is copied, the handles stay the caller's and must outlive the Index.
  static std::unique_ptr<Index> Adopt(mgx_columns* columns, mgx_index* device_index, int ngram_size,
                                      int kanji_ngram_size, bool cross_boundary_ngrams);

  // A table of a reference dump (DUMP SAVE, "MGDB" v2: src/storage/dump_format_v2.cpp:520-770) as a searchable, ranking
  // Index: what Index::LoadFromStream + DocumentStore::LoadFromStream restore in the reference (dump_format_internal.cpp:
  // 285-305) — postings, normalized texts (BM25 needs them, as in the reference), the store's id set (the NOT universe)
  // and every filter column (FILTER conditions, FACET). `table` empty: the first table. Normalisation settings come from
  // the dump's index header. nullptr + *error on failure.
  static std::unique_ptr<Index> FromDump(const void* data, size_t len, const std::string& table, std::string* error,
                                         int device = 0);

  // Doc-range shards (one Index per GPU): the table-wide BM25Stats and every gram's table-wide posting size, by this
  // shard's gram ids — what idf must be computed from so that all ranks score identically. Returns "" or an error.
  std::string SetGlobalStats(uint64_t total_docs, double avg_doc_length, std::vector<uint64_t> global_posting_sizes);
  // Doc-range shards: the grams of the TABLE this shard holds no posting for, with their table-wide sizes. The planner
  // then treats them as known (MGX_GRAM_ABSENT: an empty operand on this shard), so every rank plans the same batch —
  // without it a query with such a gram would end on this rank's host (empty_term_detected) and run on the others.
  void SetAbsentGrams(std::unordered_map<std::string, uint64_t> grams);

  // internals shared with BM25Scorer / ResultSorter / search_pipeline
  struct Impl;
  [[nodiscard]] Impl* impl() const { return impl_.get(); }
  // Builds the column arrays and the device index now (otherwise done by the first search). Returns an error message
  // or "" — searches on an index that failed to build return empty results and keep the message in LastError().
  std::string Finalize() const;
  [[nodiscard]] const std::string& LastError() const;

 private:
  int ngram_size_, kanji_ngram_size_;
  bool cross_boundary_;
  bool normalize_nfkc_ = true;
  std::string normalize_width_ = "keep";
  bool normalize_lower_ = true;
  std::unique_ptr<Impl> impl_;
  mutable bool pending_columns_ready_ = false;  // (under Impl::mu)

 public:
  void FlushPendingFilterColumns() const;  // internal: AddDocument's filter values -> device columns, after Finalize
};

}  // namespace mygramdb::index

namespace mygramdb::storage {
// The slice of storage::DocumentStore (src/storage/document_store.h) the BM25 call sites touch: they hand the store to
// BM25Scorer::ScoreDocuments for the candidates' texts (search_handler.cpp:454, http_server.cpp:572). Here the texts —
// and the tf / doc-length columns made from them — live with the device index, so the store is a view of an Index; a
// maintainer constructs it next to the Index and the call sites compile unchanged.
class DocumentStore {
 public:
  explicit DocumentStore(const index::Index& index) : index_(&index) {}
  [[nodiscard]] const index::Index& index() const { return *index_; }
  [[nodiscard]] bool IsStoreTextsEnabled() const { return true; }  // document_store.h: SORT _score / HIGHLIGHT need texts
  [[nodiscard]] size_t Size() const;                                 // documents with a text

 private:
  const index::Index* index_;
};
}  // namespace mygramdb::storage

namespace mygramdb::index {

struct BM25Params {  // src/index/bm25_scorer.h:20-23
  double k1 = 1.2;
  double b = 0.75;
};

struct ScoredDoc {  // src/index/bm25_scorer.h:28-31
  DocId doc_id;
  double score;
};

class BM25Scorer {
 public:
  static double ComputeIDF(uint64_t total_docs, uint64_t doc_freq);
  // src/index/bm25_scorer.h:79-82 — `index` stands where the reference passes the DocumentStore (see header note).
  static mygram::utils::Expected<std::vector<ScoredDoc>, mygram::utils::Error> ScoreDocuments(
      const std::vector<DocId>& candidates, const std::vector<std::string>& search_terms,
      const std::vector<uint64_t>& term_doc_freqs, const Index& index, uint64_t total_docs, double avg_doc_length,
      const BM25Params& params);
  // the reference's exact signature (src/index/bm25_scorer.h:79-82)
  static mygram::utils::Expected<std::vector<ScoredDoc>, mygram::utils::Error> ScoreDocuments(
      const std::vector<DocId>& candidates, const std::vector<std::string>& search_terms,
      const std::vector<uint64_t>& term_doc_freqs, const storage::DocumentStore& doc_store, uint64_t total_docs,
      double avg_doc_length, const BM25Params& params) {
    return ScoreDocuments(candidates, search_terms, term_doc_freqs, doc_store.index(), total_docs, avg_doc_length, params);
  }
};

}  // namespace mygramdb::index

namespace mygramdb::query {

using DocId = storage::DocId;
enum class SortOrder : uint8_t { ASC, DESC };  // src/query/query_parser.h
enum class FilterOp : uint8_t { EQ, NE, GT, GTE, LT, LTE };  // src/query/query_parser.h:93-100
struct FilterCondition {                                       // src/query/query_parser.h:123-127
  std::string column;
  FilterOp op = FilterOp::EQ;
  std::string value;
};

// query::QueryNode (src/query/query_ast.h:52-83): the boolean expression tree QueryASTParser produces.
enum class NodeType : uint8_t { AND, OR, NOT, TERM };
struct QueryNode {
  NodeType type;
  std::string term;  // TERM only
  std::vector<std::unique_ptr<QueryNode>> children;
  explicit QueryNode(std::string term_value) : type(NodeType::TERM), term(std::move(term_value)) {}
  explicit QueryNode(NodeType node_type) : type(node_type) {}
};

class ResultSorter {
 public:
  // src/query/result_sorter.h:75-76. Runs on the device of `index` (any finalised index: only its stream is used).
  static std::vector<DocId> SortByScore(const index::Index& index, const std::vector<DocId>& results,
                                        const std::vector<double>& scores, SortOrder order, uint32_t limit,
                                        uint32_t offset);
  // the reference's exact signature (src/query/result_sorter.h:75-76): the sort runs on the device of the Index this
  // thread used last (ScoreDocuments, a search, ...) — at the call sites that is the table just scored
  // (search_handler.cpp:454,469) — or, on a thread that has used none, of the Index finalised last in the process.
  static std::vector<DocId> SortByScore(const std::vector<DocId>& results, const std::vector<double>& scores,
                                        SortOrder order, uint32_t limit, uint32_t offset);
};

}  // namespace mygramdb::query

namespace mygramdb::search_pipeline {

using DocId = storage::DocId;

// The parts of query::Query (src/query/query_parser.h:207-243) the hot path consumes.
struct BatchQuery {
  std::vector<std::string> terms;      // search_text + and_terms (raw; normalised here)
  // ExecuteWithBooleanAst (src/server/search_pipeline.cpp:1408-1578): when set, the positive part of the query is this
  // tree (TERM = the term's doc set, NOT = every document of the index minus the child) and `terms` must be empty;
  // NOT terms and filters still apply to its result. With SORT _score the scored terms are the tree's TERM leaves that are
  // not under a NOT (CollectAstScoringTerms, search_pipeline.cpp:232-254), as SearchHandler scores them (:428-456).
  std::shared_ptr<const query::QueryNode> ast;
  std::vector<std::string> not_terms;
  std::vector<std::pair<uint32_t, bool>> filters;  // (bitmap id from Index::AddFilterBitmap, negate = FilterOp::NE)
  // query::Query::filters as parsed (FILTER col op value): resolved against the columns of Index::AddFilterColumn the way
  // ApplyFiltersWithBitmap does (search_pipeline.cpp:1196-1237) — all EQ / NE: the literal under every type
  // interpretation the column can hold (BuildTypeUnionBitmap :1021-1094), exact; any other operator in the list: every
  // condition by the per-document comparison of ApplyFilters (:1098-1194: NULL passes != only, doubles compare = / !=
  // with an epsilon). A condition becomes a device bitmap the first time it is seen and is cached by the Index.
  std::vector<query::FilterCondition> filter_conditions;
  uint32_t fuzzy_max_distance = 0;     // FUZZY d (query_parser_clauses.cpp:454: 1 or 2); 0 = not a fuzzy query.
                                       // ExecuteWithFuzzy (search_pipeline.cpp:1659-1744): per term "at least theta of
                                       // its n-grams", AND across terms in the order given; SORT _score scores the
                                       // exact terms. Not combined with verify_text here (edit-distance verification)
  bool sort_by_score = false;          // SORT _score
  bool verify_text = false;            // the caller's ShouldApplyVerifyText(memory.verify_text, terms) decision
                                       // (search_pipeline.cpp:42-66); mixed-script fragments force it (:858-866)
  query::SortOrder order = query::SortOrder::DESC;
  uint32_t limit = 100;                // api.default_limit (src/config/config.h:61)
  uint32_t offset = 0;
  index::BM25Params bm25;
};

// SearchPipelineResult (src/server/search_pipeline.h:58-65) plus the page the handler would format.
struct BatchResult {
  std::vector<DocId> results;          // the page, in rank order
  std::vector<double> scores;          // parallel to results for SORT _score
  uint64_t total = 0;                  // results.size() before pagination
  size_t total_candidates = 0, after_intersection = 0, after_not = 0, after_filters = 0;
  bool empty_term_detected = false;
};

// N queries planned like ExecuteFullPipeline's regular branch (GenerateTermInfos with df, sort by estimated size,
// Execute, BM25 + SortByScore) and run as ONE device batch.
mygram::utils::Expected<std::vector<BatchResult>, mygram::utils::Error> ExecuteBatch(
    const index::Index& index, const std::vector<BatchQuery>& queries);

// FACET (ExecuteFacetPipeline, src/server/search_pipeline.cpp:2061-2153): how many documents of the query's result set
// hold each value of `column`. query.terms / ast / not_terms / filters / filter_conditions select the documents (none of
// them: every document); limit / offset page the VALUES (offset first, :2141-2148); sort fields are ignored.
struct FacetOutput {
  uint64_t matched_documents = 0;
  uint64_t total_values = 0;  // values with a non-zero count, before paging
  // (SerializeFilterValue key, count), count descending; the order among equal counts is unspecified in the reference
  // (hash-map iteration + std::sort) — here ties keep the column's value order
  std::vector<std::pair<std::string, uint64_t>> value_counts;
  std::vector<std::string> display;  // DeserializeToDisplayString of each key, parallel to value_counts
};
mygram::utils::Expected<FacetOutput, mygram::utils::Error> ExecuteFacet(const index::Index& index, const BatchQuery& query,
                                                                        const std::string& column);

// A serving loop over ExecuteBatch's work. Submit queues a FRESH batch and returns at once; a pool of `planner_threads`
// workers plans it in chunks of queries (GenerateTermInfos, the size sort and idf per term — the way the reference
// plans each request on its own worker; several queued batches are planned side by side), the worker that ends a batch's
// planning compiles it into one of `depth` re-used batch objects (mgx_batch_reset: no allocation in steady state), and
// batches go to the device in ticket order, each on its object's own stream. Wait returns a ticket's results. The host
// therefore prepares up to `depth` batches while the device runs others: throughput is bounded by the slower of the
// two, not by their sum, and the host side scales with its threads.
// Submit and Wait may run on two different threads (one submitter, one waiter: MicroBatcher below); neither is
// re-entrant.
class BatchExecutor {
 public:
  struct Options {
    int depth = 2;
    int planner_threads = 4;
    // One rank of a table sharded by doc range: after every execute the shards' top-k are all-gathered
    // over this communicator (mgx_comm_create; RCCL) and merged, so Wait returns the table-wide page and total on
    // every rank. Every rank submits the same batches in the same order; the index carries the table-wide statistics
    // (Index::SetGlobalStats, and Index::SetAbsentGrams for the grams only other shards hold).
    mgx_comm* comm = nullptr;
  };
  struct Timing {  // host milliseconds of one batch
    double plan_ms = 0, compile_ms = 0, enqueue_ms = 0, wait_ms = 0;
    uint32_t device_queries = 0;  // queries of the batch that ran on the device (the rest were resolved by the planner)
  };
  BatchExecutor(const index::Index& index, Options options);
  explicit BatchExecutor(const index::Index& index) : BatchExecutor(index, Options{}) {}
  ~BatchExecutor();
  BatchExecutor(const BatchExecutor&) = delete;
  BatchExecutor& operator=(const BatchExecutor&) = delete;
  // -> ticket. Fails (kInvalidArgument) when all `depth` slots hold unfetched batches.
  mygram::utils::Expected<uint64_t, mygram::utils::Error> Submit(const std::vector<BatchQuery>& queries);
  // (no copy; on return `queries` holds the BatchQuery objects of an EARLIER batch — storage a serving loop builds its
  // next batch in without allocating; clear() or overwrite them)
  mygram::utils::Expected<uint64_t, mygram::utils::Error> Submit(std::vector<BatchQuery>&& queries);
  mygram::utils::Expected<std::vector<BatchResult>, mygram::utils::Error> Wait(uint64_t ticket, Timing* timing = nullptr);
  // Setup, outside any timed loop: runs `sample` through EVERY slot `rounds` times and throws the results away, so that
  // each slot's device arenas, pinned blocks, stream and events, the dispatcher threads' compile helpers and the index's
  // per-parameter tables (block-max bytes, length norms) exist before the first real batch — a serving process pays
  // for them at start-up, not on its first requests. Nothing of a batch's RESULT is kept: real batches are planned,
  // compiled and run from scratch. Call before the first Submit (no batch may be in flight).
  mygram::utils::Error Warm(const std::vector<BatchQuery>& sample, int rounds = 2);
  // Wait into a vector the caller keeps from batch to batch (its elements' vectors are re-used: no allocation in steady
  // state); the returned Error's code is kSuccess on success.
  mygram::utils::Error WaitInto(uint64_t ticket, std::vector<BatchResult>* results, Timing* timing = nullptr);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

// The front end a server puts between its connection threads and the executor: the reference runs one
// query per request thread (ExecuteFullPipeline per connection); the device wants them a thousand at a time. Search()
// is thread-safe and blocking: queries from any number of threads are gathered into one batch — closed when it holds
// `max_batch` queries or when its first query has waited `max_delay` — submitted to a BatchExecutor and answered when
// their batch is fetched. Two threads run inside: one forms and submits batches, one waits for tickets in order and
// hands the results out, so a batch is being formed, others are being planned / run, and one is being collected at the
// same time.
class MicroBatcher {
 public:
  struct Options {
    size_t max_batch = 1024;
    std::chrono::microseconds max_delay{200};
    BatchExecutor::Options executor;
  };
  struct Stats {
    uint64_t batches = 0, queries = 0;  // mean batch size = queries / batches
    uint64_t closed_full = 0, closed_by_delay = 0;
  };
  MicroBatcher(const index::Index& index, Options options);
  ~MicroBatcher();  // answers what is queued, then stops
  MicroBatcher(const MicroBatcher&) = delete;
  MicroBatcher& operator=(const MicroBatcher&) = delete;
  [[nodiscard]] mygram::utils::Expected<BatchResult, mygram::utils::Error> Search(BatchQuery query);
  [[nodiscard]] Stats GetStats() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace mygramdb::search_pipeline
