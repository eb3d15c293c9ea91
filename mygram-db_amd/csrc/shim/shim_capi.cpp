// shim_capi.cpp — extern "C" face of the C++17 shim (include/mygram_shim_c.h): what bench.py and the tests bind with
// ctypes to drive search_pipeline::BatchExecutor, i.e. to plan, compile, run and fetch fresh batches entirely in C++.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/mygram_shim_c.h"
#include "mygram_shim.hpp"

using mygramdb::index::Index;
using mygramdb::search_pipeline::BatchExecutor;
using mygramdb::search_pipeline::BatchQuery;
using mygramdb::search_pipeline::MicroBatcher;

struct mgxs_table {
  std::unique_ptr<Index> index;
};
struct mgxs_executor {
  std::unique_ptr<BatchExecutor> ex;
  std::vector<BatchQuery> queries;  // re-used between submits
  std::vector<mygramdb::search_pipeline::BatchResult> results;  // re-used between waits
  std::unordered_map<uint64_t, uint32_t> limits;  // ticket -> row pitch of mgxs_wait's docs / scores
};

struct mgxs_batcher {
  std::unique_ptr<MicroBatcher> mb;
};

namespace {
thread_local std::string g_err;
int Fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace

extern "C" {

const char* mgxs_last_error(void) { return g_err.c_str(); }

int mgxs_table_adopt(mgx_columns* columns, mgx_index* device_index, int ngram_size, int kanji_ngram_size,
                     int cross_boundary_ngrams, mgxs_table** out) {
  if (out) *out = nullptr;
  if (!columns || !device_index || !out) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_adopt: null argument");
  try {
    auto t = std::make_unique<mgxs_table>();
    t->index = Index::Adopt(columns, device_index, ngram_size, kanji_ngram_size, cross_boundary_ngrams != 0);
    if (!t->index->LastError().empty()) return Fail(MGX_ERR_INTERNAL, t->index->LastError());
    *out = t.release();
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_from_dump(const uint8_t* data, uint64_t len, const char* table_name, int device, mgxs_table** out) {
  if (out) *out = nullptr;
  if (!data || !out) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_from_dump: null argument");
  try {
    std::string err;
    auto t = std::make_unique<mgxs_table>();
    t->index = Index::FromDump(data, len, table_name ? table_name : "", &err, device);
    if (!t->index) return Fail(MGX_ERR_INVALID_ARGUMENT, err);
    *out = t.release();
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_set_global_stats(mgxs_table* table, uint64_t total_docs, double avg_doc_length,
                                const uint64_t* global_posting_sizes, uint64_t n_grams) {
  if (!table || (n_grams && !global_posting_sizes))
    return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_set_global_stats: null argument");
  try {
    const std::string err = table->index->SetGlobalStats(
        total_docs, avg_doc_length, std::vector<uint64_t>(global_posting_sizes, global_posting_sizes + n_grams));
    return err.empty() ? MGX_OK : Fail(MGX_ERR_INVALID_ARGUMENT, err);
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_set_normalization(mgxs_table* table, int nfkc, const char* width, int lower) {
  if (!table || !width) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_set_normalization: null argument");
  try {
    table->index->SetNormalization(nfkc != 0, width, lower != 0);
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_normalize_uses_icu(void) { return mygram::utils::NormalizeTextUsesIcu() ? 1 : 0; }

int mgxs_normalize_text(const char* text, size_t len, int nfkc, const char* width, int lower, char* out, size_t cap,
                        size_t* out_len) {
  if ((len && !text) || !width || !out_len || (cap && !out))
    return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_normalize_text: null argument");
  try {
    const std::string r = mygram::utils::NormalizeText(std::string_view(text ? text : "", len), nfkc != 0, width, lower != 0);
    *out_len = r.size();
    if (r.size() > cap) return Fail(MGX_ERR_OUT_OF_RANGE, "mgxs_normalize_text: output buffer too small (*out_len holds the size)");
    if (!r.empty()) std::memcpy(out, r.data(), r.size());
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_set_absent_grams(mgxs_table* table, uint64_t n, const char* const* grams, const uint64_t* sizes) {
  if (!table || (n && (!grams || !sizes))) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_set_absent_grams: null argument");
  try {
    std::unordered_map<std::string, uint64_t> m;
    for (uint64_t i = 0; i < n; ++i) m.emplace(grams[i], sizes[i]);
    table->index->SetAbsentGrams(std::move(m));
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

void mgxs_table_destroy(mgxs_table* table) { delete table; }

int mgxs_executor_create_sharded(mgxs_table* table, int depth, int planner_threads, mgx_comm* comm, mgxs_executor** out) {
  if (out) *out = nullptr;
  if (!table || !out) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_executor_create_sharded: null argument");
  try {
    auto e = std::make_unique<mgxs_executor>();
    BatchExecutor::Options o;
    o.depth = depth > 0 ? depth : 2;
    o.planner_threads = planner_threads > 0 ? planner_threads : 1;
    o.comm = comm;
    e->ex = std::make_unique<BatchExecutor>(*table->index, o);
    *out = e.release();
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_executor_create(mgxs_table* table, int depth, int planner_threads, mgxs_executor** out) {
  return mgxs_executor_create_sharded(table, depth, planner_threads, nullptr, out);
}

void mgxs_executor_destroy(mgxs_executor* ex) { delete ex; }

namespace {
void FillQueries(std::vector<BatchQuery>* out, uint32_t n_queries, const uint32_t* n_terms, const char* const* terms,
                 uint32_t limit, uint32_t offset, int sort_by_score, int descending) {
  out->resize(n_queries);
  size_t at = 0;
  for (uint32_t i = 0; i < n_queries; ++i) {
    BatchQuery& q = (*out)[i];  // (possibly an earlier batch's object: every field is set again)
    q.terms.resize(n_terms[i]);
    for (uint32_t t = 0; t < n_terms[i]; ++t) q.terms[t].assign(terms[at++]);
    q.ast.reset();
    q.not_terms.clear();
    q.filters.clear();
    q.fuzzy_max_distance = 0;
    q.verify_text = false;
    q.bm25 = mygramdb::index::BM25Params{};
    q.sort_by_score = sort_by_score != 0;
    q.order = descending ? mygramdb::query::SortOrder::DESC : mygramdb::query::SortOrder::ASC;
    q.limit = limit;
    q.offset = offset;
  }
}
}  // namespace

int mgxs_executor_warm(mgxs_executor* ex, uint32_t n_queries, const uint32_t* n_terms, const char* const* terms,
                       uint32_t limit, uint32_t offset, int sort_by_score, int descending, int rounds) {
  if (!ex || (n_queries && (!n_terms || !terms))) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_executor_warm: null argument");
  try {
    std::vector<BatchQuery> sample;
    FillQueries(&sample, n_queries, n_terms, terms, limit, offset, sort_by_score, descending);
    const auto e = ex->ex->Warm(sample, rounds);
    if (e.code() != mygram::utils::ErrorCode::kSuccess) return Fail(static_cast<int>(e.code()), e.message());
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_submit(mgxs_executor* ex, uint32_t n_queries, const uint32_t* n_terms, const char* const* terms,
                uint32_t limit, uint32_t offset, int sort_by_score, int descending, uint64_t* ticket) {
  if (ticket) *ticket = 0;
  if (!ex || !ticket || (n_queries && (!n_terms || !terms)))
    return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_submit: null argument");
  try {
    FillQueries(&ex->queries, n_queries, n_terms, terms, limit, offset, sort_by_score, descending);
    auto r = ex->ex->Submit(std::move(ex->queries));  // (hands an earlier batch's objects back: built into again above)
    if (!r) return Fail(static_cast<int>(r.error().code()), r.error().message());
    *ticket = *r;
    ex->limits[*r] = limit;
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_wait(mgxs_executor* ex, uint64_t ticket, uint64_t* totals, uint32_t* n_docs, uint32_t* docs, double* scores,
              double* timing_ms) {
  if (!ex || !totals || !n_docs || !docs) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_wait: null argument");
  try {
    BatchExecutor::Timing tm;
    const auto lim = ex->limits.find(ticket);
    const uint32_t limit = lim != ex->limits.end() ? lim->second : 0;
    if (lim != ex->limits.end()) ex->limits.erase(lim);
    const auto err = ex->ex->WaitInto(ticket, &ex->results, &tm);
    if (err.code() != mygram::utils::ErrorCode::kSuccess) return Fail(static_cast<int>(err.code()), err.message());
    const auto& res = ex->results;
    for (size_t i = 0; i < res.size(); ++i) {
      totals[i] = res[i].total;
      const size_t n = res[i].results.size();
      n_docs[i] = static_cast<uint32_t>(n);
      for (size_t k = 0; k < n && k < limit; ++k) {
        docs[i * limit + k] = res[i].results[k];
        if (scores) scores[i * limit + k] = k < res[i].scores.size() ? res[i].scores[k] : 0.0;
      }
    }
    if (timing_ms) {
      timing_ms[0] = tm.plan_ms;
      timing_ms[1] = tm.compile_ms;
      timing_ms[2] = tm.enqueue_ms;
      timing_ms[3] = tm.wait_ms;
      timing_ms[4] = static_cast<double>(tm.device_queries);
    }
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

namespace {
using mygramdb::query::FilterCondition;
using mygramdb::query::FilterOp;
using mygramdb::storage::FilterValue;

BatchQuery MakeQuery(uint32_t n_terms, const char* const* terms, uint32_t n_not, const char* const* not_terms,
                     uint32_t n_cond, const char* const* cond_columns, const uint32_t* cond_ops,
                     const char* const* cond_values, int sort_by_score, int descending, uint32_t limit, uint32_t offset) {
  BatchQuery q;
  for (uint32_t t = 0; t < n_terms; ++t) q.terms.emplace_back(terms[t]);
  for (uint32_t t = 0; t < n_not; ++t) q.not_terms.emplace_back(not_terms[t]);
  for (uint32_t c = 0; c < n_cond; ++c)
    q.filter_conditions.push_back(FilterCondition{cond_columns[c], static_cast<FilterOp>(cond_ops[c]), cond_values[c]});
  q.sort_by_score = sort_by_score != 0;
  q.order = descending ? mygramdb::query::SortOrder::DESC : mygramdb::query::SortOrder::ASC;
  q.limit = limit;
  q.offset = offset;
  return q;
}
}  // namespace

int mgxs_table_add_filter_column(mgxs_table* table, const char* name, int value_type, uint64_t n, const void* values,
                                 const char* const* strings, const uint8_t* is_null) {
  if (!table || !name || (n && !values && !strings)) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_add_filter_column: null argument");
  if (value_type < 1 || value_type > 12) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_add_filter_column: value_type is 1..12");
  try {
    std::vector<FilterValue> v(n);
    const int64_t* si = static_cast<const int64_t*>(values);
    const uint64_t* ui = static_cast<const uint64_t*>(values);
    const double* di = static_cast<const double*>(values);
    for (uint64_t i = 0; i < n; ++i) {
      if (is_null && is_null[i]) continue;  // std::monostate
      switch (value_type) {
        case 1: v[i] = si[i] != 0; break;
        case 2: v[i] = static_cast<int8_t>(si[i]); break;
        case 3: v[i] = static_cast<uint8_t>(ui[i]); break;
        case 4: v[i] = static_cast<int16_t>(si[i]); break;
        case 5: v[i] = static_cast<uint16_t>(ui[i]); break;
        case 6: v[i] = static_cast<int32_t>(si[i]); break;
        case 7: v[i] = static_cast<uint32_t>(ui[i]); break;
        case 8: v[i] = static_cast<int64_t>(si[i]); break;
        case 9: v[i] = static_cast<uint64_t>(ui[i]); break;
        case 10: v[i] = mygramdb::storage::TimeValue{si[i]}; break;
        case 11: v[i] = std::string(strings[i]); break;
        default: v[i] = di[i]; break;
      }
    }
    const std::string err = table->index->AddFilterColumn(name, v);
    return err.empty() ? MGX_OK : Fail(MGX_ERR_INVALID_ARGUMENT, err);
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

namespace {
mygramdb::storage::FilterMap MakeFilterMap(uint32_t n, const char* const* names, const int* types, const void* values,
                                           const char* const* strings) {
  mygramdb::storage::FilterMap m;
  const int64_t* si = static_cast<const int64_t*>(values);
  const uint64_t* ui = static_cast<const uint64_t*>(values);
  const double* di = static_cast<const double*>(values);
  for (uint32_t i = 0; i < n; ++i) {
    FilterValue v;
    switch (types[i]) {
      case 1: v = si[i] != 0; break;
      case 2: v = static_cast<int8_t>(si[i]); break;
      case 3: v = static_cast<uint8_t>(ui[i]); break;
      case 4: v = static_cast<int16_t>(si[i]); break;
      case 5: v = static_cast<uint16_t>(ui[i]); break;
      case 6: v = static_cast<int32_t>(si[i]); break;
      case 7: v = static_cast<uint32_t>(ui[i]); break;
      case 8: v = static_cast<int64_t>(si[i]); break;
      case 9: v = static_cast<uint64_t>(ui[i]); break;
      case 10: v = mygramdb::storage::TimeValue{si[i]}; break;
      case 11: v = std::string(strings[i]); break;
      case 12: v = di[i]; break;
      default: break;  // 0: NULL
    }
    m[names[i]] = std::move(v);
  }
  return m;
}
}  // namespace

int mgxs_table_add_document(mgxs_table* table, uint32_t doc_id, const char* text, size_t len, uint32_t n_filters,
                            const char* const* names, const int* types, const void* values, const char* const* strings) {
  if (!table || (len && !text)) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_add_document: null argument");
  try {
    const size_t stats_before = table->index->GetMutationStats().delta_documents;
    const std::string_view t(text ? text : "", len);
    if (n_filters) table->index->AddDocument(doc_id, t, MakeFilterMap(n_filters, names, types, values, strings));
    else table->index->AddDocument(doc_id, t);
    if (table->index->GetMutationStats().delta_documents == stats_before)
      return Fail(MGX_ERR_INVALID_ARGUMENT, table->index->LastError());
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_update_document(mgxs_table* table, uint32_t doc_id, const char* old_text, size_t old_len, const char* new_text,
                               size_t new_len, int with_filters, uint32_t n_filters, const char* const* names,
                               const int* types, const void* values, const char* const* strings) {
  if (!table || (old_len && !old_text) || (new_len && !new_text))
    return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_update_document: null argument");
  try {
    const std::string_view o(old_text ? old_text : "", old_len), n(new_text ? new_text : "", new_len);
    const std::string before = table->index->LastError();
    if (with_filters) table->index->UpdateDocument(doc_id, o, n, MakeFilterMap(n_filters, names, types, values, strings));
    else table->index->UpdateDocument(doc_id, o, n);
    // (the reference's signature returns nothing: a refused change shows in LastError())
    if (table->index->LastError() != before && !table->index->LastError().empty())
      return Fail(MGX_ERR_INVALID_ARGUMENT, table->index->LastError());
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_remove_document(mgxs_table* table, uint32_t doc_id, const char* text, size_t len) {
  if (!table || (len && !text)) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_remove_document: null argument");
  try {
    const std::string before = table->index->LastError();
    table->index->RemoveDocument(doc_id, std::string_view(text ? text : "", len));
    if (table->index->LastError() != before && !table->index->LastError().empty())
      return Fail(MGX_ERR_INVALID_ARGUMENT, table->index->LastError());
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_update_filters(mgxs_table* table, uint32_t doc_id, uint32_t n_filters, const char* const* names,
                              const int* types, const void* values, const char* const* strings) {
  if (!table) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_update_filters: null argument");
  try {
    if (!table->index->UpdateFilters(doc_id, MakeFilterMap(n_filters, names, types, values, strings)))
      return Fail(MGX_ERR_INVALID_ARGUMENT, table->index->LastError());
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_set_mutation_staleness(mgxs_table* table, uint64_t microseconds) {
  if (!table) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_set_mutation_staleness: null argument");
  table->index->SetMutationStaleness(std::chrono::microseconds(microseconds));
  return MGX_OK;
}

int mgxs_table_compact(mgxs_table* table) {
  if (!table) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_compact: null argument");
  try {
    const std::string err = table->index->Compact();
    return err.empty() ? MGX_OK : Fail(MGX_ERR_INTERNAL, err);
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_table_mutation_stats(mgxs_table* table, uint64_t* main_documents, uint64_t* delta_documents,
                              uint64_t* removed_from_main, uint64_t* epoch) {
  if (!table) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_table_mutation_stats: null argument");
  const auto st = table->index->GetMutationStats();
  if (main_documents) *main_documents = st.main_documents;
  if (delta_documents) *delta_documents = st.delta_documents;
  if (removed_from_main) *removed_from_main = st.removed_from_main;
  if (epoch) *epoch = st.epoch;
  return MGX_OK;
}

int mgxs_search(mgxs_table* table, uint32_t n_terms, const char* const* terms, uint32_t n_not, const char* const* not_terms,
                uint32_t n_cond, const char* const* cond_columns, const uint32_t* cond_ops, const char* const* cond_values,
                int sort_by_score, int descending, uint32_t limit, uint32_t offset, uint64_t* total, uint32_t* n_docs,
                uint32_t* docs, double* scores) {
  if (!table || !total || !n_docs || (limit && !docs)) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_search: null argument");
  try {
    const BatchQuery q = MakeQuery(n_terms, terms, n_not, not_terms, n_cond, cond_columns, cond_ops, cond_values,
                                   sort_by_score, descending, limit, offset);
    auto r = mygramdb::search_pipeline::ExecuteBatch(*table->index, {q});
    if (!r) return Fail(static_cast<int>(r.error().code()), r.error().message());
    const auto& res = (*r)[0];
    *total = res.total;
    const size_t n = limit ? std::min<size_t>(res.results.size(), limit) : 0;
    *n_docs = static_cast<uint32_t>(n);
    for (size_t k = 0; k < n; ++k) {
      docs[k] = res.results[k];
      if (scores) scores[k] = k < res.scores.size() ? res.scores[k] : 0.0;
    }
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_facet(mgxs_table* table, uint32_t n_terms, const char* const* terms, uint32_t n_not, const char* const* not_terms,
               uint32_t n_cond, const char* const* cond_columns, const uint32_t* cond_ops, const char* const* cond_values,
               const char* column, uint32_t limit, uint32_t offset, uint64_t* matched, uint64_t* total_values,
               uint32_t* n_out, uint64_t* counts, char* display, size_t display_cap, uint32_t* display_off) {
  if (!table || !column || !matched || !total_values || !n_out || (limit && (!counts || !display || !display_off)))
    return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_facet: null argument");
  try {
    const BatchQuery q = MakeQuery(n_terms, terms, n_not, not_terms, n_cond, cond_columns, cond_ops, cond_values, 0, 1, limit, offset);
    auto r = mygramdb::search_pipeline::ExecuteFacet(*table->index, q, column);
    if (!r) return Fail(static_cast<int>(r.error().code()), r.error().message());
    *matched = r->matched_documents;
    *total_values = r->total_values;
    *n_out = static_cast<uint32_t>(r->value_counts.size());
    size_t at = 0;
    for (size_t k = 0; k < r->value_counts.size(); ++k) {
      counts[k] = r->value_counts[k].second;
      display_off[k] = static_cast<uint32_t>(at);
      if (at + r->display[k].size() > display_cap) return Fail(MGX_ERR_OUT_OF_RANGE, "mgxs_facet: display buffer too small");
      std::memcpy(display + at, r->display[k].data(), r->display[k].size());
      at += r->display[k].size();
    }
    display_off[r->value_counts.size()] = static_cast<uint32_t>(at);
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_batcher_create(mgxs_table* table, uint32_t max_batch, uint32_t max_delay_us, int depth, int planner_threads,
                        mgxs_batcher** out) {
  if (out) *out = nullptr;
  if (!table || !out) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_batcher_create: null argument");
  try {
    MicroBatcher::Options o;
    o.max_batch = max_batch ? max_batch : 1024;
    o.max_delay = std::chrono::microseconds(max_delay_us);
    o.executor.depth = depth > 0 ? depth : 2;
    o.executor.planner_threads = planner_threads > 0 ? planner_threads : 1;
    auto b = std::make_unique<mgxs_batcher>();
    b->mb = std::make_unique<MicroBatcher>(*table->index, o);
    *out = b.release();
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

void mgxs_batcher_destroy(mgxs_batcher* b) { delete b; }

int mgxs_batcher_search(mgxs_batcher* b, uint32_t n_terms, const char* const* terms, uint32_t limit, uint32_t offset,
                        int sort_by_score, int descending, uint64_t* total, uint32_t* n_docs, uint32_t* docs,
                        double* scores) {
  if (!b || !total || !n_docs || (n_terms && !terms) || (limit && !docs))
    return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_batcher_search: null argument");
  try {
    BatchQuery q;
    for (uint32_t t = 0; t < n_terms; ++t) q.terms.emplace_back(terms[t]);
    q.sort_by_score = sort_by_score != 0;
    q.order = descending ? mygramdb::query::SortOrder::DESC : mygramdb::query::SortOrder::ASC;
    q.limit = limit;
    q.offset = offset;
    auto r = b->mb->Search(std::move(q));
    if (!r) return Fail(static_cast<int>(r.error().code()), r.error().message());
    *total = r->total;
    const size_t n = std::min<size_t>(r->results.size(), limit);
    *n_docs = static_cast<uint32_t>(n);
    for (size_t k = 0; k < n; ++k) {
      docs[k] = r->results[k];
      if (scores) scores[k] = k < r->scores.size() ? r->scores[k] : 0.0;
    }
    return MGX_OK;
  } catch (const std::exception& e) {
    return Fail(MGX_ERR_INTERNAL, e.what());
  }
}

int mgxs_batcher_stats(mgxs_batcher* b, uint64_t* batches, uint64_t* queries, uint64_t* closed_full,
                       uint64_t* closed_by_delay) {
  if (!b) return Fail(MGX_ERR_INVALID_ARGUMENT, "mgxs_batcher_stats: null argument");
  const MicroBatcher::Stats st = b->mb->GetStats();
  if (batches) *batches = st.batches;
  if (queries) *queries = st.queries;
  if (closed_full) *closed_full = st.closed_full;
  if (closed_by_delay) *closed_by_delay = st.closed_by_delay;
  return MGX_OK;
}

}  // extern "C"
