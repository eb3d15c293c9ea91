// mygram_shim.cpp — host C++17 implementation of mygram_shim.hpp over the C ABI of libmygram_gpu.so.
// Host-side rules are cited from the reference (paths relative to its tree); all posting work goes to the device.
#include "mygram_shim.hpp"

#include "../mgx_text.hpp"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <functional>
#include <iterator>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

#include <pthread.h>

namespace mygramdb {

using mygram::utils::Error;
using mygram::utils::ErrorCode;
using mygram::utils::Expected;
using mygram::utils::MakeError;
using mygram::utils::MakeUnexpected;

namespace {

// ---- n-gram rules over the shared text implementation (../mgx_text.hpp: UTF-8 decode string_utils.cpp:94-218,
//      IsCJKIdeograph :441-448, window rule :452-509) -------------------------------------------------------------------

struct CodePoints {
  std::vector<uint32_t> cp;
  std::vector<std::pair<uint32_t, uint32_t>> span;  // byte [begin, end) of every decoded code point
};

CodePoints Decode(std::string_view text) {
  CodePoints out;
  mgx::text::Decode(reinterpret_cast<const uint8_t*>(text.data()), text.size(), &out.cp, &out.span);
  return out;
}

using mgx::text::IsCjkIdeographPipeline;

// src/server/search_pipeline.cpp:80-136 HasUncoveredHybridFragment: a mixed-script term with a code point that no
// query n-gram covers (windows here follow the pipeline's own, wider ideograph predicate)
bool HasUncoveredHybridFragment(std::string_view normalized, int ngram_size, int kanji_ngram_size, bool cross) {
  if (normalized.empty() || kanji_ngram_size <= 0) return false;
  const int ascii_n = ngram_size > 0 ? ngram_size : 2;
  const CodePoints cps = Decode(normalized);
  const size_t n = cps.cp.size();
  if (n < 2) return false;
  bool has_cjk = false, has_other = false;
  for (uint32_t c : cps.cp) (IsCjkIdeographPipeline(c) ? has_cjk : has_other) = true;
  if (!has_cjk || !has_other) return false;
  std::vector<bool> covered(n, false);
  for (size_t i = 0; i < n; ++i) {
    const bool start_is_cjk = IsCjkIdeographPipeline(cps.cp[i]);
    const int w = start_is_cjk ? kanji_ngram_size : ascii_n;
    if (w <= 0 || i + static_cast<size_t>(w) > n) continue;
    bool crossed = false;
    if (!cross)
      for (int j = 1; j < w; ++j) crossed = crossed || IsCjkIdeographPipeline(cps.cp[i + j]) != start_is_cjk;
    if (crossed) continue;
    for (int j = 0; j < w; ++j) covered[i + j] = true;
  }
  for (bool c : covered)
    if (!c) return true;
  return false;
}

// bytes of code points [a, b): contiguous in the source when no invalid byte sits between them; re-assembled from the
// decoded spans otherwise (CodepointsToUtf8 :241-272 re-encodes, which is the same bytes for valid input)
std::string Window(std::string_view text, const CodePoints& cps, size_t a, size_t b) {
  std::string s;
  for (size_t i = a; i < b; ++i) s.append(text.substr(cps.span[i].first, cps.span[i].second - cps.span[i].first));
  return s;
}

std::vector<std::string> GenerateNgrams(std::string_view text, int n) {  // :382-423
  std::vector<std::string> out;
  const CodePoints cps = Decode(text);
  if (cps.cp.empty() || n <= 0 || cps.cp.size() < static_cast<size_t>(n)) return out;
  for (size_t i = 0; i + static_cast<size_t>(n) <= cps.cp.size(); ++i) out.push_back(Window(text, cps, i, i + n));
  return out;
}

std::vector<std::string> GenerateHybridNgrams(std::string_view text, int ascii_n, int kanji_n, bool cross) {  // :452-509
  std::vector<std::string> out;
  if (ascii_n <= 0 || kanji_n <= 0) return out;
  const CodePoints cps = Decode(text);
  for (size_t i = 0; i < cps.cp.size(); ++i) {
    const int w = mgx::text::WindowAt(cps.cp, i, ascii_n, kanji_n, cross);  // the builder's own window rule
    if (w != 0) out.push_back(Window(text, cps, i, i + static_cast<size_t>(w)));
  }
  return out;
}

std::vector<std::string> GenerateQueryNgrams(std::string_view normalized, int ngram, int kanji, bool cross) {  // :639-653
  if (kanji > 0) return GenerateHybridNgrams(normalized, ngram > 0 ? ngram : 2, kanji, cross);
  if (ngram == 0) return GenerateHybridNgrams(normalized, 2, 1, true);
  return GenerateNgrams(normalized, ngram);
}

void DeduplicateSorted(std::vector<std::string>& v) {  // string_utils.h:192-196
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
}

std::vector<storage::DocId> Take(uint32_t* p, uint64_t n) {
  std::vector<storage::DocId> v(p, p + n);
  mgx_free(p);
  return v;
}

}  // namespace

// =================================================================================================================
// storage::FilterValue keys
// =================================================================================================================

namespace storage {

std::string SerializeFilterValue(const FilterValue& value) {  // src/storage/filter_index.cpp:177-258 (little-endian host)
  return std::visit(
      [](const auto& val) -> std::string {
        using T = std::decay_t<decltype(val)>;
        auto raw = [](char tag, const void* p, size_t n) {
          std::string r(1 + n, '\0');
          r[0] = tag;
          std::memcpy(&r[1], p, n);
          return r;
        };
        if constexpr (std::is_same_v<T, std::monostate>) return std::string(1, '\x00');
        else if constexpr (std::is_same_v<T, bool>) return std::string{'\x01', val ? '\x01' : '\x00'};
        else if constexpr (std::is_same_v<T, std::string>) return std::string(1, '\x0B') + val;
        else if constexpr (std::is_same_v<T, double>) return raw('\x0C', &val, sizeof(double));
        else if constexpr (std::is_same_v<T, TimeValue>) return raw('\x0A', &val.seconds, sizeof(int64_t));
        else if constexpr (std::is_same_v<T, int8_t>) return raw('\x02', &val, 1);
        else if constexpr (std::is_same_v<T, uint8_t>) return raw('\x03', &val, 1);
        else if constexpr (std::is_same_v<T, int16_t>) return raw('\x04', &val, 2);
        else if constexpr (std::is_same_v<T, uint16_t>) return raw('\x05', &val, 2);
        else if constexpr (std::is_same_v<T, int32_t>) return raw('\x06', &val, 4);
        else if constexpr (std::is_same_v<T, uint32_t>) return raw('\x07', &val, 4);
        else if constexpr (std::is_same_v<T, int64_t>) return raw('\x08', &val, 8);
        else return raw('\x09', &val, 8);  // uint64_t
      },
      value);
}

std::string FilterValueToDisplayString(const FilterValue& value) {  // DeserializeToDisplayString, filter_index.cpp:314-407
  return std::visit(
      [](const auto& val) -> std::string {
        using T = std::decay_t<decltype(val)>;
        if constexpr (std::is_same_v<T, std::monostate>) return "NULL";
        else if constexpr (std::is_same_v<T, bool>) return val ? "true" : "false";
        else if constexpr (std::is_same_v<T, std::string>) return val;
        else if constexpr (std::is_same_v<T, TimeValue>) return std::to_string(val.seconds);
        else if constexpr (std::is_same_v<T, double>) {
          char buf[32];
          auto [ptr, ec] = std::to_chars(buf, buf + sizeof(buf), val);
          return ec != std::errc() ? std::string() : std::string(buf, ptr);
        } else return std::to_string(val);
      },
      value);
}

}  // namespace storage

// =================================================================================================================
// index::Index
// =================================================================================================================

namespace index {

namespace {
// the Index whose device this thread touched last, and the one finalised last in the process: where the index-less
// ResultSorter::SortByScore (the reference's signature) runs
thread_local const Index* tl_last_index = nullptr;
std::atomic<const Index*> g_last_finalized{nullptr};
thread_local std::string tl_device_error;
// every std::vector-returning method: cleared on entry ...
void ClearDeviceError() { tl_device_error.clear(); }
}  // namespace
const std::string& LastDeviceError() { return tl_device_error; }
const Index* LastUsedIndex() { return tl_last_index ? tl_last_index : g_last_finalized.load(); }
// ... and set where the C ABI (or the index build) reports a failure
static void SetDeviceError(const std::string& msg) { tl_device_error = msg.empty() ? "device failure" : msg; }
void ClearDeviceErrorForSorter() { ClearDeviceError(); }

struct Index::Impl {
  int device = 0;
  double dense_threshold = 0.0;
  int query_kanji = 0;  // table-config kanji size used for QUERY n-grams (the Index itself keeps kanji = ngram if 0)
  mutable std::mutex mu;
  mutable std::map<DocId, std::string> pending;  // documents recorded before the first search
  mutable mgx_columns* cols = nullptr;
  mutable mgx_index* dev = nullptr;
  mutable mgx_columns_view view{};
  mutable bool finalized = false;
  mutable std::string last_error;
  mutable bool has_gaps = false;       // ids inside [first, last] that were never added
  mutable uint32_t exists_bitmap = 0;  // filter bitmap of the ids that were (only when has_gaps)

  mutable bool owns_handles = true;  // false: adopted (Index::Adopt)
  // doc-range shards: table-wide statistics (empty / 0: this index is the whole table)
  std::vector<uint64_t> global_sizes;
  uint64_t global_docs = 0;
  double global_avgdl = 0.0;

  // One generation of the main index. `cols` / `dev` above are the current generation's; a compaction publishes a new one,
  // and whoever still holds batch objects compiled on the old device index (an executor's slots) keeps the old generation
  // alive through its own reference until those objects are gone.
  struct Handles {
    mgx_columns* cols = nullptr;
    mgx_index* dev = nullptr;
    bool owns = true;
    ~Handles() {
      if (dev && owns) mgx_index_destroy(dev);
      if (cols && owns) mgx_columns_destroy(cols);
    }
  };
  mutable std::shared_ptr<Handles> handles;
  void Publish() const {
    auto h = std::make_shared<Handles>();
    h->cols = cols;
    h->dev = dev;
    h->owns = owns_handles;
    handles = std::move(h);
  }
  ~Impl() {
    if (handles) return;  // (the generation object destroys what it owns)
    if (dev && owns_handles) mgx_index_destroy(dev);
    if (cols && owns_handles) mgx_columns_destroy(cols);
  }
  bool Lookup(std::string_view gram, uint32_t* id) const {
    int found = 0;
    if (!cols) return false;
    mgx_columns_lookup(cols, reinterpret_cast<const uint8_t*>(gram.data()), gram.size(), id, &found);
    return found != 0;
  }
  // posting size the planner sees (EstimatePostingSize / df): table-wide when the index is one shard of a table
  uint64_t Size(uint32_t id) const {
    return global_sizes.empty() ? view.offsets[id + 1] - view.offsets[id] : global_sizes[id];
  }
  // The planner's view of a gram: its id for the device and its table-wide size. A gram some OTHER shard holds resolves to
  // MGX_GRAM_ABSENT (an empty operand here) with that table-wide size, so every rank plans the same batch.
  bool Resolve(std::string_view gram, uint32_t* id, uint64_t* size) const {
    if (Lookup(gram, id)) {
      *size = Size(*id);
      return true;
    }
    if (!absent_grams.empty()) {
      const auto it = absent_grams.find(std::string(gram));
      if (it != absent_grams.end()) {
        *id = MGX_GRAM_ABSENT;
        *size = it->second;
        return true;
      }
    }
    if (fallback != nullptr) {  // a delta index: the main index's dictionary says what the rest of the table holds
      uint32_t mid = 0;
      if (fallback->Lookup(gram, &mid)) {
        *id = MGX_GRAM_ABSENT;
        *size = fallback->Size(mid);
        return true;
      }
    }
    return false;
  }
  std::unordered_map<std::string, uint64_t> absent_grams;  // grams of the table this shard has no posting for -> size
  const Impl* fallback = nullptr;

  // ---- mutable tables: changes recorded after the index was built (all under mu) ------------------------------------------
  struct Mutable {
    bool active = false;  // a change was ever recorded (the bookkeeping below exists)
    bool dirty = false;   // recorded changes the device has not seen
    std::vector<uint64_t> main_live;            // live documents of the main index, by slot
    std::vector<DocId> pending_clear;           // main documents removed / superseded since the last apply
    std::vector<uint32_t> pending_dead_grams;   // their distinct n-grams (main gram ids): one posting less each
    std::vector<DocId> pending_dead_docs;       // parallel to pending_dead_grams: the document of each posting
    std::map<DocId, std::string> delta_docs;    // documents of the delta index: table id -> normalized text
    std::map<DocId, storage::FilterMap> delta_filters;
    bool delta_changed = false;
    uint64_t main_docs = 0, main_len = 0;       // BM25Stats share of the main index's live documents
    uint64_t removed = 0, epoch = 0;
    std::chrono::microseconds staleness{0};                 // SetMutationStaleness
    std::chrono::steady_clock::time_point last_apply{};     // when the device state was last brought up to date
    uint32_t live_bitmap = 0;
    bool have_live = false;
    std::vector<uint64_t> applied_live;         // main_live as of the last application: what the DEVICE's live row says
    // background application (SetMutationStaleness > 0): a snapshot's delta is built by `builder` while queries keep
    // running on the old state; the result waits (build_ready) for the next entry point to install it
    std::thread builder;
    bool build_running = false, build_ready = false;
    std::shared_ptr<void> bg_snapshot, bg_built;  // (MutationSnapshot / BuiltDelta of the running build)
    std::shared_ptr<Index> delta;               // the delta index the device holds (nullptr: none)
    std::shared_ptr<const std::vector<DocId>> delta_ids;  // its doc map: local id - 1 -> table id, ascending
    std::vector<std::pair<uint32_t, uint64_t>> delta_contrib;  // (main gram id, the delta's posting count) inside global_sizes
  };
  mutable Mutable mut;
  mutable std::vector<uint64_t> exists_bits;  // has_gaps: the ids that were added, by slot
  bool LiveInMain(DocId doc) const {
    if (!mut.active || doc < view.first_doc_id || doc - view.first_doc_id >= view.n_docs) return false;
    const uint64_t slot = doc - view.first_doc_id;
    return (mut.main_live[slot >> 6] >> (slot & 63)) & 1;
  }
  // ... as the device's live row has it (changes recorded but not yet applied do not count): what the single operators
  // must filter by, so that they agree with the delta index the device holds
  bool LiveApplied(DocId doc) const {
    if (!mut.active || doc < view.first_doc_id || doc - view.first_doc_id >= view.n_docs) return false;
    const uint64_t slot = doc - view.first_doc_id;
    return (mut.applied_live[slot >> 6] >> (slot & 63)) & 1;
  }

  // ---- filter columns (DocumentStore filter values + FilterIndex, on the device by doc slot) -------------------------
  struct FilterColumn {
    std::string name;
    size_t type = 0;      // index of the FilterValue alternative every non-NULL value holds (0: the column is all NULL)
    uint32_t device_id = 0;
    std::vector<storage::FilterValue> dict;  // distinct non-NULL values, ascending (strings bytewise): value id = position
  };
  mutable std::mutex filter_mu;  // columns, the condition cache
  mutable std::vector<FilterColumn> filter_columns;
  mutable std::unordered_map<std::string, std::pair<uint32_t, bool>> condition_cache;  // key -> (bitmap id, negate)
  mutable std::map<DocId, storage::FilterMap> pending_filters;  // AddDocument(..., filters) before the first search
  mutable uint32_t empty_bitmap = 0;
  mutable bool have_empty_bitmap = false;

  // FilterIndex::ResolveColumnName (filter_index.cpp:122-147): exact name first, else the one unambiguous ASCII
  // case-insensitive match
  const FilterColumn* ResolveColumn(std::string_view name) const {
    const FilterColumn* hit = nullptr;
    for (const auto& c : filter_columns)
      if (c.name == name) return &c;
    for (const auto& c : filter_columns) {
      if (c.name.size() != name.size()) continue;
      bool same = true;
      for (size_t i = 0; same && i < name.size(); ++i)
        same = std::tolower(static_cast<unsigned char>(c.name[i])) == std::tolower(static_cast<unsigned char>(name[i]));
      if (!same) continue;
      if (hit) return nullptr;
      hit = &c;
    }
    return hit;
  }
};

Index::Index(int ngram_size, int kanji_ngram_size, double roaring_threshold, bool cross_boundary_ngrams,
             bool normalize_nfkc, const std::string& normalize_width, bool normalize_lower, int device)
    : ngram_size_(ngram_size),
      kanji_ngram_size_(kanji_ngram_size > 0 ? kanji_ngram_size : ngram_size),  // index.cpp:31
      cross_boundary_(cross_boundary_ngrams),
      normalize_nfkc_(normalize_nfkc),
      normalize_width_(normalize_width),
      normalize_lower_(normalize_lower),
      impl_(std::make_unique<Impl>()) {
  impl_->device = device;
  impl_->dense_threshold = roaring_threshold;
  impl_->query_kanji = kanji_ngram_size;
}

Index::~Index() {
  if (impl_->mut.builder.joinable()) impl_->mut.builder.join();  // (a delta build in flight reads this object's configuration)
  if (tl_last_index == this) tl_last_index = nullptr;
  const Index* me = this;
  g_last_finalized.compare_exchange_strong(me, nullptr);
}

std::unique_ptr<Index> Index::Adopt(mgx_columns* columns, mgx_index* device_index, int ngram_size,
                                    int kanji_ngram_size, bool cross_boundary_ngrams) {
  auto idx = std::make_unique<Index>(ngram_size, kanji_ngram_size, 0.0, cross_boundary_ngrams);
  Impl* im = idx->impl_.get();
  im->cols = columns;
  im->dev = device_index;
  im->owns_handles = false;
  im->finalized = true;
  if (mgx_columns_view_get(columns, &im->view) != MGX_OK) im->last_error = mgx_last_error();
  im->Publish();
  g_last_finalized.store(idx.get());
  return idx;
}

std::unique_ptr<Index> Index::FromDump(const void* data, size_t len, const std::string& table, std::string* error,
                                       int device) {
  auto fail = [&](const std::string& msg) {
    if (error) *error = msg;
    return std::unique_ptr<Index>();
  };
  mgx_dump* dump = nullptr;
  if (mgx_dump_open(static_cast<const uint8_t*>(data), len, table.empty() ? nullptr : table.c_str(), &dump) != MGX_OK)
    return fail(mgx_last_error());
  std::unique_ptr<mgx_dump, void (*)(mgx_dump*)> guard(dump, mgx_dump_destroy);
  mgx_dump_view v{};
  mgx_dump_view_get(dump, &v);
  const mgx_mgix_info& info = v.index_info;
  auto idx = std::make_unique<Index>(info.ngram_size, info.kanji_ngram_size, 0.0, info.cross_boundary_ngrams != 0,
                                     info.normalize_nfkc != 0, std::string(info.normalize_width), info.normalize_lower != 0, device);
  Impl* im = idx->impl_.get();
  if (mgx_dump_take_columns(dump, &im->cols) != MGX_OK || im->cols == nullptr) return fail(mgx_last_error());
  mgx_columns_view_get(im->cols, &im->view);
  const auto& cv = im->view;
  im->finalized = true;
  if (cv.n_docs == 0) return fail("the dump's table holds no document");
  mgx_index_desc d{sizeof(mgx_index_desc), MGX_ABI_VERSION, device, 0, cv.first_doc_id, cv.n_docs, cv.n_grams,
                   cv.offsets,              cv.docids,       cv.tf,  cv.doc_len, 0.0,
                   cv.tf_overflow_pos,      cv.tf_overflow_val, cv.n_tf_overflow};
  if (mgx_index_create(&d, &im->dev) != MGX_OK) return fail(mgx_last_error());
  im->Publish();
  if (v.has_texts && mgx_index_attach_text(im->dev, v.text_bytes, v.text_off) != MGX_OK) return fail(mgx_last_error());
  // DocumentStore::GetAllDocIds: the ids the store holds (deleted ids leave gaps in the range)
  if (v.n_existing != v.n_docs) {
    std::vector<DocId> existing;
    im->exists_bits.assign((v.n_docs + 63) / 64, 0);
    for (uint64_t i = 0; i < v.n_docs; ++i)
      if (v.exists[i]) {
        existing.push_back(v.first_doc_id + static_cast<DocId>(i));
        im->exists_bits[i >> 6] |= 1ull << (i & 63);
      }
    im->has_gaps = true;
    if (mgx_index_add_filter_bitmap(im->dev, existing.data(), existing.size(), &im->exists_bitmap) != MGX_OK)
      return fail(mgx_last_error());
  }
  g_last_finalized.store(idx.get());
  for (uint32_t c = 0; c < v.n_filter_columns; ++c) {
    mgx_dump_filter_column fc{};
    if (mgx_dump_filter_column_get(dump, c, &fc) != MGX_OK) return fail(mgx_last_error());
    std::vector<storage::FilterValue> vals(v.n_docs);
    for (uint64_t i = 0; i < v.n_docs; ++i) {
      if (fc.is_null[i]) continue;
      const uint64_t w = fc.values[i];
      switch (fc.value_type) {
        case 1: vals[i] = w != 0; break;
        case 2: vals[i] = static_cast<int8_t>(static_cast<int64_t>(w)); break;
        case 3: vals[i] = static_cast<uint8_t>(w); break;
        case 4: vals[i] = static_cast<int16_t>(static_cast<int64_t>(w)); break;
        case 5: vals[i] = static_cast<uint16_t>(w); break;
        case 6: vals[i] = static_cast<int32_t>(static_cast<int64_t>(w)); break;
        case 7: vals[i] = static_cast<uint32_t>(w); break;
        case 8: vals[i] = static_cast<int64_t>(w); break;
        case 9: vals[i] = static_cast<uint64_t>(w); break;
        case 10: vals[i] = storage::TimeValue{static_cast<int64_t>(w)}; break;
        case 11:
          vals[i] = std::string(reinterpret_cast<const char*>(fc.string_bytes + fc.string_off[i]),
                                static_cast<size_t>(fc.string_off[i + 1] - fc.string_off[i]));
          break;
        case 12: {
          double dv;
          std::memcpy(&dv, &w, 8);
          vals[i] = dv;
          break;
        }
        default: break;
      }
    }
    const std::string err = idx->AddFilterColumn(fc.name, vals);
    if (!err.empty()) return fail(err);
  }
  return idx;
}

namespace {
// Everything the first recorded change needs: which documents of the main index are live, its share of BM25Stats, and
// table-wide posting sizes the planner reads instead of the columns' own. Called with im->mu held, index finalised.
std::string EnsureMutable(Index::Impl* im) {
  Index::Impl::Mutable& m = im->mut;
  if (m.active) return "";
  if (!im->dev) return im->last_error.empty() ? "no device index" : im->last_error;
  if (!im->global_sizes.empty() || !im->absent_grams.empty() || im->fallback)
    return "a shard of a sharded table (SetGlobalStats) is static";
  const uint64_t n = im->view.n_docs;
  if (im->has_gaps) {
    m.main_live = im->exists_bits;
  } else {
    m.main_live.assign((n + 63) / 64, ~0ull);
    if (n & 63) m.main_live.back() = (1ull << (n & 63)) - 1;
  }
  m.applied_live = m.main_live;
  m.main_docs = im->view.bm25_doc_count;
  m.main_len = im->view.bm25_total_len;
  im->global_sizes.resize(im->view.n_grams);
  for (uint64_t g = 0; g < im->view.n_grams; ++g) im->global_sizes[g] = im->view.offsets[g + 1] - im->view.offsets[g];
  im->global_docs = m.main_docs;
  im->global_avgdl = m.main_docs ? static_cast<double>(m.main_len) / static_cast<double>(m.main_docs) : 0.0;
  m.active = true;
  return "";
}
}  // namespace

bool Index::AddDocument(DocId doc_id, std::string_view text) {
  std::lock_guard<std::mutex> lock(impl_->mu);
  if (impl_->finalized) {  // index.cpp:39-74 on a built index: the document joins the delta
    const std::string err = EnsureMutable(impl_.get());
    if (!err.empty()) {
      impl_->last_error = "AddDocument: " + err;
      return false;
    }
    if (impl_->LiveInMain(doc_id) || impl_->mut.delta_docs.count(doc_id)) {
      impl_->last_error = "AddDocument: the document id is live (UpdateDocument changes a document's text)";
      return false;
    }
    impl_->mut.delta_docs[doc_id] = std::string(text);
    impl_->mut.delta_changed = impl_->mut.dirty = true;
    return !GenerateHybridNgrams(text, ngram_size_, kanji_ngram_size_, cross_boundary_).empty();
  }
  impl_->pending[doc_id] = std::string(text);
  return !GenerateHybridNgrams(text, ngram_size_, kanji_ngram_size_, cross_boundary_).empty();  // index.cpp:39-74
}

bool Index::AddDocument(DocId doc_id, std::string_view text, const storage::FilterMap& filters) {
  bool was_live = false;
  {
    std::lock_guard<std::mutex> lock(impl_->mu);
    was_live = impl_->finalized && (impl_->LiveInMain(doc_id) || impl_->mut.delta_docs.count(doc_id) != 0);
  }
  const bool has_grams = AddDocument(doc_id, text);
  std::lock_guard<std::mutex> lock(impl_->mu);
  if (!impl_->finalized) {
    impl_->pending_filters[doc_id] = filters;
  } else if (!was_live && impl_->mut.delta_docs.count(doc_id)) {
    impl_->mut.delta_filters[doc_id] = filters;
  }
  return has_grams;
}

namespace {
storage::FilterValue FilterValueFromBits(size_t type, uint64_t w, const std::vector<storage::FilterValue>& dict) {
  switch (type) {
    case 1: return w != 0;
    case 2: return static_cast<int8_t>(static_cast<int64_t>(w));
    case 3: return static_cast<uint8_t>(w);
    case 4: return static_cast<int16_t>(static_cast<int64_t>(w));
    case 5: return static_cast<uint16_t>(w);
    case 6: return static_cast<int32_t>(static_cast<int64_t>(w));
    case 7: return static_cast<uint32_t>(w);
    case 8: return static_cast<int64_t>(w);
    case 9: return static_cast<uint64_t>(w);
    case 10: return storage::TimeValue{static_cast<int64_t>(w)};
    case 11: return w < dict.size() ? dict[w] : storage::FilterValue{};  // strings are stored as dictionary ranks
    case 12: {
      double d;
      std::memcpy(&d, &w, 8);
      return d;
    }
    default: return storage::FilterValue{};
  }
}

// RemoveDocument's bookkeeping (index.cpp:148-197: every distinct n-gram of the text loses the posting). im->mu held.
void RemoveLocked(const Index& index, Index::Impl* im, DocId doc_id, std::string_view text) {
  Index::Impl::Mutable& m = im->mut;
  const auto it = m.delta_docs.find(doc_id);
  if (it != m.delta_docs.end()) {
    m.delta_docs.erase(it);
    m.delta_filters.erase(doc_id);
    m.delta_changed = m.dirty = true;
    return;
  }
  if (!im->LiveInMain(doc_id)) return;  // not a document of the table
  const uint64_t slot = doc_id - im->view.first_doc_id;
  m.main_live[slot >> 6] &= ~(1ull << (slot & 63));
  m.pending_clear.push_back(doc_id);
  auto grams = GenerateHybridNgrams(text, index.GetNgramSize(), index.GetKanjiNgramSize(), index.GetCrossBoundaryNgrams());
  DeduplicateSorted(grams);
  for (const auto& g : grams) {
    uint32_t id = 0;
    if (im->Lookup(g, &id)) {
      m.pending_dead_grams.push_back(id);
      m.pending_dead_docs.push_back(doc_id);
    }
  }
  const uint32_t dl = im->view.doc_len[slot];
  if (dl > 0) {  // BM25Stats::RemoveDocument (binlog_event_processor.cpp:140-142)
    m.main_docs -= 1;
    m.main_len -= dl;
  }
  m.removed += 1;
  m.dirty = true;
}
}  // namespace

void Index::RemoveDocument(DocId doc_id, std::string_view text) {
  Finalize();
  std::lock_guard<std::mutex> lock(impl_->mu);
  const std::string err = EnsureMutable(impl_.get());
  if (!err.empty()) {
    impl_->last_error = "RemoveDocument: " + err;
    return;
  }
  RemoveLocked(*this, impl_.get(), doc_id, text);
}

void Index::UpdateDocument(DocId doc_id, std::string_view old_text, std::string_view new_text) {
  Finalize();
  FlushPendingFilterColumns();
  std::lock_guard<std::mutex> lock(impl_->mu);
  Impl* im = impl_.get();
  const std::string err = EnsureMutable(im);
  if (!err.empty()) {
    im->last_error = "UpdateDocument: " + err;
    return;
  }
  // the document keeps its filter values (the reference changes them through DocumentStore::UpdateDocument, a separate
  // call): a delta document has them on the host, a document of the main index has them in the device columns
  storage::FilterMap keep;
  const auto df = im->mut.delta_filters.find(doc_id);
  if (df != im->mut.delta_filters.end()) {
    keep = df->second;
  } else if (im->LiveInMain(doc_id)) {
    std::lock_guard<std::mutex> fl(im->filter_mu);
    for (const auto& col : im->filter_columns) {
      uint64_t w = 0;
      int nul = 1;
      if (mgx_index_filter_column_read(im->dev, col.device_id, doc_id, &w, &nul, nullptr) != MGX_OK) {
        im->last_error = mgx_last_error();
        continue;
      }
      if (!nul && col.type != 0) keep[col.name] = FilterValueFromBits(col.type, w, col.dict);
    }
  }
  RemoveLocked(*this, im, doc_id, old_text);
  im->mut.delta_docs[doc_id] = std::string(new_text);
  if (!keep.empty()) im->mut.delta_filters[doc_id] = std::move(keep);
  im->mut.delta_changed = im->mut.dirty = true;
}

void Index::UpdateDocument(DocId doc_id, std::string_view old_text, std::string_view new_text,
                           const storage::FilterMap& filters) {
  Finalize();
  std::lock_guard<std::mutex> lock(impl_->mu);
  Impl* im = impl_.get();
  const std::string err = EnsureMutable(im);
  if (!err.empty()) {
    im->last_error = "UpdateDocument: " + err;
    return;
  }
  RemoveLocked(*this, im, doc_id, old_text);
  im->mut.delta_docs[doc_id] = std::string(new_text);
  im->mut.delta_filters[doc_id] = filters;
  im->mut.delta_changed = im->mut.dirty = true;
}

bool Index::UpdateFilters(DocId doc_id, const storage::FilterMap& filters) {
  Finalize();
  std::lock_guard<std::mutex> lock(impl_->mu);
  Impl* im = impl_.get();
  const std::string err = EnsureMutable(im);
  if (!err.empty()) {
    im->last_error = "UpdateFilters: " + err;
    return false;
  }
  if (im->mut.delta_docs.count(doc_id)) {
    im->mut.delta_filters[doc_id] = filters;
    im->mut.delta_changed = im->mut.dirty = true;
    return true;
  }
  if (!im->LiveInMain(doc_id)) {
    im->last_error = "UpdateFilters: not a live document";
    return false;
  }
  uint64_t len = 0;
  if (mgx_index_read_text(im->dev, doc_id, nullptr, 0, &len) != MGX_OK) {
    im->last_error = std::string("UpdateFilters: ") + mgx_last_error();
    return false;
  }
  std::string text(len, '\0');
  if (len && mgx_index_read_text(im->dev, doc_id, reinterpret_cast<uint8_t*>(text.data()), len, &len) != MGX_OK) {
    im->last_error = std::string("UpdateFilters: ") + mgx_last_error();
    return false;
  }
  RemoveLocked(*this, im, doc_id, text);
  im->mut.delta_docs[doc_id] = std::move(text);
  im->mut.delta_filters[doc_id] = filters;
  im->mut.delta_changed = im->mut.dirty = true;
  return true;
}

Index::MutationStats Index::GetMutationStats() const {
  std::lock_guard<std::mutex> lock(impl_->mu);
  MutationStats st;
  const auto& m = impl_->mut;
  st.main_documents = m.active ? m.main_docs : impl_->view.bm25_doc_count;
  st.delta_documents = m.delta_docs.size();
  st.removed_from_main = m.removed;
  st.epoch = m.epoch;
  return st;
}

// The recorded changes reach the device: live row, delta index, table-wide statistics. The caller guarantees that no
// batch of this Index is being planned, compiled or run by another thread (header note).
namespace {
// What one application works from: everything recorded up to one moment, taken under the index lock.
struct MutationSnapshot {
  std::vector<DocId> clear, dead_docs;
  std::vector<uint32_t> dead_grams;
  bool delta_changed = false;
  std::map<DocId, std::string> docs;  // (copies, only when the delta changed)
  std::map<DocId, storage::FilterMap> filters;
  std::vector<std::string> column_names;
  uint64_t main_docs = 0, main_len = 0;
};
struct BuiltDelta {
  bool rebuilt = false;
  std::shared_ptr<Index> delta;  // nullptr with rebuilt: the delta is empty now
  std::shared_ptr<const std::vector<DocId>> ids;
  std::string error;
};

MutationSnapshot TakeSnapshot(Index::Impl* im) {  // im->mu held
  Index::Impl::Mutable& m = im->mut;
  MutationSnapshot snap;
  snap.clear.swap(m.pending_clear);
  snap.dead_docs.swap(m.pending_dead_docs);
  snap.dead_grams.swap(m.pending_dead_grams);
  snap.delta_changed = m.delta_changed;
  if (m.delta_changed) {
    snap.docs = m.delta_docs;
    snap.filters = m.delta_filters;
  }
  snap.main_docs = m.main_docs;
  snap.main_len = m.main_len;
  {
    std::lock_guard<std::mutex> fl(im->filter_mu);
    for (const auto& c : im->filter_columns) snap.column_names.push_back(c.name);
  }
  m.delta_changed = false;
  m.dirty = false;
  return snap;
}

// The delta index of a snapshot: column build, device index, doc map, filter columns. Touches nothing of `self` that changes
// (its configuration only), so it runs with or without the index lock — in a builder thread when changes may wait.
BuiltDelta BuildDelta(const Index& self, const MutationSnapshot& snap) {
  BuiltDelta out;
  if (!snap.delta_changed) return out;
  out.rebuilt = true;
  if (snap.docs.empty()) return out;
  Index::Impl* im = self.impl();
  auto nd = std::make_shared<Index>(self.GetNgramSize(), im->query_kanji, im->dense_threshold, self.GetCrossBoundaryNgrams(),
                                    self.GetNormalizeNfkc(), self.GetNormalizeWidth(), self.GetNormalizeLower(), im->device);
  std::vector<DocId> ids;
  ids.reserve(snap.docs.size());
  DocId local = 1;
  for (const auto& kv : snap.docs) {  // (ascending table id -> ascending local id: rank order and ties are kept)
    nd->AddDocument(local++, kv.second);
    ids.push_back(kv.first);
  }
  const std::string err = nd->Finalize();
  if (!err.empty() || !nd->impl()->dev) {
    out.error = "the delta index failed to build: " + err;
    return out;
  }
  if (mgx_index_set_doc_map(nd->impl()->dev, ids.data(), ids.size()) != MGX_OK) {
    out.error = mgx_last_error();
    return out;
  }
  std::vector<std::string> names = snap.column_names;
  for (const auto& kv : snap.filters)
    for (const auto& f : kv.second)
      if (std::find(names.begin(), names.end(), f.first) == names.end()) names.push_back(f.first);
  for (const auto& name : names) {
    std::vector<storage::FilterValue> values(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) {
      const auto fm = snap.filters.find(ids[i]);
      if (fm == snap.filters.end()) continue;
      const auto fv = fm->second.find(name);
      if (fv != fm->second.end()) values[i] = fv->second;
    }
    const std::string ferr = nd->AddFilterColumn(name, values);
    if (!ferr.empty()) {
      out.error = ferr;
      return out;
    }
  }
  nd->impl()->fallback = im;
  out.delta = std::move(nd);
  out.ids = std::make_shared<const std::vector<DocId>>(std::move(ids));
  return out;
}

// A snapshot and its delta reach the device: cleaned bitmaps, live row, the delta's handles, table-wide statistics. im->mu
// held; no batch of this Index is planned, compiled or in flight (the callers' contract).
std::string InstallMutations(Index::Impl* im, MutationSnapshot& snap, BuiltDelta& built) {
  Index::Impl::Mutable& m = im->mut;
  static const bool kTrace = std::getenv("MGX_TRACE_HOST") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  m.last_apply = t_prev;
  auto lap = [&](const char* what) {
    if (!kTrace) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[shim] ApplyMutations %s: %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  auto fail = [&](const std::string& msg) {
    im->last_error = "ApplyMutations: " + msg;
    return im->last_error;
  };
  if (!built.error.empty()) return fail(built.error);
  if (mgx_index_synchronize(im->dev) != MGX_OK) return fail(mgx_last_error());
  // ---- the dead documents' postings leave the bitmap form of their dense grams (a query over bitmap-form grams then
  // needs no live-row operand), then their live bits go --------------------------------------------------------------
  if (!snap.dead_grams.empty() &&
      mgx_index_clear_postings(im->dev, snap.dead_docs.data(), snap.dead_grams.data(), snap.dead_grams.size()) != MGX_OK)
    return fail(mgx_last_error());
  for (DocId d : snap.clear) {
    const uint64_t slot = d - im->view.first_doc_id;
    m.applied_live[slot >> 6] &= ~(1ull << (slot & 63));
  }
  // ---- the live row of the main index ---------------------------------------------------------------------------------
  if (!m.have_live) {
    std::vector<DocId> live;
    live.reserve(im->view.n_docs);
    for (uint64_t slot = 0; slot < im->view.n_docs; ++slot)
      if ((m.applied_live[slot >> 6] >> (slot & 63)) & 1) live.push_back(im->view.first_doc_id + static_cast<DocId>(slot));
    if (mgx_index_add_filter_bitmap(im->dev, live.data(), live.size(), &m.live_bitmap) != MGX_OK) return fail(mgx_last_error());
    if (mgx_index_set_live_bitmap(im->dev, m.live_bitmap, MGX_LIVE_BITMAPS_CLEAN) != MGX_OK) return fail(mgx_last_error());
    m.have_live = true;
    im->has_gaps = false;  // (the live row is also the NOT universe: ids never added are not in it)
  } else if (!snap.clear.empty()) {
    if (mgx_index_update_filter_bitmap(im->dev, m.live_bitmap, nullptr, 0, snap.clear.data(), snap.clear.size()) != MGX_OK)
      return fail(mgx_last_error());
  }
  lap("sync + live row + cleaned bitmaps");
  // ---- posting sizes: the main index's live postings, then the delta's on top -------------------------------------------
  for (const auto& c : m.delta_contrib) im->global_sizes[c.first] -= c.second;
  m.delta_contrib.clear();
  for (uint32_t gid : snap.dead_grams)
    if (im->global_sizes[gid] > 0) im->global_sizes[gid] -= 1;
  // ---- the delta index ------------------------------------------------------------------------------------------------
  if (built.rebuilt) {
    m.delta = built.delta;
    m.delta_ids = built.ids;
  }
  lap("delta index handles");
  // ---- table-wide statistics in both ------------------------------------------------------------------------------------
  uint64_t n_docs = snap.main_docs, total_len = snap.main_len;
  im->absent_grams.clear();
  std::vector<uint64_t> delta_sizes;
  if (m.delta) {
    const mgx_columns_view& dv = m.delta->impl()->view;
    n_docs += dv.bm25_doc_count;
    total_len += dv.bm25_total_len;
    delta_sizes.resize(dv.n_grams);
    for (uint64_t g = 0; g < dv.n_grams; ++g) {
      const std::string_view key(reinterpret_cast<const char*>(dv.key_bytes) + dv.key_off[g], dv.key_off[g + 1] - dv.key_off[g]);
      const uint64_t sz = dv.offsets[g + 1] - dv.offsets[g];
      uint32_t mid = 0;
      if (im->Lookup(key, &mid)) {
        im->global_sizes[mid] += sz;
        m.delta_contrib.emplace_back(mid, sz);
        delta_sizes[g] = im->global_sizes[mid];
      } else {
        im->absent_grams.emplace(std::string(key), sz);
        delta_sizes[g] = sz;
      }
    }
  }
  im->global_docs = n_docs;
  im->global_avgdl = n_docs ? static_cast<double>(total_len) / static_cast<double>(n_docs) : 0.0;
  if (m.delta) {
    Index::Impl* dm = m.delta->impl();
    dm->global_sizes = std::move(delta_sizes);
    dm->global_docs = im->global_docs;
    dm->global_avgdl = im->global_avgdl;
    if (mgx_index_invalidate_statistics(dm->dev) != MGX_OK) return fail(mgx_last_error());
  }
  if (mgx_index_invalidate_statistics(im->dev) != MGX_OK) return fail(mgx_last_error());
  lap("statistics");
  m.epoch += 1;
  return "";
}
}  // namespace

void Index::SetMutationStaleness(std::chrono::microseconds max_staleness) {
  std::lock_guard<std::mutex> lock(impl_->mu);
  impl_->mut.staleness = max_staleness;
}

// With no staleness bound (and always the first time): snapshot, build, install — in this call, under the lock. With a
// bound: the snapshot's delta is built by a thread while queries keep running on the old state, and installed by the first
// entry point that finds it ready — the stall a steady writer causes is the few milliseconds of the installation, not the
// delta build.
std::string Index::ApplyMutations(bool force) const {
  Impl* im = impl_.get();
  std::unique_lock<std::mutex> lock(im->mu);
  Impl::Mutable& m = im->mut;
  if (m.staleness.count() == 0) force = true;  // (no bound: every change recorded before this call is applied by it)
  auto install_background = [&]() -> std::string {
    if (m.builder.joinable()) {
      lock.unlock();  // (the builder takes the lock to publish its result)
      m.builder.join();
      lock.lock();
    }
    auto snap = std::static_pointer_cast<MutationSnapshot>(m.bg_snapshot);
    auto built = std::static_pointer_cast<BuiltDelta>(m.bg_built);
    m.bg_snapshot.reset();
    m.bg_built.reset();
    m.build_running = m.build_ready = false;
    return snap && built ? InstallMutations(im, *snap, *built) : std::string();
  };
  if (m.build_running && (m.build_ready || force)) {
    const std::string e = install_background();
    if (!e.empty()) return e;
  }
  if (m.build_running) return "";  // (a build is under way: its result is installed by a later call)
  if (!m.dirty) return "";
  const auto now = std::chrono::steady_clock::now();
  const bool background = !force && m.staleness.count() > 0 && m.epoch > 0;
  if (background && now - m.last_apply < m.staleness) return "";
  if (background) {
    auto snap = std::make_shared<MutationSnapshot>(TakeSnapshot(im));
    m.bg_snapshot = snap;
    m.build_running = true;
    m.build_ready = false;
    const Index* self = this;
    m.builder = std::thread([self, im, snap] {
      pthread_setname_np(pthread_self(), "mgx-delta");
      auto built = std::make_shared<BuiltDelta>(BuildDelta(*self, *snap));
      std::lock_guard<std::mutex> l(im->mu);
      im->mut.bg_built = built;
      im->mut.build_ready = true;
    });
    return "";
  }
  MutationSnapshot snap = TakeSnapshot(im);
  // (the lock stays held through the build: a writer thread recording a change meanwhile waits, and is not marked applied)
  BuiltDelta built = BuildDelta(*this, snap);
  return InstallMutations(im, snap, built);
}

// The main index rebuilt from the table's current documents: live documents of the old main index (their texts come back
// from the device, where BM25 keeps them) + the delta's. Same quiescence contract as ApplyMutations. Batch objects an
// executor compiled on the old device index stay valid until that executor lets go of them (Impl::Handles).
std::string Index::Compact() const {
  Finalize();
  FlushPendingFilterColumns();
  {
    const std::string e = ApplyMutations(/*force=*/true);
    if (!e.empty()) return e;
  }
  Impl* im = impl_.get();
  std::unique_lock<std::mutex> lock(im->mu);
  Impl::Mutable& m = im->mut;
  if (!m.active) return "";  // the table never changed
  auto fail = [&](const std::string& msg) {
    im->last_error = "Compact: " + msg;
    return im->last_error;
  };
  const uint64_t n_old = im->view.n_docs;
  uint64_t total = 0;
  if (mgx_index_copy_text(im->dev, nullptr, 0, nullptr, &total) != MGX_OK) return fail(mgx_last_error());
  std::vector<uint8_t> old_bytes(total + 16);
  std::vector<uint64_t> old_off(n_old + 1);
  if (mgx_index_copy_text(im->dev, old_bytes.data(), total, old_off.data(), &total) != MGX_OK) return fail(mgx_last_error());
  const DocId old_first = im->view.first_doc_id;
  DocId first = old_first, last = old_first + static_cast<DocId>(n_old) - 1;
  if (!m.delta_docs.empty()) {
    first = std::min(first, m.delta_docs.begin()->first);
    last = std::max(last, m.delta_docs.rbegin()->first);
  }
  const uint64_t n2 = static_cast<uint64_t>(last) - first + 1;
  std::vector<uint8_t> bytes;
  bytes.reserve(total + 1024);
  std::vector<uint64_t> off(n2 + 1, 0);
  std::vector<uint64_t> exists((n2 + 63) / 64, 0);
  std::vector<DocId> existing;
  {
    auto dit = m.delta_docs.begin();
    for (uint64_t i = 0; i < n2; ++i) {
      const DocId doc = first + static_cast<DocId>(i);
      while (dit != m.delta_docs.end() && dit->first < doc) ++dit;
      bool here = false;
      if (dit != m.delta_docs.end() && dit->first == doc) {
        bytes.insert(bytes.end(), dit->second.begin(), dit->second.end());
        here = true;
      } else if (im->LiveInMain(doc)) {
        const uint64_t slot = doc - old_first;
        bytes.insert(bytes.end(), old_bytes.begin() + static_cast<std::ptrdiff_t>(old_off[slot]),
                     old_bytes.begin() + static_cast<std::ptrdiff_t>(old_off[slot + 1]));
        here = true;
      }
      if (here) {
        exists[i >> 6] |= 1ull << (i & 63);
        existing.push_back(doc);
      }
      off[i + 1] = bytes.size();
    }
  }
  bytes.resize(bytes.size() + 16);
  std::vector<uint8_t>().swap(old_bytes);
  // ---- the old generation's filter columns, by doc slot ----------------------------------------------------------------
  struct OldColumn {
    Impl::FilterColumn meta;
    std::vector<uint64_t> values;
    std::vector<uint8_t> nul;
  };
  std::vector<OldColumn> old_cols;
  {
    std::lock_guard<std::mutex> fl(im->filter_mu);
    for (const auto& c : im->filter_columns) {
      OldColumn oc;
      oc.meta = c;
      oc.values.resize(n_old);
      oc.nul.resize(n_old);
      if (mgx_index_filter_column_export(im->dev, c.device_id, oc.values.data(), oc.nul.data(), nullptr) != MGX_OK)
        return fail(mgx_last_error());
      old_cols.push_back(std::move(oc));
    }
  }
  // ---- the new generation ----------------------------------------------------------------------------------------------
  mgx_build_params bp{sizeof(mgx_build_params), MGX_ABI_VERSION, ngram_size_, kanji_ngram_size_, cross_boundary_ ? 1 : 0, 0};
  mgx_columns* cols2 = nullptr;
  if (mgx_columns_build(&bp, bytes.data(), off.data(), first, n2, &cols2) != MGX_OK) return fail(mgx_last_error());
  mgx_columns_view view2{};
  mgx_columns_view_get(cols2, &view2);
  mgx_index_desc d{sizeof(mgx_index_desc), MGX_ABI_VERSION, im->device, 0, view2.first_doc_id, view2.n_docs, view2.n_grams,
                   view2.offsets,            view2.docids,    view2.tf,   view2.doc_len, im->dense_threshold,
                   view2.tf_overflow_pos,    view2.tf_overflow_val, view2.n_tf_overflow};
  mgx_index* dev2 = nullptr;
  if (mgx_index_create(&d, &dev2) != MGX_OK) {
    mgx_columns_destroy(cols2);
    return fail(mgx_last_error());
  }
  auto gen = std::make_shared<Impl::Handles>();
  gen->cols = cols2;
  gen->dev = dev2;
  gen->owns = true;
  if (mgx_index_attach_text(dev2, bytes.data(), off.data()) != MGX_OK) return fail(mgx_last_error());
  const bool gaps = existing.size() != n2;
  uint32_t exists_bitmap = 0;
  if (gaps && mgx_index_add_filter_bitmap(dev2, existing.data(), existing.size(), &exists_bitmap) != MGX_OK)
    return fail(mgx_last_error());
  // what the re-added filter columns need of the old state
  const std::vector<uint64_t> old_live = m.main_live;
  const std::map<DocId, storage::FilterMap> delta_filters = m.delta_filters;
  std::vector<DocId> delta_ids;
  for (const auto& kv : m.delta_docs) delta_ids.push_back(kv.first);
  // ---- swap ----------------------------------------------------------------------------------------------------------------
  const std::shared_ptr<Impl::Handles> old_gen = im->handles;  // (let go at the end of this call, or by the last executor slot)
  im->handles = gen;
  im->cols = cols2;
  im->dev = dev2;
  im->view = view2;
  im->owns_handles = true;
  im->has_gaps = gaps;
  im->exists_bitmap = exists_bitmap;
  im->exists_bits = gaps ? exists : std::vector<uint64_t>();
  im->global_sizes.clear();
  im->absent_grams.clear();
  im->global_docs = 0;
  im->global_avgdl = 0.0;
  const uint64_t epoch = m.epoch + 1;
  const auto staleness = m.staleness;
  m = Impl::Mutable{};
  m.epoch = epoch;
  m.staleness = staleness;
  {
    std::lock_guard<std::mutex> fl(im->filter_mu);
    im->filter_columns.clear();
    im->condition_cache.clear();
    im->have_empty_bitmap = false;
  }
  lock.unlock();
  // ---- filter columns of the new generation: a live document of the old main index keeps its value, a document of the delta
  // has the values recorded with it ----------------------------------------------------------------------------------------------
  std::vector<std::string> names;
  for (const auto& oc : old_cols) names.push_back(oc.meta.name);
  for (const auto& kv : delta_filters)
    for (const auto& f : kv.second)
      if (std::find(names.begin(), names.end(), f.first) == names.end()) names.push_back(f.first);
  for (const auto& name : names) {
    std::vector<storage::FilterValue> values(n2);
    const OldColumn* oc = nullptr;
    for (const auto& c : old_cols)
      if (c.meta.name == name) oc = &c;
    if (oc != nullptr && oc->meta.type != 0) {
      for (uint64_t slot = 0; slot < n_old; ++slot) {
        if (!((old_live[slot >> 6] >> (slot & 63)) & 1) || oc->nul[slot]) continue;
        values[static_cast<uint64_t>(old_first - first) + slot] = FilterValueFromBits(oc->meta.type, oc->values[slot], oc->meta.dict);
      }
    }
    for (DocId doc : delta_ids) {
      storage::FilterValue v;
      const auto fm = delta_filters.find(doc);
      if (fm != delta_filters.end()) {
        const auto fv = fm->second.find(name);
        if (fv != fm->second.end()) v = fv->second;
      }
      values[doc - first] = std::move(v);
    }
    const std::string ferr = AddFilterColumn(name, values);
    if (!ferr.empty()) return "Compact: " + ferr;
  }
  return "";
}

namespace {
// widened device value of a non-NULL FilterValue (strings: their dictionary rank, filled in by the caller)
uint64_t WidenFilterValue(const storage::FilterValue& v) {
  return std::visit(
      [](const auto& val) -> uint64_t {
        using T = std::decay_t<decltype(val)>;
        if constexpr (std::is_same_v<T, std::monostate> || std::is_same_v<T, std::string>) return 0;
        else if constexpr (std::is_same_v<T, storage::TimeValue>) return static_cast<uint64_t>(val.seconds);
        else if constexpr (std::is_same_v<T, double>) {
          uint64_t u;
          std::memcpy(&u, &val, 8);
          return u;
        } else if constexpr (std::is_signed_v<T> || std::is_same_v<T, bool>) return static_cast<uint64_t>(static_cast<int64_t>(val));
        else return static_cast<uint64_t>(val);
      },
      v);
}
uint32_t FilterClassOf(size_t type) {  // MGX_FC_* of a FilterValue alternative
  switch (type) {
    case 12: return MGX_FC_DOUBLE;
    case 3: case 5: case 7: case 9: case 11: return MGX_FC_UNSIGNED;
    default: return MGX_FC_SIGNED;  // bool, int8/16/32/64, TimeValue
  }
}
}  // namespace

std::string Index::AddFilterColumn(const std::string& name, const std::vector<storage::FilterValue>& values) const {
  Finalize();
  Impl* im = impl_.get();
  if (!im->dev) return im->last_error.empty() ? "AddFilterColumn: no device index" : im->last_error;
  if (values.size() != im->view.n_docs) return "AddFilterColumn: one value per doc slot of the index";
  Impl::FilterColumn col;
  col.name = name;
  for (const auto& v : values) {
    if (v.index() == 0) continue;
    if (col.type == 0) col.type = v.index();
    if (v.index() != col.type) return "AddFilterColumn: the values of one column must hold one type";
  }
  std::vector<storage::FilterValue> nn;
  for (const auto& v : values)
    if (v.index() != 0) nn.push_back(v);
  std::sort(nn.begin(), nn.end());
  nn.erase(std::unique(nn.begin(), nn.end()), nn.end());
  col.dict = std::move(nn);
  const size_t n = values.size();
  std::vector<uint64_t> wide(n, 0);
  std::vector<uint8_t> nul(n, 0);
  std::vector<uint32_t> ids(n, 0xFFFFFFFFu);
  for (size_t i = 0; i < n; ++i) {
    if (values[i].index() == 0) {
      nul[i] = 1;
      continue;
    }
    const auto it = std::lower_bound(col.dict.begin(), col.dict.end(), values[i]);
    ids[i] = static_cast<uint32_t>(it - col.dict.begin());
    wide[i] = col.type == 11 ? ids[i] : WidenFilterValue(values[i]);  // strings compare by their bytewise rank
  }
  mgx_filter_column_desc d{sizeof(mgx_filter_column_desc), MGX_ABI_VERSION, FilterClassOf(col.type),
                           static_cast<uint32_t>(col.dict.size()), wide.data(), nul.data(), ids.data()};
  if (mgx_index_add_filter_column(im->dev, &d, &col.device_id) != MGX_OK) return mgx_last_error();
  std::lock_guard<std::mutex> lock(im->filter_mu);
  for (auto& c : im->filter_columns)
    if (c.name == name) {
      c = std::move(col);  // (replaced: cached conditions of the old column are dropped)
      im->condition_cache.clear();
      return "";
    }
  im->filter_columns.push_back(std::move(col));
  return "";
}

void Index::AddDocumentBatch(const std::vector<DocumentItem>& documents) {
  for (const auto& d : documents) AddDocument(d.doc_id, d.text);
}

std::string Index::Finalize() const {
  tl_last_index = this;  // (every search / scoring entry point passes through here)
  std::lock_guard<std::mutex> lock(impl_->mu);
  if (impl_->finalized) return impl_->last_error;
  impl_->finalized = true;
  g_last_finalized.store(this);
  // dense id range [first, last]; ids never added are documents without text
  const DocId first = impl_->pending.empty() ? 1 : impl_->pending.begin()->first;
  const DocId last = impl_->pending.empty() ? 1 : impl_->pending.rbegin()->first;
  const uint64_t n = static_cast<uint64_t>(last) - first + 1;
  std::vector<uint8_t> bytes;
  std::vector<uint64_t> off(n + 1, 0);
  {
    auto it = impl_->pending.begin();
    for (uint64_t i = 0; i < n; ++i) {
      if (it != impl_->pending.end() && it->first == first + i) {
        bytes.insert(bytes.end(), it->second.begin(), it->second.end());
        ++it;
      }
      off[i + 1] = bytes.size();
    }
  }
  bytes.resize(bytes.size() + 16);
  std::vector<DocId> existing;
  impl_->has_gaps = impl_->pending.size() != n && !impl_->pending.empty();
  if (impl_->has_gaps) {
    impl_->exists_bits.assign((n + 63) / 64, 0);
    for (const auto& kv : impl_->pending) {
      existing.push_back(kv.first);
      const uint64_t slot = kv.first - first;
      impl_->exists_bits[slot >> 6] |= 1ull << (slot & 63);
    }
  }
  impl_->pending.clear();
  mgx_build_params bp{sizeof(mgx_build_params), MGX_ABI_VERSION, ngram_size_, kanji_ngram_size_, cross_boundary_ ? 1 : 0, 0};
  static const bool kTraceBuild = std::getenv("MGX_TRACE_HOST") != nullptr;
  const auto tb0 = std::chrono::steady_clock::now();
  if (mgx_columns_build(&bp, bytes.data(), off.data(), first, n, &impl_->cols) != MGX_OK) {
    impl_->last_error = mgx_last_error();
    return impl_->last_error;
  }
  const auto tb1 = std::chrono::steady_clock::now();
  mgx_columns_view_get(impl_->cols, &impl_->view);
  const auto& v = impl_->view;
  mgx_index_desc d{sizeof(mgx_index_desc), MGX_ABI_VERSION, impl_->device, 0, v.first_doc_id, v.n_docs, v.n_grams,
                   v.offsets,               v.docids,        v.tf,          v.doc_len, impl_->dense_threshold,
                   v.tf_overflow_pos,       v.tf_overflow_val, v.n_tf_overflow};
  if (mgx_index_create(&d, &impl_->dev) != MGX_OK) {
    impl_->last_error = mgx_last_error();
    impl_->dev = nullptr;
    return impl_->last_error;
  }
  impl_->Publish();
  const auto tb2 = std::chrono::steady_clock::now();
  // the shim's Index is also the DocumentStore of the texts it was given: BM25 terms longer than one n-gram are
  // counted in the text on the device
  if (mgx_index_attach_text(impl_->dev, bytes.data(), off.data()) != MGX_OK) impl_->last_error = mgx_last_error();
  if (kTraceBuild)
    fprintf(stderr, "[shim] Finalize of %llu docs: columns %.2f ms, device index %.2f ms, texts %.2f ms\n",
            static_cast<unsigned long long>(n), std::chrono::duration<double, std::milli>(tb1 - tb0).count(),
            std::chrono::duration<double, std::milli>(tb2 - tb1).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb2).count());
  // DocumentStore::GetAllDocIds (the NOT universe of boolean expressions) is the set of ids that were added
  if (impl_->has_gaps &&
      mgx_index_add_filter_bitmap(impl_->dev, existing.data(), existing.size(), &impl_->exists_bitmap) != MGX_OK)
    impl_->last_error = mgx_last_error();
  pending_columns_ready_ = !impl_->pending_filters.empty();
  return impl_->last_error;
}

// The filter values recorded by AddDocument(doc, text, filters) become device columns (outside impl_->mu: AddFilterColumn
// calls Finalize itself).
void Index::FlushPendingFilterColumns() const {
  std::map<DocId, storage::FilterMap> pending;
  {
    std::lock_guard<std::mutex> lock(impl_->mu);
    if (!pending_columns_ready_) return;
    pending_columns_ready_ = false;
    pending.swap(impl_->pending_filters);
  }
  std::vector<std::string> names;
  for (const auto& kv : pending)
    for (const auto& f : kv.second)
      if (std::find(names.begin(), names.end(), f.first) == names.end()) names.push_back(f.first);
  const DocId first = impl_->view.first_doc_id;
  for (const auto& name : names) {
    std::vector<storage::FilterValue> values(impl_->view.n_docs);
    for (const auto& kv : pending) {
      const auto it = kv.second.find(name);
      if (it != kv.second.end() && kv.first >= first && kv.first - first < values.size()) values[kv.first - first] = it->second;
    }
    const std::string err = AddFilterColumn(name, values);
    if (!err.empty()) impl_->last_error = err;
  }
}

const std::string& Index::LastError() const { return impl_->last_error; }

void Index::SetNormalization(bool nfkc, const std::string& width, bool lower) {
  normalize_nfkc_ = nfkc;
  normalize_width_ = width;
  normalize_lower_ = lower;
}

uint64_t Index::PostingSize(std::string_view term) const {  // index.cpp:580-584
  Finalize();
  bool mutable_table = false;
  {
    std::lock_guard<std::mutex> lock(impl_->mu);
    mutable_table = impl_->mut.active;
  }
  if (mutable_table) ApplyMutations();  // (the live postings of both indexes: what Count() is in the reference)
  uint32_t id = 0;
  uint64_t size = 0;
  return impl_->Resolve(term, &id, &size) ? size : 0;
}
uint64_t Index::EstimatePostingSize(std::string_view term) const { return PostingSize(term); }  // index.cpp:756-759

void Index::SetAbsentGrams(std::unordered_map<std::string, uint64_t> grams) { impl_->absent_grams = std::move(grams); }

std::string Index::SetGlobalStats(uint64_t total_docs, double avg_doc_length, std::vector<uint64_t> global_posting_sizes) {
  Finalize();
  if (global_posting_sizes.size() != impl_->view.n_grams) return "SetGlobalStats: one size per gram of this shard";
  impl_->global_sizes = std::move(global_posting_sizes);
  impl_->global_docs = total_docs;
  impl_->global_avgdl = avg_doc_length;
  return "";
}

uint64_t Index::Bm25DocCount() const {
  Finalize();
  return impl_->global_sizes.empty() ? impl_->view.bm25_doc_count : impl_->global_docs;
}
double Index::Bm25AvgDocLength() const {  // server_types.h:182-187
  Finalize();
  if (!impl_->global_sizes.empty()) return impl_->global_avgdl;
  return impl_->view.bm25_doc_count
             ? static_cast<double>(impl_->view.bm25_total_len) / static_cast<double>(impl_->view.bm25_doc_count)
             : 0.0;
}

Expected<uint32_t, Error> Index::AddFilterBitmap(const std::vector<DocId>& docs) const {
  Finalize();
  if (!impl_->dev) return MakeUnexpected(MakeError(ErrorCode::kInternalError, impl_->last_error));
  uint32_t id = 0;
  const int rc = mgx_index_add_filter_bitmap(impl_->dev, docs.data(), docs.size(), &id);
  if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
  return id;
}

// ---- single operators on a mutable table ---------------------------------------------------------------------------------
// The operators below answer from the column arrays of ONE device index. A table with recorded changes answers with the
// main index's result minus its dead documents plus the delta index's result under table ids — both ascending and
// disjoint, so one std::merge; the set semantics are per document, so evaluating each index on its own is exact.
namespace {
thread_local bool tl_raw_operators = false;  // inside OverBoth: the operators answer from this index's arrays alone
struct RawScope {
  bool prev;
  RawScope() : prev(tl_raw_operators) { tl_raw_operators = true; }
  ~RawScope() { tl_raw_operators = prev; }
};

bool IsMutable(const Index& index) {
  if (tl_raw_operators) return false;
  Index::Impl* im = index.impl();
  std::lock_guard<std::mutex> lock(im->mu);
  return im->mut.active;
}

struct DeltaView {
  std::shared_ptr<Index> delta;
  std::shared_ptr<const std::vector<DocId>> ids;
};
DeltaView ViewOfDelta(const Index& index) {
  Index::Impl* im = index.impl();
  std::lock_guard<std::mutex> lock(im->mu);
  return DeltaView{im->mut.delta, im->mut.delta_ids};
}

// op(index) -> ascending doc ids of that one index
template <typename Op>
std::vector<DocId> OverBoth(const Index& index, Op op) {
  const std::string err = index.ApplyMutations();
  if (!err.empty()) {
    SetDeviceError(err);
    return {};
  }
  RawScope raw;
  std::vector<DocId> a = op(index);
  if (!tl_device_error.empty()) return {};
  Index::Impl* im = index.impl();
  {
    std::lock_guard<std::mutex> lock(im->mu);
    a.erase(std::remove_if(a.begin(), a.end(), [&](DocId d) { return !im->LiveApplied(d); }), a.end());
  }
  const DeltaView dv = ViewOfDelta(index);
  if (!dv.delta) return a;
  std::vector<DocId> b = op(*dv.delta);
  if (!tl_device_error.empty()) return {};
  for (DocId& d : b) d = (*dv.ids)[d - 1];
  std::vector<DocId> out(a.size() + b.size());
  std::merge(a.begin(), a.end(), b.begin(), b.end(), out.begin());
  return out;
}

// The candidates of a caller's list by the index that holds them: documents of the main index that are live, documents of
// the delta index under its local ids; anything else is not a document of the table. ApplyMutations has run.
struct SplitCandidates {
  DeltaView view;
  std::vector<DocId> main_docs, delta_docs;
  std::vector<size_t> main_pos, delta_pos;  // positions in the caller's list
  SplitCandidates(const Index& index, const std::vector<DocId>& candidates) : view(ViewOfDelta(index)) {
    Index::Impl* im = index.impl();
    std::lock_guard<std::mutex> lock(im->mu);
    for (size_t i = 0; i < candidates.size(); ++i) {
      const DocId c = candidates[i];
      if (im->LiveApplied(c)) {
        main_docs.push_back(c);
        main_pos.push_back(i);
      } else if (view.ids) {
        const auto it = std::lower_bound(view.ids->begin(), view.ids->end(), c);
        if (it != view.ids->end() && *it == c) {
          delta_docs.push_back(static_cast<DocId>(it - view.ids->begin()) + 1);
          delta_pos.push_back(i);
        }
      }
    }
  }
  // `kept` is `given` with some documents taken out (every copy of a document, or none): mark what stayed
  static void MarkKept(const std::vector<DocId>& given, const std::vector<size_t>& pos, const std::vector<DocId>& kept,
                       std::vector<uint8_t>* keep) {
    size_t k = 0;
    for (size_t i = 0; i < given.size() && k < kept.size(); ++i)
      if (given[i] == kept[k]) {
        (*keep)[pos[i]] = 1;
        ++k;
      }
  }
};

// PostingList::GetTopN / the tail of Index::SearchAnd (posting_list.cpp:476-514, index.cpp:352-366) over an ascending list
std::vector<DocId> FinishTopN(std::vector<DocId> all, size_t limit, bool reverse) {
  if (limit > 0 && all.size() > limit) {
    if (reverse) {
      all.erase(all.begin(), all.begin() + static_cast<std::ptrdiff_t>(all.size() - limit));
      std::reverse(all.begin(), all.end());
    } else {
      all.resize(limit);
    }
  } else if (reverse) {
    std::reverse(all.begin(), all.end());
  }
  return all;
}
}  // namespace

std::vector<DocId> Index::SearchAnd(const std::vector<std::string>& terms, size_t limit, bool reverse) const {
  ClearDeviceError();
  if (terms.empty()) return {};  // index.cpp:203
  Finalize();
  if (!impl_->dev) {
    SetDeviceError(impl_->last_error);
    return {};
  }
  if (IsMutable(*this))
    return FinishTopN(OverBoth(*this, [&](const Index& ix) { return ix.SearchAnd(terms, 0, false); }), limit, reverse);
  std::vector<uint32_t> ids;
  for (const auto& t : terms) {
    uint32_t id = 0;
    if (!impl_->Lookup(t, &id)) return {};  // :211-215 unknown term
    ids.push_back(id);
  }
  uint32_t* out = nullptr;
  uint64_t n = 0;
  if (mgx_and(impl_->dev, ids.data(), static_cast<uint32_t>(ids.size()), limit, reverse ? 1 : 0, &out, &n) != MGX_OK) {
    impl_->last_error = mgx_last_error();
    SetDeviceError(impl_->last_error);
    return {};
  }
  return Take(out, n);
}

std::vector<DocId> Index::SearchOr(const std::vector<std::string>& terms) const {  // index.cpp:418-448
  ClearDeviceError();
  if (terms.empty()) return {};
  Finalize();
  if (!impl_->dev) {
    SetDeviceError(impl_->last_error);
    return {};
  }
  if (IsMutable(*this)) return OverBoth(*this, [&](const Index& ix) { return ix.SearchOr(terms); });
  std::vector<uint32_t> ids;
  for (const auto& t : terms) {
    uint32_t id = 0;
    if (impl_->Lookup(t, &id)) ids.push_back(id);  // unknown terms are skipped
  }
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  if (ids.empty()) return {};
  uint32_t* out = nullptr;
  uint64_t n = 0;
  if (mgx_or(impl_->dev, ids.data(), static_cast<uint32_t>(ids.size()), &out, &n) != MGX_OK) {
    impl_->last_error = mgx_last_error();
    SetDeviceError(impl_->last_error);
    return {};
  }
  return Take(out, n);
}

std::vector<DocId> Index::SearchNot(const std::vector<DocId>& all_docs, const std::vector<std::string>& terms) const {
  ClearDeviceError();
  if (terms.empty()) return all_docs;  // index.cpp:451-453
  Finalize();
  if (!impl_->dev) {
    SetDeviceError(impl_->last_error);
    return {};
  }
  if (IsMutable(*this)) {  // index.cpp:456-484 as written: the union of the terms' lists, then std::set_difference
    const std::vector<DocId> uni = OverBoth(*this, [&](const Index& ix) { return ix.SearchOr(terms); });
    if (!tl_device_error.empty()) return {};
    std::vector<DocId> res;
    std::set_difference(all_docs.begin(), all_docs.end(), uni.begin(), uni.end(), std::back_inserter(res));
    return res;
  }
  std::vector<uint32_t> ids;
  for (const auto& t : terms) {
    uint32_t id = 0;
    if (impl_->Lookup(t, &id)) ids.push_back(id);
  }
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  if (ids.empty()) return all_docs;
  // std::set_difference (index.cpp:480-484) accepts a sorted all_docs WITH repeats: of m copies of a doc the union
  // holds, m - 1 survive; of a doc it does not hold, all m. The device takes the distinct ids; repeats are put back here.
  const bool repeats = std::adjacent_find(all_docs.begin(), all_docs.end()) != all_docs.end();
  std::vector<DocId> distinct;
  if (repeats) std::unique_copy(all_docs.begin(), all_docs.end(), std::back_inserter(distinct));
  const std::vector<DocId>& in = repeats ? distinct : all_docs;
  uint32_t* out = nullptr;
  uint64_t n = 0;
  if (mgx_not(impl_->dev, in.data(), in.size(), ids.data(), static_cast<uint32_t>(ids.size()), &out, &n) != MGX_OK) {
    impl_->last_error = mgx_last_error();
    SetDeviceError(impl_->last_error);
    return {};
  }
  std::vector<DocId> kept = Take(out, n);
  if (!repeats) return kept;
  std::vector<DocId> res;
  size_t k = 0;
  for (size_t i = 0; i < all_docs.size();) {
    size_t j = i;
    while (j < all_docs.size() && all_docs[j] == all_docs[i]) ++j;
    while (k < kept.size() && kept[k] < all_docs[i]) ++k;
    const bool survives = k < kept.size() && kept[k] == all_docs[i];
    res.insert(res.end(), survives ? j - i : j - i - 1, all_docs[i]);
    i = j;
  }
  return res;
}

std::vector<DocId> Index::SearchByThreshold(const std::vector<std::string>& terms, size_t threshold) const {
  ClearDeviceError();
  if (terms.empty() || threshold == 0) return {};  // index.cpp:489-491
  std::vector<std::string> uniq = terms;
  DeduplicateSorted(uniq);                         // :496-497
  if (threshold > uniq.size()) return {};          // :499-501
  if (threshold == uniq.size()) return SearchAnd(uniq);  // :504-506
  Finalize();
  if (!impl_->dev) {
    SetDeviceError(impl_->last_error);
    return {};
  }
  if (IsMutable(*this)) return OverBoth(*this, [&](const Index& ix) { return ix.SearchByThreshold(terms, threshold); });
  std::vector<uint32_t> ids;
  for (const auto& t : uniq) {
    uint32_t id = 0;
    if (impl_->Lookup(t, &id)) ids.push_back(id);  // missing lists do not count (:512-518)
  }
  if (ids.size() < threshold) return {};           // :521-523
  uint32_t* out = nullptr;
  uint64_t n = 0;
  if (mgx_threshold(impl_->dev, ids.data(), static_cast<uint32_t>(ids.size()), static_cast<uint32_t>(threshold), &out,
                    &n) != MGX_OK) {
    impl_->last_error = mgx_last_error();
    SetDeviceError(impl_->last_error);
    return {};
  }
  return Take(out, n);
}

std::vector<DocId> Index::FilterByNgrams(const std::vector<DocId>& candidates,
                                         const std::vector<std::string>& terms) const {
  ClearDeviceError();
  if (candidates.empty()) return {};  // index.cpp:372-374
  if (terms.empty()) return candidates;  // :378-380
  Finalize();
  if (!impl_->dev) {
    SetDeviceError(impl_->last_error);
    return {};
  }
  if (IsMutable(*this)) {
    // every candidate belongs to one index (or to neither: not a live document); each index filters its own, and the
    // caller's order — repeats included — is put back from the two answers, which are subsequences of what they were given
    const std::string err = ApplyMutations();
    if (!err.empty()) {
      SetDeviceError(err);
      return {};
    }
    RawScope raw;
    SplitCandidates split(*this, candidates);
    std::vector<uint8_t> keep(candidates.size(), 0);
    const std::vector<DocId> km = split.main_docs.empty() ? std::vector<DocId>() : FilterByNgrams(split.main_docs, terms);
    if (!tl_device_error.empty()) return {};
    split.MarkKept(split.main_docs, split.main_pos, km, &keep);
    if (split.view.delta && !split.delta_docs.empty()) {
      const std::vector<DocId> kd = split.view.delta->FilterByNgrams(split.delta_docs, terms);
      if (!tl_device_error.empty()) return {};
      split.MarkKept(split.delta_docs, split.delta_pos, kd, &keep);
    }
    std::vector<DocId> out;
    for (size_t i = 0; i < candidates.size(); ++i)
      if (keep[i]) out.push_back(candidates[i]);
    return out;
  }
  std::vector<uint32_t> ids;
  for (const auto& t : terms) {
    uint32_t id = 0;
    if (!impl_->Lookup(t, &id)) return {};  // :381-385
    ids.push_back(id);
  }
  uint32_t* out = nullptr;
  uint64_t n = 0;
  if (mgx_retain(impl_->dev, candidates.data(), candidates.size(), ids.data(), static_cast<uint32_t>(ids.size()), &out,
                 &n) != MGX_OK) {
    impl_->last_error = mgx_last_error();
    SetDeviceError(impl_->last_error);
    return {};
  }
  return Take(out, n);
}

// ---- BM25Scorer ---------------------------------------------------------------------------------------------------

double BM25Scorer::ComputeIDF(uint64_t total_docs, uint64_t doc_freq) {  // bm25_scorer.cpp:14-25
  if (total_docs == 0) return 0.0;
  if (doc_freq > total_docs) doc_freq = total_docs;
  const auto n = static_cast<double>(total_docs);
  const auto df = static_cast<double>(doc_freq);
  return std::log((n - df + 0.5) / (df + 0.5) + 1.0);
}

Expected<std::vector<ScoredDoc>, Error> BM25Scorer::ScoreDocuments(const std::vector<DocId>& candidates,
                                                                  const std::vector<std::string>& search_terms,
                                                                  const std::vector<uint64_t>& term_doc_freqs,
                                                                  const Index& index, uint64_t total_docs,
                                                                  double avg_doc_length, const BM25Params& params) {
  if (search_terms.size() != term_doc_freqs.size()) {  // bm25_scorer.cpp:51-55
    return MakeUnexpected(MakeError(ErrorCode::kInvalidArgument,
                                    "BM25 search_terms and term_doc_freqs must have identical lengths"));
  }
  index.Finalize();
  Index::Impl* im = index.impl();
  if (!im->dev) return MakeUnexpected(MakeError(ErrorCode::kInternalError, im->last_error));
  if (IsMutable(index)) {
    // a document's score comes from its own text (tf, length) and the caller's statistics: each index scores its own
    // candidates; an id that is no live document scores 0.0, like a document the store does not hold (bm25_scorer.cpp:73-76)
    const std::string err = index.ApplyMutations();
    if (!err.empty()) return MakeUnexpected(MakeError(ErrorCode::kInternalError, err));
    RawScope raw;
    SplitCandidates split(index, candidates);
    std::vector<ScoredDoc> out;
    out.reserve(candidates.size());
    for (DocId c : candidates) out.push_back({c, 0.0});
    if (!split.main_docs.empty()) {
      auto r = ScoreDocuments(split.main_docs, search_terms, term_doc_freqs, index, total_docs, avg_doc_length, params);
      if (!r) return r;
      for (size_t i = 0; i < r->size(); ++i) out[split.main_pos[i]].score = (*r)[i].score;
    }
    if (split.view.delta && !split.delta_docs.empty()) {
      auto r = ScoreDocuments(split.delta_docs, search_terms, term_doc_freqs, *split.view.delta, total_docs, avg_doc_length, params);
      if (!r) return r;
      for (size_t i = 0; i < r->size(); ++i) out[split.delta_pos[i]].score = (*r)[i].score;
    }
    return out;
  }
  std::vector<uint32_t> ids;
  std::vector<double> idfs;
  bool all_single_gram = true;
  for (size_t i = 0; i < search_terms.size(); ++i) {
    uint32_t id = 0xFFFFFFFFu;  // a term no document contains: tf = 0 everywhere
    auto grams = GenerateQueryNgrams(search_terms[i], index.GetNgramSize(), im->query_kanji,
                                     index.GetCrossBoundaryNgrams());
    DeduplicateSorted(grams);
    if (grams.size() != 1 || grams[0] != search_terms[i]) all_single_gram = false;
    im->Lookup(search_terms[i], &id) || (id = 0xFFFFFFFFu);
    ids.push_back(id);
    idfs.push_back(ComputeIDF(total_docs, term_doc_freqs[i]));
  }
  std::vector<double> scores(candidates.size(), 0.0);
  int rc;
  if (all_single_gram) {  // every term is one n-gram: its tf column holds CountTermOccurrences already
    rc = mgx_score_documents(im->dev, candidates.data(), candidates.size(), ids.data(), idfs.data(),
                             static_cast<uint32_t>(ids.size()), avg_doc_length, params.k1, params.b, scores.data());
  } else {  // bm25_scorer.cpp:71-91 literally: every term counted in the text
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off(search_terms.size() + 1, 0);
    for (size_t i = 0; i < search_terms.size(); ++i) {
      bytes.insert(bytes.end(), search_terms[i].begin(), search_terms[i].end());
      off[i + 1] = static_cast<uint32_t>(bytes.size());
    }
    bytes.resize(bytes.size() + 16);
    rc = mgx_score_documents_text(im->dev, candidates.data(), candidates.size(), bytes.data(), off.data(),
                                  idfs.data(), static_cast<uint32_t>(search_terms.size()), avg_doc_length, params.k1,
                                  params.b, scores.data());
  }
  if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
  std::vector<ScoredDoc> out;
  out.reserve(candidates.size());
  for (size_t i = 0; i < candidates.size(); ++i) out.push_back({candidates[i], scores[i]});
  return out;
}

}  // namespace index

// =================================================================================================================
// query::ResultSorter
// =================================================================================================================

namespace query {

std::vector<DocId> ResultSorter::SortByScore(const index::Index& index, const std::vector<DocId>& results,
                                             const std::vector<double>& scores, SortOrder order, uint32_t limit,
                                             uint32_t offset) {
  index::ClearDeviceErrorForSorter();
  if (results.empty()) return {};  // result_sorter.cpp:663-665
  index.Finalize();
  index::Index::Impl* im = index.impl();
  if (!im->dev) {
    index::SetDeviceError(im->last_error);
    return {};
  }
  uint32_t* out = nullptr;
  uint64_t n = 0;
  if (mgx_sort_by_score(im->dev, results.data(), scores.data(), results.size(), order == SortOrder::DESC ? 1 : 0, limit,
                        offset, &out, &n) != MGX_OK) {
    im->last_error = mgx_last_error();
    index::SetDeviceError(im->last_error);
    return {};
  }
  return Take(out, n);
}

std::vector<DocId> ResultSorter::SortByScore(const std::vector<DocId>& results, const std::vector<double>& scores,
                                             SortOrder order, uint32_t limit, uint32_t offset) {
  index::ClearDeviceErrorForSorter();
  if (results.empty()) return {};
  const index::Index* idx = index::LastUsedIndex();
  if (idx == nullptr) {
    index::SetDeviceError("ResultSorter::SortByScore: no Index has been used in this process (the sort runs on its device)");
    return {};
  }
  return SortByScore(*idx, results, scores, order, limit, offset);
}

}  // namespace query

namespace storage {
size_t DocumentStore::Size() const { return static_cast<size_t>(index_->Bm25DocCount()); }
}  // namespace storage

// =================================================================================================================
// search_pipeline::ExecuteBatch / BatchExecutor
// =================================================================================================================

namespace search_pipeline {

namespace {
// The gram ids of one term: a query term is a handful of n-grams, so they live inside the object (no heap traffic on the
// planner's hot path — it plans a million terms a second); a long term spills to the heap.
class IdVec {
 public:
  IdVec() = default;
  IdVec(const IdVec& o) { assign(o.data(), o.size()); }
  IdVec(IdVec&& o) noexcept { steal(o); }
  IdVec& operator=(const IdVec& o) {
    if (this != &o) assign(o.data(), o.size());
    return *this;
  }
  IdVec& operator=(IdVec&& o) noexcept {
    if (this != &o) steal(o);
    return *this;
  }
  void push_back(uint32_t v) {
    if (!heap_.empty()) {
      heap_.push_back(v);
    } else if (n_ < kInline) {
      inl_[n_] = v;
    } else {
      heap_.assign(inl_, inl_ + n_);
      heap_.push_back(v);
    }
    ++n_;
  }
  void assign(const uint32_t* p, size_t n) {
    clear();
    for (size_t i = 0; i < n; ++i) push_back(p[i]);
  }
  void assign(size_t n, uint32_t v) {
    clear();
    for (size_t i = 0; i < n; ++i) push_back(v);
  }
  void clear() {
    heap_.clear();
    n_ = 0;
  }
  [[nodiscard]] const uint32_t* data() const { return heap_.empty() ? inl_ : heap_.data(); }
  [[nodiscard]] size_t size() const { return n_; }
  [[nodiscard]] bool empty() const { return n_ == 0; }
  [[nodiscard]] const uint32_t* begin() const { return data(); }
  [[nodiscard]] const uint32_t* end() const { return data() + n_; }

 private:
  static constexpr size_t kInline = 8;
  void steal(IdVec& o) {
    heap_ = std::move(o.heap_);
    std::copy(o.inl_, o.inl_ + kInline, inl_);
    n_ = o.n_;
    o.clear();
  }
  uint32_t inl_[kInline] = {0};
  size_t n_ = 0;
  std::vector<uint32_t> heap_;
};

struct TermInfo {  // search_pipeline.h:44-55
  IdVec gram_ids;
  size_t n_grams = 0;
  uint64_t estimated_size = 0;  // UINT64_MAX = no n-grams
  uint64_t df = 0;
  bool is_gram = false;  // the term is exactly one n-gram
  std::string normalized;
};

// One query planned on the host the way ExecuteFullPipeline's regular branch (search_pipeline.cpp:2002-2030) or its
// boolean branch (:1842-1913) plans it: everything the C ABI's mgx_query points at lives in this object.
struct PlannedQuery {
  mgx_query q{};
  bool on_device = false;          // false: resolved on the host (empty_term_detected) or failed
  bool empty_term_detected = false;
  ErrorCode error = ErrorCode::kSuccess;
  std::string error_message;
  std::vector<mgx_term> terms, not_terms;
  std::vector<IdVec> ids;       // (reserved before the first mgx_term points into it: elements must not move)
  std::vector<TermInfo> tis;    // scratch of PlanQuery, kept for its storage
  std::vector<mgx_filter> filters;
  std::vector<mgx_expr_token> expr;
  std::vector<uint32_t> score_list;  // mgx_query::score_terms of expression / FUZZY queries
  bool has_score_list = false;
  std::deque<std::string> texts;  // normalized text-level terms (addresses stay valid as the deque grows)
  // Back to the freshly constructed state WITHOUT giving memory back: an executor slot re-plans a thousand of these
  // per batch, and constructing one allocates (the deque's map and first node alone are two mallocs).
  void Reset() {
    q = mgx_query{};
    on_device = empty_term_detected = false;
    error = ErrorCode::kSuccess;
    error_message.clear();
    terms.clear();
    not_terms.clear();
    ids.clear();
    tis.clear();
    filters.clear();
    expr.clear();
    score_list.clear();
    has_score_list = false;
    texts.clear();
  }
};

TermInfo MakeInfo(const index::Index& index, const index::Index::Impl* im, const std::string& raw) {
  // GenerateTermInfos, search_pipeline.cpp:569-603
  TermInfo ti;
  ti.normalized = index.NormalizeText(raw);
  {
    // the commonest term of an n-gram index is exactly one n-gram of ASCII letters / digits: GenerateQueryNgrams would
    // return that one string (no whitespace to split at, no CJK run for the kanji size) — same answers without building
    // a vector of strings
    const int n = index.GetNgramSize();
    bool one_gram = n > 0 && ti.normalized.size() == static_cast<size_t>(n);
    for (size_t i = 0; one_gram && i < ti.normalized.size(); ++i) {
      const unsigned char c = static_cast<unsigned char>(ti.normalized[i]);
      one_gram = (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
    }
    if (one_gram) {
      uint32_t id = 0;
      uint64_t ps = 0;
      if (!im->Resolve(ti.normalized, &id, &ps)) ps = 0;
      ti.n_grams = 1;
      if (ps > 0) ti.gram_ids.push_back(id);
      ti.estimated_size = ps;
      ti.is_gram = true;
      ti.df = ps;
      return ti;
    }
  }
  auto grams = GenerateQueryNgrams(ti.normalized, index.GetNgramSize(), im->query_kanji, index.GetCrossBoundaryNgrams());
  DeduplicateSorted(grams);
  ti.n_grams = grams.size();
  uint64_t mn = UINT64_MAX;
  for (const auto& g : grams) {
    uint32_t id = 0;
    uint64_t ps = 0;
    if (!im->Resolve(g, &id, &ps)) ps = 0;
    if (ps > 0) {
      mn = std::min(mn, ps);
      ti.gram_ids.push_back(id);
    } else {
      mn = 0;
      break;
    }
  }
  ti.estimated_size = mn;
  // a term that IS its one n-gram: df = |posting list|, tf = the gram's tf column; any other term is counted in the
  // text on the device (PopulateTermDocumentFrequency / CountTermOccurrences)
  ti.is_gram = grams.size() == 1 && grams[0] == ti.normalized;
  ti.df = (ti.is_gram && mn != UINT64_MAX) ? mn : 0;
  return ti;
}

void Fail(PlannedQuery* p, ErrorCode code, const char* msg) {
  p->error = code;
  p->error_message = msg;
}

// ---- FILTER conditions -> device bitmaps -------------------------------------------------------------------------------
template <typename T>
bool ParseWhole(const std::string& s, T* out) {  // std::from_chars over the whole string (search_pipeline.cpp:962-991)
  const char* end = s.data() + s.size();
  T v{};
  auto [ptr, ec] = std::from_chars(s.data(), end, v);
  if (ec != std::errc() || ptr != end) return false;
  *out = v;
  return true;
}

struct ResolvedFilter {
  bool skip = false;     // the condition holds for every document
  uint32_t bitmap = 0;
  bool negate = false;
};

// One FilterCondition against the table's columns. bitmap_mode: every condition of the query is EQ / NE (the FilterIndex
// path, ApplyFiltersWithBitmap search_pipeline.cpp:1196-1237); otherwise the per-document semantics of ApplyFilters
// (:1098-1194) for every condition of the query. Returns an error message or "".
std::string ResolveCondition(const index::Index& index, const query::FilterCondition& cond, bool bitmap_mode,
                             ResolvedFilter* out) {
  index.FlushPendingFilterColumns();
  index::Index::Impl* im = index.impl();
  std::lock_guard<std::mutex> lock(im->filter_mu);
  const index::Index::Impl::FilterColumn* col = im->ResolveColumn(cond.column);
  std::string key;
  key.push_back(bitmap_mode ? 'b' : 'f');
  key.push_back(static_cast<char>('0' + static_cast<int>(cond.op)));
  key += col ? col->name : std::string("\x01") + cond.column;
  key.push_back('\0');
  key += cond.value;
  const auto hit = im->condition_cache.find(key);
  if (hit != im->condition_cache.end()) {
    out->bitmap = hit->second.first;
    out->negate = hit->second.second;
    out->skip = hit->second.first == 0xFFFFFFFFu;
    return "";
  }
  auto remember = [&](uint32_t bitmap, bool negate, bool skip) {
    out->bitmap = bitmap;
    out->negate = negate;
    out->skip = skip;
    im->condition_cache.emplace(key, std::make_pair(skip ? 0xFFFFFFFFu : bitmap, negate));
  };
  auto empty_bitmap = [&](uint32_t* id) -> std::string {
    if (!im->have_empty_bitmap) {
      if (mgx_index_add_filter_bitmap(im->dev, nullptr, 0, &im->empty_bitmap) != MGX_OK) return mgx_last_error();
      im->have_empty_bitmap = true;
    }
    *id = im->empty_bitmap;
    return "";
  };
  const bool is_ne = cond.op == query::FilterOp::NE;
  // a column no document has: every stored value is NULL — nothing equals or orders against it, != holds everywhere
  // (bitmap path: OrEqBitmapInto finds no column, :107-109; per-document path: NULL passes != only, :1151-1157)
  if (col == nullptr || col->type == 0) {
    if (is_ne) {
      remember(0, false, true);
      return "";
    }
    uint32_t id = 0;
    const std::string err = empty_bitmap(&id);
    if (!err.empty()) return err;
    remember(id, false, false);
    return "";
  }
  const size_t type = col->type;
  const std::string& v = cond.value;
  uint64_t lit = 0;
  bool valid = false;
  uint32_t op = 0;
  double eps = 0.0;
  if (type == 11) {  // string column: ranks in the column's bytewise-sorted dictionary
    const storage::FilterValue needle{v};
    const auto lb = std::lower_bound(col->dict.begin(), col->dict.end(), needle);
    const auto ub = std::upper_bound(col->dict.begin(), col->dict.end(), needle);
    const uint64_t lbi = static_cast<uint64_t>(lb - col->dict.begin()), ubi = static_cast<uint64_t>(ub - col->dict.begin());
    valid = true;
    switch (cond.op) {
      case query::FilterOp::EQ:
      case query::FilterOp::NE:
        op = MGX_CMP_EQ;  // (NE: the EQ set, negated below / by the null rule)
        lit = lb != ub ? lbi : ~0ull;  // a literal no document holds equals no rank
        break;
      case query::FilterOp::LT: op = MGX_CMP_LT; lit = lbi; break;   // stored <  v  <=> rank <  lower_bound
      case query::FilterOp::LTE: op = MGX_CMP_LT; lit = ubi; break;  // stored <= v  <=> rank <  upper_bound
      case query::FilterOp::GT: op = MGX_CMP_GE; lit = ubi; break;   // stored >  v  <=> rank >= upper_bound
      default: op = MGX_CMP_GE; lit = lbi; break;                     // stored >= v  <=> rank >= lower_bound
    }
  } else {
    switch (cond.op) {
      case query::FilterOp::EQ: case query::FilterOp::NE: op = MGX_CMP_EQ; break;
      case query::FilterOp::LT: op = MGX_CMP_LT; break;
      case query::FilterOp::LTE: op = MGX_CMP_LE; break;
      case query::FilterOp::GT: op = MGX_CMP_GT; break;
      default: op = MGX_CMP_GE; break;
    }
    if (type == 1) {  // bool
      if (bitmap_mode) {  // BuildTypeUnionBitmap :1043-1048
        valid = v == "1" || v == "true" || v == "0" || v == "false";
        lit = (v == "1" || v == "true") ? 1 : 0;
      } else {  // ParseFilterValue :957: anything but "1" / "true" reads as false
        valid = true;
        lit = (v == "1" || v == "true") ? 1 : 0;
      }
    } else if (type == 12) {  // double
      double d = 0.0;
      valid = ParseWhole(v, &d);
      std::memcpy(&lit, &d, 8);
      if (!bitmap_mode) eps = 1e-9;  // mygram::constants::kFilterValueEpsilon (src/utils/constants.h:104)
    } else if (type == 3 || type == 5 || type == 7 || type == 9) {  // unsigned
      uint64_t u = 0;
      valid = ParseWhole(v, &u);
      lit = u;
      if (bitmap_mode && valid) {  // the literal must be representable in the column's type (:1078-1089)
        const uint64_t mx = type == 3 ? 0xFFull : type == 5 ? 0xFFFFull : type == 7 ? 0xFFFFFFFFull : ~0ull;
        valid = u <= mx;
      }
    } else {  // signed integers, TimeValue
      int64_t i = 0;
      valid = ParseWhole(v, &i);
      lit = static_cast<uint64_t>(i);
      if (bitmap_mode && valid) {  // :1057-1067
        const int64_t lo = type == 2 ? INT8_MIN : type == 4 ? INT16_MIN : type == 6 ? INT32_MIN : INT64_MIN;
        const int64_t hi = type == 2 ? INT8_MAX : type == 4 ? INT16_MAX : type == 6 ? INT32_MAX : INT64_MAX;
        valid = i >= lo && i <= hi;
      }
    }
  }
  uint32_t id = 0;
  if (bitmap_mode) {
    // EQ: AND the union of the literal's interpretations; NE: ANDNOT it (NULL documents are in no value bitmap: they
    // pass NE, fail EQ). No interpretation in this column's type: the union is empty.
    if (!valid) {
      if (is_ne) {
        remember(0, false, true);
        return "";
      }
      const std::string err = empty_bitmap(&id);
      if (!err.empty()) return err;
      remember(id, false, false);
      return "";
    }
    if (mgx_index_filter_compare(im->dev, col->device_id, MGX_CMP_EQ, lit, 0.0, 0, 0, &id) != MGX_OK) return mgx_last_error();
    remember(id, is_ne, false);
    return "";
  }
  // per-document semantics: the set of documents that PASS the condition, ANDed
  const uint32_t dev_op = is_ne ? static_cast<uint32_t>(MGX_CMP_NE) : op;
  if (mgx_index_filter_compare(im->dev, col->device_id, dev_op, lit, eps, is_ne ? 1 : 0, valid ? 0 : 1, &id) != MGX_OK)
    return mgx_last_error();
  remember(id, false, false);
  return "";
}

void PlanQuery(const index::Index& index, const BatchQuery& q, uint64_t total_docs, double avgdl, PlannedQuery* p) {
  const index::Index::Impl* im = index.impl();
  // FILTER clauses: caller-resolved bitmaps, then the parsed conditions (resolved here, cached by the Index)
  auto add_filters = [&]() -> bool {
    for (const auto& f : q.filters) p->filters.push_back(mgx_filter{f.first, f.second ? 1u : 0u});
    bool bitmap_mode = true;  // AllFiltersHaveBitmapSupport, search_pipeline.cpp:996-1003
    for (const auto& c : q.filter_conditions)
      bitmap_mode = bitmap_mode && (c.op == query::FilterOp::EQ || c.op == query::FilterOp::NE);
    for (const auto& c : q.filter_conditions) {
      ResolvedFilter rf;
      const std::string err = ResolveCondition(index, c, bitmap_mode, &rf);
      if (!err.empty()) {
        p->error = ErrorCode::kInternalError;
        p->error_message = err;
        return false;
      }
      if (!rf.skip) p->filters.push_back(mgx_filter{rf.bitmap, rf.negate ? 1u : 0u});
    }
    if (p->filters.size() > MGX_MAX_TERMS) {
      p->error = ErrorCode::kInvalidArgument;
      p->error_message = "more than 64 filters in one query";
      return false;
    }
    return true;
  };
  auto finish = [&](uint32_t sort) {
    mgx_query& m = p->q;
    m.terms = p->terms.data();
    m.n_terms = static_cast<uint32_t>(p->terms.size());
    m.not_terms = p->not_terms.data();
    m.n_not_terms = static_cast<uint32_t>(p->not_terms.size());
    m.filters = p->filters.data();
    m.n_filters = static_cast<uint32_t>(p->filters.size());
    m.sort = sort;
    m.limit = q.limit;
    m.reverse = q.order == query::SortOrder::DESC ? 1 : 0;
    m.k1 = q.bm25.k1;
    m.b = q.bm25.b;
    m.total_docs = total_docs;
    m.avg_doc_length = avgdl;
    if (p->has_score_list) {  // (non-NULL even when empty: "no term is scored" is not "every term is")
      static const uint32_t kNone = 0;
      m.score_terms = p->score_list.empty() ? &kNone : p->score_list.data();
      m.n_score_terms = static_cast<uint32_t>(p->score_list.size());
    }
    p->on_device = true;
  };
  auto add_not_terms = [&]() -> bool {  // ApplyNotFilter :871-932
    for (const auto& t : q.not_terms) {
      TermInfo ti = MakeInfo(index, im, t);
      if (ti.n_grams == 0) {
        // shorter than one n-gram: the docs whose text contains it (SearchTermDocuments :438-446); "" matches nothing
        if (ti.normalized.empty()) continue;
        p->texts.push_back(ti.normalized);
        p->not_terms.push_back(mgx_term{nullptr, 0, 0, 0.0, reinterpret_cast<const uint8_t*>(p->texts.back().data()),
                                        static_cast<uint32_t>(p->texts.back().size())});
        continue;
      }
      if (ti.estimated_size == 0) continue;  // an unknown gram: the NOT term matches nothing
      p->ids.push_back(ti.gram_ids);
      p->not_terms.push_back(mgx_term{p->ids.back().data(), static_cast<uint32_t>(p->ids.back().size()), 0, 0.0, nullptr, 0});
    }
    return true;
  };
  if (q.ast) {
    // ---- boolean expression: distinct TERM leaves in first-use order, tree in postfix -----------------------------
    if (!q.terms.empty())
      return Fail(p, ErrorCode::kNotImplemented, "an expression query takes its terms from the tree");
    std::vector<std::string> leaves;
    std::vector<TermInfo> tis;
    bool bad = false;
    // SORT _score: the TERM leaves that are not under a NOT, in tree order, repeats kept (CollectAstScoringTerms,
    // search_pipeline.cpp:232-254; search_handler.cpp:428-456 scores exactly those, whatever branch matched)
    p->has_score_list = q.sort_by_score;
    std::function<void(const query::QueryNode&, bool)> walk = [&](const query::QueryNode& nd, bool under_not) {
      if (nd.type == query::NodeType::TERM) {
        size_t k = 0;
        while (k < leaves.size() && leaves[k] != nd.term) ++k;
        if (k == leaves.size()) {
          leaves.push_back(nd.term);
          tis.push_back(MakeInfo(index, im, nd.term));
        }
        // a term with an unknown gram is an empty doc set inside the tree; a term shorter than one n-gram is the documents
        // whose text contains it (regular_term_search -> SearchTermDocuments, search_pipeline.cpp:1446-1455, :438-446)
        const bool substring = tis[k].n_grams == 0 && !tis[k].normalized.empty();
        const bool empty = !substring && (tis[k].estimated_size == 0 || tis[k].estimated_size == UINT64_MAX);
        p->expr.push_back(empty ? mgx_expr_token{MGX_EXPR_EMPTY, 0} : mgx_expr_token{MGX_EXPR_TERM, static_cast<uint32_t>(k)});
        // (a leaf with an unknown gram occurs in no text: tf = df = 0, it adds nothing to any score)
        if (q.sort_by_score && !under_not && !empty) p->score_list.push_back(static_cast<uint32_t>(k));
        return;
      }
      if (nd.children.empty() || (nd.type == query::NodeType::NOT && nd.children.size() != 1)) {
        bad = true;
        return;
      }
      for (const auto& c : nd.children) walk(*c, under_not || nd.type == query::NodeType::NOT);
      const uint32_t op = nd.type == query::NodeType::AND ? MGX_EXPR_AND
                          : nd.type == query::NodeType::OR ? MGX_EXPR_OR
                                                           : MGX_EXPR_NOT;
      p->expr.push_back(mgx_expr_token{op, static_cast<uint32_t>(nd.children.size())});
    };
    walk(*q.ast, false);
    if (bad || leaves.empty() || leaves.size() > MGX_MAX_TERMS)
      return Fail(p, ErrorCode::kInvalidArgument, "malformed expression tree");
    p->ids.reserve(leaves.size() + q.not_terms.size());
    for (auto& ti : tis) {
      if (ti.n_grams == 0 && !ti.normalized.empty()) {  // substring leaf: a text scan on the device
        p->ids.emplace_back();
        p->texts.push_back(ti.normalized);
        p->terms.push_back(mgx_term{nullptr, 0, 0, 0.0, reinterpret_cast<const uint8_t*>(p->texts.back().data()),
                                    static_cast<uint32_t>(p->texts.back().size())});
        continue;
      }
      const bool empty_leaf = ti.gram_ids.empty() || ti.estimated_size == 0 || ti.estimated_size == UINT64_MAX;
      if (empty_leaf) ti.gram_ids.assign(1, 0u);  // placeholder of an EMPTY leaf
      p->ids.push_back(ti.gram_ids);
      mgx_term mt{p->ids.back().data(), static_cast<uint32_t>(p->ids.back().size()), 0, 0.0, nullptr, 0};
      if (q.sort_by_score && !empty_leaf) {
        if (ti.is_gram) {
          mt.idf = index::BM25Scorer::ComputeIDF(total_docs, ti.df);
        } else {  // tf / df from the text on the device
          p->texts.push_back(ti.normalized);
          mt.text = reinterpret_cast<const uint8_t*>(p->texts.back().data());
          mt.text_len = static_cast<uint32_t>(p->texts.back().size());
        }
      }
      p->terms.push_back(mt);
    }
    if (!add_not_terms()) return;
    if (!add_filters()) return;
    if (im->has_gaps) p->filters.push_back(mgx_filter{im->exists_bitmap, 0u});  // NOT universe = added ids
    finish(q.sort_by_score ? MGX_SORT_SCORE : MGX_SORT_DOCID);
    p->q.offset = q.sort_by_score ? q.offset : 0;
    p->q.expr = p->expr.data();
    p->q.n_expr = static_cast<uint32_t>(p->expr.size());
    return;
  }
  if (q.terms.empty() || q.terms.size() > MGX_MAX_TERMS || q.not_terms.size() > MGX_MAX_TERMS)
    return Fail(p, ErrorCode::kInvalidArgument, "query needs 1..64 terms");
  if (q.fuzzy_max_distance > 0) {
    // ---- ExecuteWithFuzzy (search_pipeline.cpp:1659-1744): no size sort; theta per term; unknown grams are skipped by
    // Index::SearchByThreshold (index.cpp:512-523) unless theta asks for every gram (then it is SearchAnd) ------------
    if (q.verify_text)
      return Fail(p, ErrorCode::kNotImplemented, "FUZZY with verify_text is a host path (edit distance over texts)");
    p->ids.reserve(q.terms.size() + q.not_terms.size());
    p->has_score_list = q.sort_by_score;
    for (const auto& raw : q.terms) {
      const std::string normalized = index.NormalizeText(raw);
      auto grams = GenerateQueryNgrams(normalized, index.GetNgramSize(), im->query_kanji, index.GetCrossBoundaryNgrams());
      DeduplicateSorted(grams);
      if (grams.empty()) {  // :1676-1682
        p->empty_term_detected = true;
        return;
      }
      size_t n_eff = index.GetNgramSize() > 0 ? static_cast<size_t>(index.GetNgramSize()) : 2;  // :1685-1698
      if (im->query_kanji > 0) {
        size_t short_count = 0;
        for (const auto& g : grams) short_count += g.size() <= 3 ? 1 : 0;
        if (short_count > grams.size() / 2) n_eff = static_cast<size_t>(im->query_kanji);
      }
      const size_t drop = static_cast<size_t>(q.fuzzy_max_distance) * n_eff;  // :1700-1703
      const size_t theta = grams.size() > drop ? grams.size() - drop : 1;
      IdVec known;
      for (const auto& g : grams) {
        uint32_t id = 0;
        uint64_t ps = 0;
        if (im->Resolve(g, &id, &ps) && ps > 0) known.push_back(id);
      }
      const bool empty = theta == grams.size() ? known.size() != grams.size() : known.size() < theta;
      if (empty || known.empty()) known.assign(1, MGX_GRAM_ABSENT);  // the empty doc set, as an operand
      p->ids.push_back(std::move(known));
      const auto& ids = p->ids.back();
      const uint32_t thr = (empty || theta >= ids.size()) ? 0u : static_cast<uint32_t>(theta);
      mgx_term mt{ids.data(), static_cast<uint32_t>(ids.size()), thr, 0.0, nullptr, 0};
      // SORT _score: the EXACT term is scored (GenerateTermInfos' infos, search_pipeline.cpp:1895-1899 ->
      // search_handler.cpp:428-456). A term with an unknown n-gram occurs in no text (tf = df = 0): left out.
      if (q.sort_by_score && !empty && ids.size() == grams.size()) {
        p->score_list.push_back(static_cast<uint32_t>(p->terms.size()));
        if (grams.size() == 1 && grams[0] == normalized) {
          uint32_t gid = 0;
          uint64_t ps = 0;
          im->Resolve(grams[0], &gid, &ps);
          mt.idf = index::BM25Scorer::ComputeIDF(total_docs, ps);
        } else {
          p->texts.push_back(normalized);
          mt.text = reinterpret_cast<const uint8_t*>(p->texts.back().data());
          mt.text_len = static_cast<uint32_t>(p->texts.back().size());
        }
      }
      p->terms.push_back(mt);
    }
    if (!add_not_terms()) return;
    if (!add_filters()) return;
    finish(q.sort_by_score ? MGX_SORT_SCORE : MGX_SORT_DOCID);
    p->q.offset = q.sort_by_score ? q.offset : 0;
    return;
  }
  std::vector<TermInfo>& tis = p->tis;
  tis.reserve(q.terms.size());
  for (const auto& t : q.terms) tis.push_back(MakeInfo(index, im, t));
  // search_pipeline.cpp:2012-2014 (std::sort on <=16 elements is an insertion sort: equal keys keep their order)
  // (a stable insertion sort by hand: std::stable_sort asks the heap for a merge buffer on every call, and this runs a
  // million times a second)
  for (size_t i = 1; i < tis.size(); ++i)
    for (size_t j = i; j > 0 && tis[j].estimated_size < tis[j - 1].estimated_size; --j) std::swap(tis[j], tis[j - 1]);
  for (const auto& ti : tis)  // Execute :804-810
    if ((ti.estimated_size == 0 || ti.estimated_size == UINT64_MAX) && (ti.n_grams != 0 || ti.normalized.empty())) {
      p->empty_term_detected = true;
      return;
    }
  p->ids.reserve(q.terms.size() + q.not_terms.size());
  // exact-text post-filter: the caller's verify_text decision, or a mixed-script term with an uncovered fragment
  // (search_pipeline.cpp:856-866)
  bool exact = q.verify_text;
  for (const auto& ti : tis)
    exact = exact || HasUncoveredHybridFragment(ti.normalized, index.GetNgramSize(), im->query_kanji,
                                                index.GetCrossBoundaryNgrams());
  for (auto& ti : tis) {
    p->ids.push_back(std::move(ti.gram_ids));  // (the plan owns the ids from here on: no copy)
    mgx_term mt{p->ids.back().data(), static_cast<uint32_t>(p->ids.back().size()), 0, 0.0, nullptr, 0};
    if (q.sort_by_score && ti.is_gram) mt.idf = index::BM25Scorer::ComputeIDF(total_docs, ti.df);
    // (a term shorter than one n-gram has no grams at all: the device scans the texts for it, SearchNormalizedSubstring)
    if (exact || ti.n_grams == 0 || (q.sort_by_score && !ti.is_gram)) {
      p->texts.push_back(ti.normalized);
      mt.text = reinterpret_cast<const uint8_t*>(p->texts.back().data());
      mt.text_len = static_cast<uint32_t>(p->texts.back().size());
    }
    p->terms.push_back(mt);
  }
  if (!add_not_terms()) return;
  if (!add_filters()) return;
  finish(q.sort_by_score ? MGX_SORT_SCORE : MGX_SORT_DOCID);
  p->q.offset = q.offset;
  p->q.exact_text = exact ? 1u : 0u;
}

// results of one fetched batch -> BatchResult objects, in the order the queries were given
void Collect(const std::vector<PlannedQuery>& plans, const mgx_result_view& v, std::vector<BatchResult>* out,
             const mgx_result_view* delta_view = nullptr) {
  out->resize(plans.size());  // (elements a caller hands back keep their vectors' storage: WaitInto)
  size_t k = 0;
  for (size_t qi = 0; qi < plans.size(); ++qi) {
    BatchResult& o = (*out)[qi];
    o.results.clear();
    o.scores.clear();
    o.total = 0;
    o.total_candidates = o.after_intersection = o.after_not = o.after_filters = 0;
    o.empty_term_detected = false;
    if (!plans[qi].on_device) {
      o.empty_term_detected = plans[qi].empty_term_detected;
      continue;
    }
    const mgx_query_result& r = v.queries[k];
    o.total = r.total;
    o.total_candidates = r.total_candidates;
    o.after_intersection = r.after_intersection;
    o.after_not = r.after_not;
    o.after_filters = r.after_filters;
    if (delta_view != nullptr && delta_view->queries != nullptr) {
      // a mutable table: page and total were merged on the device; the funnel counters are each index's own
      const mgx_query_result& d = delta_view->queries[k];
      o.total_candidates += d.total_candidates;
      o.after_intersection += d.after_intersection;
      o.after_not += d.after_not;
      o.after_filters += d.after_filters;
    }
    ++k;
    o.results.assign(v.docs + r.docs_begin, v.docs + r.docs_begin + r.n_docs);
    o.scores.assign(v.scores + r.docs_begin, v.scores + r.docs_begin + r.n_docs);
  }
}
}  // namespace

namespace {
// A mutable table's delta index as the device holds it now (nullptr: none). ApplyMutations has run.
std::shared_ptr<index::Index> DeltaOf(const index::Index& index) {
  index::Index::Impl* im = index.impl();
  std::lock_guard<std::mutex> lock(im->mu);
  return im->mut.delta;
}

// The same query planned against the delta index: must run where the main plan runs (both see table-wide sizes).
bool PlanForDelta(const index::Index& delta, const BatchQuery& q, uint64_t total_docs, double avgdl,
                  const PlannedQuery& main_plan, PlannedQuery* p) {
  if (!q.filters.empty()) {
    Fail(p, ErrorCode::kNotImplemented,
         "BatchQuery::filters (raw bitmap ids) belong to one device index: a table with a delta takes filter_conditions");
    return false;
  }
  PlanQuery(delta, q, total_docs, avgdl, p);
  if (p->error != ErrorCode::kSuccess) return false;
  if (p->on_device != main_plan.on_device) {
    Fail(p, ErrorCode::kInternalError, "the main and the delta index plan a query differently");
    return false;
  }
  return true;
}

// df pass + execute of both batches and the merge, all on the main batch's stream
int RunWithDelta(mgx_batch* main_batch, mgx_batch* delta_batch, void* stream) {
  mgx_batch* others[1] = {delta_batch};
  int rc = mgx_batch_df_merge_local(main_batch, others, 1, stream);
  if (rc == MGX_OK) rc = mgx_batch_execute(main_batch, stream);
  if (rc == MGX_OK) rc = mgx_batch_execute(delta_batch, stream);
  if (rc == MGX_OK) rc = mgx_batch_merge_local(main_batch, others, 1, stream);
  return rc;
}
}  // namespace

Expected<std::vector<BatchResult>, Error> ExecuteBatch(const index::Index& index,
                                                       const std::vector<BatchQuery>& queries) {
  index.Finalize();
  index::Index::Impl* im = index.impl();
  if (!im->dev) return MakeUnexpected(MakeError(ErrorCode::kInternalError, im->last_error));
  {
    const std::string merr = index.ApplyMutations();
    if (!merr.empty()) return MakeUnexpected(MakeError(ErrorCode::kInternalError, merr));
  }
  const std::shared_ptr<index::Index> delta = DeltaOf(index);
  const uint64_t total_docs = index.Bm25DocCount();
  const double avgdl = index.Bm25AvgDocLength();
  std::vector<PlannedQuery> plans(queries.size()), dplans(delta ? queries.size() : 0);
  std::vector<mgx_query> mq, dq;
  for (size_t qi = 0; qi < queries.size(); ++qi) {
    PlanQuery(index, queries[qi], total_docs, avgdl, &plans[qi]);
    if (plans[qi].error != ErrorCode::kSuccess)
      return MakeUnexpected(MakeError(plans[qi].error, plans[qi].error_message));
    if (plans[qi].on_device) mq.push_back(plans[qi].q);
    if (delta) {
      if (!PlanForDelta(*delta, queries[qi], total_docs, avgdl, plans[qi], &dplans[qi]))
        return MakeUnexpected(MakeError(dplans[qi].error, dplans[qi].error_message));
      if (dplans[qi].on_device) dq.push_back(dplans[qi].q);
    }
  }
  std::vector<BatchResult> out;
  mgx_result_view v{};
  if (mq.empty()) {
    Collect(plans, v, &out);
    return out;
  }
  mgx_batch* batch = nullptr;
  int rc = mgx_batch_prepare(im->dev, mq.data(), static_cast<uint32_t>(mq.size()), &batch);
  if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
  std::unique_ptr<mgx_batch, void (*)(mgx_batch*)> guard(batch, mgx_batch_destroy);
  if (delta) {
    mgx_batch* dbatch = nullptr;
    rc = mgx_batch_prepare(delta->impl()->dev, dq.data(), static_cast<uint32_t>(dq.size()), &dbatch);
    if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
    std::unique_ptr<mgx_batch, void (*)(mgx_batch*)> dguard(dbatch, mgx_batch_destroy);
    void* stream = nullptr;
    rc = mgx_batch_stream(batch, &stream);
    if (rc == MGX_OK) rc = RunWithDelta(batch, dbatch, stream);
    if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
    mgx_result_view dv{};
    rc = mgx_batch_fetch(batch, &v);
    if (rc == MGX_OK) rc = mgx_batch_fetch(dbatch, &dv);
    if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
    Collect(plans, v, &out, &dv);
    return out;
  }
  rc = mgx_batch_execute(batch, nullptr);
  if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
  rc = mgx_batch_fetch(batch, &v);
  if (rc != MGX_OK) return MakeUnexpected(MakeError(static_cast<ErrorCode>(rc), mgx_last_error()));
  Collect(plans, v, &out);
  return out;
}

namespace {
// FACET over one device index: the column's distinct values (ascending) and how many documents of the query's result set
// hold each. found = false: this index has no such column.
Error FacetOne(const index::Index& index, const BatchQuery& query, const std::string& column, bool* found,
               std::vector<storage::FilterValue>* dict, std::vector<uint64_t>* counts, uint64_t* matched) {
  index.FlushPendingFilterColumns();
  index::Index::Impl* im = index.impl();
  uint32_t device_column = 0;
  *found = false;
  *matched = 0;
  {
    std::lock_guard<std::mutex> lock(im->filter_mu);
    const index::Index::Impl::FilterColumn* col = im->ResolveColumn(column);
    if (col != nullptr) {
      *found = true;
      device_column = col->device_id;
      *dict = col->dict;
    }
  }
  // the search part: the query's own, or — FACET without search terms — every document (GetAllDocIds, :2118) narrowed by
  // its NOT terms and filters: the expression NOT(<empty>) is the universe
  BatchQuery q = query;
  q.sort_by_score = false;
  q.limit = 0;
  q.offset = 0;
  if (q.terms.empty() && !q.ast) {
    auto all = std::make_shared<query::QueryNode>(query::NodeType::NOT);
    all->children.push_back(std::make_unique<query::QueryNode>(std::string()));
    q.ast = std::move(all);
  }
  PlannedQuery plan;
  PlanQuery(index, q, index.Bm25DocCount(), index.Bm25AvgDocLength(), &plan);
  if (plan.error != ErrorCode::kSuccess) return MakeError(plan.error, plan.error_message);
  counts->assign(std::max<size_t>(dict->size(), 1), 0);
  if (plan.on_device && *found) {
    if (mgx_facet_counts(im->dev, &plan.q, device_column, counts->data(), matched) != MGX_OK)
      return MakeError(ErrorCode::kInternalError, mgx_last_error());
  }
  return Error{ErrorCode::kSuccess, ""};
}
}  // namespace

Expected<FacetOutput, Error> ExecuteFacet(const index::Index& index, const BatchQuery& query, const std::string& column) {
  index.Finalize();
  index::Index::Impl* im = index.impl();
  if (!im->dev) return MakeUnexpected(MakeError(ErrorCode::kInternalError, im->last_error));
  {
    const std::string merr = index.ApplyMutations();
    if (!merr.empty()) return MakeUnexpected(MakeError(ErrorCode::kInternalError, merr));
  }
  const std::shared_ptr<index::Index> delta = DeltaOf(index);
  if (delta && !query.filters.empty())
    return MakeUnexpected(MakeError(ErrorCode::kNotImplemented,
                                    "BatchQuery::filters (raw bitmap ids) belong to one device index: a table with a delta takes filter_conditions"));
  bool found = false, dfound = false;
  std::vector<storage::FilterValue> dict, ddict;
  std::vector<uint64_t> counts, dcounts;
  FacetOutput out;
  Error e = FacetOne(index, query, column, &found, &dict, &counts, &out.matched_documents);
  if (e.code() != ErrorCode::kSuccess) return MakeUnexpected(e);
  if (delta) {  // a mutable table: the delta's documents count too; values meet by value, not by per-index id
    uint64_t dmatched = 0;
    e = FacetOne(*delta, query, column, &dfound, &ddict, &dcounts, &dmatched);
    if (e.code() != ErrorCode::kSuccess) return MakeUnexpected(e);
    out.matched_documents += dmatched;
    if (dfound) {
      std::vector<storage::FilterValue> merged;
      std::vector<uint64_t> mcounts;
      size_t i = 0, j = 0;
      const size_t ni = found ? dict.size() : 0;
      while (i < ni || j < ddict.size()) {
        if (j == ddict.size() || (i < ni && dict[i] < ddict[j])) {
          merged.push_back(dict[i]);
          mcounts.push_back(counts[i]);
          ++i;
        } else if (i == ni || ddict[j] < dict[i]) {
          merged.push_back(ddict[j]);
          mcounts.push_back(dcounts[j]);
          ++j;
        } else {
          merged.push_back(dict[i]);
          mcounts.push_back(counts[i] + dcounts[j]);
          ++i;
          ++j;
        }
      }
      dict.swap(merged);
      counts.swap(mcounts);
      found = true;
    }
  }
  if (!found)  // search_pipeline.cpp:2089-2092
    return MakeUnexpected(MakeError(ErrorCode::kIndexNotFound, "Facet column \"" + column + "\" not found"));
  std::vector<size_t> order;
  for (size_t v = 0; v < dict.size(); ++v)
    if (counts[v] > 0) order.push_back(v);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return counts[a] > counts[b]; });  // :307
  out.total_values = order.size();
  const size_t first = std::min<size_t>(query.offset, order.size());                         // :2141-2148
  const size_t last = std::min<size_t>(order.size(), first + static_cast<size_t>(query.limit));
  for (size_t k = first; k < last; ++k) {
    out.value_counts.emplace_back(storage::SerializeFilterValue(dict[order[k]]), counts[order[k]]);
    out.display.push_back(storage::FilterValueToDisplayString(dict[order[k]]));
  }
  return out;
}

// ---- BatchExecutor ------------------------------------------------------------------------------------------------
//
// Submit only queues: a pool of workers plans the batch in chunks of queries (several batches at once when several are
// queued); dispatcher threads (three) compile planned batches (mgx_batch_reset: host work, one object per slot) and put
// them on the device strictly in ticket order — on a sharded table each execute is followed by a collective that every
// rank must issue in the same order. Wait blocks until its ticket is on the device, then fetches.

struct BatchExecutor::Impl {
  using clock = std::chrono::steady_clock;
  const index::Index& index;
  Options opt;
  uint64_t total_docs = 0;
  double avgdl = 0.0;
  enum State { kFree, kFilling, kPlanning, kPlanned, kCompiling, kCompiled, kEnqueued, kFailed };
  struct Slot {
    mgx_batch* batch = nullptr;
    std::vector<BatchQuery> queries;
    std::vector<PlannedQuery> plans;
    std::vector<mgx_query> mq;
    // a mutable table: the delta index this batch was submitted against, the same queries planned for it, and the batch
    // object compiled on it (bound to dbatch_index: a rebuilt delta gets a new object)
    std::shared_ptr<index::Index::Impl::Handles> handles;  // the generation of the main index `batch` was compiled on
    std::shared_ptr<index::Index> delta, dbatch_index;
    std::vector<PlannedQuery> dplans;
    std::vector<mgx_query> dq;
    mgx_batch* dbatch = nullptr;
    uint64_t total_docs = 0;  // BM25Stats the batch is planned with (a mutable table's change between batches)
    double avgdl = 0.0;
    uint64_t ticket = 0;
    State state = kFree;
    size_t chunks_left = 0;  // under mu
    Error error{ErrorCode::kSuccess, ""};
    Timing timing;
    clock::time_point t_submit, t_planned, t_compiled;
  };
  struct Chunk {
    Slot* slot;
    size_t begin, end;
  };
  static constexpr size_t kChunk = 64;
  std::vector<Slot> slots;
  uint64_t next_ticket = 1, next_compile = 1, next_enqueue = 1;
  std::vector<std::thread> workers;
  std::vector<std::thread> dispatchers;
  std::mutex mu;  // slots' state, the chunk queue, the enqueue turn
  std::condition_variable cv_work, cv_state;
  std::deque<Chunk> chunks;
  bool stop = false;
  bool enqueuing = false;  // a dispatcher is inside EnqueueReady's loop (guarded by mu)
  // Sharded tables: every rank issues the same collectives in the same order. A rank that cannot (a ticket that failed to
  // plan or compile here, an exchange call that returned an error) has left that lock-step: its peers have enqueued this
  // ticket's all-gather and would pair it with the NEXT ticket's. The communicator is then poisoned: aborted
  // (mgx_comm_abort: this rank's enqueued collectives are cancelled, the peers' fail or time out instead of pairing
  // wrongly), and every later ticket fails at once with the first error — the process is expected to exit non-zero.
  bool comm_poisoned = false;  // guarded by mu
  std::string poison_message;

  Impl(const index::Index& ix, Options o) : index(ix), opt(o) {}

  Slot* ByTicket(uint64_t ticket) {
    for (auto& s : slots)
      if (s.state != kFree && s.state != kFilling && s.ticket == ticket) return &s;
    return nullptr;
  }

  // host side of one batch after planning: the device queries, compiled into the slot's batch object
  void Compile(Slot* slot) {
    index::Index::Impl* im = index.impl();
    static const bool kTrace = std::getenv("MGX_TRACE_HOST") != nullptr;
    const auto t_in = clock::now();
    slot->mq.clear();
    for (const auto& p : slot->plans) {
      if (p.error != ErrorCode::kSuccess) {
        slot->error = MakeError(p.error, p.error_message);
        return;
      }
      if (p.on_device) slot->mq.push_back(p.q);
    }
    slot->timing.device_queries = static_cast<uint32_t>(slot->mq.size());
    slot->dq.clear();
    if (slot->delta) {
      for (const auto& p : slot->dplans) {
        if (p.error != ErrorCode::kSuccess) {
          slot->error = MakeError(p.error, p.error_message);
          return;
        }
        if (p.on_device) slot->dq.push_back(p.q);
      }
    }
    if (slot->mq.empty()) return;
    if (kTrace) {
      static std::atomic<uint64_t> n{0}, us{0};
      us += std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t_in).count();
      if (++n % 64 == 0) fprintf(stderr, "[shim] gather of the planned queries: %.3f ms per batch\n", us.exchange(0) / 64e3);
    }
    int rc;
    {
      std::shared_ptr<index::Index::Impl::Handles> cur;
      {
        std::lock_guard<std::mutex> il(im->mu);
        cur = im->handles;
      }
      if (slot->batch && slot->handles != cur) {  // the main index was compacted: the object belongs to the old generation
        mgx_batch_destroy(slot->batch);
        slot->batch = nullptr;
      }
      slot->handles = std::move(cur);
    }
    if (!slot->batch)
      rc = mgx_batch_prepare(im->dev, slot->mq.data(), static_cast<uint32_t>(slot->mq.size()), &slot->batch);
    else
      rc = mgx_batch_reset(slot->batch, slot->mq.data(), static_cast<uint32_t>(slot->mq.size()));
    if (rc != MGX_OK) {
      slot->error = MakeError(static_cast<ErrorCode>(rc), mgx_last_error());
      return;
    }
    if (slot->dbatch && slot->dbatch_index != slot->delta) {  // the delta index was rebuilt (or went away)
      mgx_batch_destroy(slot->dbatch);
      slot->dbatch = nullptr;
      slot->dbatch_index.reset();
    }
    if (!slot->delta) return;
    if (!slot->dbatch) {
      rc = mgx_batch_prepare(slot->delta->impl()->dev, slot->dq.data(), static_cast<uint32_t>(slot->dq.size()), &slot->dbatch);
      if (rc == MGX_OK) slot->dbatch_index = slot->delta;
    } else {
      rc = mgx_batch_reset(slot->dbatch, slot->dq.data(), static_cast<uint32_t>(slot->dq.size()));
    }
    if (rc != MGX_OK) slot->error = MakeError(static_cast<ErrorCode>(rc), mgx_last_error());
  }

  // device side, in ticket order. Called with mu held; the HIP calls themselves (they only enqueue, but a batch is a
  // dozen of them: ~0.1 ms) run with mu RELEASED — planners, Submit and Wait all take mu — and `enqueuing` keeps them in
  // one thread at a time: whoever finds it set leaves its batch to that thread, which looks again (under mu) after
  // every batch it has put on the device.
  void EnqueueReady(std::unique_lock<std::mutex>& lock) {
    if (enqueuing) return;
    enqueuing = true;
    for (;;) {
      Slot* slot = ByTicket(next_enqueue);
      if (!slot || (slot->state != kCompiled && slot->state != kFailed)) break;
      if (opt.comm && comm_poisoned && slot->state == kCompiled) {
        slot->error = MakeError(ErrorCode::kInternalError, "the table's communicator was aborted after an earlier failure: " + poison_message);
        slot->state = kFailed;
      } else if (opt.comm && slot->state == kFailed && !comm_poisoned) {
        comm_poisoned = true;  // (this ticket never reached its collectives on this rank)
        poison_message = slot->error.message();
        (void)mgx_comm_abort(opt.comm);
      }
      if (slot->state == kCompiled && !slot->mq.empty()) {
        lock.unlock();
        const auto t0 = clock::now();
        void* stream = nullptr;  // the batch object's own stream: slots overlap on the device
        int rc = mgx_batch_stream(slot->batch, &stream);
        mgx_comm* comm = opt.comm;
        if (rc == MGX_OK && slot->delta) {
          rc = RunWithDelta(slot->batch, slot->dbatch, stream);  // both indexes + the merge, on this one stream
        } else {
          static const bool kTrace = std::getenv("MGX_TRACE_HOST") != nullptr;
          const auto e0 = clock::now();
          if (rc == MGX_OK && comm) rc = mgx_batch_exchange_df(slot->batch, comm, stream);  // table-wide df before idf
          const auto e1 = clock::now();
          if (rc == MGX_OK)                                                                 // asynchronous
            rc = comm ? mgx_batch_execute_sharded(slot->batch, comm, stream) : mgx_batch_execute(slot->batch, stream);
          const auto e2 = clock::now();
          if (rc == MGX_OK && comm) rc = mgx_batch_exchange(slot->batch, comm, stream);     // all-gather + merge
          if (kTrace) {
            static std::atomic<uint64_t> n{0}, a{0}, b{0}, c{0};
            a += std::chrono::duration_cast<std::chrono::nanoseconds>(e1 - e0).count();
            b += std::chrono::duration_cast<std::chrono::nanoseconds>(e2 - e1).count();
            c += std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - e2).count();
            if (++n % 64 == 0)
              fprintf(stderr, "[shim] enqueue per batch: exchange_df %.1f us, execute %.1f us, exchange %.1f us\n",
                      a.exchange(0) / 64e3, b.exchange(0) / 64e3, c.exchange(0) / 64e3);
          }
        }
        if (rc != MGX_OK) slot->error = MakeError(static_cast<ErrorCode>(rc), mgx_last_error());
        slot->timing.enqueue_ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
        lock.lock();
        if (rc != MGX_OK && comm && !comm_poisoned) {
          comm_poisoned = true;
          poison_message = slot->error.message();
          (void)mgx_comm_abort(comm);
        }
      }
      slot->state = slot->error.code() == ErrorCode::kSuccess && slot->state == kCompiled ? kEnqueued : kFailed;
      ++next_enqueue;
      cv_state.notify_all();
    }
    enqueuing = false;
  }

  void Work() {
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
      cv_work.wait(lock, [&] { return stop || !chunks.empty(); });
      if (stop) return;
      const Chunk c = chunks.front();
      chunks.pop_front();
      lock.unlock();
      for (size_t i = c.begin; i < c.end; ++i) {
        c.slot->plans[i].Reset();  // (the slot's plans are re-used from batch to batch)
        PlanQuery(index, c.slot->queries[i], c.slot->total_docs, c.slot->avgdl, &c.slot->plans[i]);
        if (c.slot->delta) {
          c.slot->dplans[i].Reset();
          if (c.slot->plans[i].error == ErrorCode::kSuccess)
            PlanForDelta(*c.slot->delta, c.slot->queries[i], c.slot->total_docs, c.slot->avgdl, c.slot->plans[i],
                         &c.slot->dplans[i]);
        }
      }
      lock.lock();
      if (--c.slot->chunks_left != 0) continue;
      c.slot->t_planned = clock::now();
      c.slot->state = kPlanned;
      cv_state.notify_all();
    }
  }

  // Dedicated threads compile, ticket after ticket (the compile step walks the same arenas and pinned blocks every time:
  // on whichever planner finished last it cost 2-3x). Three of them take tickets in turn — compiling a batch is ~0.5 ms
  // of one thread, which on a small shard (0.3 ms of device per batch) was the step — and whoever finishes puts every
  // batch that is ready on the device IN TICKET ORDER (EnqueueReady: one thread at a time): the order a sharded table's
  // collectives need identical on every rank.
  void Dispatch() {
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
      Slot* slot = nullptr;
      cv_state.wait(lock, [&] {
        slot = ByTicket(next_compile);
        return stop || (slot && slot->state == kPlanned);
      });
      if (stop) return;
      slot->state = kCompiling;
      ++next_compile;
      lock.unlock();
      const auto t_begin = clock::now();  // (a planned batch may have waited for this thread: not compile time)
      Compile(slot);
      slot->t_compiled = clock::now();
      slot->timing.plan_ms = std::chrono::duration<double, std::milli>(slot->t_planned - slot->t_submit).count();
      slot->timing.compile_ms = std::chrono::duration<double, std::milli>(slot->t_compiled - t_begin).count();
      lock.lock();
      slot->state = slot->error.code() == ErrorCode::kSuccess ? kCompiled : kFailed;
      EnqueueReady(lock);
    }
  }
};

BatchExecutor::BatchExecutor(const index::Index& index, Options options) : impl_(std::make_unique<Impl>(index, options)) {
  index.Finalize();
  impl_->total_docs = index.Bm25DocCount();
  impl_->avgdl = index.Bm25AvgDocLength();
  impl_->slots.resize(static_cast<size_t>(std::max(1, options.depth)));
  // (threads are named so that a profile — or bench.py's per-thread CPU accounting over /proc/self/task — can tell the
  // roles apart)
  for (int t = 0; t < std::max(1, options.planner_threads); ++t)
    impl_->workers.emplace_back([this] {
      pthread_setname_np(pthread_self(), "mgx-plan");
      impl_->Work();
    });
  static const int kDispatchers = std::getenv("MGX_DISPATCHERS") ? atoi(std::getenv("MGX_DISPATCHERS")) : 3;
  const int n_dispatchers = std::max(1, std::min(kDispatchers, std::max(1, options.depth)));
  for (int t = 0; t < n_dispatchers; ++t)
    impl_->dispatchers.emplace_back([this] {
      pthread_setname_np(pthread_self(), "mgx-dispatch");
      impl_->Dispatch();
    });
}

BatchExecutor::~BatchExecutor() {
  {
    std::unique_lock<std::mutex> lock(impl_->mu);
    // batches still in the pipeline reach the device first: a collective half-issued would hang the other ranks
    impl_->cv_state.wait(lock, [&] {
      for (auto& s : impl_->slots)
        if (s.state == Impl::kPlanning || s.state == Impl::kPlanned || s.state == Impl::kCompiling ||
            s.state == Impl::kCompiled)
          return false;
      return true;
    });
    impl_->stop = true;
  }
  impl_->cv_work.notify_all();
  impl_->cv_state.notify_all();
  for (auto& w : impl_->workers) w.join();
  for (auto& d : impl_->dispatchers) d.join();
  for (auto& s : impl_->slots) {
    if (s.state == Impl::kEnqueued && s.batch && !s.mq.empty()) {
      mgx_result_view v{};
      (void)mgx_batch_fetch(s.batch, &v);  // (drain the device before the object goes)
    }
    if (s.dbatch) mgx_batch_destroy(s.dbatch);
    if (s.batch) mgx_batch_destroy(s.batch);
  }
}

Expected<uint64_t, Error> BatchExecutor::Submit(std::vector<BatchQuery>&& queries) {
  index::Index::Impl* im = impl_->index.impl();
  if (!im->dev) return MakeUnexpected(MakeError(ErrorCode::kInternalError, im->last_error));
  // a mutable table: changes recorded since the last batch reach the device now — once nothing of this executor is
  // being planned or compiled (batches already on the device finish first: ApplyMutations waits for it)
  // install: something is about to change on the device (a finished background build, or — without a staleness bound, and
  // the first time — everything recorded so far): nothing of this executor may be being planned or compiled meanwhile.
  // kick: with a staleness bound that has passed, a background build of the recorded changes starts; nothing changes yet.
  bool install = false, kick = false, mutable_table = false;
  {
    std::lock_guard<std::mutex> il(im->mu);
    const auto& mm = im->mut;
    const bool immediate = mm.staleness.count() == 0 || mm.epoch == 0;
    install = mm.build_ready || (immediate && (mm.dirty || mm.build_running));
    kick = !install && mm.dirty && !mm.build_running &&
           std::chrono::steady_clock::now() - mm.last_apply >= mm.staleness;
    mutable_table = mm.active;
  }
  if (install) {
    if (impl_->opt.comm)
      return MakeUnexpected(MakeError(ErrorCode::kNotImplemented, "a sharded table (Options::comm) is static"));
    {
      std::unique_lock<std::mutex> lock(impl_->mu);
      impl_->cv_state.wait(lock, [&] {
        for (auto& s : impl_->slots)
          if (s.state == Impl::kPlanning || s.state == Impl::kPlanned || s.state == Impl::kCompiling || s.state == Impl::kCompiled)
            return false;
        return true;
      });
    }
    const std::string merr = impl_->index.ApplyMutations();
    if (!merr.empty()) return MakeUnexpected(MakeError(ErrorCode::kInternalError, merr));
  } else if (kick) {
    const std::string merr = impl_->index.ApplyMutations();
    if (!merr.empty()) return MakeUnexpected(MakeError(ErrorCode::kInternalError, merr));
  }
  if (mutable_table) {  // (another entry point may have applied the changes: the statistics are read per batch)
    impl_->total_docs = impl_->index.Bm25DocCount();
    impl_->avgdl = impl_->index.Bm25AvgDocLength();
  }
  Impl::Slot* slot = nullptr;
  {
    std::lock_guard<std::mutex> lock(impl_->mu);
    for (auto& s : impl_->slots)
      if (s.state == Impl::kFree) {
        slot = &s;
        break;
      }
    if (!slot)
      return MakeUnexpected(MakeError(ErrorCode::kInvalidArgument,
                                      "BatchExecutor: every slot holds an unfetched batch (Wait for one first)"));
    slot->state = Impl::kFilling;  // reserved: nobody else touches it until its chunks are queued
  }
  // (outside the lock: freeing the slot's previous queries and plans and sizing the new ones is a few thousand
  // allocations, and the planner threads take the same lock for every chunk)
  slot->queries.swap(queries);  // (the caller gets the slot's previous batch back: storage to build its next one in)
  slot->plans.resize(slot->queries.size());  // (elements are reset by the planner that takes them)
  slot->total_docs = impl_->total_docs;
  slot->avgdl = impl_->avgdl;
  slot->delta = DeltaOf(impl_->index);
  if (slot->delta) slot->dplans.resize(slot->queries.size());
  slot->error = Error{ErrorCode::kSuccess, ""};
  slot->timing = Timing{};
  slot->t_submit = Impl::clock::now();
  const size_t n = slot->queries.size();
  std::lock_guard<std::mutex> lock(impl_->mu);
  slot->ticket = impl_->next_ticket++;
  slot->state = Impl::kPlanning;
  slot->chunks_left = std::max<size_t>(1, (n + Impl::kChunk - 1) / Impl::kChunk);
  if (n == 0) impl_->chunks.push_back(Impl::Chunk{slot, 0, 0});
  for (size_t a = 0; a < n; a += Impl::kChunk) impl_->chunks.push_back(Impl::Chunk{slot, a, std::min(n, a + Impl::kChunk)});
  impl_->cv_work.notify_all();
  return slot->ticket;
}

Expected<uint64_t, Error> BatchExecutor::Submit(const std::vector<BatchQuery>& queries) {
  return Submit(std::vector<BatchQuery>(queries));
}

Expected<std::vector<BatchResult>, Error> BatchExecutor::Wait(uint64_t ticket, Timing* timing) {
  std::vector<BatchResult> out;
  const Error e = WaitInto(ticket, &out, timing);
  if (e.code() != ErrorCode::kSuccess) return MakeUnexpected(e);
  return out;
}

Error BatchExecutor::Warm(const std::vector<BatchQuery>& sample, int rounds) {
  const size_t depth = impl_->slots.size();
  std::vector<uint64_t> tickets;
  std::vector<BatchResult> sink;
  for (int r = 0; r < std::max(1, rounds); ++r) {
    tickets.clear();
    for (size_t k = 0; k < depth; ++k) {  // all slots in flight at once: each one is touched in every round
      auto t = Submit(sample);
      if (!t) return t.error();
      tickets.push_back(*t);
    }
    for (uint64_t t : tickets) {
      const Error e = WaitInto(t, &sink);
      if (e.code() != ErrorCode::kSuccess) return e;
    }
  }
  return Error{ErrorCode::kSuccess, ""};
}

Error BatchExecutor::WaitInto(uint64_t ticket, std::vector<BatchResult>* results, Timing* timing) {
  std::vector<BatchResult>& out = *results;
  std::unique_lock<std::mutex> lock(impl_->mu);
  Impl::Slot* s = ticket ? impl_->ByTicket(ticket) : nullptr;
  if (!s) return MakeError(ErrorCode::kInvalidArgument, "BatchExecutor::Wait: unknown ticket");
  impl_->cv_state.wait(lock, [&] { return s->state == Impl::kEnqueued || s->state == Impl::kFailed; });
  if (s->state == Impl::kFailed) {
    const Error e = s->error;
    s->state = Impl::kFree;
    return e;
  }
  lock.unlock();  // (the slot is this caller's until it is freed below)
  mgx_result_view v{}, dv{};
  const auto t0 = Impl::clock::now();
  if (!s->mq.empty()) {
    int rc = mgx_batch_fetch(s->batch, &v);
    if (rc == MGX_OK && s->delta) rc = mgx_batch_fetch(s->dbatch, &dv);
    if (rc != MGX_OK) {
      const Error e = MakeError(static_cast<ErrorCode>(rc), mgx_last_error());
      lock.lock();
      s->state = Impl::kFree;
      return e;
    }
  }
  Collect(s->plans, v, &out, s->delta ? &dv : nullptr);
  s->timing.wait_ms = std::chrono::duration<double, std::milli>(Impl::clock::now() - t0).count();
  if (timing) *timing = s->timing;
  lock.lock();
  s->state = Impl::kFree;
  return Error{ErrorCode::kSuccess, ""};
}

// ---- MicroBatcher --------------------------------------------------------------------------------------------------

struct MicroBatcher::Impl {
  using clock = std::chrono::steady_clock;
  struct Waiter {  // one blocked Search() call
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    Expected<BatchResult, Error> result{MakeUnexpected(MakeError(ErrorCode::kInternalError, "unanswered"))};
  };
  struct InFlight {
    uint64_t ticket = 0;
    std::vector<Waiter*> waiters;
    Error submit_error{ErrorCode::kSuccess, ""};
  };
  Options opt;
  BatchExecutor executor;
  std::mutex mu;  // the forming batch, the in-flight queue, stats
  std::condition_variable cv_pending, cv_inflight, cv_room;
  std::vector<BatchQuery> pending;
  std::vector<Waiter*> pending_waiters;
  clock::time_point first_arrival;
  std::deque<InFlight> inflight;
  Stats stats;
  bool stop = false;
  std::thread former, completer;

  Impl(const index::Index& index, Options o) : opt(o), executor(index, o.executor) {}

  static void Answer(Waiter* w, Expected<BatchResult, Error>&& r) {
    std::lock_guard<std::mutex> lock(w->mu);
    w->result = std::move(r);
    w->done = true;
    w->cv.notify_one();
  }

  void Form() {
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
      cv_pending.wait(lock, [&] { return stop || !pending.empty(); });
      if (pending.empty()) return;  // (stop, and nothing left to answer)
      // the batch closes when it is full or when its first query has waited max_delay
      const auto deadline = first_arrival + opt.max_delay;
      const bool full = cv_pending.wait_until(lock, deadline, [&] { return stop || pending.size() >= opt.max_batch; });
      // no more batches in flight than the executor has slots
      cv_room.wait(lock, [&] { return inflight.size() < static_cast<size_t>(std::max(1, opt.executor.depth)); });
      InFlight f;
      std::vector<BatchQuery> batch;
      const size_t take = std::min(pending.size(), opt.max_batch);
      batch.assign(std::make_move_iterator(pending.begin()), std::make_move_iterator(pending.begin() + take));
      f.waiters.assign(pending_waiters.begin(), pending_waiters.begin() + take);
      pending.erase(pending.begin(), pending.begin() + take);
      pending_waiters.erase(pending_waiters.begin(), pending_waiters.begin() + take);
      if (!pending.empty()) first_arrival = clock::now();  // (what did not fit starts the next batch now)
      ++stats.batches;
      stats.queries += take;
      ++(full && take == opt.max_batch ? stats.closed_full : stats.closed_by_delay);
      lock.unlock();
      auto t = executor.Submit(std::move(batch));
      lock.lock();
      if (t) f.ticket = *t; else f.submit_error = t.error();
      inflight.push_back(std::move(f));
      cv_inflight.notify_one();
    }
  }

  void Complete() {
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
      cv_inflight.wait(lock, [&] { return !inflight.empty() || (stop && pending.empty() && former_done); });
      if (inflight.empty()) return;
      InFlight f = std::move(inflight.front());
      lock.unlock();
      if (f.ticket == 0) {
        for (Waiter* w : f.waiters) Answer(w, MakeUnexpected(f.submit_error));
      } else {
        auto r = executor.Wait(f.ticket);
        if (!r) {
          for (Waiter* w : f.waiters) Answer(w, MakeUnexpected(r.error()));
        } else {
          for (size_t i = 0; i < f.waiters.size(); ++i) Answer(f.waiters[i], std::move((*r)[i]));
        }
      }
      lock.lock();
      inflight.pop_front();  // (the slot is free again only now: Wait has returned)
      cv_room.notify_one();
    }
  }
  bool former_done = false;
};

MicroBatcher::MicroBatcher(const index::Index& index, Options options) : impl_(std::make_unique<Impl>(index, options)) {
  impl_->former = std::thread([this] {
    impl_->Form();
    std::lock_guard<std::mutex> lock(impl_->mu);
    impl_->former_done = true;
    impl_->cv_inflight.notify_all();
  });
  impl_->completer = std::thread([this] { impl_->Complete(); });
}

MicroBatcher::~MicroBatcher() {
  {
    std::lock_guard<std::mutex> lock(impl_->mu);
    impl_->stop = true;
  }
  impl_->cv_pending.notify_all();
  impl_->cv_inflight.notify_all();
  impl_->former.join();
  impl_->completer.join();
}

Expected<BatchResult, Error> MicroBatcher::Search(BatchQuery query) {
  Impl::Waiter w;
  {
    std::lock_guard<std::mutex> lock(impl_->mu);
    if (impl_->stop) return MakeUnexpected(MakeError(ErrorCode::kInvalidArgument, "MicroBatcher is shutting down"));
    if (impl_->pending.empty()) impl_->first_arrival = Impl::clock::now();
    impl_->pending.push_back(std::move(query));
    impl_->pending_waiters.push_back(&w);
    if (impl_->pending.size() == 1 || impl_->pending.size() >= impl_->opt.max_batch) impl_->cv_pending.notify_all();
  }
  std::unique_lock<std::mutex> lock(w.mu);
  w.cv.wait(lock, [&] { return w.done; });
  return std::move(w.result);
}

MicroBatcher::Stats MicroBatcher::GetStats() const {
  std::lock_guard<std::mutex> lock(impl_->mu);
  return impl_->stats;
}

}  // namespace search_pipeline
}  // namespace mygramdb
