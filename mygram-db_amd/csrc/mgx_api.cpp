// mgx_api.cpp — host side of libmygram_gpu.so: the C ABI of include/mygram_gpu.h.
//
// Responsibilities: device index construction (upload + skip rows + dense bitmaps), compiling queries into tile
// programs (the planning rules of search_pipeline::Execute, src/server/search_pipeline.cpp:795-869, and
// ApplyNotFilter :871-932 / ApplyFiltersWithBitmap :1196-1237 become instruction sequences), launching the kernels
// of mgx_kernels.hip, and returning results in the reference's shapes. No CPU compute path exists here: every
// operator either runs on the device or fails.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mygram_gpu.h"
#include "mgx_host.hpp"
#include "mgx_internal.hpp"
#include "mgx_launch.hpp"
#include "mygram_tools.h"

namespace mgx {

static thread_local std::string g_last_error;
void SetError(const std::string& msg) { g_last_error = msg; }
int Fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define MGX_HIP(expr)                                                                             \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess)                                                                        \
      return ::mgx::Fail(MGX_ERR_INTERNAL, std::string(#expr) + ": " + hipGetErrorString(e__));   \
  } while (0)
#define MGX_LAUNCH(expr)                                                                                    \
  do {                                                                                                      \
    int e__ = (expr);                                                                                       \
    if (e__ != 0)                                                                                           \
      return ::mgx::Fail(MGX_ERR_INTERNAL,                                                                  \
                         std::string(#expr) + ": " + hipGetErrorString(static_cast<hipError_t>(e__)));      \
  } while (0)

// Every device allocation of the library goes through here. The test hook (include/mygram_tools.h,
// mgxt_fail_device_allocs) makes chosen allocations fail, the way the reference injects Roaring allocation failures
// (MYGRAMDB_INDEX_TEST_HOOKS, posting_list.h:217-246): the failure must surface as an error code, never as an empty
// result.
static std::atomic<int> g_fail_after{-1};  // allocations still to succeed before the injected failures; -1 = hook off
static std::atomic<int> g_fail_count{0};   // how many allocations then fail
static hipError_t DeviceMalloc(void** p, size_t n) {
  if (g_fail_after.load(std::memory_order_relaxed) >= 0) {
    if (g_fail_after.load() == 0) {
      if (g_fail_count.fetch_sub(1) > 0) {
        if (g_fail_count.load() <= 0) g_fail_after.store(-1);
        *p = nullptr;
        return hipErrorOutOfMemory;
      }
      g_fail_after.store(-1);
    } else {
      g_fail_after.fetch_sub(1);
    }
  }
  return hipMalloc(p, n);
}

// Device memory handed out in pieces that never move, optionally mirrored in pinned host memory. A batch builds all
// of its (small) input arrays in the mirror and ships them with ONE asynchronous copy per chunk on the execute stream;
// its device-only outputs come from a second arena. Chunks are kept when the arena is reset, so a serving loop that
// re-prepares a batch object (mgx_batch_reset) allocates nothing in steady state.
struct Arena {
  struct Chunk {
    char* dev = nullptr;
    char* host = nullptr;  // pinned mirror (mirrored arenas only)
    size_t cap = 0, used = 0;
  };
  std::vector<Chunk> chunks;
  bool mirrored = false;
  size_t chunk_bytes = 4u << 20;
  explicit Arena(bool m) : mirrored(m) {}
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  ~Arena() {
    for (Chunk& c : chunks) {
      if (c.dev) (void)hipFree(c.dev);
      if (c.host) (void)hipHostFree(c.host);
    }
  }
  void Reset() {
    for (Chunk& c : chunks) c.used = 0;
  }
  // 256-byte aligned piece of n bytes; *mirror receives its host twin (mirrored arenas)
  hipError_t Alloc(size_t n, void** dev, void** mirror) {
    n = (std::max<size_t>(n, 16) + 255) & ~static_cast<size_t>(255);
    for (Chunk& c : chunks) {
      if (c.cap - c.used >= n) {
        *dev = c.dev + c.used;
        if (mirror) *mirror = c.host ? c.host + c.used : nullptr;
        c.used += n;
        return hipSuccess;
      }
    }
    Chunk c;
    c.cap = std::max(n, chunk_bytes);
    hipError_t e = DeviceMalloc(reinterpret_cast<void**>(&c.dev), c.cap);
    if (e != hipSuccess) return e;
    if (mirrored) {
      e = hipHostMalloc(reinterpret_cast<void**>(&c.host), c.cap, hipHostMallocDefault);
      if (e != hipSuccess) {
        (void)hipFree(c.dev);
        return e;
      }
    }
    c.used = n;
    *dev = c.dev;
    if (mirror) *mirror = c.host;
    chunks.push_back(c);
    return hipSuccess;
  }
};

// What a batch object keeps across mgx_batch_reset: its arenas, the pinned block its results are copied into, events.
struct BatchResources {
  Arena upload{true}, out{false};
  bool uploaded = false;
  void* h_down[2] = {nullptr, nullptr};  // pinned result blocks (score group, docid-page group)
  size_t h_down_cap[2] = {0, 0};
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;  // fork/join of the side-stream launch
  // A batch object is one pipeline slot: with a NULL stream argument it runs on its own non-blocking stream, and the
  // copy of its (small) result block to pinned memory is part of the execute, fenced by done_ev — so fetching batch i
  // waits for batch i alone while batch i+1 already runs.
  hipStream_t stream = nullptr;
  hipEvent_t done_ev = nullptr;
  hipEvent_t main_ev = nullptr;  // recorded behind the score group's main launches: the next batch's main launches wait for it
  bool copy_issued = false;
  // mgx_batch_exchange: this rank's blob (docid pages only) and every rank's. They belong to the slot, not to one batch:
  // a hipFree / hipMalloc per reset would synchronise the device once per step.
  void* xchg[4] = {nullptr, nullptr, nullptr, nullptr};  // 0/1: the top-k exchange, 2/3: the seed-bound exchange
  size_t xchg_cap[4] = {0, 0, 0, 0};
  hipError_t Exchange(int which, size_t bytes, void** out_ptr) {
    if (xchg_cap[which] < bytes) {
      if (xchg[which]) (void)hipFree(xchg[which]);
      xchg[which] = nullptr;
      xchg_cap[which] = 0;
      const size_t cap = bytes + bytes / 4;
      hipError_t e = DeviceMalloc(&xchg[which], cap);
      if (e != hipSuccess) return e;
      xchg_cap[which] = cap;
    }
    *out_ptr = xchg[which];
    return hipSuccess;
  }
  ~BatchResources() {
    for (void* x : xchg)
      if (x) (void)hipFree(x);
    if (stream) (void)hipStreamDestroy(stream);
    if (done_ev) (void)hipEventDestroy(done_ev);
    if (main_ev) (void)hipEventDestroy(main_ev);
    for (void* h : h_down)
      if (h) (void)hipHostFree(h);
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    if (join_ev) (void)hipEventDestroy(join_ev);
  }
  hipError_t Pinned(int which, size_t bytes, void** out_ptr) {
    if (h_down_cap[which] < bytes) {
      if (h_down[which]) (void)hipHostFree(h_down[which]);
      h_down[which] = nullptr;
      h_down_cap[which] = 0;
      const size_t cap = std::max<size_t>(bytes + bytes / 2, 1u << 16);
      hipError_t e = hipHostMalloc(&h_down[which], cap, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      h_down_cap[which] = cap;
    }
    *out_ptr = h_down[which];
    return hipSuccess;
  }
  void Reset() {
    upload.Reset();
    out.Reset();
    uploaded = false;
    copy_issued = false;
  }
};
// The resources DevBuf::Alloc / Upload draw from while a batch is being prepared on this thread (null: own hipMalloc).
static thread_local BatchResources* tl_res = nullptr;
struct ResourceScope {
  BatchResources* prev;
  explicit ResourceScope(BatchResources* r) : prev(tl_res) { tl_res = r; }
  ~ResourceScope() { tl_res = prev; }
};

// Device buffer: owns its memory (hipMalloc), or is a piece of the current batch's arena.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      Free();
      p = o.p;
      bytes = o.bytes;
      owned = o.owned;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() { Free(); }
  void Free() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    owned = true;
  }
  hipError_t Alloc(size_t n) {
    Free();
    bytes = n;
    if (tl_res) {
      owned = false;
      return tl_res->out.Alloc(n, &p, nullptr);
    }
    return DeviceMalloc(&p, n ? n : 16);
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

template <typename T>
static hipError_t Upload(DevBuf& b, const T* host, size_t count, size_t pad_count = 0) {
  if (tl_res) {  // into the batch's upload arena: shipped by the first execute
    b.Free();
    b.bytes = (count + pad_count) * sizeof(T);
    b.owned = false;
    void* mirror = nullptr;
    hipError_t e = tl_res->upload.Alloc(b.bytes, &b.p, &mirror);
    if (e != hipSuccess) return e;
    if (count) std::memcpy(mirror, host, count * sizeof(T));
    if (pad_count) std::memset(static_cast<char*>(mirror) + count * sizeof(T), 0xFF, pad_count * sizeof(T));
    return hipSuccess;
  }
  hipError_t e = b.Alloc((count + pad_count) * sizeof(T));
  if (e != hipSuccess) return e;
  if (pad_count) {
    e = hipMemset(static_cast<char*>(b.p) + count * sizeof(T), 0xFF, pad_count * sizeof(T));
    if (e != hipSuccess) return e;
  }
  if (count) e = hipMemcpy(b.p, host, count * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

}  // namespace mgx

using mgx::DevBuf;

// =================================================================================================================
// index
// =================================================================================================================

struct mgx_index {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side_stream = nullptr;  // the few queries the wave kernel cannot take run here, beside the main launch
  std::mutex mu;  // filter registration (rows / columns of the filter pool), text attachment
  // The single-operator entry points (Index::SearchAnd & co. through the shim: what an UNMODIFIED call site uses, from
  // every worker thread of the server at once, command_handler.cpp:66) each lease one of these for the call: its own
  // arenas, pinned block and stream — no lock is held while the operator runs, nothing is allocated in steady state.
  std::mutex pool_mu;
  std::vector<std::unique_ptr<mgx::BatchResources>> single_pool;  // free ones
  mgx::DevIndex dev{};
  DevBuf d_offsets, d_docids, d_tf, d_doc_len, d_skip_row, d_tile_off, d_gram_bitmaps, d_filter_bitmaps;
  DevBuf d_text, d_text_off;  // mgx_index_attach_text
  DevBuf d_dl8, d_tfnib, d_tf_ovf_pos, d_tf_ovf_val;
  DevBuf d_dev_index;  // `dev` in device memory: what out-of-line device functions take a pointer to (SyncDevIndex)
  std::vector<uint64_t> h_offsets;
  std::vector<uint32_t> h_skip_row;  // per gram
  std::vector<uint32_t> h_bm_row;    // per gram
  uint32_t n_filter_rows = 0, filter_cap_rows = 0;
  struct FilterColumn {  // mgx_index_add_filter_column
    uint32_t value_class = 0, n_values = 0;
    DevBuf d_values, d_null, d_value_ids;
  };
  std::vector<std::unique_ptr<FilterColumn>> filter_columns;
  std::vector<DevBuf> retired_filter_pools;
  uint64_t n_grams = 0;
  bool can_score = false;
  uint64_t words_per_row = 0;  // n_tiles * 256
  uint64_t filter_row_stride = 0;  // words; words_per_row + padding
  // BM25 contribution tables of the fast scoring path: idf*tf*(k1+1)/(tf + k1*(1-b+b*dl/avgdl)) for tf 0..14 and every
  // doc length below table_dl, one table per (dense gram, idf, k1, b, avgdl). These are table constants (idf comes from
  // the gram's posting count and N), so they are built once on first use and a query carries their addresses.
  struct TableKey {
    uint32_t row;
    uint64_t idf, k1, b, avg;
    bool operator==(const TableKey& o) const { return row == o.row && idf == o.idf && k1 == o.k1 && b == o.b && avg == o.avg; }
  };
  struct TableKeyHash {
    size_t operator()(const TableKey& k) const {
      uint64_t h = k.row * 0x9E3779B97F4A7C15ull;
      for (uint64_t v : {k.idf, k.k1, k.b, k.avg}) h = (h ^ v) * 0xFF51AFD7ED558CCDull + (h >> 29);
      return static_cast<size_t>(h);
    }
  };
  std::mutex table_mu;
  std::unordered_map<TableKey, uint32_t, TableKeyHash> table_slot;
  DevBuf d_table_pool;
  uint32_t table_cap = 0, table_used = 0, table_dl = 0;
  // New tables are built ON THE DEVICE (build_contrib_tables_kernel) on the index's table stream: the compile step of a
  // batch only reserves the slot and queues a 40-byte job (a fresh serving process meets new (gram, idf) pairs in almost
  // every batch until its working set exists; 3840 divisions + a blocking 30 KB copy per table were most of those
  // batches' compile time). The next execute ships the queued jobs through a pinned ring, launches the build and waits
  // for table_ev on its own stream — a no-op once the tables are there.
  hipStream_t table_stream = nullptr;
  hipEvent_t table_ev = nullptr;
  mgx::TableJob* job_ring = nullptr;  // pinned, kJobRing jobs; the device twin holds the same offsets
  DevBuf d_job_ring;
  uint32_t job_ring_at = 0;
  std::vector<mgx::TableJob> pending_jobs;  // under table_mu
  // Batches of a serving loop run on their own streams, and the device would time-slice the main kernels of all batches
  // in flight: every batch then finishes after (batches in flight) x (kernel time). The main launches of a batch
  // therefore wait for the main launches of the batch enqueued before it on this index (chain_ev: that batch's
  // BatchResources::main_ev) — first in, first out; uploads, clears, the merge and the result copy still overlap the
  // previous batch's kernels. Guarded by table_mu.
  hipEvent_t chain_ev = nullptr;
  std::atomic<int> fifo{1};  // mgx_index_set_batch_order
  bool table_ev_recorded = false;
  static constexpr uint32_t kJobRing = 4096;
  size_t table_doubles() const { return static_cast<size_t>(mgx::kFastPoolTf + 1) * table_dl; }
  // K[dl] = k1 * (1 - b + b * dl / avgdl), dl 0..255, per (k1, b, avgdl) of a batch (group_score_kernel)
  struct NormTable {
    uint64_t k1, b, avg;
    DevBuf d;
  };
  std::vector<std::unique_ptr<NormTable>> norm_tables;
  // block-max bytes of the dense grams' BM25 term factor per (k1, b, avgdl) (build_blockmax_kernel): top-k pruning
  struct BlockMax {
    uint64_t k1, b, avg;
    double step;
    DevBuf d, d_fine;
  };
  std::vector<std::unique_ptr<BlockMax>> block_max;
  // df of text-level terms seen so far: (term bytes, N) -> THIS SHARD's count (a shard of a table sums the ranks' counts
  // per batch, mgx_batch_exchange_df). The index is static, so the count never changes: the df pass (a text scan over
  // every candidate of the term) runs once per distinct term, not once per batch.
  std::unordered_map<std::string, uint64_t> df_cache;
  uint32_t n_bitmap_rows = 0, n_fine_rows = 0;
  // Mutable tables (mgx_index_set_live_bitmap): the filter row every batched query is ANDed with right after its first
  // term — the documents of this index that are still live (removed / superseded ones cleared) — or kNoRow.
  std::atomic<uint32_t> live_row{mgx::kNoRow};
  // the bitmap form of every dense gram excludes the dead documents too (mgx_index_clear_postings was called for each of
  // them): a term whose grams are all bitmap-form is live by construction and takes no live-row operand
  std::atomic<bool> bitmaps_clean{false};
  DevBuf d_doc_map;  // mgx_index_set_doc_map: local slot -> table doc id (a delta index), or empty
  DevBuf d_fine_rows;                 // fine row -> bitmap row
  std::vector<uint32_t> h_fine_map;   // bitmap row -> fine row (kNoRow: none)
};

namespace mgx {
// RCCL's by-value types, restated (rccl.h:40-41, :259-270) so that the library needs no RCCL header to build
struct RcclId {
  char internal[128];
};
constexpr int kRcclUint8 = 1, kRcclUint64 = 5, kRcclSum = 0;
static inline uint64_t Bits(double d) {
  uint64_t u;
  std::memcpy(&u, &d, 8);
  return u;
}

// Keeps the device-memory copy of idx->dev current (after creation and whenever a pointer inside it changes).
static hipError_t SyncDevIndex(mgx_index* idx) {
  if (!idx->d_dev_index.p) {
    ResourceScope none(nullptr);
    hipError_t e = idx->d_dev_index.Alloc(sizeof(DevIndex));
    if (e != hipSuccess) return e;
  }
  return hipMemcpy(idx->d_dev_index.p, &idx->dev, sizeof(DevIndex), hipMemcpyHostToDevice);
}

// Device address of the contribution table of (dense gram row, idf, k1, b, avgdl); built and uploaded on first use.
// 0 when the pool is exhausted (the query then runs on the general path).
static uint64_t GetContributionTable(mgx_index* idx, uint32_t bm_row, double idf, double k1, double b, double avgdl) {
  if (idx->table_cap == 0) return 0;
  const mgx_index::TableKey key{bm_row, Bits(idf), Bits(k1), Bits(b), Bits(avgdl)};
  std::lock_guard<std::mutex> lock(idx->table_mu);
  const auto hit = idx->table_slot.find(key);
  const size_t nd = idx->table_doubles();
  if (hit != idx->table_slot.end())
    return reinterpret_cast<uint64_t>(idx->d_table_pool.as<double>() + static_cast<size_t>(hit->second) * nd);
  if (idx->table_used == idx->table_cap) return 0;
  const uint32_t slot = idx->table_used;
  double* dst = idx->d_table_pool.as<double>() + static_cast<size_t>(slot) * nd;
  idx->pending_jobs.push_back(TableJob{idf, k1, b, avgdl, slot, 0});  // built by the next execute (FlushTableJobs)
  idx->table_used++;
  idx->table_slot.emplace(key, slot);
  return reinterpret_cast<uint64_t>(dst);
}

// Ships the queued table jobs and launches their build on the table stream; the caller then waits for table_ev on its
// own stream. Called with table_mu held.
static hipError_t FlushTableJobs(mgx_index* idx) {
  if (idx->pending_jobs.empty()) return hipSuccess;
  hipError_t e;
  if (!idx->job_ring) {
    ResourceScope none(nullptr);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&idx->job_ring), mgx_index::kJobRing * sizeof(TableJob),
                           hipHostMallocDefault)) != hipSuccess ||
        (e = idx->d_job_ring.Alloc(mgx_index::kJobRing * sizeof(TableJob))) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&idx->table_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&idx->table_ev, hipEventDisableTiming)) != hipSuccess)
      return e;
  }
  size_t done = 0;
  while (done < idx->pending_jobs.size()) {
    const uint32_t n = static_cast<uint32_t>(std::min<size_t>(idx->pending_jobs.size() - done, mgx_index::kJobRing));
    if (idx->job_ring_at + n > mgx_index::kJobRing) {  // wrap: the ring's earlier copies must have left
      if ((e = hipStreamSynchronize(idx->table_stream)) != hipSuccess) return e;
      idx->job_ring_at = 0;
    }
    TableJob* stage = idx->job_ring + idx->job_ring_at;
    TableJob* dev = idx->d_job_ring.as<TableJob>() + idx->job_ring_at;
    std::memcpy(stage, idx->pending_jobs.data() + done, n * sizeof(TableJob));
    if ((e = hipMemcpyAsync(dev, stage, n * sizeof(TableJob), hipMemcpyHostToDevice, idx->table_stream)) != hipSuccess) return e;
    const int le = LaunchBuildContribTables(dev, n, idx->table_dl, idx->d_table_pool.as<double>(), idx->table_stream);
    if (le != 0) return static_cast<hipError_t>(le);
    idx->job_ring_at += n;
    done += n;
  }
  idx->pending_jobs.clear();
  if ((e = hipEventRecord(idx->table_ev, idx->table_stream)) != hipSuccess) return e;
  idx->table_ev_recorded = true;
  return hipSuccess;
}

// Device address of K[0..255], K[dl] = k1 * ((1 - b) + b * dl / max(avgdl, 1)): the doc-length half of the BM25
// denominator (bm25_scorer.cpp:80-84), evaluated on the host operation by operation so that tf + K[dl] is the
// reference's denominator bit for bit. nullptr when it cannot be built (the queries then take the table-based path).
static const double* GetLengthNormTable(mgx_index* idx, double k1, double b, double avgdl) {
  std::lock_guard<std::mutex> lock(idx->table_mu);
  for (const auto& t : idx->norm_tables)
    if (t->k1 == Bits(k1) && t->b == Bits(b) && t->avg == Bits(avgdl)) return t->d.as<double>();
  if (idx->norm_tables.size() >= 64) return nullptr;
  double h[256];
  const double one_minus_b = 1.0 - b, avg = std::max(avgdl, 1.0);
  for (uint32_t dli = 0; dli < 256; ++dli) {
    const double length_norm = one_minus_b + b * static_cast<double>(dli) / avg;
    h[dli] = k1 * length_norm;
  }
  auto t = std::make_unique<mgx_index::NormTable>();
  t->k1 = Bits(k1);
  t->b = Bits(b);
  t->avg = Bits(avgdl);
  ResourceScope none(nullptr);  // index-owned memory, not a batch arena
  if (hipSetDevice(idx->device) != hipSuccess || t->d.Alloc(sizeof(h)) != hipSuccess ||
      hipMemcpy(t->d.p, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  idx->norm_tables.push_back(std::move(t));
  return idx->norm_tables.back()->d.as<double>();
}

// Block-max bytes of the index for (k1, b, avgdl), built on first use (one pass over the tf-nibble rows and doc
// lengths: a few ms, ~9 bytes per 64 docs per dense gram... 1 byte per word per row). nullptr when pruning is off for
// these parameters (it needs k1 > 0 and 0 <= b <= 1: the term factor must grow with tf and shrink with dl), when the
// index has no tf columns, or when the few parameter sets an index keeps are taken.
static const uint8_t* GetBlockMax(mgx_index* idx, double k1, double b, double avgdl, double* step, const uint8_t** fine) {
  *fine = nullptr;
  static const bool off = std::getenv("MGX_NO_PRUNE") != nullptr && atoi(std::getenv("MGX_NO_PRUNE")) != 0;
  if (off || !idx->can_score || idx->n_bitmap_rows == 0 || !(k1 > 0.0) || !(b >= 0.0 && b <= 1.0) || !(avgdl >= 0.0))
    return nullptr;
  {
    std::lock_guard<std::mutex> lock(idx->table_mu);
    for (const auto& t : idx->block_max)
      if (t->k1 == Bits(k1) && t->b == Bits(b) && t->avg == Bits(avgdl)) {
        *step = t->step;
        *fine = t->d_fine.as<uint8_t>();
        return t->d.as<uint8_t>();
      }
    if (idx->block_max.size() >= 4) return nullptr;
  }
  const double* ktab = GetLengthNormTable(idx, k1, b, avgdl);
  if (!ktab) return nullptr;
  std::lock_guard<std::mutex> lock(idx->table_mu);
  for (const auto& t : idx->block_max)  // (another planner thread may have built it meanwhile)
    if (t->k1 == Bits(k1) && t->b == Bits(b) && t->avg == Bits(avgdl)) {
      *step = t->step;
      *fine = t->d_fine.as<uint8_t>();
      return t->d.as<uint8_t>();
    }
  auto t = std::make_unique<mgx_index::BlockMax>();
  t->k1 = Bits(k1);
  t->b = Bits(b);
  t->avg = Bits(avgdl);
  t->step = (k1 + 1.0) / 253.0;  // the supremum k1 + 1 lands on 254 (build_blockmax_kernel: floor + 1)
  ResourceScope none(nullptr);
  const size_t bytes = static_cast<size_t>(idx->dev.n_tiles) * idx->n_bitmap_rows * 256u;
  if (hipSetDevice(idx->device) != hipSuccess || t->d.Alloc(bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  const size_t fine_bytes = static_cast<size_t>(idx->dev.n_tiles) * idx->n_fine_rows * 1024u;
  if (fine_bytes && t->d_fine.Alloc(fine_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (LaunchBuildBlockMax(idx->dev.tfnib, idx->dev.nib_row_stride, idx->dev.dl8, ktab, k1 + 1.0, 1.0 / t->step, nullptr,
                          idx->n_bitmap_rows, idx->dev.n_tiles, false, t->d.p, idx->stream) != 0 ||
      LaunchBuildBlockMax(idx->dev.tfnib, idx->dev.nib_row_stride, idx->dev.dl8, ktab, k1 + 1.0, 1.0 / t->step,
                          idx->d_fine_rows.as<uint32_t>(), idx->n_fine_rows, idx->dev.n_tiles, true, t->d_fine.p,
                          idx->stream) != 0 ||
      hipStreamSynchronize(idx->stream) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  *step = t->step;
  *fine = t->d_fine.as<uint8_t>();
  idx->block_max.push_back(std::move(t));
  return idx->block_max.back()->d.as<uint8_t>();
}
}  // namespace mgx


extern "C" {

int mgx_abi_version(void) { return MGX_ABI_VERSION; }
const char* mgx_last_error(void) { return mgx::g_last_error.c_str(); }
void mgx_free(void* p) { std::free(p); }

int mgx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---- columns ----------------------------------------------------------------------------------------------------

int mgx_columns_build(const mgx_build_params* params, const uint8_t* text_bytes, const uint64_t* text_off,
                      uint32_t first_doc_id, uint64_t n_docs, mgx_columns** out) {
  if (out) *out = nullptr;
  if (!params || !out || !text_off || (n_docs && !text_bytes && text_off[n_docs] != 0))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_columns_build: null argument");
  if (params->struct_size < sizeof(mgx_build_params) || params->version != MGX_ABI_VERSION)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_columns_build: bad struct_size/version");
  if (static_cast<uint64_t>(first_doc_id) + n_docs > 0xFFFFFFFFull)
    return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_columns_build: doc ids exceed uint32");
  try {
    mgx::Columns* c = nullptr;
    std::string err;
    int rc = mgx::BuildColumns(*params, text_bytes, text_off, first_doc_id, n_docs, &c, &err);
    if (rc != MGX_OK) return mgx::Fail(rc, err);
    *out = reinterpret_cast<mgx_columns*>(c);
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_columns_build: ") + e.what());
  }
}

int mgx_columns_from_mgix(const uint8_t* data, uint64_t len, uint32_t first_doc_id, uint64_t n_docs, mgx_columns** out,
                          mgx_mgix_info* info) {
  if (out) *out = nullptr;
  if (!data || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_columns_from_mgix: null argument");
  if (static_cast<uint64_t>(first_doc_id) + n_docs > 0xFFFFFFFFull)
    return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_columns_from_mgix: doc ids exceed uint32");
  try {
    mgx::Columns* c = nullptr;
    std::string err;
    const int rc = mgx::ColumnsFromMgix(data, len, first_doc_id, n_docs, &c, info, &err);
    if (rc != MGX_OK) return mgx::Fail(rc, err);
    *out = reinterpret_cast<mgx_columns*>(c);
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_columns_from_mgix: ") + e.what());
  }
}

int mgx_dump_open(const uint8_t* data, uint64_t len, const char* table, mgx_dump** out) {
  if (out) *out = nullptr;
  if (!data || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dump_open: null argument");
  try {
    mgx::DumpData* d = nullptr;
    std::string err;
    const int rc = mgx::DumpOpen(data, len, table, &d, &err);
    if (rc != MGX_OK) return mgx::Fail(rc, err);
    *out = reinterpret_cast<mgx_dump*>(d);
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_dump_open: ") + e.what());
  }
}
int mgx_dump_view_get(const mgx_dump* dump, mgx_dump_view* out) {
  if (!dump || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dump_view_get: null argument");
  mgx::DumpView(reinterpret_cast<const mgx::DumpData*>(dump), out);
  return MGX_OK;
}
int mgx_dump_filter_column_get(const mgx_dump* dump, uint32_t i, mgx_dump_filter_column* out) {
  if (!dump || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dump_filter_column_get: null argument");
  if (!mgx::DumpFilterColumn(reinterpret_cast<const mgx::DumpData*>(dump), i, out))
    return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_dump_filter_column_get: no such column");
  return MGX_OK;
}
int mgx_dump_take_columns(mgx_dump* dump, mgx_columns** out) {
  if (out) *out = nullptr;
  if (!dump || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dump_take_columns: null argument");
  *out = reinterpret_cast<mgx_columns*>(mgx::DumpTakeColumns(reinterpret_cast<mgx::DumpData*>(dump)));
  return MGX_OK;
}
void mgx_dump_destroy(mgx_dump* dump) { mgx::DumpDestroy(reinterpret_cast<mgx::DumpData*>(dump)); }

int mgx_columns_view_get(const mgx_columns* cols, mgx_columns_view* out) {
  if (!cols || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_columns_view_get: null argument");
  mgx::ColumnsView(reinterpret_cast<const mgx::Columns*>(cols), out);
  return MGX_OK;
}

int mgx_columns_lookup(const mgx_columns* cols, const uint8_t* gram, size_t len, uint32_t* gram_id, int* found) {
  if (gram_id) *gram_id = 0;
  if (found) *found = 0;
  if (!cols || !gram_id || !found || (!gram && len))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_columns_lookup: null argument");
  uint32_t id = 0;
  if (mgx::ColumnsLookup(reinterpret_cast<const mgx::Columns*>(cols), gram, len, &id)) {
    *gram_id = id;
    *found = 1;
  }
  return MGX_OK;
}

void mgx_columns_destroy(mgx_columns* cols) { mgx::DestroyColumns(reinterpret_cast<mgx::Columns*>(cols)); }

// ---- index ------------------------------------------------------------------------------------------------------

static int IndexCreateImpl(const mgx_index_desc* d, mgx_index** out) {
  auto idx = std::make_unique<mgx_index>();
  idx->device = d->device;
  const uint64_t G = d->n_grams;
  const uint64_t P = G ? d->offsets[G] : 0;
  const uint64_t n_docs = d->n_docs;
  if (n_docs == 0 || n_docs > 0xFFFFFFFFull || static_cast<uint64_t>(d->first_doc_id) + n_docs > 0x100000000ull)
    return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_create: n_docs must be in [1, 2^32) and ids fit uint32");
  // The contract of the descriptor is checked, not assumed: the build kernels turn every doc id into a bit / nibble
  // address without a bounds guard, so one id outside the owned range (a shard given the wrong first_doc_id) would be
  // an out-of-bounds device write. One host pass over the postings (threads over gram ranges).
  {
    for (uint64_t g = 0; g < G; ++g)
      if (d->offsets[g + 1] < d->offsets[g])
        return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_create: offsets must be non-decreasing");
    const uint64_t lo_id = d->first_doc_id, hi_id = lo_id + n_docs;  // [lo_id, hi_id)
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned n_thr = P < (1u << 22) ? 1u : hw;
    std::vector<int> bad(n_thr, 0);
    auto check = [&](unsigned t) {
      // thread t takes the grams whose postings start in its share of [0, P)
      const uint64_t p_lo = P * t / n_thr, p_hi = P * (t + 1) / n_thr;
      uint64_t g = static_cast<uint64_t>(std::lower_bound(d->offsets, d->offsets + G, p_lo) - d->offsets);
      for (; g < G && d->offsets[g] < p_hi; ++g) {
        const uint64_t a = d->offsets[g], b = d->offsets[g + 1];
        uint64_t prev = 0;
        for (uint64_t p = a; p < b; ++p) {
          const uint64_t id = d->docids[p];
          if (id < lo_id || id >= hi_id) bad[t] = MGX_ERR_OUT_OF_RANGE;
          else if (p > a && id <= prev) bad[t] = bad[t] ? bad[t] : MGX_ERR_INVALID_ARGUMENT;
          prev = id;
        }
      }
    };
    if (n_thr == 1) {
      check(0);
    } else {
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(check, t);
      for (auto& th : pool) th.join();
    }
    for (int b : bad) {
      if (b == MGX_ERR_OUT_OF_RANGE)
        return mgx::Fail(b, "mgx_index_create: a doc id lies outside [first_doc_id, first_doc_id + n_docs)");
      if (b) return mgx::Fail(b, "mgx_index_create: doc ids must ascend strictly inside every posting list");
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return mgx::Fail(MGX_ERR_INTERNAL, "no HIP device available: libmygram_gpu has no CPU fallback");
  }
  if (d->device < 0 || d->device >= ndev) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_create: bad device");
  MGX_HIP(hipSetDevice(d->device));
  MGX_HIP(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
  MGX_HIP(hipStreamCreateWithFlags(&idx->side_stream, hipStreamNonBlocking));
  const uint32_t n_tiles = static_cast<uint32_t>((n_docs + mgx::kTileDocs - 1) >> mgx::kTileShift);
  idx->n_grams = G;
  idx->h_offsets.assign(d->offsets, d->offsets + G + 1);
  idx->words_per_row = static_cast<uint64_t>(n_tiles) * mgx::kWordsPerTile;
  idx->can_score = d->tf != nullptr && d->doc_len != nullptr;

  MGX_HIP(mgx::Upload(idx->d_offsets, d->offsets, G + 1));
  MGX_HIP(mgx::Upload(idx->d_docids, d->docids, P, 4));
  uint32_t max_doc_len = 0;
  if (idx->can_score) {
    if (d->n_tf_overflow) {
      if (!d->tf_overflow_pos || !d->tf_overflow_val || d->n_tf_overflow > 0xFFFFFFFFull)
        return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_create: bad tf overflow table");
      for (uint64_t i = 0; i < d->n_tf_overflow; ++i)
        if (d->tf_overflow_pos[i] >= P || (i && d->tf_overflow_pos[i] <= d->tf_overflow_pos[i - 1]) ||
            d->tf[d->tf_overflow_pos[i]] != 255 || d->tf_overflow_val[i] < 255)
          return mgx::Fail(MGX_ERR_INVALID_ARGUMENT,
                           "mgx_index_create: tf overflow entries must ascend, point at tf bytes of 255 and hold counts >= 255");
      MGX_HIP(mgx::Upload(idx->d_tf_ovf_pos, d->tf_overflow_pos, d->n_tf_overflow));
      MGX_HIP(mgx::Upload(idx->d_tf_ovf_val, d->tf_overflow_val, d->n_tf_overflow));
    }
    MGX_HIP(mgx::Upload(idx->d_tf, d->tf, P, 4));
    MGX_HIP(mgx::Upload(idx->d_doc_len, d->doc_len, n_docs));
    for (uint64_t i = 0; i < n_docs; ++i) max_doc_len = std::max(max_doc_len, d->doc_len[i]);
  }

  // which grams get a skip row / a dense bitmap
  double dense_thr = d->dense_threshold == 0.0 ? 1.0 / 256.0 : d->dense_threshold;
  const uint64_t skip_min = std::max<uint64_t>(64, static_cast<uint64_t>(n_tiles) + 1);
  idx->h_skip_row.assign(G, mgx::kNoRow);
  idx->h_bm_row.assign(G, mgx::kNoRow);
  std::vector<uint32_t> skip_grams, bm_grams;
  std::vector<uint64_t> bm_lo, bm_hi;
  for (uint64_t g = 0; g < G; ++g) {
    const uint64_t len = d->offsets[g + 1] - d->offsets[g];
    if (len > 0xFFFFFFFFull) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "a posting list exceeds 2^32 entries");
    const bool dense = len > 0 && static_cast<double>(len) >= dense_thr * static_cast<double>(n_docs);
    if (dense) {
      idx->h_bm_row[g] = static_cast<uint32_t>(bm_grams.size());
      bm_grams.push_back(static_cast<uint32_t>(g));
      bm_lo.push_back(d->offsets[g]);
      bm_hi.push_back(d->offsets[g + 1]);
    }
    if (dense || len >= skip_min) {
      idx->h_skip_row[g] = static_cast<uint32_t>(skip_grams.size());
      skip_grams.push_back(static_cast<uint32_t>(g));
    }
  }
  MGX_HIP(mgx::Upload(idx->d_skip_row, idx->h_skip_row.data(), G));
  {
    DevBuf d_rows;
    MGX_HIP(mgx::Upload(d_rows, skip_grams.data(), skip_grams.size()));
    MGX_HIP(idx->d_tile_off.Alloc(skip_grams.size() * (static_cast<size_t>(n_tiles) + 1) * sizeof(uint32_t)));
    MGX_LAUNCH(mgx::LaunchBuildTileOff(idx->d_offsets.as<uint64_t>(), idx->d_docids.as<uint32_t>(),
                                       d_rows.as<uint32_t>(), static_cast<uint32_t>(skip_grams.size()), n_tiles,
                                       d->first_doc_id, idx->d_tile_off.as<uint32_t>(), idx->stream));
    MGX_HIP(hipStreamSynchronize(idx->stream));
  }
  if (!bm_grams.empty()) {
    DevBuf d_lo, d_hi;
    MGX_HIP(mgx::Upload(d_lo, bm_lo.data(), bm_lo.size()));
    MGX_HIP(mgx::Upload(d_hi, bm_hi.data(), bm_hi.size()));
    MGX_HIP(idx->d_gram_bitmaps.Alloc(bm_grams.size() * idx->words_per_row * sizeof(uint64_t)));
    MGX_HIP(hipMemsetAsync(idx->d_gram_bitmaps.p, 0, idx->d_gram_bitmaps.bytes, idx->stream));
    MGX_LAUNCH(mgx::LaunchBuildBitmaps(idx->d_docids.as<uint32_t>(), d_lo.as<uint64_t>(), d_hi.as<uint64_t>(),
                                       static_cast<uint32_t>(bm_grams.size()), d->first_doc_id, 0,
                                       /*tile_stride=*/bm_grams.size() * mgx::kWordsPerTile,
                                       /*row_stride=*/mgx::kWordsPerTile, idx->d_gram_bitmaps.as<uint64_t>(),
                                       idx->stream));
    MGX_HIP(hipStreamSynchronize(idx->stream));
    if (idx->can_score) {
      // tf of the dense grams by doc slot (4 bits per doc); rows padded off power-of-two strides
      const uint64_t stride = static_cast<uint64_t>(n_tiles) * (mgx::kTileDocs / 2) + 4352;
      MGX_HIP(idx->d_tfnib.Alloc(bm_grams.size() * stride + 256));
      MGX_HIP(hipMemsetAsync(idx->d_tfnib.p, 0, idx->d_tfnib.bytes, idx->stream));
      MGX_LAUNCH(mgx::LaunchBuildTfNib(idx->d_docids.as<uint32_t>(), idx->d_tf.as<uint8_t>(), d_lo.as<uint64_t>(),
                                       d_hi.as<uint64_t>(), static_cast<uint32_t>(bm_grams.size()), d->first_doc_id,
                                       stride, idx->d_tfnib.as<uint8_t>(), idx->stream));
      MGX_HIP(hipStreamSynchronize(idx->stream));
      idx->dev.nib_row_stride = stride;
    }
  }
  if (idx->can_score) {  // (sparse scored grams get tables too: keyed by gram id, see UploadGroup)
    idx->table_dl = mgx::FastTableDl(max_doc_len);
    idx->table_cap = static_cast<uint32_t>(std::min<uint64_t>(2 * bm_grams.size() + 4096, 1u << 20));
    MGX_HIP(idx->d_table_pool.Alloc(static_cast<size_t>(idx->table_cap) * idx->table_doubles() * sizeof(double)));
  }
  if (idx->can_score) {
    // padded to whole tiles (zeros): group_score_kernel copies a tile's 16 KiB into LDS whole
    std::vector<uint8_t> dl8((n_docs + mgx::kTileDocs - 1) / mgx::kTileDocs * mgx::kTileDocs, 0);
    for (uint64_t i = 0; i < n_docs; ++i) dl8[i] = static_cast<uint8_t>(std::min<uint32_t>(d->doc_len[i], 255u));
    MGX_HIP(mgx::Upload(idx->d_dl8, dl8.data(), dl8.size(), 16));
  }

  mgx::DevIndex& v = idx->dev;
  v.offsets = idx->d_offsets.as<uint64_t>();
  v.docids = idx->d_docids.as<uint32_t>();
  v.tf = idx->d_tf.as<uint8_t>();
  v.tf_ovf_pos = idx->d_tf_ovf_pos.as<uint64_t>();
  v.tf_ovf_val = idx->d_tf_ovf_val.as<uint32_t>();
  v.n_tf_ovf = idx->can_score ? static_cast<uint32_t>(d->n_tf_overflow) : 0u;
  v.doc_len = idx->d_doc_len.as<uint32_t>();
  v.dl8 = idx->d_dl8.as<uint8_t>();
  v.tfnib = idx->d_tfnib.as<uint8_t>();
  v.skip_row = idx->d_skip_row.as<uint32_t>();
  v.tile_off = idx->d_tile_off.as<uint32_t>();
  v.gram_bitmaps = idx->d_gram_bitmaps.as<uint64_t>();
  v.gb_tile_stride = bm_grams.size() * mgx::kWordsPerTile;
  idx->n_bitmap_rows = static_cast<uint32_t>(bm_grams.size());
  if (idx->can_score && !bm_grams.empty()) {
    // the grams held by more than a twentieth of the docs get block-max bytes per 16-doc quarter (GetBlockMax):
    // measured on the benchmark batch, 0.05 beats 0.2 (1.55 vs 1.62 ms) and no fine rows at all (1.97 ms)
    static const double kFineDensity = std::getenv("MGX_FINE_DENSITY") ? atof(std::getenv("MGX_FINE_DENSITY")) : 0.05;
    std::vector<uint32_t> fine_map(bm_grams.size(), mgx::kNoRow), fine_rows;
    for (size_t r = 0; r < bm_grams.size(); ++r) {
      const uint64_t sz = d->offsets[bm_grams[r] + 1] - d->offsets[bm_grams[r]];
      if (static_cast<double>(sz) > kFineDensity * static_cast<double>(n_docs)) {
        fine_map[r] = static_cast<uint32_t>(fine_rows.size());
        fine_rows.push_back(static_cast<uint32_t>(r));
      }
    }
    idx->n_fine_rows = static_cast<uint32_t>(fine_rows.size());
    idx->h_fine_map = fine_map;
    MGX_HIP(mgx::Upload(idx->d_fine_rows, fine_rows.data(), fine_rows.size(), 1));
  }
  v.gb_row_stride = mgx::kWordsPerTile;
  idx->filter_row_stride = idx->words_per_row + 32 * ((n_tiles % 2) ? 1 : 3);  // 256 B / 768 B of padding
  v.fb_tile_stride = mgx::kWordsPerTile;
  v.fb_row_stride = idx->filter_row_stride;
  v.filter_bitmaps = nullptr;
  v.first_doc_id = d->first_doc_id;
  v.n_docs = static_cast<uint32_t>(n_docs);
  v.n_tiles = n_tiles;
  v.max_doc_len = max_doc_len;
  MGX_HIP(mgx::SyncDevIndex(idx.get()));
  *out = idx.release();
  return MGX_OK;
}

int mgx_index_create(const mgx_index_desc* desc, mgx_index** out) {
  if (out) *out = nullptr;
  if (!desc || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_create: null argument");
  if (desc->struct_size < sizeof(mgx_index_desc) || desc->version != MGX_ABI_VERSION)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_create: bad struct_size/version");
  if (desc->tile_shift != 0 && desc->tile_shift != static_cast<uint32_t>(mgx::kTileShift))
    return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED, "mgx_index_create: only tile_shift 14 is built");
  if (!desc->offsets || (desc->n_grams && desc->offsets[desc->n_grams] && !desc->docids))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_create: null posting arrays");
  try {
    return IndexCreateImpl(desc, out);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_index_create: ") + e.what());
  }
}

void mgx_index_destroy(mgx_index* idx) {
  if (!idx) return;
  (void)hipSetDevice(idx->device);
  if (idx->stream) (void)hipStreamDestroy(idx->stream);
  if (idx->side_stream) (void)hipStreamDestroy(idx->side_stream);
  if (idx->table_stream) {
    (void)hipStreamSynchronize(idx->table_stream);
    (void)hipStreamDestroy(idx->table_stream);
  }
  if (idx->table_ev) (void)hipEventDestroy(idx->table_ev);
  if (idx->job_ring) (void)hipHostFree(idx->job_ring);
  delete idx;
}

int mgx_posting_size(const mgx_index* idx, uint32_t gram_id, uint64_t* out) {
  if (out) *out = 0;
  if (!idx || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_posting_size: null argument");
  if (gram_id >= idx->n_grams) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_posting_size: unknown gram id");
  *out = idx->h_offsets[gram_id + 1] - idx->h_offsets[gram_id];
  return MGX_OK;
}

int mgx_index_memory_bytes(const mgx_index* idx, uint64_t* out) {
  if (out) *out = 0;
  if (!idx || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_memory_bytes: null argument");
  *out = idx->d_offsets.bytes + idx->d_docids.bytes + idx->d_tf.bytes + idx->d_doc_len.bytes +
         idx->d_skip_row.bytes + idx->d_tile_off.bytes + idx->d_gram_bitmaps.bytes + idx->d_filter_bitmaps.bytes +
         idx->d_text.bytes + idx->d_text_off.bytes + idx->d_dl8.bytes + idx->d_tfnib.bytes + idx->d_table_pool.bytes +
         idx->d_tf_ovf_pos.bytes + idx->d_tf_ovf_val.bytes;
  {  // the per-(k1, b, avgdl) pruning arrays built so far
    std::lock_guard<std::mutex> lock(const_cast<mgx_index*>(idx)->table_mu);
    for (const auto& t : idx->block_max) *out += t->d.bytes + t->d_fine.bytes;
    for (const auto& t : idx->norm_tables) *out += t->d.bytes;
  }
  return MGX_OK;
}

int mgx_index_set_batch_order(mgx_index* idx, uint32_t order) {
  if (!idx || order > MGX_ORDER_CONCURRENT) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_set_batch_order: bad argument");
  idx->fifo.store(order == MGX_ORDER_FIFO ? 1 : 0);
  return MGX_OK;
}

int mgx_index_attach_text(mgx_index* idx, const uint8_t* text_bytes, const uint64_t* text_off) {
  if (!idx || !text_off) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_attach_text: null argument");
  const uint64_t n = idx->dev.n_docs;
  for (uint64_t i = 0; i < n; ++i)
    if (text_off[i + 1] < text_off[i] || text_off[i + 1] - text_off[i] > 0xFFFFFFFFull)
      return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_attach_text: offsets must ascend");
  const uint64_t total = text_off[n] - text_off[0];
  if (total && !text_bytes) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_attach_text: null text");
  std::lock_guard<std::mutex> lock(idx->mu);
  MGX_HIP(hipSetDevice(idx->device));
  std::vector<uint64_t> rel(n + 1);
  for (uint64_t i = 0; i <= n; ++i) rel[i] = text_off[i] - text_off[0];
  // (the text scan reads aligned 16-byte chunks up to 48 bytes past a doc's end)
  MGX_HIP(mgx::Upload(idx->d_text, text_bytes ? text_bytes + text_off[0] : nullptr, total, 128));
  MGX_HIP(mgx::Upload(idx->d_text_off, rel.data(), rel.size()));
  idx->dev.text = idx->d_text.as<uint8_t>();
  idx->dev.text_off = idx->d_text_off.as<uint64_t>();
  MGX_HIP(mgx::SyncDevIndex(idx));
  return MGX_OK;
}

// A fresh, zeroed row of the filter bitmap pool (the pool doubles when full). Called with idx->mu held; the row is
// counted (n_filter_rows) by the caller once its bits are in place.
static int NewFilterRow(mgx_index* idx, uint32_t* row_out) {
  if (idx->n_filter_rows == idx->filter_cap_rows) {
    const uint32_t ncap = idx->filter_cap_rows ? idx->filter_cap_rows * 2 : 8;
    DevBuf nb;
    MGX_HIP(nb.Alloc(static_cast<size_t>(ncap) * idx->filter_row_stride * sizeof(uint64_t)));
    if (idx->n_filter_rows)
      MGX_HIP(hipMemcpy(nb.p, idx->d_filter_bitmaps.p,
                        static_cast<size_t>(idx->n_filter_rows) * idx->filter_row_stride * sizeof(uint64_t),
                        hipMemcpyDeviceToDevice));
    // (rows are added while batches run — a FILTER condition becomes a row the first time it is seen — and a kernel in
    // flight still reads the old pool: it is retired, not freed; doubling keeps the waste below the pool's own size)
    idx->retired_filter_pools.push_back(std::move(idx->d_filter_bitmaps));
    idx->d_filter_bitmaps = std::move(nb);
    idx->filter_cap_rows = ncap;
    idx->dev.filter_bitmaps = idx->d_filter_bitmaps.as<uint64_t>();
    MGX_HIP(mgx::SyncDevIndex(idx));
  }
  const uint32_t row = idx->n_filter_rows;
  uint64_t* dst = idx->d_filter_bitmaps.as<uint64_t>() + static_cast<uint64_t>(row) * idx->filter_row_stride;
  MGX_HIP(hipMemsetAsync(dst, 0, idx->filter_row_stride * sizeof(uint64_t), idx->stream));
  *row_out = row;
  return MGX_OK;
}

int mgx_index_add_filter_column(mgx_index* idx, const mgx_filter_column_desc* d, uint32_t* out_column_id) {
  if (out_column_id) *out_column_id = 0;
  if (!idx || !d || !out_column_id || !d->values)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_add_filter_column: null argument");
  if (d->struct_size < sizeof(mgx_filter_column_desc) || d->version != MGX_ABI_VERSION)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_add_filter_column: bad struct_size/version");
  if (d->value_class > MGX_FC_DOUBLE) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_add_filter_column: unknown value class");
  const uint64_t n = idx->dev.n_docs;
  if (d->value_ids)
    for (uint64_t i = 0; i < n; ++i)
      if (d->value_ids[i] != 0xFFFFFFFFu && d->value_ids[i] >= d->n_values)
        return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_add_filter_column: a value id is not below n_values");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    MGX_HIP(hipSetDevice(idx->device));
    auto col = std::make_unique<mgx_index::FilterColumn>();
    col->value_class = d->value_class;
    col->n_values = d->value_ids ? d->n_values : 0;
    MGX_HIP(mgx::Upload(col->d_values, static_cast<const uint64_t*>(d->values), n));
    if (d->is_null) MGX_HIP(mgx::Upload(col->d_null, d->is_null, n));
    if (d->value_ids) MGX_HIP(mgx::Upload(col->d_value_ids, d->value_ids, n));
    idx->filter_columns.push_back(std::move(col));
    *out_column_id = static_cast<uint32_t>(idx->filter_columns.size() - 1);
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_index_add_filter_column: ") + e.what());
  }
}

int mgx_index_filter_compare(mgx_index* idx, uint32_t column_id, uint32_t op, uint64_t literal_bits, double eq_epsilon,
                             int null_matches, int never_matches, uint32_t* out_bitmap_id) {
  if (out_bitmap_id) *out_bitmap_id = 0;
  if (!idx || !out_bitmap_id) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_filter_compare: null argument");
  if (op > MGX_CMP_GE) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_filter_compare: unknown operator");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    if (column_id >= idx->filter_columns.size())
      return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_filter_compare: unknown filter column");
    MGX_HIP(hipSetDevice(idx->device));
    const mgx_index::FilterColumn& col = *idx->filter_columns[column_id];
    uint32_t row = 0;
    int rc = NewFilterRow(idx, &row);
    if (rc) return rc;
    uint64_t* dst = idx->d_filter_bitmaps.as<uint64_t>() + static_cast<uint64_t>(row) * idx->filter_row_stride;
    MGX_LAUNCH(mgx::LaunchFilterCompare(col.d_values.as<uint64_t>(), col.d_null.as<uint8_t>(), idx->dev.n_docs,
                                        col.value_class, op, literal_bits, eq_epsilon, null_matches ? 1u : 0u,
                                        never_matches ? 1u : 0u, dst, idx->stream));
    MGX_HIP(hipStreamSynchronize(idx->stream));
    idx->n_filter_rows++;
    *out_bitmap_id = row;
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_index_filter_compare: ") + e.what());
  }
}

int mgx_index_add_filter_bitmap(mgx_index* idx, const uint32_t* docids, uint64_t n, uint32_t* out_bitmap_id) {
  if (out_bitmap_id) *out_bitmap_id = 0;
  if (!idx || !out_bitmap_id || (n && !docids))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_add_filter_bitmap: null argument");
  std::lock_guard<std::mutex> lock(idx->mu);
  MGX_HIP(hipSetDevice(idx->device));
  for (uint64_t i = 0; i < n; ++i) {
    if (docids[i] < idx->dev.first_doc_id || docids[i] - idx->dev.first_doc_id >= idx->dev.n_docs)
      return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_add_filter_bitmap: doc id outside the index range");
  }
  uint32_t row = 0;
  {
    const int rc = NewFilterRow(idx, &row);
    if (rc) return rc;
  }
  if (n) {
    DevBuf d_ids, d_lo, d_hi;
    MGX_HIP(mgx::Upload(d_ids, docids, n));
    const uint64_t lo = 0, hi = n;
    MGX_HIP(mgx::Upload(d_lo, &lo, 1));
    MGX_HIP(mgx::Upload(d_hi, &hi, 1));
    MGX_LAUNCH(mgx::LaunchBuildBitmaps(d_ids.as<uint32_t>(), d_lo.as<uint64_t>(), d_hi.as<uint64_t>(), 1,
                                       idx->dev.first_doc_id, row, /*tile_stride=*/mgx::kWordsPerTile,
                                       /*row_stride=*/idx->filter_row_stride, idx->d_filter_bitmaps.as<uint64_t>(),
                                       idx->stream));
    MGX_HIP(hipStreamSynchronize(idx->stream));
  } else {
    MGX_HIP(hipStreamSynchronize(idx->stream));
  }
  idx->n_filter_rows++;
  *out_bitmap_id = row;
  return MGX_OK;
}

int mgx_index_update_filter_bitmap(mgx_index* idx, uint32_t bitmap_id, const uint32_t* set_docids, uint64_t n_set,
                                   const uint32_t* clear_docids, uint64_t n_clear) {
  if (!idx || (n_set && !set_docids) || (n_clear && !clear_docids))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_update_filter_bitmap: null argument");
  if (n_set + n_clear > 0xFFFFFFFFull) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_update_filter_bitmap: too many ids");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    if (bitmap_id >= idx->n_filter_rows) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_update_filter_bitmap: unknown bitmap id");
    std::vector<uint32_t> slots;
    slots.reserve(n_set + n_clear);
    for (int part = 0; part < 2; ++part) {
      const uint32_t* ids = part == 0 ? set_docids : clear_docids;
      const uint64_t n = part == 0 ? n_set : n_clear;
      for (uint64_t i = 0; i < n; ++i) {
        if (ids[i] < idx->dev.first_doc_id || ids[i] - idx->dev.first_doc_id >= idx->dev.n_docs)
          return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_update_filter_bitmap: doc id outside the index range");
        slots.push_back(ids[i] - idx->dev.first_doc_id);
      }
    }
    if (slots.empty()) return MGX_OK;
    MGX_HIP(hipSetDevice(idx->device));
    DevBuf d_slots;
    MGX_HIP(mgx::Upload(d_slots, slots.data(), slots.size()));
    uint64_t* row = idx->d_filter_bitmaps.as<uint64_t>() + static_cast<uint64_t>(bitmap_id) * idx->filter_row_stride;
    MGX_LAUNCH(mgx::LaunchUpdateBitmap(row, d_slots.as<uint32_t>(), static_cast<uint32_t>(n_set),
                                       static_cast<uint32_t>(slots.size()), idx->stream));
    MGX_HIP(hipStreamSynchronize(idx->stream));
    if (idx->live_row.load() == bitmap_id) {  // the live set changed: df counts of text-level terms are stale
      std::lock_guard<std::mutex> tl(idx->table_mu);
      idx->df_cache.clear();
    }
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_index_update_filter_bitmap: ") + e.what());
  }
}

int mgx_index_copy_text(mgx_index* idx, uint8_t* text_bytes, uint64_t capacity, uint64_t* text_off, uint64_t* total_bytes) {
  if (!idx || !total_bytes) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_copy_text: null argument");
  *total_bytes = 0;
  std::lock_guard<std::mutex> lock(idx->mu);
  if (!idx->d_text_off.p) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_copy_text: the index holds no texts (mgx_index_attach_text)");
  MGX_HIP(hipSetDevice(idx->device));
  const uint64_t n = idx->dev.n_docs;
  uint64_t total = 0;
  MGX_HIP(hipMemcpy(&total, idx->d_text_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost));
  *total_bytes = total;
  if (!text_off && !text_bytes) return MGX_OK;  // size query
  if (!text_off || (total && !text_bytes)) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_copy_text: null buffer");
  if (capacity < total) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_copy_text: buffer too small (*total_bytes holds the size)");
  MGX_HIP(hipMemcpy(text_off, idx->d_text_off.p, (n + 1) * 8, hipMemcpyDeviceToHost));
  if (total) MGX_HIP(hipMemcpy(text_bytes, idx->d_text.p, total, hipMemcpyDeviceToHost));
  return MGX_OK;
}

int mgx_index_read_text(mgx_index* idx, uint32_t doc_id, uint8_t* text, uint64_t capacity, uint64_t* len) {
  if (!idx || !len) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_read_text: null argument");
  *len = 0;
  std::lock_guard<std::mutex> lock(idx->mu);
  if (!idx->d_text_off.p) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_read_text: the index holds no texts (mgx_index_attach_text)");
  if (doc_id < idx->dev.first_doc_id || doc_id - idx->dev.first_doc_id >= idx->dev.n_docs)
    return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_read_text: doc id outside the index range");
  MGX_HIP(hipSetDevice(idx->device));
  const uint32_t slot = doc_id - idx->dev.first_doc_id;
  uint64_t off[2] = {0, 0};
  MGX_HIP(hipMemcpy(off, idx->d_text_off.as<uint64_t>() + slot, 16, hipMemcpyDeviceToHost));
  *len = off[1] - off[0];
  if (*len == 0 || !text) return MGX_OK;
  if (capacity < *len) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_read_text: buffer too small (*len holds the size)");
  MGX_HIP(hipMemcpy(text, idx->d_text.as<uint8_t>() + off[0], *len, hipMemcpyDeviceToHost));
  return MGX_OK;
}

int mgx_index_filter_column_export(mgx_index* idx, uint32_t column_id, uint64_t* values, uint8_t* is_null, uint32_t* value_ids) {
  if (!idx || !values || !is_null) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_filter_column_export: null argument");
  std::lock_guard<std::mutex> lock(idx->mu);
  if (column_id >= idx->filter_columns.size()) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_filter_column_export: unknown filter column");
  const mgx_index::FilterColumn& col = *idx->filter_columns[column_id];
  const uint64_t n = idx->dev.n_docs;
  MGX_HIP(hipSetDevice(idx->device));
  MGX_HIP(hipMemcpy(values, col.d_values.p, n * 8, hipMemcpyDeviceToHost));
  if (col.d_null.p) MGX_HIP(hipMemcpy(is_null, col.d_null.p, n, hipMemcpyDeviceToHost));
  else std::memset(is_null, 0, n);
  if (value_ids) {
    if (col.d_value_ids.p) MGX_HIP(hipMemcpy(value_ids, col.d_value_ids.p, n * 4, hipMemcpyDeviceToHost));
    else std::memset(value_ids, 0xFF, n * 4);
  }
  return MGX_OK;
}

int mgx_index_filter_column_read(mgx_index* idx, uint32_t column_id, uint32_t doc_id, uint64_t* value_bits, int* is_null,
                                 uint32_t* value_id) {
  if (!idx || !value_bits || !is_null) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_filter_column_read: null argument");
  std::lock_guard<std::mutex> lock(idx->mu);
  if (column_id >= idx->filter_columns.size()) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_filter_column_read: unknown filter column");
  if (doc_id < idx->dev.first_doc_id || doc_id - idx->dev.first_doc_id >= idx->dev.n_docs)
    return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_filter_column_read: doc id outside the index range");
  const uint32_t slot = doc_id - idx->dev.first_doc_id;
  const mgx_index::FilterColumn& col = *idx->filter_columns[column_id];
  MGX_HIP(hipSetDevice(idx->device));
  uint8_t nul = 0;
  uint32_t vid = 0xFFFFFFFFu;
  MGX_HIP(hipMemcpy(value_bits, col.d_values.as<uint64_t>() + slot, 8, hipMemcpyDeviceToHost));
  if (col.d_null.p) MGX_HIP(hipMemcpy(&nul, col.d_null.as<uint8_t>() + slot, 1, hipMemcpyDeviceToHost));
  if (col.d_value_ids.p) MGX_HIP(hipMemcpy(&vid, col.d_value_ids.as<uint32_t>() + slot, 4, hipMemcpyDeviceToHost));
  *is_null = nul ? 1 : 0;
  if (value_id) *value_id = vid;
  return MGX_OK;
}

int mgx_index_clear_postings(mgx_index* idx, const uint32_t* docids, const uint32_t* gram_ids, uint64_t n) {
  if (!idx || (n && (!docids || !gram_ids))) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_clear_postings: null argument");
  if (n > 0xFFFFFFFFull) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_clear_postings: too many postings in one call");
  try {
    std::vector<uint32_t> slots, rows;
    for (uint64_t i = 0; i < n; ++i) {
      if (docids[i] < idx->dev.first_doc_id || docids[i] - idx->dev.first_doc_id >= idx->dev.n_docs)
        return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_clear_postings: doc id outside the index range");
      if (gram_ids[i] >= idx->n_grams) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_clear_postings: unknown gram id");
      const uint32_t row = idx->h_bm_row[gram_ids[i]];
      if (row == mgx::kNoRow) continue;  // (a list-form gram: its queries keep the live-row operand)
      slots.push_back(docids[i] - idx->dev.first_doc_id);
      rows.push_back(row);
    }
    if (slots.empty()) return MGX_OK;
    std::lock_guard<std::mutex> lock(idx->mu);
    MGX_HIP(hipSetDevice(idx->device));
    DevBuf d_slots, d_rows;
    MGX_HIP(mgx::Upload(d_slots, slots.data(), slots.size()));
    MGX_HIP(mgx::Upload(d_rows, rows.data(), rows.size()));
    MGX_LAUNCH(mgx::LaunchClearGramBits(idx->d_gram_bitmaps.as<uint64_t>(), idx->dev.gb_tile_stride, idx->dev.gb_row_stride,
                                        d_slots.as<uint32_t>(), d_rows.as<uint32_t>(), static_cast<uint32_t>(slots.size()),
                                        idx->stream));
    MGX_HIP(hipStreamSynchronize(idx->stream));
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_index_clear_postings: ") + e.what());
  }
}

int mgx_index_set_live_bitmap(mgx_index* idx, uint32_t bitmap_id, int enable) {
  if (!idx) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_set_live_bitmap: null index");
  std::lock_guard<std::mutex> lock(idx->mu);
  if (enable && bitmap_id >= idx->n_filter_rows)
    return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_index_set_live_bitmap: unknown bitmap id");
  idx->live_row.store(enable ? bitmap_id : mgx::kNoRow);
  idx->bitmaps_clean.store(enable == MGX_LIVE_BITMAPS_CLEAN);
  std::lock_guard<std::mutex> tl(idx->table_mu);
  idx->df_cache.clear();
  return MGX_OK;
}

int mgx_index_set_doc_map(mgx_index* idx, const uint32_t* table_ids, uint64_t n) {
  if (!idx || !table_ids) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_set_doc_map: null argument");
  if (n != idx->dev.n_docs) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_set_doc_map: one id per doc slot of the index");
  for (uint64_t i = 1; i < n; ++i)
    if (table_ids[i] <= table_ids[i - 1])
      return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_set_doc_map: ids must ascend with the slot (rank order is kept by that)");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    MGX_HIP(hipSetDevice(idx->device));
    MGX_HIP(mgx::Upload(idx->d_doc_map, table_ids, n));
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_index_set_doc_map: ") + e.what());
  }
}

int mgx_index_invalidate_statistics(mgx_index* idx) {
  if (!idx) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_invalidate_statistics: null index");
  MGX_HIP(hipSetDevice(idx->device));
  MGX_HIP(hipDeviceSynchronize());  // nothing in flight reads the tables that go
  std::lock_guard<std::mutex> lock(idx->table_mu);
  idx->table_slot.clear();
  idx->table_used = 0;
  idx->pending_jobs.clear();
  idx->norm_tables.clear();
  idx->block_max.clear();
  idx->df_cache.clear();
  return MGX_OK;
}

int mgx_index_synchronize(mgx_index* idx) {
  if (!idx) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_index_synchronize: null index");
  MGX_HIP(hipSetDevice(idx->device));
  MGX_HIP(hipDeviceSynchronize());
  return MGX_OK;
}

}  // extern "C"

// =================================================================================================================
// query compilation
// =================================================================================================================

namespace mgx {

// One compiled query, before upload.
struct QuerySpec {
  std::vector<DevLeaf> leaves;
  std::vector<uint32_t> prog;
  std::vector<DevScoreTerm> score;
  std::vector<uint32_t> explicit_ids;  // for kLeafExplicit leaves (offsets are relative to this vector)
  uint32_t mode = kModeBitmap;
  uint32_t limit = 0, offset = 0, reverse = 0;
  uint32_t stack_depth = 0;
  double k1 = 1.2, b = 0.75, avgdl = 0.0;
  uint64_t list_postings = 0;  // sum of |L| over gram operands (algorithmic bytes = 4x)
  bool wave_ok = true;         // flat program + every scored term in its own register slot
  bool flat = true;            // LOAD/AND/OR/ANDNOT/COUNT only
  bool fast_score_ok = true;   // every scored term reads a dense gram's tf column (no text-level / absent terms)
  double est_density = 0.0;    // estimated fraction of docs that match (work per tile grows with it)
  // text-level scored terms (mgx_term.text): pattern bytes and the grams whose AND is the term's candidate set
  struct TextTerm {
    std::string pattern;
    std::vector<uint32_t> grams;
    uint32_t score_index = 0;  // position among the query's scored terms
  };
  std::vector<TextTerm> text_terms;
  std::vector<std::string> verify_patterns;  // kOpVerifyText: every positive term's text
  uint64_t total_docs = 0;       // N of ComputeIDF for the text-level terms
  // SORT _score pages the fused top-k does not hold (offset + limit > kMaxNeeded, or limit 0 = all): every match is
  // materialised, scored and fully sorted (ResultSorter::SortByScore's own shape, result_sorter.cpp:661-716)
  bool deep_score = false;
  std::vector<uint32_t> deep_grams;  // the scored terms' gram ids (kNoRow: a gram this shard lacks)
  std::vector<double> deep_idfs;
  uint32_t pat_off = 0, pat_len = 0;  // kModeTextDf specs: the term in the batch's pattern pool
  uint32_t vt_begin = 0;              // first of the query's verify patterns in the batch-wide array
  // back to the freshly constructed state, keeping the vectors' storage (a re-prepared batch object compiles its new
  // queries into the specs of the old ones)
  void Clear() {
    leaves.clear();
    prog.clear();
    score.clear();
    explicit_ids.clear();
    text_terms.clear();
    verify_patterns.clear();
    deep_score = false;
    deep_grams.clear();
    deep_idfs.clear();
    mode = kModeBitmap;
    limit = offset = reverse = stack_depth = 0;
    k1 = 1.2;
    b = 0.75;
    avgdl = 0.0;
    list_postings = 0;
    wave_ok = flat = fast_score_ok = true;
    est_density = 0.0;
    total_docs = 0;
    pat_off = pat_len = vt_begin = 0;
  }
};

struct Compiler {
  const mgx_index* idx;
  QuerySpec* q;
  uint32_t sp = 0;

  uint32_t GramLeaf(uint32_t g) {
    // (a query has a handful of operands: a scan beats a hash map and allocates nothing)
    for (uint32_t i = 0; i < q->leaves.size(); ++i) {
      const DevLeaf& l = q->leaves[i];
      if (g == MGX_GRAM_ABSENT ? (l.kind == kLeafRange && l.row == kNoRow && l.a == 0 && l.b == 0 && l.score_slot == kAbsentMark)
                               : ((l.kind == kLeafList || l.kind == kLeafGramBitmap) && l.a == g))
        return i;
    }
    DevLeaf lf{};
    lf.score_slot = kNoSlot;
    if (g == MGX_GRAM_ABSENT) {  // known to the table, no posting in this shard: the empty slot range
      lf.kind = kLeafRange;
      lf.row = kNoRow;
      lf.score_slot = kAbsentMark;  // (tells this operand from the expression compiler's own empty ranges)
      const uint32_t id = static_cast<uint32_t>(q->leaves.size());
      q->leaves.push_back(lf);
      return id;
    }
    lf.a = g;
    lf.row = idx->h_skip_row[g];
    if (idx->h_bm_row[g] != kNoRow) {
      lf.kind = kLeafGramBitmap;
      lf.b = idx->h_bm_row[g];
    } else {
      lf.kind = kLeafList;
    }
    const uint32_t id = static_cast<uint32_t>(q->leaves.size());
    q->leaves.push_back(lf);
    q->list_postings += idx->h_offsets[g + 1] - idx->h_offsets[g];
    return id;
  }
  void Emit(Op op, uint32_t arg = 0) {
    if (op == kOpCount) {
      arg = 1u << arg;  // slot -> mask; consecutive counts of the same accumulator share one instruction
      if (!q->prog.empty() && (q->prog.back() >> 24) == kOpCount) {
        q->prog.back() |= arg;
        return;
      }
    }
    q->prog.push_back(MakeInstr(op, arg));
    if (op != kOpLoad && op != kOpAnd && op != kOpOr && op != kOpAndNot && op != kOpCount) q->wave_ok = q->flat = false;
    if (op == kOpPush) {
      ++sp;
      q->stack_depth = std::max(q->stack_depth, sp);
    } else if (op == kOpPopAnd || op == kOpPopOr || op == kOpPopAndNot) {
      --sp;
    }
  }
  // acc = the term's doc set: AND of its grams, or "at least threshold of them"
  // (SearchTermDocuments, search_pipeline.cpp:438-446; fuzzy terms :1697-1702)
  uint32_t RangeLeaf(uint64_t lo, uint64_t hi) {  // doc slots [lo, hi) of this shard
    DevLeaf lf{};
    lf.score_slot = kNoSlot;
    lf.row = kNoRow;
    lf.kind = kLeafRange;
    lf.a = static_cast<uint32_t>(lo);
    lf.b = static_cast<uint32_t>(hi);
    q->leaves.push_back(lf);
    return static_cast<uint32_t>(q->leaves.size() - 1);
  }
  uint32_t FilterLeaf(uint32_t row) {
    DevLeaf lf{};
    lf.score_slot = kNoSlot;
    lf.row = kNoRow;
    lf.kind = kLeafFilterBitmap;
    lf.b = row;
    q->leaves.push_back(lf);
    return static_cast<uint32_t>(q->leaves.size() - 1);
  }
  // kOpVerifyText argument for ONE more pattern of the query: its index among the query's patterns | count 1 << 12
  uint32_t OnePattern(const mgx_term& t) {
    q->verify_patterns.emplace_back(reinterpret_cast<const char*>(t.text), t.text_len);
    return static_cast<uint32_t>(q->verify_patterns.size() - 1) | (1u << 12);
  }
  void LoadTerm(const mgx_term& t) {
    if (t.n_grams == 0) {
      // a term shorter than one n-gram: query::SearchNormalizedSubstring (src/query/substring_search.h:24-42) — every
      // doc whose stored text contains it (SearchTermDocuments, search_pipeline.cpp:438-446)
      Emit(kOpLoad, RangeLeaf(0, idx->dev.n_docs));
      Emit(kOpVerifyText, OnePattern(t));
      return;
    }
    if (t.threshold == 1 && t.n_grams > 1) {
      // "at least one of its n-grams" is their union (FUZZY on a short term: theta = max(1, n - d * n_eff),
      // search_pipeline.cpp:1700-1703): a flat program, which the wave kernels take
      Emit(kOpLoad, GramLeaf(t.gram_ids[0]));
      for (uint32_t i = 1; i < t.n_grams; ++i) Emit(kOpOr, GramLeaf(t.gram_ids[i]));
    } else if (t.threshold != 0 && t.threshold < t.n_grams) {
      Emit(kOpThreshBegin);
      for (uint32_t i = 0; i < t.n_grams; ++i) Emit(kOpThreshAdd, GramLeaf(t.gram_ids[i]));
      Emit(kOpThreshEnd, t.threshold);
    } else {
      Emit(kOpLoad, GramLeaf(t.gram_ids[0]));
      for (uint32_t i = 1; i < t.n_grams; ++i) Emit(kOpAnd, GramLeaf(t.gram_ids[i]));
    }
  }
};

static int ValidateTerm(const mgx_index* idx, const mgx_term& t, const char* what) {
  if (t.n_grams == 0) {  // substring term: needs its text and the docs' texts
    if (!t.text || t.text_len == 0 || t.text_len > 4096)
      return Fail(MGX_ERR_INVALID_ARGUMENT, std::string(what) + ": a term without grams needs its normalized text (1..4096 bytes)");
    if (!idx->dev.text)
      return Fail(MGX_ERR_INVALID_ARGUMENT, std::string(what) + ": a term without grams (substring search) needs mgx_index_attach_text");
    return MGX_OK;
  }
  if (!t.gram_ids) return Fail(MGX_ERR_INVALID_ARGUMENT, std::string(what) + ": term without grams");
  if (t.n_grams > 127) return Fail(MGX_ERR_OUT_OF_RANGE, std::string(what) + ": more than 127 grams in a term");
  for (uint32_t i = 0; i < t.n_grams; ++i)
    if (t.gram_ids[i] >= idx->n_grams && t.gram_ids[i] != MGX_GRAM_ABSENT)
      return Fail(MGX_ERR_OUT_OF_RANGE, std::string(what) + ": unknown gram id");
  return MGX_OK;
}

// search_pipeline::Execute (:812-853) as a tile program. Counter slots: 0 total_candidates, 1 after_intersection,
// 2 after_not, 3 after_filters.
static int CompileQuery(const mgx_index* idx, const mgx_query& in, QuerySpec* out) {
  if (in.n_terms == 0 || !in.terms) return Fail(MGX_ERR_INVALID_ARGUMENT, "query without positive terms");
  if (in.n_terms > MGX_MAX_TERMS || in.n_not_terms > MGX_MAX_TERMS || in.n_filters > MGX_MAX_TERMS)
    return Fail(MGX_ERR_OUT_OF_RANGE, "more than 64 terms / NOT terms / filters (query_parser.h:270-272)");
  Compiler c{idx, out, 0};
  for (uint32_t i = 0; i < in.n_terms; ++i) {
    int rc = ValidateTerm(idx, in.terms[i], "term");
    if (rc) return rc;
  }
  for (uint32_t i = 0; i < in.n_not_terms; ++i) {
    int rc = ValidateTerm(idx, in.not_terms[i], "NOT term");
    if (rc) return rc;
  }
  {
    double dens = 1.0;
    for (uint32_t i = 0; i < in.n_terms; ++i) {
      uint64_t mn = in.terms[i].n_grams ? ~0ull : idx->dev.n_docs;  // (a substring term filters, it does not narrow the estimate)
      for (uint32_t g = 0; g < in.terms[i].n_grams; ++g) {
        const uint32_t id = in.terms[i].gram_ids[g];
        mn = std::min<uint64_t>(mn, id == MGX_GRAM_ABSENT ? 0 : idx->h_offsets[id + 1] - idx->h_offsets[id]);
      }
      dens *= static_cast<double>(mn) / static_cast<double>(std::max<uint32_t>(idx->dev.n_docs, 1));
    }
    out->est_density = dens;
  }
  // a mutable table's main index: only its live documents exist (a removed document is in no posting list of the
  // reference, index.cpp:148-197), so every funnel counter counts live documents only
  const uint32_t live_row = idx->live_row.load(std::memory_order_relaxed);
  const bool bitmaps_clean = idx->bitmaps_clean.load(std::memory_order_relaxed);
  // (a term that is an AND of bitmap-form grams holds live documents only when the index keeps its bitmaps clean)
  auto live_by_construction = [&](const mgx_term& t) {
    if (!bitmaps_clean || t.n_grams == 0 || (t.threshold != 0 && t.threshold < t.n_grams)) return false;
    for (uint32_t g = 0; g < t.n_grams; ++g)
      if (t.gram_ids[g] != MGX_GRAM_ABSENT && idx->h_bm_row[t.gram_ids[g]] == kNoRow) return false;
    return true;
  };
  auto and_live = [&](const mgx_term* first_term = nullptr) {
    if (live_row == kNoRow) return;
    if (first_term != nullptr && live_by_construction(*first_term)) return;
    c.Emit(kOpAnd, c.FilterLeaf(live_row));
  };
  if (in.n_expr > 0) {
    // ExecuteWithBooleanAst: the positive part is a boolean tree over the terms. Postfix -> tree -> accumulator code.
    if (!in.expr) return Fail(MGX_ERR_INVALID_ARGUMENT, "n_expr without expr");
    if (in.n_expr > 512) return Fail(MGX_ERR_OUT_OF_RANGE, "expression longer than 512 tokens");
    struct Node {
      uint32_t op, arg;
      std::vector<int> kids;
    };
    std::vector<Node> nodes;
    std::vector<int> st;
    for (uint32_t k = 0; k < in.n_expr; ++k) {
      const mgx_expr_token tk = in.expr[k];
      Node n{tk.op, tk.arg, {}};
      if (tk.op == MGX_EXPR_TERM) {
        if (tk.arg >= in.n_terms) return Fail(MGX_ERR_OUT_OF_RANGE, "expression refers to a term that is not given");
      } else if (tk.op == MGX_EXPR_AND || tk.op == MGX_EXPR_OR || tk.op == MGX_EXPR_NOT) {
        const uint32_t nk = tk.op == MGX_EXPR_NOT ? 1u : tk.arg;
        if (nk == 0 || nk > st.size()) return Fail(MGX_ERR_INVALID_ARGUMENT, "malformed postfix expression");
        n.kids.assign(st.end() - nk, st.end());
        st.resize(st.size() - nk);
      } else if (tk.op != MGX_EXPR_EMPTY) {
        return Fail(MGX_ERR_INVALID_ARGUMENT, "unknown expression token");
      }
      nodes.push_back(std::move(n));
      st.push_back(static_cast<int>(nodes.size()) - 1);
    }
    if (st.size() != 1) return Fail(MGX_ERR_INVALID_ARGUMENT, "postfix expression does not reduce to one result");
    // operand for the NOT universe / the empty set: a slot range of this shard
    auto range_leaf = [&](uint64_t lo, uint64_t hi) {
      DevLeaf lf{};
      lf.score_slot = kNoSlot;
      lf.row = kNoRow;
      lf.kind = kLeafRange;
      lf.a = static_cast<uint32_t>(lo);
      lf.b = static_cast<uint32_t>(hi);
      out->leaves.push_back(lf);
      return static_cast<uint32_t>(out->leaves.size() - 1);
    };
    const uint64_t base = idx->dev.first_doc_id, span = idx->dev.n_docs;
    uint64_t ulo = 0, uhi = span;
    if (in.universe_count != 0) {
      const uint64_t a = in.universe_first, b2 = a + in.universe_count;
      ulo = a > base ? std::min(a - base, span) : 0;
      uhi = b2 > base ? std::min(b2 - base, span) : 0;
    }
    uint32_t universe = kNoRow, empty = kNoRow;
    std::function<void(int)> emit = [&](int ni) {
      const Node& n = nodes[ni];
      switch (n.op) {
        case MGX_EXPR_TERM: c.LoadTerm(in.terms[n.arg]); break;
        case MGX_EXPR_EMPTY:
          if (empty == kNoRow) empty = range_leaf(0, 0);
          c.Emit(kOpLoad, empty);
          break;
        case MGX_EXPR_AND:
        case MGX_EXPR_OR:
          emit(n.kids[0]);
          for (size_t k = 1; k < n.kids.size(); ++k) {
            c.Emit(kOpPush);
            emit(n.kids[k]);
            c.Emit(n.op == MGX_EXPR_AND ? kOpPopAnd : kOpPopOr);
          }
          break;
        default:  // NOT: universe & ~child
          if (universe == kNoRow) universe = range_leaf(ulo, uhi);
          c.Emit(kOpLoad, universe);
          c.Emit(kOpPush);
          emit(n.kids[0]);
          c.Emit(kOpPopAndNot);
          break;
      }
    };
    emit(st[0]);
    if (out->stack_depth > 40) return Fail(MGX_ERR_OUT_OF_RANGE, "expression nests deeper than 40 levels");
    and_live();
    c.Emit(kOpCount, 0);
    c.Emit(kOpCount, 1);
  } else {
    c.LoadTerm(in.terms[0]);
    and_live(&in.terms[0]);
    c.Emit(kOpCount, 0);
    for (uint32_t i = 1; i < in.n_terms; ++i) {
      const mgx_term& t = in.terms[i];
      if (t.n_grams == 0) {  // substring term: the accumulator's docs whose text contains it (search_pipeline.cpp:817-826)
        c.Emit(kOpVerifyText, c.OnePattern(t));
      } else if (t.threshold != 0 && t.threshold < t.n_grams) {
        c.Emit(kOpPush);
        c.LoadTerm(t);
        c.Emit(kOpPopAnd);
      } else {
        for (uint32_t g = 0; g < t.n_grams; ++g) c.Emit(kOpAnd, c.GramLeaf(t.gram_ids[g]));
      }
    }
    c.Emit(kOpCount, 1);
  }
  for (uint32_t i = 0; i < in.n_not_terms; ++i) {
    const mgx_term& t = in.not_terms[i];
    if (t.n_grams == 1 && !(t.threshold != 0 && t.threshold < t.n_grams)) {
      c.Emit(kOpAndNot, c.GramLeaf(t.gram_ids[0]));
    } else {
      c.Emit(kOpPush);
      c.LoadTerm(t);
      c.Emit(kOpPopAndNot);
    }
  }
  c.Emit(kOpCount, 2);
  for (uint32_t i = 0; i < in.n_filters; ++i) {
    if (in.filters[i].bitmap_id >= idx->n_filter_rows) return Fail(MGX_ERR_OUT_OF_RANGE, "unknown filter bitmap id");
    DevLeaf lf{};
    lf.score_slot = kNoSlot;
    lf.row = kNoRow;
    lf.kind = kLeafFilterBitmap;
    lf.b = in.filters[i].bitmap_id;
    const uint32_t id = static_cast<uint32_t>(out->leaves.size());
    out->leaves.push_back(lf);
    c.Emit(in.filters[i].negate ? kOpAndNot : kOpAnd, id);
  }
  c.Emit(kOpCount, 3);
  if (in.exact_text) {
    if (!idx->dev.text) return Fail(MGX_ERR_INVALID_ARGUMENT, "exact_text without mgx_index_attach_text");
    const uint32_t first_exact = static_cast<uint32_t>(out->verify_patterns.size());
    for (uint32_t i = 0; i < in.n_terms; ++i) {
      const mgx_term& t = in.terms[i];
      if (!t.text || t.text_len == 0 || t.text_len > 4096)
        return Fail(MGX_ERR_INVALID_ARGUMENT, "exact_text: every positive term needs its normalized text (1..4096 bytes)");
      out->verify_patterns.emplace_back(reinterpret_cast<const char*>(t.text), t.text_len);
    }
    // (not a flat op: the query runs on the general workgroup kernel) argument: first pattern | count << 12
    c.Emit(kOpVerifyText, first_exact | (in.n_terms << 12));
  }

  out->limit = in.limit;
  out->offset = in.offset;
  out->reverse = in.reverse;
  out->k1 = in.k1;
  out->b = in.b;
  out->avgdl = in.avg_doc_length;
  if (in.sort == MGX_SORT_SCORE) {
    if (!idx->can_score) return Fail(MGX_ERR_NOT_IMPLEMENTED, "index was created without tf/doc_len columns");
    // the scored terms: the caller's list (expression / FUZZY queries: search_handler.cpp:428-456 scores the positive
    // terms of the tree or query, whatever branch produced the result set), or every positive term in order
    const uint32_t n_scored = in.score_terms ? in.n_score_terms : in.n_terms;
    auto scored_index = [&](uint32_t k) { return in.score_terms ? in.score_terms[k] : k; };
    for (uint32_t k = 0; k < n_scored; ++k)
      if (scored_index(k) >= in.n_terms) return Fail(MGX_ERR_OUT_OF_RANGE, "score_terms refers to a term that is not given");
    const uint64_t needed = static_cast<uint64_t>(in.offset) + in.limit;
    if (in.limit == 0 || needed > kMaxNeeded) {
      // deep page (the reference is benchmarked with OFFSET 10000): the full-sort fallback. The matches are
      // materialised like a docid-ordered result, then scored and sorted whole at fetch time.
      for (uint32_t k = 0; k < n_scored; ++k) {
        const mgx_term& t = in.terms[scored_index(k)];
        if (t.text != nullptr || t.n_grams != 1)
          return Fail(MGX_ERR_NOT_IMPLEMENTED,
                      "SORT _score beyond offset+limit 1024 with a text-level term: score with "
                      "mgx_score_documents_text and page with mgx_sort_by_score");
        out->deep_grams.push_back(t.gram_ids[0] == MGX_GRAM_ABSENT ? kNoRow : t.gram_ids[0]);
        out->deep_idfs.push_back(t.idf);
      }
      out->deep_score = true;
      out->mode = kModeBitmap;
      if (out->leaves.size() > kMaxLeaves)
        return Fail(MGX_ERR_NOT_IMPLEMENTED, "more than 192 distinct operands in one query");
      return MGX_OK;
    }
    if (n_scored > kMaxScoreTerms) return Fail(MGX_ERR_NOT_IMPLEMENTED, "more than 64 scored terms");
    out->mode = kModeScore;
    out->total_docs = in.total_docs;
    if (in.score_terms) out->wave_ok = out->fast_score_ok = false;  // (a listed term need not be an AND operand of the result: the general kernel tests membership)
    for (uint32_t i = 0; i < n_scored; ++i) {
      const mgx_term& t = in.terms[scored_index(i)];
      if (t.text != nullptr) {
        // tf from the doc text, df from a scan of the term's candidates (N1: terms that are not one n-gram)
        if (!idx->dev.text)
          return Fail(MGX_ERR_INVALID_ARGUMENT, "text-level scored term without mgx_index_attach_text");
        if (t.text_len == 0 || t.text_len > 4096)
          return Fail(MGX_ERR_OUT_OF_RANGE, "text-level scored term: length must be 1..4096 bytes");
        // (a FUZZY term is scored as the exact term: tf = its occurrences in the text, df = the candidates of the AND of
        // ALL its n-grams whose text contains it — PopulateTermDocumentFrequency does not know about the threshold)
        QuerySpec::TextTerm tt;
        tt.pattern.assign(reinterpret_cast<const char*>(t.text), t.text_len);
        tt.grams.assign(t.gram_ids, t.gram_ids + t.n_grams);
        tt.score_index = i;
        out->text_terms.push_back(std::move(tt));
        DevScoreTerm st{};
        st.leaf = kNoLeaf;
        out->wave_ok = out->fast_score_ok = false;  // the wave kernels score from tf columns only
        out->score.push_back(st);
        continue;
      }
      if (t.n_grams != 1)
        return Fail(MGX_ERR_INVALID_ARGUMENT,
                    "a scored term of several n-grams needs its text (mgx_term.text) for tf and df");
      DevScoreTerm st{};
      st.leaf = c.GramLeaf(t.gram_ids[0]);
      st.idf = t.idf;
      // the same gram scored twice (a repeated term) keeps the first slot; the block kernel handles that shape
      if (t.gram_ids[0] == MGX_GRAM_ABSENT)
        out->wave_ok = out->fast_score_ok = false;  // an empty operand: never a member, contributes nothing
      else if (out->leaves[st.leaf].score_slot == kNoSlot && i < static_cast<uint32_t>(kWaveScoreSlots))
        out->leaves[st.leaf].score_slot = i;
      else
        out->wave_ok = false;
      out->score.push_back(st);
    }
  } else if (in.sort == MGX_SORT_DOCID) {
    // a bounded page in docid order needs no materialised result bitmap: count pass + page pass
    out->mode = (in.limit != 0 && in.limit <= kMaxDocPage) ? kModeDocPage : kModeBitmap;
    out->offset = 0;  // (pagination offset belongs to SORT _score, search_handler.cpp:469-470)
  } else {
    return Fail(MGX_ERR_INVALID_ARGUMENT, "unknown sort");
  }
  if (out->leaves.size() > kMaxLeaves)
    return Fail(MGX_ERR_NOT_IMPLEMENTED, "more than 192 distinct operands in one query");
  return MGX_OK;
}

}  // namespace mgx

// =================================================================================================================
// batch
// =================================================================================================================

struct mgx_batch {
  mgx_index* idx = nullptr;
  std::unique_ptr<mgx::BatchResources> res_owned;  // kept across mgx_batch_reset
  mgx::BatchResources* res = nullptr;              // = res_owned.get(), or the index's single-operator resources
  uint32_t n_queries = 0;
  std::vector<mgx::QuerySpec> specs;
  // per mode: the queries of that mode, in batch order
  struct Group {
    std::vector<uint32_t> qids;  // batch index of each member
    mgx::DevBatch dev{};
    mgx::LdsPlan plan{};
    mgx::WavePlan wplan{};
    // score mode: queries the wave kernel can run (flat program, dense scored operands) and the rest are launched
    // separately, over disjoint item lists, into the same candidate arrays
    mgx::DevBatch dev_wave{};
    DevBuf d_items_wave, d_tables;
    uint32_t n_items_wave = 0;
    // score mode: wave-kernel queries with a sorted-list operand (bigger LDS plan, launched beside the others)
    mgx::WavePlan wplan_lists{};
    mgx::DevBatch dev_wave_lists{};
    DevBuf d_items_wave_lists;
    // score mode, fast path (bitmap_score_kernel<T>): resolved query descriptors and one item list per number of scored terms
    mgx::FastPlan fplan{};
    mgx::DevBatch dev_fast[mgx::kFastMaxScore]{};
    uint32_t n_seed_fast[mgx::kFastMaxScore] = {0, 0, 0, 0, 0};  // seed items at the front of d_items_fast[t]
    uint32_t seed_k = 0;       // keys per query in the seed-bound exchange (0: none; rank-independent, see UploadGroup)
    uint64_t seed_table_docs = 0;  // table-wide doc count of those queries (BM25 N): the exchange pays on large shards only
    DevBuf d_has_seed;         // [n] u8: the query's first candidate list is a seed item's
    DevBuf d_fast_queries, d_items_fast[mgx::kFastMaxScore];
    // selective flat queries (a sparse posting list drives: cand_kernel), score and docid-page groups
    mgx::DevBatch dev_cand{};
    DevBuf d_items_cand, d_cand_skip;  // d_cand_skip [n] u8: the query is candidate-driven (docid-page group)
    uint32_t cand_leaves = 0, cand_instr = 0, cand_cap = 0;
    // SORT _score over long sorted posting arrays (merge_score_kernel: the driver's tile segments against staged bitmaps)
    mgx::DevBatch dev_merge{};
    DevBuf d_items_merge;
    uint32_t merge_leaves = 0, merge_instr = 0, merge_ops = 0, merge_cap = 0;
    // docid-page group: the page pass runs one workgroup per query; flat programs on the wave kernel
    mgx::DevBatch dev_page_wave{}, dev_page_block{};
    DevBuf d_pq_wave, d_pq_block;
    DevBuf d_queries, d_leaves, d_prog, d_score, d_explicit, d_counters, d_ident, d_items, d_list_begin;
    uint32_t n_items = 0;
    uint32_t n_items_plain = 0;  // block-kernel items [0, n_items_plain): queries of bitmap-form operands only (LaunchTileEval)
    std::vector<unsigned long long> h_counters;
  };
  Group score, bitmap;
  // df pass of the text-level scored terms: one kModeTextDf query per term (the AND of its grams + a text scan)
  Group textdf;
  std::vector<mgx::QuerySpec> df_specs;
  DevBuf d_patterns, d_text_terms, d_text_idf, d_text_df, d_verify_terms;
  // d_text_df is the array ranks all-reduce in place (table-wide df after the reduction); d_text_df_local keeps this
  // shard's own counts — the only thing the index's df cache ever holds, so what a rank contributes to the reduction is
  // the same whether a term's count came from the cache or from the df pass
  DevBuf d_text_df_local, d_text_df_known;
  std::vector<uint64_t> h_text_df, h_text_df_local, text_total_docs;
  std::vector<uint64_t> text_df_known;      // per text term: its cached LOCAL df, or ~0ull (the df query then counts it)
  std::vector<std::string> text_df_key;     // per text term: the cache key
  std::vector<double> h_text_idf;
  bool df_ready = false;  // mgx_batch_count_df ran (and the caller summed the counts) for the next execute
  // score group outputs
  DevBuf d_cand_keys, d_cand_docs, d_cand_n;
  // This shard's merged top-(offset+limit) per query in the exchange layout of mgx_batch_export_topk, written in place
  // by the merge kernel: [keys n*S u64 | totals n u64 | docs n*S u32 | counts n u32]  (S = top_stride)
  DevBuf d_export;
  size_t ex_off32 = 0, ex_bytes = 0;
  uint64_t* ex_keys() const { return d_export.as<uint64_t>(); }
  uint64_t* ex_totals(size_t n) const { return d_export.as<uint64_t>() + n * top_stride; }
  uint32_t* ex_docs() const { return reinterpret_cast<uint32_t*>(static_cast<char*>(d_export.p) + ex_off32); }
  uint32_t* ex_counts(size_t n) const { return ex_docs() + n * top_stride; }
  // Everything mgx_batch_fetch needs from a score group sits in ONE device block, copied with one async memcpy into
  // pinned host memory: [counters n*9 u64 | total_override n u64 | page_scores n*L f64 | page_docs n*L u32 | page_n n u32]
  DevBuf d_score_out;
  void* h_score_out = nullptr;  // pinned (BatchResources)
  size_t so_override = 0, so_scores = 0, so_docs = 0, so_n = 0, so_bytes = 0;
  unsigned long long* sc_counters() const { return d_score_out.as<unsigned long long>(); }
  uint64_t* sc_override() const { return reinterpret_cast<uint64_t*>(static_cast<char*>(d_score_out.p) + so_override); }
  double* sc_scores() const { return reinterpret_cast<double*>(static_cast<char*>(d_score_out.p) + so_scores); }
  uint32_t* sc_docs() const { return reinterpret_cast<uint32_t*>(static_cast<char*>(d_score_out.p) + so_docs); }
  uint32_t* sc_n() const { return reinterpret_cast<uint32_t*>(static_cast<char*>(d_score_out.p) + so_n); }
  uint32_t top_stride = 0, page_stride = 0;
  bool merged_shards = false;
  // bitmap group outputs
  DevBuf d_rbits, d_tile_cnt, d_tile_start, d_totals, d_take, d_out_off, d_reverse, d_out;
  // docid-page group (kModeDocPage): per-tile counts / rank offsets, and one packed result block
  // [counters n*8 u64 | totals n u64 | page docs n*stride u32] mirrored in pinned host memory
  Group page;
  DevBuf d_ptile_cnt, d_ptile_start, d_page_out, d_page_scratch;
  void* h_page_out = nullptr;
  size_t po_totals = 0, po_docs = 0, po_bytes = 0;
  uint32_t doc_page_stride = 0;
  // host results
  std::vector<mgx_query_result> h_results;
  std::vector<uint32_t> h_docs;
  std::vector<double> h_scores;
  bool executed = false;
  hipStream_t last_stream = nullptr;
  // kernel timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  uint64_t list_bytes = 0;
};

namespace mgx {

// MGX_TRACE_HOST: where a batch's host time goes (microseconds per section, printed by mgx_batch_reset every 64 batches)
static std::atomic<uint64_t> g_section_us[8];
static const char* const kSectionNames[8] = {"queries+leaves", "fast-path resolution", "table uploads", "wave tables",
                                              "items", "launch order", "item uploads", "rest of PrepareInto"};
struct SectionTimer {
  static bool On() {
    static const bool on = std::getenv("MGX_TRACE_HOST") != nullptr;
    return on;
  }
  std::chrono::steady_clock::time_point t;
  SectionTimer() { if (On()) t = std::chrono::steady_clock::now(); }
  void Mark(int section) {
    if (!On()) return;
    const auto now = std::chrono::steady_clock::now();
    g_section_us[section] += std::chrono::duration_cast<std::chrono::nanoseconds>(now - t).count();
    t = now;
  }
};

static int UploadGroup(mgx_batch* b, mgx_batch::Group& g, uint32_t mode, const std::vector<QuerySpec>& specs) {
  const bool score_mode = mode == kModeScore, df_mode = mode == kModeTextDf, page_mode = mode == kModeDocPage;
  const uint32_t n = static_cast<uint32_t>(g.qids.size());
  if (n == 0) return MGX_OK;
  SectionTimer sec;
  std::vector<DevQuery> dq(n);
  std::vector<DevLeaf> leaves;
  std::vector<uint32_t> prog;
  std::vector<DevScoreTerm> score;
  std::vector<uint32_t> expl;
  uint32_t max_leaves = 0, max_lds_leaves = 0, max_score = 0, max_stack = 0, max_instr = 0, max_cap = 64;
  for (uint32_t i = 0; i < n; ++i) {
    const QuerySpec& s = specs[g.qids[i]];
    DevQuery& q = dq[i];
    std::memset(&q, 0, sizeof(q));
    q.leaf_begin = static_cast<uint32_t>(leaves.size());
    q.n_leaves = static_cast<uint32_t>(s.leaves.size());
    q.prog_begin = static_cast<uint32_t>(prog.size());
    q.n_instr = static_cast<uint32_t>(s.prog.size());
    q.score_begin = static_cast<uint32_t>(score.size());
    q.n_score = static_cast<uint32_t>(s.score.size());
    q.mode = s.mode;
    q.limit = s.limit;
    q.offset = s.offset;
    // (offset + limit fits 32 bits only where it is bounded: score mode enforces needed <= kMaxNeeded; docid pages
    // carry limit alone — a caller's SearchAnd(terms, UINT32_MAX) must not wrap, nor spin a doubling loop forever)
    q.needed = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(s.offset) + s.limit, 0xFFFFFFFFull));
    uint32_t cap = 64;
    if (s.mode == kModeScore)
      while (cap < q.needed && cap < (1u << 30)) cap <<= 1;
    q.cap = cap;
    q.descending = s.reverse;
    q.stack_depth = s.stack_depth;
    q.out_slot = i;
    q.pat_off = s.pat_off;
    q.pat_len = s.pat_len;
    q.vt_begin = s.vt_begin;
    q.vt_count = static_cast<uint32_t>(s.verify_patterns.size());
    q.k1 = s.k1;
    q.b = s.b;
    q.one_minus_b = 1.0 - s.b;
    q.k1_plus_1 = s.k1 + 1.0;
    q.avgdl_clamped = std::max(s.avgdl, 1.0);  // bm25_scorer.cpp:81
    const uint32_t ebase = static_cast<uint32_t>(expl.size());
    // LDS residency (general kernel): sorted-list operands are scattered into an LDS bitmap, scored operands are
    // probed there by other threads; bitmap-form operands are read straight from HBM and take no LDS
    uint32_t n_lds = 0;
    for (size_t li = 0; li < s.leaves.size(); ++li) {
      DevLeaf lf = s.leaves[li];
      if (lf.kind == kLeafExplicit) lf.a += ebase;
      bool resident = lf.kind == kLeafList || lf.kind == kLeafExplicit;
      for (const DevScoreTerm& st : s.score) resident = resident || st.leaf == li;
      lf.lds = resident ? n_lds++ : kNoRow;
      leaves.push_back(lf);
    }
    if (n_lds > kMaxLdsLeaves)
      return Fail(MGX_ERR_NOT_IMPLEMENTED,
                  "query " + std::to_string(g.qids[i]) + ": more than 64 sorted-list / scored operands in one query");
    max_lds_leaves = std::max(max_lds_leaves, n_lds);
    prog.insert(prog.end(), s.prog.begin(), s.prog.end());
    score.insert(score.end(), s.score.begin(), s.score.end());
    expl.insert(expl.end(), s.explicit_ids.begin(), s.explicit_ids.end());
    while (expl.size() % 4) expl.push_back(0xFFFFFFFFu);
    max_leaves = std::max(max_leaves, q.n_leaves);
    max_score = std::max(max_score, q.n_score);
    max_stack = std::max(max_stack, q.stack_depth);
    max_instr = std::max(max_instr, q.n_instr);
    max_cap = std::max(max_cap, q.cap);
    if (!df_mode) b->list_bytes += 4 * s.list_postings;
  }
  g.plan = PlanLds(max_leaves, max_lds_leaves, max_score, max_stack, max_instr, max_cap, score_mode || df_mode);
  sec.Mark(0);
  std::vector<uint8_t> on_wave(n, 0);  // 0 general kernel, 1 wave kernel, 2 wave kernel with list operands, 3 fast path
  std::vector<DevFastQuery> fastq;
  std::vector<uint64_t> wave_tables;
  if (score_mode) {
    // ---- fast path (bitmap_score_kernel): flat programs over bitmap-form operands, 1..5 scored terms that are dense
    // grams, a page of at most 128 entries, one (k1, b, avgdl) for the whole set (they are table constants)
    const bool allow_fast = std::getenv("MGX_FORCE_BLOCK_KERNEL") == nullptr &&
                            !(std::getenv("MGX_FAST_PATH") && atoi(std::getenv("MGX_FAST_PATH")) == 0);
    mgx_index* idx = b->idx;
    uint32_t fcap = 64;
    bool have_params = false;
    double pk1 = 0, pb = 0, pavg = 0;
    for (uint32_t i = 0; allow_fast && i < n; ++i) {
      const QuerySpec& s = specs[g.qids[i]];
      if (!s.flat || !s.fast_score_ok || s.score.empty() || s.score.size() > static_cast<size_t>(kFastMaxScore)) continue;
      if (dq[i].cap > 128) continue;
      if (have_params && (Bits(s.k1) != Bits(pk1) || Bits(s.b) != Bits(pb) || Bits(s.avgdl) != Bits(pavg))) continue;
      DevFastQuery f{};
      bool ok = true;
      for (uint32_t ins : s.prog) {
        const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
        if (op == kOpCount) {
          if (f.n_ops == 0) { ok = false; break; }
          f.ops[f.n_ops - 1].code |= arg << 8;
          continue;
        }
        const DevLeaf& lf = s.leaves[arg];
        if (f.n_ops == static_cast<uint32_t>(kFastMaxOps) || (op == kOpLoad) != (f.n_ops == 0)) { ok = false; break; }
        FastOp o{};
        uint64_t stride = 0;
        if (lf.kind == kLeafGramBitmap) {
          o.base = reinterpret_cast<uint64_t>(idx->dev.gram_bitmaps + static_cast<uint64_t>(lf.b) * idx->dev.gb_row_stride);
          stride = idx->dev.gb_tile_stride * 8;
        } else if (lf.kind == kLeafFilterBitmap) {  // rows are appended at run time and may move: relative to the base
          o.base = static_cast<uint64_t>(lf.b) * idx->dev.fb_row_stride * 8;
          stride = idx->dev.fb_tile_stride * 8;
          o.code |= 16u;
        } else {
          ok = false;
          break;
        }
        if (stride > 0xFFFFFFFFull) { ok = false; break; }
        o.tile_stride = static_cast<uint32_t>(stride);
        o.code |= (op == kOpLoad || op == kOpOr) ? kFastOr : op == kOpAnd ? kFastAnd : kFastAndNot;
        f.ops[f.n_ops++] = o;
      }
      if (!ok || f.n_ops == 0) continue;
      for (size_t t = 0; ok && t < s.score.size(); ++t) {
        const DevLeaf& lf = s.leaves[s.score[t].leaf];
        if (lf.kind != kLeafGramBitmap) { ok = false; break; }
        FastScore& fs = f.score[t];
        fs.nib = reinterpret_cast<uint64_t>(idx->dev.tfnib + static_cast<uint64_t>(lf.b) * idx->dev.nib_row_stride);
        fs.idf = s.score[t].idf;
        fs.gram = lf.a;
        fs.skip_row = lf.row;
      }
      if (!ok) continue;
      if (s.reverse != 0) {  // SORT _score DESC: top-k pruning by block-max bounds
        double step = 0.0;
        f.blockmax = GetBlockMax(idx, s.k1, s.b, s.avgdl, &step, &f.blockmax_fine);
        if (f.blockmax) {
          f.bm_tile_stride = idx->n_bitmap_rows * 256u;
          f.bmf_tile_stride = idx->n_fine_rows * 1024u;
          // integer form of the bound: unit = step * (largest idf / 255); W_i = ceil(idf_i / (largest idf / 255))
          double idf_max = 0.0;
          for (size_t t = 0; t < s.score.size(); ++t) idf_max = std::max(idf_max, s.score[t].idf);
          const double wunit = idf_max > 0.0 ? idf_max * (1.0 + 0x1p-40) / 255.0 : 1.0;
          f.bm_inv_unit = 1.0 / (wunit * step);
          f.bm_wpack = f.bm_w4 = f.bm_cint = 0;
          for (size_t t = 0; t < s.score.size(); ++t) {
            const uint32_t w = static_cast<uint32_t>(std::min(255.0, std::ceil(s.score[t].idf / wunit)));
            if (t < 4) f.bm_wpack |= w << (8 * t); else f.bm_w4 = w;
          }
          for (size_t t = 0; t < s.score.size(); ++t) {
            const DevLeaf& lf = s.leaves[s.score[t].leaf];
            FastScore& fs = f.score[t];
            const uint32_t frow = f.blockmax_fine ? idx->h_fine_map[lf.b] : kNoRow;
            fs.bm_mode = frow != kNoRow ? 2u : 1u;
            fs.bm_off = frow != kNoRow ? frow * 1024u : lf.b * 256u;
          }
        }
      }
      if (!have_params) {
        pk1 = s.k1;
        pb = s.b;
        pavg = s.avgdl;
        have_params = true;
      }
      f.n_score = static_cast<uint32_t>(s.score.size());
      f.needed = dq[i].needed;
      f.cap = dq[i].cap;
      f.descending = dq[i].descending;
      f.k1 = dq[i].k1;
      f.b = dq[i].b;
      f.one_minus_b = dq[i].one_minus_b;
      f.k1_plus_1 = dq[i].k1_plus_1;
      f.avgdl_clamped = dq[i].avgdl_clamped;
      if (fastq.empty()) fastq.resize(n);
      fastq[i] = f;
      on_wave[i] = 3;
      fcap = std::max(fcap, f.cap);
    }
    if (have_params) {
      const double* ktab = GetLengthNormTable(idx, pk1, pb, pavg);
      g.fplan = PlanFast(fcap, ktab);
      if (ktab == nullptr || g.fplan.bytes > 64 * 1024)
        for (auto& w : on_wave) w = 0;
    }
    const bool allow = std::getenv("MGX_FORCE_BLOCK_KERNEL") == nullptr;
    uint32_t wl = 0, wsc = 0, wi = 0, wc = 64;
    bool has_list = false;
    for (uint32_t i = 0; i < n; ++i) {
      if (on_wave[i] == 3) continue;
      const QuerySpec& s = specs[g.qids[i]];
      bool ok = allow && s.wave_ok;
      // scored terms: dense grams (tf nibbles by doc slot) or sparse ones (exact posting lookup per match)
      for (const DevScoreTerm& st : s.score)
        ok = ok && (s.leaves[st.leaf].kind == kLeafGramBitmap || s.leaves[st.leaf].kind == kLeafList);
      if (!ok) continue;
      // A sorted-list operand needs the per-wave scatter scratch (16 KB per workgroup: 2 instead of 3 workgroups per
      // CU), so queries with one are launched separately and the all-bitmap majority keeps the small LDS plan.
      // the scored terms' contribution tables come from the index's pool (a sparse gram has no bitmap row: its table is
      // keyed by the gram id above the bitmap rows); without one the query runs on the general kernel
      uint64_t tabs[kWaveScoreSlots] = {0, 0, 0};
      for (size_t t = 0; ok && t < s.score.size(); ++t) {
        const DevLeaf& lf = s.leaves[s.score[t].leaf];
        const uint32_t key = lf.kind == kLeafGramBitmap ? lf.b : 0x80000000u | lf.a;
        tabs[t] = GetContributionTable(idx, key, s.score[t].idf, s.k1, s.b, s.avgdl);
        ok = tabs[t] != 0;
      }
      if (!ok) continue;
      if (wave_tables.empty()) wave_tables.assign(static_cast<size_t>(n) * kWaveScoreSlots, 0);
      for (int t = 0; t < kWaveScoreSlots; ++t) wave_tables[static_cast<size_t>(i) * kWaveScoreSlots + t] = tabs[t];
      bool lists = false;
      for (const DevLeaf& lf : s.leaves) lists = lists || lf.kind == kLeafList || lf.kind == kLeafExplicit;
      on_wave[i] = lists ? 2 : 1;
      has_list = has_list || lists;
      wl = std::max<uint32_t>(wl, dq[i].n_leaves);
      wsc = std::max<uint32_t>(wsc, dq[i].n_score);
      wi = std::max<uint32_t>(wi, dq[i].n_instr);
      wc = std::max<uint32_t>(wc, dq[i].cap);
    }
    g.wplan = PlanWave(wl, wsc, wi, wc, b->idx->dev.max_doc_len, false);
    g.wplan_lists = PlanWave(wl, wsc, wi, wc, b->idx->dev.max_doc_len, true);
    (void)has_list;
    if (g.wplan_lists.bytes > 160 * 1024)
      for (auto& w : on_wave)
        if (w != 3) w = 0;
  }
  if (page_mode || df_mode) {
    // flat programs count on the wave kernel (registers only, two tiles in flight per wave); the df pass of a
    // text-level term (always a flat AND of its grams) enumerates and scans its candidates there too
    const bool allow = std::getenv("MGX_FORCE_BLOCK_KERNEL") == nullptr;
    uint32_t wl = 1, wi = 1;
    bool has_list = false;
    for (uint32_t i = 0; i < n; ++i) {
      const QuerySpec& s = specs[g.qids[i]];
      if (!(allow && s.wave_ok)) continue;
      on_wave[i] = 1;
      for (const DevLeaf& lf : s.leaves) has_list = has_list || lf.kind == kLeafList || lf.kind == kLeafExplicit;
      wl = std::max<uint32_t>(wl, dq[i].n_leaves);
      wi = std::max<uint32_t>(wi, dq[i].n_instr);
    }
    g.wplan = WavePlan{wl, 0, wi, 0, 0, has_list ? 1u : 0u, 0};
  }
  // ---- selective queries: candidate-driven (cand_kernel). The smallest positive operand drives when it is a sorted
  // posting array (a gram too sparse for a bitmap row) that the program loads before its first COUNT — terms arrive
  // smallest-first (search_pipeline.cpp:2012-2014), so that is a gram of the first term and every funnel counter is a
  // count of candidates. Everything else of the query is probed per candidate.
  std::vector<uint32_t> cand_driver;
  g.cand_leaves = g.cand_instr = g.cand_cap = 0;
  g.merge_leaves = g.merge_instr = g.merge_ops = g.merge_cap = 0;
  if (score_mode || page_mode) {
    const bool allow_cand = std::getenv("MGX_FORCE_BLOCK_KERNEL") == nullptr &&
                                   !(std::getenv("MGX_CAND") && atoi(std::getenv("MGX_CAND")) == 0);
    static const uint64_t kMaxPerTile = std::getenv("MGX_CAND_MAX_PER_TILE") ? static_cast<uint64_t>(atoll(std::getenv("MGX_CAND_MAX_PER_TILE"))) : 64ull;
    const mgx_index* idx = b->idx;
    for (uint32_t i = 0; allow_cand && i < n; ++i) {
      const QuerySpec& s = specs[g.qids[i]];
      if (on_wave[i] == 3 || !s.flat || s.prog.empty()) continue;
      if (score_mode && (!s.fast_score_ok || s.score.empty())) continue;
      bool ok = true, counted = false;
      uint32_t driver = kNoLeaf;
      uint64_t best = ~0ull, longest_list = 0;
      for (size_t pc = 0; ok && pc < s.prog.size(); ++pc) {
        const uint32_t op = s.prog[pc] >> 24, arg = s.prog[pc] & 0xFFFFFFu;
        if (op == kOpCount) {
          counted = true;
          continue;
        }
        if ((op == kOpLoad) != (pc == 0) || (op != kOpLoad && op != kOpAnd && op != kOpAndNot)) { ok = false; break; }
        const DevLeaf& lf = s.leaves[arg];
        if (lf.kind != kLeafList && lf.kind != kLeafGramBitmap && lf.kind != kLeafFilterBitmap) { ok = false; break; }
        if (lf.kind == kLeafList) longest_list = std::max<uint64_t>(longest_list, idx->h_offsets[lf.a + 1] - idx->h_offsets[lf.a]);
        // any positive gram loaded before the first COUNT makes every counter a count of candidates; the smallest of them
        // drives (with terms in the reference's order that is the smallest gram of the whole query)
        if (op == kOpAndNot || lf.kind == kLeafFilterBitmap || counted) continue;
        const uint64_t sz = idx->h_offsets[lf.a + 1] - idx->h_offsets[lf.a];
        if (sz < best) {
          best = sz;
          driver = arg;
        }
      }
      if (!ok || driver == kNoLeaf || s.leaves[driver].kind != kLeafList) continue;
      for (uint32_t ins : s.prog)  // NOT of the driver's own gram: the tile program handles it (the result is empty)
        if ((ins >> 24) == kOpAndNot && (ins & 0xFFFFFFu) == driver) ok = false;
      if (score_mode)
        for (const DevScoreTerm& st : s.score)
          ok = ok && st.leaf != kNoLeaf && (s.leaves[st.leaf].kind == kLeafList || s.leaves[st.leaf].kind == kLeafGramBitmap);
      if (!ok) continue;
      // probing a LONG sorted list costs a binary search of its tile segment per candidate (14 dependent loads on a segment
      // of 16384): such operands are staged per tile instead (the merge kernel), whatever the driver's size
      const bool long_probe = longest_list > 8 * kMaxPerTile * b->idx->dev.n_tiles;
      if (best > kMaxPerTile * b->idx->dev.n_tiles || long_probe) {
        // a LONG driver: too many candidates to probe one by one — the tile-synchronous merge (SORT _score only; docid
        // pages of such queries stay on the counting kernels, which have no per-match work)
        const bool allow_merge = !(std::getenv("MGX_MERGE") && atoi(std::getenv("MGX_MERGE")) == 0);
        uint32_t staged = 0;
        for (size_t li = 0; li < s.leaves.size(); ++li) staged += li != driver ? 1u : 0u;
        if (!score_mode || !allow_merge || staged > kMergeMaxOps || dq[i].cap > 256) continue;
        dq[i].pat_off = driver;  // (unused by score-mode queries otherwise: where the kernel finds its driver)
        on_wave[i] = 5;
        g.merge_leaves = std::max(g.merge_leaves, dq[i].n_leaves);
        g.merge_instr = std::max(g.merge_instr, dq[i].n_instr);
        g.merge_ops = std::max(g.merge_ops, staged);
        g.merge_cap = std::max(g.merge_cap, dq[i].cap);
        continue;
      }
      if (cand_driver.empty()) cand_driver.assign(n, kNoLeaf);
      cand_driver[i] = driver;
      on_wave[i] = 4;
      g.cand_leaves = std::max(g.cand_leaves, dq[i].n_leaves);
      g.cand_instr = std::max(g.cand_instr, dq[i].n_instr);
      g.cand_cap = std::max(g.cand_cap, dq[i].cap);
    }
    if (CandLdsBytes(g.cand_leaves, g.cand_instr, score_mode ? g.cand_cap : 0) > 150 * 1024) {
      for (auto& w : on_wave)
        if (w == 4) w = 0;  // (a page of a thousand entries with many operands: the general kernel takes these)
      cand_driver.clear();
    }
    if (MergeLdsBytes(g.merge_leaves, g.merge_instr, g.merge_ops, g.merge_cap) > 150 * 1024)
      for (auto& w : on_wave)
        if (w == 5) w = 0;
  }
  if (g.plan.bytes > 160 * 1024)
    return Fail(MGX_ERR_NOT_IMPLEMENTED, "query shape exceeds the 160 KiB LDS of a CU");
  sec.Mark(1);
  MGX_HIP(Upload(g.d_queries, dq.data(), dq.size()));
  MGX_HIP(Upload(g.d_leaves, leaves.data(), leaves.size()));
  MGX_HIP(Upload(g.d_prog, prog.data(), prog.size()));
  MGX_HIP(Upload(g.d_score, score.data(), score.size()));
  MGX_HIP(Upload(g.d_explicit, expl.data(), expl.size(), 4));
  sec.Mark(2);
  // [n][8] counters followed by [n] pruning bounds: one memset clears both before every execute
  if (score_mode) {
    uint32_t max_limit = 1;
    for (const DevQuery& q : dq) max_limit = std::max(max_limit, q.limit);
    b->so_override = static_cast<size_t>(n) * 9 * 8;
    b->so_scores = b->so_override + static_cast<size_t>(n) * 8;
    b->so_docs = b->so_scores + static_cast<size_t>(n) * max_limit * 8;
    b->so_n = b->so_docs + static_cast<size_t>(n) * max_limit * 4;
    b->so_bytes = b->so_n + static_cast<size_t>(n) * 4;
    MGX_HIP(b->d_score_out.Alloc(b->so_bytes));
    MGX_HIP(b->res->Pinned(0, b->so_bytes, &b->h_score_out));
  } else if (page_mode) {
    uint32_t max_limit = 1;
    for (const DevQuery& q : dq) max_limit = std::max(max_limit, q.limit);
    b->doc_page_stride = max_limit;
    b->po_totals = static_cast<size_t>(n) * 8 * 8;
    b->po_docs = b->po_totals + static_cast<size_t>(n) * 8;
    b->po_bytes = b->po_docs + static_cast<size_t>(n) * max_limit * 4;
    MGX_HIP(b->d_page_out.Alloc(b->po_bytes));
    MGX_HIP(b->res->Pinned(1, b->po_bytes, &b->h_page_out));
    const size_t tiles = b->idx->dev.n_tiles;
    MGX_HIP(b->d_ptile_cnt.Alloc(static_cast<size_t>(n) * tiles * 4));
    MGX_HIP(b->d_ptile_start.Alloc(static_cast<size_t>(n) * tiles * 8));
  } else {
    MGX_HIP(g.d_counters.Alloc(static_cast<size_t>(n) * 9 * sizeof(unsigned long long)));
  }
  g.h_counters.assign(static_cast<size_t>(n) * 8, 0);
  std::vector<uint32_t> ident(n);
  for (uint32_t i = 0; i < n; ++i) ident[i] = i;
  MGX_HIP(Upload(g.d_ident, ident.data(), n));
  sec.Mark(3);
  // ---- work items: cut every query into runs of tiles of about equal estimated cost ------------------------------
  // cost of one tile ~ 1 (operand fetch, program) + matches/128 (enumeration + scoring); a workgroup gets ~48 units.
  std::vector<DevItem> items;
  std::vector<uint32_t> list_begin(n + 1, 0);
  uint32_t n_lists = 0;
  std::vector<uint32_t> item_begin(n + 1, 0);  // a query's own items (none for the members of a group)
  {
    const uint32_t n_tiles = b->idx->dev.n_tiles;
    for (uint32_t i = 0; i < n; ++i) {
      const QuerySpec& s = specs[g.qids[i]];
      item_begin[i] = static_cast<uint32_t>(items.size());
      static const double kMatchesPerUnit = std::getenv("MGX_ITEM_MATCHES") ? atof(std::getenv("MGX_ITEM_MATCHES")) : 256.0;
      static const double kItemCost = std::getenv("MGX_ITEM_COST") ? atof(std::getenv("MGX_ITEM_COST")) : 192.0;
      // (a text scan per candidate costs about what scoring a match does)
      static const uint32_t kMaxTiles = std::getenv("MGX_MAX_ITEM_TILES") ? static_cast<uint32_t>(atoi(std::getenv("MGX_MAX_ITEM_TILES"))) : static_cast<uint32_t>(kMaxTilesPerItem);
      // The fast path prunes most matches once a query has a k-th best score, and a longer item keeps that bound in the
      // workgroup: 4x the tiles per item (measured on the benchmark batch: 1.55 -> 1.49 ms; 8x loses to the tail).
      const bool fast = on_wave[i] == 3;
      const double per_tile = 1.0 + (score_mode || df_mode ? s.est_density * kTileDocs / (fast ? kMatchesPerUnit / 2 : kMatchesPerUnit) : 0.0);
      // (a small shard wants shorter items than a whole table: 2x..4x the base cost as the tile count goes 77 -> 611,
      // measured at 1.25M docs — 0.47 -> 0.39 ms — and at 10M)
      const double fast_scale = std::min(4.0, std::max(2.0, static_cast<double>(n_tiles) / 150.0));
      uint32_t tiles = static_cast<uint32_t>((fast ? fast_scale * kItemCost : kItemCost) / per_tile);
      tiles = std::max<uint32_t>(8, std::min<uint32_t>(tiles, fast ? 2 * kMaxTiles : kMaxTiles)) & ~7u;  // whole rounds of the waves of a workgroup
      // a small shard (few tiles per query) still wants a few workgroups per CU slot, or the launch is one ragged round
      if (score_mode) {
        static const uint64_t want_items = std::getenv("MGX_WANT_ITEMS") ? static_cast<uint64_t>(atoll(std::getenv("MGX_WANT_ITEMS"))) : 4ull * 768ull;  // ~4 rounds of the 768 workgroups a launch keeps resident
        const uint64_t cap = std::max<uint64_t>(8, (static_cast<uint64_t>(n_tiles) * n / want_items + 7) & ~7ull);
        tiles = static_cast<uint32_t>(std::min<uint64_t>(tiles, cap));
      }
      // the workgroup kernel walks its tiles one after the other (~4 us each): short items keep the few queries it
      // serves from becoming the tail of the step
      if (score_mode && !on_wave[i]) tiles = 8;
      list_begin[i] = n_lists;
      if (on_wave[i] == 4) {  // candidate-driven: one workgroup holds the whole query (tile_begin = the driver's leaf)
        items.push_back(DevItem{i, cand_driver[i], 0, n_lists++});
        continue;
      }
      uint32_t t0 = 0;
      // Fast path: a short "seed" item first (two tiles per wave), launched ahead of everything else (below): it scores
      // its tiles unpruned and publishes the query's first k-th best score, so the long items start pruning at once.
      static const uint32_t kSeedTiles = std::getenv("MGX_SEED_TILES") ? static_cast<uint32_t>(atoi(std::getenv("MGX_SEED_TILES"))) : 16u;
      const uint32_t seed_tiles = std::min<uint32_t>(kSeedTiles, (n_tiles / 8) & ~7u);  // (8 on a 77-tile shard)
      if (fast && seed_tiles && n_tiles > 4 * seed_tiles) {
        items.push_back(DevItem{i, 0, seed_tiles, n_lists++});
        t0 = seed_tiles;
      }
      for (uint32_t t = t0; t < n_tiles; t += tiles) {
        DevItem it{i, t, std::min(tiles, n_tiles - t), n_lists++};  // candidate lists stay grouped by query
        items.push_back(it);
      }
    }
    list_begin[n] = n_lists;
    item_begin[n] = static_cast<uint32_t>(items.size());
    sec.Mark(4);
    // Launch order: by doc band (kSortTiles tiles), so concurrent workgroups share operand tiles in L2/MALL; inside a
    // band by the query's largest gram (its bitmap and tf column are the lines the band re-reads most), dealt so that
    // workgroup j — which runs on XCD j % 8 under round-robin dispatch, every XCD with its own L2 — gets a contiguous
    // eighth of that order. One bucket pass (counts, offsets, scatter): a fresh batch is scheduled per step, and two
    // stable sorts of ~12,000 items were a third of the host's compile time.
    static const uint32_t kSortTiles = std::getenv("MGX_SORT_TILES") ? static_cast<uint32_t>(atoi(std::getenv("MGX_SORT_TILES"))) : 16u;
    static const bool kXcdAffinity = std::getenv("MGX_XCD_AFFINITY") ? atoi(std::getenv("MGX_XCD_AFFINITY")) != 0 : true;
    std::vector<uint32_t> order(n);  // queries in the order they appear inside a band
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    if (kXcdAffinity && score_mode) {
      std::vector<uint32_t> heavy(n, 0);
      for (uint32_t i = 0; i < n; ++i) {
        uint64_t best = 0;
        for (const DevLeaf& lf : specs[g.qids[i]].leaves) {
          if (lf.kind != kLeafGramBitmap && lf.kind != kLeafList) continue;
          const uint64_t sz = b->idx->h_offsets[lf.a + 1] - b->idx->h_offsets[lf.a];
          if (sz > best) {
            best = sz;
            heavy[i] = lf.a;
          }
        }
      }
      std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return heavy[x] < heavy[y]; });
    }
    const uint32_t n_bands = n_tiles / kSortTiles + 1;
    // A heavy query's items (thousands of matches per tile: tens of microseconds per tile visit, where most take one or
    // two) are pulled forward in the launch order — its last bands would otherwise start when the launch is nearly over
    // and finish alone, the tail of the step. They stay in doc order, so the query's bound still grows from band to band.
    static const double kHeavyDensity = std::getenv("MGX_HEAVY_DENSITY") ? atof(std::getenv("MGX_HEAVY_DENSITY")) : 0.05;
    // (measured: 0.346 -> 0.327 ms on a 77-tile shard at 0.6; on the 611-tile table the tail is a smaller share of the
    // step and a later start of the query's last bands — a better bound — is worth as much: neutral, left in order)
    static const double kSqueezeEnv = std::getenv("MGX_HEAVY_SQUEEZE") ? atof(std::getenv("MGX_HEAVY_SQUEEZE")) : -1.0;
    const double kHeavySqueeze = kSqueezeEnv >= 0.0 ? kSqueezeEnv : 0.6 + 0.4 * std::min(1.0, n_tiles / 600.0);
    auto band_of = [&](const DevItem& it) {
      if (on_wave[it.query] == 4) return 0u;  // (a candidate-driven item covers the whole query; tile_begin holds its driver)
      const uint32_t band = it.tile_begin / kSortTiles;
      if (score_mode && on_wave[it.query] == 3 && specs[g.qids[it.query]].est_density >= kHeavyDensity)
        return static_cast<uint32_t>(band * kHeavySqueeze);
      return band;
    };
    std::vector<uint32_t> band_at(n_bands + 1, 0);
    for (const DevItem& it : items) band_at[band_of(it) + 1]++;
    for (uint32_t k = 0; k < n_bands; ++k) band_at[k + 1] += band_at[k];
    std::vector<DevItem> sorted(items.size());
    {
      std::vector<uint32_t> cur(band_at.begin(), band_at.end() - 1);
      for (uint32_t oi = 0; oi < n; ++oi) {
        const uint32_t i = order[oi];
        for (uint32_t k = item_begin[i]; k < item_begin[i + 1]; ++k) sorted[cur[band_of(items[k])]++] = items[k];
      }
    }
    if (kXcdAffinity && score_mode) {
      for (uint32_t k = 0; k < n_bands; ++k) {
        const size_t lo = band_at[k], m = band_at[k + 1] - lo, per = (m + 7) / 8;
        size_t w = lo;
        for (size_t r = 0; r < per; ++r)
          for (size_t x = 0; x < 8; ++x)
            if (x * per + r < m) items[w++] = sorted[lo + x * per + r];
      }
    } else {
      items.swap(sorted);
    }
  }
  const uint32_t n_lists_all = n_lists;
  std::vector<DevItem> items_wave, items_block, items_wave_lists, items_fast[kFastMaxScore], items_cand, items_merge;
  std::vector<uint8_t> has_seed(n, 0);
  uint32_t seed_k = 0;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0: the fast path's seed items (a query's first list), then the rest
    for (const DevItem& it : items) {
      const uint8_t w = on_wave[it.query];
      const bool seed = w == 3 && it.list == list_begin[it.query] && it.tile_begin == 0 &&
                        list_begin[it.query + 1] - list_begin[it.query] > 1 && it.n_tiles <= 16;
      if (seed != (pass == 0)) continue;
      if (w == 3) items_fast[fastq[it.query].n_score - 1].push_back(it);
      else if (w == 4) items_cand.push_back(it);
      else if (w == 5) items_merge.push_back(it);
      else (w == 2 && score_mode ? items_wave_lists : w ? items_wave : items_block).push_back(it);
      if (seed && fastq[it.query].blockmax != 0) has_seed[it.query] = 1;  // (a query that prunes: its seed's keys are worth sharing)
    }
    if (pass == 0)
      for (int t = 0; t < kFastMaxScore; ++t) g.n_seed_fast[t] = static_cast<uint32_t>(items_fast[t].size());
  }
  // The shape of the seed-key exchange of a sharded table must be the same on every rank, whatever this shard's own
  // index looks like (a gram may be a bitmap here and a list there, a shard a tile shorter): it is made from the QUERIES
  // alone — keys per query = the largest page of the group's SORT _score DESC queries that fit the fast path's lists —
  // and a query without a seed on this rank contributes zeros.
  g.seed_table_docs = 0;
  if (score_mode) {
    for (uint32_t i = 0; i < n; ++i) {
      const QuerySpec& sp = specs[g.qids[i]];
      if (sp.reverse != 0 && dq[i].needed != 0 && dq[i].needed <= 128u) {
        seed_k = std::max(seed_k, dq[i].needed);
        g.seed_table_docs = std::max<uint64_t>(g.seed_table_docs, sp.total_docs);
      }
    }
  }
  g.seed_k = seed_k;
  if (seed_k) MGX_HIP(Upload(g.d_has_seed, has_seed.data(), has_seed.size()));
  sec.Mark(5);
  if (!fastq.empty()) MGX_HIP(Upload(g.d_fast_queries, fastq.data(), fastq.size()));
  for (int t = 0; t < kFastMaxScore; ++t) MGX_HIP(Upload(g.d_items_fast[t], items_fast[t].data(), items_fast[t].size()));
  MGX_HIP(Upload(g.d_items_wave_lists, items_wave_lists.data(), items_wave_lists.size()));
  MGX_HIP(Upload(g.d_items_cand, items_cand.data(), items_cand.size()));
  MGX_HIP(Upload(g.d_items_merge, items_merge.data(), items_merge.size()));
  if (page_mode && !items_cand.empty()) {
    std::vector<uint8_t> skip(n, 0);
    for (uint32_t i = 0; i < n; ++i) skip[i] = on_wave[i] == 4 ? 1 : 0;
    MGX_HIP(Upload(g.d_cand_skip, skip.data(), skip.size()));
  }
  {
    // block-kernel items: queries whose operands are all bitmap-form first (plain instantiation), then the queries with a
    // sorted list, an explicit id list or a slot range among their operands (the instantiation that skips / jumps over tiles)
    std::vector<uint8_t> listlike(n, 0);
    for (uint32_t i = 0; i < n; ++i)
      for (const DevLeaf& lf : specs[g.qids[i]].leaves)
        if (lf.kind == kLeafList || lf.kind == kLeafExplicit || lf.kind == kLeafRange) listlike[i] = 1;
    std::stable_partition(items_block.begin(), items_block.end(), [&](const DevItem& it) { return listlike[it.query] == 0; });
    g.n_items_plain = 0;
    for (const DevItem& it : items_block) g.n_items_plain += listlike[it.query] == 0 ? 1u : 0u;
  }
  g.n_items = static_cast<uint32_t>(items_block.size());
  g.n_items_wave = static_cast<uint32_t>(items_wave.size());
  MGX_HIP(Upload(g.d_items, items_block.data(), items_block.size()));
  MGX_HIP(Upload(g.d_items_wave, items_wave.data(), items_wave.size()));
  MGX_HIP(Upload(g.d_list_begin, list_begin.data(), list_begin.size()));
  sec.Mark(6);
  if (std::getenv("MGX_VERBOSE"))
    fprintf(stderr,
            "[mgx] %s group: %u queries; fast path %zu items (lds %u B, ring %u), wave kernel %u items (lds %u B), "
            "block kernel %u items (lds %u B), candidate-driven %zu queries, list merge %zu items\n",
            score_mode ? "score" : page_mode ? "docid-page" : df_mode ? "df" : "bitmap", n,
            items_fast[0].size() + items_fast[1].size() + items_fast[2].size() + items_fast[3].size() + items_fast[4].size(),
            g.fplan.bytes, g.fplan.ring, g.n_items_wave + static_cast<uint32_t>(items_wave_lists.size()), g.wplan.bytes, g.n_items,
            g.plan.bytes, items_cand.size(), items_merge.size());
  DevBatch& d = g.dev;
  d.items = g.d_items.as<DevItem>();
  d.n_items = g.n_items;
  d.queries = g.d_queries.as<DevQuery>();
  d.leaves = g.d_leaves.as<DevLeaf>();
  d.prog = g.d_prog.as<uint32_t>();
  d.score_terms = g.d_score.as<DevScoreTerm>();
  d.explicit_pool = g.d_explicit.as<uint32_t>();
  d.patterns = b->d_patterns.as<uint8_t>();
  d.text_terms = b->d_text_terms.as<DevTextTerm>();
  d.text_idf = b->d_text_idf.as<double>();
  d.verify_terms = b->d_verify_terms.as<DevTextTerm>();
  d.n_queries = n;
  d.counters = score_mode  ? b->sc_counters()
               : page_mode ? b->d_page_out.as<unsigned long long>()
                           : g.d_counters.as<unsigned long long>();
  if (page_mode) {
    d.tile_cnt = b->d_ptile_cnt.as<uint32_t>();
    d.tile_start = b->d_ptile_start.as<uint64_t>();
    d.totals = reinterpret_cast<const uint64_t*>(static_cast<const char*>(b->d_page_out.p) + b->po_totals);
    d.page_docs = reinterpret_cast<uint32_t*>(static_cast<char*>(b->d_page_out.p) + b->po_docs);
    d.page_stride = b->doc_page_stride;
  }
  d.bounds = score_mode ? d.counters + static_cast<size_t>(n) * 8 : nullptr;
#ifdef MGX_ABLATION
  d.debug_skip = std::getenv("MGX_DEBUG_SKIP") ? static_cast<uint32_t>(atoi(std::getenv("MGX_DEBUG_SKIP"))) : 0u;
#endif
  if (score_mode) {
    uint32_t max_needed = 1, max_limit = 1;
    for (const DevQuery& q : dq) {
      max_needed = std::max(max_needed, q.needed);
      max_limit = std::max(max_limit, q.limit);
    }
    const size_t n_lists_total = n_lists_all;
    d.cand_stride = max_needed;
    b->top_stride = max_needed;
    b->page_stride = max_limit;
    MGX_HIP(b->d_cand_keys.Alloc(n_lists_total * max_needed * 8));
    MGX_HIP(b->d_cand_docs.Alloc(n_lists_total * max_needed * 4));
    MGX_HIP(b->d_cand_n.Alloc(n_lists_total * 4));
    {
      const size_t elems = static_cast<size_t>(n) * max_needed + n;
      b->ex_off32 = elems * 8;
      b->ex_bytes = (elems * 12 + 7) / 8 * 8;
      MGX_HIP(b->d_export.Alloc(b->ex_bytes));
    }
    d.cand_keys = b->d_cand_keys.as<uint64_t>();
    d.cand_docs = b->d_cand_docs.as<uint32_t>();
    d.cand_n = b->d_cand_n.as<uint32_t>();
  } else if (!df_mode && !page_mode) {
    const size_t tiles = b->idx->dev.n_tiles;
    MGX_HIP(b->d_rbits.Alloc(static_cast<size_t>(n) * tiles * kWordsPerTile * 8));
    MGX_HIP(b->d_tile_cnt.Alloc(static_cast<size_t>(n) * tiles * 4));
    MGX_HIP(b->d_tile_start.Alloc(static_cast<size_t>(n) * tiles * 8));
    MGX_HIP(b->d_totals.Alloc(static_cast<size_t>(n) * 8));
    MGX_HIP(b->d_take.Alloc(static_cast<size_t>(n) * 8));
    MGX_HIP(b->d_out_off.Alloc(static_cast<size_t>(n) * 8));
    std::vector<uint32_t> rev(n);
    for (uint32_t i = 0; i < n; ++i) rev[i] = specs[g.qids[i]].deep_score ? 0u : specs[g.qids[i]].reverse;
    MGX_HIP(Upload(b->d_reverse, rev.data(), n));
    d.rbits = b->d_rbits.as<uint64_t>();
    d.tile_cnt = b->d_tile_cnt.as<uint32_t>();
  }
  if (score_mode && !wave_tables.empty()) {
    MGX_HIP(Upload(g.d_tables, wave_tables.data(), wave_tables.size()));
    d.wave_tables = g.d_tables.as<uint64_t>();
  }
  d.fast_queries = g.d_fast_queries.as<DevFastQuery>();
  d.dev_index = b->idx->d_dev_index.as<DevIndex>();
  for (int t = 0; t < kFastMaxScore; ++t) {
    g.dev_fast[t] = d;
    g.dev_fast[t].items = g.d_items_fast[t].as<DevItem>();
    g.dev_fast[t].n_items = static_cast<uint32_t>(items_fast[t].size());
  }
  g.dev_merge = d;
  g.dev_merge.items = g.d_items_merge.as<DevItem>();
  g.dev_merge.n_items = static_cast<uint32_t>(items_merge.size());
  g.dev_cand = d;
  g.dev_cand.items = g.d_items_cand.as<DevItem>();
  g.dev_cand.n_items = static_cast<uint32_t>(items_cand.size());
  g.dev_wave = d;
  g.dev_wave.items = g.d_items_wave.as<DevItem>();
  g.dev_wave.n_items = g.n_items_wave;
  g.dev_wave_lists = d;
  g.dev_wave_lists.items = g.d_items_wave_lists.as<DevItem>();
  g.dev_wave_lists.n_items = static_cast<uint32_t>(items_wave_lists.size());
  if (page_mode) {
    std::vector<DevItem> pw, pb;
    for (uint32_t i = 0; i < n; ++i)
      if (on_wave[i] != 4) (on_wave[i] ? pw : pb).push_back(DevItem{i, 0, 0, 0});  // (cand_kernel writes its own pages)
    MGX_HIP(Upload(g.d_pq_wave, pw.data(), pw.size()));
    MGX_HIP(Upload(g.d_pq_block, pb.data(), pb.size()));
    g.dev_page_wave = d;
    g.dev_page_wave.items = g.d_pq_wave.as<DevItem>();
    g.dev_page_wave.n_items = static_cast<uint32_t>(pw.size());
    g.dev_page_block = d;
    g.dev_page_block.items = g.d_pq_block.as<DevItem>();
    g.dev_page_block.n_items = static_cast<uint32_t>(pb.size());
  }
  return MGX_OK;
}

// CompileQuery over a whole batch. Queries are independent, so batches of a few hundred and more are cut into chunks for
// a small process-wide helper pool (MGX_COMPILE_THREADS helpers, default 3, beside the calling thread): on a doc-range
// shard the device finishes a batch in a fraction of a millisecond and the host's per-batch compile is what a rank's
// throughput hangs on. One caller at a time uses the pool; a second concurrent caller compiles inline.
struct CompilePool {
  std::mutex mu, gate;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> helpers;
  const mgx_index* idx = nullptr;
  const mgx_query* queries = nullptr;
  std::vector<QuerySpec>* specs = nullptr;
  uint32_t next = 0, end = 0, pending = 0;
  int rc = MGX_OK;
  std::string error;
  bool stop = false;
  static constexpr uint32_t kChunk = 64;

  void Run(std::unique_lock<std::mutex>& lock) {  // called with mu held; works until the queue is empty
    while (next < end) {
      const uint32_t a = next, b = std::min(end, a + kChunk);
      next = b;
      lock.unlock();
      int local = MGX_OK;
      std::string msg;
      for (uint32_t i = a; i < b && local == MGX_OK; ++i) {
        (*specs)[i].Clear();
        local = CompileQuery(idx, queries[i], &(*specs)[i]);
        if (local) msg = "query " + std::to_string(i) + ": " + g_last_error;
      }
      lock.lock();
      if (local && rc == MGX_OK) {
        rc = local;
        error = msg;
      }
      pending -= b - a;
    }
  }
  void Helper() {
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
      cv_work.wait(lock, [&] { return stop || next < end; });
      if (stop) return;
      Run(lock);
      if (pending == 0) cv_done.notify_all();
    }
  }
  CompilePool() {
    const int n = std::getenv("MGX_COMPILE_THREADS") ? atoi(std::getenv("MGX_COMPILE_THREADS")) : 3;
    for (int i = 0; i < n; ++i)
      helpers.emplace_back([this] {
        pthread_setname_np(pthread_self(), "mgx-compile");
        Helper();
      });
  }
  ~CompilePool() {
    {
      std::lock_guard<std::mutex> lock(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : helpers) t.join();
  }
};

static int CompileAll(const mgx_index* idx, const mgx_query* queries, uint32_t n, std::vector<QuerySpec>* specs) {
  // one pool per calling thread: an executor's dispatcher threads compile different batches at the same time, and a
  // shared pool handed its helpers to one of them while the other compiled its 1024 queries alone (4x the time)
  static thread_local CompilePool pool;
  std::unique_lock<std::mutex> gate(pool.gate, std::try_to_lock);
  if (n < 4 * CompilePool::kChunk || pool.helpers.empty() || !gate.owns_lock()) {
    for (uint32_t i = 0; i < n; ++i) {
      (*specs)[i].Clear();
      const int rc = CompileQuery(idx, queries[i], &(*specs)[i]);
      if (rc) {
        SetError("query " + std::to_string(i) + ": " + g_last_error);
        return rc;
      }
    }
    return MGX_OK;
  }
  std::unique_lock<std::mutex> lock(pool.mu);
  pool.idx = idx;
  pool.queries = queries;
  pool.specs = specs;
  pool.next = 0;
  pool.end = pool.pending = n;
  pool.rc = MGX_OK;
  pool.cv_work.notify_all();
  pool.Run(lock);  // the caller compiles too
  pool.cv_done.wait(lock, [&] { return pool.pending == 0; });
  pool.end = 0;
  if (pool.rc) SetError(pool.error);
  return pool.rc;
}

// Compiles `specs` into batch object `b` (fresh, or just emptied by ResetBatch): every device array comes from the
// batch's arenas, nothing is copied to the device here — the first execute ships the upload arena.
static int PrepareInto(mgx_batch* b, mgx_index* idx, std::vector<QuerySpec>&& specs) {
  ResourceScope scope(b->res);
  b->idx = idx;
  b->n_queries = static_cast<uint32_t>(specs.size());
  b->specs = std::move(specs);
  for (uint32_t i = 0; i < b->n_queries; ++i) {
    const uint32_t m = b->specs[i].mode;
    (m == kModeScore ? b->score : m == kModeDocPage ? b->page : b->bitmap).qids.push_back(i);
  }
  MGX_HIP(hipSetDevice(idx->device));
  // text-level scored terms: number them across the batch, pool their bytes, and give each one a df query
  {
    std::string pool;
    std::vector<DevTextTerm> tts;
    // the df of a term depends on the term (and on N), not on the query around it: one df query per distinct term
    std::map<std::pair<std::string, uint64_t>, uint32_t> seen;
    for (QuerySpec& s : b->specs) {
      for (const QuerySpec::TextTerm& tt : s.text_terms) {
        const auto key = std::make_pair(tt.pattern, s.total_docs);
        const auto hit = seen.find(key);
        if (hit != seen.end()) {
          s.score[tt.score_index].text_term = hit->second;
          continue;
        }
        const uint32_t id = static_cast<uint32_t>(tts.size());
        seen.emplace(key, id);
        DevTextTerm d{static_cast<uint32_t>(pool.size()), static_cast<uint32_t>(tt.pattern.size())};
        pool += tt.pattern;
        tts.push_back(d);
        s.score[tt.score_index].text_term = id;
        b->text_total_docs.push_back(s.total_docs);
        QuerySpec ds;
        ds.mode = kModeTextDf;
        ds.pat_off = d.pat_off;
        ds.pat_len = d.pat_len;
        Compiler c{idx, &ds, 0};
        double dens = 1.0;
        uint64_t mn = ~0ull;
        std::string ckey = tt.pattern;
        ckey.push_back('\0');
        ckey += std::to_string(s.total_docs);
        uint64_t known = ~0ull;
        {
          std::lock_guard<std::mutex> lock(idx->table_mu);
          const auto hit = idx->df_cache.find(ckey);
          if (hit != idx->df_cache.end()) known = hit->second;
        }
        b->text_df_known.push_back(tt.grams.empty() ? 0ull : known);
        b->text_df_key.push_back(std::move(ckey));
        if (tt.grams.empty() || known != ~0ull) {
          // nothing to count: a term shorter than one n-gram has df 0 (PopulateTermDocumentFrequency returns before
          // counting, search_pipeline.cpp:546-549), a term seen before has this shard's count in the index's cache — the
          // df query runs over the empty doc range and the known LOCAL value takes its place on the device
          // (gather_df_kernel), before any rank sums the counts
          c.Emit(kOpLoad, c.RangeLeaf(0, 0));
          mn = 0;
        } else {
          c.Emit(kOpLoad, c.GramLeaf(tt.grams[0]));
          for (size_t k = 1; k < tt.grams.size(); ++k) c.Emit(kOpAnd, c.GramLeaf(tt.grams[k]));
          const uint32_t live_row = idx->live_row.load(std::memory_order_relaxed);
          bool all_bitmaps = idx->bitmaps_clean.load(std::memory_order_relaxed);
          for (uint32_t gid : tt.grams) all_bitmaps = all_bitmaps && (gid == MGX_GRAM_ABSENT || idx->h_bm_row[gid] != kNoRow);
          if (live_row != kNoRow && !all_bitmaps) c.Emit(kOpAnd, c.FilterLeaf(live_row));  // (df counts live documents only)
        }
        for (uint32_t gid : tt.grams)
          mn = std::min<uint64_t>(mn, gid == MGX_GRAM_ABSENT ? 0 : idx->h_offsets[gid + 1] - idx->h_offsets[gid]);
        dens = static_cast<double>(mn) / static_cast<double>(std::max<uint32_t>(idx->dev.n_docs, 1));
        ds.est_density = dens;
        if (ds.leaves.size() > kMaxLeaves)
          return Fail(MGX_ERR_NOT_IMPLEMENTED, "more than 192 distinct n-grams in one text-level term");
        b->df_specs.push_back(std::move(ds));
      }
    }
    // exact-text filters: their patterns share the pool
    std::vector<DevTextTerm> vts;
    for (QuerySpec& s : b->specs) {
      s.vt_begin = static_cast<uint32_t>(vts.size());
      for (const std::string& pat : s.verify_patterns) {
        vts.push_back(DevTextTerm{static_cast<uint32_t>(pool.size()), static_cast<uint32_t>(pat.size())});
        pool += pat;
      }
    }
    if (!vts.empty()) {
      MGX_HIP(Upload(b->d_verify_terms, vts.data(), vts.size()));
      if (tts.empty()) MGX_HIP(Upload(b->d_patterns, reinterpret_cast<const uint8_t*>(pool.data()), pool.size(), 16));
    }
    const uint32_t n_tt = static_cast<uint32_t>(tts.size());
    if (n_tt != 0) {
      MGX_HIP(Upload(b->d_patterns, reinterpret_cast<const uint8_t*>(pool.data()), pool.size(), 16));
      MGX_HIP(Upload(b->d_text_terms, tts.data(), tts.size()));
      MGX_HIP(b->d_text_idf.Alloc(static_cast<size_t>(n_tt) * sizeof(double)));
      MGX_HIP(b->d_text_df.Alloc(static_cast<size_t>(n_tt) * sizeof(uint64_t)));
      MGX_HIP(b->d_text_df_local.Alloc(static_cast<size_t>(n_tt) * sizeof(uint64_t)));
      MGX_HIP(Upload(b->d_text_df_known, b->text_df_known.data(), b->text_df_known.size()));
      b->h_text_df.assign(n_tt, 0);
      b->h_text_df_local.assign(n_tt, 0);
      b->h_text_idf.assign(n_tt, 0.0);
      for (uint32_t i = 0; i < n_tt; ++i) b->textdf.qids.push_back(i);
    }
  }
  int rc = UploadGroup(b, b->score, kModeScore, b->specs);
  if (rc) return rc;
  rc = UploadGroup(b, b->bitmap, kModeBitmap, b->specs);
  if (rc) return rc;
  rc = UploadGroup(b, b->textdf, kModeTextDf, b->df_specs);
  if (rc) return rc;
  rc = UploadGroup(b, b->page, kModeDocPage, b->specs);
  if (rc) return rc;
  b->h_results.assign(b->n_queries, mgx_query_result{});
  return MGX_OK;
}

static int PrepareFromSpecs(mgx_index* idx, std::vector<QuerySpec>&& specs, mgx_batch** out) {
  auto b = std::make_unique<mgx_batch>();
  b->res_owned = std::make_unique<BatchResources>();
  b->res = b->res_owned.get();
  int rc = PrepareInto(b.get(), idx, std::move(specs));
  if (rc) return rc;
  *out = b.release();
  return MGX_OK;
}

// Empties a batch object for re-use; its resources (arena chunks, pinned result blocks, events) stay.
static void ResetBatch(mgx_batch* b) {
  std::unique_ptr<BatchResources> owned = std::move(b->res_owned);
  BatchResources* res = b->res;
  for (auto& ev : b->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  const bool timing = b->timing;
  b->~mgx_batch();
  new (b) mgx_batch();
  b->res_owned = std::move(owned);
  b->res = res;
  b->timing = timing;
  if (res) res->Reset();
}

// Result blocks of the score / docid-page groups -> pinned host memory, fenced by the batch's done event.
static int IssueResultCopy(mgx_batch* b, hipStream_t s) {
  if (!b->res) return MGX_OK;
  if (!b->page.qids.empty())
    MGX_HIP(hipMemcpyAsync(b->h_page_out, b->d_page_out.p, b->po_bytes, hipMemcpyDeviceToHost, s));
  if (!b->score.qids.empty())
    MGX_HIP(hipMemcpyAsync(b->h_score_out, b->d_score_out.p, b->so_bytes, hipMemcpyDeviceToHost, s));
  if (!b->res->done_ev) {
    MGX_HIP(hipEventCreateWithFlags(&b->res->done_ev, hipEventDisableTiming));
  }
  MGX_HIP(hipEventRecord(b->res->done_ev, s));
  b->res->copy_issued = true;
  return MGX_OK;
}

// hip_stream as given by the caller (NULL = the default stream, as everywhere in HIP).
static int BatchStream(mgx_batch* b, void* hip_stream, hipStream_t* out) {
  (void)b;
  *out = static_cast<hipStream_t>(hip_stream);
  return MGX_OK;
}

// The batch's input arrays travel on the stream of its first execute (or df pass): one copy per arena chunk.
static int EnsureUploaded(mgx_batch* b, hipStream_t s) {
  if (!b->res || b->res->uploaded) return MGX_OK;
  for (const Arena::Chunk& c : b->res->upload.chunks)
    if (c.used) MGX_HIP(hipMemcpyAsync(c.dev, c.host, c.used, hipMemcpyHostToDevice, s));
  b->res->uploaded = true;
  return MGX_OK;
}

// df pass of the text-level terms: counts land in d_text_df[term]
static int CountDfImpl(mgx_batch* b, hipStream_t s) {
  mgx_index* idx = b->idx;
  mgx_batch::Group& g = b->textdf;
  if (g.qids.empty()) return MGX_OK;
  MGX_HIP(hipSetDevice(idx->device));
  {
    int rc = EnsureUploaded(b, s);
    if (rc) return rc;
  }
  MGX_HIP(hipMemsetAsync(g.d_counters.p, 0, g.d_counters.bytes, s));
  MGX_LAUNCH(LaunchWaveCount(idx->dev, g.dev_wave, g.wplan, true, s));
  MGX_LAUNCH(LaunchTileEval(kModeTextDf, idx->dev, g.dev, g.plan, s, g.n_items_plain));
  // counter slot 5 of every df query (or the cached local count) -> the local array and the buffer ranks all-reduce
  MGX_LAUNCH(LaunchGatherDf(g.d_counters.as<unsigned long long>(), b->d_text_df_known.as<uint64_t>(),
                            static_cast<uint32_t>(g.qids.size()), b->d_text_df_local.as<uint64_t>(),
                            b->d_text_df.as<uint64_t>(), s));
  return MGX_OK;
}

// Sharded tables: between a score group's seed launch and its main launch, every rank's seed keys are all-gathered (by
// the caller's transport: RCCL in mgx_batch_execute_sharded, anything in mgx_batch_execute_gather) and each query's bound
// is raised to the needed-th best of their union.
struct SeedGather {
  int world = 1;
  mgx_gather_fn fn = nullptr;
  void* user = nullptr;
};
static int SeedBoundExchange(mgx_batch* b, mgx_batch::Group& g, const SeedGather& sg, hipStream_t s);

static int ExecuteImpl(mgx_batch* b, hipStream_t s, const SeedGather* sg = nullptr) {
  mgx_index* idx = b->idx;
  MGX_HIP(hipSetDevice(idx->device));
  {
    int rc = EnsureUploaded(b, s);
    if (rc) return rc;
  }
  b->merged_shards = false;
  {  // contribution tables built while this (or any earlier) batch was compiled travel on the index's table stream
    std::lock_guard<std::mutex> lock(idx->table_mu);
    MGX_HIP(FlushTableJobs(idx));
    if (idx->table_ev_recorded) MGX_HIP(hipStreamWaitEvent(s, idx->table_ev, 0));
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (b->timing) {
    MGX_HIP(hipEventCreate(&ev0));
    MGX_HIP(hipEventCreate(&ev1));
  }
  bool timed = false;
  if (!b->textdf.qids.empty()) {
    // text-level terms: df pass (unless the caller ran and reduced it), then idf on the host — log() of the host libm,
    // like the reference and the oracle, so that scores stay bit-identical — and back to the device
    if (!b->df_ready) {
      int rc = CountDfImpl(b, s);
      if (rc) return rc;
    }
    b->df_ready = false;
    const size_t n_tt = b->h_text_df.size();
    MGX_HIP(hipMemcpyAsync(b->h_text_df.data(), b->d_text_df.p, n_tt * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipMemcpyAsync(b->h_text_df_local.data(), b->d_text_df_local.p, n_tt * sizeof(uint64_t),
                           hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
    {
      // h_text_df is what idf is taken from (summed over the ranks when the caller reduced it); the cache learns this
      // shard's own counts only, and a cached count is never put in the place of a reduced one
      std::lock_guard<std::mutex> lock(idx->table_mu);
      for (size_t i = 0; i < n_tt; ++i)
        if (b->text_df_known[i] == ~0ull && idx->df_cache.size() < (1u << 20))
          idx->df_cache.emplace(b->text_df_key[i], b->h_text_df_local[i]);
    }
    for (size_t i = 0; i < n_tt; ++i) {
      // BM25Scorer::ComputeIDF, bm25_scorer.cpp:14-25
      const uint64_t total = b->text_total_docs[i];
      const uint64_t dfc = std::min<uint64_t>(b->h_text_df[i], total);
      const double nn = static_cast<double>(total), df = static_cast<double>(dfc);
      b->h_text_idf[i] = total == 0 ? 0.0 : std::log((nn - df + 0.5) / (df + 0.5) + 1.0);
    }
    MGX_HIP(hipMemcpyAsync(b->d_text_idf.p, b->h_text_idf.data(), n_tt * sizeof(double), hipMemcpyHostToDevice, s));
  }
  if (!b->score.qids.empty()) {
    mgx_batch::Group& g = b->score;
    size_t clear_bytes = b->so_override;
#ifdef MGX_ABLATION
    // (timing experiment: keep the pruning bounds of the batch's previous execute = "the final bound known from the start")
    if (std::getenv("MGX_KEEP_BOUNDS")) clear_bytes = b->score.qids.size() * 8 * 8;
#endif
    MGX_HIP(hipMemsetAsync(b->d_score_out.p, 0, clear_bytes, s));
    if (b->timing) {
      MGX_HIP(hipEventRecord(ev0, s));
    }
    static const bool kChainEnv = !(std::getenv("MGX_CHAIN") && atoi(std::getenv("MGX_CHAIN")) == 0);
    const bool kChain = kChainEnv && idx->fifo.load(std::memory_order_relaxed) != 0;
    if (kChain && b->res_owned) {
      if (!b->res->main_ev) MGX_HIP(hipEventCreateWithFlags(&b->res->main_ev, hipEventDisableTiming));
      std::lock_guard<std::mutex> lock(idx->table_mu);
      if (idx->chain_ev && idx->chain_ev != b->res->main_ev) MGX_HIP(hipStreamWaitEvent(s, idx->chain_ev, 0));
    }
    uint32_t n_fast = 0;
    for (int t = 0; t < kFastMaxScore; ++t) n_fast += g.dev_fast[t].n_items;
    const bool main_work = n_fast != 0 || g.n_items_wave != 0 || g.dev_merge.n_items != 0;
    const bool side_work = g.n_items != 0 || g.dev_wave_lists.n_items != 0 || g.dev_cand.n_items != 0;
    hipStream_t side = s;
    if (main_work && side_work) {
      // fork: the (small) shares of the general kernel and of the list-operand plan run on the side stream while the
      // main launch fills the chip
      if (!b->res->fork_ev) {
        MGX_HIP(hipEventCreateWithFlags(&b->res->fork_ev, hipEventDisableTiming));
        MGX_HIP(hipEventCreateWithFlags(&b->res->join_ev, hipEventDisableTiming));
      }
      MGX_HIP(hipEventRecord(b->res->fork_ev, s));
      MGX_HIP(hipStreamWaitEvent(idx->side_stream, b->res->fork_ev, 0));
      side = idx->side_stream;
    }
    MGX_LAUNCH(LaunchCand(kModeScore, idx->dev, g.dev_cand, g.cand_leaves, g.cand_instr, g.cand_cap, side));
    MGX_LAUNCH(LaunchWaveScore(idx->dev, g.dev_wave_lists, g.wplan_lists, side));
    MGX_LAUNCH(LaunchTileEval(kModeScore, idx->dev, g.dev, g.plan, side, g.n_items_plain));
    // (both conditions are the same on every rank: the collective is entered by all or by none)
    static const uint64_t kMinShardDocs = std::getenv("MGX_SEED_EXCHANGE_MIN_DOCS") ? static_cast<uint64_t>(atoll(std::getenv("MGX_SEED_EXCHANGE_MIN_DOCS"))) : 1000000ull;
    if (sg && sg->fn && g.seed_k != 0 && g.seed_table_docs / static_cast<uint64_t>(sg->world) >= kMinShardDocs) {
      // doc-range shards: a shard's own matches give a weak bound (a 1.25M-doc shard scores 22 % of its matches where
      // the whole table scores 7 %); the seeds of ALL shards together are a sample of the whole table
      for (int t = 0; t < kFastMaxScore; ++t) {
        DevBatch seeds = g.dev_fast[t];
        seeds.n_items = g.n_seed_fast[t];
        MGX_LAUNCH(LaunchBitmapScore(static_cast<uint32_t>(t + 1), idx->dev, seeds, g.fplan, s));
      }
      {
        const int rc = SeedBoundExchange(b, g, *sg, s);
        if (rc) return rc;
      }
      for (int t = 0; t < kFastMaxScore; ++t) {
        DevBatch rest = g.dev_fast[t];
        rest.items += g.n_seed_fast[t];
        rest.n_items -= g.n_seed_fast[t];
        MGX_LAUNCH(LaunchBitmapScore(static_cast<uint32_t>(t + 1), idx->dev, rest, g.fplan, s));
      }
    } else {
      for (int t = 0; t < kFastMaxScore; ++t)
        MGX_LAUNCH(LaunchBitmapScore(static_cast<uint32_t>(t + 1), idx->dev, g.dev_fast[t], g.fplan, s));
    }
    MGX_LAUNCH(LaunchWaveScore(idx->dev, g.dev_wave, g.wplan, s));
    MGX_LAUNCH(LaunchMergeScore(idx->dev, g.dev_merge, g.merge_leaves, g.merge_instr, g.merge_ops, g.merge_cap, s));
    if (side != s) {
      MGX_HIP(hipEventRecord(b->res->join_ev, side));
      MGX_HIP(hipStreamWaitEvent(s, b->res->join_ev, 0));
    }
    if (b->timing) {
      MGX_HIP(hipEventRecord(ev1, s));
      timed = true;
    }
    if (kChain && b->res_owned) {
      std::lock_guard<std::mutex> lock(idx->table_mu);
      MGX_HIP(hipEventRecord(b->res->main_ev, s));
      idx->chain_ev = b->res->main_ev;
    }
    const uint32_t n = static_cast<uint32_t>(g.qids.size());
    MGX_LAUNCH(LaunchMergeTopK(g.dev.queries, g.d_ident.as<uint32_t>(), n, 0, g.dev.cand_keys, g.dev.cand_docs,
                               g.dev.cand_n, /*kq=*/0, /*kj=*/g.dev.cand_stride, /*dj=*/g.dev.cand_stride, /*cq=*/0,
                               /*cj=*/1,
                               b->ex_keys(), b->ex_docs(), b->ex_counts(n), b->top_stride, b->sc_docs(),
                               b->sc_scores(), b->sc_n(), b->page_stride, g.d_list_begin.as<uint32_t>(),
                               b->sc_counters(), b->ex_totals(n), s));
  }
  if (!b->bitmap.qids.empty()) {
    mgx_batch::Group& g = b->bitmap;
    MGX_HIP(hipMemsetAsync(g.d_counters.p, 0, g.d_counters.bytes, s));
    if (b->timing && !timed) {
      MGX_HIP(hipEventRecord(ev0, s));
    }
    MGX_LAUNCH(LaunchTileEval(kModeBitmap, idx->dev, g.dev, g.plan, s, g.n_items_plain));
    if (b->timing && !timed) {
      MGX_HIP(hipEventRecord(ev1, s));
      timed = true;
    }
    MGX_LAUNCH(LaunchScanTiles(g.dev.tile_cnt, static_cast<uint32_t>(g.qids.size()), idx->dev.n_tiles,
                               b->d_tile_start.as<uint64_t>(), b->d_totals.as<uint64_t>(), s));
  }
  if (!b->page.qids.empty()) {
    mgx_batch::Group& g = b->page;
    const uint32_t n = static_cast<uint32_t>(g.qids.size());
    MGX_HIP(hipMemsetAsync(b->d_page_out.p, 0, b->po_docs, s));  // counters + totals
    if (b->timing && !timed) {
      MGX_HIP(hipEventRecord(ev0, s));
    }
    // candidate-driven queries (selective: a sparse list drives) beside the tile passes, on the side stream: they write
    // their totals and pages themselves
    const bool cand_side = g.dev_cand.n_items != 0 && (g.dev_wave.n_items != 0 || g.n_items != 0) && b->res_owned;
    if (cand_side) {
      if (!b->res->fork_ev) {
        MGX_HIP(hipEventCreateWithFlags(&b->res->fork_ev, hipEventDisableTiming));
        MGX_HIP(hipEventCreateWithFlags(&b->res->join_ev, hipEventDisableTiming));
      }
      MGX_HIP(hipEventRecord(b->res->fork_ev, s));
      MGX_HIP(hipStreamWaitEvent(idx->side_stream, b->res->fork_ev, 0));
      MGX_LAUNCH(LaunchCand(kModeDocPage, idx->dev, g.dev_cand, g.cand_leaves, g.cand_instr, 0, idx->side_stream));
      MGX_HIP(hipEventRecord(b->res->join_ev, idx->side_stream));
    }
    MGX_LAUNCH(LaunchWaveCount(idx->dev, g.dev_wave, g.wplan, false, s));
    MGX_LAUNCH(LaunchTileEval(kModeDocCount, idx->dev, g.dev, g.plan, s, g.n_items_plain));
    if (b->timing && !timed) {
      MGX_HIP(hipEventRecord(ev1, s));
      timed = true;
    }
    MGX_LAUNCH(LaunchScanTiles(g.dev.tile_cnt, n, idx->dev.n_tiles, b->d_ptile_start.as<uint64_t>(),
                               const_cast<uint64_t*>(g.dev.totals), s, g.d_cand_skip.as<uint8_t>()));
    if (!cand_side) MGX_LAUNCH(LaunchCand(kModeDocPage, idx->dev, g.dev_cand, g.cand_leaves, g.cand_instr, 0, s));
    MGX_LAUNCH(LaunchWavePage(idx->dev, g.dev_page_wave, g.wplan, s));
    MGX_LAUNCH(LaunchTileEval(kModeDocPage, idx->dev, g.dev_page_block, g.plan, s));
    if (cand_side) MGX_HIP(hipStreamWaitEvent(s, b->res->join_ev, 0));
  }
  if (b->timing) {
    if (timed) {
      b->events.emplace_back(ev0, ev1);
    } else {
      (void)hipEventDestroy(ev0);
      (void)hipEventDestroy(ev1);
    }
  }
  {
    int rc = IssueResultCopy(b, s);
    if (rc) return rc;
  }
  b->executed = true;
  b->last_stream = s;
  return MGX_OK;
}

static int FetchImpl(mgx_batch* b, mgx_result_view* out) {
  mgx_index* idx = b->idx;
  MGX_HIP(hipSetDevice(idx->device));
  hipStream_t s = b->last_stream;
  // ---- score group: one async copy of the packed result block, then the only synchronisation of the step ----
  const uint32_t* page_n = nullptr;
  const uint32_t* page_docs = nullptr;
  const double* page_scores = nullptr;
  const unsigned long long* override_tot = nullptr;
  if (b->res && b->res->copy_issued) {
    // the copies were enqueued behind the kernels: wait for THIS batch's event only (a later batch may be running).
    // hipEventSynchronize keeps a core busy for the whole wait (with or without hipEventBlockingSync on this runtime:
    // measured 0.85 ms of CPU per 0.85 ms step); a serving loop keeps several batches in flight, so the collecting
    // thread naps between queries of the event instead — tens of microseconds of added latency per batch, no CPU.
    // MGX_WAIT_POLICY=spin restores the busy wait (a caller with one batch in flight who wants the last microseconds).
    static const bool spin = std::getenv("MGX_WAIT_POLICY") != nullptr && std::string(std::getenv("MGX_WAIT_POLICY")) == "spin";
    if (spin) {
      MGX_HIP(hipEventSynchronize(b->res->done_ev));
    } else {
      for (;;) {
        const hipError_t q = hipEventQuery(b->res->done_ev);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return Fail(MGX_ERR_INTERNAL, std::string("hipEventQuery: ") + hipGetErrorString(q));
        std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
    }
  } else {
    if (!b->page.qids.empty())
      MGX_HIP(hipMemcpyAsync(b->h_page_out, b->d_page_out.p, b->po_bytes, hipMemcpyDeviceToHost, s));
    if (!b->score.qids.empty())
      MGX_HIP(hipMemcpyAsync(b->h_score_out, b->d_score_out.p, b->so_bytes, hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
  }
  if (!b->score.qids.empty()) {
    mgx_batch::Group& g = b->score;
    const size_t n = g.qids.size();
    const char* h = static_cast<const char*>(b->h_score_out);
    const unsigned long long* hc = reinterpret_cast<const unsigned long long*>(h);
    for (size_t i = 0; i < n * 8; ++i) g.h_counters[i] = hc[i];
    override_tot = reinterpret_cast<const unsigned long long*>(h + b->so_override);
    page_scores = reinterpret_cast<const double*>(h + b->so_scores);
    page_docs = reinterpret_cast<const uint32_t*>(h + b->so_docs);
    page_n = reinterpret_cast<const uint32_t*>(h + b->so_n);
  }
  // ---- bitmap group: totals -> page sizes -> expand ----
  std::vector<uint64_t> totals, take, out_off;
  std::vector<uint32_t> bm_docs;
  std::vector<std::vector<uint32_t>> deep_docs;
  std::vector<std::vector<double>> deep_scores;
  if (!b->bitmap.qids.empty()) {
    mgx_batch::Group& g = b->bitmap;
    const size_t n = g.qids.size();
    MGX_HIP(hipStreamSynchronize(s));  // (this group's sizes make a host round trip)
    totals.resize(n);
    take.resize(n);
    out_off.resize(n);
    MGX_HIP(hipMemcpy(g.h_counters.data(), g.d_counters.p, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    MGX_HIP(hipMemcpy(totals.data(), b->d_totals.p, n * 8, hipMemcpyDeviceToHost));
    uint64_t at = 0;
    for (size_t i = 0; i < n; ++i) {
      const QuerySpec& sp = b->specs[g.qids[i]];
      take[i] = (sp.limit == 0 || sp.deep_score) ? totals[i] : std::min<uint64_t>(totals[i], sp.limit);
      out_off[i] = at;
      at += take[i];
    }
    if (at > 0xFFFFFFFFull) return Fail(MGX_ERR_OUT_OF_RANGE, "batch result exceeds 2^32 doc ids");
    bm_docs.resize(at);
    if (at) {
      if (b->d_out.bytes < at * 4) MGX_HIP(b->d_out.Alloc(at * 4));
      MGX_HIP(hipMemcpyAsync(b->d_take.p, take.data(), n * 8, hipMemcpyHostToDevice, s));
      MGX_HIP(hipMemcpyAsync(b->d_out_off.p, out_off.data(), n * 8, hipMemcpyHostToDevice, s));
      MGX_LAUNCH(LaunchExpand(g.dev.rbits, b->d_tile_start.as<uint64_t>(), b->d_totals.as<uint64_t>(),
                              b->d_take.as<uint64_t>(), b->d_out_off.as<uint64_t>(), b->d_reverse.as<uint32_t>(),
                              static_cast<uint32_t>(n), idx->dev.n_tiles, idx->dev.first_doc_id,
                              b->d_out.as<uint32_t>(), s));
      MGX_HIP(hipMemcpyAsync(bm_docs.data(), b->d_out.p, at * 4, hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
    }
    // deep SORT _score pages: BM25Scorer::ScoreDocuments over every match, ResultSorter::SortByScore by a full sort
    deep_docs.assign(n, {});
    deep_scores.assign(n, {});
    for (size_t i = 0; i < n; ++i) {
      const QuerySpec& sp = b->specs[g.qids[i]];
      if (!sp.deep_score || take[i] == 0) continue;
      const uint64_t m = take[i];
      const uint32_t lo = static_cast<uint32_t>(std::min<uint64_t>(sp.offset, m));
      const uint32_t hi = sp.limit == 0 ? static_cast<uint32_t>(m)
                                        : static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(lo) + sp.limit, m));
      if (hi == lo) continue;
      uint64_t n2 = 2048;
      while (n2 < m) n2 <<= 1;
      const uint32_t* cand = b->d_out.as<uint32_t>() + out_off[i];
      DevBuf d_g, d_i, d_sc, d_k, d_d, d_pd, d_ps;
      MGX_HIP(Upload(d_g, sp.deep_grams.data(), sp.deep_grams.size()));
      MGX_HIP(Upload(d_i, sp.deep_idfs.data(), sp.deep_idfs.size()));
      MGX_HIP(d_sc.Alloc(m * 8));
      MGX_HIP(d_k.Alloc(n2 * 8));
      MGX_HIP(d_d.Alloc(n2 * 4));
      MGX_HIP(d_pd.Alloc(static_cast<size_t>(hi - lo) * 4));
      MGX_HIP(d_ps.Alloc(static_cast<size_t>(hi - lo) * 8));
      MGX_LAUNCH(LaunchScoreCandidates(idx->dev, cand, m, d_g.as<uint32_t>(), d_i.as<double>(),
                                       static_cast<uint32_t>(sp.deep_grams.size()), sp.k1, sp.b, sp.avgdl,
                                       d_sc.as<double>(), s));
      MGX_HIP(hipMemsetAsync(d_k.as<uint64_t>() + m, 0, (n2 - m) * 8, s));
      MGX_HIP(hipMemsetAsync(d_d.as<uint32_t>() + m, 0, (n2 - m) * 4, s));
      MGX_LAUNCH(LaunchMakeSortKeys(cand, d_sc.as<double>(), m, sp.reverse ? 1 : 0, d_k.as<uint64_t>(),
                                    d_d.as<uint32_t>(), s));
      MGX_LAUNCH(LaunchSortPairs(d_k.as<uint64_t>(), d_d.as<uint32_t>(), n2, s));
      MGX_LAUNCH(LaunchSortPage(d_k.as<uint64_t>(), d_d.as<uint32_t>(), lo, hi, sp.reverse ? 1 : 0,
                                d_pd.as<uint32_t>(), d_ps.as<double>(), s));
      deep_docs[i].resize(hi - lo);
      deep_scores[i].resize(hi - lo);
      MGX_HIP(hipMemcpyAsync(deep_docs[i].data(), d_pd.p, static_cast<size_t>(hi - lo) * 4, hipMemcpyDeviceToHost, s));
      MGX_HIP(hipMemcpyAsync(deep_scores[i].data(), d_ps.p, static_cast<size_t>(hi - lo) * 8, hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
    }
  }
  const uint32_t* dp_docs = nullptr;
  const unsigned long long* dp_totals = nullptr;
  if (!b->page.qids.empty()) {
    const char* h = static_cast<const char*>(b->h_page_out);
    const unsigned long long* hc = reinterpret_cast<const unsigned long long*>(h);
    for (size_t i = 0; i < b->page.qids.size() * 8; ++i) b->page.h_counters[i] = hc[i];
    dp_docs = reinterpret_cast<const uint32_t*>(h + b->po_docs);
    dp_totals = reinterpret_cast<const unsigned long long*>(h + b->po_totals);
  }
  // ---- assemble in batch order ----
  std::vector<uint32_t> pos_in_group(b->n_queries, 0);
  for (size_t i = 0; i < b->score.qids.size(); ++i) pos_in_group[b->score.qids[i]] = static_cast<uint32_t>(i);
  for (size_t i = 0; i < b->bitmap.qids.size(); ++i) pos_in_group[b->bitmap.qids[i]] = static_cast<uint32_t>(i);
  for (size_t i = 0; i < b->page.qids.size(); ++i) pos_in_group[b->page.qids[i]] = static_cast<uint32_t>(i);
  // two passes: sizes first, then bulk copies (a page-mode batch returns ~10^6 doc ids per fetch)
  uint64_t n_out = 0;
  for (uint32_t qi = 0; qi < b->n_queries; ++qi) {
    mgx_query_result& r = b->h_results[qi];
    const uint32_t gi = pos_in_group[qi];
    const bool sc = b->specs[qi].mode == kModeScore, pg = b->specs[qi].mode == kModeDocPage;
    const unsigned long long* c =
        (sc ? b->score : pg ? b->page : b->bitmap).h_counters.data() + static_cast<size_t>(gi) * 8;
    r.total_candidates = c[0];
    r.after_intersection = c[1];
    r.after_not = c[2];
    r.after_filters = c[3];
    r.total = c[4];
    if (sc && b->merged_shards) r.total = override_tot[gi];
    if (pg && b->merged_shards) r.total = dp_totals[gi];
    r.n_docs = sc                         ? page_n[gi]
               : pg                       ? static_cast<uint32_t>(std::min<uint64_t>(r.total, b->specs[qi].limit))
               : b->specs[qi].deep_score ? static_cast<uint32_t>(deep_docs[gi].size())
                                          : static_cast<uint32_t>(take[gi]);
    r.docs_begin = static_cast<uint32_t>(n_out);
    n_out += r.n_docs;
  }
  if (n_out > 0xFFFFFFFFull) return Fail(MGX_ERR_OUT_OF_RANGE, "batch result exceeds 2^32 doc ids");
  b->h_docs.resize(n_out);
  b->h_scores.assign(n_out, 0.0);
  for (uint32_t qi = 0; qi < b->n_queries; ++qi) {
    const mgx_query_result& r = b->h_results[qi];
    if (r.n_docs == 0) continue;
    const uint32_t gi = pos_in_group[qi];
    const uint32_t mode = b->specs[qi].mode;
    uint32_t* dst = b->h_docs.data() + r.docs_begin;
    if (mode == kModeScore) {
      std::memcpy(dst, page_docs + static_cast<size_t>(gi) * b->page_stride, r.n_docs * sizeof(uint32_t));
      std::memcpy(b->h_scores.data() + r.docs_begin, page_scores + static_cast<size_t>(gi) * b->page_stride,
                  r.n_docs * sizeof(double));
    } else if (mode == kModeDocPage) {
      std::memcpy(dst, dp_docs + static_cast<size_t>(gi) * b->doc_page_stride, r.n_docs * sizeof(uint32_t));
    } else if (b->specs[qi].deep_score) {
      std::memcpy(dst, deep_docs[gi].data(), r.n_docs * sizeof(uint32_t));
      std::memcpy(b->h_scores.data() + r.docs_begin, deep_scores[gi].data(), r.n_docs * sizeof(double));
    } else {
      std::memcpy(dst, bm_docs.data() + out_off[gi], r.n_docs * sizeof(uint32_t));
    }
  }
  out->n_queries = b->n_queries;
  out->queries = b->h_results.data();
  out->docs = b->h_docs.data();
  out->scores = b->h_scores.data();
  return MGX_OK;
}

}  // namespace mgx

extern "C" {

int mgx_batch_prepare(mgx_index* idx, const mgx_query* queries, uint32_t n_queries, mgx_batch** out) {
  if (out) *out = nullptr;
  if (!idx || !out || (n_queries && !queries))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_prepare: null argument");
  try {
    std::vector<mgx::QuerySpec> specs(n_queries);
    const int rc = mgx::CompileAll(idx, queries, n_queries, &specs);
    if (rc) return rc;
    return mgx::PrepareFromSpecs(idx, std::move(specs), out);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_prepare: ") + e.what());
  }
}

int mgx_batch_reset(mgx_batch* batch, const mgx_query* queries, uint32_t n_queries) {
  if (!batch || (n_queries && !queries)) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_reset: null argument");
  try {
    mgx_index* idx = batch->idx;
    std::vector<mgx::QuerySpec> specs = std::move(batch->specs);  // (their vectors' storage is compiled into again)
    specs.resize(n_queries);
    int rc = MGX_OK;
    static const bool kTrace = std::getenv("MGX_TRACE_HOST") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    rc = mgx::CompileAll(idx, queries, n_queries, &specs);
    const auto t1 = std::chrono::steady_clock::now();
    MGX_HIP(hipSetDevice(idx->device));
    mgx::ResetBatch(batch);
    if (rc) return rc;  // (the batch is empty but usable)
    rc = mgx::PrepareInto(batch, idx, std::move(specs));
    if (kTrace) {
      static std::atomic<uint64_t> n{0};
      static std::atomic<uint64_t> us_compile{0}, us_into{0};
      const auto t2 = std::chrono::steady_clock::now();
      us_compile += std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
      us_into += std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count();
      static const uint64_t period = std::max(1, atoi(std::getenv("MGX_TRACE_HOST")) > 1 ? atoi(std::getenv("MGX_TRACE_HOST")) : 64);
      if (++n % period == 0) {
        const double per = static_cast<double>(period);
        fprintf(stderr, "[mgx] reset x%llu: CompileQuery loop %.3f ms, reset + PrepareInto %.3f ms per batch\n",
                static_cast<unsigned long long>(period), us_compile.exchange(0) / (per * 1e3), us_into.exchange(0) / (per * 1e3));
        for (int k = 0; k < 8; ++k)
          fprintf(stderr, "[mgx]   %-24s %.3f ms\n", mgx::kSectionNames[k], mgx::g_section_us[k].exchange(0) / (per * 1e6));
      }
    }
    return rc;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_reset: ") + e.what());
  }
}

int mgx_batch_stream(mgx_batch* batch, void** hip_stream) {
  if (hip_stream) *hip_stream = nullptr;
  if (!batch || !hip_stream || !batch->res) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_stream: null argument");
  if (!batch->res->stream) {
    MGX_HIP(hipSetDevice(batch->idx->device));
    MGX_HIP(hipStreamCreateWithFlags(&batch->res->stream, hipStreamNonBlocking));
  }
  *hip_stream = batch->res->stream;
  return MGX_OK;
}

int mgx_batch_execute(mgx_batch* batch, void* hip_stream) {
  if (!batch) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_execute: null batch");
  try {
    hipStream_t s = nullptr;
    int rc = mgx::BatchStream(batch, hip_stream, &s);
    if (rc) return rc;
    return mgx::ExecuteImpl(batch, s);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_execute: ") + e.what());
  }
}

int mgx_batch_fetch(mgx_batch* batch, mgx_result_view* out) {
  if (out) std::memset(out, 0, sizeof(*out));
  if (!batch || !out) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_fetch: null argument");
  if (!batch->executed) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_fetch: batch was never executed");
  try {
    return mgx::FetchImpl(batch, out);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_fetch: ") + e.what());
  }
}

int mgx_batch_count_df(mgx_batch* batch, void* hip_stream) {
  if (!batch) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_count_df: null batch");
  try {
    hipStream_t s = nullptr;
    int rc = mgx::BatchStream(batch, hip_stream, &s);
    if (rc) return rc;
    rc = mgx::CountDfImpl(batch, s);
    if (rc) return rc;
    batch->df_ready = true;
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_count_df: ") + e.what());
  }
}

int mgx_batch_df_buffer(mgx_batch* batch, uint64_t** device_counts, uint32_t* n) {
  if (device_counts) *device_counts = nullptr;
  if (n) *n = 0;
  if (!batch || !device_counts || !n) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_df_buffer: null argument");
  *device_counts = batch->d_text_df.as<uint64_t>();
  *n = static_cast<uint32_t>(batch->h_text_df.size());
  return MGX_OK;
}

// One group of a batch (its SORT _score queries, or its docid-ordered pages) in the exchange layout. The public entry
// points below take uniform batches (one all-gather moves one blob); the local merge of a mutable table takes each group
// of a mixed batch in turn.
static int ExportGroup(mgx_batch* batch, bool pages, uint64_t* blob64, uint32_t* blob32, uint32_t* stride, void* hip_stream);
static int MergeGroup(mgx_batch* batch, bool pages, uint32_t n_shards, const uint64_t* blob64, uint64_t pitch64,
                      const uint32_t* blob32, uint64_t pitch32, void* hip_stream);

int mgx_batch_export_topk(mgx_batch* batch, uint64_t* blob64, uint32_t* blob32, uint32_t* stride,
                          void* hip_stream) {
  if (!batch || !stride) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_export_topk: null argument");
  const bool pages = batch->score.qids.empty() && batch->bitmap.qids.empty() && !batch->page.qids.empty();
  if (!pages && (!batch->bitmap.qids.empty() || !batch->page.qids.empty() || batch->score.qids.empty()))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT,
                     "mgx_batch_export_topk: the batch must be all MGX_SORT_SCORE or all docid-ordered pages "
                     "(0 < limit <= 16384)");
  return ExportGroup(batch, pages, blob64, blob32, stride, hip_stream);
}

static int ExportGroup(mgx_batch* batch, bool pages, uint64_t* blob64, uint32_t* blob32, uint32_t* stride, void* hip_stream) {
  *stride = pages ? batch->doc_page_stride : batch->top_stride;
  if (!blob64 && !blob32) return MGX_OK;  // size query
  if (!blob64 || !blob32) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_export_topk: null blob");
  if (!batch->executed) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_export_topk: not executed");
  hipStream_t s = nullptr;
  {
    int rc_s = mgx::BatchStream(batch, hip_stream, &s);
    if (rc_s) return rc_s;
  }
  if (pages) {
    const mgx_batch::Group& g = batch->page;
    MGX_HIP(hipSetDevice(batch->idx->device));
    MGX_LAUNCH(mgx::LaunchExportPages(g.dev.queries, static_cast<uint32_t>(g.qids.size()), batch->doc_page_stride,
                                      g.dev.page_docs, g.dev.totals, blob64, blob32, s));
    return MGX_OK;
  }
  const size_t n = batch->score.qids.size();
  const size_t ks = n * batch->top_stride;
  MGX_HIP(hipSetDevice(batch->idx->device));
  MGX_HIP(hipMemcpyAsync(blob64, batch->ex_keys(), (ks + n) * 8, hipMemcpyDeviceToDevice, s));
  MGX_HIP(hipMemcpyAsync(blob32, batch->ex_docs(), (ks + n) * 4, hipMemcpyDeviceToDevice, s));
  return MGX_OK;
}

int mgx_batch_export_buffer(mgx_batch* batch, void** device_blob, uint64_t* bytes, uint64_t* offset32) {
  if (device_blob) *device_blob = nullptr;
  if (bytes) *bytes = 0;
  if (offset32) *offset32 = 0;
  if (!batch || !device_blob || !bytes || !offset32)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_export_buffer: null argument");
  if (!batch->bitmap.qids.empty() || !batch->page.qids.empty() || batch->score.qids.empty()) return MGX_OK;  // none
  *device_blob = batch->d_export.p;
  *bytes = batch->ex_bytes;
  *offset32 = batch->ex_off32;
  return MGX_OK;
}

int mgx_batch_merge_shards(mgx_batch* batch, uint32_t n_shards, const uint64_t* blob64, uint64_t pitch64,
                           const uint32_t* blob32, uint64_t pitch32, void* hip_stream) {
  if (!batch || !blob64 || !blob32 || n_shards == 0)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_merge_shards: null argument");
  const bool pages = batch->score.qids.empty() && batch->bitmap.qids.empty() && !batch->page.qids.empty();
  if (!pages && (!batch->bitmap.qids.empty() || !batch->page.qids.empty() || batch->score.qids.empty()))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT,
                     "mgx_batch_merge_shards: the batch must be all MGX_SORT_SCORE or all docid-ordered pages");
  return MergeGroup(batch, pages, n_shards, blob64, pitch64, blob32, pitch32, hip_stream);
}

static int MergeGroup(mgx_batch* batch, bool pages, uint32_t n_shards, const uint64_t* blob64, uint64_t pitch64,
                      const uint32_t* blob32, uint64_t pitch32, void* hip_stream) {
  hipStream_t s = nullptr;
  {
    int rc_s = mgx::BatchStream(batch, hip_stream, &s);
    if (rc_s) return rc_s;
  }
  if (pages) {
    // shards hold disjoint doc ranges, but the merge needs no such knowledge: pages are best-first lists by doc id
    mgx_batch::Group& g = batch->page;
    const uint32_t n = static_cast<uint32_t>(g.qids.size());
    const uint32_t stride = batch->doc_page_stride;
    if (pitch64 == 0) pitch64 = static_cast<uint64_t>(n) * stride + n;  // blobs laid rank after rank
    if (pitch32 == 0) pitch32 = static_cast<uint64_t>(n) * stride + n;
    MGX_HIP(hipSetDevice(batch->idx->device));
    if (batch->d_page_scratch.bytes == 0)
      MGX_HIP(batch->d_page_scratch.Alloc(static_cast<size_t>(n) * stride * 8 + static_cast<size_t>(n) * 4));
    double* scratch_scores = batch->d_page_scratch.as<double>();
    uint32_t* scratch_n = reinterpret_cast<uint32_t*>(scratch_scores + static_cast<size_t>(n) * stride);
    MGX_LAUNCH(mgx::LaunchMergeTopK(g.dev.queries, g.d_ident.as<uint32_t>(), n, n_shards, blob64, blob32,
                                    blob32 + static_cast<uint64_t>(n) * stride, /*kq=*/stride, /*kj=*/pitch64,
                                    /*dj=*/pitch32, /*cq=*/1, /*cj=*/pitch32, nullptr, nullptr, nullptr, 0,
                                    g.dev.page_docs, scratch_scores, scratch_n, stride, nullptr, nullptr, nullptr, s));
    MGX_LAUNCH(mgx::LaunchSumTotals(blob64 + static_cast<uint64_t>(n) * stride, n_shards, n, pitch64,
                                    const_cast<uint64_t*>(g.dev.totals), s));
    batch->merged_shards = true;
    batch->last_stream = s;
    return mgx::IssueResultCopy(batch, s);  // (the merged page replaces the shard's own in the pinned block)
  }
  const uint32_t n = static_cast<uint32_t>(batch->score.qids.size());
  if (pitch64 == 0) pitch64 = static_cast<uint64_t>(n) * batch->top_stride + n;  // blobs laid rank after rank
  if (pitch32 == 0) pitch32 = static_cast<uint64_t>(n) * batch->top_stride + n;
  MGX_HIP(hipSetDevice(batch->idx->device));
  MGX_LAUNCH(mgx::LaunchMergeTopK(batch->score.dev.queries, batch->score.d_ident.as<uint32_t>(), n, n_shards, blob64,
                                  blob32, blob32 + static_cast<uint64_t>(n) * batch->top_stride,
                                  /*kq=*/batch->top_stride, /*kj=*/pitch64, /*dj=*/pitch32, /*cq=*/1, /*cj=*/pitch32,
                                  nullptr, nullptr, nullptr, 0, batch->sc_docs(), batch->sc_scores(), batch->sc_n(),
                                  batch->page_stride, nullptr, nullptr, nullptr, s));
  MGX_LAUNCH(mgx::LaunchSumTotals(blob64 + static_cast<uint64_t>(n) * batch->top_stride, n_shards, n, pitch64,
                                  batch->sc_override(), s));
  batch->merged_shards = true;
  batch->last_stream = s;
  return mgx::IssueResultCopy(batch, s);
}

// Mutable tables: a table is its main index plus a small delta index on the same device (documents added or changed
// since the main index was built). The same batch is compiled against both; the two calls below are the exchange of a
// sharded table with the wire taken out.
int mgx_batch_df_merge_local(mgx_batch* primary, mgx_batch* const* others, uint32_t n_others, void* hip_stream) {
  if (!primary || (n_others && !others)) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_df_merge_local: null argument");
  try {
    const uint32_t n = static_cast<uint32_t>(primary->h_text_df.size());
    for (uint32_t j = 0; j < n_others; ++j)
      if (!others[j] || others[j]->h_text_df.size() != n || others[j]->idx->device != primary->idx->device)
        return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_df_merge_local: the batches do not hold the same text-level terms");
    if (n == 0) return MGX_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    int rc = mgx_batch_count_df(primary, hip_stream);
    if (rc) return rc;
    for (uint32_t j = 0; j < n_others; ++j) {
      rc = mgx_batch_count_df(others[j], hip_stream);
      if (rc) return rc;
      MGX_LAUNCH(mgx::LaunchAddU64(primary->d_text_df.as<uint64_t>(), others[j]->d_text_df.as<uint64_t>(), n, s));
    }
    for (uint32_t j = 0; j < n_others; ++j)
      MGX_HIP(hipMemcpyAsync(others[j]->d_text_df.p, primary->d_text_df.p, static_cast<size_t>(n) * 8, hipMemcpyDeviceToDevice, s));
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_df_merge_local: ") + e.what());
  }
}

int mgx_batch_merge_local(mgx_batch* primary, mgx_batch* const* others, uint32_t n_others, void* hip_stream) {
  if (!primary || (n_others && !others)) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_merge_local: null argument");
  if (primary->n_queries == 0) return MGX_OK;
  if (!primary->res) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_merge_local: not a prepared batch object");
  if (!primary->bitmap.qids.empty())
    return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED,
                     "mgx_batch_merge_local: a query without a page bound (limit 0 or > 16384, not SORT _score) cannot be merged");
  try {
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    for (uint32_t j = 0; j < n_others; ++j)
      if (!others[j] || others[j]->n_queries != primary->n_queries || others[j]->idx->device != primary->idx->device ||
          others[j]->score.qids.size() != primary->score.qids.size() || others[j]->page.qids.size() != primary->page.qids.size())
        return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_merge_local: the batches do not hold the same queries");
    MGX_HIP(hipSetDevice(primary->idx->device));
    // a mixed batch has two groups (SORT _score queries, docid-ordered pages): each is exported, remapped and merged on its own
    uint64_t need = 0;
    for (int pass = 0; pass < 2; ++pass) {
      const bool pages = pass == 1;
      const uint64_t n = pages ? primary->page.qids.size() : primary->score.qids.size();
      if (n == 0) continue;
      const uint64_t stride = pages ? primary->doc_page_stride : primary->top_stride;
      need += ((n * stride + n) * 12 + 7) / 8 * 8 * (1ull + n_others);
    }
    void* recv = nullptr;
    MGX_HIP(primary->res->Exchange(1, need, &recv));
    char* g = static_cast<char*>(recv);
    for (int pass = 0; pass < 2; ++pass) {
      const bool pages = pass == 1;
      const uint32_t n = static_cast<uint32_t>(pages ? primary->page.qids.size() : primary->score.qids.size());
      if (n == 0) continue;
      uint32_t stride = 0;
      int rc = ExportGroup(primary, pages, nullptr, nullptr, &stride, hip_stream);
      if (rc) return rc;
      for (uint32_t j = 0; j < n_others; ++j) {
        uint32_t st = 0;
        rc = ExportGroup(others[j], pages, nullptr, nullptr, &st, hip_stream);
        if (rc) return rc;
        if (st != stride) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_merge_local: the batches do not hold the same queries");
      }
      const uint64_t elems = static_cast<uint64_t>(n) * stride + n;
      const uint64_t off32 = elems * 8, bytes = (elems * 12 + 7) / 8 * 8;
      for (uint32_t j = 0; j <= n_others; ++j) {
        mgx_batch* b = j == 0 ? primary : others[j - 1];
        uint64_t* b64 = reinterpret_cast<uint64_t*>(g + bytes * j);
        uint32_t* b32 = reinterpret_cast<uint32_t*>(g + bytes * j + off32);
        uint32_t st = 0;
        rc = ExportGroup(b, pages, b64, b32, &st, hip_stream);
        if (rc) return rc;
        const mgx_index* ix = b->idx;
        if (ix->d_doc_map.p) {
          const mgx_batch::Group& grp = pages ? b->page : b->score;
          MGX_LAUNCH(mgx::LaunchRemapBlobDocs(grp.dev.queries, n, stride, ix->d_doc_map.as<uint32_t>(), ix->dev.first_doc_id,
                                              ix->dev.n_docs, pages ? b64 : nullptr, b32, s));
        }
      }
      rc = MergeGroup(primary, pages, n_others + 1, reinterpret_cast<const uint64_t*>(g), bytes / 8,
                      reinterpret_cast<const uint32_t*>(g + off32), bytes / 4, hip_stream);
      if (rc) return rc;
      g += bytes * (1ull + n_others);
    }
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_merge_local: ") + e.what());
  }
}

// =================================================================================================================
// RCCL exchange (one rank per doc-range shard): the collective of the sharded path lives behind the C ABI
// =================================================================================================================
// librccl is loaded on first use (dlopen), so the library itself loads on hosts without it; the five entry points
// below are RCCL's own (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclAllReduce :611, ncclAllGather :678).
namespace mgx {
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
};
static Rccl& LoadRccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.error = std::string("librccl.so not found: ") + dlerror();
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.lib, n);
      if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}
static int RcclFail(const char* what, int rc) {
  Rccl& r = LoadRccl();
  return Fail(MGX_ERR_INTERNAL, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") +
                                    " (" + std::to_string(rc) + ")");
}
}  // namespace mgx

struct mgx_comm {
  void* comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

int mgx_comm_unique_id(uint8_t* id) {
  if (!id) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_comm_unique_id: null argument");
  mgx::Rccl& r = mgx::LoadRccl();
  if (!r.error.empty()) return mgx::Fail(MGX_ERR_INTERNAL, r.error);
  mgx::RcclId uid{};
  const int rc = r.GetUniqueId(&uid);
  if (rc != 0) return mgx::RcclFail("ncclGetUniqueId", rc);
  std::memcpy(id, uid.internal, MGX_COMM_ID_BYTES);
  return MGX_OK;
}

int mgx_comm_create(const uint8_t* id, int rank, int world, int device, mgx_comm** out) {
  if (out) *out = nullptr;
  if (!id || !out || world < 1 || rank < 0 || rank >= world)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_comm_create: bad argument");
  mgx::Rccl& r = mgx::LoadRccl();
  if (!r.error.empty()) return mgx::Fail(MGX_ERR_INTERNAL, r.error);
  MGX_HIP(hipSetDevice(device));
  mgx::RcclId uid{};
  std::memcpy(uid.internal, id, MGX_COMM_ID_BYTES);
  auto c = std::make_unique<mgx_comm>();
  c->rank = rank;
  c->world = world;
  c->device = device;
  const int rc = r.CommInitRank(&c->comm, world, uid, rank);
  if (rc != 0) return mgx::RcclFail("ncclCommInitRank", rc);
  *out = c.release();
  return MGX_OK;
}

void mgx_comm_destroy(mgx_comm* comm) {
  if (!comm) return;
  if (comm->comm) (void)mgx::LoadRccl().CommDestroy(comm->comm);
  delete comm;
}

int mgx_comm_abort(mgx_comm* comm) {
  if (!comm) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_comm_abort: null argument");
  if (!comm->comm) return MGX_OK;  // (already aborted)
  mgx::Rccl& r = mgx::LoadRccl();
  void* c = comm->comm;
  comm->comm = nullptr;
  const int rc = r.CommAbort ? r.CommAbort(c) : r.CommDestroy(c);
  return rc != 0 ? mgx::RcclFail("ncclCommAbort", rc) : MGX_OK;
}

int mgx_batch_exchange_df(mgx_batch* batch, mgx_comm* comm, void* hip_stream) {
  if (!batch || !comm) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_exchange_df: null argument");
  if (!comm->comm) return mgx::Fail(MGX_ERR_INTERNAL, "mgx_batch_exchange_df: the communicator was aborted");
  uint64_t* counts = nullptr;
  uint32_t n = 0;
  int rc = mgx_batch_df_buffer(batch, &counts, &n);
  if (rc || n == 0) return rc;
  rc = mgx_batch_count_df(batch, hip_stream);
  if (rc) return rc;
  hipStream_t s = nullptr;
  rc = mgx::BatchStream(batch, hip_stream, &s);
  if (rc) return rc;
  // PopulateTermDocumentFrequency over every shard's candidates: the per-shard counts are summed before any rank takes idf
  const int nrc = mgx::LoadRccl().AllReduce(counts, counts, n, mgx::kRcclUint64, mgx::kRcclSum, comm->comm, s);
  if (nrc != 0) return mgx::RcclFail("ncclAllReduce", nrc);
  return MGX_OK;
}

int mgx_batch_exchange(mgx_batch* batch, mgx_comm* comm, void* hip_stream) {
  if (!batch || !comm) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_exchange: null argument");
  if (!comm->comm) return mgx::Fail(MGX_ERR_INTERNAL, "mgx_batch_exchange: the communicator was aborted");
  if (batch->n_queries == 0) return MGX_OK;
  if (!batch->executed) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_exchange: not executed");
  hipStream_t s = nullptr;
  int rc = mgx::BatchStream(batch, hip_stream, &s);
  if (rc) return rc;
  MGX_HIP(hipSetDevice(batch->idx->device));
  if (!batch->res) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_exchange: not a prepared batch object");
  void* blob = nullptr;
  uint64_t bytes = 0, off32 = 0;
  rc = mgx_batch_export_buffer(batch, &blob, &bytes, &off32);
  if (rc) return rc;
  if (!blob) {  // docid-ordered pages: exported by copy into the exchange layout
    uint32_t stride = 0;
    rc = mgx_batch_export_topk(batch, nullptr, nullptr, &stride, s);
    if (rc) return rc;
    const uint64_t elems = static_cast<uint64_t>(batch->n_queries) * stride + batch->n_queries;
    off32 = elems * 8;
    bytes = (elems * 12 + 7) / 8 * 8;
    MGX_HIP(batch->res->Exchange(0, bytes, &blob));
    rc = mgx_batch_export_topk(batch, static_cast<uint64_t*>(blob),
                               reinterpret_cast<uint32_t*>(static_cast<char*>(blob) + off32), &stride, s);
    if (rc) return rc;
  }
  const uint64_t need = bytes * static_cast<uint64_t>(comm->world);
  void* recv = nullptr;
  MGX_HIP(batch->res->Exchange(1, need, &recv));
  // one all-gather moves every rank's keys, totals, doc ids and counts (both blobs sit in one buffer per rank)
  const int nrc = mgx::LoadRccl().AllGather(blob, recv, bytes, mgx::kRcclUint8, comm->comm, s);
  if (nrc != 0) return mgx::RcclFail("ncclAllGather", nrc);
  char* g = static_cast<char*>(recv);
  return mgx_batch_merge_shards(batch, static_cast<uint32_t>(comm->world), reinterpret_cast<const uint64_t*>(g),
                                bytes / 8, reinterpret_cast<const uint32_t*>(g + off32), bytes / 4, s);
}

namespace mgx {
static int SeedBoundExchange(mgx_batch* b, mgx_batch::Group& g, const SeedGather& sg, hipStream_t s) {
  const uint32_t n = static_cast<uint32_t>(g.qids.size()), k = g.seed_k;
  const uint64_t bytes = static_cast<uint64_t>(n) * k * 8;
  void* mine = nullptr;
  void* all = nullptr;
  MGX_HIP(b->res->Exchange(2, bytes, &mine));
  MGX_HIP(b->res->Exchange(3, bytes * static_cast<uint64_t>(sg.world), &all));
  MGX_LAUNCH(LaunchPackSeedKeys(g.d_list_begin.as<uint32_t>(), g.d_has_seed.as<uint8_t>(), g.dev.cand_keys, g.dev.cand_n,
                                g.dev.cand_stride, n, k, static_cast<uint64_t*>(mine), s));
  if (std::getenv("MGX_VERBOSE"))
    fprintf(stderr, "[mgx] seed-bound exchange: %u queries x %u keys, world %d\n", n, k, sg.world);
  const int grc = sg.fn(sg.user, mine, all, bytes, s);
  if (grc != MGX_OK) return grc == MGX_ERR_INTERNAL && !g_last_error.empty() ? grc : Fail(grc, "the seed-key all-gather failed");
  MGX_LAUNCH(LaunchApplySeedBounds(static_cast<const uint64_t*>(all), static_cast<uint32_t>(sg.world), n, k, g.dev.queries,
                                   g.dev.bounds, s));
  return MGX_OK;
}
static int RcclGather(void* user, const void* mine, void* all, uint64_t bytes, void* hip_stream) {
  const int nrc = LoadRccl().AllGather(mine, all, bytes, kRcclUint8, user, static_cast<hipStream_t>(hip_stream));
  return nrc != 0 ? RcclFail("ncclAllGather (seed keys)", nrc) : MGX_OK;
}
}  // namespace mgx

int mgx_batch_execute_gather(mgx_batch* batch, int world, mgx_gather_fn gather, void* user, void* hip_stream) {
  if (!batch || !gather || world < 1) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_execute_gather: bad argument");
  try {
    hipStream_t s = nullptr;
    int rc = mgx::BatchStream(batch, hip_stream, &s);
    if (rc) return rc;
    mgx::SeedGather sg;
    sg.world = world;
    sg.fn = gather;
    sg.user = user;
    // Splitting the launch costs ~0.03 ms per batch on a 1.25M-doc shard (the main items wait for every seed, two small
    // kernels, the collective: 0.375 -> 0.405 ms per step measured with one rank) and buys a bound made from world x 8
    // tiles instead of 8. What a better start is worth was measured on one GPU with the bounds of a finished pass
    // injected (ablation build, MGX_KEEP_BOUNDS): 0.27 -> 0.20 ms on that shard, 24 % of its kernel. Eight ranks' seeds
    // together cover 64 of a shard's 77 tiles' worth of documents — a bound close to the finished pass's — so the
    // exchange nets about 0.07 x 0.8 - 0.03 = +0.03 ms of 0.27 (10 %) at eight shards, less than its cost at two.
    // Default: ON from four ranks up (shards of a million docs and more, see ExecuteImpl); MGX_SEED_EXCHANGE=0 / 1
    // forces it off / on for every world size.
    static const int forced = std::getenv("MGX_SEED_EXCHANGE") ? atoi(std::getenv("MGX_SEED_EXCHANGE")) : -1;
    const bool on = forced >= 0 ? forced != 0 : world >= 4;
    return mgx::ExecuteImpl(batch, s, on ? &sg : nullptr);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_batch_execute_gather: ") + e.what());
  }
}

int mgx_batch_execute_sharded(mgx_batch* batch, mgx_comm* comm, void* hip_stream) {
  if (!batch || !comm) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_execute_sharded: null argument");
  if (!comm->comm) return mgx::Fail(MGX_ERR_INTERNAL, "mgx_batch_execute_sharded: the communicator was aborted");
  return mgx_batch_execute_gather(batch, comm->world, &mgx::RcclGather, comm->comm, hip_stream);
}

int mgx_batch_algorithmic_bytes(mgx_batch* batch, uint64_t* list_bytes, uint64_t* score_bytes,
                                uint64_t* topk_bytes) {
  if (!batch || !list_bytes || !score_bytes || !topk_bytes)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_algorithmic_bytes: null argument");
  *list_bytes = batch->list_bytes;
  uint64_t sb = 0, tb = 0;
  for (uint32_t qi = 0; qi < batch->n_queries; ++qi) {
    const mgx::QuerySpec& sp = batch->specs[qi];
    const uint64_t R = batch->h_results[qi].total;
    if (sp.mode == mgx::kModeScore) {
      sb += R * (sp.score.size() + 4);
      tb += 12 * std::min<uint64_t>(R, static_cast<uint64_t>(sp.offset) + sp.limit);
    } else {
      tb += 4 * std::min<uint64_t>(R, sp.limit ? sp.limit : R);
    }
  }
  *score_bytes = sb;
  *topk_bytes = tb;
  return MGX_OK;
}

int mgx_batch_kernel_time_ms(mgx_batch* batch, double* avg_ms, uint32_t* n) {
  if (avg_ms) *avg_ms = 0.0;
  if (n) *n = 0;
  if (!batch || !avg_ms || !n) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_batch_kernel_time_ms: null argument");
  MGX_HIP(hipSetDevice(batch->idx->device));
  double sum = 0.0;
  uint32_t cnt = 0;
  for (auto& ev : batch->events) {
    MGX_HIP(hipEventSynchronize(ev.second));
    float ms = 0.f;
    MGX_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
    sum += ms;
    ++cnt;
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  batch->events.clear();
  batch->timing = true;
  *avg_ms = cnt ? sum / cnt : 0.0;
  *n = cnt;
  return MGX_OK;
}

void mgx_batch_destroy(mgx_batch* batch) {
  if (!batch) return;
  (void)hipSetDevice(batch->idx->device);
  for (auto& ev : batch->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (batch->res_owned && batch->res_owned->main_ev) {  // (the index must not chain a later batch to a dead event)
    std::lock_guard<std::mutex> lock(batch->idx->table_mu);
    if (batch->idx->chain_ev == batch->res_owned->main_ev) batch->idx->chain_ev = nullptr;
  }
  delete batch;
}

// =================================================================================================================
// single operators
// =================================================================================================================

namespace {
// One set of batch resources (arenas, pinned result block, stream) leased from the index for the duration of a call.
struct SingleLease {
  mgx_index* idx;
  std::unique_ptr<mgx::BatchResources> res;
  explicit SingleLease(mgx_index* i) : idx(i) {
    {
      std::lock_guard<std::mutex> lock(idx->pool_mu);
      if (!idx->single_pool.empty()) {
        res = std::move(idx->single_pool.back());
        idx->single_pool.pop_back();
      }
    }
    if (!res) res = std::make_unique<mgx::BatchResources>();
    res->Reset();
  }
  ~SingleLease() {
    std::lock_guard<std::mutex> lock(idx->pool_mu);
    if (idx->single_pool.size() < 64) idx->single_pool.push_back(std::move(res));
  }
  hipError_t Stream(hipStream_t* s) {
    if (!res->stream) {
      const hipError_t e = hipStreamCreateWithFlags(&res->stream, hipStreamNonBlocking);
      if (e != hipSuccess) return e;
    }
    *s = res->stream;
    return hipSuccess;
  }
  // ships what Upload() staged in the lease's pinned mirror (calls that do not go through a batch's execute)
  hipError_t Ship(hipStream_t s) {
    for (const mgx::Arena::Chunk& c : res->upload.chunks)
      if (c.used) {
        const hipError_t e = hipMemcpyAsync(c.dev, c.host, c.used, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
      }
    res->uploaded = true;
    return hipSuccess;
  }
};
}  // namespace

static int RunSingle(mgx_index* idx, mgx::QuerySpec&& spec, uint32_t** out_docs, uint64_t* out_n) {
  *out_docs = nullptr;
  *out_n = 0;
  std::vector<mgx::QuerySpec> specs;
  specs.push_back(std::move(spec));
  MGX_HIP(hipSetDevice(idx->device));
  SingleLease lease(idx);
  hipStream_t stream = nullptr;
  MGX_HIP(lease.Stream(&stream));
  mgx_batch batch_obj;
  mgx_batch* b = &batch_obj;
  b->res = lease.res.get();
  int rc = mgx::PrepareInto(b, idx, std::move(specs));
  if (rc) return rc;
  rc = mgx::ExecuteImpl(b, stream);
  if (rc) return rc;
  mgx_result_view v{};
  rc = mgx::FetchImpl(b, &v);
  if (rc) return rc;
  const uint32_t n = v.queries[0].n_docs;
  uint32_t* o = static_cast<uint32_t*>(std::malloc((n ? n : 1) * sizeof(uint32_t)));
  if (!o) return mgx::Fail(MGX_ERR_INTERNAL, "out of host memory");
  if (n) std::memcpy(o, v.docs, n * sizeof(uint32_t));
  *out_docs = o;
  *out_n = n;
  return MGX_OK;
}

int mgx_facet_counts(mgx_index* idx, const mgx_query* query, uint32_t column_id, uint64_t* counts_out, uint64_t* matched) {
  if (matched) *matched = 0;
  if (!idx || !query || !counts_out || !matched)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_facet_counts: null argument");
  try {
    // the search part as a materialising query: the result bitmap of every tile, then a histogram of the column's value
    // ids over it (what roaring_bitmap_and_cardinality per value bitmap adds up to, filter_index.cpp:300-306)
    mgx_query q = *query;
    q.sort = MGX_SORT_DOCID;
    q.limit = 0;
    q.offset = 0;
    q.reverse = 0;
    q.score_terms = nullptr;
    q.n_score_terms = 0;
    mgx::QuerySpec spec;
    int rc = mgx::CompileQuery(idx, q, &spec);
    if (rc) return rc;
    const mgx_index::FilterColumn* colp = nullptr;
    {
      std::lock_guard<std::mutex> lock(idx->mu);  // (columns are added under mu; the objects themselves never move)
      if (column_id >= idx->filter_columns.size())
        return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_facet_counts: unknown filter column");
      colp = idx->filter_columns[column_id].get();
    }
    const mgx_index::FilterColumn& col = *colp;
    if (!col.d_value_ids.p) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_facet_counts: the column was added without value ids");
    std::vector<mgx::QuerySpec> specs;
    specs.push_back(std::move(spec));
    MGX_HIP(hipSetDevice(idx->device));
    SingleLease lease(idx);
    hipStream_t stream = nullptr;
    MGX_HIP(lease.Stream(&stream));
    mgx_batch batch_obj;
    mgx_batch* b = &batch_obj;
    b->res = lease.res.get();
    rc = mgx::PrepareInto(b, idx, std::move(specs));
    if (rc) return rc;
    rc = mgx::ExecuteImpl(b, stream);
    if (rc) return rc;
    if (b->bitmap.qids.size() != 1) return mgx::Fail(MGX_ERR_INTERNAL, "mgx_facet_counts: the query did not compile to a result bitmap");
    DevBuf d_counts;
    {
      mgx::ResourceScope none(nullptr);
      MGX_HIP(d_counts.Alloc(std::max<size_t>(col.n_values, 1) * sizeof(unsigned long long)));
    }
    MGX_HIP(hipMemsetAsync(d_counts.p, 0, d_counts.bytes, stream));
    MGX_LAUNCH(mgx::LaunchFacetCount(b->d_rbits.as<uint64_t>(), idx->dev.n_tiles * mgx::kWordsPerTile,
                                     col.d_value_ids.as<uint32_t>(), idx->dev.n_docs, col.n_values,
                                     d_counts.as<unsigned long long>(), stream));
    uint64_t total = 0;
    if (col.n_values)
      MGX_HIP(hipMemcpyAsync(counts_out, d_counts.p, static_cast<size_t>(col.n_values) * 8, hipMemcpyDeviceToHost, stream));
    MGX_HIP(hipMemcpyAsync(&total, b->d_totals.p, 8, hipMemcpyDeviceToHost, stream));
    MGX_HIP(hipStreamSynchronize(stream));
    *matched = total;
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_facet_counts: ") + e.what());
  }
}

static int CheckGrams(const mgx_index* idx, const uint32_t* g, uint32_t n, const char* what) {
  if (n && !g) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, std::string(what) + ": null gram ids");
  for (uint32_t i = 0; i < n; ++i)
    if (g[i] >= idx->n_grams) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, std::string(what) + ": unknown gram id");
  return MGX_OK;
}

#define MGX_SINGLE_PROLOGUE(name)                                                                    \
  if (out_docs) *out_docs = nullptr;                                                                 \
  if (out_n) *out_n = 0;                                                                             \
  if (!idx || !out_docs || !out_n) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, name ": null argument"); \
  {                                                                                                  \
    int rc_ = CheckGrams(idx, gram_ids, n, name);                                                    \
    if (rc_) return rc_;                                                                             \
  }

int mgx_and(mgx_index* idx, const uint32_t* gram_ids, uint32_t n, uint64_t limit, int reverse, uint32_t** out_docs,
            uint64_t* out_n) {
  MGX_SINGLE_PROLOGUE("mgx_and");
  if (n == 0) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_and: no grams (the caller returns {} for empty terms)");
  if (limit > 0xFFFFFFFFull) limit = 0;
  try {
    mgx::QuerySpec q;
    mgx::Compiler c{idx, &q, 0};
    c.Emit(mgx::kOpLoad, c.GramLeaf(gram_ids[0]));
    for (uint32_t i = 1; i < n; ++i) c.Emit(mgx::kOpAnd, c.GramLeaf(gram_ids[i]));
    if (q.leaves.size() > mgx::kMaxLeaves) return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED, "mgx_and: more than 192 lists");
    q.limit = static_cast<uint32_t>(limit);
    q.reverse = reverse ? 1 : 0;
    return RunSingle(idx, std::move(q), out_docs, out_n);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_and: ") + e.what());
  }
}

int mgx_or(mgx_index* idx, const uint32_t* gram_ids, uint32_t n, uint32_t** out_docs, uint64_t* out_n) {
  MGX_SINGLE_PROLOGUE("mgx_or");
  if (n == 0) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_or: no grams");
  try {
    mgx::QuerySpec q;
    mgx::Compiler c{idx, &q, 0};
    c.Emit(mgx::kOpLoad, c.GramLeaf(gram_ids[0]));
    for (uint32_t i = 1; i < n; ++i) c.Emit(mgx::kOpOr, c.GramLeaf(gram_ids[i]));
    if (q.leaves.size() > mgx::kMaxLeaves) return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED, "mgx_or: more than 192 lists");
    return RunSingle(idx, std::move(q), out_docs, out_n);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_or: ") + e.what());
  }
}

int mgx_not(mgx_index* idx, const uint32_t* all_docs, uint64_t n_all, const uint32_t* gram_ids, uint32_t n,
            uint32_t** out_docs, uint64_t* out_n) {
  MGX_SINGLE_PROLOGUE("mgx_not");
  if (n_all && !all_docs) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_not: null all_docs");
  if (n_all > 0xFFFFFFFFull) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_not: too many docs");
  try {
    mgx::QuerySpec q;
    mgx::Compiler c{idx, &q, 0};
    // ids outside the index range cannot be in any posting list: they pass through untouched, in order
    std::vector<uint32_t> below, above;
    const uint64_t lo = idx->dev.first_doc_id, hi = lo + idx->dev.n_docs;
    for (uint64_t i = 0; i < n_all; ++i) {
      if (i && all_docs[i] <= all_docs[i - 1])
        return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_not: all_docs must be strictly ascending");
      if (all_docs[i] < lo) below.push_back(all_docs[i]);
      else if (all_docs[i] >= hi) above.push_back(all_docs[i]);
      else q.explicit_ids.push_back(all_docs[i]);
    }
    mgx::DevLeaf lf{};
    lf.score_slot = mgx::kNoSlot;
    lf.row = mgx::kNoRow;
    lf.kind = mgx::kLeafExplicit;
    lf.a = 0;
    lf.b = static_cast<uint32_t>(q.explicit_ids.size());
    q.leaves.push_back(lf);
    c.Emit(mgx::kOpLoad, 0);
    for (uint32_t i = 0; i < n; ++i) c.Emit(mgx::kOpAndNot, c.GramLeaf(gram_ids[i]));
    if (q.leaves.size() > mgx::kMaxLeaves) return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED, "mgx_not: more than 192 lists");
    uint32_t* mid = nullptr;
    uint64_t n_mid = 0;
    int rc = RunSingle(idx, std::move(q), &mid, &n_mid);
    if (rc) return rc;
    if (below.empty() && above.empty()) {
      *out_docs = mid;
      *out_n = n_mid;
      return MGX_OK;
    }
    const uint64_t tot = below.size() + n_mid + above.size();
    uint32_t* o = static_cast<uint32_t*>(std::malloc((tot ? tot : 1) * 4));
    if (!o) {
      std::free(mid);
      return mgx::Fail(MGX_ERR_INTERNAL, "out of host memory");
    }
    std::copy(below.begin(), below.end(), o);
    std::copy(mid, mid + n_mid, o + below.size());
    std::copy(above.begin(), above.end(), o + below.size() + n_mid);
    std::free(mid);
    *out_docs = o;
    *out_n = tot;
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_not: ") + e.what());
  }
}

int mgx_threshold(mgx_index* idx, const uint32_t* gram_ids, uint32_t n, uint32_t threshold, uint32_t** out_docs,
                  uint64_t* out_n) {
  MGX_SINGLE_PROLOGUE("mgx_threshold");
  if (n == 0 || threshold == 0 || threshold > n)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_threshold: need 1 <= threshold <= number of distinct grams");
  if (n > 127) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_threshold: more than 127 grams");
  try {
    mgx::QuerySpec q;
    mgx::Compiler c{idx, &q, 0};
    c.Emit(mgx::kOpThreshBegin);
    for (uint32_t i = 0; i < n; ++i) c.Emit(mgx::kOpThreshAdd, c.GramLeaf(gram_ids[i]));
    c.Emit(mgx::kOpThreshEnd, threshold);
    if (q.leaves.size() != n) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_threshold: grams must be distinct");
    if (q.leaves.size() > mgx::kMaxLeaves)
      return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED, "mgx_threshold: more than 192 lists");
    return RunSingle(idx, std::move(q), out_docs, out_n);
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_threshold: ") + e.what());
  }
}

int mgx_retain(mgx_index* idx, const uint32_t* candidates, uint64_t n_cand, const uint32_t* gram_ids, uint32_t n,
               uint32_t** out_docs, uint64_t* out_n) {
  MGX_SINGLE_PROLOGUE("mgx_retain");
  if (n_cand && !candidates) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_retain: null candidates");
  try {
    MGX_HIP(hipSetDevice(idx->device));
    SingleLease lease(idx);
    hipStream_t stream = nullptr;
    MGX_HIP(lease.Stream(&stream));
    mgx::ResourceScope scope(lease.res.get());
    std::vector<uint8_t> keep(n_cand, 1);
    if (n_cand && n) {
      DevBuf d_c, d_g, d_k;
      MGX_HIP(mgx::Upload(d_c, candidates, n_cand));
      MGX_HIP(mgx::Upload(d_g, gram_ids, n));
      MGX_HIP(d_k.Alloc(n_cand));
      MGX_HIP(lease.Ship(stream));
      MGX_LAUNCH(mgx::LaunchRetain(idx->dev, d_c.as<uint32_t>(), n_cand, d_g.as<uint32_t>(), n, d_k.as<uint8_t>(),
                                   stream));
      MGX_HIP(hipMemcpyAsync(keep.data(), d_k.p, n_cand, hipMemcpyDeviceToHost, stream));
      MGX_HIP(hipStreamSynchronize(stream));
    }
    uint32_t* o = static_cast<uint32_t*>(std::malloc((n_cand ? n_cand : 1) * 4));
    if (!o) return mgx::Fail(MGX_ERR_INTERNAL, "out of host memory");
    uint64_t k = 0;
    for (uint64_t i = 0; i < n_cand; ++i)
      if (keep[i]) o[k++] = candidates[i];
    *out_docs = o;
    *out_n = k;
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_retain: ") + e.what());
  }
}

int mgx_score_documents(mgx_index* idx, const uint32_t* candidates, uint64_t n_cand, const uint32_t* gram_ids,
                        const double* idfs, uint32_t n_terms, double avg_doc_length, double k1, double b,
                        double* scores_out) {
  if (!idx || (n_cand && (!candidates || !scores_out)) || (n_terms && (!gram_ids || !idfs)))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_score_documents: null argument");
  if (!idx->can_score) return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED, "index was created without tf/doc_len columns");
  for (uint32_t i = 0; i < n_terms; ++i)
    if (gram_ids[i] != mgx::kNoRow && gram_ids[i] >= idx->n_grams)
      return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_score_documents: unknown gram id");
  if (n_cand == 0) return MGX_OK;
  try {
    MGX_HIP(hipSetDevice(idx->device));
    SingleLease lease(idx);
    hipStream_t stream = nullptr;
    MGX_HIP(lease.Stream(&stream));
    mgx::ResourceScope scope(lease.res.get());
    DevBuf d_c, d_g, d_i, d_s;
    MGX_HIP(mgx::Upload(d_c, candidates, n_cand));
    MGX_HIP(mgx::Upload(d_g, gram_ids, n_terms));
    MGX_HIP(mgx::Upload(d_i, idfs, n_terms));
    MGX_HIP(d_s.Alloc(n_cand * 8));
    MGX_HIP(lease.Ship(stream));
    MGX_LAUNCH(mgx::LaunchScoreCandidates(idx->dev, d_c.as<uint32_t>(), n_cand, d_g.as<uint32_t>(), d_i.as<double>(),
                                          n_terms, k1, b, avg_doc_length, d_s.as<double>(), stream));
    MGX_HIP(hipMemcpyAsync(scores_out, d_s.p, n_cand * 8, hipMemcpyDeviceToHost, stream));
    MGX_HIP(hipStreamSynchronize(stream));
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_score_documents: ") + e.what());
  }
}

int mgx_score_documents_text(mgx_index* idx, const uint32_t* candidates, uint64_t n_cand, const uint8_t* term_bytes,
                             const uint32_t* term_off, const double* idfs, uint32_t n_terms, double avg_doc_length,
                             double k1, double b, double* scores_out) {
  if (!idx || (n_cand && (!candidates || !scores_out)) || (n_terms && (!term_off || !idfs)))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_score_documents_text: null argument");
  if (!idx->can_score) return mgx::Fail(MGX_ERR_NOT_IMPLEMENTED, "index was created without tf/doc_len columns");
  if (!idx->dev.text) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_score_documents_text: no text attached");
  for (uint32_t i = 0; i < n_terms; ++i)
    if (term_off[i + 1] < term_off[i]) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "term offsets must ascend");
  if (n_terms && term_off[n_terms] > term_off[0] && !term_bytes)
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_score_documents_text: null term bytes");
  if (n_cand == 0) return MGX_OK;
  try {
    MGX_HIP(hipSetDevice(idx->device));
    SingleLease lease(idx);
    hipStream_t stream = nullptr;
    MGX_HIP(lease.Stream(&stream));
    mgx::ResourceScope scope(lease.res.get());
    DevBuf d_c, d_t, d_o, d_i, d_s;
    MGX_HIP(mgx::Upload(d_c, candidates, n_cand));
    MGX_HIP(mgx::Upload(d_t, term_bytes, n_terms ? term_off[n_terms] : 0, 16));
    MGX_HIP(mgx::Upload(d_o, term_off, n_terms + 1));
    MGX_HIP(mgx::Upload(d_i, idfs, n_terms));
    MGX_HIP(d_s.Alloc(n_cand * 8));
    MGX_HIP(lease.Ship(stream));
    MGX_LAUNCH(mgx::LaunchScoreCandidatesText(idx->dev, d_c.as<uint32_t>(), n_cand, d_t.as<uint8_t>(),
                                              d_o.as<uint32_t>(), d_i.as<double>(), n_terms, k1, b, avg_doc_length,
                                              d_s.as<double>(), stream));
    MGX_HIP(hipMemcpyAsync(scores_out, d_s.p, n_cand * 8, hipMemcpyDeviceToHost, stream));
    MGX_HIP(hipStreamSynchronize(stream));
    return MGX_OK;
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_score_documents_text: ") + e.what());
  }
}

int mgx_sort_by_score(mgx_index* idx, const uint32_t* results, const double* scores, uint64_t n, int descending,
                      uint32_t limit, uint32_t offset, uint32_t** out_docs, uint64_t* out_n) {
  if (out_docs) *out_docs = nullptr;
  if (out_n) *out_n = 0;
  if (!idx || !out_docs || !out_n || (n && (!results || !scores)))
    return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "mgx_sort_by_score: null argument");
  if (n > 0xFFFFFFFFull) return mgx::Fail(MGX_ERR_OUT_OF_RANGE, "mgx_sort_by_score: more than 2^32 - 1 entries");
  const uint64_t start = std::min<uint64_t>(offset, n);
  const uint64_t end = limit == 0 ? n : std::min<uint64_t>(start + limit, n);
  // (owned here until the call has succeeded: out-parameters stay NULL/0 on every failure path)
  std::unique_ptr<uint32_t, void (*)(void*)> page(
      static_cast<uint32_t*>(std::malloc(((end - start) ? (end - start) : 1) * 4)), std::free);
  uint32_t* o = page.get();
  if (!o) return mgx::Fail(MGX_ERR_INTERNAL, "out of host memory");
  auto succeed = [&]() {
    *out_docs = page.release();
    *out_n = end - start;
    return MGX_OK;
  };
  if (end == start) return succeed();
  const bool bounded_page = limit != 0 && static_cast<uint64_t>(offset) + limit <= mgx::kMaxNeeded;
  if (n > 4096 && !(bounded_page && n > 65536)) {
    // the full-sort fallback (result_sorter.cpp:661-716 sorts whatever it is given; deep OFFSETs are benchmarked,
    // docs/releases/v1.3.5.md:238): every pair sorted best first by the bitonic network, then the page is cut out
    try {
      MGX_HIP(hipSetDevice(idx->device));
      SingleLease lease(idx);
      hipStream_t stream = nullptr;
      MGX_HIP(lease.Stream(&stream));
      mgx::ResourceScope scope(lease.res.get());
      uint64_t n2 = 2048;
      while (n2 < n) n2 <<= 1;
      DevBuf d_r, d_s, d_k, d_d, d_o;
      MGX_HIP(mgx::Upload(d_r, results, n));
      MGX_HIP(mgx::Upload(d_s, scores, n));
      MGX_HIP(d_k.Alloc(n2 * 8));
      MGX_HIP(d_d.Alloc(n2 * 4));
      MGX_HIP(d_o.Alloc((end - start) * 4));
      MGX_HIP(hipMemsetAsync(d_k.as<uint64_t>() + n, 0, (n2 - n) * 8, stream));
      MGX_HIP(hipMemsetAsync(d_d.as<uint32_t>() + n, 0, (n2 - n) * 4, stream));
      MGX_HIP(lease.Ship(stream));
      MGX_LAUNCH(mgx::LaunchMakeSortKeys(d_r.as<uint32_t>(), d_s.as<double>(), n, descending, d_k.as<uint64_t>(),
                                         d_d.as<uint32_t>(), stream));
      MGX_LAUNCH(mgx::LaunchSortPairs(d_k.as<uint64_t>(), d_d.as<uint32_t>(), n2, stream));
      MGX_LAUNCH(mgx::LaunchSortPage(d_k.as<uint64_t>(), d_d.as<uint32_t>(), static_cast<uint32_t>(start),
                                     static_cast<uint32_t>(end), descending, d_o.as<uint32_t>(), nullptr, stream));
      MGX_HIP(hipMemcpyAsync(o, d_o.p, (end - start) * 4, hipMemcpyDeviceToHost, stream));
      MGX_HIP(hipStreamSynchronize(stream));
      return succeed();
    } catch (const std::exception& e) {
      return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_sort_by_score: ") + e.what());
    }
  }
  if (n > 65536) {
    // long arrays: per-wave top-(offset+limit) over strided shares, then the merge kernel of the batch path
    try {
      MGX_HIP(hipSetDevice(idx->device));
      SingleLease lease(idx);
      hipStream_t stream = nullptr;
      MGX_HIP(lease.Stream(&stream));
      mgx::ResourceScope scope(lease.res.get());
      const uint32_t needed = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(offset) + limit, n));
      uint32_t cap = 64;
      while (cap < needed) cap <<= 1;
      const uint32_t n_blocks = static_cast<uint32_t>(std::min<uint64_t>(256, (n + 16383) / 16384));
      const uint32_t n_lists = n_blocks * (mgx::kBlock / 64);
      mgx::DevQuery q{};
      q.needed = needed;
      q.cap = cap;
      q.limit = limit;
      q.offset = offset;
      q.descending = descending ? 1 : 0;
      const uint32_t zero = 0;
      DevBuf d_r, d_s, d_k, d_d, d_ck, d_cd, d_cn, d_q, d_id, d_o, d_ps, d_pn;
      MGX_HIP(mgx::Upload(d_r, results, n));
      MGX_HIP(mgx::Upload(d_s, scores, n));
      MGX_HIP(mgx::Upload(d_q, &q, 1));
      MGX_HIP(mgx::Upload(d_id, &zero, 1));
      MGX_HIP(d_k.Alloc(n * 8));
      MGX_HIP(d_d.Alloc(n * 4));
      MGX_HIP(d_ck.Alloc(static_cast<size_t>(n_lists) * needed * 8));
      MGX_HIP(d_cd.Alloc(static_cast<size_t>(n_lists) * needed * 4));
      MGX_HIP(d_cn.Alloc(static_cast<size_t>(n_lists) * 4));
      MGX_HIP(d_o.Alloc(static_cast<size_t>(limit) * 4));
      MGX_HIP(d_ps.Alloc(static_cast<size_t>(limit) * 8));
      MGX_HIP(d_pn.Alloc(4));
      MGX_HIP(lease.Ship(stream));
      MGX_LAUNCH(mgx::LaunchMakeSortKeys(d_r.as<uint32_t>(), d_s.as<double>(), n, descending, d_k.as<uint64_t>(),
                                         d_d.as<uint32_t>(), stream));
      MGX_LAUNCH(mgx::LaunchTopKScan(d_k.as<uint64_t>(), d_d.as<uint32_t>(), n, needed, cap, descending, n_blocks,
                                     d_ck.as<uint64_t>(), d_cd.as<uint32_t>(), d_cn.as<uint32_t>(), stream));
      MGX_LAUNCH(mgx::LaunchMergeTopK(d_q.as<mgx::DevQuery>(), d_id.as<uint32_t>(), 1, n_lists, d_ck.as<uint64_t>(),
                                      d_cd.as<uint32_t>(), d_cn.as<uint32_t>(), /*kq=*/0, /*kj=*/needed, /*dj=*/needed,
                                      /*cq=*/0, /*cj=*/1, nullptr, nullptr, nullptr, 0, d_o.as<uint32_t>(),
                                      d_ps.as<double>(), d_pn.as<uint32_t>(), limit, nullptr, nullptr, nullptr,
                                      stream));
      MGX_HIP(hipMemcpyAsync(o, d_o.p, (end - start) * 4, hipMemcpyDeviceToHost, stream));
      MGX_HIP(hipStreamSynchronize(stream));
      return succeed();
    } catch (const std::exception& e) {
      return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_sort_by_score: ") + e.what());
    }
  }
  try {
    MGX_HIP(hipSetDevice(idx->device));
    SingleLease lease(idx);
    hipStream_t stream = nullptr;
    MGX_HIP(lease.Stream(&stream));
    mgx::ResourceScope scope(lease.res.get());
    DevBuf d_r, d_s, d_k, d_d, d_o;
    MGX_HIP(mgx::Upload(d_r, results, n));
    MGX_HIP(mgx::Upload(d_s, scores, n));
    MGX_HIP(d_k.Alloc(n * 8));
    MGX_HIP(d_d.Alloc(n * 4));
    MGX_HIP(d_o.Alloc((end - start) * 4));
    MGX_HIP(lease.Ship(stream));
    MGX_LAUNCH(mgx::LaunchSortByScore(d_r.as<uint32_t>(), d_s.as<double>(), n, descending,
                                      static_cast<uint32_t>(start), static_cast<uint32_t>(end), d_k.as<uint64_t>(),
                                      d_d.as<uint32_t>(), d_o.as<uint32_t>(), stream));
    MGX_HIP(hipMemcpyAsync(o, d_o.p, (end - start) * 4, hipMemcpyDeviceToHost, stream));
    MGX_HIP(hipStreamSynchronize(stream));
    return succeed();
  } catch (const std::exception& e) {
    return mgx::Fail(MGX_ERR_INTERNAL, std::string("mgx_sort_by_score: ") + e.what());
  }
}

}  // extern "C"

extern "C" void mgxt_fail_device_allocs(int after, int count) {
  mgx::g_fail_count.store(count > 0 ? count : 0);
  mgx::g_fail_after.store(count > 0 && after >= 0 ? after : -1);
}

extern "C" int mgxt_measure_read_bandwidth(int device, uint64_t bytes, int iters, double* gb_per_s) {
  if (!gb_per_s || bytes < (1ull << 20) || iters < 1) return mgx::Fail(MGX_ERR_INVALID_ARGUMENT, "read probe: bad arguments");
  *gb_per_s = 0.0;
  MGX_HIP(hipSetDevice(device));
  bytes &= ~static_cast<uint64_t>(15);
  void* buf = nullptr;
  uint32_t* sink = nullptr;
  MGX_HIP(hipMalloc(&buf, bytes));
  if (hipMalloc(reinterpret_cast<void**>(&sink), 4) != hipSuccess) {
    (void)hipFree(buf);
    return mgx::Fail(MGX_ERR_INTERNAL, "read probe: hipMalloc failed");
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = MGX_OK;
  float best = 0.f;
  do {
    if (hipMemset(buf, 0x5A, bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess) {
      rc = mgx::Fail(MGX_ERR_INTERNAL, "read probe: setup failed");
      break;
    }
    for (int i = 0; i < iters + 1 && rc == MGX_OK; ++i) {  // first launch is a warm-up
      (void)hipEventRecord(e0, nullptr);
      if (mgx::LaunchReadProbe(buf, bytes, sink, nullptr) != 0) rc = mgx::Fail(MGX_ERR_INTERNAL, "read probe: launch failed");
      (void)hipEventRecord(e1, nullptr);
      if (hipEventSynchronize(e1) != hipSuccess) rc = mgx::Fail(MGX_ERR_INTERNAL, "read probe: kernel failed");
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (i > 0 && ms > 0.f && (best == 0.f || ms < best)) best = ms;
    }
  } while (false);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  (void)hipFree(buf);
  if (rc == MGX_OK && best > 0.f) *gb_per_s = static_cast<double>(bytes) / (best * 1e-3) / 1e9;
  return rc;
}
