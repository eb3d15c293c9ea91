// mgx_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the MygramDB query hot path.
//
// Kernel families (SURVEY.md §2.2 names in brackets):
//   build_tile_off_kernel / build_bitmap_kernel / build_tfnib_kernel   index-side precomputation
//   wave_score_kernel              [K1+K2+K5+K6+K7] the dominant kernel of SORT _score batches: 512-thread workgroups of
//                                  8 autonomous waves, each wave owns whole 16384-doc tiles (4 bitmap words per lane),
//                                  BM25 from doc-slot tf nibbles + LDS contribution tables, per-wave running top-k
//   tile_eval_kernel<mode>         [K1..K7] the general 256-thread workgroup kernel: any tile program (stack, bit-sliced
//                                  threshold counters, list-form scored terms, text-level terms, exact-text filter)
//   wave_count_kernel / wave_page_kernel   docid-ordered pages of flat programs (count pass, rank scan, page pass)
//   merge_topk_kernel              [K7/C1 merge] per-query merge of per-workgroup (or per-shard) sorted top-k lists
//   scan_tiles_kernel / expand_kernel  result bitmaps -> ascending/descending docid pages
//   retain_kernel                  [K4] Index::FilterByNgrams
//   score_candidates*_kernel       [K6] BM25Scorer::ScoreDocuments over an explicit candidate list
//
// All of them are HBM/LDS/issue-bound integer and fp64 work; none uses MFMA (nothing here is a dense contraction).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "mgx_internal.hpp"
#include "mgx_launch.hpp"

namespace mgx {

// Timing ablations (skip scoring / enumeration) exist only in -DMGX_ABLATION builds; the product library has no such
// switch in its hot loop.
#ifdef MGX_ABLATION
#define MGX_ABLATE(bt, bit) (((bt).debug_skip & (bit)) != 0)
#else
#define MGX_ABLATE(bt, bit) false
#endif

// ---------------------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// A value every lane of the wave holds identically, moved to scalar registers: loads through LDS or vector memory
// leave wave-uniform values in VGPRs, and the scoring kernel has none to spare.
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float wave_uniform(float v) {
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}
__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// LDS operations of one wave complete in issue order; this only has to stop the compiler from moving LDS accesses
// of other lanes' data across it (and drains the wave's own outstanding LDS traffic).
__device__ __forceinline__ void wave_lds_sync() { __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (no LDS traffic): Hillis-Steele inside each row of
// 16 lanes (row_shr 1,2,4,8; out-of-row sources read as 0), then lane 15 of each row is added to the next row
// (row_bcast:15, rows 1 and 3) and lane 31 to the upper half (row_bcast:31, rows 2 and 3).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// first index in [lo, hi) whose docid is >= x
__device__ __forceinline__ uint64_t lower_bound_u32(const uint32_t* __restrict__ a, uint64_t lo, uint64_t hi,
                                                    uint64_t x) {
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (static_cast<uint64_t>(a[mid]) < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// tf of posting p: the byte column, or — where the byte is saturated — the side table of true counts
// (CountTermOccurrences has no ceiling, bm25_scorer.cpp:27-45; a doc that repeats a gram 300 times must score as such)
__device__ __forceinline__ uint32_t posting_tf(const DevIndex& ix, uint64_t p) {
  uint32_t v = ix.tf[p];
  if (v == 255u && ix.n_tf_ovf != 0) {
    uint32_t lo = 0, hi = ix.n_tf_ovf;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (ix.tf_ovf_pos[mid] < p) lo = mid + 1; else hi = mid;
    }
    if (lo < ix.n_tf_ovf && ix.tf_ovf_pos[lo] == p) v = ix.tf_ovf_val[lo];
  }
  return v;
}

__device__ __forceinline__ uint64_t score_key(double s, bool descending) {
  uint64_t u = static_cast<uint64_t>(__double_as_longlong(s));
  u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // total order of IEEE doubles as unsigned integers
  return descending ? u : ~u;
}
__device__ __forceinline__ double key_score(uint64_t k, bool descending) {
  uint64_t u = descending ? k : ~k;
  u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
  return __longlong_as_double(static_cast<long long>(u));
}
// ResultSorter::SortByScore comparator (result_sorter.cpp:681-686) in "larger is better" form:
// DESC: higher score, then larger docid; ASC: lower score, then smaller docid (key and docid both flipped).
__device__ __forceinline__ bool better(uint64_t ka, uint32_t da, uint64_t kb, uint32_t db) {
  return ka > kb || (ka == kb && da > db);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS plan (shared with the host through PlanLds)
// ---------------------------------------------------------------------------------------------------------------

struct LdsOffsets {
  uint32_t bm, stack, pref, prog, seg_lo, seg_hi, leaf, scan_tot, match, tk_keys, tk_docs, misc, tpat, total;
};

__host__ __device__ inline uint32_t align8(uint32_t x) { return (x + 7u) & ~7u; }

__host__ __device__ inline LdsOffsets carve(const LdsPlan& p, bool score_mode) {
  LdsOffsets o;
  uint32_t at = 0;
  o.bm = at;        at += p.max_lds_leaves * kBlock * 8;
  o.stack = at;     at += p.max_stack * kBlock * 8;
  o.seg_lo = at;    at += align8(p.max_leaves * 8);
  o.seg_hi = at;    at += align8(p.max_leaves * 8);
  o.leaf = at;      at += align8(p.max_leaves * static_cast<uint32_t>(sizeof(DevLeaf)));
  o.prog = at;      at += align8(p.max_instr * 4);
  o.scan_tot = at;  at += 4 * 16 * 4;  // 4 waves x up to 16 packed scan lanes
  o.misc = at;      at += 128;  // page-pass masks (32 B), operand-presence mask (32 B), jump targets (8 B)
  o.pref = at;
  o.match = at;
  o.tk_keys = at;
  o.tk_docs = at;
  if (score_mode) {
    at += align8((1 + p.max_score) * kBlock * 2);
    o.match = at;    at += kMatchBuf * 2;
    o.tk_keys = at;  at += 4 * 2 * p.max_cap * 8;
    o.tk_docs = at;  at += 4 * 2 * p.max_cap * 4;
  }
  o.tpat = at;
  if (score_mode) at += p.max_score * 32;  // one TextPattern per scored term
  o.total = at;
  return o;
}

LdsPlan PlanLds(uint32_t max_leaves, uint32_t max_lds_leaves, uint32_t max_score, uint32_t max_stack, uint32_t max_instr,
                uint32_t max_cap, bool score_mode) {
  LdsPlan p{max_leaves ? max_leaves : 1, max_score, max_stack, max_instr ? max_instr : 1, max_cap,
            max_lds_leaves ? max_lds_leaves : 1, 0};
  p.bytes = carve(p, score_mode).total;
  return p;
}

// ---------------------------------------------------------------------------------------------------------------
// index-side precomputation
// ---------------------------------------------------------------------------------------------------------------

// tile_off[row][t] = number of postings of gram rows_gram[row] whose local slot is < t * kTileDocs.
__global__ void build_tile_off_kernel(const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ docids,
                                      const uint32_t* __restrict__ rows_gram, uint32_t n_rows, uint32_t n_tiles,
                                      uint32_t first_doc_id, uint32_t* __restrict__ tile_off) {
  uint64_t gid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  uint64_t total = static_cast<uint64_t>(n_rows) * (n_tiles + 1);
  if (gid >= total) return;
  uint32_t row = static_cast<uint32_t>(gid / (n_tiles + 1));
  uint32_t t = static_cast<uint32_t>(gid % (n_tiles + 1));
  uint32_t g = rows_gram[row];
  uint64_t lo = offsets[g], hi = offsets[g + 1];
  uint64_t x = static_cast<uint64_t>(first_doc_id) + static_cast<uint64_t>(t) * kTileDocs;
  tile_off[gid] = static_cast<uint32_t>(lower_bound_u32(docids, lo, hi, x) - lo);
}

// One workgroup-strided pass per row: bitmap[row][slot] = 1 for every posting of the row's doc list.
__global__ void build_bitmap_kernel(const uint32_t* __restrict__ docids, const uint64_t* __restrict__ row_lo,
                                    const uint64_t* __restrict__ row_hi, uint32_t first_doc_id, uint32_t row0,
                                    uint64_t tile_stride, uint64_t row_stride,
                                    unsigned long long* __restrict__ bitmaps) {
  const uint32_t row = blockIdx.y;
  const uint64_t lo = row_lo[row], hi = row_hi[row];
  unsigned long long* bm = bitmaps + static_cast<uint64_t>(row0 + row) * row_stride;
  for (uint64_t p = lo + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < hi;
       p += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint32_t slot = docids[p] - first_doc_id;
    atomicOr(&bm[static_cast<uint64_t>(slot >> kTileShift) * tile_stride + ((slot >> 6) & (kWordsPerTile - 1))],
             1ull << (slot & 63));
  }
}

// tf of the dense grams by DOC SLOT, 4 bits per doc: nib[row][slot/2] holds min(tf, 15) of slot in its low (even slot)
// or high (odd slot) nibble, 0 where the doc lacks the gram. The scoring kernel reads a match's tf from its slot
// directly — no rank, hence no prefix popcounts and no parked operand words.
__global__ void build_tfnib_kernel(const uint32_t* __restrict__ docids, const uint8_t* __restrict__ tf,
                                   const uint64_t* __restrict__ row_lo, const uint64_t* __restrict__ row_hi,
                                   uint32_t first_doc_id, uint64_t row_stride_bytes, uint8_t* __restrict__ nib) {
  const uint32_t row = blockIdx.y;
  const uint64_t lo = row_lo[row], hi = row_hi[row];
  uint32_t* out = reinterpret_cast<uint32_t*>(nib + static_cast<uint64_t>(row) * row_stride_bytes);
  for (uint64_t p = lo + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < hi;
       p += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint32_t slot = docids[p] - first_doc_id;
    const uint32_t v = tf[p] < 15 ? tf[p] : 15u;
    atomicOr(&out[slot >> 3], v << ((slot & 7u) * 4u));
  }
}

// Block-max of the BM25 term factor, for top-k pruning (the block-max idea of WAND-style retrieval on this layout):
// for every dense gram row and every 64-doc word of its bitmap, an upper bound of g(tf, dl) = tf * (k1 + 1) / (tf + K[dl])
// over the docs of the word that hold the gram, in units of `step` (one byte; 0 = no doc of the word holds the gram).
// A doc's score is sum_i idf_i * g_i, so sum_i idf_i * step * q_i bounds every score of the word from above; the scoring
// kernels skip the words whose bound is below the query's current k-th best (they still count their matches).
// q = floor(g / step * (1 + 2^-40)) + 1 > g / step even after the roundings of that product (2^-53 each), so q * step
// exceeds g by a relative 2^-41 at least — the margin the fp64 roundings of a doc's actual score (a few 2^-53) stay far
// inside; the kernels' side of the comparison is integer arithmetic (exact) with the weights rounded up and the
// threshold rounded down. (Until the bound went integer this was floor(g / step) + 2, a whole spare step per term: on
// the heavy queries of the benchmark that slack let 20 % more matches through.) A saturated nibble (tf >= 15) takes the supremum k1 + 1, a saturated doc length
// K[255] (the true length is longer, its factor smaller). Layout: [tile][row][256 words], like the gram bitmaps.
// kFine: the rows listed in `rows` (the densest grams: every 64-doc word of theirs holds some doc with a near-maximal
// factor, so the per-word bound prunes nothing) get one byte per 16-doc QUARTER of a word, four per word, as a u32.
template <bool kFine>
__global__ __launch_bounds__(256) void build_blockmax_kernel(const uint8_t* __restrict__ nib, uint64_t nib_row_stride,
                                                             const uint8_t* __restrict__ dl8,
                                                             const double* __restrict__ ktab, double k1_plus_1,
                                                             double inv_step, const uint32_t* __restrict__ rows,
                                                             uint32_t n_rows, uint64_t n_out, void* __restrict__ out) {
  const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n_out) return;
  const uint32_t word = static_cast<uint32_t>(idx & 255u);
  const uint64_t tr = idx >> 8;
  const uint32_t r = static_cast<uint32_t>(tr % n_rows);
  const uint32_t row = rows ? rows[r] : r;          // the bitmap / nibble row this output row describes
  const uint64_t gw = (tr / n_rows) * 256u + word;  // the word's index in the doc-slot space
  const uint4* np = reinterpret_cast<const uint4*>(nib + static_cast<uint64_t>(row) * nib_row_stride + gw * 32u);
  const uint4* dp = reinterpret_cast<const uint4*>(dl8 + gw * 64u);
  const uint4 n4[2] = {np[0], np[1]};
  const uint4 d4[4] = {dp[0], dp[1], dp[2], dp[3]};
  const uint32_t nw[8] = {n4[0].x, n4[0].y, n4[0].z, n4[0].w, n4[1].x, n4[1].y, n4[1].z, n4[1].w};
  const uint32_t dw[16] = {d4[0].x, d4[0].y, d4[0].z, d4[0].w, d4[1].x, d4[1].y, d4[1].z, d4[1].w,
                           d4[2].x, d4[2].y, d4[2].z, d4[2].w, d4[3].x, d4[3].y, d4[3].z, d4[3].w};
  double best[4] = {0.0, 0.0, 0.0, 0.0};  // per 16-doc quarter of the word
#pragma unroll
  for (int s = 0; s < 64; ++s) {
    const uint32_t tf = (nw[s >> 3] >> ((s & 7) * 4)) & 15u;
    if (tf == 0u) continue;
    const uint32_t d = (dw[s >> 2] >> ((s & 3) * 8)) & 255u;
    const double tfd = static_cast<double>(tf);
    const double gv = tf == 15u ? k1_plus_1 : tfd * k1_plus_1 / (tfd + ktab[d]);
    best[s >> 4] = gv > best[s >> 4] ? gv : best[s >> 4];
  }
  uint32_t packed = 0, whole = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t qv = 0;
    if (best[j] > 0.0) {
      const double steps = floor(best[j] * inv_step * (1.0 + 0x1p-40));
      qv = steps >= 254.0 ? 255u : static_cast<uint32_t>(steps) + 1u;
    }
    packed |= qv << (8 * j);
    whole = qv > whole ? qv : whole;
  }
  if (kFine) static_cast<uint32_t*>(out)[idx] = packed;
  else static_cast<uint8_t*>(out)[idx] = static_cast<uint8_t>(whole);
}

// ---------------------------------------------------------------------------------------------------------------
// tile kernel
// ---------------------------------------------------------------------------------------------------------------

// Scatter the postings [lo, hi) of one sorted list into a 16384-bit LDS bitmap. Loads are 16 B per lane; a lane
// merges its (up to 4, ascending) ids that fall in one 32-bit word before touching LDS, so a dense list costs about
// one ds_or per lane instead of four.
__device__ __forceinline__ void scatter_segment(const uint32_t* __restrict__ ids, uint64_t lo, uint64_t hi,
                                                uint32_t tile_first_doc, uint32_t* __restrict__ bm32) {
  uint64_t p0 = lo & ~3ull;
  for (uint64_t p = p0 + 4ull * threadIdx.x; p < hi; p += 4ull * kBlock) {
    const uint4 v = *reinterpret_cast<const uint4*>(ids + p);
    const uint32_t e[4] = {v.x, v.y, v.z, v.w};
    uint32_t curw = 0xFFFFFFFFu, curm = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint64_t q = p + j;
      if (q >= lo && q < hi) {
        uint32_t bit = e[j] - tile_first_doc;
        uint32_t w = bit >> 5, m = 1u << (bit & 31);
        if (w == curw) {
          curm |= m;
        } else {
          if (curm) atomicOr(&bm32[curw], curm);
          curw = w;
          curm = m;
        }
      }
    }
    if (curm) atomicOr(&bm32[curw], curm);
  }
}

// Per-wave running top-k in LDS: entries [0, cap) = best so far (sorted, best first, `have` valid),
// entries [cap, 2cap) = pending survivors. Unused entries are (0,0), which every real entry beats.
struct WaveTopK {
  uint64_t* keys;
  uint32_t* docs;
  uint32_t cap, needed;
  bool lds_sort;         // -DMGX_ABLATION builds: force the LDS bitonic truncation (A/B of the in-register one)
  uint32_t have, pend;   // wave-uniform
  uint64_t bound_key;    // valid when have >= needed
  uint32_t bound_doc;
  // Query-wide pruning bound shared by every wave working on the query: the best `needed`-th key any of them has
  // published so far. At least `needed` docs with a key >= it exist, so a candidate with a strictly smaller key can
  // never reach the page; equal keys are kept (docid decides). Stale reads only prune less.
  unsigned long long* gbound_ptr;
  uint64_t gbound;
};

// In-LDS bitonic sort (best first) of all 2*cap entries by the 64 lanes of one wave, then keep the first cap.
__device__ void wave_topk_truncate(WaveTopK& t) {
  const uint32_t n = 2 * t.cap;
  const int lane = lane_id();
  if (t.pend == 0) return;  // the kept region is already sorted (or empty)
  wave_lds_sync();
  for (uint32_t k = 2; k <= n; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = lane; i < n; i += 64) {
        uint32_t l = i ^ j;
        if (l > i) {
          uint64_t ki = t.keys[i], kl = t.keys[l];
          uint32_t di = t.docs[i], dl = t.docs[l];
          bool first_block = (i & k) == 0;  // this block sorts best-first, the mirrored one worst-first
          bool swap = first_block ? better(kl, dl, ki, di) : better(ki, di, kl, dl);
          if (swap) {
            t.keys[i] = kl; t.docs[i] = dl;
            t.keys[l] = ki; t.docs[l] = di;
          }
        }
      }
      wave_lds_sync();
    }
  }
  uint32_t total = t.have + t.pend;
  t.have = total < t.cap ? total : t.cap;
  t.pend = 0;
  for (uint32_t i = t.cap + lane; i < n; i += 64) {
    t.keys[i] = 0;
    t.docs[i] = 0;
  }
  wave_lds_sync();
  if (t.have >= t.needed) {
    t.bound_key = wave_uniform(t.keys[t.needed - 1]);
    t.bound_doc = wave_uniform(t.docs[t.needed - 1]);
    if (t.gbound_ptr && t.bound_key > t.gbound) {
      if (lane == 0) atomicMax(t.gbound_ptr, static_cast<unsigned long long>(t.bound_key));
      t.gbound = t.bound_key;
    }
  }
}

// wave_topk_truncate for cap == 64, in registers: the LDS version above sorts all 128 slots with ~28 compare-exchange
// stages of LDS round trips (~900 instructions), and every wave of every work item pays it at least twice (warm-up and
// the final list) — a third of the scoring kernel's instructions. Here the <= 64 pending entries are sorted one per
// lane with cross-lane moves (ds_swizzle / bpermute, no LDS memory), merged with the kept list by the bitonic
// "reverse, take the better, merge" step, and written back once: ~340 instructions.
__device__ __forceinline__ void lane_exchange(uint64_t& k, uint32_t& d, uint64_t pk, uint32_t pd, bool keep_better) {
  const bool mine_better = better(k, d, pk, pd);
  const bool keep_mine = keep_better == mine_better;
  k = keep_mine ? k : pk;
  d = keep_mine ? d : pd;
}
template <int J>
__device__ __forceinline__ void lane_partner(uint64_t k, uint32_t d, uint64_t* pk, uint32_t* pd) {
  uint32_t lo = static_cast<uint32_t>(k), hi = static_cast<uint32_t>(k >> 32), plo, phi;
  if (J < 32) {  // bit-mode swizzle: lane ^ J inside each half of the wave
    constexpr int pat = (J << 10) | 0x1F;
    plo = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(lo), pat));
    phi = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(hi), pat));
    *pd = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(d), pat));
  } else {
    plo = static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo), 32, 64));
    phi = static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi), 32, 64));
    *pd = static_cast<uint32_t>(__shfl_xor(static_cast<int>(d), 32, 64));
  }
  *pk = (static_cast<uint64_t>(phi) << 32) | plo;
}
template <int K, int J>
__device__ __forceinline__ void bitonic_step(uint64_t& k, uint32_t& d, uint32_t lane) {
  uint64_t pk;
  uint32_t pd;
  lane_partner<J>(k, d, &pk, &pd);
  const bool up = (lane & K) == 0, lower = (lane & J) == 0;  // K == 64: one best-first block
  lane_exchange(k, d, pk, pd, up == lower);
}
template <int K>
__device__ __forceinline__ void bitonic_merge(uint64_t& k, uint32_t& d, uint32_t lane) {
  if constexpr (K >= 64) bitonic_step<K, 32>(k, d, lane);
  if constexpr (K >= 32) bitonic_step<K, 16>(k, d, lane);
  if constexpr (K >= 16) bitonic_step<K, 8>(k, d, lane);
  if constexpr (K >= 8) bitonic_step<K, 4>(k, d, lane);
  if constexpr (K >= 4) bitonic_step<K, 2>(k, d, lane);
  bitonic_step<K, 1>(k, d, lane);
}

__device__ __forceinline__ void wave_topk_truncate64(WaveTopK& t) {
  const uint32_t lane = lane_id();
  if (t.pend == 0) return;
  wave_lds_sync();
  uint64_t k = lane < t.pend ? t.keys[64 + lane] : 0ull;
  uint32_t d = lane < t.pend ? t.docs[64 + lane] : 0u;
  bitonic_merge<2>(k, d, lane);
  bitonic_merge<4>(k, d, lane);
  bitonic_merge<8>(k, d, lane);
  bitonic_merge<16>(k, d, lane);
  bitonic_merge<32>(k, d, lane);
  bitonic_merge<64>(k, d, lane);  // lane i: the i-th best pending entry
  // kept list reversed against it: the better of each pair is the top 64 of the union, as a bitonic sequence
  const uint64_t rk = t.keys[63 - lane];
  const uint32_t rd = t.docs[63 - lane];
  if (better(rk, rd, k, d)) {
    k = rk;
    d = rd;
  }
  bitonic_merge<64>(k, d, lane);
  wave_lds_sync();
  t.keys[lane] = k;
  t.docs[lane] = d;
  const uint32_t total = t.have + t.pend;
  t.have = total < 64u ? total : 64u;
  t.pend = 0;
  wave_lds_sync();
  if (t.have >= t.needed) {
    t.bound_key = wave_uniform(t.keys[t.needed - 1]);
    t.bound_doc = wave_uniform(t.docs[t.needed - 1]);
    if (t.gbound_ptr && t.bound_key > t.gbound) {
      if (lane == 0) atomicMax(t.gbound_ptr, static_cast<unsigned long long>(t.bound_key));
      t.gbound = t.bound_key;
    }
  }
}

// the truncation that matches the list's capacity (the two keep different invariants about the pending region)
__device__ __forceinline__ void wave_topk_flush(WaveTopK& t) {
  if (t.cap == 64u && !t.lds_sort) wave_topk_truncate64(t); else wave_topk_truncate(t);
}

__device__ __forceinline__ void wave_topk_refresh_gbound(WaveTopK& t) {
  if (t.gbound_ptr) {
    const uint64_t g = wave_uniform(static_cast<uint64_t>(
        __hip_atomic_load(t.gbound_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    if (g > t.gbound) t.gbound = g;
  }
}

__device__ __forceinline__ void wave_topk_flush(WaveTopK& t);

// Offer one candidate per lane (valid=false for idle lanes). Wave-uniform control flow. kReg64: the list has cap == 64
// and is truncated in registers (wave_topk_truncate64); the kernels that were tuned around the LDS truncation keep it
// (inlining both into their loops cost them registers: wave_score_kernel went from 12 to 68 bytes of scratch).
template <bool kReg64 = false>
__device__ __forceinline__ void wave_topk_offer(WaveTopK& t, bool valid, uint64_t key, uint32_t doc) {
  bool surv = valid && key >= t.gbound && (t.have < t.needed || better(key, doc, t.bound_key, t.bound_doc));
  uint64_t mask = __ballot(surv);
  if (mask == 0) return;
  uint32_t ns = __popcll(mask);
  if (t.pend + ns > t.cap) {
    if (kReg64) wave_topk_flush(t); else wave_topk_truncate(t);
    surv = surv && key >= t.gbound && (t.have < t.needed || better(key, doc, t.bound_key, t.bound_doc));
    mask = __ballot(surv);
    if (mask == 0) return;
    ns = __popcll(mask);
  }
  if (surv) {
    uint32_t slot = t.cap + t.pend + __popcll(mask & ((1ull << lane_id()) - 1ull));
    t.keys[slot] = key;
    t.docs[slot] = doc;
  }
  t.pend += ns;
}

// BM25Scorer::CountTermOccurrences (bm25_scorer.cpp:27-45): non-overlapping occurrences of the pattern in one doc's
// text, found left to right; with first_only the scan stops at the first one (std::string::find != npos,
// search_pipeline.cpp:556-563).
//
// The doc text is streamed in aligned 16-byte chunks (the next chunk is in flight while the current one is scanned);
// for each of the 16 byte positions of a chunk an 8-byte window is cut out of the chunk registers with compile-time
// shifts and compared with the pattern's first 8 bytes (masked when the pattern is shorter). Positions that pass are
// taken in order under the non-overlap rule; only patterns longer than 8 bytes verify their tail byte by byte.
struct TextPattern {
  const uint8_t* bytes;   // the whole pattern
  uint32_t len;
  uint32_t lo, hi;        // its first 8 bytes, little-endian, zero beyond len
  uint32_t mlo, mhi;      // byte mask of those 8 bytes
};

__device__ __forceinline__ TextPattern text_pattern(const uint8_t* __restrict__ pat, uint32_t plen) {
  TextPattern p{pat, plen, 0, 0, 0, 0};
  for (uint32_t k = 0; k < 8 && k < plen; ++k) {
    const uint32_t v = pat[k];
    if (k < 4) {
      p.lo |= v << (8 * k);
      p.mlo |= 0xFFu << (8 * k);
    } else {
      p.hi |= v << (8 * (k - 4));
      p.mhi |= 0xFFu << (8 * (k - 4));
    }
  }
  return p;
}

__device__ __forceinline__ uint32_t text_count_occurrences(const uint8_t* __restrict__ text_base, uint64_t t0,
                                                           uint64_t t1, const TextPattern& pt, bool first_only) {
  const uint32_t plen = pt.len;
  if (t1 <= t0 || plen == 0 || plen > t1 - t0) return 0;
  const uint64_t last_start = t1 - plen;  // last byte offset an occurrence can start at
  uint64_t base = t0 & ~static_cast<uint64_t>(15);
  uint64_t next_ok = t0;
  uint32_t count = 0;
  uint4 c0 = *reinterpret_cast<const uint4*>(text_base + base);
  uint4 c1 = *reinterpret_cast<const uint4*>(text_base + base + 16);
  for (;;) {
    // the chunk after next is requested now and used two iterations later (the text buffer is padded for it)
    const uint4 c2 = *reinterpret_cast<const uint4*>(text_base + base + 32);
    const uint32_t d[7] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z};
    uint32_t cand = 0;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int i = p >> 2, sh = p & 3;
      const uint32_t wlo = sh ? __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh) : d[i];
      const uint32_t whi = sh ? __builtin_amdgcn_alignbyte(d[i + 2], d[i + 1], sh) : d[i + 1];
      const uint32_t diff = ((wlo ^ pt.lo) & pt.mlo) | ((whi ^ pt.hi) & pt.mhi);
      cand |= diff == 0 ? (1u << p) : 0u;
    }
    while (cand) {
      const uint32_t p = __builtin_ctz(cand);
      cand &= cand - 1;
      const uint64_t pos = base + p;
      if (pos < next_ok || pos > last_start) continue;
      bool hit = true;
      for (uint32_t k = 8; hit && k < plen; ++k) hit = text_base[pos + k] == pt.bytes[k];  // rare: long pattern
      if (hit) {
        ++count;
        if (first_only) return count;
        next_ok = pos + plen;
      }
    }
    base += 16;
    if (base > last_start) break;
    c0 = c1;
    c1 = c2;
  }
  return count;
}

__device__ __forceinline__ uint32_t exact_tf(const DevIndex& ix, uint32_t g, uint32_t row, uint32_t slot);

// ---- jumping over tiles that cannot matter (general kernel) ------------------------------------------------------------
// When every operand the program reads before its first COUNT is a sorted list (no bitmap-form operand, no slot range:
// nothing that is present in every tile), a tile can only matter if one of those lists has a posting in it: the workgroup
// goes from one such tile straight to the next (each list's thread knows its next posting's tile; one LDS min per step)
// instead of testing every tile of its item — the UNION bound. A second, usually tighter one: a list the accumulator cannot
// be non-empty without (an AND operand: REQUIRED) must have a posting in the tile, so the next tile is not before the
// latest of the required lists' next tiles — and that holds whatever else the program reads (a dense gram's bitmap beside a
// rare one).
constexpr uint32_t kJumpRef = 1u, kJumpReq = 2u, kJumpJumping = 4u, kJumpUnion = 8u, kJumpCheck = 16u;

// misc words: [16..19] jump words, u64 [10..13] required-operand mask. `stack` is the (still unused) operand stack region.
__device__ __noinline__ uint32_t tile_jump_setup(const uint32_t* prog, uint32_t n_instr, const DevLeaf* leaf, uint32_t n_leaves,
                                                 uint32_t* misc, uint64_t* stack, bool walked, bool has_off_row) {
  const uint32_t tid = threadIdx.x;
  bool listlike = false;
  if (tid < n_leaves) {
    const uint32_t kind = leaf[tid].kind;
    listlike = kind == kLeafList || kind == kLeafExplicit || kind == kLeafRange;
  }
  // (a query of bitmap-form operands only — the common text-level shape — pays this one barrier)
  if (__syncthreads_or(listlike ? 1 : 0) == 0) return 0;
  uint32_t* const jump_word = misc + 16;
  uint64_t* const req_mask = reinterpret_cast<uint64_t*>(misc) + 10;
  if (tid == 0) {
    // required operands: Load l -> {l}; AND adds, OR intersects, AND-NOT keeps the left side; a threshold term requires
    // its operands only when it asks for all of them. Sets are 4-word masks.
    uint64_t acc[4] = {0, 0, 0, 0}, ts[4] = {0, 0, 0, 0};
    uint32_t sp = 0, tcount = 0;
    for (uint32_t pc = 0; pc < n_instr; ++pc) {
      const uint32_t ins = prog[pc];
      const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
      if (op == kOpCount) break;
      const uint32_t w = (arg >> 6) & 3u;
      const uint64_t bit = 1ull << (arg & 63u);
      switch (op) {
        case kOpLoad: acc[0] = acc[1] = acc[2] = acc[3] = 0; acc[w] = bit; break;
        case kOpAnd: acc[w] |= bit; break;
        case kOpOr: {
          const bool had = acc[w] & bit;
          acc[0] = acc[1] = acc[2] = acc[3] = 0;
          if (had) acc[w] = bit;
          break;
        }
        case kOpPush:
          for (int k = 0; k < 4; ++k) stack[sp * 4 + k] = acc[k];
          ++sp;
          break;
        case kOpPopAnd: --sp; for (int k = 0; k < 4; ++k) acc[k] |= stack[sp * 4 + k]; break;
        case kOpPopOr: --sp; for (int k = 0; k < 4; ++k) acc[k] &= stack[sp * 4 + k]; break;
        case kOpPopAndNot: --sp; for (int k = 0; k < 4; ++k) acc[k] = stack[sp * 4 + k]; break;
        case kOpThreshBegin: ts[0] = ts[1] = ts[2] = ts[3] = 0; tcount = 0; break;
        case kOpThreshAdd: ts[w] |= bit; ++tcount; break;
        case kOpThreshEnd:
          for (int k = 0; k < 4; ++k) acc[k] = arg >= tcount ? ts[k] : 0;
          break;
        default: break;
      }
    }
    for (int k = 0; k < 4; ++k) req_mask[k] = acc[k];
    for (int k = 0; k < 4; ++k) jump_word[k] = (k & 1) ? 0u : 0xFFFFFFFFu;
  }
  __syncthreads();
  bool my_ref = false, my_req = false, blocks = false, absent_possible = false;
  if (tid < n_leaves) {
    for (uint32_t pc = 0; pc < n_instr; ++pc) {
      const uint32_t ins = prog[pc];
      const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
      if (op == kOpCount) break;
      if ((op == kOpLoad || op == kOpAnd || op == kOpOr || op == kOpAndNot || op == kOpThreshAdd) && arg == tid) my_ref = true;
    }
    const DevLeaf lf = leaf[tid];
    blocks = my_ref && (lf.kind == kLeafGramBitmap || lf.kind == kLeafFilterBitmap || (lf.kind == kLeafRange && lf.b > lf.a));
    my_req = ((req_mask[tid >> 6] >> (tid & 63u)) & 1ull) && (walked || (lf.kind == kLeafList && has_off_row));
    absent_possible = my_ref && (lf.kind == kLeafList || lf.kind == kLeafExplicit || lf.kind == kLeafRange);
  }
  const bool union_bound = __syncthreads_or(blocks ? 1 : 0) == 0;
  const bool jumping = union_bound || __syncthreads_or(my_req ? 1 : 0) != 0;
  const bool check = __syncthreads_or(absent_possible ? 1 : 0) != 0;
  return (my_ref ? kJumpRef : 0u) | (my_req ? kJumpReq : 0u) | (jumping ? kJumpJumping : 0u) | (union_bound ? kJumpUnion : 0u) |
         (check ? kJumpCheck : 0u);
}

// a lower bound (< tile_end, or tile_end: nothing left) of the first tile >= from that can matter; workgroup-uniform.
// `cur` is this thread's walked pointer (LDS), `ids` its id array (walked lists), `tile_off` its offset row (long lists).
__device__ __noinline__ uint32_t tile_jump_next(uint32_t flags, uint32_t from, uint32_t tile_end, const uint32_t* ids,
                                                uint64_t* cur, uint64_t l1, const uint32_t* tile_off, uint32_t first_doc_id,
                                                uint32_t* jump_word, uint32_t calls) {
  uint32_t mine = 0xFFFFFFFFu;  // the first tile >= from in which this thread's list has a posting
  if ((flags & (kJumpRef | kJumpReq)) && from < tile_end) {
    if (ids) {
      uint64_t a = *cur;
      const uint64_t from_doc = static_cast<uint64_t>(first_doc_id) + static_cast<uint64_t>(from) * kTileDocs;
      if (a < l1 && ids[a] < from_doc) {
        a = lower_bound_u32(ids, a, l1, from_doc);
        *cur = a;
      }
      if (a < l1) mine = (ids[a] - first_doc_id) >> kTileShift;
    } else if (tile_off) {
      uint32_t t = from;
      while (t < tile_end && tile_off[t + 1] == tile_off[t]) ++t;
      mine = t;
    }
    if (mine > tile_end) mine = tile_end;
  }
  uint32_t* const w = jump_word + 2u * (calls & 1u);
  if ((flags & kJumpUnion) && (flags & kJumpRef) && mine < tile_end) atomicMin(w, mine);
  if (flags & kJumpReq) atomicMax(w + 1, mine);  // (tile_end: this required list has nothing left — nor has the query)
  __syncthreads();
  uint32_t r = (flags & kJumpUnion) ? w[0] : from;
  if (w[1] > r) r = w[1];
  if (threadIdx.x == 0) {  // (the other pair is next call's: nobody touches it before a barrier)
    jump_word[2u * ((calls + 1u) & 1u)] = 0xFFFFFFFFu;
    jump_word[2u * ((calls + 1u) & 1u) + 1] = 0u;
  }
  return r < tile_end ? r : tile_end;
}

// Can the query's program reach its first COUNT with a non-empty accumulator in a tile where only the operands marked in
// `ne` (bit l = leaf l has a posting / a set range in this tile; bitmap-form operands always count as present) hold
// anything? Evaluated on one boolean per operand: AND needs both sides, OR either, AND-NOT leaves the left side, a
// threshold term needs that many present operands. Everything after the first COUNT only narrows the accumulator, so a
// tile that fails here contributes zero to every counter, to the result and to the page: it is skipped before any
// posting is scattered. (A sparse-gram index — CJK trigrams: a handful of postings per gram — fails in almost every
// tile; the general kernel used to stage and evaluate all of them: 8.4 ms of a 9.4 ms batch of BASELINE configs[2].)
__device__ __forceinline__ bool tile_may_match(const uint32_t* prog, uint32_t n_instr, const uint64_t* ne) {
  bool acc = false;
  uint64_t stk = 0;
  uint32_t sp = 0, cnt = 0;
  auto NE = [&](uint32_t l) -> bool { return (ne[l >> 6] >> (l & 63u)) & 1ull; };
  for (uint32_t pc = 0; pc < n_instr; ++pc) {
    const uint32_t ins = prog[pc];
    const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
    switch (op) {
      case kOpLoad: acc = NE(arg); break;
      case kOpAnd: acc = acc && NE(arg); break;
      case kOpOr: acc = acc || NE(arg); break;
      case kOpPush: stk = (stk & ~(1ull << sp)) | (static_cast<uint64_t>(acc) << sp); ++sp; break;
      case kOpPopAnd: --sp; acc = ((stk >> sp) & 1ull) && acc; break;
      case kOpPopOr: --sp; acc = ((stk >> sp) & 1ull) || acc; break;
      case kOpPopAndNot: --sp; acc = (stk >> sp) & 1ull; break;
      case kOpThreshBegin: cnt = 0; break;
      case kOpThreshAdd: cnt += NE(arg) ? 1u : 0u; break;
      case kOpThreshEnd: acc = cnt >= arg; break;
      case kOpCount: return acc;
      default: break;  // AND-NOT, text verification: the accumulator can only shrink
    }
  }
  return acc;
}

// JUMP: the query has an operand that can be absent from a tile (a sorted list, an explicit id list, a slot range): tiles
// are tested / jumped over as described above. Queries of bitmap-form operands only run the plain instantiation — the
// same tile loop without that machinery (its registers cost the text-level scoring path 6 % when they shared one kernel).
template <int MODE, bool JUMP>
__device__ __forceinline__ void tile_eval_body(const DevIndex& ix, const DevBatch& bt, const LdsPlan& plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const LdsOffsets lo_ = carve(plan, MODE == kModeScore || MODE == kModeTextDf);
  uint64_t* const bm64 = reinterpret_cast<uint64_t*>(smem + lo_.bm);
  uint64_t* const stack = reinterpret_cast<uint64_t*>(smem + lo_.stack);
  uint64_t* const seg_lo = reinterpret_cast<uint64_t*>(smem + lo_.seg_lo);
  uint64_t* const seg_hi = reinterpret_cast<uint64_t*>(smem + lo_.seg_hi);
  DevLeaf* const leaf = reinterpret_cast<DevLeaf*>(smem + lo_.leaf);
  uint32_t* const prog = reinterpret_cast<uint32_t*>(smem + lo_.prog);
  uint32_t* const scan_tot = reinterpret_cast<uint32_t*>(smem + lo_.scan_tot);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + lo_.misc);
  uint16_t* const pref = reinterpret_cast<uint16_t*>(smem + lo_.pref);
  uint16_t* const matchbuf = reinterpret_cast<uint16_t*>(smem + lo_.match);

  const uint32_t tid = threadIdx.x;
  // items are ordered by doc range first: neighbouring workgroups walk the same tiles for different queries, so
  // shared lists / doc_len hit in L2
  // the page pass has one workgroup per query and finds its own tile range; every other mode walks a host-made item
  DevItem it = bt.items[blockIdx.x];  // (page pass: only .query is meaningful)
  const uint32_t qi = it.query;
  const DevQuery q = bt.queries[qi];
  if (q.mode != MODE && !(MODE == kModeDocCount && q.mode == kModeDocPage)) return;
  if (MODE == kModeDocPage) {
    // tiles that hold ranks [lo, hi) of the matches in doc order: tile_start is non-decreasing, so two binary searches
    const uint64_t total = bt.totals[q.out_slot];
    const uint64_t take = total < q.limit ? total : q.limit;
    if (take == 0) return;
    const uint64_t lo = q.descending ? total - take : 0, hi = q.descending ? total : take;
    const uint64_t* ts = bt.tile_start + static_cast<uint64_t>(q.out_slot) * ix.n_tiles;
    uint32_t a = 0, b = ix.n_tiles;  // last tile whose start rank is <= lo
    while (b - a > 1) {
      const uint32_t mid = (a + b) >> 1;
      if (ts[mid] <= lo) a = mid; else b = mid;
    }
    uint32_t c = a, d = ix.n_tiles;  // first tile past `a` whose start rank is >= hi
    while (c < d) {
      const uint32_t mid = (c + d) >> 1;
      if (ts[mid] >= hi) d = mid; else c = mid + 1;
    }
    it.tile_begin = a;
    it.n_tiles = max(c, a + 1) - a;
    // a sparse result's page spans up to `limit` tiles, one full tile evaluation each: the query's page pass is shared by
    // gridDim.y workgroups (each takes a slice of the tile range; ranks are global, so their outputs do not meet)
    if (gridDim.y > 1) {
      const uint32_t per = (it.n_tiles + gridDim.y - 1) / gridDim.y;
      const uint32_t skip = blockIdx.y * per;
      if (skip >= it.n_tiles) return;
      it.tile_begin += skip;
      it.n_tiles = min(per, it.n_tiles - skip);
    }
  }

  const uint32_t n_leaves = q.n_leaves;
  for (uint32_t i = tid; i < n_leaves; i += kBlock) leaf[i] = bt.leaves[q.leaf_begin + i];
  for (uint32_t i = tid; i < q.n_instr; i += kBlock) prog[i] = bt.prog[q.prog_begin + i];

  WaveTopK tk;
  if (MODE == kModeScore) {
    tk.cap = q.cap;
    tk.needed = q.needed;
    tk.lds_sort = false;
    tk.keys = reinterpret_cast<uint64_t*>(smem + lo_.tk_keys) + static_cast<size_t>(wave_id()) * 2 * q.cap;
    tk.docs = reinterpret_cast<uint32_t*>(smem + lo_.tk_docs) + static_cast<size_t>(wave_id()) * 2 * q.cap;
    tk.have = 0;
    tk.pend = 0;
    tk.bound_key = 0;
    tk.bound_doc = 0;
    tk.gbound_ptr = bt.bounds ? bt.bounds + qi : nullptr;
    tk.gbound = 0;
    for (uint32_t i = lane_id(); i < 2 * q.cap; i += 64) {
      tk.keys[i] = 0;
      tk.docs[i] = 0;
    }
  }
  uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0, cnt_res = 0, cnt_df = 0;
  const uint32_t n_score = MODE == kModeScore ? q.n_score : 0u;
  TextPattern df_pattern{nullptr, 0, 0, 0, 0, 0};
  if (MODE == kModeTextDf) df_pattern = text_pattern(bt.patterns + q.pat_off, q.pat_len);
  static_assert(sizeof(TextPattern) == 32, "LDS plan reserves 32 bytes per text pattern");
  TextPattern* const tpat = reinterpret_cast<TextPattern*>(smem + lo_.tpat);
  if (MODE == kModeScore && tid < n_score) {
    const DevScoreTerm st = bt.score_terms[q.score_begin + tid];
    if (st.leaf == kNoLeaf) {
      const DevTextTerm tt = bt.text_terms[st.text_term];
      tpat[tid] = text_pattern(bt.patterns + tt.pat_off, tt.pat_len);
    }
  }

  const uint32_t tile_begin = it.tile_begin;
  const uint32_t tile_end = min(tile_begin + it.n_tiles, ix.n_tiles);
  __syncthreads();

  // Page pass: only tiles with matches are visited. A sparse query's page spans hundreds of tiles, almost all empty;
  // their counts are read 256 at a time into four ballot masks instead of one dependent load per tile.
  uint64_t* const pmask = reinterpret_cast<uint64_t*>(misc);
  uint32_t loaded_chunk = 0xFFFFFFFFu;
  auto next_tile = [&](uint32_t from) -> uint32_t {  // first tile >= from to visit (workgroup-uniform)
    if (MODE != kModeDocPage) return from;
    while (from < tile_end) {
      const uint32_t ck = (from - tile_begin) >> 8;
      if (ck != loaded_chunk) {
        __syncthreads();
        const uint32_t t = tile_begin + ck * 256 + tid;
        const uint32_t c = t < tile_end ? bt.tile_cnt[static_cast<uint64_t>(q.out_slot) * ix.n_tiles + t] : 0u;
        const uint64_t m = __ballot(c != 0);
        if (lane_id() == 0) pmask[wave_id()] = m;
        __syncthreads();
        loaded_chunk = ck;
      }
      const uint32_t rel = (from - tile_begin) & 255u;
      for (uint32_t w = rel >> 6; w < 4; ++w) {
        uint64_t m = pmask[w];
        if (w == (rel >> 6)) m &= ~0ull << (rel & 63u);
        if (m) return tile_begin + ck * 256 + w * 64 + static_cast<uint32_t>(__builtin_ctzll(m));
      }
      from = tile_begin + (ck + 1) * 256;
    }
    return tile_end;
  };
  // What thread l < n_leaves needs about ITS leaf in every tile, fetched once: the posting range of the gram and its
  // per-tile offset row (three dependent loads in front of a barrier, per tile, became two independent ones).
  uint64_t my_l0 = 0, my_l1 = 0;
  const uint32_t* my_tile_off = nullptr;
  // A list without a per-tile offset row (short) and an explicit id list are WALKED: `my_cur` is the first entry not before
  // the tile being visited — one load tells that a tile holds none of them (two binary searches of the whole list per
  // tile before).
  // (the pointer lives in seg_hi[tid] — step A leaves the tile's end there anyway —, the id array follows from the kind:
  //  registers are what keeps four waves per SIMD resident here)
  const uint32_t* my_ids = nullptr;
  uint64_t* const ne_mask = reinterpret_cast<uint64_t*>(misc) + 4;  // 4 words: operands present in the tile (above)
  if (tid < n_leaves) {
    const DevLeaf lf = leaf[tid];
    if (lf.kind == kLeafList || lf.kind == kLeafGramBitmap) {
      my_l0 = ix.offsets[lf.a];
      my_l1 = ix.offsets[lf.a + 1];
      const uint32_t row = ix.skip_row[lf.a];
      if (row != kNoRow) my_tile_off = ix.tile_off + static_cast<uint64_t>(row) * (ix.n_tiles + 1);
      else if (lf.kind == kLeafList) my_ids = ix.docids;
    } else if (lf.kind == kLeafExplicit) {
      my_ids = bt.explicit_pool;
      my_l0 = lf.a;
      my_l1 = static_cast<uint64_t>(lf.a) + lf.b;
    }
    seg_hi[tid] = my_l0;
  }
  // JUMPING (tile_jump_setup / tile_jump_next above the kernel; out of line, so that the queries that never jump keep the
  // register budget of the tile loop).
  uint32_t* const jump_word = reinterpret_cast<uint32_t*>(misc) + 16;  // [2][2]: min of the referenced, max of the required
  uint32_t jflags = 0, jump_calls = 0;
  if (JUMP && MODE != kModeDocPage)
    jflags = tile_jump_setup(prog, q.n_instr, leaf, n_leaves, misc, stack, my_ids != nullptr, my_tile_off != nullptr);
  const bool jumping = JUMP && (jflags & kJumpJumping) != 0, check_tiles = JUMP && (jflags & kJumpCheck) != 0;
  auto next_candidate = [&](uint32_t from) -> uint32_t {
    const uint32_t r = tile_jump_next(jflags, from, tile_end, my_ids, seg_hi + tid, my_l1, my_tile_off, ix.first_doc_id,
                                      jump_word, jump_calls);
    ++jump_calls;
    return r;
  };
  // tiles [from, to) are not visited: their outputs are zero
  auto zero_skipped = [&](uint32_t from, uint32_t to) {
    if (MODE == kModeBitmap)
      for (uint32_t t = from; t < to; ++t)
        bt.rbits[(static_cast<uint64_t>(q.out_slot) * ix.n_tiles + t) * kWordsPerTile + tid] = 0;
    if (MODE == kModeBitmap || MODE == kModeDocCount)
      for (uint32_t t = from + tid; t < to; t += kBlock) bt.tile_cnt[static_cast<uint64_t>(q.out_slot) * ix.n_tiles + t] = 0;
  };
  auto advance = [&](uint32_t done) -> uint32_t {  // the tile to visit after `done`
    if (!jumping) return next_tile(done + 1);
    const uint32_t nt = next_candidate(done + 1);
    zero_skipped(done + 1, nt);
    return nt;
  };
  uint32_t first_tile = next_tile(tile_begin);
  if (jumping) {
    first_tile = next_candidate(tile_begin);
    zero_skipped(tile_begin, first_tile);
  }
  for (uint32_t tile = first_tile; tile < tile_end; tile = advance(tile)) {
    const uint64_t tile_first = static_cast<uint64_t>(ix.first_doc_id) + static_cast<uint64_t>(tile) * kTileDocs;
    if (MODE == kModeScore) wave_topk_refresh_gbound(tk);
    // page pass: ranks [page_lo, page_hi) of the query's matches in doc order are wanted (ascending: the first `limit`,
    // descending: the last `limit`); a tile whose rank range misses the page is skipped before any operand is read
    uint64_t page_lo = 0, page_hi = 0, page_total = 0, tile_rank0 = 0;
    if (MODE == kModeDocPage) {
      page_total = bt.totals[q.out_slot];
      const uint64_t take = page_total < q.limit ? page_total : q.limit;
      page_lo = q.descending ? page_total - take : 0;
      page_hi = q.descending ? page_total : take;
      tile_rank0 = bt.tile_start[static_cast<uint64_t>(q.out_slot) * ix.n_tiles + tile];
      const uint32_t c = bt.tile_cnt[static_cast<uint64_t>(q.out_slot) * ix.n_tiles + tile];
      if (c == 0 || tile_rank0 >= page_hi || tile_rank0 + c <= page_lo) continue;  // workgroup-uniform
    }

    // ---- A. operand setup: segment bounds, and bitmaps that need no scatter -----------------------------------
    bool present = false;
    if (tid < n_leaves) {
      const DevLeaf lf = leaf[tid];
      uint64_t a = 0, b = 0;
      present = true;  // (bitmap-form operands: a dense gram, a filter row)
      if (my_ids) {    // walked list: entries of this tile are [a, b), the pointer (seg_hi[tid]) moves on to b
        a = seg_hi[tid];
        if (a < my_l1 && my_ids[a] < tile_first) a = lower_bound_u32(my_ids, a, my_l1, tile_first);  // (tiles were skipped)
        b = a;
        if (a < my_l1 && my_ids[a] < tile_first + kTileDocs) b = lower_bound_u32(my_ids, a + 1, my_l1, tile_first + kTileDocs);
        present = b > a;
      } else if (lf.kind == kLeafList) {
        a = my_l0 + my_tile_off[tile];
        b = my_l0 + my_tile_off[tile + 1];
        present = b > a;
      } else if (lf.kind == kLeafGramBitmap) {
        // rank base for tf lookups: postings of the gram before this tile
        a = my_l0 + (my_tile_off ? my_tile_off[tile] : 0u);
      } else if (lf.kind == kLeafRange) {
        const uint64_t s0 = static_cast<uint64_t>(tile) * kTileDocs;
        present = lf.b > lf.a && lf.b > s0 && lf.a < s0 + kTileDocs;
      }
      seg_lo[tid] = a;
      seg_hi[tid] = b;
    }
    if (check_tiles) {
      const uint64_t m = __ballot(present);
      if (lane_id() == 0) ne_mask[wave_id()] = m;
      __syncthreads();
    }
    if (check_tiles && !tile_may_match(prog, q.n_instr, ne_mask)) {  // workgroup-uniform
      if (MODE == kModeBitmap)
        bt.rbits[(static_cast<uint64_t>(q.out_slot) * ix.n_tiles + tile) * kWordsPerTile + tid] = 0;
      if ((MODE == kModeBitmap || MODE == kModeDocCount) && tid == 0)
        bt.tile_cnt[static_cast<uint64_t>(q.out_slot) * ix.n_tiles + tile] = 0;
      __syncthreads();  // (the next tile rewrites the mask)
      continue;
    }
    // this thread's 64-bit word of a bitmap-form operand (dense gram, filter, slot range), straight from HBM
    auto direct_word = [&](const DevLeaf lf) -> uint64_t {
      if (lf.kind == kLeafGramBitmap) return ix.gram_bitmaps[tile * ix.gb_tile_stride + lf.b * ix.gb_row_stride + tid];
      if (lf.kind == kLeafFilterBitmap) return ix.filter_bitmaps[tile * ix.fb_tile_stride + lf.b * ix.fb_row_stride + tid];
      if (lf.kind == kLeafRange) {
        // slots [a, b) of the shard; this thread's word covers slots [s0, s0+64)
        const uint64_t s0 = static_cast<uint64_t>(tile) * kTileDocs + static_cast<uint64_t>(tid) * 64;
        const uint64_t ra = lf.a > s0 ? lf.a - s0 : 0;
        const uint64_t rb = lf.b > s0 ? lf.b - s0 : 0;
        const uint64_t hi_mask = rb >= 64 ? ~0ull : ((1ull << rb) - 1ull);
        const uint64_t lo_mask = ra >= 64 ? ~0ull : ((1ull << ra) - 1ull);
        return hi_mask & ~lo_mask;
      }
      return 0;
    };
    // Only operands that need it live in LDS: sorted lists (scattered there) and scored operands (probed by other
    // threads); every other operand is one global load per thread at the instruction that uses it.
    for (uint32_t l = 0; l < n_leaves; ++l) {
      const DevLeaf lf = leaf[l];
      if (lf.lds != kNoRow) bm64[lf.lds * kBlock + tid] = direct_word(lf);
    }
    __syncthreads();

    // ---- B. scatter sorted segments into their bitmaps ---------------------------------------------------------
    for (uint32_t l = 0; l < n_leaves; ++l) {
      const uint32_t kind = leaf[l].kind;
      if (kind == kLeafList) {
        scatter_segment(ix.docids, seg_lo[l], seg_hi[l], static_cast<uint32_t>(tile_first),
                        reinterpret_cast<uint32_t*>(bm64 + leaf[l].lds * kBlock));
      } else if (kind == kLeafExplicit) {
        scatter_segment(bt.explicit_pool, seg_lo[l], seg_hi[l], static_cast<uint32_t>(tile_first),
                        reinterpret_cast<uint32_t*>(bm64 + leaf[l].lds * kBlock));
      }
    }
    __syncthreads();

    // ---- C. evaluate the query's program on this thread's 64-bit word ------------------------------------------
    uint64_t acc = 0;
    {
      uint32_t sp = 0;
      uint64_t cs[7] = {0, 0, 0, 0, 0, 0, 0};  // bit-sliced per-doc counters (one bit lane per doc slot)
      auto W = [&](uint32_t l) -> uint64_t {
        const DevLeaf lf = leaf[l];
        return lf.lds != kNoRow ? bm64[lf.lds * kBlock + tid] : direct_word(lf);
      };
      for (uint32_t pc = 0; pc < q.n_instr; ++pc) {
        const uint32_t ins = prog[pc];
        const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
        switch (op) {
          case kOpLoad: acc = W(arg); break;
          case kOpAnd: acc &= W(arg); break;
          case kOpOr: acc |= W(arg); break;
          case kOpAndNot: acc &= ~W(arg); break;
          case kOpPush: stack[sp * kBlock + tid] = acc; ++sp; break;
          case kOpPopAnd: --sp; acc = stack[sp * kBlock + tid] & acc; break;
          case kOpPopOr: --sp; acc = stack[sp * kBlock + tid] | acc; break;
          case kOpPopAndNot: --sp; acc = stack[sp * kBlock + tid] & ~acc; break;
          case kOpCount: {
            const uint32_t pcnt = __popcll(acc);
            cnt0 += (arg & 1u) ? pcnt : 0;
            cnt1 += (arg & 2u) ? pcnt : 0;
            cnt2 += (arg & 4u) ? pcnt : 0;
            cnt3 += (arg & 8u) ? pcnt : 0;
            break;
          }
          case kOpThreshBegin:
#pragma unroll
            for (int b = 0; b < 7; ++b) cs[b] = 0;
            break;
          case kOpThreshAdd: {
            uint64_t carry = W(arg);
#pragma unroll
            for (int b = 0; b < 7; ++b) {
              const uint64_t t = cs[b] & carry;
              cs[b] ^= carry;
              carry = t;
            }
            break;
          }
          case kOpThreshEnd: {
            // per-bit-lane compare of the 7-bit counters with the constant arg: ge = (count >= arg)
            uint64_t gt = 0, eq = ~0ull;
#pragma unroll
            for (int b = 6; b >= 0; --b) {
              const uint64_t tb = ((arg >> b) & 1u) ? ~0ull : 0ull;
              gt |= eq & cs[b] & ~tb;
              eq &= ~(cs[b] ^ tb);
            }
            acc = gt | eq;
            break;
          }
          case kOpVerifyText: {
            // text filter: a doc stays only if its text contains every pattern of the instruction (rare query shapes only —
            // the exact-text post-filter of mixed-script terms / verify_text, a term shorter than one n-gram; one doc at a time)
            uint64_t bits = acc;
            while (bits) {
              const uint32_t bpos = __builtin_ctzll(bits);
              bits &= bits - 1;
              const uint32_t slot = tile * kTileDocs + tid * 64 + bpos;
              const uint64_t t0 = ix.text_off[slot], t1 = ix.text_off[slot + 1];
              bool all = true;
              const uint32_t vfirst = arg & 0xFFFu, vcount = arg >> 12;  // the patterns this instruction checks
              for (uint32_t v = 0; all && v < vcount; ++v) {
                const DevTextTerm vt = bt.verify_terms[q.vt_begin + vfirst + v];
                all = text_count_occurrences(ix.text, t0, t1, text_pattern(bt.patterns + vt.pat_off, vt.pat_len), true) != 0;
              }
              if (!all) acc &= ~(1ull << bpos);
            }
            break;
          }
          default: break;
        }
      }
    }
    const uint32_t my_cnt = __popcll(acc);
    cnt_res += my_cnt;

    if (MODE == kModeDocPage) {
      const uint32_t inc = wave_incl_scan(my_cnt);
      if (lane_id() == 63) scan_tot[wave_id()] = inc;
      __syncthreads();
      uint32_t off = 0;
      for (int w = 0; w < wave_id(); ++w) off += scan_tot[w];
      uint64_t rank = tile_rank0 + off + inc - my_cnt;
      uint64_t bits = acc;
      uint32_t* out = bt.page_docs + static_cast<uint64_t>(q.out_slot) * bt.page_stride;
      while (bits) {
        const uint32_t bpos = __builtin_ctzll(bits);
        bits &= bits - 1;
        if (rank >= page_lo && rank < page_hi)
          out[q.descending ? page_total - 1 - rank : rank] =
              static_cast<uint32_t>(tile_first) + static_cast<uint32_t>(tid) * 64 + bpos;
        ++rank;
      }
      __syncthreads();  // scan_tot and the operand bitmaps are rewritten by the next tile
      continue;
    }
    if (MODE == kModeBitmap || MODE == kModeDocCount) {
      if (MODE == kModeBitmap) {
        const uint64_t obase = (static_cast<uint64_t>(q.out_slot) * ix.n_tiles + tile) * kWordsPerTile;
        bt.rbits[obase + tid] = acc;
      }
      // per-tile count
      uint32_t s = wave_incl_scan(my_cnt);
      if (lane_id() == 63) scan_tot[wave_id()] = s;
      __syncthreads();
      if (tid == 0) {
        bt.tile_cnt[static_cast<uint64_t>(q.out_slot) * ix.n_tiles + tile] =
            scan_tot[0] + scan_tot[1] + scan_tot[2] + scan_tot[3];
      }
      __syncthreads();
      continue;
    }

    // ---- D. (score mode) ranks: exclusive prefix popcounts of the result and of every scored operand ----------
    // quantities are packed two per u32 (a tile holds at most 16384 set bits, so 16 bits each)
    // (text-level terms have no operand of their own: their rows stay unused; kModeTextDf ranks the result only)
    const uint32_t n_q = 1 + n_score;
    const uint32_t n_packed = (n_q + 1) >> 1;
    uint32_t excl_res = 0;
    auto operand_pop = [&](uint32_t term) -> uint32_t {
      const uint32_t lf = bt.score_terms[q.score_begin + term].leaf;
      return lf == kNoLeaf ? 0u : static_cast<uint32_t>(__popcll(bm64[leaf[lf].lds * kBlock + tid]));
    };
    for (uint32_t pk = 0; pk < n_packed; ++pk) {
      const uint32_t qa = 2 * pk, qb = 2 * pk + 1;
      const uint32_t va = qa == 0 ? my_cnt : operand_pop(qa - 1);
      const uint32_t vb = qb < n_q ? operand_pop(qb - 1) : 0u;
      const uint32_t v = va | (vb << 16);
      const uint32_t inc = wave_incl_scan(v);
      if (lane_id() == 63) scan_tot[wave_id() * 16 + pk] = inc;
      // stash the wave-local exclusive value; the cross-wave offset is added after the barrier
      const uint32_t ex = inc - v;
      pref[qa * kBlock + tid] = static_cast<uint16_t>(ex & 0xFFFFu);
      if (qb < n_q) pref[qb * kBlock + tid] = static_cast<uint16_t>(ex >> 16);
    }
    __syncthreads();
    uint32_t total_matches = 0;
    for (uint32_t pk = 0; pk < n_packed; ++pk) {
      uint32_t off = 0, tot = 0;
      for (int w = 0; w < 4; ++w) {
        const uint32_t t = scan_tot[w * 16 + pk];
        if (w < wave_id()) off += t;
        tot += t;
      }
      const uint32_t qa = 2 * pk, qb = 2 * pk + 1;
      pref[qa * kBlock + tid] = static_cast<uint16_t>(pref[qa * kBlock + tid] + (off & 0xFFFFu));
      if (qb < n_q) pref[qb * kBlock + tid] = static_cast<uint16_t>(pref[qb * kBlock + tid] + (off >> 16));
      if (pk == 0) {
        total_matches = tot & 0xFFFFu;
        excl_res = pref[tid];
      }
    }
    // (pref rows are read by other threads only after the barrier that follows the first enumeration)

    // ---- E. enumerate matches into LDS in rank order, then score them one per lane -----------------------------
    for (uint32_t rb = 0; rb < total_matches; rb += kMatchBuf) {
      {
        uint64_t bits = acc;
        uint32_t r = excl_res;
        while (bits) {
          const uint32_t bpos = __builtin_ctzll(bits);
          bits &= bits - 1;
          if (r >= rb && r < rb + kMatchBuf) matchbuf[r - rb] = static_cast<uint16_t>(tid * 64 + bpos);
          ++r;
        }
      }
      __syncthreads();
      const uint32_t nm = min(kMatchBuf, total_matches - rb);
      for (uint32_t j0 = 0; j0 < nm; j0 += kBlock) {
        const uint32_t j = j0 + tid;
        const bool valid = j < nm;
        if (MODE == kModeTextDf) {
          // PopulateTermDocumentFrequency (search_pipeline.cpp:556-563): candidates whose text contains the term
          if (valid) {
            const uint32_t slot = tile * kTileDocs + matchbuf[j];
            const uint64_t t0 = ix.text_off[slot], t1 = ix.text_off[slot + 1];
            cnt_df += text_count_occurrences(ix.text, t0, t1, df_pattern, true);
          }
          continue;
        }
        double score = 0.0;
        uint32_t doc = 0;
        if (valid) {
          const uint32_t d = matchbuf[j];
          const uint32_t word = d >> 6, bit = d & 63;
          const uint64_t below = (1ull << bit) - 1ull;
          const uint32_t slot = tile * kTileDocs + d;
          doc = ix.first_doc_id + slot;
          const double dl = static_cast<double>(ix.doc_len[slot]);
          // bm25_scorer.cpp:80-84, same operation order
          const double length_norm = q.one_minus_b + q.b * dl / q.avgdl_clamped;
          for (uint32_t i = 0; i < n_score; ++i) {
            const DevScoreTerm st = bt.score_terms[q.score_begin + i];
            double tf = 0.0, idf = st.idf;
            if (st.leaf == kNoLeaf) {
              const uint64_t t0 = ix.text_off[slot], t1 = ix.text_off[slot + 1];
              tf = static_cast<double>(text_count_occurrences(ix.text, t0, t1, tpat[i], false));
              idf = bt.text_idf[st.text_term];
            } else {
              const DevLeaf slf = leaf[st.leaf];
              const uint64_t wbits = bm64[slf.lds * kBlock + word];
              if ((wbits >> bit) & 1ull) {
                if (slf.kind == kLeafGramBitmap && ix.tfnib != nullptr) {
                  // by doc slot, not by posting rank: a mutable table clears dead documents' bits in the bitmap form
                  // (mgx_index_clear_postings), after which the popcount below a bit is no longer the posting's rank
                  const uint32_t nb = ix.tfnib[static_cast<uint64_t>(slf.b) * ix.nib_row_stride + (slot >> 1)];
                  uint32_t tfv = (nb >> ((slot & 1u) * 4u)) & 15u;
                  if (tfv == 15u) tfv = exact_tf(ix, slf.a, slf.row, slot);
                  tf = static_cast<double>(tfv);
                } else {
                  const uint32_t rank = pref[(1 + i) * kBlock + word] + __popcll(wbits & below);
                  tf = static_cast<double>(posting_tf(ix, seg_lo[st.leaf] + rank));
                }
              }
            }
            if (tf > 0.0) {
              const double numerator = tf * q.k1_plus_1;
              const double denominator = tf + q.k1 * length_norm;
              score += idf * numerator / denominator;
            }
          }
        }
        const uint64_t key = score_key(score, q.descending != 0);
        wave_topk_offer(tk, valid, key, q.descending ? doc : ~doc);
      }
      __syncthreads();
    }
    __syncthreads();  // bitmaps / pref / matchbuf are rewritten by the next tile
  }

  // ---- funnel counters: one atomic per wave per slot -----------------------------------------------------------
  if (MODE != kModeDocPage) {  // (the page pass re-evaluates tiles pass 1 has already counted)
    uint32_t v[6] = {cnt0, cnt1, cnt2, cnt3, cnt_res, cnt_df};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      uint32_t x = v[s];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
      if (lane_id() == 0 && x) atomicAdd(&bt.counters[static_cast<uint64_t>(qi) * 8 + s], (unsigned long long)x);
    }
  }
  if (MODE != kModeScore) return;

  // ---- F. merge the four waves' lists into this workgroup's best `needed`, best first, to HBM ------------------
  wave_topk_truncate(tk);
  if (lane_id() == 0) misc[wave_id()] = tk.have;
  __syncthreads();
  {
    const uint64_t* all_keys = reinterpret_cast<const uint64_t*>(smem + lo_.tk_keys);
    const uint32_t* all_docs = reinterpret_cast<const uint32_t*>(smem + lo_.tk_docs);
    const uint32_t cap = q.cap;
    uint32_t have[4];
    uint32_t total = 0;
    for (int w = 0; w < 4; ++w) {
      have[w] = min(misc[w], q.needed);
      total += have[w];
    }
    const uint64_t obase = static_cast<uint64_t>(it.list) * bt.cand_stride;
    for (uint32_t e = tid; e < 4 * cap; e += kBlock) {
      const uint32_t w = e / cap, i = e % cap;
      if (i >= have[w]) continue;
      const uint64_t k = all_keys[static_cast<size_t>(w) * 2 * cap + i];
      const uint32_t d = all_docs[static_cast<size_t>(w) * 2 * cap + i];
      // rank = entries of the other waves' (sorted) lists that beat this one, plus its own position
      uint32_t rank = i;
      for (uint32_t w2 = 0; w2 < 4; ++w2) {
        if (w2 == w) continue;
        const uint64_t* kk = all_keys + static_cast<size_t>(w2) * 2 * cap;
        const uint32_t* dd = all_docs + static_cast<size_t>(w2) * 2 * cap;
        uint32_t lo = 0, hi = have[w2];
        while (lo < hi) {  // first index whose entry does NOT beat (k,d)
          const uint32_t mid = (lo + hi) >> 1;
          if (better(kk[mid], dd[mid], k, d)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < q.needed) {
        bt.cand_keys[obase + rank] = k;
        bt.cand_docs[obase + rank] = d;
      }
    }
    if (tid == 0) bt.cand_n[it.list] = min(total, q.needed);
  }
}

// One launch for both kinds of item (two launches meant two tails): the host puts the plain items first, the workgroup
// picks its instantiation by its index. The two bodies share no registers in flight; the kernel's budget is the larger one's.
template <int MODE>
__global__ __launch_bounds__(kBlock, (MODE == kModeScore ? 3 : 4)) void tile_eval_kernel(DevIndex ix, DevBatch bt, LdsPlan plan,
                                                                                         uint32_t n_plain) {
  if (blockIdx.x < n_plain) tile_eval_body<MODE, false>(ix, bt, plan);
  else tile_eval_body<MODE, true>(ix, bt, plan);
}

// ---------------------------------------------------------------------------------------------------------------
// wave-autonomous scoring kernel (the fast path of SORT _score batches)
// ---------------------------------------------------------------------------------------------------------------
//
// Same work as tile_eval_kernel<kModeScore> for "flat" programs (LOAD/AND/OR/ANDNOT/COUNT only) whose scored terms
// (at most kWaveScoreSlots) all have the dense form (bitmap row + doc-slot tf nibbles). Built for occupancy: the
// kernel is bound by memory latency and issue slots, not bandwidth, so what counts is how many tiles a CU has in flight.
//   * a workgroup is 8 waves that share nothing but the query's BM25 table; each WAVE owns whole 16384-doc tiles
//     (tile = tile_begin + wave, +8, ...) and never meets a workgroup barrier inside the tile loop;
//   * A: a lane owns 4 consecutive 64-bit words (256 doc slots) of every operand (two 16-byte loads per bitmap-form
//     operand; sorted lists go through a 2 KiB per-wave LDS bitmap) and evaluates the program in registers;
//   * B: every lane appends its matches (14-bit slot inside the tile) to the wave's match buffer;
//   * C: matches are scored one per lane, kScoreUnroll per lane in flight. A match's tf under scored term i is the
//     nibble tfnib[row_i][slot/2] and its doc length dl8[slot] — both addressed by the doc slot alone, so there are no
//     ranks, no prefix popcounts and no parked operand words: T+1 byte gathers, then one LDS table read per term.
//     Nibble 15 (tf >= 15) and dl8 255 take an exact slow path (binary search of the posting segment / the u32
//     doc_len), both rare;
//   * BM25 term contributions idf*tf*(k1+1)/(tf + k1*(1-b+b*dl/avgdl)) are tabulated once per workgroup in LDS for
//     tf <= kTableTf and dl < table_dl with the reference's exact operation order (bm25_scorer.cpp:80-84); anything
//     outside the table is computed directly;
//   * every wave keeps its own running top-k (WaveTopK) pruned by the query-wide bound; the lists are merged at the end.
// 80 VGPRs (12 B spill), ~42 KB LDS: 3 workgroups = 24 waves per CU. (An earlier version located tf through the
// posting rank — prefix popcounts, operand words parked in LDS, per-step re-fetch — and was 1.4x slower.)

uint32_t FastTableDl(uint32_t max_doc_len);

constexpr int kScoreUnroll = 2;          // matches in flight per lane in phase C
constexpr uint32_t kWaveMatchBuf = 512;  // matches of one tile buffered per wave and round

struct WaveOffsets {
  uint32_t leaf, prog, misc, table, scratch, mbuf, tk_keys, tk_docs, total;
};

__host__ __device__ inline WaveOffsets carve_wave(const WavePlan& p) {
  WaveOffsets o;
  uint32_t at = 0;
  o.leaf = at;     at += align8(p.max_leaves * static_cast<uint32_t>(sizeof(DevLeaf)));
  o.prog = at;     at += align8(p.max_instr * 4);
  o.misc = at;     at += 128;
  at = (at + 15u) & ~15u;
  o.table = at;    at += ((p.max_score * kTableTf * p.table_dl + 1u) & ~1u) * 8;
  o.scratch = at;  at += p.has_list ? kWavesPerBlock * kWordsPerTile * 8 : 0;
  o.mbuf = at;     at += kWavesPerBlock * kWaveMatchBuf * 2;
  o.tk_keys = at;  at += kWavesPerBlock * 2 * p.max_cap * 8;
  o.tk_docs = at;  at += kWavesPerBlock * 2 * p.max_cap * 4;
  o.total = at;
  return o;
}

WavePlan PlanWave(uint32_t max_leaves, uint32_t max_score, uint32_t max_instr, uint32_t max_cap, uint32_t max_doc_len,
                  bool has_list) {
  WavePlan p{max_leaves ? max_leaves : 1, max_score, max_instr ? max_instr : 1, max_cap, 0, has_list ? 1u : 0u, 0};
  p.table_dl = FastTableDl(max_doc_len);  // the extent of the pool's tables
  p.bytes = carve_wave(p).total;
  return p;
}

__device__ __forceinline__ uint32_t wave_excl_scan_total(uint32_t v, uint32_t* total) {
  const uint32_t inc = wave_incl_scan(v);
  *total = __builtin_amdgcn_readlane(inc, 63);
  return inc - v;
}

// scatter one sorted segment into a per-wave 16384-bit LDS bitmap, 64 lanes
__device__ __forceinline__ void wave_scatter_segment(const uint32_t* __restrict__ ids, uint64_t lo, uint64_t hi,
                                                     uint32_t tile_first_doc, uint32_t* __restrict__ bm32) {
  const uint64_t p0 = lo & ~3ull;
  for (uint64_t p = p0 + 4ull * lane_id(); p < hi; p += 4ull * 64) {
    const uint4 v = *reinterpret_cast<const uint4*>(ids + p);
    const uint32_t e[4] = {v.x, v.y, v.z, v.w};
    uint32_t curw = 0xFFFFFFFFu, curm = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t q = p + j;
      if (q >= lo && q < hi) {
        const uint32_t bit = e[j] - tile_first_doc;
        const uint32_t w = bit >> 5, m = 1u << (bit & 31);
        if (w == curw) {
          curm |= m;
        } else {
          if (curm) atomicOr(&bm32[curw], curm);
          curw = w;
          curm = m;
        }
      }
    }
    if (curm) atomicOr(&bm32[curw], curm);
  }
}

// wave_scatter_segment with four 16-byte loads per lane in flight (1024 postings per wave and step): a long segment — a
// dense list without a bitmap row holds thousands of postings per tile — was a chain of one load round trip per 256.
__device__ __forceinline__ void wave_scatter_segment_x4(const uint32_t* __restrict__ ids, uint64_t lo, uint64_t hi,
                                                        uint32_t tile_first_doc, uint32_t* __restrict__ bm32) {
  const uint64_t p0 = lo & ~3ull;
  for (uint64_t base = p0; base < hi; base += 4ull * 64 * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t p = base + (static_cast<uint64_t>(u) * 64 + lane_id()) * 4;
      v[u] = p < hi ? *reinterpret_cast<const uint4*>(ids + p) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t p = base + (static_cast<uint64_t>(u) * 64 + lane_id()) * 4;
      if (p >= hi) continue;
      const uint32_t e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      // (ids merged per 32-bit word before the ds_or: merging per 64-bit word and ds_or_b64 measured slower, 3.4 -> 4.4 ms
      // of staging on the lists-only benchmark batch)
      uint32_t curw = 0xFFFFFFFFu, curm = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p + j >= lo && p + j < hi) {
          const uint32_t bit = e[j] - tile_first_doc;
          const uint32_t w = bit >> 5, m = 1u << (bit & 31);
          if (w == curw) {
            curm |= m;
          } else {
            if (curm) atomicOr(&bm32[curw], curm);
            curw = w;
            curm = m;
          }
        }
      }
      if (curm) atomicOr(&bm32[curw], curm);
    }
  }
}

// This lane's four 64-bit words (256 doc slots) of one operand for `tile`; for scored posting-list operands
// *seg_rel = list-relative index of the tile's first posting (the rank base of the tf column).
__device__ __forceinline__ void wave_fetch_operand(const DevIndex& ix, const DevBatch& bt, const DevLeaf lf,
                                                   uint32_t tile, uint64_t tile_first, uint64_t* scratch,
                                                   uint64_t (&w)[4], uint32_t* seg_rel) {
  const uint32_t lane = lane_id();
  *seg_rel = 0;
  if (lf.kind == kLeafGramBitmap || lf.kind == kLeafFilterBitmap) {
    const uint64_t* rowp = lf.kind == kLeafGramBitmap
                               ? ix.gram_bitmaps + tile * ix.gb_tile_stride + lf.b * ix.gb_row_stride + lane * 4
                               : ix.filter_bitmaps + tile * ix.fb_tile_stride + lf.b * ix.fb_row_stride + lane * 4;
    const uint4 v0 = *reinterpret_cast<const uint4*>(rowp);
    const uint4 v1 = *reinterpret_cast<const uint4*>(rowp + 2);
    w[0] = (static_cast<uint64_t>(v0.y) << 32) | v0.x;
    w[1] = (static_cast<uint64_t>(v0.w) << 32) | v0.z;
    w[2] = (static_cast<uint64_t>(v1.y) << 32) | v1.x;
    w[3] = (static_cast<uint64_t>(v1.w) << 32) | v1.z;
    if (lf.kind == kLeafGramBitmap && lf.score_slot != kNoSlot)
      *seg_rel = ix.tile_off[static_cast<uint64_t>(lf.row) * (ix.n_tiles + 1) + tile];
  } else if (lf.kind == kLeafRange) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t s0 = static_cast<uint64_t>(tile) * kTileDocs + (static_cast<uint64_t>(lane) * 4 + k) * 64;
      const uint64_t ra = lf.a > s0 ? lf.a - s0 : 0, rb = lf.b > s0 ? lf.b - s0 : 0;
      const uint64_t hi_mask = rb >= 64 ? ~0ull : ((1ull << rb) - 1ull);
      const uint64_t lo_mask = ra >= 64 ? ~0ull : ((1ull << ra) - 1ull);
      w[k] = hi_mask & ~lo_mask;
    }
  } else {  // sorted list (posting list or explicit ids): scatter into this wave's LDS bitmap
    uint64_t a, b;
    const uint32_t* ids;
    if (lf.kind == kLeafList) {
      ids = ix.docids;
      const uint64_t l0 = ix.offsets[lf.a], l1 = ix.offsets[lf.a + 1];
      if (lf.row != kNoRow) {
        const uint32_t* r = ix.tile_off + static_cast<uint64_t>(lf.row) * (ix.n_tiles + 1);
        a = l0 + r[tile];
        b = l0 + r[tile + 1];
      } else {
        a = lower_bound_u32(ids, l0, l1, tile_first);
        b = lower_bound_u32(ids, a, l1, tile_first + kTileDocs);
      }
      *seg_rel = static_cast<uint32_t>(a - l0);
    } else {
      ids = bt.explicit_pool;
      a = lower_bound_u32(ids, lf.a, static_cast<uint64_t>(lf.a) + lf.b, tile_first);
      b = lower_bound_u32(ids, a, static_cast<uint64_t>(lf.a) + lf.b, tile_first + kTileDocs);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) scratch[lane * 4 + k] = 0;
    wave_lds_sync();
    wave_scatter_segment(ids, a, b, static_cast<uint32_t>(tile_first), reinterpret_cast<uint32_t*>(scratch));
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = scratch[lane * 4 + k];
    wave_lds_sync();
  }
}

// exact tf of gram `g` in the doc at `slot` (the nibble saturated, or the gram has no nibble row because its list is
// sparse): binary search of the tile's posting segment; 0 when the doc lacks the gram
__device__ __forceinline__ uint32_t exact_tf(const DevIndex& ix, uint32_t g, uint32_t row, uint32_t slot) {
  const uint32_t tile = slot >> kTileShift;
  const uint64_t l0 = ix.offsets[g];
  uint64_t lo = l0, hi = ix.offsets[g + 1];
  if (row != kNoRow) {  // short lists have no skip row: the whole list is the segment
    const uint32_t* r = ix.tile_off + static_cast<uint64_t>(row) * (ix.n_tiles + 1);
    hi = l0 + r[tile + 1];
    lo = l0 + r[tile];
  }
  const uint32_t d = ix.first_doc_id + slot;
  const uint64_t p = lower_bound_u32(ix.docids, lo, hi, d);
  return (p < hi && ix.docids[p] == d) ? posting_tf(ix, p) : 0u;
}

__device__ __forceinline__ void wave_score_body(const DevIndex ix, const DevBatch bt, const WavePlan plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WaveOffsets wo = carve_wave(plan);
  DevLeaf* const leaf = reinterpret_cast<DevLeaf*>(smem + wo.leaf);
  uint32_t* const prog = reinterpret_cast<uint32_t*>(smem + wo.prog);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + wo.misc);
  double* const table = reinterpret_cast<double*>(smem + wo.table);
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  uint64_t* const scratch = reinterpret_cast<uint64_t*>(smem + wo.scratch) + static_cast<size_t>(wave) * kWordsPerTile;
  uint16_t* const mbuf = reinterpret_cast<uint16_t*>(smem + wo.mbuf) + static_cast<size_t>(wave) * kWaveMatchBuf;

  const DevItem it = bt.items[blockIdx.x];
  const uint32_t qi = it.query;
  const DevQuery q = bt.queries[qi];
  const uint32_t tdl = plan.table_dl;
  for (uint32_t i = tid; i < q.n_leaves; i += kWaveBlock) leaf[i] = bt.leaves[q.leaf_begin + i];
  for (uint32_t i = tid; i < q.n_instr; i += kWaveBlock) prog[i] = bt.prog[q.prog_begin + i];
  // The query's BM25 contribution tables come from the index's pool (one table per (gram, idf, k1, b, avgdl), built on
  // first use): rows tf 1..kTableTf of every scored term are staged, tdl doubles each (tdl is even).
  for (uint32_t i = 0; i < q.n_score; ++i) {
    const double2* src = reinterpret_cast<const double2*>(bt.wave_tables[static_cast<uint64_t>(qi) * kWaveScoreSlots + i]) + tdl / 2;
    double2* dst = reinterpret_cast<double2*>(table + i * kTableTf * tdl);
    for (uint32_t e = tid; e < kTableTf * tdl / 2; e += kWaveBlock) dst[e] = src[e];
  }
  WaveTopK tk;
  tk.cap = q.cap;
  tk.needed = q.needed;
  tk.lds_sort = false;
  tk.keys = reinterpret_cast<uint64_t*>(smem + wo.tk_keys) + static_cast<size_t>(wave) * 2 * q.cap;
  tk.docs = reinterpret_cast<uint32_t*>(smem + wo.tk_docs) + static_cast<size_t>(wave) * 2 * q.cap;
  tk.have = 0;
  tk.pend = 0;
  tk.bound_key = 0;
  tk.bound_doc = 0;
  tk.gbound_ptr = bt.bounds ? bt.bounds + qi : nullptr;
  tk.gbound = 0;
  for (uint32_t i = lane; i < 2 * q.cap; i += 64) {
    tk.keys[i] = 0;
    tk.docs[i] = 0;
  }
  // per scored term (LDS, misc[16..]): gram id, skip row (exact-tf path), bitmap row (nibble row)
  uint32_t* const sterm = misc + 16;
  double* const zero_slot = reinterpret_cast<double*>(misc + 28);  // contribution of a term the doc does not hold
  if (tid == 0) *zero_slot = 0.0;
  if (tid < q.n_score) {
    const DevLeaf lf = bt.leaves[q.leaf_begin + bt.score_terms[q.score_begin + tid].leaf];
    sterm[tid * 3] = lf.a;
    sterm[tid * 3 + 1] = lf.row;
    sterm[tid * 3 + 2] = lf.kind == kLeafGramBitmap ? lf.b : kNoRow;
  }
  __syncthreads();
  // A scored term whose gram is sparse (list form) has no nibble row: it reads as "saturated" and takes the exact path.
  const uint8_t* nib[kWaveScoreSlots];
  uint32_t no_nib = 0;
#pragma unroll
  for (int i = 0; i < kWaveScoreSlots; ++i) {
    nib[i] = ix.tfnib;
    if (static_cast<uint32_t>(i) < q.n_score) {
      const uint32_t brow = wave_uniform(sterm[i * 3 + 2]);
      if (brow == kNoRow) no_nib |= 1u << i;
      else nib[i] = ix.tfnib + static_cast<uint64_t>(brow) * ix.nib_row_stride;
    }
  }

  uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0, cnt_res = 0;
  const uint32_t tile_begin = it.tile_begin;
  const uint32_t tile_end = min(tile_begin + it.n_tiles, ix.n_tiles);
  const bool desc = q.descending != 0;

  for (uint32_t tile = tile_begin + wave; tile < tile_end; tile += kWavesPerBlock) {
    const uint64_t tile_first = static_cast<uint64_t>(ix.first_doc_id) + static_cast<uint64_t>(tile) * kTileDocs;
    wave_topk_refresh_gbound(tk);
    // ---- A. the program on this lane's four words ------------------------------------------------------------------
    uint64_t acc[4] = {0, 0, 0, 0};
    for (uint32_t pc = 0; pc < q.n_instr; ++pc) {
      const uint32_t ins = prog[pc];
      const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
      if (op == kOpCount) {
        const uint32_t pcnt = __popcll(acc[0]) + __popcll(acc[1]) + __popcll(acc[2]) + __popcll(acc[3]);
        cnt0 += (arg & 1u) ? pcnt : 0;
        cnt1 += (arg & 2u) ? pcnt : 0;
        cnt2 += (arg & 4u) ? pcnt : 0;
        cnt3 += (arg & 8u) ? pcnt : 0;
        continue;
      }
      DevLeaf lf = leaf[arg];
      lf.score_slot = kNoSlot;  // (no rank base needed here)
      uint64_t w[4];
      uint32_t seg_rel;
      wave_fetch_operand(ix, bt, lf, tile, tile_first, scratch, w, &seg_rel);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (op == kOpLoad) acc[k] = w[k];
        else if (op == kOpAnd) acc[k] &= w[k];
        else if (op == kOpOr) acc[k] |= w[k];
        else if (op == kOpAndNot) acc[k] &= ~w[k];
      }
    }
    cnt_res += __popcll(acc[0]) + __popcll(acc[1]) + __popcll(acc[2]) + __popcll(acc[3]);
    if (MGX_ABLATE(bt, 2u)) continue;

    // Rounds of at most kWaveMatchBuf matches: B consumes bits of acc (a lane resumes where it stopped), C scores.
    for (;;) {
      uint32_t n_left;
      const uint32_t mine = __popcll(acc[0]) + __popcll(acc[1]) + __popcll(acc[2]) + __popcll(acc[3]);
      const uint32_t my_first = wave_excl_scan_total(mine, &n_left);
      if (n_left == 0) break;  // wave-uniform
      // ---- B. (owner lane, word, bit) of this lane's next matches ---------------------------------------------------
      {
        uint32_t r = my_first;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          while (acc[k] != 0 && r < kWaveMatchBuf) {
            const uint32_t bit = __builtin_ctzll(acc[k]);
            acc[k] &= acc[k] - 1;
            mbuf[r] = static_cast<uint16_t>((lane << 8) | (k << 6) | bit);  // = the doc's slot inside the tile
            ++r;
          }
        }
      }
      wave_lds_sync();
      const uint32_t nm = min(kWaveMatchBuf, n_left);
      // ---- C. one match per lane, kScoreUnroll in flight -------------------------------------------------------------
      for (uint32_t j0 = 0; j0 < nm && !MGX_ABLATE(bt, 1u); j0 += 64 * kScoreUnroll) {
        bool valid[kScoreUnroll];
        uint32_t slot[kScoreUnroll], dli[kScoreUnroll];
        uint32_t tfv[kScoreUnroll][kWaveScoreSlots];
#pragma unroll
        for (int m = 0; m < kScoreUnroll; ++m) {
          valid[m] = false;
          slot[m] = 0;
          dli[m] = 0;
#pragma unroll
          for (int i = 0; i < kWaveScoreSlots; ++i) tfv[m][i] = 0;
          if (j0 + m * 64 < nm) {  // wave-uniform
            const uint32_t j = j0 + m * 64 + lane;
            valid[m] = j < nm;
            if (valid[m]) {
              slot[m] = tile * kTileDocs + mbuf[j];
              // every gather of the iteration is issued before any of them is looked at
#pragma unroll
              for (int i = 0; i < kWaveScoreSlots; ++i)
                if (static_cast<uint32_t>(i) < q.n_score)
                  tfv[m][i] = (no_nib >> i) & 1u ? 0xFFu : static_cast<uint32_t>(nib[i][slot[m] >> 1]);
              dli[m] = ix.dl8[slot[m]];
            }
          }
        }
        bool slow = false;
#pragma unroll
        for (int m = 0; m < kScoreUnroll; ++m) {
          if (j0 + m * 64 < nm) {  // wave-uniform
            const uint32_t sh = (slot[m] & 1u) * 4u;
#pragma unroll
            for (int i = 0; i < kWaveScoreSlots; ++i) {
              tfv[m][i] = (tfv[m][i] >> sh) & 15u;
              slow = slow || tfv[m][i] == 15u;
            }
            slow = slow || (valid[m] && dli[m] == 255u);
          }
        }
        if (__ballot(slow) != 0) {  // wave-uniform, rare: saturated nibble or doc length
#pragma unroll
          for (int m = 0; m < kScoreUnroll; ++m) {
            if (valid[m]) {
#pragma unroll
              for (int i = 0; i < kWaveScoreSlots; ++i)
                if (tfv[m][i] == 15u) tfv[m][i] = exact_tf(ix, sterm[i * 3], sterm[i * 3 + 1], slot[m]);
              if (dli[m] == 255u) dli[m] = ix.doc_len[slot[m]];
            }
          }
        }
#pragma unroll
        for (int m = 0; m < kScoreUnroll; ++m) {
          if (j0 + m * 64 < nm) {  // wave-uniform
            // Contributions without per-term branches: a term the doc lacks (or one outside the table) reads the 0.0
            // slot — score + 0.0 is score bit for bit, contributions being > 0 — so the table reads go out together;
            // tf above the table or a doc longer than it (rare) are then evaluated directly.
            double contrib[kWaveScoreSlots];
            bool direct = false;
#pragma unroll
            for (int i = 0; i < kWaveScoreSlots; ++i) {
              const bool in_table = tfv[m][i] - 1u < kTableTf && dli[m] < tdl;  // (tf 0 wraps around: not in the table)
              direct = direct || (tfv[m][i] != 0 && !in_table);
              const double* src = in_table ? table + ((i * kTableTf + tfv[m][i] - 1) * tdl + dli[m]) : zero_slot;
              contrib[i] = *src;
            }
            if (__ballot(direct) != 0) {  // wave-uniform
#pragma unroll
              for (int i = 0; i < kWaveScoreSlots; ++i) {
                if (tfv[m][i] != 0 && !(tfv[m][i] - 1u < kTableTf && dli[m] < tdl)) {
                  // bm25_scorer.cpp:80-84, same operation order as the table
                  const double dl = static_cast<double>(dli[m]), tf = static_cast<double>(tfv[m][i]);
                  const double length_norm = q.one_minus_b + q.b * dl / q.avgdl_clamped;
                  const double numerator = tf * q.k1_plus_1;
                  const double denominator = tf + q.k1 * length_norm;
                  contrib[i] = bt.score_terms[q.score_begin + i].idf * numerator / denominator;
                }
              }
            }
            double score = 0.0;
#pragma unroll
            for (int i = 0; i < kWaveScoreSlots; ++i) score += contrib[i];
            const uint32_t doc = ix.first_doc_id + slot[m];
            wave_topk_offer(tk, valid[m], score_key(score, desc), desc ? doc : ~doc);
          }
        }
      }
      wave_lds_sync();
    }
  }

  {
    uint32_t v[5] = {cnt0, cnt1, cnt2, cnt3, cnt_res};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      uint32_t x = v[s];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
      if (lane == 0 && x) atomicAdd(&bt.counters[static_cast<uint64_t>(qi) * 8 + s], (unsigned long long)x);
    }
  }

  // ---- merge the waves' lists into this workgroup's best `needed`, best first, to HBM ---------------------------------
  wave_topk_truncate(tk);
  if (lane == 0) misc[wave] = tk.have;
  __syncthreads();
  {
    const uint64_t* all_keys = reinterpret_cast<const uint64_t*>(smem + wo.tk_keys);
    const uint32_t* all_docs = reinterpret_cast<const uint32_t*>(smem + wo.tk_docs);
    const uint32_t cap = q.cap;
    uint32_t have[kWavesPerBlock];
    uint32_t total = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) {
      have[w] = min(misc[w], q.needed);
      total += have[w];
    }
    const uint64_t obase = static_cast<uint64_t>(it.list) * bt.cand_stride;
    for (uint32_t e = tid; e < kWavesPerBlock * cap; e += kWaveBlock) {
      const uint32_t w = e / cap, i = e % cap;
      if (i >= have[w]) continue;
      const uint64_t k = all_keys[static_cast<size_t>(w) * 2 * cap + i];
      const uint32_t d = all_docs[static_cast<size_t>(w) * 2 * cap + i];
      uint32_t rank = i;
      for (uint32_t w2 = 0; w2 < kWavesPerBlock; ++w2) {
        if (w2 == w) continue;
        const uint64_t* kk = all_keys + static_cast<size_t>(w2) * 2 * cap;
        const uint32_t* dd = all_docs + static_cast<size_t>(w2) * 2 * cap;
        uint32_t lo = 0, hi = have[w2];
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (better(kk[mid], dd[mid], k, d)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < q.needed) {
        bt.cand_keys[obase + rank] = k;
        bt.cand_docs[obase + rank] = d;
      }
    }
    if (tid == 0) bt.cand_n[it.list] = min(total, q.needed);
  }
}

// Two entry points over the same body: the all-bitmap launch of a step and the (small) launch of the queries that need
// the sorted-list scratch run side by side; separate symbols keep them apart in profiles.
__global__ __launch_bounds__(kWaveBlock, 6) void wave_score_kernel(DevIndex ix, DevBatch bt, WavePlan plan) {
  wave_score_body(ix, bt, plan);
}
__global__ __launch_bounds__(kWaveBlock, 6) void wave_score_lists_kernel(DevIndex ix, DevBatch bt, WavePlan plan) {
  wave_score_body(ix, bt, plan);
}

// ---------------------------------------------------------------------------------------------------------------
// bitmap_score_kernel: the fast path of SORT _score batches (flat AND/OR/ANDNOT programs over bitmap-form operands)
// ---------------------------------------------------------------------------------------------------------------
//
// Same results as wave_score_kernel. What the round-2 profile of that kernel says (profiles/r02_headline_*): it moves
// 12.8 GB of L2 misses per 1024-query launch at 5.8 TB/s — 84 % of the streaming bandwidth this box reaches — and four
// fifths of that is the per-match byte gathers (three tf nibbles + one doc length), not the operand bitmaps (the
// intersection alone, wave_count_kernel, runs AT the streaming bandwidth). It is bound by bytes, so this kernel
// removes bytes — by not scoring matches that cannot reach the page:
//   * block-max pruning. The index keeps, per dense gram and per 64-doc word (the densest grams: per 16-doc quarter),
//     one byte bounding the gram's BM25 term factor over the docs of the block (build_blockmax_kernel). A lane reads
//     the bytes of its 256 doc slots with its operands, sums idf-weighted bounds per quarter and drops the quarters
//     whose bound is below the query's current k-th best score (its own list's, or any wave's of the query through the
//     shared bound) — in integers: idf weights rounded up to 8 bits, the terms' bytes transposed with v_perm_b32, one
//     v_dot4_u32_u8 per quarter. The matches are still counted (total_results is exact); only survivors are enumerated,
//     gathered and scored. On the benchmark batch about a fifth of the matches survive.
//   * no per-query BM25 tables: a contribution is idf * (tf * (k1 + 1)) / (tf + K[dl]) with K[dl] = k1 * (1 - b + b *
//     dl / avgdl) — a 2 KiB table of the BATCH (k1, b, avgdl are table constants), built on the host operation by
//     operation like bm25_scorer.cpp:80-84 — and one correctly rounded fp64 division per (match, term) on the device:
//     no staging of up to 36 KB per workgroup, every nibble value 1..14 covered (the tables stopped at tf 6).
//   * the resolved-query form: an operand is an address + tile stride in scalar registers (no interpreter, no operand-
//     kind dispatch), byte gathers go through buffer descriptors, survivors wait in a per-wave buffer from tile to tile
//     and are scored in full rounds (a round's gathers cost one memory round trip however few lanes carry a match).
// One workgroup = 8 autonomous waves on one query; a wave owns whole 16384-doc tiles (a lane owns 256 doc slots =
// eight 32-bit words = sixteen 16-doc quarters). No workgroup barrier inside the tile loop.

#ifndef MGX_FPL
#define MGX_FPL 4
#endif
constexpr int kFastPerLane = MGX_FPL;               // matches per lane scored in one round: their gathers fly together
constexpr uint32_t kFastRound = 64 * kFastPerLane;  // matches per round
// Surviving 32-doc words wait as (first doc slot, bits) pairs until 64 of them can be expanded by 64 lanes at once: up to
// 63 left from earlier tiles plus the 256 a half tile can add.
constexpr uint32_t kFastPairs = 320;

struct FastOffsets {
  uint32_t ktab, pairs, ring, tk_keys, tk_docs, misc, total;
};
__host__ __device__ inline FastOffsets carve_fast(const FastPlan& p) {
  FastOffsets o;
  uint32_t at = 0;
  o.ktab = at;     at += 256 * 8;
  o.pairs = at;    at += kFastWaves * kFastPairs * 8;
  o.ring = at;     at += kFastWaves * p.ring * 4;
  o.tk_keys = at;  at += kFastWaves * 2 * p.max_cap * 8;
  o.tk_docs = at;  at += kFastWaves * 2 * p.max_cap * 4;
  o.misc = at;     at += 64;
  o.total = at;
  return o;
}
uint32_t FastTableDl(uint32_t max_doc_len) {
  const uint32_t t = max_doc_len + 1 < kTableDlMax ? max_doc_len + 1 : kTableDlMax;
  return (t + 1u) & ~1u;  // even: table rows are copied 16 bytes at a time
}
FastPlan PlanFast(uint32_t max_cap, const double* ktab) {
  FastPlan p{};
  p.ring = kFastRound + 128;
  p.max_cap = max_cap;
  p.ktab = ktab;
  p.bytes = carve_fast(p).total;
  return p;
}

// Global-address-space views of resolved addresses: the compiler cannot see that an integer turned pointer is global
// memory and would emit flat loads (which also tick the LDS counter).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1)))* gptr_u4;
typedef const uint32_t __attribute__((address_space(1)))* gptr_u1;
// A query's resolved description, read through the CONSTANT address space: the kernel never writes it, and only there
// does the compiler fetch its (wave-uniform) fields with scalar loads. Through a generic pointer every `fq->field`
// inside the tile loop became a vector load + s_waitcnt vmcnt(0) + v_readfirstlane — the kernel's own stores and atomics
// forbid the "not clobbered" proof scalar loads need — i.e. a serialized L2 round trip per field group, about ten per
// tile visit in front of the operand loads.
typedef const DevFastQuery __attribute__((address_space(4)))* FastQueryPtr;

// The cold corner of scoring, out of line so that its needs (posting arrays, skip rows, the overflow table) stay out of
// the hot loop's registers: the exact tf of a saturated nibble (tf >= 15). `ix` is the index's descriptor in device
// memory (DevBatch::dev_index).
__device__ __noinline__ uint32_t fast_exact_tf(const DevIndex* ix, uint32_t gram, uint32_t row, uint32_t slot) {
  return exact_tf(*ix, gram, row, slot);
}

// The block-max bytes of this lane's 256 doc slots of `tile`, one 16-byte load per scored term and NO control flow, so
// that the T loads and the operand loads behind them are in flight together (with a branch per mode the compiler parked
// an s_waitcnt vmcnt(0) between the terms and in front of the operand loads: three serialized round trips per visit).
// Mode 2 (a byte per 16-doc quarter): the lane's own four words. Mode 1 (a byte per 64-doc word, 256 bytes per tile row):
// the 16 bytes that hold this lane's four — lanes 4j..4j+3 read the same 16 — picked apart in fast_quarter_mask.
template <int T>
__device__ __forceinline__ void fast_load_blockmax(const FastQueryPtr fq, uint32_t tile, uint32_t lane, u32x4 (&raw)[T]) {
  const uint64_t bm = reinterpret_cast<uint64_t>(fq->blockmax), bmf = reinterpret_cast<uint64_t>(fq->blockmax_fine);
  const uint32_t st = fq->bm_tile_stride, stf = fq->bmf_tile_stride;
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const bool fine = fq->score[i].bm_mode == 2u;  // wave-uniform
    const uint64_t row = (fine ? bmf : bm) + static_cast<uint64_t>(tile) * (fine ? stf : st) + fq->score[i].bm_off;
    const uint32_t off = fine ? lane << 4 : (lane >> 2) << 4;
    raw[i] = *reinterpret_cast<gptr_u4>(row + off);
  }
}

// One bit per 16-doc quarter of this lane's 256 slots: "a doc of this quarter can still reach the page" against the
// k-th best key `bound`. Integer form: a quarter survives iff sum_i W_i * q_i >= floor(theta * bm_inv_unit), W_i the idf
// weights rounded UP to 8 bits (the host), theta's conversion rounded DOWN: the integer bound stays above the exact one,
// which itself sits a relative 2^-41 above any score (build_blockmax_kernel). Per 64-doc word the terms' bytes are transposed
// (v_perm_b32) so that one v_dot4_u32_u8 per quarter does the weighted sum.
template <int T>
__device__ __forceinline__ uint32_t fast_quarter_mask(const FastQueryPtr fq, uint32_t lane, const u32x4 (&raw)[T],
                                                      uint64_t bound) {
  const uint32_t tint = static_cast<uint32_t>(
      fmin(floor(key_score(bound, true) * fq->bm_inv_unit * (1.0 - 0x1p-30)), 4294967040.0));
  const uint32_t wpack = fq->bm_wpack, w4 = fq->bm_w4, cint = fq->bm_cint;
  uint32_t bmw[T][4];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    bmw[i][0] = raw[i].x;
    bmw[i][1] = raw[i].y;
    bmw[i][2] = raw[i].z;
    bmw[i][3] = raw[i].w;
    const uint32_t mode = fq->score[i].bm_mode;
    if (mode == 1u) {  // wave-uniform: this lane's four bytes are dword (lane & 3); a word's byte stands for its four quarters
      const uint32_t l3 = lane & 3u;
      const uint32_t c = l3 == 0u ? raw[i].x : l3 == 1u ? raw[i].y : l3 == 2u ? raw[i].z : raw[i].w;
      bmw[i][0] = __builtin_amdgcn_perm(0u, c, 0x00000000u);
      bmw[i][1] = __builtin_amdgcn_perm(0u, c, 0x01010101u);
      bmw[i][2] = __builtin_amdgcn_perm(0u, c, 0x02020202u);
      bmw[i][3] = __builtin_amdgcn_perm(0u, c, 0x03030303u);
    } else if (mode != 2u) {
      bmw[i][0] = bmw[i][1] = bmw[i][2] = bmw[i][3] = 0;
    }
  }
  uint32_t mk = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t t0 = bmw[0][k], t1 = T > 1 ? bmw[T > 1 ? 1 : 0][k] : 0u, t2 = T > 2 ? bmw[T > 2 ? 2 : 0][k] : 0u,
                   t3 = T > 3 ? bmw[T > 3 ? 3 : 0][k] : 0u;
    // [t0.0 t1.0 t0.1 t1.1], [t0.2 t1.2 t0.3 t1.3] and the same of terms 2, 3
    const uint32_t lo01 = __builtin_amdgcn_perm(t1, t0, 0x05010400u), hi01 = __builtin_amdgcn_perm(t1, t0, 0x07030602u);
    const uint32_t lo23 = T > 2 ? __builtin_amdgcn_perm(t3, t2, 0x05010400u) : 0u;
    const uint32_t hi23 = T > 2 ? __builtin_amdgcn_perm(t3, t2, 0x07030602u) : 0u;
    uint32_t qb[4];  // quarter j: the bytes of terms 0..3
    qb[0] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
    qb[1] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
    qb[2] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
    qb[3] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t ub = __builtin_amdgcn_udot4(qb[j], wpack, cint, false);
      if (T > 4) ub += w4 * ((bmw[T > 4 ? 4 : 0][k] >> (8 * j)) & 255u);
      if (ub >= tint) mk |= 1u << (4 * k + j);
    }
  }
  return mk;
}

// The quarter bits of a tile once more, against a bound that moved while the tile's matches were being scored (a tile
// with more matches than the wave's buffer holds is scored in chunks): out of line, the rare path of heavy queries.
template <int T>
__device__ __noinline__ uint32_t fast_remask(const FastQueryPtr fq, uint32_t tile, uint32_t lane, uint64_t bound) {
  u32x4 raw[T];
  fast_load_blockmax<T>(fq, tile, lane, raw);
  return fast_quarter_mask<T>(fq, lane, raw, bound);
}

template <int T>
__device__ __forceinline__ void bitmap_score_body(const DevIndex& ix, const DevBatch& bt, const FastPlan& plan,
                                                  const FastQueryPtr fq, const DevItem it) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const FastOffsets fo = carve_fast(plan);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = wave_uniform(tid >> 6);
  double* const ktab = reinterpret_cast<double*>(smem + fo.ktab);
  uint32_t* const ring = reinterpret_cast<uint32_t*>(smem + fo.ring) + wave * plan.ring;
  uint2* const pairs = reinterpret_cast<uint2*>(smem + fo.pairs) + wave * kFastPairs;
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + fo.misc);
  const uint32_t n_ops = fq->n_ops;
  const bool desc = fq->descending != 0;
  const uint32_t tile_begin = it.tile_begin;
  const uint32_t tile_end = min(tile_begin + it.n_tiles, ix.n_tiles);

  // the batch's K[dl] = k1 * (1 - b + b * dl / avgdl), dl 0..254 (entry 255 unused: the escape value)
  if (tid < 256) ktab[tid] = plan.ktab[tid];

  WaveTopK tk;
  tk.cap = fq->cap;
  tk.needed = fq->needed;
  tk.lds_sort = true;
  tk.keys = reinterpret_cast<uint64_t*>(smem + fo.tk_keys) + static_cast<size_t>(wave) * 2 * fq->cap;
  tk.docs = reinterpret_cast<uint32_t*>(smem + fo.tk_docs) + static_cast<size_t>(wave) * 2 * fq->cap;
  tk.have = 0;
  tk.pend = 0;
  tk.bound_key = 0;
  tk.bound_doc = 0;
  tk.gbound_ptr = bt.bounds ? bt.bounds + it.query : nullptr;
  tk.gbound = 0;
  for (uint32_t i = lane; i < 2 * tk.cap; i += 64) {
    tk.keys[i] = 0;
    tk.docs[i] = 0;
  }
  __syncthreads();

  // buffer descriptors of the byte columns (wave-uniform: built from kernel arguments and scalar loads only)
  __amdgpu_buffer_rsrc_t nib_rsrc[T];
  double idf[T];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    nib_rsrc[i] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(fq->score[i].nib), 0,
                                                    static_cast<int>((ix.n_docs + 1u) >> 1), 0x00020000);
    idf[i] = fq->score[i].idf;
  }
  const __amdgpu_buffer_rsrc_t dl_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(ix.dl8), 0, static_cast<int>(ix.n_docs), 0x00020000);
  const double k1_plus_1 = fq->k1_plus_1;
  const bool prune = fq->blockmax != 0;  // wave-uniform

  uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0, cnt_res = 0;
  uint32_t pend = 0;  // matches waiting in the ring (wave-uniform)
  uint32_t np = 0;    // pairs waiting to be expanded (wave-uniform)

  // scores ring[g0 .. g0+n), n <= kFastRound: one match per lane and step, kFastPerLane steps whose T + 1 byte gathers
  // are all requested before the first is looked at
  auto score_round = [&](uint32_t g0, uint32_t n) {
    uint32_t slot[kFastPerLane], dli[kFastPerLane], nb[kFastPerLane][T];
#pragma unroll
    for (int m = 0; m < kFastPerLane; ++m) {
      const uint32_t j = m * 64 + lane;
      slot[m] = j < n ? ring[g0 + j] : 0u;
    }
#pragma unroll
    for (int m = 0; m < kFastPerLane; ++m) {
#pragma unroll
      for (int i = 0; i < T; ++i) nb[m][i] = __builtin_amdgcn_raw_buffer_load_b8(nib_rsrc[i], slot[m] >> 1, 0, 0);
      dli[m] = __builtin_amdgcn_raw_buffer_load_b8(dl_rsrc, slot[m], 0, 0);
    }
#pragma nounroll
    for (int m = 0; m < kFastPerLane; ++m) {  // (one copy of the scoring + offer code; the selects below are cheap)
      if (static_cast<uint32_t>(m) * 64u >= n) break;  // wave-uniform
      uint32_t sl = slot[0], dl = dli[0], nbm[T];
#pragma unroll
      for (int i = 0; i < T; ++i) nbm[i] = nb[0][i];
#pragma unroll
      for (int mm = 1; mm < kFastPerLane; ++mm) {
        if (m == mm) {  // wave-uniform
          sl = slot[mm];
          dl = dli[mm];
#pragma unroll
          for (int i = 0; i < T; ++i) nbm[i] = nb[mm][i];
        }
      }
      const bool valid = static_cast<uint32_t>(m) * 64u + lane < n;
      double kd = ktab[dl];
      const uint32_t sh = (sl & 1u) << 2;
      uint32_t tf[T], mx = 0;
#pragma unroll
      for (int i = 0; i < T; ++i) {
        tf[i] = (nbm[i] >> sh) & 15u;
        mx = max(mx, tf[i]);
      }
      if (__ballot(valid && (mx == 15u || dl == 255u)) != 0) {  // wave-uniform, rare: saturated nibble / doc length
#pragma unroll
        for (int i = 0; i < T; ++i)
          if (valid && tf[i] == 15u) tf[i] = fast_exact_tf(bt.dev_index, fq->score[i].gram, fq->score[i].skip_row, sl);
        if (valid && dl == 255u) {
          const double dld = static_cast<double>(ix.doc_len[sl]);
          kd = fq->k1 * (fq->one_minus_b + fq->b * dld / fq->avgdl_clamped);  // bm25_scorer.cpp:80-84
        }
      }
      double score = 0.0;
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const double tfd = static_cast<double>(tf[i]);
        // bm25_scorer.cpp:80-84: idf * numerator / denominator, numerator = tf * (k1 + 1), denominator = tf + k1 * norm;
        // a term the doc lacks adds +0.0 where bm25_scorer.cpp:86 skips it
        const double c = tf[i] != 0u ? idf[i] * (tfd * k1_plus_1) / (tfd + kd) : 0.0;
        score = i == 0 ? c : score + c;
      }
      const uint32_t doc = ix.first_doc_id + sl;
      wave_topk_offer(tk, valid, score_key(score, desc), desc ? doc : ~doc);
    }
  };

  // This lane's 16 bytes of half `h` of operand `o` for `tile` (issued and not waited for)
  auto load_half = [&](uint32_t o, uint32_t tile, uint32_t h) {
    const uint64_t op_base = fq->ops[o].base;
    const uint32_t op_code = fq->ops[o].code, op_stride = fq->ops[o].tile_stride;
    const uint64_t base = op_base + ((op_code & 16u) ? reinterpret_cast<uint64_t>(ix.filter_bitmaps) : 0ull) +
                          static_cast<uint64_t>(tile) * op_stride;
    const gptr_u4 p = reinterpret_cast<gptr_u4>(base) + lane * 2 + h;
    return p[0];
  };

  // The tile loop runs one extra, tile-less pass at the end that only scores the wave's last partial round, so the
  // scoring code exists once.
  for (uint32_t tile = tile_begin + wave;; tile += kFastWaves) {
    const bool flush = tile >= tile_end;  // wave-uniform
    uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t bound = 0;  // the k-th best key this tile's quarter bits were made against (wave-uniform)
    if (!flush) {
      wave_topk_refresh_gbound(tk);
      // ---- pruning: one bit per 16-doc quarter of this lane's 256 slots — "can still enter the page". The block-max
      // bytes are requested first, the operands right behind them, and the bits are made while the operands fly.
      uint32_t qmask = 0xFFFFu;
      if (prune) {
        // the k-th best so far — this wave's own list or any wave's of the query: a quarter whose bound is below it
        // holds no doc that can enter the page (equal scores are kept: the docid decides)
        bound = tk.gbound;
        if (tk.have >= tk.needed && tk.bound_key > bound) bound = tk.bound_key;
      }
      u32x4 raw[T];
#pragma unroll
      for (int i = 0; i < T; ++i) raw[i] = u32x4{0u, 0u, 0u, 0u};
      if (prune) fast_load_blockmax<T>(fq, tile, lane, raw);  // (also before the first bound exists: no branch on it)
      // ---- A. the operands of this lane's 256 doc slots, combined in registers; in two halves of 128 slots, so that
      //         a load is 16 bytes per lane. The first three operands (both halves: six loads) are requested back to back
      //         and without control flow between them (an operand the query does not have re-reads its last one: same
      //         address, no new traffic), so that they — and the block-max loads in front of them — are ONE round trip; a
      //         wave-uniform branch or a register copy between two loads made the compiler wait for the first before it
      //         issued the second (1.03 -> 0.90 ms), and the halves one after the other were two round trips (-> 0.87).
      auto combine = [&](int h, uint32_t o, const u32x4 w) {
        const uint32_t code = fq->ops[o].code;
        const uint32_t x[4] = {w.x, w.y, w.z, w.w};
        const uint32_t kind = wave_uniform(code & 15u);
        // (scalar branches on the wave-uniform kind: written as selects this was three ALU ops + two v_cndmask per word)
        if (kind == kFastAnd) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[h * 4 + j] &= x[j];
        } else if (kind == kFastOr) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[h * 4 + j] |= x[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[h * 4 + j] &= ~x[j];
        }
        const uint32_t cmask = wave_uniform(code >> 8);
        if (cmask) {  // funnel counters taken after this operand (search_pipeline.h:58-65)
          uint32_t pc = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) pc += __popc(a[h * 4 + j]);
          if (cmask & 1u) cnt0 += pc;
          if (cmask & 2u) cnt1 += pc;
          if (cmask & 4u) cnt2 += pc;
          if (cmask & 8u) cnt3 += pc;
        }
      };
      const uint32_t o1 = min(1u, n_ops - 1u), o2 = min(2u, n_ops - 1u);
      u32x4 w[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        w[h][0] = load_half(0, tile, h);
        w[h][1] = load_half(o1, tile, h);
        w[h][2] = load_half(o2, tile, h);
      }
      if (bound != 0) qmask = fast_quarter_mask<T>(fq, lane, raw, bound);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        combine(h, 0, w[h][0]);
        if (n_ops > 1u) combine(h, 1, w[h][1]);
        if (n_ops > 2u) combine(h, 2, w[h][2]);
        for (uint32_t o = 3; o < n_ops; ++o) combine(h, o, load_half(o, tile, h));  // (rare: one at a time)
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) cnt_res += __popc(a[j]);  // every match counts ...
      if (qmask != 0xFFFFu && !MGX_ABLATE(bt, 128u)) {       // ... only the quarters that can reach the page are scored
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t m2 = qmask >> (2 * j);
          a[j] &= ((m2 & 1u) ? 0x0000FFFFu : 0u) | ((m2 & 2u) ? 0xFFFF0000u : 0u);
        }
      }
      if (MGX_ABLATE(bt, 2u)) {  // (ablation: operands + counts + pruning only)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0;
      }
    }

    // ---- B. queue the surviving 32-doc words as (first doc slot, bits) pairs — a ballot and a compacted write per word,
    //         nothing per match; C. whenever 64 pairs wait, every lane expands ONE pair into the slot buffer and the full
    //         rounds are scored. (Expanding per tile visit, as this kernel first did, ran the per-lane bit loops with a
    //         handful of lanes busy — after pruning most visits leave a few matches in a few lanes — and cost a prefix
    //         sum per visit.)
    const uint32_t slot0 = tile * kTileDocs + lane * 256u;
    bool scored = false;  // wave-uniform: a round was scored since this tile's quarter bits were made
#pragma nounroll
    for (uint32_t part = 0; part < 2u; ++part) {
      if (!flush) {
#ifndef MGX_NO_REMASK
        // The page's bound has moved while the first half's matches were scored (a heavy tile): the second half is held
        // against the new one before it is queued.
        if (part != 0u && scored && prune) {
          wave_topk_refresh_gbound(tk);
          uint64_t nb = tk.gbound;
          if (tk.have >= tk.needed && tk.bound_key > nb) nb = tk.bound_key;
          if (nb > bound) {
            bound = nb;
            const uint32_t qm = fast_remask<T>(fq, tile, lane, nb);
#pragma unroll
            for (int j = 4; j < 8; ++j) {
              const uint32_t m2 = qm >> (2 * j);
              a[j] &= ((m2 & 1u) ? 0x0000FFFFu : 0u) | ((m2 & 2u) ? 0xFFFF0000u : 0u);
            }
          }
        }
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = part != 0u ? a[4 + j] : a[j];
          const uint64_t m = __ballot(w != 0u);
          if (m != 0) {  // wave-uniform
            if (w != 0u) {
              const uint32_t at = np + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
              pairs[at] = make_uint2(slot0 + (part * 4u + j) * 32u, w);
            }
            np += static_cast<uint32_t>(__popcll(m));
          }
        }
      }
      for (;;) {  // batches of 64 pairs, newest first (the order of matches is immaterial)
        if (np < 64u && !(flush && (np != 0u || pend != 0u))) break;
        const uint32_t take = min(np, 64u);
        wave_lds_sync();
        uint32_t base = 0, bits = 0;
        if (lane < take) {
          const uint2 pv = pairs[np - take + lane];
          base = pv.x;
          bits = pv.y;
        }
        np -= take;
        for (;;) {  // ... expanded in chunks the slot buffer has room for
          uint32_t n_new;
          const uint32_t my_first = wave_excl_scan_total(static_cast<uint32_t>(__popc(bits)), &n_new);
          bool more = false;  // wave-uniform: bits of this batch left for the next chunk
          if (n_new != 0) {
            const uint32_t space = plan.ring - pend;
            uint32_t r = my_first;
            while (bits != 0 && r < space) {
              ring[pend + r] = base + static_cast<uint32_t>(__builtin_ctz(bits));
              bits &= bits - 1;
              ++r;
            }
            more = n_new > space;
            pend = more ? plan.ring : pend + n_new;
            wave_lds_sync();
          }
          const bool fin = flush && !more && np == 0u;  // the wave's last chunk: the partial round is scored too
          uint32_t g0 = 0;
          for (;;) {
            const uint32_t n = min(pend - g0, kFastRound);
            if (n == 0 || (n < kFastRound && !fin)) break;
            if (!MGX_ABLATE(bt, 1u)) score_round(g0, n);  // (ablation: enumerate, do not score)
#ifdef MGX_ABLATION
            if (MGX_ABLATE(bt, 64u) && lane == 0) cnt3 += n;  // (debug: after_filters reports the matches that were scored)
#endif
            g0 += n;
            scored = true;
          }
          if (g0 != 0) {  // move the (< kFastRound) leftover to the front: sources sit at >= kFastRound, destinations below
            const uint32_t left = pend - g0;
            uint32_t v[kFastPerLane];
#pragma unroll
            for (int m = 0; m < kFastPerLane; ++m) v[m] = m * 64 + lane < left ? ring[g0 + m * 64 + lane] : 0u;
            wave_lds_sync();
#pragma unroll
            for (int m = 0; m < kFastPerLane; ++m)
              if (m * 64 + lane < left) ring[m * 64 + lane] = v[m];
            pend = left;
            wave_lds_sync();
          }
          if (!more) break;
        }
      }
      if (flush) break;
    }
    if (flush) break;
  }

  {
    uint32_t v[5] = {cnt0, cnt1, cnt2, cnt3, cnt_res};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      uint32_t x = v[s];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
      if (lane == 0 && x) atomicAdd(&bt.counters[static_cast<uint64_t>(it.query) * 8 + s], (unsigned long long)x);
    }
  }

  // ---- merge the waves' lists into this workgroup's best `needed`, best first, to HBM ---------------------------------
  wave_topk_truncate(tk);
  if (lane == 0) misc[wave] = tk.have;
  __syncthreads();
  {
    const uint64_t* all_keys = reinterpret_cast<const uint64_t*>(smem + fo.tk_keys);
    const uint32_t* all_docs = reinterpret_cast<const uint32_t*>(smem + fo.tk_docs);
    const uint32_t cap = tk.cap, needed = tk.needed;
    uint32_t have[kFastWaves];
    uint32_t total = 0;
    for (int w = 0; w < kFastWaves; ++w) {
      have[w] = min(misc[w], needed);
      total += have[w];
    }
    const uint64_t obase = static_cast<uint64_t>(it.list) * bt.cand_stride;
    for (uint32_t e = tid; e < kFastWaves * cap; e += kFastBlock) {
      const uint32_t w = e / cap, i = e % cap;
      if (i >= have[w]) continue;
      const uint64_t k = all_keys[static_cast<size_t>(w) * 2 * cap + i];
      const uint32_t d = all_docs[static_cast<size_t>(w) * 2 * cap + i];
      uint32_t rank = i;  // entries of the other waves' (sorted) lists that beat this one, plus its own position
      for (uint32_t w2 = 0; w2 < kFastWaves; ++w2) {
        if (w2 == w) continue;
        const uint64_t* kk = all_keys + static_cast<size_t>(w2) * 2 * cap;
        const uint32_t* dd = all_docs + static_cast<size_t>(w2) * 2 * cap;
        uint32_t lo = 0, hi = have[w2];
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (better(kk[mid], dd[mid], k, d)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < needed) {
        bt.cand_keys[obase + rank] = k;
        bt.cand_docs[obase + rank] = d;
        // the needed-th best of the whole item (8 waves' matches merged) is a far tighter query-wide bound than any one
        // wave's: later items of the query start from it
        if (rank == needed - 1 && tk.gbound_ptr) atomicMax(tk.gbound_ptr, static_cast<unsigned long long>(k));
      }
    }
    if (tid == 0) bt.cand_n[it.list] = min(total, needed);
  }
}

// One kernel per number of scored terms (own register allocation each); a batch launches the ones it has items for.
template <int T>
#ifndef MGX_FOCC
#define MGX_FOCC 6
#endif
__global__ __launch_bounds__(kFastBlock, MGX_FOCC) void bitmap_score_kernel(DevIndex ix, DevBatch bt, FastPlan plan) {
  const DevItem it = bt.items[blockIdx.x];
  bitmap_score_body<T>(ix, bt, plan,
                       reinterpret_cast<FastQueryPtr>(reinterpret_cast<uint64_t>(bt.fast_queries + it.query)), it);
}

// ---------------------------------------------------------------------------------------------------------------
// wave-autonomous count pass of docid-ordered pages (kModeDocPage queries with flat programs)
// ---------------------------------------------------------------------------------------------------------------
//
// Pass 1 of a docid page only needs the number of matches per tile (and the funnel counters): the program runs in
// registers on 4 words per lane exactly as in wave_score_kernel's operand phase, nothing is written but one u32 per
// tile. Without sorted-list operands a wave keeps TWO tiles in flight (every operand's loads for both tiles are issued
// before either is combined), which is what hides the load latency at this register budget.

// This lane's four words of operand `lf` for `tile` (zeros when !active, a wave-uniform flag). Without kLists only
// bitmap-form and range operands occur, and nothing but the loads themselves stands between two calls.
template <bool kLists>
__device__ __forceinline__ void wave_operand_words(const DevIndex& ix, const DevBatch& bt, const DevLeaf lf,
                                                   uint32_t tile, bool active, uint64_t* scratch, uint64_t (&w)[4]) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = 0;
  if (!active) return;
  if (kLists) {
    uint32_t seg_rel;
    wave_fetch_operand(ix, bt, lf, tile, static_cast<uint64_t>(ix.first_doc_id) + static_cast<uint64_t>(tile) * kTileDocs,
                       scratch, w, &seg_rel);
  } else if (lf.kind == kLeafRange) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t s0 = static_cast<uint64_t>(tile) * kTileDocs + (static_cast<uint64_t>(lane) * 4 + k) * 64;
      const uint64_t ra = lf.a > s0 ? lf.a - s0 : 0, rb = lf.b > s0 ? lf.b - s0 : 0;
      const uint64_t hi_mask = rb >= 64 ? ~0ull : ((1ull << rb) - 1ull);
      const uint64_t lo_mask = ra >= 64 ? ~0ull : ((1ull << ra) - 1ull);
      w[k] = hi_mask & ~lo_mask;
    }
  } else {  // gram or filter bitmap
    const uint64_t* rowp = lf.kind == kLeafGramBitmap
                               ? ix.gram_bitmaps + tile * ix.gb_tile_stride + lf.b * ix.gb_row_stride + lane * 4
                               : ix.filter_bitmaps + tile * ix.fb_tile_stride + lf.b * ix.fb_row_stride + lane * 4;
    const uint4 v0 = *reinterpret_cast<const uint4*>(rowp);
    const uint4 v1 = *reinterpret_cast<const uint4*>(rowp + 2);
    w[0] = (static_cast<uint64_t>(v0.y) << 32) | v0.x;
    w[1] = (static_cast<uint64_t>(v0.w) << 32) | v0.z;
    w[2] = (static_cast<uint64_t>(v1.y) << 32) | v1.x;
    w[3] = (static_cast<uint64_t>(v1.w) << 32) | v1.z;
  }
}

// kTextDf: the df pass of a text-level term (kModeTextDf queries with flat programs) — instead of a count per tile, the
// tile's candidates are enumerated into a per-wave LDS buffer and every lane looks for the term in one candidate's text
// (PopulateTermDocumentFrequency, search_pipeline.cpp:556-563); the hits are summed into counter slot 5.
constexpr uint32_t kDfMatchBuf = 512;

template <bool kLists, bool kTextDf>
__global__ __launch_bounds__(kBlock) void wave_count_kernel(DevIndex ix, DevBatch bt, WavePlan plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  DevLeaf* const leaf = reinterpret_cast<DevLeaf*>(smem);
  uint32_t* const prog = reinterpret_cast<uint32_t*>(smem + align8(plan.max_leaves * static_cast<uint32_t>(sizeof(DevLeaf))));
  uint64_t* const scratch_all = reinterpret_cast<uint64_t*>(
      smem + align8(plan.max_leaves * static_cast<uint32_t>(sizeof(DevLeaf))) + align8(plan.max_instr * 4));
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  uint64_t* const scratch = scratch_all + (kLists ? static_cast<size_t>(wave) * kWordsPerTile : 0);
  constexpr int kT = (kLists || kTextDf) ? 1 : 2;  // tiles in flight per wave
  constexpr int kWaves = kBlock / 64;
  uint16_t* const mbuf = reinterpret_cast<uint16_t*>(scratch_all + (kLists ? kWaves * kWordsPerTile : 0)) +
                         static_cast<size_t>(wave) * kDfMatchBuf;

  const DevItem it = bt.items[blockIdx.x];
  const uint32_t qi = it.query;
  const DevQuery q = bt.queries[qi];
  for (uint32_t i = tid; i < q.n_leaves; i += kBlock) leaf[i] = bt.leaves[q.leaf_begin + i];
  for (uint32_t i = tid; i < q.n_instr; i += kBlock) prog[i] = bt.prog[q.prog_begin + i];
  __syncthreads();
  TextPattern df_pattern{nullptr, 0, 0, 0, 0, 0};
  if (kTextDf) df_pattern = text_pattern(bt.patterns + q.pat_off, q.pat_len);

  uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0, cnt_res = 0, cnt_df = 0;
  const uint32_t tile_begin = it.tile_begin;
  const uint32_t tile_end = min(tile_begin + it.n_tiles, ix.n_tiles);
  for (uint32_t t0 = tile_begin + wave * kT; t0 < tile_end; t0 += kWaves * kT) {
    uint64_t acc[kT][4];
#pragma unroll
    for (int u = 0; u < kT; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[u][k] = 0;
    for (uint32_t pc = 0; pc < q.n_instr; ++pc) {
      const uint32_t ins = prog[pc];
      const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
      if (op == kOpCount) {
        uint32_t pcnt = 0;
#pragma unroll
        for (int u = 0; u < kT; ++u)
          pcnt += __popcll(acc[u][0]) + __popcll(acc[u][1]) + __popcll(acc[u][2]) + __popcll(acc[u][3]);
        cnt0 += (arg & 1u) ? pcnt : 0;
        cnt1 += (arg & 2u) ? pcnt : 0;
        cnt2 += (arg & 4u) ? pcnt : 0;
        cnt3 += (arg & 8u) ? pcnt : 0;
        continue;
      }
      const DevLeaf lf = leaf[arg];
      uint64_t w[kT][4];
#pragma unroll
      for (int u = 0; u < kT; ++u) wave_operand_words<kLists>(ix, bt, lf, t0 + u, t0 + u < tile_end, scratch, w[u]);
#pragma unroll
      for (int u = 0; u < kT; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (op == kOpLoad) acc[u][k] = w[u][k];
          else if (op == kOpAnd) acc[u][k] &= w[u][k];
          else if (op == kOpOr) acc[u][k] |= w[u][k];
          else if (op == kOpAndNot) acc[u][k] &= ~w[u][k];
        }
      }
    }
    for (int u = 0; u < kT; ++u) {  // (kT is 1 or 2: unrolled by the compiler where it can be)
      const uint32_t tile = t0 + u;
      if (tile < tile_end) {  // wave-uniform
        uint32_t c = __popcll(acc[u][0]) + __popcll(acc[u][1]) + __popcll(acc[u][2]) + __popcll(acc[u][3]);
        cnt_res += c;
        if (kTextDf) {
          for (;;) {  // rounds of at most kDfMatchBuf candidates
            uint32_t n_left;
            const uint32_t mine =
                __popcll(acc[u][0]) + __popcll(acc[u][1]) + __popcll(acc[u][2]) + __popcll(acc[u][3]);
            uint32_t r = wave_excl_scan_total(mine, &n_left);
            if (n_left == 0) break;  // wave-uniform
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              while (acc[u][k] != 0 && r < kDfMatchBuf) {
                const uint32_t bit = __builtin_ctzll(acc[u][k]);
                acc[u][k] &= acc[u][k] - 1;
                mbuf[r] = static_cast<uint16_t>((lane << 8) | (k << 6) | bit);
                ++r;
              }
            }
            wave_lds_sync();
            const uint32_t nm = min(kDfMatchBuf, n_left);
            for (uint32_t j = lane; j < nm; j += 64) {
              const uint32_t slot = tile * kTileDocs + mbuf[j];
              cnt_df += text_count_occurrences(ix.text, ix.text_off[slot], ix.text_off[slot + 1], df_pattern, true);
            }
            wave_lds_sync();
          }
        } else {
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
          if (lane == 0) bt.tile_cnt[static_cast<uint64_t>(q.out_slot) * ix.n_tiles + tile] = c;
        }
      }
    }
  }
  {
    uint32_t v[6] = {cnt0, cnt1, cnt2, cnt3, cnt_res, cnt_df};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      uint32_t x = v[s];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
      if (lane == 0 && x) atomicAdd(&bt.counters[static_cast<uint64_t>(qi) * 8 + s], (unsigned long long)x);
    }
  }
}

// Pass 2 of a docid page for flat programs: one workgroup per query, its four waves take the page's non-empty tiles in
// turn (tile counts are read 256 at a time into ballot masks, every wave for itself: no barrier in the kernel after the
// program is staged), re-evaluate them in registers and write the doc ids at their rank positions.
template <bool kLists>
__global__ __launch_bounds__(kBlock) void wave_page_kernel(DevIndex ix, DevBatch bt, WavePlan plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  DevLeaf* const leaf = reinterpret_cast<DevLeaf*>(smem);
  uint32_t* const prog = reinterpret_cast<uint32_t*>(smem + align8(plan.max_leaves * static_cast<uint32_t>(sizeof(DevLeaf))));
  uint64_t* const scratch_all = reinterpret_cast<uint64_t*>(
      smem + align8(plan.max_leaves * static_cast<uint32_t>(sizeof(DevLeaf))) + align8(plan.max_instr * 4));
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  uint64_t* const scratch = scratch_all + (kLists ? static_cast<size_t>(wave) * kWordsPerTile : 0);
  constexpr uint32_t kWaves = kBlock / 64;

  const uint32_t qi = bt.items[blockIdx.x].query;
  const DevQuery q = bt.queries[qi];
  const uint64_t total = bt.totals[q.out_slot];
  const uint64_t take = total < q.limit ? total : q.limit;
  if (take == 0) return;
  const uint64_t lo = q.descending ? total - take : 0, hi = q.descending ? total : take;
  const uint64_t* ts = bt.tile_start + static_cast<uint64_t>(q.out_slot) * ix.n_tiles;
  const uint32_t* tc = bt.tile_cnt + static_cast<uint64_t>(q.out_slot) * ix.n_tiles;
  uint32_t a = 0, b = ix.n_tiles;  // last tile whose start rank is <= lo
  while (b - a > 1) {
    const uint32_t mid = (a + b) >> 1;
    if (ts[mid] <= lo) a = mid; else b = mid;
  }
  uint32_t c = a, d = ix.n_tiles;  // first tile past `a` whose start rank is >= hi
  while (c < d) {
    const uint32_t mid = (c + d) >> 1;
    if (ts[mid] >= hi) d = mid; else c = mid + 1;
  }
  const uint32_t tile_begin = a, tile_end = max(c, a + 1);
  for (uint32_t i = tid; i < q.n_leaves; i += kBlock) leaf[i] = bt.leaves[q.leaf_begin + i];
  for (uint32_t i = tid; i < q.n_instr; i += kBlock) prog[i] = bt.prog[q.prog_begin + i];
  __syncthreads();

  uint32_t* const out = bt.page_docs + static_cast<uint64_t>(q.out_slot) * bt.page_stride;
  uint32_t seen = 0;  // non-empty tiles passed so far; this wave takes those with seen % kWaves == wave
  for (uint32_t base = tile_begin; base < tile_end; base += 256) {
#pragma unroll 1
    for (uint32_t sub = 0; sub < 4; ++sub) {
      const uint32_t t = base + sub * 64 + lane;
      uint64_t m = __ballot(t < tile_end && tc[t] != 0);
      while (m) {  // wave-uniform
        const uint32_t tile = base + sub * 64 + static_cast<uint32_t>(__builtin_ctzll(m));
        m &= m - 1;
        if ((seen++ % kWaves) != wave) continue;
        uint64_t acc[4] = {0, 0, 0, 0};
        for (uint32_t pc = 0; pc < q.n_instr; ++pc) {
          const uint32_t ins = prog[pc];
          const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
          if (op == kOpCount) continue;
          uint64_t w[4];
          wave_operand_words<kLists>(ix, bt, leaf[arg], tile, true, scratch, w);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (op == kOpLoad) acc[k] = w[k];
            else if (op == kOpAnd) acc[k] &= w[k];
            else if (op == kOpOr) acc[k] |= w[k];
            else if (op == kOpAndNot) acc[k] &= ~w[k];
          }
        }
        const uint32_t mine = __popcll(acc[0]) + __popcll(acc[1]) + __popcll(acc[2]) + __popcll(acc[3]);
        uint64_t rank = ts[tile] + (wave_incl_scan(mine) - mine);
        const uint32_t doc0 = ix.first_doc_id + tile * kTileDocs + lane * 256;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint64_t bits = acc[k];
          while (bits) {
            const uint32_t bpos = __builtin_ctzll(bits);
            bits &= bits - 1;
            if (rank >= lo && rank < hi) out[q.descending ? total - 1 - rank : rank] = doc0 + k * 64 + bpos;
            ++rank;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// cand_kernel: candidate-driven evaluation of SELECTIVE flat queries (a sparse sorted posting list drives)
// ---------------------------------------------------------------------------------------------------------------
//
// The reference stops scanning as soon as a conjunction is small: Execute probes the remaining terms instead of
// intersecting them once <= 1000 candidates are left (search_pipeline.cpp:828-829 -> Index::FilterByNgrams ->
// PostingList::RetainPresent, posting_list.cpp:432-474), SearchAnd sorts its lists smallest-first and exits when the
// accumulator is empty (index.cpp:228-240,338-351), Intersect gallops (posting_list.cpp:634-715). The tile kernels of
// this file evaluate every tile of every operand whatever the sizes: a query whose smallest list has a thousand postings
// walked all 611 tiles of all its operands. Here the smallest positive operand — a sorted u32 posting array that is too
// sparse to have a bitmap row — IS the candidate set:
//   * one workgroup of sixteen waves per query, each wave takes a contiguous share of the driver list, 128 postings per
//     step, two per lane (coalesced 256-byte loads; the driver's tf byte sits at the same index);
//   * every other operand is PROBED per candidate, in program order: a bitmap-form operand (dense gram, filter) by one
//     8-byte gather of the word that holds the doc's bit; another sorted list by a binary search inside the candidate's
//     tile segment (skip row) — the galloping step of a merge whose other side is a few postings long; NOT terms and NE
//     filters invert the test. A wave whose 64 candidates are all dead skips the rest of the program (the early exit);
//   * funnel counters are popcounts of the alive mask where the tile program has its COUNT instructions — exact, because
//     the host only sends a query here when its driver is loaded before the first COUNT (terms arrive smallest-first,
//     search_pipeline.cpp:2012-2014, so the driver is a gram of the first term);
//   * SORT _score: survivors are scored in place — tf from the driver's own tf column, from the doc-slot nibble row of a
//     dense gram, or by the same segment search for another sparse gram; BM25 in the reference's fp64 operation order
//     (bm25_scorer.cpp:73-88) — and offered to the per-wave top-k; docid-ordered pages: pass 1 counts, a workgroup
//     barrier turns the waves' counts into rank offsets, pass 2 re-probes only the waves that hold ranks of the page and
//     writes the doc ids at their rank positions.
// Bytes touched per query: 5 B per driver posting + one 64-byte sector per (candidate still alive, operand) instead of
// 2 KiB per (tile, operand).
// 16 waves per query and two candidates per lane in flight: a probe is a dependent global load (a microsecond under
// load), and with four waves and one candidate per lane a 48,000-posting driver was a chain of ~1,000 such loads per wave.
constexpr int kCandBlock = 1024;
constexpr int kCandWaves = kCandBlock / 64;
constexpr int kCandU = 2;

struct CandOffsets {
  uint32_t leaf, prog, misc, tk_keys, tk_docs, total;
};
__host__ __device__ inline CandOffsets carve_cand(uint32_t max_leaves, uint32_t max_instr, uint32_t max_cap) {
  CandOffsets o;
  uint32_t at = 0;
  o.leaf = at;     at += align8(max_leaves * static_cast<uint32_t>(sizeof(DevLeaf)));
  o.prog = at;     at += align8(max_instr * 4);
  o.misc = at;     at += kCandWaves * 8 * 4;
  o.tk_keys = at;  at += kCandWaves * 2 * max_cap * 8;
  o.tk_docs = at;  at += kCandWaves * 2 * max_cap * 4;
  o.total = at;
  return o;
}
uint32_t CandLdsBytes(uint32_t max_leaves, uint32_t max_instr, uint32_t max_cap) {
  return carve_cand(max_leaves ? max_leaves : 1, max_instr ? max_instr : 1, max_cap).total;
}

// Is the doc at `slot` (tile `tile`, doc id `d`) a member of operand `lf`?
__device__ __forceinline__ bool cand_member(const DevIndex& ix, const DevLeaf lf, uint32_t d, uint32_t slot, uint32_t tile) {
  if (lf.kind == kLeafGramBitmap || lf.kind == kLeafFilterBitmap) {
    const uint32_t w = (slot & (kTileDocs - 1)) >> 6;
    const uint64_t* wp = lf.kind == kLeafGramBitmap
                             ? ix.gram_bitmaps + tile * ix.gb_tile_stride + lf.b * ix.gb_row_stride + w
                             : ix.filter_bitmaps + tile * ix.fb_tile_stride + lf.b * ix.fb_row_stride + w;
    return ((*wp >> (slot & 63)) & 1ull) != 0;
  }
  if (lf.kind == kLeafList) {
    const uint64_t l0 = ix.offsets[lf.a];
    uint64_t lo = l0, hi = ix.offsets[lf.a + 1];
    if (lf.row != kNoRow) {
      const uint32_t* r = ix.tile_off + static_cast<uint64_t>(lf.row) * (ix.n_tiles + 1);
      hi = l0 + r[tile + 1];
      lo = l0 + r[tile];
    }
    const uint64_t p = lower_bound_u32(ix.docids, lo, hi, d);
    return p < hi && ix.docids[p] == d;
  }
  if (lf.kind == kLeafRange) return slot >= lf.a && slot < lf.b;
  return false;
}

// The flat program on kCandU candidates per lane (alive[u]: the lane holds a posting of the driver). On return alive[u]
// marks the survivors; the funnel counts of the wave are added to cnt[0..3].
__device__ __forceinline__ void cand_run_program(const DevIndex& ix, const DevLeaf* leaf, const uint32_t* prog,
                                                 uint32_t n_instr, uint32_t driver_leaf, bool (&alive)[kCandU],
                                                 const uint32_t (&d)[kCandU], uint32_t (&cnt)[4]) {
  for (uint32_t pc = 0; pc < n_instr; ++pc) {
    const uint32_t ins = prog[pc];
    const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
    uint64_t any = 0;
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < kCandU; ++u) {
      const uint64_t m = __ballot(alive[u]);
      any |= m;
      c += static_cast<uint32_t>(__popcll(m));
    }
    if (op == kOpCount) {
      cnt[0] += (arg & 1u) ? c : 0;
      cnt[1] += (arg & 2u) ? c : 0;
      cnt[2] += (arg & 4u) ? c : 0;
      cnt[3] += (arg & 8u) ? c : 0;
      continue;
    }
    if (arg == driver_leaf) continue;  // (LOAD / AND of the candidate set itself)
    if (any == 0) break;               // nothing left: the rest of the program counts zeros
    const DevLeaf lf = leaf[arg];
    bool in[kCandU];
#pragma unroll
    for (int u = 0; u < kCandU; ++u) {  // (the probes of the lane's candidates are independent loads: in flight together)
      const uint32_t slot = d[u] - ix.first_doc_id;
      in[u] = alive[u] && cand_member(ix, lf, d[u], slot, slot >> kTileShift);
    }
#pragma unroll
    for (int u = 0; u < kCandU; ++u) alive[u] = op == kOpAndNot ? (alive[u] && !in[u]) : in[u];
  }
}

template <int MODE>  // kModeScore | kModeDocPage
__global__ __launch_bounds__(kCandBlock) void cand_kernel(DevIndex ix, DevBatch bt, uint32_t max_leaves, uint32_t max_instr,
                                                          uint32_t max_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const CandOffsets co = carve_cand(max_leaves, max_instr, max_cap);
  DevLeaf* const leaf = reinterpret_cast<DevLeaf*>(smem + co.leaf);
  uint32_t* const prog = reinterpret_cast<uint32_t*>(smem + co.prog);
  uint32_t* const wcnt = reinterpret_cast<uint32_t*>(smem + co.misc);  // [kCandWaves][8]
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  constexpr uint32_t kStep = 64 * kCandU;

  const DevItem it = bt.items[blockIdx.x];
  const uint32_t qi = it.query, driver_leaf = it.tile_begin;
  const DevQuery q = bt.queries[qi];
  for (uint32_t i = tid; i < q.n_leaves; i += kCandBlock) leaf[i] = bt.leaves[q.leaf_begin + i];
  for (uint32_t i = tid; i < q.n_instr; i += kCandBlock) prog[i] = bt.prog[q.prog_begin + i];
  __syncthreads();
  const uint32_t dgram = leaf[driver_leaf].a;
  const uint64_t l0 = ix.offsets[dgram], l1 = ix.offsets[dgram + 1];
  // this wave's contiguous share of the driver list, in whole steps
  const uint64_t steps = (l1 - l0 + kStep - 1) / kStep, per = (steps + kCandWaves - 1) / kCandWaves;
  const uint64_t wa = min(l1, l0 + wave * per * kStep), wb = min(l1, wa + per * kStep);

  uint32_t cnt[4] = {0, 0, 0, 0};
  uint32_t cnt_res = 0;
  const bool desc = q.descending != 0;

  WaveTopK tk;
  if (MODE == kModeScore) {
    tk.cap = q.cap;
    tk.needed = q.needed;
    tk.lds_sort = false;
    tk.keys = reinterpret_cast<uint64_t*>(smem + co.tk_keys) + static_cast<size_t>(wave) * 2 * q.cap;
    tk.docs = reinterpret_cast<uint32_t*>(smem + co.tk_docs) + static_cast<size_t>(wave) * 2 * q.cap;
    tk.have = 0;
    tk.pend = 0;
    tk.bound_key = 0;
    tk.bound_doc = 0;
    tk.gbound_ptr = nullptr;  // (one workgroup holds the whole query: nothing to share)
    tk.gbound = 0;
    for (uint32_t i = lane; i < 2 * q.cap; i += 64) {
      tk.keys[i] = 0;
      tk.docs[i] = 0;
    }
    wave_lds_sync();
  }

  for (uint64_t p0 = wa; p0 < wb; p0 += kStep) {
    bool alive[kCandU];
    uint32_t d[kCandU];
#pragma unroll
    for (int u = 0; u < kCandU; ++u) {
      const uint64_t p = p0 + u * 64 + lane;
      alive[u] = p < wb;
      d[u] = alive[u] ? ix.docids[p] : ix.first_doc_id;
    }
    cand_run_program(ix, leaf, prog, q.n_instr, driver_leaf, alive, d, cnt);
#pragma unroll
    for (int u = 0; u < kCandU; ++u) {
      const uint64_t amask = __ballot(alive[u]);
      cnt_res += static_cast<uint32_t>(__popcll(amask));
      if (MODE == kModeScore && amask != 0) {
        // BM25Scorer::ScoreDocuments for the survivors, term by term in the query's order (bm25_scorer.cpp:73-88)
        const uint64_t p = p0 + u * 64 + lane;
        const uint32_t slot = d[u] - ix.first_doc_id;
        double score = 0.0;
        uint32_t dl = 0;
        if (alive[u]) {
          dl = ix.dl8[slot];
          if (dl == 255u) dl = ix.doc_len[slot];
        }
        const double length_norm = q.one_minus_b + q.b * static_cast<double>(dl) / q.avgdl_clamped;
        for (uint32_t i = 0; i < q.n_score; ++i) {
          const DevScoreTerm st = bt.score_terms[q.score_begin + i];
          const DevLeaf lf = leaf[st.leaf];
          uint32_t tfv = 0;
          if (alive[u]) {
            if (st.leaf == driver_leaf) {
              tfv = posting_tf(ix, p);
            } else if (lf.kind == kLeafGramBitmap) {
              const uint32_t nb = ix.tfnib[static_cast<uint64_t>(lf.b) * ix.nib_row_stride + (slot >> 1)];
              tfv = (nb >> ((slot & 1u) * 4u)) & 15u;
              if (tfv == 15u) tfv = exact_tf(ix, lf.a, lf.row, slot);
            } else if (lf.kind == kLeafList) {
              tfv = exact_tf(ix, lf.a, lf.row, slot);
            }
          }
          if (tfv != 0) {
            const double tf = static_cast<double>(tfv);
            const double numerator = tf * q.k1_plus_1;
            const double denominator = tf + q.k1 * length_norm;
            score += st.idf * numerator / denominator;
          }
        }
        wave_topk_offer(tk, alive[u], score_key(score, desc), desc ? d[u] : ~d[u]);
      }
    }
  }

  // ---- counters: one workgroup holds the whole query ------------------------------------------------------------------
  if (lane == 0) {
    wcnt[wave * 8 + 0] = cnt[0];
    wcnt[wave * 8 + 1] = cnt[1];
    wcnt[wave * 8 + 2] = cnt[2];
    wcnt[wave * 8 + 3] = cnt[3];
    wcnt[wave * 8 + 4] = cnt_res;
  }
  if (MODE == kModeScore) {
    wave_topk_truncate(tk);
    if (lane == 0) wcnt[wave * 8 + 5] = tk.have;
  }
  __syncthreads();
  if (tid < 5) {
    unsigned long long v = 0;
    for (int w = 0; w < kCandWaves; ++w) v += wcnt[w * 8 + tid];
    bt.counters[static_cast<uint64_t>(qi) * 8 + tid] = v;
  }

  if (MODE == kModeScore) {
    // the waves' lists -> this query's one candidate list, best first (rank by counting the better entries elsewhere)
    const uint64_t* all_keys = reinterpret_cast<const uint64_t*>(smem + co.tk_keys);
    const uint32_t* all_docs = reinterpret_cast<const uint32_t*>(smem + co.tk_docs);
    const uint32_t cap = q.cap;
    uint32_t total = 0;
    for (int w = 0; w < kCandWaves; ++w) total += min(wcnt[w * 8 + 5], q.needed);
    const uint64_t obase = static_cast<uint64_t>(it.list) * bt.cand_stride;
    for (uint32_t e = tid; e < kCandWaves * cap; e += kCandBlock) {
      const uint32_t w = e / cap, i = e % cap;
      if (i >= min(wcnt[w * 8 + 5], q.needed)) continue;
      const uint64_t k = all_keys[static_cast<size_t>(w) * 2 * cap + i];
      const uint32_t dd0 = all_docs[static_cast<size_t>(w) * 2 * cap + i];
      uint32_t rank = i;
      for (uint32_t w2 = 0; w2 < static_cast<uint32_t>(kCandWaves) && rank < q.needed; ++w2) {
        if (w2 == w) continue;
        const uint64_t* kk = all_keys + static_cast<size_t>(w2) * 2 * cap;
        const uint32_t* dd = all_docs + static_cast<size_t>(w2) * 2 * cap;
        uint32_t lo = 0, hi = min(wcnt[w2 * 8 + 5], q.needed);
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (better(kk[mid], dd[mid], k, dd0)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < q.needed) {
        bt.cand_keys[obase + rank] = k;
        bt.cand_docs[obase + rank] = dd0;
      }
    }
    if (tid == 0) bt.cand_n[it.list] = min(total, q.needed);
    return;
  }

  // ---- docid-ordered page: ranks of the survivors, then the page -----------------------------------------------------
  uint64_t total = 0, my_start = 0;
  for (int w = 0; w < kCandWaves; ++w) {
    if (static_cast<uint32_t>(w) == wave) my_start = total;
    total += wcnt[w * 8 + 4];
  }
  if (tid == 0) const_cast<uint64_t*>(bt.totals)[q.out_slot] = total;
  const uint64_t take = total < q.limit ? total : q.limit;
  if (take == 0) return;
  const uint64_t lo = desc ? total - take : 0, hi = desc ? total : take;
  if (my_start >= hi || my_start + cnt_res <= lo) return;  // wave-uniform: none of this wave's ranks is on the page
  uint32_t* const out = bt.page_docs + static_cast<uint64_t>(q.out_slot) * bt.page_stride;
  uint64_t rank0 = my_start;
  uint32_t dummy[4] = {0, 0, 0, 0};
  for (uint64_t p0 = wa; p0 < wb && rank0 < hi; p0 += kStep) {
    bool alive[kCandU];
    uint32_t d[kCandU];
#pragma unroll
    for (int u = 0; u < kCandU; ++u) {
      const uint64_t p = p0 + u * 64 + lane;
      alive[u] = p < wb;
      d[u] = alive[u] ? ix.docids[p] : ix.first_doc_id;
    }
    cand_run_program(ix, leaf, prog, q.n_instr, driver_leaf, alive, d, dummy);
#pragma unroll
    for (int u = 0; u < kCandU; ++u) {
      const uint64_t amask = __ballot(alive[u]);
      if (alive[u]) {
        const uint64_t rank = rank0 + static_cast<uint64_t>(__popcll(amask & ((1ull << lane) - 1ull)));
        if (rank >= lo && rank < hi) out[desc ? total - 1 - rank : rank] = d[u];
      }
      rank0 += static_cast<uint64_t>(__popcll(amask));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// merge_score_kernel: SORT _score over LONG sorted posting arrays (a list-form operand too long to probe per candidate)
// ---------------------------------------------------------------------------------------------------------------
//
// The reference intersects sorted arrays by merging / galloping (PostingList::Intersect, posting_list.cpp:634-715) and
// then scores every result by scanning its text. Round 2 ran these queries on wave_score_kernel's list path: every
// operand scattered into an LDS bitmap, the matches enumerated out of the AND of the bitmaps — and the tf of a match
// under a list-form term found by a BINARY SEARCH of the posting array per (match, term): ~12 dependent global loads,
// 182M matches x 3 terms per benchmark batch on an index without bitmaps (24.7 ms, 16 % of the HBM roofline on the
// algorithmic bytes, 1.73x re-read traffic). This kernel is the tile-synchronous merge that carries posting RANKS:
//   * a wave owns whole 16384-doc tiles; the smallest positive list of the query DRIVES: its tile segment (skip row) is
//     read 64 postings at a time, one per lane — doc id and, at the same index, tf byte: coalesced, read once;
//   * every other operand of the tile is staged in this wave's LDS as a 2 KiB bitmap: a sorted list by scattering its
//     segment (coalesced 16-byte loads, ids of one word merged per lane before the ds_or), a bitmap-form gram or filter
//     by copying its row; for a scored list the per-word prefix popcounts go beside it (one DPP scan per tile);
//   * a candidate is tested against the staged bitmaps in program order (LDS bit tests, NOT terms and NE filters
//     inverted; funnel counters are popcounts of the alive mask at the program's COUNT positions — the host sends a
//     query here only when its driver is loaded before the first COUNT);
//   * a survivor's tf under a list-form term is tf[segment base + prefix[word] + popcount(word below its bit)]: two LDS
//     reads and ONE byte gather into the tile's own tf segment, no search; under a bitmap-form term the doc-slot nibble;
//     BM25 in the reference's fp64 operation order; per-wave top-k with the query-wide bound.
// Every posting of every operand is read exactly once per tile visit; nothing is searched.
constexpr int kMergeBlock = 512;
constexpr int kMergeWaves = kMergeBlock / 64;
constexpr int kMergeU = 4;               // driver postings per lane and step
constexpr uint32_t kMergeQueue = 64 + 64 * kMergeU;  // survivors waiting for a full scoring round

struct MergeOffsets {
  uint32_t leaf, prog, slot_of, wave0, w_bm, w_pref, w_seg, w_queue, w_keys, w_docs, wave_bytes, total;
};
__host__ __device__ inline MergeOffsets carve_merge(uint32_t max_leaves, uint32_t max_instr, uint32_t max_ops, uint32_t max_cap) {
  MergeOffsets o;
  uint32_t at = 0;
  o.leaf = at;     at += align8(max_leaves * static_cast<uint32_t>(sizeof(DevLeaf)));
  o.prog = at;     at += align8(max_instr * 4);
  o.slot_of = at;  at += align8(max_leaves);
  at = (at + 15u) & ~15u;
  o.wave0 = at;
  uint32_t w = 0;
  o.w_bm = w;      w += max_ops * kWordsPerTile * 8;
  o.w_pref = w;    w += max_ops * kWordsPerTile * 2;
  o.w_seg = w;     w += align8(max_ops * 8);
  o.w_queue = w;   w += kMergeQueue * 4;
  o.w_keys = w;    w += 2 * max_cap * 8;
  o.w_docs = w;    w += 2 * max_cap * 4;
  o.wave_bytes = (w + 15u) & ~15u;
  o.total = at + kMergeWaves * o.wave_bytes + 64;
  return o;
}
uint32_t MergeLdsBytes(uint32_t max_leaves, uint32_t max_instr, uint32_t max_ops, uint32_t max_cap) {
  return carve_merge(max_leaves ? max_leaves : 1, max_instr ? max_instr : 1, max_ops ? max_ops : 1, max_cap).total;
}

__global__ __launch_bounds__(kMergeBlock) void merge_score_kernel(DevIndex ix, DevBatch bt, uint32_t max_leaves,
                                                                  uint32_t max_instr, uint32_t max_ops, uint32_t max_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const MergeOffsets mo = carve_merge(max_leaves, max_instr, max_ops, max_cap);
  DevLeaf* const leaf = reinterpret_cast<DevLeaf*>(smem + mo.leaf);
  uint32_t* const prog = reinterpret_cast<uint32_t*>(smem + mo.prog);
  uint8_t* const slot_of = smem + mo.slot_of;  // leaf -> staged operand slot (0xFF: the driver / unused)
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  unsigned char* const wbase = smem + mo.wave0 + static_cast<size_t>(wave) * mo.wave_bytes;
  uint64_t* const bm = reinterpret_cast<uint64_t*>(wbase + mo.w_bm);      // [ops][256]
  uint16_t* const pref = reinterpret_cast<uint16_t*>(wbase + mo.w_pref);  // [ops][256]
  uint64_t* const seg = reinterpret_cast<uint64_t*>(wbase + mo.w_seg);    // [ops] posting index of the tile's first posting
  uint32_t* const wq = reinterpret_cast<uint32_t*>(wbase + mo.w_queue);   // [kMergeQueue] survivors waiting to be scored
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + mo.wave0 + kMergeWaves * mo.wave_bytes);

  const DevItem it = bt.items[blockIdx.x];
  const uint32_t qi = it.query;
  const DevQuery q = bt.queries[qi];
  for (uint32_t i = tid; i < q.n_leaves; i += kMergeBlock) leaf[i] = bt.leaves[q.leaf_begin + i];
  for (uint32_t i = tid; i < q.n_instr; i += kMergeBlock) prog[i] = bt.prog[q.prog_begin + i];
  __syncthreads();
  // the driver: the host put its leaf index into DevQuery::pat_off (unused by score-mode queries)
  const uint32_t driver_leaf = q.pat_off;
  if (tid == 0) {
    uint32_t n = 0;  // (at most kMergeMaxOps: the host checked)
    for (uint32_t i = 0; i < q.n_leaves; ++i) slot_of[i] = 0xFF;
    for (uint32_t pc = 0; pc < q.n_instr; ++pc) {
      const uint32_t op = prog[pc] >> 24, arg = prog[pc] & 0xFFFFFFu;
      if (op == kOpCount || arg == driver_leaf || slot_of[arg] != 0xFF) continue;
      slot_of[arg] = static_cast<uint8_t>(n++);
    }
  }
  __syncthreads();
  // which staged operands need ranks (a scored list-form term)
  uint32_t need_rank = 0;
  for (uint32_t i = 0; i < q.n_score; ++i) {
    const uint32_t lfi = bt.score_terms[q.score_begin + i].leaf;
    if (lfi != driver_leaf && leaf[lfi].kind == kLeafList) need_rank |= 1u << slot_of[lfi];
  }

  WaveTopK tk;
  tk.cap = q.cap;
  tk.needed = q.needed;
  tk.lds_sort = false;
  tk.keys = reinterpret_cast<uint64_t*>(wbase + mo.w_keys);
  tk.docs = reinterpret_cast<uint32_t*>(wbase + mo.w_docs);
  tk.have = 0;
  tk.pend = 0;
  tk.bound_key = 0;
  tk.bound_doc = 0;
  tk.gbound_ptr = bt.bounds ? bt.bounds + qi : nullptr;
  tk.gbound = 0;
  for (uint32_t i = lane; i < 2 * q.cap; i += 64) {
    tk.keys[i] = 0;
    tk.docs[i] = 0;
  }
  wave_lds_sync();

  const DevLeaf dlf = leaf[driver_leaf];
  const uint64_t dl0 = ix.offsets[dlf.a], dl1 = ix.offsets[dlf.a + 1];
  const uint32_t* const drow = dlf.row != kNoRow ? ix.tile_off + static_cast<uint64_t>(dlf.row) * (ix.n_tiles + 1) : nullptr;
  uint32_t cnt[4] = {0, 0, 0, 0};
  uint32_t cnt_res = 0;
  const bool desc = q.descending != 0;
  const uint32_t tile_end = min(it.tile_begin + it.n_tiles, ix.n_tiles);

  for (uint32_t tile = it.tile_begin + wave; tile < tile_end; tile += kMergeWaves) {
    const uint64_t tile_first = static_cast<uint64_t>(ix.first_doc_id) + static_cast<uint64_t>(tile) * kTileDocs;
    uint64_t da, db;
    if (drow) {
      da = dl0 + drow[tile];
      db = dl0 + drow[tile + 1];
    } else {
      da = lower_bound_u32(ix.docids, dl0, dl1, tile_first);
      db = lower_bound_u32(ix.docids, da, dl1, tile_first + kTileDocs);
    }
    if (da == db) continue;  // wave-uniform: no candidate in this tile
    wave_topk_refresh_gbound(tk);
    // ---- stage the other operands of the tile ---------------------------------------------------------------------------
    for (uint32_t li = 0; li < q.n_leaves; ++li) {
      const uint32_t j = slot_of[li];
      if (j == 0xFF) continue;
      const DevLeaf lf = leaf[li];
      uint64_t* const b = bm + static_cast<size_t>(j) * kWordsPerTile;
      if (lf.kind == kLeafList) {
        const uint64_t l0 = ix.offsets[lf.a], l1 = ix.offsets[lf.a + 1];
        uint64_t a, e;
        if (lf.row != kNoRow) {
          const uint32_t* r = ix.tile_off + static_cast<uint64_t>(lf.row) * (ix.n_tiles + 1);
          a = l0 + r[tile];
          e = l0 + r[tile + 1];
        } else {
          a = lower_bound_u32(ix.docids, l0, l1, tile_first);
          e = lower_bound_u32(ix.docids, a, l1, tile_first + kTileDocs);
        }
        if (MGX_ABLATE(bt, 4u)) e = a;  // (timing ablation: no scatter)
        if (lane == 0) seg[j] = a;
#pragma unroll
        for (int k = 0; k < 4; ++k) b[lane * 4 + k] = 0;
        wave_lds_sync();
        wave_scatter_segment_x4(ix.docids, a, e, static_cast<uint32_t>(tile_first), reinterpret_cast<uint32_t*>(b));
        wave_lds_sync();
        if ((need_rank >> j) & 1u) {  // per-word prefix popcounts: rank of a member = prefix[word] + bits below it
          uint32_t c[4], mine = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            c[k] = static_cast<uint32_t>(__popcll(b[lane * 4 + k]));
            mine += c[k];
          }
          uint32_t run = wave_incl_scan(mine) - mine;
          uint16_t* const pr = pref + static_cast<size_t>(j) * kWordsPerTile;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pr[lane * 4 + k] = static_cast<uint16_t>(run);
            run += c[k];
          }
        }
      } else {  // bitmap-form gram or filter: the tile's 2 KiB row
        const uint64_t* rowp = lf.kind == kLeafGramBitmap
                                   ? ix.gram_bitmaps + tile * ix.gb_tile_stride + lf.b * ix.gb_row_stride + lane * 4
                                   : ix.filter_bitmaps + tile * ix.fb_tile_stride + lf.b * ix.fb_row_stride + lane * 4;
        const uint4 v0 = *reinterpret_cast<const uint4*>(rowp);
        const uint4 v1 = *reinterpret_cast<const uint4*>(rowp + 2);
        *reinterpret_cast<uint4*>(b + lane * 4) = v0;
        *reinterpret_cast<uint4*>(b + lane * 4 + 2) = v1;
      }
    }
    wave_lds_sync();
    // ---- the driver's candidates, kMergeU per lane and step; survivors wait in the queue until 64 can be scored at once --
    uint32_t n_wait = 0;  // wave-uniform: queued survivors (slot in tile << 16 | driver posting offset in the tile segment)
    if (MGX_ABLATE(bt, 2u)) db = da;  // (timing ablation: staging only)
    for (uint64_t p0 = da; p0 < db || n_wait != 0; p0 += 64 * kMergeU) {
      if (p0 < db) {
        bool alive[kMergeU];
        uint32_t sl[kMergeU];
#pragma unroll
        for (int u = 0; u < kMergeU; ++u) {  // (the loads of a step are issued together)
          const uint64_t p = p0 + u * 64 + lane;
          alive[u] = p < db;
          sl[u] = (alive[u] ? ix.docids[p] : static_cast<uint32_t>(tile_first)) - static_cast<uint32_t>(tile_first);
        }
        for (uint32_t pc = 0; pc < q.n_instr; ++pc) {
          const uint32_t ins = prog[pc];
          const uint32_t op = ins >> 24, arg = ins & 0xFFFFFFu;
          if (op == kOpCount) {  // (the only place the alive lanes are counted: a ballot per candidate row)
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < kMergeU; ++u) c += static_cast<uint32_t>(__popcll(__ballot(alive[u])));
            cnt[0] += (arg & 1u) ? c : 0;
            cnt[1] += (arg & 2u) ? c : 0;
            cnt[2] += (arg & 4u) ? c : 0;
            cnt[3] += (arg & 8u) ? c : 0;
            if (c == 0) break;  // nothing left: the rest of the program counts zeros
            continue;
          }
          if (arg == driver_leaf) continue;
          const uint64_t* const ob = bm + static_cast<size_t>(slot_of[arg]) * kWordsPerTile;
#pragma unroll
          for (int u = 0; u < kMergeU; ++u) {
            const bool in = (ob[sl[u] >> 6] >> (sl[u] & 63)) & 1ull;
            alive[u] = op == kOpAndNot ? (alive[u] && !in) : (alive[u] && in);
          }
        }
#pragma unroll
        for (int u = 0; u < kMergeU; ++u) {
          const uint64_t am = __ballot(alive[u]);
          if (alive[u])
            wq[n_wait + static_cast<uint32_t>(__popcll(am & ((1ull << lane) - 1ull)))] =
                (sl[u] << 16) | static_cast<uint32_t>(p0 + u * 64 + lane - da);
          n_wait += static_cast<uint32_t>(__popcll(am));
        }
        cnt_res += 0;  // (counted below, from the queue)
        wave_lds_sync();
      }
      // ---- BM25 of the queued survivors, 64 at a time (the last, partial round when the segment is exhausted) ----------
      if (MGX_ABLATE(bt, 1u)) {  // (timing ablation: no scoring)
        cnt_res += n_wait;
        n_wait = 0;
      }
      while (n_wait >= 64 || (p0 + 64 * kMergeU >= db && n_wait != 0)) {
        const uint32_t take = n_wait < 64 ? n_wait : 64;
        const bool alive = lane < take;
        const uint32_t ent = alive ? wq[n_wait - take + lane] : 0u;  // (taken from the tail: no shifting of the rest)
        n_wait -= take;
        cnt_res += take;
        const uint32_t sl = ent >> 16;
        const uint64_t p = da + (ent & 0xFFFFu);
        const uint32_t w = sl >> 6;
        const uint64_t bit = 1ull << (sl & 63);
        const uint32_t slot = tile * kTileDocs + sl;
        const uint32_t d = static_cast<uint32_t>(tile_first) + sl;
        double score = 0.0;
        constexpr uint32_t kUn = 4;
        if (q.n_score <= kUn) {
          // every gather of the round — doc length and each term's tf byte — is issued before any of them is looked at:
          // with the loads inside a loop over the terms the round was a chain of one memory round trip per term
          uint32_t dl = alive ? static_cast<uint32_t>(ix.dl8[slot]) : 0u;
          uint32_t raw[kUn];   // the byte (or nibble byte) as loaded
          uint32_t how[kUn];   // 0 absent, 1 tf byte of a posting (driver / ranked list), 2 nibble byte
          uint64_t at[kUn];    // posting index (how == 1)
          double idf[kUn];
#pragma unroll
          for (uint32_t i = 0; i < kUn; ++i) {
            raw[i] = 0;
            how[i] = 0;
            at[i] = 0;
            idf[i] = 0.0;
            if (i < q.n_score) {  // wave-uniform
              const DevScoreTerm st = bt.score_terms[q.score_begin + i];
              const DevLeaf lf = leaf[st.leaf];
              idf[i] = st.idf;
              if (alive) {
                if (st.leaf == driver_leaf) {
                  how[i] = 1;
                  at[i] = p;
                } else if (lf.kind == kLeafList) {
                  const uint32_t j = slot_of[st.leaf];
                  const uint64_t word = bm[static_cast<size_t>(j) * kWordsPerTile + w];
                  if (word & bit) {
                    how[i] = 1;
                    at[i] = seg[j] + pref[static_cast<size_t>(j) * kWordsPerTile + w] +
                            static_cast<uint32_t>(__popcll(word & (bit - 1ull)));
                  }
                } else if (lf.kind == kLeafGramBitmap) {
                  how[i] = 2;
                  at[i] = static_cast<uint64_t>(lf.b) * ix.nib_row_stride + (slot >> 1);
                }
                if (how[i] == 1) raw[i] = ix.tf[at[i]];
                else if (how[i] == 2) raw[i] = ix.tfnib[at[i]];
              }
            }
          }
          if (dl == 255u) dl = ix.doc_len[slot];
          const double length_norm = q.one_minus_b + q.b * static_cast<double>(dl) / q.avgdl_clamped;
#pragma unroll
          for (uint32_t i = 0; i < kUn; ++i) {
            if (i < q.n_score) {
              uint32_t tfv = 0;
              if (how[i] == 1) {
                tfv = raw[i];
                if (tfv == 255u && ix.n_tf_ovf != 0) tfv = posting_tf(ix, at[i]);  // the saturated byte: the side table
              } else if (how[i] == 2) {
                tfv = (raw[i] >> ((slot & 1u) * 4u)) & 15u;
                if (tfv == 15u) {
                  const DevLeaf lf = leaf[bt.score_terms[q.score_begin + i].leaf];
                  tfv = exact_tf(ix, lf.a, lf.row, slot);
                }
              }
              if (tfv != 0) {
                const double tf = static_cast<double>(tfv);
                const double numerator = tf * q.k1_plus_1;
                const double denominator = tf + q.k1 * length_norm;
                score += idf[i] * numerator / denominator;
              }
            }
          }
        } else {
          uint32_t dl = 0;
          if (alive) {
            dl = ix.dl8[slot];
            if (dl == 255u) dl = ix.doc_len[slot];
          }
          const double length_norm = q.one_minus_b + q.b * static_cast<double>(dl) / q.avgdl_clamped;
          for (uint32_t i = 0; i < q.n_score; ++i) {
            const DevScoreTerm st = bt.score_terms[q.score_begin + i];
            const DevLeaf lf = leaf[st.leaf];
            uint32_t tfv = 0;
            if (alive) {
              if (st.leaf == driver_leaf) {
                tfv = posting_tf(ix, p);
              } else if (lf.kind == kLeafList) {
                const uint32_t j = slot_of[st.leaf];
                const uint64_t word = bm[static_cast<size_t>(j) * kWordsPerTile + w];
                if (word & bit) {
                  const uint32_t rank = pref[static_cast<size_t>(j) * kWordsPerTile + w] +
                                        static_cast<uint32_t>(__popcll(word & (bit - 1ull)));
                  tfv = posting_tf(ix, seg[j] + rank);
                }
              } else if (lf.kind == kLeafGramBitmap) {
                const uint32_t nb = ix.tfnib[static_cast<uint64_t>(lf.b) * ix.nib_row_stride + (slot >> 1)];
                tfv = (nb >> ((slot & 1u) * 4u)) & 15u;
                if (tfv == 15u) tfv = exact_tf(ix, lf.a, lf.row, slot);
              }
            }
            if (tfv != 0) {
              const double tf = static_cast<double>(tfv);
              const double numerator = tf * q.k1_plus_1;
              const double denominator = tf + q.k1 * length_norm;
              score += st.idf * numerator / denominator;
            }
          }
        }
        wave_topk_offer(tk, alive, score_key(score, desc), desc ? d : ~d);
      }
    }
    wave_lds_sync();
  }

  {
    uint32_t v[5] = {cnt[0], cnt[1], cnt[2], cnt[3], cnt_res};
    if (lane == 0) {
#pragma unroll
      for (int sidx = 0; sidx < 5; ++sidx)
        if (v[sidx]) atomicAdd(&bt.counters[static_cast<uint64_t>(qi) * 8 + sidx], (unsigned long long)v[sidx]);
    }
  }

  // ---- the waves' lists -> this item's candidate list, best first ------------------------------------------------------
  wave_topk_truncate(tk);
  if (lane == 0) misc[wave] = tk.have;
  __syncthreads();
  {
    const uint32_t cap = q.cap;
    uint32_t total = 0;
    for (int w = 0; w < kMergeWaves; ++w) total += min(misc[w], q.needed);
    const uint64_t obase = static_cast<uint64_t>(it.list) * bt.cand_stride;
    for (uint32_t e = tid; e < kMergeWaves * cap; e += kMergeBlock) {
      const uint32_t w = e / cap, i = e % cap;
      if (i >= min(misc[w], q.needed)) continue;
      const unsigned char* wb = smem + mo.wave0 + static_cast<size_t>(w) * mo.wave_bytes;
      const uint64_t k = reinterpret_cast<const uint64_t*>(wb + mo.w_keys)[i];
      const uint32_t dd0 = reinterpret_cast<const uint32_t*>(wb + mo.w_docs)[i];
      uint32_t rank = i;
      for (uint32_t w2 = 0; w2 < static_cast<uint32_t>(kMergeWaves) && rank < q.needed; ++w2) {
        if (w2 == w) continue;
        const unsigned char* wb2 = smem + mo.wave0 + static_cast<size_t>(w2) * mo.wave_bytes;
        const uint64_t* kk = reinterpret_cast<const uint64_t*>(wb2 + mo.w_keys);
        const uint32_t* dd = reinterpret_cast<const uint32_t*>(wb2 + mo.w_docs);
        uint32_t lo = 0, hi = min(misc[w2], q.needed);
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (better(kk[mid], dd[mid], k, dd0)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < q.needed) {
        bt.cand_keys[obase + rank] = k;
        bt.cand_docs[obase + rank] = dd0;
      }
    }
    if (tid == 0) bt.cand_n[it.list] = min(total, q.needed);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// merge of sorted candidate lists (per-workgroup lists of one shard, or per-shard lists of one query)
// ---------------------------------------------------------------------------------------------------------------

// lists: n_lists sorted (best-first) lists per query; list j of query slot qq starts at keys[qq*kq + j*kj] (same for
// docs) and holds cnt[qq*cq + j*cj] valid entries. Per-workgroup lists of one shard: kq = n_lists*stride, kj = stride,
// cq = n_lists, cj = 1. Per-shard lists gathered rank by rank: kq = stride, kj = elements per rank blob, cq = 1,
// cj = elements per rank blob (dj: the same pitch for the doc-id array, which may differ from the keys' when both live in
// one packed buffer per rank). Writes the merged best `needed` (best first) to top_keys/top_docs[qq*top_stride..]
// and the page [offset, offset+limit) to page_docs/page_scores[qq*page_stride ..].
__global__ __launch_bounds__(kBlock) void merge_topk_kernel(const DevQuery* __restrict__ queries, uint32_t n_lists,
                                                            const uint64_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ docs,
                                                            const uint32_t* __restrict__ cnt, uint64_t kq,
                                                            uint64_t kj, uint64_t dj, uint64_t cq, uint64_t cj,
                                                            uint64_t* __restrict__ top_keys,
                                                            uint32_t* __restrict__ top_docs,
                                                            uint32_t* __restrict__ top_n, uint32_t top_stride,
                                                            uint32_t* __restrict__ page_docs,
                                                            double* __restrict__ page_scores,
                                                            uint32_t* __restrict__ page_n, uint32_t page_stride,
                                                            const uint32_t* __restrict__ query_ids,
                                                            const uint32_t* __restrict__ list_begin,
                                                            const unsigned long long* __restrict__ counters,
                                                            uint64_t* __restrict__ totals_out) {
  const uint32_t slot = blockIdx.x;
  // (per-shard merge: this shard's match count goes next to its keys, where the exchange expects it)
  if (totals_out && threadIdx.x == 0) totals_out[slot] = counters[static_cast<uint64_t>(slot) * 8 + 4];
  const DevQuery q = queries[query_ids[slot]];
  // with list_begin (CSR over query slots) the lists of slot qq are lists list_begin[qq] .. list_begin[qq+1]-1 of
  // one launch-wide array: list L at keys[L*kj], cnt[L*cj]
  const uint64_t l0 = list_begin ? list_begin[slot] : 0;
  if (list_begin) n_lists = list_begin[slot + 1] - list_begin[slot];
#define MGX_K(j) (list_begin ? (l0 + (j)) * kj : static_cast<uint64_t>(slot) * kq + static_cast<uint64_t>(j) * kj)
#define MGX_D(j) (list_begin ? (l0 + (j)) * kj : static_cast<uint64_t>(slot) * kq + static_cast<uint64_t>(j) * dj)
#define MGX_C(j) (list_begin ? (l0 + (j)) * cj : static_cast<uint64_t>(slot) * cq + static_cast<uint64_t>(j) * cj)
  __shared__ uint32_t s_total;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  uint32_t local_total = 0;
  for (uint32_t j = threadIdx.x; j < n_lists; j += kBlock) local_total += min(cnt[MGX_C(j)], q.needed);
  if (local_total) atomicAdd(&s_total, local_total);
  __syncthreads();
  const uint32_t total = s_total;
  const uint32_t merged = min(total, q.needed);
  const uint32_t page_lo = min(q.offset, merged);
  const uint32_t page_hi = q.limit == 0 ? merged : min(q.offset + q.limit, merged);

  const uint64_t total_slots = static_cast<uint64_t>(n_lists) * q.needed;
  // Small merges (the usual page: ~12 lists x 10 entries) are staged in LDS first: ranking an entry is a chain of
  // dependent probes into every other list, which costs a memory latency each when the lists stay in HBM.
  constexpr uint32_t kStage = 1024;
  __shared__ uint64_t s_keys[kStage];
  __shared__ uint32_t s_docs[kStage];
  __shared__ uint32_t s_cnt[kStage];
  const bool staged = total_slots <= kStage;
  if (staged) {
    for (uint32_t j = threadIdx.x; j < n_lists; j += kBlock) s_cnt[j] = min(cnt[MGX_C(j)], q.needed);
    for (uint32_t e = threadIdx.x; e < total_slots; e += kBlock) {
      const uint32_t j = e / q.needed, i = e % q.needed;
      const bool live = i < cnt[MGX_C(j)];
      s_keys[e] = live ? keys[MGX_K(j) + i] : 0;
      s_docs[e] = live ? docs[MGX_D(j) + i] : 0;
    }
    __syncthreads();
  }
  // A list that is full holds `needed` entries at or above its last key, so nothing below the largest such key can
  // reach the merged top: most entries of a many-list merge are dropped here without being ranked.
  // Second bound, the one that bites when many lists are full (top-100 over ~100 workgroup lists): with F full lists,
  // their first a = ceil(needed / F) entries are `needed` entries at or above the smallest a-th key among them.
  __shared__ unsigned long long s_thr, s_thr2;
  __shared__ uint32_t s_full;
  if (threadIdx.x == 0) {
    s_thr = 0;
    s_thr2 = ~0ull;
    s_full = 0;
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < n_lists; j += kBlock) {
    const uint32_t c = staged ? s_cnt[j] : min(cnt[MGX_C(j)], q.needed);
    if (c >= q.needed && q.needed > 0) {
      const uint64_t k = staged ? s_keys[j * q.needed + q.needed - 1] : keys[MGX_K(j) + q.needed - 1];
      atomicMax(&s_thr, static_cast<unsigned long long>(k));
      atomicAdd(&s_full, 1u);
    }
  }
  __syncthreads();
  if (s_full > 1) {
    const uint32_t a = (q.needed + s_full - 1) / s_full;
    for (uint32_t j = threadIdx.x; j < n_lists; j += kBlock) {
      const uint32_t c = staged ? s_cnt[j] : min(cnt[MGX_C(j)], q.needed);
      if (c >= q.needed) {
        const uint64_t k = staged ? s_keys[j * q.needed + a - 1] : keys[MGX_K(j) + a - 1];
        atomicMin(&s_thr2, static_cast<unsigned long long>(k));
      }
    }
  }
  __syncthreads();
  const uint64_t thr = (s_full > 1 && s_thr2 > s_thr) ? static_cast<uint64_t>(s_thr2) : static_cast<uint64_t>(s_thr);
  // Everything below thr is worse than every entry at or above it, so the survivors can be ranked among themselves:
  // they are compacted and each counts the survivors that beat it (a few dozen LDS reads instead of a binary search in
  // every other list, which is what made merges of ~70 lists slow).
  constexpr uint32_t kSurv = 2048;
  __shared__ uint64_t s_sk[kSurv];
  __shared__ uint32_t s_sd[kSurv];
  __shared__ uint32_t s_nsurv;
  if (threadIdx.x == 0) s_nsurv = 0;
  __syncthreads();
  for (uint64_t e = threadIdx.x; e < total_slots; e += kBlock) {
    const uint32_t j = static_cast<uint32_t>(e / q.needed), i = static_cast<uint32_t>(e % q.needed);
    if (i >= (staged ? s_cnt[j] : cnt[MGX_C(j)])) continue;
    const uint64_t k = staged ? s_keys[e] : keys[MGX_K(j) + i];
    if (k < thr) continue;
    const uint32_t at = atomicAdd(&s_nsurv, 1u);
    if (at < kSurv) {
      s_sk[at] = k;
      s_sd[at] = staged ? s_docs[e] : docs[MGX_D(j) + i];
    }
  }
  __syncthreads();
  const uint32_t n_surv = s_nsurv;
  if (n_surv <= kSurv) {
    for (uint32_t t = threadIdx.x; t < n_surv; t += kBlock) {
      const uint64_t k = s_sk[t];
      const uint32_t d = s_sd[t];
      uint32_t rank = 0;
      for (uint32_t u = 0; u < n_surv; ++u) rank += better(s_sk[u], s_sd[u], k, d) ? 1u : 0u;
      if (rank < q.needed) {
        if (top_keys) {
          top_keys[static_cast<uint64_t>(slot) * top_stride + rank] = k;
          top_docs[static_cast<uint64_t>(slot) * top_stride + rank] = d;
        }
        if (page_docs && rank >= page_lo && rank < page_hi) {
          page_docs[static_cast<uint64_t>(slot) * page_stride + rank - page_lo] = q.descending ? d : ~d;
          page_scores[static_cast<uint64_t>(slot) * page_stride + rank - page_lo] = key_score(k, q.descending != 0);
        }
      }
    }
  } else
  for (uint64_t e = threadIdx.x; e < total_slots; e += kBlock) {
    const uint32_t j = static_cast<uint32_t>(e / q.needed), i = static_cast<uint32_t>(e % q.needed);
    if (i >= (staged ? s_cnt[j] : cnt[MGX_C(j)])) continue;
    const uint64_t k = staged ? s_keys[e] : keys[MGX_K(j) + i];
    if (k < thr) continue;
    const uint32_t d = staged ? s_docs[e] : docs[MGX_D(j) + i];
    uint32_t rank = i;
    for (uint32_t j2 = 0; j2 < n_lists && rank < q.needed; ++j2) {
      if (j2 == j) continue;
      uint32_t lo = 0;
      if (staged) {
        const uint64_t* kk = s_keys + j2 * q.needed;
        const uint32_t* dd = s_docs + j2 * q.needed;
        uint32_t hi = s_cnt[j2];
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (better(kk[mid], dd[mid], k, d)) lo = mid + 1; else hi = mid;
        }
      } else {
        const uint64_t* kk = keys + MGX_K(j2);
        const uint32_t* dd = docs + MGX_D(j2);
        uint32_t hi = min(cnt[MGX_C(j2)], q.needed);
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (better(kk[mid], dd[mid], k, d)) lo = mid + 1; else hi = mid;
        }
      }
      rank += lo;
    }
    if (rank < q.needed) {
      if (top_keys) {
        top_keys[static_cast<uint64_t>(slot) * top_stride + rank] = k;
        top_docs[static_cast<uint64_t>(slot) * top_stride + rank] = d;
      }
      if (page_docs && rank >= page_lo && rank < page_hi) {
        page_docs[static_cast<uint64_t>(slot) * page_stride + rank - page_lo] = q.descending ? d : ~d;
        page_scores[static_cast<uint64_t>(slot) * page_stride + rank - page_lo] = key_score(k, q.descending != 0);
      }
    }
  }
  if (threadIdx.x == 0) {
    if (top_n) top_n[slot] = merged;
    if (page_n) page_n[slot] = page_hi - page_lo;
  }
#undef MGX_K
#undef MGX_C
#undef MGX_D
}

// A shard's docid-ordered pages in the exchange layout of mgx_batch_export_topk: a page is already a best-first list
// under the key "doc id" (descending pages) or "complement of the doc id" (ascending pages).
__global__ void export_pages_kernel(const DevQuery* __restrict__ queries, uint32_t n, uint32_t stride,
                                    const uint32_t* __restrict__ page_docs, const uint64_t* __restrict__ totals,
                                    uint64_t* __restrict__ blob64, uint32_t* __restrict__ blob32) {
  const uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= static_cast<uint64_t>(n) * stride) return;
  const uint32_t qi = static_cast<uint32_t>(e / stride), k = static_cast<uint32_t>(e % stride);
  const DevQuery q = queries[qi];
  const uint64_t total = totals[qi];
  const uint32_t cnt = static_cast<uint32_t>(total < q.limit ? total : q.limit);
  uint32_t dprime = 0;
  if (k < cnt) {
    const uint32_t d = page_docs[e];
    dprime = q.descending ? d : ~d;
  }
  blob64[e] = dprime;
  blob32[e] = dprime;
  if (k == 0) {
    blob64[static_cast<uint64_t>(n) * stride + qi] = total;
    blob32[static_cast<uint64_t>(n) * stride + qi] = cnt;
  }
}

// out[q] = sum over shards of totals[shard*pitch + q]
__global__ void sum_totals_kernel(const uint64_t* __restrict__ totals, uint32_t n_shards, uint32_t n_queries,
                                  uint64_t pitch, uint64_t* __restrict__ out) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_queries) return;
  uint64_t s = 0;
  for (uint32_t r = 0; r < n_shards; ++r) s += totals[static_cast<uint64_t>(r) * pitch + q];
  out[q] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// result bitmaps -> docid pages
// ---------------------------------------------------------------------------------------------------------------

// Exclusive scan of tile_cnt per bitmap-mode query (one workgroup each): tile_start[slot][t], total[slot].
__global__ __launch_bounds__(kBlock) void scan_tiles_kernel(const uint32_t* __restrict__ tile_cnt, uint32_t n_tiles,
                                                            uint64_t* __restrict__ tile_start,
                                                            uint64_t* __restrict__ totals,
                                                            const uint8_t* __restrict__ skip) {
  __shared__ uint64_t s_carry;
  __shared__ uint32_t s_w[4];
  const uint32_t slot = blockIdx.x;
  if (skip != nullptr && skip[slot] != 0) return;  // a query without tile counts (cand_kernel writes its total itself)
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_tiles; base += kBlock) {
    const uint32_t t = base + threadIdx.x;
    const uint32_t v = t < n_tiles ? tile_cnt[static_cast<uint64_t>(slot) * n_tiles + t] : 0;
    const uint32_t inc = wave_incl_scan(v);
    if (lane_id() == 63) s_w[wave_id()] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave_id(); ++w) off += s_w[w];
    if (t < n_tiles) tile_start[static_cast<uint64_t>(slot) * n_tiles + t] = s_carry + off + inc - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) s_carry += off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[slot] = s_carry;
}

// One workgroup per (tile, query): writes the tile's matches that fall inside the requested page.
// Page = the first `take[slot]` ids in ascending order, or (reverse) the last `take[slot]` ids, descending.
__global__ __launch_bounds__(kBlock) void expand_kernel(const uint64_t* __restrict__ rbits,
                                                        const uint64_t* __restrict__ tile_start,
                                                        const uint64_t* __restrict__ totals,
                                                        const uint64_t* __restrict__ take,
                                                        const uint64_t* __restrict__ out_off,
                                                        const uint32_t* __restrict__ reverse, uint32_t n_tiles,
                                                        uint32_t first_doc_id, uint32_t* __restrict__ out) {
  const uint32_t tile = blockIdx.x, slot = blockIdx.y;
  const uint64_t total = totals[slot], want = take[slot];
  const uint64_t start = tile_start[static_cast<uint64_t>(slot) * n_tiles + tile];
  const uint64_t end = tile + 1 < n_tiles ? tile_start[static_cast<uint64_t>(slot) * n_tiles + tile + 1] : total;
  if (start == end) return;
  const bool rev = reverse[slot] != 0;
  // ranks wanted: ascending [0, want) ; reverse [total-want, total)
  const uint64_t want_lo = rev ? total - want : 0, want_hi = rev ? total : want;
  if (end <= want_lo || start >= want_hi) return;
  __shared__ uint32_t s_w[4];
  const uint64_t w = rbits[(static_cast<uint64_t>(slot) * n_tiles + tile) * kWordsPerTile + threadIdx.x];
  const uint32_t c = __popcll(w);
  const uint32_t inc = wave_incl_scan(c);
  if (lane_id() == 63) s_w[wave_id()] = inc;
  __syncthreads();
  uint32_t off = 0;
  for (int i = 0; i < wave_id(); ++i) off += s_w[i];
  uint64_t r = start + off + inc - c;
  uint64_t bits = w;
  while (bits) {
    const uint32_t b = __builtin_ctzll(bits);
    bits &= bits - 1;
    if (r >= want_lo && r < want_hi) {
      const uint32_t doc = first_doc_id + tile * kTileDocs + threadIdx.x * 64 + b;
      const uint64_t pos = rev ? (total - 1 - r) : r;
      out[out_off[slot] + pos] = doc;
    }
    ++r;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// small-candidate operators
// ---------------------------------------------------------------------------------------------------------------

// Index::FilterByNgrams: keep[i] = candidate i is present in every list.
__global__ void retain_kernel(DevIndex ix, const uint32_t* __restrict__ cand, uint64_t n_cand,
                              const uint32_t* __restrict__ grams, uint32_t n_grams, uint8_t* __restrict__ keep) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_cand) return;
  const uint32_t d = cand[i];
  bool ok = d >= ix.first_doc_id && static_cast<uint64_t>(d) - ix.first_doc_id < ix.n_docs;
  for (uint32_t t = 0; ok && t < n_grams; ++t) {
    const uint32_t g = grams[t];
    uint64_t lo = ix.offsets[g], hi = ix.offsets[g + 1];
    const uint32_t row = ix.skip_row[g];
    if (row != kNoRow) {
      const uint32_t tile = (d - ix.first_doc_id) >> kTileShift;
      const uint32_t* r = ix.tile_off + static_cast<uint64_t>(row) * (ix.n_tiles + 1);
      hi = lo + r[tile + 1];
      lo = lo + r[tile];
    }
    const uint64_t p = lower_bound_u32(ix.docids, lo, hi, d);
    ok = p < hi && ix.docids[p] == d;
  }
  keep[i] = ok ? 1 : 0;
}

// BM25Scorer::ScoreDocuments for single-gram terms over explicit candidates (bm25_scorer.cpp:71-91).
__global__ void score_candidates_kernel(DevIndex ix, const uint32_t* __restrict__ cand, uint64_t n_cand,
                                        const uint32_t* __restrict__ grams, const double* __restrict__ idfs,
                                        uint32_t n_terms, double k1, double b, double one_minus_b, double k1_plus_1,
                                        double avgdl_clamped, double* __restrict__ scores) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_cand) return;
  const uint32_t d = cand[i];
  double score = 0.0;
  if (d >= ix.first_doc_id && static_cast<uint64_t>(d) - ix.first_doc_id < ix.n_docs) {
    const uint32_t slot = d - ix.first_doc_id;
    const uint32_t dli = ix.doc_len[slot];
    if (dli > 0) {  // empty / missing text => 0.0
      const double dl = static_cast<double>(dli);
      const double length_norm = one_minus_b + b * dl / avgdl_clamped;
      for (uint32_t t = 0; t < n_terms; ++t) {
        const uint32_t g = grams[t];
        if (g == kNoRow) continue;  // term unknown to the index: tf = 0 everywhere
        uint64_t lo = ix.offsets[g], hi = ix.offsets[g + 1];
        const uint32_t row = ix.skip_row[g];
        if (row != kNoRow) {
          const uint32_t* r = ix.tile_off + static_cast<uint64_t>(row) * (ix.n_tiles + 1);
          hi = lo + r[(slot >> kTileShift) + 1];
          lo = lo + r[slot >> kTileShift];
        }
        const uint64_t p = lower_bound_u32(ix.docids, lo, hi, d);
        if (p < hi && ix.docids[p] == d) {
          const double tf = static_cast<double>(posting_tf(ix, p));
          const double numerator = tf * k1_plus_1;
          const double denominator = tf + k1 * length_norm;
          score += idfs[t] * numerator / denominator;
        }
      }
    }
  }
  scores[i] = score;
}

// BM25Scorer::ScoreDocuments as the reference runs it — every term's tf counted in the doc text (bm25_scorer.cpp:71-91)
// — for terms of any length. One thread per candidate.
__global__ void score_candidates_text_kernel(DevIndex ix, const uint32_t* __restrict__ cand, uint64_t n_cand,
                                             const uint8_t* __restrict__ term_bytes,
                                             const uint32_t* __restrict__ term_off, const double* __restrict__ idfs,
                                             uint32_t n_terms, double k1, double b, double one_minus_b,
                                             double k1_plus_1, double avgdl_clamped, double* __restrict__ scores) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_cand) return;
  const uint32_t d = cand[i];
  double score = 0.0;
  if (d >= ix.first_doc_id && static_cast<uint64_t>(d) - ix.first_doc_id < ix.n_docs) {
    const uint32_t slot = d - ix.first_doc_id;
    const uint64_t t0 = ix.text_off[slot], t1 = ix.text_off[slot + 1];
    if (t1 > t0) {  // empty / missing text => 0.0
      const double dl = static_cast<double>(ix.doc_len[slot]);
      const double length_norm = one_minus_b + b * dl / avgdl_clamped;
      for (uint32_t t = 0; t < n_terms; ++t) {
        const double tf = static_cast<double>(text_count_occurrences(
            ix.text, t0, t1, text_pattern(term_bytes + term_off[t], term_off[t + 1] - term_off[t]), false));
        if (tf > 0.0) {
          const double numerator = tf * k1_plus_1;
          const double denominator = tf + k1 * length_norm;
          score += idfs[t] * numerator / denominator;
        }
      }
    }
  }
  scores[i] = score;
}

// (key, docid') pairs for ResultSorter::SortByScore on arbitrary inputs, and the trivial one-list "merge" input.
__global__ void make_sort_keys_kernel(const uint32_t* __restrict__ docs, const double* __restrict__ scores,
                                      uint64_t n, int descending, uint64_t* __restrict__ keys,
                                      uint32_t* __restrict__ dprime) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double s = scores[i] + 0.0;  // -0.0 ties with +0.0 in the reference comparator
  keys[i] = score_key(s, descending != 0);
  dprime[i] = descending ? docs[i] : ~docs[i];
}

// rank-by-counting sort for SortByScore on device-resident (key, docid') pairs: rank[i] = entries that beat i.
// O(n^2 / threads); used for result sets up to a few tens of thousands (larger ones are sorted on the host side of
// the shim after scoring on the device).
__global__ void rank_count_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ dprime, uint64_t n,
                                  uint32_t lo, uint32_t hi, int descending, uint32_t* __restrict__ out) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  const uint32_t d = dprime[i];
  uint32_t rank = 0;
  for (uint64_t j = 0; j < n; ++j) rank += better(keys[j], dprime[j], k, d) ? 1u : 0u;
  if (rank >= lo && rank < hi) out[rank - lo] = descending ? d : ~d;
}

// ---------------------------------------------------------------------------------------------------------------
// full sort of (key, docid') pairs, best first: ResultSorter::SortByScore without a bound on offset + limit
// ---------------------------------------------------------------------------------------------------------------
//
// The reference sorts (partial_sort / sort) whatever it is given (result_sorter.cpp:661-716) and is benchmarked with
// OFFSET 10000 (docs/releases/v1.3.5.md:238); the fused top-k stops at offset+limit = 1024. Past that, and for
// unbounded pages, the pairs are sorted whole by a bitonic network: runs of kSortRun elements in LDS (every stage
// with j < kSortRun), the wider stages one compare-exchange per thread in HBM. O(n log^2 n) traffic — a fallback, not
// a hot path (a 1M-entry sort moves ~5 GB). The comparator is better(): key, then docid'; padding entries (0,0) lose
// to every real entry and end up at the tail.
constexpr uint32_t kSortRun = 2048;  // elements sorted / merged per workgroup in LDS (24 KB)

__device__ __forceinline__ void sort_cmpxchg(uint64_t* k, uint32_t* d, uint32_t i, uint32_t l, bool best_first) {
  const uint64_t ki = k[i], kl = k[l];
  const uint32_t di = d[i], dl = d[l];
  const bool swap = best_first ? better(kl, dl, ki, di) : better(ki, di, kl, dl);
  if (swap) {
    k[i] = kl; d[i] = dl;
    k[l] = ki; d[l] = di;
  }
}

// Stages (k = kfirst..klast doubling, j = min(k/2, kSortRun/2)..1) of the network on this block's run, in LDS.
// kfirst == 2: sorts the run from scratch; kfirst == klast >= kSortRun * 2... : finishes a wide stage's small strides.
__global__ __launch_bounds__(kBlock) void sort_local_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ docs,
                                                            uint64_t kfirst, uint64_t klast) {
  __shared__ uint64_t sk[kSortRun];
  __shared__ uint32_t sd[kSortRun];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kSortRun;
  for (uint32_t e = threadIdx.x; e < kSortRun; e += kBlock) {
    sk[e] = keys[base + e];
    sd[e] = docs[base + e];
  }
  __syncthreads();
  for (uint64_t k = kfirst; k <= klast; k <<= 1) {
    for (uint32_t j = static_cast<uint32_t>(k / 2 < kSortRun / 2 ? k / 2 : kSortRun / 2); j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < kSortRun / 2; t += kBlock) {
        const uint32_t i = 2 * t - (t & (j - 1));  // the lower index of pair t at stride j
        const bool best_first = ((base + i) & k) == 0;
        sort_cmpxchg(sk, sd, i, i + j, best_first);
      }
      __syncthreads();
    }
  }
  for (uint32_t e = threadIdx.x; e < kSortRun; e += kBlock) {
    keys[base + e] = sk[e];
    docs[base + e] = sd[e];
  }
}

// One compare-exchange per thread at stride j >= kSortRun of stage k, in HBM.
__global__ void sort_global_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ docs, uint64_t n_pairs,
                                   uint64_t k, uint64_t j) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_pairs) return;
  const uint64_t i = 2 * t - (t & (j - 1));
  const uint64_t l = i + j;
  const uint64_t ki = keys[i], kl = keys[l];
  const uint32_t di = docs[i], dl = docs[l];
  const bool best_first = (i & k) == 0;
  const bool swap = best_first ? better(kl, dl, ki, di) : better(ki, di, kl, dl);
  if (swap) {
    keys[i] = kl; docs[i] = dl;
    keys[l] = ki; docs[l] = di;
  }
}

// page[i] = docid of rank lo + i, scores likewise (scores may be null)
__global__ void sort_page_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ dprime, uint32_t lo,
                                 uint32_t hi, int descending, uint32_t* __restrict__ out_docs,
                                 double* __restrict__ out_scores) {
  const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hi) return;
  out_docs[i - lo] = descending ? dprime[i] : ~dprime[i];
  if (out_scores) out_scores[i - lo] = key_score(keys[i], descending != 0);
}

// SortByScore over an arbitrarily long (key, docid') array when only a bounded page is wanted: every wave keeps the
// best `needed` of a strided share of the array (WaveTopK, as in the scoring kernels) and leaves them as one sorted
// candidate list; merge_topk_kernel then merges the lists like the per-workgroup lists of a query.
__global__ __launch_bounds__(kBlock) void topk_scan_kernel(const uint64_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ dprime, uint64_t n,
                                                           uint32_t needed, uint32_t cap, int descending,
                                                           uint64_t* __restrict__ cand_keys,
                                                           uint32_t* __restrict__ cand_docs,
                                                           uint32_t* __restrict__ cand_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = lane_id(), wave = wave_id();
  const uint32_t gw = blockIdx.x * (kBlock / 64) + wave, n_waves = gridDim.x * (kBlock / 64);
  WaveTopK tk;
  tk.cap = cap;
  tk.needed = needed;
  tk.lds_sort = false;
  tk.keys = reinterpret_cast<uint64_t*>(smem) + static_cast<size_t>(wave) * 2 * cap;
  tk.docs = reinterpret_cast<uint32_t*>(smem + static_cast<size_t>(kBlock / 64) * 2 * cap * 8) +
            static_cast<size_t>(wave) * 2 * cap;
  tk.have = 0;
  tk.pend = 0;
  tk.bound_key = 0;
  tk.bound_doc = 0;
  tk.gbound_ptr = nullptr;
  tk.gbound = 0;
  for (uint32_t i = lane; i < 2 * cap; i += 64) {
    tk.keys[i] = 0;
    tk.docs[i] = 0;
  }
  wave_lds_sync();
  for (uint64_t base = static_cast<uint64_t>(gw) * 64; base < n; base += static_cast<uint64_t>(n_waves) * 64) {
    const uint64_t i = base + lane;
    const bool valid = i < n;
    wave_topk_offer(tk, valid, valid ? keys[i] : 0ull, valid ? dprime[i] : 0u);
  }
  wave_topk_truncate(tk);
  const uint32_t have = tk.have < needed ? tk.have : needed;
  for (uint32_t i = lane; i < have; i += 64) {
    cand_keys[static_cast<uint64_t>(gw) * needed + i] = tk.keys[i];
    cand_docs[static_cast<uint64_t>(gw) * needed + i] = tk.docs[i];
  }
  if (lane == 0) cand_n[gw] = have;
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------

#define MGX_KCHECK()                                   \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) return static_cast<int>(e_); \
  } while (0)

int LaunchBuildTileOff(const uint64_t* offsets, const uint32_t* docids, const uint32_t* rows_gram, uint32_t n_rows,
                       uint32_t n_tiles, uint32_t first_doc_id, uint32_t* tile_off, hipStream_t s) {
  const uint64_t total = static_cast<uint64_t>(n_rows) * (n_tiles + 1);
  if (total == 0) return 0;
  const uint32_t blocks = static_cast<uint32_t>((total + 255) / 256);
  hipLaunchKernelGGL(build_tile_off_kernel, dim3(blocks), dim3(256), 0, s, offsets, docids, rows_gram, n_rows,
                     n_tiles, first_doc_id, tile_off);
  MGX_KCHECK();
  return 0;
}

int LaunchBuildBlockMax(const uint8_t* nib, uint64_t nib_row_stride, const uint8_t* dl8, const double* ktab,
                        double k1_plus_1, double inv_step, const uint32_t* rows, uint32_t n_rows, uint32_t n_tiles,
                        bool fine, void* out, hipStream_t s) {
  const uint64_t n_out = static_cast<uint64_t>(n_tiles) * n_rows * 256u;
  if (n_out == 0) return 0;
  const dim3 grid(static_cast<uint32_t>((n_out + 255) / 256));
  if (fine)
    hipLaunchKernelGGL(build_blockmax_kernel<true>, grid, dim3(256), 0, s, nib, nib_row_stride, dl8, ktab, k1_plus_1,
                       inv_step, rows, n_rows, n_out, out);
  else
    hipLaunchKernelGGL(build_blockmax_kernel<false>, grid, dim3(256), 0, s, nib, nib_row_stride, dl8, ktab, k1_plus_1,
                       inv_step, rows, n_rows, n_out, out);
  MGX_KCHECK();
  return 0;
}

int LaunchBuildTfNib(const uint32_t* docids, const uint8_t* tf, const uint64_t* row_lo, const uint64_t* row_hi,
                     uint32_t n_rows, uint32_t first_doc_id, uint64_t row_stride_bytes, uint8_t* nib, hipStream_t s) {
  for (uint32_t r0 = 0; r0 < n_rows; r0 += 32768) {
    const uint32_t nr = n_rows - r0 < 32768 ? n_rows - r0 : 32768;
    hipLaunchKernelGGL(build_tfnib_kernel, dim3(64, nr), dim3(256), 0, s, docids, tf, row_lo + r0, row_hi + r0,
                       first_doc_id, row_stride_bytes, nib + static_cast<uint64_t>(r0) * row_stride_bytes);
    MGX_KCHECK();
  }
  return 0;
}

int LaunchBuildBitmaps(const uint32_t* docids, const uint64_t* row_lo, const uint64_t* row_hi, uint32_t n_rows,
                       uint32_t first_doc_id, uint32_t first_row, uint64_t tile_stride, uint64_t row_stride,
                       uint64_t* bitmaps, hipStream_t s) {
  if (n_rows == 0) return 0;
  for (uint32_t r0 = 0; r0 < n_rows; r0 += 32768) {
    const uint32_t nr = n_rows - r0 < 32768 ? n_rows - r0 : 32768;
    hipLaunchKernelGGL(build_bitmap_kernel, dim3(64, nr), dim3(256), 0, s, docids, row_lo + r0, row_hi + r0,
                       first_doc_id, first_row + r0, tile_stride, row_stride,
                       reinterpret_cast<unsigned long long*>(bitmaps));
    MGX_KCHECK();
  }
  return 0;
}

// Items [0, n_plain) belong to queries of bitmap-form operands only (plain instantiation), the rest to queries with an
// operand that can be absent from a tile (jumping instantiation); n_plain >= n_items: everything plain.
int LaunchTileEval(int mode, const DevIndex& ix, const DevBatch& bt, const LdsPlan& plan, hipStream_t s, uint32_t n_plain) {
  const uint64_t grid = bt.n_items;
  if (grid == 0) return 0;
  if (grid > 0x7FFFFFFFull) return static_cast<int>(hipErrorInvalidValue);
  auto* kernel = mode == kModeScore      ? &tile_eval_kernel<kModeScore>
                 : mode == kModeBitmap   ? &tile_eval_kernel<kModeBitmap>
                 : mode == kModeDocCount ? &tile_eval_kernel<kModeDocCount>
                 : mode == kModeDocPage  ? &tile_eval_kernel<kModeDocPage>
                                         : &tile_eval_kernel<kModeTextDf>;
  if (plan.bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan.bytes));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  // (the page pass has one item per query: eight workgroups share each query's tiles)
  const uint32_t parts = mode == kModeDocPage ? 8u : 1u;
  hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(grid), parts), dim3(kBlock), plan.bytes, s, ix, bt, plan, n_plain);
  MGX_KCHECK();
  return 0;
}

uint32_t WaveCountLdsBytes(const WavePlan& plan) {
  return align8(plan.max_leaves * static_cast<uint32_t>(sizeof(DevLeaf))) + align8(plan.max_instr * 4) +
         (plan.has_list ? (kBlock / 64) * kWordsPerTile * 8 : 0);
}

int LaunchWaveCount(const DevIndex& ix, const DevBatch& bt, const WavePlan& plan, bool text_df, hipStream_t s) {
  const uint64_t grid = bt.n_items;
  if (grid == 0) return 0;
  if (grid > 0x7FFFFFFFull) return static_cast<int>(hipErrorInvalidValue);
  const uint32_t lds = WaveCountLdsBytes(plan) + (text_df ? (kBlock / 64) * kDfMatchBuf * 2 : 0);
  auto* kernel = text_df ? (plan.has_list ? &wave_count_kernel<true, true> : &wave_count_kernel<false, true>)
                         : (plan.has_list ? &wave_count_kernel<true, false> : &wave_count_kernel<false, false>);
  hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(grid)), dim3(kBlock), lds, s, ix, bt, plan);
  MGX_KCHECK();
  return 0;
}

int LaunchWavePage(const DevIndex& ix, const DevBatch& bt, const WavePlan& plan, hipStream_t s) {
  const uint64_t grid = bt.n_items;
  if (grid == 0) return 0;
  if (grid > 0x7FFFFFFFull) return static_cast<int>(hipErrorInvalidValue);
  const uint32_t lds = WaveCountLdsBytes(plan);
  if (plan.has_list)
    hipLaunchKernelGGL(wave_page_kernel<true>, dim3(static_cast<uint32_t>(grid)), dim3(kBlock), lds, s, ix, bt, plan);
  else
    hipLaunchKernelGGL(wave_page_kernel<false>, dim3(static_cast<uint32_t>(grid)), dim3(kBlock), lds, s, ix, bt, plan);
  MGX_KCHECK();
  return 0;
}

int LaunchWaveScore(const DevIndex& ix, const DevBatch& bt, const WavePlan& plan, hipStream_t s) {
  const uint64_t grid = bt.n_items;
  if (grid == 0) return 0;
  if (grid > 0x7FFFFFFFull) return static_cast<int>(hipErrorInvalidValue);
  auto* kernel = plan.has_list ? &wave_score_lists_kernel : &wave_score_kernel;
  if (plan.bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan.bytes));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(grid)), dim3(kWaveBlock), plan.bytes, s, ix, bt, plan);
  MGX_KCHECK();
  return 0;
}

int LaunchCand(int mode, const DevIndex& ix, const DevBatch& bt, uint32_t max_leaves, uint32_t max_instr, uint32_t max_cap,
               hipStream_t s) {
  if (bt.n_items == 0) return 0;
  if (max_leaves == 0) max_leaves = 1;
  if (max_instr == 0) max_instr = 1;
  const uint32_t lds = carve_cand(max_leaves, max_instr, mode == kModeScore ? max_cap : 0).total;
  auto* kernel = mode == kModeScore ? &cand_kernel<kModeScore> : &cand_kernel<kModeDocPage>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(kernel, dim3(bt.n_items), dim3(kCandBlock), lds, s, ix, bt, max_leaves, max_instr,
                     mode == kModeScore ? max_cap : 0u);
  MGX_KCHECK();
  return 0;
}

int LaunchMergeScore(const DevIndex& ix, const DevBatch& bt, uint32_t max_leaves, uint32_t max_instr, uint32_t max_ops,
                     uint32_t max_cap, hipStream_t s) {
  if (bt.n_items == 0) return 0;
  if (max_leaves == 0) max_leaves = 1;
  if (max_instr == 0) max_instr = 1;
  if (max_ops == 0) max_ops = 1;
  const uint32_t lds = carve_merge(max_leaves, max_instr, max_ops, max_cap).total;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&merge_score_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(merge_score_kernel, dim3(bt.n_items), dim3(kMergeBlock), lds, s, ix, bt, max_leaves, max_instr, max_ops,
                     max_cap);
  MGX_KCHECK();
  return 0;
}

template <int T>
static int LaunchBitmapScoreT(const DevIndex& ix, const DevBatch& bt, const FastPlan& plan, hipStream_t s) {
  if (plan.bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bitmap_score_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan.bytes));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL((bitmap_score_kernel<T>), dim3(bt.n_items), dim3(kFastBlock), plan.bytes, s, ix, bt, plan);
  MGX_KCHECK();
  return 0;
}

// bt.items: the fast-path items of the queries with `n_score` scored terms
int LaunchBitmapScore(uint32_t n_score, const DevIndex& ix, const DevBatch& bt, const FastPlan& plan, hipStream_t s) {
  if (bt.n_items == 0) return 0;
  switch (n_score) {
    case 1: return LaunchBitmapScoreT<1>(ix, bt, plan, s);
    case 2: return LaunchBitmapScoreT<2>(ix, bt, plan, s);
    case 3: return LaunchBitmapScoreT<3>(ix, bt, plan, s);
    case 4: return LaunchBitmapScoreT<4>(ix, bt, plan, s);
    case 5: return LaunchBitmapScoreT<5>(ix, bt, plan, s);
    default: return static_cast<int>(hipErrorInvalidValue);
  }
}

// ---- seed-bound exchange of a sharded table (mgx_batch_execute_sharded) -----------------------------------------------
// out[q][k]: the best min(k, cand_n) keys of query q's seed list (its first candidate list, best first), 0 beyond — and
// all 0 for a query without a seed.
__global__ __launch_bounds__(256) void pack_seed_keys_kernel(const uint32_t* __restrict__ list_begin,
                                                             const uint8_t* __restrict__ has_seed,
                                                             const uint64_t* __restrict__ cand_keys,
                                                             const uint32_t* __restrict__ cand_n, uint32_t cand_stride,
                                                             uint32_t n, uint32_t k, uint64_t* __restrict__ out) {
  const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<uint64_t>(n) * k) return;
  const uint32_t q = static_cast<uint32_t>(idx / k), j = static_cast<uint32_t>(idx % k);
  uint64_t v = 0;
  if (has_seed[q]) {
    const uint32_t list = list_begin[q];
    if (j < cand_n[list] && j < cand_stride) v = cand_keys[static_cast<uint64_t>(list) * cand_stride + j];
  }
  out[idx] = v;
}

// One wave per query. all[r][q][k]: rank r's seed keys of query q, best first, zeros at the end. The query's bound is
// raised to the needed-th best key of the union: at least `needed` docs of the TABLE score that or better, so a doc with
// a strictly smaller key is on no rank's share of the page (equal keys are kept everywhere: the docid decides in the merge).
__global__ __launch_bounds__(64) void apply_seed_bounds_kernel(const uint64_t* __restrict__ all, uint32_t world, uint32_t n,
                                                               uint32_t k, const DevQuery* __restrict__ queries,
                                                               unsigned long long* __restrict__ bounds) {
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  if (q >= n) return;
  const uint32_t needed = queries[q].needed;
  if (needed == 0 || needed > k * world) return;
  uint64_t best = 0;
  for (uint32_t c = lane; c < world * k; c += 64) {
    const uint64_t x = all[(static_cast<uint64_t>(c / k) * n + q) * k + c % k];
    if (x == 0) continue;
    uint32_t count = 0;  // keys of the union that are >= x
    for (uint32_t r = 0; r < world; ++r) {
      const uint64_t* lst = all + (static_cast<uint64_t>(r) * n + q) * k;  // descending, zeros last
      uint32_t lo = 0, hi = k;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lst[mid] >= x) lo = mid + 1; else hi = mid;
      }
      count += lo;
    }
    if (count >= needed && x > best) best = x;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint64_t o = __shfl_down(best, d, 64);
    best = o > best ? o : best;
  }
  if (lane == 0 && best != 0) atomicMax(&bounds[q], static_cast<unsigned long long>(best));
}

int LaunchPackSeedKeys(const uint32_t* list_begin, const uint8_t* has_seed, const uint64_t* cand_keys, const uint32_t* cand_n,
                       uint32_t cand_stride, uint32_t n, uint32_t k, uint64_t* out, hipStream_t s) {
  const uint64_t total = static_cast<uint64_t>(n) * k;
  if (total == 0) return 0;
  hipLaunchKernelGGL(pack_seed_keys_kernel, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, s, list_begin,
                     has_seed, cand_keys, cand_n, cand_stride, n, k, out);
  MGX_KCHECK();
  return 0;
}

int LaunchApplySeedBounds(const uint64_t* all, uint32_t world, uint32_t n, uint32_t k, const DevQuery* queries,
                          unsigned long long* bounds, hipStream_t s) {
  if (n == 0 || k == 0) return 0;
  hipLaunchKernelGGL(apply_seed_bounds_kernel, dim3(n), dim3(64), 0, s, all, world, n, k, queries, bounds);
  MGX_KCHECK();
  return 0;
}

// Contribution tables of the wave kernel's pool, one workgroup per table (see TableJob). Built here instead of on the
// host: a fresh serving process meets new (gram, idf) pairs in almost every batch until its working set exists, and
// 3840 fp64 divisions + a 30 KB copy per table sat in the compile step of those batches.
__global__ __launch_bounds__(256) void build_contrib_tables_kernel(const TableJob* __restrict__ jobs, uint32_t table_dl,
                                                                   double* __restrict__ pool) {
  const TableJob j = jobs[blockIdx.x];
  const uint32_t nd = (kFastPoolTf + 1) * table_dl;
  double* t = pool + static_cast<uint64_t>(j.slot) * nd;
  const double one_minus_b = 1.0 - j.b, k1_plus_1 = j.k1 + 1.0, avg = j.avgdl > 1.0 ? j.avgdl : 1.0;  // std::max(avgdl, 1.0)
  for (uint32_t e = threadIdx.x; e < nd; e += 256) {
    const uint32_t tfi = e / table_dl, dli = e - tfi * table_dl;
    double v = 0.0;  // row 0 (a term the doc lacks) is +0.0
    if (tfi != 0) {
      const double dl = static_cast<double>(dli), tf = static_cast<double>(tfi);
      const double length_norm = one_minus_b + j.b * dl / avg;
      const double numerator = tf * k1_plus_1;
      const double denominator = tf + j.k1 * length_norm;
      v = j.idf * numerator / denominator;
    }
    t[e] = v;
  }
}

int LaunchBuildContribTables(const TableJob* jobs, uint32_t n_jobs, uint32_t table_dl, double* pool, hipStream_t s) {
  if (n_jobs == 0 || table_dl == 0) return 0;
  hipLaunchKernelGGL(build_contrib_tables_kernel, dim3(n_jobs), dim3(256), 0, s, jobs, table_dl, pool);
  MGX_KCHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// typed filter columns: FilterCondition -> doc bitmap, FACET value counts
// ---------------------------------------------------------------------------------------------------------------
// A filter column (DocumentStore filter values, src/storage/document_store.h:73-87) lives by doc slot, every value widened
// to 8 bytes: signed integers / bool / TimeValue as int64, unsigned integers and string ranks as uint64, FLOAT/DOUBLE as
// their bits; a byte per doc says NULL. One pass turns a condition (op, literal) into a filter bitmap row:
//   * the per-document comparison of ApplyFilters (search_pipeline.cpp:1136-1187): CompareValues' six operators on the
//     widened value, CompareDoubleValues' epsilon for = / != on doubles (eq_epsilon > 0), NULL a member iff null_matches
//     (the fallback lets NULL pass != only, :1151-1157); never_matches: the literal did not parse for this column's type
//     (":invalid number -> false" for every non-NULL doc);
//   * the EQ bitmap of the FilterIndex path (BuildTypeUnionBitmap, :1021-1094): op EQ, eq_epsilon 0 (doubles by their
//     bits, as the serialized keys compare), null_matches 0.
// One thread per doc, one ballot per 64 docs: coalesced 8-byte loads, one 8-byte store per wave.
__global__ __launch_bounds__(256) void filter_compare_kernel(const uint64_t* __restrict__ values,
                                                             const uint8_t* __restrict__ is_null, uint32_t n_docs,
                                                             uint32_t value_class, uint32_t op, uint64_t literal,
                                                             double eq_epsilon, uint32_t null_matches,
                                                             uint32_t never_matches, uint64_t* __restrict__ dst) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  bool m = false;
  if (slot < n_docs) {
    if (is_null != nullptr && is_null[slot] != 0) {
      m = null_matches != 0;
    } else if (never_matches == 0) {
      const uint64_t v = values[slot];
      bool lt, gt, le, ge, eq, ne;
      if (value_class == 0) {  // signed
        const long long a = static_cast<long long>(v), b = static_cast<long long>(literal);
        lt = a < b; gt = a > b; le = a <= b; ge = a >= b; eq = a == b; ne = a != b;
      } else if (value_class == 1) {  // unsigned / string rank
        lt = v < literal; gt = v > literal; le = v <= literal; ge = v >= literal; eq = v == literal; ne = v != literal;
      } else {  // double: CompareDoubleValues (comparison_utils.h:56-70), or the exact key comparison of the bitmap path
        const double a = __longlong_as_double(static_cast<long long>(v)), b = __longlong_as_double(static_cast<long long>(literal));
        lt = a < b; gt = a > b; le = a <= b; ge = a >= b;
        if (eq_epsilon > 0.0) {
          eq = fabs(a - b) < eq_epsilon;
          ne = fabs(a - b) >= eq_epsilon;
        } else {
          eq = v == literal;
          ne = !eq;
        }
      }
      m = op == 0u ? eq : op == 1u ? ne : op == 2u ? lt : op == 3u ? le : op == 4u ? gt : ge;
    }
  }
  const uint64_t mask = __ballot(m);
  if ((threadIdx.x & 63) == 0 && slot < ((n_docs + 63u) & ~63u)) dst[slot >> 6] = mask;
}

int LaunchFilterCompare(const uint64_t* values, const uint8_t* is_null, uint32_t n_docs, uint32_t value_class, uint32_t op,
                        uint64_t literal, double eq_epsilon, uint32_t null_matches, uint32_t never_matches, uint64_t* dst,
                        hipStream_t s) {
  if (n_docs == 0) return 0;
  hipLaunchKernelGGL(filter_compare_kernel, dim3((n_docs + 255) / 256), dim3(256), 0, s, values, is_null, n_docs, value_class,
                     op, literal, eq_epsilon, null_matches, never_matches, dst);
  MGX_KCHECK();
  return 0;
}

// FACET (FilterIndex::GetColumnValueCountsFiltered, src/storage/filter_index.cpp:284-312): how many docs of a result
// bitmap hold each distinct value of a column. value_ids[slot] = dense id of the doc's value (0xFFFFFFFF: NULL — NULLs have
// no bitmap in the reference's index and are not counted). One thread per 64-doc word of the result; a column of up to
// kFacetLdsValues distinct values is counted in an LDS histogram per workgroup and flushed once, larger ones with global
// atomics.
constexpr uint32_t kFacetLdsValues = 8192;
__global__ __launch_bounds__(256) void facet_count_kernel(const uint64_t* __restrict__ rbits, uint32_t n_words,
                                                          const uint32_t* __restrict__ value_ids, uint32_t n_docs,
                                                          uint32_t n_values, uint32_t use_lds,
                                                          unsigned long long* __restrict__ counts) {
  extern __shared__ uint32_t hist[];
  if (use_lds) {
    for (uint32_t i = threadIdx.x; i < n_values; i += blockDim.x) hist[i] = 0;
    __syncthreads();
  }
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += gridDim.x * blockDim.x) {
    uint64_t bits = rbits[w];
    while (bits) {
      const uint32_t slot = w * 64u + static_cast<uint32_t>(__builtin_ctzll(bits));
      bits &= bits - 1;
      if (slot >= n_docs) continue;
      const uint32_t id = value_ids[slot];
      if (id >= n_values) continue;
      if (use_lds) atomicAdd(&hist[id], 1u); else atomicAdd(&counts[id], 1ull);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_values; i += blockDim.x)
      if (hist[i]) atomicAdd(&counts[i], static_cast<unsigned long long>(hist[i]));
  }
}

int LaunchFacetCount(const uint64_t* rbits, uint32_t n_words, const uint32_t* value_ids, uint32_t n_docs, uint32_t n_values,
                     unsigned long long* counts, hipStream_t s) {
  if (n_words == 0 || n_values == 0) return 0;
  const uint32_t use_lds = n_values <= kFacetLdsValues ? 1u : 0u;
  const uint32_t blocks = std::min<uint32_t>((n_words + 255) / 256, 2048);
  hipLaunchKernelGGL(facet_count_kernel, dim3(blocks), dim3(256), use_lds ? n_values * 4 : 0, s, rbits, n_words, value_ids,
                     n_docs, n_values, use_lds, counts);
  MGX_KCHECK();
  return 0;
}

// df pass of the text-level terms (CountDfImpl): slot 5 of every df query's counters is this shard's LOCAL count; a term
// whose local count the index already knows (its df query ran over the empty range) takes the known value. Both the
// local array (what the index caches) and the exchange array (what ranks all-reduce in place) receive it, so a rank's
// contribution to the reduction never depends on what its cache happened to hold.
__global__ __launch_bounds__(256) void gather_df_kernel(const unsigned long long* __restrict__ counters,
                                                        const uint64_t* __restrict__ known, uint32_t n,
                                                        uint64_t* __restrict__ local, uint64_t* __restrict__ exchange) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = known[i];
  const uint64_t v = k != ~0ull ? k : counters[static_cast<uint64_t>(i) * 8 + 5];
  local[i] = v;
  exchange[i] = v;
}

int LaunchGatherDf(const unsigned long long* counters, const uint64_t* known, uint32_t n, uint64_t* local,
                   uint64_t* exchange, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gather_df_kernel, dim3((n + 255) / 256), dim3(256), 0, s, counters, known, n, local, exchange);
  MGX_KCHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// mutable tables: live-set row updates, the delta index's doc map, df sums of two local indexes
// ---------------------------------------------------------------------------------------------------------------

// Bits of one filter row set (the first n_set slots) and cleared (the rest) in place: the live set of a table whose
// documents are removed or superseded after the index was built (Index::RemoveDocument / UpdateDocument, index.cpp:148-233).
__global__ __launch_bounds__(256) void update_bitmap_kernel(unsigned long long* __restrict__ row,
                                                            const uint32_t* __restrict__ slots, uint32_t n_set,
                                                            uint32_t n_total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  const uint32_t slot = slots[i];
  const unsigned long long bit = 1ull << (slot & 63u);
  if (i < n_set) atomicOr(&row[slot >> 6], bit); else atomicAnd(&row[slot >> 6], ~bit);
}

int LaunchUpdateBitmap(uint64_t* row, const uint32_t* slots, uint32_t n_set, uint32_t n_total, hipStream_t s) {
  if (n_total == 0) return 0;
  hipLaunchKernelGGL(update_bitmap_kernel, dim3((n_total + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<unsigned long long*>(row), slots, n_set, n_total);
  MGX_KCHECK();
  return 0;
}

// Postings of dead documents taken out of the bitmap form of their dense grams: pair i = (doc slot, bitmap row). A query
// whose operands are all bitmap-form then needs no live-row operand at all.
__global__ __launch_bounds__(256) void clear_gram_bits_kernel(unsigned long long* __restrict__ bitmaps, uint64_t tile_stride,
                                                              uint64_t row_stride, const uint32_t* __restrict__ slots,
                                                              const uint32_t* __restrict__ rows, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t slot = slots[i];
  const uint64_t tile = slot / kTileDocs, w = (slot % kTileDocs) >> 6;
  atomicAnd(&bitmaps[tile * tile_stride + static_cast<uint64_t>(rows[i]) * row_stride + w], ~(1ull << (slot & 63u)));
}

int LaunchClearGramBits(uint64_t* bitmaps, uint64_t tile_stride, uint64_t row_stride, const uint32_t* slots,
                        const uint32_t* rows, uint32_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(clear_gram_bits_kernel, dim3((n + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<unsigned long long*>(bitmaps), tile_stride, row_stride, slots, rows, n);
  MGX_KCHECK();
  return 0;
}

// The doc ids of one exchange blob (mgx_batch_export_topk layout) through the index's doc map: a delta index numbers its
// documents 1..n in ascending order of their table ids, so best-first order and ties are kept. Entries are in ordering
// form (the id for descending queries, its complement for ascending ones); page blobs carry the same value as their key.
__global__ __launch_bounds__(256) void remap_blob_docs_kernel(const DevQuery* __restrict__ queries, uint32_t n,
                                                              uint32_t stride, const uint32_t* __restrict__ map,
                                                              uint32_t first_doc_id, uint32_t n_docs,
                                                              uint64_t* __restrict__ blob64, uint32_t* __restrict__ blob32) {
  const uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= static_cast<uint64_t>(n) * stride) return;
  const uint32_t qi = static_cast<uint32_t>(e / stride), k = static_cast<uint32_t>(e % stride);
  const uint32_t cnt = blob32[static_cast<uint64_t>(n) * stride + qi];
  if (k >= cnt) return;
  const bool desc = queries[qi].descending != 0;
  const uint32_t d = blob32[e];
  const uint32_t local = (desc ? d : ~d) - first_doc_id;
  if (local >= n_docs) return;  // (cannot happen for a blob this index exported)
  const uint32_t real = map[local];
  const uint32_t o = desc ? real : ~real;
  blob32[e] = o;
  if (blob64) blob64[e] = o;
}

int LaunchRemapBlobDocs(const DevQuery* queries, uint32_t n, uint32_t stride, const uint32_t* map, uint32_t first_doc_id,
                        uint32_t n_docs, uint64_t* blob64, uint32_t* blob32, hipStream_t s) {
  const uint64_t total = static_cast<uint64_t>(n) * stride;
  if (total == 0) return 0;
  hipLaunchKernelGGL(remap_blob_docs_kernel, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, s, queries, n,
                     stride, map, first_doc_id, n_docs, blob64, blob32);
  MGX_KCHECK();
  return 0;
}

__global__ __launch_bounds__(256) void add_u64_kernel(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

int LaunchAddU64(uint64_t* dst, const uint64_t* src, uint32_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_u64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
  MGX_KCHECK();
  return 0;
}

// read-only streaming probe: the box's attainable HBM read bandwidth, the second roofline denominator of bench.py
typedef uint32_t probe_vec4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_probe_kernel(const probe_vec4* __restrict__ src, uint64_t n_vec, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_vec;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const probe_vec4 v = __builtin_nontemporal_load(src + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9E3779B9u) *sink = acc;  // never true for the probe's fill; keeps the loads alive
}

int LaunchReadProbe(const void* src, uint64_t bytes, uint32_t* sink, hipStream_t s) {
  hipLaunchKernelGGL(read_probe_kernel, dim3(256 * 16), dim3(256), 0, s, static_cast<const probe_vec4*>(src), bytes / 16, sink);
  MGX_KCHECK();
  return 0;
}

int LaunchMergeTopK(const DevQuery* queries, const uint32_t* query_ids, uint32_t n_slots, uint32_t n_lists,
                    const uint64_t* keys, const uint32_t* docs, const uint32_t* cnt, uint64_t kq, uint64_t kj,
                    uint64_t dj, uint64_t cq, uint64_t cj, uint64_t* top_keys, uint32_t* top_docs, uint32_t* top_n,
                    uint32_t top_stride, uint32_t* page_docs, double* page_scores, uint32_t* page_n,
                    uint32_t page_stride, const uint32_t* list_begin, const unsigned long long* counters,
                    uint64_t* totals_out, hipStream_t s) {
  if (n_slots == 0) return 0;
  hipLaunchKernelGGL(merge_topk_kernel, dim3(n_slots), dim3(kBlock), 0, s, queries, n_lists, keys, docs, cnt, kq, kj,
                     dj, cq, cj, top_keys, top_docs, top_n, top_stride, page_docs, page_scores, page_n, page_stride,
                     query_ids, list_begin, counters, totals_out);
  MGX_KCHECK();
  return 0;
}

int LaunchExportPages(const DevQuery* queries, uint32_t n, uint32_t stride, const uint32_t* page_docs,
                      const uint64_t* totals, uint64_t* blob64, uint32_t* blob32, hipStream_t s) {
  const uint64_t total = static_cast<uint64_t>(n) * stride;
  if (total == 0) return 0;
  hipLaunchKernelGGL(export_pages_kernel, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, s, queries, n,
                     stride, page_docs, totals, blob64, blob32);
  MGX_KCHECK();
  return 0;
}

int LaunchSumTotals(const uint64_t* totals, uint32_t n_shards, uint32_t n_queries, uint64_t pitch, uint64_t* out,
                    hipStream_t s) {
  if (n_queries == 0) return 0;
  hipLaunchKernelGGL(sum_totals_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, s, totals, n_shards, n_queries,
                     pitch, out);
  MGX_KCHECK();
  return 0;
}

int LaunchScanTiles(const uint32_t* tile_cnt, uint32_t n_slots, uint32_t n_tiles, uint64_t* tile_start,
                    uint64_t* totals, hipStream_t s, const uint8_t* skip) {
  if (n_slots == 0) return 0;
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(n_slots), dim3(kBlock), 0, s, tile_cnt, n_tiles, tile_start, totals, skip);
  MGX_KCHECK();
  return 0;
}

int LaunchExpand(const uint64_t* rbits, const uint64_t* tile_start, const uint64_t* totals, const uint64_t* take,
                 const uint64_t* out_off, const uint32_t* reverse, uint32_t n_slots, uint32_t n_tiles,
                 uint32_t first_doc_id, uint32_t* out, hipStream_t s) {
  if (n_slots == 0 || n_tiles == 0) return 0;
  for (uint32_t s0 = 0; s0 < n_slots; s0 += 32768) {
    const uint32_t ns = n_slots - s0 < 32768 ? n_slots - s0 : 32768;
    hipLaunchKernelGGL(expand_kernel, dim3(n_tiles, ns), dim3(kBlock), 0, s,
                       rbits + static_cast<uint64_t>(s0) * n_tiles * kWordsPerTile,
                       tile_start + static_cast<uint64_t>(s0) * n_tiles, totals + s0, take + s0, out_off + s0,
                       reverse + s0, n_tiles, first_doc_id, out);
    MGX_KCHECK();
  }
  return 0;
}

int LaunchRetain(const DevIndex& ix, const uint32_t* cand, uint64_t n_cand, const uint32_t* grams, uint32_t n_grams,
                 uint8_t* keep, hipStream_t s) {
  if (n_cand == 0) return 0;
  hipLaunchKernelGGL(retain_kernel, dim3(static_cast<uint32_t>((n_cand + 255) / 256)), dim3(256), 0, s, ix, cand,
                     n_cand, grams, n_grams, keep);
  MGX_KCHECK();
  return 0;
}

int LaunchScoreCandidates(const DevIndex& ix, const uint32_t* cand, uint64_t n_cand, const uint32_t* grams,
                          const double* idfs, uint32_t n_terms, double k1, double b, double avgdl, double* scores,
                          hipStream_t s) {
  if (n_cand == 0) return 0;
  const double avg = avgdl > 1.0 ? avgdl : 1.0;
  hipLaunchKernelGGL(score_candidates_kernel, dim3(static_cast<uint32_t>((n_cand + 255) / 256)), dim3(256), 0, s, ix,
                     cand, n_cand, grams, idfs, n_terms, k1, b, 1.0 - b, k1 + 1.0, avg, scores);
  MGX_KCHECK();
  return 0;
}

int LaunchScoreCandidatesText(const DevIndex& ix, const uint32_t* cand, uint64_t n_cand, const uint8_t* term_bytes,
                              const uint32_t* term_off, const double* idfs, uint32_t n_terms, double k1, double b,
                              double avgdl, double* scores, hipStream_t s) {
  if (n_cand == 0) return 0;
  const double avg = avgdl > 1.0 ? avgdl : 1.0;
  hipLaunchKernelGGL(score_candidates_text_kernel, dim3(static_cast<uint32_t>((n_cand + 255) / 256)), dim3(256), 0, s,
                     ix, cand, n_cand, term_bytes, term_off, idfs, n_terms, k1, b, 1.0 - b, k1 + 1.0, avg, scores);
  MGX_KCHECK();
  return 0;
}

// keys/dprime: the (key, docid') pairs of make_sort_keys_kernel. Returns the number of candidate lists written
// (negative: a HIP error code negated).
int LaunchTopKScan(const uint64_t* keys, const uint32_t* dprime, uint64_t n, uint32_t needed, uint32_t cap,
                   int descending, uint32_t n_blocks, uint64_t* cand_keys, uint32_t* cand_docs, uint32_t* cand_n,
                   hipStream_t s) {
  const uint32_t lds = (kBlock / 64) * 2 * cap * 12;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_scan_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(topk_scan_kernel, dim3(n_blocks), dim3(kBlock), lds, s, keys, dprime, n, needed, cap, descending,
                     cand_keys, cand_docs, cand_n);
  MGX_KCHECK();
  return 0;
}

// Sorts the first n_pow2 (a power of two >= kSortRun, padding = (0,0)) pairs best first, in place.
int LaunchSortPairs(uint64_t* keys, uint32_t* dprime, uint64_t n_pow2, hipStream_t s) {
  if (n_pow2 < kSortRun || (n_pow2 & (n_pow2 - 1)) != 0) return static_cast<int>(hipErrorInvalidValue);
  const uint32_t runs = static_cast<uint32_t>(n_pow2 / kSortRun);
  hipLaunchKernelGGL(sort_local_kernel, dim3(runs), dim3(kBlock), 0, s, keys, dprime, 2ull, static_cast<uint64_t>(kSortRun));
  MGX_KCHECK();
  for (uint64_t k = 2ull * kSortRun; k <= n_pow2; k <<= 1) {
    for (uint64_t j = k / 2; j >= kSortRun; j >>= 1) {
      const uint64_t pairs = n_pow2 / 2;
      hipLaunchKernelGGL(sort_global_kernel, dim3(static_cast<uint32_t>((pairs + 255) / 256)), dim3(256), 0, s, keys,
                         dprime, pairs, k, j);
      MGX_KCHECK();
    }
    hipLaunchKernelGGL(sort_local_kernel, dim3(runs), dim3(kBlock), 0, s, keys, dprime, k, k);
    MGX_KCHECK();
  }
  return 0;
}

int LaunchSortPage(const uint64_t* keys, const uint32_t* dprime, uint32_t lo, uint32_t hi, int descending,
                   uint32_t* out_docs, double* out_scores, hipStream_t s) {
  if (hi <= lo) return 0;
  hipLaunchKernelGGL(sort_page_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, s, keys, dprime, lo, hi, descending,
                     out_docs, out_scores);
  MGX_KCHECK();
  return 0;
}

int LaunchMakeSortKeys(const uint32_t* docs, const double* scores, uint64_t n, int descending, uint64_t* keys,
                       uint32_t* dprime, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(make_sort_keys_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, docs, scores,
                     n, descending, keys, dprime);
  MGX_KCHECK();
  return 0;
}

int LaunchSortByScore(const uint32_t* docs, const double* scores, uint64_t n, int descending, uint32_t lo,
                      uint32_t hi, uint64_t* keys_tmp, uint32_t* dprime_tmp, uint32_t* out, hipStream_t s) {
  if (n == 0) return 0;
  const uint32_t blocks = static_cast<uint32_t>((n + 255) / 256);
  hipLaunchKernelGGL(make_sort_keys_kernel, dim3(blocks), dim3(256), 0, s, docs, scores, n, descending, keys_tmp,
                     dprime_tmp);
  MGX_KCHECK();
  hipLaunchKernelGGL(rank_count_kernel, dim3(blocks), dim3(256), 0, s, keys_tmp, dprime_tmp, n, lo, hi, descending,
                     out);
  MGX_KCHECK();
  return 0;
}

}  // namespace mgx
