// mgx_internal.hpp — structures shared by the host side of libmygram_gpu.so and its HIP kernels.
//
// Data layout in HBM (one index = one doc-range shard):
//   offsets[G+1] u64, docids[P(+pad)] u32 (ascending per gram), tf[P] u8, doc_len[n_docs] u32 (by local slot),
//   tile_off[rows][n_tiles+1] u32  : for every gram with a "skip row", the list-relative index of the first posting
//                                    whose local slot is >= tile*16384 (the docid space is cut into 16384-doc tiles);
//   gram_bitmaps[rows][n_tiles*256] u64 : dense grams additionally as one bit per local slot;
//   filter_bitmaps[n][n_tiles*256] u64  : FilterIndex (column,value) doc sets.
// Every set-algebra kernel works tile by tile on 16384-bit bitmaps held in LDS (one u64 word per thread of a
// 256-thread workgroup), whatever representation the operand came from.
#pragma once

#include <cstdint>

namespace mgx {

constexpr int kBlock = 256;                       // threads per workgroup (4 wave64)
constexpr int kTileShift = 14;                    // 16384 doc slots per tile
constexpr uint32_t kTileDocs = 1u << kTileShift;  // = kBlock * 64 bits: one u64 bitmap word per thread
constexpr int kWordsPerTile = kTileDocs / 64;     // 256
constexpr int kMaxTilesPerItem = 64;              // most tiles one workgroup walks (cheap queries)
constexpr uint32_t kNoRow = 0xFFFFFFFFu;
constexpr uint32_t kMaxLeaves = 192;              // distinct operands of one query (64 terms/NOT terms/filters, query_parser.h:270-272)
constexpr uint32_t kMaxLdsLeaves = 64;            // of them resident in LDS at once: sorted-list and scored operands
constexpr uint32_t kMaxScoreTerms = 64;           // scored terms (= the 64 AND terms of query_parser.h:270)
constexpr uint32_t kMaxNeeded = 1024;             // offset+limit handled by the fused top-k
constexpr uint32_t kMergeMaxOps = 6;           // operands besides the driver that merge_score_kernel stages per tile
constexpr uint32_t kMatchBuf = 2048;              // matches enumerated per scoring round

// ---- operand kinds --------------------------------------------------------------------------------------------
enum LeafKind : uint32_t {
  kLeafList = 0,          // a = gram id (sorted u32 posting list, scattered into the LDS bitmap)
  kLeafGramBitmap = 1,    // a = gram id, b = row in gram_bitmaps (precomputed dense bitmap, copied)
  kLeafFilterBitmap = 2,  // b = row in filter_bitmaps
  kLeafRange = 3,         // a = first local slot, b = one past the last local slot
  kLeafExplicit = 4,      // a = offset, b = length of an ascending docid list in the batch's explicit pool
};

struct DevLeaf {
  uint32_t kind, a, b;
  uint32_t score_slot;  // index of the scored term whose tf column this operand carries, or kNoSlot
  uint32_t row;         // the gram's skip row (tile_off), or kNoRow; filled by the host so kernels need not chase it
  uint32_t lds;         // general kernel: the operand's tile bitmap in LDS (sorted lists are scattered there, scored
                        // operands are probed there), or kNoRow: a bitmap-form operand read straight from HBM
};
constexpr uint32_t kNoSlot = 0xFFu;
constexpr uint32_t kAbsentMark = 0xFEu;  // score_slot of the empty-range operand that stands for MGX_GRAM_ABSENT
constexpr int kWaveScoreSlots = 3;   // scored terms the wave kernel handles
constexpr int kWaveBlock = 512;      // threads per workgroup of the wave kernel (8 autonomous waves share one BM25 table)
constexpr int kWavesPerBlock = kWaveBlock / 64;
constexpr uint32_t kTableTf = 6;     // BM25 contribution tables cover tf 1..6 ...
constexpr uint32_t kTableDlMax = 256;  // ... and doc lengths below min(max_doc_len+1, 256)

// ---- tile program: an accumulator machine over 64-bit bitmap words -------------------------------------------
enum Op : uint32_t {
  kOpLoad = 0,      // acc = W(leaf)
  kOpAnd,           // acc &= W(leaf)
  kOpOr,            // acc |= W(leaf)
  kOpAndNot,        // acc &= ~W(leaf)
  kOpPush,          // stack[sp++] = acc
  kOpPopAnd,        // acc = stack[--sp] & acc
  kOpPopOr,         // acc = stack[--sp] | acc
  kOpPopAndNot,     // acc = stack[--sp] & ~acc
  kOpCount,         // counter[s] += popcount(acc) for every slot s whose bit is set in arg (funnel counters)
  kOpThreshBegin,   // bit-sliced counters = 0
  kOpThreshAdd,     // counters += W(leaf)
  kOpThreshEnd,     // acc = (counters >= arg)
  kOpVerifyText,    // acc &= docs whose text contains every pattern of the query (PostFilterByText,
                    // search_pipeline.cpp:1239-1246); general workgroup kernel only
};
inline constexpr uint32_t MakeInstr(Op op, uint32_t arg) { return (static_cast<uint32_t>(op) << 24) | (arg & 0xFFFFFFu); }

// A scored term takes its tf either from the tf column of ONE gram operand (term == that gram) or, for terms that span
// several grams, from the doc text (CountTermOccurrences, bm25_scorer.cpp:27-45); the latter's idf comes from the
// batch's df pass (DevBatch::text_idf).
constexpr uint32_t kNoLeaf = 0xFFFFFFFFu;
struct DevScoreTerm {
  uint32_t leaf;       // operand whose posting list carries the tf column, or kNoLeaf for a text-level term
  uint32_t text_term;  // text-level term: index into DevBatch::text_terms / text_idf
  double idf;          // column terms only
};

struct DevTextTerm {
  uint32_t pat_off, pat_len;  // the normalized term in DevBatch::patterns
};

enum QueryMode : uint32_t {
  kModeScore = 0,   // fused BM25 + per-workgroup top-k
  kModeBitmap = 1,  // result bitmaps + per-tile counts to HBM (expanded to docids afterwards)
  kModeTextDf = 2,  // df of one text-level term: candidates of its gram AND whose text contains it (counter slot 5)
  // docid-ordered page (no SORT _score, 0 < limit <= kMaxDocPage) without materialising result bitmaps:
  kModeDocCount = 3,  // pass 1: per-tile match counts only (scan_tiles_kernel turns them into rank offsets)
  kModeDocPage = 4,   // pass 2: re-evaluates only the tiles that hold ranks of the page and writes the page's docids
};
constexpr uint32_t kMaxDocPage = 16384;

struct DevQuery {
  uint32_t leaf_begin, n_leaves;
  uint32_t prog_begin, n_instr;
  uint32_t score_begin, n_score;
  uint32_t mode;
  uint32_t cap;       // C' : power of two >= needed, >= 32 (score mode)
  uint32_t needed;    // offset + limit (0 => unbounded: not handled by the fused path)
  uint32_t limit, offset;
  uint32_t descending;
  uint32_t stack_depth;
  uint32_t out_slot;  // row of this query in the per-mode output arrays
  uint32_t pat_off, pat_len;  // kModeTextDf: the term searched in the candidates' text
  uint32_t vt_begin, vt_count;  // kOpVerifyText: the query's patterns in DevBatch::verify_terms
  double k1, b, one_minus_b, k1_plus_1, avgdl_clamped;  // BM25 constants, pre-evaluated on the host
};

struct DevIndex {
  const uint64_t* offsets;
  const uint32_t* docids;
  const uint8_t* tf;
  const uint64_t* tf_ovf_pos;  // postings whose tf byte is saturated (255): index, ascending ...
  const uint32_t* tf_ovf_val;  // ... and true count
  uint32_t n_tf_ovf;
  const uint32_t* doc_len;
  const uint8_t* dl8;         // [n_docs] min(doc_len, 255)
  const uint8_t* tfnib;       // [bitmap rows][nib_row_stride] min(tf,15) by doc slot, two docs per byte (0 = absent)
  uint64_t nib_row_stride;    // bytes
  const uint32_t* skip_row;   // [G] row in tile_off, or kNoRow
  const uint32_t* tile_off;   // [rows][n_tiles+1]
  // bitmap word (tile, row, w) lives at base[tile*tile_stride + row*row_stride + w]. Gram bitmaps are TILE-major
  // (row_stride = 256): workgroups that walk the same doc range for different queries then read one contiguous
  // region instead of addresses that differ by a multiple of the row size, which would pile onto few HBM channels.
  const uint64_t* gram_bitmaps;
  uint64_t gb_tile_stride, gb_row_stride;
  const uint64_t* filter_bitmaps;  // row-major (rows are appended at run time), rows padded off power-of-two strides
  uint64_t fb_tile_stride, fb_row_stride;
  const uint8_t* text;        // normalized doc text by local slot (mgx_index_attach_text), or null
  const uint64_t* text_off;   // [n_docs+1]
  uint32_t first_doc_id;
  uint32_t n_docs;
  uint32_t n_tiles;
  uint32_t max_doc_len;
};

// ---- fast path of SORT _score batches: flat programs over bitmap-form operands, <= kFastMaxScore dense scored terms --
// The query arrives at the kernel fully resolved (bitmap_score_kernel): every operand is a device address + tile stride
// (no operand-kind dispatch, no leaf table), every scored term is the address of its tf-nibble row, its idf and where
// its block-max bytes are (top-k pruning).
constexpr int kFastMaxOps = 8;
constexpr int kFastMaxScore = 5;
constexpr uint32_t kFastPoolTf = 14;  // contribution-table pool of the wave kernel: rows tf 0..14
#ifndef MGX_FWAVES
#define MGX_FWAVES 8
#endif
constexpr int kFastBlock = 64 * MGX_FWAVES;  // autonomous waves per workgroup, all on one query
constexpr int kFastWaves = kFastBlock / 64;

enum FastOpKind : uint32_t { kFastOr = 0, kFastAnd = 1, kFastAndNot = 2 };  // (LOAD = OR into the empty accumulator)
struct FastOp {
  uint64_t base;         // address of the operand's words of tile 0; filter bitmaps: byte offset from DevIndex::filter_bitmaps
  uint32_t tile_stride;  // bytes from one tile's 2 KiB of this row to the next tile's
  uint32_t code;         // kind | (relative-to-filter-base flag << 4) | (funnel counter mask counted AFTER this op << 8)
};
struct FastScore {
  uint64_t nib;          // address of the term's tf-nibble row (DevIndex::tfnib + row * nib_row_stride)
  double idf;
  uint32_t gram, skip_row;  // exact tf lookup of a saturated nibble
  // block-max bytes of the term (pruning): mode 0 none (its bound is inside DevFastQuery::bm_cint), 1 one byte per
  // 64-doc word at blockmax + tile * bm_tile_stride + bm_off, 2 one byte per 16-doc quarter at blockmax_fine + ...
  uint32_t bm_mode, bm_off;
};
struct DevFastQuery {
  uint32_t n_ops, n_score, needed, cap;
  uint32_t descending, pad0;
  uint32_t pad1, pad1b;
  // pruning, integer form: a quarter's bound is bm_cint + sum_i W_i * q_i in units of 1 / bm_inv_unit (W_i = the term's
  // idf rounded UP to 8 bits of the largest idf, q_i its block-max byte), evaluated with one v_dot4_u32_u8 per quarter
  uint32_t bm_wpack, bm_w4;  // W_0..W_3 packed, W_4
  uint32_t bm_cint, pad2;
  double bm_inv_unit;        // theta * bm_inv_unit, rounded down, is the integer threshold
  double k1, b, one_minus_b, k1_plus_1, avgdl_clamped;
  const uint8_t* blockmax;       // null: score every match (SORT _score ASC, or pruning unavailable)
  const uint8_t* blockmax_fine;
  uint32_t bm_tile_stride, bmf_tile_stride;  // bytes per tile of the two arrays
  FastOp ops[kFastMaxOps];
  FastScore score[kFastMaxScore];
};
static_assert(sizeof(DevFastQuery) % 8 == 0, "DevFastQuery is read with scalar loads");

struct FastPlan {
  uint32_t ring;       // match-buffer entries per wave
  uint32_t max_cap;
  uint32_t bytes, pad;
  const double* ktab;  // [256] K[dl] = k1 * (1 - b + b * dl / avgdl) of the batch's (k1, b, avgdl)
};
// One contribution table of the wave kernel's pool, to be built on the device (build_contrib_tables_kernel):
// pool[slot][tf][dl] = idf * (tf * (k1 + 1)) / (tf + k1 * ((1 - b) + b * dl / max(avgdl, 1))) — bm25_scorer.cpp:80-84
// operation by operation in fp64 (the device code is compiled with -ffp-contract=off like the host's).
struct TableJob {
  double idf, k1, b, avgdl;
  uint32_t slot, pad;
};
FastPlan PlanFast(uint32_t max_cap, const double* ktab);
uint32_t FastTableDl(uint32_t max_doc_len);  // tdl of an index's contribution-table pool

// One workgroup's share of a query: tiles [tile_begin, tile_begin + n_tiles). The host cuts every query into items
// of about equal estimated cost (expensive queries — many matches per tile — get more, shorter items), so no
// workgroup is a long tail. `list` is the item's index in the launch-wide candidate arrays.
struct DevItem {
  uint32_t query, tile_begin, n_tiles, list;
};

struct DevBatch {
  const DevItem* items;
  uint32_t n_items;
  const DevQuery* queries;
  const DevLeaf* leaves;
  const uint32_t* prog;
  const DevScoreTerm* score_terms;
  const uint32_t* explicit_pool;
  const uint8_t* patterns;         // text-level terms of the batch, concatenated
  const DevTextTerm* text_terms;
  const double* text_idf;          // [n_text_terms] ComputeIDF(N, df) of the current execute's df pass
  const DevTextTerm* verify_terms;  // patterns of the queries that filter their result by exact text
  uint32_t n_queries;
  // outputs
  unsigned long long* counters;  // [n_queries][8]: funnel slots 0..3, slot 4 = final result count
  unsigned long long* bounds;    // [n_queries] query-wide top-k pruning bound (score mode; zeroed per execute)
  // score mode: per item candidates, best first
  uint64_t* cand_keys;   // [n_items][cand_stride]
  uint32_t* cand_docs;
  uint32_t* cand_n;      // [n_items]
  uint32_t cand_stride;
  const uint64_t* wave_tables;  // [n_queries][kWaveScoreSlots] addresses of the scored terms' pool tables (wave kernel)
  const DevFastQuery* fast_queries;  // [n_queries] resolved descriptors of the fast-path queries (others: unused rows)
  const struct DevIndex* dev_index;  // the index's descriptor in device memory (out-of-line cold paths take a pointer)
  uint32_t debug_skip;   // -DMGX_ABLATION builds only (MGX_DEBUG_SKIP: 1 = no scoring, 2 = no enumeration + scoring); 0 otherwise
  // bitmap mode
  uint64_t* rbits;       // [n_bitmap_queries][n_tiles][256]
  uint32_t* tile_cnt;    // [n_bitmap_queries][n_tiles]
  // docid-page mode
  const uint64_t* tile_start;  // [n][n_tiles] matches before each tile (scan of tile_cnt)
  const uint64_t* totals;      // [n]
  uint32_t* page_docs;         // [n][page_stride]
  uint32_t page_stride;
};

// LDS bytes the tile kernel needs for a launch whose queries have at most these shapes.
struct LdsPlan {
  uint32_t max_leaves, max_score, max_stack, max_instr, max_cap;
  uint32_t max_lds_leaves;  // operand bitmaps resident in LDS
  uint32_t bytes;
};
LdsPlan PlanLds(uint32_t max_leaves, uint32_t max_lds_leaves, uint32_t max_score, uint32_t max_stack, uint32_t max_instr,
                uint32_t max_cap, bool score_mode);

// LDS plan of the wave-autonomous scoring kernel (flat programs, <= kWaveScoreSlots scored terms).
struct WavePlan {
  uint32_t max_leaves, max_score, max_instr, max_cap;
  uint32_t table_dl;      // doc-length extent of the BM25 tables (0 => no tables)
  uint32_t has_list;      // some operand needs the per-wave scatter bitmap
  uint32_t bytes;
};
WavePlan PlanWave(uint32_t max_leaves, uint32_t max_score, uint32_t max_instr, uint32_t max_cap, uint32_t max_doc_len,
                  bool has_list);

// Order-preserving key of a BM25 score for "larger is better" comparisons (scores are >= 0, so the IEEE bit pattern
// orders like the value); ASC sorts flip both key and docid.
inline uint64_t ScoreKeyHost(double s, bool descending) {
  uint64_t u;
  __builtin_memcpy(&u, &s, 8);
  return descending ? u : ~u;
}

}  // namespace mgx
