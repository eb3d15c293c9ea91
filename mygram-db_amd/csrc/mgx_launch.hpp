// mgx_launch.hpp — host-callable launchers of the kernels in mgx_kernels.hip. Return 0 or a hipError_t value.
#pragma once
#include <hip/hip_runtime_api.h>

#include "mgx_internal.hpp"

namespace mgx {

int LaunchBuildTileOff(const uint64_t* offsets, const uint32_t* docids, const uint32_t* rows_gram, uint32_t n_rows,
                       uint32_t n_tiles, uint32_t first_doc_id, uint32_t* tile_off, hipStream_t s);
int LaunchBuildBlockMax(const uint8_t* nib, uint64_t nib_row_stride, const uint8_t* dl8, const double* ktab,
                        double k1_plus_1, double inv_step, const uint32_t* rows, uint32_t n_rows, uint32_t n_tiles,
                        bool fine, void* out, hipStream_t s);
int LaunchBuildTfNib(const uint32_t* docids, const uint8_t* tf, const uint64_t* row_lo, const uint64_t* row_hi,
                     uint32_t n_rows, uint32_t first_doc_id, uint64_t row_stride_bytes, uint8_t* nib, hipStream_t s);
int LaunchBuildBitmaps(const uint32_t* docids, const uint64_t* row_lo, const uint64_t* row_hi, uint32_t n_rows,
                       uint32_t first_doc_id, uint32_t first_row, uint64_t tile_stride, uint64_t row_stride,
                       uint64_t* bitmaps, hipStream_t s);
int LaunchTileEval(int mode, const DevIndex& ix, const DevBatch& bt, const LdsPlan& plan, hipStream_t s,
                   uint32_t n_plain = 0xFFFFFFFFu);
int LaunchWaveScore(const DevIndex& ix, const DevBatch& bt, const WavePlan& plan, hipStream_t s);
int LaunchBitmapScore(uint32_t n_score, const DevIndex& ix, const DevBatch& bt, const FastPlan& plan, hipStream_t s);
int LaunchWavePage(const DevIndex& ix, const DevBatch& bt, const WavePlan& plan, hipStream_t s);
int LaunchWaveCount(const DevIndex& ix, const DevBatch& bt, const WavePlan& plan, bool text_df, hipStream_t s);
// candidate-driven evaluation of selective flat queries (items: query, tile_begin = the driver's leaf index, list)
int LaunchCand(int mode, const DevIndex& ix, const DevBatch& bt, uint32_t max_leaves, uint32_t max_instr, uint32_t max_cap,
               hipStream_t s);
uint32_t CandLdsBytes(uint32_t max_leaves, uint32_t max_instr, uint32_t max_cap);
// tile-synchronous merge over long sorted posting arrays with rank-carried tf (SORT _score; DevQuery::pat_off = driver leaf)
int LaunchMergeScore(const DevIndex& ix, const DevBatch& bt, uint32_t max_leaves, uint32_t max_instr, uint32_t max_ops,
                     uint32_t max_cap, hipStream_t s);
uint32_t MergeLdsBytes(uint32_t max_leaves, uint32_t max_instr, uint32_t max_ops, uint32_t max_cap);
int LaunchReadProbe(const void* src, uint64_t bytes, uint32_t* sink, hipStream_t s);

int LaunchMergeTopK(const DevQuery* queries, const uint32_t* query_ids, uint32_t n_slots, uint32_t n_lists,
                    const uint64_t* keys, const uint32_t* docs, const uint32_t* cnt, uint64_t kq, uint64_t kj,
                    uint64_t dj, uint64_t cq, uint64_t cj, uint64_t* top_keys, uint32_t* top_docs, uint32_t* top_n,
                    uint32_t top_stride, uint32_t* page_docs, double* page_scores, uint32_t* page_n,
                    uint32_t page_stride, const uint32_t* list_begin, const unsigned long long* counters,
                    uint64_t* totals_out, hipStream_t s);
int LaunchExportPages(const DevQuery* queries, uint32_t n, uint32_t stride, const uint32_t* page_docs,
                      const uint64_t* totals, uint64_t* blob64, uint32_t* blob32, hipStream_t s);
int LaunchSumTotals(const uint64_t* totals, uint32_t n_shards, uint32_t n_queries, uint64_t pitch, uint64_t* out,
                    hipStream_t s);
// seed-bound exchange of a sharded table: pack every query's seed keys / raise its bound to the needed-th best of all ranks'
int LaunchPackSeedKeys(const uint32_t* list_begin, const uint8_t* has_seed, const uint64_t* cand_keys, const uint32_t* cand_n,
                       uint32_t cand_stride, uint32_t n, uint32_t k, uint64_t* out, hipStream_t s);
int LaunchApplySeedBounds(const uint64_t* all, uint32_t world, uint32_t n, uint32_t k, const DevQuery* queries,
                          unsigned long long* bounds, hipStream_t s);
int LaunchBuildContribTables(const TableJob* jobs, uint32_t n_jobs, uint32_t table_dl, double* pool, hipStream_t s);
int LaunchFilterCompare(const uint64_t* values, const uint8_t* is_null, uint32_t n_docs, uint32_t value_class, uint32_t op,
                        uint64_t literal, double eq_epsilon, uint32_t null_matches, uint32_t never_matches, uint64_t* dst,
                        hipStream_t s);
int LaunchFacetCount(const uint64_t* rbits, uint32_t n_words, const uint32_t* value_ids, uint32_t n_docs, uint32_t n_values,
                     unsigned long long* counts, hipStream_t s);
int LaunchUpdateBitmap(uint64_t* row, const uint32_t* slots, uint32_t n_set, uint32_t n_total, hipStream_t s);
int LaunchClearGramBits(uint64_t* bitmaps, uint64_t tile_stride, uint64_t row_stride, const uint32_t* slots,
                        const uint32_t* rows, uint32_t n, hipStream_t s);
int LaunchRemapBlobDocs(const DevQuery* queries, uint32_t n, uint32_t stride, const uint32_t* map, uint32_t first_doc_id,
                        uint32_t n_docs, uint64_t* blob64, uint32_t* blob32, hipStream_t s);
int LaunchAddU64(uint64_t* dst, const uint64_t* src, uint32_t n, hipStream_t s);
int LaunchGatherDf(const unsigned long long* counters, const uint64_t* known, uint32_t n, uint64_t* local,
                   uint64_t* exchange, hipStream_t s);
int LaunchScanTiles(const uint32_t* tile_cnt, uint32_t n_slots, uint32_t n_tiles, uint64_t* tile_start,
                    uint64_t* totals, hipStream_t s, const uint8_t* skip = nullptr);
int LaunchExpand(const uint64_t* rbits, const uint64_t* tile_start, const uint64_t* totals, const uint64_t* take,
                 const uint64_t* out_off, const uint32_t* reverse, uint32_t n_slots, uint32_t n_tiles,
                 uint32_t first_doc_id, uint32_t* out, hipStream_t s);
int LaunchRetain(const DevIndex& ix, const uint32_t* cand, uint64_t n_cand, const uint32_t* grams, uint32_t n_grams,
                 uint8_t* keep, hipStream_t s);
int LaunchScoreCandidates(const DevIndex& ix, const uint32_t* cand, uint64_t n_cand, const uint32_t* grams,
                          const double* idfs, uint32_t n_terms, double k1, double b, double avgdl, double* scores,
                          hipStream_t s);
int LaunchScoreCandidatesText(const DevIndex& ix, const uint32_t* cand, uint64_t n_cand, const uint8_t* term_bytes,
                              const uint32_t* term_off, const double* idfs, uint32_t n_terms, double k1, double b,
                              double avgdl, double* scores, hipStream_t s);
int LaunchTopKScan(const uint64_t* keys, const uint32_t* dprime, uint64_t n, uint32_t needed, uint32_t cap,
                   int descending, uint32_t n_blocks, uint64_t* cand_keys, uint32_t* cand_docs, uint32_t* cand_n,
                   hipStream_t s);
int LaunchSortPairs(uint64_t* keys, uint32_t* dprime, uint64_t n_pow2, hipStream_t s);
int LaunchSortPage(const uint64_t* keys, const uint32_t* dprime, uint32_t lo, uint32_t hi, int descending,
                   uint32_t* out_docs, double* out_scores, hipStream_t s);
int LaunchMakeSortKeys(const uint32_t* docs, const double* scores, uint64_t n, int descending, uint64_t* keys,
                       uint32_t* dprime, hipStream_t s);
int LaunchSortByScore(const uint32_t* docs, const double* scores, uint64_t n, int descending, uint32_t lo,
                      uint32_t hi, uint64_t* keys_tmp, uint32_t* dprime_tmp, uint32_t* out, hipStream_t s);

}  // namespace mgx
