/*
 * mygram_tools.h — bench/test tooling exported by libmygram_gpu.so next to the product ABI (mygram_gpu.h).
 * Not a reference interface: the reference benchmarks against a MySQL-loaded table (support/seed/benchmark.py,
 * e2e/lib/data_generator.py); with no database here the corpus is synthetic, deterministic (seed 42 by default,
 * like e2e/lib/data_generator.py:26) and identical for every consumer (device index, CPU oracle, every rank).
 */
#ifndef MYGRAM_TOOLS_H_
#define MYGRAM_TOOLS_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgxt_corpus mgxt_corpus;

/* Documents global_first .. global_first+n_docs-1 of the infinite synthetic ASCII corpus `seed`:
 * doc = 4..16 words, word = Zipf(s=1) rank over a 20,000-word vocabulary (word length 2..10, letters Zipf-weighted),
 * single spaces, already normalized (lower-case ASCII). A doc's text depends only on (seed, its global number), so
 * any shard of the corpus can be generated independently. */
int mgxt_corpus_generate(uint64_t seed, uint64_t global_first, uint64_t n_docs, int n_threads, mgxt_corpus** out);
int mgxt_corpus_view(const mgxt_corpus* c, const uint8_t** text_bytes, const uint64_t** text_off, uint64_t* n_docs);
void mgxt_corpus_destroy(mgxt_corpus* c);

/* Streaming-read bandwidth of `device` in GB/s: a read-only kernel over a `bytes`-sized buffer (larger than the
 * 256 MiB Infinity Cache), best of `iters` timed launches (HIP events). bench.py reports it next to the datasheet peak
 * as the measured roofline denominator. */
int mgxt_measure_read_bandwidth(int device, uint64_t bytes, int iters, double* gb_per_s);

/* Test hook (the counterpart of the reference's MYGRAMDB_INDEX_TEST_HOOKS allocation-failure injection,
 * src/index/posting_list.h:217-246): after `after` more successful device allocations, the next `count` allocations of
 * the library fail with an out-of-memory error. count <= 0 switches the hook off. Process-wide. */
void mgxt_fail_device_allocs(int after, int count);

#ifdef __cplusplus
}
#endif
#endif
