#!/usr/bin/env python3
"""bench.py — queries/sec + p50 latency of batched 3-term AND + BM25 top-10 on a 10M-doc bigram index (MI355X).

One step = one batch of 1024 queries: tile kernel (set algebra + fused BM25 + per-workgroup top-k), merge kernel,
(N > 1: one RCCL all-gather of per-shard top-k + merge kernel), results copied back to the host. The query batch and
the index are resident in HBM before the timed region starts; four different batches are cycled.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1 is STRONG scaling: the same 10M-doc table is cut into N contiguous doc-range shards, one per rank.
Environment knobs (for rehearsals only; the defaults are the benchmark): MGX_BENCH_DOCS, MGX_BENCH_BATCH,
MGX_BENCH_CPU_SECONDS, MGX_BENCH_DENSE (dense_threshold; >=2 disables the bitmap form of dense lists).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X datasheet HBM3E peak (/opt/skills/guides/MI355X_MICROARCH.md)


def make_queries(mg, table, n_batches, batch, seed=42, limit=10):
    """3 distinct letter-only bigrams per query, sampled proportionally to their (global) document frequency, among the
    grams every shard knows."""
    import torch.distributed as dist
    from mygram_db_amd import dist as mdist
    keys, sizes = table.keys, table.global_sizes
    # a gram is usable if it has no space and exists on every shard
    present = {k for k in keys}
    if table.world > 1:
        gathered = [None] * table.world
        dist.all_gather_object(gathered, sorted(present))
        for ks in gathered:
            present &= set(ks)
    cand = sorted(k for k in present if b" " not in k)
    size_of = dict(zip(keys, sizes.tolist()))
    w = np.asarray([size_of[k] for k in cand], dtype=np.float64)
    p = w / w.sum()
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_batches):
        qs = []
        for _ in range(batch):
            pick = rng.choice(len(cand), size=3, replace=False, p=p)
            qs.append(mg.engine.Query([cand[i].decode() for i in pick], sort_score=True, limit=limit))
        out.append(qs)
    return out


def cpu_baseline(mg, table, corpus, queries, gpu_results, seconds):
    """The CPU oracle (oracle/mygram_oracle.c: a C restatement of the reference's per-query path, text scans and all)
    timed single-threaded on a bounded sample of the same batch, and used to check the GPU results of that sample."""
    from oracle import oracle as O
    c = table.index.columns
    oidx = O.Index.from_csr(2, 0, True, c.key_bytes, c.key_off, c.offsets, c.docids)
    ostore = O.DocumentStore.from_arrays(corpus.text_bytes, corpus.text_off)
    n, avg = table.index.total_docs, table.index.avg_doc_length
    done, t0 = 0, time.perf_counter()
    mismatches = 0
    while done < len(queries) and (done < 4 or time.perf_counter() - t0 < seconds):
        q = queries[done]
        total, docs, scores = O.search_scored(oidx, ostore, q.terms, n, avg, q.k1, q.b, q.descending, q.limit, q.offset)
        g = gpu_results[done]
        if g.total != total or g.docs.tolist() != docs.tolist() or not np.array_equal(g.scores, scores):
            mismatches += 1
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d queries of batch 0 (same 10M-doc corpus, same BM25 top-10), %.1f s, 1 thread; "
                      "per query: df by text scan of every term's candidates + Execute + ScoreDocuments + SortByScore"
                      % (done, dt),
            "parity_checked": done, "parity_mismatches": mismatches}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmygram_gpu has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("MGX_FORCE_EXCHANGE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    entry.build() if not os.path.exists(os.path.join(ROOT, "mygram-db_amd", "libmygram_gpu.so")) else None
    mg = entry.load_package()
    from mygram_db_amd import dist as mdist

    n_docs_total = int(os.environ.get("MGX_BENCH_DOCS", "10000000"))
    batch_size = int(os.environ.get("MGX_BENCH_BATCH", "1024"))
    cpu_seconds = float(os.environ.get("MGX_BENCH_CPU_SECONDS", "15"))
    dense = float(os.environ.get("MGX_BENCH_DENSE", "0"))
    n_batches = 4

    t_setup = time.perf_counter()
    before, mine = mdist.shard_range(n_docs_total, rank, world)
    corpus = mg.Corpus.synthetic(mine, seed=42, global_first=before)
    table = mdist.ShardedTable(corpus, first_doc_id=1 + before, device=local_rank, ngram_size=2, kanji_ngram_size=0,
                               dense_threshold=dense)
    cols = table.index.columns
    batches_q = make_queries(mg, table, n_batches, batch_size)
    batches = [table.prepare(qs) for qs in batches_q]
    setup_s = time.perf_counter() - t_setup

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def step(i):
        b = batches[i % n_batches]
        table.run(b)
        b.fetch_raw()

    for i in range(args.warmup):
        step(i)
    for b in batches:
        b.kernel_time_ms()  # start recording HIP events around the tile kernel from here on
    sync()
    lat = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        s0 = time.perf_counter()
        step(i)
        lat.append(time.perf_counter() - s0)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: average launch duration (HIP events on the launch stream) and algorithmic bytes per launch
    k_ms, k_n, alg = 0.0, 0, [0, 0, 0]
    for b in batches:
        ms, n = b.kernel_time_ms()
        if n:
            k_ms += ms * n
            k_n += n
    k_ms = k_ms / k_n if k_n else 0.0
    # bytes of an average batch, this rank's shard (totals come from the last fetch of every batch)
    per_batch = []
    for b in batches:
        if world > 1:  # totals in h_results are table-wide after a merge; re-run locally for this shard's own counts
            b.execute(torch.cuda.current_stream().cuda_stream)
            b.fetch_raw()
        per_batch.append(b.algorithmic_bytes())
    alg = [sum(x[i] for x in per_batch) / len(per_batch) for i in range(3)]
    alg_total = sum(alg)
    achieved = alg_total / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    # HBM bytes per launch from the committed PMC summary of this same command (profiles/pmc_traffic_current.json:
    # rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH_SIZE doubled as the gfx950 guide says)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_current.json")) as f:
            traffic = json.load(f)["traffic_bytes_per_launch_dominant_kernels"] if world == 1 else None
    except (OSError, KeyError, ValueError):
        traffic = None

    # second denominator: this box's streaming-read bandwidth, measured in the same job (read-only kernel, 2 GiB)
    measured_peak = None
    if rank == 0:
        import ctypes
        gbs = ctypes.c_double(0.0)
        if mg._capi.load().mgxt_measure_read_bandwidth(local_rank, 2 << 30, 5, ctypes.byref(gbs)) == 0:
            measured_peak = gbs.value

    results0 = None
    cpu = None
    if rank == 0 and world == 1 and cpu_seconds > 0:
        table.run(batches[0])
        results0 = batches[0].fetch()
        cpu = cpu_baseline(mg, table, corpus, batches_q[0], results0, cpu_seconds)

    if rank == 0:
        qps = batch_size * args.steps / elapsed
        sizes = np.diff(cols.offsets.astype(np.int64))
        line = {
            "metric": "queries/sec + p50 latency, 10M-doc bigram index, batch=1024 3-term AND",
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "p50_ms": 1e3 * statistics.median(lat),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 set algebra + f64 BM25",
            "data": "synthetic",
            "config": {"workload": "10M-doc synthetic ASCII corpus (seed 42), bigram index, 3-term AND + BM25 top-10, "
                                   "batch=1024 (BASELINE.json configs[1])",
                       "n_docs": n_docs_total, "batch": batch_size, "limit": 10, "k1": 1.2, "b": 0.75,
                       "parallelism": "doc-range shards x%d, top-k all-gather + merge" % world,
                       "shard_docs": mine, "shard_grams": cols.n_grams, "shard_postings": cols.n_postings,
                       "index_bytes_hbm": table.index.device_index.memory_bytes(),
                       "mean_list_len_of_queries": alg[0] / 4 / batch_size / 3,
                       "dense_threshold": dense if dense else 1.0 / 256, "setup_s": setup_s},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_gbs": (traffic / (k_ms * 1e-3) / 1e9) if traffic and k_ms > 0 else None,
                         "traffic_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and k_ms > 0 else None,
                         "peak_measured_read": measured_peak,
                         "frac_of_measured": (achieved / measured_peak) if measured_peak else None,
                         "kernel": "mgx::wave_score_kernel (set algebra + fused BM25 from doc-slot tf nibbles + per-wave top-k; "
                                   "queries with a sorted-list operand run beside it as mgx::wave_score_lists_kernel on a "
                                   "side stream inside the same timed region)",
                         "kernel_ms": k_ms, "launches_timed": k_n,
                         "algorithmic_bytes_per_launch": alg_total,
                         "algorithmic_breakdown": {"lists_4B_per_posting": alg[0], "score_R_times_T_plus_4": alg[1],
                                                   "topk_12B": alg[2]},
                         "note": "achieved = algorithmic bytes (SURVEY.md 8d: 4*sum|L| + R*(T+4) + 12*min(k,R)) of "
                                 "this rank's shard per launch / HIP-event kernel time; dense lists are read as "
                                 "bitmaps and lists are shared between concurrent queries through L2/MALL, so "
                                 "physical HBM traffic is far below the algorithmic figure (see DESIGN.md, profiles/)"},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
