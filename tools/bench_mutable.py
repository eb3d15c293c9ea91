#!/usr/bin/env python3
"""Serving throughput of a table that changes: the headline workload of bench.py — 10M docs, fresh
1024-query batches of 3-term AND + BM25 top-10 through search_pipeline::BatchExecutor — measured on the static table and
again after documents were updated, removed and added (live row on the main index + delta index, merged on the device).
Also timed: the mutation calls themselves and the first batch after them (ApplyMutations: live row, delta build,
statistics). One JSON line per delta size. Not used by the driver.

    python tools/bench_mutable.py [--docs 10000000] [--steps 100] [--deltas 1000,10000,100000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402


HOST_MS = np.zeros(4)  # plan, compile, enqueue, wait of the last run(), mean per batch


def run(ex, qbs, steps, depth, outs):
    """`steps` fresh batches with `depth` in flight -> seconds."""
    t0 = time.perf_counter()
    pending = []
    acc = np.zeros(4)
    for j in range(steps):
        if len(pending) == depth:
            acc += ex.wait(pending.pop(0), outs[j % depth])[4][:4]
        pending.append(ex.submit(qbs[j % len(qbs)], limit=10))
    for k, t in enumerate(pending):
        acc += ex.wait(t, outs[k % depth])[4][:4]
    HOST_MS[:] = acc / steps
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--deltas", default="1000,10000,100000")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--write-rate", type=float, default=0.0, help="steady writer: changes per second (0: skip)")
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    mg = entry.load_package()
    from mygram_db_amd import _shim_capi as S
    corpus = mg.Corpus.synthetic(args.docs, seed=42)
    index = mg.Index(corpus=corpus, first_doc_id=1, ngram_size=2, kanji_ngram_size=0)
    index.ensure_text()
    index.device_index.set_batch_order(True)
    cols = index.columns

    class _Whole:  # what bench.make_queries reads of a table
        keys = [cols.gram(g) for g in range(cols.n_grams)]
        global_sizes = np.diff(cols.offsets.astype(np.int64))
        world = 1
    term_batches = bench.make_queries(mg, _Whole, 16, args.batch)
    shim_table = S.Table(index)
    depth = args.depth
    ex = S.Executor(shim_table, depth=depth, planner_threads=6)
    qbs = [S.QueryBatch(tb) for tb in term_batches]
    ex.warm(qbs[-1], limit=10, rounds=3)
    outs = [(np.zeros(args.batch, np.uint64), np.zeros(args.batch, np.uint32), np.zeros((args.batch, 10), np.uint32),
             np.zeros((args.batch, 10), np.float64), np.zeros(5, np.float64)) for _ in range(depth)]
    run(ex, qbs, 20, depth, outs)
    dt = run(ex, qbs, args.steps, depth, outs)
    static_qps = args.batch * args.steps / dt
    print(json.dumps({"table": "static", "docs": args.docs, "queries_per_s": static_qps, "ms_per_step": 1e3 * dt / args.steps,
                      "host_ms_plan_compile_enqueue_wait": HOST_MS.round(3).tolist()}), flush=True)
    rng = np.random.default_rng(7)
    texts = {}
    live_main = np.ones(args.docs + 1, bool)
    next_id = args.docs + 1
    done = 0
    for target in [int(x) for x in args.deltas.split(",")]:
        n_new = target - done
        # of the documents that change: 60% updates, 20% removals, 20% additions (the additions and updates form the delta)
        n_upd, n_rm, n_add = int(n_new * 0.6), int(n_new * 0.2), n_new - int(n_new * 0.6) - int(n_new * 0.2)
        t0 = time.perf_counter()
        cand = rng.choice(args.docs, size=2 * (n_upd + n_rm), replace=False) + 1
        cand = [int(d) for d in cand if live_main[d]][: n_upd + n_rm]
        for d in cand[:n_upd]:
            new = corpus.text(int(rng.integers(0, args.docs)))
            shim_table.update_document(d, texts.get(d, corpus.text(d - 1)), new)
            texts[d] = new
            live_main[d] = False
        for d in cand[n_upd:]:
            shim_table.remove_document(d, texts.pop(d, None) or corpus.text(d - 1))
            live_main[d] = False
        for _ in range(n_add):
            new = corpus.text(int(rng.integers(0, args.docs)))
            shim_table.add_document(next_id, new)
            texts[next_id] = new
            next_id += 1
        calls_s = time.perf_counter() - t0
        done = target
        t0 = time.perf_counter()
        ex.wait(ex.submit(qbs[0], limit=10), outs[0])  # the first batch applies the changes
        apply_s = time.perf_counter() - t0
        run(ex, qbs, 10, depth, outs)
        dt = run(ex, qbs, args.steps, depth, outs)
        st = shim_table.mutation_stats()
        print(json.dumps({"table": "mutable", "changed_documents": target, "delta_documents": st["delta_documents"],
                          "removed_from_main": st["removed_from_main"], "mutation_calls_s": calls_s,
                          "calls_per_s": n_new / calls_s, "first_batch_after_changes_s": apply_s,
                          "queries_per_s": args.batch * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                          "vs_static": args.batch * args.steps / dt / static_qps,
                          "host_ms_plan_compile_enqueue_wait": HOST_MS.round(3).tolist()}), flush=True)
    # a steady writer: changes keep arriving (a thread, ~args.write_rate per second) while the executor serves; with a
    # staleness bound the delta is rebuilt once per bound, not once per batch
    if args.write_rate > 0:
        import threading
        for bound in (0.0, 0.25, 1.0):
            shim_table.set_mutation_staleness(bound)
            stop = threading.Event()
            made = [0]

            def writer():
                nonlocal next_id
                r = np.random.default_rng(11)
                t_next = time.perf_counter()
                while not stop.is_set():
                    new = corpus.text(int(r.integers(0, args.docs)))
                    shim_table.add_document(next_id, new)
                    texts[next_id] = new
                    next_id += 1
                    made[0] += 1
                    t_next += 1.0 / args.write_rate
                    d = t_next - time.perf_counter()
                    if d > 0:
                        time.sleep(d)
            th = threading.Thread(target=writer)
            th.start()
            t0 = time.perf_counter()
            steps = 0
            while time.perf_counter() - t0 < 4.0:
                run(ex, qbs, 20, depth, outs)
                steps += 20
            dt = time.perf_counter() - t0
            stop.set()
            th.join()
            print(json.dumps({"table": "steady writer", "staleness_s": bound, "changes_per_s": made[0] / dt,
                              "queries_per_s": args.batch * steps / dt, "ms_per_step": 1e3 * dt / steps,
                              "vs_static": args.batch * steps / dt / static_qps,
                              "delta_documents": shim_table.mutation_stats()["delta_documents"]}), flush=True)
        shim_table.set_mutation_staleness(0.0)
    # compaction: the main index rebuilt from the current documents (texts back from HBM, columns rebuilt, uploaded)
    t0 = time.perf_counter()
    shim_table.compact()
    compact_s = time.perf_counter() - t0
    run(ex, qbs, 10, depth, outs)
    dt = run(ex, qbs, args.steps, depth, outs)
    print(json.dumps({"table": "compacted", "compact_s": compact_s, "queries_per_s": args.batch * args.steps / dt,
                      "ms_per_step": 1e3 * dt / args.steps, "vs_static": args.batch * args.steps / dt / static_qps,
                      "stats": shim_table.mutation_stats()}), flush=True)


if __name__ == "__main__":
    main()
