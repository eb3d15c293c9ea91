#!/usr/bin/env python3
"""bench.py — queries/sec + p50 latency of batched 3-term AND + BM25 top-10 on a 10M-doc bigram index (MI355X).

One step = one FRESH batch of 1024 queries, end to end, in C++ (search_pipeline::BatchExecutor behind
libmygram_shim.so): per-query planning as ExecuteFullPipeline's regular branch does it (normalise, n-grams, dictionary
lookup, estimated sizes, sort, idf — src/server/search_pipeline.cpp:2004-2014) -> query compilation + item scheduling
(mgx_batch_reset) -> one asynchronous upload of the batch -> kernels (set algebra + fused BM25 + per-workgroup top-k,
merge) -> results copied to pinned host memory -> BatchResult objects. Four batches are in flight (six on a shard): the
host plans and compiles the next ones while the device runs one. 32 distinct batches are cycled; nothing of a batch is cached between steps.
The index is resident in HBM; the query strings are resident in host memory (they are what a front end hands over).
`value` is that end-to-end rate. The kernel-replay rate of round 1 (prepared batches re-executed) is reported beside it
as `replay_qps`, and the dominant kernel's duration comes from HIP events around it in the replay loop.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1 is STRONG scaling: the same 10M-doc table is cut into N contiguous doc-range shards, one per rank; every rank
runs the whole batch on its shard, per-shard top-k lists are exchanged (RCCL all-gather) and merged.
Environment knobs (rehearsals only; the defaults are the benchmark): MGX_BENCH_DOCS, MGX_BENCH_BATCH,
MGX_BENCH_CPU_SECONDS, MGX_BENCH_DENSE (dense_threshold; >=2 disables the bitmap form of dense lists),
MGX_BENCH_DEPTH (batches in flight), MGX_BENCH_PLANNERS (host planner threads).
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X datasheet HBM3E peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_DISTINCT_BATCHES = 32


def make_queries(mg, table, n_batches, batch, seed=42, limit=10):
    """3 distinct letter-only bigrams per query, sampled proportionally to their (global) document frequency, among the
    grams every shard knows. -> list of batches of term lists."""
    import torch.distributed as dist
    keys, sizes = table.keys, table.global_sizes
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
            qs.append([cand[i].decode() for i in pick])
        out.append(qs)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Physical cores this process may run on (distinct (package, core) pairs of its affinity set)."""
    allowed = sorted(os.sched_getaffinity(0))
    seen = set()
    for cpu in allowed:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            seen.add(("?", str(cpu)))
    return max(1, len(seen)), len(allowed)


def cpu_quota():
    """CPUs' worth of time this container may use per period (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited:
    a GPU box shows all 256 logical CPUs of its host to every tenant but schedules only its share of them."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def usable_cores():
    """Threads worth running: the physical cores of the affinity set, capped by the container's CPU quota."""
    cores = physical_cores()[0]
    quota = cpu_quota()
    return max(1, min(cores, int(math.ceil(quota)))) if quota else cores


def name_thread(name):
    """prctl(PR_SET_NAME) for the calling thread (so that thread_cpu_ms can tell the driver's threads from the runtime's)."""
    try:
        import ctypes
        ctypes.CDLL(None).prctl(15, name.encode()[:15], 0, 0, 0)
    except (OSError, AttributeError):
        pass


def thread_cpu_ms():
    """CPU time of every thread of this process so far, by thread name (ms): /proc/self/task/<tid>/schedstat holds the
    nanoseconds the thread has run. The host layer names its threads (mgx-plan, mgx-dispatch, mgx-compile)."""
    out = {}
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                name = open("/proc/self/task/%s/comm" % tid).read().strip()
                ns = int(open("/proc/self/task/%s/schedstat" % tid).read().split()[0])
            except (OSError, ValueError, IndexError):
                continue
            out[name] = out.get(name, 0.0) + ns / 1e6
    except OSError:
        pass
    return out


def cpu_baseline(mg, table, corpus, term_lists, gpu_rows, seconds):
    """The CPU oracle (oracle/mygram_oracle.c: a C restatement of the reference's per-query path — df by text scan of
    every term's candidates, Execute, ScoreDocuments with tf by text scan, SortByScore) on the host cores of this box:
    first single-threaded, then one worker per physical core, each worker running whole queries.
    The same queries check the GPU results (docids, totals, scores bit for bit)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    c = table.index.columns
    oidx = O.Index.from_csr(2, 0, True, c.key_bytes, c.key_off, c.offsets, c.docids)
    ostore = O.DocumentStore.from_arrays(corpus.text_bytes, corpus.text_off)
    n, avg = table.index.total_docs, table.index.avg_doc_length
    visible, logical = physical_cores()
    cores = usable_cores()

    def one(i):
        t0 = time.perf_counter()
        r = O.search_scored(oidx, ostore, term_lists[i], n, avg, 1.2, 0.75, True, 10, 0)  # (ctypes drops the GIL)
        return i, r, time.perf_counter() - t0

    # 1 thread: a few queries, bounded by time
    lat1, done, t0 = [], 0, time.perf_counter()
    mismatches = checked = 0

    def check(i, r):
        nonlocal mismatches, checked
        total, docs, scores = r
        g_total, g_docs, g_scores = gpu_rows(i)
        checked += 1
        if g_total != total or g_docs.tolist() != docs.tolist() or not np.array_equal(g_scores, scores):
            mismatches += 1

    while done < len(term_lists) and (done < 4 or time.perf_counter() - t0 < seconds * 0.35):
        i, r, dt = one(done)
        lat1.append(dt)
        check(i, r)
        done += 1
    dt1 = time.perf_counter() - t0
    qps1 = done / dt1
    # all cores: enough queries for >= ~seconds*0.65 of wall time, at least 256 when they fit
    budget = max(seconds * 0.65, 1.0)
    want = int(min(len(term_lists) - done, max(4 * cores, min(512, qps1 * cores * budget))))
    want = max(want, min(len(term_lists) - done, cores))
    latn = []
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        for i, r, dt in pool.map(one, range(done, done + want)):
            latn.append(dt)
            check(i, r)
    dtn = time.perf_counter() - t0
    latn.sort()
    return {"value": want / dtn, "unit": "queries/s", "cores": cores, "threads": cores, "physical_cores_visible": visible,
            "logical_cpus": logical, "cpu_quota": cpu_quota(), "cpu_model": cpu_model(), "kind": "port",
            "value_1_thread": qps1, "queries_1_thread": done, "queries_all_cores": want,
            "p50_ms_per_query": 1e3 * latn[len(latn) // 2], "p99_ms_per_query": 1e3 * latn[min(len(latn) - 1, int(len(latn) * 0.99))],
            "p50_ms_per_query_1_thread": 1e3 * statistics.median(lat1),
            "sample": "batch 0 of the benchmark (same 10M-doc corpus, same 3-term AND + BM25 top-10): %d queries on 1 "
                      "thread (%.1f s), then %d queries on %d threads = one per core this container may use (its "
                      "cgroup CPU quota, or the physical cores of its affinity set) (%.1f s); per query: df "
                      "by text scan of every term's candidates + Execute + ScoreDocuments (tf by text scan) + "
                      "SortByScore, as search_pipeline.cpp:2004-2019 + search_handler.cpp:454-470 do" %
                      (done, dt1, want, cores, dtn),
            "parity_checked": checked, "parity_mismatches": mismatches}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
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

    if not (os.path.exists(os.path.join(ROOT, "mygram-db_amd", "libmygram_gpu.so")) and
            os.path.exists(os.path.join(ROOT, "mygram-db_amd", "libmygram_shim.so"))):
        entry.build()
    mg = entry.load_package()
    from mygram_db_amd import _shim_capi as S
    from mygram_db_amd import dist as mdist

    n_docs_total = int(os.environ.get("MGX_BENCH_DOCS", "10000000"))
    batch_size = int(os.environ.get("MGX_BENCH_BATCH", "1024"))
    cpu_seconds = float(os.environ.get("MGX_BENCH_CPU_SECONDS", "24"))
    dense = float(os.environ.get("MGX_BENCH_DENSE", "0"))
    # batches in flight: a batch spends ~1.3 ms of host stages (submit, plan, compile, enqueue, collect) around its device
    # time, so a small shard (0.3 ms of device per batch) needs six slots to keep the device busy, the whole table four
    # (round 3: with the host stages at ~0.5 ms per batch and batches running first-in-first-out on the device, TWO in
    # flight keep the whole table's device busy — 1.17M q/s at 1.7 ms p50 against 1.22M at 3.4 ms with four time-sliced
    # ones; a 1.25M-doc shard, 0.26 ms of device per batch, still wants six, time-sliced: 0.27 vs 0.29 ms per step)
    sharded_run = int(os.environ.get("WORLD_SIZE", "1")) > 1 or bool(os.environ.get("MGX_FORCE_EXCHANGE"))
    depth = int(os.environ.get("MGX_BENCH_DEPTH", "0")) or (6 if sharded_run else 2)
    fifo = (os.environ.get("MGX_BENCH_FIFO", "") or ("0" if sharded_run else "1")) != "0"
    planners = int(os.environ.get("MGX_BENCH_PLANNERS", "0")) or max(1, min(8, (usable_cores() - 1) // max(1, world)))  # (8 plan a batch in 0.15-0.2 ms; 15 only add contention on a 16-CPU box)
    exchange = world > 1 or bool(os.environ.get("MGX_FORCE_EXCHANGE"))
    # profiling variant (never the headline): MGX_BENCH_SORT=docid runs the same 3-term AND batches WITHOUT scoring —
    # the intersection-only path (mgx::wave_count_kernel + page emit), docid-DESC pages of 10
    by_score = os.environ.get("MGX_BENCH_SORT", "score") != "docid"

    t_setup = time.perf_counter()
    before, mine = mdist.shard_range(n_docs_total, rank, world)
    corpus = mg.Corpus.synthetic(mine, seed=42, global_first=before)
    table = mdist.ShardedTable(corpus, first_doc_id=1 + before, device=local_rank, ngram_size=2, kanji_ngram_size=0,
                               dense_threshold=dense)
    cols = table.index.columns
    table.index.device_index.set_batch_order(fifo)
    term_batches = make_queries(mg, table, N_DISTINCT_BATCHES, batch_size)
    setup_s = time.perf_counter() - t_setup

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---------------------------------------------------------------------------------------------------------------
    # (1) replay loop: 4 prepared batches re-executed — isolates the device side; the dominant kernel is timed by HIP
    #     events on its launch stream, algorithmic bytes come from the fetched match counts
    # ---------------------------------------------------------------------------------------------------------------
    replay_q = [[mg.engine.Query(t, sort_score=by_score, limit=10) for t in tb] for tb in term_batches[:4]]
    batches = [table.prepare(qs) for qs in replay_q]
    replay_steps = max(8, min(args.steps, 40))

    def replay(i):
        b = batches[i % len(batches)]
        table.run(b)
        b.fetch_raw()

    for i in range(4):
        replay(i)
    for b in batches:
        b.kernel_time_ms()  # start recording HIP events around the tile kernel from here on
    sync()
    t0 = time.perf_counter()
    for i in range(replay_steps):
        replay(i)
    sync()
    replay_elapsed = time.perf_counter() - t0
    k_ms, k_n = 0.0, 0
    for b in batches:
        ms, n = b.kernel_time_ms()
        if n:
            k_ms += ms * n
            k_n += n
    k_ms = k_ms / k_n if k_n else 0.0
    per_batch = []
    for b in batches:
        if world > 1:  # totals in h_results are table-wide after a merge; re-run locally for this shard's own counts
            b.execute(torch.cuda.current_stream().cuda_stream)
            b.fetch_raw()
        per_batch.append(b.algorithmic_bytes())
    alg = [sum(x[i] for x in per_batch) / len(per_batch) for i in range(3)]
    alg_total = sum(alg)
    achieved = alg_total / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    results0 = batches[0].fetch() if rank == 0 else None

    # ---------------------------------------------------------------------------------------------------------------
    # (2) the timed region: fresh batches end to end in C++; N > 1 adds the RCCL exchange inside the same C++ executor
    # ---------------------------------------------------------------------------------------------------------------
    lat, timings = [], []
    cxx_exchange = exchange and dist.get_backend() == "nccl"  # (gloo: the CPU rehearsal of dist.py, tests only)
    if not exchange or cxx_exchange:
        shim_table = S.Table(table.index)
        comm = mdist.Comm(device=local_rank) if cxx_exchange else None  # RCCL communicator behind the C ABI (mgx_comm_create)
        ex = S.Executor(shim_table, depth=depth, planner_threads=planners, comm=comm)
        qbs = [S.QueryBatch(tb) for tb in term_batches]
        # setup (outside the clock, before the driver's own warm-up steps): every slot's arenas, pinned blocks, streams and
        # helper threads exist before the first step — BatchExecutor::Warm runs one sample batch through each slot and
        # discards the results; the timed steps still plan, compile and run every batch from scratch
        ex.warm(qbs[-1], limit=10, sort_by_score=by_score, rounds=3)
        outs = [(np.zeros(batch_size, np.uint64), np.zeros(batch_size, np.uint32), np.zeros((batch_size, 10), np.uint32),
                 np.zeros((batch_size, 10), np.float64), np.zeros(5, np.float64)) for _ in range(depth)]

        call_ms = [0.0, 0.0]  # time the driving thread spends inside submit / wait calls (sum over the timed steps)

        def run_steps(k, record):
            """k steps with `depth` batches in flight. Two host threads drive the executor, as its interface intends (one
            submitter, one waiter — search_pipeline::MicroBatcher does the same): building 1024 BatchQuery objects and
            unpacking 1024 BatchResults are each ~0.1 ms of host work per step, and on a small shard (0.3 ms of device per
            batch) one thread doing both in turn was the step."""
            import queue
            import threading
            in_flight = threading.Semaphore(depth)
            tickets = queue.Queue()
            sub_t = {}
            failure = []

            def submitter():
                name_thread("bench-submit")
                try:
                    for j in range(k):
                        in_flight.acquire()
                        sub_t[j] = time.perf_counter()
                        tickets.put((j, ex.submit(qbs[j % len(qbs)], limit=10, sort_by_score=by_score)))
                        if record:
                            call_ms[0] += 1e3 * (time.perf_counter() - sub_t[j])
                except BaseException as e:  # noqa: BLE001 (handed to the waiting thread)
                    failure.append(e)
                    tickets.put(None)

            th = threading.Thread(target=submitter, daemon=True)
            th.start()
            for _ in range(k):
                item = tickets.get()
                if item is None:
                    raise failure[0]
                j, ticket = item
                w0 = time.perf_counter()
                out = ex.wait(ticket, outs[j % depth])
                in_flight.release()
                if record:
                    lat.append(time.perf_counter() - sub_t[j])
                    call_ms[1] += 1e3 * (time.perf_counter() - w0)
                    timings.append(out[4].copy())
            th.join()

        name_thread("bench-wait")
        run_steps(args.warmup, False)
        sync()
        cpu0 = time.process_time()
        thr0 = thread_cpu_ms()
        t0 = time.perf_counter()
        run_steps(args.steps, True)
        sync()
        elapsed = time.perf_counter() - t0
        host_cpu_ms = 1e3 * (time.process_time() - cpu0) / max(1, args.steps)  # CPU time of ALL threads of this rank
        thr1 = thread_cpu_ms()
        host_cpu_by_thread = {k: round((v - thr0.get(k, 0.0)) / max(1, args.steps), 4) for k, v in thr1.items()
                              if v - thr0.get(k, 0.0) > 0.0005 * args.steps}
    else:
        ex_batches = batches

        def step(i):
            s0 = time.perf_counter()
            b = ex_batches[i % len(ex_batches)]
            table.run(b)
            b.fetch_raw()
            lat.append(time.perf_counter() - s0)

        for i in range(args.warmup):
            step(i)
        lat.clear()
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        sync()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # second denominator: this box's streaming-read bandwidth, measured in the same job (read-only kernel, 2 GiB)
    measured_peak = None
    if rank == 0:
        import ctypes
        gbs = ctypes.c_double(0.0)
        if mg._capi.load().mgxt_measure_read_bandwidth(local_rank, 2 << 30, 5, ctypes.byref(gbs)) == 0:
            measured_peak = gbs.value

    cpu = None
    if rank == 0 and world == 1 and cpu_seconds > 0:
        def gpu_rows(i):
            r = results0[i]
            return r.total, r.docs, r.scores
        cpu = cpu_baseline(mg, table, corpus, term_batches[0], gpu_rows, cpu_seconds)

    if rank == 0:
        qps = batch_size * args.steps / elapsed
        tm = np.asarray(timings) if timings else np.zeros((1, 5))
        line = {
            "metric": "queries/sec + p50 latency, 10M-doc bigram index, batch=1024 3-term AND",
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "p50_ms": 1e3 * statistics.median(lat),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 set algebra + f64 BM25",
            "data": "synthetic",
            "end_to_end": {
                "what": ("fresh batch per step, in C++: plan (GenerateTermInfos, size sort, idf) -> compile + schedule "
                         "(mgx_batch_reset) -> async upload -> kernels -> pinned results -> BatchResult; %d batches in "
                         "flight, %d distinct batches cycled, %d host planner threads" % (depth, N_DISTINCT_BATCHES, planners))
                        + ("; every batch's per-shard top-k all-gathered over RCCL (mgx_batch_exchange) and merged on "
                           "the batch's stream, in C++" if cxx_exchange else "")
                        if (not exchange or cxx_exchange) else "prepared batches + per-step gloo exchange (dist.ShardedTable.run)",
                "end_to_end_qps": qps,
                "prepare_ms": float(tm[:, 0].mean() + tm[:, 1].mean()), "plan_ms": float(tm[:, 0].mean()),
                "compile_ms": float(tm[:, 1].mean()), "enqueue_ms": float(tm[:, 2].mean()),
                "wait_ms": float(tm[:, 3].mean()),
                "driver_submit_call_ms": (call_ms[0] / args.steps) if (not exchange or cxx_exchange) else None,
                "driver_wait_call_ms": (call_ms[1] / args.steps) if (not exchange or cxx_exchange) else None,
                "host_cpu_ms_per_step": host_cpu_ms if (not exchange or cxx_exchange) else None,
                "host_cpu_ms_per_step_by_thread": host_cpu_by_thread if (not exchange or cxx_exchange) else None,
                "execute_ms": 1e3 * replay_elapsed / replay_steps,
                "replay_qps": batch_size * replay_steps / replay_elapsed, "replay_steps": replay_steps,
                "batch_latency_p50_ms": 1e3 * statistics.median(lat), "batches_in_flight": depth,
                "batch_order": "fifo" if fifo else "concurrent"},
            "config": {"workload": ("10M-doc synthetic ASCII corpus (seed 42), bigram index, 3-term AND + BM25 top-10, "
                                    "batch=1024 (BASELINE.json configs[1])") if by_score else
                                   ("PROFILING VARIANT, not the headline: the same batches without scoring "
                                    "(intersection only, docid-DESC top-10)"),
                       "n_docs": n_docs_total, "batch": batch_size, "limit": 10, "k1": 1.2, "b": 0.75,
                       "parallelism": "doc-range shards x%d, top-k all-gather + merge" % world,
                       "shard_docs": mine, "shard_grams": cols.n_grams, "shard_postings": cols.n_postings,
                       "index_bytes_hbm": table.index.device_index.memory_bytes(),
                       "mean_list_len_of_queries": alg[0] / 4 / batch_size / 3,
                       "dense_threshold": dense if dense else 1.0 / 256, "setup_s": setup_s},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None,
                         "peak_measured_read": measured_peak,
                         "frac_of_measured": (achieved / measured_peak) if measured_peak else None,
                         "kernel": ("mgx::bitmap_score_kernel<3> (set algebra on tile bitmaps, block-max top-k pruning, fused BM25 "
                                    "of the surviving matches from doc-slot tf nibbles, per-wave top-k; queries with a "
                                    "sorted-list operand run beside it as mgx::wave_score_lists_kernel on a side stream "
                                    "inside the same timed region)") if by_score else
                                   "mgx::wave_count_kernel (+ tile_eval doc-count share): set algebra + per-tile counts",
                         "kernel_ms": k_ms, "launches_timed": k_n,
                         "algorithmic_bytes_per_launch": alg_total,
                         "algorithmic_breakdown": {"lists_4B_per_posting": alg[0], "score_R_times_T_plus_4": alg[1],
                                                   "topk_12B": alg[2]},
                         "note": "ALGORITHMIC figure: bytes the reference's full-scan semantics would move (SURVEY.md 8d: "
                                 "4*sum|L| + R*(T+4) + 12*min(k,R)) per launch / HIP-event kernel time. Dense lists are "
                                 "read as 1-bit-per-doc bitmaps and tiles are shared through L2/MALL, so it exceeds 1.0 "
                                 "and is NOT a fraction of physical bandwidth; the counter-measured traffic of this "
                                 "kernel (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) is in "
                                 "profiles/r02_*_summary.json with the build it was taken on, not replayed here"},
            "cpu_baseline": cpu,
        }
        if os.environ.get("MGX_BENCH_SERIES"):  # rehearsal: per-step host timings (plan, compile, enqueue, wait ms)
            line["series"] = {"timings": np.round(tm[:, :4], 3).tolist(), "latency_ms": [round(1e3 * x, 3) for x in lat]}
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
