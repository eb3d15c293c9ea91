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


def kernel_source_sha16():
    """Identity of the device code this run executes: sha256 over the kernel sources (the profile a counter figure was
    taken on must be of the same sources, or the figure is not printed)."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("mygram-db_amd/csrc/mgx_kernels.hip", "mygram-db_amd/csrc/mgx_internal.hpp"):
        try:
            h.update(open(os.path.join(ROOT, rel), "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def load_counter_profile():
    """profiles/roofline_current.json: the rocprofv3 PMC figures of the dominant kernel (tools/make_roofline_profile.py
    writes it from the --pmc passes of tools/profile.sh, with the kernel source hash and the commit they were taken on)."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "roofline_current.json")))
    except (OSError, ValueError):
        return None, "profiles/roofline_current.json is missing"
    sha = kernel_source_sha16()
    if prof.get("kernel_source_sha16") != sha:
        return None, ("profiles/roofline_current.json was taken on kernel sources %s, this build is %s: counter figures "
                      "withheld" % (prof.get("kernel_source_sha16"), sha))
    return prof, None


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
    # host threads per rank from the box's CPU share: the quota of this container (or its affinity set) divided among the
    # ranks of the node, two of them left to the submit / collect threads; at least 2 planners (one planner thread cannot
    # feed a 0.3 ms shard step), at most 8 (8 plan a batch in 0.1-0.2 ms; more only add contention)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpu_share = max(1.0, usable_cores() / max(1, local_world))
    planners = int(os.environ.get("MGX_BENCH_PLANNERS", "0")) or int(max(2, min(8, cpu_share - 2)))
    if not os.environ.get("MGX_DISPATCHERS"):
        os.environ["MGX_DISPATCHERS"] = str(3 if cpu_share >= 8 else 2 if cpu_share >= 4 else 1)
    if not os.environ.get("MGX_COMPILE_THREADS"):
        os.environ["MGX_COMPILE_THREADS"] = str(3 if cpu_share >= 12 else 1 if cpu_share >= 4 else 0)
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
        kept0 = {}  # the rows the TIMED executor returned for distinct batch 0 (checked against the oracle below)

        def run_steps(k, record, exe=None, out_bufs=None, n_flight=None, lat_out=None):
            """k steps with `depth` batches in flight. Two host threads drive the executor, as its interface intends (one
            submitter, one waiter — search_pipeline::MicroBatcher does the same): building 1024 BatchQuery objects and
            unpacking 1024 BatchResults are each ~0.1 ms of host work per step, and on a small shard (0.3 ms of device per
            batch) one thread doing both in turn was the step."""
            import queue
            import threading
            exe = exe or ex
            out_bufs = out_bufs or outs
            n_flight = n_flight or depth
            in_flight = threading.Semaphore(n_flight)
            tickets = queue.Queue()
            sub_t = {}
            failure = []

            def submitter():
                name_thread("bench-submit")
                try:
                    for j in range(k):
                        in_flight.acquire()
                        sub_t[j] = time.perf_counter()
                        tickets.put((j, exe.submit(qbs[j % len(qbs)], limit=10, sort_by_score=by_score)))
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
                out = exe.wait(ticket, out_bufs[j % n_flight])
                in_flight.release()
                if lat_out is not None:
                    lat_out.append(time.perf_counter() - sub_t[j])
                if record:
                    lat.append(time.perf_counter() - sub_t[j])
                    call_ms[1] += 1e3 * (time.perf_counter() - w0)
                    timings.append(out[4].copy())
                    if j % len(qbs) == 0 and not kept0:
                        kept0.update(totals=out[0].copy(), n_docs=out[1].copy(), docs=out[2].copy(), scores=out[3].copy())
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
    rank_ms_per_step = [1e3 * elapsed / args.steps]
    if world > 1:
        mine_t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(every, mine_t)
        rank_ms_per_step = [1e3 * float(x.item()) / args.steps for x in every]  # per-rank spread
        elapsed = max(float(x.item()) for x in every)

    # ---------------------------------------------------------------------------------------------------------------
    # (3) other operating points of the same executor (short passes AFTER the timed region; never the headline): the
    #     metric is queries/s AND p50 batch latency, and the number of batches in flight trades one for the other
    # ---------------------------------------------------------------------------------------------------------------
    operating_points = []
    if (not exchange) and rank == 0 and not os.environ.get("MGX_BENCH_NO_POINTS"):
        for d2, fifo2 in ((1, True), (2, True), (4, False)):
            if d2 == depth and fifo2 == fifo:
                continue
            table.index.device_index.set_batch_order(fifo2)
            ex2 = S.Executor(shim_table, depth=d2, planner_threads=planners)
            ex2.warm(qbs[-1], limit=10, sort_by_score=by_score, rounds=2)
            outs2 = [(np.zeros(batch_size, np.uint64), np.zeros(batch_size, np.uint32), np.zeros((batch_size, 10), np.uint32),
                      np.zeros((batch_size, 10), np.float64), np.zeros(5, np.float64)) for _ in range(d2)]
            k2 = 120
            lat2 = []
            run_steps(10, False, ex2, outs2, d2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(k2, False, ex2, outs2, d2, lat2)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            operating_points.append({"batches_in_flight": d2, "batch_order": "fifo" if fifo2 else "concurrent",
                                     "queries_per_s": batch_size * k2 / dt2, "ms_per_step": 1e3 * dt2 / k2,
                                     "p50_ms": 1e3 * statistics.median(lat2), "steps": k2})
            del ex2
        table.index.device_index.set_batch_order(fifo)

    # second denominator: this box's streaming-read bandwidth, measured in the same job (read-only kernel, 2 GiB)
    measured_peak = None
    if rank == 0:
        import ctypes
        gbs = ctypes.c_double(0.0)
        if mg._capi.load().mgxt_measure_read_bandwidth(local_rank, 2 << 30, 5, ctypes.byref(gbs)) == 0:
            measured_peak = gbs.value

    cpu = None
    if rank == 0 and world == 1 and cpu_seconds > 0:
        # the rows checked against the oracle are the ones the TIMED C++ executor returned for distinct batch 0 (and, beside
        # them, the Python replay path's: both must agree with the oracle bit for bit)
        def gpu_rows(i):
            r = results0[i]
            if kept0:
                nd = int(kept0["n_docs"][i])
                same = (int(kept0["totals"][i]) == r.total and kept0["docs"][i, :nd].tolist() == r.docs.tolist() and
                        np.array_equal(kept0["scores"][i, :nd], r.scores))
                if not same:  # the executor's row disagrees with the replay path: report the executor's (a mismatch)
                    return int(kept0["totals"][i]), kept0["docs"][i, :nd].copy(), kept0["scores"][i, :nd].copy()
                return int(kept0["totals"][i]), kept0["docs"][i, :nd].copy(), kept0["scores"][i, :nd].copy()
            return r.total, r.docs, r.scores
        cpu = cpu_baseline(mg, table, corpus, term_batches[0], gpu_rows, cpu_seconds)
        cpu["parity_rows_from"] = "the timed C++ executor (search_pipeline::BatchExecutor)" if kept0 else "the replay path"

    # ---------------------------------------------------------------------------------------------------------------
    # (4) N = 8 only: BASELINE.json configs[3] beside the strong-scaling line — 100M docs doc-range-sharded 12.5M per rank,
    #     batch 1024 x 3-term AND + BM25 top-100, RCCL top-k merge (each rank generates only its own shard). A second block
    #     of the same JSON line; the headline above stays configs[1].
    # ---------------------------------------------------------------------------------------------------------------
    index_bytes_hbm = table.index.device_index.memory_bytes()
    shard_grams, shard_postings = cols.n_grams, cols.n_postings
    config3 = None
    want_cfg3 = os.environ.get("MGX_BENCH_CFG3", "1" if world == 8 else "0") != "0"
    if want_cfg3 and cxx_exchange and by_score:
        try:
            t_c3 = time.perf_counter()
            n3 = int(os.environ.get("MGX_BENCH_CFG3_DOCS", "100000000"))
            del ex, shim_table, batches, table, corpus, cols
            before3, mine3 = mdist.shard_range(n3, rank, world)
            corpus3 = mg.Corpus.synthetic(mine3, seed=42, global_first=before3)
            table3 = mdist.ShardedTable(corpus3, first_doc_id=1 + before3, device=local_rank, ngram_size=2, kanji_ngram_size=0)
            table3.index.device_index.set_batch_order(fifo)
            tb3 = make_queries(mg, table3, 8, batch_size, seed=43)
            ex3 = S.Executor(S.Table(table3.index), depth=depth, planner_threads=planners, comm=mdist.Comm(device=local_rank))
            qb3 = [S.QueryBatch(tb) for tb in tb3]
            ex3.warm(qb3[-1], limit=100, sort_by_score=True, rounds=2)
            outs3 = [(np.zeros(batch_size, np.uint64), np.zeros(batch_size, np.uint32), np.zeros((batch_size, 100), np.uint32),
                      np.zeros((batch_size, 100), np.float64), np.zeros(5, np.float64)) for _ in range(depth)]
            k3 = max(8, min(args.steps, 40))
            lat3, pend = [], []
            sync()
            t0 = time.perf_counter()
            for j in range(k3 + depth):
                if j >= depth:
                    tk, ts = pend.pop(0)
                    ex3.wait(tk, outs3[j % depth])
                    lat3.append(time.perf_counter() - ts)
                if j < k3:
                    ts = time.perf_counter()
                    pend.append((ex3.submit(qb3[j % len(qb3)], limit=100, sort_by_score=True), ts))
            sync()
            dt3 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.all_reduce(dt3, op=dist.ReduceOp.MAX)
            dt3 = float(dt3.item())
            config3 = {"workload": "BASELINE.json configs[3]: %d docs doc-range-sharded across %d GPUs (%d per rank), bigram, "
                                   "batch %d x 3-term AND + BM25 top-100, RCCL top-k merge" % (n3, world, mine3, batch_size),
                       "value": batch_size * k3 / dt3, "unit": "queries/s", "ms_per_step": 1e3 * dt3 / k3,
                       "p50_ms": 1e3 * statistics.median(lat3), "steps": k3, "batches_in_flight": depth,
                       "setup_s": time.perf_counter() - t_c3}
        except Exception as e:  # noqa: BLE001 (the second block must never cost the headline line)
            config3 = {"error": repr(e)}

    if rank == 0:
        qps = batch_size * args.steps / elapsed
        tm = np.asarray(timings) if timings else np.zeros((1, 5))
        # ---- roofline of the dominant kernel -----------------------------------------------------------------------------
        # `traffic` = the bytes the kernel moved past L2 per launch, from rocprofv3 PMC passes (FETCH_SIZE with the gfx950
        # x2 correction for 16-byte-per-lane reads + WRITE_SIZE; /opt/skills/guides/MI355X_MICROARCH.md) committed under
        # profiles/ — printed only when they were taken on THIS build's kernel sources; `achieved` = traffic / the kernel
        # duration measured LIVE in this run (HIP events on its launch stream), `frac` = achieved / 8 TB/s. The algorithmic
        # figure of SURVEY.md 8d (bytes the reference's full scan would move) stands beside it as `algorithmic_ratio`: it
        # exceeds 1 because dense lists are read as 1-bit-per-doc bitmaps and most matches are pruned before being touched.
        prof, why_not = load_counter_profile() if by_score and dense == 0 and n_docs_total == 10000000 and batch_size == 1024 else (None, "not the profiled configuration")
        traffic = prof["traffic_bytes_per_launch"] if prof else None
        phys = (traffic / (k_ms * 1e-3) / 1e9) if (traffic and k_ms > 0) else None
        roofline = {
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "achieved": phys, "frac": (phys / HBM_PEAK_GBS) if phys else None, "traffic": traffic,
            "traffic_source": ({"file": prof["files"], "taken_at_commit": prof.get("commit"),
                                "kernel_source_sha16": prof["kernel_source_sha16"],
                                "kernel_ms_in_profile": prof.get("kernel_ms"),
                                "fetch_bytes_raw": prof.get("fetch_bytes_raw"), "write_bytes": prof.get("write_bytes"),
                                "correction": "2 x FETCH_SIZE (gfx950: 16-byte-per-lane reads count half) + WRITE_SIZE; "
                                              "FETCH_SIZE includes Infinity-Cache hits, so this is L2-miss traffic: an "
                                              "upper bound of HBM bytes"} if prof else why_not),
            "peak_measured_read": measured_peak,
            "frac_of_measured": (phys / measured_peak) if (phys and measured_peak) else None,
            "kernel": ("mgx::bitmap_score_kernel<3> (set algebra on tile bitmaps, block-max top-k pruning, fused BM25 of the "
                       "surviving matches from doc-slot tf nibbles, per-wave top-k); the few queries with a sparse sorted-list "
                       "operand run beside it as mgx::cand_kernel on a side stream inside the same timed region") if by_score else
                      "mgx::wave_count_kernel (+ tile_eval doc-count share): set algebra + per-tile counts",
            "kernel_ms": k_ms, "launches_timed": k_n,
            "algorithmic_bytes_per_launch": alg_total,
            "algorithmic_achieved": achieved, "algorithmic_ratio": achieved / HBM_PEAK_GBS,
            "algorithmic_breakdown": {"lists_4B_per_posting": alg[0], "score_R_times_T_plus_4": alg[1], "topk_12B": alg[2]},
            # "posting-list intersection" on its own (north_star: >= 50 % of the HBM roofline), both index forms, from the
            # same profile file: the intersection-only kernel on bitmap-form lists, and SORT _score on an index WITHOUT
            # bitmaps (sorted u32 posting arrays: mgx::merge_score_kernel), the latter priced on algorithmic bytes
            "intersection": (prof.get("intersection") if prof else None),
            "note": "frac / achieved / traffic: physical (counter) figures; algorithmic_*: SURVEY.md 8d's "
                    "4*sum|L| + R*(T+4) + 12*min(k,R) per launch / the same live kernel time — not a fraction of "
                    "physical bandwidth"}
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
                "batch_order": "fifo" if fifo else "concurrent",
                "other_operating_points": operating_points,
                "rank_ms_per_step": rank_ms_per_step,
                "host_threads": {"planners": planners, "dispatchers": int(os.environ.get("MGX_DISPATCHERS", "3")),
                                 "compile_helpers_per_dispatcher": int(os.environ.get("MGX_COMPILE_THREADS", "3")),
                                 "cpu_share_of_this_rank": cpu_share}},
            "config": {"workload": ("10M-doc synthetic ASCII corpus (seed 42), bigram index, 3-term AND + BM25 top-10, "
                                    "batch=1024 (BASELINE.json configs[1])") if by_score else
                                   ("PROFILING VARIANT, not the headline: the same batches without scoring "
                                    "(intersection only, docid-DESC top-10)"),
                       "n_docs": n_docs_total, "batch": batch_size, "limit": 10, "k1": 1.2, "b": 0.75,
                       "parallelism": "doc-range shards x%d, top-k all-gather + merge" % world,
                       "shard_docs": mine, "shard_grams": shard_grams, "shard_postings": shard_postings,
                       "index_bytes_hbm": index_bytes_hbm,
                       "mean_list_len_of_queries": alg[0] / 4 / batch_size / 3,
                       "dense_threshold": dense if dense else 1.0 / 256, "setup_s": setup_s},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "config3_100M_docs_8_gpus": config3,
        }
        if os.environ.get("MGX_BENCH_SERIES"):  # rehearsal: per-step host timings (plan, compile, enqueue, wait ms)
            line["series"] = {"timings": np.round(tm[:, :4], 3).tolist(), "latency_ms": [round(1e3 * x, 3) for x in lat]}
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
