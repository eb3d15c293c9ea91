#!/usr/bin/env python3
"""Writes profiles/roofline_current.json — the counter figures bench.py prints in its `roofline` block — from the
summaries tools/profile.sh produced on the GPU box, tagged with the kernel source hash and the commit they belong to.
bench.py prints them only while the kernel sources still hash to the same value.

    python tools/make_roofline_profile.py --main gpurun_out/prof_r03_final --docid gpurun_out/prof_r03_docid \
        --lists gpurun_out/prof_r03_lists [--tag r03_final]

--main:  tools/profile.sh <tag> bitmap_score_kernel                       (the benchmark's dominant kernel)
--docid: tools/profile.sh <tag> wave_count_kernel MGX_BENCH_SORT=docid    (intersection only, bitmap-form lists)
--lists: tools/profile.sh <tag> merge_score_kernel MGX_BENCH_DENSE=2      (index WITHOUT bitmaps: sorted posting arrays)
--listsdoc: tools/profile.sh <tag> wave_count_kernel MGX_BENCH_SORT=docid MGX_BENCH_DENSE=2 MGX_CAND=0
                                                                           (the same arrays, intersection only)
The summaries (and the kernel-trace stats) are copied into profiles/ under the tag."""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import HBM_PEAK_GBS, kernel_source_sha16  # noqa: E402


def bench_line(prof_dir):
    """The bench JSON line printed inside the --stats pass (algorithmic bytes of the profiled configuration)."""
    try:
        for ln in reversed(open(os.path.join(prof_dir, "stats.log")).read().splitlines()):
            if ln.startswith("{") and '"metric"' in ln:
                return json.loads(ln)
    except (OSError, ValueError):
        pass
    return None


def stats_ms(prof_dir, kernel_substr):
    """Mean duration (ms) of the kernel in the --stats pass: one batch in flight, no counters — the isolated launch time."""
    import csv
    try:
        for r in csv.DictReader(open(os.path.join(prof_dir, "stats.csv"), newline="")):
            if kernel_substr in r["Name"]:
                return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    except (OSError, KeyError, ValueError):
        pass
    return None, 0


def take(prof_dir, tag, name):
    d = json.load(open(os.path.join(prof_dir, "summary.json")))
    dst = os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (tag, name))
    shutil.copy(os.path.join(prof_dir, "summary.json"), dst)
    st = os.path.join(prof_dir, "stats.csv")
    files = [os.path.relpath(dst, ROOT)]
    if os.path.exists(st):
        dst2 = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, name))
        shutil.copy(st, dst2)
        files.append(os.path.relpath(dst2, ROOT))
    return d, files


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--main", required=True)
    ap.add_argument("--docid")
    ap.add_argument("--lists")
    ap.add_argument("--listsdoc")
    ap.add_argument("--tag", default="r03_final")
    a = ap.parse_args()
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    main_d, files = take(a.main, a.tag, "main")
    dv = main_d["derived"]
    ms, calls = stats_ms(a.main, "bitmap_score_kernel<3>")
    out = {"kernel": "mgx::bitmap_score_kernel<3>", "kernel_source_sha16": kernel_source_sha16(), "commit": commit,
           "files": files, "kernel_ms": ms if ms else dv["kernel_ms_mean"], "kernel_ms_launches": calls,
           "kernel_ms_in_counter_passes": dv["kernel_ms_mean"], "fetch_bytes_raw": dv["fetch_bytes_raw"],
           "write_bytes": dv["write_bytes"], "traffic_bytes_per_launch": dv["traffic_bytes_corrected"],
           "l2_hit_rate": dv.get("l2_hit_rate"), "wave_cycles_waiting_fraction": dv.get("wave_cycles_waiting_fraction"),
           "intersection": {}}
    if a.docid:
        d, f = take(a.docid, a.tag, "docid")
        x = d["derived"]
        ms, _ = stats_ms(a.docid, "wave_count_kernel")
        x["kernel_ms_mean"] = ms or x["kernel_ms_mean"]
        gbps = x["traffic_bytes_corrected"] / (x["kernel_ms_mean"] * 1e-3) / 1e9
        out["intersection"]["bitmap_form"] = {
            "kernel": "mgx::wave_count_kernel (the benchmark batch without scoring: MGX_BENCH_SORT=docid)",
            "kernel_ms": x["kernel_ms_mean"], "traffic_bytes_per_launch": x["traffic_bytes_corrected"],
            "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBS, "files": f}
    if a.lists:
        d, f = take(a.lists, a.tag, "lists")
        x = d["derived"]
        ms, _ = stats_ms(a.lists, "merge_score_kernel")
        x["kernel_ms_mean"] = ms or x["kernel_ms_mean"]
        bl = bench_line(a.lists)
        alg = bl["roofline"]["algorithmic_bytes_per_launch"] if bl else None
        e = {"kernel": "mgx::merge_score_kernel (the benchmark batch on an index WITHOUT bitmaps: MGX_BENCH_DENSE=2, "
                       "sorted u32 posting arrays only; intersection + BM25 + top-k)",
             "kernel_ms": x["kernel_ms_mean"], "traffic_bytes_per_launch": x["traffic_bytes_corrected"],
             "traffic_GBps": x["traffic_bytes_corrected"] / (x["kernel_ms_mean"] * 1e-3) / 1e9, "files": f}
        if alg:
            e["algorithmic_bytes_per_launch"] = alg
            e["algorithmic_GBps"] = alg / (x["kernel_ms_mean"] * 1e-3) / 1e9
            e["frac"] = e["algorithmic_GBps"] / HBM_PEAK_GBS
            e["traffic_over_algorithmic"] = x["traffic_bytes_corrected"] / alg
        out["intersection"]["posting_arrays"] = e
    if a.listsdoc:
        d, f = take(a.listsdoc, a.tag, "listsdoc")
        x = d["derived"]
        ms, _ = stats_ms(a.listsdoc, "wave_count_kernel")
        x["kernel_ms_mean"] = ms or x["kernel_ms_mean"]
        bl = bench_line(a.listsdoc)
        alg = bl["roofline"]["algorithmic_bytes_per_launch"] if bl else None
        e = {"kernel": "mgx::wave_count_kernel<lists> (the benchmark batch WITHOUT scoring on an index WITHOUT bitmaps: "
                       "MGX_BENCH_SORT=docid MGX_BENCH_DENSE=2 MGX_CAND=0 — intersection of sorted u32 posting arrays, "
                       "exact totals, docid-DESC pages of 10)",
             "kernel_ms": x["kernel_ms_mean"], "traffic_bytes_per_launch": x["traffic_bytes_corrected"],
             "traffic_GBps": x["traffic_bytes_corrected"] / (x["kernel_ms_mean"] * 1e-3) / 1e9, "files": f}
        e["traffic_frac"] = e["traffic_GBps"] / HBM_PEAK_GBS
        if alg:
            e["algorithmic_bytes_per_launch"] = alg
            e["algorithmic_GBps"] = alg / (x["kernel_ms_mean"] * 1e-3) / 1e9
            e["frac"] = e["algorithmic_GBps"] / HBM_PEAK_GBS
            e["traffic_over_algorithmic"] = x["traffic_bytes_corrected"] / alg
        out["intersection"]["posting_arrays_intersection_only"] = e
    json.dump(out, open(os.path.join(ROOT, "profiles", "roofline_current.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
