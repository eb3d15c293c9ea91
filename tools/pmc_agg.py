#!/usr/bin/env python3
"""Aggregates rocprofv3 CSV output (kernel trace and/or counter collection) into one small JSON summary per kernel.

    python tools/pmc_agg.py <dir> [<dir> ...] --kernel and_score_kernel [--skip 2] [--out profiles/x.json]

Every *counter_collection.csv under the given directories contributes (counter name -> mean value per dispatch of the
kernels whose name contains --kernel, after skipping the first --skip dispatches of each pass: warm-up launches);
every *kernel_trace.csv contributes the mean duration. Separate --pmc passes (rocprofv3 cannot collect FETCH_SIZE and
WRITE_SIZE together, /opt/skills/guides/MI355X_MICROARCH.md 'rocprofv3 PMC slots') are simply given as several
directories. Derived figures follow that guide: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request, so the corrected read traffic is 2 x FETCH_SIZE; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS).
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(path):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            yield r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--skip", type=int, default=0, help="dispatches of the kernel to skip at the start of every pass")
    ap.add_argument("--out")
    ap.add_argument("--note", default="")
    a = ap.parse_args()

    counters = defaultdict(list)   # name -> per-dispatch values
    durations = []
    grid = wg = vgpr = sgpr = lds = scratch = None
    files = []
    for d in a.dirs:
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            files.append(path)
            per_dispatch = defaultdict(dict)  # dispatch id -> {counter: value}
            order = []
            for r in rows(path):
                if a.kernel not in r.get("Kernel_Name", ""):
                    continue
                did = r.get("Dispatch_Id") or r.get("Correlation_Id")
                if did not in per_dispatch:
                    order.append(did)
                per_dispatch[did][r["Counter_Name"]] = per_dispatch[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                grid, wg = r.get("Grid_Size", grid), r.get("Workgroup_Size", wg)
                vgpr, sgpr = r.get("VGPR_Count", vgpr), r.get("SGPR_Count", sgpr)
                lds, scratch = r.get("LDS_Block_Size", lds), r.get("Scratch_Size", scratch)
            for did in order[a.skip:]:
                for k, v in per_dispatch[did].items():
                    counters[k].append(v)
        for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            files.append(path)
            seen = 0
            for r in rows(path):
                if a.kernel not in r.get("Kernel_Name", ""):
                    continue
                seen += 1
                if seen <= a.skip:
                    continue
                durations.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not counters and not durations:
        sys.exit("no dispatch of a kernel matching %r under %s" % (a.kernel, a.dirs))
    mean = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
    out = {"kernel": a.kernel, "files": files, "dispatches_per_counter": {k: len(v) for k, v in sorted(counters.items())},
           "counters_mean_per_dispatch": mean, "note": a.note,
           "launch": {"grid": grid, "workgroup": wg, "vgpr": vgpr, "sgpr": sgpr, "lds": lds, "scratch": scratch}}
    der = {}
    if durations:
        der["kernel_ms_mean"] = sum(durations) / len(durations) / 1e6
        der["kernel_launches"] = len(durations)
    if "FETCH_SIZE" in mean:
        der["fetch_bytes_raw"] = mean["FETCH_SIZE"] * 1024
        der["fetch_bytes_gfx950_corrected_x2"] = 2 * mean["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in mean:
        der["write_bytes"] = mean["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean and mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"] > 0:
        der["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in mean and mean["SQ_WAVE_CYCLES"] > 0:
        for k, name in (("SQ_WAIT_ANY", "wave_cycles_waiting_fraction"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stall_fraction"),
                        ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing_fraction")):
            if k in mean:
                der[name] = mean[k] / mean["SQ_WAVE_CYCLES"]
    if durations and "fetch_bytes_gfx950_corrected_x2" in der:
        t = der["kernel_ms_mean"] * 1e-3
        der["traffic_bytes_corrected"] = der["fetch_bytes_gfx950_corrected_x2"] + der.get("write_bytes", 0.0)
        der["traffic_GBps_corrected"] = der["traffic_bytes_corrected"] / t / 1e9
        der["traffic_GBps_raw"] = (der["fetch_bytes_raw"] + der.get("write_bytes", 0.0)) / t / 1e9
    out["derived"] = der
    text = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
