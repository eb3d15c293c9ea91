#!/bin/bash
# rocprofv3 passes over one short bench.py run each (run on the GPU box, from the repo root):
#   tools/profile.sh <tag> [kernel-substring] [extra env assignments for bench.py ...]
# writes raw CSVs under gpurun_out/prof_<tag>/{stats,sq1,sq2,tcc,fetch,write} and the aggregated summaries
# gpurun_out/prof_<tag>/summary.json (tools/pmc_agg.py) + stats.csv. Counters are collected in their own passes with
# --kernel-trace only (no --sys-trace / --hip-trace beside --pmc: refused on this pool), FETCH_SIZE and WRITE_SIZE in
# separate passes (TCC slots). The program after `--` is python3 itself (no env/bash hop under the profiler).
set -u
TAG=${1:?tag}; KERNEL=${2:-bitmap_score_kernel}; shift; shift || true
for kv in "$@"; do export "$kv"; done
export MGX_BENCH_CPU_SECONDS=${MGX_BENCH_CPU_SECONDS:-0}
export MGX_BENCH_NO_POINTS=1   # (the extra operating points of bench.py run four batches time-sliced: not what is profiled)
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps ${PROFILE_STEPS:-6} --warmup 2"
run() { # name, rocprofv3 options...
  local name=$1; shift
  (cd /tmp && rocprofv3 "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1) || echo "pass $name failed"
  echo "pass $name done"
}
# (one batch in flight for the duration pass: with the default four, launches of different batches overlap on the device
#  and stretch each other — the trace then averages 1.3 ms for a kernel that takes 0.86 ms alone)
MGX_BENCH_DEPTH=${MGX_BENCH_DEPTH:-1} run stats --stats
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
python3 tools/pmc_agg.py "$OUT" --kernel "$KERNEL" --skip 2 --out "$OUT/summary.json" > /dev/null
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/stats.csv" \;
tail -n +1 "$OUT/summary.json" | head -80
