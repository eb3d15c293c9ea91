#!/bin/bash
# tools/sweep.sh [-n DOCS] "<env assignments>" ... : one kernel_sweep.py run per argument, appended to gpurun_out/sweep.jsonl
DOCS=10000000
if [ "$1" = "-n" ]; then DOCS=$2; shift; shift; fi
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout -k 10 240 python3 tools/kernel_sweep.py $DOCS 2>/dev/null | tail -1 >> gpurun_out/sweep.jsonl || echo "{\"failed\": \"$cfg\"}" >> gpurun_out/sweep.jsonl
done
cat gpurun_out/sweep.jsonl
