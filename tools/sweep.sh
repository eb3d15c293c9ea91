#!/bin/bash
# tools/sweep.sh "<env assignments>" ... : one kernel_sweep.py run per argument, appended to gpurun_out/sweep.jsonl
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout -k 10 240 python3 tools/kernel_sweep.py 2>/dev/null | tail -1 >> gpurun_out/sweep.jsonl || echo "{\"failed\": \"$cfg\"}" >> gpurun_out/sweep.jsonl
done
cat gpurun_out/sweep.jsonl
