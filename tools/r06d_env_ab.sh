#!/bin/bash
# Round 6d: runtime knobs that could touch the 5-15 us gaps between dependent launches (2 ms idle per step, mostly behind stream switches):
# hardware-queue count and interrupt-free completion signals.  Interleaved, bench.py --steps 12 --warmup 4.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r06d_env_ab.txt
: > $OUT
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$label', d['value'], d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['gated_block']['ms'])" >> $OUT
}
for r in 1 2; do
  run default X=1
  run hwq2 GPU_MAX_HW_QUEUES=2
  run hwq8 GPU_MAX_HW_QUEUES=8
  run nointr HSA_ENABLE_INTERRUPT=0
done
cat $OUT
