#!/bin/bash
# Round 6d: the training step with M0 written once per four LDS-DMA pieces (default build) against one s_mov per piece (libotter_hip_m0g0.so), interleaved.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OLD=$PWD/otter_amd/lib/libotter_hip_m0g0.so
OUT=gpurun_out/r06d_m0group_step_ab.txt
: > $OUT
for r in 1 2 3 4; do
  for leg in grouped per_piece; do
    if [ $leg = per_piece ]; then export OTTER_LIB_PATH=$OLD; else unset OTTER_LIB_PATH; fi
    timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$leg', $r, d['value'], d['ms_per_step'], r['avg_us'], r['by_layout']['k_contiguous']['avg_us'], r['by_layout']['k_major']['avg_us'], r['gated_block']['ms'])" >> $OUT
  done
done
cat $OUT
