#!/bin/bash
# Round 6d: everything the round-6d K-loop work changed in csrc/gemm.hip (M0 per four pieces and a slot early, offset arithmetic on schedule slots, pieces spread
# wider, zero-C first k-step, K-major early M0 + offset slots) against the round-6c kernels (gemm.o of commit 6b81fa1 linked with today's other objects:
# libotter_hip_r6c.so), one box, interleaved: the six launch forms on cold operands, then the training step.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OLD=$PWD/otter_amd/lib/libotter_hip_r6c.so
OUT=gpurun_out/r06d_total_ab.txt
: > $OUT
for r in 1 2 3; do
  for v in r6d r6c; do
    if [ $v = r6c ]; then export OTTER_LIB_PATH=$OLD; else unset OTTER_LIB_PATH; fi
    echo "== launches, round $r $v" >> $OUT
    timeout 300 python tools/gemm_xt_ab.py 3 3 2>/dev/null | grep "bf16\|f32" >> $OUT
  done
done
for r in 1 2 3 4; do
  for v in r6d r6c; do
    if [ $v = r6c ]; then export OTTER_LIB_PATH=$OLD; else unset OTTER_LIB_PATH; fi
    timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('step $v', $r, d['value'], d['ms_per_step'], r['avg_us'], r['by_layout']['k_contiguous']['avg_us'], r['by_layout']['k_major']['avg_us'], r['gated_block']['ms'], r['gated_block']['frac'])" >> $OUT
  done
done
cat $OUT
