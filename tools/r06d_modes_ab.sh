#!/bin/bash
# Round 6d: frozen-decoder GEMM modes after the round-6d K-loop changes (M0 per four pieces, offset arithmetic on schedule slots): default "mlp" against
# "1t" (every decoder GEMM on the own kernels, dgrad against transposed copies) and "attn" (mlp + attention projections).  Interleaved, 16 steps.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r06d_modes_ab.txt
: > $OUT
for r in 1 2 3; do
  for mode in mlp 1t attn 1; do
    OTTER_OWN_DECODER_GEMM=$mode timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$mode', $r, d['value'], d['ms_per_step'], r['avg_us'], r['gated_block']['ms'])" >> $OUT
  done
done
cat $OUT
