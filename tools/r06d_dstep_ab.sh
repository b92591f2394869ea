#!/bin/bash
# Round 6d: cache-policy bits on the LDS-DMA operand loads of the cross-tile kernels (nt / sc1 / sc0 sc1) against none, cold and warm operands
# (tools/gemm_cold_ab.py), interleaved.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r06d_dma_step_ab.txt
: > $OUT
for r in 1 2 3; do
  for v in none dstep5 dstep5e; do
    if [ $v = none ]; then unset OTTER_LIB_PATH; else export OTTER_LIB_PATH=$PWD/otter_amd/lib/libotter_hip_$v.so; fi
    echo "== round $r $v" >> $OUT
    timeout 200 python tools/gemm_cold_ab.py 2>/dev/null | grep "own" >> $OUT
  done
done
cat $OUT
