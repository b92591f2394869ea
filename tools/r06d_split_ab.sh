#!/bin/bash
# Round 6d: split-wait K-tile schedule (-DOTTER_T4_SPLIT: barrier #2 waits for the first eight DMA pieces of K-tile t+1 only, a third barrier
# publishes the rest) against the default build: correctness of the split build on the GEMM tests, then interleaved timing legs on cold operands.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
SPLIT=$PWD/otter_amd/lib/libotter_hip_split.so
OUT=gpurun_out/r06d_split_ab.txt
: > $OUT
echo "== correctness of the split build (GEMM tests through OTTER_LIB_PATH)" >> $OUT
OTTER_LIB_PATH=$SPLIT timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" -x 2>&1 | tail -5 >> $OUT
for r in 1 2 3; do
  echo "== round $r default" >> $OUT
  timeout 300 python tools/gemm_xt_ab.py 3 3 2>&1 | grep -v "^$" | tail -16 >> $OUT
  echo "== round $r split" >> $OUT
  OTTER_LIB_PATH=$SPLIT timeout 300 python tools/gemm_xt_ab.py 3 3 2>&1 | grep -v "^$" | tail -16 >> $OUT
done
cat $OUT
