#!/bin/bash
# Round 6d: M0 written once per four LDS-DMA pieces (cross-tile K-contiguous instantiations of variant 26, -DOTTER_T4_M0GROUP=1, the default) against
# M0 in the first piece's slot (libotter_hip_m0late.so): correctness (GEMM tests + the cross-tile race screen), then interleaved timing on cold operands.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OLD=$PWD/otter_amd/lib/libotter_hip_m0late.so
OUT=gpurun_out/r06d_m0early_ab.txt
: > $OUT
echo "== GEMM tests, grouped build" >> $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" -x 2>&1 | tail -2 >> $OUT
echo "== race screen (cross-tile form bit for bit against the plain form)" >> $OUT
timeout 900 python tools/gemm_xt_stress.py 2 2>&1 | tail -4 >> $OUT
for r in 1 2 3; do
  echo "== round $r M0 a slot early (default build)" >> $OUT
  timeout 300 python tools/gemm_xt_ab.py 3 3 2>&1 | grep -v "^$" | tail -8 >> $OUT
  echo "== round $r M0 in the first piece's slot" >> $OUT
  OTTER_LIB_PATH=$OLD timeout 300 python tools/gemm_xt_ab.py 3 3 2>&1 | grep -v "^$" | tail -8 >> $OUT
done
cat $OUT
