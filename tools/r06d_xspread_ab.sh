#!/bin/bash
# Round 6d: the K-major schedule (KTILE_X0) with its DMA pieces spread as far apart as the K-contiguous one's (libotter_hip_xs.so) against the product
# schedule (steps 3 / 4, all sixteen in front of barrier #2): the K-major legs of tools/gemm_xt_ab.py, interleaved.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r06d_xm0early_ab.txt
: > $OUT
OTTER_LIB_PATH=$PWD/otter_amd/lib/libotter_hip_xs.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" -x 2>&1 | tail -1 >> $OUT
for r in 1 2 3; do
  for v in product spread; do
    if [ $v = product ]; then unset OTTER_LIB_PATH; else export OTTER_LIB_PATH=$PWD/otter_amd/lib/libotter_hip_xs.so; fi
    echo "== round $r $v" >> $OUT
    timeout 300 python tools/gemm_xt_ab.py 3 3 2>/dev/null | grep "^dW2\|^dU\|^df" >> $OUT
  done
done
cat $OUT
