#!/bin/bash
# Round 6d: where the K loop of variant 26 spends its cycles above the 2048 of back-to-back MFMAs: ablation builds (tools/gemm_abl_time.py), interleaved.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r06d_abl.txt
: > $OUT
for r in 1 2; do
  for m in ${MASKS:-0 1 2 4 3 5 6 7}; do
    if [ $m = 0 ]; then unset OTTER_LIB_PATH; else export OTTER_LIB_PATH=$PWD/otter_amd/lib/libotter_hip_abl$m.so; fi
    timeout 300 python tools/gemm_abl_time.py 5 2>/dev/null | sed "s/^/mask $m  /" >> $OUT
  done
done
cat $OUT
