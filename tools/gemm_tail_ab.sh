#!/bin/bash
# Fused-tail experiments on the GPU box (round 4): (1) per-tile timeline of variant 26 for every epilogue kind with the product library
# (non-temporal output stores) and with the diag-128 build (plain write-back stores); (2) the gated block's anatomy on both libraries.
# usage: tools/gemm_tail_ab.sh <outfile>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${1:-$ROOT/gpurun_out/r04_tail_ab.txt}
cd $ROOT; mkdir -p gpurun_out
python -m otter_amd.build --diag 128 > /dev/null 2>&1
D=$ROOT/otter_amd/lib/libotter_hip_diag128.so
{
for lib in "" "$D"; do
  echo "=== library: ${lib:-product (nt stores)}"
  for epi in store store_f32 gelu gate_bwd res; do
    OTTER_LIB_PATH=$lib python tools/gemm_timeline.py 26 4096 16384 4096 $epi 2>/dev/null | grep -E "epilogue|tile"
  done
  for i in 1 2; do OTTER_LIB_PATH=$lib python tools/block_profile.py 30 2>/dev/null | tail -1; done
done
} > $OUT 2>&1
cat $OUT
