#!/bin/bash
# step-level A/B of the frozen decoder's GEMM paths on one box: library (default) / fused MLP legs on own kernels / every GEMM on own kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2; do
  for mode in "" mlp 1; do
    OTTER_OWN_DECODER_GEMM=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTTER_OWN_DECODER_GEMM=%-4s' % '$mode', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')"
  done
done
