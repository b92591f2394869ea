#!/bin/bash
# Round 6c: derivative stash of the frozen MLP (OTTER_MLP_STASH_DGELU=1) -- parity tests, then the interleaved A/B on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q -m gpu -x -k "stash or frozen_decoder_block or gemm_epilogues or full_tile_fast_tail" 2>&1 | grep -v "^Librccl\|^RCCL\|^HIP ver\|^ROCm\|^Hostname" | tail -30
for rep in 1 2 3 4; do
  for st in 0 1; do
    OTTER_MLP_STASH_DGELU=$st python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06c6_stash${st}_$rep.json
  done
done
for f in gpurun_out/r06c6_stash*; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['gated_block']['ms'], d.get('loss'))"; done
