#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -x -k "diverse" 2>&1 | grep -v "^Librccl\|^RCCL\|^HIP ver\|^ROCm\|^Hostname" | tail -40
for rep in 1 2 3 4 5; do
  for mode in mlp attn attn_qkv; do
    OTTER_OWN_DECODER_GEMM=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06c3_own_${mode}_$rep.json
  done
done
for f in gpurun_out/r06c3_own_*; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['gated_block']['ms'])"; done
