#!/bin/bash
# Round 6c: which of the frozen decoder's plain products the library wins in situ -- default "mlp" against "mlp" + the attention projections on the own kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "diverse or beam_search" 2>&1 | tail -3
for rep in 1 2; do
  for mode in mlp attn attn_qkv attn_out; do
    OTTER_OWN_DECODER_GEMM=$mode python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06c_own_${mode}_$rep.json
  done
done
for f in gpurun_out/r06c_own_*; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['gated_block']['ms'])"; done
