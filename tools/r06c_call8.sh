#!/bin/bash
# Round 6c: runtime knobs that touch launch overhead (1 410 launches per step, GPU idle ~1.5 % of the step): kernel arguments in device memory
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do
  for k in 0 1; do
    HIP_FORCE_DEV_KERNARG=$k python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06c8_kernarg${k}_$rep.json
  done
done
for f in gpurun_out/r06c8_kernarg*; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['gated_block']['ms'])"; done
