#!/bin/bash
# three-way: the runtime's own default (OTTER_NO_RUNTIME_DEFAULTS=1, variable unset) / explicit 0 / the package default (1), interleaved on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
unset HIP_FORCE_DEV_KERNARG
for rep in 1 2 3 4; do
  OTTER_NO_RUNTIME_DEFAULTS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06c9_unset_$rep.json
  HIP_FORCE_DEV_KERNARG=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06c9_zero_$rep.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06c9_default_$rep.json
done
for f in gpurun_out/r06c9_*; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['gated_block']['ms'], d['config'].get('dev_kernarg'), d.get('loss'))"; done
python -m pytest tests/test_gpu_modules.py -q -m gpu -k "bench_line or two_ranks" 2>&1 | tail -2
