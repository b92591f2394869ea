#!/bin/bash
# non-temporal loads in the LayerNorm row kernels (A/B builds -DOTTER_NORM_NT=mask): cold-operand kernel times and the step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for m in 0 1 3 7; do
  L=""; [ $m != 0 ] && L="OTTER_LIB_PATH=otter_amd/lib/libotter_hip_normnt$m.so"
  echo "== mask $m"; env $L python tools/norm_cold_bench.py 2>/dev/null | grep "cold" | grep -v "copy\|add_f32"
done
for rep in 1 2; do for m in 0 7 3; do
  L=""; [ $m != 0 ] && L="OTTER_LIB_PATH=otter_amd/lib/libotter_hip_normnt$m.so"
  env $L python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask $m:', d['value'], d['ms_per_step'])"
done; done
