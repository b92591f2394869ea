python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sum_of_squares" > gpurun_out/r06b_t3.log 2>&1
for rep in 1 2 3; do
  python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06b_kc0_$rep.json
  OTTER_MLP_DGRAD_KCONTIG=1 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06b_kc1_$rep.json
done
tail -n 3 gpurun_out/r06b_t3.log
for f in gpurun_out/r06b_kc*; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['gated_block']['ms'])"; done
