for rep in 1 2 3; do
  for mode in mlp 1t 1; do
    OTTER_OWN_DECODER_GEMM=$mode python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06b_own_${mode}_$rep.json
  done
done
for f in gpurun_out/r06b_own_*; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['gated_block']['ms'], d.get('peak_mem_gb'))"; done
