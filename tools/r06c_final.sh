#!/bin/bash
# Round 6c: validation of HEAD on one box -- full GPU suite, the bench line, the rocprofv3 summary of the same command, flash PMC traffic.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06c_smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --steps 20 --warmup 5 2>gpurun_out/r06c_bench.err | grep '^{"metric' > gpurun_out/r06c_bench.json; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r06c_bench.json').read()); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], r.get('avg_us'), r['frac'], r['gated_block']['ms'], r['gated_block']['frac'], d['cpu_baseline'].get('value'))
PY
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06c_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06c_gpu_tests.log
bash tools/prof_bench.sh r06c > /dev/null 2>&1; head -12 gpurun_out/r06c_kernel_stats.txt
bash tools/pmc_flash_traffic.sh gpurun_out/r06c_pmc_flash_traffic > gpurun_out/r06c_pmc_flash.log 2>&1; tail -30 gpurun_out/r06c_pmc_flash.log
