#!/bin/bash
# Round 6d: does PyTorch's TunableOp find faster hipBLASLt / rocBLAS solutions for the library GEMMs of the step (QKV / out_proj run a stream-K kernel
# at 0.57 of nominal, the FFN-shape plain products a 0.64 kernel)?  Leg T: tune while running 4 steps (CSV -> gpurun_out/), then A/B interleaved:
# default heuristics against the tuned CSV (tuning off).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r06d_tunableop_ab.txt
CSV=$PWD/gpurun_out/r06d_tunableop_results.csv
: > $OUT
echo "== tuning run" >> $OUT
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$CSV PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=512 \
  timeout 1500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>gpurun_out/r06d_tune.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tune-run', d['value'], d['ms_per_step'])" >> $OUT
ls -la gpurun_out/r06d_tunableop_results*.csv >> $OUT 2>&1
for r in 1 2 3; do
  for leg in default tuned; do
    if [ $leg = tuned ]; then export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$CSV; else unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_TUNING PYTORCH_TUNABLEOP_FILENAME; fi
    timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$leg', $r, d['value'], d['ms_per_step'], d['roofline']['avg_us'])" >> $OUT
  done
done
cat $OUT; head -40 gpurun_out/r06d_tunableop_results*.csv
