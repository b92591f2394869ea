#!/bin/bash
# What a resident communication kernel costs the step, and what reserving its CUs buys (run on the GPU box).
# Legs: baseline | 16 CUs pinned by otter_debug_occupy_cus | + one-workgroup-per-tile grids for the own GEMMs | + hipBLASLt stream-K grids
# capped at 240 CUs | the cap alone.   usage: tools/occupy_ab.sh [n_cus] [out]
N=${1:-16}; OUT=${2:-gpurun_out/occupy_ab.txt}; CAP=$((256 - N))
run() { echo -n "$1 : " >> $OUT; env $2 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'pairs/s',d['ms_per_step'],'ms/step, FFN GEMM',d['roofline']['avg_us'],'us')" >> $OUT; }
: > $OUT
run "baseline                                            " "A=1"
run "$N CUs occupied                                     " "OTTER_BENCH_OCCUPY_CUS=$N"
run "$N CUs occupied, own GEMMs one workgroup per tile   " "OTTER_BENCH_OCCUPY_CUS=$N OTTER_BENCH_NONPERSISTENT=1"
run "$N occupied, per-tile grids, stream-K capped at $CAP" "OTTER_BENCH_OCCUPY_CUS=$N OTTER_BENCH_NONPERSISTENT=1 TENSILE_STREAMK_MAX_CUS=$CAP"
run "stream-K capped at $CAP, nothing occupied           " "TENSILE_STREAMK_MAX_CUS=$CAP"
cat $OUT
