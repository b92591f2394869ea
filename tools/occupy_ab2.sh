#!/bin/bash
# Second part of tools/occupy_ab.sh: the frozen decoder's GEMMs on the own per-tile kernels (OTTER_OWN_DECODER_GEMM=1) against hipBLASLt while
# n CUs are held by another kernel.   usage: tools/occupy_ab2.sh [out]
OUT=${1:-gpurun_out/occupy_ab2.txt}
run() { echo -n "$1 : " >> $OUT; env $2 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'pairs/s',d['ms_per_step'],'ms/step, FFN GEMM',d['roofline']['avg_us'],'us')" >> $OUT; }
: > $OUT
run "nothing occupied, library decoder GEMMs, persistent own grids (the N=1 default)" "A=1"
run "nothing occupied, own decoder GEMMs, per-tile grids                            " "OTTER_BENCH_OCCUPY_CUS=0 OTTER_OWN_DECODER_GEMM=1 OTTER_BENCH_FORCE_NONPERSISTENT=1"
for n in 16 32; do
run "$n CUs occupied, library decoder GEMMs, per-tile own grids                     " "OTTER_BENCH_OCCUPY_CUS=$n OTTER_BENCH_NONPERSISTENT=1"
run "$n CUs occupied, own decoder GEMMs, per-tile grids                             " "OTTER_BENCH_OCCUPY_CUS=$n OTTER_BENCH_NONPERSISTENT=1 OTTER_OWN_DECODER_GEMM=1"
done
cat $OUT
