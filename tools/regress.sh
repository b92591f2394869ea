#!/bin/bash
# tools/regress.sh -- ONE gpurun call that re-measures the figures DESIGN.md quotes, so that a kernel change is checked against the same
# table every time: per-shape GEMM times (own vs hipBLASLt, warm and cold operands), the gated block's
# per-kernel anatomy under rocprofv3, the flash kernels at C2, the LayerNorm stream kernels, and the default bench line with its calibration.
# usage (GPU box, from the repo root):   bash tools/regress.sh [tag]      ->  gpurun_out/<tag>_regress_*.{txt,json}   (copy to profiles/)
# e.g.  gpurun --timeout 900 -- 'bash tools/regress.sh r05'
TAG=${1:-r05}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
echo "== bench (default line, calibration inside) =="
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_regress_bench.json 2> $OUT/${TAG}_regress_bench.err; cut -c1-1500 $OUT/${TAG}_regress_bench.json
echo "== per-shape GEMMs: own vs hipBLASLt, warm / cold operands (tools/gemm_cold_ab.py) =="
timeout 200 python tools/gemm_cold_ab.py > $OUT/${TAG}_regress_gemm_cold_ab.txt 2>&1; cat $OUT/${TAG}_regress_gemm_cold_ab.txt
echo "== every GEMM launch of the gated block, one by one (tools/block_gemm_times.py) =="
timeout 200 python tools/block_gemm_times.py > $OUT/${TAG}_regress_block_gemm_times.txt 2>&1; tail -30 $OUT/${TAG}_regress_block_gemm_times.txt
echo "== gated block anatomy under rocprofv3 (tools/prof_block.sh) =="
timeout 300 bash tools/prof_block.sh ${TAG}_regress_block 20 > /dev/null 2>&1; head -45 $OUT/${TAG}_regress_block_kernel_stats.txt
echo "== flash attention at C2 (tools/flash_bench.py) =="
timeout 200 python tools/flash_bench.py > $OUT/${TAG}_regress_flash.txt 2>&1; tail -12 $OUT/${TAG}_regress_flash.txt
echo "== LayerNorm stream kernels, warm / cold (tools/norm_cold_bench.py) =="
timeout 200 python tools/norm_cold_bench.py > $OUT/${TAG}_regress_norm.txt 2>&1; tail -12 $OUT/${TAG}_regress_norm.txt
echo "== verdict: this call against the last committed regress table (tools/regress_check.py; non-zero exit on a > 3 % regression on a comparable box) =="
python tools/regress_check.py $TAG | tee $OUT/${TAG}_regress_check.txt
exit ${PIPESTATUS[0]}
