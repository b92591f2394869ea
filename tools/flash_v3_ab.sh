#!/bin/bash
# forward version 3 (flash variant 6: 16 queries per wave, 16x16x32 MFMA, four waves per SIMD) vs the default forward, one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2 3; do
  echo "default : $(python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  echo "variant6: $(FLASH_VARIANT=6 python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
done
echo "S=2048 default : $(python tools/flash_bench.py 2 2048 1 0 2>/dev/null)"
echo "S=2048 variant6: $(FLASH_VARIANT=6 python tools/flash_bench.py 2 2048 1 0 2>/dev/null)"
echo "padded default : $(python tools/flash_bench.py 8 512 1 0 1 2>/dev/null)"
echo "padded variant6: $(FLASH_VARIANT=6 python tools/flash_bench.py 8 512 1 0 1 2>/dev/null)"
