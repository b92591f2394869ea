#!/bin/bash
# round 4 flash A/B on one box: default (lean LDS-DMA, delta inside dQ, bit-mask diagonal tiles) vs variant 4 (separate flash_delta launch)
# vs the -DOTTER_FLASH_SAFE_DMA build (save / restore M0 around every piece); C2 shape and the C5 head-pair shape.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
S=$ROOT/otter_amd/lib/libotter_hip_safedma.so
for rep in 1 2 3; do
  echo "default        : $(python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  echo "variant 4      : $(FLASH_VARIANT=4 python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  echo "safe-dma build : $(OTTER_LIB_PATH=$S python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  echo "safe-dma + v4  : $(OTTER_LIB_PATH=$S FLASH_VARIANT=4 python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
done
echo "padded default : $(python tools/flash_bench.py 8 512 1 0 1 2>/dev/null)"
echo "padded safe+v4 : $(OTTER_LIB_PATH=$S FLASH_VARIANT=4 python tools/flash_bench.py 8 512 1 0 1 2>/dev/null)"
