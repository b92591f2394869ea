#!/bin/bash
# persistent per-head dK/dV kernel (default at C2: B x H = 256 workgroups) vs one workgroup per key block (flash variant 7), one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2 3; do
  echo "persistent: $(python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  echo "per-block : $(FLASH_VARIANT=7 python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
done
echo "S=2048 B=8 persistent: $(python tools/flash_bench.py 8 2048 1 0 2>/dev/null)"
echo "S=2048 B=8 per-block : $(FLASH_VARIANT=7 python tools/flash_bench.py 8 2048 1 0 2>/dev/null)"
echo "padded persistent: $(python tools/flash_bench.py 8 512 1 0 1 2>/dev/null)"
echo "padded per-block : $(FLASH_VARIANT=7 python tools/flash_bench.py 8 512 1 0 1 2>/dev/null)"
