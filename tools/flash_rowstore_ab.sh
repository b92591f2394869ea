#!/bin/bash
# O / dQ / dK / dV through the LDS transpose as whole rows (default) vs the 8-byte-per-row stores (-DOTTER_FLASH_ROWSTORE=0 build), one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OLD=otter_amd/lib/libotter_hip_norowstore.so
for rep in 1 2 3; do
  echo "row stores : $(python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  echo "8-B stores : $(OTTER_LIB_PATH=$OLD python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
done
echo "S=2048 B=2 row stores : $(python tools/flash_bench.py 2 2048 1 0 2>/dev/null)"
echo "S=2048 B=2 8-B stores : $(OTTER_LIB_PATH=$OLD python tools/flash_bench.py 2 2048 1 0 2>/dev/null)"
echo "head_dim 64 (C5 shape) row stores : $(python tools/flash64_bench.py 2>/dev/null | tail -1)"
echo "head_dim 64 (C5 shape) 8-B stores : $(OTTER_LIB_PATH=$OLD python tools/flash64_bench.py 2>/dev/null | tail -1)"
