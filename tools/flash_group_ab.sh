#!/bin/bash
# heads per longest-first group of the flash launches (OTTER_FLASH_LPT_GROUP): does running a head's four query blocks close together (its
# K / V still in the XCD's L2) cut the re-read traffic and the time?  C2 shape, one box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2; do for g in 256 64 32 16 8; do
  echo "group $g: $(OTTER_FLASH_LPT_GROUP=$g python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
done; done
