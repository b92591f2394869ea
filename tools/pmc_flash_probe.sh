#!/bin/bash
# which flash variant survives the 8-counter --pmc passes of tools/pmc_flash.sh? (diagnostic)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"
for v in 0 7; do
  rm -rf /tmp/pp_$v
  FLASH_VARIANT=$v timeout -k 5 45 rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/pp_$v -o p -- python $ROOT/tools/flash_bench.py 8 512 1 0 > /tmp/pp_$v.log 2>&1
  echo "variant $v rc=$? : $(grep -c flash_ $(find /tmp/pp_$v -name '*counter_collection.csv' | head -1) 2>/dev/null) counter rows; log: $(grep -v "^W2026" /tmp/pp_$v.log | tail -3 | cut -c1-300)"
done
