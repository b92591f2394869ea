#!/bin/bash
# round 6b: s_setprio 1 around the MFMA clusters of the flash forward (bit 0) / dQ (bit 1) kernels -- A/B builds
# (python -m otter_amd.build --flash-define OTTER_FLASH_PRIO=n prioN), interleaved processes on one box; C2 shape, us per launch.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2 3; do
  echo "default : $(python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  for n in 1 2; do
    echo "prio $n  : $(OTTER_LIB_PATH=$ROOT/otter_amd/lib/libotter_hip_prio$n.so python tools/flash_bench.py 8 512 1 0 2>/dev/null)"
  done
done
