#!/bin/bash
# Round-5 tail restructure of the large-grid GEMM (HASIN template parameter + ping-pong input buffers two stripes ahead) against the round-4 tail
# on one box: correctness of the new build, sustained / cold-operand launch time and tile phases, the gated block, every GEMM of the block.
# The old build: check out the parent of the commit that introduced the change, `python -m otter_amd.build --define OTTER_OLDTAIL=1 oldtail`,
# come back and rebuild.   usage (GPU box): bash tools/gemm_tail_r5_ab.sh   ->  profiles/r05_gemm_tail_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
echo "== correctness (new default lib)"; timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_modules.py -m gpu -q -p no:cacheprovider -k "gated or grad_sink or frozen_decoder" 2>&1 | tail -2
for rep in 1 2; do
  for L in new oldtail; do
    if [ $L = new ]; then unset OTTER_LIB_PATH; else export OTTER_LIB_PATH=$ROOT/otter_amd/lib/libotter_hip_$L.so; fi
    for a in "store 0" "store_f32 ab"; do
      echo "[$L rep $rep] $a: $(python tools/gemm_timeline_sustained.py $a 2>&1 | grep -A1 'sustained' | tr '\n' ' ' | sed 's/the stamped launch itself.*block   0 wave 0, 4 tiles://' | cut -c1-250)"
    done
    echo "[$L rep $rep] block: $(python tools/block_profile.py 40 2>&1 | tail -1)"
    echo "[$L rep $rep] block gemms: $(python tools/block_gemm_times.py 2>&1 | grep -E 'gate_bwd|store|gelu|res' | awk '{printf "%s/%s/%s:%s ", $2,$3,$4,$(NF-3)}' | cut -c1-900)"
  done
done
