#!/bin/bash
# Stage (or remove) an UNCOMMITTED scratch copy of the five files of the reference's training script under oracle/_ref/reference_loop/ so that
# ONE gpurun call can run tests/test_gpu_dropin.py (the reference's own train_one_epoch on the HIP kernels).  oracle/_ref/ is git-ignored and
# travels with the gpurun snapshot; nothing under it is ever committed.  Usage: tools/stage_reference_loop.sh stage|clean
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
dst="$root/oracle/_ref/reference_loop"
case "$1" in
  stage)
    src=/root/reference/pipeline/train
    [ -d "$src" ] || { echo "no $src (build container only)"; exit 1; }
    mkdir -p "$dst/pipeline/train"
    for f in __init__.py instruction_following.py train_args.py train_utils.py distributed.py; do cp "$src/$f" "$dst/pipeline/train/$f"; done
    echo "staged -> $dst (remove with: $0 clean)";;
  stage-models)
    # the reference's own model modules, for oracle/calibrate_cpu_baseline.py ON THE GPU NODE (cpu_baseline calibrated on its host cores)
    src=/root/reference/src/otter_ai/models
    [ -d "$src" ] || { echo "no $src (build container only)"; exit 1; }
    for d in otter mpt mpt_redpajama falcon; do
      mkdir -p "$dst/src/otter_ai/models/$d"
      cp "$src/$d"/*.py "$dst/src/otter_ai/models/$d/"
    done
    echo "staged models -> $dst (OTTER_REF_ROOT=$dst; remove with: $0 clean)";;
  clean) rm -rf "$dst"; echo "removed $dst";;
  *) echo "usage: $0 stage|stage-models|clean"; exit 2;;
esac
