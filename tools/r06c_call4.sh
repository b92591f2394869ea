#!/bin/bash
# Does the flash forward's time follow its fabric traffic?  PMC passes with the LPT head-group size forced (smaller groups = more K/V reuse in the L2s).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -x -k "diverse" 2>&1 | tail -2
for g in 0 8 16 32 64; do
  OTTER_FLASH_LPT_GROUP=$g bash tools/pmc_flash_traffic.sh gpurun_out/r06c_pmc_flash_g$g > /dev/null 2>&1
  python - $g <<'PY'
import json,sys
g=sys.argv[1]
try:
    F=json.load(open('gpurun_out/r06c_pmc_flash_g%s/FETCH_SIZE.json'%g)); W=json.load(open('gpurun_out/r06c_pmc_flash_g%s/WRITE_SIZE.json'%g))
    for k in F:
        r=F[k]["mean_KB"]*2*1024/1e6; w=W[k]["mean_KB"]*1024/1e6; us=(F[k]["mean_us"]+W[k]["mean_us"])/2
        print("group=%s %-45s read %.1f MB write %.1f MB  %.1f us  %.2f TB/s" % (g, k[:45], r, w, us, (r+w)/us))
except Exception as e: print(g, 'failed', e)
PY
done
