#!/bin/bash
# PMC passes over the decoder-host flash attention kernels at the C2 shape (run on the GPU box; counters in their own runs,
# --kernel-trace only).  Per kernel: mean of each counter and of the duration.   usage: tools/pmc_flash.sh <outdir> [hd64]
# hd64: the head-pair kernels at the C5 shape (tools/flash64_bench.py: B=4, 64 heads x 64, 1396 tokens) instead
OUT=${1:-gpurun_out/pmc_flash}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CMD="$ROOT/tools/flash_bench.py 8 512 1 0"; [ "$2" = "hd64" ] && CMD="$ROOT/tools/flash64_bench.py"
mkdir -p $ROOT/$OUT; cd /tmp; export TMPDIR=/tmp
C[1]="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"
C[2]="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"
for i in 1 2; do
  rm -rf /tmp/pf_$i
  timeout -k 5 60 rocprofv3 --kernel-trace --pmc ${C[$i]} --output-format csv -d /tmp/pf_$i -o p -- python $CMD > /tmp/pf_$i.log 2>&1
  f=$(find /tmp/pf_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pf_$i.log; continue; fi
  python - "$f" "$ROOT/$OUT/pass$i.json" <<'PY'
import csv, sys, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "flash_" not in k:
        continue
    import re
    k = re.search(r"flash_\w+(<[^>]*>)?", k).group(0)
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, _launches=len(dur[k]), _mean_us=sum(dur[k].values()) / len(dur[k])) for k, cs in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
done
