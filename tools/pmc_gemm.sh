#!/bin/bash
# PMC passes over the FFN-shape GEMM (run on the GPU box; counters in their own runs, --kernel-trace only).
# usage: [KMAJOR=b|ab] tools/pmc_gemm.sh <variant|-1 = torch/hipBLASLt> <outdir> [pass numbers...]
V=${1:-0}; OUT=${2:-gpurun_out/pmc}; shift 2; PASSES=${@:-1 2 3 4}; KM=${KMAJOR:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT; cd /tmp; export TMPDIR=/tmp
C[1]="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
C[2]="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"
C[3]="TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
C[4]="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"
for i in $PASSES; do
  rm -rf /tmp/pmc_$i
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc ${C[$i]} --output-format csv -d /tmp/pmc_$i -o p -- python $ROOT/tools/gemm_one.py $V 3 4096 16384 4096 $KM > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pmc_$i.log; continue; fi
  python - "$f" "$ROOT/$OUT/pass$i.json" <<'PY'
import csv, sys, json, collections
acc = collections.defaultdict(list); dur = {}; name = ''
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    d_us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "gemm_bf16" in k or ("at::native" not in k and d_us > 100):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = k[:80]
name = name if acc else "none"
out = {k: sum(v) / len(v) for k, v in acc.items()}
out["_kernel"] = name; out["_launches"] = len(dur); out["_mean_us"] = sum(dur.values()) / max(len(dur), 1)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
done
