#!/bin/bash
# HBM-side traffic of the FFN-shape GEMM (default variant): FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (kernel-trace only), per the MI355X guide; FETCH_SIZE is doubled afterwards (gfx950 counts 128-B requests at 64 B).
# usage: [KMAJOR=b|ab] tools/pmc_traffic.sh <outdir>
OUT=${1:-gpurun_out/pmc_traffic}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; KM=${KMAJOR:-0}
mkdir -p $ROOT/$OUT; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$C
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pt_$C -o p -- python $ROOT/tools/gemm_one.py 0 4 4096 16384 4096 $KM > /tmp/pt_$C.log 2>&1
  f=$(find /tmp/pt_$C -name "*counter_collection.csv" | head -1)
  python - "$f" "$ROOT/$OUT/$C.json" "$C" <<'PY'
import csv, sys, json
vals, dur, name = [], [], ""
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_bf16" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]:
        vals.append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3); name = r["Kernel_Name"][:90]
json.dump({"kernel": name, "launches": len(vals), "mean_KB": sum(vals) / max(len(vals), 1), "mean_us": sum(dur) / max(len(dur), 1)}, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
done
