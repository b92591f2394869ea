#!/bin/bash
# rocprofv3 kernel stats of tools/flash_bench.py at C2 (B=8, S=512), per flash variant: tools/prof_flash.sh <variant> [B] [S]
V=${1:-0}; B=${2:-8}; S=${3:-512}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_flash_$V
FLASH_VARIANT=$V timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_flash_$V -o p -- python $ROOT/tools/flash_bench.py $B $S 1 0 > /tmp/prof_flash_$V.log 2>&1
tail -1 /tmp/prof_flash_$V.log
f=$(find /tmp/prof_flash_$V -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:6]:
    print("%-90s calls %5d avg_us %8.1f min_us %8.1f" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
