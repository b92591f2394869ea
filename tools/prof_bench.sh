#!/bin/bash
# rocprofv3 kernel trace of bench.py (run on the GPU box): per-kernel stats -> gpurun_out/<name>_kernel_stats.txt
# usage: tools/prof_bench.sh <name> [bench args...]
NAME=${1:-prof}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/prof_$NAME.log 2>&1
grep '^{"metric' /tmp/prof_$NAME.log | tail -1 > $ROOT/gpurun_out/${NAME}_bench.json   # (the last line of the log is rocprofv3's own)
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
python - "$f" "$ROOT/gpurun_out/${NAME}_kernel_stats.txt" "$ROOT/gpurun_out/${NAME}_bench.json" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w")
out.write(open(sys.argv[3]).read().strip()[:400] + "\n")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out.write("# total kernel time %.3f ms over %d dispatches\n" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
out.write("%-110s %7s %12s %10s %10s %10s %6s\n" % ("name", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:70]:
    out.write("%-110s %7d %12.1f %10.1f %10.1f %10.1f %6.2f\n" % (r["Name"][:110], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
              float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
out.close()
print(open(sys.argv[2]).read()[:6000])
PY
