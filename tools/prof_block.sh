#!/bin/bash
# rocprofv3 kernel trace of tools/block_profile.py (one gated cross-attention block, forward + backward, C2 shapes) -> per-kernel anatomy
# usage: tools/prof_block.sh <name> [iters]
NAME=${1:-block}; IT=${2:-20}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o p -- python $ROOT/tools/block_profile.py $IT > /tmp/prof_$NAME.log 2>&1
tail -1 /tmp/prof_$NAME.log
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
python - "$f" "$ROOT/gpurun_out/${NAME}_kernel_stats.txt" "$IT" "$(tail -1 /tmp/prof_$NAME.log)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = int(sys.argv[3]) + 2
out = open(sys.argv[2], "w")
out.write("# tools/block_profile.py under rocprofv3 --kernel-trace --stats; %s (under the profiler)\n" % sys.argv[4])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out.write("# kernel time per block forward+backward: %.3f ms (%d launches per iteration)\n" % (tot / 1e6 / it, sum(int(r["Calls"]) for r in rows) // it))
out.write("%-100s %9s %12s %10s %6s\n" % ("name", "calls/it", "us per it", "avg_us", "pct"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    out.write("%-100s %9.1f %12.1f %10.1f %6.2f\n" % (r["Name"][:100], int(r["Calls"]) / it, float(r["TotalDurationNs"]) / 1e3 / it, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
out.close()
print(open(sys.argv[2]).read()[:5000])
PY
