"""GPU idle time inside the training steps: runs bench.py under rocprofv3 --kernel-trace (csv), takes the kernel intervals of the last steps and
reports busy time (union of intervals), idle gaps, and the largest gaps with the kernels around them.  Run on the GPU box: python tools/gpu_idle.py"""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/prof_idle"
subprocess.run("rm -rf %s; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d %s -o p -- python %s/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/prof_idle.log 2>&1" % (out, out, ROOT), shell=True)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows), key=lambda e: e[0])
# the timed steps: find AdamW launches (one per step) and take the window between the 3rd-last and the last
ad = [i for i, e in enumerate(ev) if "adamw_stream_kernel" in e[2]]
lo, hi = ad[-4] + 1, ad[-1] + 1          # three whole steps
win = ev[lo:hi]
span = win[-1][1] - win[0][0]
busy, cur_end, gaps = 0, win[0][0], []
for s, e, n in win:
    if s > cur_end:
        gaps.append((s - cur_end, n))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
print("3 steps: span %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps; kernels %d" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps), len(win)))
import collections
by = collections.Counter()
for g, n in gaps:
    by[n] += g
print("idle time attributed to the kernel that FOLLOWS the gap (top 15), ms per step:")
for n, g in by.most_common(15):
    print("  %8.3f  %s" % (g / 3e6, n))
hist = collections.Counter(min(int(g / 1000) // 5 * 5, 100) for g, _ in gaps)
print("gap histogram (us bucket: count):", sorted(hist.items()))
