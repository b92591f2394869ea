"""Per-kernel time PER TRAINING STEP: bench.py under rocprofv3 --kernel-trace (csv), windowed on the last three whole steps (between
AdamW launches), so the model build, warm-up and the stand-alone block timing of bench.py are left out.  Run on the GPU box."""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/prof_steptable"
extra = " ".join(sys.argv[1:])
subprocess.run("rm -rf %s; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d %s -o p -- python %s/bench.py --steps 4 --warmup 2 --no-cpu-baseline %s > /tmp/prof_steptable.log 2>&1"
               % (out, out, ROOT, extra), shell=True)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
ad = [i for i, e in enumerate(ev) if "adamw_stream_kernel" in e[2]]
# warm-up 2 + 4 timed steps, then bench.py's calibration extras (one more step, the stand-alone block, the MFMA probe):
# the window is the timed steps 1-3, counted from the front
lo, hi = ad[2] + 1, ad[5] + 1
win = ev[lo:hi]
span = (win[-1][1] - win[0][0]) / 3e6
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in win:
    tot[n] += e - s
    cnt[n] += 1
allk = sum(tot.values()) / 3e6
print("# per training step (mean of 3): wall %.2f ms, sum of kernel durations %.2f ms, %d launches" % (span, allk, len(win) // 3))
print("%-118s %8s %10s %8s %6s" % ("kernel", "calls", "ms/step", "avg_us", "pct"))
for n, t in tot.most_common(45):
    print("%-118s %8.1f %10.3f %8.1f %6.2f" % (n[:118], cnt[n] / 3, t / 3e6, t / cnt[n] / 1e3, 100.0 * t / 3e6 / allk))
lib = sum(t for n, t in tot.items() if "Cijk_" in n) / 3e6
own = sum(t for n, t in tot.items() if "gemm_bf16" in n) / 3e6
print("# hipBLASLt (Cijk_*) %.2f ms = %.1f %%; own GEMM kernels %.2f ms = %.1f %%" % (lib, 100 * lib / allk, own, 100 * own / allk))
